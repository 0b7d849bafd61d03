#!/bin/bash
# usage (on a node with N GPUs, from the repo root): bash tools/scale_matrix.sh [max_gpus, default 8] [out-dir, default gpurun_out]
# The 1 / 2 / 4 / 8-GPU lines of the three workloads that BASELINE.md quotes a scaling figure for, exactly as the driver
# launches bench.py (torch.distributed.run, one rank per GPU, RCCL):
#   c2      sample-set split (weak scaling: every rank renders its own frames of the chain; one reduce(sum) at the end)
#   c4      tile split       (strong scaling: every rank renders its 32x32 tiles of every frame; gather of owned tiles)
#   c5 4K   sample-set split (weak; the 132.7 MB accumulator reduce is the exchange BASELINE.md's C5 row names)
# One JSON line per run in <out-dir>/scale_<workload>_<n>.json.  `--rehearse` as a third argument runs every rank on GPU 0
# with gloo (a one-GPU box: checks the commands, the numbers mean nothing).
set -u
MAXN=${1:-8}; OUT=${2:-gpurun_out}; REH=${3:-}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29700
run() { # name n args...
  local name=$1 n=$2; shift 2
  port=$((port + 1))
  if [ "$n" = 1 ]; then
    timeout -k 10 600 python3 bench.py --gpus 1 --no-cpu --no-alone "$@" > "$OUT/scale_${name}_1.json" 2> "$OUT/scale_${name}_1.err"
  else
    timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus "$n" --no-cpu --no-alone $REH "$@" > "$OUT/scale_${name}_${n}.json" 2> "$OUT/scale_${name}_${n}.err"
  fi || { echo "$name x$n failed"; tail -3 "$OUT/scale_${name}_${n}.err"; return 1; }
  python3 - "$OUT/scale_${name}_${n}.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-28s n=%d  %9.0f Mrays/s  %8.3f ms/step  exchange %s ms (%s)" % (d["config"]["workload"][:28], d["n_gpus"], d["value"], d["ms_per_step"],
      "-" if d["reduce_ms"] is None else "%.3f" % d["reduce_ms"], (d.get("exchange") or {}).get("kind", "none")))
PY
}
for n in 1 2 4 8; do
  [ "$n" -le "$MAXN" ] || break
  run c2 $n --steps 200 --warmup 10 || exit 1
  run c4 $n --workload c4 --split tile --steps 20 --warmup 2 || exit 1
  run c5_4k $n --workload c5 --width 3840 --height 2160 --split sample --steps 32 --warmup 2 || exit 1
done
