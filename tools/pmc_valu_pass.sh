#!/bin/bash
# usage: tools/pmc_valu_pass.sh <tag> <bench.py args...>   -- the SQ instruction pass alone (per-kernel VALU instructions and busy cycles)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/pmcv_${tag} -o p --output-format csv -- python3 bench.py --no-cpu --no-alone "$@" > gpurun_out/pmcv_${tag}.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmcv_${tag}.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, sys, collections, re
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open("gpurun_out/pmcv_%s/p_counter_collection.csv" % tag)):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"]); k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:12]:
    print("%-48s launches %6d  VALU insts total %.3e (%.1f %%)  per launch %.3e  cycles/inst %.2f" % (k, len(n[k]), v["SQ_INSTS_VALU"], 100 * v["SQ_INSTS_VALU"] / tot, v["SQ_INSTS_VALU"] / len(n[k]), 4 * v["SQ_ACTIVE_INST_VALU"] / max(1.0, v["SQ_INSTS_VALU"])))
PY
