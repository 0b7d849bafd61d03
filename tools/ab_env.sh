#!/bin/bash
# usage: tools/ab_env.sh <out-file> "<ENV=val ...>" ... -- <bench args>   one bench.py run per environment setting ("-" = none), each twice
out=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
: > "$out"
for rep in 1 2; do
for e in "${envs[@]}"; do
  ee=$e; [ "$e" = "-" ] && ee="RT_NOOP=1"
  line=$(env $ee timeout -k 10 300 python bench.py --no-cpu --no-alone "$@" 2>/dev/null | tail -1) || { echo "$e: failed" >> "$out"; exit 1; }
  echo "$e $* :: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("Mrays/s %.0f ms/step %.4f primary %s secondary %s launch_ms %.4f" % (d["value"], d["ms_per_step"], d["primary_Mrays_s"] and round(d["primary_Mrays_s"]), d["secondary_Mrays_s"] and round(d["secondary_Mrays_s"]), d["roofline"]["mean_launch_ms"]))')" >> "$out"
done
done
cat "$out"
