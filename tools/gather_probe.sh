#!/bin/bash
# usage: tools/gather_probe.sh "<ENV=val ...>" ... : one C4 bench line per environment setting, the gather figures of each
for e in "$@"; do
  ee=$e; [ "$e" = "-" ] && ee="RT_NOOP=1"
  line=$(env $ee timeout -k 10 300 python bench.py --no-cpu --no-alone --workload c4 --steps 20 2>/dev/null | tail -1) || { echo "$e: failed"; exit 1; }
  echo "$e :: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); g=d["gather"]; print("Mrays/s %.0f ms/step %.3f | gather launch %.3f ms, %.0f G pairs/s, share %.3f | traverse launch %.4f ms" % (d["value"], d["ms_per_step"], g["mean_launch_ms"], g["pairs_per_s"]/1e9, g["share_of_step"], d["roofline"]["mean_launch_ms"]))')"
done
