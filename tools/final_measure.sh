#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/final_measure.sh <tag>
# bench lines of all workloads + rocprofv3 kernel stats of the default command -> gpurun_out/<tag>_*
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err || { echo "c2 failed"; tail -5 gpurun_out/${tag}_c2.err; exit 1; }
echo "c2 done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_c2 -o c2 --output-format csv -- python3 bench.py --no-cpu --no-alone > gpurun_out/${tag}_prof_c2.log 2>&1 || { echo "rocprof c2 failed"; exit 1; }
echo "rocprof c2 done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_c3 -o c3 --output-format csv -- python3 bench.py --no-cpu --no-alone --workload c3 --spp 4 > gpurun_out/${tag}_prof_c3.log 2>&1 || { echo "rocprof c3 failed"; exit 1; }
echo "rocprof c3 done"
for w in c3 c3t c4; do
  timeout -k 10 300 python3 bench.py --no-cpu --workload $w > gpurun_out/${tag}_$w.json 2> gpurun_out/${tag}_$w.err || { echo "$w failed"; tail -5 gpurun_out/${tag}_$w.err; exit 1; }
  echo "$w done"
done
timeout -k 10 300 python3 bench.py --no-cpu --workload c5 --width 3840 --height 2160 > gpurun_out/${tag}_c5_4k.json 2> gpurun_out/${tag}_c5.err || { echo "c5 failed"; exit 1; }
echo "c5 4k done"
timeout -k 10 300 python3 bench.py --no-cpu --workload c5 > gpurun_out/${tag}_c5_1080p.json 2>> gpurun_out/${tag}_c5.err || { echo "c5 failed"; exit 1; }
timeout -k 10 200 python3 tools/build_bench.py > gpurun_out/${tag}_build.txt 2>&1 || { echo "build bench failed"; exit 1; }
echo "all done"
