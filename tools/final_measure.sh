#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/final_measure.sh <stage> <tag>
#   pmc    counter passes of every BASELINE configuration -> gpurun_out/pmc_<key>.json (merge: tools/pmc_merge.py, then commit)
#   bench  the bench lines (with the CPU leg and the roofline from the merged profiles/pmc_traffic.json) -> gpurun_out/<tag>_<config>.json
#   prof   rocprofv3 --kernel-trace --stats of each configuration's bench command -> gpurun_out/<tag>_<config>_kernel_stats.csv,
#          traversal counters, BVH build throughput
set -u
stage=$1; tag=${2:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
case $stage in
pmc)
  bash tools/pmc_passes.sh c2 --steps 40 --warmup 5 && bash tools/pmc_passes.sh c3 --workload c3 --steps 3 --warmup 1 &&
  bash tools/pmc_passes.sh c4 --workload c4 --steps 3 --warmup 1 && bash tools/pmc_passes.sh c5 --workload c5 --steps 16 --warmup 2 &&
  bash tools/pmc_passes.sh c5_3840x2160 --workload c5 --width 3840 --height 2160 --steps 16 --warmup 2 &&
  bash tools/pmc_passes.sh c3t --workload c3t --steps 40 --warmup 5 || exit 1 ;;
bench)
  run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err || { echo "$name failed"; tail -5 gpurun_out/${tag}_$name.err; exit 1; }; echo "$name done"; }
  run c2 ; run c2_driver --steps 20 --warmup 5 --no-cpu ; run c3 --workload c3 --steps 10 ; run c4 --workload c4 --steps 20
  run c5_4k --workload c5 --width 3840 --height 2160 --steps 60 ; run c5_1080p --workload c5 --steps 100 ; run c3t --workload c3t --no-cpu ;;
prof)
  prof() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_$name -o $name --output-format csv -- python3 bench.py --no-cpu --no-alone "$@" > gpurun_out/${tag}_prof_$name.log 2>&1 || { echo "rocprof $name failed"; tail -5 gpurun_out/${tag}_prof_$name.log; exit 1; }
    cp gpurun_out/${tag}_prof_$name/*/${name}_kernel_stats.csv gpurun_out/${tag}_${name}_kernel_stats.csv 2>/dev/null || cp gpurun_out/${tag}_prof_$name/${name}_kernel_stats.csv gpurun_out/${tag}_${name}_kernel_stats.csv || exit 1
    rm -rf gpurun_out/${tag}_prof_$name; echo "rocprof $name done"; }
  prof c2 ; prof c3 --workload c3 --steps 4 ; prof c4 --workload c4 --steps 6 ; prof c5_4k --workload c5 --width 3840 --height 2160 --steps 24
  timeout -k 10 200 python3 tools/count_traversal.py c2 c3 c4 c5 c5_4k > gpurun_out/${tag}_traversal_counters.txt 2>&1 || { echo "counters failed"; exit 1; }
  timeout -k 10 200 python3 tools/build_bench.py > gpurun_out/${tag}_bvh_build_throughput.txt 2>&1 || { echo "build bench failed"; exit 1; } ;;
*) echo "unknown stage $stage"; exit 2 ;;
esac
echo "stage $stage done"
