#!/bin/bash
# usage: tools/install_profiles.sh <tag> <version> [round prefix, default r01]  -- copies gpurun_out/<tag>_* (tools/final_measure.sh, tools/pmc_passes.sh <tag>_c2) into profiles/
set -e
T=$1; V=$2; R=${3:-r01}
cp gpurun_out/${T}_c2.json profiles/${R}_c2_bench_${V}_final.json
cp gpurun_out/${T}_c3.json profiles/${R}_c3_bench_${V}_final.json
cp gpurun_out/${T}_c3t.json profiles/${R}_c3t_bench_${V}_final.json
cp gpurun_out/${T}_c4.json profiles/${R}_c4_bench_${V}_final.json
cp gpurun_out/${T}_c5_4k.json profiles/${R}_c5_4k_bench_${V}_final.json
cp gpurun_out/${T}_c5_1080p.json profiles/${R}_c5_1080p_bench_${V}_final.json
cp gpurun_out/${T}_prof_c2/c2_kernel_stats.csv profiles/${R}_c2_bunny_kernel_stats_${V}_final.csv
cp gpurun_out/${T}_prof_c3/c3_kernel_stats.csv profiles/${R}_c3_cornell_4spp_kernel_stats_${V}_final.csv
cp gpurun_out/${T}_build.txt profiles/${R}_bvh_build_throughput.txt
cp gpurun_out/${T}_counts.txt profiles/${R}_traversal_counters_${V}.txt
python3 tools/pmc_summary.py c2 profiles/pmc_traffic.json gpurun_out/pmc_${T}_c2_*/p_counter_collection.csv > /dev/null
