#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <bench.py args...>   -- one rocprofv3 --pmc run per counter group; the per-kernel, per-launch
# means of all passes end up in gpurun_out/pmc_<tag>.json under the key <tag> (tools/pmc_summary.py; bench.py's pmc_key: the
# workload, with _WxH unless 1080p), stamped with the hash of the kernel sources; merge into profiles/pmc_traffic.json with
# tools/pmc_merge.py.  The csv files (tens of MB) stay on the box.
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
groups=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum TCC_CYCLE_sum"
)
python3 -c "import sys; sys.path.insert(0, 'tools'); import pmc_summary; print(pmc_summary.source_hash('.'))" > gpurun_out/pmc_${tag}.hash || exit 1  # the kernel sources these passes belong to (PMC_SOURCE_HASH for tools/pmc_summary.py)
n=0
for g in "${groups[@]}"; do
  n=$((n+1))
  timeout -s KILL 150 rocprofv3 --pmc $g -d gpurun_out/pmc_${tag}_$n -o p --output-format csv -- python3 bench.py --no-cpu --no-alone "$@" > gpurun_out/pmc_${tag}_$n.log 2>&1 || { echo "pass $n failed"; tail -5 gpurun_out/pmc_${tag}_$n.log; exit 1; }
  echo "pass $n ok"
done
PMC_SOURCE_HASH=$(cat gpurun_out/pmc_${tag}.hash) python3 tools/pmc_summary.py $tag gpurun_out/pmc_${tag}.json gpurun_out/pmc_${tag}_*/p_counter_collection.csv > gpurun_out/pmc_${tag}.txt || exit 1
rm -rf gpurun_out/pmc_${tag}_[0-9]*
echo "summary gpurun_out/pmc_${tag}.json"
