#!/usr/bin/env python3
"""usage: python tools/pmc_report.py [profiles/pmc_traffic.json | gpurun_out/pmc_<tag>.json ...]
One line per kernel and entry: what the counters of tools/pmc_passes.sh say about it (per launch, kernels one at a time)."""
import json, os, sys
files = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")]
for f in files:
    d = json.load(open(f))
    for w in sorted(d):
        print("== %s (sources %s)" % (w, d[w].get("source_hash")))
        for k, v in sorted(d[w]["kernels"].items()):
            if not v.get("SQ_INSTS_VALU") or v.get("duration_ns", 0) < 3000 or not v.get("GRBM_GUI_ACTIVE") or k.startswith("k_traverse<true"):
                continue
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0          # the counter sums the 8 XCDs
            simd_cyc = 1024 * cyc
            g = lambda name: v.get(name, 0.0)
            print("%-46s n=%4d %8.1f us  VALU %7.2f M inst (%.2f SIMD-cycles each)  valu_busy %.2f  lanes %.2f  TA_busy %.2f  L2 hit %.2f  HBM %7.1f MB = %.2f TB/s  wave: mem-wait %.2f issue-wait %.2f  LDS conflict %.2f  vgpr %s" % (
                k, v["launches_sampled"], v["duration_ns"] / 1e3, v["SQ_INSTS_VALU"] / 1e6, simd_cyc / v["SQ_INSTS_VALU"],
                4 * g("SQ_ACTIVE_INST_VALU") / simd_cyc, g("SQ_THREAD_CYCLES_VALU") / (v["SQ_INSTS_VALU"] * 64), g("TA_TA_BUSY_sum") / (256 * cyc),
                g("TCC_HIT_sum") / max(1.0, g("TCC_REQ_sum")), g("traffic_bytes_per_launch") / 1e6, g("traffic_bytes_per_launch") / v["duration_ns"] / 1e3,
                g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), g("SQ_WAIT_INST_ANY") / max(1.0, g("SQ_WAVE_CYCLES")),
                g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")), (v.get("registers") or {}).get("vgpr")))
