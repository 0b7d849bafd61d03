#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, csv output) into per-kernel, per-launch means.

usage: python tools/pmc_summary.py <workload> <out.json> <counter_collection.csv>...

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes, so reads are doubled; WRITE_SIZE is taken as is.
    traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
bench.py reads the resulting file (profiles/pmc_traffic.json) for `roofline.traffic`.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def source_hash(root):
    """sha1 over the sources every kernel is built from (csrc/*.hip, csrc/*.h, include/rt_abi.h, the Makefile's flags):
    a PMC entry is only used by bench.py while this still matches the tree"""
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(root, "raytracer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "raytracer_amd", "csrc", "*.h")))
    for f in files + [os.path.join(root, "include", "rt_abi.h"), os.path.join(root, "Makefile")]:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def short_name(kernel):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", kernel)
    return (m.group(1) + (m.group(2) or "")) if m else kernel[:40]


def main():
    workload, out_path, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    durations = collections.defaultdict(dict)  # kernel -> (file, dispatch) -> ns: launch time under the profiler (kernels run one at a time)
    regs = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short_name(r["Kernel_Name"])
            names = [k]
            if "<" in k and not k.startswith("k_traverse<true"):
                names.append(k.split("<")[0])  # all instantiations of a kernel together (not the counting variant)
            for k in names:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
                try:
                    durations[k][(f, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
            if "VGPR_Count" in r:
                regs[names[0]] = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")}
    kernels = {}
    for k, counters in acc.items():
        if not k.startswith("k_"):
            continue
        per = {c: v / max(1, len(launches[k][c])) for c, v in counters.items()}
        per["launches_sampled"] = max(len(s) for s in launches[k].values())
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            per["traffic_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        if durations[k]:
            per["duration_ns"] = sum(durations[k].values()) / len(durations[k])
        if k in regs:
            per["registers"] = regs[k]
        kernels[k] = per
    try:
        doc = json.load(open(out_path))
    except (OSError, ValueError):
        doc = {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc[workload] = {"note": "per-launch means; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction); "
                             "duration_ns = launch time under the profiler (one kernel at a time)",
                     "source_hash": os.environ.get("PMC_SOURCE_HASH") or source_hash(root), "kernels": kernels}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    for k, per in sorted(kernels.items()):
        print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in per.items()})


if __name__ == "__main__":
    main()
