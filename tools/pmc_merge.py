#!/usr/bin/env python3
"""usage: python tools/pmc_merge.py gpurun_out/pmc_<tag>.json ...  -- merges the entries of per-tag PMC summaries (tools/pmc_passes.sh)
into profiles/pmc_traffic.json"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", "pmc_traffic.json")
doc = json.load(open(out)) if os.path.exists(out) else {}
for f in sys.argv[1:]:
    for k, v in json.load(open(f)).items():
        doc[k] = v
        print("merged", k, v.get("source_hash"))
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
