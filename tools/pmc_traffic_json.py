#!/usr/bin/env python3
"""tools/pmc_traffic_json.py KEY gpurun_out/TAG_pmc_traffic.txt -- fold a PMC summary (tools/pmc_summary.py output of
the FETCH_SIZE and WRITE_SIZE passes) into profiles/pmc_traffic.json as HBM-side bytes per dispatch by kernel name:
FETCH_SIZE x 2 + WRITE_SIZE, counters in KB (gfx950 correction of MI355X_MICROARCH.md, section HBM)."""
import json, os, re, sys
key, path = sys.argv[1], sys.argv[2]
per = {}
name = None
for line in open(path):
    m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
    if m:
        name = m.group(1).strip()
        continue
    m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+total=\S+\s+per_dispatch=(\S+)", line)
    if m and name:
        per.setdefault(name, {})[m.group(1)] = float(m.group(2))
out = {k: (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for k, v in per.items() if "rh::" in k}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", "pmc_traffic.json")
try:
    doc = json.load(open(dst))
except (OSError, ValueError):
    doc = {}
doc["note"] = ("HBM-side bytes per DISPATCH by kernel from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KB counters, gfx950 "
               "correction); key = <model>_n<seqlen>_b<pairs per step>; bench.py reports the dominant kernel's entry as roofline.traffic")
sys.path.insert(0, root)
import bench
out["source_hash"] = bench.source_hash()   # bench.py reports an entry only for the kernel sources it was measured on
doc[key] = out
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
