#!/usr/bin/env python3
"""tools/pmc_traffic_json.py KEY gpurun_out/TAG_pmc_traffic.txt COMPUTES -- fold a PMC summary (tools/pmc_summary.py output of the
FETCH_SIZE and WRITE_SIZE passes over tools/pmc_workload.py, which runs COMPUTES rh_batch_compute calls) into
profiles/pmc_traffic.json: per kernel the HBM-side bytes per DISPATCH and the dispatches per STEP.
bytes = FETCH_SIZE x read_factor + WRITE_SIZE x write_factor (KB counters), factors from profiles/pmc_calibration.json -- measured on
this repository's own access patterns by tools/calibrate_pmc.sh (8 B/lane and 16 B/lane streaming reads: 2.0, the strip kernels'
row-segment pattern: counter x 2 = bytes that crossed the fabric; 8 / 16 B/lane stores: 1.0); MI355X_MICROARCH.md, section HBM."""
import json, os, re, sys
key, path, computes = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rf, wf = 2.0, 1.0
try:
    cal = json.load(open(os.path.join(root, "profiles", "pmc_calibration.json")))
    rf = float(cal["factors"]["read"]); wf = float(cal["factors"]["write"])
except (OSError, ValueError, KeyError):
    pass
per, disp = {}, {}
name = None
for line in open(path):
    m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
    if m:
        name = m.group(1).strip()
        disp[name] = max(disp.get(name, 0), int(m.group(2)))
        continue
    m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+total=\S+\s+per_dispatch=(\S+)", line)
    if m and name:
        per.setdefault(name, {})[m.group(1)] = float(m.group(2))
out = {"kernels": {}}
for k, v in per.items():
    if "rh::" not in k and not k.startswith("cand_") and not k.startswith("collect_"):
        continue
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out["kernels"][k] = {"fetch_KB": f, "write_KB": w, "bytes_per_dispatch": (rf * f + wf * w) * 1024.0,
                         "dispatches_per_step": disp[k] / float(computes)}
out["read_factor"], out["write_factor"], out["computes"] = rf, wf, computes
dst = os.path.join(root, "profiles", "pmc_traffic.json")
try:
    doc = json.load(open(dst))
except (OSError, ValueError):
    doc = {}
doc["note"] = ("HBM-side bytes per DISPATCH and dispatches per step, by kernel, from rocprofv3 PMC passes over tools/pmc_workload.py "
               "(FETCH_SIZE x read_factor + WRITE_SIZE x write_factor, KB counters; factors: profiles/pmc_calibration.json); "
               "key = <model>_n<seqlen>_b<pairs per step>; bench.py builds roofline.traffic from the entry whose source_hash matches")
sys.path.insert(0, root)
import bench
out["source_hash"] = bench.source_hash()   # bench.py reports an entry only for the kernel sources it was measured on
doc[key] = out
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}), len(out["kernels"]), "kernels")
