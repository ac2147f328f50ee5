#!/usr/bin/env python3
"""tools/pmc_calibrate.py BYTES.txt FETCH_DIR WRITE_DIR OUT.json -- bytes per counter unit of FETCH_SIZE / WRITE_SIZE for each access
pattern of tools/probes/fetch_calib.hip (counters in KB as rocprofv3 reports them on gfx950: factor 1.0 = the counter x 1024 is the
byte count; MI355X_MICROARCH.md states 2.0 for 16 B/lane streaming reads)."""
import csv, glob, json, re, sys, collections

known = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(\S+)\s+(.*)$", line.strip())
    if m:
        known[m.group(1)] = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", m.group(2))}
val = collections.defaultdict(dict)
for d, cname in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != cname:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            val[k][cname] = val[k].get(cname, 0.0) + float(r["Counter_Value"])
out = {}
print("# bytes moved / (counter x 1024) per access pattern; buffer 2 GiB (8 x the Infinity Cache), each byte touched once")
print("%-28s %14s %14s %10s %14s %10s" % ("kernel", "known bytes", "FETCH_SIZE KB", "rd factor", "WRITE_SIZE KB", "wr factor"))
for k, b in known.items():
    name = next((n for n in val if n.replace(" ", "") == k.replace(" ", "")), None)
    if name is None:
        name = next((n for n in val if k.split("<")[0] in n and (("<" not in k) or k.split("<")[1].rstrip(">") in n)), None)
    v = val.get(name, {})
    rd = b.get("bytes_read") or b.get("bytes_unique")
    wr = b.get("bytes_written")
    e = {"known": b, "FETCH_SIZE_KB": v.get("FETCH_SIZE"), "WRITE_SIZE_KB": v.get("WRITE_SIZE")}
    e["read_factor"] = rd / (v["FETCH_SIZE"] * 1024.0) if rd and v.get("FETCH_SIZE") else None
    e["write_factor"] = wr / (v["WRITE_SIZE"] * 1024.0) if wr and v.get("WRITE_SIZE") else None
    if "bytes_requested" in b and v.get("FETCH_SIZE"):
        e["requested_over_counter"] = b["bytes_requested"] / (v["FETCH_SIZE"] * 1024.0)
    out[k] = e
    print("%-28s %14d %14.6g %10s %14.6g %10s" % (k, rd or wr or 0, v.get("FETCH_SIZE", float("nan")),
          "%.3f" % e["read_factor"] if e["read_factor"] else "-", v.get("WRITE_SIZE", float("nan")),
          "%.3f" % e["write_factor"] if e["write_factor"] else "-"))
json.dump(out, open(sys.argv[4], "w"), indent=1)
