#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: sum of each counter and dispatch count."""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -len(cnt[k])):
    print("%-60s dispatches=%d" % (k[:60], len(cnt[k])))
    for c, v in sorted(agg[k].items()):
        print("    %-28s total=%.4g  per_dispatch=%.4g" % (c, v, v / max(1, len(cnt[k]))))
