#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 --pmc csv output(s): python tools/pmc_summary.py dir [filter]"""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:60]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, cs in acc.items():
    if flt in n:
        print(n, {k: round(sum(v) / len(v)) for k, v in sorted(cs.items())})
