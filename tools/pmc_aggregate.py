#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes into the per-kernel csv kept under profiles/:
    python tools/pmc_aggregate.py <dir with the passes> <out.csv>"""
import csv, glob, sys, collections, re, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsarec_amd import _lib as Lb
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    o.write("# rocprofv3 --pmc <counters> -- python3 tools/pmc_run.py 4   (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*), C1 shape, averages per launch\n")
    o.write("# FETCH_SIZE / WRITE_SIZE in KiB as reported; on gfx950 FETCH_SIZE counts 1/2 of wide streaming reads -> HBM read bytes ~= 2 * FETCH_SIZE * 1024\n")
    o.write("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES is cycles\n")
    o.write(f"# library_sha16={Lb.source_sha16()}\n")
    o.write("Kernel,Counter,AvgPerLaunch,Launches\n")
    for n in sorted(acc):
        if n.startswith("at::") or "elementwise" in n or "Cijk" in n:
            continue
        for c in sorted(acc[n]):
            v = acc[n][c]
            o.write(f'"{n}",{c},{round(sum(v) / len(v))},{len(v)}\n')
