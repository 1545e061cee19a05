#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv: python tools/prof_summary.py <kernel_stats.csv> [steps]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = re.sub(r'\(.*', '', r['Name'].replace('void ', ''))
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls/step={float(r['Calls'])/steps:5.1f} avg={float(r['AverageNs'])/1e3:8.2f}us  {n[:110]}")
print('total ms', tot / 1e6, ' per step us', tot / 1e3 / steps)
