#!/usr/bin/env python3
"""Step timeline from a rocprofv3 --kernel-trace csv: python tools/timeline.py <kernel_trace.csv> [steps_to_average]
Steps are delimited by the first kernel of a step (embed_fwd_kernel, or fused_layer_fwd_kernel when the embedding rides in it); for each kernel position in the step prints the average start offset,
duration and the gap to the previous kernel's end (negative = overlap), plus the queue it ran on."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")) for r in rows]
first = "embed_fwd_kernel" if any(n.startswith("embed_fwd_kernel") for n in names) else "fused_layer_fwd_kernel"
steps, cur = [], None
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:44]
    if n.startswith(first):
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
steps = [s for s in steps[:-1] if len(s) == len(steps[len(steps) // 2])][-nst:]
k = len(steps[0])
print(f"{len(steps)} steps of {k} kernels; step period avg "
      f"{(steps[-1][0][1] - steps[0][0][1]) / (len(steps) - 1) / 1e3:.2f} us")
for i in range(k):
    st = sum(s[i][1] - s[0][1] for s in steps) / len(steps) / 1e3
    du = sum(s[i][2] - s[i][1] for s in steps) / len(steps) / 1e3
    prev_end = sum(max(x[2] for x in s[:i]) - s[0][1] for s in steps) / len(steps) / 1e3 if i else 0.0
    print(f"{i:2d} q{steps[0][i][3]:>2s} start {st:8.2f}  dur {du:7.2f}  gap {st - prev_end:6.2f}  {steps[0][i][0]}")
