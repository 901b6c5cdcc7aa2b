"""Kernels of ONE steady-state training step by total time, from a rocprofv3 kernel trace of bench.py (step
boundaries = the Adam kernels).  Usage: step_top_kernels.py <kernel_trace.csv> [rows] [sequence-file]
(the optional third argument writes the step's launches in issue order: start offset, duration, gap, name)"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
ends = []
for i in adam:
    if not ends or i - ends[-1] > 50:
        ends.append(i)
seg = rows[ends[-3] + 1:ends[-2] + 1]
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in seg:
    k = n.replace("void ", "")[:100]
    tot[k] += e - s
    cnt[k] += 1
span = seg[-1][1] - seg[0][0]
print(f"step: {len(seg)} launches, span {span / 1e6:.2f} ms, busy {sum(tot.values()) / 1e6:.2f} ms")
for k, v in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print(f"{cnt[k]:5d} x {v / cnt[k] / 1e3:9.1f} us = {v / 1e6:7.3f} ms  {k}")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        prev_end = seg[0][0]
        for s, e, n in seg:
            fh.write(f"{(s - seg[0][0]) / 1e3:10.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {n.replace('void ', '')[:110]}\n")
            prev_end = max(prev_end, e)
