"""Start / end of the kernels of the last D + E + F launches in a rocprofv3 kernel trace (csv): do the walk and the gather overlap?
    python scripts/r5_def_trace.py <p_kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(k in n for k in ("mark_owners", "compact_owners", "strip_list", "pixel_map_strip", "gather_kernel<true, true, true>")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("mr::", "")[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
strips = [i for i, r in enumerate(rows) if "pixel_map_strip" in r[2]]
for i in strips[-3:]:
    t0 = rows[max(i - 4, 0)][0]
    print("--- launch")
    for r in rows[max(i - 4, 0): i + 2]:
        print("  %-42s queue %s stream %s  start %7.1f us  end %7.1f us  (%.1f us)" % (r[2], r[3], r[4], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3))

# everything on the device around the last launch (all kernels, all queues)
allk = []
for r in csv.DictReader(open(sys.argv[1])):
    allk.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
allk.sort()
if strips:
    i = strips[-1]
    lo, hi = rows[max(i - 4, 0)][0] - 250000, rows[min(i + 1, len(rows) - 1)][1] + 20000
    print("--- every kernel between %.0f us before the last launch's first kernel and its end" % 250)
    for a, b, n, q in allk:
        if b >= lo and a <= hi:
            print("  q%-2s %9.1f .. %9.1f  (%7.1f us)  %s" % (q, (a - lo) / 1e3, (b - lo) / 1e3, (b - a) / 1e3, n))
