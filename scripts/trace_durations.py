"""Durations (us) of every dispatch of kernels matching a substring, in launch order, from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
group = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for i in range(0, len(d), group):
    g = d[i:i + group]
    print(f"[{i:4d}] n={len(g):3d} min {min(g):8.1f} med {sorted(g)[len(g) // 2]:8.1f} max {max(g):8.1f}")
