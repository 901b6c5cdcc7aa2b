"""Per-(kernel, grid size) dispatch statistics of the mr:: kernels from a rocprofv3 kernel_trace.csv
(`--kernel-trace --stats` aggregates by name only; the vertex-colour render runs per frame -- B meshes -- in
the per-kernel benchmark and for both frames of a pair -- 2B meshes -- inside the training step).
Usage: stats_by_grid.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "mr::" not in name:
        continue
    name = name.split("(")[0].replace("void ", "")
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':48s} {'grid':>10s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'median':>9s} {'max us':>9s}")
for (name, grid), d in sorted(acc.items()):
    d.sort()
    print(f"{name[:48]:48s} {grid:10d} {len(d):6d} {sum(d) / len(d):9.1f} {d[0]:9.1f} {d[len(d) // 2]:9.1f} {d[-1]:9.1f}")
