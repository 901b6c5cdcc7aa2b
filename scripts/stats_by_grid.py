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
# a kernel launched over a list whose length the host guesses (the forward tile kernel: a different grid nearly every
# training step) would fill the table with one-call rows: grids with fewer than three calls are shown as one row per kernel
n_rare = defaultdict(int)
for (name, _), d in acc.items():
    n_rare[name] += len(d) < 3
rare = defaultdict(list)
for (name, grid), d in list(acc.items()):
    if len(d) < 3 and n_rare[name] > 3:
        rare[name].append((grid, d))
        del acc[(name, grid)]
print(f"{'kernel':48s} {'grid':>26s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'median':>9s} {'max us':>9s}")
rows = [(name, grid, str(grid), d) for (name, grid), d in acc.items()]
for name, lst in rare.items():
    d = [x for _, dd in lst for x in dd]
    lo, hi = min(g for g, _ in lst), max(g for g, _ in lst)
    rows.append((name, lo, f"{lo}..{hi} ({len(lst)} grids)", d))
for name, _, gtxt, d in sorted(rows, key=lambda r: (r[0], r[1])):
    d.sort()
    print(f"{name[:48]:48s} {gtxt:>26s} {len(d):6d} {sum(d) / len(d):9.1f} {d[0]:9.1f} {d[len(d) // 2]:9.1f} {d[-1]:9.1f}")
