"""Aggregate rocprofv3 counter_collection CSVs (one per PMC pass) per kernel: mean per dispatch.
A kernel launched with more than one grid size (the vertex-colour render runs per frame, B meshes, and for
both frames of a pair, 2B meshes) is listed once per grid.  `--json <file>` also writes the FETCH_SIZE /
WRITE_SIZE means converted to HBM bytes per launch (what bench.py reads for `roofline.traffic`): under the
bare kernel name for its LARGEST grid, under `name @grid=N` for the others."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
acc = defaultdict(lambda: defaultdict(list))  # (name, grid) -> counter -> values
for f in sorted(glob.glob(os.path.join(d, "pass*_counters.csv"))):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "mr::" not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = (k.split("(")[0].replace("void ", ""), int(row["Grid_Size"]))
        for (disp, cname), v in per_dispatch.items():
            acc[names[disp]][cname].append(v)
grids = defaultdict(list)
for name, grid in acc:
    grids[name].append(grid)
label = {}
for name, gs in grids.items():
    for g in gs:
        label[(name, g)] = name if g == max(gs) else f"{name} @grid={g}"
        if len(gs) > 1 and g == max(gs):
            label[(name, g)] = name  # bare name = largest grid
for key in sorted(acc, key=lambda k: label[k]):
    multi = len(grids[key[0]]) > 1
    print(label[key] + (f"   [grid {key[1]}{', largest' if key[1] == max(grids[key[0]]) else ''}]" if multi else ""))
    for c in sorted(acc[key]):
        v = acc[key][c]
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
if json_out:
    rec = {"_doc": "mean per dispatch from scripts/pmc.sh (separate --pmc passes). hbm_bytes = 1024 * (2 * FETCH_SIZE + "
                   "WRITE_SIZE): gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM "
                   "section), so this is the upper estimate; hbm_bytes_low uses FETCH_SIZE as is.  A kernel launched "
                   "with several grid sizes is listed under its bare name for the largest grid and under "
                   "'name @grid=N' for the others."}
    for key in sorted(acc, key=lambda k: label[k]):
        c = acc[key]
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fe, wr = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        rec[label[key]] = {"grid": key[1], "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                           "hbm_bytes": int(1024 * (2 * fe + wr)), "hbm_bytes_low": int(1024 * (fe + wr))}
    with open(json_out, "w") as fh:
        json.dump(rec, fh, indent=1)
