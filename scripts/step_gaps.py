"""GPU timeline of the steady-state training steps from a rocprofv3 kernel trace: busy time, idle gaps
and what follows the big gaps.  Usage: step_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# step boundaries: the fused / foreach Adam kernels
adam = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
ends = []
for i in adam:
    if not ends or i - ends[-1] > 50:
        ends.append(i)
print("launches", len(rows), "steps found", len(ends))
ends = ends[-9:-1]  # eight steady-state steps before the last
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[k + 1][0] - seg[k][1], seg[k + 1][2][:60], seg[k][2][:40]) for k in range(len(seg) - 1)]
    idle = sum(max(g[0], 0) for g in gaps)
    big = sorted(gaps, reverse=True)[:3]
    print(f"step: {len(seg)} launches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms; biggest gaps:",
          "; ".join(f"{g[0] / 1e3:.0f} us before {g[1]}" for g in big))
seg = rows[ends[-2] + 1:ends[-1] + 1]
acc = collections.Counter()
for s, e, n in seg:
    key = "encoder/conv+bn" if any(t in n for t in ("Conv", "igemm", "batch_norm", "BatchNorm", "max_pool", "transpose", "Im2", "Col2", "SubTensor")) else \
          ("mr:: kernels" if "mr::" in n else ("elementwise" if "elementwise" in n else ("gemm" if "Cijk" in n else "other")))
    acc[key] += e - s
print({k: round(v / 1e6, 2) for k, v in acc.items()})
cnt = collections.Counter()
for s, e, n in seg:
    if not any(t in n for t in ("Conv", "igemm", "batch_norm", "BatchNorm", "max_pool", "transpose", "Im2", "Col2", "SubTensor")):
        cnt[n[:90]] += 1
print("non-encoder launches in the last step:", sum(cnt.values()))
for k, v in cnt.most_common(25):
    print(f"{v:4d}  {k}")
