"""Where do DistributedDataParallel's RCCL all-reduce kernels sit in a training step?  Reads a rocprofv3
kernel_trace.csv of `HOC_FORCE_DDP=1 python bench.py ...` and, for the last steady-state step (steps are delimited
by the fused Adam kernel), prints every RCCL kernel with its start offset, duration, and the compute kernels that
run concurrently (other streams), plus the share of the all-reduce time that overlaps compute.
Usage: ddp_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?")))
rows.sort()
is_comm = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
is_adam = lambda n: "adam" in n.lower() and "multi_tensor" in n.lower()
adam_idx = [i for i, r in enumerate(rows) if is_adam(r[2])]
if len(adam_idx) < 3:
    sys.exit("fewer than three optimiser kernels in the trace")
# one step = (end of the previous Adam launch group, end of this one]; take the last full step
ends = [rows[i][1] for i in adam_idx]
t1 = ends[-1]
t0 = max(e for e in ends if e < t1 - 5_000_000) if any(e < t1 - 5_000_000 for e in ends) else ends[0]
step = [r for r in rows if t0 < r[0] <= t1]
comm = [r for r in step if is_comm(r[2])]
comp = [r for r in step if not is_comm(r[2])]
print(f"step: {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches, {len(comm)} RCCL kernels, "
      f"{sum(e - s for s, e, _, _ in comm) / 1e3:.1f} us of collective kernel time")
busy = sorted((s, e) for s, e, _, _ in comp)
tot_overlap = 0
for s, e, name, stream in comm:
    ov = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce in busy)
    tot_overlap += min(ov, e - s)
    during = sorted({n.split("(")[0][:50] for cs, ce, n, _ in comp if ce > s and cs < e})
    print(f"  +{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  stream {stream}  {name.split('(')[0][:60]}")
    print(f"      concurrent compute ({min(ov, e - s) / 1e3:.1f} us overlapped): {', '.join(during[:4])}{' ...' if len(during) > 4 else ''}")
if comm:
    tc = sum(e - s for s, e, _, _ in comm)
    print(f"all-reduce kernel time overlapped with compute kernels: {100.0 * tot_overlap / tc:.0f} %")
    last_comp_end = max(e for s, e, _, _ in comp)
    print(f"last RCCL kernel ends {(max(e for s, e, _, _ in comm) - t0) / 1e6:.3f} ms into the step; "
          f"last compute kernel at {(last_comp_end - t0) / 1e6:.3f} ms")
