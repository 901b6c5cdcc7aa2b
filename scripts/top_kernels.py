"""Print the top-N rows of a rocprofv3 *_kernel_stats.csv (name shortened)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = re.sub(r"at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_\d+_NS::detail::", "", r["Name"])
    name = re.sub(r"<.*", "", name)[:60]
    print(f"{name:60s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs']) / 1e3:10.1f} us  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
