"""summary of a rocprofv3 kernel trace: per-kernel totals, union of the kernel intervals, and how many kernels overlap on average.
usage: python scratch/trace_summary.py <dir with *_kernel_trace.csv> [skip_first_n_launches_of_k_prep]"""
import csv, glob, os, sys, collections
tr = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(tr))]
rows.sort()
# steady state: from the start of the 3rd-from-last ... keep the last 5 calls: a call starts with its first k_prep (8 classes -> 8 k_prep per call)
preps = [a for a, b, k in rows if k == "k_prep"]
ncls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t_first = preps[-5 * ncls] if len(preps) >= 5 * ncls else rows[0][0]
rows = [r for r in rows if r[0] >= t_first]
per = collections.defaultdict(lambda: [0, 0])
for a, b, k in rows: per[k][0] += b - a; per[k][1] += 1
busy, cur_a, cur_b = 0, rows[0][0], rows[0][1]
for a, b, k in rows[1:]:
    if a <= cur_b: cur_b = max(cur_b, b)
    else: busy += cur_b - cur_a; cur_a, cur_b = a, b
busy += cur_b - cur_a
span = max(b for a, b, k in rows) - rows[0][0]
tot = sum(v[0] for v in per.values())
print(f"last 5 calls: span {span / 5e6:.3f} ms per call; union of kernel intervals {busy / 5e6:.3f} ms per call ({busy / span:.3f} of the span); sum of kernel durations {tot / 5e6:.3f} ms per call = {tot / busy:.2f} kernels in flight on average")
for k, (ns_, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:9]:
    print(f"  {k[:44]:44s} {ns_ / 5e6:8.3f} ms per call  {n // 5:5d} launches per call")
