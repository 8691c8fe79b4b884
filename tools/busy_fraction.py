"""GPU busy fraction of a rocprofv3 --kernel-trace run: union of the kernel intervals over the span of the last
`tail` fraction of the trace (skips warm-up), and the average number of kernels in flight."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
lo, hi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3, float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
n = len(rows)
rows = rows[int(lo * n): int(hi * n)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, total = 0, None, None, 0
for s, e, _ in rows:
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms, busy %.1f %%, kernel-time sum / span = %.2f, kernels %d" % ((t1 - t0) / 1e6, 100.0 * busy / (t1 - t0), total / (t1 - t0), len(rows)))
gaps = []
cur_e = None
for s, e, name in rows:
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, name))
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort(reverse=True)
print("largest idle gaps (us) before kernel:", [(round(g / 1e3, 1), nm[:40]) for g, nm in gaps[:8]])
print("idle total ms", sum(g for g, _ in gaps) / 1e6)
