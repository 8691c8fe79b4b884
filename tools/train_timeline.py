"""Timeline analysis of a rocprofv3 kernel trace of bench_train.py: per hardware queue busy time, time with no kernel at
all, and the kernels that run while only ONE queue is busy, over the last steps.   python tools/train_timeline.py TRACE.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundaries: the fused Adam launches (multi_tensor_apply ... FusedOptimizer) mark the end of a step
ends = [int(r["End_Timestamp"]) for r in rows if "FusedOptimizer" in r["Kernel_Name"] or "fused_adam" in r["Kernel_Name"].lower()]
# group Adam launches that are close together
marks = []
for t in ends:
    if not marks or t - marks[-1] > 2_000_000:
        marks.append(t)
    else:
        marks[-1] = t
assert len(marks) > nsteps + 1, len(marks)
t0, t1 = marks[-nsteps - 1], marks[-1]
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
span = (t1 - t0) / 1e6
print("%d steps, %.3f ms per step, %d launches per step" % (nsteps, span / nsteps, len(sel) // nsteps))
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print("queue %s: %5d launches/step, busy %.3f ms/step (sum of durations %.3f)" % (q, len(iv) // nsteps, union([(s, e) for s, e, _ in iv]) / 1e6 / nsteps, sum(e - s for s, e, _ in iv) / 1e6 / nsteps))
allbusy = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel])
print("some kernel running: %.3f ms/step; idle: %.3f ms/step" % (allbusy / 1e6 / nsteps, (t1 - t0 - allbusy) / 1e6 / nsteps))
# idle gaps by the kernel that follows them
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in sel)
gaps = collections.Counter(); ce = ev[0][1]
for s, e, n in ev[1:]:
    if s > ce: gaps[n[:70]] += s - ce
    ce = max(ce, e)
print("idle time by the kernel that ends the gap (us/step):")
for n, g in gaps.most_common(12):
    print("  %8.1f  %s" % (g / 1e3 / nsteps, n))
