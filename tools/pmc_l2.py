"""Per-kernel sums of rocprofv3 --pmc counters (counter_collection.csv) -> hit rates; usage: pmc_l2.py DIR [name filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv_gemm"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    if flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    n = max(cnt[(k, c)] for c in v)
    print(k, "launches", n, {c: round(x / n) for c, x in v.items()})
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print("   L2 hit rate %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
