"""Mean counter values per (kernel symbol, grid size) of one rocprofv3 --pmc pass.
usage: python tools/pmc_summary.py <output dir> [kernel-name substring]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:70], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    n = len(next(iter(cs.values())))
    print("%s grid %s (%d launches)" % (key[0], key[1], n))
    for c, v in sorted(cs.items()):
        print("   %-28s %.4g" % (c, sum(v) / len(v)))
