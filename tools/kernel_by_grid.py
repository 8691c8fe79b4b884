"""Average duration per (kernel, grid) of a rocprofv3 --kernel-trace run; usage: kernel_by_grid.py DIR [filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv_gemm"
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if flt in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:64], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
    v = v[len(v) // 3:]
    print("%-66s grid %7s x %s wg %4s  n=%4d avg %7.1f us  total %8.1f" % (*k, len(v), sum(v) / len(v), sum(v)))
