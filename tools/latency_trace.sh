# kernel-by-kernel trace of one batch-1 forward: usage (on the GPU box)  bash tools/latency_trace.sh OUT.txt [n_phones]   (120: BASELINE config 1)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/lat_trace; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/tools/latency_profile.py f16x3 ${2:-120} > $O/log.txt 2>&1
cd $R
python3 - $O > ${1:-$O/seq.txt} <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "embed_posenc" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1]]          # the last complete forward
t0, tot = int(seq[0]["Start_Timestamp"]), 0.0
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("void fs2::", "").replace("fs2::", "")[:70]
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"])
    print("%7.1f +%5.1f  wgs %5d x %4d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, wgs, int(r["Workgroup_Size_X"]), n))
print("sum of kernel durations: %.1f us in %d launches" % (tot, len(seq)))
P
rm -rf $O/prof
