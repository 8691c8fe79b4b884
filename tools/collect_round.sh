#!/bin/bash
# End-of-round evidence on the MI355X box: bench line, rocprofv3 kernel stats (serial + overlapped), training line,
# config table.  Run through gpurun from the repo root; results land in gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py > $O/${TAG:-r02_m}_bench.json 2> $O/${TAG:-r02_m}_bench.err; echo bench ok
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --serial --steps 20 --warmup 5 --no-cpu-baseline --no-pure > $O/${TAG:-r02_m}_bench_serial.json 2> $O/prof_serial.log; echo serial ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_overlap -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pure > $O/${TAG:-r02_m}_bench_overlap.json 2> $O/prof_overlap.log; echo overlap ok
cd $R
timeout -k 10 300 python3 bench_train.py --steps 30 --warmup 5 > $O/${TAG:-r02_m}_train_dropout.json 2> $O/train.err; echo train ok
timeout -k 10 400 python3 tools/bench_configs.py > $O/${TAG:-r02_m}_configs.txt 2>&1; echo configs ok
rm -f $O/prof_*/*/*kernel_trace.csv $O/prof_*/*/*agent_info.csv
find $O -name "*kernel_stats.csv"
