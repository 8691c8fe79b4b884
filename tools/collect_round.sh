#!/bin/bash
# End-of-round evidence on the MI355X box: bench line, rocprofv3 kernel stats (serial + overlapped), training line,
# config table, HBM traffic of the dominant kernel (two --pmc passes).  Run through gpurun from the repo root
# (TAG=r03_z bash tools/collect_round.sh); results land in gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
T=${TAG:-r03_z}
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo bench ok
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --serial --steps 20 --warmup 5 --no-cpu-baseline --no-pure > $O/${T}_bench_serial.json 2> $O/prof_serial.log; echo serial ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_overlap -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pure > $O/${T}_bench_overlap.json 2> $O/prof_overlap.log; echo overlap ok
export FS2_REPLAY=dec
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/ffn_conv_replay.py > $O/pmc_fetch.log 2>&1; echo fetch ok
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/ffn_conv_replay.py > $O/pmc_write.log 2>&1; echo write ok
unset FS2_REPLAY
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${T}_ffn_conv_traffic.json 24 Li9E > $O/pmc_traffic.log 2>&1; echo traffic ok
timeout -k 10 300 python3 bench_train.py --steps 30 --warmup 5 > $O/${T}_train_dropout.json 2> $O/train.err; echo train ok
timeout -k 10 400 python3 tools/bench_configs.py > $O/${T}_configs.txt 2>&1; echo configs ok
timeout -k 10 300 python3 tools/bench_latency.py > $O/${T}_batch1_latency.txt 2>&1; echo latency ok
timeout -k 10 300 python3 bench.py --dtype f16x3 --no-cpu-baseline > $O/${T}_bench_f16x3.json 2>> $O/${T}_bench.err; echo f16x3 ok
rm -f $O/prof_*/*/*kernel_trace.csv $O/prof_*/*/*agent_info.csv; rm -rf $O/pmc_fetch $O/pmc_write
cp $O/prof_serial/*/*kernel_stats.csv $O/${T}_bench_serial_kernel_stats.csv
cp $O/prof_overlap/*/*kernel_stats.csv $O/${T}_bench_overlap_kernel_stats.csv
ls $O | grep $T
