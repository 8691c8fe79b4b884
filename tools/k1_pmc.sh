#!/bin/bash
# PMC passes over the QKV-shaped k = 1 GEMM (tools/k1_probe.py); $1 = tag under gpurun_out/
out=gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
export K1_SHAPE=${2:-qkv} TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d $R/$out/pmc1 -- python3 $R/tools/k1_probe.py > $R/$out/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR -d $R/$out/pmc2 -- python3 $R/tools/k1_probe.py > $R/$out/pmc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $R/$out/pmc3 -- python3 $R/tools/k1_probe.py > $R/$out/pmc3.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$out/pmc4 -- python3 $R/tools/k1_probe.py > $R/$out/pmc4.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/$out/pmc5 -- python3 $R/tools/k1_probe.py > $R/$out/pmc5.log 2>&1 || exit 1
cd $R
for p in pmc1 pmc2 pmc3 pmc4 pmc5; do python tools/pmc_summary.py $out/$p ${3:-conv_gemm} >> $out/pmc_summary_$K1_SHAPE.txt 2>&1; done
rm -rf $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 $out/pmc5
