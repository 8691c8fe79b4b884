#!/bin/bash
# PMC passes over tools/conv256_check.py (new kernel only); $1 = output tag under gpurun_out/
out=gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
export C256_ONLY=new TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $R/$out/pmc1 -- python3 $R/tools/conv256_check.py 25611 > $R/$out/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES -d $R/$out/pmc2 -- python3 $R/tools/conv256_check.py 25611 > $R/$out/pmc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $R/$out/pmc3 -- python3 $R/tools/conv256_check.py 25611 > $R/$out/pmc3.log 2>&1 || exit 1
cd $R
for p in pmc1 pmc2 pmc3; do python tools/pmc_summary.py $out/$p conv_gemm256 >> $out/pmc_summary.txt 2>&1; done
rm -rf $out/pmc1 $out/pmc2 $out/pmc3
