#!/bin/bash
# PMC passes over fs2_attention alone on one case of tools/bench_attention.py:  bash tools/attn_pmc.sh TAG "config 4 B=4"
out=gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
export FS2_BENCH_CASE="${2:-config 4 B=4}" TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d $R/$out/pmc1 -- python3 $R/tools/bench_attention.py bf16 > $R/$out/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU -d $R/$out/pmc2 -- python3 $R/tools/bench_attention.py bf16 > $R/$out/pmc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_IFETCH -d $R/$out/pmc3 -- python3 $R/tools/bench_attention.py bf16 > $R/$out/pmc3.log 2>&1 || true
cd $R
for p in pmc1 pmc2 pmc3; do python tools/pmc_summary.py $out/$p attention32 >> $out/pmc_summary.txt 2>&1; done
rm -rf $out/pmc1 $out/pmc2 $out/pmc3
