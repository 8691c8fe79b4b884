set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03_k; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_zy_determinism.py -x -q --timeout 180 --timeout-method thread > $O/test_model.log 2>&1 || { tail -30 $O/test_model.log; exit 1; }
tail -2 $O/test_model.log
rm -f $O/ab.txt
for rep in 1 2 3; do for sa in 0 1; do timeout -k 10 100 python bench.py --no-cpu-baseline --engine-set fused_proj_ln=$sa 2>/dev/null | python3 -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('fused_proj_ln', d['config']['launch_env']['engine_set'], d['ms_per_step'], 'pure bf16', d['config']['pure_bf16']['ms_per_step'])" >> $O/ab.txt; done; done
cat $O/ab.txt
timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep "f16x3\|bf16  " 
