set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03_k; mkdir -p $O
timeout -k 10 200 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "proj_add" --timeout 120 --timeout-method thread > $O/test.log 2>&1 || { tail -40 $O/test.log; exit 1; }
tail -1 $O/test.log
timeout -k 10 100 python tools/pl_sweep.py 2>&1 | grep -v amdgpu
K1_SHAPE=proj_ln timeout -k 10 100 python tools/k1_probe.py 2>&1 | grep -v amdgpu | tee $O/probe.txt
