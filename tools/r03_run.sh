set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03_j; mkdir -p $O
timeout -k 10 200 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "conv_gemm256" --timeout 120 --timeout-method thread > $O/test.log 2>&1 || { tail -40 $O/test.log; exit 1; }
tail -3 $O/test.log
