set -e
timeout -k 10 200 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "attention" --timeout 120 --timeout-method thread -s 2>&1 | grep "split-fp16\|passed\|failed"
timeout -k 10 100 python tools/bench_phone_attention.py 2>&1 | grep -v amdgpu
