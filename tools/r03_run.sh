set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_model.py -x -q -k "rescaled or full_workloads" --timeout 180 --timeout-method thread -s 2>&1 | grep -v amdgpu | tail -15
