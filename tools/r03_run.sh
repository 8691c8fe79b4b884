set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -k "embed_kernels or training_forward_loss" --timeout 180 --timeout-method thread 2>&1 | grep -v amdgpu | tail -25
