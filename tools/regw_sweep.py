"""fs2_gemm_regw over token counts (N = 768 and 256, bf16 out): fixed cost against per-row cost."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
K = 256
for N in (768, 256):
    for M in (2048, 5440, 10880, 21760, 25611, 43520, 87040, 174080):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        run = lambda: call("fs2_gemm_regw", ptr(x), ptr(w), ptr(b), ptr(out), M, N, K, K, N, 0, 1, stream())
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        nbytes = M * K * 2 + N * K * 2 + M * N * 2
        print("N=%d M=%6d: %6.1f us  %5.0f GB/s  %5.0f TFLOP/s" % (N, M, us, nbytes / us * 1e-3, 2.0 * M * N * K / us * 1e-6), flush=True)
