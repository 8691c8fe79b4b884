"""The variance predictors' frame-level convs (k = 3, 256 -> 384 and 384 -> 384, ReLU, fp32 out) alone through
fs2_conv_gemm; use with FS2_CONV_MT=2|3|4 to compare tile heights.    python tools/pred_conv_sweep.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25603
lens = [M // 32] * 31 + [M - 31 * (M // 32)]
cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
seg = torch.empty((M, 2), dtype=torch.int32, device=dev); pos = torch.empty(M, dtype=torch.int32, device=dev)
call("fs2_build_seg", ptr(cu), 32, ptr(seg), ptr(pos), M, stream())
for (N, K, taps) in [(384, 256, 3), (384, 384, 3), (256, 256, 3)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    run = lambda: call("fs2_conv_gemm", ptr(x), ptr(w), ptr(b), None, ptr(out), ptr(seg), M, N, K, taps, K, N, N, 1, 1, 1, 1, M * N, stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("M=%d N=%d K=%d taps=%d: %.1f us %.0f TF" % (M, N, K, taps, us, 2.0 * M * N * K * taps / us * 1e-6), flush=True)
