"""k = 1 GEMM shapes of the model (QKV, attention output projection, linear FFN of the shipped YAML) timed alone through
fs2_conv_gemm; use with FS2_CONV_MT / FS2_CONV_LOWTILES to compare tile heights.
    python tools/lin_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for (M, N, K, f32) in [(25603, 768, 256, 0), (25603, 256, 256, 1), (27057, 1536, 384, 0), (27057, 384, 1536, 1), (27057, 1152, 384, 0), (3767, 768, 256, 0)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    run = lambda: call("fs2_conv_gemm", ptr(x), ptr(w), ptr(b), None, ptr(out), None, M, N, K, 1, K, N, N, 0, f32, 1, 1, M * N, stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("M=%d N=%d K=%d f32out=%d: %.1f us %.0f TF" % (M, N, K, f32, us, 2.0 * M * N * K / us * 1e-6), flush=True)
