"""The k = 1 GEMM shapes of the frame-level half timed alone (bf16): QKV, output projection, feat_out and the k = 3
predictor convs, through fs2_conv_gemm (the KT=0 symbol).  K1_SHAPE=name runs only that shape (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
M = int(os.environ.get("K1_M", "25611"))
SHAPES = {"qkv": (768, 256, 1, 0, 0, 1), "out_proj": (256, 256, 1, 1, 1, 1), "feat_out": (80, 256, 1, 1, 0, 1),
          "pred1": (384, 256, 3, 1, 0, 1), "pred2": (384, 384, 3, 1, 0, 1)}
only = os.environ.get("K1_SHAPE")
cu = torch.tensor(np.arange(0, M + 799, 800).clip(0, M).astype(np.int32), device=dev)
seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
call("fs2_build_seg", ptr(cu), cu.numel() - 1, ptr(seg), None, M, stream())
for name, (N, K, taps, f32, has_res, ks) in SHAPES.items():
    if only and name != only:
        continue
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev) if has_res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    run = lambda: call("fs2_conv_gemm", ptr(x), ptr(w), ptr(b), ptr(res), ptr(out), ptr(seg) if taps > 1 else None, M, N, K, taps,
                       K, N, N, 1 if taps > 1 else 0, f32, 1, ks, M * N, stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 12 if only else 50
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    nbytes = M * K * 2 + taps * N * K * 2 + M * N * (4 if f32 else 2) + (M * N * 4 if has_res else 0)
    print("%-9s M=%d N=%d K=%d x %d taps: %.1f us  %.0f TFLOP/s  %.0f GB/s of algorithmic bytes (%.1f MB)"
          % (name, M, N, K, taps, us, 2.0 * M * N * K * taps / us * 1e-6, nbytes / us * 1e-3, nbytes * 1e-6), flush=True)
if not only or only == "qkv_regw":
    N, K = 768, 256
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: call("fs2_gemm_regw", ptr(x), ptr(w), ptr(b), ptr(out), M, N, K, K, N, 0, 1, stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 12 if only else 50
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    nbytes = M * K * 2 + N * K * 2 + M * N * 2
    print("qkv_regw  M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s  %.0f GB/s of algorithmic bytes" % (M, N, K, us, 2.0 * M * N * K / us * 1e-6, nbytes / us * 1e-3), flush=True)
if not only or only == "proj_ln":
    # output projection + residual + norm1 as one kernel (csrc/proj_ln.hip) against the two launches it replaces
    N = K = 256
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    b, gam, bet = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev)
    part, o32 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ot = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def fused():
        call("fs2_proj_add_layernorm", ptr(x), ptr(w), ptr(b), ptr(res), ptr(gam), ptr(bet), ptr(o32), ptr(ot), M, N, K, K, N, N,
             1e-12, 1, stream())
    def pair():
        call("fs2_conv_gemm", ptr(x), ptr(w), ptr(b), ptr(res), ptr(part), None, M, N, K, 1, K, N, N, 0, 1, 1, 1, 0, stream())
        call("fs2_add_layernorm", None, ptr(part), 1, M * N, ptr(gam), ptr(bet), None, ptr(o32), ptr(ot), M, N, 1e-12, 1, stream())
    nbytes = M * N * (2 + 4 + 4 + 2) + N * K * 2
    for name, run in (("pair", pair), ("fused", fused), ("pair", pair), ("fused", fused)):
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        n = 12 if only else 50
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print("out_proj + residual + norm1, %-5s M=%d: %.1f us  %.0f GB/s of the fused kernel's algorithmic bytes (%.1f MB)"
              % (name, M, us, nbytes / us * 1e-3, nbytes * 1e-6), flush=True)
