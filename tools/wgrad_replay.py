"""Replays the decoder-level FFN weight-gradient launches alone (for rocprofv3 --pmc passes and timing):
dY [M, 1024] x X [M, 256], 9 taps, and the transposed pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
lens = [int(v) for v in np.random.Generator(np.random.PCG64(0)).integers(600, 1000, size=32)]
M = sum(lens)
cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
call("fs2_build_seg", ptr(cu), len(lens), ptr(seg), None, M, stream())
for N, C in ((1024, 256), (256, 1024)):
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    grad = torch.empty(N, C, 9, device=dev)
    dbias = torch.empty(N, device=dev)
    ks = 5
    ws = torch.empty(hip.load().fs2_conv_wgrad_workspace(N, C, 9, ks) // 4 + 4, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            call("fs2_conv_wgrad", ptr(dy), ptr(x), ptr(grad), ptr(dbias), ptr(ws), ptr(seg), M, N, C, 9, N, C, 1, ks, 1.0, 0, stream())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("M=%d N=%d C=%d: %.1f us per launch (kernel + reduce), %.0f TFLOP/s" % (M, N, C, dt * 1e6, 2.0 * M * N * C * 9 / dt / 1e12))
