import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
N = K = 256
for pieces in (1, 2, 4, 6, 7, 8, 16, 32):
    M = 256 * 16 * pieces
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    b, gam, bet = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev)
    o32 = torch.empty(M, N, device=dev); ot = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: call("fs2_proj_add_layernorm", ptr(x), ptr(w), ptr(b), ptr(res), ptr(gam), ptr(bet), ptr(o32), ptr(ot), M, N, K, K, N, N, 1e-12, 1, stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("pieces per WG %2d  M=%6d: %.1f us  %.0f GB/s" % (pieces, M, us, M * 3072 / us * 1e-3), flush=True)
