import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
dev = "cuda:0"; M, C = 24388, 256
x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev)
w = torch.randn(C, device=dev)
dw = torch.empty(2 * C, device=dev)
of = torch.empty(M, C, device=dev); ot = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
ws = torch.empty(hip.load().fs2_layernorm_bwd_workspace(M, C) // 4 + 4, device=dev)
for nparts in (0, 1, 3):
    parts = torch.randn(max(nparts, 1), M, C, device=dev)
    for drop in (0.0, 0.1):
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                call("fs2_layernorm_bwd", ptr(x), ptr(dy), ptr(parts) if nparts else None, nparts, M * C, ptr(w), None, 0, ptr(of),
                     ptr(ot), ptr(dw), hip.c_void_p(dw.data_ptr() + 4 * C), ptr(ws), M, C, 1e-12, 0, 1, 7, 1, drop, 2, drop, stream())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        mb = (2 + nparts) * M * C * 4 + M * C * 6
        print("nparts %d dropout %.1f: %.1f us, %.2f TB/s algorithmic" % (nparts, drop, dt * 1e6, mb / dt / 1e12))
