"""Does fs2_conv_wgrad write outside its outputs / workspace?  Every buffer it writes sits inside a larger allocation
filled with a sentinel; after many launches the guard bands must be untouched."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call
dev = "cuda:0"
lib = hip.load()
M = 26976
G = 1 << 20            # guard floats on each side (4 MB)
SENT = 12345.0
for (N, Cin, taps, ks) in [(256, 256, 1, 32), (256, 256, 1, 8), (768, 256, 1, 8), (256, 1024, 9, 5), (1024, 256, 9, 5)]:
    gy = torch.randn(M, N, device=dev).to(torch.bfloat16); gx = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
    nws = lib.fs2_conv_wgrad_workspace(N, Cin, taps, ks) // 4
    bufs = {}
    for name, n in (("grad", N * Cin * taps), ("dbias", N), ("ws", nws)):
        t = torch.full((n + 2 * G,), SENT, device=dev)
        bufs[name] = t
    seg = None
    if taps > 1:
        import numpy as np
        lens = [M // 32] * 31 + [M - 31 * (M // 32)]
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
        seg = torch.empty((M, 2), dtype=torch.int32, device=dev); pos = torch.empty(M, dtype=torch.int32, device=dev)
        call("fs2_build_seg", hip.ptr(cu), 32, hip.ptr(seg), hip.ptr(pos), M, hip.stream())
    p = lambda name: hip.c_void_p(bufs[name].data_ptr() + 4 * G)
    for _ in range(20):
        call("fs2_conv_wgrad", hip.ptr(gy), hip.ptr(gx), p("grad"), p("dbias"), p("ws"), hip.ptr(seg) if seg is not None else None,
             M, N, Cin, taps, N, Cin, 1, ks, 1.0, 0, hip.stream())
    torch.cuda.synchronize()
    for name, t in bufs.items():
        lo, hi = t[:G], t[-G:]
        nlo, nhi = int((lo != SENT).sum()), int((hi != SENT).sum())
        print("N=%d Cin=%d taps=%d ks=%d %-6s: guard floats touched below %d, above %d%s" % (
            N, Cin, taps, ks, name, nlo, nhi,
            "" if not (nlo or nhi) else "  first above at +%d" % int((hi != SENT).nonzero()[0]) if nhi else ""), flush=True)
