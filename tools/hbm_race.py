"""The inference-side HBM kernels with many loads in flight (fs2_layernorm_rowdot, fs2_lr_gather, fs2_add_layernorm,
fs2_variance_add_posenc) on one stream while fs2_conv_wgrad / fs2_conv_gemm (LDS-DMA kernels) keep the chip busy on
another: repeated launches on the same inputs must be bit-identical.   python tools/hbm_race.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr
dev = "cuda:0"
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M, C = 24659, 384
g = torch.Generator(device=dev).manual_seed(0)
lib = hip.load()
y = torch.randn(M, C, device=dev, generator=g)
lw, lb = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
W4, hb4 = torch.randn(4, C, device=dev, generator=g), torch.randn(4, device=dev, generator=g)
x256 = torch.randn(M, 256, device=dev, generator=g)
parts = torch.randn(3, M, 256, device=dev, generator=g)
w2, b2 = torch.randn(256, device=dev, generator=g), torch.randn(256, device=dev, generator=g)
# aggressors
N, Cin = 256, 256
gy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16); gx = torch.randn(M, Cin, device=dev, generator=g).to(torch.bfloat16)
gw = torch.empty(N, Cin, device=dev); gb = torch.empty(N, device=dev)
wws = torch.empty(lib.fs2_conv_wgrad_workspace(N, Cin, 1, 8) // 4 + 4, device=dev)
cx = torch.randn(M, 256, device=dev, generator=g).to(torch.bfloat16)
cw = (torch.randn(1, 768, 256, device=dev, generator=g) / 16).to(torch.bfloat16)
cb = torch.zeros(768, device=dev); co = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for busy in (1, 2):
    outs = []
    torch.cuda.synchronize()
    for it in range(NL):
        for _ in range(3):
            if busy == 1:
                call("fs2_conv_wgrad", ptr(gy), ptr(gx), ptr(gw), ptr(gb), ptr(wws), None, M, N, Cin, 1, N, Cin, 1, 8, 1.0, 0,
                     hip.c_void_p(sb.cuda_stream))
            else:
                call("fs2_conv_gemm", ptr(cx), ptr(cw), ptr(cb), None, ptr(co), None, M, 768, 256, 1, 256, 768, 768, 0, 0, 1, 1,
                     M * 768, hip.c_void_p(sb.cuda_stream))
        o4 = torch.empty(M, 4, device=dev)
        call("fs2_layernorm_rowdot", ptr(y), ptr(lw), ptr(lb), ptr(W4), ptr(hb4), ptr(o4), M, C, 4, 1e-12, hip.c_void_p(sa.cuda_stream))
        of = torch.empty(M, 256, device=dev); ot = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
        call("fs2_add_layernorm", ptr(x256), ptr(parts), 3, M * 256, ptr(w2), ptr(b2), None, ptr(of), ptr(ot), M, 256, 1e-12, 1,
             hip.c_void_p(sa.cuda_stream))
        outs.append((o4, of, ot))
    torch.cuda.synchronize()
    bad = 0
    for k, cur in enumerate(outs[1:], 1):
        for name, a, b in zip(("layernorm_rowdot", "add_layernorm f32", "add_layernorm bf16"), cur, outs[0]):
            if not torch.equal(a, b):
                d = (a.float() - b.float()).abs()
                print("aggressor %d launch %d: %s differs in %d elements (max %.3g)" % (busy, k, name, int((d > 0).sum()), float(d.max())), flush=True)
                bad += 1
    print("aggressor %s: launches %d, differing outputs %d" % ("conv_wgrad" if busy == 1 else "conv_gemm", NL, bad), flush=True)
    del outs
