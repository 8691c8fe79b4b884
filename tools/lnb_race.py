"""fs2_layernorm_bwd on one stream while fs2_conv_wgrad keeps the chip busy on another: the outputs of repeated launches
on the same inputs must be bit-identical (a kernel-level check, no engine logic involved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr
dev = "cuda:0"; M, C = int(os.environ.get("M", "26976")), 256
busy = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, device=dev, generator=g); dy = torch.randn(M, C, device=dev, generator=g)
parts = torch.randn(2, M, C, device=dev, generator=g)
w = torch.randn(C, device=dev, generator=g)
lib = hip.load()
ws = torch.empty(lib.fs2_layernorm_bwd_workspace(M, C) // 4 + 4, device=dev)
N, Cin = int(os.environ.get("WN", "768")), 256
gy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16); gx = torch.randn(M, Cin, device=dev, generator=g).to(torch.bfloat16)
gw = torch.empty(N, Cin, device=dev); gb = torch.empty(N, device=dev)
wdt = int(os.environ.get("WDT", "1"))            # aggressor operand type: 1 bf16, 0 fp32
nobias = os.environ.get("WNOBIAS", "0") == "1"   # aggressor without the bias-gradient path
gy32, gx32 = (gy.float(), gx.float()) if wdt == 0 else (None, None)
ks = int(os.environ.get("WKS", "8"))
wws = torch.empty(lib.fs2_conv_wgrad_workspace(N, Cin, 1, ks) // 4 + 4, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# the producer of `parts` in a training step: da = dqkv . Wqkv as a split-K GEMM (fp32 partials)
dq = torch.randn(M, 3 * C, device=dev, generator=g).to(torch.bfloat16)
wq = (torch.randn(1, C, 3 * C, device=dev, generator=g) / 16).to(torch.bfloat16)
ksg = int(lib.fs2_conv_gemm_ksplit(M, C, 3 * C, 1, 1))
print("split-K factor of the producer GEMM:", ksg)
ta = torch.randn(4096, 2048, device=dev).to(torch.bfloat16); tb = torch.randn(2048, 4096, device=dev).to(torch.bfloat16)
tc = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
cg_out = torch.empty(M, C, device=dev)
import numpy as np
aq = torch.randn(M, 768, device=dev).to(torch.bfloat16); actx = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
acu = torch.tensor(np.arange(33) * 843, dtype=torch.int32, device=dev)
fixed_parts = os.environ.get("FIXED_PARTS", "0") == "1"
parts0 = torch.randn(1, M, C, device=dev, generator=g)
outs = []
torch.cuda.synchronize()
NL = int(os.environ.get("NL", "200"))
for it in range(NL):
    if busy == 3:          # another LDS-DMA aggressor: the k = 1 conv_gemm (QKV shape)
        for _ in range(3):
            call("fs2_conv_gemm", ptr(dq), ptr(wq), None, None, ptr(cg_out), None, M, C, 3 * C, 1, 3 * C, C, C, 0, 1, 1, 1, M * C,
                 hip.c_void_p(sb.cuda_stream))
    elif busy == 4:        # attention (LDS-DMA + MFMA + transposed reads)
        call("fs2_attention", ptr(aq), ptr(actx), ptr(acu), 32, 843, 2, 128, 768, 256, 1, hip.c_void_p(sb.cuda_stream))
    elif busy == 2:          # another aggressor: plain torch GEMMs on the other stream
        with torch.cuda.stream(sb):
            for _ in range(3):
                torch.mm(ta, tb, out=tc)
    elif busy:
        for _ in range(3):
            call("fs2_conv_wgrad", ptr(gy32 if wdt == 0 else gy), ptr(gx32 if wdt == 0 else gx), ptr(gw),
                 None if nobias else ptr(gb), ptr(wws), None, M, N, Cin, 1, N, Cin, wdt, ks, 1.0, 0, hip.c_void_p(sb.cuda_stream))
    of = torch.empty(M, C, device=dev); ot = torch.empty(M, C, device=dev, dtype=torch.bfloat16); dw = torch.empty(2 * C, device=dev)
    with torch.cuda.stream(sa):
        parts = torch.empty(ksg, M, C, device=dev)
    if fixed_parts:
        parts = parts0
    else:
        call("fs2_conv_gemm", ptr(dq), ptr(wq), None, None, ptr(parts), None, M, C, 3 * C, 1, 3 * C, C, C, 0, 1, 1, ksg, M * C,
             hip.c_void_p(sa.cuda_stream))
    call("fs2_layernorm_bwd", ptr(x), ptr(dy), ptr(parts), ksg, M * C, ptr(w), None, 0, ptr(of), ptr(ot), ptr(dw),
         hip.c_void_p(dw.data_ptr() + 4 * C), ptr(ws), M, C, 1e-12, 0, 1, 7, 1, 0.0, 2, 0.0, hip.c_void_p(sa.cuda_stream))
    outs.append((of, ot, dw, parts))
torch.cuda.synchronize()
bad = 0
for k, (of, ot, dw, pp) in enumerate(outs[1:], 1):
    for name, a, b in (("parts", pp.view(-1, C), outs[0][3].view(-1, C)), ("dx", of, outs[0][0]), ("dx_t", ot, outs[0][1]), ("dw", dw, outs[0][2])):
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            rows = (d.view(d.shape[0], -1).amax(1) > 0).nonzero().flatten() if d.dim() == 2 else None
            print("launch %d: %s differs in %d elements (max %.3g)%s" % (k, name, int((d > 0).sum()), float(d.max()),
                  "" if rows is None else ", rows %s" % rows[:8].tolist()), flush=True)
            bad += 1
print("busy %d: launches %d, differing outputs %d" % (busy, NL, bad))
