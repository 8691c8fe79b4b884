"""Micro-benchmark of the HBM-bound frame-level kernels at BASELINE config-2 shapes (one process, interleaved
rounds, random data): lr_gather, variance_add_posenc, layernorm_rowdot, add_layernorm.  GB/s = algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastspeech2_amd import hip
from fastspeech2_amd.engine import posenc_table
from fastspeech2_amd.hip import call, ptr, stream

DEV = "cuda:0"
ROUNDS, INNER = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("INNER", "10"))
NBUF = 6           # rotate over copies of the big operands so that a launch does not find its input in L2


def timeit(fn):
    for k in range(NBUF):
        fn(k)
    torch.cuda.synchronize()
    t = []
    for _ in range(ROUNDS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(INNER):
            fn(k % NBUF)
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / INNER * 1e3)
    return float(np.median(t)), float(np.min(t))


def main():
    rng = np.random.Generator(np.random.PCG64(0))
    B, d = 32, 256
    plens = np.clip(np.rint(rng.normal(120, 15, B)), 60, 180).astype(np.int64)
    durs = [rng.integers(3, 11, size=int(L)).astype(np.int32) for L in plens]
    Mp = int(plens.sum())
    cu_p = torch.from_numpy(np.concatenate([[0], np.cumsum(plens)]).astype(np.int32)).to(DEV)
    dur = torch.from_numpy(np.concatenate(durs)).to(DEV)
    dur_used, cum = torch.empty_like(dur), torch.empty_like(dur)
    olens, cu_f = torch.empty(B, dtype=torch.int32, device=DEV), torch.empty(B + 1, dtype=torch.int32, device=DEV)
    call("fs2_lr_plan", ptr(dur), ptr(cu_p), 1.0, ptr(dur_used), ptr(cum), ptr(olens), ptr(cu_f), B, stream())
    Mf = int(cu_f[-1])
    hs = [torch.randn(Mp, d, device=DEV) for _ in range(NBUF)]
    o32 = [torch.empty(Mf, d, device=DEV) for _ in range(NBUF)]
    ot = [torch.empty(Mf, d, device=DEV, dtype=torch.bfloat16) for _ in range(NBUF)]
    seg, pos, lri = (torch.empty((Mf, 2), dtype=torch.int32, device=DEV), torch.empty(Mf, dtype=torch.int32, device=DEV),
                     torch.empty(Mf, dtype=torch.int32, device=DEV))

    def gather(k):
        call("fs2_lr_gather", ptr(hs[k]), ptr(cum), ptr(cu_p), ptr(cu_f), B, ptr(o32[k]), ptr(ot[k]), ptr(seg), ptr(pos),
             ptr(lri), Mf, d, hip.BF16, stream())
    med, mn = timeit(gather)
    nb = Mp * (4 * d + 12) + Mf * (6 * d + 16)
    print("lr_gather            Mf=%d  %6.1f us (min %6.1f)  %6.0f GB/s" % (Mf, med, mn, nb / med / 1e3), flush=True)

    pe = posenc_table(2048, d).to(DEV)
    pitch, energy = torch.randn(Mf, 4, device=DEV), torch.randn(Mf, 1, device=DEV)
    Wp, bp, We, be = (torch.randn(d, 4, device=DEV), torch.randn(d, device=DEV), torch.randn(d, 1, device=DEV),
                      torch.randn(d, device=DEV))
    x32 = [torch.randn(Mf, d, device=DEV) for _ in range(NBUF)]

    def vap(k):
        call("fs2_variance_add_posenc", ptr(x32[k]), ptr(pitch), ptr(energy), ptr(Wp), ptr(bp), ptr(We), ptr(be), 1.0, 1.0,
             ptr(pe), ptr(pos), 1.0, 1.0, None, ptr(o32[k]), ptr(ot[k]), Mf, d, 4, 1, hip.BF16, stream())
    med, mn = timeit(vap)
    nb = Mf * (4 * d + 20 + 4 + 4 * d + 2 * d)
    print("variance_add_posenc  Mf=%d  %6.1f us (min %6.1f)  %6.0f GB/s" % (Mf, med, mn, nb / med / 1e3), flush=True)

    C = 384
    y = [torch.randn(Mf, C, device=DEV) for _ in range(NBUF)]
    lw, lb = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    for nout in (4, 1):
        W, hb = torch.randn(nout, C, device=DEV), torch.randn(nout, device=DEV)
        out = torch.empty(Mf, nout, device=DEV)

        def lrd(k):
            call("fs2_layernorm_rowdot", ptr(y[k]), ptr(lw), ptr(lb), ptr(W), ptr(hb), ptr(out), Mf, C, nout, 1e-12, stream())
        med, mn = timeit(lrd)
        nb = Mf * (4 * C + 4 * nout)
        print("layernorm_rowdot n=%d Mf=%d  %6.1f us (min %6.1f)  %6.0f GB/s" % (nout, Mf, med, mn, nb / med / 1e3), flush=True)

    parts = [torch.randn(3, Mf, d, device=DEV) for _ in range(NBUF)]
    w, b = torch.randn(d, device=DEV), torch.randn(d, device=DEV)
    for nparts in (1, 3):
        def aln(k):
            call("fs2_add_layernorm", ptr(x32[k]), ptr(parts[k]), nparts, Mf * d, ptr(w), ptr(b), None, ptr(o32[k]), ptr(ot[k]),
                 Mf, d, 1e-12, hip.BF16, stream())
        med, mn = timeit(aln)
        nb = Mf * d * (4 * (1 + nparts) + 6)
        print("add_layernorm parts=%d Mf=%d  %6.1f us (min %6.1f)  %6.0f GB/s" % (nparts, Mf, med, mn, nb / med / 1e3), flush=True)


if __name__ == "__main__":
    main()
