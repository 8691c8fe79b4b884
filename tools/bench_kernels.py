"""Micro-benchmarks of the hot kernels at BASELINE config-2 shapes (one process, interleaved rounds,
random data — cdna guide §5.4 rules 24/25).  Prints TFLOP/s (or GB/s) per shape."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream

DEV = "cuda:0"
ROUNDS = int(os.environ.get("ROUNDS", "5"))
INNER = int(os.environ.get("INNER", "5"))


def seg_tables(lens):
    cu = np.zeros(len(lens) + 1, np.int32)
    cu[1:] = np.cumsum(lens)
    M = int(cu[-1])
    cu_d = torch.from_numpy(cu).to(DEV)
    seg = torch.empty((M, 2), dtype=torch.int32, device=DEV)
    call("fs2_build_seg", ptr(cu_d), len(lens), ptr(seg), None, M, stream())
    return cu_d, seg, M


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(ROUNDS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(INNER):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / INNER * 1e3)
    return float(np.median(best)), float(np.min(best))


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    rng = np.random.Generator(np.random.PCG64(0))
    flens = np.clip(np.rint(rng.normal(800, 100, 32)), 400, 1200).astype(int).tolist()
    plens = np.clip(np.rint(rng.normal(120, 15, 32)), 60, 180).astype(int).tolist()
    shapes = [  # name, lens, taps, Cin, N, relu, res, out_f32
        ("dec ffn conv1 k9 256->1024", flens, 9, 256, 1024, True, False, False),
        ("dec ffn conv2 k9 1024->256", flens, 9, 1024, 256, False, True, True),
        ("dec qkv 256->768", flens, 1, 256, 768, False, False, False),
        ("dec out 256->256 +res", flens, 1, 256, 256, False, True, True),
        ("var conv k3 256->384", flens, 3, 256, 384, True, False, True),
        ("var conv k3 384->384", flens, 3, 384, 384, True, False, True),
        ("feat_out 256->80", flens, 1, 256, 80, False, False, True),
        ("enc ffn conv1 k9 256->1024", plens, 9, 256, 1024, True, False, False),
        ("enc ffn conv2 k9 1024->256", plens, 9, 1024, 256, False, True, True),
        ("enc qkv 256->768", plens, 1, 256, 768, False, False, False),
    ]
    only = os.environ.get("ONLY")
    for name, lens, taps, cin, n, relu, use_res, out_f32 in shapes:
        if only and only not in name:
            continue
        cu_d, seg, M = seg_tables(lens)
        x = torch.randn(M, cin, device=DEV).to(tdt)
        w = (torch.randn(taps, n, cin, device=DEV) / math.sqrt(cin * taps)).to(tdt)
        b = torch.randn(n, device=DEV)
        res = torch.randn(M, n, device=DEV) if use_res else None
        out = torch.empty((M, n), device=DEV, dtype=torch.float32 if out_f32 else tdt)

        def fn():
            call("fs2_conv_gemm", ptr(x), ptr(w), ptr(b), ptr(res), ptr(out), ptr(seg) if taps > 1 else None, M, n, cin,
                 taps, cin, n, n, int(relu), int(out_f32), code, 1, 0, stream())
        med, mn = timeit(fn)
        fl = 2.0 * M * n * cin * taps
        print("%-30s M=%6d  %8.1f us (min %8.1f)  %7.1f TFLOP/s" % (name, M, med, mn, fl / med / 1e6), flush=True)

    # attention (decoder / encoder)
    for name, lens in (("dec attention h2 dk128", flens), ("enc attention h2 dk128", plens)):
        if only and only not in name:
            continue
        cu_d, seg, M = seg_tables(lens)
        qkv = torch.randn(M, 768, device=DEV).to(tdt)
        ctx = torch.empty((M, 256), device=DEV, dtype=tdt)

        def fn():
            call("fs2_attention", ptr(qkv), ptr(ctx), ptr(cu_d), len(lens), max(lens), 2, 128, 768, 256, code, stream())
        med, mn = timeit(fn)
        fl = sum(4.0 * L * L * 256 for L in lens)
        print("%-30s M=%6d  %8.1f us (min %8.1f)  %7.1f TFLOP/s" % (name, M, med, mn, fl / med / 1e6), flush=True)

    # layernorm (HBM-bound)
    cu_d, seg, M = seg_tables(flens)
    x = torch.randn(M, 256, device=DEV)
    wv, bv = torch.randn(256, device=DEV), torch.randn(256, device=DEV)
    o32 = torch.empty_like(x)
    ot = torch.empty((M, 256), device=DEV, dtype=tdt)

    def fn():
        call("fs2_layernorm", ptr(x), ptr(wv), ptr(bv), ptr(o32), ptr(ot) if code else None, M, 256, 1e-12, code, stream())
    med, mn = timeit(fn)
    byt = M * 256 * (4 + 4 + (2 if code else 0))
    print("%-30s M=%6d  %8.1f us (min %8.1f)  %7.1f GB/s" % ("layernorm 256 (f32+T out)", M, med, mn, byt / med / 1e3), flush=True)


if __name__ == "__main__":
    main()
