"""Old (fs2_conv_gemm) against new (fs2_conv_gemm256) FFN-conv kernels over a sweep of token counts: us per launch, best of
interleaved rounds, both decoder conv shapes.  python tools/conv256_sweep.py [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream


def main():
    dev = torch.device("cuda", 0)
    lib = hip.load()
    Ms = [int(a) for a in sys.argv[1:]] or list(range(12288, 53248, 2048)) + [25611]
    d, U, taps = 256, 1024, 9
    g = torch.Generator(device="cpu").manual_seed(1)
    for M in sorted(Ms):
        cu = torch.tensor(np.arange(0, M + 799, 800).clip(0, M).astype(np.int32), device=dev)
        seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
        call("fs2_build_seg", ptr(cu), cu.numel() - 1, ptr(seg), None, M, stream())
        line = "M=%6d (%5.1f units per workgroup of conv1)" % (M, (M + 15) // 16 / 64.0)
        for (Cin, N, relu, split) in ((d, U, 1, 0), (U, d, 0, 1)):
            x = torch.randn(M, Cin, generator=g).to(dev).to(torch.bfloat16)
            if split:
                x = torch.relu(x)
            w = (torch.randn(taps, N, Cin, generator=g) * (1.0 / np.sqrt(Cin * taps))).to(dev).to(torch.bfloat16)
            bias = torch.randn(N, generator=g).to(dev)
            res = torch.randn(M, N, generator=g).to(dev) if split else None
            wt = torch.empty_like(w)
            call("fs2_pack_weight_tiled", ptr(w), ptr(wt), N, Cin, taps, 1, stream())
            ks_new = max(1, lib.fs2_conv_gemm256_plan(1 << 20, N, Cin, taps, 1, 0))
            if split:
                ks_new = 2
            ks_old = lib.fs2_conv_gemm_ksplit(M, N, Cin, taps, 1) if split else 1
            dt = torch.float32 if split else torch.bfloat16
            o_old = torch.zeros((ks_old, M, N), dtype=dt, device=dev)
            o_new = torch.zeros((ks_new, M, N), dtype=dt, device=dev)
            fo = lambda: call("fs2_conv_gemm", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(o_old), ptr(seg), M, N, Cin, taps, Cin,
                              N, N, relu, split, 1, ks_old, M * N, stream())
            fn = lambda: call("fs2_conv_gemm256", ptr(x), ptr(wt), ptr(bias), ptr(res), ptr(o_new), ptr(seg), M, N, Cin, taps,
                              Cin, N, N, relu, split, 1, ks_new, M * N, stream())
            best = {}
            for rnd in range(3):
                for name, f in (("old", fo), ("new", fn)):
                    for _ in range(3):
                        f()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 10 * 1e3)
            line += " | %d->%d old(ks %d) %6.1f new(ks %d) %6.1f us  new/old %.3f" % (Cin, N, ks_old, best["old"], ks_new, best["new"],
                                                                                 best["new"] / best["old"])
        print(line, flush=True)


if __name__ == "__main__":
    main()
