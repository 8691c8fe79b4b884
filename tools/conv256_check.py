"""fs2_conv_gemm256 (256 x 256 persistent ping-pong kernel) against fs2_conv_gemm on the decoder FFN conv shapes:
results (ragged utterances, so edge tiles are exercised) and time per launch, interleaved A/B in one process.
    python tools/conv256_check.py [M ...]"""
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
    Ms = [int(a) for a in sys.argv[1:]] or [25611]
    d, U, taps = 256, 1024, int(os.environ.get("C256_TAPS", "9"))
    g = torch.Generator(device="cpu").manual_seed(1)
    for M in Ms:
        # ragged utterances of 500..1100 frames
        lens, tot = [], 0
        rng = np.random.default_rng(M)
        while tot < M:
            n = int(min(rng.integers(500, 1100), M - tot))
            lens.append(n)
            tot += n
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
        seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
        call("fs2_build_seg", ptr(cu), len(lens), ptr(seg), None, M, stream())
        for (Cin, N, relu, split) in ((d, U, 1, 0), (U, d, 0, 1)):
            x = torch.randn(M, Cin, generator=g).to(dev).to(torch.bfloat16)
            w = (torch.randn(taps, N, Cin, generator=g) * (1.0 / np.sqrt(Cin * taps))).to(dev).to(torch.bfloat16)
            if os.environ.get("C256_ZERO"):          # all-zero operands: same instruction stream, far less switching power
                x.zero_(); w.zero_()
            bias = torch.randn(N, generator=g).to(dev)
            res = torch.randn(M, N, generator=g).to(dev) if split else None
            wt = torch.empty_like(w)
            call("fs2_pack_weight_tiled", ptr(w), ptr(wt), N, Cin, taps, 1, stream())
            ks_new = lib.fs2_conv_gemm256_plan(M, N, Cin, taps, 1, split)
            ks_old = lib.fs2_conv_gemm_ksplit(M, N, Cin, taps, 1) if split else 1
            if ks_new < 1:      # the plan prefers fs2_conv_gemm for this M: measure the new kernel all the same
                ks_new = 2 if split else 1
            out_f32 = 1 if split else 0
            dt = torch.float32 if out_f32 else torch.bfloat16
            o_old = torch.zeros((ks_old, M, N), dtype=dt, device=dev)
            o_new = torch.zeros((ks_new, M, N), dtype=dt, device=dev)

            def run_old():
                call("fs2_conv_gemm", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(o_old), ptr(seg), M, N, Cin, taps, Cin, N, N,
                     relu, out_f32, 1, ks_old, M * N, stream())

            def run_new():
                call("fs2_conv_gemm256", ptr(x), ptr(wt), ptr(bias), ptr(res), ptr(o_new), ptr(seg), M, N, Cin, taps, Cin, N,
                     N, relu, out_f32, 1, ks_new, M * N, stream())

            only = os.environ.get("C256_ONLY")
            if only == "new":          # profiling / ablation builds: the new kernel alone, no comparison
                for _ in range(12):
                    run_new()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run_new()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 20 * 1e3
                print("M=%d %d->%d new only: %.1f us = %.0f TFLOP/s" % (M, Cin, N, t, 2.0 * M * N * Cin * taps / t * 1e-6), flush=True)
                if os.environ.get("C256_STAMPS"):      # B2_DBG_STAMP build: s_memtime stamps of one chunk, waves 0 and 4 of block 0
                    st = o_new.view(-1).view(torch.int32)[:2048].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                    names = ["reads issued", "vmcnt", "lgkmcnt+mask", "B1", "M end", "B2+tail"]
                    order = [4, 5, 0, 1, 2, 3]
                    for w in (0, 1):
                        v = st[w * 1024: w * 1024 + 2 * taps * 6]
                        print(" wave %d: half-step length %s" % (4 * w, " ".join(str(int(v[(h + 1) * 6 + 4] - v[h * 6 + 4])) for h in range(2 * taps - 1))))
                        for h in range(1, 2 * taps, 3):
                            prev_end = v[(h - 1) * 6 + 3]
                            sg = [v[h * 6 + k] for k in order]
                            t, out = prev_end, []
                            for k, n in enumerate(names):
                                out.append("%s+%d" % (n, int(sg[k] - t)))
                                t = sg[k]
                            print("   hs %2d: " % h + " ".join(out))
                        gs = st[w * 1024 + 900: w * 1024 + 932]
                        real = ((gs[31] - gs[30]) & 0xFFFFFFFF) * 10.0       # s_memrealtime: 100 MHz
                        print("   block 0 from first to last stamp: %d ticks in %.1f us -> %.2f GHz" % (gs[20] - gs[0], real * 1e-3, (gs[20] - gs[0]) / real))
                        print("   kernel: prologue %d, tile0 setup..mainloop end %d (setup stamp %d), epilogue+setup of tile 1 %d, tile1 mainloop %d, "
                              "last epilogue %d" % (gs[1] - gs[0], gs[2] - gs[1], gs[1] - gs[1], gs[4] - gs[2], gs[5] - gs[4], gs[20] - gs[5]))
                continue
            run_old()
            run_new()
            torch.cuda.synchronize()
            a, b = o_old.float().sum(0), o_new.float().sum(0)
            err = (a - b).abs().max().item()
            scale = a.abs().max().item()
            bad = ((a - b).abs() > 0.02 * scale).sum().item()
            print("M=%d %d->%d  ks old %d / new %d: max |old - new| = %.4g (max |out| %.3g), elements off by > 2 %% of max: %d"
                  % (M, Cin, N, ks_old, ks_new, err, scale, bad), flush=True)
            times = {"old": [], "new": []}
            for rnd in range(4):
                for name, fn in (("old", run_old), ("new", run_new)):
                    for _ in range(3):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
            fl = 2.0 * M * N * Cin * taps
            for name in ("old", "new"):
                t = min(times[name])
                print("   %s: %s us -> best %.1f us = %.0f TFLOP/s (%.3f of 2.5 PF)"
                      % (name, " ".join("%.1f" % v for v in times[name]), t, fl / t * 1e-6, fl / t * 1e-6 / 2500.0), flush=True)


if __name__ == "__main__":
    main()
