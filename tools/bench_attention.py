"""fs2_attention alone at the BASELINE decoder shapes (config 2: 32 utterances of ~800 frames; config 4: one / four
utterances of ~4100 frames), bf16 and fp32, random data, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import hip, synth
from fastspeech2_amd.hip import call, ptr, stream
from tools.bench_kernels import timeit

DEV = "cuda:0"
h, dk = 2, 128
d = h * dk
rng = np.random.Generator(np.random.PCG64(0))
_fix = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "durations_bench32.npz")
cases = [("bench.py batch", np.load(_fix)["frames"].astype(int).tolist()),     # the frame counts of bench.py's own batch
         ("config 2 decoder", np.clip(np.rint(rng.normal(800, 100, 32)), 400, 1200).astype(int).tolist()),
         ("config 2 encoder", np.clip(np.rint(rng.normal(120, 15, 32)), 60, 180).astype(int).tolist()),
         ("config 1 B=1", [810]), ("one of 1600", [1600]), ("config 4 B=1 (512 ph)", [3138]),
         ("config 4 B=1", [4165]), ("config 4 B=4", [4165, 4100, 3900, 4000]),
         ("uniform 64 x 1024", [1024] * 64)]      # no imbalance: the tile code's own rate
only = os.environ.get("FS2_BENCH_CASE", "")       # substring: run this case alone (PMC passes: tools/attn_pmc.sh)
if only:
    cases = [c for c in cases if only in c[0]]
order = os.environ.get("FS2_BENCH_SORT", "")      # "desc" / "asc": utterances sorted by length (decode() hands over descending)
if order:
    cases = [(n, sorted(l, reverse=order == "desc")) for n, l in cases]
for dtype in (sys.argv[1:] or ["bf16", "fp32"]):
    code, tdt = hip.DTYPE_CODE[dtype], hip.TORCH_DTYPE[hip.DTYPE_CODE[dtype]]
    for name, lens in cases:
        M = sum(lens)
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=DEV)
        qkv = torch.randn(M, 3 * d, device=DEV).to(tdt)
        ctx = torch.empty(M, d, device=DEV, dtype=tdt)
        def fn():
            call("fs2_attention", ptr(qkv), ptr(ctx), ptr(cu), len(lens), max(lens), h, dk, 3 * d, d, code, stream())
        med, mn = timeit(fn)
        fl = 4.0 * d * sum(float(n) * n for n in lens)
        extra = ""
        ks = hip.load().fs2_attention_ksplit(len(lens), max(lens), h, dk, code)
        forced = int(os.environ.get("FS2_BENCH_KSPLIT", "0"))
        if (ks > 1 or forced > 1) and code == hip.BF16:      # what the engine launches for this shape: keys split + merge
            ks = forced if forced > 1 else ks
            ws = torch.empty(hip.load().fs2_attention_keysplit_workspace(M, h, dk, ks) // 4, device=DEV)
            def fn2():
                call("fs2_attention_keysplit", ptr(qkv), ptr(ctx), ptr(cu), len(lens), max(lens), M, h, dk, 3 * d, d, code, ks,
                     ptr(ws), stream())
            m2, _ = timeit(fn2)
            extra = "   keys over %d workgroups + merge: %8.1f us" % (ks, m2)
        print("%-5s %-18s M=%6d  %8.1f us (min %8.1f)  %7.1f TFLOP/s%s" % (dtype, name, M, med, mn, fl / med / 1e6, extra), flush=True)
