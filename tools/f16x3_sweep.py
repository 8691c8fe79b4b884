"""The phone-level f16x3 GEMM shapes alone (config-2 batch: 3767 phones), per tile shape (FS2_SPLIT_TILE is read once per
process: run once per value) and split-K factor."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import hip
from fastspeech2_amd.engine import pack_gemm_weight
from fastspeech2_amd.hip import call, ptr, stream
from tools.bench_kernels import seg_tables, timeit

DEV = "cuda:0"
rng = np.random.Generator(np.random.PCG64(0))
scale_m = int(os.environ.get("MSCALE", "1"))
plens = np.clip(np.rint(rng.normal(120, 15, 32 * scale_m)), 60, 180).astype(int).tolist()
shapes = [("ffn1 k9 256->1024", 9, 256, 1024, 1, [1]), ("ffn2 k9 1024->256", 9, 1024, 256, 0, [1, 2, 3, 4, 6, 8]),
          ("qkv 256->768", 1, 256, 768, 0, [1]), ("out 256->256", 1, 256, 256, 0, [1, 2]), ("dur k3 256->256", 3, 256, 256, 1, [1])]
cu_d, seg, M = seg_tables(plens)
print("FS2_SPLIT_TILE=%s M=%d" % (os.environ.get("FS2_SPLIT_TILE"), M))
for name, taps, cin, n, relu, kss in shapes:
    x32 = torch.randn(M, cin, device=DEV)
    x = torch.empty_like(x32)
    call("fs2_cast", ptr(x32), ptr(x), M * cin, hip.F16X3, stream())
    w = pack_gemm_weight(torch.randn(n, cin, taps, device=DEV) / math.sqrt(cin * taps), hip.F16X3)
    b = torch.randn(n, device=DEV)
    for ks in kss:
        out = torch.empty((ks, M, n), device=DEV)
        def fn():
            call("fs2_conv_gemm_f16x3", ptr(x), ptr(w), w.fs2_scale, ptr(b), None, ptr(out), ptr(seg) if taps > 1 else None,
                 M, n, cin, taps, cin, n, n, relu if ks == 1 else 0, 1, ks, M * n, stream())
        med, mn = timeit(fn)
        fl = 2.0 * M * n * cin * taps
        print("  %-20s ks=%d  %7.1f us (min %7.1f)  %6.1f TFLOP/s algorithmic (x3 MFMA)" % (name, ks, med, mn, fl / med / 1e6), flush=True)
print("  model rule ks(ffn2) =", hip.load().fs2_conv_gemm_ksplit(M, 256, 1024, 9, 0), " ks(out) =", hip.load().fs2_conv_gemm_ksplit(M, 256, 256, 1, 0))
