"""Second FFN conv (k=9, 1024 -> 256, bf16) timed alone for every split-K factor over a sweep of token counts: the data
the cost model of fs2_conv_gemm_ksplit (csrc/conv_gemm.hip) is fitted to.  `rule` is what the library picks.
    python tools/ks_sweep.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
model, _, _ = bench.build_model("bf16", None, dev)
eng = model.engine()
from fastspeech2_amd.hip import call, ptr, stream
L, st, d = eng.dec.layers[0], eng.dec, eng.d
Ms = [int(a) for a in sys.argv[1:]] or [3072, 6144, 9216, 12288, 18432, 25603, 40000]
for M in Ms:
    cu = torch.tensor([0, M], dtype=torch.int32, device=dev)
    seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
    call("fs2_build_seg", ptr(cu), 1, ptr(seg), None, M, stream())
    h = torch.randn(M, L["units"], device=dev).to(torch.bfloat16)
    line = "M=%6d rule=%d |" % (M, eng._ksplit(M, d, L["units"], L["taps"], st.code))
    for ks in (1, 2, 3, 4, 6, 8):
        run = lambda: eng._gemm(h, L["w2"], L["b2"], M, d, L["units"], L["taps"], st.code, seg=seg, out_f32=True, ksplit=ks)
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        line += " ks%d %6.1f" % (ks, e0.elapsed_time(e1) / 30 * 1e3)
    print(line, flush=True)
