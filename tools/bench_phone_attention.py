"""Phone-level attention launches alone (us per launch): the exact-fp32 MFMA kernel (fp32 q | k | v, dtype 2) against the
split-fp16 kernel (dtype 3) on B utterances of ~120 phones (config 2's phone-level shape, and 4 steps grouped), one
512-phone utterance (config 4) and B = 1 (config 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
dev = torch.device("cuda", 0)
h, dk = 2, 128
d = h * dk
rng = np.random.Generator(np.random.PCG64(0))
for name, lens in (("B=32 x ~120", rng.integers(90, 150, 32).tolist()), ("B=128 x ~120", rng.integers(90, 150, 128).tolist()),
                   ("B=1 x 120", [120]), ("B=1 x 512", [512]), ("B=4 x 512", [512] * 4), ("B=1 x 810", [810]), ("B=1 x 3138", [3138])):
    M = sum(lens)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
    q32 = torch.randn(M, 3 * d, device=dev)
    qs = torch.empty_like(q32)
    call("fs2_cast", ptr(q32), ptr(qs), M * 3 * d, hip.F16X3, stream())
    ctx = torch.empty(M, d, device=dev)
    res = {}
    for rnd in range(3):
        for tag, src, code in (("fp32", q32, hip.F16X3), ("split", qs, hip.F16X3_QKV)):
            run = lambda: call("fs2_attention", ptr(src), ptr(ctx), ptr(cu), len(lens), max(lens), h, dk, 3 * d, d, code, stream())
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res[tag] = min(res.get(tag, 1e9), e0.elapsed_time(e1) / 20 * 1e3)
    extra = ""
    ks = hip.load().fs2_attention_ksplit(len(lens), max(lens), h, dk, hip.F16X3_QKV)
    if ks > 1:          # what the engine launches for this shape: keys over ks workgroups + merge
        ws = torch.empty(hip.load().fs2_attention_keysplit_workspace(M, h, dk, ks) // 4, device=dev)
        run = lambda: call("fs2_attention_keysplit", ptr(qs), ptr(ctx), ptr(cu), len(lens), max(lens), M, h, dk, 3 * d, d,
                           hip.F16X3_QKV, ks, ptr(ws), stream())
        best = 1e9
        for rnd in range(3):
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        extra = ", keys over %d workgroups + merge %6.1f us" % (ks, best)
    print("%-14s M=%6d: exact-fp32 MFMA %6.1f us, split-fp16 %6.1f us%s" % (name, M, res["fp32"], res["split"], extra), flush=True)
