"""Batch-1 latency (BASELINE config 1: one utterance of 120 phones -> ~810 frames, BASELINE dims): model.inference()
issued launch by launch from Python against the same call replayed from HIP graphs (enable_inference_graphs).
    python tools/bench_latency.py [engine_switch=0|1 ...]        e.g. latency_split=0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastspeech2_amd import synth

def timeit(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

import bench
model, _, _ = bench.build_model("fp32", None, "cuda:0")
ids = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).integers(1, synth.BENCH_IDIM, size=120))
for prec in (("bf16", "f16x3"), ("bf16", "bf16"), ("f16x3", None), ("fp32", None)):
    model.set_precision(prec[0], prec[1])
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        assert hasattr(model.engine(), k), k
        setattr(model.engine(), k, bool(int(v)))
    with torch.no_grad():
        model.enable_inference_graphs(0)
        mel = model.inference(ids)
        eager = timeit(lambda: model.inference(ids))
        model.enable_inference_graphs(16)
        graphed = timeit(lambda: model.inference(ids))
        same = torch.equal(mel, model.inference(ids))
    print("%-14s frames %4d  eager %.3f ms  graphs %.3f ms  identical %s" % ("/".join(str(p) for p in prec), mel.shape[0], eager, graphed, same), flush=True)
