"""Latency / throughput of the other BASELINE.json configs (not the bench line): config 1 (B=1, 120
phones), config 4 (long utterance: 512 phones -> ~4000 frames, B=1 and B=4), fp32 and bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fastspeech2_amd import synth

def run(eng, xs, iters=20):
    ilens = [len(x) for x in xs]
    ids = torch.from_numpy(np.concatenate(xs)).cuda()
    for _ in range(3):
        out = eng.forward(ids, ilens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = eng.forward(ids, ilens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    fr = sum(out["olens"])
    return dt * 1e3, fr, fr / dt

for dtype, enc in (("bf16", "f16x3"), ("bf16", "bf16"), ("f16x3", "f16x3"), ("fp32", "fp32")):
    model, _, _ = bench.build_model(dtype, enc, torch.device("cuda", 0))
    eng = model.engine()
    dtype = dtype if enc == dtype else "%s+%s" % (dtype, enc)
    rng = np.random.Generator(np.random.PCG64(0))
    cases = {"config1 B=1 120 phones": [rng.integers(1, 80, size=120)],
             "config4 B=1 512 phones": [rng.integers(1, 80, size=512)],
             "config4 B=4 512 phones": [rng.integers(1, 80, size=512) for _ in range(4)],
             "config2 B=32": synth.synth_utterances(32, 0)}
    for name, xs in cases.items():
        ms, fr, fps = run(eng, xs)
        print("%-10s %-24s %8.3f ms  frames=%6d  %10.0f frames/s  RTF=%.2e" % (dtype, name, ms, fr, fps, ms * 1e-3 / (fr * bench.HOP_SECONDS)), flush=True)

# the shipped param/fastspeech.yaml (6+6 linear-FFN blocks, d=384, 4 heads, PostNet 5x256), random init
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from fastspeech2_amd.param.load_param import load_config
for dtype, enc in (("bf16", "f16x3"), ("bf16", None), ("f16x3", None), ("fp32", None)):
    cfg = load_config()
    sd = synth.synth_state_dict(80, 80, 1234, dur_bias=float(np.log(7.67)), dur_w_scale=0.2, **cfg)
    model = FeedForwardTransformer(80, 80, **cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval().cuda()
    model.set_precision(dtype, enc)
    eng = model.engine()
    dtype = dtype if enc is None else "%s+%s" % (dtype, enc)
    for name, xs in {"yaml B=1 120 phones": [np.random.Generator(np.random.PCG64(0)).integers(1, 80, size=120)],
                     "yaml B=32": synth.synth_utterances(32, 0)}.items():
        ms, fr, fps = run(eng, xs)
        print("%-5s %-24s %8.3f ms  frames=%6d  %10.0f frames/s  RTF=%.2e" % (dtype, name, ms, fr, fps, ms * 1e-3 / (fr * bench.HOP_SECONDS)), flush=True)
