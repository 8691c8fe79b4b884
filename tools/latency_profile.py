"""One utterance (default: BASELINE config 1, 120 phones; config 4: 512) run eagerly 60 times, for `rocprofv3 --kernel-trace`
(tools/latency_trace.sh): where a batch-1 forward's kernel time goes.
    python tools/latency_profile.py [encoder_dtype [n_phones]]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastspeech2_amd import synth
import bench
model, _, _ = bench.build_model("bf16", sys.argv[1] if len(sys.argv) > 1 else "f16x3", "cuda:0")
ids = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).integers(1, synth.BENCH_IDIM, size=int(sys.argv[2]) if len(sys.argv) > 2 else 120))
with torch.no_grad():
    for _ in range(60):
        model.inference(ids)
torch.cuda.synchronize()
