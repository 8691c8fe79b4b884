"""Run-to-run determinism of the pipelined decode loop (config 3, 256 utterances): every repeat must give bit-identical
durations and mel for every utterance.  A difference means a race (stream ordering, LDS reuse), not arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from fastspeech2_amd.run.fastspeech_decode import decode
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
enc = sys.argv[2] if len(sys.argv) > 2 else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kw = dict(synth.BASELINE_OVERRIDES)
model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()})
model.eval().cuda().set_precision(prec, enc)
utts = {"utt%03d" % i: x for i, x in enumerate(synth.synth_utterances(256, 0, sort=False))}
cfg = Namespace(DATASET=Namespace(VAL=Namespace(UTTERANCES=utts)), DECODE=Namespace(BATCH_SIZE=32))
ref = decode(cfg, model, 0, None)
bad = 0
for rep in range(n):
    for r, ws in ((0, 1), (rep % 8, 8)):
        out = decode(cfg, model, 0, None, rank=r, world_size=ws) if ws > 1 else decode(cfg, model, 0, None)
        if ws == 1:
            for k, a in out.items():
                b = ref[k]
                if not np.array_equal(a["durations"], b["durations"]) or a["mel_postnet"].shape != b["mel_postnet"].shape \
                        or not np.array_equal(a["mel_postnet"], b["mel_postnet"]):
                    d = np.abs(a["durations"] - b["durations"]).max()
                    m = np.abs(a["mel_postnet"] - b["mel_postnet"]).max() if a["mel_postnet"].shape == b["mel_postnet"].shape else -1
                    print("rep %d: %s differs: max |d dur| %d, max |d mel| %g" % (rep, k, d, m), flush=True)
                    bad += 1
print("repeats %d, differing utterances %d" % (n, bad))
