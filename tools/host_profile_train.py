"""cProfile of the host side of the training step (config 5): where the 7 ms of issue time go."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_train
from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from fastspeech2_amd.network.optimizer import get_std_opt
from fastspeech2_amd.run.train import Updater
kw = dict(synth.BASELINE_OVERRIDES)
model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}, strict=True)
model.cuda().train(); model.set_precision("bf16")
up = Updater(model, 1.0, get_std_opt(model, kw["adim"], 4000, 1.0))
packed = model.pack_batch(**bench_train.make_batch(1, 0))
for _ in range(5):
    up.update({"packed": packed})
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    up.update({"packed": packed})
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
