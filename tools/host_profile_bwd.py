"""cProfile of the HIP backward schedule called directly (outside autograd's thread, which cProfile cannot see)."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_train
from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
kw = dict(synth.BASELINE_OVERRIDES)
model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}, strict=True)
model.cuda().train(); model.set_precision("bf16")
eng = model.train_engine()
packed = model.pack_batch(**bench_train.make_batch(1, 0))
up = torch.ones(1, device="cuda")
def step():
    outs, saved = eng.forward(*packed)
    eng.backward(saved, up)
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
