"""Host-side issue time of the phases of one training update (no syncs inside), against the GPU-inclusive time."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench_train
from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from fastspeech2_amd.network.optimizer import get_std_opt
kw = dict(synth.BASELINE_OVERRIDES)
model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}, strict=True)
model.cuda().train(); model.set_precision("bf16")
opt = get_std_opt(model, kw["adim"], 4000, 1.0)
packed = model.pack_batch(**bench_train.make_batch(1, 0))
params = list(model.parameters())
def step(acc):
    t = [time.perf_counter()]
    loss = model(packed=packed); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    gn = torch.nn.utils.clip_grad_norm_(params, 1.0); t.append(time.perf_counter())
    base = opt.optimizer
    base.grad_scale, base.found_inf = None, torch.isnan(gn).float().reshape(())
    opt.step(); base.found_inf = None; t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    for i in range(5):
        acc[i] += t[i + 1] - t[i]
for _ in range(5):
    step([0] * 5)
torch.cuda.synchronize()
acc = [0.0] * 5
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step(acc)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host ms/step: fwd %.2f bwd %.2f clip %.2f adam %.2f zero %.2f | host total %.2f, with drain %.2f" % (
    *[a / n * 1e3 for a in acc], (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
