import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench_train
from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
kw = dict(synth.BASELINE_OVERRIDES)
model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}, strict=True)
model.cuda().train(); model.set_precision("bf16")
packed = model.pack_batch(**bench_train.make_batch(1, 0))
for _ in range(3):
    model(packed=packed).backward(); model.zero_grad()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    loss = model(packed=packed)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    model.zero_grad()
    print("host fwd %.2f ms, host bwd %.2f ms, drain %.2f ms, total %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
