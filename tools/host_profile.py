"""cProfile of the host side of one B=1 forward (config 1): where the launch overhead goes."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
model, _, _ = bench.build_model("bf16", None, torch.device("cuda", 0))
eng = model.engine()
x = np.random.Generator(np.random.PCG64(0)).integers(1, 80, size=120)
ids = torch.from_numpy(x).cuda()
for _ in range(5):
    eng.forward(ids, [120])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    eng.forward(ids, [120])
torch.cuda.synchronize()
print("ms per forward", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    eng.forward(ids, [120])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
