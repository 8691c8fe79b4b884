"""How much of a throughput-loop step is host work?  Runs bench.py's loop and subtracts the time the host spends blocked
in the one synchronisation per batch (Event.synchronize on the phone-level half's frame counts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
model, _, _ = bench.build_model("bf16", sys.argv[1] if len(sys.argv) > 1 else "f16x3", "cuda:0")
eng = model.engine()
b = bench.rank_batch(1, 0)
feed = (torch.from_numpy(np.concatenate(b)).cuda(), [len(x) for x in b])
blocked = [0.0]
orig = torch.cuda.Event.synchronize
def sync(self):
    t = time.perf_counter(); orig(self); blocked[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = sync
def run(n):
    for o in eng.forward_pipelined((feed for _ in range(n)), frame_streams=3, phone_group=2):
        pass
run(10); torch.cuda.synchronize()
for _ in range(3):
    blocked[0] = 0.0
    t0 = time.perf_counter(); run(40); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("per step: wall %.3f ms, host enqueue %.3f ms (blocked in the sync %.3f ms), drain after the loop %.3f ms" %
          ((t2 - t0) / 40 * 1e3, (t1 - t0 - blocked[0]) / 40 * 1e3, blocked[0] / 40 * 1e3, (t2 - t1) * 1e3), flush=True)
