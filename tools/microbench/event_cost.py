"""Host cost of the event fork used per weight-gradient launch (train_engine._wgrad): Event() + record + wait_event on a
second stream, with a small kernel launched on each stream in between (us per iteration, host time only)."""
import time, torch
a = torch.zeros(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
side = torch.cuda.Stream()
def loop(n, fork, raw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a.add_(1.0)
        if fork:
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with torch.cuda.stream(side):
            b.add_(1.0)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6
for _ in range(2):
    print("no fork %.1f us/iter   fork %.1f us/iter" % (loop(2000, False, False), loop(2000, True, False)), flush=True)
# pieces
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(2000): e = torch.cuda.Event()
t1 = time.perf_counter()
for _ in range(2000): ev.record()
t2 = time.perf_counter()
for _ in range(2000): side.wait_event(ev)
t3 = time.perf_counter()
torch.cuda.synchronize()
print("Event() %.2f us, record %.2f us, wait_event %.2f us" % ((t1 - t0) / 2000 * 1e6, (t2 - t1) / 2000 * 1e6, (t3 - t2) / 2000 * 1e6))
