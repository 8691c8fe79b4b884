"""CPU checks of the training plumbing: Noam schedule (vs the reference's own class when it is
importable), the update rule of CustomUpdater.update_core, and the bucketed data-parallel gradient
averaging on 2 gloo ranks (the N > 1 training path)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastspeech2_amd.network import optimizer as opt
from fastspeech2_amd.run.train import GradReducer, Updater
from oracle import ref_harness


def test_noam_schedule_formula_and_state_dict():
    lin = torch.nn.Linear(4, 4)
    o = opt.get_std_opt(lin, 256, 4000, 1.0)
    assert o.optimizer.defaults["betas"] == (0.9, 0.98) and o.optimizer.defaults["eps"] == 1e-9
    lin.weight.grad = torch.ones_like(lin.weight)
    lin.bias.grad = torch.ones_like(lin.bias)
    for s in range(1, 6):
        o.step()
        want = 256 ** -0.5 * min(s ** -0.5, s * 4000 ** -1.5)
        assert abs(o.param_group[0]["lr"] - want) <= 1e-15 and o._rate == o.param_group[0]["lr"]
    sd = o.state_dict()
    assert sorted(sd) == sorted(["_step", "warmup", "factor", "model_size", "_rate", "optimizer"])
    o2 = opt.get_std_opt(lin, 1, 1, 1.0)
    o2.load_state_dict(sd)
    assert o2._step == 5 and o2.warmup == 4000 and o2.model_size == 256
    so = opt.ScheduledOptimizer(torch.optim.SGD(lin.parameters(), lr=0.0), 256, 4000, 10)
    so.update_lr()
    assert abs(so.optimizer.param_groups[0]["lr"] - 256 ** -0.5 * min(11 ** -0.5, 4000 ** -1.5 * 11)) <= 1e-15


@pytest.mark.reference
def test_noam_schedule_equals_reference_class():
    if not ref_harness.available():
        pytest.skip("reference not mounted")
    ref_harness.load()
    import importlib
    ref_opt = importlib.import_module("network.optimizer")
    a, b = torch.nn.Linear(3, 3), torch.nn.Linear(3, 3)
    b.load_state_dict(a.state_dict())
    oa, ob = opt.get_std_opt(a, 384, 50, 2.0), ref_opt.get_std_opt(b, 384, 50, 2.0)
    g = torch.Generator().manual_seed(0)
    for _ in range(120):
        gw, gb = torch.randn(3, 3, generator=g), torch.randn(3, generator=g)
        for m in (a, b):
            m.weight.grad, m.bias.grad = gw.clone(), gb.clone()
        oa.step()
        ob.step()
        assert oa._rate == ob._rate
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0]))

    def forward(self, xs, **_):
        return (self.w * xs).sum()


def test_updater_accumulates_clips_and_skips_nan():
    m = _Toy()
    sgd = torch.optim.SGD(m.parameters(), lr=1.0)
    up = Updater(m, grad_clip=1.0, optimizer=sgd, accum_grad=2)
    up.update({"xs": torch.tensor([3.0, 0.0])})
    assert up.iteration == 0 and torch.equal(m.w.data, torch.tensor([1.0, -2.0]))      # first half: no step yet
    up.update({"xs": torch.tensor([3.0, 0.0])})
    # grad = (3/2 + 3/2, 0) = (3, 0) -> clipped to norm 1 -> w -= (1, 0)
    assert up.iteration == 1 and torch.allclose(m.w.data, torch.tensor([0.0, -2.0]), atol=1e-5)
    assert m.w.grad is None or float(m.w.grad.abs().sum()) == 0.0
    up2 = Updater(m, grad_clip=1.0, optimizer=sgd, accum_grad=1)
    before = m.w.data.clone()
    up2.update({"xs": torch.tensor([float("nan"), 1.0])})
    assert torch.equal(m.w.data, before) and up2.iteration == 1                       # NaN norm: update skipped


class _FakeEngine:
    def __init__(self, rank):
        self.flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        self.bucket_slices = [(0, 4), (4, 4), (4, 10)]
        self.grad_ready = None
        self.grad_finish = None


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine(rank)
    GradReducer(eng)
    for i in (0, 1, 2):                  # what TrainEngine.backward does as buckets complete
        eng.grad_ready(i)
    eng.grad_finish()
    q.put((rank, eng.flat.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_averaging():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [1.5 * i for i in range(10)]          # mean of (i, 2 i)
    for _, flat in got:
        assert flat == want
