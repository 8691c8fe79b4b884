"""N > 1 path on CPU: world_size-2 gloo processes shard the utterances as bench.py / decode() do
(no data-path collective), run their share through the CPU oracle (test infrastructure standing in
for the GPU engine), and the union of the per-rank results must equal the single-process result.
Also checks the MAX-over-ranks timing reduction bench.py uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastspeech2_amd import synth
from fastspeech2_amd.run import shard
from oracle import fs2_oracle

KW = dict(adim=64, aheads=2, positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3, eunits=64,
          dunits=64, elayers=1, dlayers=1, duration_predictor_chans=32, postnet_layers=0, use_scaled_pos_enc=True)


def _utts():
    return synth.synth_utterances(10, 3, idim=30, mean_len=9, std_len=3, min_len=3, max_len=15, sort=False)


def _run_share(indices, xs):
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(30, 12, 5, dur_bias=float(np.log(3.0)), **KW).items()}
    cfg = fs2_oracle.full_cfg(30, 12, **KW)
    return {i: fs2_oracle.infer_one(sd, cfg, torch.from_numpy(xs[i]))["mel"].numpy() for i in indices}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    xs = _utts()
    lengths = [len(x) for x in xs]
    mine = [i for b in shard.shard_buckets(lengths, world, rank, batch_size=2) for i in b]
    local = _run_share(mine, xs)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # bench.py's time reduction
    frames = torch.tensor([float(sum(v.shape[0] for v in local.values()))], dtype=torch.float64)
    dist.all_reduce(frames, op=dist.ReduceOp.SUM)              # and its frame count
    merged = shard.gather_results({i: v.shape[0] for i, v in local.items()}, world)
    dist.barrier()
    q.put((rank, mine, {i: v for i, v in local.items()}, float(t), float(frames), merged))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
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
    xs = _utts()
    ref = _run_share(range(len(xs)), xs)
    seen = {}
    for rank, mine, local, tmax, frames, merged in got:
        assert tmax == 2.0
        assert frames == float(sum(v.shape[0] for v in ref.values()))
        assert merged == {i: v.shape[0] for i, v in ref.items()}
        for i in mine:
            assert i not in seen
            seen[i] = local[i]
    assert sorted(seen) == list(range(len(xs)))                 # a partition: nothing lost, nothing twice
    for i, mel in seen.items():
        assert mel.shape == ref[i].shape and np.abs(mel - ref[i]).max() <= 1e-5   # thread counts differ


def test_shard_helpers():
    lengths = [5, 9, 3, 9, 7, 1, 8]
    order = shard.shard_round_robin(lengths, 1, 0)
    assert [lengths[i] for i in order] == sorted(lengths, reverse=True)
    r0, r1 = shard.shard_round_robin(lengths, 2, 0), shard.shard_round_robin(lengths, 2, 1)
    assert sorted(r0 + r1) == list(range(7)) and abs(len(r0) - len(r1)) <= 1
    b = shard.make_buckets(lengths, 3)
    assert [len(x) for x in b] == [3, 3, 1] and lengths[b[0][0]] == 9


@pytest.mark.parametrize("script", ["bench.py", "bench_train.py"])
def test_bench_gpus_2_launches_two_ranks_by_itself(script):
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape) must become two ranks:
    the parent spawns them (fastspeech2_amd/run/launch.py), rank 0 prints ONE line with n_gpus == 2.
    --rendezvous-only swaps the GPU work for placeholder numbers (gloo), the launch / rendezvous / reduction path is
    the real one.  A rank that fails makes the parent exit non-zero."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["max_rank_time"] == 2.0          # MAX over ranks of (1 + rank)
    # mismatched launch is refused by every rank -> the parent reports failure
    bad = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2", "--rendezvous-only"],
                         env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


def test_rank_cpu_binding_splits_the_allowed_cpus():
    """run/launch.py:bind_rank_cpus: every rank keeps a disjoint, equal share of the CPUs the job may use (checked in
    child processes, so that this process keeps its own mask)."""
    import os
    import subprocess
    import sys
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from fastspeech2_amd.run.launch import bind_rank_cpus; import os; "
            "mine = bind_rank_cpus(); print(sorted(os.sched_getaffinity(0)) == mine, mine)" % root)
    shares = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", RANK=str(r), WORLD_SIZE="2")
        out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip()
        ok, mine = out.split(" ", 1)
        assert ok == "True", out
        shares.append(set(eval(mine)))
    assert shares[0] and shares[1] and not (shares[0] & shares[1])
    assert len(shares[0]) == len(shares[1]) == len(os.sched_getaffinity(0)) // 2
