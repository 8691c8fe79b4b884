#!/usr/bin/env python3
"""Training-step benchmark, BASELINE.json config 5: the BASELINE architecture, 32 utterances per GPU,
teacher-forced durations ds ~ randint(1, 13), ys/ps/es ~ N(0, 1); one step = forward + backward +
(N > 1: bucketed RCCL gradient all-reduce overlapped with the backward) + clip + Adam (Noam schedule),
dropout on at the reference's rates (counter-based masks; --no-dropout for the deterministic step).

    python bench_train.py --gpus N --steps K --warmup W
    (N > 1: under torchrun, or bare — without WORLD_SIZE in the environment the process spawns the N ranks itself,
     fastspeech2_amd/run/launch.py)

Prints ONE JSON line on rank 0 (same shape as bench.py's).  bench.py stays the headline (config 2) bench.
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The data-parallel step
# uses compute, weight-gradient, RCCL-internal and optimizer-side streams; when two of them share a queue, one
# stream's event wait stalls the other (12.5 vs 9.0 ms per step with the reducer on, measured with one rank).
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fastspeech2_amd.run.launch import bind_rank_cpus, maybe_self_launch  # noqa: E402  (no torch, no GPU)

if __name__ == "__main__":
    _n = 1
    for _i, _a in enumerate(sys.argv[1:]):
        if _a == "--gpus" and _i + 2 < len(sys.argv):
            _n = int(sys.argv[_i + 2])
        elif _a.startswith("--gpus="):
            _n = int(_a.split("=", 1)[1])
    maybe_self_launch(_n)          # N > 1 without a launcher: become N ranks (the parent never touches a GPU)
    CPU_SHARE = bind_rank_cpus()   # a rank (under torchrun or self-launched): its share of the host cores
else:
    CPU_SHARE = None

import numpy as np                                                        # noqa: E402
import torch                                                              # noqa: E402

from fastspeech2_amd import synth                                        # noqa: E402
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer   # noqa: E402
from fastspeech2_amd.network.optimizer import get_std_opt                # noqa: E402
from fastspeech2_amd.run.train import GradReducer, Updater               # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
BATCH = 32


def make_batch(world, rank, seed=0):
    utts = synth.synth_utterances(BATCH * world, seed, idim=synth.BENCH_IDIM)[rank::world]
    rng = np.random.Generator(np.random.PCG64(seed + 100 + rank))
    B, Tp = len(utts), max(len(u) for u in utts)
    xs = np.zeros((B, Tp), np.int64)
    ds = np.zeros((B, Tp), np.int64)
    for b, u in enumerate(utts):
        xs[b, :len(u)] = u
        ds[b, :len(u)] = rng.integers(1, 13, size=len(u))
    ilens = np.array([len(u) for u in utts])
    olens = ds.sum(1)
    Tm = int(olens.max())
    f = torch.from_numpy
    return dict(xs=f(xs), ilens=f(ilens), ys=f(rng.standard_normal((B, Tm, synth.BENCH_ODIM), dtype=np.float32)),
                olens=f(olens), ds=f(ds), ps=f(rng.standard_normal((B, Tm, 4), dtype=np.float32)),
                es=f(rng.standard_normal((B, Tm, 1), dtype=np.float32)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--breakdown", action="store_true", help="per-tag kernel times (HIP events) of one extra step")
    ap.add_argument("--no-dropout", action="store_true", help="deterministic step (every dropout site off)")
    ap.add_argument("--rccl-single", action="store_true",
                    help="with one GPU: still create the nccl (RCCL) group and run the bucketed all-reduces")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="no GPU, no model: gloo process group, barrier, one reduction, the line's skeleton "
                         "(checks the N-rank launch path on a CPU-only machine)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rendezvous_only:
        import torch.distributed as dist
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "training mel frames/sec (node) at batch 32 per GPU", "value": None,
                              "n_gpus": world, "rendezvous_only": True, "max_rank_time": float(t[0])}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.rccl_single:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29571")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    kw = dict(synth.BASELINE_OVERRIDES)
    model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}, strict=True)
    model.to(dev).train()
    model.set_precision(args.dtype)
    model.dropout_enabled = not args.no_dropout
    eng = model.train_engine()
    reducer = GradReducer(eng, single_rank_too=args.rccl_single) if dist is not None else None
    optimizer = get_std_opt(model, kw["adim"], 4000, 1.0)
    updater = Updater(model, grad_clip=1.0, optimizer=optimizer, reducer=reducer)
    batch = make_batch(world, rank)
    packed = model.pack_batch(**batch)                 # inputs resident in HBM before the timed region
    frames = float(batch["olens"].sum())
    nparams = sum(p.numel() for p in model.parameters())

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        updater.update({"packed": packed})
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        updater.update({"packed": packed})
    host_elapsed = time.perf_counter() - t0           # every launch enqueued: ~ ms_per_step means the step is host-bound
    fence()
    elapsed = time.perf_counter() - t0
    loss = model.reporter.observation["loss"]
    stats = torch.tensor([elapsed, frames, host_elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed, frames, host_elapsed = float(mx[0]), float(stats[1]), float(mx[2])

    breakdown = None
    if rank == 0 and args.breakdown:
        eng.profile = []
        model(packed=packed).backward()
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, e0, e1 in eng.profile:
            a = agg.setdefault(tag, [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += flops
            a[2] += 1
        eng.profile = None
        model.zero_grad()
        breakdown = {k: dict(ms=round(v[0], 3), launches=v[2], tflops=round(v[1] / (v[0] * 1e-3) / 1e12, 1))
                     for k, v in agg.items()}

    if rank == 0:
        il, ol = [int(v) for v in batch["ilens"]], [int(v) for v in batch["olens"]]
        flops = 3.0 * synth.algorithmic_flops(il, ol) * world       # forward + input-gradient + weight-gradient GEMMs
        value = frames * args.steps / elapsed
        line = {
            "metric": "training mel frames/sec (node) at batch 32 per GPU", "value": round(value, 1),
            "unit": "mel frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 5: 32 utterances/GPU, teacher-forced ds~randint(1,13), fwd + bwd + "
                                   "grad all-reduce (N>1) + clip + Adam(Noam); 4+4 FFT blocks d=256 h=2 conv-k9 U=1024",
                       "frames_per_step": frames, "params": nparams, "final_loss": round(loss, 5),
                       "host_ms_per_step": round(host_elapsed / args.steps * 1e3, 4),
                       "launch_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                                      "cpus_of_rank0": None if CPU_SHARE is None else len(CPU_SHARE)},
                       "model_tflops": round(flops * args.steps / elapsed / 1e12, 2),
                       "model_mfma_frac": round(flops * args.steps / elapsed / 1e12 / world / PEAK_TFLOPS[args.dtype], 4),
                       "dropout": "off" if args.no_dropout else
                       "on (reference rates: decoder 0.1 / pos 0.1 / attention 0.1, duration predictor 0.1, "
                       "pitch+energy predictors 0.5, encoder 0)"},
            "breakdown": breakdown,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
