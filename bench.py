#!/usr/bin/env python3
"""Headline benchmark: mel frames/s of the FastSpeech2 forward path (phone ids resident in HBM ->
mel resident in HBM) on BASELINE.json config 2: the BASELINE architecture (4+4 FFT blocks, d=256,
2 heads, conv k=9 FFN, U=1024), random-init, one batch of 32 length-bucketed synthetic utterances
(~120 phones -> ~800 mel frames) per GPU, bf16 MFMA operands.

    python bench.py --gpus N --steps K --warmup W [--scaling strong]
    (N > 1: either under torchrun — python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... —
     or bare: without WORLD_SIZE in the environment the process spawns the N ranks itself, fastspeech2_amd/run/launch.py)

One "step" = one forward over one 32-utterance batch.  Utterances are independent, so N GPUs run N
different batches with no data-path collective (weak scaling); only the timing is reduced (MAX).
--scaling strong is BASELINE config 3: the SAME 256 utterances at every N, length-sorted and dealt to the ranks in
32-utterance batches (8 / N batches per rank and step); value = the 256 utterances' frames / the slowest rank's time.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the conv-FFN implicit GEMM, timed
with HIP events on its launch stream in a second, separately timed pass) and `cpu_baseline` (the CPU
oracle, kind "port", timed on this box's host cores on the same batch).
"""
import argparse
import collections
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The throughput loop uses
# the default stream, the phone-level side stream and three frame-level streams; with four queues two of them share
# one and an event wait of one stalls the other (3 frame streams: 1.77 ms per step with 4 queues, 1.65 with 8; 2 frame
# streams: 1.67 either way).  Must be set before the HIP runtime starts.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fastspeech2_amd.run.launch import bind_rank_cpus, maybe_self_launch   # noqa: E402  (no torch, no GPU)


def _gpus_arg(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__":
    maybe_self_launch(_gpus_arg(sys.argv[1:]))     # N > 1 without a launcher: become N ranks (parent never touches a GPU)
    CPU_SHARE = bind_rank_cpus()                   # a rank (under torchrun or self-launched): its share of the host cores
else:
    CPU_SHARE = None

import numpy as np                                                   # noqa: E402
import torch                                                         # noqa: E402

from fastspeech2_amd import synth                                   # noqa: E402
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "f16x3": 2500.0}   # MI355X_MICROARCH.md chip table (dense; f16x3: the fp16 pipe, 3 MFMAs per product)
PEAK_HBM_GBS = 8000.0                            # HBM3E spec (6.3 TB/s measured achievable, same guide)
COPY_RATE_GBS = 6290.0                           # measured float4 copy on MI355X (same guide): what a streaming kernel can reach
HOP_SECONDS = 256.0 / 22050.0                    # assumed hop (reference defines none)
BATCH = 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # the pipelined loop pays one fill + drain (~1.5 ms) per timed run: DESIGN.md 5.1
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "f16x3"],
                    help="frame-level half; f16x3 = fp32-grade results on error-compensated fp16 operand pairs (3 MFMAs per product)")
    ap.add_argument("--encoder-dtype", default="f16x3", choices=["bf16", "fp32", "f16x3"],
                    help="phone-level half (encoder + duration predictor). f16x3 (default): error-compensated fp16 GEMM "
                         "operands, durations bit-exact vs the fp32 engine / reference goldens; bf16: ~1 %% of the "
                         "durations differ by one frame; fp32: exact-fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="no cross-step overlap: one forward after the other")
    ap.add_argument("--frame-streams", type=int, default=3, help="alternate the frame-level halves over this many streams")
    ap.add_argument("--phone-group", type=int, default=4,
                    help="run the phone-level half of this many consecutive steps as one ragged batch (Engine.forward_pipelined)")
    ap.add_argument("--no-pure", action="store_true", help="skip the extra timed loop with the phone-level half in --dtype")
    ap.add_argument("--engine-set", action="append", default=[], metavar="NAME=0|1",
                    help="set a kernel-dispatch switch of the engine (reg_w, big_tiles): A/B runs; recorded in the line")
    ap.add_argument("--cpu-passes", type=int, default=12)   # ~10-15 s of host CPU work
    ap.add_argument("--force-dist", action="store_true",
                    help="with one GPU: still create the nccl (RCCL) group and run the barrier / reductions")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 32 utterances per GPU (config 2, the headline); strong: the same 256 utterances "
                         "split over the GPUs (config 3)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="no GPU, no model: build the process group on gloo, run the barrier and the MAX / SUM "
                         "reductions on placeholder numbers and print the line's skeleton (checks the N-rank launch "
                         "path on a CPU-only machine; the value is null)")
    return ap.parse_args()


def build_model(dtype, encoder_dtype, device):
    kw = dict(synth.BASELINE_OVERRIDES)
    model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    sd = synth.bench_state_dict()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval().to(device)
    model.set_precision(dtype, encoder_dtype)
    return model, sd, kw


def rank_batch(world, rank):
    """world*32 utterances (seed 0), length-sorted, dealt round-robin: every rank gets one
    32-utterance bucket with a near-identical length profile."""
    utts = synth.synth_utterances(BATCH * world, 0, idim=synth.BENCH_IDIM)
    return utts[rank::world]


STRONG_UTTERANCES = 256


def rank_batches_strong(world, rank):
    """BASELINE config 3: the same 256 utterances (seed 0) whatever the world size; this rank's share = its
    round-robin deal of the length-sorted list, cut into 32-utterance batches (run/shard.py)."""
    from fastspeech2_amd.run.shard import shard_buckets
    utts = synth.synth_utterances(STRONG_UTTERANCES, 0, idim=synth.BENCH_IDIM, sort=False)
    return [[utts[i] for i in bucket] for bucket in shard_buckets([len(u) for u in utts], world, rank, BATCH)]


def rendezvous_only(args, world, rank):
    """--rendezvous-only: the launch / rendezvous / reduction skeleton of main() on gloo, nothing else."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    stats = torch.tensor([1.0 + rank, 100.0], dtype=torch.float64)
    mx = stats.clone()
    if world > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "mel frames/sec (node) at batch 32 per GPU", "value": None, "unit": "mel frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                          "rendezvous_only": True, "max_rank_time": float(mx[0]), "sum_frames": float(stats[1])}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sd_np, kw, xs, passes):
    from oracle import fs2_oracle                  # checker used as the CPU baseline ("port")
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    cfg = fs2_oracle.full_cfg(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    fs2_oracle.infer_one(sd, cfg, torch.from_numpy(xs[-1]))         # warm-up
    t0 = time.perf_counter()
    frames = 0
    for _ in range(passes):
        for x in xs:
            frames += fs2_oracle.infer_one(sd, cfg, torch.from_numpy(x))["mel"].shape[0]
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="mel frames/s", cores=threads, host_cpus=os.cpu_count(), kind="port",
                sample="%d pass(es) over the same 32-utterance batch, fp32 torch-CPU oracle, %.1f s"
                       % (passes, dt))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29572")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    model, sd_np, kw = build_model(args.dtype, args.encoder_dtype, dev)
    eng = model.engine()
    for kv in args.engine_set:
        k, v = kv.split("=")
        if not isinstance(getattr(eng, k, None), bool):
            raise SystemExit("--engine-set: the engine has no switch %r" % k)
        setattr(eng, k, bool(int(v)))
    strong = args.scaling == "strong"
    batches = rank_batches_strong(world, rank) if strong else [rank_batch(world, rank)]
    feed = [(torch.from_numpy(np.concatenate(b)).to(dev), [len(x) for x in b]) for b in batches]
    xs = batches[0]
    ids, ilens = feed[0]

    def step():
        return eng.forward(ids, ilens)

    def run_steps(n):
        # throughput loop: the phone-level half of batch k+1 is enqueued on a side stream underneath the
        # frame-level half of batch k (Engine.forward_pipelined); every step does the full forward of every batch
        # of this rank (one batch in weak scaling, 8 / N in strong scaling).  Returns the outputs of the last step.
        last = []
        if args.serial:
            for _ in range(n):
                last = [eng.forward(i, l) for (i, l) in feed]
        else:
            last = collections.deque(maxlen=len(feed))     # keep only the last step's outputs alive
            for o in eng.forward_pipelined((feed[j % len(feed)] for j in range(n * len(feed))),
                                           frame_streams=args.frame_streams, phone_group=args.phone_group):
                last.append(o)
            last = list(last)
        return last

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run_steps(args.warmup)
    else:
        step()
    fence()
    t0 = time.perf_counter()
    outs = run_steps(args.steps)
    host_elapsed = time.perf_counter() - t0          # every launch of the timed steps enqueued (incl. the one blocking read
    fence()                                          # of the frame counts per batch): ~ ms_per_step means host-bound
    elapsed = time.perf_counter() - t0
    frames = float(sum(sum(o["olens"]) for o in outs))
    flops = sum(synth.algorithmic_flops(l, o["olens"]) for (_, l), o in zip(feed, outs))
    phones = float(sum(sum(l) for _, l in feed))
    stats = torch.tensor([elapsed, frames, flops, phones, host_elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed, host_elapsed = float(mx[0]), float(mx[4])
        frames, flops, phones = float(stats[1]), float(stats[2]), float(stats[3])

    # ---- rooflines: HIP events on the launch stream around every tagged launch, one forward at a time (a second pass
    #      over the same steps, outside the timed region).  `roofline` = the dominant kernel (the bf16 k=9 FFN conv
    #      implicit GEMM of the frame-level half); `rooflines` = the other kernels the north_star names.
    roofline, rooflines = None, None
    if rank == 0:
        eng.profile = []
        for _ in range(max(3, min(args.steps, 10))):
            step()
        torch.cuda.synchronize()
        recs = eng.profile
        eng.profile = None
        agg = {}
        for tag, kind, work, e0, e1 in recs:
            a = agg.setdefault(tag, [kind, 0.0, 0.0, 0])
            a[1] += work
            a[2] += e0.elapsed_time(e1)
            a[3] += 1
        peak = PEAK_TFLOPS[args.dtype]

        def entry(tags, note, mfma_per_product=1):
            kind = agg[tags[0]][0]
            work = sum(agg[t][1] for t in tags)
            t_ms = sum(agg[t][2] for t in tags)
            n = sum(agg[t][3] for t in tags)
            if kind == "flop":
                ach = work / (t_ms * 1e-3) / 1e12
                e = dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach * mfma_per_product / peak, 4),
                         flops_per_launch=work / n)
                if mfma_per_product != 1:
                    e["mfma_per_product"] = mfma_per_product
            else:
                ach = work / (t_ms * 1e-3) / 1e9
                # frac is against the 8 TB/s spec; a float4 copy reaches 6.29 TB/s on this part (MI355X_MICROARCH.md),
                # frac_of_copy_rate is against that.  HIP events around ONE short launch also time its launch gap.
                e = dict(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4),
                         frac_of_copy_rate=round(ach / COPY_RATE_GBS, 4), bytes_per_launch=work / n)
            e.update(kernel=note, launches=n, avg_launch_us=round(t_ms * 1e3 / n, 2))
            return e

        main_tags = [t for t in ("ffn1:" + args.dtype, "ffn2:" + args.dtype) if t in agg]
        roofline = entry(main_tags, "fs2::conv_gemm_kernel<%s, ..., KT=9> (the k=9 FFN convs of the frame-level half%s; own "
                         "kernel symbol; timed one forward at a time, i.e. without the cross-batch overlap of the "
                         "throughput loop)" % (args.dtype, "" if args.encoder_dtype != args.dtype else " and the phone-level half"),
                         3 if args.dtype == "f16x3" else 1)
        traffic = None     # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
        # (a PMC pass cannot run inside this process: the figure is the newest committed measurement of the same kernel
        #  on the same launches - tools/collect_round.sh: FETCH_SIZE / WRITE_SIZE passes over tools/ffn_conv_replay.py)
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_ffn_conv_traffic.json")))
        if args.dtype == "bf16" and tfiles:
            traffic = json.load(open(tfiles[-1])).get("hbm_bytes_per_launch")
            roofline["traffic_source"] = "profiles/" + os.path.basename(tfiles[-1])
        roofline["traffic"] = traffic
        rooflines = {}
        d = args.dtype
        others = [
            ("conv_ffn_f16x3", ["ffn1:f16x3", "ffn2:f16x3"], "fs2::conv_gemm_kernel<float, ..., SPLIT> (phone-level FFN convs "
             "on error-compensated fp16 operands: 3 MFMAs per product; achieved = algorithmic FLOP/s, frac counts the 3 MFMAs)", 3),
            ("gemm_k1", ["qkv:" + d, "out_proj:" + d, "feat_out:" + d], "fs2::conv_gemm_kernel<%s, ..., KT=0>: the k=1 GEMMs of the "
             "frame-level half (QKV, output projection, feat_out)" % d, 1),
            ("attention", ["attention:" + d], "fs2::attention_kernel<%s> (frame-level self-attention, 4 sum(T^2) d FLOP)" % d,
             3 if d == "f16x3" else 1),
            ("proj_add_layernorm", ["proj_ln"], "fs2::proj_ln_kernel (attention output projection + residual + norm1 of the frame-level "
             "post-LN blocks as one stream over the rows, W in registers; algorithmic bytes X + res + fp32 out + bf16 out + W)", 1),
            ("add_layernorm", ["add_layernorm"], "fs2::add_layernorm_kernel (frame level: residual + split-K partial sum + "
             "LayerNorm, fp32 + bf16 outputs)", 1),
            ("lr_gather", ["lr_gather"], "fs2::lr_gather_kernel (LengthRegulator expansion)", 1),
            ("variance_add_posenc", ["variance_add_posenc"], "fs2::variance_add_posenc_kernel", 1),
            ("layernorm_rowdot", ["layernorm_rowdot"], "fs2::layernorm_rowdot_kernel (pitch / energy heads)", 1),
        ]
        for key, tags, note, mpp in others:
            tags = [t for t in tags if t in agg]
            if tags:
                rooflines[key] = entry(tags, note, mpp)
        # the k=1 GEMMs move almost as many bytes as they compute on: the same launches against the HBM roof
        btags = [t + "/bytes" for t in ("qkv:" + d, "out_proj:" + d, "feat_out:" + d) if t + "/bytes" in agg]
        if btags and "gemm_k1" in rooflines:
            hb = entry(btags, "")
            rooflines["gemm_k1"].update(hbm_achieved=hb["achieved"], hbm_unit="GB/s", hbm_frac=hb["frac"],
                                        bytes_per_launch=hb["bytes_per_launch"],
                                        note="short-K GEMMs (K = 256): algorithmic bytes X + W + out (+ residual) against 8 TB/s")

    # ---- the pure-bf16 figure beside the exact-duration headline (same loop, phone-level half in bf16 too)
    pure = None
    if rank == 0 and world == 1 and args.encoder_dtype != args.dtype and not args.no_pure:
        model.set_precision(args.dtype, args.dtype)
        eng = model.engine()
        run_steps(max(args.warmup, 3))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pouts = run_steps(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        pframes = float(sum(sum(o["olens"]) for o in pouts))
        pure = dict(value=round(pframes * args.steps / dt, 1), ms_per_step=round(dt / args.steps * 1e3, 4),
                    note="encoder_dtype=%s: about 1 %% of the integer durations differ by one frame from the fp32 engine / "
                         "reference, so this mode does not meet the bit-exact bar" % args.dtype)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_np, kw, xs, args.cpu_passes)

    if rank == 0:
        value = frames * args.steps / elapsed
        line = {
            "metric": "mel frames/sec (node) at batch 32 per GPU",
            "value": round(value, 1), "unit": "mel frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": ("BASELINE config 3: the same 256 length-sorted utterances dealt to the GPUs in "
                                    "32-utterance batches (strong scaling), " if strong else
                                    "BASELINE config 2: 32 length-bucketed utterances/GPU (~120 phones -> ~800 mel "
                                    "frames), ") + "4+4 FFT blocks d=256 h=2 conv-k9 U=1024, random-init, idim=odim=80",
                       "frames_per_step": frames, "phones_per_step": phones, "batches_per_rank_and_step": len(feed),
                       "encoder_dtype": args.encoder_dtype,
                       "durations": ("equal to the REFERENCE's on all 3 767 phones of this batch and all 30 727 of the 256-utterance "
                                     "config-3 set (fixtures tests/golden/durations_*.npz generated by the reference itself; "
                                     "tests/test_gpu_model.py::test_durations_of_the_full_workloads_equal_the_reference); exact "
                                     "on the fixtures - a checkpoint whose activations leave the fp16 range is not covered"
                                     if args.encoder_dtype != "bf16" else "~1 % differ by one frame (bf16 encoder)"),
                       "pure_%s" % args.dtype: pure,
                       "step_overlap": "none" if args.serial else "phone-level halves of the next %d step(s) as one ragged batch on a side stream under the frame-level halves of the current ones; frame-level halves alternate over %d stream(s)" % (args.phone_group, args.frame_streams),
                       "host_ms_per_step": round(host_elapsed / args.steps * 1e3, 4),
                       "launch_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "phone_group": args.phone_group,
                                      "frame_streams": args.frame_streams, "engine_set": args.engine_set or None,
                                      "cpus_of_rank0": None if CPU_SHARE is None else len(CPU_SHARE)},
                       "rtf": round((elapsed / args.steps) / (frames * HOP_SECONDS), 8),
                       "model_tflops": round(flops * args.steps / elapsed / 1e12, 2),
                       "model_mfma_frac": round(flops * args.steps / elapsed / 1e12 / world / PEAK_TFLOPS[args.dtype], 4)},
            "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu,
        }
        if cpu:
            line["config"]["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
