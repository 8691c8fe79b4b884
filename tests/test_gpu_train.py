"""-m gpu: the training step (forward + backward HIP schedules behind `loss.backward()`) against the
gradient goldens of the repaired reference and against the oracle's autograd on ragged batches."""
import math
import numpy as np
import pytest
import torch

from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from oracle import fs2_oracle
from tests import helpers

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GRAD_TOL = 2e-4        # fp32 path: |g - g_ref| <= tol * max|g_ref| per tensor (floor 1e-2 on the scale)


def _model(meta, dtype="fp32"):
    idim, odim, kw = helpers.split_cfg(meta["cfg"])
    m = FeedForwardTransformer(idim, odim, **kw)
    m.load_state_dict(helpers.golden_state_dict(meta), strict=True)
    m.to(DEV).train()
    m.set_precision(dtype)
    m.dropout_enabled = False          # goldens / oracle are dropout-free (eval-mode reference modules)
    return m


@pytest.mark.parametrize("name", helpers.GRAD_CASES)
def test_backward_matches_reference_gradients(name):
    z, meta = helpers.load_golden(name)
    model = _model(meta)
    t = lambda k: torch.from_numpy(z[k])
    loss = model(t("xs"), t("ilens"), t("ys"), t("olens"), t("ds"), t("ps"), t("es"))
    assert abs(loss.item() - meta["loss"]) <= 2e-5 * max(1.0, abs(meta["loss"]))
    loss.backward()
    obs = model.reporter.observation
    for k in ("l1_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(obs[k] - meta["losses"][k]) <= 2e-5 * max(1.0, abs(meta["losses"][k])), k
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert sorted(grads) == sorted(meta["grad_names"])
    for k in meta["grad_names"]:
        assert grads[k] is not None, k
        want, got, wstat, gstat = helpers.golden_grad_sample(z, meta, k, grads[k])
        tol = GRAD_TOL * max(np.abs(want).max(), 1e-2)
        assert np.abs(want - got).max() <= tol, (k, np.abs(want - got).max(), np.abs(want).max())
        assert abs(wstat[1] - gstat[1]) <= 1e-3 * max(wstat[1], 1e-2), k


def test_baseline_size_step_matches_reference_gradients():
    """BASELINE config 5 shapes: B=32 utterances of 120 phones -> 781 frames each, 4+4 blocks, d=256, conv k=9 — all
    142 parameter gradients against samples of what the repaired reference's loss.backward() left, then the bf16 step
    against the same samples (stated tolerance: cosine of the whole sampled gradient vector).
    fp32 bar here: 5e-4 * max|g| per tensor (2e-4 on the small fixtures).  Measured (tools/grad_err_big.py): median
    1e-5, every l2 norm within 6e-5, all but two tensors under 2e-4; the two are encoder FFN w1 weights / biases
    (3.2e-4, 2.0e-4): with 3840 phones x 1024 units some ReLU pre-activations sit within fp32 noise of 0, and a token
    whose gate differs from the reference's contributes (or not) one whole dY x X term to single entries."""
    z, meta = helpers.load_golden(helpers.BIG_TRAIN_CASES[0])
    batch = [torch.from_numpy(a) for a in helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])]
    assert np.array_equal(batch[4].numpy(), z["ds"])
    model = _model(meta)
    loss = model(*batch)
    assert abs(loss.item() - meta["loss"]) <= 2e-5 * max(1.0, abs(meta["loss"]))
    loss.backward()
    obs = model.reporter.observation
    for k in ("l1_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(obs[k] - meta["losses"][k]) <= 2e-5 * max(1.0, abs(meta["losses"][k])), k
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert sorted(grads) == sorted(meta["grad_names"])
    wants = []
    for k in meta["grad_names"]:
        want, got, wstat, gstat = helpers.golden_grad_sample(z, meta, k, grads[k])
        tol = 5e-4 * max(np.abs(want).max(), 1e-2)
        assert np.abs(want - got).max() <= tol, (k, np.abs(want - got).max(), np.abs(want).max())
        assert abs(wstat[1] - gstat[1]) <= 1e-4 * max(wstat[1], 1e-2), k
        wants.append(want.astype(np.float64))
    del model, grads
    model = _model(meta, "bf16")
    model(*batch).backward()
    gots = [helpers.golden_grad_sample(z, meta, k, p.grad)[1].astype(np.float64) for k, p in model.named_parameters()]
    names = [k for k, _ in model.named_parameters()]
    a = np.concatenate([wants[meta["grad_names"].index(k)] for k in names])
    b = np.concatenate(gots)
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos >= 0.99, cos
    assert abs(np.linalg.norm(b) / np.linalg.norm(a) - 1.0) <= 0.05


def _ragged_batch(cfg, seed, ilens):
    rng = np.random.Generator(np.random.PCG64(seed))
    B, Tp = len(ilens), max(ilens)
    xs = np.zeros((B, Tp), np.int64)
    ds = np.zeros((B, Tp), np.int64)
    for b, L in enumerate(ilens):
        xs[b, :L] = rng.integers(1, cfg["idim"], size=L)
        ds[b, :L] = rng.integers(0, 6, size=L)
        ds[b, 0] += 1
    olens = ds.sum(1)
    Tm = int(olens.max())
    ys = rng.standard_normal((B, Tm, cfg["odim"]), dtype=np.float32)
    ps = rng.standard_normal((B, Tm, 4), dtype=np.float32)
    es = rng.standard_normal((B, Tm, 1), dtype=np.float32)
    f = torch.from_numpy
    return f(xs), torch.tensor(ilens), f(ys), f(olens), f(ds), f(ps), f(es)


@pytest.mark.parametrize("name,ilens", [("train_tiny_conv9", [13, 1, 40, 7]), ("train_tiny_conv3_mixed", [70, 9, 22]),
                                        ("train_tiny_linear", [5, 64])])
def test_ragged_batch_gradients_match_oracle(name, ilens):
    _, meta = helpers.load_golden(name)
    cfg = helpers.oracle_cfg(meta)
    sd = helpers.golden_state_dict(meta)
    xs, il, ys, ol, ds, ps, es = _ragged_batch(cfg, 3, ilens)
    terms, ref = fs2_oracle.train_grads(sd, cfg, xs, il, ys, ol, ds, ps, es)
    model = _model(meta)
    loss = model(xs, il, ys, ol, ds, ps, es)
    (loss * 0.5).backward()                                     # upstream gradient != 1 (accum_grad scaling)
    assert abs(loss.item() - terms["loss"]) <= 2e-5 * max(1.0, terms["loss"])
    for k, p in model.named_parameters():
        want = 0.5 * ref[k].double()
        got = p.grad.detach().cpu().double()
        tol = GRAD_TOL * max(want.abs().max().item(), 1e-2)
        assert (want - got).abs().max().item() <= tol, (k, (want - got).abs().max().item(), want.abs().max().item())
    # a second backward accumulates into .grad (run/train_decode.py:94-102, accum_grad)
    loss2 = model(packed=model.pack_batch(xs, il, ys, ol, ds, ps, es))
    (loss2 * 0.25).backward()
    k = "decoder.encoders.0.feed_forward.w1.weight"
    want = 0.75 * ref[k].double()
    got = dict(model.named_parameters())[k].grad.cpu().double()
    assert (want - got).abs().max().item() <= 2 * GRAD_TOL * max(want.abs().max().item(), 1e-2)


@pytest.mark.parametrize("name", ["postnet_tiny_bn", "postnet_tiny_nobn"])
def test_postnet_training_gradients_match_oracle(name):
    """Training with the repaired PostNet (network/fastspeech2.py:222-225, :271-274): loss, every gradient and the
    BatchNorm running statistics against the oracle's autograd through torch's train()-mode batch_norm over the frames
    of the batch (dropout off).  Parity vs the literal reference is unpinned here: its PostNet cannot run (D7)."""
    _, meta = helpers.load_golden(name)
    cfg = helpers.oracle_cfg(meta)
    sd = helpers.golden_state_dict(meta)
    xs, il, ys, ol, ds, ps, es = _ragged_batch(cfg, 12, [23, 4, 37, 1])   # (seeds 11, 14: a ReLU gate of the tiny pitch predictor sits on zero)
    terms, ref = fs2_oracle.train_grads(sd, cfg, xs, il, ys, ol, ds, ps, es)
    model = _model(meta)
    loss = model(xs, il, ys, ol, ds, ps, es)
    (loss * 0.5).backward()
    assert abs(loss.item() - terms["loss"]) <= 2e-5 * max(1.0, terms["loss"])
    for k, p in model.named_parameters():
        want = 0.5 * ref[k].double()
        got = p.grad.detach().cpu().double()
        tol = GRAD_TOL * max(want.abs().max().item(), 1e-2)
        assert (want - got).abs().max().item() <= tol, (k, (want - got).abs().max().item(), want.abs().max().item())
    bufs = dict(model.named_buffers())
    for k, want in terms.get("running", {}).items():
        got = bufs[k].cpu()
        assert (want - got).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), k
    if cfg["use_batch_norm"]:
        assert int(bufs["postnet.postnet.0.1.num_batches_tracked"]) == 1
        # eval() after the step uses the UPDATED running statistics (the engine's folded weights are rebuilt)
        model.eval()
        with torch.no_grad():
            ev = model(xs, il, ys, ol, ds, ps, es).item()
        sd2 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        want_ev = fs2_oracle.train_loss(sd2, cfg, xs, il, ys, ol, ds, ps, es)["loss"]
        assert abs(ev - want_ev) <= 5e-5 * max(1.0, want_ev)
    # dropout on (rate 0.5 after every PostNet layer): another loss, finite gradients; bf16 runs
    model.train()
    model.dropout_enabled = True
    model.zero_grad()
    l2 = model(xs, il, ys, ol, ds, ps, es)
    l2.backward()
    assert math.isfinite(l2.item()) and abs(l2.item() - loss.item()) > 1e-6
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    mb = _model(meta, "bf16")
    lb = mb(xs, il, ys, ol, ds, ps, es)
    lb.backward()
    assert abs(lb.item() - terms["loss"]) <= 0.05 * max(1.0, terms["loss"])
    g = dict(mb.named_parameters())["postnet.postnet.0.0.weight"].grad.cpu().double()
    w = ref["postnet.postnet.0.0.weight"].double()
    assert (g - w).abs().max().item() <= 0.1 * max(w.abs().max().item(), 1e-2)


def test_bf16_gradients_track_fp32():
    _, meta = helpers.load_golden("train_tiny_conv3_mixed")
    cfg = helpers.oracle_cfg(meta)
    batch = _ragged_batch(cfg, 5, [50, 31, 12])
    grads = {}
    for dtype in ("fp32", "bf16"):
        model = _model(meta, dtype)
        model(*batch).backward()
        grads[dtype] = {k: p.grad.detach().cpu().double().reshape(-1) for k, p in model.named_parameters()}
    a = torch.cat([grads["fp32"][k] for k in sorted(grads["fp32"])])
    b = torch.cat([grads["bf16"][k] for k in sorted(grads["bf16"])])
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos >= 0.995, cos       # bf16 operands, fp32 accumulation: stated tolerance on the whole gradient vector
    assert abs(float(b.norm() / a.norm()) - 1.0) <= 0.03


def test_yaml_dims_gradients_match_oracle():
    """shipped-YAML shapes (d=384, 4 heads of 96, linear FFN 1536, scaled pos-enc) on 2+2 blocks, PostNet off."""
    _, meta = helpers.load_golden("yaml_dims_2x2")
    cfg = helpers.oracle_cfg(meta)
    sd = helpers.golden_state_dict(meta)
    xs, il, ys, ol, ds, ps, es = _ragged_batch(cfg, 21, [33, 70, 5])
    terms, ref = fs2_oracle.train_grads(sd, cfg, xs, il, ys, ol, ds, ps, es)
    model = _model(meta)
    loss = model(xs, il, ys, ol, ds, ps, es)
    loss.backward()
    assert abs(loss.item() - terms["loss"]) <= 2e-5 * max(1.0, terms["loss"])
    for k, p in model.named_parameters():
        want, got = ref[k].double(), p.grad.detach().cpu().double()
        tol = GRAD_TOL * max(want.abs().max().item(), 1e-2)
        assert (want - got).abs().max().item() <= tol, (k, (want - got).abs().max().item(), want.abs().max().item())


def test_updater_trains_and_reducer_streams():
    """A few optimiser steps through the reference's update rule lower the loss; the bucketed reducer's stream /
    event plumbing runs on the device (world size 1: the collective itself is skipped)."""
    from fastspeech2_amd.network.optimizer import get_std_opt
    from fastspeech2_amd.run.train import Updater
    _, meta = helpers.load_golden("train_tiny_linear")
    model = _model(meta, "bf16")
    cfg = helpers.oracle_cfg(meta)
    xs, il, ys, ol, ds, ps, es = _ragged_batch(cfg, 2, [30, 18, 25, 9])
    batch = dict(xs=xs, ilens=il, ys=ys, olens=ol, ds=ds, ps=ps, es=es, spkids=torch.zeros(4), labels=torch.zeros(4, 3))
    eng = model.train_engine()
    seen = []
    eng.grad_ready = seen.append
    up = Updater(model, grad_clip=1.0, optimizer=get_std_opt(model, cfg["adim"], 20, 1.0), device=DEV)
    losses = []
    for _ in range(12):
        up.update(batch)
        losses.append(model.reporter.observation["loss"])
    assert up.iteration == 12 and losses[-1] < losses[0] - 0.2, losses
    n_buckets = len(eng.bucket_slices)
    assert seen[:n_buckets] == list(range(n_buckets))                 # buckets are announced in flat-buffer order
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_two_process_data_parallel_step(tmp_path):
    """Two data-parallel ranks (two processes on this GPU, gloo transport standing in for RCCL, which cannot put two
    ranks on one device) with the bucketed reducer hooked into the backward: both end with the mean of the two
    single-rank gradients.  A rank that stops fails the test with the ranks' stage lines and (after 60 s) their
    Python stacks attached — no second attempt."""
    import os
    import socket
    import subprocess
    import sys
    name = "train_tiny_conv3_mixed"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    stage_log = os.path.join(str(tmp_path), "stages")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FS2_DP_STAGE_LOG=stage_log)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dp_worker.py"), str(tmp_path),
                                       name], env=env, cwd=root))
    codes = None
    try:
        codes = [p.wait(timeout=150) for p in procs]
    except subprocess.TimeoutExpired:
        pass
    finally:
        for p in procs:                  # never leave a rank behind on the GPU (it would stall this process's exit)
            if p.poll() is None:
                p.kill()
                p.wait()
    if codes != [0, 0]:
        report = []
        for rank in range(2):
            path = "%s.rank%d" % (stage_log, rank)
            report.append("---- rank %d ----\n%s" % (rank, open(path).read() if os.path.isfile(path) else "(no stage log)"))
        pytest.fail("data-parallel ranks %s\n%s" % ("stalled" if codes is None else "exited with %s" % codes,
                                                    "\n".join(report)))
    got = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(2)]
    # single-process truth: the two ranks' gradients computed here, then averaged
    _, meta = helpers.load_golden(name)
    cfg = helpers.oracle_cfg(meta)
    want = None
    for rank in range(2):
        model = _model(meta)
        model(*_ragged_batch(cfg, 100 + rank, [17 + 5 * rank, 30, 8])).backward()
        g = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
        want = g if want is None else {k: 0.5 * (want[k] + g[k]) for k in g}
    for k, w in want.items():
        for r in range(2):
            a = torch.from_numpy(got[r][k.replace(".", "__")]).double()
            assert (a - w).abs().max().item() <= 1e-6 * max(1.0, w.abs().max().item()), (k, r)
    assert got[0]["loss"] != got[1]["loss"]          # different shares of the data


def test_async_nan_guard_skips_the_update_and_takes_back_the_schedule_tick():
    from fastspeech2_amd.network.optimizer import get_std_opt
    from fastspeech2_amd.run.train import Updater
    _, meta = helpers.load_golden("train_tiny_linear")
    model = _model(meta)
    cfg = helpers.oracle_cfg(meta)
    xs, il, ys, ol, ds, ps, es = _ragged_batch(cfg, 2, [12, 7])
    good = dict(xs=xs, ilens=il, ys=ys, olens=ol, ds=ds, ps=ps, es=es)
    bad = dict(good, ps=ps.clone())
    bad["ps"][0, 0] = float("nan")                           # MSE target -> NaN gradient (an L1 target would give sign(NaN) = 0)
    opt = get_std_opt(model, cfg["adim"], 20, 1.0)
    up = Updater(model, grad_clip=1.0, optimizer=opt, device=DEV)
    up.update(good)
    up.finalize()
    w0 = {k: p.detach().clone() for k, p in model.named_parameters()}
    steps0 = [int(st["step"]) for st in opt.optimizer.state.values()]
    up.update(bad)                                           # NaN gradient norm -> no update
    up.finalize()
    assert up.skipped_steps == 1 and opt._step == 1          # the tick counted by NoamOpt.step() was taken back
    assert all(torch.equal(w0[k], p.detach()) for k, p in model.named_parameters())
    assert [int(st["step"]) for st in opt.optimizer.state.values()] == steps0
    up.update(good)
    up.finalize()
    assert opt._step == 2 and up.skipped_steps == 1
    assert any(not torch.equal(w0[k], p.detach()) for k, p in model.named_parameters())
    # literal host-side check gives the same decisions
    up2 = Updater(model, grad_clip=1.0, optimizer=opt, device=DEV, sync_nan_check=True)
    up2.update(bad)
    assert up2.skipped_steps == 1 and opt._step == 2


def test_train_driver_runs_evaluates_snapshots_and_resumes(tmp_path):
    """fastspeech2_amd.run.train_decode.main: two epochs on a synthetic corpus, validation, snapshot, resume"""
    import json
    import os
    from fastspeech2_amd.run import train_decode
    out = str(tmp_path / "exp")
    common = ["--outdir", out, "--synthetic", "24", "--adim", "64", "--aheads", "2", "--elayers", "1", "--dlayers", "1",
              "--eunits", "128", "--dunits", "128", "--positionwise-layer-type", "conv1d",
              "--positionwise-conv-kernel-size", "3", "--duration-predictor-chans", "64", "--postnet-layers", "0",
              "--batch-bins", "60000", "--transformer-warmup-steps", "50", "--report-interval-iters", "2",
              "--precision", "fp32", "--sortagrad", "1"]
    args = train_decode.get_parser().parse_args(common + ["--epochs", "2"])
    args.adim = int(args.adim)
    s1 = train_decode.train(args)
    assert s1["iterations"] >= 2 and s1["skipped_steps"] == 0
    assert math.isfinite(s1["validation/main/loss"]) and math.isfinite(s1["main/loss"])
    for f in ("model.json", "snapshot.ep.1", "snapshot.ep.2", "model.loss.best"):
        assert os.path.isfile(os.path.join(out, f)), f
    with open(os.path.join(out, "model.json")) as fh:
        idim, odim, conf = json.load(fh)
    assert (idim, odim) == (80, 80) and conf["adim"] == 64
    args2 = train_decode.get_parser().parse_args(common + ["--epochs", "3", "--resume", os.path.join(out, "snapshot.ep.2")])
    s2 = train_decode.train(args2)
    assert s2["iterations"] > s1["iterations"]                      # continued from the restored counters: one more epoch
    assert s2["validation/main/loss"] < 1.5 * s1["validation/main/loss"]
