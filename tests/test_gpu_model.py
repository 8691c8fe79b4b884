"""-m gpu: the whole forward path (FeedForwardTransformer on the HIP engine) against the golden
vectors of the repaired reference and against the CPU oracle.

Bars (BASELINE.json north_star): durations and LengthRegulator indices bit-exact, mel within
1e-3 max-abs in fp32.  bf16: durations are teacher-forced from the oracle (a bf16 encoder may
legitimately flip a rounding), mel tolerance BF16_MEL_TOL stated below."""
import numpy as np
import pytest
import torch

from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from oracle import fs2_oracle
from tests import helpers

pytestmark = pytest.mark.gpu

FP32_MEL_TOL = 1e-3       # north_star
BF16_MEL_TOL = 0.05       # max-abs on mel values of |.|max ~3.5; bf16 operands (8-bit mantissa) through the blocks; measured 1.3e-2..3.8e-2 on the fixtures
BF16_MEL_RMS_TOL = 0.012  # measured rms 4e-3..7.8e-3: a 2 x accuracy regression fails either this or the max-abs bar


def build_model(meta, precision="fp32", encoder_dtype=None):
    idim, odim, kw = helpers.split_cfg(meta["cfg"])
    model = FeedForwardTransformer(idim, odim, **kw)
    model.load_state_dict(helpers.golden_state_dict(meta), strict=True)
    model.eval().cuda()
    model.set_precision(precision, encoder_dtype)
    return model


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", helpers.INFER_CASES)
def test_fp32_matches_reference_golden(name, precision):
    """Both fp32-grade modes against the reference's golden vectors at the north_star's fp32 bars: 'fp32' (exact-fp32
    MFMA) and 'f16x3' (every GEMM and the attention on error-compensated fp16 operand pairs, frame level included)."""
    z, meta = helpers.load_golden(name)
    try:
        model = build_model(meta, precision)
        model.engine()
    except NotImplementedError as e:          # f16x3: channel counts that are not whole 32-element runs
        assert precision == "f16x3"
        pytest.skip(str(e))
    dc = meta["duration_control"]
    xs = [torch.from_numpy(z["ids_%d" % i]) for i in range(meta["n"])]
    # per utterance through the reference API
    for i, x in enumerate(xs):
        mel = model.inference(x.cuda(), None, duration_control=dc).cpu().numpy()
        assert mel.shape == z["mel_%d" % i].shape
        assert np.abs(mel - z["mel_%d" % i]).max() <= FP32_MEL_TOL
    # the same utterances as ONE ragged batch: every row must equal its batch-1 result
    out = model.infer_batch(xs, duration_control=dc)
    pk = out["packed"]
    cu_p = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    cu_f = np.concatenate([[0], np.cumsum(pk["olens"])])          # decoder frames (pitch, energy, LengthRegulator map)
    cu_m = np.concatenate([[0], np.cumsum(pk["mel_lens"])])       # mel rows = reduction_factor x those
    r = int(meta["cfg"].get("reduction_factor", 1))
    for i in range(meta["n"]):
        assert pk["mel_lens"][i] == int(z["frames_%d" % i]) == r * pk["olens"][i] == int(out["olens"][i])
        assert np.array_equal(pk["durations_pred"].cpu().numpy()[cu_p[i]:cu_p[i + 1]], z["dur_pred_%d" % i])
        if dc == 1.0:
            assert np.array_equal(pk["durations"].cpu().numpy()[cu_p[i]:cu_p[i + 1]], z["dur_after_lr_%d" % i])
        idx, _ = fs2_oracle.length_regulate_index(fs2_oracle.scale_durations(z["dur_pred_%d" % i], dc))
        assert np.array_equal(pk["lr_index"].cpu().numpy()[cu_f[i]:cu_f[i + 1]], idx)
        T = pk["mel_lens"][i]
        assert np.abs(out["mel"][i, :T].cpu().numpy() - z["mel_%d" % i]).max() <= FP32_MEL_TOL
        assert (out["mel"][i, T:] == 0).all()
        if "mel_before_%d" % i in z.files:      # (before_outs, after_outs) of _forward, network/fastspeech2.py:219-225
            assert np.abs(pk["mel_before"].cpu().numpy()[cu_m[i]:cu_m[i + 1]] - z["mel_before_%d" % i]).max() <= FP32_MEL_TOL
        assert np.abs(pk["enc_out"].cpu().numpy()[cu_p[i]:cu_p[i + 1]] - z["enc_out_%d" % i]).max() <= 1e-3
        assert np.abs(pk["pitch"].cpu().numpy()[cu_f[i]:cu_f[i + 1]] - z["pitch_%d" % i]).max() <= 1e-3
        assert np.abs(pk["energy"].cpu().numpy()[cu_f[i]:cu_f[i + 1]] - z["energy_%d" % i]).max() <= 1e-3


# (the d=64 plain-PosEnc toy config is left out: its sqrt(d)-scaled embeddings and 64-wide LayerNorms
#  amplify bf16 rounding to ~0.6 max-abs on a few frames; it is covered in fp32)
@pytest.mark.parametrize("name", ["tiny_linear_scaled", "tiny_conv3_mixed_ln", "yaml_dims_2x2", "baseline_4x4",
                                  "yaml_shipped_full"])
def test_bf16_teacher_forced_durations(name):
    z, meta = helpers.load_golden(name)
    model = build_model(meta, "bf16")
    eng = model.engine()
    for i in range(meta["n"]):
        x = torch.from_numpy(z["ids_%d" % i]).cuda()
        ds = torch.from_numpy(z["dur_after_lr_%d" % i].astype(np.int32)).cuda()
        out = eng.forward(x, [x.numel()], ds=ds)
        mel = out["mel"].cpu().numpy()
        ref = z["mel_%d" % i]
        assert mel.shape == ref.shape
        print("bf16 %s utt %d: mel max-abs err %.3g, rms %.3g" % (name, i, np.abs(mel - ref).max(), np.sqrt(np.mean((mel - ref) ** 2))))
        assert np.abs(mel - ref).max() <= BF16_MEL_TOL
        assert np.sqrt(np.mean((mel - ref) ** 2)) <= BF16_MEL_RMS_TOL


def test_bf16_with_fp32_encoder_keeps_durations_exact():
    z, meta = helpers.load_golden("baseline_4x4")
    model = build_model(meta, "bf16", encoder_dtype="fp32")
    xs = [torch.from_numpy(z["ids_%d" % i]) for i in range(meta["n"])]
    out = model.infer_batch(xs)
    cu_p = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    for i in range(meta["n"]):
        assert np.array_equal(out["packed"]["durations_pred"].cpu().numpy()[cu_p[i]:cu_p[i + 1]], z["dur_pred_%d" % i])
        T = int(z["frames_%d" % i])
        assert np.abs(out["mel"][i, :T].cpu().numpy() - z["mel_%d" % i]).max() <= BF16_MEL_TOL


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", helpers.LONG_CASES)
def test_long_utterance_config4(name, precision):
    """BASELINE config 4: 512 phones -> 4165 frames at BASELINE dims against the reference's own output (sampled mel
    rows + row / column sums of every frame), alone and as one row of a 4-utterance batch; both fp32-grade modes (exact
    fp32 MFMA, f16x3) to the north_star's fp32 bars, bf16 (teacher-forced durations) to the stated bf16 tolerance."""
    z, meta = helpers.load_golden(name)
    ids = torch.from_numpy(z["ids_0"])
    T = int(z["frames_0"])
    rows = helpers.long_mel_rows(T)
    idx, _ = fs2_oracle.length_regulate_index(z["dur_pred_0"])
    model = build_model(meta, precision)
    # a short utterance first: the cached positional table (1024 rows) has to re-grow for 4165 frames
    assert model.inference(ids[:9].cuda(), None).shape[1] == meta["cfg"]["odim"]
    others = [ids[:300], ids[100:], torch.flip(ids, [0])]
    for xs in ([ids], [ids] + others):
        out = model.infer_batch(xs)
        pk = out["packed"]
        assert pk["olens"][0] == T
        assert np.array_equal(pk["durations_pred"].cpu().numpy()[:512], z["dur_pred_0"])
        assert np.array_equal(pk["durations"].cpu().numpy()[:512], z["dur_after_lr_0"])
        assert np.array_equal(pk["lr_index"].cpu().numpy()[:T], idx)
        mel = out["mel"][0, :T].cpu().numpy()
        assert np.abs(mel[rows] - z["mel_rows_0"]).max() <= FP32_MEL_TOL
        assert np.abs(mel.astype(np.float64).sum(1) - z["mel_rowsum_0"]).max() <= 80 * FP32_MEL_TOL
        assert np.abs(mel.astype(np.float64).sum(0) - z["mel_colsum_0"]).max() <= T * FP32_MEL_TOL
        assert np.abs(pk["enc_out"].cpu().numpy()[:512][::8] - z["enc_out_0"]).max() <= 1e-3
        assert np.abs(pk["pitch"].cpu().numpy()[:T] - z["pitch_0"]).max() <= 1e-3
        assert np.abs(pk["energy"].cpu().numpy()[:T] - z["energy_0"]).max() <= 1e-3
    if precision != "fp32":
        return
    # bf16, durations teacher-forced from the reference
    eng = build_model(meta, "bf16").engine()
    ds = torch.from_numpy(z["dur_after_lr_0"].astype(np.int32)).cuda()
    out = eng.forward(ids.cuda(), [512], ds=ds)
    mel = out["mel"].cpu().numpy()
    assert mel.shape[0] == T
    print("bf16 %s: mel max-abs err %.3g, rms %.3g" % (name, np.abs(mel[rows] - z["mel_rows_0"]).max(),
                                                       np.sqrt(np.mean((mel[rows] - z["mel_rows_0"]) ** 2))))
    assert np.abs(mel[rows] - z["mel_rows_0"]).max() <= BF16_MEL_TOL
    assert np.sqrt(np.mean((mel[rows] - z["mel_rows_0"]) ** 2)) <= BF16_MEL_RMS_TOL
    assert np.array_equal(out["lr_index"].cpu().numpy(), idx)


@pytest.mark.parametrize("precision,tol,flips", [("fp32", FP32_MEL_TOL, 0.001), ("bf16", BF16_MEL_TOL, 0.02)])
def test_config3_eight_shards_equal_one_rank(precision, tol, flips):
    """BASELINE config 3: 256 utterances decoded by 8 ranks (decode(..., rank=r, world_size=8), here one after the
    other on this GPU) against the same 256 decoded by a single rank: the shards partition the set, and every
    utterance has the same durations and mel whichever rank and batch it went through (other batch compositions =
    other tile / split-K shapes = other fp32 summation orders: at most `flips` of the phones may differ by one frame,
    utterances with equal durations agree to the precision's tolerance).  In fp32, sampled utterances are also held
    against the CPU oracle (durations exact, mel <= 1e-3)."""
    from argparse import Namespace
    from fastspeech2_amd import synth
    from fastspeech2_amd.run.fastspeech_decode import decode
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.bench_state_dict()
    model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval().cuda().set_precision(precision)
    utts = {"utt%03d" % i: x for i, x in enumerate(synth.synth_utterances(256, 0, sort=False))}
    cfg = Namespace(DATASET=Namespace(VAL=Namespace(UTTERANCES=utts)), DECODE=Namespace(BATCH_SIZE=32))
    one = decode(cfg, model, 0, None)
    assert sorted(one) == sorted(utts)
    shards = [decode(cfg, model, 0, None, rank=r, world_size=8) for r in range(8)]
    assert all(len(s) == 32 for s in shards)
    union = {}
    for s in shards:
        assert not (set(s) & set(union))
        union.update(s)
    assert sorted(union) == sorted(utts)
    frames = [sum(v["mel_length"] for v in s.values()) for s in shards]
    assert max(frames) <= 1.05 * min(frames)                 # length-sorted round-robin deal balances the ranks
    phones = differing = compared = 0
    for k, a in one.items():
        b = union[k]
        delta = a["durations"] - b["durations"]
        phones += delta.size
        differing += int((delta != 0).sum())
        assert np.abs(delta).max() <= 1
        if (delta != 0).any():
            continue
        assert a["mel_length"] == b["mel_length"]
        assert np.abs(a["mel_postnet"] - b["mel_postnet"]).max() <= tol, k
        compared += 1
    assert differing <= flips * phones, (differing, phones)
    assert compared >= (240 if precision == "fp32" else 100), compared
    if precision == "fp32":
        cfg_o = fs2_oracle.full_cfg(80, 80, use_scaled_pos_enc=True, **kw)
        sd_t = {k: torch.from_numpy(v) for k, v in sd.items()}
        for k in ("utt000", "utt101", "utt255"):
            o = fs2_oracle.infer_one(sd_t, cfg_o, torch.from_numpy(utts[k]))
            assert np.array_equal(union[k]["durations"], o["durations"].numpy())
            assert np.abs(union[k]["mel_postnet"] - o["mel"].numpy()).max() <= FP32_MEL_TOL


@pytest.mark.parametrize("name", ["baseline_4x4", "long_512", "tiny_conv3_mixed_ln", "yaml_dims_2x2", "yaml_shipped_full"])
def test_bf16_with_f16x3_encoder_keeps_durations_exact(name):
    """The throughput mode that keeps the integer results exact: frame-level half in bf16, phone-level half (encoder +
    duration predictor) on error-compensated fp16 GEMM operands (encoder_dtype="f16x3").  Durations, frame counts and
    the LengthRegulator map equal the REFERENCE's on its golden vectors; mel to the bf16 tolerance."""
    z, meta = helpers.load_golden(name)
    model = build_model(meta, "bf16", encoder_dtype="f16x3")
    xs = [torch.from_numpy(z["ids_%d" % i]) for i in range(meta["n"])]
    out = model.infer_batch(xs)
    pk = out["packed"]
    cu_p = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    cu_f = np.concatenate([[0], np.cumsum(pk["olens"])])
    for i in range(meta["n"]):
        assert np.array_equal(pk["durations_pred"].cpu().numpy()[cu_p[i]:cu_p[i + 1]], z["dur_pred_%d" % i])
        T = int(z["frames_%d" % i])
        assert pk["olens"][i] == T
        idx, _ = fs2_oracle.length_regulate_index(z["dur_pred_%d" % i])
        assert np.array_equal(pk["lr_index"].cpu().numpy()[cu_f[i]:cu_f[i + 1]], idx)
        enc_ref = z["enc_out_%d" % i]
        enc = pk["enc_out"].cpu().numpy()[cu_p[i]:cu_p[i + 1]]
        assert np.abs((enc[::8] if name in helpers.LONG_CASES else enc) - enc_ref).max() <= 1e-4
        mel = out["mel"][i, :T].cpu().numpy()
        if name in helpers.LONG_CASES:
            assert np.abs(mel[helpers.long_mel_rows(T)] - z["mel_rows_%d" % i]).max() <= BF16_MEL_TOL
        else:
            assert np.abs(mel - z["mel_%d" % i]).max() <= BF16_MEL_TOL


def test_f16x3_encoder_durations_equal_the_fp32_engine_on_the_bench_batch():
    """BASELINE config 2 batch (32 utterances, 3767 phones): the f16x3 phone-level half against the exact-fp32 engine
    and the CPU oracle — every duration equal (a pure-bf16 encoder flips ~1 % of them)."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.bench_state_dict()
    xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
    durs = {}
    for enc in ("fp32", "f16x3", "bf16"):
        model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval().cuda().set_precision("bf16", enc)
        pk = model.infer_batch(xs)["packed"]
        durs[enc] = pk["durations"].cpu().numpy()
        if enc == "f16x3":
            logd = pk["log_durations"].cpu().numpy()
    assert np.array_equal(durs["f16x3"], durs["fp32"])
    assert 0 < (durs["bf16"] != durs["fp32"]).sum() < 0.05 * durs["fp32"].size
    cfg = fs2_oracle.full_cfg(80, 80, use_scaled_pos_enc=True, **kw)
    sd_t = {k: torch.from_numpy(v) for k, v in sd.items()}
    cu = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    worst = 0.0
    for b in (0, 9, 20, 31):
        o = fs2_oracle.infer_one(sd_t, cfg, xs[b])
        assert np.array_equal(durs["f16x3"][cu[b]:cu[b + 1]], o["durations"].numpy())
        worst = max(worst, np.abs(logd[cu[b]:cu[b + 1]] - o["log_durations"].numpy()).max())
    assert worst <= 5e-6, worst          # fp32 summation-order noise level (measured 1e-6; bf16 encoder: 4e-3)


def test_f16x3_everywhere_is_fp32_grade_on_the_bench_batch():
    """set_precision("f16x3"): frame-level half on error-compensated fp16 operands as well.  BASELINE config 2 batch against
    the exact-fp32 engine: every duration equal, mel / pitch / energy at fp32 summation-noise level (the north_star's fp32
    bar is 1e-3 on mel; a bf16 frame-level half measures 1e-2), and against the CPU oracle on four utterances."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.bench_state_dict()
    xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
    res = {}
    for prec in ("fp32", "f16x3"):
        model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval().cuda().set_precision(prec)
        pk = model.infer_batch(xs)["packed"]
        res[prec] = {k: pk[k].float().cpu().numpy() for k in ("durations", "mel", "pitch", "energy")}
        res[prec]["olens"] = list(pk["olens"])
    assert np.array_equal(res["fp32"]["durations"], res["f16x3"]["durations"])
    assert res["fp32"]["olens"] == res["f16x3"]["olens"]
    worst = {k: float(np.abs(res["fp32"][k] - res["f16x3"][k]).max()) for k in ("mel", "pitch", "energy")}
    print("f16x3 everywhere vs exact fp32 on the bench batch:", worst)
    assert max(worst.values()) <= 5e-5, worst          # measured 8-9e-6
    cfg = fs2_oracle.full_cfg(80, 80, use_scaled_pos_enc=True, **kw)
    sd_t = {k: torch.from_numpy(v) for k, v in sd.items()}
    cu = np.concatenate([[0], np.cumsum(res["f16x3"]["olens"])])
    for b in (0, 13, 31):
        o = fs2_oracle.infer_one(sd_t, cfg, xs[b])
        assert np.abs(res["f16x3"]["mel"][cu[b]:cu[b + 1]] - o["mel"].numpy()).max() <= FP32_MEL_TOL


@pytest.mark.parametrize("n_phones", [120, 37, 450, (200, 33, 7)])
def test_single_utterance_latency_schedule_equals_the_batch_schedule(n_phones):
    """BASELINE config 1 (one utterance): with a few hundred tokens the engine runs the ReLU convs split-K and applies
    the ReLU to the summed partials (Engine.latency_split, fs2_conv_gemm_ksplit_act / fs2_sum_act_layernorm).  Same
    durations as the schedule without it and as the CPU oracle; log-durations to fp32 summation noise; mel to the bf16
    tolerance against the oracle."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.bench_state_dict()
    lens = n_phones if isinstance(n_phones, tuple) else (n_phones,)        # (a small ragged batch takes the same schedule)
    n_phones = sum(lens)
    rng = np.random.Generator(np.random.PCG64(n_phones))
    utts = [torch.from_numpy(rng.integers(1, synth.BENCH_IDIM, size=n)) for n in lens]
    ids = utts[0]
    model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval().cuda().set_precision("bf16", "f16x3")
    from fastspeech2_amd import hip
    eng = model.engine()
    assert eng.latency_split
    assert hip.load().fs2_conv_gemm_ksplit_act(n_phones, 1024, 256, 9, hip.F16X3, 3.5) > 1      # the path under test is taken
    res = {}
    for on in (True, False):
        eng.latency_split = on
        pk = model.infer_batch(utts)["packed"]
        res[on] = (pk["durations_pred"].cpu().numpy(), pk["log_durations"].cpu().numpy(), pk["mel"].float().cpu().numpy())
    eng.latency_split = True
    assert np.array_equal(res[True][0], res[False][0])
    assert np.abs(res[True][1] - res[False][1]).max() <= 5e-6
    assert res[True][2].shape == res[False][2].shape
    cfg = fs2_oracle.full_cfg(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    o = fs2_oracle.infer_one({k: torch.from_numpy(v) for k, v in sd.items()}, cfg, ids)      # first utterance
    n0, t0 = lens[0], int(o["mel"].shape[0])
    assert np.array_equal(res[True][0][:n0], o["durations"].numpy())
    assert np.abs(res[True][1][:n0] - o["log_durations"].numpy()).max() <= 5e-6
    for on in (True, False):
        assert np.abs(res[on][2][:t0] - o["mel"].numpy()).max() <= BF16_MEL_TOL, on


@pytest.mark.parametrize("variant", ["small_ln_outputs", "large_ffn_hidden"])
def test_f16x3_encoder_on_rescaled_weights(variant):
    """The f16x3 mode splits activations UNSCALED into fp16 pairs (include/fs2_hip.h): operands far from 1 lose low-half
    bits (|x| < ~1e-3: the low half goes subnormal) or overflow (|x| > 65504).  Two function-preserving rescalings of the
    bench model push the phone-level operands towards those edges: LayerNorm outputs 64 x smaller with the weights that
    read them 64 x larger, and FFN hidden activations 16 x larger with the second conv 16 x smaller.  Durations must still
    equal the exact-fp32 engine's on the bench batch; the log-duration distance is printed (parity on real checkpoints
    remains unpinned: there are none here)."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = {k: v.copy() for k, v in synth.bench_state_dict().items()}
    for li in range(kw["elayers"]):
        p = "encoder.encoders.%d." % li
        if variant == "small_ln_outputs":
            # x = norm(...) feeds the next block's q / k / v projections and FFN conv 1 (and the residual stream, which the
            # next LayerNorm renormalises): scale norm2's affine down and every reader of the block input up
            for n in ("norm1", "norm2"):
                sd[p + n + ".weight"] /= 64.0
                sd[p + n + ".bias"] /= 64.0
            sd[p + "feed_forward.w1.weight"] *= 64.0      # reads norm1's output (64 x smaller)
            sd[p + "feed_forward.w2.weight"] /= 64.0      # its output joins a 64 x smaller residual stream
            sd[p + "feed_forward.w2.bias"] /= 64.0
            if li > 0:          # block input 64 x smaller (block 0 reads the embedding, left as it is)
                for n in "qkv":
                    sd[p + "self_atten.linear_%s.weight" % n] *= 64.0
                sd[p + "self_atten.linear_out.weight"] /= 64.0
                sd[p + "self_atten.linear_out.bias"] /= 64.0
        else:
            sd[p + "feed_forward.w1.weight"] *= 16.0
            sd[p + "feed_forward.w1.bias"] *= 16.0
            sd[p + "feed_forward.w2.weight"] /= 16.0
    if variant == "small_ln_outputs":      # the duration predictor reads the 64 x smaller encoder output
        sd["variance_adaptor.duration_predictor.conv.0.0.conv.weight"] *= 64.0
    xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
    res = {}
    for enc in ("fp32", "f16x3"):
        model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval().cuda().set_precision("bf16", enc)
        pk = model.infer_batch(xs)["packed"]
        res[enc] = (pk["durations_pred"].cpu().numpy(), pk["log_durations"].cpu().numpy())
    differ = int((res["fp32"][0] != res["f16x3"][0]).sum())
    dist = float(np.abs(res["fp32"][1] - res["f16x3"][1]).max())
    print("f16x3 on %s: %d of %d durations differ from the fp32 engine, max |logd difference| %.2e"
          % (variant, differ, res["fp32"][0].size, dist))
    assert np.isfinite(res["f16x3"][1]).all()
    assert differ == 0 and dist <= 2e-5


@pytest.mark.parametrize("precision", [("fp32", None), ("bf16", "f16x3")])
@pytest.mark.parametrize("name", sorted(helpers.DURATION_CASES))
def test_durations_of_the_full_workloads_equal_the_reference(name, precision):
    """BASELINE config 2 (bench.py's 32 utterances, 3 767 phones) and config 3 (256 utterances, 30 727 phones, as eight
    batches of 32): the durations the HIP path predicts against the REFERENCE's own, phone by phone (fixtures from
    tests/golden/make_golden.py:run_durations_case), in the exact-fp32 mode and in the bench's headline mode (bf16
    frame level, f16x3 phone level).  Equal on every phone whose rounding margin exceeds MARGIN_FLOOR; the phones below
    it are counted and may differ by one frame at most."""
    from fastspeech2_amd import synth
    utts, logd_ref, dur_ref, margin, frames_ref, meta = helpers.duration_case(name)
    kw = dict(synth.BASELINE_OVERRIDES)
    model = FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()})
    model.eval().cuda().set_precision(*precision)
    dur, logd, frames = [], [], []
    for b0 in range(0, len(utts), 32):
        pk = model.infer_batch([torch.from_numpy(x) for x in utts[b0:b0 + 32]])["packed"]
        dur.append(pk["durations_pred"].cpu().numpy().astype(np.int64))
        logd.append(pk["log_durations"].cpu().numpy())
        frames += [int(v) for v in pk["olens"]]
    dur, logd = np.concatenate(dur), np.concatenate(logd)
    assert dur.shape == dur_ref.shape
    safe = margin > helpers.MARGIN_FLOOR
    differing = int((dur != dur_ref).sum())
    print("%s %s: %d phones, %d below the margin floor, %d durations differ, max |logd - ref| %.2e"
          % (name, precision, dur.size, int((~safe).sum()), differing, np.abs(logd - logd_ref).max()))
    assert np.array_equal(dur[safe], dur_ref[safe])
    assert np.abs(dur - dur_ref).max() <= 1
    assert np.abs(logd - logd_ref).max() <= 2e-5
    if differing == 0:
        assert frames == frames_ref.tolist()


@pytest.mark.parametrize("name", helpers.TRAIN_CASES)
def test_training_forward_loss_matches_reference(name):
    z, meta = helpers.load_golden(name)
    model = build_model(meta)
    t = lambda k: torch.from_numpy(z[k])
    loss = model(t("xs"), t("ilens"), t("ys"), t("olens"), t("ds"), t("ps"), t("es"))
    obs = model.reporter.observation
    for k in ("l1_loss", "duration_loss", "pitch_loss", "energy_loss", "loss"):
        assert abs(obs[k] - meta["losses"][k]) <= 1e-4 * max(1.0, abs(meta["losses"][k])), k
    assert abs(float(loss) - meta["loss"]) <= 1e-4 * abs(meta["loss"])
    # _forward (non-inference) hands back padded (before, after, d_outs, p_outs, e_outs)
    outs = model._forward(t("xs"), t("ilens"), t("olens"), t("ds"), t("ps"), t("es"))
    assert outs[0].shape == (z["xs"].shape[0], int(z["olens"].max()), meta["cfg"]["odim"])
    assert outs[2].shape == tuple(z["xs"].shape) and outs[3].shape[-1] == 4 and outs[4].shape[-1] == 1


def test_ragged_batch_vs_oracle_and_attention_dump():
    """Seeded inputs beyond the goldens: the HIP path vs the CPU oracle on a ragged batch, plus
    calculate_all_attentions against the oracle's softmax matrices."""
    z, meta = helpers.load_golden("tiny_conv9_plain")
    model = build_model(meta)
    sd, cfg = helpers.golden_state_dict(meta), helpers.oracle_cfg(meta)
    rng = np.random.Generator(np.random.PCG64(99))
    xs = [torch.from_numpy(rng.integers(1, cfg["idim"], size=L)) for L in (13, 2, 29, 8)]
    out = model.infer_batch(xs)
    for i, x in enumerate(xs):
        o = fs2_oracle.infer_one(sd, cfg, x)
        T = o["mel"].shape[0]
        assert out["olens"][i] == T
        assert torch.equal(out["durations"][i, :len(x)].cpu(), o["durations"])
        assert (out["mel"][i, :T].cpu() - o["mel"]).abs().max().item() <= FP32_MEL_TOL
    # teacher-forced attention dump
    ils = [len(x) for x in xs]
    ds = [torch.from_numpy(rng.integers(0, 4, size=L)) for L in ils]
    for dd in ds:
        dd[0] += 1
    ols = [int(dd.sum()) for dd in ds]
    B, Tp, Tm = len(xs), max(ils), max(ols)
    pad = lambda lst, T, shape=(): torch.stack([torch.cat([v, v.new_zeros((T - v.shape[0],) + shape)]) for v in lst])
    ps = [torch.from_numpy(rng.standard_normal((L, 4)).astype(np.float32)) for L in ols]
    es = [torch.from_numpy(rng.standard_normal((L, 1)).astype(np.float32)) for L in ols]
    att = model.calculate_all_attentions(pad(xs, Tp), torch.tensor(ils), torch.zeros(B, Tm, cfg["odim"]),
                                         torch.tensor(ols), pad(ds, Tp), pad(ps, Tm, (4,)), pad(es, Tm, (1,)))
    assert set(att) == {"encoder.encoders.0.self_atten", "encoder.encoders.1.self_atten",
                        "decoder.encoders.0.self_atten", "decoder.encoders.1.self_atten", "predicted_fbank"}
    for b in range(B):
        ea, da, mel = fs2_oracle.attentions_one(sd, cfg, xs[b], ds[b], ps[b], es[b])
        for li in range(2):
            assert np.abs(att["encoder.encoders.%d.self_atten" % li][b] - ea[li].numpy()).max() <= 1e-4
            assert np.abs(att["decoder.encoders.%d.self_atten" % li][b] - da[li].numpy()).max() <= 1e-4
        assert np.abs(att["predicted_fbank"][b] - mel.numpy().T).max() <= FP32_MEL_TOL


def test_full_size_properties_bf16():
    """BASELINE config-2 size (32 length-bucketed utterances, BASELINE dims): size-independent
    properties — batch result == per-utterance result for sampled rows, frame counts == sum of
    durations, LengthRegulator index map monotone and consistent with the durations."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.synth_state_dict(80, 80, 1234, dur_bias=float(np.log(7.67)), dur_w_scale=0.15, **kw)
    model = FeedForwardTransformer(80, 80, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval().cuda().set_precision("bf16")
    xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
    out = model.infer_batch(xs)
    pk = out["packed"]
    dur = pk["durations"].cpu().numpy()
    cu_p = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    cu_f = np.concatenate([[0], np.cumsum(pk["olens"])])
    lri = pk["lr_index"].cpu().numpy()
    for b in range(32):
        d_b = dur[cu_p[b]:cu_p[b + 1]]
        assert pk["olens"][b] == int(d_b.sum())
        idx, _ = fs2_oracle.length_regulate_index(d_b)
        assert np.array_equal(lri[cu_f[b]:cu_f[b + 1]], idx)
    assert torch.isfinite(out["mel"]).all()
    # batch row == the same utterance alone.  Not bit-equal: a 1-utterance launch picks other tile /
    # split-K shapes, i.e. another fp32 summation order, and in bf16 that can flip activation roundings and, with a
    # bf16 phone-level half, a duration (then the frames shift and the row cannot be compared).
    compared = 0
    for b in (0, 13, 31):
        alone = model.infer_batch([xs[b]])
        if np.array_equal(alone["packed"]["durations"].cpu().numpy(), dur[cu_p[b]:cu_p[b + 1]]):
            T = pk["olens"][b]
            assert (alone["mel"][0, :T] - out["mel"][b, :T]).abs().max().item() <= BF16_MEL_TOL
            compared += 1
    # the throughput mode with exact durations (f16x3 phone-level half): every sampled row MUST be comparable
    model.set_precision("bf16", "f16x3")
    out = model.infer_batch(xs)
    pk = out["packed"]
    dur = pk["durations"].cpu().numpy()
    for b in (0, 13, 31):
        alone = model.infer_batch([xs[b]])
        assert np.array_equal(alone["packed"]["durations"].cpu().numpy(), dur[cu_p[b]:cu_p[b + 1]])
        T = pk["olens"][b]
        assert alone["olens"][0] == T
        assert (alone["mel"][0, :T] - out["mel"][b, :T]).abs().max().item() <= BF16_MEL_TOL


@pytest.mark.parametrize("precision,tol,flips", [("bf16", BF16_MEL_TOL, 0.02), ("fp32", FP32_MEL_TOL, 0.001)])
def test_sub_batches_agree_with_the_full_batch(precision, tol, flips):
    """BASELINE dims: the 32-utterance batch against the same utterances in groups of 4 / 8 / 16.  The token counts of the
    groups (≈3 k / 6 k / 13 k frames) make the launch rules pick other split-K factors and tile heights
    (fs2_conv_gemm_ksplit's cost model, 96-row tiles), i.e. other fp32 summation orders.  A duration is
    round(exp(x) - 1), so a last-bit difference in x can flip one that sits on a rounding edge: at most `flips` of the
    phones may differ, by one frame; utterances whose durations all agree must agree in the mel to the precision's
    tolerance (an utterance with a flipped duration has shifted frames and is not compared)."""
    from fastspeech2_amd import synth
    kw = dict(synth.BASELINE_OVERRIDES)
    sd = synth.synth_state_dict(80, 80, 1234, dur_bias=float(np.log(7.67)), dur_w_scale=0.15, **kw)
    model = FeedForwardTransformer(80, 80, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval().cuda().set_precision(precision)
    xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
    full = model.infer_batch(xs)
    dur = full["packed"]["durations"].cpu().numpy()
    cu_p = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    compared = phones = differing = 0
    for group in (4, 8, 16):
        for g0 in range(0, 32, group):
            part = model.infer_batch(xs[g0:g0 + group])
            pd = part["packed"]["durations"].cpu().numpy()
            cu_q = np.concatenate([[0], np.cumsum([len(x) for x in xs[g0:g0 + group]])])
            for j in range(group):
                b = g0 + j
                delta = pd[cu_q[j]:cu_q[j + 1]].astype(np.int64) - dur[cu_p[b]:cu_p[b + 1]].astype(np.int64)
                phones += delta.size
                differing += int((delta != 0).sum())
                assert np.abs(delta).max() <= 1
                if (delta != 0).any():
                    continue
                T = full["packed"]["olens"][b]
                assert part["packed"]["olens"][j] == T
                assert (part["mel"][j, :T] - full["mel"][b, :T]).abs().max().item() <= tol, (group, b)
                compared += 1
    assert differing <= flips * phones, (differing, phones)
    assert compared >= (90 if precision == "fp32" else 24), compared


def test_decode_entry_point():
    """run/fastspeech_decode.py:decode counterpart: eval+no_grad, controls, 8 fields, restores train()."""
    from argparse import Namespace
    from fastspeech2_amd.run.fastspeech_decode import decode
    z, meta = helpers.load_golden("tiny_linear_scaled")
    model = build_model(meta)
    model.train()
    utts = {"utt%d" % i: z["ids_%d" % i] for i in range(meta["n"])}
    cfg = Namespace(DATASET=Namespace(VAL=Namespace(UTTERANCES=utts)), DECODE=Namespace(BATCH_SIZE=2))
    res = decode(cfg, model, 0, None)
    assert model.training
    for i in range(meta["n"]):
        r = res["utt%d" % i]
        assert set(r) == {"mel", "mel_postnet", "durations", "pitch", "energy", "src_mask", "mel_mask", "mel_length"}
        assert r["mel_length"] == int(z["frames_%d" % i])
        assert np.array_equal(r["durations"], z["dur_after_lr_%d" % i])
        assert np.abs(r["mel"] - z["mel_%d" % i]).max() <= FP32_MEL_TOL
    # two "ranks" partition the set
    r0 = decode(cfg, model, 0, None, rank=0, world_size=2)
    r1 = decode(cfg, model, 0, None, rank=1, world_size=2)
    assert sorted(list(r0) + list(r1)) == sorted(utts)


@pytest.mark.parametrize("name", ["reduction2_tiny", "reduction3_postnet"])
def test_reduction_factor_decode_and_refusals(name):
    """reduction_factor r > 1 (network/fastspeech2.py:219: feat_out gives r mel frames per decoder frame): the decode
    driver returns r x sum(durations) mel rows per utterance, equal to the reference's; pitch / energy stay on the
    decoder frames.  Teacher-forced / training calls and graph capture are refused."""
    from argparse import Namespace
    from fastspeech2_amd.run.fastspeech_decode import decode
    z, meta = helpers.load_golden(name)
    r = meta["cfg"]["reduction_factor"]
    model = build_model(meta)
    utts = {"utt%d" % i: z["ids_%d" % i] for i in range(meta["n"])}
    cfg = Namespace(DATASET=Namespace(VAL=Namespace(UTTERANCES=utts)), DECODE=Namespace(BATCH_SIZE=2))
    res = decode(cfg, model, 0, None)
    for i in range(meta["n"]):
        u = res["utt%d" % i]
        assert u["mel_length"] == int(z["frames_%d" % i]) == r * int(u["durations"].sum())
        assert u["pitch"].shape[0] * r == u["mel_length"] and u["mel_mask"].shape[0] == u["mel_length"]
        assert np.abs(u["mel_postnet"] - z["mel_%d" % i]).max() <= FP32_MEL_TOL
        assert np.abs(u["pitch"] - z["pitch_%d" % i]).max() <= 1e-3
    il = torch.tensor([len(z["ids_0"])])
    with pytest.raises(NotImplementedError):
        model._forward(torch.from_numpy(z["ids_0"]).unsqueeze(0), il, olens=torch.tensor([4]), ds=torch.ones(1, len(z["ids_0"])),
                       ps=torch.zeros(1, 4, 4), es=torch.zeros(1, 4, 1))
    with pytest.raises(NotImplementedError):
        model.enable_inference_graphs()


def test_pipelined_forward_equals_serial():
    """Engine.forward_pipelined (phone phase of batch k+1 on a side stream) gives the serial results."""
    z, meta = helpers.load_golden("tiny_conv3_mixed_ln")
    model = build_model(meta)
    eng = model.engine()
    rng = np.random.Generator(np.random.PCG64(4))
    batches = []
    for k in range(5):
        xs = [rng.integers(1, meta["cfg"]["idim"], size=int(L)) for L in rng.integers(3, 40, size=3)]
        batches.append((torch.from_numpy(np.concatenate(xs)).cuda(), [len(x) for x in xs]))
    serial = [eng.forward(ids, il) for ids, il in batches]
    for fs in (1, 2, 3):               # frame phases on one stream / alternating over 2 or 3 streams
        piped = list(eng.forward_pipelined(batches, frame_streams=fs))
        assert len(piped) == len(serial)
        for a, b in zip(serial, piped):
            if fs > 1:
                b["ready"].synchronize()
            assert a["olens"] == b["olens"]
            assert torch.equal(a["durations"], b["durations"])
            assert torch.equal(a["mel"], b["mel"])
    # phone-level halves of 2 / 3 consecutive batches run as one ragged batch (other launch shapes: fp32 summation
    # order may differ, the integer results may not)
    for group in (2, 3):
        piped = list(eng.forward_pipelined(batches, frame_streams=2, phone_group=group))
        assert len(piped) == len(serial)
        for a, b in zip(serial, piped):
            b["ready"].synchronize()
            assert a["olens"] == b["olens"]
            assert torch.equal(a["durations"], b["durations"]) and torch.equal(a["lr_index"], b["lr_index"])
            assert torch.equal(a["cu_frames"], b["cu_frames"]) and torch.equal(a["cu_phone"], b["cu_phone"])
            assert (a["mel"] - b["mel"]).abs().max().item() <= 1e-4
            assert (a["enc_out"] - b["enc_out"]).abs().max().item() <= 1e-4


def test_graphed_inference_is_bit_identical_to_the_eager_forward():
    """`enable_inference_graphs()`: batch-1 inference replayed from HIP graphs (one per phone count, one per frame
    count) - same kernels and arguments as the eager schedule, so the mel must be equal bit for bit, on first sight of a
    shape (eager + capture), on replays, with another utterance of the same shape, and across shapes."""
    z, meta = helpers.load_golden("baseline_4x4")
    model = build_model(meta, "bf16", "f16x3")
    rng = np.random.Generator(np.random.PCG64(5))
    idim = meta["cfg"]["idim"]
    utts = [torch.from_numpy(rng.integers(1, idim, size=n)) for n in (37, 90, 37, 90, 37)]
    with torch.no_grad():
        want = [model.inference(u).cpu() for u in utts]
        model.enable_inference_graphs(8)
        for rnd in range(3):
            for u, w in zip(utts, want):
                got = model.inference(u).cpu()
                assert got.shape == w.shape and torch.equal(got, w), (rnd, u.shape)
        gi = model._graphed
        assert gi.hits >= 8 and len(gi.phone) == 2
        # the controls are part of the graph key
        a = model.inference(utts[0], duration_control=1.3).cpu()
        model.enable_inference_graphs(0)
        b = model.inference(utts[0], duration_control=1.3).cpu()
        assert torch.equal(a, b)
