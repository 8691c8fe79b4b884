"""Pins the CPU oracle (oracle/fs2_oracle.py) against the golden vectors produced by the
repaired reference (tests/golden/make_golden.py).  Bar: mel <= 1e-5 max-abs, durations and
LengthRegulator outputs exact (BASELINE.md acceptance)."""
import numpy as np
import pytest
import torch

from oracle import fs2_oracle
from tests import helpers

MEL_TOL = 1e-5
# the BASELINE-size stack accumulates more fp32 reassociation noise (8 blocks, K up to 9216)
MEL_TOL_BIG = 5e-5


@pytest.mark.parametrize("name", helpers.INFER_CASES)
def test_oracle_matches_reference_inference(name):
    z, meta = helpers.load_golden(name)
    sd = helpers.golden_state_dict(meta)
    cfg = helpers.oracle_cfg(meta)
    tol = MEL_TOL_BIG if name in ("baseline_4x4", "yaml_shipped_full") else MEL_TOL
    for i in range(meta["n"]):
        x = torch.from_numpy(z["ids_%d" % i])
        out = fs2_oracle.infer_one(sd, cfg, x, duration_control=meta["duration_control"])
        assert np.array_equal(out["durations_pred"].numpy(), z["dur_pred_%d" % i])
        assert np.array_equal(out["durations_reported"].numpy(), z["dur_after_lr_%d" % i])
        assert out["mel"].shape[0] == int(z["frames_%d" % i])
        assert np.abs(out["enc_out"].numpy() - z["enc_out_%d" % i]).max() <= tol
        assert np.abs(out["log_durations"].numpy() - z["logdur_%d" % i]).max() <= tol
        assert np.abs(out["pitch"].numpy() - z["pitch_%d" % i]).max() <= tol
        assert np.abs(out["energy"].numpy() - z["energy_%d" % i]).max() <= tol
        assert np.abs(out["mel"].numpy() - z["mel_%d" % i]).max() <= tol
        if "mel_before_%d" % i in z.files:
            assert np.abs(out["mel_before"].numpy() - z["mel_before_%d" % i]).max() <= tol


@pytest.mark.parametrize("name", helpers.LONG_CASES)
def test_oracle_matches_reference_long_utterance(name):
    """BASELINE config 4 (512 phones -> 4165 frames): sampled mel rows, row / column sums of all frames."""
    z, meta = helpers.load_golden(name)
    sd, cfg = helpers.golden_state_dict(meta), helpers.oracle_cfg(meta)
    out = fs2_oracle.infer_one(sd, cfg, torch.from_numpy(z["ids_0"]))
    assert np.array_equal(out["durations_pred"].numpy(), z["dur_pred_0"])
    assert np.array_equal(out["durations_reported"].numpy(), z["dur_after_lr_0"])
    T = int(z["frames_0"])
    mel = out["mel"].numpy()
    assert mel.shape[0] == T
    assert np.abs(out["enc_out"].numpy()[::8] - z["enc_out_0"]).max() <= MEL_TOL_BIG
    assert np.abs(out["pitch"].numpy() - z["pitch_0"]).max() <= MEL_TOL_BIG
    assert np.abs(out["energy"].numpy() - z["energy_0"]).max() <= MEL_TOL_BIG
    assert np.abs(mel[helpers.long_mel_rows(T)] - z["mel_rows_0"]).max() <= MEL_TOL_BIG
    assert np.abs(mel.astype(np.float64).sum(1) - z["mel_rowsum_0"]).max() <= 80 * MEL_TOL_BIG
    assert np.abs(mel.astype(np.float64).sum(0) - z["mel_colsum_0"]).max() <= T * MEL_TOL_BIG


@pytest.mark.parametrize("name", sorted(helpers.DURATION_CASES))
def test_oracle_durations_of_the_full_workloads_equal_the_reference(name):
    """BASELINE configs 2 and 3, every utterance (32 / 256; 3 767 / 30 727 phones): the oracle's encoder + duration
    predictor against the reference's own per-phone outputs (network/variance_adaptor.py:117-138).  Durations must be
    equal on every phone whose rounding margin exceeds MARGIN_FLOOR; the count below it is asserted, not hidden."""
    utts, logd_ref, dur_ref, margin, frames_ref, meta = helpers.duration_case(name)
    _, sd, cfg = helpers.bench_cfg_and_weights()
    logd, dur = [], []
    with torch.no_grad():
        for x in utts:
            hs = fs2_oracle.encode(sd, cfg, torch.from_numpy(x))
            ld = fs2_oracle.duration_log(hs, sd)
            logd.append(ld.numpy())
            dur.append(fs2_oracle.duration_round(ld).numpy())
    logd, dur = np.concatenate(logd), np.concatenate(dur)
    assert np.abs(logd - logd_ref).max() <= MEL_TOL_BIG
    safe = margin > helpers.MARGIN_FLOOR
    assert int((~safe).sum()) == meta["below_1e5"] <= 1
    assert np.array_equal(dur[safe], dur_ref[safe])
    assert np.abs(dur - dur_ref).max() <= 1                 # the phones below the floor may flip by one frame, no more
    cu = np.concatenate([[0], np.cumsum([len(x) for x in utts])])
    if safe.all():
        assert [int(dur[cu[i]:cu[i + 1]].sum()) for i in range(len(utts))] == frames_ref.tolist()


def test_oracle_matches_reference_baseline_size_training_step():
    """BASELINE config 5 shapes (B=32, 120 phones -> 781 frames each, 4+4 blocks): loss terms and gradient samples."""
    z, meta = helpers.load_golden(helpers.BIG_TRAIN_CASES[0])
    sd, cfg = helpers.golden_state_dict(meta), helpers.oracle_cfg(meta)
    batch = helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])
    assert np.array_equal(batch[4], z["ds"])
    terms, grads = fs2_oracle.train_grads(sd, cfg, *[torch.from_numpy(a) for a in batch])
    for k in ("l1_loss", "duration_loss", "pitch_loss", "energy_loss", "loss"):
        assert abs(terms[k] - meta["losses"][k]) <= 2e-5 * max(1.0, abs(meta["losses"][k])), k
    for k in meta["grad_names"]:
        want, got, wstat, gstat = helpers.golden_grad_sample(z, meta, k, grads[k])
        assert np.abs(want - got).max() <= 1e-4 * max(np.abs(want).max(), 1e-2), k
        assert abs(wstat[1] - gstat[1]) <= 1e-4 * max(wstat[1], 1e-2), k


@pytest.mark.parametrize("name", helpers.TRAIN_CASES)
def test_oracle_matches_reference_loss(name):
    z, meta = helpers.load_golden(name)
    sd = helpers.golden_state_dict(meta)
    cfg = helpers.oracle_cfg(meta)
    t = lambda k: torch.from_numpy(z[k])
    got = fs2_oracle.train_loss(sd, cfg, t("xs"), t("ilens"), t("ys"), t("olens"), t("ds"), t("ps"), t("es"))
    for k in ("l1_loss", "duration_loss", "pitch_loss", "energy_loss", "loss"):
        assert abs(got[k] - meta["losses"][k]) <= 2e-5 * max(1.0, abs(meta["losses"][k])), k


@pytest.mark.parametrize("name", helpers.GRAD_CASES)
def test_oracle_gradients_match_reference_backward(name):
    """d loss / d parameter of the restatement (autograd) vs what the repaired reference's
    loss.backward() left in .grad; tolerance relative to the largest gradient entry of the tensor
    (plus a floor for tensors whose true gradient is zero, e.g. linear_k.bias)."""
    z, meta = helpers.load_golden(name)
    sd = helpers.golden_state_dict(meta)
    cfg = helpers.oracle_cfg(meta)
    t = lambda k: torch.from_numpy(z[k])
    _, grads = fs2_oracle.train_grads(sd, cfg, t("xs"), t("ilens"), t("ys"), t("olens"), t("ds"), t("ps"), t("es"))
    assert sorted(grads) == sorted(meta["grad_names"])
    for k in meta["grad_names"]:
        want, got, wstat, gstat = helpers.golden_grad_sample(z, meta, k, grads[k])
        tol = 1e-4 * max(np.abs(want).max(), 1e-2)
        assert np.abs(want - got).max() <= tol, k
        assert abs(wstat[1] - gstat[1]) <= 1e-4 * max(wstat[1], 1e-2), k


def test_length_regulator_integer_rules():
    # zero-duration phones are skipped; an all-zero row becomes all ones (D9)
    idx, d = fs2_oracle.length_regulate_index(np.array([2, 0, 3, 1]))
    assert idx.tolist() == [0, 0, 2, 2, 2, 3] and d.tolist() == [2, 0, 3, 1]
    idx, d = fs2_oracle.length_regulate_index(np.array([0, 0, 0]))
    assert idx.tolist() == [0, 1, 2] and d.tolist() == [1, 1, 1]
    # duration_control: fp32 multiply then round-half-to-even (torch.round)
    assert fs2_oracle.scale_durations(np.array([1, 2, 3, 5]), 0.5).tolist() == [0, 1, 2, 2]
    assert fs2_oracle.scale_durations(np.array([1, 2, 3]), 1.0).tolist() == [1, 2, 3]


def test_duration_round_half_even():
    x = torch.log(torch.tensor([1.5, 2.5, 3.5, 0.2, 1.0]))
    assert fs2_oracle.duration_round(x).tolist() == [0, 2, 2, 0, 0]
