"""Shared test plumbing: golden fixture loading and synthetic-weight state dicts."""
import glob
import json
import os

import numpy as np
import torch

from fastspeech2_amd import synth
from oracle import fs2_oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

LONG_CASES = ["long_512"]            # BASELINE config 4: 512 phones -> ~4000 frames, mel stored as sampled rows + sums
BIG_TRAIN_CASES = ["train_baseline_b32"]   # BASELINE config 5 shapes: inputs regenerated from the seed (train_inputs)
INFER_CASES = ["tiny_conv9_plain", "tiny_linear_scaled", "tiny_conv3_mixed_ln", "yaml_dims_2x2",
               "d9_zero_durations", "duration_control_0p8", "duration_control_1p3", "baseline_4x4",
               "postnet_tiny_bn", "postnet_tiny_nobn", "yaml_shipped_full",
               "reduction2_tiny", "reduction3_postnet"]      # reduction_factor 2 / 3 (the latter through the PostNet)
TRAIN_CASES = ["train_tiny_conv9", "train_tiny_linear", "train_postnet_tiny", "train_tiny_conv3_mixed"]
GRAD_CASES = ["train_tiny_conv9", "train_tiny_linear", "train_tiny_conv3_mixed"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def all_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def split_cfg(cfg):
    kw = dict(cfg)
    idim, odim = kw.pop("idim"), kw.pop("odim")
    return idim, odim, kw


def golden_state_dict(meta):
    idim, odim, kw = split_cfg(meta["cfg"])
    sd = synth.synth_state_dict(idim, odim, meta["seed"], dur_bias=meta["dur_bias"],
                                alpha=meta.get("alpha", 1.0), dur_w_scale=meta.get("dur_w_scale", 1.0), **kw)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def oracle_cfg(meta):
    idim, odim, kw = split_cfg(meta["cfg"])
    return fs2_oracle.full_cfg(idim, odim, **kw)


def golden_grad_sample(z, meta, name, grad: torch.Tensor):
    """(expected sample, got sample, expected (sum, l2)) for one parameter of a GRAD_CASES fixture."""
    j = meta["grad_names"].index(name)
    g = grad.detach().reshape(-1).double().cpu()
    step = max(1, -(-g.numel() // meta["grad_sample"]))
    return z["grad_%d" % j], g[::step].float().numpy(), z["grad_stats"][j], np.array([float(g.sum()), float(g.norm())])


def train_inputs(cfg, seed, B, Tp, ds_lo=0, ds_hi=5):
    """Equal-length teacher-forced batch (no padding, every row sums to the same frame count) from numpy PCG64:
    (xs, ilens, ys, olens, ds, ps, es) as numpy arrays.  Shared by tests/golden/make_golden.py and the tests that
    regenerate the inputs of a fixture too large to store (BIG_TRAIN_CASES)."""
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    xs = rng.integers(1, cfg["idim"], size=(B, Tp), dtype=np.int64)
    ds = rng.integers(ds_lo, ds_hi, size=(B, Tp), dtype=np.int64)
    total = int(ds[0].sum())
    for b in range(1, B):                       # make every row sum to the same frame count
        diff = total - int(ds[b].sum())
        ds[b, 0] += max(diff, 0)
        j = 0
        while ds[b].sum() > total:
            take = min(ds[b, j] - ds_lo, ds[b].sum() - total) if ds_lo else min(ds[b, j], ds[b].sum() - total)
            ds[b, j] -= take
            j += 1
    assert all(int(ds[b].sum()) == total for b in range(B)) and total > 0
    ys = rng.standard_normal((B, total, cfg["odim"]), dtype=np.float32)
    ps = rng.standard_normal((B, total, 4), dtype=np.float32)
    es = rng.standard_normal((B, total, 1), dtype=np.float32)
    ilens = np.full((B,), Tp, np.int64)
    olens = np.full((B,), total, np.int64)
    return xs, ilens, ys, olens, ds, ps, es


def long_mel_rows(T, step=16):
    """Frame rows of a LONG_CASES fixture that are stored in full."""
    return np.unique(np.concatenate([np.arange(0, T, step), np.arange(max(T - 8, 0), T)]))
