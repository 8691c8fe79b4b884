"""Shared test plumbing: golden fixture loading and synthetic-weight state dicts."""
import glob
import json
import os

import numpy as np
import torch

from fastspeech2_amd import synth
from oracle import fs2_oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

INFER_CASES = ["tiny_conv9_plain", "tiny_linear_scaled", "tiny_conv3_mixed_ln", "yaml_dims_2x2",
               "d9_zero_durations", "duration_control_0p8", "duration_control_1p3", "baseline_4x4",
               "postnet_tiny_bn", "postnet_tiny_nobn", "yaml_shipped_full"]
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
                                alpha=meta.get("alpha", 1.0), **kw)
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
