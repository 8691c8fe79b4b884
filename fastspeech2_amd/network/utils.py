"""Host helpers with the reference's names and argument meaning (network/utils.py:50-115).
The kernels consume row ranges, not masks; these exist for callers (loss path, plots) that
still want the boolean tensors."""
from typing import List, Union

import torch
import torch.nn as nn


def pad_list(xs, pad_value):
    """list of (T_i, ...) tensors -> (B, Tmax, ...) filled with pad_value (network/utils.py:50-56)."""
    n_batch = len(xs)
    max_len = max(x.shape[0] for x in xs)
    out = xs[0].new_full((n_batch, max_len) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, :x.shape[0]] = x
    return out


def make_pad_mask(lengths: Union[torch.LongTensor, List[int]], xs: torch.Tensor = None, length_dim: int = -1):
    """True at padded positions (network/utils.py:59-82)."""
    if length_dim == 0:
        raise ValueError(f"length_dim cannot be {length_dim}")
    if not isinstance(lengths, list):
        lengths = lengths.tolist()
    bs = len(lengths)
    maxlen = int(max(lengths)) if xs is None else xs.shape[length_dim]
    steps = torch.arange(0, maxlen, dtype=torch.int64).unsqueeze(0).expand(bs, maxlen)
    mask = steps >= torch.tensor(lengths, dtype=torch.int64).unsqueeze(-1)
    if xs is not None:
        assert xs.shape[0] == bs, (xs.shape[0], bs)
        if length_dim < 0:
            length_dim = xs.dim() + length_dim
        ind = tuple(slice(None) if i in (0, length_dim) else None for i in range(xs.dim()))
        mask = mask[ind].expand_as(xs).to(xs.device)
    return mask


def make_non_pad_mask(lengths, xs: torch.Tensor = None, length_dim: int = -1):
    return ~make_pad_mask(lengths, xs, length_dim)


def initialize(model: nn.Module, init_type: str = "pytorch"):
    """transformer_init handling (network/utils.py:89-109).  As in the reference, the bias-zeroing
    loop is inert (`p.dim == 1` compares a bound method) and Embedding/LayerNorm are re-initialised."""
    if init_type == "pytorch":
        return
    for p in model.parameters():
        if p.dim() > 1:
            if init_type == "xavier_uniform":
                nn.init.xavier_uniform_(p.data)
            elif init_type == "xavier_normal":
                nn.init.xavier_normal_(p.data)
            elif init_type == "kaiming_uniform":
                nn.init.kaiming_uniform_(p.data, nonlinearity="relu")
            elif init_type == "kaiming_normal":
                nn.init.kaiming_normal_(p.data, nonlinearity="relu")
            else:
                raise ValueError(f"unknown initialization: {init_type}")
    for m in model.modules():
        if isinstance(m, (nn.Embedding, nn.LayerNorm)):
            m.reset_parameters()


class Reporter:
    """Stand-in for the chainer.Chain reporter (network/utils.py:112-115): keeps the last
    observation as a plain dict (chainer is not a dependency of this package)."""

    def __init__(self):
        self._raw = {}

    def report(self, dicts):
        """Values may be floats or 0-dim device tensors; tensors are read back only when
        `observation` is looked at, so a training step enqueues without a host sync."""
        for d in dicts:
            self._raw.update(d)

    @property
    def observation(self):
        return {k: (float(v) if not isinstance(v, float) else v) for k, v in self._raw.items()}
