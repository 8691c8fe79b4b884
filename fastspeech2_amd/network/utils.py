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
    """True at padded positions (network/utils.py:59-82).  Without `xs`: (B, max(lengths)).  With `xs`: a mask of
    xs's shape whose axis `length_dim` is the time axis (broadcast over every other non-batch axis)."""
    if length_dim == 0:
        raise ValueError(f"length_dim cannot be {length_dim}")
    lens = torch.as_tensor(lengths, dtype=torch.int64).reshape(-1).cpu()
    if xs is None:
        return torch.arange(int(lens.max())) >= lens[:, None]
    assert xs.shape[0] == lens.numel(), (xs.shape[0], lens.numel())
    axis = length_dim % xs.dim()
    shape = [1] * xs.dim()
    shape[axis] = xs.shape[axis]
    time = torch.arange(xs.shape[axis]).view(shape)                     # 1 x .. x T x .. x 1
    shape[axis], shape[0] = 1, lens.numel()
    return (time >= lens.view(shape)).expand(xs.shape).to(xs.device)


def make_non_pad_mask(lengths, xs: torch.Tensor = None, length_dim: int = -1):
    return ~make_pad_mask(lengths, xs, length_dim)


_INITS = {
    "xavier_uniform": lambda w: nn.init.xavier_uniform_(w),
    "xavier_normal": lambda w: nn.init.xavier_normal_(w),
    "kaiming_uniform": lambda w: nn.init.kaiming_uniform_(w, nonlinearity="relu"),
    "kaiming_normal": lambda w: nn.init.kaiming_normal_(w, nonlinearity="relu"),
}


def initialize(model: nn.Module, init_type: str = "pytorch"):
    """transformer_init handling (network/utils.py:89-109): every parameter with more than one axis is re-drawn by the
    named rule, then Embedding / LayerNorm modules go back to their own defaults.  (The reference also means to zero
    the biases but tests `p.dim == 1`, a bound method, so that never happens — kept: biases stay as constructed.)"""
    if init_type == "pytorch":
        return
    if init_type not in _INITS:
        raise ValueError(f"unknown initialization: {init_type}")
    for p in model.parameters():
        if p.dim() > 1:
            _INITS[init_type](p.data)
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
