"""-m gpu: counter-based dropout of the training path.  The mask function is restated in numpy here,
so every kernel that applies or re-derives a mask is checked exactly against torch math with that mask;
the whole step is checked by directional finite differences with the step seed pinned.
The random stream is not torch's: comparisons with the reference in dropout mode are statistical only
("parity unpinned" for the stochastic part; with dropout off the step is pinned by tests/test_gpu_train.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from tests import helpers
from tests.test_gpu_backward import _conv_ref, _seg_tables, _ws
from tests.test_gpu_train import _ragged_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def drop_mul(seed, site, rows, cols, rate, row0=0):
    """numpy restatement of fs2::drop_mul (csrc/common.h): multiplier of element (row, col)"""
    with np.errstate(over="ignore"):
        r = (np.arange(rows, dtype=np.uint32) + np.uint32(row0))[:, None]
        c = np.arange(cols, dtype=np.uint32)[None, :]
        x = (r * np.uint32(0x9E3779B1)) ^ ((c + np.uint32(0x7F4A7C15)) * np.uint32(0x85EBCA77)) ^ np.uint32(seed) \
            ^ np.uint32((site * 0xC2B2AE3D) & 0xFFFFFFFF)
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    thresh = min(int(float(np.float32(rate)) * 4294967296.0), 4294967295)
    return torch.from_numpy((x >= np.uint32(thresh)).astype(np.float64) / (1.0 - float(np.float32(rate))))


def test_dropout_kernel_and_keep_fraction():
    M, C, rate = 1000, 256, 0.3
    x = torch.randn(M, C)
    xd = x.to(DEV)
    xt = torch.empty((M, C), device=DEV, dtype=torch.bfloat16)
    call("fs2_dropout", ptr(xd), ptr(xt), M, C, 123, 7, rate, 1, stream())
    m = drop_mul(123, 7, M, C, rate)
    assert (xd.cpu().double() - x.double() * m).abs().max().item() <= 1e-6
    assert (xt.float().cpu().double() - x.double() * m).abs().max().item() <= 2e-2
    keep = float((m > 0).double().mean())
    assert abs(keep - (1 - rate)) < 0.01
    m2 = drop_mul(124, 7, M, C, rate)
    assert 0.4 < float(((m > 0) == (m2 > 0)).double().mean()) < 0.7      # another seed: an independent mask


def test_add_layernorm_dropout_and_backward_masks():
    M, C = 130, 192
    g = torch.Generator().manual_seed(3)
    x, parts = torch.randn(M, C, generator=g), torch.randn(2, M, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    mp, mo = drop_mul(9, 11, M, C, 0.1), drop_mul(9, 12, M, C, 0.5)
    xd, pd, wd, bd = x.to(DEV), parts.to(DEV), w.to(DEV), b.to(DEV)
    s_out, of = torch.empty((M, C), device=DEV), torch.empty((M, C), device=DEV)
    call("fs2_add_layernorm_dropout", ptr(xd), ptr(pd), 2, M * C, ptr(wd), ptr(bd), ptr(s_out), ptr(of), None, M, C, 1e-12,
         0, 9, 11, 0.1, 12, 0.5, stream())
    ssum = x.double() + parts.double().sum(0) * mp
    assert (s_out.cpu().double() - ssum).abs().max().item() <= 1e-5
    ref = F.layer_norm(ssum, (C,), w.double(), b.double(), 1e-12) * mo
    assert (of.cpu().double() - ref).abs().max().item() <= 1e-4
    # backward: incoming-gradient mask (site 12) and a masked T copy (site 11) beside the unmasked fp32 result
    dy = torch.randn(M, C, generator=g)
    xr, wr, br = ssum.clone().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    (F.layer_norm(xr, (C,), wr, br, 1e-12) * mo * dy.double()).sum().backward()
    sd, dyd = ssum.float().to(DEV), dy.to(DEV)
    dx32, dxt = torch.empty((M, C), device=DEV), torch.empty((M, C), device=DEV)
    dw, db = torch.empty((C,), device=DEV), torch.empty((C,), device=DEV)
    ws = _ws(hip.load().fs2_layernorm_bwd_workspace(M, C))
    call("fs2_layernorm_bwd", ptr(sd), ptr(dyd), None, 0, 0, ptr(wd), None, 0, ptr(dx32), ptr(dxt), ptr(dw), ptr(db), ptr(ws),
         M, C, 1e-12, 0, 0, 9, 12, 0.5, 11, 0.1, stream())
    assert (dx32.cpu().double() - xr.grad).abs().max().item() <= 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert (dxt.cpu().double() - xr.grad * mp).abs().max().item() <= 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert (dw.cpu().double() - wr.grad).abs().max().item() <= 1e-3 * max(1.0, wr.grad.abs().max().item())


def test_conv_gemm_dropout_epilogue():
    lens, taps, cin, n, rate = [70, 31], 3, 64, 128, 0.2
    M = sum(lens)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(M, cin, generator=g)
    w = torch.randn(n, cin, taps, generator=g) / math.sqrt(cin * taps)
    b = torch.randn(n, generator=g) * 0.1
    _, _, seg, _ = _seg_tables(lens)
    xd, wd, bd = x.to(DEV), w.permute(2, 0, 1).contiguous().to(DEV), b.to(DEV)
    out = torch.empty((M, n), device=DEV)
    call("fs2_conv_gemm_dropout", ptr(xd), ptr(wd), ptr(bd), ptr(out), ptr(seg), M, n, cin, taps, cin, n, 1, 1, 0, 77, 5,
         rate, stream())
    ref = (_conv_ref(x.double(), w.double(), lens, taps) + b.double()).clamp_min(0) * drop_mul(77, 5, M, n, rate)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_attention_dropout_forward_backward(dtype):
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    h, dk, lens, rate, seed, site = 2, 64, [100, 37], 0.25, 31, 300
    g = torch.Generator().manual_seed(8)
    d, M, B = h * dk, sum(lens), len(lens)
    qkv = (torch.randn(M, 3 * d, generator=g) * 0.7).to(tdt).float()
    dctx = torch.randn(M, d, generator=g).to(tdt).float()
    qr = qkv.double().requires_grad_(True)
    outs, r0 = [], 0
    for L in lens:
        q, k, v = (qr[r0:r0 + L, j * d:(j + 1) * d].view(L, h, dk).transpose(0, 1) for j in range(3))
        p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(dk), -1)
        mask = torch.stack([drop_mul(seed, site + hh, L, L, rate, row0=r0) for hh in range(h)], 0)
        outs.append(((p * mask) @ v).transpose(0, 1).reshape(L, d))
        r0 += L
    ctx_ref = torch.cat(outs, 0)
    (ctx_ref * dctx.double()).sum().backward()
    _, cu_d, _, _ = _seg_tables(lens)
    qd, dod = qkv.to(DEV, tdt).contiguous(), dctx.to(DEV, tdt).contiguous()
    ctx = torch.empty((M, d), device=DEV, dtype=tdt)
    lse, delta = torch.empty((h, M), device=DEV), torch.empty((h, M), device=DEV)
    call("fs2_attention_fwd", ptr(qd), ptr(ctx), ptr(lse), ptr(cu_d), B, max(lens), M, h, dk, 3 * d, d, code, seed, site,
         rate, stream())
    tol = 2e-5 if dtype == "fp32" else 2e-2
    assert (ctx.float().cpu().double() - ctx_ref.detach()).abs().max().item() <= tol * max(1.0, ctx_ref.abs().max().item())
    dqkv = torch.empty((M, 3 * d), device=DEV, dtype=tdt)
    call("fs2_attention_bwd", ptr(qd), ptr(ctx), ptr(dod), ptr(lse), ptr(delta), ptr(dqkv), ptr(cu_d), B, max(lens), M, h,
         dk, 3 * d, d, code, seed, site, rate, stream())
    err = (dqkv.float().cpu().double() - qr.grad).abs().max().item()
    assert err <= (5e-5 if dtype == "fp32" else 3e-2) * max(1.0, qr.grad.abs().max().item()), err


@pytest.mark.parametrize("name", ["train_tiny_conv9", "train_tiny_conv3_mixed", "postnet_tiny_bn"])
def test_step_with_dropout_is_consistent(name):
    """With every dropout site on and the step seed pinned, the loss is a fixed function of the weights: its
    directional derivative (central differences) must equal <grad, direction>."""
    _, meta = helpers.load_golden(name)
    idim, odim, kw = helpers.split_cfg(meta["cfg"])
    kw = dict(kw, transformer_enc_dropout_rate=0.1, transformer_enc_positional_dropout_rate=0.1,
              transformer_enc_atten_dropout_rate=0.1, pitch_embed_dropout=0.2,
              energy_embed_dropout=0.3)           # decoder / predictor / PostNet rates: constructor defaults
    model = FeedForwardTransformer(idim, odim, **kw)
    model.load_state_dict(helpers.golden_state_dict(meta), strict=True)
    model.to(DEV).train()
    eng = model.train_engine()
    batch = _ragged_batch(helpers.oracle_cfg(meta), 11, [21, 40, 9])
    packed = model.pack_batch(*batch)

    def loss_at(step):
        eng.step = step - 1
        return model(packed=packed)

    l0 = loss_at(5)
    l0.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    model.zero_grad()
    assert abs(loss_at(5).item() - l0.item()) == 0.0                  # same seed: bit-identical
    assert abs(loss_at(6).item() - l0.item()) > 1e-4                  # another step: other masks
    model.dropout_enabled = False
    l_off = loss_at(5).item()
    model.dropout_enabled = True
    assert abs(l_off - l0.item()) > 1e-4
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    eps = 2e-3 / gn
    params = dict(model.named_parameters())
    with torch.no_grad():
        for k, p in params.items():
            p.add_(grads[k], alpha=eps)
    lp = loss_at(5).item()
    with torch.no_grad():
        for k, p in params.items():
            p.add_(grads[k], alpha=-2 * eps)
    lm = loss_at(5).item()
    fd, an = (lp - lm) / (2 * eps), gn * gn
    assert abs(fd - an) <= 0.03 * abs(an), (fd, an)
