"""-m gpu: the training-path kernels, called through the C ABI, against torch autograd in fp64 on the
same seeded inputs (bf16: same math on bf16-rounded operands, looser tolerance)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _cu(lens):
    cu = np.zeros(len(lens) + 1, np.int32)
    cu[1:] = np.cumsum(lens)
    return cu


def _seg_tables(lens):
    cu = _cu(lens)
    M = int(cu[-1])
    cu_d = torch.from_numpy(cu).to(DEV)
    seg = torch.empty((max(M, 1), 2), dtype=torch.int32, device=DEV)
    pos = torch.empty((max(M, 1),), dtype=torch.int32, device=DEV)
    call("fs2_build_seg", ptr(cu_d), len(lens), ptr(seg), ptr(pos), M, stream())
    return cu, cu_d, seg, pos


def _ws(nbytes):
    return torch.empty((max(int(nbytes), 16) + 3) // 4, dtype=torch.float32, device=DEV)


def _conv_ref(x, w, lens, taps):
    """per-utterance conv1d with zero "same" padding, fp64, differentiable"""
    outs, r0 = [], 0
    for L in lens:
        xi = x[r0:r0 + L].t().unsqueeze(0)
        outs.append(F.conv1d(xi, w, None, padding=(taps - 1) // 2).squeeze(0).t())
        r0 += L
    return torch.cat(outs, 0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("taps,cin,n,lens,ksplit", [
    (1, 64, 192, [5, 130, 1], 1),
    (9, 64, 128, [1, 3, 200, 17, 4], 3),
    (3, 96, 96, [40, 2, 90], 2),
    (9, 256, 1024, [300, 75], 4),
    (9, 1024, 256, [150, 140], 5),
    (1, 256, 80, [257], 2),
    (5, 128, 24, [33, 31], 1),
    (3, 256, 384, [64, 64, 64], 7),      # more splits than token blocks
    (9, 128, 256, [2000, 1, 777, 63], 2),  # 23 token blocks per workgroup: the stage ring wraps many times
    (5, 64, 64, [900, 131], 1),          # one-tap-per-workgroup form with utterance bounds, 17 blocks in a row
])
def test_conv_wgrad_and_dgrad(dtype, taps, cin, n, lens, ksplit):
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(taps * 77 + cin + n)
    M = sum(lens)
    x = torch.randn(M, cin, generator=g).to(tdt).float()
    w = (torch.randn(n, cin, taps, generator=g) / math.sqrt(cin * taps)).to(tdt).float()
    dy = torch.randn(M, n, generator=g).to(tdt).float()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (_conv_ref(xr, wr, lens, taps) * dy.double()).sum().backward()
    _, _, seg, _ = _seg_tables(lens)
    xd, dyd, wsrc = x.to(DEV, tdt).contiguous(), dy.to(DEV, tdt).contiguous(), w.to(DEV).contiguous()
    # ---- weight gradient, parameter layout (n, cin, taps)
    grad = torch.full((n, cin, taps), 7.0, device=DEV)
    dbias = torch.full((n,), 7.0, device=DEV)
    ws = _ws(hip.load().fs2_conv_wgrad_workspace(n, cin, taps, ksplit))
    call("fs2_conv_wgrad", ptr(dyd), ptr(xd), ptr(grad), ptr(dbias), ptr(ws), ptr(seg) if taps > 1 else None, M, n, cin,
         taps, n, cin, code, ksplit, 0.5, 0, stream())
    exp = 0.5 * wr.grad
    expb = 0.5 * dy.double().sum(0)                # bias gradient: column sums of dY
    tol = 2e-5 if dtype == "fp32" else 1e-4       # operands already bf16-exact: only accumulation order differs
    err = (grad.cpu().double() - exp).abs().max().item()
    assert err <= tol * max(1.0, exp.abs().max().item()), err
    assert (dbias.cpu().double() - expb).abs().max().item() <= tol * max(1.0, expb.abs().max().item())
    call("fs2_conv_wgrad", ptr(dyd), ptr(xd), ptr(grad), None, ptr(ws), ptr(seg) if taps > 1 else None, M, n, cin, taps,
         n, cin, code, ksplit, 0.5, 1, stream())                       # accumulate: doubles; no bias this time
    err = (grad.cpu().double() - 2 * exp).abs().max().item()
    assert err <= 2 * tol * max(1.0, exp.abs().max().item()), err
    assert (dbias.cpu().double() - expb).abs().max().item() <= tol * max(1.0, expb.abs().max().item())
    # ---- packing + input gradient = conv over flipped / transposed weights
    wf = torch.empty((taps, n, cin), device=DEV, dtype=tdt)
    wb = torch.empty((taps, cin, n), device=DEV, dtype=tdt)
    call("fs2_pack_weight", ptr(wsrc), ptr(wf), ptr(wb), n, cin, taps, cin, n, n * cin, cin * n, code, stream())
    assert torch.equal(wf.float().cpu(), w.permute(2, 0, 1))
    assert torch.equal(wb.float().cpu(), w.flip(2).permute(2, 1, 0))
    if n % (4 if dtype == "fp32" else 8) == 0:
        gate = torch.randn(M, cin, generator=g)
        gd = gate.to(DEV, tdt).contiguous()
        dx = torch.empty((M, cin), device=DEV)
        call("fs2_conv_gemm_gated", ptr(dyd), ptr(wb), None, None, ptr(gd), 1.25, ptr(dx), ptr(seg) if taps > 1 else None,
             M, cin, n, taps, n, 0, cin, cin, 1, code, stream())
        expx = 1.25 * xr.grad * (gate.to(tdt).double() > 0)
        err = (dx.cpu().double() - expx).abs().max().item()
        assert err <= (2e-5 if dtype == "fp32" else 2e-3) * max(1.0, expx.abs().max().item()), err


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pack_weights_batched_equals_one_by_one(dtype):
    """fs2_pack_weights (one launch, descriptor table on the device) against fs2_pack_weight per parameter"""
    import numpy as np
    from fastspeech2_amd.train_engine import PACK_DESC
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(5)
    shapes = [(24, 16, 9), (7, 40, 1), (64, 64, 3), (1, 8, 5), (300, 33, 1)]
    rows, refs, outs = [], [], []
    for n, c, k in shapes:
        w = torch.randn(n, c, k, generator=g).to(DEV).contiguous()
        ld_b = n + 3                                                     # padded bwd rows
        f1, b1 = torch.zeros((k, n, c), device=DEV, dtype=tdt), torch.zeros((k, c, ld_b), device=DEV, dtype=tdt)
        f2, b2 = torch.zeros_like(f1), torch.zeros_like(b1)
        call("fs2_pack_weight", ptr(w), ptr(f1), ptr(b1), n, c, k, c, ld_b, n * c, c * ld_b, code, stream())
        rows.append((w.data_ptr(), f2.data_ptr(), b2.data_ptr(), n, c, k, c, ld_b, 0, n * c, c * ld_b))
        refs.append((w, f1, b1))
        outs.append((f2, b2))
    start = np.zeros(len(rows) + 1, np.int32)
    start[1:] = np.cumsum([2 * (-(-(r[3] * r[4]) // 256)) for r in rows])
    table = torch.from_numpy(np.array(rows, dtype=PACK_DESC).view(np.uint8).copy()).to(DEV)
    start_d = torch.from_numpy(start).to(DEV)
    call("fs2_pack_weights", ptr(table), ptr(start_d), len(rows), int(start[-1]), code, stream())
    for (w, f1, b1), (f2, b2) in zip(refs, outs):
        assert torch.equal(f1, f2) and torch.equal(b1, b2)
        assert torch.equal(f2.float(), w.to(tdt).float().permute(2, 0, 1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,C,nparts,mode", [(37, 64, 0, "pre"), (300, 256, 3, "post"), (129, 384, 1, "relu"),
                                             (50, 192, 2, "sum")])
def test_layernorm_bwd(dtype, M, C, nparts, mode):
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    if mode == "relu":
        x = x.clamp_min(0)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = torch.randn(M, C, generator=g)
    parts = torch.randn(max(nparts, 1), M, C, generator=g)
    dres = torch.randn(M, C, generator=g)
    gtot = dy.double() + (parts[:nparts].double().sum(0) if nparts else 0)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    if mode == "sum":
        exp_dx, exp_dw, exp_db = gtot + dres.double(), None, None
    else:
        (F.layer_norm(xr, (C,), wr, br, 1e-12) * gtot).sum().backward()
        exp_dx = xr.grad + (dres.double() if mode == "pre" else 0)
        if mode == "relu":
            exp_dx = exp_dx * (x.double() > 0)
        exp_dw, exp_db = wr.grad, br.grad
    xd, wd, dyd, pd, rd = x.to(DEV), w.to(DEV), dy.to(DEV), parts.to(DEV), dres.to(DEV)
    dx32 = torch.empty((M, C), device=DEV)
    dxt = torch.empty((M, C), device=DEV, dtype=tdt)
    dw, db = torch.full((C,), 3.0, device=DEV), torch.full((C,), 3.0, device=DEV)
    ws = _ws(hip.load().fs2_layernorm_bwd_workspace(M, C))
    call("fs2_layernorm_bwd", ptr(xd), ptr(dyd), ptr(pd) if nparts else None, nparts, M * C,
         ptr(wd) if mode != "sum" else None, ptr(rd) if mode in ("pre", "sum") else None, int(mode == "relu"),
         ptr(dx32), ptr(dxt), ptr(dw), ptr(db), ptr(ws), M, C, 1e-12, 0, code, 0, 0, 0.0, 0, 0.0, stream())
    sc = max(1.0, exp_dx.abs().max().item())
    assert (dx32.cpu().double() - exp_dx).abs().max().item() <= 2e-5 * sc
    assert (dxt.float().cpu().double() - exp_dx).abs().max().item() <= (2e-5 if dtype == "fp32" else 1e-2) * sc
    if exp_dw is not None:
        assert (dw.cpu().double() - exp_dw).abs().max().item() <= 1e-4 * max(1.0, exp_dw.abs().max().item())
        assert (db.cpu().double() - exp_db).abs().max().item() <= 1e-4 * max(1.0, exp_db.abs().max().item())


def test_colreduce_rowdot_bwd():
    g = torch.Generator().manual_seed(5)
    M, C, R = 777, 384, 5
    G = torch.randn(M, C, generator=g)
    S = torch.randn(M, R, generator=g)
    Gd, Sd = G.to(DEV), S.to(DEV)
    out = torch.empty((R, C), device=DEV)
    ws = _ws(hip.load().fs2_colreduce_workspace(M, C, R))
    call("fs2_colreduce", ptr(Gd), 0, C, ptr(Sd), R, R, None, None, ptr(out), ptr(ws), M, C, 2.0, 0, stream())
    exp = 2.0 * S.double().t() @ G.double()
    assert (out.cpu().double() - exp).abs().max().item() <= 1e-4 * exp.abs().max().item()
    # column sums of a bf16 matrix with a padded leading dimension
    Gb = torch.randn(M, 96, generator=g).to(torch.bfloat16)
    Gbd = Gb.to(DEV)
    out1 = torch.empty((80,), device=DEV)
    call("fs2_colreduce", ptr(Gbd), 1, 96, None, 0, 1, None, None, ptr(out1), ptr(ws), M, 80, 1.0, 0, stream())
    exp1 = Gb[:, :80].double().sum(0)
    assert (out1.cpu().double() - exp1).abs().max().item() <= 1e-4 * exp1.abs().max().item()
    # positional-encoding mode
    pe = torch.randn(40, C, generator=g)
    pos = torch.randint(0, 40, (M,), generator=g, dtype=torch.int32)
    outp = torch.empty((C,), device=DEV)
    ped, posd = pe.to(DEV), pos.to(DEV)
    call("fs2_colreduce", ptr(Gd), 0, C, None, 0, 1, ptr(ped), ptr(posd), ptr(outp), ptr(ws), M, C, 1.0, 0, stream())
    expp = (G.double() * pe[pos.long()].double()).sum(0)
    assert (outp.cpu().double() - expp).abs().max().item() <= 1e-4 * expp.abs().max().item()
    # rowdot backward
    W = torch.randn(4, C, generator=g)
    dy = torch.randn(M, 4, generator=g)
    Wd, dyd = W.to(DEV), dy.to(DEV)
    dx = torch.empty((M, C), device=DEV)
    call("fs2_rowdot_bwd", ptr(dyd), ptr(Wd), ptr(dx), M, C, 4, stream())
    expx = dy.double() @ W.double()
    assert (dx.cpu().double() - expx).abs().max().item() <= 1e-5 * expx.abs().max().item()


def test_lr_scatter_embed_bwd_loss_grad():
    g = torch.Generator().manual_seed(9)
    d = 64
    ilens = [4, 1, 9]
    durs = [np.array([2, 0, 3, 1]), np.array([5]), np.array([0, 1, 1, 4, 0, 2, 2, 1, 3])]
    dur = np.concatenate(durs).astype(np.int32)
    olens = [int(x.sum()) for x in durs]
    cum = np.concatenate([np.cumsum(x) for x in durs]).astype(np.int32)
    Mp, Mf = sum(ilens), sum(olens)
    dfr = torch.randn(Mf, d, generator=g)
    base = torch.randn(Mp, d, generator=g)
    exp = base.double().clone()
    f0 = np.concatenate([[0], np.cumsum(olens)])
    p = 0
    for b, x in enumerate(durs):
        f = f0[b]
        for dv in x:
            exp[p] += 0.5 * dfr[f:f + dv].double().sum(0)
            f += dv
            p += 1
    out = torch.empty((Mp, d), device=DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dfr_d, base_d = dfr.to(DEV), base.to(DEV)
    a_dur, a_cum, a_cup, a_cuf = t(dur), t(cum), t(_cu(ilens)), t(_cu(olens))
    call("fs2_lr_scatter", ptr(dfr_d), ptr(a_dur), ptr(a_cum), ptr(a_cup), ptr(a_cuf), len(ilens), ptr(base_d), 0.5,
         ptr(out), Mp, d, stream())
    assert (out.cpu().double() - exp).abs().max().item() <= 1e-5
    # embedding backward
    idim, M = 11, 50
    ids = torch.randint(0, idim, (M,), generator=g)
    dx = torch.randn(M, d, generator=g)
    emb = torch.zeros(idim, d, dtype=torch.float64, requires_grad=True)
    (F.embedding(ids, emb, padding_idx=0) * 3.0 * dx.double()).sum().backward()
    ids_d, dx_d = ids.to(DEV), dx.to(DEV)
    demb = torch.empty((idim, d), device=DEV)
    call("fs2_embed_bwd", ptr(ids_d), ptr(dx_d), 3.0, ptr(demb), M, d, idim, 0, stream())
    assert (demb.cpu().double() - emb.grad).abs().max().item() <= 1e-5
    # loss gradients
    pred, tgt = torch.randn(M, 20, generator=g), torch.randn(M, 20, generator=g)
    pd_, td_ = pred.to(DEV), tgt.to(DEV)
    g32 = torch.empty((M, 20), device=DEV)
    two = torch.full((1,), 2.0, device=DEV)
    gt = torch.full((M, 24), 9.0, device=DEV, dtype=torch.bfloat16)
    call("fs2_loss_grad", ptr(pd_), ptr(td_), None, 0, 0.125, ptr(two), ptr(g32), ptr(gt), M, 20, 24, 1, stream())
    assert torch.equal(g32.cpu(), 0.25 * torch.sign(pred - tgt))
    assert torch.equal(gt.float().cpu()[:, :20], 0.25 * torch.sign(pred - tgt)) and (gt[:, 20:] == 0).all()
    call("fs2_loss_grad", ptr(pd_), ptr(td_), None, 1, 0.25, None, ptr(g32), None, M, 20, 20, 0, stream())
    assert (g32.cpu() - 0.5 * (pred - tgt)).abs().max().item() <= 1e-6
    durd = torch.randint(0, 9, (M, 1), generator=g, dtype=torch.int32)
    logd = torch.randn(M, 1, generator=g)
    ld_, dd_ = logd.to(DEV), durd.to(DEV)
    gd = torch.empty((M, 1), device=DEV)
    call("fs2_loss_grad", ptr(ld_), None, ptr(dd_), 2, 0.5, None, ptr(gd), None, M, 1, 1, 0, stream())
    assert (gd.cpu() - (logd - torch.log(durd.float() + 1.0))).abs().max().item() <= 1e-6


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("h,dk,lens", [(2, 32, [5, 70, 1]), (2, 64, [130, 64]), (4, 96, [33, 200]), (2, 128, [300, 17, 129]),
                                       (2, 64, [130, 64, 257, 1, 90])])   # 10 pairs: XCD-aware order with a padded grid
def test_attention_fwd_lse_and_bwd(dtype, h, dk, lens):
    code = hip.DTYPE_CODE[dtype]
    tdt = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(h * 100 + dk)
    d, M, B = h * dk, sum(lens), len(lens)
    qkv = (torch.randn(M, 3 * d, generator=g) * 0.7).to(tdt).float()
    dctx = torch.randn(M, d, generator=g).to(tdt).float()
    qr = qkv.double().requires_grad_(True)
    outs, lses, r0 = [], [], 0
    for L in lens:
        q, k, v = (qr[r0:r0 + L, j * d:(j + 1) * d].view(L, h, dk).transpose(0, 1) for j in range(3))
        s = q @ k.transpose(1, 2) / math.sqrt(dk)
        lses.append(torch.logsumexp(s, -1).detach())          # (h, L)
        outs.append((torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, d))
        r0 += L
    ctx_ref = torch.cat(outs, 0)
    (ctx_ref * dctx.double()).sum().backward()
    cu, cu_d, _, _ = _seg_tables(lens)
    qd, dod = qkv.to(DEV, tdt).contiguous(), dctx.to(DEV, tdt).contiguous()
    ctx = torch.empty((M, d), device=DEV, dtype=tdt)
    lse = torch.empty((h, M), device=DEV)
    call("fs2_attention_fwd", ptr(qd), ptr(ctx), ptr(lse), ptr(cu_d), B, max(lens), M, h, dk, 3 * d, d, code, 0, 0, 0.0,
         stream())
    tol = 2e-5 if dtype == "fp32" else 2e-2
    assert (ctx.float().cpu().double() - ctx_ref.detach()).abs().max().item() <= tol * max(1.0, ctx_ref.abs().max().item())
    lse_ref = torch.cat(lses, 1) / math.log(2.0)
    assert (lse.cpu().double() - lse_ref).abs().max().item() <= (1e-4 if dtype == "fp32" else 2e-2)
    delta = torch.empty((h, M), device=DEV)
    dqkv = torch.full((M, 3 * d), 5.0, device=DEV, dtype=tdt)
    call("fs2_attention_bwd", ptr(qd), ptr(ctx), ptr(dod), ptr(lse), ptr(delta), ptr(dqkv), ptr(cu_d), B, max(lens), M,
         h, dk, 3 * d, d, code, 0, 0, 0.0, stream())
    exp = qr.grad
    err = (dqkv.float().cpu().double() - exp).abs().max().item()
    assert err <= (5e-5 if dtype == "fp32" else 3e-2) * max(1.0, exp.abs().max().item()), err
