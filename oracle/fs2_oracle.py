"""TEST INFRASTRUCTURE — CPU oracle for the FastSpeech2 forward path.  NOT product code.

A functional fp32 CPU restatement (torch CPU ops + numpy integer code) of the reference
algorithm with the repair set of SURVEY.md §0.1.  It works from a plain `state_dict`
(reference key names) and a config dict (the reference constructor kwargs), so it does not
share a line of module code with either the reference or the product package.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file; the product path (`fastspeech2_amd/`) never does and fails loudly when the HIP
library is missing.

Parity status: PINNED.  `tests/golden/*.npz` were produced by importing the repaired
reference itself in the build container (`tests/golden/make_golden.py`, via
`oracle/ref_harness.py`); `tests/test_oracle_golden.py` checks this file against every
one of them (mel <= 1e-5 max-abs, durations / LengthRegulator indices exact).

Reference citations are file:line under /root/reference.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-12          # network/utils.py:41
DUR_OFFSET = 1.0        # network/variance_adaptor.py:100

# constructor defaults, network/fastspeech2.py:58-104
DEFAULT_CFG = dict(
    adim=384, aheads=4, positionwise_layer_type="linear", positionwise_conv_kernel_size=3,
    eunits=1536, elayers=6, encoder_normalized_before=True, encoder_concat_after=False,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3,
    dlayers=6, dunits=1536, decoder_normalized_before=False, decoder_concat_after=False,
    reduction_factor=1, postnet_layers=5, use_batch_norm=True, use_masking=True, use_scaled_pos_enc=True,
)


def full_cfg(idim: int, odim: int, **kw) -> dict:
    cfg = dict(DEFAULT_CFG)
    cfg.update({k: v for k, v in kw.items() if k in DEFAULT_CFG})
    cfg["idim"], cfg["odim"] = idim, odim
    return cfg


# ----------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------
def posenc_table(T: int, d: int) -> torch.Tensor:
    """network/encoder.py:116-121 — the fp32 sin/cos recipe, rebuilt at length T (D10)."""
    pe = torch.zeros(T, d)
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def pos_encode(x: torch.Tensor, sd, prefix: str, scaled: bool) -> torch.Tensor:
    """network/encoder.py:125-128 (plain: x*sqrt(d)+pe) and :142-145 with D1 (scaled: x+alpha*pe)."""
    T, d = x.shape[-2], x.shape[-1]
    pe = posenc_table(T, d)
    if scaled:
        return x + sd[prefix + "alpha"] * pe
    return x * math.sqrt(d) + pe


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """network/utils.py:39-47 — LayerNorm over the channel (last) axis, eps 1e-12."""
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def mha(x: torch.Tensor, sd, p: str, h: int, return_atten: bool = False):
    """network/encoder.py:64-89 on ONE unpadded utterance x:(T,d) (mask == all ones)."""
    T, d = x.shape
    dk = d // h
    q = F.linear(x, sd[p + "linear_q.weight"], sd[p + "linear_q.bias"]).view(T, h, dk).transpose(0, 1)
    k = F.linear(x, sd[p + "linear_k.weight"], sd[p + "linear_k.bias"]).view(T, h, dk).transpose(0, 1)
    v = F.linear(x, sd[p + "linear_v.weight"], sd[p + "linear_v.bias"]).view(T, h, dk).transpose(0, 1)
    scores = torch.matmul(q, k.transpose(-2, -1) / math.sqrt(dk))      # :77 scale on K
    atten = torch.softmax(scores, dim=-1)                              # :84
    ctx = torch.matmul(atten, v).transpose(0, 1).contiguous().view(T, h * dk)
    out = F.linear(ctx, sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])
    if return_atten:
        return out, atten
    return out


def ffn(x: torch.Tensor, sd, p: str, ff_type: str, k: int) -> torch.Tensor:
    """network/encoder.py:28-30 (conv1d, zero 'same' padding, both convs kernel k) /
    :43-44 (linear)."""
    if ff_type == "conv1d":
        pad = (k - 1) // 2
        y = x.t().unsqueeze(0)
        y = torch.relu(F.conv1d(y, sd[p + "w1.weight"], sd[p + "w1.bias"], padding=pad))
        y = F.conv1d(y, sd[p + "w2.weight"], sd[p + "w2.bias"], padding=pad)
        return y.squeeze(0).t()
    if ff_type == "linear":
        y = torch.relu(F.linear(x, sd[p + "w1.weight"], sd[p + "w1.bias"]))
        return F.linear(y, sd[p + "w2.weight"], sd[p + "w2.bias"])
    raise NotImplementedError("not implemented")                       # network/encoder.py:232


def fft_block(x, sd, p, h, pre_ln, concat_after, ff_type, k, attens=None):
    """network/encoder.py:168-189."""
    residual = x
    if pre_ln:
        x = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    if attens is not None:
        att, aw = mha(x, sd, p + "self_atten.", h, return_atten=True)
        attens.append(aw)
    else:
        att = mha(x, sd, p + "self_atten.", h)
    if concat_after:
        x = residual + F.linear(torch.cat((x, att), dim=-1),
                                sd[p + "concat_linear.weight"], sd[p + "concat_linear.bias"])
    else:
        x = residual + att
    if not pre_ln:
        x = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    residual = x
    if pre_ln:
        x = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    x = residual + ffn(x, sd, p + "feed_forward.", ff_type, k)
    if not pre_ln:
        x = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return x


def fft_stack(x, sd, p, n_layers, h, pre_ln, concat_after, ff_type, k, scaled, has_embedding,
              attens=None):
    """network/encoder.py:248-254 with D3; embed = [Embedding,] PosEnc (:211-219)."""
    pe_prefix = p + ("embed.1." if has_embedding else "embed.0.")
    x = pos_encode(x, sd, pe_prefix, scaled)
    for i in range(n_layers):
        x = fft_block(x, sd, "%sencoders.%d." % (p, i), h, pre_ln, concat_after, ff_type, k, attens)
    if pre_ln:
        x = layer_norm(x, sd[p + "after_norm.weight"], sd[p + "after_norm.bias"])
    return x


def predictor_trunk(x: torch.Tensor, sd, p: str, n_layers: int) -> torch.Tensor:
    """Conv1d(k, same pad)+bias -> ReLU -> LayerNorm(channels) [-> dropout(eval)] x n_layers.
    network/variance_adaptor.py:66-77 and :103-114 with D4."""
    y = x.t().unsqueeze(0)
    for j in range(n_layers):
        w = sd["%sconv.%d.0.conv.weight" % (p, j)]
        b = sd.get("%sconv.%d.0.conv.bias" % (p, j))
        y = torch.relu(F.conv1d(y, w, b, padding=(w.shape[-1] - 1) // 2))
        y = layer_norm(y.transpose(1, 2), sd["%sconv.%d.2.weight" % (p, j)],
                       sd["%sconv.%d.2.bias" % (p, j)]).transpose(1, 2)
    return y.squeeze(0).t()


def n_pred_layers(sd, p: str) -> int:
    n = 0
    while "%sconv.%d.0.conv.weight" % (p, n) in sd:
        n += 1
    return n


def duration_log(x, sd) -> torch.Tensor:
    """DurationPredictor.forward with D4/D5: (T,) log-domain floats (variance_adaptor.py:117-126)."""
    p = "variance_adaptor.duration_predictor."
    y = predictor_trunk(x, sd, p, n_pred_layers(sd, p))
    return F.linear(y, sd[p + "linear.weight"], sd[p + "linear.bias"]).squeeze(-1)


def duration_round(logd: torch.Tensor) -> torch.Tensor:
    """DurationPredictor.inference (:135): clamp(round(exp(x) - 1), min=0).long(); torch.round is
    round-half-to-even."""
    return torch.clamp(torch.round(logd.exp() - DUR_OFFSET), min=0).long()


def variance_predict(x, sd, which: str) -> torch.Tensor:
    """VariancePredictor.forward (:80-90): (T, out_dim)."""
    p = "variance_adaptor.%s_predictor." % which
    y = predictor_trunk(x, sd, p, n_pred_layers(sd, p))
    return F.linear(y, sd[p + "linear.weight"], sd[p + "linear.bias"])


def variance_embed(v: torch.Tensor, sd, which: str) -> torch.Tensor:
    """VarianceEmbedding / ConvNorm (:35-51; utils.py:7-36): Conv1d over time, 'same' padding."""
    w = sd["variance_adaptor.%s_embed.conv.conv.weight" % which]
    b = sd["variance_adaptor.%s_embed.conv.conv.bias" % which]
    return F.conv1d(v.t().unsqueeze(0), w, b, padding=(w.shape[-1] - 1) // 2).squeeze(0).t()


# ----------------------------------------------------------------------------------
# LengthRegulator — integer work, numpy (network/variance_adaptor.py:13-31)
# ----------------------------------------------------------------------------------
def scale_durations(d: np.ndarray, duration_control: float) -> np.ndarray:
    """:18-19 — round(d.float() * control).long(), fp32 arithmetic, half-to-even."""
    if duration_control == 1.0:
        return d.astype(np.int64)
    return np.rint(d.astype(np.float32) * np.float32(duration_control)).astype(np.int64)


def length_regulate_index(d: np.ndarray):
    """One utterance.  Returns (index map of length sum(d'), d') where d' is d after the D9
    fix-up (all-zero row -> all ones, :28-30) and index[t] = the phone that frame t copies
    (:31, zero-duration phones are skipped)."""
    d = np.asarray(d, dtype=np.int64).copy()
    if d.sum() == 0:
        d[:] = 1
    return np.repeat(np.arange(d.shape[0], dtype=np.int64), d), d


# ----------------------------------------------------------------------------------
# whole forward, one utterance
# ----------------------------------------------------------------------------------
def _enc_kw(cfg):
    return dict(n_layers=cfg["elayers"], h=cfg["aheads"], pre_ln=cfg["encoder_normalized_before"],
                concat_after=cfg["encoder_concat_after"], ff_type=cfg["positionwise_layer_type"],
                k=cfg["positionwise_conv_kernel_size"], scaled=cfg["use_scaled_pos_enc"])


def _dec_kw(cfg):
    return dict(n_layers=cfg["dlayers"], h=cfg["aheads"], pre_ln=cfg["decoder_normalized_before"],
                concat_after=cfg["decoder_concat_after"], ff_type=cfg["positionwise_layer_type"],
                k=cfg["positionwise_conv_kernel_size"], scaled=cfg["use_scaled_pos_enc"])


def encode(sd, cfg, x_ids: torch.Tensor, attens=None) -> torch.Tensor:
    """Embedding(padding_idx=0) + encoder stack (network/fastspeech2.py:116-133, :197-198)."""
    emb = F.embedding(x_ids, sd["encoder.embed.0.weight"])
    return fft_stack(emb, sd, "encoder.", has_embedding=True, attens=attens, **_enc_kw(cfg))


BN_EPS = 1e-5   # nn.BatchNorm1d default (network/fastspeech2.py:35)


def postnet(sd, cfg, before: torch.Tensor) -> torch.Tensor:
    """REPAIRED PostNet (network/fastspeech2.py:13-54 builds only n-1 layers, feeds idim
    channels and never emits odim, so the literal code cannot run): n layers of Conv1d(k, bias=False) ->
    [BatchNorm1d, eval] -> tanh (not on the last layer) -> dropout(eval), first conv from odim channels,
    last conv to odim.  Returns the residual branch postnet(before) as (T, odim)."""
    y = before.t().unsqueeze(0)
    n = cfg["postnet_layers"]
    for j in range(n):
        w = sd["postnet.postnet.%d.0.weight" % j]
        y = F.conv1d(y, w, None, padding=(w.shape[-1] - 1) // 2)
        if cfg["use_batch_norm"]:
            p = "postnet.postnet.%d.1." % j
            y = F.batch_norm(y, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                             False, 0.0, BN_EPS)
        if j != n - 1:
            y = torch.tanh(y)
    return y.squeeze(0).t()


def postnet_train(sd, cfg, befores, running=None):
    """The repaired PostNet in train() mode (dropout off) on a batch: `befores` is a list of (T_b, odim) tensors, one per
    utterance.  Every conv sees one utterance at a time (zero padding at its edges, as in the padded reference batch
    once padding frames are excluded); nn.BatchNorm1d takes its statistics over the frames of ALL utterances.  Returns
    the list of residual branches; `running` (dict) receives the updated running_mean / running_var
    (momentum 0.1, unbiased variance) when given.  Padded frames do not exist here: the literal reference would also
    average over them (it cannot run at all) - parity of this piece is unpinned."""
    n = cfg["postnet_layers"]
    lens = [int(b.shape[0]) for b in befores]
    ys = [b.t().unsqueeze(0) for b in befores]
    for j in range(n):
        w = sd["postnet.postnet.%d.0.weight" % j]
        ys = [F.conv1d(y, w, None, padding=(w.shape[-1] - 1) // 2) for y in ys]
        if cfg["use_batch_norm"]:
            p = "postnet.postnet.%d.1." % j
            allf = torch.cat(ys, 2)
            rm = sd[p + "running_mean"].detach().clone()
            rv = sd[p + "running_var"].detach().clone()
            allf = F.batch_norm(allf, rm, rv, sd[p + "weight"], sd[p + "bias"], True, 0.1, BN_EPS)
            if running is not None:
                running[p + "running_mean"], running[p + "running_var"] = rm, rv
            ys = list(torch.split(allf, lens, 2))
        if j != n - 1:
            ys = [torch.tanh(y) for y in ys]
    return [y.squeeze(0).t() for y in ys]


def decode(sd, cfg, hs: torch.Tensor, attens=None, return_before: bool = False):
    """decoder stack + feat_out (+ repaired postnet): network/fastspeech2.py:218-225."""
    zs = fft_stack(hs, sd, "decoder.", has_embedding=False, attens=attens, **_dec_kw(cfg))
    before = F.linear(zs, sd["feat_out.weight"], sd["feat_out.bias"]).view(-1, cfg["odim"])
    after = before if cfg["postnet_layers"] == 0 else before + postnet(sd, cfg, before)
    return (before, after) if return_before else after


def infer_one(sd, cfg, x_ids: torch.Tensor, duration_control: float = 1.0,
              ds: Optional[torch.Tensor] = None, ps: Optional[torch.Tensor] = None,
              es: Optional[torch.Tensor] = None, pitch_control: float = 1.0,
              energy_control: float = 1.0) -> Dict[str, torch.Tensor]:
    """FeedForwardTransformer.inference (network/fastspeech2.py:297-302) through
    VarianceAdaptor.inference with D6 (variance_adaptor.py:195-207), one unpadded utterance.
    `ds` / `ps` / `es`, when given, replace the predicted durations / pitch / energy that feed the
    LengthRegulator and the embeddings (teacher forcing, VarianceAdaptor.forward :178-193)."""
    with torch.no_grad():
        hs = encode(sd, cfg, x_ids)
        logd = duration_log(hs, sd)
        d_pred = duration_round(logd)
        d_used = d_pred if ds is None else ds.long()
        d_used = torch.from_numpy(scale_durations(d_used.numpy(), duration_control))
        idx, d_fixed = length_regulate_index(d_used.numpy())
        idx_t = torch.from_numpy(idx)
        hs_lr = hs[idx_t]
        p_outs = variance_predict(hs_lr, sd, "pitch")
        e_outs = variance_predict(hs_lr, sd, "energy")
        p_in = p_outs * pitch_control if ps is None else ps
        e_in = e_outs * energy_control if es is None else es
        hs_va = hs_lr + (variance_embed(e_in, sd, "energy") + variance_embed(p_in, sd, "pitch"))
        before, mel = decode(sd, cfg, hs_va, return_before=True)
    d_report = torch.from_numpy(d_fixed) if ds is None and duration_control == 1.0 else d_pred
    return dict(mel=mel, mel_before=before, durations=torch.from_numpy(d_fixed), durations_pred=d_pred,
                durations_reported=d_report, log_durations=logd, lr_index=idx_t,
                enc_out=hs, hs_lr=hs_lr, hs_va=hs_va, pitch=p_outs, energy=e_outs)


def infer_batch(sd, cfg, xs: List[torch.Tensor], **kw) -> List[Dict[str, torch.Tensor]]:
    """The truth for a ragged batch is one batch-1 call per utterance."""
    return [infer_one(sd, cfg, x, **kw) for x in xs]


def attentions_one(sd, cfg, x_ids, ds, ps, es):
    """calculate_all_attentions (network/fastspeech2.py:305-341) for one utterance: list of
    (h,T,T) arrays in module order encoder.* then decoder.*, plus the predicted mel."""
    enc_att, dec_att = [], []
    with torch.no_grad():
        hs = encode(sd, cfg, x_ids, attens=enc_att)
        idx, _ = length_regulate_index(ds.numpy())
        hs_lr = hs[torch.from_numpy(idx)]
        hs_va = hs_lr + (variance_embed(ps, sd, "pitch") + variance_embed(es, sd, "energy"))
        mel = decode(sd, cfg, hs_va, attens=dec_att)
    return enc_att, dec_att, mel


# ----------------------------------------------------------------------------------
# training-step forward (loss) — network/fastspeech2.py:232-295, network/loss.py:5-14
# ----------------------------------------------------------------------------------
def train_loss(sd, cfg, xs, ilens, ys, olens, ds, ps, es) -> Dict[str, float]:
    """Loss terms of FeedForwardTransformer.forward with use_masking=True, computed per utterance
    on unpadded data (equal to the masked_select formulation at :259-268 because every term is a
    mean over the selected elements).  NOTE: the reference runs the padded batch through the
    FFT blocks, whose conv-FFN sees pad frames; for ragged batches this oracle follows the
    per-utterance (batch-1) semantics of SURVEY fact 6, as the product does."""
    assert cfg["use_masking"]
    l1_num = l1_den = lb_num = 0.0
    d_num = d_den = 0.0
    p_num = p_den = 0.0
    e_num = e_den = 0.0
    with torch.no_grad():
        for b in range(len(ilens)):
            il, ol = int(ilens[b]), int(olens[b])
            x = xs[b, :il]
            hs = encode(sd, cfg, x)
            logd = duration_log(hs, sd)
            idx, _ = length_regulate_index(ds[b, :il].numpy())
            hs_lr = hs[torch.from_numpy(idx)]
            p_outs = variance_predict(hs_lr, sd, "pitch")
            e_outs = variance_predict(hs_lr, sd, "energy")
            hs_va = hs_lr + (variance_embed(ps[b, :ol], sd, "pitch") + variance_embed(es[b, :ol], sd, "energy"))
            before, mel = decode(sd, cfg, hs_va, return_before=True)
            l1_num += (mel - ys[b, :ol]).abs().sum().item(); l1_den += mel.numel()
            lb_num += (before - ys[b, :ol]).abs().sum().item()
            tgt = torch.log(ds[b, :il].float() + DUR_OFFSET)
            d_num += ((logd - tgt) ** 2).sum().item(); d_den += il
            p_num += ((p_outs - ps[b, :ol]) ** 2).sum().item(); p_den += p_outs.numel()
            e_num += ((e_outs - es[b, :ol]) ** 2).sum().item(); e_den += e_outs.numel()
    l1 = l1_num / l1_den
    if cfg["postnet_layers"] != 0:              # network/fastspeech2.py:271-274
        l1 += lb_num / l1_den
    out = dict(l1_loss=l1, duration_loss=d_num / d_den,
               pitch_loss=p_num / p_den, energy_loss=e_num / e_den)
    out["loss"] = out["l1_loss"] + out["duration_loss"] + out["pitch_loss"] + out["energy_loss"]
    return out


def train_grads(sd, cfg, xs, ilens, ys, olens, ds, ps, es):
    """Loss of `train_loss` with autograd through the functional restatement: returns
    (loss terms, {parameter name: d loss / d parameter}).  What `loss.backward()` leaves in
    `.grad` in run/train_decode.py:94-97 (accum_grad = 1), dropout off (eval-mode modules),
    With a PostNet the BatchNorm statistics couple the utterances (`postnet_train`): the graph of the whole batch is
    kept and differentiated once; `terms["running"]` then holds the updated running statistics."""
    assert cfg["use_masking"]
    leaf = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    B = len(ilens)
    n_mel = sum(int(olens[b]) for b in range(B)) * cfg["odim"]
    n_dur = sum(int(ilens[b]) for b in range(B))
    n_fr = sum(int(olens[b]) for b in range(B))
    terms = dict(l1_loss=0.0, duration_loss=0.0, pitch_loss=0.0, energy_loss=0.0)
    total = 0.0
    has_post = cfg["postnet_layers"] != 0
    befores, pending = [], []
    for b in range(B):
        il, ol = int(ilens[b]), int(olens[b])
        hs = encode(leaf, cfg, xs[b, :il])
        logd = duration_log(hs, leaf)
        idx, _ = length_regulate_index(ds[b, :il].numpy())
        hs_lr = hs[torch.from_numpy(idx)]
        p_outs = variance_predict(hs_lr, leaf, "pitch")
        e_outs = variance_predict(hs_lr, leaf, "energy")
        hs_va = hs_lr + (variance_embed(ps[b, :ol], leaf, "pitch") + variance_embed(es[b, :ol], leaf, "energy"))
        if has_post:
            zs = fft_stack(hs_va, leaf, "decoder.", has_embedding=False, attens=None, **_dec_kw(cfg))
            mel = F.linear(zs, leaf["feat_out.weight"], leaf["feat_out.bias"]).view(-1, cfg["odim"])
            befores.append(mel)
        else:
            mel = decode(leaf, cfg, hs_va)
        tgt = torch.log(ds[b, :il].float() + DUR_OFFSET)
        parts = dict(l1_loss=(mel - ys[b, :ol]).abs().sum() / n_mel,
                     duration_loss=((logd - tgt) ** 2).sum() / n_dur,
                     pitch_loss=((p_outs - ps[b, :ol]) ** 2).sum() / (n_fr * p_outs.shape[-1]),
                     energy_loss=((e_outs - es[b, :ol]) ** 2).sum() / (n_fr * e_outs.shape[-1]))
        loss_b = sum(parts.values())
        if has_post:
            pending.append(loss_b)
        else:
            loss_b.backward()                      # per utterance: keeps the graph small
        for k, v in parts.items():
            terms[k] += float(v.detach())
        total += float(loss_b.detach())
    if has_post:                                   # network/fastspeech2.py:222-225, 271-274
        running = {}
        res = postnet_train(leaf, cfg, befores, running)
        l1_after = sum(((bf + r) - ys[b, :int(olens[b])]).abs().sum() for b, (bf, r) in enumerate(zip(befores, res))) / n_mel
        (sum(pending) + l1_after).backward()
        terms["l1_loss"] += float(l1_after.detach())
        total += float(l1_after.detach())
        terms["running"] = running
    terms["loss"] = total
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()
             if v.dtype.is_floating_point}
    return terms, grads
