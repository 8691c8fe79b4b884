"""`FeedForwardTransformer` — drop-in for the reference model class (network/fastspeech2.py:57-352).

Same constructor kwargs, same parameter names/shapes (state_dict-compatible with the repaired
reference) and the same default initialisation order, so `torch.manual_seed(s)`
gives the same random-init weights.  The sub-modules are PARAMETER HOLDERS: the forward pass does
not go through them but through `fastspeech2_amd.engine.Engine`, i.e. the gfx950 HIP kernels of
libfs2_gfx950.so.  There is no CPU fallback — without the library or a GPU the forward raises.

Semantics = the repaired reference: every utterance of a batch is computed as the
reference computes it alone (batch-1 `inference`), padded frames are returned as zeros.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import synth
from .loss import DurationPredictorLoss
from ..trace import span
from .utils import Reporter, initialize, make_non_pad_mask

LN_EPS = 1e-12  # network/utils.py:41


class _Holder(nn.Module):
    """Plain container; children are attached by the builder functions below."""


class _PosEnc(nn.Module):
    """Holds `alpha` for ScaledPositionalEncoding (network/encoder.py:131-145, repair D1).  The
    positional table itself is built by the engine (network/encoder.py:116-121 recipe)."""

    def __init__(self, scaled: bool):
        super().__init__()
        if scaled:
            self.alpha = nn.Parameter(torch.tensor(1.0))


class _ConvNorm(nn.Module):
    """Conv1d + xavier_normal_ init, gain('linear') = 1 (network/utils.py:7-36)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, padding=(k - 1) // 2)
        nn.init.xavier_normal_(self.conv.weight, gain=nn.init.calculate_gain("linear"))


def _fft_stack(d, heads, units, n_layers, pre_ln, concat_after, ff_type, k, scaled, input_layer):
    """Parameter tree of network/encoder.py:192-246 (with D2: layers in a ModuleList)."""
    assert d % heads == 0                                             # network/encoder.py:53
    st = _Holder()
    st.embed = nn.Sequential(input_layer, _PosEnc(scaled)) if input_layer is not None \
        else nn.Sequential(_PosEnc(scaled))
    if ff_type not in ("linear", "conv1d"):
        raise NotImplementedError("not implemented")                  # network/encoder.py:232
    layers = []
    for _ in range(n_layers):
        att = _Holder()
        att.linear_q, att.linear_k = nn.Linear(d, d), nn.Linear(d, d)
        att.linear_v, att.linear_out = nn.Linear(d, d), nn.Linear(d, d)
        ff = _Holder()
        if ff_type == "conv1d":
            ff.w1 = nn.Conv1d(d, units, k, padding=(k - 1) // 2)
            ff.w2 = nn.Conv1d(units, d, k, padding=(k - 1) // 2)
        else:
            ff.w1, ff.w2 = nn.Linear(d, units), nn.Linear(units, d)
        layer = _Holder()
        layer.self_atten, layer.feed_forward = att, ff
        layer.norm1, layer.norm2 = nn.LayerNorm(d, eps=LN_EPS), nn.LayerNorm(d, eps=LN_EPS)
        if concat_after:
            layer.concat_linear = nn.Linear(2 * d, d)
        layers.append(layer)
    st.encoders = nn.ModuleList(layers)
    if pre_ln:
        st.after_norm = nn.LayerNorm(d, eps=LN_EPS)
    return st


def _predictor(in_dim, n_layers, chans, k, out_dim, dropout):
    """Conv stack of Duration/VariancePredictor (network/variance_adaptor.py:55-78, :93-115)."""
    pr = _Holder()
    pr.conv = nn.ModuleList([
        nn.Sequential(_ConvNorm(in_dim if j == 0 else chans, chans, k), nn.ReLU(),
                      nn.LayerNorm(chans, eps=LN_EPS), nn.Dropout(dropout))
        for j in range(n_layers)])
    pr.linear = nn.Linear(chans, out_dim)
    return pr


def _embedder(in_dim, out_dim, k):
    e = _Holder()
    e.conv = _ConvNorm(in_dim, out_dim, k)
    return e


def _postnet(odim, n_layers, chans, filts, use_batch_norm, dropout):
    pn = _Holder()
    layers = []
    for j in range(n_layers):
        cin = odim if j == 0 else chans
        cout = odim if j == n_layers - 1 else chans
        mods = [nn.Conv1d(cin, cout, filts, padding=(filts - 1) // 2, bias=False)]
        if use_batch_norm:
            mods.append(nn.BatchNorm1d(cout))
        if j != n_layers - 1:
            mods.append(nn.Tanh())
        mods.append(nn.Dropout(dropout))
        layers.append(nn.Sequential(*mods))
    pn.postnet = nn.ModuleList(layers)
    return pn


class _TrainStep(torch.autograd.Function):
    """One autograd node for the whole teacher-forced step: forward and backward are explicit HIP
    launch schedules (fastspeech2_amd/train_engine.py); the parameters are its differentiable inputs."""

    @staticmethod
    def forward(ctx, model, eng, batch, names, *params):
        ids, il, ol, ds_p, ps_p, es_p, ys_p = batch
        with span("fs2.train_forward"):
            outs, saved = eng.forward(ids, il, ol, ds_p, ps_p, es_p, ys_p)
        l1_loss = F.l1_loss(outs["before"], ys_p)                              # network/fastspeech2.py:271-274
        if "after" in outs:
            l1_loss = F.l1_loss(outs["after"], ys_p) + l1_loss
        duration_loss = model.duration_criterion(outs["log_durations"], ds_p)  # network/loss.py:5-14
        pitch_loss = model.mse_criterion(outs["pitch"], ps_p)
        energy_loss = model.mse_criterion(outs["energy"], es_p)
        loss = l1_loss + duration_loss + pitch_loss + energy_loss
        report_keys = [{"l1_loss": l1_loss}, {"duration_loss": duration_loss}, {"pitch_loss": pitch_loss},
                       {"energy_loss": energy_loss}, {"loss": loss.detach()}]
        if model.use_scaled_pos_enc:
            report_keys += [{"encoder_alpha": model.encoder.embed[-1].alpha.data},
                            {"decoder_alpha": model.decoder.embed[-1].alpha.data}]
        model.reporter.report(report_keys)
        ctx.eng, ctx.saved, ctx.names = eng, saved, names
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        """Parameter gradients are handed over WITHOUT a copy: `p.grad` becomes a view into the engine's flat
        gradient buffer (the buffer the data-parallel reducer all-reduces), which the next backward
        overwrites — after `zero_grad()` (set_to_none, the torch default) that is invisible.  If gradients
        are still held (accum_grad > 1, run/train_decode.py:94-102) the old values are added back."""
        eng = ctx.eng
        up = grad_loss.detach().reshape(1).to(torch.float32).contiguous()
        params = [(n, eng.P[n]) for n in ctx.names if eng.P[n].requires_grad]
        held = [(n, p) for n, p in params if p.grad is not None]
        aliased = any(p.grad.data_ptr() == eng.G[n].data_ptr() for n, p in held)
        keep = eng.flat.clone() if aliased else None
        with span("fs2.train_backward"):
            G = eng.backward(ctx.saved, up)
        ctx.saved = None
        if keep is not None:
            eng.flat.add_(keep)
        for n, p in params:
            if p.grad is None:
                p.grad = G[n]
            elif p.grad.data_ptr() != G[n].data_ptr():
                p.grad.add_(G[n])
        return (None, None, None, None) + (None,) * len(ctx.names)


class FeedForwardTransformer(nn.Module):
    def __init__(self,
                 idim: int,
                 odim: int,
                 adim: int = 384,
                 aheads: int = 4,
                 positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 3,
                 eunits: int = 1536,
                 elayers: int = 6,
                 transformer_enc_dropout_rate: float = 0.0,
                 transformer_enc_positional_dropout_rate: float = 0.0,
                 transformer_enc_atten_dropout_rate: float = 0.0,
                 encoder_normalized_before: bool = True,
                 encoder_concat_after: bool = False,
                 pitch_embed_kernel_size: int = 1,
                 pitch_embed_dropout: float = 0.0,
                 energy_embed_kernel_size: int = 1,
                 energy_embed_dropout: float = 0.0,
                 duration_predictor_layers: int = 2,
                 duration_predictor_chans: int = 256,
                 duration_predictor_kernel_size: int = 3,
                 duration_predictor_dropout_rate: float = 0.1,
                 dlayers: int = 6,
                 dunits: int = 1536,
                 transformer_dec_dropout_rate: float = 0.1,
                 transformer_dec_positional_dropout_rate: float = 0.1,
                 transformer_dec_atten_dropout_rate: float = 0.1,
                 decoder_normalized_before: bool = False,
                 decoder_concat_after: bool = False,
                 reduction_factor: int = 1,
                 postnet_layers: int = 5,
                 postnet_filts: int = 5,
                 postnet_chans: int = 256,
                 postnet_dropout_rate: float = 0.5,
                 transformer_init: str = "pytorch",
                 initial_encoder_alpha: float = 1.0,
                 initial_decoder_alpha: float = 1.0,
                 use_masking: bool = True,
                 use_batch_norm: bool = True,
                 use_scaled_pos_enc: bool = True):
        super().__init__()
        self.idim, self.odim = idim, odim
        self.use_scaled_pos_enc = use_scaled_pos_enc
        self.reduction_factor = reduction_factor
        self.use_masking = use_masking
        self.reporter = Reporter()
        # nn.Dropout rates of the reference's sub-modules (training path only; network/fastspeech2.py:122-167)
        self.dropout_rates = dict(enc=transformer_enc_dropout_rate, enc_pos=transformer_enc_positional_dropout_rate,
                                  enc_att=transformer_enc_atten_dropout_rate, dec=transformer_dec_dropout_rate,
                                  dec_pos=transformer_dec_positional_dropout_rate,
                                  dec_att=transformer_dec_atten_dropout_rate, dur=duration_predictor_dropout_rate,
                                  var=0.5, pitch_embed=pitch_embed_dropout, energy_embed=energy_embed_dropout,
                                  postnet=postnet_dropout_rate)
        self.dropout_enabled = True       # False: deterministic training step (no dropout)
        self.cfg = dict(
            idim=idim, odim=odim, adim=adim, aheads=aheads, positionwise_layer_type=positionwise_layer_type,
            positionwise_conv_kernel_size=positionwise_conv_kernel_size, eunits=eunits, elayers=elayers,
            encoder_normalized_before=encoder_normalized_before, encoder_concat_after=encoder_concat_after,
            duration_predictor_layers=duration_predictor_layers, duration_predictor_chans=duration_predictor_chans,
            duration_predictor_kernel_size=duration_predictor_kernel_size, dlayers=dlayers, dunits=dunits,
            decoder_normalized_before=decoder_normalized_before, decoder_concat_after=decoder_concat_after,
            reduction_factor=reduction_factor, postnet_layers=postnet_layers, use_batch_norm=use_batch_norm,
            use_masking=use_masking,
            use_scaled_pos_enc=use_scaled_pos_enc)

        # same construction order as network/fastspeech2.py:116-167 -> same RNG consumption
        emb = nn.Embedding(idim, adim, padding_idx=0)
        self.encoder = _fft_stack(adim, aheads, eunits, elayers, encoder_normalized_before, encoder_concat_after,
                                  positionwise_layer_type, positionwise_conv_kernel_size, use_scaled_pos_enc, emb)
        va = _Holder()
        va.duration_predictor = _predictor(adim, duration_predictor_layers, duration_predictor_chans,
                                           duration_predictor_kernel_size, 1, duration_predictor_dropout_rate)
        va.pitch_predictor = _predictor(adim, synth.VAR_PRED_LAYERS, synth.VAR_PRED_CHANS, synth.VAR_PRED_KERNEL,
                                        synth.PITCH_DIM, 0.5)
        va.pitch_embed = _embedder(synth.PITCH_DIM, adim, pitch_embed_kernel_size)
        va.energy_predictor = _predictor(adim, synth.VAR_PRED_LAYERS, synth.VAR_PRED_CHANS, synth.VAR_PRED_KERNEL,
                                         synth.ENERGY_DIM, 0.5)
        va.energy_embed = _embedder(synth.ENERGY_DIM, adim, energy_embed_kernel_size)
        self.variance_adaptor = va
        self.decoder = _fft_stack(adim, aheads, dunits, dlayers, decoder_normalized_before, decoder_concat_after,
                                  positionwise_layer_type, positionwise_conv_kernel_size, use_scaled_pos_enc, None)
        self.feat_out = nn.Linear(adim, odim * reduction_factor)
        # The reference PostNet cannot run as written (n-1 layers, idim input channels, never
        # emits odim).  This is the repaired ESPnet layout: n x [Conv1d(k, bias=False), BatchNorm1d, tanh
        # (not on the last layer), dropout], odim -> chans -> ... -> odim; `after = before + postnet(before)`.
        self.postnet = None if postnet_layers == 0 else _postnet(odim, postnet_layers, postnet_chans, postnet_filts,
                                                                 use_batch_norm, postnet_dropout_rate)
        self.postnet_layers = postnet_layers

        initialize(self, transformer_init)
        if use_scaled_pos_enc:
            self.encoder.embed[-1].alpha.data = torch.tensor(initial_encoder_alpha)
            self.decoder.embed[-1].alpha.data = torch.tensor(initial_decoder_alpha)
        self.duration_criterion = DurationPredictorLoss()
        self.mse_criterion = nn.MSELoss()

        self._precision = ("fp32", None)
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------ engine plumbing
    def set_precision(self, dtype: str = "fp32", encoder_dtype: Optional[str] = None):
        """'fp32' (parity mode, exact-fp32 MFMA), 'bf16' (bf16 MFMA operands) or 'f16x3' (every GEMM and the attention
        on error-compensated fp16 operand pairs: fp32-grade results - mel within the fp32 bar - at a fraction of the
        exact-fp32 time).  encoder_dtype='fp32' / 'f16x3' keeps the phone-level half - which fixes the integer
        durations - at that precision under dtype='bf16'."""
        assert dtype in ("fp32", "bf16", "f16x3") and encoder_dtype in (None, "fp32", "bf16", "f16x3")
        self._precision = (dtype, encoder_dtype)
        self._engine = None
        return self

    def engine(self):
        from ..engine import Engine
        params = list(self.parameters()) + list(self.buffers())
        dev = params[0].device
        key = (self._precision, str(dev), tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if self._engine is None or key != self._engine_key:
            if dev.type != "cuda":
                from ..hip import HipError
                raise HipError("model parameters are on %s: move the model to the MI355X (`model.cuda()`); the "
                               "forward path has no CPU implementation" % dev)
            cfg = dict(self.cfg)
            self._engine = Engine(self.state_dict(), cfg, dtype=self._precision[0],
                                  encoder_dtype=self._precision[1], device=dev)
            self._engine_key = key
        return self._engine

    @staticmethod
    def _pack(x: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
        """padded (B, T, ...) -> packed (sum lens, ...)"""
        B, T = x.shape[0], x.shape[1]
        lens_t = torch.as_tensor(list(lens), dtype=torch.int64)
        mask = (torch.arange(T).unsqueeze(0) < lens_t.unsqueeze(1)).to(x.device)
        return x[mask]

    def _source_mask(self, ilens: torch.LongTensor):
        """(B, T, T) outer-product mask (network/fastspeech2.py:174-176) — kept for callers; the
        kernels use row ranges instead."""
        x_masks = make_non_pad_mask(ilens).to(self.feat_out.weight.device)
        return x_masks.unsqueeze(-2) & x_masks.unsqueeze(-1)

    # ------------------------------------------------------------------ reference API
    def _forward(self, xs, ilens, olens=None, ds=None, ps=None, es=None, in_masks=None, out_masks=None,
                 is_inference=False, duration_control: float = 1.0, pitch_control: float = 1.0,
                 energy_control: float = 1.0, _raw=False):
        """network/fastspeech2.py:187-230.  Masks are accepted for signature compatibility; validity
        comes from ilens (and the frame counts).  Padded positions of every output are zero."""
        eng = self.engine()
        il = [int(v) for v in ilens]
        B = len(il)
        ids = self._pack(xs.to(eng.dev), il).to(torch.int64).contiguous()
        with torch.no_grad():
            if is_inference:
                out = eng.forward(ids, il, duration_control=duration_control, pitch_control=pitch_control,
                                  energy_control=energy_control)
            else:
                if self.reduction_factor != 1:
                    # the reference builds the decoder masks from olens // r but length-regulates to olens frames
                    # (network/fastspeech2.py:209-215): not self-consistent, nothing to pin a result to
                    raise NotImplementedError("reduction_factor > 1 is supported for inference only")
                ol = [int(v) for v in olens]
                ds_p = self._pack(ds.to(eng.dev), il).to(torch.int32).contiguous()
                if [int(v) for v in self._segment_sums(ds_p, il)] != ol:
                    raise ValueError("sum(ds) per utterance must equal olens")
                ps_p = self._pack(ps.to(eng.dev), ol).float().contiguous()
                es_p = self._pack(es.to(eng.dev), ol).float().contiguous()
                out = eng.forward(ids, il, ds=ds_p, ps=ps_p, es=es_p)
            Tmax = max(out["olens"]) if B else 0
            Tmel = max(out["mel_lens"]) if B else 0          # = reduction_factor x the decoder frames
            after = eng.unpack(out["mel"], out["cu_mel"], B, Tmel)
            before = after if self.postnet is None else eng.unpack(out["mel_before"], out["cu_mel"], B, Tmel)
        if _raw:
            return before, after, out
        if is_inference:
            return before, after
        Tp = max(il)
        d_outs = eng.unpack(out["log_durations"].view(-1, 1), out["cu_phone"], B, Tp).squeeze(-1)
        p_outs = eng.unpack(out["pitch"], out["cu_frames"], B, Tmax)
        e_outs = eng.unpack(out["energy"], out["cu_frames"], B, Tmax)
        return before, after, d_outs, p_outs, e_outs

    @staticmethod
    def _segment_sums(packed: torch.Tensor, lens: Sequence[int]) -> List[int]:
        v = packed.cpu().numpy().astype(np.int64)
        cu = np.concatenate([[0], np.cumsum(lens)])
        return [int(v[cu[i]:cu[i + 1]].sum()) for i in range(len(lens))]

    def train_engine(self):
        """Training-step launch schedule (forward + backward kernels) bound to this module's parameters."""
        from ..train_engine import TrainEngine
        plist = getattr(self, "_param_list", None)
        if plist is None:                                  # the Parameter objects are fixed after construction;
            plist = self._param_list = list(self.parameters())   # walking the module tree costs 0.3 ms per call
        key = (self._precision[0], tuple(p.data_ptr() for p in plist))
        if getattr(self, "_train_engine", None) is None or self._train_engine_key != key:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                from ..hip import HipError
                raise HipError("model parameters are on %s: move the model to the MI355X (`model.cuda()`); the "
                               "training path has no CPU implementation" % dev)
            self._train_engine = TrainEngine(self, dtype=self._precision[0])
            self._train_engine_key = key
        return self._train_engine

    def pack_batch(self, xs, ilens, ys, olens, ds, ps, es):
        """Padded training batch (the converter's tensors, run/train_decode.py:126-149) -> packed device
        tensors, validated once (sum(ds) == olens per utterance).  The result can be fed to `forward` as
        `packed=` on every step of a loop that re-uses the batch, skipping the re-packing."""
        eng = self.train_engine()
        il, ol = [int(v) for v in ilens], [int(v) for v in olens]
        dev = eng.dev
        ids = self._pack(xs[:, :max(il)].to(dev), il).to(torch.int64).contiguous()
        ds_p = self._pack(ds[:, :max(il)].to(dev), il).to(torch.int32).contiguous()
        if [int(v) for v in self._segment_sums(ds_p, il)] != ol:
            raise ValueError("sum(ds) per utterance must equal olens")
        ys_p = self._pack(ys[:, :max(ol)].to(dev), ol).float().contiguous()
        ps_p = self._pack(ps[:, :max(ol)].to(dev), ol).float().contiguous()
        es_p = self._pack(es[:, :max(ol)].to(dev), ol).float().contiguous()
        return (ids, il, ol, ds_p, ps_p, es_p, ys_p)

    def _forward_train(self, packed):
        """Teacher-forced step with a backward: the loss tensor is the output of one autograd node whose
        backward runs the HIP backward schedule and hands every parameter its gradient, so the reference's
        `loss.backward(); clip_grad_norm_; optimizer.step()` (run/train_decode.py:94-110) works unchanged."""
        eng = self.train_engine()
        names = list(eng.P)
        return _TrainStep.apply(self, eng, packed, names, *[eng.P[n] for n in names])

    def forward(self, xs=None, ilens=None, ys=None, olens=None, ds=None, ps=None, es=None, packed=None, **_ignored):
        """Loss of one teacher-forced step (network/fastspeech2.py:232-295).  Extra keys produced by
        the reference's converter (`labels`, `spkids`, run/train_decode.py:136-149) are ignored.
        In training mode with grad enabled the returned loss carries a backward (HIP kernels, see
        `_forward_train`) and dropout follows the constructor's rates (`dropout_enabled = False` turns it off);
        under `torch.no_grad()` / `eval()` it is the plain value (evaluator path, run/train_decode.py:42-57)."""
        if self.training and torch.is_grad_enabled():
            return self._forward_train(packed if packed is not None else self.pack_batch(xs, ilens, ys, olens, ds, ps, es))
        if xs is None:
            raise ValueError("`packed=` batches are for the training path (train mode, grad enabled)")
        il, ol = [int(v) for v in ilens], [int(v) for v in olens]
        xs, ds = xs[:, :max(il)], ds[:, :max(il)]
        ys, ps, es = ys[:, :max(ol)], ps[:, :max(ol)], es[:, :max(ol)]
        _, _, out = self._forward(xs, ilens, olens, ds, ps, es, is_inference=False, _raw=True)
        dev = out["mel"].device
        if not self.use_masking:
            raise NotImplementedError("use_masking=False (loss over padded frames) is not supported: "
                                      "padded frames do not exist in the packed layout")
        ys_p = self._pack(ys.to(dev), ol).float()
        ps_p = self._pack(ps.to(dev), ol).float()
        es_p = self._pack(es.to(dev), ol).float()
        ds_p = self._pack(ds.to(dev), il)
        l1_loss = F.l1_loss(out["mel"], ys_p)
        if self.postnet is not None:                                   # network/fastspeech2.py:271-274
            l1_loss = l1_loss + F.l1_loss(out["mel_before"], ys_p)
        duration_loss = self.duration_criterion(out["log_durations"], ds_p)
        pitch_loss = self.mse_criterion(out["pitch"], ps_p)
        energy_loss = self.mse_criterion(out["energy"], es_p)
        loss = l1_loss + duration_loss + pitch_loss + energy_loss
        report_keys = [{"l1_loss": l1_loss.item()}, {"duration_loss": duration_loss.item()},
                       {"pitch_loss": pitch_loss.item()}, {"energy_loss": energy_loss.item()},
                       {"loss": loss.item()}]
        if self.use_scaled_pos_enc:
            report_keys += [{"encoder_alpha": self.encoder.embed[-1].alpha.data.item()},
                            {"decoder_alpha": self.decoder.embed[-1].alpha.data.item()}]
        self.reporter.report(report_keys)
        return loss

    def inference(self, x: torch.LongTensor, y: torch.Tensor = None, duration_control: float = 1.0,
                  pitch_control: float = 1.0, energy_control: float = 1.0):
        """One utterance, phone ids (Tp,) -> mel (Tm, odim)  (network/fastspeech2.py:297-302)."""
        if getattr(self, "_graph_cap", 0) and x.shape[0] > 0:
            eng = self.engine()
            gi = getattr(self, "_graphed", None)
            if gi is None or gi.eng is not eng:          # the engine was rebuilt (weights / precision changed)
                from ..engine import GraphedInference
                gi = self._graphed = GraphedInference(eng, self._graph_cap)
            with torch.no_grad():
                out = gi(x.to(eng.dev, torch.int64), duration_control, pitch_control, energy_control)
            return out["mel"].clone()
        ilens = torch.LongTensor([x.shape[0]])
        _, outs = self._forward(x.unsqueeze(0), ilens, is_inference=True, duration_control=duration_control,
                                pitch_control=pitch_control, energy_control=energy_control)
        return outs[0]

    def enable_inference_graphs(self, max_graphs: int = 64):
        """Extension: `inference()` replays HIP graphs captured per utterance shape (engine.GraphedInference) instead
        of issuing its ~100 launches from Python: same kernels, same results, about half the batch-1 latency for
        shapes seen before.  max_graphs = 0 turns it off again."""
        if int(max_graphs) and self.reduction_factor != 1:
            raise NotImplementedError("inference graphs are captured for reduction_factor = 1")
        self._graph_cap = int(max_graphs)
        self._graphed = None
        return self

    def infer_batch(self, xs: Sequence[torch.Tensor], duration_control: float = 1.0, pitch_control: float = 1.0,
                    energy_control: float = 1.0):
        """Extension: a ragged batch in one pass; every row equals the batch-1
        `inference` result.  xs: list of 1-D id tensors.  Returns a dict with padded `mel`
        (B, Tmax, odim), `olens`, padded `durations` (B, Tp), `pitch`, `energy`, masks."""
        eng = self.engine()
        il = [int(x.shape[0]) for x in xs]
        ids = torch.cat([x.reshape(-1) for x in xs]).to(eng.dev, torch.int64).contiguous()
        with torch.no_grad():
            out = eng.forward(ids, il, duration_control=duration_control, pitch_control=pitch_control,
                              energy_control=energy_control)
            B, Tmax, Tp = len(il), max(out["olens"]), max(il)
            mel = eng.unpack(out["mel"], out["cu_mel"], B, max(out["mel_lens"]))
            dur = eng.unpack(out["durations"].float().view(-1, 1), out["cu_phone"], B, Tp).squeeze(-1).long()
            pitch = eng.unpack(out["pitch"], out["cu_frames"], B, Tmax)
            energy = eng.unpack(out["energy"], out["cu_frames"], B, Tmax)
        # olens: mel frames per utterance (= reduction_factor x the decoder frames that pitch / energy are given on)
        olens = torch.tensor(out["mel_lens"], dtype=torch.int64)
        return dict(mel=mel, olens=olens, durations=dur, pitch=pitch, energy=energy,
                    src_mask=make_non_pad_mask(il), mel_mask=make_non_pad_mask(out["mel_lens"]), packed=out)

    def calculate_all_attentions(self, xs, ilens, ys, olens, ds, ps, es, **_ignored):
        """network/fastspeech2.py:305-341: dict name -> list over utterances of (h, T, T) arrays, plus
        'predicted_fbank'.  Uses the debug kernel that materialises the softmax matrices."""
        eng = self.engine()
        il, ol = [int(v) for v in ilens], [int(v) for v in olens]
        xs, ds = xs[:, :max(il)], ds[:, :max(il)]
        ps, es = ps[:, :max(ol)], es[:, :max(ol)]
        with torch.no_grad():
            ids = self._pack(xs.to(eng.dev), il).to(torch.int64).contiguous()
            out = eng.forward(ids, il, ds=self._pack(ds.to(eng.dev), il).to(torch.int32).contiguous(),
                              ps=self._pack(ps.to(eng.dev), ol).float().contiguous(),
                              es=self._pack(es.to(eng.dev), ol).float().contiguous(), want_attention=True)
        att_ws_dict = {}
        for side, key in (("encoder", "enc_attention"), ("decoder", "dec_attention")):
            for i, per_utt in enumerate(out[key]):
                att_ws_dict["%s.encoders.%d.self_atten" % (side, i)] = [a.cpu().numpy() for a in per_utt]
        mel = out["mel"].cpu().numpy()
        cu = np.concatenate([[0], np.cumsum(out["olens"])])
        att_ws_dict["predicted_fbank"] = [mel[cu[b]:cu[b + 1]].T for b in range(len(il))]
        return att_ws_dict

    @property
    def attention_plot_class(self):
        """The reference returns its chainer TTSPlot extension (network/fastspeech2.py:343-345);
        plotting/Chainer glue is out of scope, so there is no class to hand back."""
        return None

    @property
    def base_plot_keys(self):
        plot_keys = ["loss", "l1_loss", "duration_loss"]
        if self.use_scaled_pos_enc:
            plot_keys += ["encoder_alpha", "decoder_alpha"]
        return plot_keys
