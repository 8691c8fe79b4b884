"""Launch schedule of one teacher-forced training step (forward + backward) on the gfx950 kernels.

Forward follows `FeedForwardTransformer.forward` / `_forward` (network/fastspeech2.py:187-295) and
`VarianceAdaptor.forward` (network/variance_adaptor.py:178-193, repair D8); backward is what
`loss.backward()` (run/train_decode.py:94-97) computes through those modules, written out as an
explicit sequence of C-ABI launches (include/fs2_hip.h "Training path").  Dropout (every nn.Dropout of the
reference's encoder / decoder / predictors) is counter based: masks are re-derived from (step seed, site, row,
col) in the backward kernels, nothing is stored; `model.dropout_enabled = False` gives the deterministic step.

Weights stay in the model's fp32 `nn.Parameter`s (the optimiser's master copy).  Every step the
GEMM operands are re-packed from them by `fs2_pack_weight` into persistent buffers: `fwd`
([tap][N][C], forward and weight-gradient layout) and `bwd` ([tap][C][N], flipped taps: the operand of
the input-gradient GEMM).  Parameter gradients are written (fp32, the parameters' own layouts) into
one flat buffer laid out in the order the backward finishes them, so data-parallel training can
all-reduce contiguous buckets while the rest of the backward is still running (`grad_ready` hook).
"""
import math
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hip
from .engine import LN_EPS, Engine, posenc_table
from .hip import BF16, F32, call, ptr, stream


class _Weight:
    """One GEMM weight: parameter(s) in reference layout -> packed fwd / bwd operands."""

    def __init__(self, params: Sequence[torch.nn.Parameter], tdt, dev, pad_n: int = 0):
        p0 = params[0]
        self.params = list(params)
        self.k = p0.shape[2] if p0.dim() == 3 else 1
        self.n_each = p0.shape[0]
        self.N = self.n_each * len(params)
        self.C = p0.shape[1]
        self.Np = max(self.N, pad_n)             # row length of the bwd operand (zero padded)
        self.fwd = torch.zeros((self.k, self.N, self.C), device=dev, dtype=tdt)
        self.bwd = torch.zeros((self.k, self.C, self.Np), device=dev, dtype=tdt)

    def pack(self, code):
        for j, p in enumerate(self.params):
            esz = self.fwd.element_size()
            f = self.fwd.data_ptr() + j * self.n_each * self.C * esz
            b = self.bwd.data_ptr() + j * self.n_each * esz
            call("fs2_pack_weight", ptr(p.data), hip.c_void_p(f), hip.c_void_p(b), self.n_each, self.C, self.k, self.C,
                 self.Np, self.N * self.C, self.C * self.Np, code, stream())

    def descriptors(self):
        """(src, fwd, bwd, N, C, k, ld_fwd, ld_bwd, 0, tap_stride_fwd, tap_stride_bwd) per parameter: fs2_pack_desc"""
        esz = self.fwd.element_size()
        return [(p.data.data_ptr(), self.fwd.data_ptr() + j * self.n_each * self.C * esz,
                 self.bwd.data_ptr() + j * self.n_each * esz, self.n_each, self.C, self.k, self.C, self.Np, 0,
                 self.N * self.C, self.C * self.Np) for j, p in enumerate(self.params)]


PACK_DESC = np.dtype([("src", "<u8"), ("fwd", "<u8"), ("bwd", "<u8"), ("N", "<i4"), ("C", "<i4"), ("k", "<i4"),
                      ("ld_fwd", "<i4"), ("ld_bwd", "<i4"), ("reserved", "<i4"), ("ts_fwd", "<i8"), ("ts_bwd", "<i8")])
assert PACK_DESC.itemsize == 64          # include/fs2_pack_desc.h


class _PackGroup:
    """The weights of one refresh group as one fs2_pack_weights launch; the device-side table is rebuilt when a
    parameter's storage moved (model.to(), load_state_dict(assign=True), ...)."""

    def __init__(self, weights: Sequence[_Weight], dev):
        self.weights = list(weights)
        self.dev = dev
        self.key = None
        self.table = self.start = None
        self.n = self.blocks = 0

    def launch(self, code):
        key = tuple(p.data.data_ptr() for w in self.weights for p in w.params)
        if key != self.key:
            rows = [r for w in self.weights for r in w.descriptors()]
            desc = np.array(rows, dtype=PACK_DESC)
            start = np.zeros(len(rows) + 1, np.int32)
            start[1:] = np.cumsum([2 * (-(-(r[3] * r[4]) // 256)) for r in rows])
            self.table = torch.from_numpy(desc.view(np.uint8).copy()).to(self.dev)
            self.start = torch.from_numpy(start).to(self.dev)
            self.n, self.blocks, self.key = len(rows), int(start[-1]), key
        call("fs2_pack_weights", ptr(self.table), ptr(self.start), self.n, self.blocks, code, stream())


_HOLD = [] if os.environ.get("FS2_TRAIN_HOLD", "0") == "1" else None


class TrainEngine:
    SITE_ENC_POS, SITE_DEC_POS = (2 << 16) + 1, (2 << 16) + 2
    SITE_PITCH_EMB, SITE_ENERGY_EMB = (2 << 16) + 3, (2 << 16) + 4

    def __init__(self, model, dtype: str = "fp32"):
        self.dev = hip.require_gpu()
        hip.load()
        self.model = model
        cfg = model.cfg
        self.cfg = cfg
        if dtype not in ("fp32", "bf16"):
            raise NotImplementedError("the training step runs in 'fp32' or 'bf16' (set_precision(%r) is an inference mode)" % dtype)
        self.code = hip.DTYPE_CODE[dtype]
        self.tdt = hip.TORCH_DTYPE[self.code]
        self.epv = 4 if self.code == F32 else 8
        d = cfg["adim"]
        self.d, self.h = d, cfg["aheads"]
        self.dk = d // self.h
        if self.dk not in (32, 64, 96, 128):
            raise NotImplementedError("head dim %d: the attention kernels support 32/64/96/128" % self.dk)
        if cfg["reduction_factor"] != 1:
            raise NotImplementedError("reduction_factor != 1")
        if not cfg["use_masking"]:
            raise NotImplementedError("use_masking=False")
        self.P: Dict[str, torch.nn.Parameter] = dict(model.named_parameters())
        self.buffers: Dict[str, torch.Tensor] = dict(model.named_buffers())      # BatchNorm running statistics
        for k, p in self.P.items():
            if p.device != self.dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise hip.HipError("parameter %s must be a contiguous fp32 tensor on %s" % (k, self.dev))
        self.scaled = cfg["use_scaled_pos_enc"]
        self.odim = cfg["odim"]
        self.idim = cfg["idim"]
        self.odim_pad = -(-self.odim // self.epv) * self.epv
        self._build_weights()
        self._build_grad_buffer()
        # dropout sites (rates of the reference's constructor; 0 disables a site).  model.dropout_enabled = False
        # switches every site off (deterministic step, used for the parity tests against eval-mode goldens)
        self.rates = dict(model.dropout_rates)
        # dropout masks = hash(seed of the step, site, row, col); the step seed mixes a user seed (`set_seed`), the
        # data-parallel rank (ranks must not draw the same masks) and the step counter (saved in snapshots)
        self.base_seed = 0x5EED
        self.rank_seed = int(os.environ.get("RANK", "0") or 0)
        self.step = 0
        self.seed = 0
        self._pe = None
        self.grad_ready: Optional[Callable[[int], None]] = None   # called with the index of a finished bucket
        self.grad_finish: Optional[Callable[[], None]] = None     # called when every bucket has been announced
        self.profile = None
        # None (FS2_TRAIN_SIDE_STREAM=0): weight gradients and weight packing on the main stream (debugging aid)
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) if os.environ.get("FS2_TRAIN_SIDE_STREAM", "1") != "0" else None
        self._pack_groups = None
        self._side_raw = self.wgrad_stream.cuda_stream if self.wgrad_stream is not None else None   # raw hipStream_t of the side stream
        self._ws_cache = {}
        self._wgrad_pending = False
        # Weight-gradient launches wait in a queue and go to the side stream a few at a time behind ONE event: a fork
        # (event record + wait on the other stream) costs ~24 us of host time with kernels in flight
        # (tools/microbench/event_cost.py) and the step is bound by the host; FS2_TRAIN_WGRAD_BATCH=1: one fork per launch
        self._wq = []
        self.wgrad_batch = max(1, int(os.environ.get("FS2_TRAIN_WGRAD_BATCH", "4") or 4))
        self._wgrad_sync = os.environ.get("FS2_TRAIN_WGRAD_SYNC", "0") == "1"   # debugging aid: the streams never run side by side
        self._size_cache = {}

    # ------------------------------------------------------------------ set-up
    def _stack_weights(self, prefix, n_layers, concat_after):
        P, tdt, dev = self.P, self.tdt, self.dev
        layers = []
        for i in range(n_layers):
            p = "%sencoders.%d." % (prefix, i)
            L = dict(prefix=p)
            L["wqkv"] = _Weight([P[p + "self_atten.linear_%s.weight" % n] for n in "qkv"], tdt, dev)
            L["wo"] = _Weight([P[p + "self_atten.linear_out.weight"]], tdt, dev)
            L["w1"] = _Weight([P[p + "feed_forward.w1.weight"]], tdt, dev)
            L["w2"] = _Weight([P[p + "feed_forward.w2.weight"]], tdt, dev)
            L["bqkv"] = torch.empty((3 * self.d,), device=dev)
            if concat_after:
                L["wc_x"] = _Weight([torch.nn.Parameter(P[p + "concat_linear.weight"][:, :self.d].contiguous())], tdt, dev)
                L["wc_a"] = _Weight([torch.nn.Parameter(P[p + "concat_linear.weight"][:, self.d:].contiguous())], tdt, dev)
            layers.append(L)
        return layers

    def _pred_weights(self, prefix):
        convs, j = [], 0
        while "%sconv.%d.0.conv.weight" % (prefix, j) in self.P:
            convs.append(dict(w=_Weight([self.P["%sconv.%d.0.conv.weight" % (prefix, j)]], self.tdt, self.dev),
                              p="%sconv.%d." % (prefix, j)))
            j += 1
        return dict(prefix=prefix, convs=convs, nout=self.P[prefix + "linear.weight"].shape[0])

    def _build_weights(self):
        cfg = self.cfg
        self.enc_layers = self._stack_weights("encoder.", cfg["elayers"], cfg["encoder_concat_after"])
        self.dec_layers = self._stack_weights("decoder.", cfg["dlayers"], cfg["decoder_concat_after"])
        va = "variance_adaptor."
        self.dur_pred = dict(self._pred_weights(va + "duration_predictor."), rate_key="dur", site=(2 << 16) + 16)
        self.pitch_pred = dict(self._pred_weights(va + "pitch_predictor."), rate_key="var", site=(2 << 16) + 32)
        self.energy_pred = dict(self._pred_weights(va + "energy_predictor."), rate_key="var", site=(2 << 16) + 48)
        self.w_out = _Weight([self.P["feat_out.weight"]], self.tdt, self.dev, pad_n=self.odim_pad)
        wp, we = self.P[va + "pitch_embed.conv.conv.weight"], self.P[va + "energy_embed.conv.conv.weight"]
        self.pdim, self.edim = wp.shape[1], we.shape[1]
        # pitch / energy_embed_kernel_size (network/variance_adaptor.py:35-51, any odd k): 1 = the reference default, folded
        # into the add kernel; larger kernels are convs over time inside each utterance (fp32 fs2_conv_gemm over inputs
        # padded to 4 channels, fs2_conv_wgrad for the gradients)
        self.embed_taps = (int(wp.shape[-1]), int(we.shape[-1]))
        if any(k % 2 == 0 or k > 9 for k in self.embed_taps):
            raise NotImplementedError("pitch/energy embed kernel sizes must be odd and <= 9")
        # repaired PostNet (network/fastspeech2.py:13-54): conv (no bias) -> BatchNorm -> tanh -> dropout
        self.post = []
        n = cfg["postnet_layers"]
        if n and self.odim % self.epv:
            raise NotImplementedError("PostNet training needs odim %% %d == 0" % self.epv)
        for j in range(n):
            pre = "postnet.postnet.%d." % j
            w = self.P[pre + "0.weight"]
            last = j == n - 1
            self.post.append(dict(p=pre, w=_Weight([w], self.tdt, self.dev, pad_n=self.odim_pad if last else 0),
                                  cin=w.shape[1], cout=w.shape[0], k=w.shape[2], bn=bool(cfg.get("use_batch_norm", True)),
                                  last=last, site=(3 << 16) + j))

    def _all_weights(self) -> List[_Weight]:
        out = [self.w_out] + [L["w"] for L in self.post]
        for L in self.enc_layers + self.dec_layers:
            out += [L[k] for k in ("wqkv", "wo", "w1", "w2", "wc_x", "wc_a") if k in L]
        for pr in (self.dur_pred, self.pitch_pred, self.energy_pred):
            out += [c["w"] for c in pr["convs"]]
        return out

    def _layer_param_names(self, L, concat_after):
        p = L["prefix"]
        names = [p + "self_atten.linear_%s.weight" % n for n in "qkv"]      # adjacent: one fused gradient GEMM
        names += [p + "self_atten.linear_%s.bias" % n for n in "qkv"]
        names += [p + "self_atten.linear_out.weight", p + "self_atten.linear_out.bias",
                  p + "feed_forward.w1.weight", p + "feed_forward.w1.bias",
                  p + "feed_forward.w2.weight", p + "feed_forward.w2.bias",
                  p + "norm1.weight", p + "norm1.bias", p + "norm2.weight", p + "norm2.bias"]
        if concat_after:
            names += [p + "concat_linear.weight", p + "concat_linear.bias"]
        return names

    def _pred_param_names(self, pr):
        names = [pr["prefix"] + "linear.weight", pr["prefix"] + "linear.bias"]
        for c in pr["convs"]:
            names += [c["p"] + "0.conv.weight", c["p"] + "0.conv.bias", c["p"] + "2.weight", c["p"] + "2.bias"]
        return names

    def _build_grad_buffer(self):
        """Flat fp32 gradient buffer, bucketed in backward-completion order."""
        cfg = self.cfg
        va = "variance_adaptor."
        buckets: List[List[str]] = [[]]
        for L in reversed(self.post):
            buckets[0].append(L["p"] + "0.weight")
            if L["bn"]:
                buckets[0] += [L["p"] + "1.weight", L["p"] + "1.bias"]
        buckets[0] += ["feat_out.weight", "feat_out.bias"]
        for L in reversed(self.dec_layers):
            buckets.append(self._layer_param_names(L, cfg["decoder_concat_after"]))
        adaptor = []
        if cfg["decoder_normalized_before"]:
            buckets[0] += ["decoder.after_norm.weight", "decoder.after_norm.bias"]
        if self.scaled:
            adaptor.append("decoder.embed.0.alpha")
        adaptor += [va + "pitch_embed.conv.conv.weight", va + "pitch_embed.conv.conv.bias",
                    va + "energy_embed.conv.conv.weight", va + "energy_embed.conv.conv.bias"]
        adaptor += self._pred_param_names(self.pitch_pred) + self._pred_param_names(self.energy_pred)
        adaptor += self._pred_param_names(self.dur_pred)
        if cfg["encoder_normalized_before"]:
            adaptor += ["encoder.after_norm.weight", "encoder.after_norm.bias"]
        buckets.append(adaptor)
        for L in reversed(self.enc_layers):
            buckets.append(self._layer_param_names(L, cfg["encoder_concat_after"]))
        last = ["encoder.embed.0.weight"]
        if self.scaled:
            last.append("encoder.embed.1.alpha")
        buckets.append(last)
        names = [n for b in buckets for n in b]
        assert sorted(names) == sorted(self.P), (sorted(set(self.P) - set(names)), sorted(set(names) - set(self.P)))
        off, self.G, self.bucket_slices = 0, {}, []
        offsets = {}
        for b in buckets:
            start = off
            for n in b:
                off = -(-off // 4) * 4               # 16-byte alignment of every gradient
                offsets[n] = off
                off += self.P[n].numel()
            self.bucket_slices.append((start, off))
        self.flat = torch.zeros((off,), device=self.dev)
        for n, o in offsets.items():
            self.G[n] = self.flat[o:o + self.P[n].numel()].view(self.P[n].shape)

    # ------------------------------------------------------------------ helpers
    def _pe_table(self, T):
        if self._pe is None or self._pe.shape[0] < T:
            n = 1 << (max(T, 1024) - 1).bit_length()
            self._pe = posenc_table(n, self.d).to(self.dev)
        return self._pe

    def _empty(self, shape, dtype=None):
        t = torch.empty(shape, device=self.dev, dtype=dtype or self.tdt)
        if _HOLD is not None:             # FS2_TRAIN_HOLD=1 (debugging aid): no block is handed out twice
            _HOLD.append(t)
        return t

    def _ws(self, nbytes, raw_stream=None):
        """Kernel scratch (split partials, per-workgroup sums).  One buffer per stream, grown on demand: the kernels
        that use it run in order on that stream, each done with it before the next one starts."""
        key = stream() if raw_stream is None else raw_stream
        n = (max(int(nbytes), 16) + 3) // 4
        buf = self._ws_cache.get(key)
        if buf is None or buf.numel() < n:
            buf = self._ws_cache[key] = torch.empty(n, device=self.dev, dtype=torch.float32)
        return buf

    def _ksplit(self, M, N, Cin, taps):
        return Engine._ksplit(M, N, Cin, taps, self.code)       # same split-K rule as the inference schedule

    def _rate(self, key):
        return float(self.rates.get(key, 0.0)) if self.drop_on else 0.0

    def _gemm(self, x, w, bias, M, N, Cin, taps, seg=None, res=None, relu=0, out_f32=False, ksplit=1, out=None,
              ldx=None, gate=None, gate_scale=1.0, drop=None, tag=None):
        """drop = (site, rate): dropout after the activation (training forward)"""
        if out is None:
            shape = (ksplit, M, N) if ksplit > 1 else (M, N)
            out = self._empty(shape, dtype=torch.float32 if out_f32 else self.tdt)
        ldx = Cin if ldx is None else ldx
        t0 = self._tic(tag)
        if gate is not None:
            call("fs2_conv_gemm_gated", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(gate), float(gate_scale), ptr(out),
                 ptr(seg), M, N, Cin, taps, ldx, N, N, N, int(out_f32), self.code, stream())
        elif drop is not None and drop[1] > 0.0:
            assert res is None and ksplit == 1
            call("fs2_conv_gemm_dropout", ptr(x), ptr(w), ptr(bias), ptr(out), ptr(seg), M, N, Cin, taps, ldx, N,
                 int(relu), int(out_f32), self.code, self.seed, drop[0], drop[1], stream())
        else:
            call("fs2_conv_gemm", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(out), ptr(seg), M, N, Cin, taps, ldx, N, N,
                 int(relu), int(out_f32), self.code, ksplit, M * N, stream())
        self._toc(tag, t0, 2.0 * M * N * Cin * taps)
        return out

    def _tic(self, tag):
        if self.profile is None or tag is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _toc(self, tag, e0, flops):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.profile.append((tag, flops, e0, e1))

    def _ln(self, x, w, b, M, C, want_f32, want_t, parts=None, sum_out=None, pdrop=None, odrop=None):
        """pdrop / odrop = (site, rate): dropout of the summed parts / of the LayerNorm output"""
        of = self._empty((M, C), torch.float32) if (want_f32 or (want_t and self.code == F32)) else None
        ot = self._empty((M, C)) if (want_t and self.code != F32) else None
        nparts = 0 if parts is None else (parts.shape[0] if parts.dim() == 3 else 1)
        pd = pdrop if (pdrop is not None and pdrop[1] > 0.0 and nparts > 0) else (0, 0.0)
        od = odrop if (odrop is not None and odrop[1] > 0.0) else (0, 0.0)
        if pd[1] > 0.0 or od[1] > 0.0:
            call("fs2_add_layernorm_dropout", ptr(x), ptr(parts), nparts, M * C, ptr(w), ptr(b), ptr(sum_out), ptr(of),
                 ptr(ot), M, C, LN_EPS, self.code, self.seed, pd[0], pd[1], od[0], od[1], stream())
        else:
            call("fs2_add_layernorm", ptr(x), ptr(parts), nparts, M * C, ptr(w), ptr(b), ptr(sum_out), ptr(of), ptr(ot),
                 M, C, LN_EPS, self.code, stream())
        return of, (of if self.code == F32 else ot)

    def _ln_bwd(self, x, dy, parts, w, dres, M, C, dw=None, db=None, relu_gate=False, want_f32=True, want_t=True,
                in_drop=None, t_drop=None):
        """returns (dx fp32 or None, dx as T or None).  in_drop = (site, rate) masks the incoming gradient,
        t_drop masks only the T copy (the gradient of a dropped-out sub-layer output)."""
        f32 = self.code == F32
        ind = in_drop if (in_drop is not None and in_drop[1] > 0.0) else (0, 0.0)
        td = t_drop if (t_drop is not None and t_drop[1] > 0.0 and want_t) else (0, 0.0)
        split_t = td[1] > 0.0                      # the T copy differs from the fp32 result
        of = self._empty((M, C), torch.float32) if (want_f32 or (want_t and f32) or split_t) else None
        ot = self._empty((M, C)) if (want_t and (not f32 or split_t)) else None
        nparts = 0 if parts is None else (parts.shape[0] if parts.dim() == 3 else 1)
        ws = self._ws(self._size("fs2_layernorm_bwd_workspace", M, C)) if w is not None else None
        call("fs2_layernorm_bwd", ptr(x), ptr(dy), ptr(parts), nparts, M * C, ptr(w), ptr(dres), int(relu_gate), ptr(of),
             ptr(ot), ptr(dw), ptr(db), ptr(ws), M, C, LN_EPS, 0, self.code, self.seed, ind[0], ind[1], td[0], td[1],
             stream())
        return of, (ot if ot is not None else of)

    def _wgrad_split(self, M, N, Cin, taps):
        bch = 64 if self.code == F32 else 128
        tw3 = taps % 3 == 0                      # 3 taps per 12-wave workgroup, one per CU; else 4 waves, two per CU
        tiles = -(-N // bch) * -(-Cin // bch) * (taps // 3 if tw3 else taps)
        nblocks = max(1, -(-M // 64))
        slots = 256 if tw3 else 512              # resident workgroups per round
        target = slots if tiles <= slots else 2 * slots
        ks = max(1, min(64, nblocks, target // tiles if tiles <= target else 1))
        return ks

    def _wgrad(self, dy, x, grad, seg, M, N, Cin, taps, ldy=None, ldx=None, tag="wgrad", dbias=None):
        """weight gradient (+ bias gradient = column sums of dy when dbias is given)"""
        ks = self._wgrad_split(M, N, Cin, taps)
        if self.wgrad_stream is None or self.profile is not None:
            ws = self._ws(self._size("fs2_conv_wgrad_workspace", N, Cin, taps, ks))
            t0 = self._tic(tag)
            call("fs2_conv_wgrad", ptr(dy), ptr(x), ptr(grad), ptr(dbias), ptr(ws), ptr(seg) if taps > 1 else None, M, N,
                 Cin, taps, N if ldy is None else ldy, Cin if ldx is None else ldx, self.code, ks, 1.0, 0, stream())
            self._toc(tag, t0, 2.0 * M * N * Cin * taps)
            return
        # weight gradients are leaves of the backward: they run on a side stream, beside the input-gradient
        # chain on the main stream (ordered after the kernel that produced dy; joined before the buckets are used)
        self._wq.append((dy, x, grad, seg if taps > 1 else None, M, N, Cin, taps, N if ldy is None else ldy,
                         Cin if ldx is None else ldx, dbias, ks))
        if len(self._wq) >= self.wgrad_batch or self._wgrad_sync:
            self._flush_wgrad()
        if self._wgrad_sync:
            self._join_wgrad()

    def _size(self, name, *args):
        """workspace-size query of the library, memoised (pure function of the shape)"""
        key = (name,) + args
        v = self._size_cache.get(key)
        if v is None:
            v = self._size_cache[key] = int(getattr(hip.load(), name)(*args))
        return v

    def _flush_wgrad(self):
        """queued weight-gradient launches -> side stream, ordered after everything enqueued on the main stream so far"""
        q = self._wq
        if not q:
            return
        self._wq = []
        side = self.wgrad_stream
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        raw = hip.c_void_p(self._side_raw)
        for dy, x, grad, seg, M, N, Cin, taps, ldy, ldx, dbias, ks in q:
            nbytes = self._size("fs2_conv_wgrad_workspace", N, Cin, taps, ks)
            buf = self._ws_cache.get(self._side_raw)
            if buf is None or buf.numel() * 4 < nbytes:
                with torch.cuda.stream(side):          # (the caching allocator ties the block to the side stream)
                    buf = self._ws(nbytes, self._side_raw)
            call("fs2_conv_wgrad", ptr(dy), ptr(x), ptr(grad), ptr(dbias), ptr(buf), ptr(seg), M, N, Cin, taps, ldy, ldx,
                 self.code, ks, 1.0, 0, raw)
            dy.record_stream(side)
            x.record_stream(side)
        self._wgrad_pending = True

    def producer_streams(self):
        """streams whose enqueued work writes the gradient buffer (a bucket consumer must order itself after both)"""
        out = [torch.cuda.current_stream()]
        if self.wgrad_stream is not None:
            out.append(self.wgrad_stream)
        return out

    def _join_wgrad(self):
        if self.wgrad_stream is not None:
            self._flush_wgrad()
        if self.wgrad_stream is not None and self._wgrad_pending:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._wgrad_pending = False

    def _colsum(self, G, out, M, C, ldg=None, g_is_t=True, S=None, R=1, scale=1.0, pe=None, pos=None):
        ws = self._ws(self._size("fs2_colreduce_workspace", M, C, R))
        gd = 1 if (g_is_t and self.code == BF16) else 0
        call("fs2_colreduce", ptr(G), gd, C if ldg is None else ldg, ptr(S), 0 if S is None else S.shape[1], R, ptr(pe),
             ptr(pos), ptr(out), ptr(ws), M, C, float(scale), 0, stream())

    def _pack_layers(self, layers, concat_after):
        for L in layers:
            p = L["prefix"]
            torch.cat([self.P[p + "self_atten.linear_%s.bias" % n].data for n in "qkv"], 0, out=L["bqkv"])
            if concat_after:
                wc = self.P[p + "concat_linear.weight"].data
                L["wc_x"].params[0].data.copy_(wc[:, :self.d])
                L["wc_a"].params[0].data.copy_(wc[:, self.d:])

    def refresh_weights(self):
        """Re-pack every GEMM operand from the current parameter values (once per step).  Two groups on a side
        stream, in the order the forward needs them: phone level (encoder stack, duration predictor) and frame
        level (pitch / energy predictors, decoder stack, feat_out) — the second group is packed underneath the
        phone-level compute.  Returns the two events the forward waits for."""
        main = torch.cuda.current_stream()
        side = self.wgrad_stream if self.wgrad_stream is not None else main
        if side is not main:
            side.wait_stream(main)                 # the optimiser's update of the parameters
        if self._pack_groups is None:
            lw = lambda layers: [L[k] for L in layers for k in ("wqkv", "wo", "w1", "w2", "wc_x", "wc_a") if k in L]
            self._pack_groups = (
                _PackGroup(lw(self.enc_layers) + [c["w"] for c in self.dur_pred["convs"]], self.dev),
                _PackGroup([c["w"] for pr in (self.pitch_pred, self.energy_pred) for c in pr["convs"]] +
                           lw(self.dec_layers) + [self.w_out] + [L["w"] for L in self.post], self.dev))
        with torch.cuda.stream(side):
            self._pack_layers(self.enc_layers, self.cfg["encoder_concat_after"])      # bias concatenation, concat_after split
            self._pack_groups[0].launch(self.code)
            ev_phone = torch.cuda.Event()
            ev_phone.record()
            self._pack_layers(self.dec_layers, self.cfg["decoder_concat_after"])
            self._pack_groups[1].launch(self.code)
            ev_frame = torch.cuda.Event()
            ev_frame.record()
        return ev_phone, ev_frame

    # ------------------------------------------------------------------ forward pieces
    @staticmethod
    def _sites(side, li):
        """dropout site ids of one FFT block: attention probabilities (+ head), attention residual, FFN inner, FFN residual"""
        base = (0 if side == "enc" else 1 << 16) + 256 * (li + 1)
        return dict(att=base, res_a=base + 64, inner=base + 65, res_f=base + 66)

    def _stack_fwd(self, layers, prefix, pre_ln, concat_after, x32, xt, seg, cu, B, max_len, M):
        d, P = self.d, self.P
        side = "enc" if prefix.startswith("enc") else "dec"
        r_res, r_att = self._rate(side), self._rate(side + "_att")
        recs, pend, pend_drop = [], None, None
        for li, L in enumerate(layers):
            p = L["prefix"]
            st = self._sites(side, li)
            r = {}
            if pre_ln:
                xs = x32 if pend is None else self._empty((M, d), torch.float32)
                a_t = self._ln(x32, P[p + "norm1.weight"], P[p + "norm1.bias"], M, d, False, True, parts=pend,
                               sum_out=xs if pend is not None else None, pdrop=pend_drop)[1]
                x32 = xs
                r["x_ln1"] = x32
            else:
                a_t = xt
            r["a_t"] = a_t
            qkv = self._gemm(a_t, L["wqkv"].fwd, L["bqkv"], M, 3 * d, d, 1)
            ctx = self._empty((M, d))
            lse = self._empty((self.h, M), torch.float32)
            call("fs2_attention_fwd", ptr(qkv), ptr(ctx), ptr(lse), ptr(cu), B, max_len, M, self.h, self.dk, 3 * d, d,
                 self.code, self.seed, st["att"], r_att, stream())
            r.update(qkv=qkv, ctx=ctx, lse=lse)
            if concat_after:
                att = self._gemm(ctx, L["wo"].fwd, P[p + "self_atten.linear_out.bias"], M, d, d, 1)
                pend = self._empty((2, M, d), torch.float32)
                self._gemm(a_t, L["wc_x"].fwd, P[p + "concat_linear.bias"], M, d, d, 1, out_f32=True, out=pend[0])
                self._gemm(att, L["wc_a"].fwd, None, M, d, d, 1, out_f32=True, out=pend[1])
                r["att"] = att
            else:
                pend = self._gemm(ctx, L["wo"].fwd, P[p + "self_atten.linear_out.bias"], M, d, d, 1, out_f32=True,
                                  ksplit=self._ksplit(M, d, d, 1))
            pend_drop = (st["res_a"], r_res)
            if pre_ln:
                x1 = self._empty((M, d), torch.float32)
                f_t = self._ln(x32, P[p + "norm2.weight"], P[p + "norm2.bias"], M, d, False, True, parts=pend,
                               sum_out=x1, pdrop=pend_drop)[1]
                x32 = x1
                r["x_ln2"] = x1
            else:
                s1 = self._empty((M, d), torch.float32)
                x32, xt = self._ln(x32, P[p + "norm1.weight"], P[p + "norm1.bias"], M, d, True, True, parts=pend,
                                   sum_out=s1, pdrop=pend_drop)
                r["s1"] = s1
                f_t = xt
            r["f_t"] = f_t
            w1, w2 = L["w1"], L["w2"]
            sg = seg if w1.k > 1 else None
            hdn = self._gemm(f_t, w1.fwd, P[p + "feed_forward.w1.bias"], M, w1.N, d, w1.k, seg=sg, relu=1, tag="ffn1",
                             drop=(st["inner"], r_res))
            pend = self._gemm(hdn, w2.fwd, P[p + "feed_forward.w2.bias"], M, d, w1.N, w2.k, seg=sg, out_f32=True,
                              ksplit=self._ksplit(M, d, w1.N, w2.k), tag="ffn2")
            pend_drop = (st["res_f"], r_res)
            r["hdn"] = hdn
            if not pre_ln:
                s2 = self._empty((M, d), torch.float32)
                x32, xt = self._ln(x32, P[p + "norm2.weight"], P[p + "norm2.bias"], M, d, True, True, parts=pend,
                                   sum_out=s2, pdrop=pend_drop)
                r["s2"] = s2
                pend = None
            recs.append(r)
        after = None
        if pre_ln:
            after = self._empty((M, d), torch.float32)
            x32, xt = self._ln(x32, P[prefix + "after_norm.weight"], P[prefix + "after_norm.bias"], M, d, True, True,
                               parts=pend, sum_out=after, pdrop=pend_drop)
        return x32, xt, dict(layers=recs, after=after)

    def _stack_bwd(self, layers, prefix, pre_ln, concat_after, saved, dy, dparts, seg, cu, B, max_len, M, first_ready):
        """dy (+ dparts): gradient wrt the stack output.  Returns the gradient wrt the stack input as (dy, parts).
        first_ready: gradient bucket that is complete once the stack's after_norm has been processed."""
        d, P, G = self.d, self.P, self.G
        side = "enc" if prefix.startswith("enc") else "dec"
        r_res, r_att = self._rate(side), self._rate(side + "_att")
        inner_scale = 1.0 / (1.0 - r_res) if r_res > 0.0 else 1.0
        n = len(layers)
        if pre_ln:
            dy, dyt = self._ln_bwd(saved["after"], dy, dparts, P[prefix + "after_norm.weight"], None, M, d,
                                   dw=G[prefix + "after_norm.weight"], db=G[prefix + "after_norm.bias"],
                                   t_drop=(self._sites(side, n - 1)["res_f"], r_res))
            dparts = None
        if self.grad_ready is not None:
            self._flush_wgrad()
            self.grad_ready(first_ready)
        for li in range(n - 1, -1, -1):
            L, r = layers[li], saved["layers"][li]
            st = self._sites(side, li)
            p = L["prefix"]
            w1, w2 = L["w1"], L["w2"]
            U = w1.N
            sg = seg if w1.k > 1 else None
            if not pre_ln:       # x2 = LN2(s2)
                dy, dyt = self._ln_bwd(r["s2"], dy, dparts, P[p + "norm2.weight"], None, M, d,
                                       dw=G[p + "norm2.weight"], db=G[p + "norm2.bias"], t_drop=(st["res_f"], r_res))
                dparts = None
            # ---- feed-forward: s2 = x1 + w2(relu(w1(f)))       (dy = d s2, dyt its T copy)
            self._wgrad(dyt, r["hdn"], G[p + "feed_forward.w2.weight"], sg, M, d, U, w2.k, tag="wgrad_ffn2",
                        dbias=G[p + "feed_forward.w2.bias"])
            dh = self._gemm(dyt, w2.bwd, None, M, U, d, w2.k, seg=sg, gate=r["hdn"], gate_scale=inner_scale,
                            tag="dgrad_ffn2")
            self._wgrad(dh, r["f_t"], G[p + "feed_forward.w1.weight"], sg, M, U, d, w1.k, tag="wgrad_ffn1",
                        dbias=G[p + "feed_forward.w1.bias"])
            df = self._gemm(dh, w1.bwd, None, M, d, U, w1.k, seg=sg, out_f32=True, ksplit=self._ksplit(M, d, U, w1.k),
                            tag="dgrad_ffn1")
            if pre_ln:           # f = LN2(x1): d x1 = dy + LN2'(df)
                dy, dyt = self._ln_bwd(r["x_ln2"], None, df, P[p + "norm2.weight"], dy, M, d,
                                       dw=G[p + "norm2.weight"], db=G[p + "norm2.bias"], t_drop=(st["res_a"], r_res))
            else:                # f = x1 = LN1(s1): d s1 = LN1'(dy + df)
                dy, dyt = self._ln_bwd(r["s1"], dy, df, P[p + "norm1.weight"], None, M, d,
                                       dw=G[p + "norm1.weight"], db=G[p + "norm1.bias"], t_drop=(st["res_a"], r_res))
            # ---- attention: s1 = x + out(ctx)  [or x + concat_linear(cat(a, out(ctx)))]
            da_extra = None
            if concat_after:
                gc = G[p + "concat_linear.weight"]
                tmp_x, tmp_a = self._empty((d, d), torch.float32), self._empty((d, d), torch.float32)
                self._wgrad(dyt, r["a_t"], tmp_x, None, M, d, d, 1, dbias=G[p + "concat_linear.bias"])
                self._wgrad(dyt, r["att"], tmp_a, None, M, d, d, 1)
                self._join_wgrad()                      # the two halves are copied on the main stream
                gc[:, :d].copy_(tmp_x)
                gc[:, d:].copy_(tmp_a)
                da_extra = self._gemm(dyt, L["wc_x"].bwd, None, M, d, d, 1, out_f32=True)
                datt = self._gemm(dyt, L["wc_a"].bwd, None, M, d, d, 1)
            else:
                datt = dyt
            self._wgrad(datt, r["ctx"], G[p + "self_atten.linear_out.weight"], None, M, d, d, 1,
                        dbias=G[p + "self_atten.linear_out.bias"])
            dctx = self._gemm(datt, L["wo"].bwd, None, M, d, d, 1)
            dqkv = self._empty((M, 3 * d))
            delta = self._empty((self.h, M), torch.float32)
            call("fs2_attention_bwd", ptr(r["qkv"]), ptr(r["ctx"]), ptr(dctx), ptr(r["lse"]), ptr(delta), ptr(dqkv),
                 ptr(cu), B, max_len, M, self.h, self.dk, 3 * d, d, self.code, self.seed, st["att"], r_att, stream())
            gq = G[p + "self_atten.linear_q.weight"]
            gqkv = self.flat[gq.storage_offset():gq.storage_offset() + 3 * d * d].view(3 * d, d)
            gb = G[p + "self_atten.linear_q.bias"]
            self._wgrad(dqkv, r["a_t"], gqkv, None, M, 3 * d, d, 1,
                        dbias=self.flat[gb.storage_offset():gb.storage_offset() + 3 * d])
            da = self._gemm(dqkv, L["wqkv"].bwd, None, M, d, 3 * d, 1, out_f32=True, ksplit=self._ksplit(M, d, 3 * d, 1))
            if da_extra is not None:
                da = torch.stack([da, da_extra], 0) if da.dim() == 2 else torch.cat([da, da_extra.unsqueeze(0)], 0)
            if pre_ln:           # a = LN1(x): d x = dy + LN1'(da)
                prev = (self._sites(side, li - 1)["res_f"], r_res) if li > 0 else None
                dy, dyt = self._ln_bwd(r["x_ln1"], None, da, P[p + "norm1.weight"], dy, M, d,
                                       dw=G[p + "norm1.weight"], db=G[p + "norm1.bias"], t_drop=prev)
                dparts = None
            else:                # a = x: d x = dy + da, folded into the next LayerNorm backward
                dparts = da
            if self.grad_ready is not None:
                self._flush_wgrad()
                self.grad_ready(self._bucket_of[p])
        return dy, dparts

    def _pred_fwd(self, pr, x_t, seg, M, cin):
        P = self.P
        rate, site0 = self._rate(pr["rate_key"]), pr["site"]
        cur, recs = x_t, []
        n = len(pr["convs"])
        y32 = None
        for j, c in enumerate(pr["convs"]):
            w = c["w"]
            y = self._gemm(cur, w.fwd, P[c["p"] + "0.conv.bias"], M, w.N, cin, w.k, seg=seg if w.k > 1 else None, relu=1,
                           out_f32=True)
            recs.append(dict(inp=cur, y=y))
            last = j == n - 1
            y32, cur = self._ln(y, P[c["p"] + "2.weight"], P[c["p"] + "2.bias"], M, w.N, last, not last,
                                odrop=(site0 + j, rate))
            cin = w.N
        out = self._empty((M, pr["nout"]), torch.float32)
        call("fs2_rowdot", ptr(y32), ptr(P[pr["prefix"] + "linear.weight"]), ptr(P[pr["prefix"] + "linear.bias"]),
             ptr(out), M, cin, pr["nout"], stream())
        return out, dict(convs=recs, y32=y32)

    def _pred_bwd(self, pr, saved, dout, seg, M, cin0, acc):
        """dout fp32 [M, nout]; adds the gradient wrt the predictor input into `acc` (fp32 [M, cin0])."""
        P, G = self.P, self.G
        pre = pr["prefix"]
        nout = pr["nout"]
        rate, site0 = self._rate(pr["rate_key"]), pr["site"]
        C = pr["convs"][-1]["w"].N
        self._colsum(saved["y32"], G[pre + "linear.weight"], M, C, g_is_t=False, S=dout, R=nout)
        self._colsum(dout, G[pre + "linear.bias"], M, nout, g_is_t=False)
        dcur = self._empty((M, C), torch.float32)
        call("fs2_rowdot_bwd", ptr(dout), ptr(P[pre + "linear.weight"]), ptr(dcur), M, C, nout, stream())
        dparts = None
        for j in range(len(pr["convs"]) - 1, -1, -1):
            c, r = pr["convs"][j], saved["convs"][j]
            w = c["w"]
            cin = cin0 if j == 0 else pr["convs"][j - 1]["w"].N
            sg = seg if w.k > 1 else None
            _, dpre = self._ln_bwd(r["y"], dcur, dparts, P[c["p"] + "2.weight"], None, M, w.N, dw=G[c["p"] + "2.weight"],
                                   db=G[c["p"] + "2.bias"], relu_gate=True, want_f32=False, in_drop=(site0 + j, rate))
            self._wgrad(dpre, r["inp"], G[c["p"] + "0.conv.weight"], sg, M, w.N, cin, w.k, dbias=G[c["p"] + "0.conv.bias"])
            if j == 0:
                self._gemm(dpre, w.bwd, None, M, cin, w.N, w.k, seg=sg, res=acc, out_f32=True, out=acc)
            else:
                dcur = None
                dparts = self._gemm(dpre, w.bwd, None, M, cin, w.N, w.k, seg=sg, out_f32=True,
                                    ksplit=self._ksplit(M, cin, w.N, w.k))

    # ------------------------------------------------------------------ PostNet (training-mode BatchNorm)
    BN_EPS, BN_MOMENTUM = 1e-5, 0.1          # nn.BatchNorm1d defaults

    def _postnet_fwd(self, before, seg, M):
        """after = before + postnet(before) (network/fastspeech2.py:222-225); BatchNorm takes its statistics over
        the packed frames of the batch and updates running_mean / running_var / num_batches_tracked as
        nn.BatchNorm1d does in train() mode."""
        P, lib = self.P, hip.load()
        rate = self._rate("postnet")
        cur = self._empty((M, self.odim))
        call("fs2_cast", ptr(before), ptr(cur), M * self.odim, self.code, stream())
        recs, after = [], None
        for L in self.post:
            w = L["w"]
            y = self._gemm(cur, w.fwd, None, M, L["cout"], L["cin"], L["k"], seg=seg if L["k"] > 1 else None, out_f32=True,
                           tag="postnet")
            stats = None
            if L["bn"]:
                stats = self._empty((3, L["cout"]), torch.float32)
                ws = self._ws(lib.fs2_bn_workspace(M, L["cout"]))
                call("fs2_bn_stats", ptr(y), ptr(stats), ptr(ws), M, L["cout"], self.BN_EPS, stream())
                rm, rv = self.buffers[L["p"] + "1.running_mean"], self.buffers[L["p"] + "1.running_var"]
                rm.mul_(1.0 - self.BN_MOMENTUM).add_(stats[0], alpha=self.BN_MOMENTUM)
                rv.mul_(1.0 - self.BN_MOMENTUM).add_(stats[2], alpha=self.BN_MOMENTUM * M / max(M - 1, 1))
                self.buffers[L["p"] + "1.num_batches_tracked"].add_(1)
            gam = P[L["p"] + "1.weight"] if L["bn"] else None
            bet = P[L["p"] + "1.bias"] if L["bn"] else None
            recs.append(dict(inp=cur, y=y, stats=stats))
            if L["last"]:
                after = self._empty((M, self.odim), torch.float32)
                call("fs2_bn_apply", ptr(y), ptr(stats), ptr(gam), ptr(bet), 0, ptr(before), self.odim, ptr(after),
                     self.odim, None, 0, M, L["cout"], self.code, self.seed, L["site"], rate, stream())
            else:
                nxt = self._empty((M, L["cout"]))
                call("fs2_bn_apply", ptr(y), ptr(stats), ptr(gam), ptr(bet), 1, None, 0, None, 0, ptr(nxt), L["cout"], M,
                     L["cout"], self.code, self.seed, L["site"], rate, stream())
                cur = nxt
        return after, recs

    def _postnet_bwd(self, recs, dafter, seg, M):
        """dafter: gradient of `after` (fp32).  Fills the PostNet gradients; returns the gradient wrt the PostNet input
        (fp32 [M, odim]; the residual path `after = before + ...` is added by the caller)."""
        P, G, lib = self.P, self.G, hip.load()
        rate = self._rate("postnet")
        dout = dafter
        for L, r in zip(reversed(self.post), reversed(recs)):
            w = L["w"]
            C = L["cout"]
            ldt = self.odim_pad if L["last"] else C
            dy = torch.zeros((M, ldt), device=self.dev, dtype=self.tdt) if ldt != C else self._empty((M, C))
            gam = P[L["p"] + "1.weight"] if L["bn"] else None
            bet = P[L["p"] + "1.bias"] if L["bn"] else None
            ws = self._ws(lib.fs2_bn_workspace(M, C)) if L["bn"] else None
            call("fs2_bn_bwd", ptr(r["y"]), ptr(r["stats"]), ptr(gam), ptr(bet), 0 if L["last"] else 1, ptr(dout), C,
                 ptr(G[L["p"] + "1.weight"]) if L["bn"] else None, ptr(G[L["p"] + "1.bias"]) if L["bn"] else None, ptr(dy),
                 ldt, ptr(ws), M, C, self.code, self.seed, L["site"], rate, stream())
            sg = seg if L["k"] > 1 else None
            self._wgrad(dy, r["inp"], G[L["p"] + "0.weight"], sg, M, C, L["cin"], L["k"], ldy=ldt, tag="wgrad_postnet")
            dout = self._gemm(dy, w.bwd, None, M, L["cin"], ldt, L["k"], seg=sg, out_f32=True, tag="dgrad_postnet")
        return dout

    # ------------------------------------------------------------------ one step
    def forward(self, ids, ilens, olens, ds, ps, es, ys):
        """ids int64 [Mp], ds int32 [Mp], ps fp32 [Mf,4], es fp32 [Mf,1], ys fp32 [Mf,odim] — packed, on the device.
        Returns (loss terms as device scalars, saved state for `backward`)."""
        dev, d, P, cfg = self.dev, self.d, self.P, self.cfg
        self.drop_on = bool(getattr(self.model, "dropout_enabled", True))
        self.step += 1
        self.seed = (self.base_seed * 0x9E3779B1 + self.step * 0x85EBCA77 + self.rank_seed * 0xC2B2AE3D) & 0xFFFFFFFF
        ev_phone, ev_frame = self.refresh_weights()
        torch.cuda.current_stream().wait_event(ev_phone)
        B = len(ilens)
        ilens = [int(v) for v in ilens]
        Mp = int(sum(ilens))
        cu_p = np.zeros(B + 1, np.int32)
        cu_p[1:] = np.cumsum(ilens)
        cu_phone = torch.from_numpy(cu_p).to(dev, non_blocking=True)
        max_il = max(ilens)
        seg_p = self._empty((Mp, 2), torch.int32)
        pos_p = self._empty((Mp,), torch.int32)
        call("fs2_build_seg", ptr(cu_phone), B, ptr(seg_p), ptr(pos_p), Mp, stream())
        pe = self._pe_table(max_il)
        xscale = 1.0 if self.scaled else math.sqrt(d)
        enc_pre, dec_pre = cfg["encoder_normalized_before"], cfg["decoder_normalized_before"]
        x32 = self._empty((Mp, d), torch.float32)
        xt = self._empty((Mp, d)) if (not enc_pre and self.code != F32) else None
        if self.scaled:        # alpha is a learnable parameter: read on the device, no host sync in the step
            call("fs2_embed_posenc_dev", ptr(ids), ptr(pos_p), ptr(P["encoder.embed.0.weight"]), ptr(pe), xscale,
                 ptr(P["encoder.embed.1.alpha"]), ptr(x32), ptr(xt), Mp, d, self.idim, self.code, stream())
        else:
            call("fs2_embed_posenc", ptr(ids), ptr(pos_p), ptr(P["encoder.embed.0.weight"]), ptr(pe), xscale, 1.0,
                 ptr(x32), ptr(xt), Mp, d, self.idim, self.code, stream())
        if self._rate("enc_pos") > 0.0:
            call("fs2_dropout", ptr(x32), ptr(xt), Mp, d, self.seed, self.SITE_ENC_POS, self._rate("enc_pos"), self.code,
                 stream())
        if self.code == F32:
            xt = x32
        hs32, hst, enc_saved = self._stack_fwd(self.enc_layers, "encoder.", enc_pre, cfg["encoder_concat_after"], x32, xt,
                                               seg_p, cu_phone, B, max_il, Mp)
        logd, dur_saved = self._pred_fwd(self.dur_pred, hst, seg_p, Mp, d)
        torch.cuda.current_stream().wait_event(ev_frame)          # frame-level operands are packed
        # LengthRegulator with the ground-truth durations (network/variance_adaptor.py:187)
        dur_used = self._empty((Mp,), torch.int32)
        cum = self._empty((Mp,), torch.int32)
        olens_dev = self._empty((B,), torch.int32)
        cu_frames = self._empty((B + 1,), torch.int32)
        call("fs2_lr_plan", ptr(ds), ptr(cu_phone), 1.0, ptr(dur_used), ptr(cum), ptr(olens_dev), ptr(cu_frames), B,
             stream())
        Mf = int(ys.shape[0])      # the caller guarantees sum(ds) == olens per utterance (no device read needed)
        olens = [int(v) for v in olens]
        assert sum(olens) == Mf
        max_ol = max(olens)
        hs_lr32 = self._empty((Mf, d), torch.float32)
        hs_lrt = self._empty((Mf, d)) if self.code != F32 else None
        seg_f = self._empty((Mf, 2), torch.int32)
        pos_f = self._empty((Mf,), torch.int32)
        lr_index = self._empty((Mf,), torch.int32)
        call("fs2_lr_gather", ptr(hs32), ptr(cum), ptr(cu_phone), ptr(cu_frames), B, ptr(hs_lr32), ptr(hs_lrt),
             ptr(seg_f), ptr(pos_f), ptr(lr_index), Mf, d, self.code, stream())
        if self.code == F32:
            hs_lrt = hs_lr32
        pitch, pitch_saved = self._pred_fwd(self.pitch_pred, hs_lrt, seg_f, Mf, d)
        energy, energy_saved = self._pred_fwd(self.energy_pred, hs_lrt, seg_f, Mf, d)
        va = "variance_adaptor."
        pe_f = self._pe_table(max_ol)
        y32 = self._empty((Mf, d), torch.float32)
        yt = self._empty((Mf, d)) if (not dec_pre and self.code != F32) else None
        r_pe, r_ee = self._rate("pitch_embed"), self._rate("energy_embed")
        hs_in = hs_lr32
        w_pe, b_pe = P[va + "pitch_embed.conv.conv.weight"], P[va + "pitch_embed.conv.conv.bias"]
        w_ee, b_ee = P[va + "energy_embed.conv.conv.weight"], P[va + "energy_embed.conv.conv.bias"]
        conv_emb = self.embed_taps != (1, 1)
        emb_x4 = None
        if conv_emb:
            # embedding convs with kernel size > 1: each embedding = fp32 fs2_conv_gemm ("same" zero padding inside each
            # utterance) over its input padded to 4 channels (16-byte rows); dropped out with their own sites, added to hs_lr
            emb = self._empty((2, Mf, d), torch.float32)
            emb_x4 = []
            for j, (vals, w, b, site, rate) in enumerate(((ps, w_pe, b_pe, self.SITE_PITCH_EMB, r_pe),
                                                          (es, w_ee, b_ee, self.SITE_ENERGY_EMB, r_ee))):
                taps, c = int(w.shape[-1]), int(w.shape[1])
                c4 = (c + 3) // 4 * 4
                x4 = torch.zeros((Mf, c4), device=self.dev, dtype=torch.float32)
                x4[:, :c].copy_(vals)
                wk = torch.zeros((taps, d, c4), device=self.dev, dtype=torch.float32)
                wk[:, :, :c].copy_(w.detach().permute(2, 0, 1))
                call("fs2_conv_gemm", ptr(x4), ptr(wk), ptr(b), None, ptr(emb[j]), ptr(seg_f) if taps > 1 else None, Mf, d, c4,
                     taps, c4, d, d, 0, 1, F32, 1, 0, stream())
                if rate > 0.0:
                    call("fs2_dropout", ptr(emb[j]), None, Mf, d, self.seed, site, rate, F32, stream())
                emb_x4.append(x4)
            hs_in = self._empty((Mf, d), torch.float32)
            call("fs2_add_layernorm", ptr(hs_lr32), ptr(emb), 2, Mf * d, None, None, ptr(hs_in), None, None, Mf, d, LN_EPS,
                 F32, stream())
            w_pe, b_pe, w_ee, b_ee = (torch.zeros((d, t.shape[1]), device=self.dev) if t.dim() == 3 else torch.zeros_like(t)
                                      for t in (w_pe, b_pe, w_ee, b_ee))
        elif r_pe > 0.0 or r_ee > 0.0:
            # nn.Dropout inside VarianceEmbedding (network/variance_adaptor.py:35-51; reference default 0): the two
            # embeddings are formed apart ([Mf, 4] x [4, d] and [Mf, 1] x [1, d]), dropped out and added to hs_lr; the
            # fused kernel then runs with zero embedding weights
            # (each embedding alone = the fused kernel over a zero residual with the other embedding's weights zeroed and
            # the positional term switched off; their sum with hs_lr = fs2_add_layernorm without a norm)
            zw_p, zb_p, zw_e, zb_e = (torch.zeros_like(t) for t in (w_pe, b_pe, w_ee, b_ee))
            zero_hs = torch.zeros((Mf, d), device=self.dev, dtype=torch.float32)
            emb = self._empty((2, Mf, d), torch.float32)
            ep, ee = emb[0], emb[1]
            call("fs2_variance_add_posenc", ptr(zero_hs), ptr(ps), ptr(es), ptr(w_pe), ptr(b_pe), ptr(zw_e), ptr(zb_e), 1.0, 1.0,
                 ptr(pe_f), ptr(pos_f), 1.0, 0.0, None, ptr(ep), None, Mf, d, self.pdim, self.edim, F32, stream())
            call("fs2_variance_add_posenc", ptr(zero_hs), ptr(ps), ptr(es), ptr(zw_p), ptr(zb_p), ptr(w_ee), ptr(b_ee), 1.0, 1.0,
                 ptr(pe_f), ptr(pos_f), 1.0, 0.0, None, ptr(ee), None, Mf, d, self.pdim, self.edim, F32, stream())
            if r_pe > 0.0:
                call("fs2_dropout", ptr(ep), None, Mf, d, self.seed, self.SITE_PITCH_EMB, r_pe, F32, stream())
            if r_ee > 0.0:
                call("fs2_dropout", ptr(ee), None, Mf, d, self.seed, self.SITE_ENERGY_EMB, r_ee, F32, stream())
            hs_in = self._empty((Mf, d), torch.float32)
            call("fs2_add_layernorm", ptr(hs_lr32), ptr(emb), 2, Mf * d, None, None, ptr(hs_in), None, None, Mf, d, LN_EPS,
                 F32, stream())
            w_pe, b_pe, w_ee, b_ee = (torch.zeros_like(t) for t in (w_pe, b_pe, w_ee, b_ee))
        vargs = (ptr(hs_in), ptr(ps), ptr(es), ptr(w_pe), ptr(b_pe), ptr(w_ee), ptr(b_ee), 1.0, 1.0, ptr(pe_f),
                 ptr(pos_f), xscale)
        vtail = (None, ptr(y32), ptr(yt), Mf, d, self.pdim, self.edim, self.code, stream())
        if self.scaled:
            call("fs2_variance_add_posenc_dev", *vargs, ptr(P["decoder.embed.0.alpha"]), *vtail)
        else:
            call("fs2_variance_add_posenc", *vargs, 1.0, *vtail)
        if self._rate("dec_pos") > 0.0:
            call("fs2_dropout", ptr(y32), ptr(yt), Mf, d, self.seed, self.SITE_DEC_POS, self._rate("dec_pos"), self.code,
                 stream())
        if self.code == F32:
            yt = y32
        z32, zt, dec_saved = self._stack_fwd(self.dec_layers, "decoder.", dec_pre, cfg["decoder_concat_after"], y32, yt,
                                             seg_f, cu_frames, B, max_ol, Mf)
        before = self._gemm(zt, self.w_out.fwd, P["feat_out.bias"], Mf, self.odim, d, 1, out_f32=True)
        after, post_saved = self._postnet_fwd(before, seg_f, Mf) if self.post else (None, None)
        saved = dict(B=B, Mp=Mp, Mf=Mf, max_il=max_il, max_ol=max_ol, ids=ids, ds=ds, ps=ps, es=es, ys=ys, after=after,
                     post=post_saved,
                     cu_phone=cu_phone, cu_frames=cu_frames, seg_p=seg_p, seg_f=seg_f, pos_p=pos_p, pos_f=pos_f,
                     dur_used=dur_used, cum=cum, enc=enc_saved, dec=dec_saved, dur=dur_saved, pitch=pitch_saved,
                     energy=energy_saved, hst=hst, hs_lrt=hs_lrt, zt=zt, emb_x4=emb_x4, before=before, logd=logd, pitch_out=pitch,
                     energy_out=energy, xscale=xscale, seed=self.seed, drop_on=self.drop_on, keep=(olens_dev, lr_index, hs_lr32, x32, y32, z32))
        outs = dict(before=before, log_durations=logd.view(-1), pitch=pitch, energy=energy)
        if after is not None:
            outs["after"] = after
        return outs, saved

    def backward(self, s, upstream: Optional[torch.Tensor] = None):
        """Fills self.G (all parameter gradients, fp32) for loss = L1(before, ys) + MSE(logd, log(ds+1)) +
        MSE(pitch, ps) + MSE(energy, es), each a mean over the packed elements, times `upstream` (device scalar)."""
        d, P, G, cfg = self.d, self.P, self.G, self.cfg
        B, Mp, Mf = s["B"], s["Mp"], s["Mf"]
        self.seed, self.drop_on = s["seed"], s["drop_on"]       # the masks of THIS step's forward
        va = "variance_adaptor."
        up = ptr(upstream) if upstream is not None else None
        self._bucket_of = {}
        bi = 1
        for L in reversed(self.dec_layers):
            self._bucket_of[L["prefix"]] = bi
            bi += 1
        adaptor_bucket = bi
        bi += 1
        for L in reversed(self.enc_layers):
            self._bucket_of[L["prefix"]] = bi
            bi += 1
        # ---- mel loss -> feat_out
        dbt = self._empty((Mf, self.odim_pad))
        if self.post:       # loss = L1(after) + L1(before), after = before + postnet(before)
            dbefore = self._empty((Mf, self.odim), torch.float32)
            call("fs2_loss_grad", ptr(s["before"]), ptr(s["ys"]), None, 0, 1.0 / (Mf * self.odim), up, ptr(dbefore), None,
                 Mf, self.odim, self.odim, self.code, stream())
            dafter = self._empty((Mf, self.odim), torch.float32)
            call("fs2_loss_grad", ptr(s["after"]), ptr(s["ys"]), None, 0, 1.0 / (Mf * self.odim), up, ptr(dafter), None,
                 Mf, self.odim, self.odim, self.code, stream())
            dx0 = self._postnet_bwd(s["post"], dafter, s["seg_f"], Mf)
            dbefore.add_(dafter).add_(dx0)
            if self.odim_pad != self.odim:
                dbt.zero_()
            dbt[:, :self.odim].copy_(dbefore)
        else:
            call("fs2_loss_grad", ptr(s["before"]), ptr(s["ys"]), None, 0, 1.0 / (Mf * self.odim), up, None, ptr(dbt), Mf,
                 self.odim, self.odim_pad, self.code, stream())
        self._wgrad(dbt, s["zt"], G["feat_out.weight"], None, Mf, self.odim, d, 1, ldy=self.odim_pad,
                    dbias=G["feat_out.bias"])
        dz = self._gemm(dbt, self.w_out.bwd, None, Mf, d, self.odim_pad, 1, out_f32=True)
        # ---- decoder stack
        dy, dparts = self._stack_bwd(self.dec_layers, "decoder.", cfg["decoder_normalized_before"],
                                     cfg["decoder_concat_after"], s["dec"], dz, None, s["seg_f"], s["cu_frames"], B,
                                     s["max_ol"], Mf, 0)
        if dparts is not None or self._rate("dec_pos") > 0.0:      # materialise the sum, undo the positional dropout
            dy, _ = self._ln_bwd(None, dy, dparts, None, None, Mf, d, want_t=False,
                                 in_drop=(self.SITE_DEC_POS, self._rate("dec_pos")))
        # ---- decoder positional encoding + variance embeddings (y = xscale * (hs_lr + emb) + alpha pe)
        if self.scaled:
            tmp = self._empty((d,), torch.float32)
            self._colsum(dy, tmp, Mf, d, g_is_t=False, pe=self._pe_table(s["max_ol"]), pos=s["pos_f"])
            torch.sum(tmp, 0, out=G["decoder.embed.0.alpha"])
        r_pe, r_ee = self._rate("pitch_embed"), self._rate("energy_embed")
        ones = torch.ones((Mf, 1), device=self.dev)
        if s["emb_x4"] is not None:        # embedding convs with kernel size > 1: weight + bias gradients by fs2_conv_wgrad (fp32)
            for key, x4, dim, site, rate in (("pitch_embed", s["emb_x4"][0], self.pdim, self.SITE_PITCH_EMB, r_pe),
                                             ("energy_embed", s["emb_x4"][1], self.edim, self.SITE_ENERGY_EMB, r_ee)):
                g = dy
                if rate > 0.0:
                    g = dy.clone()
                    call("fs2_dropout", ptr(g), None, Mf, d, self.seed, site, rate, F32, stream())
                taps = int(G[va + key + ".conv.conv.weight"].shape[-1])
                c4 = x4.shape[1]
                gw = self._empty((d, c4, taps), torch.float32)
                gb = self._empty((d,), torch.float32)
                ks = max(1, min(16, Mf // 256))
                ws = self._ws(hip.load().fs2_conv_wgrad_workspace(d, c4, taps, ks))
                call("fs2_conv_wgrad", ptr(g), ptr(x4), ptr(gw), ptr(gb), ptr(ws), ptr(s["seg_f"]) if taps > 1 else None, Mf, d,
                     c4, taps, d, c4, F32, ks, float(s["xscale"]), 0, stream())
                G[va + key + ".conv.conv.weight"].copy_(gw[:, :dim, :])
                G[va + key + ".conv.conv.bias"].copy_(gb)
        elif r_pe > 0.0 or r_ee > 0.0:     # every embedding has its own mask: gradient = column sums of mask o dy
            for key, vals, dim, site, rate in (("pitch_embed", s["ps"], self.pdim, self.SITE_PITCH_EMB, r_pe),
                                               ("energy_embed", s["es"], self.edim, self.SITE_ENERGY_EMB, r_ee)):
                g = dy
                if rate > 0.0:
                    g = dy.clone()
                    call("fs2_dropout", ptr(g), None, Mf, d, self.seed, site, rate, F32, stream())
                emb = self._empty((dim + 1, d), torch.float32)
                self._colsum(g, emb, Mf, d, g_is_t=False, S=torch.cat([vals, ones], 1).contiguous(), R=dim + 1,
                             scale=s["xscale"])
                G[va + key + ".conv.conv.weight"].copy_(emb[:dim].t().unsqueeze(-1))
                G[va + key + ".conv.conv.bias"].copy_(emb[dim])
        else:
            S = torch.cat([s["ps"], s["es"], ones], 1).contiguous()
            R = self.pdim + self.edim + 1
            emb = self._empty((R, d), torch.float32)
            self._colsum(dy, emb, Mf, d, g_is_t=False, S=S, R=R, scale=s["xscale"])
            G[va + "pitch_embed.conv.conv.weight"].copy_(emb[:self.pdim].t().unsqueeze(-1))
            G[va + "energy_embed.conv.conv.weight"].copy_(emb[self.pdim:self.pdim + self.edim].t().unsqueeze(-1))
            G[va + "pitch_embed.conv.conv.bias"].copy_(emb[R - 1])
            G[va + "energy_embed.conv.conv.bias"].copy_(emb[R - 1])
        if s["xscale"] != 1.0:
            acc = self._empty((Mf, d), torch.float32)
            call("fs2_scale", ptr(dy), float(s["xscale"]), ptr(acc), Mf * d, stream())
        else:
            acc = dy
        # ---- pitch / energy predictors (inputs: hs_lr before the embedding add)
        for pr, saved, out, tgt in ((self.pitch_pred, s["pitch"], s["pitch_out"], s["ps"]),
                                    (self.energy_pred, s["energy"], s["energy_out"], s["es"])):
            n = pr["nout"]
            dout = self._empty((Mf, n), torch.float32)
            call("fs2_loss_grad", ptr(out), ptr(tgt), None, 1, 1.0 / (Mf * n), up, ptr(dout), None, Mf, n, n, self.code,
                 stream())
            self._pred_bwd(pr, saved, dout, s["seg_f"], Mf, d, acc)
        # ---- LengthRegulator backward, duration predictor
        dhs = self._empty((Mp, d), torch.float32)
        call("fs2_lr_scatter", ptr(acc), ptr(s["dur_used"]), ptr(s["cum"]), ptr(s["cu_phone"]), ptr(s["cu_frames"]), B,
             None, 1.0, ptr(dhs), Mp, d, stream())
        dlog = self._empty((Mp, 1), torch.float32)
        call("fs2_loss_grad", ptr(s["logd"]), None, ptr(s["ds"]), 2, 1.0 / Mp, up, ptr(dlog), None, Mp, 1, 1, self.code,
             stream())
        self._pred_bwd(self.dur_pred, s["dur"], dlog, s["seg_p"], Mp, d, dhs)
        # ---- encoder stack
        dx, dparts = self._stack_bwd(self.enc_layers, "encoder.", cfg["encoder_normalized_before"],
                                     cfg["encoder_concat_after"], s["enc"], dhs, None, s["seg_p"], s["cu_phone"], B,
                                     s["max_il"], Mp, adaptor_bucket)
        if dparts is not None or self._rate("enc_pos") > 0.0:
            dx, _ = self._ln_bwd(None, dx, dparts, None, None, Mp, d, want_t=False,
                                 in_drop=(self.SITE_ENC_POS, self._rate("enc_pos")))
        call("fs2_embed_bwd", ptr(s["ids"]), ptr(dx), float(s["xscale"]), ptr(G["encoder.embed.0.weight"]), Mp, d,
             self.idim, 0, stream())
        if self.scaled:
            tmp = self._empty((d,), torch.float32)
            self._colsum(dx, tmp, Mp, d, g_is_t=False, pe=self._pe_table(s["max_il"]), pos=s["pos_p"])
            torch.sum(tmp, 0, out=G["encoder.embed.1.alpha"])
        self._join_wgrad()
        if self.grad_ready is not None:
            self.grad_ready(len(self.bucket_slices) - 1)
        if self.grad_finish is not None:
            self.grad_finish()
        return G
