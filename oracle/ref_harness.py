"""TEST INFRASTRUCTURE — never imported by the product path.

Imports the *reference* (`/root/reference/network/*`) in THIS container only, with the
repair set of SURVEY.md §0.1 applied from the outside (monkeypatch / subclass — the
reference tree is never edited, nothing from it is copied here).  Used by
`tests/golden/make_golden.py` to produce golden vectors and by the container-only
oracle-vs-reference tests.  `/root/reference` does not exist on the GPU box, so everything
here is gated on `available()`.

Repairs applied (numbers are SURVEY.md §0.1 rows; file:line are in /root/reference):
  D13  `import chainer` (network/utils.py:1, tools/plot.py:8-9)    -> stub modules in sys.modules
  D1   ScaledPositionalEncoding.__int__ typo (network/encoder.py:132-137) -> real __init__ that adds alpha
  D2   Encoder.encoders is a python list (network/encoder.py:234-244)     -> wrapped in nn.ModuleList
  D3   `xs = encoder(xs, masks)` (network/encoder.py:250-251)             -> unpack the tuple
  D4   DurationPredictor LayerNorm(dim=-1) on (B,C,T) (variance_adaptor.py:110) -> dim=1
  D5   DurationPredictor.linear on (B,C,T) (variance_adaptor.py:123,134)  -> transpose + squeeze(-1)
  D6   VarianceAdaptor.inference calls .forward (variance_adaptor.py:200) -> calls .inference
  D8   in-place `hs +=` (variance_adaptor.py:192) breaks backward         -> out-of-place (training only)
  D7   PostNet builds n-1 layers, takes idim channels, never emits odim (fastspeech2.py:24-26,:161)
       -> replaced by a repaired class (ESPnet layout: n conv layers k=filts bias=False, BatchNorm1d, tanh on
          all but the last, first conv from odim, last conv to odim); the literal PostNet cannot run, so
          parity for postnet_layers > 0 is pinned against THIS repaired definition only.
D9 (fill_(1) on an all-zero duration row) is observable behaviour and is kept.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FS2_REFERENCE_ROOT", "/root/reference")

_loaded = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "network", "fastspeech2.py"))


def _install_chainer_stub():
    if "chainer" in sys.modules:
        return
    chainer = types.ModuleType("chainer")

    class Chain(object):
        def __init__(self, *a, **k):
            pass

    class _Reporter(types.ModuleType):
        def report(self, values, observer=None):
            return None

    reporter = _Reporter("chainer.reporter")
    training = types.ModuleType("chainer.training")
    extension = types.ModuleType("chainer.training.extension")

    class Trainer(object):
        pass

    class Extension(object):
        pass

    training.Trainer = Trainer
    extension.Extension = Extension
    training.extension = extension
    chainer.Chain = Chain
    chainer.reporter = reporter
    chainer.training = training
    sys.modules["chainer"] = chainer
    sys.modules["chainer.reporter"] = reporter
    sys.modules["chainer.training"] = training
    sys.modules["chainer.training.extension"] = extension


def load():
    """Returns a namespace with the repaired reference classes."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    _install_chainer_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch
    import torch.nn as nn
    import network.encoder as enc
    import network.variance_adaptor as va
    import network.fastspeech2 as fs2

    # ---- D1 -------------------------------------------------------------
    def _spe_init(self, d_model, dropout_rate, max_len=5000):
        enc.PositionalEncoding.__init__(self, d_model, dropout_rate, max_len)
        self.alpha = nn.Parameter(torch.tensor(1.0))
    enc.ScaledPositionalEncoding.__init__ = _spe_init

    # ---- D2 + D3 ----------------------------------------------------------
    _enc_init = enc.Encoder.__init__

    def _encoder_init(self, *a, **k):
        _enc_init(self, *a, **k)
        self.encoders = nn.ModuleList(self.encoders)
    enc.Encoder.__init__ = _encoder_init

    def _encoder_forward(self, xs, masks):
        xs = self.embed(xs)
        for layer in self.encoders:
            xs, masks = layer(xs, masks)
        if self.normalized_before:
            xs = self.after_norm(xs)
        return xs, masks
    enc.Encoder.forward = _encoder_forward

    # ---- D4 + D5 ----------------------------------------------------------
    _dp_init = va.DurationPredictor.__init__

    def _dp_init_fixed(self, *a, **k):
        _dp_init(self, *a, **k)
        for seq in self.conv:
            seq[2].dim = 1
    va.DurationPredictor.__init__ = _dp_init_fixed

    def _dp_body(self, xs):
        xs = xs.transpose(1, 2)
        for f in self.conv:
            xs = f(xs)
        return self.linear(xs.transpose(1, -1)).squeeze(-1)

    def _dp_forward(self, xs, x_masks=None):
        xs = _dp_body(self, xs)
        if x_masks is not None:
            xs = xs.masked_fill(x_masks, 0.0)
        return xs

    def _dp_inference(self, xs, x_masks=None):
        xs = _dp_body(self, xs)
        xs = torch.clamp(torch.round(xs.exp() - self.offset), min=0).long()
        if x_masks is not None:
            xs = xs.masked_fill(x_masks, 0)
        return xs
    va.DurationPredictor.forward = _dp_forward
    va.DurationPredictor.inference = _dp_inference

    # ---- D6 ---------------------------------------------------------------
    def _va_inference(self, hs, input_lengths, duration_masks, variance_masks):
        duration_outs = self.duration_predictor.inference(hs, duration_masks)
        hs = self.length_regulator.forward(hs, duration_outs, input_lengths)
        pitch_outs = self.pitch_predictor.forward(hs, variance_masks)
        pitch_embed = self.pitch_embed.forward(pitch_outs.transpose(1, 2)).transpose(1, 2)
        energy_outs = self.energy_predictor.forward(hs, variance_masks)
        energy_embed = self.energy_embed.forward(energy_outs.transpose(1, 2)).transpose(1, 2)
        hs += energy_embed + pitch_embed
        return hs, duration_outs, pitch_outs, energy_outs
    va.VarianceAdaptor.inference = _va_inference

    # ---- D8 (training forward only) ---------------------------------------
    def _va_forward(self, hs, durations, input_lengths, pitch_target, energy_target,
                    duration_mask, variance_mask):
        duration_outs = self.duration_predictor.forward(hs, duration_mask)
        hs = self.length_regulator.forward(hs, durations, input_lengths)
        pitch_outs = self.pitch_predictor.forward(hs, variance_mask)
        pitch_embed = self.pitch_embed.forward(pitch_target.transpose(1, 2)).transpose(1, 2)
        energy_outs = self.energy_predictor.forward(hs, variance_mask)
        energy_embed = self.energy_embed(energy_target.transpose(1, 2)).transpose(1, 2)
        hs = hs + (pitch_embed + energy_embed)
        return hs, duration_outs, pitch_outs, energy_outs
    va.VarianceAdaptor.forward = _va_forward

    # ---- D7 ---------------------------------------------------------------
    class RepairedPostNet(nn.Module):
        def __init__(self, in_dim, out_dim, n_layers=5, n_chans=512, n_filts=5, dropout_rate=0.5,
                     use_batch_norm=True):
            super().__init__()
            self.postnet = nn.ModuleList()
            for layer in range(n_layers):
                ichans = out_dim if layer == 0 else n_chans          # the mel has odim channels, not idim
                ochans = out_dim if layer == n_layers - 1 else n_chans
                mods = [nn.Conv1d(ichans, ochans, n_filts, stride=1, padding=(n_filts - 1) // 2, bias=False)]
                if use_batch_norm:
                    mods.append(nn.BatchNorm1d(ochans))
                if layer != n_layers - 1:
                    mods.append(nn.Tanh())
                mods.append(nn.Dropout(dropout_rate))
                self.postnet.append(nn.Sequential(*mods))

        def forward(self, xs):
            for layer in self.postnet:
                xs = layer(xs)
            return xs
    fs2.PostNet = RepairedPostNet

    ns = types.SimpleNamespace(enc=enc, va=va, fs2=fs2,
                               FeedForwardTransformer=fs2.FeedForwardTransformer)
    _loaded = ns
    return ns


def build_reference_model(idim, odim, seed, dur_bias=None, **kwargs):
    """Repaired reference model, eval mode, fp32 CPU.  `dur_bias` overrides
    variance_adaptor.duration_predictor.linear.bias so random-init durations are not
    degenerate."""
    import torch
    ns = load()
    torch.manual_seed(seed)
    model = ns.FeedForwardTransformer(idim, odim, **kwargs)
    if dur_bias is not None:
        with torch.no_grad():
            model.variance_adaptor.duration_predictor.linear.bias.fill_(float(dur_bias))
    model.eval()
    return model
