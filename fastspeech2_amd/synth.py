"""Deterministic synthetic weights and utterance batches (numpy PCG64, portable across
torch builds).  Used by bench.py (random-init model of the BASELINE architecture — there are
no checkpoints), by the tests and by tests/golden/make_golden.py, so that a golden fixture
only has to store a seed + inputs + expected outputs instead of a 170 MB state_dict.

`state_dict_spec` enumerates the parameter names/shapes of the reference model
(network/fastspeech2.py:113-167 with repairs D1/D2 of SURVEY.md §0.1); the container-only
test `tests/test_reference_equivalence.py` checks the list against the repaired reference.
"""
import math
from typing import Dict, List, Tuple

import numpy as np

# constructor defaults the spec needs (network/fastspeech2.py:58-104)
_DEFAULTS = dict(
    adim=384, aheads=4, positionwise_layer_type="linear", positionwise_conv_kernel_size=3,
    eunits=1536, elayers=6, encoder_normalized_before=True, encoder_concat_after=False,
    pitch_embed_kernel_size=1, energy_embed_kernel_size=1,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3,
    dlayers=6, dunits=1536, decoder_normalized_before=False, decoder_concat_after=False,
    reduction_factor=1, postnet_layers=5, postnet_filts=5, postnet_chans=256, use_batch_norm=True,
    use_scaled_pos_enc=True,
)

# VariancePredictor defaults that the model constructor does not forward
# (network/variance_adaptor.py:59-63, :164-172)
VAR_PRED_LAYERS = 2
VAR_PRED_CHANS = 384
VAR_PRED_KERNEL = 3
PITCH_DIM = 4          # network/fastspeech2.py:135
ENERGY_DIM = 1         # network/fastspeech2.py:136

# the measured architecture (BASELINE.json configs / SURVEY.md §8d)
BASELINE_OVERRIDES = dict(
    adim=256, aheads=2, elayers=4, dlayers=4, eunits=1024, dunits=1024,
    positionwise_layer_type="conv1d", positionwise_conv_kernel_size=9, postnet_layers=0,
)


def _cfg(kw: dict) -> dict:
    c = dict(_DEFAULTS)
    c.update({k: v for k, v in kw.items() if k in _DEFAULTS})
    return c


def state_dict_spec(idim: int, odim: int, **kw) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(key, shape, kind)] in state_dict order.  kind in {emb, w, b, ln_w, ln_b, alpha}."""
    c = _cfg(kw)
    d = c["adim"]
    spec: List[Tuple[str, Tuple[int, ...], str]] = []

    def stack(prefix, n_layers, units, pre_ln, concat_after, has_emb):
        if has_emb:
            spec.append((prefix + "embed.0.weight", (idim, d), "emb"))
        if c["use_scaled_pos_enc"]:
            spec.append((prefix + ("embed.1.alpha" if has_emb else "embed.0.alpha"), (), "alpha"))
        for i in range(n_layers):
            p = "%sencoders.%d." % (prefix, i)
            for n in ("q", "k", "v", "out"):
                spec.append((p + "self_atten.linear_%s.weight" % n, (d, d), "w"))
                spec.append((p + "self_atten.linear_%s.bias" % n, (d,), "b"))
            if c["positionwise_layer_type"] == "conv1d":
                k = c["positionwise_conv_kernel_size"]
                spec.append((p + "feed_forward.w1.weight", (units, d, k), "w"))
                spec.append((p + "feed_forward.w1.bias", (units,), "b"))
                spec.append((p + "feed_forward.w2.weight", (d, units, k), "w"))
                spec.append((p + "feed_forward.w2.bias", (d,), "b"))
            else:
                spec.append((p + "feed_forward.w1.weight", (units, d), "w"))
                spec.append((p + "feed_forward.w1.bias", (units,), "b"))
                spec.append((p + "feed_forward.w2.weight", (d, units), "w"))
                spec.append((p + "feed_forward.w2.bias", (d,), "b"))
            for n in ("norm1", "norm2"):
                spec.append((p + n + ".weight", (d,), "ln_w"))
                spec.append((p + n + ".bias", (d,), "ln_b"))
            if concat_after:
                spec.append((p + "concat_linear.weight", (d, 2 * d), "w"))
                spec.append((p + "concat_linear.bias", (d,), "b"))
        if pre_ln:
            spec.append((prefix + "after_norm.weight", (d,), "ln_w"))
            spec.append((prefix + "after_norm.bias", (d,), "ln_b"))

    def predictor(prefix, in_dim, n_layers, chans, k, out_dim):
        for j in range(n_layers):
            cin = in_dim if j == 0 else chans
            spec.append(("%sconv.%d.0.conv.weight" % (prefix, j), (chans, cin, k), "w"))
            spec.append(("%sconv.%d.0.conv.bias" % (prefix, j), (chans,), "b"))
            spec.append(("%sconv.%d.2.weight" % (prefix, j), (chans,), "ln_w"))
            spec.append(("%sconv.%d.2.bias" % (prefix, j), (chans,), "ln_b"))
        spec.append((prefix + "linear.weight", (out_dim, chans), "w"))
        spec.append((prefix + "linear.bias", (out_dim,), "b"))

    stack("encoder.", c["elayers"], c["eunits"], c["encoder_normalized_before"],
          c["encoder_concat_after"], True)
    va = "variance_adaptor."
    predictor(va + "duration_predictor.", d, c["duration_predictor_layers"],
              c["duration_predictor_chans"], c["duration_predictor_kernel_size"], 1)
    predictor(va + "pitch_predictor.", d, VAR_PRED_LAYERS, VAR_PRED_CHANS, VAR_PRED_KERNEL, PITCH_DIM)
    spec.append((va + "pitch_embed.conv.conv.weight", (d, PITCH_DIM, c["pitch_embed_kernel_size"]), "w"))
    spec.append((va + "pitch_embed.conv.conv.bias", (d,), "b"))
    predictor(va + "energy_predictor.", d, VAR_PRED_LAYERS, VAR_PRED_CHANS, VAR_PRED_KERNEL, ENERGY_DIM)
    spec.append((va + "energy_embed.conv.conv.weight", (d, ENERGY_DIM, c["energy_embed_kernel_size"]), "w"))
    spec.append((va + "energy_embed.conv.conv.bias", (d,), "b"))
    stack("decoder.", c["dlayers"], c["dunits"], c["decoder_normalized_before"],
          c["decoder_concat_after"], False)
    spec.append(("feat_out.weight", (odim * c["reduction_factor"], d), "w"))
    spec.append(("feat_out.bias", (odim * c["reduction_factor"],), "b"))
    # repaired PostNet: n layers, first conv takes odim channels, the last one emits odim
    n = c["postnet_layers"]
    for j in range(n):
        cin = odim if j == 0 else c["postnet_chans"]
        cout = odim if j == n - 1 else c["postnet_chans"]
        spec.append(("postnet.postnet.%d.0.weight" % j, (cout, cin, c["postnet_filts"]), "w"))
        if c["use_batch_norm"]:
            spec.append(("postnet.postnet.%d.1.weight" % j, (cout,), "ln_w"))
            spec.append(("postnet.postnet.%d.1.bias" % j, (cout,), "ln_b"))
            spec.append(("postnet.postnet.%d.1.running_mean" % j, (cout,), "ln_b"))
            spec.append(("postnet.postnet.%d.1.running_var" % j, (cout,), "var"))
            spec.append(("postnet.postnet.%d.1.num_batches_tracked" % j, (), "nbt"))
    return spec


def synth_state_dict(idim: int, odim: int, seed: int, dur_bias: float = None, alpha: float = 1.0,
                     dur_w_scale: float = 1.0, **kw) -> Dict[str, np.ndarray]:
    """fp32 numpy weights: W ~ N(0, 1/fan_in), b ~ N(0, 0.05²), LN weight 1+0.1 N, LN bias
    0.1 N, embedding N(0,1) with row 0 zero (padding_idx, network/fastspeech2.py:116-120)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for key, shape, kind in state_dict_spec(idim, odim, **kw):
        if kind == "emb":
            w = rng.standard_normal(shape, dtype=np.float32)
            w[0] = 0.0
        elif kind == "w":
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / math.sqrt(fan_in))
        elif kind == "b":
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "ln_w":
            w = np.float32(1.0) + np.float32(0.1) * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "ln_b":
            w = np.float32(0.1) * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "var":
            w = np.float32(0.5) + np.abs(rng.standard_normal(shape, dtype=np.float32))
        elif kind == "nbt":
            out[key] = np.array(0, dtype=np.int64)
            continue
        elif kind == "alpha":
            w = np.array(alpha, dtype=np.float32)
        else:
            raise ValueError(kind)
        out[key] = np.ascontiguousarray(w, dtype=np.float32)
    if dur_bias is not None:
        out["variance_adaptor.duration_predictor.linear.bias"] = np.full((1,), dur_bias, np.float32)
    if dur_w_scale != 1.0:   # narrows the spread of random-init log-durations (bench: ~6.7 frames/phone)
        out["variance_adaptor.duration_predictor.linear.weight"] *= np.float32(dur_w_scale)
    return out


def synth_utterances(n: int, seed: int, idim: int = 80, mean_len: float = 120.0, std_len: float = 15.0,
                     min_len: int = 60, max_len: int = 180, sort: bool = True) -> List[np.ndarray]:
    """Config-2 style batch: lengths clip(round(N(mean,std)),min,max), ids uniform
    in [1, idim), length-sorted (bucketed) when `sort`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.clip(np.rint(rng.normal(mean_len, std_len, size=n)), min_len, max_len).astype(np.int64)
    if sort:
        lens = np.sort(lens)[::-1]
    return [rng.integers(1, idim, size=int(L), dtype=np.int64) for L in lens]


# bench / config-2 model: BASELINE dims, idim = odim = 80, seed 1234 (run/fastspeech_train.py:17);
# duration head tuned so random-init durations average ~6.7 frames per phone (~800 frames / 120 phones)
BENCH_IDIM = 80
BENCH_ODIM = 80
BENCH_SEED = 1234
BENCH_DUR_BIAS = math.log(6.7)
BENCH_DUR_W_SCALE = 0.2


def bench_state_dict() -> Dict[str, np.ndarray]:
    return synth_state_dict(BENCH_IDIM, BENCH_ODIM, BENCH_SEED, dur_bias=BENCH_DUR_BIAS,
                            dur_w_scale=BENCH_DUR_W_SCALE, use_scaled_pos_enc=True, **BASELINE_OVERRIDES)


def algorithmic_flops(ilens, olens, d=256, U=1024, k=9, Ne=4, Nd=4, Cd=256, Cv=384, kp=3, odim=80) -> float:
    """Closed form of SURVEY.md §8d (multiply-add = 2 FLOP), summed over utterances."""
    f = 0.0
    for tp, tm in zip(ilens, olens):
        f += Ne * (tp * (8 * d * d + 4 * k * d * U) + 4 * tp * tp * d)
        f += Nd * (tm * (8 * d * d + 4 * k * d * U) + 4 * tm * tm * d)
        f += tp * (2 * kp * d * Cd + 2 * kp * Cd * Cd + 2 * Cd)
        f += 2 * tm * (2 * kp * d * Cv + 2 * kp * Cv * Cv) + 2 * tm * Cv * 5 + 2 * tm * d * 5
        f += 2 * tm * d * odim
    return f
