"""Generates tests/golden/*.npz by running the REPAIRED REFERENCE itself (imported from
/root/reference through oracle/ref_harness.py — SURVEY.md §0.1 repairs) in the build
container.  The reference cannot travel to the GPU box, these vectors can.

Every fixture stores: the model kwargs, the synthetic-weight seed (weights are regenerated
with fastspeech2_amd.synth.synth_state_dict — numpy PCG64, portable), the inputs and the
reference's outputs.  Nothing of the reference's source text is stored.

    python tests/golden/make_golden.py            # rewrites every fixture
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fastspeech2_amd import synth          # noqa: E402
from oracle import ref_harness             # noqa: E402
from tests import helpers                  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

TINY_A = dict(idim=40, odim=20, adim=64, aheads=2, positionwise_layer_type="conv1d",
              positionwise_conv_kernel_size=9, eunits=128, dunits=128, elayers=2, dlayers=2,
              duration_predictor_chans=64, postnet_layers=0, use_scaled_pos_enc=False)
TINY_B = dict(idim=30, odim=16, adim=128, aheads=2, positionwise_layer_type="linear",
              eunits=256, dunits=256, elayers=1, dlayers=1, duration_predictor_chans=128,
              postnet_layers=0, use_scaled_pos_enc=True)
TINY_C = dict(idim=50, odim=24, adim=192, aheads=2, positionwise_layer_type="conv1d",
              positionwise_conv_kernel_size=3, eunits=192, dunits=320, elayers=1, dlayers=2,
              duration_predictor_chans=96, duration_predictor_layers=3,
              encoder_normalized_before=False, decoder_normalized_before=True,
              encoder_concat_after=True, decoder_concat_after=False,
              postnet_layers=0, use_scaled_pos_enc=True)
YAML_DIMS = dict(idim=60, odim=80, adim=384, aheads=4, positionwise_layer_type="linear",
                 eunits=1536, dunits=1536, elayers=2, dlayers=2, postnet_layers=0,
                 use_scaled_pos_enc=True)
POSTNET_TINY = dict(idim=30, odim=16, adim=128, aheads=2, positionwise_layer_type="linear",
                    eunits=256, dunits=256, elayers=1, dlayers=1, duration_predictor_chans=128,
                    postnet_layers=3, postnet_chans=32, postnet_filts=5, use_batch_norm=True, use_scaled_pos_enc=True)
POSTNET_NOBN = dict(POSTNET_TINY, postnet_layers=2, postnet_chans=64, postnet_filts=3, use_batch_norm=False)
# the shipped param/fastspeech.yaml, all 37 keys as they are (6+6 linear-FFN blocks, d=384, 4 heads, postnet 5x256)
YAML_SHIPPED = dict(idim=70, odim=80)
BASELINE = dict(idim=80, odim=80, use_scaled_pos_enc=True, **synth.BASELINE_OVERRIDES)
GRAD_SAMPLE = 1024


def build(cfg, seed, dur_bias, alpha=1.0, dur_w_scale=1.0):
    kw = dict(cfg)
    idim, odim = kw.pop("idim"), kw.pop("odim")
    ns = ref_harness.load()
    model = ns.FeedForwardTransformer(idim, odim, **kw)
    sd = synth.synth_state_dict(idim, odim, seed, dur_bias=dur_bias, alpha=alpha, dur_w_scale=dur_w_scale, **kw)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    model.eval()
    return model


def ids_for(cfg, lengths, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(1, cfg["idim"], size=L, dtype=np.int64) for L in lengths]


def run_inference_case(name, cfg, seed, dur_bias, lengths, alpha=1.0, duration_control=1.0):
    model = build(cfg, seed, dur_bias, alpha)
    xs = ids_for(cfg, lengths, seed + 1)
    arrays = {}
    margins = []
    for i, x in enumerate(xs):
        xt = torch.from_numpy(x)
        with torch.no_grad():
            ilens = torch.LongTensor([xt.shape[0]])
            x_masks = model._source_mask(ilens)
            hs, _ = model.encoder.forward(xt.unsqueeze(0), x_masks)
            va = model.variance_adaptor
            logd = va.duration_predictor.forward(hs, None)
            d = va.duration_predictor.inference(hs, None)
            d_in = d.clone()
            hs_lr = va.length_regulator.forward(hs, d_in, ilens, duration_control)
            p = va.pitch_predictor.forward(hs_lr, None)
            e = va.energy_predictor.forward(hs_lr, None)
            if duration_control == 1.0:
                mel = model.inference(xt, None)
                before = model._forward(xt.unsqueeze(0), ilens, in_masks=None, is_inference=True)[0][0]
                arrays["mel_before_%d" % i] = before.numpy().astype(np.float32)
            else:
                pe = va.pitch_embed.forward(p.transpose(1, 2)).transpose(1, 2)
                ee = va.energy_embed.forward(e.transpose(1, 2)).transpose(1, 2)
                zs, _ = model.decoder.forward(hs_lr + (ee + pe), None)
                mel = model.feat_out(zs).view(1, -1, model.odim)[0]
        frac = (logd.exp() - 1.0).numpy().ravel()
        margins.append(float(np.min(np.abs(frac - np.floor(frac) - 0.5))))
        arrays["ids_%d" % i] = x
        arrays["mel_%d" % i] = mel.numpy().astype(np.float32)
        arrays["dur_pred_%d" % i] = d[0].numpy().astype(np.int64)      # before LengthRegulator
        arrays["dur_after_lr_%d" % i] = d_in[0].numpy().astype(np.int64)  # D9-visible value
        arrays["logdur_%d" % i] = logd[0].numpy().astype(np.float32)
        arrays["enc_out_%d" % i] = hs[0].numpy().astype(np.float32)
        arrays["pitch_%d" % i] = p[0].numpy().astype(np.float32)
        arrays["energy_%d" % i] = e[0].numpy().astype(np.float32)
        arrays["frames_%d" % i] = np.array(mel.shape[0], dtype=np.int64)
    meta = dict(name=name, cfg=cfg, seed=seed, dur_bias=dur_bias, alpha=alpha,
                duration_control=duration_control, n=len(xs), rounding_margin=min(margins),
                torch=torch.__version__)
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("%-28s n=%d frames=%s margin=%.2e" % (name, len(xs),
          [int(arrays["frames_%d" % i]) for i in range(len(xs))], min(margins)))


def run_train_case(name, cfg, seed, dur_bias, B, Tp, alpha=1.0, ds_lo=0, ds_hi=5, store_inputs=True):
    """Equal-length batch (no padding) so the reference's padded-batch arithmetic equals the
    per-utterance semantics; loss terms read back through a reporter spy.
    store_inputs=False (BASELINE-size case): only the recipe (seed, B, Tp, duration range) is stored, the test
    regenerates the inputs with tests/helpers.py:train_inputs."""
    model = build(cfg, seed, dur_bias, alpha)
    xs, ilens, ys, olens, ds, ps, es = helpers.train_inputs(cfg, seed, B, Tp, ds_lo, ds_hi)
    got = {}

    class Spy:
        def report(self, dicts):
            for d in dicts:
                got.update(d)
    model.reporter = Spy()
    extra = {}
    grad_names = []
    if cfg["postnet_layers"] == 0:
        # gradients left by loss.backward() (run/train_decode.py:94-97; eval-mode modules = dropout off,
        # repair D8): per parameter a strided sample of <= GRAD_SAMPLE elements plus (sum, l2 norm).
        loss = model.forward(torch.from_numpy(xs), torch.from_numpy(ilens), torch.from_numpy(ys),
                             torch.from_numpy(olens), torch.from_numpy(ds), torch.from_numpy(ps),
                             torch.from_numpy(es))
        loss.backward()
        stats = []
        for j, (k, p) in enumerate(model.named_parameters()):
            g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).double()
            step = max(1, -(-g.numel() // GRAD_SAMPLE))
            extra["grad_%d" % j] = g[::step].float().numpy()
            stats.append([float(g.sum()), float(g.norm())])
            grad_names.append(k)
        extra["grad_stats"] = np.array(stats, np.float64)
    else:
        with torch.no_grad():
            loss = model.forward(torch.from_numpy(xs), torch.from_numpy(ilens), torch.from_numpy(ys),
                                 torch.from_numpy(olens), torch.from_numpy(ds), torch.from_numpy(ps),
                                 torch.from_numpy(es))
    meta = dict(name=name, cfg=cfg, seed=seed, dur_bias=dur_bias, alpha=alpha, torch=torch.__version__,
                losses={k: float(v) for k, v in got.items()}, loss=float(loss), grad_names=grad_names,
                grad_sample=GRAD_SAMPLE, inputs=dict(B=B, Tp=Tp, ds_lo=ds_lo, ds_hi=ds_hi), frames=int(olens.sum()))
    stored = dict(xs=xs, ilens=ilens, ys=ys, olens=olens, ds=ds, ps=ps, es=es) if store_inputs else dict(ds=ds)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(json.dumps(meta)), **stored, **extra)
    print("%-28s loss=%.6f %s" % (name, float(loss), {k: round(float(v), 5) for k, v in got.items()}))


def run_long_case(name, cfg, seed, dur_bias, Tp, dur_w_scale):
    """BASELINE config 4 (512 phones -> ~4000 frames) through the reference's own `inference`.  The mel is stored as
    every 16th frame row + the last 8 rows in full, plus the per-frame row sums and the per-bin column sums of ALL
    frames (sampled values + checksums keep the fixture small)."""
    model = build(cfg, seed, dur_bias, dur_w_scale=dur_w_scale)
    x = ids_for(cfg, [Tp], seed + 1)[0]
    xt = torch.from_numpy(x)
    with torch.no_grad():
        ilens = torch.LongTensor([Tp])
        hs, _ = model.encoder.forward(xt.unsqueeze(0), model._source_mask(ilens))
        va = model.variance_adaptor
        logd = va.duration_predictor.forward(hs, None)
        d = va.duration_predictor.inference(hs, None)
        d_in = d.clone()
        hs_lr = va.length_regulator.forward(hs, d_in, ilens, 1.0)
        p = va.pitch_predictor.forward(hs_lr, None)
        e = va.energy_predictor.forward(hs_lr, None)
        mel = model.inference(xt, None).numpy().astype(np.float32)
    frac = (logd.exp() - 1.0).numpy().ravel()
    margin = float(np.min(np.abs(frac - np.floor(frac) - 0.5)))
    T = mel.shape[0]
    rows = helpers.long_mel_rows(T)
    arrays = dict(ids_0=x, dur_pred_0=d[0].numpy().astype(np.int64), dur_after_lr_0=d_in[0].numpy().astype(np.int64),
                  logdur_0=logd[0].numpy().astype(np.float32), enc_out_0=hs[0].numpy().astype(np.float32)[::8],
                  pitch_0=p[0].numpy().astype(np.float32), energy_0=e[0].numpy().astype(np.float32),
                  frames_0=np.array(T, dtype=np.int64), mel_rows_0=mel[rows],
                  mel_rowsum_0=mel.astype(np.float64).sum(1), mel_colsum_0=mel.astype(np.float64).sum(0))
    meta = dict(name=name, cfg=cfg, seed=seed, dur_bias=dur_bias, alpha=1.0, dur_w_scale=dur_w_scale, duration_control=1.0,
                n=1, rounding_margin=margin, torch=torch.__version__)
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("%-28s Tp=%d frames=%d margin=%.2e max dur %d" % (name, Tp, T, margin, int(d.max())))


def run_durations_case(name, utts, note):
    """Durations of a whole BASELINE workload from the reference itself (encoder + duration predictor per utterance, the
    path of network/fastspeech2.py:197-205 and network/variance_adaptor.py:117-138 with repairs D3-D6): per phone the
    log-domain predictor output (fp32) and the rounded duration, per utterance the frame count.  The utterances are
    regenerated by fastspeech2_amd.synth (numpy PCG64), the weights by synth.bench_state_dict(); nothing else is
    stored.  A phone's rounding margin = distance of exp(logd) - 1 from the nearest .5 edge, derivable from logd."""
    kw = dict(synth.BASELINE_OVERRIDES)
    ns = ref_harness.load()
    model = ns.FeedForwardTransformer(synth.BENCH_IDIM, synth.BENCH_ODIM, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.bench_state_dict().items()}, strict=True)
    model.eval()
    logds, durs, frames = [], [], []
    with torch.no_grad():
        for x in utts:
            xt = torch.from_numpy(x)
            ilens = torch.LongTensor([xt.shape[0]])
            hs, _ = model.encoder.forward(xt.unsqueeze(0), model._source_mask(ilens))
            va = model.variance_adaptor
            logds.append(va.duration_predictor.forward(hs, None)[0].numpy().astype(np.float32))
            d = va.duration_predictor.inference(hs, None)[0].numpy().astype(np.int64)
            durs.append(d)
            frames.append(int(d.sum()))
    logd, dur = np.concatenate(logds), np.concatenate(durs)
    assert dur.max() < 32767 and dur.min() >= 0
    frac = np.exp(logd.astype(np.float64)) - 1.0
    margin = np.abs(frac - np.floor(frac) - 0.5)
    meta = dict(name=name, note=note, n=len(utts), phones=int(dur.size), frames=int(sum(frames)), torch=torch.__version__,
                min_margin=float(margin.min()), below_1e5=int((margin < 1e-5).sum()), below_1e4=int((margin < 1e-4).sum()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(json.dumps(meta)), logd=logd,
                        dur=dur.astype(np.int16), frames=np.array(frames, np.int32),
                        ilens=np.array([len(x) for x in utts], np.int32))
    print("%-28s n=%d phones=%d frames=%d min margin %.2e, phones with margin < 1e-5: %d, < 1e-4: %d"
          % (name, len(utts), dur.size, sum(frames), margin.min(), meta["below_1e5"], meta["below_1e4"]))


def durations_cases(only=None):
    if not only or "durations_bench32" in only:
        run_durations_case("durations_bench32", synth.synth_utterances(32, 0, idim=synth.BENCH_IDIM),
                           "BASELINE config 2: the 32 utterances of bench.py (synth.synth_utterances(32, 0), length-sorted)")
    if not only or "durations_config3_256" in only:
        run_durations_case("durations_config3_256", synth.synth_utterances(256, 0, idim=synth.BENCH_IDIM, sort=False),
                           "BASELINE config 3: the 256 utterances of bench.py --scaling strong "
                           "(synth.synth_utterances(256, 0, sort=False))")


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    if only & {"durations_bench32", "durations_config3_256"}:
        durations_cases(only)
        only -= {"durations_bench32", "durations_config3_256"}
        if not only:
            return
    if only:            # python tests/golden/make_golden.py long_512 train_baseline_b32 : just these (new) fixtures
        if "long_512" in only:
            run_long_case("long_512", BASELINE, 1234, math.log(8.8), 512, 0.15)
        if "reduction2_tiny" in only:
            run_inference_case("reduction2_tiny", dict(TINY_B, reduction_factor=2), 18, math.log(3.0), [3, 19, 28], alpha=0.9)
        if "reduction3_postnet" in only:
            run_inference_case("reduction3_postnet", dict(POSTNET_TINY, reduction_factor=3), 19, math.log(2.5), [5, 22])
        if "embed_kernels_tiny" in only:
            run_inference_case("embed_kernels_tiny", dict(TINY_B, pitch_embed_kernel_size=3, energy_embed_kernel_size=5), 20,
                               math.log(3.0), [2, 17, 33], alpha=0.8)
        if "train_embed_kernels_tiny" in only:
            run_train_case("train_embed_kernels_tiny", dict(TINY_B, pitch_embed_kernel_size=3, energy_embed_kernel_size=5), 21,
                           math.log(3.0), B=3, Tp=12, alpha=0.8)
        if "train_baseline_b32" in only:
            run_train_case("train_baseline_b32", BASELINE, 1234, math.log(7.67), B=32, Tp=120, ds_lo=1, ds_hi=13,
                           store_inputs=False)
        return
    run_inference_case("tiny_conv9_plain", TINY_A, 11, math.log(4.0), [1, 7, 23, 40])
    run_inference_case("tiny_linear_scaled", TINY_B, 12, math.log(3.0), [2, 17, 33], alpha=0.8)
    run_inference_case("tiny_conv3_mixed_ln", TINY_C, 13, math.log(5.0), [5, 12, 31], alpha=1.1)
    run_inference_case("yaml_dims_2x2", YAML_DIMS, 14, math.log(4.0), [9, 26], alpha=1.0)
    run_inference_case("d9_zero_durations", TINY_A, 11, -10.0, [6, 15])
    run_inference_case("duration_control_0p8", TINY_A, 11, math.log(4.0), [7, 23], duration_control=0.8)
    run_inference_case("duration_control_1p3", TINY_A, 11, math.log(4.0), [7, 23], duration_control=1.3)
    run_inference_case("baseline_4x4", BASELINE, 1234, math.log(7.67), [120, 61])
    run_inference_case("postnet_tiny_bn", POSTNET_TINY, 15, math.log(3.0), [4, 21, 30], alpha=0.9)
    run_inference_case("postnet_tiny_nobn", POSTNET_NOBN, 16, math.log(3.0), [9, 17], alpha=1.0)
    run_inference_case("reduction2_tiny", dict(TINY_B, reduction_factor=2), 18, math.log(3.0), [3, 19, 28], alpha=0.9)
    run_inference_case("reduction3_postnet", dict(POSTNET_TINY, reduction_factor=3), 19, math.log(2.5), [5, 22])
    run_inference_case("embed_kernels_tiny", dict(TINY_B, pitch_embed_kernel_size=3, energy_embed_kernel_size=5), 20,
                       math.log(3.0), [2, 17, 33], alpha=0.8)
    import yaml
    shipped = dict(YAML_SHIPPED, **yaml.safe_load(open(os.path.join(ROOT, "fastspeech2_amd", "param", "fastspeech.yaml"))))
    run_inference_case("yaml_shipped_full", shipped, 17, math.log(4.0), [25, 12])
    run_train_case("train_postnet_tiny", POSTNET_TINY, 15, math.log(3.0), B=2, Tp=11, alpha=0.9)
    run_train_case("train_tiny_conv9", TINY_A, 11, math.log(4.0), B=3, Tp=9)
    run_train_case("train_tiny_linear", TINY_B, 12, math.log(3.0), B=2, Tp=14, alpha=0.8)
    run_train_case("train_tiny_conv3_mixed", TINY_C, 13, math.log(5.0), B=2, Tp=10, alpha=1.1)
    run_train_case("train_embed_kernels_tiny", dict(TINY_B, pitch_embed_kernel_size=3, energy_embed_kernel_size=5), 21,
                   math.log(3.0), B=3, Tp=12, alpha=0.8)
    run_long_case("long_512", BASELINE, 1234, math.log(8.8), 512, 0.15)
    run_train_case("train_baseline_b32", BASELINE, 1234, math.log(7.67), B=32, Tp=120, ds_lo=1, ds_hi=13, store_inputs=False)
    durations_cases()


if __name__ == "__main__":
    main()
