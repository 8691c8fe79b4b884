"""Training driver: what `train(args)` of the reference's run/train_decode.py:226-393 does, without Chainer and Kaldi.

Kept from the reference: `build_model(idim, odim, args)` (:153-192), `CustomConverter` (:118-149), the json
"utts" description of a corpus and `make_batchset(..., batch_bins=...)` (:289-304), model.json in `outdir`
(:242-247), the Noam / Adam optimiser choice (:268-273), the update rule (`Updater` = update_core, :82-110),
evaluation every `eval_interval_epochs` (`CustomEvaluator`, :31-59: mean of the reported losses over the validation
batches, no gradients), a snapshot every `save_interval_epochs` and `model.loss.best` (:342-347), the report line
every `report_interval_iters` (:378-381), sortagrad (:284-286, :388-390: shortest batches first for the first
epochs, shuffled afterwards).  `resume` works here (the reference raises NotImplementedError, :334-335).

Different on purpose:
* features come from an `.npz` store or are synthetic — `LoadInputsAndTargets` reads Kaldi ark files and kaldi_io is
  not part of this build.  Store layout: `<utt>/tokens` int [L], `<utt>/feat` float [T, odim], `<utt>/dur` int [L]
  (sum == T), `<utt>/var` float [T, 5] (4 pitch columns + energy, :146-148);
* multi-GPU is one process per GPU (`torchrun --nproc-per-node N -m fastspeech2_amd.run.train_decode ...`) with the
  bucketed RCCL gradient average of `GradReducer` instead of `nn.DataParallel` (:257-262); every rank takes every
  world-th minibatch;
* plots, TensorBoard and attention figures are left to the caller (`model.calculate_all_attentions` is there).

    python -m fastspeech2_amd.run.train_decode --synthetic 256 --outdir exp/demo --epochs 2
"""
import argparse
import json
import logging
import os
import time
from typing import Dict, List, Sequence

import numpy as np
import torch

from ..network.fastspeech2 import FeedForwardTransformer
from ..network.optimizer import get_std_opt
from ..network.utils import pad_list
from ..param.load_param import load_config
from .batchfy import make_batchset
from .snapshot import load_snapshot, save_snapshot
from .train import GradReducer, Updater

MODEL_KEYS = (
    "adim aheads positionwise_layer_type positionwise_conv_kernel_size eunits elayers transformer_enc_dropout_rate "
    "transformer_enc_positional_dropout_rate transformer_enc_atten_dropout_rate encoder_normalized_before "
    "encoder_concat_after pitch_embed_kernel_size pitch_embed_dropout energy_embed_kernel_size energy_embed_dropout "
    "duration_predictor_layers duration_predictor_chans duration_predictor_kernel_size duration_predictor_dropout_rate "
    "dlayers dunits transformer_dec_dropout_rate transformer_dec_positional_dropout_rate "
    "transformer_dec_atten_dropout_rate decoder_normalized_before decoder_concat_after reduction_factor postnet_layers "
    "postnet_filts postnet_chans postnet_dropout_rate transformer_init initial_encoder_alpha initial_decoder_alpha "
    "use_masking use_batch_norm use_scaled_pos_enc").split()


def build_model(idim: int, odim: int, args) -> FeedForwardTransformer:
    """run/train_decode.py:153-192 (plus `elayers`, which the reference forgets to pass on)"""
    kw = {k: getattr(args, k) for k in MODEL_KEYS if getattr(args, k, None) is not None}
    return FeedForwardTransformer(idim=idim, odim=odim, **kw)


class CustomConverter:
    """One minibatch of loaded arrays -> the padded tensors `forward` takes (run/train_decode.py:118-149)."""

    def __call__(self, batch, device=torch.device("cpu")):
        assert len(batch) == 1
        xs, spkids, ys, durs, variances = batch[0]
        ilens = torch.tensor([x.shape[0] for x in xs], dtype=torch.long, device=device)
        olens = torch.tensor([y.shape[0] for y in ys], dtype=torch.long, device=device)
        xs = pad_list([torch.from_numpy(np.asarray(x)).long() for x in xs], 0).to(device)
        ys = pad_list([torch.from_numpy(np.asarray(y)).float() for y in ys], 0).to(device)
        labels = ys.new_zeros(ys.shape[0], ys.shape[1])
        for i, l in enumerate(olens):
            labels[i, int(l) - 1:] = 1.0
        var = pad_list([torch.from_numpy(np.asarray(v)).float() for v in variances], 0).to(device)
        return {"xs": xs, "ilens": ilens, "ys": ys, "labels": labels, "olens": olens,
                "ds": pad_list([torch.from_numpy(np.asarray(d)).long() for d in durs], 0).to(device),
                "spkids": torch.as_tensor(list(spkids), dtype=torch.long, device=device),
                "ps": var[:, :, :4], "es": var[:, :, 4:]}


class NpzCorpus:
    """`utts` (the json description make_batchset works on) + arrays, from an .npz store or synthetic."""

    def __init__(self, arrays: Dict[str, np.ndarray]):
        self.arrays = arrays
        ids = sorted({k.rsplit("/", 1)[0] for k in arrays})
        self.utts = {}
        for u in ids:
            tok, feat = arrays[u + "/tokens"], arrays[u + "/feat"]
            if int(np.sum(arrays[u + "/dur"])) != feat.shape[0] or arrays[u + "/var"].shape != (feat.shape[0], 5):
                raise ValueError("utterance %s: sum(dur) / var rows do not match the %d feature frames" % (u, feat.shape[0]))
            self.utts[u] = {"input": [{"name": "input1", "shape": [int(feat.shape[0]), int(feat.shape[1])]}],
                            "output": [{"name": "target1", "shape": [int(tok.shape[0]), int(self.idim_hint(arrays))],
                                        "tokenid": " ".join(str(int(t)) for t in tok)}]}

    @staticmethod
    def idim_hint(arrays) -> int:
        return int(max(int(v.max()) for k, v in arrays.items() if k.endswith("/tokens"))) + 1

    @classmethod
    def load(cls, path: str) -> "NpzCorpus":
        z = np.load(path, allow_pickle=False)
        return cls({k: z[k] for k in z.files})

    @classmethod
    def synthetic(cls, n: int, seed: int = 0, idim: int = 80, odim: int = 80, mean_len: float = 60.0) -> "NpzCorpus":
        rng = np.random.Generator(np.random.PCG64(seed))
        arrays = {}
        for j in range(n):
            L = int(np.clip(round(rng.normal(mean_len, mean_len / 8)), 4, 4 * mean_len))
            dur = rng.integers(1, 13, size=L)
            T = int(dur.sum())
            u = "syn%05d" % j
            arrays[u + "/tokens"] = np.concatenate([rng.integers(1, idim - 1, size=L - 1), [idim - 1]]).astype(np.int64)
            arrays[u + "/feat"] = rng.standard_normal((T, odim), dtype=np.float32)
            arrays[u + "/dur"] = dur.astype(np.int64)
            arrays[u + "/var"] = rng.standard_normal((T, 5), dtype=np.float32)
        return cls(arrays)

    def save(self, path: str):
        np.savez_compressed(path, **self.arrays)

    def load_batch(self, minibatch: Sequence) -> tuple:
        """the tuple CustomConverter expects from LoadInputsAndTargets (run/train_decode.py:123)"""
        ids = [u for u, _ in minibatch]
        a = self.arrays
        return ([a[u + "/tokens"] for u in ids], [0] * len(ids), [a[u + "/feat"] for u in ids],
                [a[u + "/dur"] for u in ids], [a[u + "/var"] for u in ids])


def evaluate(model, corpus: NpzCorpus, batchset: List, converter, device) -> Dict[str, float]:
    """CustomEvaluator.evaluate (run/train_decode.py:31-59): forward only, reported values averaged over the batches"""
    was_training = model.training
    model.eval()
    sums: Dict[str, float] = {}
    with torch.no_grad():
        for mb in batchset:
            model(**converter([corpus.load_batch(mb)], device))
            for k, v in model.reporter.observation.items():
                sums[k] = sums.get(k, 0.0) + float(v)
    if was_training:
        model.train()
    return {"validation/main/" + k: v / max(len(batchset), 1) for k, v in sums.items()}


def train(args) -> Dict[str, float]:
    if not torch.cuda.is_available():
        raise RuntimeError("cuda (HIP) is not available: the training path runs only on the MI355X kernels")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed)

    train_c = NpzCorpus.synthetic(args.synthetic, args.seed) if args.synthetic else NpzCorpus.load(args.train_npz)
    valid_c = NpzCorpus.synthetic(max(args.synthetic // 8, 4), args.seed + 1) if args.synthetic else \
        (NpzCorpus.load(args.valid_npz) if args.valid_npz else None)
    first = next(iter(train_c.utts.values()))
    idim = max(int(first["output"][0]["shape"][1]), args.idim or 0)      # run/train_decode.py:235-236 (swapped io)
    odim = int(first["input"][0]["shape"][1])
    os.makedirs(args.outdir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.outdir, "model.json"), "wb") as f:      # :242-247
            f.write(json.dumps((idim, odim, vars(args)), indent=4, ensure_ascii=False, sort_keys=True).encode("utf_8"))

    model = build_model(idim, odim, args).to(device)
    model.set_precision(args.precision)
    model.train()
    if args.opt == "adam":                                                   # :268-273
        optimizer = torch.optim.Adam(model.parameters(), args.lr, eps=args.eps, weight_decay=args.weight_decay,
                                     fused=True)
    elif args.opt == "noam":
        optimizer = get_std_opt(model, args.adim, args.transformer_warmup_steps, args.transformer_lr)
    else:
        raise NotImplementedError(f"unknown optimizer: {args.opt}")
    reducer = GradReducer(model.train_engine()) if world > 1 else None
    updater = Updater(model, args.grad_clip, optimizer, device, args.accum_grad, reducer=reducer)
    start_epoch, trainer_state = 0, {}
    model.train_engine().base_seed = (0x5EED ^ int(args.seed)) & 0xFFFFFFFF          # dropout masks follow --seed
    if args.resume:
        trainer_state = load_snapshot(args.resume, model, optimizer, updater, map_location=device)
        start_epoch = int(trainer_state.get("epoch", 0))

    use_sortagrad = args.sortagrad == -1 or args.sortagrad > 0               # :284-286
    mk = lambda c: make_batchset(c.utts, args.minibatches, min_batch_size=1, shortest_first=use_sortagrad,
                                 batch_sort_key="input", swap_io=True, count=args.batch_count, batch_bins=args.batch_bins)
    train_batches = mk(train_c)
    valid_batches = mk(valid_c) if valid_c is not None else []
    converter = CustomConverter()
    rng = np.random.Generator(np.random.PCG64(args.seed))
    if "shuffle_rng" in trainer_state:                     # resume: go on with the shuffle sequence, do not replay it
        st = rng.bit_generator.state
        st["state"] = {"state": int(trainer_state["shuffle_rng"][0]), "inc": int(trainer_state["shuffle_rng"][1])}
        rng.bit_generator.state = st
    best = float(trainer_state.get("best", float("inf")))  # resume: a worse validation must not replace model.loss.best
    t0, summary = time.time(), {}
    for epoch in range(start_epoch, args.epochs):
        order = np.arange(len(train_batches))
        shuffle = not use_sortagrad or (args.sortagrad != -1 and epoch >= args.sortagrad)     # :388-390
        if shuffle:
            rng.shuffle(order)
        order = order[: len(order) // world * world][rank::world] if world > 1 else order
        for j in order:
            updater.update(converter([train_c.load_batch(train_batches[j])], device))
            if rank == 0 and updater.forward_count == 0 and updater.iteration % args.report_interval_iters == 0:
                obs = {k: float(v) for k, v in model.reporter.observation.items()}   # (reads the device: a sync)
                logging.info("epoch %d iteration %d elapsed %.1fs %s", epoch, updater.iteration, time.time() - t0,
                             " ".join("main/%s=%.4f" % kv for kv in sorted(obs.items())))
                summary.update({"main/" + k: v for k, v in obs.items()})
        done = epoch + 1
        if valid_batches and done % args.eval_interval_epochs == 0:
            val = evaluate(model, valid_c, valid_batches, converter, device)
            summary.update(val)
            if rank == 0:
                logging.info("epoch %d %s", done, " ".join("%s=%.4f" % kv for kv in sorted(val.items())))
                if val.get("validation/main/loss", float("inf")) < best:                  # :346-347
                    best = val["validation/main/loss"]
                    torch.save(model.state_dict(), os.path.join(args.outdir, "model.loss.best"))
        if rank == 0 and done % args.save_interval_epochs == 0:
            pcg = rng.bit_generator.state["state"]
            save_snapshot(os.path.join(args.outdir, "snapshot.ep.%d" % done), model, optimizer, updater,
                          extra={"epoch": done, "best": best, "shuffle_rng": [str(pcg["state"]), str(pcg["inc"])]})
    updater.finalize()
    summary["iterations"] = updater.iteration
    summary["skipped_steps"] = updater.skipped_steps
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return summary


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--train-npz", default=None)
    ap.add_argument("--valid-npz", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many random utterances instead of --train-npz")
    ap.add_argument("--idim", type=int, default=0, help="vocabulary size when it is larger than the largest token id + 1")
    ap.add_argument("--model-config", default=None, help="yaml of FeedForwardTransformer arguments (param/fastspeech.yaml)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--opt", default="noam", choices=["adam", "noam"])
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--eps", type=float, default=1e-6)
    ap.add_argument("--weight-decay", type=float, default=0.0)
    ap.add_argument("--transformer-warmup-steps", type=int, default=4000)
    ap.add_argument("--transformer-lr", type=float, default=1.0)
    ap.add_argument("--grad-clip", type=float, default=1.0)
    ap.add_argument("--accum-grad", type=int, default=1)
    ap.add_argument("--batch-bins", type=int, default=2000000)
    ap.add_argument("--batch-count", default="auto")
    ap.add_argument("--minibatches", type=int, default=0)
    ap.add_argument("--sortagrad", type=int, default=0)
    ap.add_argument("--eval-interval-epochs", type=int, default=1)
    ap.add_argument("--save-interval-epochs", type=int, default=1)
    ap.add_argument("--report-interval-iters", type=int, default=100)
    for k in MODEL_KEYS:
        ap.add_argument("--" + k.replace("_", "-"), dest=k, default=None,
                        type=(lambda s: s if not s.lstrip("-").replace(".", "", 1).isdigit() else (float(s) if "." in s else int(s))))
    return ap


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    args = get_parser().parse_args(argv)
    if args.model_config:
        for k, v in load_config(args.model_config).items():
            if k in MODEL_KEYS and getattr(args, k) is None:
                setattr(args, k, v)
    for k in ("encoder_normalized_before", "encoder_concat_after", "decoder_normalized_before", "decoder_concat_after",
              "use_masking", "use_batch_norm", "use_scaled_pos_enc"):
        v = getattr(args, k)
        if isinstance(v, str):
            setattr(args, k, v.lower() in ("1", "true", "yes"))
    if args.adim is None:
        args.adim = 384                                   # the constructor's default, needed by the Noam schedule
    if not args.synthetic and not args.train_npz:
        raise SystemExit("give --train-npz or --synthetic N")
    summary = train(args)
    print(json.dumps(summary, sort_keys=True))


if __name__ == "__main__":
    main()
