"""Decode entry point with the reference's name and argument list (run/fastspeech_decode.py:9).

The reference body is an unfinished stub written against an older model API: it puts the model in
eval mode under no_grad, walks a Kaldi-backed validation set with batch size 1, calls
`model.inference(token_ids, src_lengths, duration_control=1.0, pitch_control=1.0,
energy_control=1.0)` expecting eight return values, and ends in `# WIP` before restoring train mode
(:12-37).  This counterpart keeps that contract — eval + no_grad, the three control knobs, the eight
fields, `model.train()` at the end — but decodes length-bucketed batches on the HIP engine and takes
its utterances from an in-memory / .npz source (Kaldi I/O is out of scope: kaldi_io is not a
dependency).

cfg attributes used (all optional, reference-style nesting):
    cfg.DATASET.VAL.UTTERANCES : {utt_id: 1-D int array of phone ids}  or a path to an .npz of such arrays
    cfg.DECODE.BATCH_SIZE (32), cfg.DECODE.DURATION_CONTROL / PITCH_CONTROL / ENERGY_CONTROL (1.0)
    cfg.DECODE.OUT_NPZ : path to write {utt_id: mel[T, odim]} (optional)
"""
from typing import Dict

import numpy as np
import torch

from .shard import shard_buckets


def _get(ns, path, default=None):
    cur = ns
    for p in path.split("."):
        if cur is None:
            return default
        cur = cur.get(p) if isinstance(cur, dict) else getattr(cur, p, None)
    return default if cur is None else cur


def load_utterances(src) -> Dict[str, np.ndarray]:
    if isinstance(src, dict):
        return {k: np.asarray(v, dtype=np.int64) for k, v in src.items()}
    z = np.load(src, allow_pickle=False)
    return {k: z[k].astype(np.int64) for k in z.files}


def decode(cfg, model, current_step=0, criterion=None, rank: int = 0, world_size: int = 1):
    """Returns {utt_id: dict(mel, mel_postnet, durations, pitch, energy, src_mask, mel_mask, mel_length)}
    for the utterances of this rank (all of them when world_size == 1)."""
    utts = load_utterances(_get(cfg, "DATASET.VAL.UTTERANCES"))
    bs = int(_get(cfg, "DECODE.BATCH_SIZE", 32))
    dc = float(_get(cfg, "DECODE.DURATION_CONTROL", 1.0))
    pc = float(_get(cfg, "DECODE.PITCH_CONTROL", 1.0))
    ec = float(_get(cfg, "DECODE.ENERGY_CONTROL", 1.0))
    ids = sorted(utts.keys())
    lengths = [len(utts[k]) for k in ids]
    was_training = model.training
    model.eval()
    results = {}
    with torch.no_grad():
        for bucket in shard_buckets(lengths, world_size, rank, bs):
            xs = [torch.from_numpy(utts[ids[i]]) for i in bucket]
            out = model.infer_batch(xs, duration_control=dc, pitch_control=pc, energy_control=ec)
            mel = out["mel"].cpu().numpy()                                   # after the PostNet (== before when there is none)
            before = out["packed"]["mel_before"].cpu().numpy()
            cu = np.concatenate([[0], np.cumsum([int(v) for v in out["olens"]])])
            for j, i in enumerate(bucket):
                T, L = int(out["olens"][j]), lengths[i]
                results[ids[i]] = dict(
                    mel=before[cu[j]:cu[j + 1]], mel_postnet=mel[j, :T],
                    durations=out["durations"][j, :L].cpu().numpy(),
                    pitch=out["pitch"][j, :T].cpu().numpy(), energy=out["energy"][j, :T].cpu().numpy(),
                    src_mask=out["src_mask"][j, :L].numpy(), mel_mask=out["mel_mask"][j, :T].numpy(),
                    mel_length=T)
    out_npz = _get(cfg, "DECODE.OUT_NPZ")
    if out_npz:
        np.savez_compressed(out_npz, **{k: v["mel_postnet"] for k, v in results.items()})
    if was_training:
        model.train()                                                  # run/fastspeech_decode.py:37
    return results
