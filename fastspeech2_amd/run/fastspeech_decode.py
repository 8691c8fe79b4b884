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
    buckets = list(shard_buckets(lengths, world_size, rank, bs))
    with torch.no_grad():
        eng = model.engine()
        # the whole list of buckets goes through the throughput path: phone-level half of bucket k+1 under the
        # frame-level half of bucket k, frame-level halves alternating over two streams (Engine.forward_pipelined)
        feed = []
        for bucket in buckets:
            xs = [utts[ids[i]] for i in bucket]
            feed.append((torch.from_numpy(np.concatenate(xs)).to(eng.dev), [len(x) for x in xs]))
        outs = eng.forward_pipelined(feed, frame_streams=2, phone_group=2, duration_control=dc, pitch_control=pc,
                                     energy_control=ec)
        for bucket, out in zip(buckets, outs):
            out["ready"].synchronize()
            mel = out["mel"].cpu().numpy()                                   # after the PostNet (== before without one)
            before = out["mel_before"].cpu().numpy()
            dur = out["durations"].cpu().numpy()
            pitch, energy = out["pitch"].cpu().numpy(), out["energy"].cpu().numpy()
            cu_f = np.concatenate([[0], np.cumsum(out["olens"])])         # decoder frames: pitch / energy rows
            cu_m = np.concatenate([[0], np.cumsum(out["mel_lens"])])      # mel rows (= reduction_factor x those)
            cu_p = np.concatenate([[0], np.cumsum([lengths[i] for i in bucket])])
            for j, i in enumerate(bucket):
                T, L = int(out["mel_lens"][j]), lengths[i]
                f0, f1, p0, p1, m0, m1 = cu_f[j], cu_f[j + 1], cu_p[j], cu_p[j + 1], cu_m[j], cu_m[j + 1]
                results[ids[i]] = dict(
                    mel=before[m0:m1], mel_postnet=mel[m0:m1], durations=dur[p0:p1].astype(np.int64),
                    pitch=pitch[f0:f1], energy=energy[f0:f1], src_mask=np.ones(L, bool), mel_mask=np.ones(T, bool),
                    mel_length=T)
    out_npz = _get(cfg, "DECODE.OUT_NPZ")
    if out_npz:
        np.savez_compressed(out_npz, **{k: v["mel_postnet"] for k, v in results.items()})
    if was_training:
        model.train()                                                  # run/fastspeech_decode.py:37
    return results
