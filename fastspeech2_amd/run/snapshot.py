"""Snapshot / resume without Chainer.  The reference's tools/snapshot.py:11-47 saves
{"trainer": <chainer trainer state>, "model": model.state_dict(), "optimizer": optimizer.state_dict()} through
a temporary file that is moved into place.  This keeps the same three keys ("trainer" holds the plain
counters of fastspeech2_amd.run.train.Updater instead of a Chainer serializer dump), so the "model" and
"optimizer" parts interchange with a snapshot written by the reference (state_dict names are the reference's)."""
import os
import shutil
import tempfile

import torch


def _train_engine_of(model):
    return getattr(model, "_train_engine", None)


def save_snapshot(path, model, optimizer, updater=None, save_func=torch.save, extra=None):
    """extra: further plain values for the "trainer" dict (epoch, best validation loss, shuffle RNG state), written in
    the same single pass."""
    model = model.module if hasattr(model, "module") else model              # tools/snapshot.py:22-31
    if hasattr(updater, "finalize"):
        updater.finalize()            # outstanding device-side NaN flags: a skipped step gives back its schedule tick
    trainer = {"iteration": getattr(updater, "iteration", 0), "forward_count": getattr(updater, "forward_count", 0)}
    eng = _train_engine_of(model)
    if eng is not None:
        trainer["dropout_step"] = int(eng.step)          # counter behind the dropout masks: a resumed run goes on
    trainer.update(extra or {})
    snap = {"trainer": trainer, "model": model.state_dict(), "optimizer": optimizer.state_dict()}
    out_dir = os.path.dirname(os.path.abspath(path))
    os.makedirs(out_dir, exist_ok=True)
    tmpdir = tempfile.mkdtemp(prefix="tmp" + os.path.basename(path), dir=out_dir)
    try:
        tmp = os.path.join(tmpdir, os.path.basename(path))
        save_func(snap, tmp)
        shutil.move(tmp, path)                                              # atomic within one file system
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return path


def load_snapshot(path, model, optimizer=None, updater=None, map_location="cpu"):
    """Restores what save_snapshot wrote (tensors only: weights_only load).  Returns the "trainer" dict."""
    snap = torch.load(path, map_location=map_location, weights_only=True)
    model = model.module if hasattr(model, "module") else model
    model.load_state_dict(snap["model"], strict=True)
    if optimizer is not None and "optimizer" in snap:
        optimizer.load_state_dict(snap["optimizer"])
    trainer = snap.get("trainer", {}) or {}
    if updater is not None and isinstance(trainer, dict):
        updater.iteration = int(trainer.get("iteration", 0))
        updater.forward_count = int(trainer.get("forward_count", 0))
    if isinstance(trainer, dict) and "dropout_step" in trainer and hasattr(model, "train_engine") \
            and next(model.parameters()).is_cuda:
        model.train_engine().step = int(trainer["dropout_step"])
    return trainer
