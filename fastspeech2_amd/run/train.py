"""Training plumbing without Chainer: the update rule of `CustomUpdater.update_core`
(run/train_decode.py:72-116) and data-parallel gradient averaging in place of `nn.DataParallel`
(run/train_decode.py:257-262).

One process per GPU.  The model's backward (fastspeech2_amd/train_engine.py) writes all parameter
gradients into one flat fp32 buffer, bucketed in the order the backward finishes them; `GradReducer`
all-reduces each bucket over RCCL (torch.distributed backend "nccl") on a side stream as soon as it is
complete, underneath the rest of the backward.  xGMI is a point-to-point mesh, so the buckets are kept
large (one per FFT block, ~4.7 M parameters / 19 MB at BASELINE dims): few, large collectives.
"""
import logging
import math
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    """Averages the engine's flat gradient buffer across ranks, bucket by bucket.

    Two transports:
    * backend "nccl" (RCCL, the production path): every bucket is all-reduced on the device as soon as the backward
      announces it — stream-ordered collectives, no host thread touches the GPU.
    * any other backend with device gradients (gloo: the one-GPU test rig, where RCCL cannot put two ranks on one
      device): the reduced region is staged through ONE pinned host buffer at the end of the backward — copy out,
      host wait, CPU all-reduce on the calling thread, copy back.  torch's own gloo-on-device-tensor path is not
      used: it runs every collective on helper threads that allocate pinned memory, take a high-priority stream from
      the pool per collective (one more hardware queue each) and call hipStreamSynchronize beside the thread that
      is still issuing the backward — with two ranks and the test process sharing one GPU that is the part of the
      old data-parallel test nobody could account for when it stalled once."""

    def __init__(self, engine, process_group=None, single_rank_too: bool = False):
        """single_rank_too: run the collectives even with one rank (exercises the RCCL path on a one-GPU box)"""
        self.engine = engine
        self.single_rank_too = single_rank_too
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.cuda = engine.flat.is_cuda
        self.host_staged = self.cuda and dist.get_backend(process_group) != "nccl"
        # The collectives are issued on the engine's weight-gradient stream when it has one: that stream already runs
        # behind the compute stream, and one stream fewer matters — HIP multiplexes streams onto GPU_MAX_HW_QUEUES
        # (4) hardware queues, and a stream that sits in an event wait stalls whatever shares its queue (a separate
        # communication stream cost the step its compute / weight-gradient overlap: 12.5 vs 8.7 ms with one rank).
        side = getattr(engine, "wgrad_stream", None)
        self.comm = (side if side is not None else torch.cuda.Stream(device=engine.flat.device)) if self.cuda else None
        self.works: List = []
        self._staged = []            # (lo, hi) of the buckets announced in this backward (host-staged transport)
        self._host = None
        engine.grad_ready = self.bucket_ready
        engine.grad_finish = self.finish

    def _producers(self):
        return self.engine.producer_streams() if hasattr(self.engine, "producer_streams") else \
            [torch.cuda.current_stream()]

    def bucket_ready(self, index: int):
        lo, hi = self.engine.bucket_slices[index]
        if hi <= lo or (self.world == 1 and not self.single_rank_too):
            return
        buf = self.engine.flat[lo:hi]
        if self.host_staged:
            self._staged.append((lo, hi))
        elif self.cuda:
            # order the collective after the bucket's last gradient kernels on every producer stream other than
            # the one it is issued on — without joining the streams
            for st in self._producers():
                if st != self.comm:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                buf.mul_(1.0 / self.world)
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            buf.mul_(1.0 / self.world)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def _reduce_through_host(self):
        lo, hi = min(a for a, _ in self._staged), max(b for _, b in self._staged)
        self._staged = []
        cur = torch.cuda.current_stream()
        for st in self._producers():
            if st != cur:
                cur.wait_stream(st)
        if self._host is None or self._host.numel() < hi - lo:
            self._host = torch.empty(hi - lo, dtype=self.engine.flat.dtype, pin_memory=True)
        host = self._host[:hi - lo]
        host.copy_(self.engine.flat[lo:hi], non_blocking=True)
        cur.synchronize()
        host.mul_(1.0 / self.world)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        self.engine.flat[lo:hi].copy_(host, non_blocking=True)

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (called at the end of backward)."""
        if self._staged:
            self._reduce_through_host()
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm)


class Updater:
    """update_core of run/train_decode.py:82-110: forward, loss / accum_grad, backward; every accum_grad
    calls: clip_grad_norm_, skip the update when the norm is NaN, optimizer.step(), zero_grad().

    The reference reads the norm on the host (`math.isnan(grad_norm)`), which drains the GPU once per step.
    With a fused torch Adam on the device (what `get_std_opt` builds) the same rule runs without that read: the
    NaN flag stays a device scalar and is handed to the optimiser as `found_inf` (the GradScaler hook of fused
    Adam: parameters, moments and step counts are left untouched when it is set).  The flag reaches the host
    later through a pinned copy; a skipped step then takes back the schedule tick `NoamOpt.step()` had counted,
    so the learning rates that follow are the reference's (in between, at most two updates see a rate one tick
    ahead).  `sync_nan_check=True` restores the literal host-side check."""

    def __init__(self, model, grad_clip, optimizer, device=None, accum_grad=1, reducer: Optional[GradReducer] = None,
                 sync_nan_check: bool = False):
        self.model = model
        self.grad_clip = grad_clip
        self.optimizer = optimizer
        self.device = device
        self.accum_grad = accum_grad
        self.forward_count = 0
        self.iteration = 0
        self.reducer = reducer
        self.last_grad_norm = None
        self.sync_nan_check = sync_nan_check
        self.skipped_steps = 0
        self._pending = []           # (event, pinned flag) of updates whose NaN flag has not been read yet
        self._params = None          # list(model.parameters()), built once (the walk costs ~0.3 ms)

    def _base_optimizer(self):
        return getattr(self.optimizer, "optimizer", self.optimizer)

    def _resolve_pending(self, wait: bool = False):
        keep = []
        for ev, flag in self._pending:
            if wait:
                ev.synchronize()
            if ev.query():
                if float(flag) != 0.0:
                    self.skipped_steps += 1
                    logging.warning("grad norm is nan. Do not update model")
                    if hasattr(self.optimizer, "_step"):
                        self.optimizer._step -= 1      # NoamOpt.step() was not called by the reference
            else:
                keep.append((ev, flag))
        self._pending = keep

    def update_core(self, batch: Dict[str, torch.Tensor]):
        x = {k: (v.to(self.device) if (self.device is not None and hasattr(v, "to")) else v) for k, v in batch.items()}
        loss = self.model(**x).mean() / self.accum_grad
        loss.backward()
        self.forward_count += 1
        if self.forward_count != self.accum_grad:
            return
        self.forward_count = 0
        if self._params is None:
            self._params = list(self.model.parameters())
        grad_norm = torch.nn.utils.clip_grad_norm_(self._params, self.grad_clip)
        self.last_grad_norm = grad_norm
        base = self._base_optimizer()
        on_device = isinstance(grad_norm, torch.Tensor) and grad_norm.is_cuda and bool(base.defaults.get("fused"))
        if on_device and not self.sync_nan_check:
            self._resolve_pending()
            found = torch.isnan(grad_norm).to(torch.float32).reshape(())
            base.grad_scale, base.found_inf = None, found
            try:
                self.optimizer.step()                  # a no-op on the device when `found` is set
            finally:
                base.found_inf = None
            host = torch.empty((), dtype=torch.float32, pin_memory=True)
            host.copy_(found, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((ev, host))
        else:
            logging.debug("grad norm=%s", grad_norm)
            if math.isnan(float(grad_norm)):
                self.skipped_steps += 1
                logging.warning("grad norm is nan. Do not update model")
            else:
                self.optimizer.step()
        self.optimizer.zero_grad()

    def update(self, batch):
        self.update_core(batch)
        if self.forward_count == 0:
            self.iteration += 1

    def finalize(self):
        """read the outstanding NaN flags (end of training / before a snapshot)"""
        self._resolve_pending(wait=True)
