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
    """Averages the engine's flat gradient buffer across ranks, bucket by bucket."""

    def __init__(self, engine, process_group=None, single_rank_too: bool = False):
        """single_rank_too: run the collectives even with one rank (exercises the RCCL path on a one-GPU box)"""
        self.engine = engine
        self.single_rank_too = single_rank_too
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.cuda = engine.flat.is_cuda
        self.comm = torch.cuda.Stream(device=engine.flat.device) if self.cuda else None
        self.works: List = []
        engine.grad_ready = self.bucket_ready
        engine.grad_finish = self.finish

    def bucket_ready(self, index: int):
        lo, hi = self.engine.bucket_slices[index]
        if hi <= lo or (self.world == 1 and not self.single_rank_too):
            return
        buf = self.engine.flat[lo:hi]
        if self.cuda:
            # order the collective after the bucket's last gradient kernels: the compute stream and, when the
            # engine runs its weight-gradient GEMMs beside it, that stream too — without joining the two
            streams = self.engine.producer_streams() if hasattr(self.engine, "producer_streams") else \
                [torch.cuda.current_stream()]
            for st in streams:
                ev = torch.cuda.Event()
                ev.record(st)
                self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                buf.mul_(1.0 / self.world)
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            buf.mul_(1.0 / self.world)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (called at the end of backward)."""
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm)


class Updater:
    """update_core of run/train_decode.py:82-110: forward, loss / accum_grad, backward; every accum_grad
    calls: clip_grad_norm_, skip the update when the norm is NaN, optimizer.step(), zero_grad()."""

    def __init__(self, model, grad_clip, optimizer, device=None, accum_grad=1, reducer: Optional[GradReducer] = None):
        self.model = model
        self.grad_clip = grad_clip
        self.optimizer = optimizer
        self.device = device
        self.accum_grad = accum_grad
        self.forward_count = 0
        self.iteration = 0
        self.reducer = reducer
        self.last_grad_norm = None

    def update_core(self, batch: Dict[str, torch.Tensor]):
        x = {k: (v.to(self.device) if self.device is not None else v) for k, v in batch.items()}
        loss = self.model(**x).mean() / self.accum_grad
        loss.backward()
        self.forward_count += 1
        if self.forward_count != self.accum_grad:
            return
        self.forward_count = 0
        grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
        self.last_grad_norm = grad_norm
        logging.debug("grad norm=%s", grad_norm)
        if math.isnan(float(grad_norm)):
            logging.warning("grad norm is nan. Do not update model")
        else:
            self.optimizer.step()
        self.optimizer.zero_grad()

    def update(self, batch):
        self.update_core(batch)
        if self.forward_count == 0:
            self.iteration += 1
