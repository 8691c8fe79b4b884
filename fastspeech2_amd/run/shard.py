"""Utterance sharding for multi-GPU decoding: one process per GPU, no data-path collective.

The reference's only multi-device construct is single-process nn.DataParallel
(run/train_decode.py:257-262).  Inference utterances are independent, so here each rank takes
length-bucketed batches of its own and nothing is exchanged but, optionally, timings / counts.
Batching follows the idea of run/batchfy.py:6-58 (sort by length, pack into batches)."""
from typing import List, Sequence


def make_buckets(lengths: Sequence[int], batch_size: int = 32) -> List[List[int]]:
    """Indices sorted by descending length and cut into buckets of `batch_size` (length-bucketed
    batches: utterances of similar length share a batch, run/batchfy.py sorts the same way)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]


def shard_round_robin(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Length-sorted utterances dealt round-robin: every rank sees the same length profile, so the
    per-rank cost (which grows with frames and frames^2) is balanced to within one utterance."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return order[rank::world_size]


def shard_buckets(lengths: Sequence[int], world_size: int, rank: int, batch_size: int = 32) -> List[List[int]]:
    """Batches of this rank: its round-robin share of the utterances, bucketed by length."""
    mine = shard_round_robin(lengths, world_size, rank)
    return [mine[i:i + batch_size] for i in range(0, len(mine), batch_size)]


def gather_results(local: dict, world_size: int):
    """Optional: collect {utt_index: python object} dicts on every rank (timings, frame counts).
    Uses torch.distributed (RCCL on GPUs, gloo on CPU); not on the data path."""
    if world_size == 1:
        return dict(local)
    import torch.distributed as dist
    parts = [None] * world_size
    dist.all_gather_object(parts, local)
    out = {}
    for p in parts:
        out.update(p)
    return out
