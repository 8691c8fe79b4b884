"""Minibatches under a "bins" budget for length-sorted TTS corpora.

Counterpart of run/batchfy.py:6-123 (`batchfy_by_bin`, `make_batchset`): same arguments, same ESPnet-style json
({utt_id: {"input": [{"shape": [T, D]}], "output": [{"shape": [T, D]}], "category": ...}}) and — checked against
the reference on random corpora in tests/test_train_host.py — the same batches, so a training script can swap the
import.  Only the modes the reference implements exist: count "bin" (or "auto" with batch_bins != 0),
swap_io=True, sort key input / output.

The contract, restated (this file is written from it, not from the reference's loops):
  * the corpus arrives sorted by length; batches are consecutive runs of it;
  * a run may grow while   (largest output size in the run + input size of the newest member) x members
    stays within `batch_bins` (sizes in elements = frames x feature dim); a run that hits the budget exactly is
    closed there;
  * every run holds at least `min_batch_size` utterances, budget or not; when the corpus ends inside a run, the
    short tail is filled from the FRONT of the run before it, and the deficit this opens is passed on the same way
    towards the first run; a first run that ends up short is joined to the second.
"""
import logging

import numpy as np

BATCH_COUNT_CHOICES = ["auto", "seq", "bin", "frame"]
BATCH_SORT_KEY_CHOICES = ["input", "output", "shuffle"]


def _sizes(items, key):
    """elements (frames x dim of the FIRST utterance, as the reference sizes them) per utterance"""
    dim = int(items[0][1][key][0]["shape"][1])
    return np.array([int(e[key][0]["shape"][0]) for _, e in items], dtype=np.int64) * dim


def _run_length(in_sz, out_sz, budget):
    """How many utterances of this (remaining) stretch fit one batch: the longest prefix whose every step stays
    within the budget, closed early at a step that meets the budget exactly."""
    members = np.arange(1, in_sz.size + 1, dtype=np.int64)
    cost = (np.maximum.accumulate(out_sz) + in_sz) * members
    reached = np.flatnonzero(cost >= budget)
    if reached.size == 0:
        return int(in_sz.size)
    first = int(reached[0])
    return first + 1 if cost[first] == budget else first


def _cut_points(in_sz, out_sz, budget, min_batch_size):
    """[start, stop) of consecutive batches over the sorted corpus"""
    total, cuts, start = in_sz.size, [], 0
    while start < total:
        # a window of `budget // smallest possible cost` would do; the stretch is short enough to scan whole
        n = _run_length(in_sz[start:], out_sz[start:], budget)
        stop = min(total, start + max(min_batch_size, n))
        cuts.append((start, stop))
        start = stop
    return cuts


def _refill_short_tail(batches, min_batch_size):
    """Only the last batch can be short (the corpus ended inside it).  It takes what it lacks from the front of its
    predecessor, which may then be short itself, and so on towards the first batch."""
    j = len(batches) - 1
    while j >= 0 and len(batches[j]) < min_batch_size:
        if j == 0:
            if len(batches) > 1:                     # a short first batch is joined to the second
                batches[1].extend(batches[0])
                del batches[0]
            break
        lack = min_batch_size - len(batches[j])
        batches[j].extend(batches[j - 1][:lack])
        del batches[j - 1][:lack]
        j -= 1
    return batches


def batchfy_by_bin(sorted_data, batch_bins, num_batches=0, min_batch_size=1, shortest_first=False,
                   ikey="input", okey="output"):
    if batch_bins <= 0:
        raise ValueError(f"invalid batch_bins: {batch_bins}")
    items = list(sorted_data)
    logging.info(f"#utts: {len(items)}")
    if not items:
        return []
    in_sz, out_sz = _sizes(items, ikey), _sizes(items, okey)
    if int((in_sz + out_sz).min()) <= 0:
        raise ValueError(f"Can't fit one sample in batch_bins ({batch_bins}): Please increase the value")
    batches = []
    for start, stop in _cut_points(in_sz, out_sz, int(batch_bins), int(min_batch_size)):
        run = items[start:stop]
        batches.append(run[::-1] if shortest_first else run)
    batches = _refill_short_tail(batches, int(min_batch_size))
    if num_batches > 0:
        batches = batches[:num_batches]
    sizes = [len(b) for b in batches]
    logging.info(f"{len(batches)} batches containing from {min(sizes)} to {max(sizes)} samples "
                 f"(avg {np.mean(sizes)} samples)")
    return batches


def _check_modes(count, batch_sort_key, swap_io, batch_bins):
    if count not in BATCH_COUNT_CHOICES:
        raise ValueError(f"arg 'count' ({count}) should be one of {BATCH_COUNT_CHOICES}")
    if batch_sort_key not in BATCH_SORT_KEY_CHOICES:
        raise ValueError(f"arg 'batch_sort_key' ({batch_sort_key}) should be one of {BATCH_SORT_KEY_CHOICES}")
    if not swap_io:
        raise NotImplementedError("swap_io=False not implemented")
    if count == "auto" and batch_bins != 0:
        count = "bin"
        logging.info(f"count is auto detected as {count}")
    if batch_sort_key == "shuffle" and count != "seq":
        raise ValueError("batch_sort_key=shuffle is only available if batch_count=seq")
    if count != "bin":
        raise NotImplementedError("count other than bin is not implemented")


def make_batchset(data, num_batches=0, min_batch_size=1, shortest_first=False, batch_sort_key="input",
                  swap_io=True, count="auto", batch_bins=0):
    _check_modes(count, batch_sort_key, swap_io, batch_bins)
    # TTS reads the json the other way round (swap_io): its "output" (text) feeds the model, its "input" (features)
    # is the target — so the model-input key is "output", and a sort "by input" sorts by the json's "output"
    model_in, model_out = "output", "input"
    length_of = model_in if batch_sort_key == "input" else model_out
    groups = {}
    for utt, entry in data.items():
        groups.setdefault(entry.get("category"), []).append((utt, entry))
    batches = []
    for members in groups.values():
        members.sort(key=lambda kv: int(kv[1][length_of][0]["shape"][0]), reverse=not shortest_first)
        batches += batchfy_by_bin(members, batch_bins=batch_bins, min_batch_size=min_batch_size,
                                  shortest_first=shortest_first, ikey=model_in, okey=model_out)
    if num_batches > 0:
        batches = batches[:num_batches]
    logging.info(f"#minibatches: {len(batches)}")
    return batches
