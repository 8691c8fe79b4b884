"""Length-sorted minibatch construction by a "bins" budget — the counterpart of run/batchfy.py:6-123
(`batchfy_by_bin`, `make_batchset`), same arguments, same data format (the ESPnet-style json dict
{utt_id: {"input": [{"shape": [T, D]}], "output": [{"shape": [T, D]}], "category": ...}}) and the same
resulting batches, so a training script can swap the import.  Only the modes the reference implements exist:
count "bin" (or "auto" with batch_bins != 0), swap_io=True, sort key input/output."""
import itertools
import logging

import numpy as np

BATCH_COUNT_CHOICES = ["auto", "seq", "bin", "frame"]
BATCH_SORT_KEY_CHOICES = ["input", "output", "shuffle"]


def _shape(entry, key):
    return entry[1][key][0]["shape"]


def batchfy_by_bin(sorted_data, batch_bins, num_batches=0, min_batch_size=1, shortest_first=False,
                   ikey="input", okey="output"):
    """Greedy packing of the (already sorted) utterances: a batch grows while
    (longest output so far + this input) * batch size <= batch_bins, sizes counted in elements (frames * dim)."""
    if batch_bins <= 0:
        raise ValueError(f"invalid batch_bins: {batch_bins}")
    total = len(sorted_data)
    idim = int(_shape(sorted_data[0], ikey)[1])
    odim = int(_shape(sorted_data[0], okey)[1])
    logging.info(f"#utts: {total}")
    batches = []
    start = 0
    while True:
        count, cost, longest_out = 0, 0, 0
        while cost < batch_bins and start + count < total:
            item = sorted_data[start + count]
            in_size = int(_shape(item, ikey)[0]) * idim
            longest_out = max(longest_out, int(_shape(item, okey)[0]) * odim)
            cost = (longest_out + in_size) * (count + 1)
            if cost <= batch_bins:
                count += 1
            elif cost == 0:
                raise ValueError(f"Can't fit one sample in batch_bins ({batch_bins}): Please increase the value")
        stop = min(total, start + max(min_batch_size, count))
        batch = sorted_data[start:stop]
        if shortest_first:
            batch.reverse()
        batches.append(batch)
        # top up a short last batch from its predecessors (or merge it away when it is the only other one)
        k = -1
        while len(batches[k]) < min_batch_size:
            lack = min_batch_size - len(batches[k])
            if -k == len(batches):
                batches[k + 1].extend(batches[k])
                batches = batches[1:]
                break
            batches[k].extend(batches[k - 1][:lack])
            batches[k - 1] = batches[k - 1][lack:]
            k -= 1
        if stop == total:
            break
        start = stop
    if num_batches > 0:
        batches = batches[:num_batches]
    sizes = [len(b) for b in batches]
    logging.info(f"{len(batches)} batches containing from {min(sizes)} to {max(sizes)} samples "
                 f"(avg {np.mean(sizes)} samples)")
    return batches


def make_batchset(data, num_batches=0, min_batch_size=1, shortest_first=False, batch_sort_key="input",
                  swap_io=True, count="auto", batch_bins=0):
    if count not in BATCH_COUNT_CHOICES:
        raise ValueError(f"arg 'count' ({count}) should be one of {BATCH_COUNT_CHOICES}")
    if batch_sort_key not in BATCH_SORT_KEY_CHOICES:
        raise ValueError(f"arg 'batch_sort_key' ({batch_sort_key}) should be one of {BATCH_SORT_KEY_CHOICES}")
    if not swap_io:
        raise NotImplementedError("swap_io=False not implemented")
    # TTS: the json's "output" (text) is the model input and vice versa
    ikey, okey = "output", "input"
    sort_key = {"input": "output", "output": "input"}.get(batch_sort_key, batch_sort_key)
    if count == "auto":
        if batch_bins == 0:
            raise NotImplementedError("count other than bin is not implemented")
        count = "bin"
        logging.info(f"count is auto detected as {count}")
    if count != "seq" and sort_key == "shuffle":
        raise ValueError("batch_sort_key=shuffle is only available if batch_count=seq")
    if count != "bin":
        raise NotImplementedError("count other than bin is not implemented")
    by_category = {}
    for utt, entry in data.items():
        by_category.setdefault(entry.get("category"), {})[utt] = entry
    per_category = []
    for group in by_category.values():
        ordered = sorted(group.items(), key=lambda kv: int(kv[1][sort_key][0]["shape"][0]), reverse=not shortest_first)
        per_category.append(batchfy_by_bin(ordered, batch_bins=batch_bins, min_batch_size=min_batch_size,
                                           shortest_first=shortest_first, ikey=ikey, okey=okey))
    batches = per_category[0] if len(per_category) == 1 else list(itertools.chain(*per_category))
    if num_batches > 0:
        batches = batches[:num_batches]
    logging.info(f"#minibatches: {len(batches)}")
    return batches
