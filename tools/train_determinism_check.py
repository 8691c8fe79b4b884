"""Run-to-run determinism of the config-5 training step (B = 32, BASELINE dims, bf16, dropout off by default): loss and
every gradient of N fresh runs of the same step must be bit-identical; prints the tensors that differ.
    python tools/train_determinism_check.py [repeats] [dropout 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
from tests.test_gpu_train import _model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
drop = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
z, meta = helpers.load_golden(helpers.BIG_TRAIN_CASES[0])
batch = [torch.from_numpy(a) for a in helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])]
ref, bad = None, 0
for rep in range(n):
    model = _model(meta, "bf16")
    model.dropout_enabled = drop
    loss = model(*batch)
    loss.backward()
    torch.cuda.synchronize()
    cur = {"loss": loss.detach().clone(), **{k: p.grad.clone() for k, p in model.named_parameters()}}
    del model
    if ref is None:
        ref = cur
        continue
    for k in ref:
        if not torch.equal(ref[k], cur[k]):
            d = (ref[k].float() - cur[k].float()).abs()
            print("rep %d: %s differs: %d of %d elements, max |d| %.3g (max |g| %.3g)" %
                  (rep, k, int((d > 0).sum()), d.numel(), float(d.max()), float(ref[k].abs().max())), flush=True)
            bad += 1
print("repeats %d, differing tensors %d" % (n, bad))
