"""Which launch of the config-5 backward first differs between repeats?  Wraps TrainEngine._gemm / _ln_bwd and keeps a
copy of every output; prints the first differing call of each repeat (index, kind, shape, how many elements)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
from tests.test_gpu_train import _model
from fastspeech2_amd import train_engine as te
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
z, meta = helpers.load_golden(helpers.BIG_TRAIN_CASES[0])
batch = [torch.from_numpy(a) for a in helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])]
log = []
og, ol = te.TrainEngine._gemm, te.TrainEngine._ln_bwd
def gemm(self, *a, **k):
    out = og(self, *a, **k)
    log.append(("gemm tag=%s M=%s N=%s K=%s taps=%s ks=%s" % (k.get("tag"), a[3], a[4], a[5], a[6], k.get("ksplit", 1)), out.clone()))
    return out
def lnb(self, *a, **k):
    of, ot = ol(self, *a, **k)
    log.append(("ln_bwd f32 M=%s C=%s" % (a[5], a[6]), of.clone() if of is not None else None))
    log.append(("ln_bwd T", ot.clone() if ot is not None else None))
    return of, ot
what = os.environ.get("WHAT", "both")
if what in ("both", "gemm"):
    te.TrainEngine._gemm = gemm
if what in ("both", "ln"):
    te.TrainEngine._ln_bwd = lnb
ref = None
for rep in range(n):
    log.clear()
    model = _model(meta, "bf16")
    model.dropout_enabled = False
    model(*batch).backward()
    torch.cuda.synchronize()
    cur = list(log)
    del model
    if ref is None:
        ref = cur
        print("calls per step:", len(ref))
        continue
    first = None
    for i, ((na, a), (nb, b)) in enumerate(zip(ref, cur)):
        if a is None:
            continue
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            bad = (d > 0)
            rows = bad.view(-1, bad.shape[-1]).any(1).nonzero().flatten()
            cols = bad.view(-1, bad.shape[-1]).any(0).nonzero().flatten()
            print("rep %d: first differing call %d: %s shape %s: %d elements, %d rows (%s..), cols %s..%s, max %.3g; previous call: %s" %
                  (rep, i, na, tuple(a.shape), int(bad.sum()), rows.numel(), rows[:6].tolist(), int(cols.min()), int(cols.max()),
                   float(d.max()), ref[i - 1][0]), flush=True)
            first = i
            break
    if first is None:
        print("rep %d: identical" % rep, flush=True)
