"""Per-parameter gradient error of the BASELINE-size training step against the reference golden (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers
from tests.test_gpu_train import _model
z, meta = helpers.load_golden("train_baseline_b32")
batch = [torch.from_numpy(a) for a in helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])]
model = _model(meta)
loss = model(*batch); loss.backward()
print("loss", loss.item(), meta["loss"])
rows = []
for k, p in model.named_parameters():
    want, got, ws, gs = helpers.golden_grad_sample(z, meta, k, p.grad)
    rows.append((np.abs(want - got).max() / max(np.abs(want).max(), 1e-2), k, np.abs(want).max(), abs(ws[1]-gs[1])/max(ws[1],1e-2)))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.2e %-60s max|g| %.3e  norm rel %.2e" % r)
print("median", np.median([r[0] for r in rows]))
