"""Diagnostic: bf16-mode mel error vs the golden fp32 reference vectors (teacher-forced durations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers
from tests.test_gpu_model import build_model

for name in helpers.INFER_CASES:
    z, meta = helpers.load_golden(name)
    if meta["duration_control"] != 1.0:
        continue
    for prec in ("fp32", "bf16"):
        model = build_model(meta, prec)
        eng = model.engine()
        for i in range(meta["n"]):
            x = torch.from_numpy(z["ids_%d" % i]).cuda()
            ds = torch.from_numpy(z["dur_after_lr_%d" % i].astype(np.int32)).cuda()
            out = eng.forward(x, [x.numel()], ds=ds)
            mel, ref = out["mel"].cpu().numpy(), z["mel_%d" % i]
            dmatch = (out["durations_pred"].cpu().numpy() == z["dur_pred_%d" % i]).mean()
            enc = np.abs(out["enc_out"].cpu().numpy() - z["enc_out_%d" % i]).max()
            print("%-22s %s utt%d T=%4d mel|max|=%.2f err max=%.3e rms=%.3e enc_err=%.2e dur_match=%.3f" % (
                name, prec, i, mel.shape[0], np.abs(ref).max(), np.abs(mel - ref).max(),
                np.sqrt(np.mean((mel - ref) ** 2)), enc, dmatch))
