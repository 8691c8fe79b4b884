"""Feasibility study (CPU, oracle only — not product code): how far do error-compensated bf16 GEMMs move the
log-durations of the phone-level half?  F.linear / F.conv1d of the oracle are replaced by sums of bf16 x bf16
products of split operands (x = x0 + x1 [+ x2], each term a bf16), accumulated in fp64:
  x3 : x0w0 + x0w1 + x1w0            (2-term split, 3 MFMAs per fragment pair)
  x6 : terms with i + j <= 2 of a 3-term split (6 MFMAs)
Attention's QK^T / PV stay fp32 (the fp32 attention kernel is cheap at phone level)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from fastspeech2_amd import synth
from oracle import fs2_oracle

def split(t, n):
    out, r = [], t.float()
    for _ in range(n):
        h = r.to(torch.bfloat16).float()
        out.append(h.double()); r = r - h
    return out

def make(nterm, maxsum):
    def lin(x, w, b=None):
        xs, ws = split(x, nterm), split(w, nterm)
        acc = 0
        for i in range(nterm):
            for j in range(nterm):
                if i + j <= maxsum:
                    acc = acc + xs[i] @ ws[j].t()
        if b is not None: acc = acc + b.double()
        return acc.float()
    def conv(x, w, b=None, padding=0):
        xs, ws = split(x, nterm), split(w, nterm)
        acc = 0
        for i in range(nterm):
            for j in range(nterm):
                if i + j <= maxsum:
                    acc = acc + real_conv(xs[i], ws[j], None, padding=padding)
        if b is not None: acc = acc + b.double().view(1, -1, 1)
        return acc.float()
    return lin, conv

kw = dict(synth.BASELINE_OVERRIDES)
sd = {k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()}
cfg = fs2_oracle.full_cfg(80, 80, use_scaled_pos_enc=True, **kw)
xs = [torch.from_numpy(x) for x in synth.synth_utterances(32, 0)]
torch.set_num_threads(8)
def run():
    logd = []
    for x in xs:
        hs = fs2_oracle.encode(sd, cfg, x)
        logd.append(fs2_oracle.duration_log(hs, sd))
    return torch.cat(logd)
ref = run()
dref = fs2_oracle.duration_round(ref)
frac = (ref.exp() - 1.0).numpy(); marg = np.abs(frac - np.floor(frac) - 0.5)
print("phones", ref.numel(), "min margin %.2e" % marg.min(), "phones within 1e-4/1e-3 of an edge:", (marg < 1e-4).sum(), (marg < 1e-3).sum())
real_lin, real_conv = F.linear, F.conv1d
for name, (n, ms) in dict(x1=(1, 0), x3=(2, 1), x6=(3, 2), exact64=(3, 4)).items():
    lin, conv = make(n, ms)
    fs2_oracle.F.linear, fs2_oracle.F.conv1d = lin, conv
    # attention matmuls inside mha use torch.matmul (fp32): unchanged
    got = run()
    fs2_oracle.F.linear, fs2_oracle.F.conv1d = real_lin, real_conv
    d = fs2_oracle.duration_round(got)
    print("%-8s max|dlogd| %.3e  rms %.3e  flipped durations %d / %d" % (name, (got - ref).abs().max(), (got - ref).pow(2).mean().sqrt(), int((d != dref).sum()), d.numel()))
