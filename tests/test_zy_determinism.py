"""Run-to-run determinism at BASELINE sizes.  No kernel here uses atomics and every reduction has a fixed order, so a
repeat of the same launches must give the same bits; a difference is a race (an LDS tile read before its LDS-DMA has
landed, a stream ordered wrongly), not arithmetic.  This is how the unwaited tile read of attention32_kernel showed
(tools/determinism_check.py is the longer, 256-utterance form of the inference half;
tools/train_determinism_check.py the training half with the names of the tensors that differ).  Collected late on
purpose (pytest -x): a timing-dependent failure here must not hide the parity tests."""
import numpy as np
import pytest
import torch

from fastspeech2_amd import synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from tests import helpers

pytestmark = pytest.mark.gpu


def _bench_model():
    kw = dict(synth.BASELINE_OVERRIDES)
    model = FeedForwardTransformer(80, 80, use_scaled_pos_enc=True, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.bench_state_dict().items()})
    return model.eval().cuda()


@pytest.mark.parametrize("precision,encoder", [("bf16", "f16x3"), ("bf16", None), ("f16x3", None), ("fp32", None)])
def test_inference_repeats_are_bit_identical(precision, encoder):
    """BASELINE config 2 (32 utterances, ~25 k frames) through the throughput loop (side-stream phone-level halves,
    three frame-level streams): 4 batches x 3 repeats, every output tensor equal bit for bit."""
    model = _bench_model().set_precision(precision, encoder)
    eng = model.engine()
    batches = []
    for k in range(4):
        xs = synth.synth_utterances(32, k)
        batches.append((torch.from_numpy(np.concatenate(xs)).cuda(), [len(x) for x in xs]))

    def run():
        outs = []
        for o in eng.forward_pipelined(batches, frame_streams=3, phone_group=2):
            o["ready"].synchronize()
            outs.append({k: o[k].clone() for k in ("mel", "durations", "log_durations", "pitch", "energy", "lr_index")})
        return outs

    ref = run()
    for rep in range(2):
        for a, b in zip(ref, run()):
            for k in a:
                assert torch.equal(a[k], b[k]), (rep, k)


def test_training_step_repeats_are_bit_identical():
    """BASELINE config 5 (B = 32, 4+4 blocks, d = 256, conv k = 9), bf16, dropout off: loss and all gradients of three
    runs of the same step, bit for bit (every gradient has one writer and a fixed summation order)."""
    z, meta = helpers.load_golden(helpers.BIG_TRAIN_CASES[0])
    batch = [torch.from_numpy(a) for a in helpers.train_inputs(meta["cfg"], meta["seed"], **meta["inputs"])]
    from tests.test_gpu_train import _model
    runs = []
    for _ in range(3):
        model = _model(meta, "bf16")
        model.dropout_enabled = False
        loss = model(*batch)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
        del model
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0])
        for k in runs[0][1]:
            assert torch.equal(runs[0][1][k], other[1][k]), k


def test_layernorm_bwd_beside_bf16_conv_wgrad_is_bit_stable():
    """Reduced tools/lnb_race.py: fs2_layernorm_bwd on one stream while fs2_conv_wgrad (bf16: its MFMAs are what the
    effect needs) keeps the chip busy on another, same inputs every launch - every output must equal the first launch's
    bit for bit.  A build with hipcc's SLP vectoriser on (fastspeech2_amd/build.py turns it off: DESIGN.md section 4.3)
    gives 15-30 differing outputs in 400 launches here; the static side of the guard is tools/isa_check.py."""
    import ctypes
    from fastspeech2_amd import hip
    from fastspeech2_amd.hip import call, ptr
    dev, M, C, N = "cuda:0", 26976, 256, 768
    g = torch.Generator(device=dev).manual_seed(0)
    x, dy = torch.randn(M, C, device=dev, generator=g), torch.randn(M, C, device=dev, generator=g)
    parts = torch.randn(1, M, C, device=dev, generator=g)
    w = torch.randn(C, device=dev, generator=g)
    lib = hip.load()
    ws = torch.empty(lib.fs2_layernorm_bwd_workspace(M, C) // 4 + 4, device=dev)
    gy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    gx = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    gw, gb = torch.empty(N, C, device=dev), torch.empty(N, device=dev)
    ks = 16
    wws = torch.empty(lib.fs2_conv_wgrad_workspace(N, C, 1, ks) // 4 + 4, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    first, differing = None, 0
    outs = []
    torch.cuda.synchronize()
    for it in range(300):
        for _ in range(3):
            call("fs2_conv_wgrad", ptr(gy), ptr(gx), ptr(gw), ptr(gb), ptr(wws), None, M, N, C, 1, N, C, 1, ks, 1.0, 0,
                 ctypes.c_void_p(sb.cuda_stream))
        of = torch.empty(M, C, device=dev)
        ot = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(2 * C, device=dev)
        call("fs2_layernorm_bwd", ptr(x), ptr(dy), ptr(parts), 1, M * C, ptr(w), None, 0, ptr(of), ptr(ot), ptr(dw),
             ctypes.c_void_p(dw.data_ptr() + 4 * C), ptr(ws), M, C, 1e-12, 0, 1, 7, 1, 0.0, 2, 0.0, ctypes.c_void_p(sa.cuda_stream))
        outs.append((of, ot, dw))
        if len(outs) == 60:                # compare in batches: bounded memory, launches stay back to back in between
            torch.cuda.synchronize()
            if first is None:
                first = outs[0]
            differing += sum(1 for o in outs for a, b in zip(o, first) if not torch.equal(a, b))
            outs = []
    assert differing == 0, "%d outputs of 300 launches differ from the first launch's" % differing
