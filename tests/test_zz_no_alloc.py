"""The entry point that allocates nothing: fs2_colreduce_static (include/fs2_hip.h) against fs2_colreduce.

Same kernels and summation order, so the bar is bit-identity; two calls in a row, the second on a smaller
input (the static partials of the first call are still in the array), plus a torch fp64 check and the
size limit.  Collected last (file name) so that a failure here hides no other test under `pytest -x`.
"""
import pytest
import torch

from fastspeech2_amd import hip
from fastspeech2_amd.hip import call, ptr, stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(G, g_dtype, ldg, S, R, C, M, scale, accumulate=0, pe=None, pos=None, seed_out=None):
    lib = hip.load()
    outs = []
    for static in (False, True):
        out = (seed_out.clone() if seed_out is not None else torch.full((R, C), float("nan"), device=DEV))
        lds = S.shape[1] if S is not None else 0
        if static:
            call("fs2_colreduce_static", ptr(G), g_dtype, ldg, ptr(S), lds, R, ptr(pe), ptr(pos), ptr(out), M, C, scale,
                 accumulate, stream())
        else:
            ws = torch.empty((max(int(lib.fs2_colreduce_workspace(M, C, R)), 16),), dtype=torch.uint8, device=DEV)
            call("fs2_colreduce", ptr(G), g_dtype, ldg, ptr(S), lds, R, ptr(pe), ptr(pos), ptr(out), ptr(ws), M, C, scale,
                 accumulate, stream())
        outs.append(out)
    torch.cuda.synchronize()
    return outs


def test_static_scratch_colreduce_matches_workspace_colreduce():
    g = torch.Generator().manual_seed(11)
    # call 1: large (fills 5.2 MB of the 8 MiB array), call 2: smaller shape over the stale partials
    for (M, C, R) in ((8191, 256, 5), (777, 384, 3), (33, 64, 1)):
        G = torch.randn(M, C, generator=g)
        S = torch.randn(M, R, generator=g)
        Gd, Sd = G.to(DEV), S.to(DEV)
        a, b = _both(Gd, 0, C, Sd if R > 1 else None, R, C, M, 0.5)
        assert torch.equal(a, b), (M, C, R)
        exp = 0.5 * (S.double().t() @ G.double() if R > 1 else G.double().sum(0, keepdim=True))
        assert (b.cpu().double() - exp).abs().max().item() <= 1e-4 * exp.abs().max().item()
    # bf16 input with a padded leading dimension, accumulate into an existing gradient
    Gb = torch.randn(4096, 96, generator=g).to(torch.bfloat16).to(DEV)
    seed = torch.randn(1, 80, generator=g).to(DEV)
    a, b = _both(Gb, 1, 96, None, 1, 80, 4096, 1.0, accumulate=1, seed_out=seed)
    assert torch.equal(a, b)
    # positional-encoding mode (d alpha)
    pe = torch.randn(50, 256, generator=g).to(DEV)
    pos = torch.randint(0, 50, (3000,), generator=g, dtype=torch.int32).to(DEV)
    G = torch.randn(3000, 256, generator=g).to(DEV)
    a, b = _both(G, 0, 256, None, 1, 256, 3000, 1.0, pe=pe, pos=pos)
    assert torch.equal(a, b)
    # empty input: zeros
    a, b = _both(G, 0, 256, None, 1, 256, 0, 1.0)
    assert torch.equal(a, b) and float(b.abs().max()) == 0.0


def test_static_scratch_rejects_oversize():
    lib = hip.load()
    cap = int(lib.fs2_colreduce_static_max_bytes())
    assert cap == 8 << 20
    M, C, R = 1 << 20, 256, 8
    assert int(lib.fs2_colreduce_workspace(M, C, R)) > cap
    G = torch.zeros(16, device=DEV)
    out = torch.zeros((R, C), device=DEV)
    with pytest.raises(hip.HipError, match="static scratch"):
        call("fs2_colreduce_static", ptr(G), 0, C, ptr(G), R, R, None, None, ptr(out), M, C, 1.0, 0, stream())


def test_static_colreduce_allocates_nothing_and_captures():
    """After fs2_colreduce_static_init() a call is launches only: the device's free memory does not move across calls and
    a call recorded into a HIP graph replays to the eager result."""
    lib = hip.load()
    assert lib.fs2_colreduce_static_init() == 0
    g = torch.Generator().manual_seed(5)
    M, C = 4096, 256
    G = torch.randn(M, C, generator=g).to(DEV)
    out = torch.empty((1, C), device=DEV)
    want = torch.empty((1, C), device=DEV)
    ws = torch.empty((max(int(lib.fs2_colreduce_workspace(M, C, 1)), 16),), dtype=torch.uint8, device=DEV)
    call("fs2_colreduce", ptr(G), 0, C, None, 0, 1, None, None, ptr(want), ptr(ws), M, C, 1.0, 0, stream())
    call("fs2_colreduce_static", ptr(G), 0, C, None, 0, 1, None, None, ptr(out), M, C, 1.0, 0, stream())
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        call("fs2_colreduce_static", ptr(G), 0, C, None, 0, 1, None, None, ptr(out), M, C, 1.0, 0, stream())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.equal(out, want)
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call("fs2_colreduce_static", ptr(G), 0, C, None, 0, 1, None, None, ptr(out), M, C, 1.0, 0, stream())
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
