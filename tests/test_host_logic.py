"""CPU-only checks of the host side: the drop-in boundary (constructor, state_dict layout, config
loading), the C-ABI library (loads, exports every symbol of include/fs2_hip.h — no compute calls),
and that the product path refuses to run without the HIP device instead of falling back."""
import os
import re

import pytest
import torch

from fastspeech2_amd import hip, synth
from fastspeech2_amd.network.fastspeech2 import FeedForwardTransformer
from fastspeech2_amd.network.utils import make_non_pad_mask, make_pad_mask, pad_list
from fastspeech2_amd.param.load_param import load_config, to_namespace_recursive
from oracle import ref_harness
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fs2_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fs2_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    lib = hip.load()
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert sorted(hip.EXPORTS) == declared
    assert lib.fs2_abi_version() == 1


def test_yaml_loads_and_constructs():
    cfg = load_config()
    assert cfg["adim"] == 384 and cfg["postnet_layers"] == 5 and cfg["use_scaled_pos_enc"] is True
    ns = to_namespace_recursive({"a": {"b": 1}, "c": 2})
    assert ns.a.b == 1 and ns.c == 2
    model = FeedForwardTransformer(80, 80, **cfg)          # shipped YAML constructs (D1 fixed, PostNet repaired)
    assert model.encoder.embed[-1].alpha.item() == 1.0
    assert len(model.postnet.postnet) == 5 and model.postnet.postnet[0][0].in_channels == 80
    assert model.postnet.postnet[4][0].out_channels == 80


@pytest.mark.parametrize("name", ["tiny_conv9_plain", "tiny_linear_scaled", "tiny_conv3_mixed_ln", "baseline_4x4",
                                  "postnet_tiny_bn", "postnet_tiny_nobn", "yaml_shipped_full"])
def test_state_dict_layout_matches_spec(name):
    _, meta = helpers.load_golden(name)
    idim, odim, kw = helpers.split_cfg(meta["cfg"])
    model = FeedForwardTransformer(idim, odim, **kw)
    sd = model.state_dict()
    spec = synth.state_dict_spec(idim, odim, **kw)
    assert sorted(sd.keys()) == sorted(k for k, _, _ in spec)
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
    model.load_state_dict(helpers.golden_state_dict(meta), strict=True)


def test_constructor_errors_follow_reference():
    with pytest.raises(NotImplementedError):
        FeedForwardTransformer(10, 10, positionwise_layer_type="conv2d", postnet_layers=0)
    with pytest.raises(AssertionError):
        FeedForwardTransformer(10, 10, adim=30, aheads=4, postnet_layers=0)
    with pytest.raises(ValueError):
        make_pad_mask([1, 2], torch.zeros(2, 2), length_dim=0)


def test_forward_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = FeedForwardTransformer(10, 8, adim=64, aheads=2, elayers=1, dlayers=1, eunits=64, dunits=64,
                                   postnet_layers=0).eval()
    with pytest.raises(hip.HipError):
        model.inference(torch.tensor([1, 2, 3]), None)


def test_mask_helpers():
    m = make_non_pad_mask([3, 1])
    assert m.tolist() == [[True, True, True], [True, False, False]]
    xs = torch.zeros(2, 4, 5)
    assert make_pad_mask([4, 2], xs, 1)[1, :, 0].tolist() == [False, False, True, True]
    p = pad_list([torch.ones(2, 3), torch.ones(4, 3)], 0.0)
    assert p.shape == (2, 4, 3) and p[0, 2:].sum() == 0


def test_synth_utterances_are_bucketed():
    xs = synth.synth_utterances(32, 0)
    lens = [len(x) for x in xs]
    assert lens == sorted(lens, reverse=True) and 60 <= min(lens) and max(lens) <= 180
    assert all(x.min() >= 1 and x.max() < 80 for x in xs)


@pytest.mark.reference
@pytest.mark.parametrize("name", ["tiny_conv3_mixed_ln", "tiny_linear_scaled"])
def test_same_seed_same_init_as_reference(name):
    """Parameter holders are created in the reference's order, so torch.manual_seed(s) yields
    bit-identical random-init weights and identical state_dict keys (repairs D1/D2)."""
    _, meta = helpers.load_golden(name)
    idim, odim, kw = helpers.split_cfg(meta["cfg"])
    torch.manual_seed(5)
    ours = FeedForwardTransformer(idim, odim, **kw).state_dict()
    ref = ref_harness.build_reference_model(idim, odim, seed=5, **kw).state_dict()
    assert sorted(ours.keys()) == sorted(ref.keys())
    for k in ref:
        assert torch.equal(ours[k], ref[k]), k


def test_training_on_cpu_fails_loudly_with_a_postnet_too():
    """param/fastspeech.yaml has postnet_layers: 5; its training step exists (tests/test_gpu_train.py) and, like every
    other path, has no CPU implementation: a CPU model must say so instead of handing back a loss without a backward."""
    from fastspeech2_amd.hip import HipError
    m = FeedForwardTransformer(10, 8, adim=32, aheads=2, elayers=1, dlayers=1, eunits=32, dunits=32, postnet_layers=2,
                               postnet_chans=8)
    m.train()
    z = torch.zeros
    with pytest.raises(HipError):
        m(z(1, 3, dtype=torch.long), torch.tensor([3]), z(1, 3, 8), torch.tensor([3]), torch.ones(1, 3, dtype=torch.long),
          z(1, 3, 4), z(1, 3, 1))
