"""The MFMA weight stream produced by the C ABI, replayed on the CPU with the documented MFMA lane maps
(tests/mfma_emulator.py) and compared with the oracle's MLP.  Catches any mismatch between the packer
(nwe_abi.hip) and the kernel's consumption order without a GPU."""
import numpy as np
import pytest
import torch

import nwe_amd
from oracle import nerf_oracle as O
from tests import mfma_emulator as E


@pytest.mark.parametrize("D,W,seed,folded,n_tiles,n_chunks", [
    (8, 256, 1001, True, 2080, 69 + 9), (8, 256, 1001, False, 2368, 78), (4, 128, 1000, True, 272, 19 + 5), (4, 128, 1000, False, 352, 24),
    (6, 256, 7, True, 1568, 53 + 9), (6, 128, 8, True, 432, 27 + 5), (8, 128, 9, True, 560, 35 + 5), (4, 256, 10, True, 992, 37 + 9)])
def test_stream_replay_matches_oracle(D, W, seed, folded, n_tiles, n_chunks):
    """n_chunks = rows of the bias table: one per chunk of the stream, and for the folded stream (which has no alpha tile)
    the W/32 + 1 dot rows of _alpha_linear behind them."""
    sd = nwe_amd.synthetic.make_state_dict(seed, D, W)
    r = nwe_amd.Renderer(host_only=True)
    r.debug_set_fold(folded)
    shape = r.set_network(0, sd)
    stream, bias, scale = r.packed_stream(0), r.packed_bias(0), r.packed_scale(0)
    assert stream.size == n_tiles * 1024 and bias.shape == (n_chunks, 32)
    packed = {k: v for k, v in sd.items() if k.endswith("weight")}
    if folded:   # _feature_linear is multiplied into the view layer (nerf_model.py:64-70): the product is what gets packed
        wv = packed.pop("_views_linears.0.weight").astype(np.float64)
        wf = packed.pop("_feature_linear.weight").astype(np.float64)
        packed["folded"] = np.concatenate([wv[:, :W] @ wf, wv[:, W:]], 1)
        packed.pop("_alpha_linear.weight")       # evaluated in fp32 from the dot rows, not part of the fp16 stream
    wmax = max(np.abs(v).max() for v in packed.values())
    assert 2.0 ** 13 <= wmax * scale <= 2.0 ** 14 and np.log2(scale) == round(np.log2(scale))   # power of two, fp16 headroom
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(32, 3, generator=g) * 2 - 1) * torch.tensor([8.0, 3.0, 1.0])
    dirs = torch.nn.functional.normalize(torch.randn(32, 3, generator=g), dim=-1)
    x = torch.cat([O.embed(pts, 10, 10), O.embed(dirs, 4, 1)], -1)
    ref = O.mlp_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x).numpy()
    got = E.mlp_eval(stream, bias, scale, (pts / 10).numpy(), dirs.numpy(), D, W, shape[4], three_pass=True, folded=folded)
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    # single-pass fp16 is visibly worse but still close: the split is what buys fp32-grade results
    got1 = E.mlp_eval(stream, bias, scale, (pts / 10).numpy(), dirs.numpy(), D, W, shape[4], three_pass=False, folded=folded)
    assert 2e-5 < np.abs(got1 - ref).max() < 5e-2


@pytest.mark.parametrize("D,W,seed", [(8, 256, 21), (4, 128, 22), (6, 256, 23), (4, 256, 24), (8, 128, 25), (6, 128, 26)])
def test_stream_replay_without_view_dirs(D, W, seed):
    """use_view_dirs=False (nerf_model.py:42-43,78-79): layer 0, D-1 trunk layers and ONE chunk of _output_linear - rows 0..3,
    the fifth channel is dropped as the reference drops it (model_utils.py:62,71)."""
    sd = nwe_amd.synthetic.make_state_dict(seed, D, W, use_view_dirs=False)
    r = nwe_amd.Renderer(host_only=True)
    shape = r.set_network(0, sd)
    assert r.mfma_supported(0)
    stream, bias, scale = r.packed_stream(0), r.packed_bias(0), r.packed_scale(0)
    NT, KH = W // 32, W // 16
    skip_tiles = NT * 2 * 4 if D > 5 else 0                                 # the skip layer's four gamma(x) k-steps
    assert stream.size == (NT * 2 * 4 + (D - 1) * NT * 2 * KH + skip_tiles + 2 * KH) * 1024
    assert bias.shape == (D * NT + 1, 32)
    wo, bo = sd["_output_linear.weight"], sd["_output_linear.bias"]
    assert np.array_equal(bias[-1, :4], bo[:4]) and np.array_equal(bias[-1, 4:8], bo[:4]) and not bias[-1, 8:].any()
    # the ignored fifth row must not shrink the scale: make it huge and pack again
    big = dict(sd)
    big["_output_linear.weight"] = wo.copy()
    big["_output_linear.weight"][4] *= 1e4
    r2 = nwe_amd.Renderer(host_only=True)
    r2.set_network(0, big)
    assert r2.packed_scale(0) == scale and np.array_equal(r2.packed_stream(0), stream)
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(32, 3, generator=g) * 2 - 1) * torch.tensor([8.0, 3.0, 1.0])
    ref = O.mlp_forward({k: torch.from_numpy(v) for k, v in sd.items()}, O.embed(pts, 10, 10)).numpy()[:, :4]
    got = E.mlp_eval(stream, bias, scale, (pts / 10).numpy(), None, D, W, shape[4], three_pass=True, no_view_dirs=True)
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    got1 = E.mlp_eval(stream, bias, scale, (pts / 10).numpy(), None, D, W, shape[4], three_pass=False, no_view_dirs=True)
    assert 2e-5 < np.abs(got1 - ref).max() < 5e-2


def test_no_view_dirs_unsupported_shape_has_no_stream():
    r = nwe_amd.Renderer(host_only=True)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(5, 6, 64, use_view_dirs=False))
    assert r.packed_stream(0).size == 0 and not r.mfma_supported(0)      # the fp32 kernel serves it


def test_unfolded_stream_only_for_the_baseline_shapes():
    r = nwe_amd.Renderer(host_only=True)
    r.debug_set_fold(False)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(5, 6, 256))
    assert r.packed_stream(0).size == 0
    r.debug_set_fold(True)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(5, 6, 256))
    assert r.packed_stream(0).size == 1568 * 1024


def test_unsupported_shape_has_no_stream():
    r = nwe_amd.Renderer(host_only=True)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(5, 6, 64))
    assert r.packed_stream(0).size == 0      # only the fp32 kernel serves this shape
    assert r.flops_per_eval(0) == 2 * (63 * 64 + 4 * 64 * 64 + (64 + 63) * 64 + 64 + 64 * 64 + (64 + 27) * 32 + 32 * 3)
