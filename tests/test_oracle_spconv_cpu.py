"""Pins oracle/spconv_ref.py (restatement of spconv v1.x gather-GEMM-scatter, SURVEY.md
App. A) against torch.nn.functional.conv3d on the densified tensor -- the only
executable ground truth available for the un-vendored spconv dependency."""
import numpy as np
import pytest
import torch

from oracle import spconv_ref as R


def _random_sparse(rng, batch, shape, n_per, c):
    idx = []
    for b in range(batch):
        lin = rng.choice(shape[0] * shape[1] * shape[2], size=n_per, replace=False)
        z, r = lin // (shape[1] * shape[2]), lin % (shape[1] * shape[2])
        idx.append(np.stack([np.full(n_per, b), z, r // shape[2], r % shape[2]], 1))
    idx = np.concatenate(idx).astype(np.int32)
    rng.shuffle(idx)  # voxelizer order is not sorted
    feats = torch.from_numpy(rng.standard_normal((len(idx), c)).astype(np.float32))
    return idx, feats


GEOMS = [
    # ksize, stride, padding, subm
    ([3, 3, 3], [1, 1, 1], [0, 0, 0], True),
    ([3, 3, 3], [2, 2, 2], [1, 1, 1], False),
    ([3, 3, 3], [2, 2, 2], [0, 1, 1], False),
    ([3, 1, 1], [2, 1, 1], [0, 0, 0], False),
    ([1, 1, 1], [1, 1, 1], [0, 0, 0], True),
]


@pytest.mark.parametrize("ksize,stride,padding,subm", GEOMS)
def test_restatement_equals_dense_conv3d(ksize, stride, padding, subm):
    rng = np.random.default_rng(hash((tuple(ksize), tuple(stride), subm)) % 2 ** 31)
    shape = [9, 14, 12]
    idx, feats = _random_sparse(rng, 2, shape, 150, 5)
    w = torch.from_numpy(rng.standard_normal((*ksize, 5, 7)).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(7).astype(np.float32))
    out_idx, out_shape, pairs = R.get_indice_pairs(idx, 2, shape, ksize, stride, padding, subm)
    out = R.indice_conv(feats, w, pairs, len(out_idx), bias)
    ref, ref_idx = R.dense_conv_reference(feats, idx, 2, shape, w, ksize, stride, padding, subm, bias)
    assert np.array_equal(out_idx, ref_idx)  # canonical (sorted) output order / input order for subm
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)
    if not subm:
        lin = R._lin(out_idx.astype(np.int64), out_shape)
        assert np.all(np.diff(lin) > 0)


def test_subm_output_sites_equal_input_sites():
    rng = np.random.default_rng(5)
    idx, feats = _random_sparse(rng, 1, [5, 8, 8], 60, 3)
    out_idx, _, pairs = R.get_indice_pairs(idx, 1, [5, 8, 8], [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    assert np.array_equal(out_idx, idx)
    centre = pairs[13]
    assert np.array_equal(centre[0], np.arange(len(idx))) and np.array_equal(centre[1], np.arange(len(idx)))
    # symmetry used by the HIP dgrad: (j -> i at k) <=> (i -> j at 26-k)
    t = R.canonical_pairs(pairs)
    fwd = {(k, i, o) for k, i, o in t.tolist()}
    assert all((26 - k, o, i) in fwd for k, i, o in fwd)


def test_autograd_through_restatement_matches_dense():
    rng = np.random.default_rng(11)
    shape = [6, 10, 10]
    idx, feats = _random_sparse(rng, 1, shape, 120, 4)
    feats.requires_grad_(True)
    w = torch.from_numpy(rng.standard_normal((3, 3, 3, 4, 6)).astype(np.float32)).requires_grad_(True)
    out_idx, _, pairs = R.get_indice_pairs(idx, 1, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    g = torch.from_numpy(rng.standard_normal((len(out_idx), 6)).astype(np.float32))
    (R.indice_conv(feats, w, pairs, len(out_idx)) * g).sum().backward()
    gf, gw = feats.grad.clone(), w.grad.clone()
    feats.grad = None
    w.grad = None
    ref, _ = R.dense_conv_reference(feats, idx, 1, shape, w, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    (ref * g).sum().backward()
    assert torch.allclose(gf, feats.grad, atol=1e-4) and torch.allclose(gw, w.grad, atol=1e-4)


def test_dense_layout():
    idx = np.array([[0, 1, 2, 3], [1, 0, 0, 0]], np.int32)
    f = torch.tensor([[1., 2.], [3., 4.]])
    d = R.dense(f, idx, 2, [2, 3, 4])
    assert d.shape == (2, 2, 2, 3, 4)
    assert d[0, :, 1, 2, 3].tolist() == [1., 2.] and d[1, :, 0, 0, 0].tolist() == [3., 4.]
    assert float(d.abs().sum()) == 10.0
