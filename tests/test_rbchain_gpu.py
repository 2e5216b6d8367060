"""csrc/d3d_rbchain.hip: every strided rulebook of a stack in one chain call + rank-probe submanifold tables.
Bit-identical to the per-layer builds (d3d_conv_rulebook + d3d_dgrad_plan_build, d3d_subm_rulebook) and to the
oracle's get_indice_pairs (oracle/spconv_ref.py, SURVEY.md App. A.3 canonical order); the full-size C4 / C5 grids go
through the same path in tests/test_fullsize_geometry_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import spconv_ref as R

pytestmark = pytest.mark.gpu


def _cloud(rng, batch, shape, n_per):
    idx = []
    for b in range(batch):
        ctr = rng.integers(0, shape, size=(max(n_per // 10, 1), 3))
        pts = ctr[rng.integers(0, len(ctr), n_per * 2)] + rng.integers(-3, 4, size=(n_per * 2, 3))
        pts = np.unique(np.clip(pts, 0, np.array(shape) - 1), axis=0)
        rng.shuffle(pts)  # voxel rows arrive in first-point order, not in cell order
        pts = pts[:n_per]
        idx.append(np.concatenate([np.full((len(pts), 1), b), pts], 1))
    return np.concatenate(idx).astype(np.int32)


def _stack():
    """the layer sequence of SpMiddleFHD (scn.py:106-157): subm level, k3 s2 p1, subm, k3 s2 p1, subm, k3 s2 p[0,1,1],
    subm, (3,1,1) s(2,1,1)"""
    from det3d_amd import spconv

    S, C = spconv.SubMConv3d, spconv.SparseConv3d
    return spconv.SparseSequential(
        S(4, 16, 3, bias=False, indice_key="s0"), C(16, 32, 3, 2, padding=1, bias=False),
        S(32, 32, 3, bias=False, indice_key="s1"), C(32, 64, 3, 2, padding=1, bias=False),
        S(64, 64, 3, bias=False, indice_key="s2"), C(64, 64, 3, 2, padding=[0, 1, 1], bias=False),
        S(64, 64, 3, bias=False, indice_key="s3"), C(64, 64, (3, 1, 1), (2, 1, 1), bias=False))


def _plan(seq, idx, batch, shape, chain):
    from det3d_amd import spconv

    prev = spconv.RB_CHAIN
    spconv.RB_CHAIN = chain
    try:
        x = spconv.SparseConvTensor(torch.zeros((len(idx), 4), device="cuda"), torch.from_numpy(idx).cuda(), shape, batch)
        d = seq.plan(x)
        torch.cuda.synchronize()
    finally:
        spconv.RB_CHAIN = prev
    return d


@pytest.mark.parametrize("seed,batch,shape,n_per", [(0, 3, [41, 160, 136], 4000), (1, 1, [41, 40, 36], 700),
                                                    (2, 2, [41, 1600, 1408], 9000), (3, 4, [41, 64, 64], 1)])
def test_chain_equals_the_per_layer_builds_and_the_oracle(seed, batch, shape, n_per):
    rng = np.random.default_rng(seed)
    idx = _cloud(rng, batch, shape, n_per)
    seq = _stack()
    a = _plan(seq, idx, batch, shape, chain=True)
    b = _plan(seq, idx, batch, shape, chain=False)
    assert set(a) == set(b) and len(a) == 8
    cur, cur_shape = idx, list(shape)
    for m in seq.conv_layers():
        key = m._rulebook_key()
        ra, rb = a[key], b[key]
        assert (ra.n_in, ra.n_out, ra.kvol, ra.subm, list(ra.out_shape)) == (rb.n_in, rb.n_out, rb.kvol, rb.subm, list(rb.out_shape))
        ta = ra.nbr_out.view(ra.kvol, ra.cap_out)[:, :ra.n_out]
        tb = rb.nbr_out.view(rb.kvol, rb.cap_out)[:, :rb.n_out]
        assert torch.equal(ta, tb), (key, "output-stationary table")
        oi, oshape, pairs = R.get_indice_pairs(cur, batch, cur_shape, m.kernel_size, m.stride, m.padding, m.subm)
        ref = np.full((ra.kvol, len(oi)), -1, np.int32)
        for k, (i, o) in enumerate(pairs):
            ref[k, o] = i
        assert ra.n_out == len(oi) and np.array_equal(ta.cpu().numpy(), ref), (key, "oracle")
        if not m.subm:
            assert torch.equal(ra.out_indices, rb.out_indices) and np.array_equal(ra.out_indices.cpu().numpy(), oi)
            assert torch.equal(ra.nbr_in.view(ra.kvol, ra.cap_in)[:, :ra.n_in], rb.nbr_in.view(rb.kvol, rb.cap_in)[:, :rb.n_in])
            assert (ra.dgrad_plan is None) == (rb.dgrad_plan is None)
            if ra.dgrad_plan is not None:
                # d3d_rowplan.h: [0] classes, [1 .. 1 + C] class starts, per class c (and c = C: every offset) at 16 + 32 c:
                # count, the offsets, words 28..31 the packed list; [304 ...] the permutation (unwritten words are garbage)
                pa, pb = ra.dgrad_plan.cpu().numpy(), rb.dgrad_plan.cpu().numpy()
                ncls, n = int(pa[0]), ra.n_in
                assert np.array_equal(pa[:2 + ncls], pb[:2 + ncls]), (key, "row plan: classes and starts")
                for c in range(ncls + 1):
                    o = 16 + 32 * c
                    cnt = int(pa[o])
                    assert np.array_equal(pa[o:o + 1 + cnt], pb[o:o + 1 + cnt]) and np.array_equal(pa[o + 28:o + 32], pb[o + 28:o + 32]), (key, c)
                assert np.array_equal(pa[304:304 + n], pb[304:304 + n]), (key, "row plan permutation")
            cur, cur_shape = oi, list(oshape)


def test_chain_entry_point_and_ranked_pair_counts():
    """the raw chain entry point again on the same input (deterministic: same tables), and the ranked submanifold build
    with its in-kernel pair counters"""
    from det3d_amd import _lib, spconv

    rng = np.random.default_rng(5)
    shape, batch = [41, 80, 72], 2
    idx = _cloud(rng, batch, shape, 1500)
    seq = _stack()
    d = _plan(seq, idx, batch, shape, chain=True)
    convs = [m for m in seq.conv_layers() if not m.subm]
    # the same chain again through the raw entry point, with in-kernel pair counters on every layer
    L = _lib.lib()
    ind = torch.from_numpy(idx).cuda()
    n_dev = torch.tensor([len(idx)], dtype=torch.int32, device="cuda")
    built, ranks, ws = spconv.build_conv_rulebook_chain(ind, n_dev, len(idx), batch, shape, convs)
    torch.cuda.synchronize()
    for m, (rb, n_out_dev) in zip(convs, built):
        ref = d[m._rulebook_key()]
        n_out = int(n_out_dev.item())
        assert n_out == ref.n_out
        assert torch.equal(rb.nbr_out.view(rb.kvol, rb.cap_out)[:, :n_out], ref.nbr_out.view(ref.kvol, ref.cap_out)[:, :n_out])
    # ranked submanifold table with pair counts == the counts of the planned table
    lvl1 = d[convs[0]._rulebook_key()]
    n1 = lvl1.n_out
    nbr = torch.empty((27 * n1,), dtype=torch.int32, device="cuda")
    pc = torch.empty((27,), dtype=torch.int32, device="cuda")
    rc = L.d3d_subm_rulebook_ranked(lvl1.out_indices.data_ptr(), built[0][1].data_ptr(), n1, batch, _lib.i32_array(lvl1.out_shape),
                                    _lib.i32_array([3, 3, 3]), ranks[0][0], ranks[0][1], nbr.data_ptr(), pc.data_ptr(),
                                    _lib.stream_ptr(ind.device))
    _lib.check(rc, "d3d_subm_rulebook_ranked")
    s1 = d["s1"]
    assert torch.equal(nbr.view(27, n1), s1.nbr_out.view(27, s1.cap_out)[:, :n1])
    assert pc.tolist() == (s1.nbr_out.view(27, s1.cap_out)[:, :n1] >= 0).sum(1).tolist()
