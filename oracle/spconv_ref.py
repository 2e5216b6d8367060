"""TEST INFRASTRUCTURE -- CPU oracle for the sparse-conv rows.

PARITY UNPINNED at the source: the arithmetic lives in `spconv`, an un-vendored and
un-pinned third-party dependency of Det3D (INSTALLATION.md:10,13,26-32; call sites
det3d/models/backbones/scn.py:22-43,46-89,106-157,187-195,323-355).  This file
restates spconv v1.x's published algorithm:

  get_indice_pairs : App. A.1/A.3 -- rulebook per kernel offset, canonical order
                     (outputs ascending in ((b*D+z)*H+y)*W+x, pairs by (k, in_row))
  indice_conv      : App. A.2 -- per offset: gather rows, `mm` with W[k], scatter-add
  dense()          : scatter to [B, C, D, H, W]

and pins it the only way available: `dense_conv_reference` computes the same layer
with torch.nn.functional.conv3d on the densified tensor (strided conv == dense conv at
active outputs; submanifold == dense conv, pad k//2, masked to the input sites).
tests/test_oracle_cpu.py checks restatement == dense conv3d; the HIP kernels are then
checked against the restatement (full frames) and the dense conv (cropped grids).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _lin(idx, shape):
    idx = idx.astype(np.int64)
    return ((idx[:, 0] * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


def conv_out_shape(in_shape, ksize, stride, padding):
    return [(in_shape[j] + 2 * padding[j] - ksize[j]) // stride[j] + 1 for j in range(3)]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, subm):
    """indices: int array [N,4] (b,z,y,x).  Returns (out_indices [M,4] int32, out_shape,
    pairs) with pairs = list over kernel offsets k=(kz*kH+ky)*kW+kx of (in_rows, out_rows)
    int64 arrays, in_rows ascending."""
    indices = np.asarray(indices).astype(np.int64)
    ksize, stride, padding = list(ksize), list(stride), list(padding)
    kvol = ksize[0] * ksize[1] * ksize[2]
    if subm:
        out_shape = list(spatial_shape)
        lin_in = _lin(indices, spatial_shape)
        order = np.argsort(lin_in, kind="stable")
        sorted_lin = lin_in[order]
        pairs = []
        for k in range(kvol):
            kz, ky, kx = k // (ksize[1] * ksize[2]), (k // ksize[2]) % ksize[1], k % ksize[2]
            # input = output + (k - k//2): for every OUTPUT site o find its input neighbour
            nb = indices.copy()
            nb[:, 1] += kz - ksize[0] // 2
            nb[:, 2] += ky - ksize[1] // 2
            nb[:, 3] += kx - ksize[2] // 2
            ok = np.ones(len(nb), dtype=bool)
            for j in range(3):
                ok &= (nb[:, 1 + j] >= 0) & (nb[:, 1 + j] < spatial_shape[j])
            out_rows = np.nonzero(ok)[0]
            lin = _lin(nb[ok], spatial_shape)
            pos = np.searchsorted(sorted_lin, lin)
            pos_c = np.minimum(pos, len(sorted_lin) - 1) if len(sorted_lin) else pos
            hit = (pos < len(sorted_lin)) & (sorted_lin[pos_c] == lin) if len(sorted_lin) else np.zeros(0, bool)
            in_rows = order[pos_c[hit]]
            out_rows = out_rows[hit]
            o2 = np.argsort(in_rows, kind="stable")
            pairs.append((in_rows[o2], out_rows[o2]))
        return indices.astype(np.int32), out_shape, pairs

    out_shape = conv_out_shape(spatial_shape, ksize, stride, padding)
    cand = []
    per_k = []
    for k in range(kvol):
        kk = [k // (ksize[1] * ksize[2]), (k // ksize[2]) % ksize[1], k % ksize[2]]
        t = indices[:, 1:] + np.array(padding) - np.array(kk)
        ok = np.ones(len(indices), dtype=bool)
        q = np.zeros_like(t)
        for j in range(3):
            q[:, j] = t[:, j] // stride[j]
            ok &= (t[:, j] >= 0) & (t[:, j] % stride[j] == 0) & (q[:, j] < out_shape[j])
        in_rows = np.nonzero(ok)[0]
        oc = np.concatenate([indices[ok, :1], q[ok]], axis=1)
        lin = _lin(oc, out_shape)
        per_k.append((in_rows, lin))
        cand.append(lin)
    all_lin = np.unique(np.concatenate(cand)) if cand else np.zeros(0, np.int64)
    # decode the sorted unique cells
    x = all_lin % out_shape[2]
    r = all_lin // out_shape[2]
    y = r % out_shape[1]
    r = r // out_shape[1]
    z = r % out_shape[0]
    b = r // out_shape[0]
    out_indices = np.stack([b, z, y, x], axis=1).astype(np.int32)
    pairs = []
    for in_rows, lin in per_k:
        out_rows = np.searchsorted(all_lin, lin)
        pairs.append((in_rows.astype(np.int64), out_rows.astype(np.int64)))
    return out_indices, out_shape, pairs


def canonical_pairs(pairs):
    """sorted (k, in_row, out_row) triples."""
    rows = []
    for k, (i, o) in enumerate(pairs):
        if len(i):
            rows.append(np.stack([np.full(len(i), k, np.int64), i, o], axis=1))
    if not rows:
        return np.zeros((0, 3), np.int64)
    t = np.concatenate(rows)
    order = np.lexsort((t[:, 2], t[:, 1], t[:, 0]))
    return t[order]


def indice_conv(features, weight, pairs, n_out, bias=None):
    """features [N,Cin] torch (cpu); weight [kD,kH,kW,Cin,Cout]; autograd-capable
    gather -> mm -> scatter-add (App. A.2)."""
    kvol = len(pairs)
    cin, cout = weight.shape[-2], weight.shape[-1]
    w = weight.reshape(kvol, cin, cout)
    out = features.new_zeros((n_out, cout))
    for k, (i, o) in enumerate(pairs):
        if len(i) == 0:
            continue
        it = torch.from_numpy(np.ascontiguousarray(i))
        ot = torch.from_numpy(np.ascontiguousarray(o))
        out = out.index_add(0, ot, features.index_select(0, it) @ w[k])
    if bias is not None:
        out = out + bias
    return out


def dense(features, indices, batch_size, spatial_shape):
    """[N,C] -> [B,C,D,H,W]"""
    c = features.shape[1]
    out = features.new_zeros((batch_size, c, *spatial_shape))
    idx = torch.as_tensor(np.asarray(indices)).long()
    out[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = features
    return out


def dense_conv_reference(features, indices, batch_size, spatial_shape, weight, ksize, stride,
                         padding, subm, bias=None):
    """The same layer via F.conv3d on the densified tensor.  Returns (out_features,
    out_indices) with strided outputs in canonical (sorted) order / subm outputs in input
    order.  Only feasible on small grids."""
    x = dense(features, indices, batch_size, spatial_shape)
    mask = dense(torch.ones((features.shape[0], 1), dtype=features.dtype), indices, batch_size,
                 spatial_shape)
    w = weight.permute(4, 3, 0, 1, 2).contiguous()  # [Cout, Cin, kD, kH, kW]
    idx = torch.as_tensor(np.asarray(indices)).long()
    if subm:
        pad = [k // 2 for k in ksize]
        y = F.conv3d(x, w, bias, stride=1, padding=pad)
        return y[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]], np.asarray(indices).astype(np.int32)
    y = F.conv3d(x, w, bias, stride=stride, padding=padding)
    ones = torch.ones((1, 1, *ksize), dtype=features.dtype)
    act = F.conv3d(mask, ones, None, stride=stride, padding=padding) > 0.5
    oi = act[:, 0].nonzero()  # sorted lexicographically by (b, z, y, x) == ascending linear index
    return y[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]], oi.numpy().astype(np.int32)


class SparseConvRef(torch.nn.Module):
    """nn.Module form used by the CPU baseline model: caches rulebooks by indice_key."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0, bias=False, subm=False,
                 indice_key=None):
        super().__init__()
        t = lambda v: [int(q) for q in v] if isinstance(v, (list, tuple)) else [int(v)] * 3
        self.ksize, self.stride, self.padding = t(ksize), t(stride), t(padding)
        if subm:
            self.stride, self.padding = [1, 1, 1], [0, 0, 0]
        self.subm = subm
        self.indice_key = indice_key
        self.weight = torch.nn.Parameter(torch.empty(*self.ksize, cin, cout))
        torch.nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = torch.nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, st):
        feats, indices, shape, batch, cache = st
        key = self.indice_key if self.indice_key is not None else id(self)
        if key not in cache:
            cache[key] = get_indice_pairs(indices, batch, shape, self.ksize, self.stride,
                                          self.padding, self.subm)
        out_idx, out_shape, pairs = cache[key]
        out = indice_conv(feats, self.weight, pairs, len(out_idx), self.bias)
        return (out, out_idx, out_shape, batch, cache)
