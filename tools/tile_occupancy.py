"""How many (tile, kernel offset) stages of the sparse layers are empty?  (CPU, numpy; no GPU needed)

VERDICT round 2 item 4 proposed a 27-bit "non-empty offset" mask per 16-site tile so that the gather kernel skips
empty (tile, offset) pairs.  The kernel already skips a stage when none of a wave's sites has the neighbour
(d3d_spconv.hip, `okm`).  This script measures what masks could add on the benched geometry -- the KITTI-like
synthetic frame through the SpMiddleFHD stages (scn.py:106-157), rows in the order the rulebooks produce them:

  * submanifold layers: fraction of non-empty (T-row tile, offset) stages for T = 16, 32, 64;
  * strided layers' data gradient (input-stationary table): the same in cell order and in parity-class order
    (csrc/d3d_rowplan.h).

    python tools/tile_occupancy.py > profiles/round3_tile_occupancy.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from det3d_amd.synthetic import KITTI_CAR, kitti_like_cloud  # noqa: E402


def lin(c, shape):
    return (c[:, 0] * shape[1] + c[:, 1]) * shape[2] + c[:, 2]


def voxels(seed=0, n_points=20000):
    vs, r = np.array(KITTI_CAR["voxel_size"]), np.array(KITTI_CAR["pc_range"])
    pts, _ = kitti_like_cloud(seed, n_points, 8)
    c = np.floor((pts[:, :3] - r[:3]) / vs).astype(np.int64)
    grid = np.round((r[3:] - r[:3]) / vs).astype(np.int64)
    c = c[((c >= 0) & (c < grid)).all(1)][:, ::-1]
    shape = np.array([grid[2] + 1, grid[1], grid[0]])
    _, first = np.unique(lin(c, shape), return_index=True)
    return c[np.sort(first)], shape  # first-point order, like the voxelizer


def offsets(k):
    return [(a, b, c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])]


def strided(coords, shape, k, s, p):
    """output cells (cell order) and the input-stationary table [kvol, n_in] of a strided layer"""
    k, s, p = np.array(k), np.array(s), np.array(p)
    oshape = (shape + 2 * p - k) // s + 1
    cand = []
    for kk in offsets(k):
        t = coords + p - np.array(kk)
        ok = (t >= 0).all(1) & (t % s == 0).all(1) & (t // s < oshape).all(1)
        cand.append((ok, t // s))
    keys = np.unique(np.concatenate([lin(o[ok], oshape) for ok, o in cand]))
    out = np.stack([keys // (oshape[1] * oshape[2]), (keys // oshape[2]) % oshape[1], keys % oshape[2]], 1)
    nbr_in = np.full((len(cand), len(coords)), -1)
    for j, (ok, o) in enumerate(cand):
        nbr_in[j, ok] = np.searchsorted(keys, lin(o[ok], oshape))
    return out, oshape, nbr_in


def subm_table(coords, shape):
    keys = lin(coords, shape)
    order = np.argsort(keys)
    ks = keys[order]
    n = len(coords)
    nbr = np.full((27, n), -1)
    for j, kk in enumerate(offsets((3, 3, 3))):
        q = coords + np.array(kk) - 1
        ok = ((q >= 0) & (q < shape)).all(1)
        ql = lin(q, shape)
        pos = np.minimum(np.searchsorted(ks, ql), n - 1)
        hit = ok & (ks[pos] == ql)
        nbr[j, hit] = order[pos[hit]]
    return nbr


def report(name, nbr):
    v = nbr >= 0
    n = v.shape[1]
    line = "%-34s rows %6d  offsets/row %5.2f of %2d " % (name, n, v.sum() / n, v.shape[0])
    for t in (16, 32, 64):
        vv = np.pad(v, ((0, 0), (0, (-n) % t))).reshape(v.shape[0], -1, t).any(2)
        line += "  T%-2d non-empty stages %.3f" % (t, vv.mean())
    print(line)


def report_ranges(name, nbr, t=32):
    """for tiles of `t` consecutive rows and each kz slice (nine offsets), the span of the valid
    neighbour indices -- the rows a coalesced range load would stage -- against the row requests of the per-offset gather"""
    v = nbr >= 0
    n = v.shape[1]
    nt = (n + t - 1) // t
    pad = (-n) % t
    lo = np.where(v, nbr, np.iinfo(np.int64).max)
    hi = np.where(v, nbr, -1)
    lo = np.pad(lo, ((0, 0), (0, pad)), constant_values=np.iinfo(np.int64).max).reshape(27, nt, t)
    hi = np.pad(hi, ((0, 0), (0, pad)), constant_values=-1).reshape(27, nt, t)
    spans = []
    for dz in range(3):
        a = lo[dz * 9:(dz + 1) * 9].min((0, 2))
        b = hi[dz * 9:(dz + 1) * 9].max((0, 2))
        spans.append(np.where(b >= 0, b - a + 1, 0))
    spans = np.stack(spans)  # [3, tiles]
    valid = v.sum()
    print("%-34s rows %6d  T%d: span per (tile, kz) median %d  mean %.1f  p90 %d  p99 %d;  <= 48 rows: %.3f  <= 64: %.3f;  "
          "rows staged / valid row requests %.2f, / all row requests %.2f" % (
              name, n, t, np.median(spans), spans.mean(), np.percentile(spans, 90), np.percentile(spans, 99),
              (spans <= 48).mean(), (spans <= 64).mean(), spans.sum() / valid, spans.sum() / (27.0 * n)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ranges":
        c, shape = voxels()
        geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
        print("# tools/tile_occupancy.py ranges: one KITTI-like frame (20k points), rows in the order the rulebooks produce them "
              "(stage 1: first-point order; later stages: cell order)")
        for stage, g in enumerate(geoms + [None], 1):
            report_ranges("stage %d submanifold 3x3x3" % stage, subm_table(c, shape))
            if stage == 1:  # what cell order would give the first stage
                o = np.argsort(lin(c, shape))
                report_ranges("stage 1, rows sorted by cell", subm_table(c[o], shape))
            if g is None:
                break
            c, shape, _ = strided(c, shape, *g)
        return
    c, shape = voxels()
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
             ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    for stage, (k, s, p) in enumerate(geoms, 1):
        report("stage %d submanifold 3x3x3" % stage, subm_table(c, shape))
        out, oshape, nbr_in = strided(c, shape, k, s, p)
        report("stage %d -> %d dgrad, cell order" % (stage, stage + 1), nbr_in)
        r = (c + np.array(p)) % np.array(s)
        cls = (r[:, 0] * s[1] + r[:, 1]) * s[2] + r[:, 2]
        report("stage %d -> %d dgrad, class order" % (stage, stage + 1), nbr_in[:, np.argsort(cls, kind="stable")])
        c, shape = out, oshape


if __name__ == "__main__":
    main()
