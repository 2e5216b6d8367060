"""spconv-shaped API over the HIP sparse-conv kernels (C ABI in include/d3d.h).

Mirrors what det3d/models/backbones/scn.py uses from the (un-vendored) `spconv`
package: SparseConvTensor (with mutable .features and .dense()), SparseSequential,
SparseModule, SubMConv3d, SparseConv3d -- same constructor arguments, weight
layout [kD, kH, kW, Cin, Cout] and state-dict names, so reference checkpoints and
the unmodified scn.py-style module stacks work (seam B3).

MI355X-native differences under the hood:
  * rulebooks are output-stationary neighbour tables built on device
    (d3d_subm_rulebook / d3d_conv_rulebook); strided outputs come out sorted by
    linear cell index (canonical order);
  * `SparseSequential.plan()` enqueues every rulebook of a backbone as one chain
    (device-side counts) and pays ONE host sync for all output sizes instead of one
    per strided layer;
  * conv fwd / dgrad / wgrad are single launches per layer (d3d_sparse_conv_*).
There is no CPU implementation here: tensors must live on the GPU.
"""
import ctypes
import math
import os
import sys

import numpy as np
import torch
from torch import nn

from . import _lib, arena as mem, sparse_stack

_SELF = sys.modules[__name__]
_DT = {torch.float32: _lib.D3D_F32, torch.bfloat16: _lib.D3D_BF16}

# bench.py sets this to a list to time every d3d_sparse_conv_fwd launch with HIP events
# recorded on the launch stream: entries are (kind, cin, cout, n_out, rulebook, ev0, ev1)
FUSE_BN_RELU = True  # SparseSequential / SparseBasicBlock route BatchNorm1d(+ReLU) to d3d_bn_relu_*
LAUNCH_LOG = None
LAUNCH_LOG_LIMIT = 256
LAUNCH_FILTER = None  # optional predicate (kind, cin, cout) -> bool: a timing event pair costs
#                       ~0.4 ms of stream time on ROCm, so bench.py times only a sample


def _timed(kind, cin, cout, n_out, rb, dev, fn, stream=None):
    """fn() = one launch; `stream`: the torch stream it goes to when that is not the current one"""
    if LAUNCH_LOG is None or len(LAUNCH_LOG) >= LAUNCH_LOG_LIMIT or \
            (LAUNCH_FILTER is not None and not LAUNCH_FILTER(kind, cin, cout)):
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    e0.record(st)
    rc = fn()
    e1.record(st)
    LAUNCH_LOG.append((kind, cin, cout, n_out, rb, e0, e1))
    return rc


DGRAD_PLAN = os.environ.get("D3D_DGRAD_PLAN", "1") != "0"  # strided data gradients in parity-class row order (csrc/d3d_rowplan.h)
FUSE_BN_STATS = os.environ.get("D3D_FUSE_BN_STATS", "1") != "0"  # BN batch statistics from the producing conv's epilogue, BN-backward reductions from the
#                       consuming conv's dgrad epilogue (d3d_sparse_conv_fwd_ex): 2 instead of 4 BN sweeps per layer


class _ScratchArena:
    """One flat f32 buffer per device handing out the fold workspaces of the conv-epilogue reductions (BN
    statistics / BN-backward sums: [2C] result + per-block slabs, see csrc/d3d_fold.h).  Nothing needs zeroing
    (round 2's atomically accumulated replicas did).  `reset()` -- called by the backbones right before they run
    their sparse stack -- rewinds the buffer and bumps `generation`; views handed out before a reset must not be
    written afterwards (the users compare generations).  Without resets (layers used on their own) the arena
    simply runs out and hands out private buffers."""

    WORDS = 1 << 24  # 64 MB

    def __init__(self):
        self.buf = {}
        self.off = {}
        self.generation = 0

    def reset(self, dev):
        self.off[dev] = 0
        self.generation += 1

    def take(self, dev, n):
        n = (int(n) + 63) // 64 * 64  # 256-byte aligned slices
        b = self.buf.get(dev)
        off = self.off.get(dev, 0)
        if b is None:
            b = self.buf[dev] = torch.empty((self.WORDS,), dtype=torch.float32, device=dev)
        if off + n > b.numel():  # arena exhausted (very deep stack / huge clouds): a private buffer
            return torch.empty((n,), dtype=torch.float32, device=dev)
        self.off[dev] = off + n
        return b[off: off + n]


def check_gather_source(t, what):
    """every matrix a conv kernel GATHERS rows from is addressed through a 2 GB buffer descriptor (rows beyond it
    would read as zeros): features in the forward, the output gradient in the data gradient, the padded features
    in the weight gradient"""
    if t.numel() * t.element_size() >= (1 << 31) - 16:
        raise _lib.D3dError("sparse conv: %s %s must stay below 2 GB (buffer-descriptor gathers)" % (what, tuple(t.shape)))


def epilogue_ws(dev, n_out, cout):
    """buffer for D3dConvEpilogue.stats / .bn_partial of a launch producing n_out rows of cout channels"""
    return ARENA.take(dev, _lib.lib().d3d_conv_epilogue_ws_floats(int(n_out), int(cout)))


def reduce_ws(dev, n, c, dtype, arena=None):
    """workspace of d3d_bn_stats / d3d_bn_bwd_reduce on an [n, c] matrix; the [2c] result sits at its start"""
    return mem.empty((_lib.lib().d3d_bn_reduce_ws_floats(int(n), int(c), _DT[dtype]),), torch.float32, dev, arena)


def has_epilogue(cin, cout, dtype):
    """channel pairs whose conv kernel can carry a BN reduction in its epilogue (the MFMA tilings)"""
    return dtype in _DT and bool(_lib.lib().d3d_sparse_conv_has_epilogue(int(cin), int(cout), _DT[dtype]))


ARENA = _ScratchArena()


class _BnLink:
    """Shared between a fused BN(+ReLU) op and the ONE sparse conv that consumes its output: lets the conv's data
    gradient launch accumulate the BN-backward reductions (sum dz*xhat, sum dz) in its epilogue."""

    __slots__ = ("x", "save", "gamma", "beta", "relu", "partial", "generation", "reduced")

    def __init__(self):
        self.x = self.save = self.gamma = self.beta = self.partial = None
        self.relu = False
        self.generation = -1
        self.reduced = False


_PACK_CACHE = {}  # id(weight) -> (weakref, version, dtype, flip, pad, wpack, wdgrad, (data_ptr, device))


def _packed_weights(weight, kvol, cin, cout, dtype, subm, need_dgrad, pad_cin):
    """[kD,kH,kW,Cin,Cout] f32 parameter -> the kernels' packed forms (one launch), cached until the parameter
    changes (optimizer steps bump `weight._version`): repeated forwards -- evaluation, several micro-batches per
    step -- do not repack."""
    import weakref

    key = id(weight)
    hit = _PACK_CACHE.get(key)
    # (`_version` does not move for writes through `.data` -- parameter broadcasts, EMA copies, `.to()` re-pointing
    # `.data` -- so the storage pointer and device are part of the identity as well; dist.broadcast_params also
    # clears the cache)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == dtype and \
            hit[3] == subm and hit[4] == pad_cin and (hit[6] is not None or not need_dgrad) and \
            hit[7] == (weight.data_ptr(), weight.device):
        return hit[5], hit[6]
    L = _lib.lib()
    w32 = weight.detach().float().reshape(kvol, cin, cout)
    if pad_cin != cin:
        w32 = torch.nn.functional.pad(w32, (0, 0, 0, pad_cin - cin))
    w32 = w32.contiguous()
    wpack = torch.empty((kvol, cout, pad_cin), dtype=dtype, device=weight.device)
    wd = torch.empty((kvol, pad_cin, cout), dtype=dtype, device=weight.device) if need_dgrad else None
    rc = L.d3d_pack_conv_weights(w32.data_ptr(), kvol, pad_cin, cout, _DT[dtype], int(subm), wpack.data_ptr(),
                                 _lib.ptr(wd), _lib.stream_ptr(weight.device))
    _lib.check(rc, "d3d_pack_conv_weights")
    if len(_PACK_CACHE) > 512:
        for k in [k for k, v in _PACK_CACHE.items() if v[0]() is None]:
            del _PACK_CACHE[k]
    _PACK_CACHE[key] = (weakref.ref(weight), weight._version, dtype, subm, pad_cin, wpack, wd,
                        (weight.data_ptr(), weight.device))
    return wpack, wd


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def _need_cuda(t, what):
    if not t.is_cuda:
        raise _lib.D3dError("%s: tensor is on %s; det3d_amd sparse ops run on the GPU only "
                            "(no CPU fallback)" % (what, t.device))


class Rulebook:
    """Neighbour tables of one conv geometry (shared through `indice_key`)."""

    def __init__(self, kvol, cap_in, cap_out, nbr_out, nbr_in, pair_count, n_in, n_out,
                 out_indices, out_shape, subm):
        self.kvol = kvol
        self.cap_in = cap_in
        self.cap_out = cap_out
        self.nbr_out = nbr_out      # int32 [kvol*cap_out]
        self.nbr_in = nbr_in        # int32 [kvol*cap_in] (strided only)
        self.pair_count = pair_count  # int32 [kvol] device
        self.n_in = n_in
        self.n_out = n_out
        self.out_indices = out_indices
        self.out_shape = out_shape
        self.subm = subm
        self.dgrad_plan = None      # int32 row plan of the strided data gradient (csrc/d3d_rowplan.h) or None
        self._pair_tables = {}      # "out" / "in" -> uint8 buffer of d3d_pair_table_build

    def pair_table(self, side="out"):
        """the pair-compacted form of a neighbour table (include/d3d.h, d3d_pair_table_build): "out" = the
        output-stationary table (forward; data gradient of submanifold layers), "in" = the input-stationary twin of a
        strided layer (its data gradient).  Built on first use on the current stream -- SparseSequential.plan() does
        that ahead of time on the geometry stream -- and kept with the rulebook."""
        hit = self._pair_tables.get(side)
        # (a table that was allocated from a bump arena dies with that arena's generation: rebuilt if the rulebook outlives it)
        t = hit[0] if hit is not None and (hit[1].arena is None or hit[1].arena.generation == hit[1].generation) else None
        if t is None:
            L = _lib.lib()
            lease = mem.Lease()
            nbr, cap, n = (self.nbr_out, self.cap_out, self.n_out) if side == "out" else (self.nbr_in, self.cap_in, self.n_in)
            nb = L.d3d_pair_table_bytes(int(n), self.kvol)
            t = mem.empty((nb,), torch.uint8, nbr.device)
            _lib.check(L.d3d_pair_table_build(nbr.data_ptr(), cap, int(n), self.kvol, t.data_ptr(), nb,
                                              _lib.stream_ptr(nbr.device)), "d3d_pair_table_build")
            self._pair_tables[side] = (t, lease)
        return t

    def pair_counts(self):
        """pairs per kernel offset, int64 [kvol] on the device.  Counted on demand from the
        table (the kernels skip their atomic counters on the hot path)."""
        if self.pair_count is None:
            t = self.nbr_out.view(self.kvol, self.cap_out)[:, : self.n_out]
            self.pair_count = (t >= 0).sum(1)
        return self.pair_count

    def pairs(self):
        """number of rulebook pairs P (host int; one small D2H)."""
        return int(self.pair_counts().sum().item())

    def canonical_pairs(self):
        """(k, in_row, out_row) triples as a sorted numpy array -- test helper."""
        t = self.nbr_out.view(self.kvol, self.cap_out)[:, : self.n_out].cpu().numpy()
        k, o = np.nonzero(t >= 0)
        trip = np.stack([k, t[k, o], o], axis=1).astype(np.int64)
        order = np.lexsort((trip[:, 2], trip[:, 1], trip[:, 0]))
        return trip[order]


def conv_out_shape(in_shape, ksize, stride, padding):
    return [(in_shape[j] + 2 * padding[j] - ksize[j]) // stride[j] + 1 for j in range(3)]


def _count_tensor(n, device):
    return torch.tensor([int(n)], dtype=torch.int32, device=device)


def build_subm_rulebook(indices, n_dev, n, cap, batch, spatial_shape, ksize, count_pairs=False):
    """Enqueue d3d_subm_rulebook.  `n` may be None while the count is still only on
    the device (planning); `cap` bounds it."""
    L = _lib.lib()
    dev = indices.device
    kvol = ksize[0] * ksize[1] * ksize[2]
    nbr = mem.empty((kvol * cap,), torch.int32, dev)
    pc = torch.empty((kvol,), dtype=torch.int32, device=dev) if count_pairs else None
    wsb = L.d3d_subm_rulebook_workspace_bytes(cap)
    ws = mem.empty((wsb,), torch.uint8, dev)
    rc = L.d3d_subm_rulebook(indices.data_ptr(), n_dev.data_ptr(), cap, batch,
                             _lib.i32_array(spatial_shape), _lib.i32_array(ksize), nbr.data_ptr(),
                             _lib.ptr(pc), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_subm_rulebook")
    return Rulebook(kvol, cap, cap, nbr, None, pc, n, n, indices, list(spatial_shape), True)


def build_conv_rulebook(indices, n_dev, n, cap_in, batch, in_shape, ksize, stride, padding,
                        count_pairs=False):
    """Enqueue d3d_conv_rulebook; returns (rulebook, n_out_dev).  rulebook.n_out and
    .out_indices are provisional (capacity views) until `finalize_conv_rulebook`."""
    L = _lib.lib()
    dev = indices.device
    out_shape = conv_out_shape(in_shape, ksize, stride, padding)
    kvol = ksize[0] * ksize[1] * ksize[2]
    fan = 1
    for j in range(3):
        fan *= -(-ksize[j] // stride[j])  # ceil(k/s): outputs one input can reach per axis
    cells = batch * out_shape[0] * out_shape[1] * out_shape[2]
    cap_out = int(max(1, min(cap_in * fan, cells)))
    coords_out = mem.empty((cap_out, 4), torch.int32, dev)
    n_out_dev = mem.empty((1,), torch.int32, dev)
    nbr_out = mem.empty((kvol * cap_out,), torch.int32, dev)
    nbr_in = mem.empty((kvol * cap_in,), torch.int32, dev)
    pc = torch.empty((kvol,), dtype=torch.int32, device=dev) if count_pairs else None
    oshape = _lib.i32_array(out_shape)
    wsb = L.d3d_conv_rulebook_workspace_bytes(batch, oshape)
    ws = mem.empty((wsb,), torch.uint8, dev)
    rc = L.d3d_conv_rulebook(indices.data_ptr(), n_dev.data_ptr(), cap_in, batch,
                             _lib.i32_array(in_shape), _lib.i32_array(ksize),
                             _lib.i32_array(stride), _lib.i32_array(padding), oshape,
                             coords_out.data_ptr(), n_out_dev.data_ptr(), cap_out,
                             nbr_out.data_ptr(), nbr_in.data_ptr(), _lib.ptr(pc), ws.data_ptr(),
                             wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv_rulebook")
    rb = Rulebook(kvol, cap_in, cap_out, nbr_out, nbr_in, pc, n, None, coords_out, out_shape, False)
    if DGRAD_PLAN and L.d3d_dgrad_plan_supported(_lib.i32_array(ksize), _lib.i32_array(stride)):
        # rows of the data gradient in parity-class order: 27 / 8 offsets per block instead of 27 (three small
        # launches next to the rulebook's, on the same stream)
        words = L.d3d_dgrad_plan_ints(cap_in)
        plan = mem.empty((words,), torch.int32, dev)
        rc = L.d3d_dgrad_plan_build(indices.data_ptr(), n_dev.data_ptr(), cap_in, _lib.i32_array(ksize),
                                    _lib.i32_array(stride), _lib.i32_array(padding), plan.data_ptr(), words,
                                    _lib.stream_ptr(dev))
        _lib.check(rc, "d3d_dgrad_plan_build")
        rb.dgrad_plan = plan
    return rb, n_out_dev


RB_CHAIN = os.environ.get("D3D_RB_CHAIN", "1") != "0"  # SparseSequential.plan(): all strided layers in one chain call


def build_conv_rulebook_chain(indices, n_dev, cap0, batch, in_shape, convs):
    """Enqueue d3d_conv_rulebook_chain for the strided layers `convs` (in execution order; the first one reads
    `indices` / `n_dev` / `cap0`).  Returns ([(rulebook, n_out_dev)] as build_conv_rulebook would per layer, the row
    plans included; [(bitmap pointer, prefix pointer)] per produced level for build_subm_rulebook_ranked; the
    workspace tensor those pointers live in -- keep it until the ranked builds are enqueued)."""
    L = _lib.lib()
    dev = indices.device
    arr = (_lib.ChainLayer * len(convs))()
    out, cap_in, shape = [], int(cap0), list(in_shape)
    for j, m in enumerate(convs):
        ksize, stride, padding = m.kernel_size, m.stride, m.padding
        out_shape = conv_out_shape(shape, ksize, stride, padding)
        kvol = ksize[0] * ksize[1] * ksize[2]
        fan = 1
        for a in range(3):
            fan *= -(-ksize[a] // stride[a])
        cells = batch * out_shape[0] * out_shape[1] * out_shape[2]
        cap_out = int(max(1, min(cap_in * fan, cells)))
        coords_out = mem.empty((cap_out, 4), torch.int32, dev)
        n_out_dev = mem.empty((1,), torch.int32, dev)
        nbr_out = mem.empty((kvol * cap_out,), torch.int32, dev)
        nbr_in = mem.empty((kvol * cap_in,), torch.int32, dev)
        plan = None
        if DGRAD_PLAN and L.d3d_dgrad_plan_supported(_lib.i32_array(ksize), _lib.i32_array(stride)):
            plan = mem.empty((L.d3d_dgrad_plan_ints(cap_in),), torch.int32, dev)
        y = arr[j]
        for a in range(3):
            y.in_shape[a], y.ksize[a], y.stride[a], y.padding[a], y.out_shape[a] = shape[a], ksize[a], stride[a], padding[a], out_shape[a]
        y.cap_out = cap_out
        y.coords_out, y.n_out_dev, y.nbr_out, y.nbr_in = coords_out.data_ptr(), n_out_dev.data_ptr(), nbr_out.data_ptr(), nbr_in.data_ptr()
        y.pair_count = None
        y.plan, y.plan_ints = (plan.data_ptr(), plan.numel()) if plan is not None else (None, 0)
        rb = Rulebook(kvol, cap_in, cap_out, nbr_out, nbr_in, None, None, None, coords_out, out_shape, False)
        rb.dgrad_plan = plan
        out.append((rb, n_out_dev))
        cap_in, shape = cap_out, out_shape
    wsb = L.d3d_conv_rulebook_chain_workspace_bytes(len(convs), batch, ctypes.cast(arr, ctypes.c_void_p))
    if wsb == 0:
        raise _lib.D3dError("d3d_conv_rulebook_chain: unsupported chain of %d layers" % len(convs))
    ws = mem.empty((wsb,), torch.uint8, dev)
    rc = L.d3d_conv_rulebook_chain(len(convs), ctypes.cast(arr, ctypes.c_void_p), batch, indices.data_ptr(),
                                   n_dev.data_ptr(), int(cap0), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv_rulebook_chain")
    ranks = [(arr[j].rank_bitmap, arr[j].rank_prefix) for j in range(len(convs))]
    return out, ranks, ws


def build_subm_rulebook_ranked(indices, n_dev, n, batch, spatial_shape, ksize, rank):
    """submanifold table of a level produced by a chain layer (rows in ascending cell order): d3d_subm_rulebook_ranked
    on that layer's rank structure `rank` = (bitmap pointer, prefix pointer).  Exact-size table ([kvol * n])."""
    L = _lib.lib()
    dev = indices.device
    kvol = ksize[0] * ksize[1] * ksize[2]
    cap = max(int(n), 1)
    nbr = mem.empty((kvol * cap,), torch.int32, dev)
    rc = L.d3d_subm_rulebook_ranked(indices.data_ptr(), n_dev.data_ptr(), cap, batch, _lib.i32_array(spatial_shape),
                                    _lib.i32_array(ksize), rank[0], rank[1], nbr.data_ptr(), None, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_subm_rulebook_ranked")
    return Rulebook(kvol, cap, cap, nbr, None, None, n, n, indices, list(spatial_shape), True)


def finalize_conv_rulebook(rb, n_in, n_out):
    if n_out > rb.cap_out:
        raise _lib.D3dError("conv rulebook overflow: %d outputs > capacity %d" % (n_out, rb.cap_out))
    rb.n_in = n_in
    rb.n_out = n_out
    rb.out_indices = rb.out_indices[:n_out]
    return rb


# Pair-compacted kernel (d3d_sparse_conv_fwd_pairs, round 6): which launches take it.
#   "off"     none (the output-stationary tile kernels everywhere)
#   "all"     every f32 launch whose channel pair has the tiling (tests, tools/bench_pairs.py)
#   "measured" the launches it measured faster on MI355X alone on the device (profiles/round6_pairs_kernel.txt): the data
#             gradients of strided layers into 32 channels.  At two waves per SIMD (the 16 KB accumulator slab per wave
#             fills the LDS with 8 waves) it is bound by the instruction issue of its own waves, not by the MFMA pipe or
#             the gathers it saves: DESIGN.md section 10.
# Default "off": inside the training step those data gradients share the CUs with the weight-gradient stream, which a
# 160 KB block locks out.
from . import runtime as _runtime  # noqa: E402

PAIR_KERNELS = _runtime.get("pair_kernels")  # (det3d_amd.runtime.configure(pair_kernels=...) sets it)
_PAIRS_OK = {}


def pairs_ok(rb, cin, cout, dtype, strided_dgrad=False):
    """does the launch (cin -> cout channels, features of `dtype`) over rulebook `rb` take the pair-compacted kernel?"""
    if PAIR_KERNELS == "off" or not PAIR_KERNELS or dtype != torch.float32 or rb.kvol > 32:
        return False
    if PAIR_KERNELS == "measured" and not (strided_dgrad and cout == 32):
        return False
    L = _lib.lib()
    key = (cin, cout, L.d3d_sparse_conv_has_epilogue(cin, cout, _lib.D3D_F32))  # (knob 10 switches both answers off)
    ok = _PAIRS_OK.get(key)
    if ok is None:
        ok = _PAIRS_OK[key] = bool(L.d3d_sparse_conv_pairs_supported(cin, cout, _lib.D3D_F32))
    return ok


def launch_fwd(rb, x, wpack, bias, pad_cin, cout, out, ep, stream):
    """forward of one sparse conv over the output-stationary table: the pair-compacted kernel where the channel pair has
    one, else the tile kernel that walks every offset"""
    L = _lib.lib()
    epp = ctypes.byref(ep) if ep is not None else None
    if pairs_ok(rb, pad_cin, cout, x.dtype):
        table = rb.pair_table("out")
        return _timed("fwd", pad_cin, cout, rb.n_out, rb, x.device, lambda: L.d3d_sparse_conv_fwd_pairs(
            x.data_ptr(), wpack.data_ptr(), _lib.ptr(bias), table.data_ptr(), rb.n_out, rb.kvol, pad_cin, cout,
            _DT[x.dtype], out.data_ptr(), epp, stream))
    return _timed("fwd", pad_cin, cout, rb.n_out, rb, x.device, lambda: L.d3d_sparse_conv_fwd_ex(
        x.data_ptr(), wpack.data_ptr(), _lib.ptr(bias), rb.nbr_out.data_ptr(), rb.cap_out, rb.n_out, rb.kvol, pad_cin,
        cout, _DT[x.dtype], out.data_ptr(), epp, stream))


def launch_dgrad(rb, gout, wd, cout, pad_cin, gin, ep, stream):
    """data gradient of one sparse conv: the gather kernel over the table that is stationary in the INPUT rows
    (submanifold: the forward table with mirrored weights; strided: nbr_in -- pair-compacted where the channel pair has
    that kernel, else in parity-class row order when the rulebook carries a plan)"""
    L = _lib.lib()
    dt = _DT[gout.dtype]
    nbr, cap = (rb.nbr_out, rb.cap_out) if rb.subm else (rb.nbr_in, rb.cap_in)
    epp = ctypes.byref(ep) if ep is not None else None
    if pairs_ok(rb, cout, pad_cin, gout.dtype, strided_dgrad=not rb.subm):
        table = rb.pair_table("out" if rb.subm else "in")
        rc = _timed("dgrad", cout, pad_cin, rb.n_in, rb, gout.device, lambda: L.d3d_sparse_conv_fwd_pairs(
            gout.data_ptr(), wd.data_ptr(), None, table.data_ptr(), rb.n_in, rb.kvol, cout, pad_cin, dt, gin.data_ptr(),
            epp, stream))
        _lib.check(rc, "d3d_sparse_conv_fwd_pairs(dgrad)")
        return
    if not rb.subm and rb.dgrad_plan is not None and L.d3d_sparse_conv_dgrad_plan_supported(cout, pad_cin, dt):
        rc = _timed("dgrad", cout, pad_cin, rb.n_in, rb, gout.device, lambda: L.d3d_sparse_conv_dgrad_planned(
            gout.data_ptr(), wd.data_ptr(), nbr.data_ptr(), cap, rb.n_in, rb.kvol, cout, pad_cin, dt,
            gin.data_ptr(), epp, rb.dgrad_plan.data_ptr(), stream))
    else:
        rc = _timed("dgrad", cout, pad_cin, rb.n_in, rb, gout.device, lambda: L.d3d_sparse_conv_fwd_ex(
            gout.data_ptr(), wd.data_ptr(), None, nbr.data_ptr(), cap, rb.n_in, rb.kvol, cout, pad_cin, dt,
            gin.data_ptr(), epp, stream))
    _lib.check(rc, "d3d_sparse_conv_fwd(dgrad)")


# --------------------------------------------------------------------- autograd
class _SparseConvFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, feats, weight, bias, rb, stats, link):
        """stats: epilogue_ws() buffer or None -- the launch leaves sum / sum-of-squares of its output rows (the
        batch statistics of the BatchNorm that follows) in stats[:2*cout].  link: _BnLink of the fused BN(+ReLU)
        that produced `feats`, or None -- the backward's dgrad launch then also reduces that BN's gradient sums."""
        _need_cuda(feats, "sparse conv")
        L = _lib.lib()
        if feats.dtype not in _DT:
            raise _lib.D3dError("sparse conv supports float32 / bfloat16 features, got %s" % feats.dtype)
        feats = feats.contiguous()
        kvol, cin, cout = rb.kvol, weight.shape[-2], weight.shape[-1]
        assert feats.shape[0] == rb.n_in and feats.shape[1] == cin, (feats.shape, rb.n_in, cin)
        if weight.numel() != kvol * cin * cout:
            # e.g. two convs of different kernel size sharing one indice_key: the packer would read
            # kvol*cin*cout floats from a smaller weight
            raise _lib.D3dError("sparse conv: weight %s does not match the rulebook's kernel volume %d "
                                "(indice_key shared between different kernel sizes?)" % (tuple(weight.shape), kvol))
        # stem layers (3..5 input channels, scn.py:106-107 / :324): zero-pad the rows to 16 channels so the layer
        # runs on the MFMA kernels like every other one (the VALU fallbacks were the slowest launches of the
        # step for the smallest layer: 41 us forward, 105 us weight gradient)
        pad_cin = 16 if (cin < 16 and cout % 16 == 0) else cin
        if pad_cin != cin:
            feats = torch.nn.functional.pad(feats, (0, pad_cin - cin))
        check_gather_source(feats, "the feature matrix")
        need_dgrad = ctx.needs_input_grad[0]
        wpack, wd = _packed_weights(weight, kvol, cin, cout, feats.dtype, rb.subm, need_dgrad, pad_cin)
        lease = mem.Lease()
        out = mem.empty((rb.n_out, cout), feats.dtype, feats.device, lease.arena)
        b = bias.detach().float().contiguous() if bias is not None else None
        mfma = has_epilogue(pad_cin, cout, feats.dtype)
        ep = None
        if stats is not None and mfma and rb.n_out > 0:
            ep = _lib.ConvEpilogue(stats=stats.data_ptr(), ws_floats=stats.numel())
        rc = launch_fwd(rb, feats, wpack, b, pad_cin, cout, out, ep, _lib.stream_ptr(feats.device))
        _lib.check(rc, "d3d_sparse_conv_fwd")
        ctx.stats_done = ep is not None
        ctx.save_for_backward(feats, weight)
        ctx.wd = wd
        from . import repack
        ctx.repack_stamp = repack.stamp((weight,))  # `wd` is rewritten in place by Repacker.run()
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.pad = (cin, pad_cin)
        ctx.link = link
        ctx.mfma = mfma
        ctx.lease = lease
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, gout):
        L = _lib.lib()
        feats, weight = ctx.saved_tensors
        rb = ctx.rb
        kvol, cout = rb.kvol, weight.shape[-1]
        cin, pad_cin = ctx.pad
        gout = gout.contiguous().to(feats.dtype)
        check_gather_source(gout, "the output gradient")
        dt = _DT[feats.dtype]
        stream = _lib.stream_ptr(feats.device)
        A = ctx.lease.check("sparse conv")
        gin = gw = gb = None
        if ctx.needs_input_grad[0]:
            from . import repack
            repack.check(ctx.repack_stamp, "sparse conv")
            wd = ctx.wd  # packed in the forward: W[k] as is; k mirrored for submanifold layers because
            #              (j -> i at offset k)  <=>  (i -> j at offset kvol-1-k) on the same table
            gin = mem.empty((rb.n_in, pad_cin), feats.dtype, feats.device, A)
            ep = None
            link = ctx.link
            if link is not None and has_epilogue(cout, pad_cin, feats.dtype) and pad_cin == cin and rb.n_in > 0 and \
                    link.generation == ARENA.generation and not link.reduced and link.partial is not None:
                # gin is dL/dy of the BN(+ReLU) that produced `feats`: reduce its gradient sums here
                ep = _lib.ConvEpilogue(bn_x=link.x.data_ptr(), bn_mean=link.save[0].data_ptr(),
                                       bn_invstd=link.save[1].data_ptr(), bn_gamma=_lib.ptr(link.gamma),
                                       bn_beta=_lib.ptr(link.beta), bn_relu=int(link.relu),
                                       bn_partial=link.partial.data_ptr(), ws_floats=link.partial.numel())
            launch_dgrad(rb, gout, wd, cout, pad_cin, gin, ep, stream)
            if ep is not None:
                link.reduced = True
            if pad_cin != cin:
                gin = gin[:, :cin].contiguous()
        if ctx.needs_input_grad[1]:
            gw = torch.empty((kvol, pad_cin, cout), dtype=torch.float32, device=feats.device)
            wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(rb.n_out, kvol, pad_cin, cout, dt)
            ws = mem.empty((wsb,), torch.uint8, feats.device, A)
            rc = _timed("wgrad", pad_cin, cout, rb.n_out, rb, feats.device, lambda: L.d3d_sparse_conv_wgrad(
                feats.data_ptr(), gout.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, rb.n_out, kvol,
                pad_cin, cout, dt, gw.data_ptr(), ws.data_ptr(), wsb, stream))
            _lib.check(rc, "d3d_sparse_conv_wgrad")
            if pad_cin != cin:
                gw = gw[:, :cin]
            gw = gw.reshape(weight.shape).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gout.float().sum(0)
        return gin, gw, gb, None, None, None


class _BNReLUFn(torch.autograd.Function):
    """BatchNorm1d (+ReLU) on [N, C] features through the d3d_bn_* kernels.  `stats` (optional): the batch sums
    already accumulated by the producing conv's epilogue -> only the apply pass runs.  `link` (optional): filled
    for the single conv consuming the output, whose dgrad epilogue then does this op's backward reduction."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, training, relu, stats, link):
        L = _lib.lib()
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        lease = mem.Lease()
        y = mem.empty((n, c), x.dtype, dev, lease.arena)
        save = torch.empty((2, c), dtype=torch.float32, device=dev)  # mean, invstd
        st = _lib.stream_ptr(dev)
        copies = 1  # every producer (conv epilogue, d3d_bn_stats) folds its sums into one [2c] vector
        if training and stats is None:
            stats = reduce_ws(dev, n, c, x.dtype, lease.arena)
            _lib.check(L.d3d_bn_stats(x.data_ptr(), n, c, _DT[x.dtype], stats.data_ptr(), st), "d3d_bn_stats")
        rc = L.d3d_bn_apply(x.data_ptr(), n, c, _DT[x.dtype], _lib.ptr(weight), _lib.ptr(bias),
                            _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), float(eps),
                            int(training), int(relu), _lib.ptr(stats) if training else None, copies, y.data_ptr(),
                            save[0].data_ptr(), save[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_apply")
        ctx.save_for_backward(x, weight, bias, save)
        ctx.relu = relu
        ctx.training = training
        ctx.link = link
        ctx.lease = lease
        if link is not None and training:
            link.x, link.save, link.gamma, link.beta, link.relu = x, save, weight, bias, bool(relu)
            link.partial = epilogue_ws(dev, n, c)
            link.generation = ARENA.generation
            link.reduced = False
        return y

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        if not ctx.training:
            raise _lib.D3dError("fused sparse BN backward is implemented for training mode only")
        L = _lib.lib()
        x, weight, bias, save = ctx.saved_tensors
        n, c = x.shape
        dy = dy.contiguous().to(x.dtype)
        A = ctx.lease.check("sparse BatchNorm")
        dx = mem.empty((n, c), x.dtype, x.device, A)
        g = torch.empty((2, c), dtype=torch.float32, device=x.device)
        st = _lib.stream_ptr(x.device)
        link = ctx.link
        copies = 1
        if link is not None and link.reduced and link.generation == ARENA.generation and link.partial is not None:
            partial = link.partial  # reduced by the consuming conv's dgrad epilogue
        else:
            partial = reduce_ws(x.device, n, c, x.dtype, A)
            rc = L.d3d_bn_bwd_reduce(x.data_ptr(), dy.data_ptr(), n, c, _DT[x.dtype], _lib.ptr(weight),
                                     _lib.ptr(bias), save[0].data_ptr(), save[1].data_ptr(), int(ctx.relu),
                                     partial.data_ptr(), st)
            _lib.check(rc, "d3d_bn_bwd_reduce")
        rc = L.d3d_bn_bwd_dx(x.data_ptr(), dy.data_ptr(), n, c, _DT[x.dtype], _lib.ptr(weight), _lib.ptr(bias),
                             save[0].data_ptr(), save[1].data_ptr(), int(ctx.relu), partial.data_ptr(), copies,
                             dx.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_bwd_dx")
        if link is not None:
            link.x = link.save = link.partial = None  # drop the references held for the epilogue
            link.reduced = False  # a second backward over the same graph runs its own reduction sweep
        gw = g[0].to(weight.dtype) if weight is not None else None
        gb = g[1].to(bias.dtype) if bias is not None else None
        return dx, gw, gb, None, None, None, None, None, None, None, None


def _fusable_bn(bn, feats, channels=None):
    """plain affine nn.BatchNorm1d with running stats, on CUDA f32/bf16 features whose channel
    count the kernel supports (c/8 | 256 for bf16, c/4 | 256 for f32).  `channels`: judge a BN that will see
    `channels`-wide rows of the same dtype / row count class as `feats` (the output of a conv not yet run)."""
    if type(bn) is not nn.BatchNorm1d or not feats.is_cuda or feats.dtype not in _DT or feats.dim() != 2:
        return False
    if bn.momentum is None or not bn.track_running_stats or bn.weight is None:
        return False
    # the kernels read gamma / beta / running statistics as float32 (after model.bfloat16() they are not)
    if any(t is not None and t.dtype != torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)):
        return False
    if feats.shape[0] < 2 or (not bn.training and torch.is_grad_enabled() and feats.requires_grad):
        return False
    c = feats.shape[1] if channels is None else channels
    if bn.num_features != c:
        return False
    epv = 8 if feats.dtype == torch.bfloat16 else 4
    return c % epv == 0 and c <= 256 and 256 % (c // epv) == 0


def bn_relu(bn, feats, relu, stats=None, link=None):
    """apply `bn` (an nn.BatchNorm1d) and optionally ReLU to sparse features with the fused HIP
    kernels; state (running stats) lives in the module exactly as with F.batch_norm."""
    return _BNReLUFn.apply(feats, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                           bn.momentum, bn.eps, bn.training, relu, stats, link)


def conv_bn_relu(conv, bn, relu, x, link_in=None, want_link=False):
    """conv -> BatchNorm1d [-> ReLU] on a SparseConvTensor with the BN batch statistics taken from the conv's
    epilogue.  link_in: _BnLink of the fused BN that produced x.features (this conv being its only consumer).
    want_link: return a _BnLink for the single conv that will consume the result.  Returns (tensor, link)."""
    fuse = FUSE_BN_STATS and bn.training and x.features.is_cuda and x.features.shape[0] > 0
    stats = None
    if fuse:
        stats = True  # the conv takes an epilogue_ws() slice once it knows its output row count
    out = conv(x, stats=stats, link=link_in if FUSE_BN_STATS else None)
    stats = getattr(out, "_stats", None)
    link = _BnLink() if (want_link and fuse) else None
    out.features = bn_relu(bn, out.features, relu, stats, link)
    return out, link


def fusable_bn2d(bn, x):
    """plain affine nn.BatchNorm2d with running stats on a CUDA f32/bf16 feature map stored channels-last,
    i.e. a row-major [B*H*W, C] matrix in memory -- exactly what d3d_bn_relu_* sweeps."""
    if type(bn) is not nn.BatchNorm2d or not x.is_cuda or x.dtype not in _DT or x.dim() != 4:
        return False
    if bn.momentum is None or not bn.track_running_stats or bn.weight is None or not bn.training:
        return False
    if any(t is not None and t.dtype != torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)):
        return False
    if x.shape[1] == 1 or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    c = x.shape[1]
    epv = 8 if x.dtype == torch.bfloat16 else 4
    return c % epv == 0 and c <= 256 and 256 % (c // epv) == 0 and x.numel() // c >= 2


def bn_relu_2d(bn, x, relu):
    """BatchNorm2d (+ReLU) of a channels-last [B, C, H, W] map through the sparse-feature kernels."""
    b, c, h, w = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)  # a view: channels-last memory is [B,H,W,C]
    y = _BNReLUFn.apply(rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                        bn.training, relu, None, None)
    # like bn_relu(): num_batches_tracked is not advanced (one launch per layer for a counter that
    # momentum-based running statistics never read)
    return y.view(b, h, w, c).permute(0, 3, 1, 2)


class _DenseFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, feats, indices, batch, shape, channels_last):
        _need_cuda(feats, "SparseConvTensor.dense")
        L = _lib.lib()
        feats = feats.contiguous()
        n, c = feats.shape
        D, H, W = shape
        if channels_last:  # memory [B, H, W, C*D], presented as a channels-last [B, C*D, H, W]
            phys = torch.empty((batch, H, W, c * D), dtype=feats.dtype, device=feats.device)
            out = phys.permute(0, 3, 1, 2)
        else:
            phys = out = torch.empty((batch, c, D, H, W), dtype=feats.dtype, device=feats.device)
        rc = L.d3d_dense_scatter(feats.data_ptr(), indices.data_ptr(), n, c, batch,
                                 _lib.i32_array(shape), _DT[feats.dtype], int(channels_last),
                                 phys.data_ptr(), _lib.stream_ptr(feats.device))
        _lib.check(rc, "d3d_dense_scatter")
        ctx.save_for_backward(indices)
        ctx.meta = (n, c, batch, list(shape), channels_last)
        ctx.lease = mem.Lease()
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        L = _lib.lib()
        (indices,) = ctx.saved_tensors
        n, c, batch, shape, channels_last = ctx.meta
        g = g.contiguous(memory_format=torch.channels_last) if channels_last else g.contiguous()
        gf = mem.empty((n, c), g.dtype, g.device, ctx.lease.check("dense()"))
        rc = L.d3d_dense_gather(g.data_ptr(), indices.data_ptr(), n, c, batch,
                                _lib.i32_array(shape), _DT[g.dtype], int(channels_last),
                                gf.data_ptr(), _lib.stream_ptr(g.device))
        _lib.check(rc, "d3d_dense_gather")
        return gf, None, None, None, None


# ------------------------------------------------------------------- public API
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """features [N, C]; indices [N, 4] int32 (batch, z, y, x)."""
        self.features = features
        self.indices = indices.int().contiguous() if indices.dtype != torch.int32 else indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        out = _DenseFn.apply(self.features, self.indices, self.batch_size, self.spatial_shape, False)
        if not channels_first:
            return out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    def dense_2d(self, channels_last=False):
        """dense().view(B, C*D, H, W) in one step (what scn.py:192-195 does); with
        channels_last=True the result is a channels-last tensor written directly by the kernel."""
        if not channels_last:
            d = self.dense()
            n, c, dd, h, w = d.shape
            return d.view(n, c * dd, h, w)
        return _DenseFn.apply(self.features, self.indices, self.batch_size, self.spatial_shape, True)


class SparseModule(nn.Module):
    """marker base class, like spconv.SparseModule"""
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, **kwargs):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse
        assert _triple(dilation) == [1, 1, 1], "dilation is not used on the Det3D hot path"
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # spconv v1.x: kaiming_uniform_(a=sqrt(5)) on the [kD,kH,kW,Cin,Cout] tensor
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%s, padding=%s, subm=%s, indice_key=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
            self.subm, self.indice_key)

    # -- geometry ---------------------------------------------------------------
    def out_spatial_shape(self, in_shape):
        if self.subm:
            return list(in_shape)
        return conv_out_shape(in_shape, self.kernel_size, self.stride, self.padding)

    def _rulebook_key(self):
        return self.indice_key if self.indice_key is not None else ("_anon", id(self))

    def _get_rulebook(self, x):
        """Planned (SparseSequential.plan) or cached-by-indice_key rulebook, else build it
        now -- a strided build on this unplanned path costs one host sync."""
        key = self._rulebook_key()
        rb = x.indice_dict.get(key)
        if rb is not None:
            return rb
        n = x.indices.shape[0]
        n_dev = _count_tensor(n, x.indices.device)
        if self.subm:
            rb = build_subm_rulebook(x.indices, n_dev, n, max(n, 1), x.batch_size, x.spatial_shape,
                                     self.kernel_size)
        else:
            rb, n_out_dev = build_conv_rulebook(x.indices, n_dev, n, max(n, 1), x.batch_size,
                                                x.spatial_shape, self.kernel_size, self.stride,
                                                self.padding)
            finalize_conv_rulebook(rb, n, int(n_out_dev.item()))
        x.indice_dict[key] = rb
        return rb

    def forward(self, x, stats=None, link=None):
        assert isinstance(x, SparseConvTensor)
        rb = self._get_rulebook(x)
        if rb.subm != self.subm:
            raise _lib.D3dError("indice_key %r is shared between a submanifold and a strided convolution" % (self.indice_key,))
        cin = self.in_channels
        pad_cin = 16 if (cin < 16 and self.out_channels % 16 == 0) else cin
        if stats is not None:
            if not has_epilogue(pad_cin, self.out_channels, x.features.dtype) or rb.n_out == 0:
                stats = None  # no MFMA tiling for this channel pair: the BatchNorm runs its own statistics sweep
            elif stats is True:
                stats = epilogue_ws(x.features.device, rb.n_out, self.out_channels)
        out_feats = _SparseConvFn.apply(x.features, self.weight, self.bias, rb, stats, link)
        out = SparseConvTensor(out_feats, rb.out_indices if not self.subm else x.indices,
                               self.out_spatial_shape(x.spatial_shape), x.batch_size, x.grid)
        out.indice_dict = x.indice_dict
        out._stats = stats  # the epilogue's workspace: [2 * cout] sums first
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, **kwargs):
        # like spconv, stride/padding are accepted and ignored for submanifold convs
        super().__init__(3, in_channels, out_channels, kernel_size, 1, 0, dilation, groups, bias,
                         subm=True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, subm=False, indice_key=indice_key)


def is_spconv_module(m):
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    """spconv.SparseSequential: sparse modules get the SparseConvTensor, plain
    nn.Modules (BatchNorm1d, ReLU) are applied to `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        from collections import OrderedDict

        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        link = None  # _BnLink of the fused BN whose output is `input.features` (if this stack produced it)
        stack_ok = FUSE_BN_RELU and FUSE_BN_STATS and sparse_stack.FUSE_STACK and torch.is_grad_enabled()
        while i < len(mods):
            module = mods[i]
            if stack_ok and isinstance(module, SparseConvolution) and isinstance(input, SparseConvTensor) and \
                    input.features.is_cuda and input.features.dtype in _DT and \
                    (input.indices.shape[0] > 1 or (i + 1 < len(mods) and type(mods[i + 1]).__name__ == "DistributedSyncBN")):
                # a run of conv -> BatchNorm1d [-> ReLU] triples: one autograd node, a plain loop of C-ABI calls
                layers, nxt = sparse_stack.segment(_SELF, mods, i, input)
                if layers:
                    input = sparse_stack.run(_SELF, layers, input)
                    link = None
                    i = nxt
                    continue
            if isinstance(module, SparseConvolution) and FUSE_BN_RELU and i + 1 < len(mods) and \
                    isinstance(input, SparseConvTensor) and input.indices.shape[0] != 0 and \
                    _fusable_bn(mods[i + 1], input.features, channels=module.out_channels):
                # conv -> BatchNorm1d [-> ReLU]: statistics from the conv epilogue, one apply pass; if the next
                # module is again a conv it is the only consumer and reduces this BN's backward sums
                relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
                nxt = i + (3 if relu else 2)
                want_link = nxt < len(mods) and isinstance(mods[nxt], SparseConvolution)
                input, link = conv_bn_relu(module, mods[i + 1], relu, input, link_in=link, want_link=want_link)
                i = nxt
                continue
            if is_spconv_module(module):
                if isinstance(module, SparseConvolution):
                    input = module(input, link=link if FUSE_BN_STATS else None)
                else:
                    input = module(input)
                link = None
            elif isinstance(input, SparseConvTensor):
                link = None
                if input.indices.shape[0] != 0:
                    # BatchNorm1d [+ ReLU] on the active sites -> one fused HIP op
                    if FUSE_BN_RELU and _fusable_bn(module, input.features):
                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        input.features = bn_relu(module, input.features, relu)
                        i += 2 if relu else 1
                        continue
                    if FUSE_BN_RELU and type(module).__name__ == "DistributedSyncBN":
                        # cross-rank BN: the same sweeps split around its packed all-reduce, ReLU folded in
                        from .syncbn import sync_bn_relu

                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        y = sync_bn_relu(module, input.features, relu)
                        if y is not None:
                            input.features = y
                            i += 2 if relu else 1
                            continue
                    input.features = module(input.features)
            else:
                link = None
                input = module(input)
            i += 1
        return input

    # -- one-sync geometry planning ----------------------------------------------
    def conv_layers(self):
        """SparseConvolution modules in execution order (descends into children that
        define `conv_layers`, e.g. residual blocks, and nested SparseSequentials)."""
        out = []
        for m in self._modules.values():
            if isinstance(m, SparseConvolution):
                out.append(m)
            elif hasattr(m, "conv_layers"):
                out.extend(m.conv_layers())
        return out

    def plan(self, x):
        """Build every rulebook this stack needs for input `x` with a single host
        sync: the strided chain is enqueued against capacity-sized buffers with
        device-side counts, the counts are read once, then the submanifold tables
        are enqueued with exact sizes.  Fills x.indice_dict."""
        layers = self.conv_layers()
        dev = x.indices.device
        n0 = x.indices.shape[0]
        # ---- phase A: strided chain
        cur_idx, cur_n_dev, cur_cap, cur_shape = x.indices, _count_tensor(n0, dev), max(n0, 1), list(x.spatial_shape)
        chain = []  # (layer, rb, n_out_dev, level_before)
        levels = [dict(indices=cur_idx, n_dev=cur_n_dev, cap=cur_cap, shape=cur_shape)]
        strided = [m for m in layers if not m.subm]
        ranks = None  # per level >= 1: (bitmap, prefix) pointers of the chain's rank structure, kept alive by `chain_ws`
        if RB_CHAIN and 1 <= len(strided) <= 8:
            # every strided layer in ONE C call: 1 + 4 launches per layer, the row plans included (csrc/d3d_rbchain.hip)
            built, ranks, chain_ws = build_conv_rulebook_chain(cur_idx, cur_n_dev, cur_cap, x.batch_size, cur_shape, strided)
            for m, (rb, n_out_dev) in zip(strided, built):
                chain.append((m, rb, n_out_dev))
                levels.append(dict(indices=rb.out_indices, n_dev=n_out_dev, cap=rb.cap_out, shape=rb.out_shape))
        else:
            for m in strided:
                rb, n_out_dev = build_conv_rulebook(cur_idx, cur_n_dev, None, cur_cap, x.batch_size,
                                                    cur_shape, m.kernel_size, m.stride, m.padding)
                chain.append((m, rb, n_out_dev))
                cur_idx, cur_n_dev, cur_cap, cur_shape = rb.out_indices, n_out_dev, rb.cap_out, rb.out_shape
                levels.append(dict(indices=cur_idx, n_dev=cur_n_dev, cap=cur_cap, shape=cur_shape))
        # ---- the one sync
        if chain:
            counts = torch.cat([c[2] for c in chain]).cpu().tolist()
        else:
            counts = []
        ns = [n0] + [int(c) for c in counts]
        for li, (m, rb, _) in enumerate(chain):
            finalize_conv_rulebook(rb, ns[li], ns[li + 1])
            key = m._rulebook_key()
            x.indice_dict[key] = rb
            levels[li + 1]["indices"] = rb.out_indices
            levels[li + 1]["cap"] = max(ns[li + 1], 1)
        # ---- phase B: submanifold tables with exact sizes
        level = 0
        for m in layers:
            if not m.subm:
                level += 1
                continue
            key = m._rulebook_key()
            if key in x.indice_dict:
                continue
            lv = levels[level]
            n = ns[level]
            if ranks is not None and level >= 1 and n > 0:
                # rows of this level came out of a chain layer in ascending cell order: 27 rank probes per site into
                # that layer's bitmap + prefix instead of a hash table (one launch)
                x.indice_dict[key] = build_subm_rulebook_ranked(lv["indices"], lv["n_dev"], n, x.batch_size, lv["shape"],
                                                                m.kernel_size, ranks[level - 1])
            else:
                x.indice_dict[key] = build_subm_rulebook(lv["indices"], _count_tensor(n, dev), n,
                                                         max(n, 1), x.batch_size, lv["shape"],
                                                         m.kernel_size)
        # ---- phase C: pair-compacted tables of the layers that have that kernel (one launch per table, here on the
        # stream that builds the geometry instead of in front of the first layer that needs them)
        for li, m in enumerate(layers):
            rb = x.indice_dict.get(m._rulebook_key())
            if rb is None or not rb.n_out or not rb.n_in:
                continue
            cin = 16 if (m.in_channels < 16 and m.out_channels % 16 == 0) else m.in_channels
            if pairs_ok(rb, cin, m.out_channels, torch.float32):
                rb.pair_table("out")
            if li > 0 and pairs_ok(rb, m.out_channels, cin, torch.float32, strided_dgrad=not rb.subm):
                rb.pair_table("out" if rb.subm else "in")
        return x.indice_dict


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()
