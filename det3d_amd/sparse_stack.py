"""Runs of `SparseConvolution -> BatchNorm1d [-> ReLU]` layers as ONE autograd node.

scn.py builds its backbones as long SparseSequentials of exactly these triples (SpMiddleFHD: 14 of them,
det3d/models/backbones/scn.py:106-157).  Going through one `torch.autograd.Function` per conv and one per
BatchNorm costs the host ~140 us per layer in the forward and ~100 us in the backward (Python wrappers, autograd
graph nodes, SparseConvTensor objects) -- 2.0 + 1.5 ms of a 5.7 ms step, more than the GPU needs for those
layers, i.e. the step was host-bound (profiles/round2_host_profile.txt).  Here the layer loop is a plain Python
loop over C-ABI calls inside a single Function: the same kernels with the same arguments in the same order
(conv forward with the BN-statistics epilogue, BN apply, and in the backward BN dx, conv dgrad with the
BN-backward-sums epilogue for the layer in front, conv wgrad), no per-layer autograd bookkeeping.

`run(seq, x)` is called by SparseSequential.forward; layers that do not fit the pattern (residual blocks -- they have
their own node, sparse_block.py --, eval mode, empty tensors, VALU-fallback channel counts) keep the per-layer path.
Round 5: the BatchNorm may be the cross-rank DistributedSyncBN (`norm_cfg = dict(type="SyncBN")`,
det3d/ops/syncbn/syncbn.py:15-118): same kernels, the two [2C] vectors of a layer (moments forward, backward sums) take
one packed all-reduce each (syncbn.exchange_moments / exchange_backward_sums, d3d_bn_apply with training = 2).
D3D_FUSE_STACK=0 disables the node.
"""
import ctypes
import os

import torch
from torch import nn

from . import _lib, arena as mem, repack, syncbn

FUSE_STACK = os.environ.get("D3D_FUSE_STACK", "1") != "0"
# weight gradients of the stack's backward on a second stream (round 5): a layer's weight gradient depends on nothing the
# data-gradient chain produces after it, and every sparse kernel leaves CUs idle in its ramp and tail (its waves are resident
# 61-63 % of its duration: profiles/round5_fwd_anatomy.txt) -- the weight gradient of layer l fills them while the chain goes
# on with the BatchNorm sweeps and data gradient of layer l - 1.  Same kernels, same arguments: bit-identical gradients.
from . import runtime  # noqa: E402  (the authoritative switches; the module variable below is what the loops read)

WGRAD_STREAM = runtime.get("wgrad_stream") != "off"
_SIDE = {}


def wgrad_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE:
        # D3D_WGRAD_PRIORITY: the chain on the main stream is the critical path; a lower priority (1) lets its blocks go first
        _SIDE[key] = torch.cuda.Stream(device=dev, priority=int(runtime.get("wgrad_priority")))
    return _SIDE[key]


class _Layer:
    __slots__ = ("conv", "bn", "relu", "rb", "cin", "cout", "pad_cin")

    def __init__(self, conv, bn, relu, rb):
        self.conv, self.bn, self.relu, self.rb = conv, bn, relu, rb
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.pad_cin = 16 if (self.cin < 16 and self.cout % 16 == 0) else self.cin


def _eligible(sp, conv, bn, feats_dtype, rb):
    if conv.weight.dtype != torch.float32 or (conv.bias is not None and conv.bias.dtype != torch.float32):
        return False
    cin, cout = conv.in_channels, conv.out_channels
    pad_cin = 16 if (cin < 16 and cout % 16 == 0) else cin
    if not sp.has_epilogue(pad_cin, cout, feats_dtype):
        return False  # VALU fallback kernels have no epilogues
    if rb is None or rb.subm != conv.subm:
        return False
    sync = syncbn.fused_sync(bn)  # cross-rank BN: the same kernels around two packed all-reduces, syncbn.exchange_*
    if type(bn) is not nn.BatchNorm1d and not sync:
        return False
    if sync:
        # The decision must be the SAME ON EVERY RANK: a rank that left the fused node would issue its
        # SyncBN all-reduces layer by layer, with the gradient buckets' all-reduces in between, while the others issue
        # them after the whole stack -- different collective orders on one process group.  So nothing rank-local may
        # decline here; a rank without active sites at a cross-rank BatchNorm is an error (the reference divides by the
        # local count as well: det3d/ops/syncbn/syncbn.py:39-46).
        if rb.n_out < 1 or rb.n_in < 1:
            raise _lib.D3dError("sparse stack: this rank has no active sites (%d -> %d rows) at a SyncBN layer; its "
                                "statistics are undefined and the other ranks are waiting in the exchange" % (rb.n_in, rb.n_out))
    elif rb.n_out <= 1 or rb.n_in <= 0:
        return False
    if not bn.training or bn.momentum is None or not bn.track_running_stats:
        return False
    if bn.weight is None or bn.num_features != cout:
        return False
    if any(t is not None and t.dtype != torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)):
        return False
    epv = 8 if feats_dtype == torch.bfloat16 else 4
    return cout % epv == 0 and 256 % (cout // epv) == 0


def segment(sp, mods, start, x):
    """the longest run of fusable (conv, bn[, relu]) triples beginning at mods[start]; returns (layers, next index)"""
    layers = []
    i = start
    n_in = x.indices.shape[0]
    while i + 1 < len(mods) and isinstance(mods[i], sp.SparseConvolution):
        conv, bn = mods[i], mods[i + 1]
        rb = x.indice_dict.get(conv._rulebook_key())
        if rb is None or rb.n_in != n_in or not _eligible(sp, conv, bn, x.features.dtype, rb):
            break
        relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
        layers.append(_Layer(conv, bn, relu, rb))
        n_in = rb.n_out
        i += 3 if relu else 2
    return layers, i


class _StackFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, feats, layers, sp, *params):
        L = _lib.lib()
        dev = feats.device
        dtype = feats.dtype
        dt = sp._DT[dtype]
        st = _lib.stream_ptr(dev)
        arena = sp.ARENA
        lease = mem.Lease()  # the step arena (arena.py) if the caller runs under one: the backward allocates from it too
        A = lease.arena
        feats = feats.contiguous()
        need_in_grad = ctx.needs_input_grad[0]
        rec = []
        x = feats
        ctx.desc = None
        if _c_executor_ok(sp, layers):
            # one foreign call for the whole run (csrc/d3d_stack.hip): this loop only allocates and fills descriptors
            desc = (_lib.StackLayer * len(layers))()
            for li, ly in enumerate(layers):
                rb, conv, bn = ly.rb, ly.conv, ly.bn
                cin, cout, pad_cin = ly.cin, ly.cout, ly.pad_cin
                if x.shape[0] != rb.n_in or x.shape[1] != cin:
                    raise _lib.D3dError("sparse stack: features %s do not match the rulebook (%d rows) / the conv "
                                        "(%d channels)" % (tuple(x.shape), rb.n_in, cin))
                if pad_cin != cin:
                    x = torch.nn.functional.pad(x, (0, pad_cin - cin))
                sp.check_gather_source(x, "the feature matrix")
                wpack, wd = sp._packed_weights(conv.weight, rb.kvol, cin, cout, dtype, rb.subm,
                                               li > 0 or need_in_grad, pad_cin)
                out = mem.empty((rb.n_out, cout), dtype, dev, A)
                stats = sp.epilogue_ws(dev, rb.n_out, cout)
                y = mem.empty((rb.n_out, cout), dtype, dev, A)
                save = torch.empty((2, cout), dtype=torch.float32, device=dev)
                partial = sp.epilogue_ws(dev, rb.n_out, cout) if li + 1 < len(layers) else None
                d = desc[li]
                d.nbr_out, d.cap_out = rb.nbr_out.data_ptr(), rb.cap_out
                if not rb.subm:
                    d.nbr_in, d.cap_in = rb.nbr_in.data_ptr(), rb.cap_in
                    if rb.dgrad_plan is not None:
                        d.dgrad_plan = rb.dgrad_plan.data_ptr()
                d.n_in, d.n_out, d.kvol, d.subm = rb.n_in, rb.n_out, rb.kvol, int(rb.subm)
                d.cin, d.cout, d.relu = pad_cin, cout, int(ly.relu)
                d.momentum, d.eps = float(bn.momentum), float(bn.eps)
                d.x, d.out, d.y = x.data_ptr(), out.data_ptr(), y.data_ptr()
                d.wpack = wpack.data_ptr()
                d.wd = wd.data_ptr() if wd is not None else None
                d.stats, d.stats_floats = stats.data_ptr(), stats.numel()
                d.gamma, d.beta = bn.weight.data_ptr(), _lib.ptr(bn.bias)
                d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                sm = save.data_ptr()
                d.save_mean, d.save_invstd = sm, sm + 4 * cout
                rec.append((x, out, save, wd, partial, None))
                x = y
            _lib.check(L.d3d_sparse_stack_forward(desc, len(layers), dt, st), "d3d_sparse_stack_forward")
            ctx.desc = desc
        for li, ly in enumerate(layers if ctx.desc is None else ()):  # call by call (what the executor cannot take)
            rb, conv, bn = ly.rb, ly.conv, ly.bn
            cin, cout, pad_cin = ly.cin, ly.cout, ly.pad_cin
            if x.shape[0] != rb.n_in or x.shape[1] != cin:
                raise _lib.D3dError("sparse stack: features %s do not match the rulebook (%d rows) / the conv "
                                    "(%d channels)" % (tuple(x.shape), rb.n_in, cin))
            if pad_cin != cin:
                x = torch.nn.functional.pad(x, (0, pad_cin - cin))
            sp.check_gather_source(x, "the feature matrix")
            wpack, wd = sp._packed_weights(conv.weight, rb.kvol, cin, cout, dtype, rb.subm,
                                           li > 0 or need_in_grad, pad_cin)
            out = mem.empty((rb.n_out, cout), dtype, dev, A)
            stats = sp.epilogue_ws(dev, rb.n_out, cout)
            ep = _lib.ConvEpilogue(stats=stats.data_ptr(), ws_floats=stats.numel())
            b = conv.bias.detach() if conv.bias is not None else None
            rc = sp.launch_fwd(rb, x, wpack, b, pad_cin, cout, out, ep, st)
            _lib.check(rc, "d3d_sparse_conv_fwd")
            y = mem.empty((rb.n_out, cout), dtype, dev, A)
            save = torch.empty((2, cout), dtype=torch.float32, device=dev)
            mode, cf = 1, None
            if type(bn) is not nn.BatchNorm1d:  # DistributedSyncBN: the moments travel through one packed all-reduce
                stats, cf = syncbn.exchange_moments(bn, stats, rb.n_out)
                mode = 2
            rc = L.d3d_bn_apply(out.data_ptr(), rb.n_out, cout, dt, bn.weight.data_ptr(), _lib.ptr(bn.bias),
                                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum),
                                float(bn.eps), mode, int(ly.relu), stats.data_ptr(), 1, y.data_ptr(),
                                save[0].data_ptr(), save[1].data_ptr(), st)
            _lib.check(rc, "d3d_bn_apply")
            # the conv behind this layer reduces this BatchNorm's backward sums in its dgrad epilogue
            partial = sp.epilogue_ws(dev, rb.n_out, cout) if li + 1 < len(layers) else None
            rec.append((x, out, save, wd, partial, cf))
            x = y
        ctx.layers, ctx.rec, ctx.sp = layers, rec, sp
        ctx.repack_stamp = repack.stamp([ly.conv.weight for ly in layers])  # the packed weights in `rec` are rewritten in place by Repacker.run()
        ctx.generation = arena.generation
        ctx.lease = lease
        ctx.dtype = dtype
        return x

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        L = _lib.lib()
        repack.check(ctx.repack_stamp, "sparse conv stack")
        sp, layers, rec = ctx.sp, ctx.layers, ctx.rec
        dtype = ctx.dtype
        dt = sp._DT[dtype]
        dev = g.device
        st = _lib.stream_ptr(dev)
        A = ctx.lease.check("sparse conv stack")
        # arena slices are only ours until the next ARENA.reset(): a second forward in between -> separate sweeps
        fused = sp.ARENA.generation == ctx.generation
        grads = []  # per layer, in parameter order
        reduced_next = False  # the previous iteration's dgrad filled rec[li][4]
        main = torch.cuda.current_stream(dev)
        side = wgrad_stream(dev) if WGRAD_STREAM and len(layers) > 1 else None
        keep = []  # what the side stream reads / writes stays referenced until it has been joined
        defer = side is not None and runtime.wgrad_after_dgrad("sparse")
        if ctx.desc is not None and fused and not defer and _c_executor_ok(sp, layers):
            return _StackFn._backward_c(ctx, g, L, sp, layers, rec, dtype, dt, dev, st, A, side)
        for li in range(len(layers) - 1, -1, -1):
            ly = layers[li]
            rb, conv, bn = ly.rb, ly.conv, ly.bn
            cin, cout, pad_cin = ly.cin, ly.cout, ly.pad_cin
            x, out, save, wd, partial, cf = rec[li]
            dy = g.contiguous()
            if dy.dtype != dtype:
                dy = dy.to(dtype)
            if not reduced_next:
                partial = sp.reduce_ws(dev, rb.n_out, cout, dtype, A)
                rc = L.d3d_bn_bwd_reduce(out.data_ptr(), dy.data_ptr(), rb.n_out, cout, dt, bn.weight.data_ptr(),
                                         _lib.ptr(bn.bias), save[0].data_ptr(), save[1].data_ptr(), int(ly.relu),
                                         partial.data_ptr(), st)
                _lib.check(rc, "d3d_bn_bwd_reduce")
            sync = None
            if cf is not None:  # cross-rank BN: sum the two vectors over ranks, hand d3d_bn_bwd_dx the rank means
                partial, g_gamma, g_beta = syncbn.exchange_backward_sums(bn, partial, cf, save[1])
                sync = (g_gamma, g_beta)
            gout = mem.empty((rb.n_out, cout), dtype, dev, A)
            sp.check_gather_source(gout, "the output gradient")
            gbn = torch.empty((2, cout), dtype=torch.float32, device=dev)
            # (everything the side stream will touch is allocated BEFORE `gout_ready` is recorded: a block the main
            # stream's pool recycles after that event could still be in use by an earlier main-stream kernel)
            gw = torch.empty((rb.kvol, pad_cin, cout), dtype=torch.float32, device=dev)
            wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(rb.n_out, rb.kvol, pad_cin, cout, dt)
            ws = mem.empty((wsb,), torch.uint8, dev, A)
            rc = L.d3d_bn_bwd_dx(out.data_ptr(), dy.data_ptr(), rb.n_out, cout, dt, bn.weight.data_ptr(),
                                 _lib.ptr(bn.bias), save[0].data_ptr(), save[1].data_ptr(), int(ly.relu),
                                 partial.data_ptr(), 1, gout.data_ptr(), gbn[0].data_ptr(), gbn[1].data_ptr(), st)
            _lib.check(rc, "d3d_bn_bwd_dx")
            gout_ready = main.record_event() if side is not None else None
            # ---- conv backward
            gin = None
            reduced_next = False
            if li > 0 or ctx.needs_input_grad[0]:
                gin = mem.empty((rb.n_in, pad_cin), dtype, dev, A)
                ep = None
                if li > 0 and fused and pad_cin == cin and sp.has_epilogue(cout, pad_cin, dtype):
                    pl = layers[li - 1]
                    _, pout, psave, _, ppartial, _ = rec[li - 1]
                    ep = _lib.ConvEpilogue(bn_x=pout.data_ptr(), bn_mean=psave[0].data_ptr(),
                                           bn_invstd=psave[1].data_ptr(), bn_gamma=pl.bn.weight.data_ptr(),
                                           bn_beta=_lib.ptr(pl.bn.bias), bn_relu=int(pl.relu),
                                           bn_partial=ppartial.data_ptr(), ws_floats=ppartial.numel())
                    reduced_next = True
                sp.launch_dgrad(rb, gout, wd, cout, pad_cin, gin, ep, st)
                if pad_cin != cin:
                    gin = gin[:, :cin].contiguous()
            if side is not None:  # (buffers from the main stream's pool; `gout` was finished on the main stream)
                if defer:  # runtime.wgrad_after_dgrad: start when this layer's data gradient has finished
                    gout_ready = main.record_event()
                side.wait_event(gout_ready)
                keep.append((gout, ws, gw, x))
                # (the launch takes the side stream's handle; no `with torch.cuda.stream(side)`: nothing is allocated in here,
                # and entering / leaving that context costs the host ~12 us per layer)
                rc = sp._timed("wgrad", pad_cin, cout, rb.n_out, rb, dev, lambda: L.d3d_sparse_conv_wgrad(
                    x.data_ptr(), gout.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, rb.n_out, rb.kvol, pad_cin, cout,
                    dt, gw.data_ptr(), ws.data_ptr(), wsb, side.cuda_stream), stream=side)
            else:
                rc = sp._timed("wgrad", pad_cin, cout, rb.n_out, rb, dev, lambda: L.d3d_sparse_conv_wgrad(
                    x.data_ptr(), gout.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, rb.n_out, rb.kvol, pad_cin, cout,
                    dt, gw.data_ptr(), ws.data_ptr(), wsb, st))
            _lib.check(rc, "d3d_sparse_conv_wgrad")
            if pad_cin != cin:
                gw = gw[:, :cin]
            lg = [gw.reshape(conv.weight.shape)]
            if conv.bias is not None:
                lg.append(gout.float().sum(0))
            if sync is not None:
                gbn = sync  # grad_gamma / grad_beta summed over ranks (syncbn.py:99-103)
            lg.append(gbn[0])
            if bn.bias is not None:
                lg.append(gbn[1])
            grads.append(lg)
            g = gin
        if side is not None:
            main.wait_stream(side)  # the weight gradients are the caller's from here on
            keep.clear()
        flat = [t for lg in reversed(grads) for t in lg]
        return (g, None, None) + tuple(flat)


def _c_executor_ok(sp, layers):
    """may this run go through d3d_sparse_stack_forward / _backward?  Not with a collective between the launches
    (DistributedSyncBN), convolution biases, pair-compacted kernels or per-launch timing events (bench.py's sampled step)."""
    if runtime.get("stack_executor") != "c" or sp.LAUNCH_LOG is not None or sp.PAIR_KERNELS != "off":
        return False
    return all(type(ly.bn) is nn.BatchNorm1d and ly.conv.bias is None for ly in layers)


def _backward_c(ctx, g, L, sp, layers, rec, dtype, dt, dev, st, A, side):
    """the backward of _StackFn as one call: allocate every buffer, complete the forward's descriptors, d3d_sparse_stack_backward"""
    desc = ctx.desc
    n = len(layers)
    dy = g.contiguous()
    if dy.dtype != dtype:
        dy = dy.to(dtype)
    bufs = []
    for li in range(n - 1, -1, -1):
        ly = layers[li]
        rb, cin, cout, pad_cin = ly.rb, ly.cin, ly.cout, ly.pad_cin
        d = desc[li]
        gout = mem.empty((rb.n_out, cout), dtype, dev, A)
        sp.check_gather_source(gout, "the output gradient")
        gbn = torch.empty((2, cout), dtype=torch.float32, device=dev)
        gw = torch.empty((rb.kvol, pad_cin, cout), dtype=torch.float32, device=dev)
        wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(rb.n_out, rb.kvol, pad_cin, cout, dt)
        ws = mem.empty((wsb,), torch.uint8, dev, A)
        gin = None
        if li > 0 or ctx.needs_input_grad[0]:
            gin = mem.empty((rb.n_in, pad_cin), dtype, dev, A)
            if li > 0 and pad_cin == cin and sp.has_epilogue(cout, pad_cin, dtype):
                p = desc[li - 1]  # this layer's data gradient reduces the BatchNorm sums of the layer below
                ppartial = rec[li - 1][4]
                p.partial_by_epilogue, p.partial, p.partial_floats = 1, ppartial.data_ptr(), ppartial.numel()
        if not d.partial_by_epilogue:
            partial = sp.reduce_ws(dev, rb.n_out, cout, dtype, A)
            d.partial, d.partial_floats = partial.data_ptr(), partial.numel()
            bufs.append(partial)
        d.gout, d.gbn, d.gw = gout.data_ptr(), gbn.data_ptr(), gw.data_ptr()
        d.gin = gin.data_ptr() if gin is not None else None
        d.wgrad_ws, d.wgrad_ws_bytes = ws.data_ptr(), wsb
        bufs.append((gout, gbn, gw, ws, gin))
    # (every buffer the second stream touches exists before the call records its first event, see backward())
    rc = L.d3d_sparse_stack_backward(desc, n, dt, dy.data_ptr(), st, side.cuda_stream if side is not None else None)
    _lib.check(rc, "d3d_sparse_stack_backward")
    grads, gfirst = [], None
    for item in bufs:
        if not isinstance(item, tuple):
            continue
        li = n - 1 - len(grads)
        ly = layers[li]
        gout, gbn, gw, ws, gin = item
        if ly.pad_cin != ly.cin:
            gw = gw[:, :ly.cin]
            if gin is not None:
                gin = gin[:, :ly.cin].contiguous()
        lg = [gw.reshape(ly.conv.weight.shape), gbn[0]]
        if ly.bn.bias is not None:
            lg.append(gbn[1])
        grads.append(lg)
        if li == 0:
            gfirst = gin
    flat = [t for lg in reversed(grads) for t in lg]
    return (gfirst, None, None) + tuple(flat)


_StackFn._backward_c = staticmethod(_backward_c)


def params_of(layers):
    out = []
    for ly in layers:
        out.append(ly.conv.weight)
        if ly.conv.bias is not None:
            out.append(ly.conv.bias)
        out.append(ly.bn.weight)
        if ly.bn.bias is not None:
            out.append(ly.bn.bias)
    return out


def run(sp, layers, x):
    """x: SparseConvTensor entering the first layer of `layers`; returns the SparseConvTensor after the last one."""
    feats = _StackFn.apply(x.features, layers, sp, *params_of(layers))
    last = layers[-1]
    shape = x.spatial_shape
    for ly in layers:
        shape = ly.conv.out_spatial_shape(shape)
    indices = x.indices
    for ly in layers:
        if not ly.conv.subm:
            indices = ly.rb.out_indices
    out = sp.SparseConvTensor(feats, indices, shape, x.batch_size, x.grid)
    out.indice_dict = x.indice_dict
    return out
