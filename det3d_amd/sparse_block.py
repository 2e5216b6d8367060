"""SparseBasicBlock (det3d/models/backbones/scn.py:46-89) as ONE autograd node.

    out = relu(bn1(conv1(x)));  out = bn2(conv2(out));  out += identity;  out = relu(out)

The per-layer path ran the residual tail as PyTorch element-wise ops: an add and a clamp in the forward, a
threshold_backward and the gradient accumulation `d(identity) + d(conv1 input)` in the backward -- 33 adds and 9 clamps
per nuScenes-CBGS step, 6 % of it (profiles/round3_kernel_stats_c4.txt) -- plus two `gout.sum(0)` reductions per block
for the conv biases.  Here the block is a plain loop over C-ABI calls inside one `torch.autograd.Function`:

  forward   conv1 (+BN statistics epilogue) -> bn_apply(ReLU) -> conv2 (+statistics) -> bn_apply_residual(x, ReLU)
  backward  bn_bwd_reduce_residual -> bn_bwd_dx_residual (dx and d(identity) = dy * mask in one sweep)
            -> conv2 wgrad, conv2 dgrad (+bn1's backward sums in its epilogue) -> bn_bwd_dx -> conv1 wgrad
            -> conv1 dgrad with D3dConvEpilogue.addend = d(identity): the block's input gradient leaves the kernel
               already summed over both branches

no element-wise PyTorch kernel, no per-layer autograd bookkeeping.  The gradients of the two conv biases are returned
as exact zeros: each bias feeds a train-mode BatchNorm, whose batch mean removes it -- dL/db = sum_i dL/d(conv out)_i
= 0 identically (a BatchNorm's input gradient sums to zero over the batch); what the per-layer path (and the
reference) compute there is the rounding noise of that sum.

D3D_FUSE_BLOCK=0 keeps the per-layer path; eval mode, a downsample branch and channel counts without an MFMA tiling keep
it as well.  Cross-rank BatchNorm (DistributedSyncBN while it exchanges statistics) runs through this node since round 5:
the same kernels, each BatchNorm's moments and backward sums through one packed all-reduce (syncbn.exchange_*); the conv
biases' exact-zero gradients then hold for the SUM over ranks, which is what the data-parallel gradient mean forms.
"""
import ctypes
import os

import torch
from torch import nn

from . import _lib, arena as mem, repack, syncbn

FUSE_BLOCK = os.environ.get("D3D_FUSE_BLOCK", "1") != "0"


def _plain_bn(bn, c):
    """nn.BatchNorm1d in training mode -- or the cross-rank DistributedSyncBN while it exchanges statistics: the node
    then runs the same kernels around syncbn.exchange_moments / exchange_backward_sums"""
    if type(bn) is not nn.BatchNorm1d and not syncbn.fused_sync(bn):
        return False
    if not bn.training or bn.momentum is None or not bn.track_running_stats:
        return False
    if bn.weight is None or bn.num_features != c:
        return False
    return all(t is None or t.dtype == torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))


def rulebook_of(sp, block, x):
    """the submanifold rulebook both convs of `block` share, if the block can run fused on `x`; else None"""
    if not (FUSE_BLOCK and sp.FUSE_BN_RELU and sp.FUSE_BN_STATS) or block.downsample is not None:
        return None
    c1, c2 = block.conv1, block.conv2
    feats = x.features
    sync = syncbn.fused_sync(block.bn1) or syncbn.fused_sync(block.bn2)
    if not (feats.is_cuda and feats.dtype in sp._DT and feats.dim() == 2):
        return None
    if sync and feats.shape[0] < 1:
        # (as in sparse_stack._eligible: with a cross-rank BatchNorm the decision must not depend on this rank's row count)
        raise _lib.D3dError("SparseBasicBlock: this rank has no active sites at a SyncBN layer; its statistics are undefined "
                            "and the other ranks are waiting in the exchange")
    if not sync and feats.shape[0] <= 1:
        return None
    c = c1.in_channels
    if not (c1.subm and c2.subm and c1.out_channels == c and c2.in_channels == c and c2.out_channels == c):
        return None
    if c1._rulebook_key() != c2._rulebook_key() or feats.shape[1] != c:
        return None
    for conv in (c1, c2):
        if conv.weight.dtype != torch.float32 or (conv.bias is not None and conv.bias.dtype != torch.float32):
            return None
    if not (sp.has_epilogue(c, c, feats.dtype) and _plain_bn(block.bn1, c) and _plain_bn(block.bn2, c)):
        return None
    epv = 8 if feats.dtype == torch.bfloat16 else 4
    if c % epv or 256 % (c // epv):
        return None
    rb = x.indice_dict.get(c1._rulebook_key())
    if rb is None or not rb.subm or rb.n_in != feats.shape[0] or rb.n_out != rb.n_in:
        return None
    return rb


class _BlockFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, feats, block, rb, sp, *params):
        L = _lib.lib()
        dev, dtype = feats.device, feats.dtype
        dt = sp._DT[dtype]
        st = _lib.stream_ptr(dev)
        conv1, bn1, conv2, bn2 = block.conv1, block.bn1, block.conv2, block.bn2
        x = feats.contiguous()
        n, c = rb.n_out, conv1.in_channels
        sp.check_gather_source(x, "the feature matrix")
        need_in_grad = ctx.needs_input_grad[0]
        lease = mem.Lease()  # the step arena, if the caller runs under one (arena.py)
        A = lease.arena
        wp1, wd1 = sp._packed_weights(conv1.weight, rb.kvol, c, c, dtype, True, need_in_grad, c)
        wp2, wd2 = sp._packed_weights(conv2.weight, rb.kvol, c, c, dtype, True, True, c)

        def conv(src, wpack, bias):
            out = mem.empty((n, c), dtype, dev, A)
            stats = sp.epilogue_ws(dev, n, c)
            ep = _lib.ConvEpilogue(stats=stats.data_ptr(), ws_floats=stats.numel())
            b = bias.detach() if bias is not None else None
            rc = sp.launch_fwd(rb, src, wpack, b, c, c, out, ep, st)
            _lib.check(rc, "d3d_sparse_conv_fwd")
            return out, stats

        def moments(bn, stats):
            """(training argument of d3d_bn_apply, its statistics, cf): local sums as they are, or -- cross-rank BN --
            the finished mean | variance behind one packed all-reduce"""
            if type(bn) is nn.BatchNorm1d:
                return 1, stats, None
            mv, cf = syncbn.exchange_moments(bn, stats, n)
            return 2, mv, cf

        out1, stats1 = conv(x, wp1, conv1.bias)
        y1 = mem.empty((n, c), dtype, dev, A)
        save1 = torch.empty((2, c), dtype=torch.float32, device=dev)
        mode1, stats1, cf1 = moments(bn1, stats1)
        rc = L.d3d_bn_apply(out1.data_ptr(), n, c, dt, bn1.weight.data_ptr(), _lib.ptr(bn1.bias),
                            bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(), float(bn1.momentum), float(bn1.eps),
                            mode1, 1, stats1.data_ptr(), 1, y1.data_ptr(), save1[0].data_ptr(), save1[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_apply")
        partial1 = sp.epilogue_ws(dev, n, c)  # conv2's dgrad launch reduces bn1's backward sums into this
        sp.check_gather_source(y1, "the feature matrix")
        out2, stats2 = conv(y1, wp2, conv2.bias)
        y = mem.empty((n, c), dtype, dev, A)
        save2 = torch.empty((2, c), dtype=torch.float32, device=dev)
        mode2, stats2, cf2 = moments(bn2, stats2)
        rc = L.d3d_bn_apply_residual(out2.data_ptr(), x.data_ptr(), n, c, dt, bn2.weight.data_ptr(), _lib.ptr(bn2.bias),
                                     bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), float(bn2.momentum),
                                     float(bn2.eps), mode2, 1, stats2.data_ptr(), 1, y.data_ptr(), save2[0].data_ptr(),
                                     save2[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_apply_residual")
        ctx.block, ctx.rb, ctx.sp = block, rb, sp
        ctx.cf = (cf1, cf2)
        # activations through save_for_backward: autograd then refuses a backward after an in-place change of any of them
        # (x is the caller's tensor); the packed weights are not autograd's to version -- repack.check guards them
        ctx.save_for_backward(x, out1, y1, out2, save1, save2)
        ctx.rec = (wd1, wd2, partial1)
        ctx.repack_stamp = repack.stamp((conv1.weight, conv2.weight))
        ctx.backward_runs = 0
        ctx.generation = sp.ARENA.generation
        ctx.lease = lease
        ctx.dtype = dtype
        return y

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        L = _lib.lib()
        sp, block, rb = ctx.sp, ctx.block, ctx.rb
        conv1, bn1, conv2, bn2 = block.conv1, block.bn1, block.conv2, block.bn2
        repack.check(ctx.repack_stamp, "SparseBasicBlock")
        x, out1, y1, out2, save1, save2 = ctx.saved_tensors
        wd1, wd2, partial1 = ctx.rec
        dtype = ctx.dtype
        dt = sp._DT[dtype]
        dev = g.device
        st = _lib.stream_ptr(dev)
        n, c = rb.n_out, conv1.in_channels
        A = ctx.lease.check("SparseBasicBlock")
        dy = g.contiguous()
        if dy.dtype != dtype:
            dy = dy.to(dtype)
        # ---- relu(bn2(.) + x): the mask needs the residual; d(identity) comes out of the same sweep as dx
        partial2 = sp.reduce_ws(dev, n, c, dtype, A)
        rc = L.d3d_bn_bwd_reduce_residual(out2.data_ptr(), x.data_ptr(), dy.data_ptr(), n, c, dt, bn2.weight.data_ptr(),
                                          _lib.ptr(bn2.bias), save2[0].data_ptr(), save2[1].data_ptr(), 1,
                                          partial2.data_ptr(), st)
        _lib.check(rc, "d3d_bn_bwd_reduce_residual")
        cf1, cf2 = ctx.cf
        sync1 = sync2 = None
        if cf2 is not None:  # cross-rank BN: the sums over ranks; d3d_bn_bwd_dx takes their rank means
            partial2, gg, gb = syncbn.exchange_backward_sums(bn2, partial2, cf2, save2[1])
            sync2 = (gg, gb)
        gout2 = mem.empty((n, c), dtype, dev, A)
        dres = mem.empty((n, c), dtype, dev, A)
        gbn2 = torch.empty((2, c), dtype=torch.float32, device=dev)
        rc = L.d3d_bn_bwd_dx_residual(out2.data_ptr(), x.data_ptr(), dy.data_ptr(), n, c, dt, bn2.weight.data_ptr(),
                                      _lib.ptr(bn2.bias), save2[0].data_ptr(), save2[1].data_ptr(), 1,
                                      partial2.data_ptr(), 1, gout2.data_ptr(), dres.data_ptr(), gbn2[0].data_ptr(),
                                      gbn2[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_bwd_dx_residual")
        sp.check_gather_source(gout2, "the output gradient")

        # weight gradients on the second stream (sparse_stack.WGRAD_STREAM): they depend on nothing the data-gradient chain
        # produces afterwards; joined before the gradients are handed back
        from . import sparse_stack
        main = torch.cuda.current_stream(dev)
        side = sparse_stack.wgrad_stream(dev) if sparse_stack.WGRAD_STREAM else None
        keep = []

        def wgrad(src, gout):
            gw = torch.empty((rb.kvol, c, c), dtype=torch.float32, device=dev)
            wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(n, rb.kvol, c, c, dt)
            ws = mem.empty((wsb,), torch.uint8, dev, A)
            launch = lambda s: L.d3d_sparse_conv_wgrad(src.data_ptr(), gout.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n,
                                                       rb.kvol, c, c, dt, gw.data_ptr(), ws.data_ptr(), wsb, s)
            if side is not None:  # `gout` was finished on the main stream just before this call
                side.wait_event(main.record_event())
                keep.append((src, gout, ws, gw))
                rc = sp._timed("wgrad", c, c, n, rb, dev, lambda: launch(side.cuda_stream), stream=side)  # (no stream context: sparse_stack.py)
            else:
                rc = sp._timed("wgrad", c, c, n, rb, dev, lambda: launch(st))
            _lib.check(rc, "d3d_sparse_conv_wgrad")
            return gw

        # ---- conv2: data gradient (reducing bn1's backward sums while the arena slice is still this pass's), weights
        gy1 = mem.empty((n, c), dtype, dev, A)
        # (a second backward over a retained graph takes the stand-alone reduction: the arena slice belongs to the first)
        fused = sp.ARENA.generation == ctx.generation and ctx.backward_runs == 0
        ctx.backward_runs += 1
        ep = None
        if fused:
            ep = _lib.ConvEpilogue(bn_x=out1.data_ptr(), bn_mean=save1[0].data_ptr(), bn_invstd=save1[1].data_ptr(),
                                   bn_gamma=bn1.weight.data_ptr(), bn_beta=_lib.ptr(bn1.bias), bn_relu=1,
                                   bn_partial=partial1.data_ptr(), ws_floats=partial1.numel())
        gw2 = wgrad(y1, gout2)  # (first: on the second stream it then runs next to the data gradient)
        sp.launch_dgrad(rb, gout2, wd2, c, c, gy1, ep, st)
        if not fused:
            partial1 = sp.reduce_ws(dev, n, c, dtype, A)
            rc = L.d3d_bn_bwd_reduce(out1.data_ptr(), gy1.data_ptr(), n, c, dt, bn1.weight.data_ptr(), _lib.ptr(bn1.bias),
                                     save1[0].data_ptr(), save1[1].data_ptr(), 1, partial1.data_ptr(), st)
            _lib.check(rc, "d3d_bn_bwd_reduce")
        if cf1 is not None:
            partial1, gg, gb = syncbn.exchange_backward_sums(bn1, partial1, cf1, save1[1])
            sync1 = (gg, gb)
        gout1 = mem.empty((n, c), dtype, dev, A)
        gbn1 = torch.empty((2, c), dtype=torch.float32, device=dev)
        rc = L.d3d_bn_bwd_dx(out1.data_ptr(), gy1.data_ptr(), n, c, dt, bn1.weight.data_ptr(), _lib.ptr(bn1.bias),
                             save1[0].data_ptr(), save1[1].data_ptr(), 1, partial1.data_ptr(), 1, gout1.data_ptr(),
                             gbn1[0].data_ptr(), gbn1[1].data_ptr(), st)
        _lib.check(rc, "d3d_bn_bwd_dx")
        sp.check_gather_source(gout1, "the output gradient")
        gw1 = wgrad(x, gout1)
        gin = None
        if ctx.needs_input_grad[0]:
            gin = mem.empty((n, c), dtype, dev, A)
            ep = _lib.ConvEpilogue(addend=dres.data_ptr())  # + the gradient that arrived through the identity branch
            sp.launch_dgrad(rb, gout1, wd1, c, c, gin, ep, st)
        if side is not None:
            main.wait_stream(side)
            keep.clear()
        zero = lambda p: torch.zeros_like(p)  # the batch mean removes a bias in front of a train-mode BatchNorm
        grads = []
        for conv, bn, gw, gbn in ((conv1, bn1, gw1, sync1 or gbn1), (conv2, bn2, gw2, sync2 or gbn2)):
            grads.append(gw.reshape(conv.weight.shape))
            if conv.bias is not None:
                grads.append(zero(conv.bias))
            grads.append(gbn[0])
            if bn.bias is not None:
                grads.append(gbn[1])
        return (gin, None, None, None) + tuple(grads)


def _params(block):
    out = []
    for conv, bn in ((block.conv1, block.bn1), (block.conv2, block.bn2)):
        out.append(conv.weight)
        if conv.bias is not None:
            out.append(conv.bias)
        out.append(bn.weight)
        if bn.bias is not None:
            out.append(bn.bias)
    return out


def run(sp, block, x, rb):
    feats = _BlockFn.apply(x.features, block, rb, sp, *_params(block))
    out = sp.SparseConvTensor(feats, x.indices, x.spatial_shape, x.batch_size, x.grid)
    out.indice_dict = x.indice_dict
    return out
