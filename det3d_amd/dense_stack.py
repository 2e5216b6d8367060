"""Runs of `[ZeroPad2d] -> Conv2d -> BatchNorm2d [-> ReLU]` on a channels-last map as ONE autograd node on the
library's own kernels (csrc/d3d_conv2d.hip + the BatchNorm sweeps of csrc/d3d_bn.hip).

Reference: the RPN blocks and deblocks, det3d/models/necks/rpn.py:58-100 (`Sequential(ZeroPad2d(1), Conv2d(3x3, stride
1 | 2), BN, ReLU, [Conv2d(3x3, padding=1), BN, ReLU] * n)`), cuDNN kernels behind nn.Conv2d / nn.BatchNorm2d.  Per layer:

  forward   conv (BN statistics in its epilogue)  ->  bn_apply (+ReLU), written in the next conv's operand format
  backward  bn_bwd_dx (written in the conv operand format) -> wgrad (two-pass) -> dgrad (= the forward kernel on the
            mirrored taps, with the previous layer's BatchNorm-backward sums in its epilogue; only the last layer
            of a run needs the stand-alone bn_bwd_reduce sweep)

Two precisions, chosen by the dtype of the incoming map:
  bfloat16 (the step runs under autocast)  : bf16 operands and activations, f32 accumulation
  float32                                  : the SPLIT format -- every operand as two bf16 planes hi + lo, three bf16
            MFMAs per product, f32 activations between the kernels: raw box regressions stay within north_star's
            1e-3 of the f32 oracle (tests/test_e2e_gpu.py) at a fraction of the exact-f32 cost
            (D3D_RPN_FP32=exact keeps such maps on MIOpen's f32 kernels).
Everything is deterministic (fixed-order folds, two-pass weight gradients): two identical steps are bit-equal.
"""
import os

import torch
from torch import nn

from . import _lib, repack, syncbn
from . import dense_conv as dc

ENABLED = os.environ.get("D3D_DENSE_CONV", "1") != "0"
FP32_MODE = os.environ.get("D3D_RPN_FP32", "split")  # "split" | "exact" (MIOpen f32)
# BN-backward sums in the dgrad epilogue (split 3x3 layers = conv2d_halo_kernel).  Round 3 measured the epilogue at +35 us
# per launch on the [4, 200, 176, 128] maps against 29 us for the streaming d3d_bn_bwd_reduce it replaces: a lane owns four
# channels of one pixel and re-read x with 16-byte accesses at pixel stride.  Round 6 (last session) stages the block's x tile
# through LDS in whole pixel rows (d3d_conv2d_halo.h): with the weight-gradient stream beside it the step is 1.0-1.3 %
# faster with the sums in the epilogue (670.6 / 671.0 -> 679.5 / 679.4 frames/s, same box, alternating;
# profiles/round6_session4_ab.txt), so it is ON for those layers.  The bf16 and the 1x1 layers (conv2d_nhwc_kernel, whose
# epilogue still reads x at pixel stride: 37 -> 89 us on the 1x1 deblock) keep the sweep.  The forward statistics epilogue
# (+6 us against 11-20 us for d3d_bn_stats) stays on as before.
FUSE_BN_BACKWARD = os.environ.get("D3D_DENSE_FUSE_BN_BWD", "1") == "1"
STATS_EPILOGUE_BF16 = os.environ.get("D3D_DENSE_STATS_EPILOGUE_BF16", "1") == "1"


class _Layer:
    __slots__ = ("conv", "bn", "relu", "ksz", "pad", "stride", "cin", "cout")

    def __init__(self, conv, bn, relu, pad):
        self.conv, self.bn, self.relu = conv, bn, relu
        self.ksz = conv.kernel_size[0]
        self.pad = pad
        self.stride = conv.stride[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels


def _conv_ok(conv, extra_pad, split=False):
    if type(conv) is not nn.Conv2d or conv.bias is not None or conv.groups != 1 or conv.weight.dtype != torch.float32:
        return None
    k = conv.kernel_size
    if k[0] != k[1] or conv.stride not in ((1, 1), (2, 2)) or conv.dilation != (1, 1) or conv.padding_mode != "zeros":
        return None
    if not isinstance(conv.padding, tuple) or conv.padding[0] != conv.padding[1]:
        return None
    pad = conv.padding[0] + extra_pad
    cin, cout = conv.in_channels, conv.out_channels
    if conv.stride == (2, 2):
        # the first convolution of an RPN block (rpn.py:64-68: ZeroPad2d(1) + Conv2d(3, stride 2)): forward and weight
        # gradient on the strided implicit GEMM, data gradient in parity classes (d3d_conv2d_nhwc_dgrad_s2)
        if k[0] != 3 or pad >= 3 or cin not in (128, 256) or cout not in (128, 256):
            return None
        return pad if dc.supported(cin, cout, 3, 2) and dc.dgrad_s2_supported(cin, cout, 3) else None
    if pad >= k[0] or 2 * pad != k[0] - 1:  # "same" convolutions: the data gradient is the mirrored forward
        return None
    if split and k[0] == 3 and cin in (64, 128, 256) and cout in (64, 128, 256):
        # the halo-patch kernel: 64-channel layers too (forward and data gradient; weight gradient: dc.wgrad_raw)
        return pad if dc.supported(cin, cout, 3, 1, "split") else None
    if cin not in (128, 256) or cout not in (128, 256) or not dc.supported(cin, cout, k[0], 1):
        return None
    return pad


def _bn_ok(bn, c):
    # nn.BatchNorm2d, or the cross-rank DistributedSyncBN while it exchanges statistics (the same kernels around one packed
    # all-reduce per direction: syncbn.exchange_moments / exchange_backward_sums)
    if type(bn) is not nn.BatchNorm2d and not syncbn.fused_sync(bn):
        return False
    if bn.momentum is None or not bn.track_running_stats or bn.weight is None:
        return False
    if any(t is not None and t.dtype != torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)):
        return False
    return bn.num_features == c and bn.bias is not None


def collect(mods, i, x):
    """longest run of eligible layers starting at mods[i]; returns (layers, index behind the run)"""
    if not ENABLED or not torch.is_tensor(x) or not x.is_cuda or x.dim() != 4:
        return [], i
    if x.dtype == torch.float32:
        if FP32_MODE != "split" or torch.is_autocast_enabled("cuda"):
            return [], i
    elif x.dtype != torch.bfloat16:
        return [], i
    if x.shape[1] == 1 or not x.is_contiguous(memory_format=torch.channels_last):
        return [], i
    if x.numel() * 2 >= (1 << 31) - 16:
        return [], i
    layers = []
    c = x.shape[1]
    training = None
    L = _lib.lib()
    bsz, hh, ww = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])  # the map entering the next layer
    while i < len(mods):
        j, extra = i, 0
        if type(mods[j]) is nn.ZeroPad2d:
            p = mods[j].padding
            p = (p, p, p, p) if isinstance(p, int) else tuple(p)
            if len(set(p)) != 1 or j + 1 >= len(mods):
                break
            extra, j = p[0], j + 1
        conv = mods[j]
        pad = _conv_ok(conv, extra, x.dtype == torch.float32)
        if pad is None or conv.in_channels != c or j + 1 >= len(mods) or not _bn_ok(mods[j + 1], conv.out_channels):
            break
        bn = mods[j + 1]
        if training is None:
            training = bn.training
        if bn.training != training:
            break
        # what the kernels would refuse later must end the run HERE (the caller then takes the module path): an
        # output map of 2 GB or more, or more pixel tiles than the statistics fold handles
        st = conv.stride[0]
        ho, wo = (hh + 2 * pad - conv.kernel_size[0]) // st + 1, (ww + 2 * pad - conv.kernel_size[1]) // st + 1
        if ho < 1 or wo < 1 or bsz * ho * wo * conv.out_channels * 2 >= (1 << 31) - 16:
            break
        if L.d3d_conv2d_stats_ws_floats(bsz, ho, wo, conv.out_channels) == 0:
            break
        hh, ww = ho, wo
        relu = j + 2 < len(mods) and type(mods[j + 2]) is nn.ReLU
        layers.append(_Layer(conv, bn, relu, pad))
        c = conv.out_channels
        i = j + (3 if relu else 2)
    return layers, i


def _params(layers):
    out = []
    for ly in layers:
        out += [ly.conv.weight, ly.bn.weight, ly.bn.bias]
    return out


class _DenseStackFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, layers, *params):
        L = _lib.lib()
        dev = x.device
        st = _lib.stream_ptr(dev)
        split = x.dtype == torch.float32
        mode = "split" if split else "bf16"
        dt = _lib.D3D_F32 if split else _lib.D3D_BF16
        b, _, h, w = x.shape
        rows = x.permute(0, 2, 3, 1)  # [B,H,W,C] view of the channels-last memory
        if not rows.is_contiguous():
            rows = rows.contiguous()
        a_hi, a_lo = dc.split_planes(rows) if split else (rows, None)
        training = layers[0].bn.training
        need_in_grad = ctx.needs_input_grad[0]
        rec = []
        out = None
        for li, ly in enumerate(layers):
            bn = ly.bn
            w_hi, w_lo = dc._packed(ly.conv.weight, "fwd", mode)
            # BatchNorm statistics from the conv's epilogue.  Alone the bf16 launch pays +22 us for it against 11 us for
            # the stand-alone sweep (profiles/round3_conv2d_variants.txt `fwd_stats_us`), but end to end the step is
            # faster with the epilogue (789 vs 717 frames/s, gpurun_out r3j17 / r3j18): 7 launches and 7 allocations
            # less on a step whose host side is 70 % of its wall time
            epi = training and (split or STATS_EPILOGUE_BF16)
            y, ws = dc.conv2d_raw(a_hi, a_lo, w_hi, w_lo, ly.ksz, ly.stride, ly.pad, mode, want_stats=epi)
            n = y.shape[0] * y.shape[1] * y.shape[2]
            cout = ly.cout
            if training and not epi:
                ws = torch.empty((L.d3d_bn_reduce_ws_floats(n, cout, dt),), dtype=torch.float32, device=dev)
                _lib.check(L.d3d_bn_stats(y.data_ptr(), n, cout, dt, ws.data_ptr(), st), "d3d_bn_stats")
            save = torch.empty((2, cout), dtype=torch.float32, device=dev) if training else None
            last = li + 1 == len(layers)
            common = (n, cout)
            bn_mode, cf = int(training), None
            if training and type(bn) is not nn.BatchNorm2d:  # DistributedSyncBN: moments through one packed all-reduce
                ws, cf = syncbn.exchange_moments(bn, ws, n)
                bn_mode = 2
            tail = (bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                    float(bn.momentum), float(bn.eps), bn_mode, int(ly.relu), _lib.ptr(ws), 1)
            if split and not last:
                z_hi = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
                z_lo = torch.empty_like(z_hi)
                rc = L.d3d_bn_apply_split(y.data_ptr(), *common, *tail, z_hi.data_ptr(), z_lo.data_ptr(),
                                          _lib.ptr(save[0]) if training else None,
                                          _lib.ptr(save[1]) if training else None, st)
                _lib.check(rc, "d3d_bn_apply_split")
                nxt = (z_hi, z_lo)
            else:
                z = torch.empty_like(y)
                rc = L.d3d_bn_apply(y.data_ptr(), *common, dt, *tail, z.data_ptr(),
                                    _lib.ptr(save[0]) if training else None,
                                    _lib.ptr(save[1]) if training else None, st)
                _lib.check(rc, "d3d_bn_apply")
                nxt = (z, None)
                if last:
                    out = z
            rec.append((a_hi, a_lo, y, save, cf))
            a_hi, a_lo = nxt
        ctx.layers, ctx.rec, ctx.mode, ctx.training = layers, rec, mode, training
        ctx.repack_stamp = repack.stamp([ly.conv.weight for ly in layers])  # the backward reads the layers' cached packs, rewritten in place by Repacker.run()
        ctx.in_dtype = x.dtype
        ctx.need_in_grad = need_in_grad
        return out.permute(0, 3, 1, 2)  # channels-last [B, C, H, W]

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        if not ctx.training:
            raise _lib.D3dError("dense conv stack: backward is implemented for training-mode BatchNorm only")
        L = _lib.lib()
        repack.check(ctx.repack_stamp, "dense conv stack")
        layers, rec, mode = ctx.layers, ctx.rec, ctx.mode
        split = mode == "split"
        dt = _lib.D3D_F32 if split else _lib.D3D_BF16
        tdt = torch.float32 if split else torch.bfloat16
        dev = g.device
        st = _lib.stream_ptr(dev)
        dy = g.permute(0, 2, 3, 1)
        if dy.dtype != tdt:
            dy = dy.to(tdt)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        grads = []
        partial = None  # BN-backward sums of layer li, left behind by the dgrad epilogue of layer li + 1
        # weight gradients on the second stream (sparse_stack.WGRAD_STREAM): they fill the partial last round of the data
        # gradient's blocks (572 work items on 256 CUs) and its statistics epilogue
        from . import sparse_stack
        main = torch.cuda.current_stream(dev)
        from . import runtime
        side = sparse_stack.wgrad_stream(dev) if sparse_stack.WGRAD_STREAM and runtime.wgrad_side("dense") and len(layers) > 1 else None
        keep = []  # the gradient planes the side stream reads stay referenced until it has been joined
        defer = side is not None and runtime.wgrad_after_dgrad("dense")
        for li in range(len(layers) - 1, -1, -1):
            ly = layers[li]
            bn = ly.bn
            a_hi, a_lo, y, save, cf = rec[li]
            n, cout = y.shape[0] * y.shape[1] * y.shape[2], ly.cout
            if partial is None:
                partial = torch.empty((L.d3d_bn_reduce_ws_floats(n, cout, dt),), dtype=torch.float32, device=dev)
                rc = L.d3d_bn_bwd_reduce(y.data_ptr(), dy.data_ptr(), n, cout, dt, bn.weight.data_ptr(),
                                         bn.bias.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), int(ly.relu),
                                         partial.data_ptr(), st)
                _lib.check(rc, "d3d_bn_bwd_reduce")
            sync = None
            if cf is not None:  # cross-rank BN: the sums over ranks; the dx sweep takes their rank means
                partial, gg, gb = syncbn.exchange_backward_sums(bn, partial, cf, save[1])
                sync = (gg, gb)
            gbn = torch.empty((2, cout), dtype=torch.float32, device=dev)
            if split:
                g_hi = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
                g_lo = torch.empty_like(g_hi)
                rc = L.d3d_bn_bwd_dx_split(y.data_ptr(), dy.data_ptr(), n, cout, bn.weight.data_ptr(),
                                           bn.bias.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), int(ly.relu),
                                           partial.data_ptr(), 1, g_hi.data_ptr(), g_lo.data_ptr(), gbn[0].data_ptr(),
                                           gbn[1].data_ptr(), st)
                _lib.check(rc, "d3d_bn_bwd_dx_split")
            else:
                g_hi, g_lo = torch.empty_like(y), None
                rc = L.d3d_bn_bwd_dx(y.data_ptr(), dy.data_ptr(), n, cout, dt, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                     save[0].data_ptr(), save[1].data_ptr(), int(ly.relu), partial.data_ptr(), 1,
                                     g_hi.data_ptr(), gbn[0].data_ptr(), gbn[1].data_ptr(), st)
                _lib.check(rc, "d3d_bn_bwd_dx")
            def side_wgrad():
                side.wait_event(main.record_event())
                keep.append((g_hi, g_lo))
                with torch.cuda.stream(side):  # (its workspace comes from, and returns to, the side stream's pool)
                    w = dc.wgrad_raw(g_hi, g_lo, a_hi, a_lo, ly.ksz, ly.stride, ly.pad, mode)
                w.record_stream(main)
                return w

            gw = None
            if side is None:
                gw = dc.wgrad_raw(g_hi, g_lo, a_hi, a_lo, ly.ksz, ly.stride, ly.pad, mode)
            elif not defer:
                gw = side_wgrad()
            partial = None
            if li > 0 or ctx.need_in_grad:
                wd_hi, wd_lo = dc._packed(ly.conv.weight, "dgrad", mode)
                if ly.stride == 2:  # the taps that reach each parity class of the input map (no epilogue)
                    dy = dc.conv2d_dgrad_s2(g_hi, g_lo, wd_hi, wd_lo, a_hi.shape[1], a_hi.shape[2], ly.ksz, ly.pad, mode)
                else:
                    bnb = None
                    if li > 0 and FUSE_BN_BACKWARD and split and ly.ksz == 3:
                        # the rows this launch produces are dL/dz of layer li - 1's BatchNorm + ReLU: reduce its sums here
                        pl = layers[li - 1]
                        _, _, py, psave, _ = rec[li - 1]
                        bnb = (py, psave[0], psave[1], pl.bn.weight, pl.bn.bias, pl.relu)
                    dy, partial = dc.conv2d_raw(g_hi, g_lo, wd_hi, wd_lo, ly.ksz, 1, ly.ksz - 1 - ly.pad, mode,
                                                bn_backward=bnb)
            else:
                dy = None
            if gw is None:
                # wgrad_after_dgrad: the weight gradient starts when this layer's data gradient has FINISHED, i.e. next to
                # the BatchNorm sweeps of the layer below (HBM-bound) instead of next to its own data gradient (MFMA-bound
                # like itself: the two only time-share the CUs, and the sweeps then run with the second stream idle)
                gw = side_wgrad()
            if sync is not None:
                gbn = sync  # grad_gamma / grad_beta summed over ranks (syncbn.py:99-103)
            grads.append([gw, gbn[0], gbn[1]])
        if side is not None:
            main.wait_stream(side)
            keep.clear()
        gin = None
        if dy is not None:
            gin = dy.permute(0, 3, 1, 2)
            if gin.dtype != ctx.in_dtype:
                gin = gin.to(ctx.in_dtype)
        flat = [t for lg in reversed(grads) for t in lg]
        return (gin, None) + tuple(flat)


def run(layers, x):
    return _DenseStackFn.apply(x, layers, *_params(layers))
