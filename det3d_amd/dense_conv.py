"""Dense channels-last 2D convolution of the RPN on the library's own MFMA kernels (csrc/d3d_conv2d.hip) -- the
host side: operand formats, weight packing, the autograd node.

Reference: det3d/models/necks/rpn.py:70-159 builds nn.Conv2d(3x3) + BatchNorm2d + ReLU stacks and runs them on
cuDNN.  Here a conv layer is `conv2d(x, weight, stride, pad, mode)` on a channels-last map:

  mode "bf16"  : x bf16 [B,H,W,C], packed bf16 weights, bf16 output            (the fast mode)
  mode "split" : x = (hi, lo) bf16 planes with hi + lo = the f32 value to 16 mantissa bits, weights likewise,
                 f32 output: three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi), relative error ~2^-16 -- the
                 precision that meets north_star's 1e-3 box tolerance at bf16-MFMA speed.

Weights keep nn.Conv2d's parameter layout [cout, cin, kh, kw] f32 (checkpoint compatible); the packed forms
[tap][cout][cin] (forward) and [mirrored tap][cin][cout] (data gradient of stride-1 layers) are cached per
parameter version like the sparse conv's.
"""
import ctypes
import weakref

import torch

from . import _lib

_MODES = {"bf16": _lib.D3D_CONV_BF16, "split": _lib.D3D_CONV_SPLIT}


def split_planes(t):
    """f32 tensor -> (hi, lo) bf16 planes, hi = bf16(t), lo = bf16(t - hi); one launch on the GPU"""
    t = t.float()
    if t.is_cuda and t.is_contiguous() and t.numel() % 4 == 0:
        hi = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        lo = torch.empty_like(hi)
        _lib.check(_lib.lib().d3d_split_planes(t.data_ptr(), t.numel(), hi.data_ptr(), lo.data_ptr(),
                                               _lib.stream_ptr(t.device)), "d3d_split_planes")
        return hi, lo
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


def join_planes(hi, lo):
    return hi.float() + lo.float()


_CAPTURE_PACK = {}  # same key -> (weakref, capturing stream handle, None, packs): see _packed
_PACK = {}  # (id(weight), mode) -> (weakref, version, data_ptr, {"fwd": (hi, lo), "dgrad": (hi, lo)})


def _packed(weight, kind, mode):
    """kind 'fwd': [tap][cout][cin]; kind 'dgrad': [mirrored tap][cin][cout] (the forward layout of the transposed
    problem).  Returns (hi, lo-or-None).  Both kinds are produced by ONE launch and cached until the parameter
    changes (optimizer steps bump `_version`; writes through `.data` move no version, so the pointer is checked too)."""
    key = (id(weight), mode)
    hit = _PACK.get(key)
    # under hipGraph capture the pack launch must be PART of the graph (a replay has to see the weights of its own
    # step, and the captured kernels must read capture-owned buffers): one pack per (weight, capture), kept in a side
    # table that dies with the capture -- never the eager cache, whose buffers an eager step would free under the graph
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
    if capturing:
        hit = _CAPTURE_PACK.get(key)
        if hit is not None and hit[0]() is weight and hit[1] == torch.cuda.current_stream(weight.device).cuda_stream:
            return hit[3][kind]
    elif hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr():
        return hit[3][kind]
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    co, ci, kh, kw = w.shape
    dev = w.device
    f_hi = torch.empty((kh * kw, co, ci), dtype=torch.bfloat16, device=dev)
    d_hi = torch.empty((kh * kw, ci, co), dtype=torch.bfloat16, device=dev)
    f_lo = torch.empty_like(f_hi) if mode == "split" else None
    d_lo = torch.empty_like(d_hi) if mode == "split" else None
    rc = _lib.lib().d3d_conv2d_pack_weights(w.data_ptr(), co, ci, kh, f_hi.data_ptr(), _lib.ptr(f_lo), d_hi.data_ptr(),
                                            _lib.ptr(d_lo), _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv2d_pack_weights")
    out = {"fwd": (f_hi, f_lo), "dgrad": (d_hi, d_lo)}
    if capturing:
        # the graph's private memory pool keeps f_* / d_* alive for the graph's lifetime; the entry only serves the
        # second request (fwd, then dgrad) of the same capture
        _CAPTURE_PACK[key] = (weakref.ref(weight), torch.cuda.current_stream(dev).cuda_stream, None, out)
        return out[kind]
    if len(_PACK) > 256:
        for k in [k for k, v in _PACK.items() if v[0]() is None]:
            del _PACK[k]
    _PACK[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), out)
    return out[kind]


def supported(cin, cout, ksz, stride, mode=None):
    """channel pairs / geometries with a library kernel.  mode "split": the 3x3 stride-1 layers also come with 64 output
    channels (conv2d_halo_kernel); their weight gradient then takes wgrad_raw's fallback"""
    L = _lib.lib()
    if mode == "split" and int(ksz) == 3 and int(stride) == 1:
        return bool(L.d3d_conv2d_supported_split3x3(int(cin), int(cout)))
    return bool(L.d3d_conv2d_supported(int(cin), int(cout), int(ksz), int(stride)))


def _out_hw(h, w, ksz, stride, pad):
    return (h + 2 * pad - ksz) // stride + 1, (w + 2 * pad - ksz) // stride + 1


def conv2d_raw(x_hi, x_lo, w_hi, w_lo, ksz, stride, pad, mode, want_stats=False, bn_backward=None):
    """x_hi (and x_lo in split mode): [B,H,W,Cin] bf16 contiguous; w_*: [ksz*ksz, Cout, Cin] bf16.
    Returns (y [B,Ho,Wo,Cout] bf16 | f32, workspace or None: its first 2*Cout floats are sum | sum of squares of y).
    bn_backward = (x, mean, invstd, gamma, beta, relu): a data-gradient launch -- the workspace then starts with
    sum dz*xhat | sum dz of the BatchNorm(+ReLU) whose output fed the convolution (d3d_bn_bwd_reduce's result)."""
    L = _lib.lib()
    b, h, w, cin = x_hi.shape
    cout = w_hi.shape[1]
    ho, wo = _out_hw(h, w, ksz, stride, pad)
    dev = x_hi.device
    y = torch.empty((b, ho, wo, cout), dtype=torch.float32 if mode == "split" else torch.bfloat16, device=dev)
    ws = None
    nws = 0
    bn = None
    if bn_backward is not None:
        bx, bmean, binv, bgamma, bbeta, brelu = bn_backward
        assert tuple(bx.shape) == (b, ho, wo, cout) and bx.dtype == (torch.float32 if mode == "split" else torch.bfloat16)
        bn = _lib.DenseBnBackward(x=bx.data_ptr(), mean=bmean.data_ptr(), invstd=binv.data_ptr(),
                                  gamma=_lib.ptr(bgamma), beta=_lib.ptr(bbeta), relu=int(brelu))
        want_stats = True
    if want_stats:
        nws = L.d3d_conv2d_stats_ws_floats(b, ho, wo, cout)
        if nws == 0:
            raise _lib.D3dError("conv2d statistics epilogue: unsupported shape")
        ws = torch.empty((nws,), dtype=torch.float32, device=dev)
    rc = L.d3d_conv2d_nhwc_fwd(x_hi.data_ptr(), _lib.ptr(x_lo), w_hi.data_ptr(), _lib.ptr(w_lo), y.data_ptr(), b, h, w,
                               cin, cout, ksz, stride, pad, _MODES[mode], _lib.ptr(ws), nws,
                               ctypes.byref(bn) if bn is not None else None, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv2d_nhwc_fwd")
    return y, ws


def dgrad_s2_supported(cin, cout, ksz):
    """stride-2 layers whose data gradient d3d_conv2d_nhwc_dgrad_s2 takes"""
    return int(ksz) == 3 and int(cout) in (128, 256) and int(cin) >= 128 and int(cin) % 128 == 0


def conv2d_dgrad_s2(g_hi, g_lo, wd_hi, wd_lo, h, w, ksz, pad, mode):
    """data gradient of y = conv(x [B,h,w,Cin], stride 2, pad): g_* [B,Ho,Wo,Cout] in the operand format, wd_* = the
    layer's 'dgrad' planes ([mirrored tap][Cin][Cout]); returns dx [B,h,w,Cin] f32 (split) | bf16"""
    L = _lib.lib()
    b, ho, wo, cout = g_hi.shape
    cin = wd_hi.shape[1]
    assert (ho, wo) == _out_hw(h, w, ksz, 2, pad) and wd_hi.shape[2] == cout
    dx = torch.empty((b, h, w, cin), dtype=torch.float32 if mode == "split" else torch.bfloat16, device=g_hi.device)
    rc = L.d3d_conv2d_nhwc_dgrad_s2(g_hi.data_ptr(), _lib.ptr(g_lo), wd_hi.data_ptr(), _lib.ptr(wd_lo), dx.data_ptr(), b, h,
                                    w, cin, cout, ksz, pad, _MODES[mode], _lib.stream_ptr(g_hi.device))
    _lib.check(rc, "d3d_conv2d_nhwc_dgrad_s2")
    return dx


def wgrad_raw(g_hi, g_lo, x_hi, x_lo, ksz, stride, pad, mode):
    """dw [Cout, Cin, ksz, ksz] f32 (nn.Conv2d's layout) from g = dL/dy [B,Ho,Wo,Cout] and the layer input x"""
    L = _lib.lib()
    b, h, w, cin = x_hi.shape
    cout = g_hi.shape[3]
    ho, wo = _out_hw(h, w, ksz, stride, pad)
    assert tuple(g_hi.shape[:3]) == (b, ho, wo)
    dev = x_hi.device
    narrow = cin == 64 and cout == 64 and ksz == 3 and stride == 1 and mode == "split"  # conv2d_wgrad3n_kernel
    if (cin % 128 or cout % 128) and not narrow:
        # other 64-channel pairs (64 -> 128 with stride 1 does not occur in the shipped necks): the f32 values back from
        # their planes, MIOpen's weight gradient on channels-last views of them
        if mode == "split":
            xf, gf = join_planes(x_hi, x_lo), join_planes(g_hi, g_lo)
        else:
            xf, gf = x_hi.float(), g_hi.float()
        return torch.nn.grad.conv2d_weight(xf.permute(0, 3, 1, 2), (cout, cin, ksz, ksz), gf.permute(0, 3, 1, 2),
                                           stride, pad).contiguous()
    dw = torch.empty((cout, cin, ksz, ksz), dtype=torch.float32, device=dev)
    nb = L.d3d_conv2d_wgrad_workspace_bytes(b, ho, wo, cin, cout, ksz)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    rc = L.d3d_conv2d_nhwc_wgrad(g_hi.data_ptr(), _lib.ptr(g_lo), x_hi.data_ptr(), _lib.ptr(x_lo), dw.data_ptr(), b, h,
                                 w, cin, cout, ksz, stride, pad, _MODES[mode], ws.data_ptr(), nb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv2d_nhwc_wgrad")
    return dw


# ---------------------------------------------------------------------------------------------------------------------
# kernel == stride layers of the necks' deblocks (det3d/models/necks/rpn.py:83-100): ConvTranspose2d(k = s) for the
# up-sampling strides, Conv2d(k = s) for the fractional ones.  Neither mixes neighbouring pixels: after a space-to-depth
# (Conv2d) / before a depth-to-space (ConvTranspose2d) re-arrangement they are 1x1 convolutions -- the library's per-pixel
# GEMM kernels with s*s*C channels on one side -- so the reference's cuDNN deconvolution needs no kernel of its own.
KS_CONV = __import__("os").environ.get("D3D_KS_CONV", "1") != "0"
_KS_MAX_C = 512  # widest channel count of the 1x1 kernels (conv2d_nhwc_kernel<512>, conv2d_wgrad_kernel)


def _pack_now(w2d, mode="split"):
    """packed forward / data-gradient planes of a [cout, cin] matrix used as a 1x1 convolution (no cache: the matrix is a
    per-step re-arrangement of the layer's parameter)"""
    co, ci = w2d.shape
    dev = w2d.device
    w = w2d.contiguous()
    f_hi = torch.empty((1, co, ci), dtype=torch.bfloat16, device=dev)
    d_hi = torch.empty((1, ci, co), dtype=torch.bfloat16, device=dev)
    f_lo, d_lo = (torch.empty_like(f_hi), torch.empty_like(d_hi)) if mode == "split" else (None, None)
    rc = _lib.lib().d3d_conv2d_pack_weights(w.data_ptr(), co, ci, 1, f_hi.data_ptr(), _lib.ptr(f_lo), d_hi.data_ptr(),
                                            _lib.ptr(d_lo), _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_conv2d_pack_weights")
    return (f_hi, f_lo), (d_hi, d_lo)


def _ks_groups(s, cout):
    """ConvTranspose2d(k = s): how many of the s output-row phases one 1x1 launch covers (its s * cout channels per
    phase must stay within the widest kernel)"""
    ga = max(1, min(s, _KS_MAX_C // (s * cout)))
    while s % ga:
        ga -= 1
    return ga


def ks_channels_ok(m):
    """the static part of ks_eligible: a kernel == stride layer whose channel counts the 1x1 kernels take on both sides"""
    if type(m) not in (torch.nn.Conv2d, torch.nn.ConvTranspose2d):
        return False
    k, s = m.kernel_size, m.stride
    if k[0] != k[1] or s != k or k[0] < 2 or k[0] > 4 or m.bias is not None or m.groups != 1 or m.dilation != (1, 1):
        return False
    s = s[0]
    cin, cout = m.in_channels, m.out_channels
    wide, narrow = (s * s * cin, cout) if type(m) is torch.nn.Conv2d else (_ks_groups(s, cout) * s * cout, cin)
    return wide in (128, 256, 384, 512) and narrow in (128, 256, 384, 512)


def ks_eligible(m, x):
    """Conv2d / ConvTranspose2d with kernel == stride on a channels-last f32 CUDA map, with channel counts the 1x1 kernels
    take on both sides (128, 256, 384, 512: each side is the input channels of one launch -- forward or data gradient --
    and the output channels of the other; d3d_conv2d_nhwc_fwd's 1x1 layers, d3d_conv2d_nhwc_wgrad)"""
    if not KS_CONV or not ks_channels_ok(m):
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32) or torch.is_autocast_enabled("cuda"):
        return False
    if m.padding != (0, 0) or m.weight.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if type(m) is torch.nn.ConvTranspose2d and m.output_padding != (0, 0):
        return False
    b, c, h, w = x.shape
    s = m.stride[0]
    if c != m.in_channels or (type(m) is torch.nn.Conv2d and (h % s or w % s)):
        return False
    pixels = b * (h // s) * (w // s) if type(m) is torch.nn.Conv2d else b * h * w
    return pixels * s * s * max(m.in_channels, m.out_channels) * 2 < (1 << 31) - 16


def _s2d(rows, s):
    """[B, H, W, C] -> [B, H/s, W/s, (a, b, C)] (a, b = row / column phase)"""
    b, h, w, c = rows.shape
    return rows.reshape(b, h // s, s, w // s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h // s, w // s, s * s * c)


def _d2s(rows, s):
    """inverse of _s2d"""
    b, h, w, cc = rows.shape
    c = cc // (s * s)
    return rows.reshape(b, h, w, s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h * s, w * s, c)


class _KsConvFn(torch.autograd.Function):
    """y = Conv2d(k = s)(x) or ConvTranspose2d(k = s)(x) on [B, C, H, W] channels-last f32 maps, split arithmetic"""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, weight, transposed, s):
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        b, h, w, cin = rows.shape
        ctx.transposed, ctx.s, ctx.in_shape = transposed, s, (b, h, w, cin)
        ctx.need_in_grad = ctx.needs_input_grad[0]
        wt = weight.detach()
        if not transposed:
            cout = wt.shape[0]
            xs = _s2d(rows, s)
            x_hi, x_lo = split_planes(xs)
            w2 = wt.permute(0, 2, 3, 1).reshape(cout, s * s * cin)  # [cout, (a, b, ci)]
            fwd, dgr = _pack_now(w2)
            y, _ = conv2d_raw(x_hi, x_lo, fwd[0], fwd[1], 1, 1, 0, "split")
            ctx.save_for_backward(x_hi, x_lo, dgr[0], dgr[1])
            return y.permute(0, 3, 1, 2)
        cout = wt.shape[1]
        ga = _ks_groups(s, cout)
        x_hi, x_lo = split_planes(rows)
        y = torch.empty((b, h, s, w, s, cout), dtype=torch.float32, device=x.device)  # = [B, H*s, W*s, cout]
        saved = [x_hi, x_lo]
        for a0 in range(0, s, ga):
            w2 = wt[:, :, a0:a0 + ga, :].permute(2, 3, 1, 0).reshape(ga * s * cout, cin)  # [(a, b, co), ci]
            fwd, dgr = _pack_now(w2)
            o, _ = conv2d_raw(x_hi, x_lo, fwd[0], fwd[1], 1, 1, 0, "split")  # [B, H, W, (a, b, co)]
            y[:, :, a0:a0 + ga] = o.reshape(b, h, w, ga, s, cout).permute(0, 1, 3, 2, 4, 5)
            saved += [dgr[0], dgr[1]]
        ctx.save_for_backward(*saved)
        return y.reshape(b, h * s, w * s, cout).permute(0, 3, 1, 2)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        s = ctx.s
        b, h, w, cin = ctx.in_shape
        grow = g.permute(0, 2, 3, 1)
        if grow.dtype != torch.float32:
            grow = grow.float()
        if not ctx.transposed:
            x_hi, x_lo, d_hi, d_lo = ctx.saved_tensors
            cout = grow.shape[3]
            g_hi, g_lo = split_planes(grow.contiguous())
            dw2 = wgrad_raw(g_hi, g_lo, x_hi, x_lo, 1, 1, 0, "split")  # [cout, (a, b, ci), 1, 1]
            dw = dw2.reshape(cout, s, s, cin).permute(0, 3, 1, 2).contiguous()
            dx = None
            if ctx.need_in_grad:
                dxs, _ = conv2d_raw(g_hi, g_lo, d_hi, d_lo, 1, 1, 0, "split")  # [B, H/s, W/s, (a, b, ci)]
                dx = _d2s(dxs, s).permute(0, 3, 1, 2)
            return dx, dw, None, None
        saved = ctx.saved_tensors
        x_hi, x_lo = saved[0], saved[1]
        cout = grow.shape[3]
        ga = _ks_groups(s, cout)
        g6 = grow.reshape(b, h, s, w, s, cout)
        dw = torch.empty((cin, cout, s, s), dtype=torch.float32, device=g.device)
        dx = None
        for gi, a0 in enumerate(range(0, s, ga)):
            gg = g6[:, :, a0:a0 + ga].permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, ga * s * cout)  # [B, H, W, (a, b, co)]
            g_hi, g_lo = split_planes(gg.contiguous())
            dw2 = wgrad_raw(g_hi, g_lo, x_hi, x_lo, 1, 1, 0, "split")  # [(a, b, co), ci, 1, 1]
            dw[:, :, a0:a0 + ga, :] = dw2.reshape(ga, s, cout, cin).permute(3, 2, 0, 1)
            if ctx.need_in_grad:
                d_hi, d_lo = saved[2 + 2 * gi], saved[3 + 2 * gi]
                part, _ = conv2d_raw(g_hi, g_lo, d_hi, d_lo, 1, 1, 0, "split")  # [B, H, W, ci]
                dx = part if dx is None else dx.add_(part)
        return (dx.permute(0, 3, 1, 2) if dx is not None else None), dw, None, None


def ks_conv(m, x):
    """the module's forward on the library's 1x1 kernels (call only when ks_eligible(m, x))"""
    return _KsConvFn.apply(x, m.weight, type(m) is torch.nn.ConvTranspose2d, int(m.stride[0]))
