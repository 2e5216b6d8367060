"""ctypes binding of libd3d_hip.so -- the C ABI declared in include/d3d.h.

There is no CPU fallback: if the library is missing or a call fails the caller
gets an exception.  PyTorch is only used by callers for device memory and
streams; every argument crossing this boundary is a raw pointer or an int.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd3d_hip.so")

_lib = None

vp = ctypes.c_void_p
i32 = ctypes.c_int
i64 = ctypes.c_int64
f32 = ctypes.c_float
sz = ctypes.c_size_t
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)

# name -> (restype, argtypes); mirrors include/d3d.h one to one
SIGNATURES = {
    "d3d_abi_version": (i32, []),
    "d3d_last_error": (ctypes.c_char_p, []),
    "d3d_voxelize_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "d3d_voxelize": (i32, [vp, i64, i32, vp, i32, _f32p, _f32p, _i32p, i32, i32, vp, vp, vp, vp,
                           i32, vp, vp, sz, vp]),
    "d3d_subm_rulebook_workspace_bytes": (sz, [i32]),
    "d3d_subm_rulebook": (i32, [vp, vp, i32, i32, _i32p, _i32p, vp, vp, vp, sz, vp]),
    "d3d_conv_rulebook_workspace_bytes": (sz, [i32, _i32p]),
    "d3d_conv_rulebook": (i32, [vp, vp, i32, i32, _i32p, _i32p, _i32p, _i32p, _i32p, vp, vp, i32,
                                vp, vp, vp, vp, sz, vp]),
    "d3d_pack_conv_weights": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "d3d_sparse_conv_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "d3d_set_tuning": (i32, [i32, i32]),
    "d3d_ctx_create": (i32, [vp]),
    "d3d_ctx_bind": (i32, [vp]),
    "d3d_ctx_destroy": (i32, [vp]),
    "d3d_ctx_get_tuning": (i32, [i32, vp]),
    "d3d_ctx_current": (i32, [vp]),
    "d3d_sparse_conv_has_epilogue": (i32, [i32, i32, i32]),
    "d3d_conv_epilogue_ws_floats": (sz, [i32, i32]),
    "d3d_bn_reduce_ws_floats": (sz, [i32, i32, i32]),
    "d3d_sparse_conv_fwd_ex": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "d3d_pair_table_bytes": (sz, [i32, i32]),
    "d3d_pair_table_build": (i32, [vp, i32, i32, i32, vp, sz, vp]),
    "d3d_sparse_conv_pairs_supported": (i32, [i32, i32, i32]),
    "d3d_sparse_conv_fwd_pairs": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "d3d_dgrad_plan_ints": (sz, [i32]),
    "d3d_dgrad_plan_supported": (i32, [vp, vp]),
    "d3d_dgrad_plan_build": (i32, [vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "d3d_conv_rulebook_chain_workspace_bytes": (sz, [i32, i32, vp]),
    "d3d_conv_rulebook_chain": (i32, [i32, vp, i32, vp, vp, i32, vp, sz, vp]),
    "d3d_subm_rulebook_ranked": (i32, [vp, vp, i32, i32, _i32p, _i32p, vp, vp, vp, vp, vp]),
    "d3d_sparse_conv_dgrad_plan_supported": (i32, [i32, i32, i32]),
    "d3d_sparse_conv_dgrad_planned": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "d3d_adam_step": (i32, [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "d3d_sparse_conv_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "d3d_sparse_conv_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "d3d_bn_relu_fwd": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp, vp, vp, vp]),
    "d3d_bn_relu_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "d3d_bn_stats": (i32, [vp, i32, i32, i32, vp, vp]),
    "d3d_bn_apply": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, f32, f32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "d3d_bn_apply_split": (i32, [vp, i32, i32, vp, vp, vp, vp, f32, f32, i32, i32, vp, i32, vp, vp, vp, vp, vp]),
    "d3d_bn_bwd_dx_split": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "d3d_bn_bwd_reduce": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "d3d_bn_apply_residual": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, f32, f32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "d3d_bn_bwd_reduce_residual": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "d3d_bn_bwd_dx_residual": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "d3d_bn_bwd_dx": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp]),
    "d3d_pack_all": (i32, [vp, i32, vp]),
    "d3d_sparse_stack_forward": (i32, [vp, i32, i32, vp]),
    "d3d_sparse_stack_backward": (i32, [vp, i32, i32, vp, vp, vp]),
    "d3d_conv2d_supported": (i32, [i32, i32, i32, i32]),
    "d3d_conv2d_supported_split3x3": (i32, [i32, i32]),
    "d3d_conv2d_stats_ws_floats": (sz, [i32, i32, i32, i32]),
    "d3d_conv2d_nhwc_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
    "d3d_conv2d_nhwc_dgrad_s2": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "d3d_conv2d_pack_weights": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "d3d_split_planes": (i32, [vp, i64, vp, vp, vp]),
    "d3d_conv2d_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "d3d_conv2d_nhwc_wgrad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "d3d_head_conv_supported": (i32, [i32, i32]),
    "d3d_head_conv_fwd": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i32, vp]),
    "d3d_head_conv_dgrad": (i32, [vp, vp, i32, i64, i32, i32, vp, vp, vp]),
    "d3d_head_conv_wgrad_ws_floats": (sz, [i64]),
    "d3d_head_conv_wgrad": (i32, [vp, vp, vp, i32, i64, i32, i32, vp, sz, vp]),
    "d3d_channel_affine": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp]),
    "d3d_channel_affine_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "d3d_pfn_ws_floats": (sz, [i32, i32, i32]),
    "d3d_pfn_fwd": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, vp, i32, vp, vp, vp, vp, f32, f32,
                          i32, vp, vp, vp, vp, vp, sz, vp]),
    "d3d_pfn_bwd": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, vp, i32, vp, vp, vp, vp, vp, vp,
                          vp, vp, vp, vp, sz, vp]),
    "d3d_dense_scatter": (i32, [vp, vp, i32, i32, i32, _i32p, i32, i32, vp, vp]),
    "d3d_dense_gather": (i32, [vp, vp, i32, i32, i32, _i32p, i32, i32, vp, vp]),
    "d3d_assign_targets_workspace_bytes": (sz, [i32, i32, i32]),
    "d3d_assign_targets": (i32, [vp, i32, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, sz, vp]),
    "d3d_assign_targets_task": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp]),
    "d3d_head_loss_workspace_bytes": (sz, [i32]),
    "d3d_head_loss": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, f32, f32, f32,
                            f32, f32, f32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "d3d_rotate_nms_workspace_bytes": (sz, [i32]),
    "d3d_rotate_nms": (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    "d3d_decode_filter_topk_workspace_bytes": (sz, [i32, i32]),
    "d3d_decode_filter_topk": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp,
                                     vp, vp, sz, vp]),
    "d3d_rotate_nms_batched_workspace_bytes": (sz, [i32, i32]),
    "d3d_rotate_nms_batched": (i32, [vp, i32, vp, i32, i32, f32, i32, vp, vp, vp, sz, vp]),
    "d3d_rotate_iou": (i32, [vp, i32, vp, i32, vp, vp]),
    "d3d_nms": (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    "d3d_iou3d_pairwise": (i32, [vp, i32, vp, i32, i32, vp, vp]),
    "d3d_iou3d_nms_workspace_bytes": (sz, [i32]),
    "d3d_iou3d_nms": (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    "d3d_rotate_iou_eval": (i32, [vp, i32, vp, i32, i32, vp, vp]),
}

ABI_VERSION = 12  # include/d3d.h D3D_ABI_VERSION
D3D_F32 = 0
D3D_BF16 = 1
D3D_CONV_BF16, D3D_CONV_SPLIT = 0, 1
D3D_BN_COPIES = 1
D3D_LOSS_SCALARS = 18
D3D_IOU3D_OVERLAP, D3D_IOU3D_IOU_BEV, D3D_IOU3D_IOU_NORMAL = 0, 1, 2


class D3dError(RuntimeError):
    pass


class ConvEpilogue(ctypes.Structure):
    """D3dConvEpilogue of include/d3d.h (device pointers as integers)"""
    _fields_ = [("stats", vp), ("bn_x", vp), ("bn_mean", vp), ("bn_invstd", vp), ("bn_gamma", vp),
                ("bn_beta", vp), ("bn_relu", i32), ("bn_partial", vp), ("ws_floats", i32), ("addend", vp)]


class StackLayer(ctypes.Structure):
    """D3dStackLayer of include/d3d.h"""
    _fields_ = [("nbr_out", vp), ("nbr_in", vp), ("dgrad_plan", vp),
                ("cap_out", i32), ("cap_in", i32), ("n_in", i32), ("n_out", i32), ("kvol", i32), ("subm", i32),
                ("cin", i32), ("cout", i32), ("relu", i32), ("partial_by_epilogue", i32),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float),
                ("x", vp), ("out", vp), ("y", vp), ("wpack", vp), ("wd", vp), ("stats", vp),
                ("stats_floats", i32), ("partial_floats", i32),
                ("gamma", vp), ("beta", vp), ("running_mean", vp), ("running_var", vp), ("save_mean", vp), ("save_invstd", vp),
                ("gout", vp), ("gin", vp), ("partial", vp), ("gbn", vp), ("gw", vp), ("wgrad_ws", vp),
                ("wgrad_ws_bytes", ctypes.c_size_t)]


class PackDesc(ctypes.Structure):
    """D3dPackDesc of include/d3d.h"""
    _fields_ = [("weight", vp), ("fwd", vp), ("fwd_lo", vp), ("dgrad", vp), ("dgrad_lo", vp), ("kind", i32), ("a", i32),
                ("b", i32), ("c", i32), ("flip", i32), ("dtype", i32)]


class ChainLayer(ctypes.Structure):
    """D3dChainLayer of include/d3d.h"""
    _i3 = ctypes.c_int32 * 3
    _fields_ = [("in_shape", _i3), ("ksize", _i3), ("stride", _i3), ("padding", _i3), ("out_shape", _i3),
                ("cap_out", ctypes.c_int32), ("coords_out", vp), ("n_out_dev", vp), ("nbr_out", vp), ("nbr_in", vp),
                ("pair_count", vp), ("plan", vp), ("plan_ints", ctypes.c_uint64), ("rank_bitmap", vp), ("rank_prefix", vp)]


class DenseBnBackward(ctypes.Structure):
    """D3dDenseBnBackward of include/d3d.h"""
    _fields_ = [("x", vp), ("mean", vp), ("invstd", vp), ("gamma", vp), ("beta", vp), ("relu", i32)]


class AdamHyper(ctypes.Structure):
    """D3dAdamHyper of include/d3d.h"""
    _fields_ = [("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("decay", f32), ("max_norm", f32),
                ("grad_scale", f32), ("step", i32)]


def lib():
    """Load libd3d_hip.so (once).  Raises if it was not built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise D3dError(
                "%s not found: build it with `python -m det3d_amd.csrc.build` "
                "(__graft_entry__.build()); det3d_amd has no CPU fallback" % LIB_PATH)
        # CDLL, not PyDLL: the calls release the GIL.  Every entry point is a short non-blocking enqueue, but the training
        # thread and the geometry thread issue ~250 of them per step between them and their HIP launches overlap only with
        # the GIL released: bf16 step (host-bound) 970 / 966 / 898 frames/s against 840 / 836 / 878 with the GIL kept
        # (round 6, same box, alternating)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.d3d_abi_version() != ABI_VERSION:
            raise D3dError("libd3d_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().d3d_last_error()
        raise D3dError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def f32_array(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def i32_array(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def ptr(t):
    """data pointer of a (contiguous) torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI needs contiguous buffers"
    return t.data_ptr()


_RAW_STREAM = None


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` as an int.  Called once per kernel launch (~60
    times per training step), so it goes through the raw C accessor (0.3 us) rather than building a
    torch.cuda.Stream object each time (5 us)."""
    global _RAW_STREAM
    import torch

    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _RAW_STREAM:
        idx = device.index if isinstance(device, torch.device) else device
        if idx is None:
            idx = torch.cuda.current_device()
        return _RAW_STREAM(idx)
    return torch.cuda.current_stream(device).cuda_stream


class Context:
    """A caller-owned library context (include/d3d.h, ABI 10): tuning knobs + fold ticket rings.  `with Context():` binds
    it to the calling thread for the block and destroys it afterwards (the caller synchronises its streams first)."""

    def __init__(self):
        self.handle = ctypes.c_void_p()
        check(lib().d3d_ctx_create(ctypes.byref(self.handle)), "d3d_ctx_create")

    def bind(self):
        check(lib().d3d_ctx_bind(self.handle), "d3d_ctx_bind")
        return self

    def destroy(self):
        if self.handle:
            check(lib().d3d_ctx_destroy(self.handle), "d3d_ctx_destroy")
            self.handle = ctypes.c_void_p()

    def __enter__(self):
        return self.bind()

    def __exit__(self, *exc):
        self.destroy()
        return False


def current_ctx():
    """handle (int) of the library context bound to the calling thread, None for the process default"""
    h = ctypes.c_void_p()
    check(lib().d3d_ctx_current(ctypes.byref(h)), "d3d_ctx_current")
    return h.value


def ctx_forward(fn):
    """decorator of an autograd Function's forward: remember the library context the forward ran under"""
    import functools

    @functools.wraps(fn)
    def wrap(ctx, *args, **kwargs):
        ctx._d3d_ctx = current_ctx()
        return fn(ctx, *args, **kwargs)
    return wrap


def ctx_backward(fn):
    """decorator of an autograd Function's backward: PyTorch runs it on the autograd engine's thread, which has no
    library context bound -- re-bind the forward's for the duration (its knobs can change operand formats: knob 10 makes
    the forward pack plain-f32 weights that a default-context backward would read as hi/lo bf16 planes)"""
    import functools

    @functools.wraps(fn)
    def wrap(ctx, *grads):
        want = getattr(ctx, "_d3d_ctx", None)
        have = current_ctx()
        if want == have:
            return fn(ctx, *grads)
        L = lib()
        check(L.d3d_ctx_bind(ctypes.c_void_p(want)), "d3d_ctx_bind")
        try:
            return fn(ctx, *grads)
        finally:
            L.d3d_ctx_bind(ctypes.c_void_p(have))
    return wrap
