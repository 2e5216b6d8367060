"""Arithmetic switch for the f32 mode of the hot path -- a TEST facility, not a product mode.

The product multiplies f32 operands in SPLIT arithmetic (hi / lo bf16 pairs, three bf16 MFMAs per product, ~2^-16
relative per product).  `exact_f32()` runs the same model, the same host code and the same
BatchNorm / head / loss kernels with every CONVOLUTION product as one exact f32 multiply-add instead:

  sparse convs   libd3d_hip's VALU kernels (d3d_set_tuning knob 10: spconv_fwd_generic_kernel for forward and data
                 gradient, spconv_wgrad_generic_kernel with one split per offset: deterministic sums)
  dense RPN      MIOpen's f32 kernels (dense_stack.FP32_MODE = "exact", kernel == stride layers as nn modules,
                 the multi-task heads as one nn.Conv2d per prediction)
  unchanged      d3d_head_conv_* and the PillarFeatureNet kernels (exact f32 FMAs on the VALU already), BatchNorm
                 sweeps, loss, decode, NMS, optimizer

so that "split vs exact f32 on the same GPU" isolates the product format: tests/test_precision_gpu.py measures how far
the split mode's predictions, losses and gradients sit from the exact mode's and from the float64 oracle, and how a
300-step training run of one compares with the other.  10-50x slower than the product path.
"""
import contextlib


def _clear_packs():
    from . import dense_conv, spconv

    spconv._PACK_CACHE.clear()
    dense_conv._PACK.clear()


@contextlib.contextmanager
def exact_f32():
    from . import _lib, bbox_heads, dense_conv, dense_stack

    L = _lib.lib()
    saved = (dense_stack.FP32_MODE, bbox_heads.FUSE_HEADS, dense_conv.KS_CONV)
    _lib.check(L.d3d_set_tuning(10, 1), "d3d_set_tuning")
    dense_stack.FP32_MODE, bbox_heads.FUSE_HEADS, dense_conv.KS_CONV = "exact", False, False
    _clear_packs()  # packed weights are cached per parameter version; the knob changes their format
    try:
        yield
    finally:
        _lib.check(L.d3d_set_tuning(10, 0), "d3d_set_tuning")
        dense_stack.FP32_MODE, bbox_heads.FUSE_HEADS, dense_conv.KS_CONV = saved
        _clear_packs()


def is_exact():
    from . import dense_stack

    return dense_stack.FP32_MODE == "exact"
