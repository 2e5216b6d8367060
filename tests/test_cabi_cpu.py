"""The C-ABI shared library loads (no GPU needed) and exports exactly the entry points
include/d3d.h declares; the product fails loudly without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "d3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(hip_lib, n), n


def test_binding_table_matches_header(hip_lib):
    from det3d_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    assert hip_lib.d3d_abi_version() == _lib.ABI_VERSION == 12


def test_rulebook_chain_planning_and_argument_checks_without_gpu(hip_lib):
    """d3d_conv_rulebook_chain (ABI 9): the workspace is two words per 32 output cells of every layer (bitmap + prefix)
    plus the tile sums; geometry that does not chain, too many layers, key spaces beyond 32 bits and short workspaces are
    refused before any launch; the test knob of the exact-f32 sparse kernels flips the epilogue predicate."""
    from det3d_amd import _lib

    err = hip_lib.d3d_last_error
    arr = (_lib.ChainLayer * 2)()

    def layer(y, ins, k, s, p):
        outs = [(ins[a] + 2 * p[a] - k[a]) // s[a] + 1 for a in range(3)]
        for a in range(3):
            y.in_shape[a], y.ksize[a], y.stride[a], y.padding[a], y.out_shape[a] = ins[a], k[a], s[a], p[a], outs[a]
        y.cap_out, y.coords_out, y.n_out_dev, y.nbr_out, y.nbr_in = 100, 1, 1, 1, 1
        return outs

    o1 = layer(arr[0], [41, 1600, 1408], [3, 3, 3], [2, 2, 2], [1, 1, 1])
    o2 = layer(arr[1], o1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert o1 == [21, 800, 704] and o2 == [11, 400, 352]
    p = ctypes.cast(arr, ctypes.c_void_p)
    words = lambda o: ((4 * o[0] * o[1] * o[2] + 31) // 32 + 7) // 8 * 8
    need = hip_lib.d3d_conv_rulebook_chain_workspace_bytes(2, 4, p)
    assert 8 * (words(o1) + words(o2)) <= need <= 8 * (words(o1) + words(o2)) + 8192
    assert hip_lib.d3d_conv_rulebook_chain_workspace_bytes(9, 4, p) == 0
    assert hip_lib.d3d_conv_rulebook_chain(2, p, 4, 1, 1, 10, 1, 10, None) == -2 and b"workspace" in err()
    assert hip_lib.d3d_conv_rulebook_chain(9, p, 4, 1, 1, 10, 1, 1 << 30, None) == -1 and b"layers" in err()
    assert hip_lib.d3d_conv_rulebook_chain(2, p, 4, None, 1, 10, 1, 1 << 30, None) == -1
    arr[1].in_shape[1] = 799  # (799 + 2 - 3) // 2 + 1 is still 400: only the link to layer 0's grid is broken
    assert hip_lib.d3d_conv_rulebook_chain(2, p, 4, 1, 1, 10, 1, 1 << 30, None) == -1 and b"continue" in err()
    arr[1].in_shape[1] = 800
    assert hip_lib.d3d_conv_rulebook_chain(2, p, 400, 1, 1, 10, 1, 1 << 40, None) == -2 and b"u32" in err()
    shape, k3 = (ctypes.c_int32 * 3)(21, 800, 704), (ctypes.c_int32 * 3)(3, 3, 3)
    assert hip_lib.d3d_subm_rulebook_ranked(1, 1, 10, 4, shape, k3, None, 1, 1, None, None) == -1 and b"null" in err()
    assert hip_lib.d3d_subm_rulebook_ranked(1, 1, 10, 400, shape, k3, 1, 1, 1, None, None) == -2
    # knob 10 (test knob): exact-f32 VALU kernels for f32 features -> no epilogues, no planned data gradient
    assert hip_lib.d3d_sparse_conv_has_epilogue(64, 64, _lib.D3D_F32) == 1
    try:
        assert hip_lib.d3d_set_tuning(10, 1) == 0
        assert hip_lib.d3d_sparse_conv_has_epilogue(64, 64, _lib.D3D_F32) == 0
        assert hip_lib.d3d_sparse_conv_has_epilogue(64, 64, _lib.D3D_BF16) == 1
        assert hip_lib.d3d_sparse_conv_dgrad_plan_supported(64, 64, _lib.D3D_F32) == 0
    finally:
        hip_lib.d3d_set_tuning(10, 0)
    assert hip_lib.d3d_sparse_conv_dgrad_plan_supported(64, 64, _lib.D3D_F32) == 1


def test_argument_validation_without_gpu(hip_lib):
    # argument checks happen before any HIP call, so they are testable on CPU
    rc = hip_lib.d3d_sparse_conv_fwd(None, None, None, None, 10, 20, 27, 16, 16, 0, None, None)
    assert rc == -1
    assert b"n_out" in hip_lib.d3d_last_error()
    rc = hip_lib.d3d_sparse_conv_fwd(None, None, None, None, 10, 5, 27, 16, 16, 7, None, None)
    assert rc == -1 and b"dtype" in hip_lib.d3d_last_error()
    shape = (ctypes.c_int32 * 3)(41, 1600, 1408)
    k = (ctypes.c_int32 * 3)(2, 3, 3)
    rc = hip_lib.d3d_subm_rulebook(1, 1, 10, 1, shape, k, 1, 1, None, 0, None)
    assert rc == -1 and b"odd" in hip_lib.d3d_last_error()
    big = (ctypes.c_int32 * 3)(4000, 4000, 4000)
    k3 = (ctypes.c_int32 * 3)(3, 3, 3)
    assert hip_lib.d3d_subm_rulebook(1, 1, 10, 1, big, k3, 1, 1, None, 0, None) == -2


def test_argument_validation_of_the_round1_additions(hip_lib):
    """iou3d / evaluation IoU / task-group AssignTarget / head loss / weight repack / BN: every argument check
    runs before the first HIP call."""
    err = hip_lib.d3d_last_error
    assert hip_lib.d3d_iou3d_pairwise(1, 3, 1, 3, 7, 1, None) == -1 and b"mode" in err()
    assert hip_lib.d3d_iou3d_pairwise(None, 0, None, 5, 1, None, None) == 0          # empty: nothing to do
    assert hip_lib.d3d_iou3d_nms(1, 5, -0.5, 0, 1, 1, 1, 1 << 20, None) == -1 and b"threshold" in err()
    assert hip_lib.d3d_iou3d_nms(1, 20000, 0.1, 0, 1, 1, 1, 1 << 30, None) == -2     # > 16384 boxes
    assert hip_lib.d3d_iou3d_nms(1, 100, 0.1, 0, 1, 1, 1, 8, None) == -2 and b"workspace" in err()
    assert hip_lib.d3d_rotate_iou_eval(1, 3, 1, 3, 5, 1, None) == -1 and b"criterion" in err()
    thr = (ctypes.c_float * 2)(0.5, 0.5)
    args = lambda box_dim, k, per, a: (1, a, box_dim, k, per, thr, thr, 1, 1, 1, 1, 1, 4, 1, 1, 1, 1, 1 << 20, None)
    assert hip_lib.d3d_assign_targets_task(*args(8, 2, 2, 16)) == -1 and b"box_dim" in err()
    assert hip_lib.d3d_assign_targets_task(*args(9, 9, 2, 36)) == -1 and b"num_classes" in err()
    assert hip_lib.d3d_assign_targets_task(*args(9, 2, 2, 18)) == -1 and b"multiple" in err()
    assert hip_lib.d3d_pack_conv_weights(1, 27, 16, 16, 5, 0, 1, 1, None) == -1 and b"dtype" in err()
    assert hip_lib.d3d_pack_conv_weights(1, 27, 16, 16, 0, 0, None, None, None) == -1
    loss = lambda code, nc: hip_lib.d3d_head_loss(1, 1, None, 0, 1, 1, None, 0, 2, 100, code, nc, 7, 1, 0.0, 1.0, 2.0,
                                                  0.25, 2.0, 3.0, 1.0, 1.0, 0.2, 1, 1, None, 1, 1, 1 << 10, None)
    assert loss(11, 1) == -1 and b"code_size" in err()
    assert loss(7, 9) == -1 and b"num_class" in err()
    assert hip_lib.d3d_head_loss_workspace_bytes(4) >= 16
    assert hip_lib.d3d_bn_relu_fwd(1, 100, 24, 1, None, None, None, None, 0.01, 1e-3, 1, 1, 1, None, None, 1, None) == -1
    assert b"unsupported" in err()


def test_workspace_queries(hip_lib):
    assert hip_lib.d3d_voxelize_workspace_bytes(60000, 4, 5, 20000) > 60000 * 4
    assert hip_lib.d3d_rotate_nms_workspace_bytes(1000) >= 1000 * 16 * 8
    assert hip_lib.d3d_subm_rulebook_workspace_bytes(1000) >= 2048 * 8


def test_no_cpu_fallback():
    from det3d_amd import _lib, point_cloud_ops, spconv

    pts = np.random.rand(10, 4).astype(np.float32)
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            point_cloud_ops.points_to_voxel(pts, [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1], 5, True, 100)
        with pytest.raises(_lib.D3dError):
            point_cloud_ops.voxelize_batch(torch.from_numpy(pts), torch.tensor([0, 10]), [0.05, 0.05, 0.1],
                                           [0, -40, -3, 70.4, 40, 1], 5, 100)
        x = spconv.SparseConvTensor(torch.zeros(3, 4), torch.zeros(3, 4, dtype=torch.int32), [4, 4, 4], 1)
        with pytest.raises(Exception):
            spconv.SubMConv3d(4, 16, 3)(x)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "det3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libd3d_oracle" not in src, f


def test_host_side_planning_helpers(hip_lib):
    """size / support predicates that run on the host alone (no device call): row plans of the strided data gradients,
    PillarFeatureNet and conv-epilogue scratch sizes, the chunk plan of the fused optimizer step"""
    import ctypes

    from det3d_amd import _lib
    from det3d_amd.fused_adam import plan_chunks

    arr = lambda v: (ctypes.c_int32 * 3)(*v)
    L = hip_lib
    assert L.d3d_dgrad_plan_supported(arr([3, 3, 3]), arr([2, 2, 2])) == 1      # scn.py:113 / :127 / :141
    assert L.d3d_dgrad_plan_supported(arr([3, 1, 1]), arr([2, 1, 1])) == 1      # the last (3,1,1) / (2,1,1) layer
    assert L.d3d_dgrad_plan_supported(arr([3, 3, 3]), arr([1, 1, 1])) == 0      # nothing to sort
    assert L.d3d_dgrad_plan_supported(arr([3, 3, 3]), arr([3, 2, 2])) == 0      # 12 classes > 8
    assert L.d3d_dgrad_plan_supported(arr([4, 4, 4]), arr([2, 2, 2])) == 0      # 64 offsets > 27
    assert L.d3d_dgrad_plan_ints(1000) == 304 + 1000 + 8
    assert L.d3d_dgrad_plan_ints(1025) == 304 + 1025 + 16 and L.d3d_dgrad_plan_ints(0) == 0
    for cin, cout, ok in [(64, 32, 1), (32, 16, 1), (64, 64, 1), (128, 64, 1), (128, 128, 1), (16, 16, 0), (32, 64, 0)]:
        assert L.d3d_sparse_conv_dgrad_plan_supported(cin, cout, _lib.D3D_F32) == ok
        assert L.d3d_sparse_conv_dgrad_plan_supported(cin, cout, _lib.D3D_BF16) == ok
    # one slab of max(2f, f (c + 5)) floats per wave (<= 8192 waves, rounded to blocks of 4) behind the 2f results, + 16 rows for the two-level fold
    assert L.d3d_pfn_ws_floats(100, 4, 64) == 128 + (100 + 16) * 64 * 9
    assert L.d3d_pfn_ws_floats(50000, 5, 64) == 128 + (8192 + 16) * 64 * 10
    assert L.d3d_pfn_ws_floats(10, 6, 64) == 0 and L.d3d_pfn_ws_floats(10, 4, 65) == 0
    assert L.d3d_conv_epilogue_ws_floats(0, 64) == 0 and L.d3d_conv_epilogue_ws_floats(1000, 64) >= 128 * (1 + 16)
    # optimizer chunks: tensor by tensor, multiples of 256 elements, at most ~4096 chunks of >= 4096 elements
    chunk, chunks, start = plan_chunks([1, 4096, 4097, 10])
    assert chunk == 4096 and chunks == [(0, 0), (1, 0), (2, 0), (2, 4096), (3, 0)] and start == [0, 1, 2, 4, 5]
    chunk, chunks, start = plan_chunks([40_000_000, 5])
    assert chunk % 256 == 0 and chunk >= 40_000_005 / 4096 and len(chunks) <= 4097 and start[-1] == len(chunks)
    assert all(o % chunk == 0 for _, o in chunks)


def test_round4_planners_and_argument_checks_without_gpu(hip_lib):
    """The host-side planners of the round-4 entry points run before any HIP call: the sparse weight gradient's
    workspace follows its split plan (and the tuning knob that moves it), the strided data gradient and the stacked
    conv entry points refuse what they do not implement."""
    from det3d_amd import _lib

    err = hip_lib.d3d_last_error
    ws = hip_lib.d3d_sparse_conv_wgrad_workspace_bytes
    for cin, cout in ((16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)):
        slab = 27 * cin * cout * 4
        for dt in (_lib.D3D_F32, _lib.D3D_BF16):
            for n in (1, 100, 65083, 1500000):
                b = ws(n, 27, cin, cout, dt)
                assert b > 256 and (b - 256) % slab == 0, (cin, cout, dt, n, b)  # whole partial slabs + the header
            assert ws(100, 27, cin, cout, dt) <= ws(65083, 27, cin, cout, dt) <= ws(1500000, 27, cin, cout, dt)
    assert ws(1000, 27, 48, 48, _lib.D3D_F32) == 256  # channel pair without a tile kernel: the one-pass fallback
    # knob 1: value % 100000 = target blocks, value / 100000 = kernel variant (1 = the round-3 kernel on 16 waves)
    base = ws(126683, 27, 64, 64, _lib.D3D_F32)
    try:
        assert hip_lib.d3d_set_tuning(1, 100000) == 0
        old = ws(126683, 27, 64, 64, _lib.D3D_F32)
        assert hip_lib.d3d_set_tuning(1, 100000 + 128) == 0
        few = ws(126683, 27, 64, 64, _lib.D3D_F32)
    finally:
        hip_lib.d3d_set_tuning(1, 0)
    assert few < old <= base and ws(126683, 27, 64, 64, _lib.D3D_F32) == base
    assert hip_lib.d3d_set_tuning(99, 0) == -1 and b"knob" in err()
    # the stride-2 data gradient: 3x3 layers, cout 128 | 256, cin a multiple of 128
    dg = hip_lib.d3d_conv2d_nhwc_dgrad_s2
    assert dg(1, 1, 1, 1, 1, 2, 16, 16, 128, 256, 3, 1, 7, None) == -1 and b"mode" in err()
    assert dg(1, 1, 1, 1, 1, 2, 16, 16, 64, 64, 3, 1, _lib.D3D_CONV_SPLIT, None) == -1 and b"channels" in err()
    assert dg(1, 1, 1, 1, 1, 2, 16, 16, 128, 256, 2, 0, _lib.D3D_CONV_SPLIT, None) == -1 and b"3x3" in err()
    assert dg(None, None, None, None, None, 2, 16, 16, 128, 256, 3, 1, _lib.D3D_CONV_SPLIT, None) == -1 and b"null" in err()
    assert hip_lib.d3d_conv2d_supported(128, 256, 3, 2) == 1 and hip_lib.d3d_conv2d_supported(512, 128, 3, 1) == 0
    assert hip_lib.d3d_conv2d_supported(512, 128, 1, 1) == 1 and hip_lib.d3d_conv2d_supported(384, 256, 1, 1) == 1  # stacked heads
    assert hip_lib.d3d_conv2d_supported_split3x3(64, 64) == 1 and hip_lib.d3d_conv2d_supported_split3x3(96, 128) == 0
    assert hip_lib.d3d_pack_all(None, 0, None) in (0, -1)  # nothing to pack: no launch either way


def test_contexts_own_the_tuning_state_per_thread(hip_lib):
    """ABI 10 (no process-global state): the d3d_set_tuning knobs belong to the context the CALLING THREAD has
    bound -- another thread, or the same thread after d3d_ctx_bind(NULL), sees the process default context; a context is
    created, bound, used and destroyed by the caller; the default context cannot be destroyed.  (The ticket rings of a
    context are allocated on the first reducing launch: nothing here touches a GPU.)"""
    import threading

    def knob(k):
        v = ctypes.c_int(12345)
        assert hip_lib.d3d_ctx_get_tuning(k, ctypes.byref(v)) == 0
        return v.value

    base = knob(11)
    ctx = ctypes.c_void_p()
    assert hip_lib.d3d_ctx_create(ctypes.byref(ctx)) == 0 and ctx.value
    assert hip_lib.d3d_ctx_bind(ctx) == 0
    assert knob(11) == base                      # a new context starts from the environment defaults
    assert hip_lib.d3d_set_tuning(11, 2) == 0 and hip_lib.d3d_set_tuning(10, 1) == 0
    assert (knob(11), knob(10)) == (2, 1)
    assert hip_lib.d3d_sparse_conv_has_epilogue(64, 64, 0) == 0   # knob 10 of THIS context: exact-f32 kernels
    seen = {}

    def other():
        seen["default"] = (knob(11), knob(10), hip_lib.d3d_sparse_conv_has_epilogue(64, 64, 0))
        c2 = ctypes.c_void_p()
        hip_lib.d3d_ctx_create(ctypes.byref(c2))
        hip_lib.d3d_ctx_bind(c2)
        hip_lib.d3d_set_tuning(11, 3)
        seen["own"] = knob(11)
        assert hip_lib.d3d_ctx_destroy(c2) == 0  # unbinds the calling thread
        seen["after"] = knob(11)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == {"default": (base, 0, 1), "own": 3, "after": base}
    assert (knob(11), knob(10)) == (2, 1)        # untouched by the other thread
    assert hip_lib.d3d_ctx_bind(None) == 0
    assert (knob(11), knob(10)) == (base, 0)
    assert hip_lib.d3d_ctx_destroy(ctx) == 0
    assert hip_lib.d3d_ctx_destroy(None) == -1 and b"d3d_ctx_destroy" in hip_lib.d3d_last_error()
    assert hip_lib.d3d_ctx_get_tuning(99, ctypes.byref(ctypes.c_int())) == -1
    assert hip_lib.d3d_set_tuning(12, 0) == -1 and b"unknown knob" in hip_lib.d3d_last_error()
