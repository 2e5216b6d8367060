// The fused sparse stack (conv -> BatchNorm -> ReLU layers of SpMiddleFHD, det3d/models/backbones/scn.py:106-157) as ONE
// call per direction -- round 6, VERDICT round 5 item 7 ("move the launch sequence into one C entry").
//
// det3d_amd/sparse_stack.py issued the same launches from Python: per layer and direction three to five ctypes calls, an
// epilogue structure, two events, a handful of attribute walks -- 0.65 ms (forward) + 1.35 ms (backward) of interpreter time
// per KITTI step behind 6.0 ms of GPU time (profiles/round6_host_profile_fp32.txt), i.e. the part of the enqueue path that
// makes the step host-bound on a slow or busy host and always bounds the bf16 step.  Here the host side fills one
// descriptor per layer (buffers it allocated, rulebook pointers, parameters) and this file walks them: the SAME entry
// points in the SAME order on the same streams, so results are bit-identical to the Python sequence
// (tests/test_graph_tail_gpu.py::test_fused_sparse_stack_equals_the_per_layer_path, tests/test_stack_executor_gpu.py).
//   forward   per layer: d3d_sparse_conv_fwd_ex with the statistics epilogue, d3d_bn_apply (training)
//   backward  per layer, last to first: [d3d_bn_bwd_reduce unless the layer above reduced it in its data-gradient
//             epilogue], d3d_bn_bwd_dx, the data gradient (planned strided / gather kernel, reducing the BatchNorm sums of
//             the layer below in its epilogue where that layer allows), and the weight gradient on the second stream behind
//             an event recorded after d3d_bn_bwd_dx; the second stream is joined into the first at the end
// Not handled here (the host keeps its per-layer path for them): DistributedSyncBN layers (a collective between the launches),
// pair-compacted kernels, convolution biases, per-launch timing events.
#include "d3d_common.h"

#define STK_EVENTS 64
struct StkEvents {
  hipEvent_t ev[STK_EVENTS];
  int next;
  bool ready;
};
static thread_local StkEvents g_stk_events = {{}, 0, false};

static int stk_event(hipEvent_t *out) {
  StkEvents &E = g_stk_events;
  if (!E.ready) {
    for (int i = 0; i < STK_EVENTS; ++i) D3D_HIP_TRY(hipEventCreateWithFlags(&E.ev[i], hipEventDisableTiming));
    E.ready = true;
  }
  *out = E.ev[E.next];
  E.next = (E.next + 1) % STK_EVENTS;
  return D3D_OK;
}

extern "C" int d3d_sparse_stack_forward(const D3dStackLayer *layers, int n, int dtype, d3d_stream_t stream) {
  D3D_REQUIRE(layers && n >= 1, "d3d_sparse_stack_forward: no layers");
  for (int i = 0; i < n; ++i) {
    const D3dStackLayer &Y = layers[i];
    D3D_REQUIRE(Y.x && Y.out && Y.y && Y.wpack && Y.nbr_out && Y.stats && Y.gamma && Y.running_mean && Y.running_var &&
                    Y.save_mean && Y.save_invstd,
                "d3d_sparse_stack_forward: layer %d: null pointer", i);
    D3dConvEpilogue ep = {};
    ep.stats = Y.stats;
    ep.ws_floats = Y.stats_floats;
    if (int rc = d3d_sparse_conv_fwd_ex(Y.x, Y.wpack, nullptr, Y.nbr_out, Y.cap_out, Y.n_out, Y.kvol, Y.cin, Y.cout, dtype,
                                        Y.out, &ep, stream))
      return rc;
    if (int rc = d3d_bn_apply(Y.out, Y.n_out, Y.cout, dtype, Y.gamma, Y.beta, Y.running_mean, Y.running_var, Y.momentum,
                              Y.eps, 1, Y.relu, Y.stats, 1, Y.y, Y.save_mean, Y.save_invstd, stream))
      return rc;
  }
  return D3D_OK;
}

extern "C" int d3d_sparse_stack_backward(const D3dStackLayer *layers, int n, int dtype, const void *dy_last,
                                         d3d_stream_t main_, d3d_stream_t side_) {
  D3D_REQUIRE(layers && n >= 1 && dy_last, "d3d_sparse_stack_backward: arguments");
  hipStream_t main = (hipStream_t)main_, side = (hipStream_t)side_;
  const void *dy = dy_last;
  bool reduced = false;  // the previous iteration's data gradient filled layers[i].partial
  for (int i = n - 1; i >= 0; --i) {
    const D3dStackLayer &Y = layers[i];
    D3D_REQUIRE(Y.out && Y.gout && Y.partial && Y.gbn && Y.gw && Y.x && Y.nbr_out, "d3d_sparse_stack_backward: layer %d: null pointer", i);
    if (!reduced)
      if (int rc = d3d_bn_bwd_reduce(Y.out, dy, Y.n_out, Y.cout, dtype, Y.gamma, Y.beta, Y.save_mean, Y.save_invstd, Y.relu,
                                     Y.partial, main_))
        return rc;
    if (int rc = d3d_bn_bwd_dx(Y.out, dy, Y.n_out, Y.cout, dtype, Y.gamma, Y.beta, Y.save_mean, Y.save_invstd, Y.relu,
                               Y.partial, 1, Y.gout, Y.gbn, Y.gbn + Y.cout, main_))
      return rc;
    hipEvent_t ready = nullptr;
    if (side) {
      if (int rc = stk_event(&ready)) return rc;
      D3D_HIP_TRY(hipEventRecord(ready, main));
    }
    reduced = false;
    if (Y.gin) {
      D3D_REQUIRE(Y.wd, "d3d_sparse_stack_backward: layer %d: data gradient without packed weights", i);
      D3dConvEpilogue ep = {};
      const D3dConvEpilogue *epp = nullptr;
      if (i > 0 && layers[i - 1].partial_by_epilogue) {  // dL/dy of the layer below leaves this launch: reduce its BN sums here
        const D3dStackLayer &P = layers[i - 1];
        ep.bn_x = P.out;
        ep.bn_mean = P.save_mean;
        ep.bn_invstd = P.save_invstd;
        ep.bn_gamma = P.gamma;
        ep.bn_beta = P.beta;
        ep.bn_relu = P.relu;
        ep.bn_partial = P.partial;
        ep.ws_floats = P.partial_floats;
        epp = &ep;
        reduced = true;
      }
      int rc;
      if (!Y.subm && Y.dgrad_plan && d3d_sparse_conv_dgrad_plan_supported(Y.cout, Y.cin, dtype))
        rc = d3d_sparse_conv_dgrad_planned(Y.gout, Y.wd, Y.nbr_in, Y.cap_in, Y.n_in, Y.kvol, Y.cout, Y.cin, dtype, Y.gin, epp,
                                           Y.dgrad_plan, main_);
      else
        rc = d3d_sparse_conv_fwd_ex(Y.gout, Y.wd, nullptr, Y.subm ? Y.nbr_out : Y.nbr_in, Y.subm ? Y.cap_out : Y.cap_in,
                                    Y.n_in, Y.kvol, Y.cout, Y.cin, dtype, Y.gin, epp, main_);
      if (rc) return rc;
    }
    if (side) D3D_HIP_TRY(hipStreamWaitEvent(side, ready, 0));
    if (int rc = d3d_sparse_conv_wgrad(Y.x, Y.gout, Y.nbr_out, Y.cap_out, Y.n_out, Y.kvol, Y.cin, Y.cout, dtype, Y.gw,
                                       Y.wgrad_ws, Y.wgrad_ws_bytes, side ? side_ : main_))
      return rc;
    dy = Y.gin;
    if (!dy && i > 0) return d3d_set_error(D3D_EINVAL, "d3d_sparse_stack_backward: layer %d has no input gradient buffer", i);
  }
  if (side) {  // the weight gradients are the caller's from here on
    hipEvent_t done = nullptr;
    if (int rc = stk_event(&done)) return rc;
    D3D_HIP_TRY(hipEventRecord(done, side));
    D3D_HIP_TRY(hipStreamWaitEvent(main, done, 0));
  }
  return D3D_OK;
}
