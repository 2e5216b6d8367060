// Anchor target assignment on the device (the first "next" row).
//
// Reference (CPU, inside DataLoader workers): AssignTarget
// det3d/datasets/pipelines/preprocess.py:346-483 -> TargetAssigner.assign_v2
// det3d/core/anchor/target_assigner.py:68-142 -> create_target_np
// det3d/core/anchor/target_ops.py:28-222 with NearestIouSimilarity
// (det3d/core/bbox/region_similarity.py:73-93: rbbox2d_to_near_bbox box_np_ops.py:343-355 +
// iou_jit(eps=0) :955-994) and second_box_encode (box_np_ops.py second_box_encode).
//
// Two passes over (frame, anchor), the gt boxes of a frame (<= a few dozen) are looped in
// registers; the per-gt maximum over all anchors (the "force match" rule, incl. ties) goes
// through atomicMax on the float bits (IoU >= 0, so the int order is the float order).
//   pass 1: best gt per anchor (max IoU, first argmax) + per-gt maxima
//   pass 2: labels (positive: IoU >= matched or tied with a gt's maximum; negative: IoU <
//           unmatched; else -1 = ignored), box encodings for positives, regression weights
// Integer outputs are bit-exact w.r.t. the numpy restatement (same float32 op order, no FMA
// contraction); encodings differ by libm-vs-device log (1 ulp).
#include "d3d_common.h"

#define ASG_PI 3.14159274101257324f  // float32(np.pi)

struct AsgBox {
  float x0, y0, x1, y1;
};

// rbbox2d_to_near_bbox on (x, y, w, l, r): standing or lying axis-aligned box
__device__ __forceinline__ AsgBox asg_near(const float *b7) {
  float r = b7[6];
  float lp = r - floorf(r / ASG_PI + 0.5f) * ASG_PI;  // limit_period(r, 0.5, pi)
  bool swap = fabsf(lp) > (ASG_PI / 4.f);
  float dx = swap ? b7[4] : b7[3], dy = swap ? b7[3] : b7[4];
  AsgBox o;
  o.x0 = b7[0] - dx / 2.f;
  o.y0 = b7[1] - dy / 2.f;
  o.x1 = b7[0] + dx / 2.f;
  o.y1 = b7[1] + dy / 2.f;
  return o;
}

// iou_jit(eps = 0): query box q against box a
__device__ __forceinline__ float asg_iou(const AsgBox &a, const AsgBox &q) {
  float iw = fminf(a.x1, q.x1) - fmaxf(a.x0, q.x0);
  float ih = fminf(a.y1, q.y1) - fmaxf(a.y0, q.y0);
  if (!(iw > 0.f && ih > 0.f)) return 0.f;
  float inter = iw * ih;
  float ua = (a.x1 - a.x0) * (a.y1 - a.y0) + (q.x1 - q.x0) * (q.y1 - q.y0) - inter;
  return inter / ua;
}

__global__ void __launch_bounds__(256)
    asg_pass1_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt,
                     const int32_t *__restrict__ gt_count, int batch, int gmax,
                     float *__restrict__ a_max, int32_t *__restrict__ a_arg,
                     int32_t *__restrict__ g_max_bits) {
  const int64_t total = (int64_t)batch * A;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(t / A), a = (int)(t - (int64_t)b * A);
    AsgBox ab = asg_near(anchors + (int64_t)a * 7);
    int n = gt_count[b];
    n = n < gmax ? n : gmax;
    float best = -1.f;
    int arg = 0;
    for (int g = 0; g < n; ++g) {
      AsgBox gb = asg_near(gt + ((int64_t)b * gmax + g) * 7);
      float iou = asg_iou(ab, gb);
      if (iou > best) {  // argmax = first maximum, like np.argmax
        best = iou;
        arg = g;
      }
      if (iou > 0.f) atomicMax(&g_max_bits[b * gmax + g], __float_as_int(iou));
    }
    a_max[t] = n > 0 ? best : 0.f;
    a_arg[t] = arg;
  }
}

__global__ void __launch_bounds__(256)
    asg_pass2_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt,
                     const int32_t *__restrict__ gt_classes, const int32_t *__restrict__ gt_count,
                     int batch, int gmax, float matched, float unmatched,
                     const float *__restrict__ a_max, const int32_t *__restrict__ a_arg,
                     const int32_t *__restrict__ g_max_bits, int32_t *__restrict__ labels,
                     float *__restrict__ reg_targets, float *__restrict__ reg_weights) {
  const int64_t total = (int64_t)batch * A;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(t / A), a = (int)(t - (int64_t)b * A);
    const float *an = anchors + (int64_t)a * 7;
    int n = gt_count[b];
    n = n < gmax ? n : gmax;
    int label = n > 0 ? -1 : 0;  // no gt in the frame: everything is background (:150-151)
    float enc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n > 0) {
      const float best = a_max[t];
      const int arg = a_arg[t];
      const int cls = gt_classes ? gt_classes[b * gmax + arg] : 1;
      // force match: this anchor attains some gt's maximum overlap (ties included, :118-124)
      bool force = false;
      AsgBox ab = asg_near(an);
      for (int g = 0; g < n && !force; ++g) {
        int mb = g_max_bits[b * gmax + g];
        if (mb <= 0) continue;  // a gt no anchor touches matches nothing (:109-110)
        AsgBox gb = asg_near(gt + ((int64_t)b * gmax + g) * 7);
        force = __float_as_int(asg_iou(ab, gb)) == mb;
      }
      if (best >= matched) label = cls;
      if (best < unmatched) label = 0;
      if (force) label = cls;
      if (label > 0) {  // second_box_encode(gt[arg], anchor), 7-dim, log sizes
        const float *gb = gt + ((int64_t)b * gmax + arg) * 7;
        float diag = sqrtf(an[4] * an[4] + an[3] * an[3]);
        enc[0] = (gb[0] - an[0]) / diag;
        enc[1] = (gb[1] - an[1]) / diag;
        enc[2] = (gb[2] - an[2]) / an[5];
        enc[3] = logf(gb[3] / an[3]);
        enc[4] = logf(gb[4] / an[4]);
        enc[5] = logf(gb[5] / an[5]);
        enc[6] = gb[6] - an[6];
      }
    }
    labels[t] = label;
    reg_weights[t] = label > 0 ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) reg_targets[t * 7 + j] = enc[j];
  }
}

extern "C" size_t d3d_assign_targets_workspace_bytes(int num_anchors, int batch, int gmax) {
  size_t ba = (size_t)batch * num_anchors;
  return d3d_align_up(ba * 4) * 2 + d3d_align_up((size_t)batch * gmax * 4) + 256;
}

extern "C" int d3d_assign_targets(const float *anchors, int num_anchors, const float *gt,
                                  const int32_t *gt_classes, const int32_t *gt_count, int batch,
                                  int gmax, float matched_threshold, float unmatched_threshold,
                                  int32_t *labels, float *reg_targets, float *reg_weights,
                                  void *workspace, size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(num_anchors >= 1 && batch >= 1 && gmax >= 1, "d3d_assign_targets: sizes");
  D3D_REQUIRE(anchors && gt && gt_count && labels && reg_targets && reg_weights && workspace,
              "d3d_assign_targets: null pointer");
  if (workspace_bytes < d3d_assign_targets_workspace_bytes(num_anchors, batch, gmax))
    return d3d_set_error(D3D_ECAPACITY, "d3d_assign_targets: workspace too small");
  D3dArena A(workspace, workspace_bytes);
  const size_t ba = (size_t)batch * num_anchors;
  float *a_max = A.take<float>(ba);
  int32_t *a_arg = A.take<int32_t>(ba);
  int32_t *g_max = A.take<int32_t>((size_t)batch * gmax);
  D3D_HIP_TRY(hipMemsetAsync(g_max, 0, (size_t)batch * gmax * 4, stream));
  int nb = (int)((ba + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(asg_pass1_kernel, dim3(nb), dim3(256), 0, stream, anchors, num_anchors, gt,
                     gt_count, batch, gmax, a_max, a_arg, g_max);
  D3D_LAUNCH_TRY("asg_pass1_kernel");
  hipLaunchKernelGGL(asg_pass2_kernel, dim3(nb), dim3(256), 0, stream, anchors, num_anchors, gt,
                     gt_classes, gt_count, batch, gmax, matched_threshold, unmatched_threshold, a_max,
                     a_arg, g_max, labels, reg_targets, reg_weights);
  D3D_LAUNCH_TRY("asg_pass2_kernel");
  return D3D_OK;
}
