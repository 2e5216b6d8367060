// Anchor target assignment on the device (the first "next" row).
//
// Reference (CPU, inside DataLoader workers): AssignTarget
// det3d/datasets/pipelines/preprocess.py:346-483 -> TargetAssigner.assign_v2
// det3d/core/anchor/target_assigner.py:68-142 (one create_target_np call PER CLASS of the task: the
// class's own anchors against the gt boxes of that class, per-class thresholds, results interleaved
// along the anchors-per-location axis) -> create_target_np det3d/core/anchor/target_ops.py:28-222 with
// NearestIouSimilarity (det3d/core/bbox/region_similarity.py:73-93: rbbox2d_to_near_bbox
// box_np_ops.py:343-355 + iou_jit(eps=0) :955-994) and second_box_encode (box_np_ops.py:52-113; 7-dim
// KITTI / Lyft boxes, 9-dim nuScenes boxes with velocities, optional angle-vector encoding).
//
// Anchor a of a task with K classes and R rotations per class belongs to class (a / R) % K -- the
// layout assign_v2 produces by concatenating the per-class anchor sets on axis -2.  Two passes over
// (frame, anchor); the gt boxes of a frame (<= a few dozen) are looped in registers and only those of
// the anchor's class take part.  The per-gt maximum over all anchors (the "force match" rule, ties
// included) goes through atomicMax on the float bits (IoU >= 0, so the int order is the float order).
//   pass 1: best gt per anchor (max IoU, first argmax among the class's gts) + per-gt maxima
//   pass 2: labels (positive: IoU >= matched or tied with a gt's maximum; negative: IoU < unmatched;
//           else -1 = ignored; a class without gt in the frame: all background), box encodings for
//           positives, regression weights
// Integer outputs are bit-exact w.r.t. the numpy original (same float32 op order, no FMA contraction);
// encodings differ by libm-vs-device log / cos / sin (1-2 ulp).
#include "d3d_common.h"
#include "d3d_fill.h"

#define ASG_PI 3.14159274101257324f  // float32(np.pi)
#define ASG_MAX_CLASSES 8

struct AsgBox {
  float x0, y0, x1, y1;
};

struct AsgTask {
  int nd;          // box dimension: 7 (x,y,z,w,l,h,r) or 9 (x,y,z,w,l,h,vx,vy,r)
  int code;        // encoding size
  int vec;         // encode_angle_vector
  int k;           // classes in this task
  int r;           // anchors per location and class (rotations x sizes)
  float matched[ASG_MAX_CLASSES], unmatched[ASG_MAX_CLASSES];
};

// np.nan_to_num on the gt boxes (target_assigner.py:96)
__device__ __forceinline__ float asg_num(float v) {
  if (v != v) return 0.f;
  if (v > 3.402823466e38f) return 3.402823466e38f;
  if (v < -3.402823466e38f) return -3.402823466e38f;
  return v;
}

// rbbox2d_to_near_bbox on (x, y, w, l, r): standing or lying axis-aligned box
__device__ __forceinline__ AsgBox asg_near(float x, float y, float w, float l, float r) {
  float lp = r - floorf(r / ASG_PI + 0.5f) * ASG_PI;  // limit_period(r, 0.5, pi)
  bool swap = fabsf(lp) > (ASG_PI / 4.f);
  float dx = swap ? l : w, dy = swap ? w : l;
  AsgBox o;
  o.x0 = x - dx / 2.f;
  o.y0 = y - dy / 2.f;
  o.x1 = x + dx / 2.f;
  o.y1 = y + dy / 2.f;
  return o;
}

__device__ __forceinline__ AsgBox asg_near_box(const float *b, int nd, bool clean) {
  if (clean) return asg_near(asg_num(b[0]), asg_num(b[1]), asg_num(b[3]), asg_num(b[4]), asg_num(b[nd - 1]));
  return asg_near(b[0], b[1], b[3], b[4], b[nd - 1]);
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

__device__ __forceinline__ int asg_class_of(const int32_t *gt_classes, int idx) {
  return gt_classes ? gt_classes[idx] : 1;
}

// wave-wide maximum of a non-negative float, returned to every lane
__device__ __forceinline__ float asg_wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

__global__ void __launch_bounds__(256)
    asg_pass1_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt,
                     const int32_t *__restrict__ gt_classes, const int32_t *__restrict__ gt_count, int batch,
                     int gmax, AsgTask T, float *__restrict__ a_max, int32_t *__restrict__ a_arg,
                     int32_t *__restrict__ g_max_bits) {
  const int64_t total = (int64_t)batch * A;
  // (the loop bound is wave-uniform: every lane of a wave runs the same number of iterations, dead lanes carry IoU 0 into
  // the wave-wide maxima below)
  for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < total; t0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = t0 + threadIdx.x;
    const bool live = t < total;
    const int64_t tc = live ? t : total - 1;
    int b = (int)(tc / A), a = (int)(tc - (int64_t)b * A);
    const int cls = T.k > 1 ? (a / T.r) % T.k + 1 : 0;  // 0: single-class task, every gt takes part
    AsgBox ab = asg_near_box(anchors + (int64_t)a * T.nd, T.nd, false);
    // the per-gt maxima: neighbouring anchors overlap the same gt boxes, and one atomicMax per overlapping (anchor, gt)
    // pair serialised a few thousand L2 atomics on a few dozen addresses (72 us for the KITTI head).  A wave whose lanes
    // all sit in one frame reduces each gt's IoU over its lanes first and issues one atomic per (wave, gt) that has overlap.
    const bool one_frame = __all(b == __builtin_amdgcn_readfirstlane(b));
    int n = gt_count[b];
    n = n < gmax ? n : gmax;
    float best = -1.f;
    int arg = -1;
    if (one_frame) {
      for (int g = 0; g < n; ++g) {  // n is wave-uniform here
        const bool mine = !(cls && asg_class_of(gt_classes, b * gmax + g) != cls);
        float iou = 0.f;
        if (mine) {
          AsgBox gb = asg_near_box(gt + ((int64_t)b * gmax + g) * T.nd, T.nd, true);
          iou = asg_iou(ab, gb);
          if (iou > best) {  // argmax = first maximum, like np.argmax
            best = iou;
            arg = g;
          }
        }
        const float wmax = asg_wave_max(live && mine ? iou : 0.f);
        if (wmax > 0.f && (threadIdx.x & 63) == 0) atomicMax(&g_max_bits[b * gmax + g], __float_as_int(wmax));
      }
    } else {
      for (int g = 0; g < n; ++g) {
        if (cls && asg_class_of(gt_classes, b * gmax + g) != cls) continue;
        AsgBox gb = asg_near_box(gt + ((int64_t)b * gmax + g) * T.nd, T.nd, true);
        float iou = asg_iou(ab, gb);
        if (iou > best) {
          best = iou;
          arg = g;
        }
        if (live && iou > 0.f) atomicMax(&g_max_bits[b * gmax + g], __float_as_int(iou));
      }
    }
    if (live) {
      a_max[t] = arg >= 0 ? best : 0.f;
      a_arg[t] = arg;  // -1: no gt of this anchor's class in the frame
    }
  }
}

__global__ void __launch_bounds__(256)
    asg_pass2_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt,
                     const int32_t *__restrict__ gt_classes, const int32_t *__restrict__ gt_count,
                     int batch, int gmax, AsgTask T, const float *__restrict__ a_max,
                     const int32_t *__restrict__ a_arg, const int32_t *__restrict__ g_max_bits,
                     int32_t *__restrict__ labels, float *__restrict__ reg_targets,
                     float *__restrict__ reg_weights) {
  const int64_t total = (int64_t)batch * A;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(t / A), a = (int)(t - (int64_t)b * A);
    const float *an = anchors + (int64_t)a * T.nd;
    const int ci = T.k > 1 ? (a / T.r) % T.k : 0;
    const int cls = T.k > 1 ? ci + 1 : 0;
    int n = gt_count[b];
    n = n < gmax ? n : gmax;
    const int arg = a_arg[t];
    int label = 0;  // no gt of the class in the frame: everything is background (:150-151)
    float enc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (arg >= 0) {
      label = -1;
      const float best = a_max[t];
      const int gcls = asg_class_of(gt_classes, b * gmax + arg);
      // force match: this anchor attains some gt's maximum overlap (ties included, :118-124)
      bool force = false;
      AsgBox ab = asg_near_box(an, T.nd, false);
      for (int g = 0; g < n && !force; ++g) {
        if (cls && asg_class_of(gt_classes, b * gmax + g) != cls) continue;
        int mb = g_max_bits[b * gmax + g];
        if (mb <= 0) continue;  // a gt no anchor touches matches nothing (:109-110)
        AsgBox gb = asg_near_box(gt + ((int64_t)b * gmax + g) * T.nd, T.nd, true);
        force = __float_as_int(asg_iou(ab, gb)) == mb;
      }
      if (best >= T.matched[ci]) label = gcls;
      if (best < T.unmatched[ci]) label = 0;
      if (force) label = gcls;
      if (label > 0) {  // second_box_encode(gt[arg], anchor), log sizes
        const float *gp = gt + ((int64_t)b * gmax + arg) * T.nd;
        float gb[9];
        for (int j = 0; j < T.nd; ++j) gb[j] = asg_num(gp[j]);
        float diag = sqrtf(an[4] * an[4] + an[3] * an[3]);
        enc[0] = (gb[0] - an[0]) / diag;
        enc[1] = (gb[1] - an[1]) / diag;
        enc[2] = (gb[2] - an[2]) / an[5];
        enc[3] = logf(gb[3] / an[3]);
        enc[4] = logf(gb[4] / an[4]);
        enc[5] = logf(gb[5] / an[5]);
        int o = 6;
        if (T.nd == 9) {
          enc[6] = gb[6] - an[6];
          enc[7] = gb[7] - an[7];
          o = 8;
        }
        const float rg = gb[T.nd - 1], ra = an[T.nd - 1];
        if (T.vec) {
          enc[o] = cosf(rg) - cosf(ra);
          enc[o + 1] = sinf(rg) - sinf(ra);
        } else {
          enc[o] = rg - ra;
        }
      }
    }
    labels[t] = label;
    reg_weights[t] = label > 0 ? 1.f : 0.f;
    for (int j = 0; j < T.code; ++j) reg_targets[t * T.code + j] = enc[j];
  }
}

extern "C" size_t d3d_assign_targets_workspace_bytes(int num_anchors, int batch, int gmax) {
  size_t ba = (size_t)batch * num_anchors;
  return d3d_align_up(ba * 4) * 2 + d3d_align_up((size_t)batch * gmax * 4) + 256;
}

static int asg_run(const float *anchors, int num_anchors, const AsgTask &T, const float *gt,
                   const int32_t *gt_classes, const int32_t *gt_count, int batch, int gmax, int32_t *labels,
                   float *reg_targets, float *reg_weights, void *workspace, size_t workspace_bytes,
                   hipStream_t stream, const char *who) {
  D3D_REQUIRE(num_anchors >= 1 && batch >= 1 && gmax >= 1, "%s: sizes", who);
  D3D_REQUIRE(anchors && gt && gt_count && labels && reg_targets && reg_weights && workspace,
              "%s: null pointer", who);
  if (workspace_bytes < d3d_assign_targets_workspace_bytes(num_anchors, batch, gmax))
    return d3d_set_error(D3D_ECAPACITY, "%s: workspace too small", who);
  D3dArena A(workspace, workspace_bytes);
  const size_t ba = (size_t)batch * num_anchors;
  float *a_max = A.take<float>(ba);
  int32_t *a_arg = A.take<int32_t>(ba);
  int32_t *g_max = A.take<int32_t>((size_t)batch * gmax);
  D3D_HIP_TRY(d3d_fill_async(stream, g_max, 0, (size_t)batch * gmax * 4));
  int nb = (int)((ba + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(asg_pass1_kernel, dim3(nb), dim3(256), 0, stream, anchors, num_anchors, gt, gt_classes,
                     gt_count, batch, gmax, T, a_max, a_arg, g_max);
  D3D_LAUNCH_TRY("asg_pass1_kernel");
  hipLaunchKernelGGL(asg_pass2_kernel, dim3(nb), dim3(256), 0, stream, anchors, num_anchors, gt,
                     gt_classes, gt_count, batch, gmax, T, a_max, a_arg, g_max, labels, reg_targets,
                     reg_weights);
  D3D_LAUNCH_TRY("asg_pass2_kernel");
  return D3D_OK;
}

extern "C" int d3d_assign_targets(const float *anchors, int num_anchors, const float *gt,
                                  const int32_t *gt_classes, const int32_t *gt_count, int batch,
                                  int gmax, float matched_threshold, float unmatched_threshold,
                                  int32_t *labels, float *reg_targets, float *reg_weights,
                                  void *workspace, size_t workspace_bytes, d3d_stream_t stream_) {
  AsgTask T;
  T.nd = 7, T.code = 7, T.vec = 0, T.k = 1, T.r = 1;
  T.matched[0] = matched_threshold;
  T.unmatched[0] = unmatched_threshold;
  return asg_run(anchors, num_anchors, T, gt, gt_classes, gt_count, batch, gmax, labels, reg_targets,
                 reg_weights, workspace, workspace_bytes, (hipStream_t)stream_, "d3d_assign_targets");
}

extern "C" int d3d_assign_targets_task(const float *anchors, int num_anchors, int box_dim, int num_classes,
                                       int anchors_per_class, const float *matched_thresholds,
                                       const float *unmatched_thresholds, int encode_angle_vector,
                                       const float *gt, const int32_t *gt_classes, const int32_t *gt_count,
                                       int batch, int gmax, int32_t *labels, float *reg_targets,
                                       float *reg_weights, void *workspace, size_t workspace_bytes,
                                       d3d_stream_t stream_) {
  D3D_REQUIRE(box_dim == 7 || box_dim == 9, "d3d_assign_targets_task: box_dim %d (7 or 9)", box_dim);
  D3D_REQUIRE(num_classes >= 1 && num_classes <= ASG_MAX_CLASSES,
              "d3d_assign_targets_task: num_classes %d (1..%d)", num_classes, ASG_MAX_CLASSES);
  D3D_REQUIRE(anchors_per_class >= 1, "d3d_assign_targets_task: anchors_per_class");
  D3D_REQUIRE(num_anchors % (num_classes * anchors_per_class) == 0,
              "d3d_assign_targets_task: %d anchors is not a multiple of %d classes x %d", num_anchors,
              num_classes, anchors_per_class);
  D3D_REQUIRE(matched_thresholds && unmatched_thresholds, "d3d_assign_targets_task: null thresholds");
  D3D_REQUIRE(num_classes == 1 || gt_classes, "d3d_assign_targets_task: gt_classes required for > 1 class");
  AsgTask T;
  T.nd = box_dim, T.vec = encode_angle_vector ? 1 : 0, T.k = num_classes, T.r = anchors_per_class;
  T.code = box_dim - 1 + (T.vec ? 2 : 1);
  for (int i = 0; i < ASG_MAX_CLASSES; ++i) {
    T.matched[i] = matched_thresholds[i < num_classes ? i : 0];
    T.unmatched[i] = unmatched_thresholds[i < num_classes ? i : 0];
  }
  return asg_run(anchors, num_anchors, T, gt, gt_classes, gt_count, batch, gmax, labels, reg_targets,
                 reg_weights, workspace, workspace_bytes, (hipStream_t)stream_, "d3d_assign_targets_task");
}
