// MultiGroupHead loss of one task group, forward AND gradient in two launches.
//
// Reference: MultiGroupHead.loss det3d/models/bbox_heads/mg_head.py:580-695 with
//   prepare_loss_weights :147-188 (NormByNumPositives), create_loss :535-578, add_sin_difference :29-34,
//   get_direction_target :54-63, _get_pos_neg_loss :36-52 and the loss classes
//   SigmoidFocalLoss losses.py:269-336, WeightedSmoothL1Loss :146-211 (code_weights are ignored there, :173),
//   WeightedSoftmaxClassificationLoss :407-446.
// In PyTorch that is ~140 element-wise / reduction kernels forward and ~60 backward per task group and
// training step (a third of all launches of the KITTI step, 2-5 us each plus the dispatch gap), all over the
// same [B, A] anchors.  Here one thread owns one anchor: it reads its label, targets and predictions once,
// evaluates the three losses, writes d(loss)/d(prediction) straight away (the loss is a plain weighted sum,
// so the gradients are local once the per-frame positive counts are known) and the block-reduced scalars
// (total, per-part, per-code and pos/neg logging sums) go out through a handful of atomics.
//   launch 1: positives per frame (+ the logged num_pos / num_neg of frame 0)
//   launch 2: everything else
#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_fold.h"

#define LOSS_MAX_CODE 10
#define LOSS_MAX_CLS 8
#define LOSS_PI 3.14159265358979323846f

struct LossCfg {
  int batch, A, code, nc, nd, anchor_bstride;  // anchors [*, A, nd], bstride 0 when shared by the batch
  int has_dir, sin_diff;
  float pos_w, neg_w;                // NormByNumPositives class weights
  float alpha, gamma, sigma;         // focal / smooth-L1 parameters
  float dir_offset;
  float w_loc, w_cls, w_dir;         // _loss_weight of the three losses
};

__device__ __forceinline__ float loss_ld(const void *p, int64_t i, int bf16) {
  return bf16 ? __uint_as_float((uint32_t)((const uint16_t *)p)[i] << 16) : ((const float *)p)[i];
}
__device__ __forceinline__ void loss_st(void *p, int64_t i, int bf16, float v) {
  if (bf16) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) {
      ((uint16_t *)p)[i] = (uint16_t)((u >> 16) | 0x40);
    } else {
      u += 0x7fffu + ((u >> 16) & 1u);
      ((uint16_t *)p)[i] = (uint16_t)(u >> 16);
    }
  } else {
    ((float *)p)[i] = v;
  }
}

// scalars layout (floats): [0] loss  [1] loc_reduced  [2] cls_reduced  [3] dir_loss  [4] cls_pos  [5] cls_neg
//                          [6] num_pos(frame 0)  [7] num_neg(frame 0)  [8 .. 8+code) loc_loss_elem
#define LOSS_NSCALAR (8 + LOSS_MAX_CODE)
static_assert(LOSS_NSCALAR == D3D_LOSS_SCALARS, "scalar layout is part of the ABI (include/d3d.h)");

__global__ void __launch_bounds__(256)
    loss_count_kernel(const int32_t *__restrict__ labels, int batch, int A, float *__restrict__ npos,
                      float *__restrict__ scalars) {
  const int b = blockIdx.y;
  int pos = 0, neg = 0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) {
    int l = labels[(int64_t)b * A + a];
    pos += l > 0;
    neg += l == 0;
  }
  __shared__ int sp, sn;
  if (threadIdx.x == 0) sp = sn = 0;
  __syncthreads();
  if (pos) atomicAdd(&sp, pos);
  if (neg) atomicAdd(&sn, neg);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sp) atomicAdd(&npos[b], (float)sp);
    if (b == 0) {
      if (sp) atomicAdd(&scalars[6], (float)sp);
      if (sn) atomicAdd(&scalars[7], (float)sn);
    }
  }
}

__global__ void __launch_bounds__(256)
    loss_main_kernel(const void *__restrict__ box_preds, const void *__restrict__ cls_preds,
                     const void *__restrict__ dir_preds, int bf16, const int32_t *__restrict__ labels,
                     const float *__restrict__ reg_targets, const float *__restrict__ anchors, LossCfg C,
                     const float *__restrict__ npos, void *__restrict__ g_box, void *__restrict__ g_cls,
                     void *__restrict__ g_dir, float *__restrict__ scalars, float *__restrict__ fold_ws,
                     unsigned *__restrict__ counters, int fold_g) {
  __shared__ float rows[4][LOSS_NSCALAR];  // one row per wave
  __shared__ float tot[LOSS_NSCALAR];
  __shared__ int fold_flag;
  __syncthreads();
  float acc[LOSS_NSCALAR];
#pragma unroll
  for (int i = 0; i < LOSS_NSCALAR; ++i) acc[i] = 0.f;
  const float inv_b = 1.f / (float)C.batch;
  const float inv_s2 = 1.f / (C.sigma * C.sigma);
  const int64_t total = (int64_t)C.batch * C.A;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(t / C.A);
    const int label = labels[t];
    const float norm = 1.f / fmaxf(npos[b], 1.f);  // NormByNumPositives: clamp(#positives, min 1)
    const bool is_pos = label > 0;
    // ---- classification: sigmoid focal loss against one-hot(label) without the background column
    const float cw = (label == 0 ? C.neg_w : 0.f) * norm + (is_pos ? C.pos_w : 0.f) * norm;
    float cls_sum = 0.f, cls_col0 = 0.f;
    for (int c = 0; c < C.nc; ++c) {
      const float x = loss_ld(cls_preds, t * C.nc + c, bf16);
      const float tg = (label == c + 1) ? 1.f : 0.f;
      const float e = expf(-fabsf(x));
      const float ce = fmaxf(x, 0.f) - x * tg + log1pf(e);
      const float p = 1.f / (1.f + expf(-x));
      const float pt = tg * p + (1.f - tg) * (1.f - p);
      const float om = 1.f - pt;
      const float mod = C.gamma == 2.f ? om * om : (C.gamma == 0.f ? 1.f : powf(om, C.gamma));
      const float aw = tg * C.alpha + (1.f - tg) * (1.f - C.alpha);
      const float l = mod * aw * ce * cw;
      cls_sum += l;
      if (c == 0) cls_col0 = l;
      // d/dx [ (1-pt)^g * ce ]:  dce/dx = p - tg ;  d(1-pt)/dx = -(2 tg - 1) p (1-p)
      const float dmod = C.gamma == 0.f ? 0.f
                         : (C.gamma == 2.f ? 2.f * om : C.gamma * powf(om, C.gamma - 1.f)) *
                               (-(2.f * tg - 1.f) * p * (1.f - p));
      const float g = aw * cw * (dmod * ce + mod * (p - tg)) * inv_b * C.w_cls;
      loss_st(g_cls, t * C.nc + c, bf16, g);
    }
    acc[2] += cls_sum * inv_b * C.w_cls;
    if (C.nc == 1) {  // _get_pos_neg_loss, single-column branch
      acc[4] += is_pos ? cls_sum * inv_b / C.pos_w : 0.f;
      acc[5] += label == 0 ? cls_sum * inv_b / C.neg_w : 0.f;
    } else {          // multi-column branch: columns 1.. count as "pos", column 0 as "neg"
      acc[4] += (cls_sum - cls_col0) * inv_b / C.pos_w;
      acc[5] += cls_col0 * inv_b / C.neg_w;
    }
    // ---- localisation: smooth L1 (sigma), optional sin-difference on the last code element
    const float rw = is_pos ? norm : 0.f;
    float tgt_r = 0.f;
#pragma unroll
    for (int j = 0; j < LOSS_MAX_CODE; ++j) {  // static trip count: acc[8 + j] stays in registers
      if (j >= C.code) break;
      const float pr = loss_ld(box_preds, t * C.code + j, bf16);
      const float tg = reg_targets[t * C.code + j];
      float diff = pr - tg, ddiff = 1.f;
      if (j == C.code - 1) {
        tgt_r = tg;
        if (C.sin_diff) {  // sin(a) cos(b) - cos(a) sin(b); d/da = cos(a) cos(b) + sin(a) sin(b)
          float sa, ca, sb, cb;
          sincosf(pr, &sa, &ca);
          sincosf(tg, &sb, &cb);
          diff = sa * cb - ca * sb;
          ddiff = ca * cb + sa * sb;
        }
      }
      const float ad = fabsf(diff);
      const bool lt = ad <= inv_s2;
      const float as = ad * C.sigma;
      const float l = (lt ? 0.5f * as * as : ad - 0.5f * inv_s2) * rw;
      acc[1] += l * inv_b * C.w_loc;
      acc[8 + j] += l * inv_b;
      const float dl = lt ? C.sigma * C.sigma * diff : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
      loss_st(g_box, t * C.code + j, bf16, dl * ddiff * rw * inv_b * C.w_loc);
    }
    // ---- direction classifier: softmax CE on [rot_gt - offset wrapped to (-pi, pi] > 0]
    if (C.has_dir) {
      const int a = (int)(t - (int64_t)b * C.A);
      const float ra = anchors[((int64_t)b * C.anchor_bstride + a) * C.nd + C.nd - 1];
      const float v = tgt_r + ra - C.dir_offset;
      const float wrapped = v - floorf(v / (2.f * LOSS_PI) + 0.5f) * (2.f * LOSS_PI);
      const int cls = wrapped > 0.f ? 1 : 0;
      const float x0 = loss_ld(dir_preds, t * 2, bf16), x1 = loss_ld(dir_preds, t * 2 + 1, bf16);
      const float m = fmaxf(x0, x1);
      const float e0 = expf(x0 - m), e1 = expf(x1 - m), s = e0 + e1;
      const float lse = m + logf(s);
      const float w = is_pos ? norm : 0.f;  // (label > 0) / clamp(sum(label > 0), 1)
      const float l = (lse - (cls ? x1 : x0)) * w;
      acc[3] += l * inv_b;
      const float k = w * inv_b * C.w_dir;
      loss_st(g_dir, t * 2, bf16, (e0 / s - (cls ? 0.f : 1.f)) * k);
      loss_st(g_dir, t * 2 + 1, bf16, (e1 / s - (cls ? 1.f : 0.f)) * k);
    }
  }
  acc[0] = acc[1] + acc[2] + acc[3] * C.w_dir;
  // logged scalars: lanes by shuffles, the block's four waves in wave order, the blocks through the fixed-order fold
  // (d3d_fold.h) -- the loss VALUE repeats bit for bit like the gradients do (round 2 summed it with float atomics)
#pragma unroll
  for (int i = 0; i < LOSS_NSCALAR; ++i) {
    float v = (i == 6 || i == 7) ? 0.f : acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) rows[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < LOSS_NSCALAR) tot[threadIdx.x] = rows[0][threadIdx.x] + rows[1][threadIdx.x] + rows[2][threadIdx.x] + rows[3][threadIdx.x];
  __syncthreads();
  if (d3d_fold_block<256>(tot, LOSS_NSCALAR, fold_ws, counters, fold_g, &fold_flag)) {
    const int i = threadIdx.x;
    if (i < LOSS_NSCALAR && i != 6 && i != 7) scalars[i] = fold_ws[i];  // [6], [7]: exact counts from loss_count_kernel
  }
}

#define LOSS_MAX_BLOCKS 1024
extern "C" size_t d3d_head_loss_workspace_bytes(int batch) {
  // per-frame positive counts, then the fold workspace of the logged scalars
  return d3d_align_up((size_t)batch * sizeof(float)) + d3d_fold_ws_floats(LOSS_MAX_BLOCKS, LOSS_NSCALAR) * sizeof(float) + 256;
}

extern "C" int d3d_head_loss(const void *box_preds, const void *cls_preds, const void *dir_preds, int dtype,
                             const int32_t *labels, const float *reg_targets, const float *anchors,
                             int anchors_per_batch, int batch, int num_anchors, int code_size, int num_class,
                             int box_dim, int sin_difference, float dir_offset, float pos_cls_weight,
                             float neg_cls_weight, float focal_alpha, float focal_gamma, float smooth_l1_sigma,
                             float loc_weight, float cls_weight, float dir_weight, void *grad_box,
                             void *grad_cls, void *grad_dir, float *scalars, void *workspace,
                             size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_head_loss: dtype %d", dtype);
  D3D_REQUIRE(batch >= 1 && num_anchors >= 1, "d3d_head_loss: sizes");
  D3D_REQUIRE(code_size >= 1 && code_size <= LOSS_MAX_CODE, "d3d_head_loss: code_size %d (1..%d)", code_size,
              LOSS_MAX_CODE);
  D3D_REQUIRE(num_class >= 1 && num_class <= LOSS_MAX_CLS, "d3d_head_loss: num_class %d (1..%d)", num_class,
              LOSS_MAX_CLS);
  D3D_REQUIRE(box_preds && cls_preds && labels && reg_targets && grad_box && grad_cls && scalars && workspace,
              "d3d_head_loss: null pointer");
  D3D_REQUIRE(!dir_preds || (anchors && grad_dir && box_dim >= 1), "d3d_head_loss: direction head needs anchors");
  D3D_REQUIRE(pos_cls_weight > 0.f && neg_cls_weight > 0.f && smooth_l1_sigma > 0.f,
              "d3d_head_loss: weights / sigma must be positive");
  if (workspace_bytes < d3d_head_loss_workspace_bytes(batch))
    return d3d_set_error(D3D_ECAPACITY, "d3d_head_loss: workspace too small");
  float *npos = (float *)workspace;
  {
    const D3dFillSeg segs[2] = {{npos, 0, (size_t)batch * sizeof(float)},
                                {scalars, 0, (size_t)D3D_LOSS_SCALARS * sizeof(float)}};
    D3D_HIP_TRY(d3d_fill_async(stream, segs, 2));
  }
  int nbx = (num_anchors + 255) / 256;
  if (nbx > 256) nbx = 256;
  hipLaunchKernelGGL(loss_count_kernel, dim3(nbx, batch), dim3(256), 0, stream, labels, batch, num_anchors,
                     npos, scalars);
  D3D_LAUNCH_TRY("loss_count_kernel");
  LossCfg C;
  C.batch = batch, C.A = num_anchors, C.code = code_size, C.nc = num_class, C.nd = box_dim;
  C.anchor_bstride = anchors_per_batch ? num_anchors : 0;
  C.has_dir = dir_preds != nullptr, C.sin_diff = sin_difference;
  C.pos_w = pos_cls_weight, C.neg_w = neg_cls_weight;
  C.alpha = focal_alpha, C.gamma = focal_gamma, C.sigma = smooth_l1_sigma;
  C.dir_offset = dir_offset;
  C.w_loc = loc_weight, C.w_cls = cls_weight, C.w_dir = dir_weight;
  int64_t items = (int64_t)batch * num_anchors;
  int nb = (int)((items + 255) / 256);
  if (nb > LOSS_MAX_BLOCKS) nb = LOSS_MAX_BLOCKS;
  const D3dFoldPlan fp = d3d_fold_plan(nb, LOSS_NSCALAR);
  float *fold_ws = (float *)((char *)workspace + d3d_align_up((size_t)batch * sizeof(float)));
  unsigned *counters = d3d_fold_counters(d3d_fold_counter_words(fp));
  if (!counters) return D3D_EHIP;
  hipLaunchKernelGGL(loss_main_kernel, dim3(nb), dim3(256), 0, stream, box_preds, cls_preds, dir_preds,
                     dtype == D3D_BF16, labels, reg_targets, anchors, C, npos, grad_box, grad_cls, grad_dir,
                     scalars, fold_ws, counters, fp.g);
  D3D_LAUNCH_TRY("loss_main_kernel");
  return D3D_OK;
}
