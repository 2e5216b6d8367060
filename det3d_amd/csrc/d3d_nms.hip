// Rotated / axis-aligned NMS and pairwise rotated BEV IoU for gfx950.
//
// The reference runs rotated NMS on the CPU (D2H copy, boost::geometry polygon
// clipping, serial O(n^2): det3d/core/bbox/box_torch_ops.py:528-549 ->
// det3d/ops/nms/nms_cpu.py:34-45 -> nms_cpu.h:73-169).  Here it stays on the device:
//   1. nms_prep:   corners + stand-up AABB per box (reference float32 arithmetic)
//   2. nms_mask:   64x64 tiles, one wavefront per tile; lane i tests box i against
//                  the tile's 64 column boxes held in LDS and emits one 64-bit word
//                  (a wave is exactly as wide as the reference's bitmask word)
//   3. nms_reduce: a single wavefront walks the boxes in score order, keeps the
//                  `removed` bit-vector in LDS and ORs in mask rows of kept boxes
// Semantics follow the CPU rule the shipped configs execute: skip a pair when the
// stand-up IoU <= 0 (nms_cpu.h:105), suppress when polygon IoU >= thresh (:157).
#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_nms_reduce.h"

struct Quad {
  float x[4], y[4];
};

// center_to_corner_box2d (box_np_ops.py:477-497): corners (-.5,-.5)(-.5,.5)(.5,.5)(.5,-.5)
// scaled by (w,l), rotated clockwise for positive r, then translated.
__device__ __forceinline__ Quad box_quad(const float *b) {
  const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
  const float ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  float s = sinf(b[4]), c = cosf(b[4]);
  Quad q;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float px = b[2] * nx[k], py = b[3] * ny[k];
    q.x[k] = (px * c + py * s) + b[0];
    q.y[k] = (-px * s + py * c) + b[1];
  }
  return q;
}

__device__ __forceinline__ float quad_area(const Quad &q) {
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) & 3;
    a += q.x[i] * q.y[j] - q.x[j] * q.y[i];
  }
  return fabsf(a) * 0.5f;
}

// area of A n B by Sutherland-Hodgman clipping of A against B's four half-planes
__device__ float quad_intersection(const Quad &A, const Quad &B) {
  // 8 vertices is the bound for exactly convex input; after rounding, near-collinear / duplicated boxes can
  // produce extra sign changes per clip edge, so the scratch holds 16 and appends stop there
  constexpr int MAXV = 16;
  float px[MAXV], py[MAXV], tx[MAXV], ty[MAXV];
  int ns = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    px[i] = A.x[i];
    py[i] = A.y[i];
  }
  float sgn = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) & 3;
    sgn += B.x[i] * B.y[j] - B.x[j] * B.y[i];
  }
  const float orient = sgn >= 0.f ? 1.f : -1.f;
  for (int e = 0; e < 4 && ns > 0; ++e) {
    float ax = B.x[e], ay = B.y[e];
    float bx = B.x[(e + 1) & 3], by = B.y[(e + 1) & 3];
    int nt = 0;
    for (int i = 0; i < ns; ++i) {
      int j = (i + 1 == ns) ? 0 : i + 1;
      float dp = orient * ((bx - ax) * (py[i] - ay) - (by - ay) * (px[i] - ax));
      float dq = orient * ((bx - ax) * (py[j] - ay) - (by - ay) * (px[j] - ax));
      if (dp >= 0.f && nt < MAXV) {
        tx[nt] = px[i];
        ty[nt] = py[i];
        ++nt;
      }
      if ((dp >= 0.f) != (dq >= 0.f) && nt < MAXV) {
        float t = dp / (dp - dq);
        tx[nt] = px[i] + t * (px[j] - px[i]);
        ty[nt] = py[i] + t * (py[j] - py[i]);
        ++nt;
      }
    }
    ns = nt;
    for (int i = 0; i < ns; ++i) {
      px[i] = tx[i];
      py[i] = ty[i];
    }
  }
  if (ns < 3) return 0.f;
  float a = 0.f;
  for (int i = 0; i < ns; ++i) {
    int j = (i + 1 == ns) ? 0 : i + 1;
    a += px[i] * py[j] - px[j] * py[i];
  }
  return fabsf(a) * 0.5f;
}

__device__ __forceinline__ float rotate_iou_quads(const Quad &A, const Quad &B) {
  float inter = quad_intersection(A, B);
  if (!(inter > 0.f)) return 0.f;
  float uni = quad_area(A) + quad_area(B) - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// workspace record per box: 8 corner floats + 4 stand-up floats
#define NMS_REC 12

__global__ void __launch_bounds__(256)
    nms_prep_kernel(const float *__restrict__ boxes, int n, float *__restrict__ rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Quad q = box_quad(boxes + 5 * i);
  float x0 = q.x[0], x1 = q.x[0], y0 = q.y[0], y1 = q.y[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    x0 = fminf(x0, q.x[k]);
    x1 = fmaxf(x1, q.x[k]);
    y0 = fminf(y0, q.y[k]);
    y1 = fmaxf(y1, q.y[k]);
  }
  float *r = rec + (int64_t)i * NMS_REC;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[2 * k] = q.x[k];
    r[2 * k + 1] = q.y[k];
  }
  r[8] = x0;
  r[9] = y0;
  r[10] = x1;
  r[11] = y1;
}

// iou_jit(eps = 0) of the stand-up boxes (box_np_ops.py:955-994), float32
__device__ __forceinline__ float standup_iou(const float *a, const float *q) {
  float iw = fminf(a[2], q[2]) - fmaxf(a[0], q[0]);
  if (iw > 0.f) {
    float ih = fminf(a[3], q[3]) - fmaxf(a[1], q[1]);
    if (ih > 0.f) {
      float ua = (a[2] - a[0]) * (a[3] - a[1]) + (q[2] - q[0]) * (q[3] - q[1]) - iw * ih;
      return iw * ih / ua;
    }
  }
  return 0.f;
}

template <bool ROTATED>
__global__ void __launch_bounds__(64)
    nms_mask_kernel(const float *__restrict__ rec, const float *__restrict__ boxes, int n,
                    float thresh, int nblk, unsigned long long *__restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  const int tx = threadIdx.x;
  if (cb < rb) {  // strictly lower tiles are never read by the reduce
    int i = rb * 64 + tx;
    if (i < n) mask[(int64_t)i * nblk + cb] = 0ull;
    return;
  }
  __shared__ float col[64 * NMS_REC];
  const int cj = cb * 64 + tx;
  if (cj < n) {
    if (ROTATED) {
#pragma unroll
      for (int k = 0; k < NMS_REC; ++k) col[tx * NMS_REC + k] = rec[(int64_t)cj * NMS_REC + k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) col[tx * NMS_REC + k] = boxes[(int64_t)cj * 4 + k];
    }
  }
  __syncthreads();
  const int i = rb * 64 + tx;
  if (i >= n) return;
  const int ncol = min(64, n - cb * 64);
  unsigned long long bits = 0ull;
  if (ROTATED) {
    float mine[NMS_REC];
#pragma unroll
    for (int k = 0; k < NMS_REC; ++k) mine[k] = rec[(int64_t)i * NMS_REC + k];
    Quad A;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A.x[k] = mine[2 * k];
      A.y[k] = mine[2 * k + 1];
    }
    for (int j = (rb == cb) ? tx + 1 : 0; j < ncol; ++j) {
      const float *o = col + j * NMS_REC;
      if (standup_iou(mine + 8, o + 8) <= 0.f) continue;
      Quad B;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        B.x[k] = o[2 * k];
        B.y[k] = o[2 * k + 1];
      }
      float inter = quad_intersection(A, B);
      if (!(inter > 0.f)) continue;
      float uni = quad_area(A) + quad_area(B) - inter;
      if (!(uni > 0.f)) continue;
      if (inter / uni >= thresh) bits |= 1ull << j;
    }
  } else {
    float a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = boxes[(int64_t)i * 4 + k];
    const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    for (int j = (rb == cb) ? tx + 1 : 0; j < ncol; ++j) {
      const float *b = col + j * NMS_REC;
      float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, 0.f);
      float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f, 0.f);
      float inter = w * h;
      float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
      if (inter / (sa + sb - inter) > thresh) bits |= 1ull << j;  // nms_gpu.py:96 strict >
    }
  }
  mask[(int64_t)i * nblk + cb] = bits;
}

extern "C" size_t d3d_rotate_nms_workspace_bytes(int n) {
  size_t nblk = (size_t)(n + 63) / 64;
  return d3d_align_up((size_t)n * NMS_REC * 4) + d3d_align_up((size_t)n * nblk * 8) + 256;
}

static int nms_common(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                      int32_t *n_keep, void *workspace, size_t workspace_bytes, hipStream_t stream,
                      bool rotated) {
  D3D_REQUIRE(n >= 0 && max_keep >= 0, "d3d_nms: n/max_keep");
  D3D_REQUIRE(keep && n_keep, "d3d_nms: null output");
  if (n == 0 || max_keep == 0) {
    D3D_HIP_TRY(d3d_fill_async(stream, n_keep, 0, 4));
    return D3D_OK;
  }
  const int nblk = (n + 63) / 64;
  if (nblk > NMS_MAX_BLK)
    return d3d_set_error(D3D_ECAPACITY, "d3d_nms: n=%d exceeds %d boxes", n, NMS_MAX_BLK * 64);
  if (workspace_bytes < d3d_rotate_nms_workspace_bytes(n))
    return d3d_set_error(D3D_ECAPACITY, "d3d_nms: workspace too small");
  D3D_REQUIRE(boxes && workspace, "d3d_nms: null pointer");
  D3dArena A(workspace, workspace_bytes);
  float *rec = A.take<float>((size_t)n * NMS_REC);
  unsigned long long *mask = A.take<unsigned long long>((size_t)n * nblk);
  if (rotated) {
    hipLaunchKernelGGL(nms_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, boxes, n, rec);
    D3D_LAUNCH_TRY("nms_prep_kernel");
    hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(nblk, nblk), dim3(64), 0, stream, rec, boxes, n,
                       thresh, nblk, mask);
  } else {
    hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(nblk, nblk), dim3(64), 0, stream, rec, boxes,
                       n, thresh, nblk, mask);
  }
  D3D_LAUNCH_TRY("nms_mask_kernel");
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, stream, mask, n, nblk, max_keep, keep,
                     n_keep);
  D3D_LAUNCH_TRY("nms_reduce_kernel");
  return D3D_OK;
}

extern "C" int d3d_rotate_nms(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                              int32_t *n_keep, void *workspace, size_t workspace_bytes,
                              d3d_stream_t stream) {
  return nms_common(boxes, n, thresh, max_keep, keep, n_keep, workspace, workspace_bytes,
                    (hipStream_t)stream, true);
}

extern "C" int d3d_nms(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                       int32_t *n_keep, void *workspace, size_t workspace_bytes,
                       d3d_stream_t stream) {
  return nms_common(boxes, n, thresh, max_keep, keep, n_keep, workspace, workspace_bytes,
                    (hipStream_t)stream, false);
}

// ---- batched rotated NMS: S independent box lists (frames x task groups) in one grid.
//   boxes  dev [S, K, nd] f32, list s holds counts[s] <= K boxes SORTED by descending score; the BEV box is
//          columns (0, 1, 3, 4, nd-1) = (x, y, w, l, r) as mg_head.py:1008 picks them
//   keep   dev [S, max_keep] i32 (indices into the list), n_keep dev [S]
__global__ void __launch_bounds__(256)
    nms_prep_batched_kernel(const float *__restrict__ boxes, int nd, const int32_t *__restrict__ counts, int K,
                            float *__restrict__ rec) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counts[s]) return;
  const float *bx = boxes + ((int64_t)s * K + i) * nd;
  const float b5[5] = {bx[0], bx[1], bx[3], bx[4], bx[nd - 1]};
  Quad q = box_quad(b5);
  float x0 = q.x[0], x1 = q.x[0], y0 = q.y[0], y1 = q.y[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    x0 = fminf(x0, q.x[k]);
    x1 = fmaxf(x1, q.x[k]);
    y0 = fminf(y0, q.y[k]);
    y1 = fmaxf(y1, q.y[k]);
  }
  float *r = rec + ((int64_t)s * K + i) * NMS_REC;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[2 * k] = q.x[k];
    r[2 * k + 1] = q.y[k];
  }
  r[8] = x0;
  r[9] = y0;
  r[10] = x1;
  r[11] = y1;
}

// One block = one 64 x 64 tile of (row box, column box) pairs of one list, NMS_MASK_WAVES waves: every wave takes the 64
// rows and a slice of 64 / NMS_MASK_WAVES columns, the partial mask words are OR-ed through LDS.  (Round 4 ran one wave per
// tile with a 64-iteration column loop: the ~540 live tiles of a saturated 4 x 1000-box batch were 2 waves per CU, each
// walking 64 divergent polygon clippings -- 1.2 of the 1.3 ms of the batched NMS.  Same per-pair arithmetic, same bits.)
#define NMS_MASK_WAVES 8
__global__ void __launch_bounds__(64 * NMS_MASK_WAVES)
    nms_mask_batched_kernel(const float *__restrict__ rec_all, const int32_t *__restrict__ counts, int K,
                            float thresh, int nblk_max, unsigned long long *__restrict__ mask_all) {
  const int s = blockIdx.z;
  const int n = counts[s];
  const int nblk = (n + 63) / 64;
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (rb >= nblk || cb >= nblk) return;
  const float *rec = rec_all + (int64_t)s * K * NMS_REC;
  unsigned long long *mask = mask_all + (int64_t)s * K * nblk_max;
  const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (cb < rb) {
    int i = rb * 64 + tx;
    if (wv == 0 && i < n) mask[(int64_t)i * nblk_max + cb] = 0ull;
    return;
  }
  __shared__ float col[64 * NMS_REC];
  __shared__ unsigned long long part[NMS_MASK_WAVES][64];
  for (int e = threadIdx.x; e < 64 * NMS_REC; e += 64 * NMS_MASK_WAVES) {
    const int cj = cb * 64 + e / NMS_REC;
    if (cj < n) col[e] = rec[(int64_t)cj * NMS_REC + e % NMS_REC];
  }
  __syncthreads();
  const int i = rb * 64 + tx;
  const int ncol = min(64, n - cb * 64);
  unsigned long long bits = 0ull;
  if (i < n) {
    float mine[NMS_REC];
#pragma unroll
    for (int k = 0; k < NMS_REC; ++k) mine[k] = rec[(int64_t)i * NMS_REC + k];
    Quad A;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A.x[k] = mine[2 * k];
      A.y[k] = mine[2 * k + 1];
    }
    constexpr int SL = 64 / NMS_MASK_WAVES;
    const int j0 = wv * SL, j1 = min(j0 + SL, ncol);
    for (int j = max(j0, (rb == cb) ? tx + 1 : 0); j < j1; ++j) {
      const float *o = col + j * NMS_REC;
      if (standup_iou(mine + 8, o + 8) <= 0.f) continue;
      Quad B;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        B.x[k] = o[2 * k];
        B.y[k] = o[2 * k + 1];
      }
      float inter = quad_intersection(A, B);
      if (!(inter > 0.f)) continue;
      float uni = quad_area(A) + quad_area(B) - inter;
      if (!(uni > 0.f)) continue;
      if (inter / uni >= thresh) bits |= 1ull << j;
    }
  }
  part[wv][tx] = bits;
  __syncthreads();
  if (wv == 0 && i < n) {
#pragma unroll
    for (int w = 1; w < NMS_MASK_WAVES; ++w) bits |= part[w][tx];
    mask[(int64_t)i * nblk_max + cb] = bits;
  }
}

__global__ void __launch_bounds__(64)
    nms_reduce_batched_kernel(const unsigned long long *__restrict__ mask_all, const int32_t *__restrict__ counts,
                              int K, int nblk_max, int max_keep, int32_t *__restrict__ keep_all,
                              int32_t *__restrict__ n_keep) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int n = counts[s];
  const int nblk = (n + 63) / 64;
  const unsigned long long *mask = mask_all + (int64_t)s * K * nblk_max;
  int32_t *keep = keep_all + (int64_t)s * max_keep;
  const int nk = d3d_nms_reduce_list(mask, n, nblk, nblk_max, max_keep, keep);
  for (int i = nk + lane; i < max_keep; i += 64) keep[i] = -1;
  if (lane == 0) n_keep[s] = nk;
}

extern "C" size_t d3d_rotate_nms_batched_workspace_bytes(int lists, int k) {
  size_t nblk = (size_t)(k + 63) / 64;
  return d3d_align_up((size_t)lists * k * NMS_REC * 4) + d3d_align_up((size_t)lists * k * nblk * 8) + 256;
}

extern "C" int d3d_rotate_nms_batched(const float *boxes, int box_ndim, const int32_t *counts, int lists, int k,
                                      float thresh, int max_keep, int32_t *keep, int32_t *n_keep,
                                      void *workspace, size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(lists >= 0 && k >= 0 && max_keep >= 0 && box_ndim >= 5, "d3d_rotate_nms_batched: sizes");
  if (lists == 0) return D3D_OK;
  D3D_REQUIRE(keep && n_keep && counts, "d3d_rotate_nms_batched: null pointer");
  if (k == 0 || max_keep == 0) {
    D3D_HIP_TRY(d3d_fill_async(stream, n_keep, 0, (size_t)lists * 4));
    return D3D_OK;
  }
  const int nblk = (k + 63) / 64;
  if (nblk > NMS_MAX_BLK)
    return d3d_set_error(D3D_ECAPACITY, "d3d_rotate_nms_batched: k=%d exceeds %d boxes", k, NMS_MAX_BLK * 64);
  if (workspace_bytes < d3d_rotate_nms_batched_workspace_bytes(lists, k))
    return d3d_set_error(D3D_ECAPACITY, "d3d_rotate_nms_batched: workspace too small");
  D3D_REQUIRE(boxes && workspace, "d3d_rotate_nms_batched: null pointer");
  D3dArena A(workspace, workspace_bytes);
  float *rec = A.take<float>((size_t)lists * k * NMS_REC);
  unsigned long long *mask = A.take<unsigned long long>((size_t)lists * k * nblk);
  hipLaunchKernelGGL(nms_prep_batched_kernel, dim3((k + 255) / 256, lists), dim3(256), 0, stream, boxes, box_ndim,
                     counts, k, rec);
  D3D_LAUNCH_TRY("nms_prep_batched_kernel");
  hipLaunchKernelGGL(nms_mask_batched_kernel, dim3(nblk, nblk, lists), dim3(64 * NMS_MASK_WAVES), 0, stream, rec, counts,
                     k, thresh, nblk, mask);
  D3D_LAUNCH_TRY("nms_mask_batched_kernel");
  hipLaunchKernelGGL(nms_reduce_batched_kernel, dim3(lists), dim3(64), 0, stream, mask, counts, k, nblk, max_keep,
                     keep, n_keep);
  D3D_LAUNCH_TRY("nms_reduce_batched_kernel");
  return D3D_OK;
}

__global__ void __launch_bounds__(256)
    rotate_iou_kernel(const float *__restrict__ a, int m, const float *__restrict__ b, int n,
                      float *__restrict__ iou) {
  const int64_t total = (int64_t)m * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(t / n), j = (int)(t - (int64_t)i * n);
    Quad A = box_quad(a + 5 * i), B = box_quad(b + 5 * j);
    iou[t] = rotate_iou_quads(A, B);
  }
}

extern "C" int d3d_rotate_iou(const float *boxes_a, int m, const float *boxes_b, int n, float *iou,
                              d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(m >= 0 && n >= 0, "d3d_rotate_iou: m/n");
  if (m == 0 || n == 0) return D3D_OK;
  D3D_REQUIRE(boxes_a && boxes_b && iou, "d3d_rotate_iou: null pointer");
  int64_t items = (int64_t)m * n;
  int nb = (int)((items + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(rotate_iou_kernel, dim3(nb), dim3(256), 0, stream, boxes_a, m, boxes_b, n,
                     iou);
  D3D_LAUNCH_TRY("rotate_iou_kernel");
  return D3D_OK;
}
