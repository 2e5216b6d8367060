// iou3d (BEV overlap / IoU / NMS on (x1, y1, x2, y2, ry) boxes) and the evaluation-side rotated IoU
// with a criterion, for gfx950.
//
// Reference:
//   det3d/ops/iou3d/src/iou3d_kernel.cu:108-221 (box_overlap, iou_bev), :250-352 (nms kernels),
//   det3d/ops/iou3d/src/iou3d.cpp:73-120 (host keep loop after a D2H copy of the mask),
//   det3d/ops/iou3d/iou3d_utils.py (boxes_iou_bev, nms_gpu, nms_normal_gpu);
//   det3d/datasets/utils/kitti_object_eval_python/rotate_iou.py:18-341 (rotate_iou_gpu_eval, numba-CUDA,
//   which does not exist on ROCm).
// Both restate the reference's own polygon algorithms in float32 with the same operation order (the
// library is built with -ffp-contract=off and correctly rounded division / sqrt), so results differ from
// the CPU oracle only through sinf / cosf / atan2f (<= 2 ulp).  NMS: the same wavefront-wide bitmask
// scheme as d3d_nms.hip (one wave per 64x64 tile, column boxes in LDS, one 64-bit word per lane, upper
// triangle only), keep pass on the device (d3d_nms_reduce.h) instead of on the host.
#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_nms_reduce.h"

#define IOU3D_EPS 1e-8f

struct P2 {
  float x, y;
};

__device__ __forceinline__ float i3_cross(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// segment (p0,p1) against segment (q0,q1): strict straddle test, then the intersection point
__device__ __forceinline__ bool i3_seg(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {
  bool rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
              fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rect) return false;
  float s1 = i3_cross(q0, p1, p0), s2 = i3_cross(p1, q1, p0);
  float s3 = i3_cross(p0, q1, q0), s4 = i3_cross(q1, p1, q0);
  if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
  float s5 = i3_cross(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU3D_EPS) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// a box prepared once per thread: corners (closed ring), centre, cos/sin of -angle for the in-box test
struct I3Box {
  float x1, y1, x2, y2;
  float cx, cy, cn, sn;  // cos(-ry), sin(-ry)
  P2 c[5];
};

__device__ __forceinline__ I3Box i3_prepare(const float *b) {
  I3Box B;
  B.x1 = b[0], B.y1 = b[1], B.x2 = b[2], B.y2 = b[3];
  B.cx = (b[0] + b[2]) / 2, B.cy = (b[1] + b[3]) / 2;
  B.cn = cosf(-b[4]), B.sn = sinf(-b[4]);
  float co = cosf(b[4]), si = sinf(b[4]);
  const float xs[4] = {b[0], b[2], b[2], b[0]}, ys[4] = {b[1], b[1], b[3], b[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    B.c[k].x = (xs[k] - B.cx) * co + (ys[k] - B.cy) * si + B.cx;
    B.c[k].y = -(xs[k] - B.cx) * si + (ys[k] - B.cy) * co + B.cy;
  }
  B.c[4] = B.c[0];
  return B;
}

__device__ __forceinline__ bool i3_inside(const I3Box &B, P2 p) {
  const float margin = 1e-5f;
  float rx = (p.x - B.cx) * B.cn + (p.y - B.cy) * B.sn + B.cx;
  float ry = -(p.x - B.cx) * B.sn + (p.y - B.cy) * B.cn + B.cy;
  return rx > B.x1 - margin && rx < B.x2 + margin && ry > B.y1 - margin && ry < B.y2 + margin;
}

// overlap area: edge-edge intersections + corners of one box inside the other, sorted by polar angle
// around their mean, shoelace fan (iou3d_kernel.cu:108-212)
__device__ float i3_overlap(const I3Box &A, const I3Box &B) {
  P2 pts[16];
  float sx = 0.f, sy = 0.f;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (cnt < 16 && i3_seg(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], x)) {
        sx += x.x;
        sy += x.y;
        pts[cnt++] = x;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (cnt < 16 && i3_inside(A, B.c[k])) {
      sx += B.c[k].x;
      sy += B.c[k].y;
      pts[cnt++] = B.c[k];
    }
    if (cnt < 16 && i3_inside(B, A.c[k])) {
      sx += A.c[k].x;
      sy += A.c[k].y;
      pts[cnt++] = A.c[k];
    }
  }
  if (cnt < 3) return 0.f;  // the fan below is empty or a single zero-area term
  sx /= cnt;
  sy /= cnt;
  float ang[16];
  for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - sy, pts[i].x - sx);
  for (int j = 0; j < cnt - 1; ++j)  // bubble sort, strict >, like the reference (stable for ties)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        float t = ang[i];
        ang[i] = ang[i + 1];
        ang[i + 1] = t;
        P2 p = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = p;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float i3_iou_bev(const I3Box &A, const I3Box &B) {
  float sa = (A.x2 - A.x1) * (A.y2 - A.y1), sb = (B.x2 - B.x1) * (B.y2 - B.y1);
  float so = i3_overlap(A, B);
  return so / fmaxf(sa + sb - so, IOU3D_EPS);
}

__device__ __forceinline__ float i3_iou_normal(const float *a, const float *b) {
  float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  float in = w * h;
  float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return in / fmaxf(sa + sb - in, IOU3D_EPS);
}

__global__ void __launch_bounds__(256)
    iou3d_pairwise_kernel(const float *__restrict__ a, int m, const float *__restrict__ b, int n, int mode,
                          float *__restrict__ out) {
  const int64_t total = (int64_t)m * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(t / n), j = (int)(t - (int64_t)i * n);
    const float *pa = a + 5 * (int64_t)i, *pb = b + 5 * (int64_t)j;
    float v;
    if (mode == D3D_IOU3D_IOU_NORMAL) {
      v = i3_iou_normal(pa, pb);
    } else {
      I3Box A = i3_prepare(pa), B = i3_prepare(pb);
      v = mode == D3D_IOU3D_OVERLAP ? i3_overlap(A, B) : i3_iou_bev(A, B);
    }
    out[t] = v;
  }
}

extern "C" int d3d_iou3d_pairwise(const float *boxes_a, int m, const float *boxes_b, int n, int mode,
                                  float *out, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(m >= 0 && n >= 0, "d3d_iou3d_pairwise: m/n");
  D3D_REQUIRE(mode == D3D_IOU3D_OVERLAP || mode == D3D_IOU3D_IOU_BEV || mode == D3D_IOU3D_IOU_NORMAL,
              "d3d_iou3d_pairwise: mode %d", mode);
  if (m == 0 || n == 0) return D3D_OK;
  D3D_REQUIRE(boxes_a && boxes_b && out, "d3d_iou3d_pairwise: null pointer");
  int64_t items = (int64_t)m * n;
  int nb = (int)((items + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(iou3d_pairwise_kernel, dim3(nb), dim3(256), 0, stream, boxes_a, m, boxes_b, n, mode,
                     out);
  D3D_LAUNCH_TRY("iou3d_pairwise_kernel");
  return D3D_OK;
}

// one wavefront per 64x64 tile of the upper triangle; lane i = row box, columns from LDS
template <bool NORMAL>
__global__ void __launch_bounds__(64)
    iou3d_mask_kernel(const float *__restrict__ boxes, int n, float thresh, int nblk,
                      unsigned long long *__restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x, tx = threadIdx.x;
  if (cb < rb) return;  // never read by the keep pass
  __shared__ float col[64 * 5];
  const int cj = cb * 64 + tx;
  if (cj < n) {
#pragma unroll
    for (int k = 0; k < 5; ++k) col[tx * 5 + k] = boxes[(int64_t)cj * 5 + k];
  }
  __syncthreads();
  const int i = rb * 64 + tx;
  if (i >= n) return;
  const int ncol = min(64, n - cb * 64);
  float mine[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) mine[k] = boxes[(int64_t)i * 5 + k];
  unsigned long long bits = 0ull;
  if (NORMAL) {
    for (int j = (rb == cb) ? tx + 1 : 0; j < ncol; ++j)
      if (i3_iou_normal(mine, col + j * 5) > thresh) bits |= 1ull << j;
  } else {
    const I3Box A = i3_prepare(mine);
    // cheap reject: circumscribed circles apart => the polygons cannot meet and the IoU is exactly 0
    const float ra = 0.5f * sqrtf((A.x2 - A.x1) * (A.x2 - A.x1) + (A.y2 - A.y1) * (A.y2 - A.y1));
    for (int j = (rb == cb) ? tx + 1 : 0; j < ncol; ++j) {
      const float *o = col + j * 5;
      float dx = (o[0] + o[2]) / 2 - A.cx, dy = (o[1] + o[3]) / 2 - A.cy;
      float rb2 = 0.5f * sqrtf((o[2] - o[0]) * (o[2] - o[0]) + (o[3] - o[1]) * (o[3] - o[1]));
      float reach = ra + rb2 + 1e-3f;
      if (dx * dx + dy * dy > reach * reach) continue;  // IoU == 0 never exceeds thresh >= 0
      const I3Box B = i3_prepare(o);
      if (i3_iou_bev(A, B) > thresh) bits |= 1ull << j;
    }
  }
  mask[(int64_t)i * nblk + cb] = bits;
}

extern "C" size_t d3d_iou3d_nms_workspace_bytes(int n) {
  size_t nblk = (size_t)(n + 63) / 64;
  return d3d_align_up((size_t)n * nblk * 8) + 256;
}

extern "C" int d3d_iou3d_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep,
                             int32_t *n_keep, void *workspace, size_t workspace_bytes,
                             d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(n >= 0, "d3d_iou3d_nms: n");
  D3D_REQUIRE(thresh >= 0.f, "d3d_iou3d_nms: negative threshold");
  D3D_REQUIRE(keep && n_keep, "d3d_iou3d_nms: null output");
  if (n == 0) {
    D3D_HIP_TRY(d3d_fill_async(stream, n_keep, 0, 4));
    return D3D_OK;
  }
  const int nblk = (n + 63) / 64;
  if (nblk > NMS_MAX_BLK)
    return d3d_set_error(D3D_ECAPACITY, "d3d_iou3d_nms: n=%d exceeds %d boxes", n, NMS_MAX_BLK * 64);
  if (workspace_bytes < d3d_iou3d_nms_workspace_bytes(n))
    return d3d_set_error(D3D_ECAPACITY, "d3d_iou3d_nms: workspace too small");
  D3D_REQUIRE(boxes && workspace, "d3d_iou3d_nms: null pointer");
  D3dArena A(workspace, workspace_bytes);
  unsigned long long *mask = A.take<unsigned long long>((size_t)n * nblk);
  if (normal)
    hipLaunchKernelGGL(iou3d_mask_kernel<true>, dim3(nblk, nblk), dim3(64), 0, stream, boxes, n, thresh,
                       nblk, mask);
  else
    hipLaunchKernelGGL(iou3d_mask_kernel<false>, dim3(nblk, nblk), dim3(64), 0, stream, boxes, n, thresh,
                       nblk, mask);
  D3D_LAUNCH_TRY("iou3d_mask_kernel");
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, stream, mask, n, nblk, n, keep, n_keep);
  D3D_LAUNCH_TRY("nms_reduce_kernel");
  return D3D_OK;
}

// ------------------------------------------------------------------ evaluation rotated IoU
// rotate_iou.py: corners (rbbox_to_corners :203-224), point_in_quadrilateral :158-175,
// line_segment_intersection :77-117, sort_vertex_in_convex_polygon :37-74, area :23-34.
struct EvQuad {
  float c[8];
};

__device__ __forceinline__ EvQuad ev_corners(const float *r) {
  float co = cosf(r[4]), si = sinf(r[4]);
  const float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2};
  const float ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
  EvQuad q;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q.c[2 * i] = co * xs[i] + si * ys[i] + r[0];
    q.c[2 * i + 1] = -si * xs[i] + co * ys[i] + r[1];
  }
  return q;
}

__device__ __forceinline__ bool ev_in_quad(float px, float py, const float *c) {
  float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  float ap0 = px - c[0], ap1 = py - c[1];
  float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

__device__ __forceinline__ bool ev_seg(const float *p1, const float *p2, int i, int j, float &tx, float &ty) {
  const int i2 = (i + 1) & 3, j2 = (j + 1) & 3;
  float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * i2], B1 = p1[2 * i2 + 1];
  float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * j2], D1 = p2[2 * j2 + 1];
  float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  bool acd = DA1 * CA0 > CA1 * DA0;
  bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd == bcd) return false;
  bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  float DC0 = D0 - C0, DC1 = D1 - C1, ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
  float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
  tx = Dx / DH;
  ty = Dy / DH;
  return true;
}

__device__ float ev_inter(const EvQuad &Q1, const EvQuad &Q2) {
  float px[8], py[8];
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (n < 8 && ev_in_quad(Q1.c[2 * i], Q1.c[2 * i + 1], Q2.c)) {
      px[n] = Q1.c[2 * i];
      py[n] = Q1.c[2 * i + 1];
      ++n;
    }
    if (n < 8 && ev_in_quad(Q2.c[2 * i], Q2.c[2 * i + 1], Q1.c)) {
      px[n] = Q2.c[2 * i];
      py[n] = Q2.c[2 * i + 1];
      ++n;
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float tx, ty;
      if (n < 8 && ev_seg(Q1.c, Q2.c, i, j, tx, ty)) {  // the reference's scratch holds 8 points
        px[n] = tx;
        py[n] = ty;
        ++n;
      }
    }
  if (n < 3) return 0.f;
  float cx = 0.f, cy = 0.f, vs[8];
  for (int i = 0; i < n; ++i) {
    cx += px[i];
    cy += py[i];
  }
  cx /= n;
  cy /= n;
  for (int i = 0; i < n; ++i) {
    float v0 = px[i] - cx, v1 = py[i] - cy;
    float d = sqrtf(v0 * v0 + v1 * v1);
    v0 = v0 / d;
    v1 = v1 / d;
    if (v1 < 0.f) v0 = -2.f - v0;
    vs[i] = v0;
  }
  for (int i = 1; i < n; ++i)
    if (vs[i - 1] > vs[i]) {
      float t = vs[i], tx = px[i], ty = py[i];
      int j = i;
      while (j > 0 && vs[j - 1] > t) {
        vs[j] = vs[j - 1];
        px[j] = px[j - 1];
        py[j] = py[j - 1];
        --j;
      }
      vs[j] = t;
      px[j] = tx;
      py[j] = ty;
    }
  float area = 0.f;
  for (int i = 0; i < n - 2; ++i)
    area += fabsf(((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) - (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) / 2.0f);
  return area;
}

__global__ void __launch_bounds__(256)
    rotate_iou_eval_kernel(const float *__restrict__ boxes, int n, const float *__restrict__ query, int k,
                           int criterion, float *__restrict__ out) {
  const int64_t total = (int64_t)n * k;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(t / k), j = (int)(t - (int64_t)i * k);
    const float *r1 = query + 5 * (int64_t)j, *r2 = boxes + 5 * (int64_t)i;  // devRotateIoUEval(query, box)
    float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
    float in = ev_inter(ev_corners(r1), ev_corners(r2));
    float v = criterion == -1 ? in / (a1 + a2 - in) : criterion == 0 ? in / a1 : criterion == 1 ? in / a2 : in;
    out[t] = v;
  }
}

extern "C" int d3d_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion,
                                   float *out, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(n >= 0 && k >= 0, "d3d_rotate_iou_eval: n/k");
  D3D_REQUIRE(criterion >= -1 && criterion <= 2, "d3d_rotate_iou_eval: criterion %d", criterion);
  if (n == 0 || k == 0) return D3D_OK;
  D3D_REQUIRE(boxes && query_boxes && out, "d3d_rotate_iou_eval: null pointer");
  int64_t items = (int64_t)n * k;
  int nb = (int)((items + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3(nb), dim3(256), 0, stream, boxes, n, query_boxes, k,
                     criterion, out);
  D3D_LAUNCH_TRY("rotate_iou_eval_kernel");
  return D3D_OK;
}
