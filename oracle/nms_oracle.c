/*
 * TEST INFRASTRUCTURE -- CPU oracle, not product code.
 *
 * Restatement of the reference rotated NMS
 *   det3d/ops/nms/nms_cpu.py:34-45      rotate_nms_cc: corners, standup boxes,
 *                                       standup IoU matrix (iou_jit, eps = 0)
 *   det3d/ops/nms/nms_cpu.h:73-169      rotate_non_max_suppression_cpu: greedy
 *                                       loop, `standup_iou <= 0` skip (:105),
 *                                       polygon IoU, suppress if IoU >= thresh (:157)
 *   det3d/core/bbox/box_np_ops.py:267-297,419-432,477-497  corner construction
 *   det3d/core/bbox/box_np_ops.py:955-994                 iou_jit
 * and of the axis-aligned NMS det3d/ops/nms/nms_gpu.py:22-33,109-126
 * (+1 pixel convention, strict `>`).
 *
 * PARITY UNPINNED for the polygon step: the reference uses boost::geometry
 * (absent here, and the reference ships no test vectors).  Intersection area
 * is computed by exact convex clipping (Sutherland-Hodgman) in double on the
 * float32 corners; union = |A| + |B| - |A n B| (what boost's union_ yields for
 * two overlapping convex quads).  Cross-checked in tests against two other
 * formulations that live in the reference tree (the numba-CUDA rotated IoU run
 * as plain Python -> tests/golden, and detectron2's box_iou_rotated_utils.h
 * compiled into oracle/_ref).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* boxes [n,5] (x, y, w, l, r) -> corners [n,4,2], float32 arithmetic like numpy */
void d3d_oracle_box_corners(const float *boxes, int n, float *corners) {
  static const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
  static const float ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  for (int i = 0; i < n; ++i) {
    const float *b = boxes + i * 5;
    float s = sinf(b[4]), c = cosf(b[4]);
    for (int k = 0; k < 4; ++k) {
      float px = b[2] * nx[k], py = b[3] * ny[k];
      /* rotation_2d: x' = x cos + y sin ; y' = -x sin + y cos (clockwise for +r) */
      corners[(i * 4 + k) * 2 + 0] = (px * c + py * s) + b[0];
      corners[(i * 4 + k) * 2 + 1] = (-px * s + py * c) + b[1];
    }
  }
}

static void standup(const float *c, float *s) {
  s[0] = s[2] = c[0];
  s[1] = s[3] = c[1];
  for (int k = 1; k < 4; ++k) {
    if (c[2 * k] < s[0]) s[0] = c[2 * k];
    if (c[2 * k] > s[2]) s[2] = c[2 * k];
    if (c[2 * k + 1] < s[1]) s[1] = c[2 * k + 1];
    if (c[2 * k + 1] > s[3]) s[3] = c[2 * k + 1];
  }
}

/* iou_jit with eps = 0, float32 */
static float standup_iou(const float *a, const float *q) {
  float box_area = (q[2] - q[0]) * (q[3] - q[1]);
  float iw = fminf(a[2], q[2]) - fmaxf(a[0], q[0]);
  if (iw > 0) {
    float ih = fminf(a[3], q[3]) - fmaxf(a[1], q[1]);
    if (ih > 0) {
      float ua = (a[2] - a[0]) * (a[3] - a[1]) + box_area - iw * ih;
      return iw * ih / ua;
    }
  }
  return 0.f;
}

static double poly_area(const double *p, int n) {
  double a = 0;
  for (int i = 0; i < n; ++i) {
    int j = (i + 1) % n;
    a += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
  }
  return fabs(a) * 0.5;
}

/* area of (quad A) n (quad B), convex clipping in double */
double d3d_oracle_quad_intersection(const float *ca, const float *cb) {
  double subj[32], tmp[32];
  int ns = 4;
  for (int i = 0; i < 8; ++i) subj[i] = ca[i];
  /* orientation of the clip polygon */
  double cl[8];
  for (int i = 0; i < 8; ++i) cl[i] = cb[i];
  double sgn = 0;
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) % 4;
    sgn += cl[2 * i] * cl[2 * j + 1] - cl[2 * j] * cl[2 * i + 1];
  }
  double orient = sgn >= 0 ? 1.0 : -1.0;
  for (int e = 0; e < 4 && ns > 0; ++e) {
    double ax = cl[2 * e], ay = cl[2 * e + 1];
    double bx = cl[2 * ((e + 1) % 4)], by = cl[2 * ((e + 1) % 4) + 1];
    int nt = 0;
    for (int i = 0; i < ns; ++i) {
      double px = subj[2 * i], py = subj[2 * i + 1];
      double qx = subj[2 * ((i + 1) % ns)], qy = subj[2 * ((i + 1) % ns) + 1];
      double dp = orient * ((bx - ax) * (py - ay) - (by - ay) * (px - ax));
      double dq = orient * ((bx - ax) * (qy - ay) - (by - ay) * (qx - ax));
      if (dp >= 0 && nt < 16) {
        tmp[2 * nt] = px;
        tmp[2 * nt + 1] = py;
        ++nt;
      }
      if ((dp >= 0) != (dq >= 0) && nt < 16) {
        double t = dp / (dp - dq);
        tmp[2 * nt] = px + t * (qx - px);
        tmp[2 * nt + 1] = py + t * (qy - py);
        ++nt;
      }
    }
    ns = nt;
    memcpy(subj, tmp, sizeof(double) * 2 * ns);
  }
  if (ns < 3) return 0.0;
  return poly_area(subj, ns);
}

double d3d_oracle_quad_area(const float *c) {
  double p[8];
  for (int i = 0; i < 8; ++i) p[i] = c[i];
  return poly_area(p, 4);
}

/* rotated IoU of two (x,y,w,l,r) boxes, polygon route only (no prefilter) */
double d3d_oracle_rotate_iou_pair(const float *a, const float *b) {
  float ca[8], cb[8];
  d3d_oracle_box_corners(a, 1, ca);
  d3d_oracle_box_corners(b, 1, cb);
  double inter = d3d_oracle_quad_intersection(ca, cb);
  double uni = d3d_oracle_quad_area(ca) + d3d_oracle_quad_area(cb) - inter;
  return uni > 0 ? inter / uni : 0.0;
}

void d3d_oracle_rotate_iou(const float *a, int m, const float *b, int n, double *iou) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) iou[(size_t)i * n + j] = d3d_oracle_rotate_iou_pair(a + 5 * i, b + 5 * j);
}

/*
 * boxes [n,5]; order [n] = indices by descending score; keep [n] out.
 * min_margin (optional): smallest |IoU - thresh| met on a decided pair, so a
 * test can tell a genuine mismatch from a threshold tie.
 */
int d3d_oracle_rotate_nms(const float *boxes, const int32_t *order, int n, float thresh,
                          int32_t *keep, double *min_margin) {
  float *corners = (float *)malloc(sizeof(float) * 8 * (size_t)n);
  float *stand = (float *)malloc(sizeof(float) * 4 * (size_t)n);
  char *suppressed = (char *)calloc((size_t)n, 1);
  d3d_oracle_box_corners(boxes, n, corners);
  for (int i = 0; i < n; ++i) standup(corners + 8 * i, stand + 4 * i);
  int nk = 0;
  double mm = 1e30;
  for (int _i = 0; _i < n; ++_i) {
    int i = order[_i];
    if (suppressed[i]) continue;
    keep[nk++] = i;
    for (int _j = _i + 1; _j < n; ++_j) {
      int j = order[_j];
      if (suppressed[j]) continue;
      if (standup_iou(stand + 4 * i, stand + 4 * j) <= 0.0f) continue;
      double inter = d3d_oracle_quad_intersection(corners + 8 * i, corners + 8 * j);
      if (inter <= 0.0) continue; /* poly_inter.empty() */
      double uni = d3d_oracle_quad_area(corners + 8 * i) + d3d_oracle_quad_area(corners + 8 * j) - inter;
      if (uni <= 0.0) continue;
      float overlap = (float)(inter / uni);
      double mg = fabs((double)overlap - (double)thresh);
      if (mg < mm) mm = mg;
      if (overlap >= thresh) suppressed[j] = 1;
    }
  }
  if (min_margin) *min_margin = mm;
  free(corners);
  free(stand);
  free(suppressed);
  return nk;
}

/* axis-aligned NMS, nms_gpu.py:22-33 (+1), :96 (`>`), :109-126 greedy reduce.
 * boxes [n,4] sorted by descending score already (nms_gpu sorts before). */
int d3d_oracle_nms(const float *boxes, int n, float thresh, int32_t *keep) {
  char *suppressed = (char *)calloc((size_t)n, 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (suppressed[i]) continue;
    keep[nk++] = i;
    const float *a = boxes + 4 * i;
    for (int j = i + 1; j < n; ++j) {
      if (suppressed[j]) continue;
      const float *b = boxes + 4 * j;
      float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
      float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
      float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
      float inter = width * height;
      float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
      float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
      if (inter / (sa + sb - inter) > thresh) suppressed[j] = 1;
    }
  }
  free(suppressed);
  return nk;
}
