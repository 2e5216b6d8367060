/* TEST INFRASTRUCTURE -- CPU restatement (plain C, float32 like the originals) of
 *   (1) the reference's iou3d BEV overlap / IoU / NMS: det3d/ops/iou3d/src/iou3d_kernel.cu:13-304,
 *       host loop det3d/ops/iou3d/src/iou3d.cpp:73-120;  boxes are (x1, y1, x2, y2, ry)
 *   (2) the evaluation-side rotated IoU with a criterion:
 *       det3d/datasets/utils/kitti_object_eval_python/rotate_iou.py:18-263 (devRotateIoUEval), boxes are
 *       (cx, cy, dx, dy, r) with r clockwise-positive.
 * Pinned against the reference itself: (1) oracle/_ref/d3d_ref_iou3d (the .cu device functions compiled for
 * the host), (2) tests/golden/rotate_iou_eval.npz (the numba device functions run as Python).
 * Never imported by the product (det3d_amd/). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IOU3D_EPS 1e-8f

typedef struct {
  float x, y;
} P2;

/* cross(p1, p2, p0) of iou3d_kernel.cu:38-40 */
static float cross3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

/* :67-96 segment (p0,p1) x segment (q0,q1) */
static int seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2 *ans) {
  int rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
             fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rect) return 0;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU3D_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* :50-65 point inside the (un-rotated) box after rotating the point back by -angle, 1e-5 margin */
static int in_box(const float *box, P2 p) {
  const float margin = 1e-5f;
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float c = cosf(-box[4]), s = sinf(-box[4]);
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > box[0] - margin && rx < box[2] + margin && ry > box[1] - margin && ry < box[3] + margin;
}

static void corners_of(const float *b, P2 *c) {
  P2 ctr = {(b[0] + b[2]) / 2, (b[1] + b[3]) / 2};
  float co = cosf(b[4]), si = sinf(b[4]);
  const float xs[4] = {b[0], b[2], b[2], b[0]}, ys[4] = {b[1], b[1], b[3], b[3]};
  for (int k = 0; k < 4; ++k) { /* rotate_around_center :98-102 */
    c[k].x = (xs[k] - ctr.x) * co + (ys[k] - ctr.y) * si + ctr.x;
    c[k].y = -(xs[k] - ctr.x) * si + (ys[k] - ctr.y) * co + ctr.y;
  }
  c[4] = c[0];
}

/* box_overlap :108-212 */
float d3d_oracle_iou3d_overlap(const float *a, const float *b) {
  P2 ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  int cnt = 0;
  corners_of(a, ca);
  corners_of(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_x(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        ctr.x += pts[cnt].x;
        ctr.y += pts[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, cb[k])) {
      ctr.x += cb[k].x;
      ctr.y += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (in_box(b, ca[k])) {
      ctr.x += ca[k].x;
      ctr.y += ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  ctr.x /= cnt; /* cnt == 0 -> NaN centre, loops below do not run, area 0 (as in the reference) */
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j) /* bubble sort by polar angle around the centre :183-191 */
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
        P2 t = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

/* iou_bev :214-221 */
float d3d_oracle_iou3d_iou_bev(const float *a, const float *b) {
  float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  float so = d3d_oracle_iou3d_overlap(a, b);
  return so / fmaxf(sa + sb - so, IOU3D_EPS);
}

/* iou_normal :295-304 (angle ignored) */
float d3d_oracle_iou3d_iou_normal(const float *a, const float *b) {
  float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  float in = w * h, sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return in / fmaxf(sa + sb - in, IOU3D_EPS);
}

/* mode 0: overlap area, 1: iou_bev, 2: iou_normal */
void d3d_oracle_iou3d_pairwise(const float *a, int m, const float *b, int n, int mode, float *out) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      const float *p = a + 5 * i, *q = b + 5 * j;
      out[(size_t)i * n + j] = mode == 0 ? d3d_oracle_iou3d_overlap(p, q)
                               : mode == 1 ? d3d_oracle_iou3d_iou_bev(p, q) : d3d_oracle_iou3d_iou_normal(p, q);
    }
}

/* nms_gpu / nms_normal_gpu: boxes sorted by descending score, suppress j > i when IoU > thresh (kernel :284 /
 * :340, host loop iou3d.cpp:101-116).  margin <- smallest |IoU - thresh| seen (tie detector for the tests). */
int d3d_oracle_iou3d_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep, float *margin) {
  char *gone = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  float mg = 1e30f;
  for (int i = 0; i < n; ++i) {
    if (gone[i]) continue;
    keep[nk++] = i;
    for (int j = i + 1; j < n; ++j) {
      if (gone[j]) continue;
      float v = normal ? d3d_oracle_iou3d_iou_normal(boxes + 5 * i, boxes + 5 * j)
                       : d3d_oracle_iou3d_iou_bev(boxes + 5 * i, boxes + 5 * j);
      float d = fabsf(v - thresh);
      if (d < mg) mg = d;
      if (v > thresh) gone[j] = 1;
    }
  }
  free(gone);
  if (margin) *margin = mg;
  return nk;
}

/* ---------------------------------------------------------------- evaluation rotated IoU (rotate_iou.py) */
/* rbbox_to_corners :203-224 */
static void eval_corners(const float *r, float *c) {
  float co = cosf(r[4]), si = sinf(r[4]);
  const float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2}, ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = co * xs[i] + si * ys[i] + r[0];
    c[2 * i + 1] = -si * xs[i] + co * ys[i] + r[1];
  }
}

/* point_in_quadrilateral :158-175 */
static int eval_in_quad(float px, float py, const float *c) {
  float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  float ap0 = px - c[0], ap1 = py - c[1];
  float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

/* line_segment_intersection :77-117 */
static int eval_seg_x(const float *p1, const float *p2, int i, int j, float *t) {
  float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
  float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
  float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  int acd = DA1 * CA0 > CA1 * DA0;
  int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd == bcd) return 0;
  int abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return 0;
  float DC0 = D0 - C0, DC1 = D1 - C1, ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
  float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
  t[0] = Dx / DH;
  t[1] = Dy / DH;
  return 1;
}

/* inter :227-243 = quadrilateral_intersection :178-200 + sort_vertex_in_convex_polygon :37-74 + area :23-34.
 * The reference's scratch holds 8 points; more than 8 would overrun it there, so we stop at 8 too. */
float d3d_oracle_rotate_inter_eval(const float *r1, const float *r2) {
  float c1[8], c2[8], pts[16 + 16], tmp[2];
  int n = 0;
  eval_corners(r1, c1);
  eval_corners(r2, c2);
  for (int i = 0; i < 4; ++i) {
    if (eval_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
    if (eval_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (n < 16 && eval_seg_x(c1, c2, i, j, tmp)) { pts[2 * n] = tmp[0]; pts[2 * n + 1] = tmp[1]; ++n; }
  if (n > 8) n = 8;
  if (n > 0) {
    float cx = 0.f, cy = 0.f, vs[16];
    for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
    cx /= n;
    cy /= n;
    for (int i = 0; i < n; ++i) {
      float v0 = pts[2 * i] - cx, v1 = pts[2 * i + 1] - cy;
      float d = sqrtf(v0 * v0 + v1 * v1);
      v0 = v0 / d;
      v1 = v1 / d;
      if (v1 < 0) v0 = -2 - v0;
      vs[i] = v0;
    }
    for (int i = 1; i < n; ++i) /* insertion sort, ascending pseudo-angle */
      if (vs[i - 1] > vs[i]) {
        float t = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
        int j = i;
        while (j > 0 && vs[j - 1] > t) {
          vs[j] = vs[j - 1];
          pts[2 * j] = pts[2 * j - 2];
          pts[2 * j + 1] = pts[2 * j - 1];
          --j;
        }
        vs[j] = t;
        pts[2 * j] = tx;
        pts[2 * j + 1] = ty;
      }
  }
  float area = 0.f;
  for (int i = 0; i < n - 2; ++i) { /* triangle fan from vertex 0 */
    const float *a = pts, *b = pts + 2 * i + 2, *c = pts + 2 * i + 4;
    area += fabsf(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f);
  }
  return area;
}

/* rotate_iou_gpu_eval :305-341: out[i, j] = devRotateIoUEval(query[j], boxes[i], criterion) (:246-258) */
void d3d_oracle_rotate_iou_eval(const float *boxes, int n, const float *query, int k, int criterion, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < k; ++j) {
      const float *r1 = query + 5 * j, *r2 = boxes + 5 * i;
      float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3], in = d3d_oracle_rotate_inter_eval(r1, r2);
      float v = criterion == -1 ? in / (a1 + a2 - in) : criterion == 0 ? in / a1 : criterion == 1 ? in / a2 : in;
      out[(size_t)i * k + j] = v;
    }
}
