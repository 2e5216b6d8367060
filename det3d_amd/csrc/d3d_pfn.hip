// PillarFeatureNet with a single (last) PFNLayer, fused for gfx950.
//
// Reference: det3d/models/readers/pillar_encoder.py:115-155 (decorate: cluster offsets, pillar-centre
// offsets, padding mask) and :42-57 (Linear(K->F, no bias) -> BatchNorm1d over all M*T rows -> ReLU
// -> max over T); ~15 PyTorch kernels over the padded [M, T, K] tensor (43 MB per pass at KITTI).
// Here nothing of size [M, T, *] is materialised: one wavefront per pillar, lane = output channel,
// the lane keeps its weight row W[c][0..K) in registers and walks only the pillar's REAL points
// (padded rows are exact zeros after the mask: they add nothing to the BN sums but count in N = M*T,
// and they compete in the max with relu(shift)).
//   fwd  : pfn_stats (sum, sumsq per channel)  ->  fold  ->  pfn_apply (normalise, ReLU, max/argmax over T)
//   bwd  : pfn_bwd_reduce (dbeta, dgamma from the argmax rows)  ->  fold  ->  pfn_bwd_weight (dW)  ->  fold
// Round 3: the per-channel sums are deterministic.  Every wave leaves ONE slab of partial sums (it walks its pillars in
// a fixed order) and pfn_fold_kernel adds the slabs in ascending wave order; rounds 1-2 added them with float atomics
// (2048 waves x 640 same-address atomics made pfn_bwd_weight the slowest launch of the PointPillars step: 620 us).
// Round 4: 8192 waves (pfn_blocks): the kernels are bound by per-pillar latency, not by bytes.
// All f32 (the reader runs before any bf16 cast); HBM-bound on reading the voxel rows once per pass.
#include <stdlib.h>

#include "d3d_common.h"
#include "d3d_fill.h"

#define PFN_MAXK 16

struct PfnGeom {
  int m, t, c, k, f;  // pillars, max points, point channels, K = c + 5, out channels
  float vx, vy, xo, yo;
};

// decorated features of point p of a pillar (pillar_encoder.py:119-137)
template <int C>
__device__ __forceinline__ void pfn_decorate(const float *pt, const float *mean3, float cx, float cy,
                                             float *feat) {
#pragma unroll
  for (int j = 0; j < C; ++j) feat[j] = pt[j];
  feat[C + 0] = pt[0] - mean3[0];
  feat[C + 1] = pt[1] - mean3[1];
  feat[C + 2] = pt[2] - mean3[2];
  feat[C + 3] = pt[0] - cx;
  feat[C + 4] = pt[1] - cy;
}

template <int C>
__device__ __forceinline__ void pfn_pillar_setup(const PfnGeom &G, const float *vox,
                                                 const int32_t *num_points, const int4 *coors,
                                                 int pillar, int &n, float *mean3, float &cx,
                                                 float &cy) {
  n = num_points[pillar];
  n = n < G.t ? n : G.t;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const float *p = vox + (int64_t)pillar * G.t * C;
  for (int i = 0; i < n; ++i) {
    s0 += p[i * C + 0];
    s1 += p[i * C + 1];
    s2 += p[i * C + 2];
  }
  const float fn = (float)num_points[pillar];
  mean3[0] = s0 / fn;
  mean3[1] = s1 / fn;
  mean3[2] = s2 / fn;
  int4 q = coors[pillar];
  cx = (float)q.w * G.vx + G.xo;
  cy = (float)q.z * G.vy + G.yo;
}

// grid-stride over pillars, one wave per pillar, lane = channel (f <= 64)
template <int C>
__global__ void __launch_bounds__(256)
    pfn_stats_kernel(PfnGeom G, const float *__restrict__ vox, const int32_t *__restrict__ num_points,
                     const int4 *__restrict__ coors, const float *__restrict__ weight,
                     float *__restrict__ slabs) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  constexpr int K = C + 5;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = lane < G.f ? weight[lane * K + j] : 0.f;
  float s = 0.f, q = 0.f;
  for (int pillar = wave; pillar < G.m; pillar += nwaves) {
    int n;
    float mean3[3], cx, cy;
    pfn_pillar_setup<C>(G, vox, num_points, coors, pillar, n, mean3, cx, cy);
    const float *p = vox + (int64_t)pillar * G.t * C;
    for (int i = 0; i < n; ++i) {
      float feat[K];
      pfn_decorate<C>(p + i * C, mean3, cx, cy, feat);
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) x = fmaf(w[j], feat[j], x);
      s += x;
      q = fmaf(x, x, q);
    }
  }
  if (lane < G.f) {  // slab of this wave: [sum f | sumsq f]
    slabs[(int64_t)wave * 2 * G.f + lane] = s;
    slabs[(int64_t)wave * 2 * G.f + G.f + lane] = q;
  }
}

// out[i] = sum over slabs w = 0 .. nslabs-1 (ascending) of slabs[w][i]: 16 waves per block share the slab range, each
// adds its contiguous share in order, the shares are combined in wave order
#define PFN_FOLD_WAVES 16
// blockIdx.y = segment of the slab range (gridDim.y segments of `seg` slabs; the last may be short): segment y writes row y
// of `out` ([gridDim.y][nval]) -- the first level of a two-level fold -- or, with one segment, the final vectors
__global__ void __launch_bounds__(64 * PFN_FOLD_WAVES)
    pfn_fold_kernel(const float *__restrict__ slabs, int nslabs, int nval, float *__restrict__ out, int split,
                    float *__restrict__ out2, int seg) {  // values i >= split go to out2[i - split]
  __shared__ float part[PFN_FOLD_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = (int)blockIdx.x * 64 + lane;
  const int s0 = (int)blockIdx.y * seg;
  const int s1 = (s0 + seg < nslabs) ? s0 + seg : nslabs;
  const int per = (s1 - s0 + PFN_FOLD_WAVES - 1) / PFN_FOLD_WAVES;
  const int w0 = s0 + wave * per, w1 = (w0 + per < s1) ? w0 + per : s1;
  if (gridDim.y > 1) out += (size_t)blockIdx.y * nval;
  float s = 0.f;
  if (i < nval) {
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slabs[(int64_t)(w + j) * nval + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; w < w1; ++w) s += slabs[(int64_t)w * nval + i];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < nval) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < PFN_FOLD_WAVES; ++j) t += part[j][lane];
    if (i < split) out[i] = t;
    else out2[i - split] = t;
  }
}

template <int C>
__global__ void __launch_bounds__(256)
    pfn_apply_kernel(PfnGeom G, const float *__restrict__ vox, const int32_t *__restrict__ num_points,
                     const int4 *__restrict__ coors, const float *__restrict__ weight,
                     const float *__restrict__ gamma, const float *__restrict__ beta,
                     float *running_mean, float *running_var, float momentum, float eps, int training,
                     const float *__restrict__ stats, float *__restrict__ out,
                     int32_t *__restrict__ argmax, float *save_mean, float *save_invstd) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool on = lane < G.f;
  const float ntot = (float)G.m * (float)G.t;
  float mean = 0.f, var = 1.f;
  if (on) {
    if (training) {
      mean = stats[lane] / ntot;
      var = fmaxf(stats[G.f + lane] / ntot - mean * mean, 0.f);
    } else {
      mean = running_mean[lane];
      var = running_var[lane];
    }
  }
  const float invstd = rsqrtf(var + eps);
  const float g = (on && gamma) ? gamma[lane] : 1.f, b = (on && beta) ? beta[lane] : 0.f;
  const float sc = g * invstd, sh = b - mean * g * invstd;
  if (on && wave == 0) {
    if (save_mean) save_mean[lane] = mean;
    if (save_invstd) save_invstd[lane] = invstd;
    if (training && running_mean) {
      float unbiased = ntot > 1.f ? var * (ntot / (ntot - 1.f)) : var;
      running_mean[lane] = (1.f - momentum) * running_mean[lane] + momentum * mean;
      running_var[lane] = (1.f - momentum) * running_var[lane] + momentum * unbiased;
    }
  }
  constexpr int K = C + 5;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = on ? weight[lane * K + j] : 0.f;
  const float pad_y = fmaxf(sh, 0.f);  // a padded row: x = 0 -> relu(shift)
  for (int pillar = wave; pillar < G.m; pillar += nwaves) {
    int n;
    float mean3[3], cx, cy;
    pfn_pillar_setup<C>(G, vox, num_points, coors, pillar, n, mean3, cx, cy);
    const float *p = vox + (int64_t)pillar * G.t * C;
    float best = -1.f;  // relu output is >= 0, so any row beats it
    int arg = 0;
    for (int i = 0; i < n; ++i) {
      float feat[K];
      pfn_decorate<C>(p + i * C, mean3, cx, cy, feat);
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) x = fmaf(w[j], feat[j], x);
      float y = fmaxf(fmaf(x, sc, sh), 0.f);
      if (y > best) {  // first occurrence wins, like torch.max
        best = y;
        arg = i;
      }
    }
    if (n < G.t && pad_y > best) {
      best = pad_y;
      arg = n;  // first padded row
    }
    if (on) {
      out[(int64_t)pillar * G.f + lane] = best;
      if (argmax) argmax[(int64_t)pillar * G.f + lane] = arg;
    }
  }
}

// x of row `arg` of a pillar for this lane's channel (0 for padded rows)
template <int C>
__device__ __forceinline__ float pfn_row_x(const float *p, int n, int arg, const float *mean3,
                                           float cx, float cy, const float *w, float *feat) {
  constexpr int K = C + 5;
  if (arg >= n) {
#pragma unroll
    for (int j = 0; j < K; ++j) feat[j] = 0.f;
    return 0.f;
  }
  pfn_decorate<C>(p + arg * C, mean3, cx, cy, feat);
  float x = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) x = fmaf(w[j], feat[j], x);
  return x;
}

template <int C>
__global__ void __launch_bounds__(256)
    pfn_bwd_reduce_kernel(PfnGeom G, const float *__restrict__ vox,
                          const int32_t *__restrict__ num_points, const int4 *__restrict__ coors,
                          const float *__restrict__ weight, const float *__restrict__ mean_,
                          const float *__restrict__ invstd_, const float *__restrict__ out,
                          const float *__restrict__ dout, const int32_t *__restrict__ argmax,
                          float *__restrict__ slabs) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool on = lane < G.f;
  constexpr int K = C + 5;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = on ? weight[lane * K + j] : 0.f;
  const float mu = on ? mean_[lane] : 0.f, is = on ? invstd_[lane] : 0.f;
  float sg = 0.f, sb = 0.f;
  for (int pillar = wave; pillar < G.m; pillar += nwaves) {
    int n;
    float mean3[3], cx, cy;
    pfn_pillar_setup<C>(G, vox, num_points, coors, pillar, n, mean3, cx, cy);
    if (!on) continue;
    const int64_t o = (int64_t)pillar * G.f + lane;
    float dz = out[o] > 0.f ? dout[o] : 0.f;  // ReLU gate at the max row
    float feat[K];
    float x = pfn_row_x<C>(vox + (int64_t)pillar * G.t * C, n, argmax[o], mean3, cx, cy, w, feat);
    sb += dz;
    sg = fmaf(dz, (x - mu) * is, sg);
  }
  if (on) {  // slab of this wave: [dgamma f | dbeta f]
    slabs[(int64_t)wave * 2 * G.f + lane] = sg;
    slabs[(int64_t)wave * 2 * G.f + G.f + lane] = sb;
  }
}

// dW[c][j] = sum over real rows of dx_row[c] * f_row[j],
// dx_row = gamma*invstd*(dz_row - dbeta/N - xhat_row*dgamma/N), dz_row != 0 only on the argmax row
template <int C>
__global__ void __launch_bounds__(256)
    pfn_bwd_weight_kernel(PfnGeom G, const float *__restrict__ vox,
                          const int32_t *__restrict__ num_points, const int4 *__restrict__ coors,
                          const float *__restrict__ weight, const float *__restrict__ gamma,
                          const float *__restrict__ mean_, const float *__restrict__ invstd_,
                          const float *__restrict__ out, const float *__restrict__ dout,
                          const int32_t *__restrict__ argmax, const float *__restrict__ dgamma,
                          const float *__restrict__ dbeta, float *__restrict__ slabs) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool on = lane < G.f;
  constexpr int K = C + 5;
  float w[K], acc[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = on ? weight[lane * K + j] : 0.f;
    acc[j] = 0.f;
  }
  const float ntot = (float)G.m * (float)G.t;
  const float mu = on ? mean_[lane] : 0.f, is = on ? invstd_[lane] : 0.f;
  const float g = (on && gamma) ? gamma[lane] : 1.f;
  const float c1 = on ? dbeta[lane] / ntot : 0.f, c2 = on ? dgamma[lane] / ntot : 0.f;
  for (int pillar = wave; pillar < G.m; pillar += nwaves) {
    int n;
    float mean3[3], cx, cy;
    pfn_pillar_setup<C>(G, vox, num_points, coors, pillar, n, mean3, cx, cy);
    const float *p = vox + (int64_t)pillar * G.t * C;
    const int64_t o = (int64_t)pillar * G.f + lane;
    const int arg = on ? argmax[o] : -1;
    const float dzmax = (on && out[o] > 0.f) ? dout[o] : 0.f;
    for (int i = 0; i < n; ++i) {
      float feat[K];
      pfn_decorate<C>(p + i * C, mean3, cx, cy, feat);
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) x = fmaf(w[j], feat[j], x);
      float dz = i == arg ? dzmax : 0.f;
      float dx = g * is * (dz - c1 - (x - mu) * is * c2);
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] = fmaf(dx, feat[j], acc[j]);
    }
  }
  if (on)  // slab of this wave: dW in nn.Linear layout [f][K]
#pragma unroll
    for (int j = 0; j < K; ++j) slabs[(int64_t)wave * G.f * K + lane * K + j] = acc[j];
}

static int pfn_geom(PfnGeom &G, int m, int t, int c, int f, float vx, float vy, float xo, float yo,
                    const char *who) {
  D3D_REQUIRE(m >= 1 && t >= 1, "%s: m=%d t=%d", who, m, t);
  D3D_REQUIRE(c >= 3 && c <= 5, "%s: point channels c=%d unsupported (3..5)", who, c);
  D3D_REQUIRE(f >= 1 && f <= 64, "%s: num_filters=%d unsupported (<= 64)", who, f);
  G.m = m;
  G.t = t;
  G.c = c;
  G.k = c + 5;
  G.f = f;
  G.vx = vx;
  G.vy = vy;
  G.xo = xo;
  G.yo = yo;
  return D3D_OK;
}

// point channels: 4 (KITTI x,y,z,r), 5 (nuScenes + dt), 3 (xyz only)
#define PFN_DISPATCH(kern, ...)                                                              \
  do {                                                                                       \
    if (c == 4)                                                                              \
      hipLaunchKernelGGL(kern<4>, dim3(nb), dim3(256), 0, stream, __VA_ARGS__);              \
    else if (c == 5)                                                                         \
      hipLaunchKernelGGL(kern<5>, dim3(nb), dim3(256), 0, stream, __VA_ARGS__);              \
    else                                                                                     \
      hipLaunchKernelGGL(kern<3>, dim3(nb), dim3(256), 0, stream, __VA_ARGS__);              \
  } while (0)

static int pfn_blocks(int m) {
  // waves of a launch (D3D_PFN_WAVES), each leaves one slab.  A wave walks its pillars one after the other with two
  // dependent passes over a pillar's points, i.e. the kernels are latency-bound and want every wave slot of the chip: 32
  // per CU (nuScenes PointPillars step, 120k pillars: 2048 waves 505 frames/s, 4096 519, 8192 527, 16384 521)
  static const int cap = getenv("D3D_PFN_WAVES") ? atoi(getenv("D3D_PFN_WAVES")) : 8192;
  const int lim = cap < 64 ? 64 : cap;
  int waves = m < lim ? m : lim;
  int nb = (waves + 3) / 4;
  return nb < 1 ? 1 : nb;
}

static size_t pfn_slab(int c, int f) { return (size_t)f * (size_t)(c + 5 > 2 ? c + 5 : 2); }

// floats of the scratch either direction needs: [2f] folded statistics | one slab of max(2f, f (c + 5)) per wave
extern "C" size_t d3d_pfn_ws_floats(int m, int c, int f) {
  if (m < 1 || c < 3 || c > 5 || f < 1 || f > 64) return 0;
  const size_t slab = (size_t)f * (size_t)(c + 5 > 2 ? c + 5 : 2);
  return (size_t)2 * f + ((size_t)pfn_blocks(m) * 4 + 16 /* PFN_FOLD_SEGS rows of the two-level fold */) * slab;
}

// out = sum of the slabs in ascending order.  More than 1024 slabs: two levels -- PFN_FOLD_SEGS segments in parallel into
// `tmp` ([PFN_FOLD_SEGS][nval], behind the slabs in the caller's workspace), then those rows (a fixed order again;
// one block of 16 waves walking 8192 slabs took 35 us, i.e. as long as the kernels it serves)
#define PFN_FOLD_SEGS 16
static int pfn_fold(const float *slabs, int nslabs, int nval, float *out, int split, float *out2, float *tmp,
                    hipStream_t stream) {
  const dim3 blk(64 * PFN_FOLD_WAVES);
  if (nslabs > 1024 && tmp) {
    const int seg = (nslabs + PFN_FOLD_SEGS - 1) / PFN_FOLD_SEGS;
    const int nseg = (nslabs + seg - 1) / seg;
    hipLaunchKernelGGL(pfn_fold_kernel, dim3((nval + 63) / 64, nseg), blk, 0, stream, slabs, nslabs, nval, tmp, nval,
                       nullptr, seg);
    D3D_LAUNCH_TRY("pfn_fold_kernel");
    hipLaunchKernelGGL(pfn_fold_kernel, dim3((nval + 63) / 64, 1), blk, 0, stream, (const float *)tmp, nseg, nval, out,
                       split, out2, nseg);
  } else {
    hipLaunchKernelGGL(pfn_fold_kernel, dim3((nval + 63) / 64, 1), blk, 0, stream, slabs, nslabs, nval, out, split, out2,
                       nslabs);
  }
  D3D_LAUNCH_TRY("pfn_fold_kernel");
  return D3D_OK;
}

extern "C" int d3d_pfn_fwd(const float *voxels, const int32_t *num_points, const int32_t *coors, int m,
                           int t, int c, float vx, float vy, float x_offset, float y_offset,
                           const float *weight, int f, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, float momentum, float eps,
                           int training, float *out, int32_t *argmax, float *save_mean,
                           float *save_invstd, float *stats_ws, size_t stats_ws_floats, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PfnGeom G;
  int rc = pfn_geom(G, m, t, c, f, vx, vy, x_offset, y_offset, "d3d_pfn_fwd");
  if (rc) return rc;
  D3D_REQUIRE(voxels && num_points && coors && weight && out, "d3d_pfn_fwd: null pointer");
  D3D_REQUIRE(training ? stats_ws != nullptr : (running_mean && running_var),
              "d3d_pfn_fwd: training needs stats_ws, eval needs running stats");
  const int nb = pfn_blocks(m);
  if (training) {
    if (stats_ws_floats < d3d_pfn_ws_floats(m, c, f))
      return d3d_set_error(D3D_ECAPACITY, "d3d_pfn_fwd: stats_ws of %zu floats < %zu (d3d_pfn_ws_floats)",
                           stats_ws_floats, d3d_pfn_ws_floats(m, c, f));
    float *slabs = stats_ws + 2 * f;
    PFN_DISPATCH(pfn_stats_kernel, G, voxels, num_points, (const int4 *)coors, weight, slabs);
    D3D_LAUNCH_TRY("pfn_stats_kernel");
    if (int frc = pfn_fold(slabs, nb * 4, 2 * f, stats_ws, 2 * f, nullptr, slabs + (size_t)nb * 4 * pfn_slab(c, f), stream)) return frc;
  }
  PFN_DISPATCH(pfn_apply_kernel, G, voxels, num_points, (const int4 *)coors, weight, gamma, beta,
               running_mean, running_var, momentum, eps, training, stats_ws, out, argmax, save_mean,
               save_invstd);
  D3D_LAUNCH_TRY("pfn_apply_kernel");
  return D3D_OK;
}

extern "C" int d3d_pfn_bwd(const float *voxels, const int32_t *num_points, const int32_t *coors, int m,
                           int t, int c, float vx, float vy, float x_offset, float y_offset,
                           const float *weight, int f, const float *gamma, const float *save_mean,
                           const float *save_invstd, const float *out, const float *dout,
                           const int32_t *argmax, float *dweight, float *dgamma, float *dbeta,
                           float *workspace, size_t workspace_floats, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PfnGeom G;
  int rc = pfn_geom(G, m, t, c, f, vx, vy, x_offset, y_offset, "d3d_pfn_bwd");
  if (rc) return rc;
  D3D_REQUIRE(voxels && num_points && coors && weight && save_mean && save_invstd && out && dout &&
                  argmax && dweight && dgamma && dbeta && workspace,
              "d3d_pfn_bwd: null pointer");
  if (workspace_floats < d3d_pfn_ws_floats(m, c, f))
    return d3d_set_error(D3D_ECAPACITY, "d3d_pfn_bwd: workspace of %zu floats < %zu (d3d_pfn_ws_floats)",
                         workspace_floats, d3d_pfn_ws_floats(m, c, f));
  const int nb = pfn_blocks(m);
  float *slabs = workspace + 2 * f;  // (the first 2 f floats mirror the forward's layout and stay unused here)
  PFN_DISPATCH(pfn_bwd_reduce_kernel, G, voxels, num_points, (const int4 *)coors, weight, save_mean,
               save_invstd, out, dout, argmax, slabs);
  D3D_LAUNCH_TRY("pfn_bwd_reduce_kernel");
  float *tmp = slabs + (size_t)nb * 4 * pfn_slab(c, f);
  if (int frc = pfn_fold(slabs, nb * 4, 2 * f, dgamma, f, dbeta, tmp, stream)) return frc;
  PFN_DISPATCH(pfn_bwd_weight_kernel, G, voxels, num_points, (const int4 *)coors, weight, gamma,
               save_mean, save_invstd, out, dout, argmax, dgamma, dbeta, slabs);
  D3D_LAUNCH_TRY("pfn_bwd_weight_kernel");
  return pfn_fold(slabs, nb * 4, f * G.k, dweight, f * G.k, nullptr, tmp, stream);
}
