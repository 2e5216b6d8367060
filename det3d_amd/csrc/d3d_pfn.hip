// PillarFeatureNet with a single (last) PFNLayer, fused for gfx950.
//
// Reference: det3d/models/readers/pillar_encoder.py:115-155 (decorate: cluster offsets, pillar-centre
// offsets, padding mask) and :42-57 (Linear(K->F, no bias) -> BatchNorm1d over all M*T rows -> ReLU
// -> max over T); ~15 PyTorch kernels over the padded [M, T, K] tensor (43 MB per pass at KITTI).
// Here nothing of size [M, T, *] is materialised: one wavefront per pillar, lane = output channel,
// the lane keeps its weight row W[c][0..K) in registers and walks only the pillar's REAL points
// (padded rows are exact zeros after the mask: they add nothing to the BN sums but count in N = M*T,
// and they compete in the max with relu(shift)).
//   fwd  : pfn_stats (sum, sumsq per channel)  ->  pfn_apply (normalise, ReLU, max/argmax over T)
//   bwd  : pfn_bwd_reduce (dbeta, dgamma from the argmax rows)  ->  pfn_bwd_weight (dW)
// All f32 (the reader runs before any bf16 cast); HBM-bound on reading the voxel rows once per pass.
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
                     float *__restrict__ stats) {
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
  if (lane < G.f) {
    atomicAdd(&stats[lane], s);
    atomicAdd(&stats[G.f + lane], q);
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
                          float *__restrict__ dgamma, float *__restrict__ dbeta) {
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
  if (on) {
    atomicAdd(&dgamma[lane], sg);
    atomicAdd(&dbeta[lane], sb);
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
                          const float *__restrict__ dbeta, float *__restrict__ dweight) {
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
  if (on)
#pragma unroll
    for (int j = 0; j < K; ++j) atomicAdd(&dweight[lane * K + j], acc[j]);
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
  int waves = m < 2048 ? m : 2048;  // <= 2048 waves: few same-address atomics, 8 waves per CU
  int nb = (waves + 3) / 4;
  return nb < 1 ? 1 : nb;
}

extern "C" int d3d_pfn_fwd(const float *voxels, const int32_t *num_points, const int32_t *coors, int m,
                           int t, int c, float vx, float vy, float x_offset, float y_offset,
                           const float *weight, int f, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, float momentum, float eps,
                           int training, float *out, int32_t *argmax, float *save_mean,
                           float *save_invstd, float *stats_ws, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PfnGeom G;
  int rc = pfn_geom(G, m, t, c, f, vx, vy, x_offset, y_offset, "d3d_pfn_fwd");
  if (rc) return rc;
  D3D_REQUIRE(voxels && num_points && coors && weight && out, "d3d_pfn_fwd: null pointer");
  D3D_REQUIRE(training ? stats_ws != nullptr : (running_mean && running_var),
              "d3d_pfn_fwd: training needs stats_ws, eval needs running stats");
  const int nb = pfn_blocks(m);
  if (training) {
    D3D_HIP_TRY(d3d_fill_async(stream, stats_ws, 0, (size_t)2 * f * sizeof(float)));
    PFN_DISPATCH(pfn_stats_kernel, G, voxels, num_points, (const int4 *)coors, weight, stats_ws);
    D3D_LAUNCH_TRY("pfn_stats_kernel");
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
                           d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PfnGeom G;
  int rc = pfn_geom(G, m, t, c, f, vx, vy, x_offset, y_offset, "d3d_pfn_bwd");
  if (rc) return rc;
  D3D_REQUIRE(voxels && num_points && coors && weight && save_mean && save_invstd && out && dout &&
                  argmax && dweight && dgamma && dbeta,
              "d3d_pfn_bwd: null pointer");
  const int nb = pfn_blocks(m);
  {
    const D3dFillSeg segs[3] = {{dweight, 0, (size_t)f * G.k * sizeof(float)},
                                {dgamma, 0, (size_t)f * sizeof(float)},
                                {dbeta, 0, (size_t)f * sizeof(float)}};
    D3D_HIP_TRY(d3d_fill_async(stream, segs, 3));
  }
  PFN_DISPATCH(pfn_bwd_reduce_kernel, G, voxels, num_points, (const int4 *)coors, weight, save_mean,
               save_invstd, out, dout, argmax, dgamma, dbeta);
  D3D_LAUNCH_TRY("pfn_bwd_reduce_kernel");
  PFN_DISPATCH(pfn_bwd_weight_kernel, G, voxels, num_points, (const int4 *)coors, weight, gamma,
               save_mean, save_invstd, out, dout, argmax, dgamma, dbeta, dweight);
  D3D_LAUNCH_TRY("pfn_bwd_weight_kernel");
  return D3D_OK;
}
