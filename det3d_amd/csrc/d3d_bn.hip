// BatchNorm1d (+ReLU) over the active sites of a sparse tensor, gfx950.
//
// The reference applies nn.BatchNorm1d and nn.ReLU as separate modules to `.features`
// ([N, C], N ~ 10^5) after every sparse conv (det3d/models/backbones/scn.py:106-157,76-87).
// In PyTorch-ROCm that is 3 + 3 kernels per layer at ~60 us for the two reductions
// (profiles/round1_kernel_stats_v1.txt: 2.3 ms of a 13.3 ms step).  Here:
//   fwd  = stats pass (per-channel sum / sum of squares, f32) + apply pass that derives
//          mean / invstd / scale / shift in its prologue, updates the running statistics
//          (block 0) and writes relu(x * scale + shift)
//   bwd  = reduce pass (dbeta, dgamma; the ReLU mask is recomputed from x) + dx pass
// Every pass is one coalesced 16 B/lane sweep over [N, C]: pure HBM streaming.
// The reductions are DETERMINISTIC (round 3): lanes fold with shuffles, waves through per-wave LDS rows summed in
// wave order, blocks through per-block slabs folded in block order by the last block to arrive (d3d_fold.h) --
// no float atomics anywhere, two identical launches are bit-equal.
#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_fold.h"
#include <stdlib.h>

__device__ __forceinline__ float bn_bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t bn_f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

template <bool BF16> struct BnVec {
  static constexpr int EPV = BF16 ? 8 : 4;  // elements per 16-byte vector
  __device__ static void load(const void *base, int64_t vec_index, float *v) {
    uint4 r = ((const uint4 *)base)[vec_index];
    if (BF16) {
      v[0] = bn_bf16_to_f32(r.x & 0xffffu); v[1] = bn_bf16_to_f32(r.x >> 16);
      v[2] = bn_bf16_to_f32(r.y & 0xffffu); v[3] = bn_bf16_to_f32(r.y >> 16);
      v[4] = bn_bf16_to_f32(r.z & 0xffffu); v[5] = bn_bf16_to_f32(r.z >> 16);
      v[6] = bn_bf16_to_f32(r.w & 0xffffu); v[7] = bn_bf16_to_f32(r.w >> 16);
    } else {
      v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
      v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    }
  }
  __device__ static void store(void *base, int64_t vec_index, const float *v) {
    uint4 r;
    if (BF16) {
      r.x = bn_f32_to_bf16(v[0]) | (bn_f32_to_bf16(v[1]) << 16);
      r.y = bn_f32_to_bf16(v[2]) | (bn_f32_to_bf16(v[3]) << 16);
      r.z = bn_f32_to_bf16(v[4]) | (bn_f32_to_bf16(v[5]) << 16);
      r.w = bn_f32_to_bf16(v[6]) | (bn_f32_to_bf16(v[7]) << 16);
    } else {
      r.x = __float_as_uint(v[0]); r.y = __float_as_uint(v[1]);
      r.z = __float_as_uint(v[2]); r.w = __float_as_uint(v[3]);
    }
    ((uint4 *)base)[vec_index] = r;
  }
};

#define BN_MAXC 256

// store of the streaming sweeps.  f32 data may instead be written as the two bf16 planes of the dense conv kernels'
// split format (d3d_conv2d.hip): hi = bf16(v) -> `y`, lo = bf16(v - hi) -> `y_lo`, 8 bytes per plane and lane
template <bool BF16>
__device__ __forceinline__ void bn_store(void *y, void *y_lo, int64_t vec_index, const float *v) {
  if (BF16 || y_lo == nullptr) {
    BnVec<BF16>::store(y, vec_index, v);
  } else {
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = bn_f32_to_bf16(v[j]);
      l[j] = bn_f32_to_bf16(v[j] - bn_bf16_to_f32(h[j]));
    }
    ((uint2 *)y)[vec_index] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    ((uint2 *)y_lo)[vec_index] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
  }
}

// thread -> (row lane, 16-byte segment of the row); segs = c / EPV divides 256
// NT = threads per block, U = rows in flight per lane.  The two REDUCTION sweeps end with 2c global atomics per
// block, and same-address float atomics retire at ~12-20 ns each on MI355X: on the RPN's [140800, 128] maps 2048
// blocks x 256 atomics cost more than the sweep itself, so large inputs run fewer, larger blocks (the variant table
// below; tools/bench_bn.py measures it).  Lanes of a wave that hold the same channels are folded with shuffles
// before the LDS atomics.
struct BnRaw { uint4 r; };
template <bool BF16> __device__ __forceinline__ void bn_unpack(const uint4 &r, float *v) {
  if (BF16) {
    v[0] = bn_bf16_to_f32(r.x & 0xffffu); v[1] = bn_bf16_to_f32(r.x >> 16);
    v[2] = bn_bf16_to_f32(r.y & 0xffffu); v[3] = bn_bf16_to_f32(r.y >> 16);
    v[4] = bn_bf16_to_f32(r.z & 0xffffu); v[5] = bn_bf16_to_f32(r.z >> 16);
    v[6] = bn_bf16_to_f32(r.w & 0xffffu); v[7] = bn_bf16_to_f32(r.w >> 16);
  } else {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
    v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
}

// fold the lanes of a wave that work on the same 16-byte segment (lane % segs) with shuffles, the waves of the block
// through one LDS row per wave summed in wave order, the blocks through d3d_fold_block: a fixed summation tree
template <int EPV, int NT>
__device__ __forceinline__ void bn_block_reduce(float *a, float *b, int c, int segs, int seg,
                                                float (*rows)[2 * BN_MAXC], float *tot, int *flag,
                                                float *__restrict__ ws, unsigned *__restrict__ counters, int g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (segs < 64) {
    for (int m = 32; m >= segs; m >>= 1) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        a[j] += __shfl_xor(a[j], m, 64);
        b[j] += __shfl_xor(b[j], m, 64);
      }
    }
  }
  if (segs >= 64 || lane < segs) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      rows[wave][seg * EPV + j] = a[j];
      rows[wave][c + seg * EPV + j] = b[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * c; i += NT) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += rows[w][i];
    tot[i] = t;
  }
  __syncthreads();
  d3d_fold_block<NT>(tot, 2 * c, ws, counters, g, flag);
}

template <bool BF16, int NT, int U>
__global__ void __launch_bounds__(NT)
    bn_stats_kernel(const void *__restrict__ x, int n, int c, float *__restrict__ stats,
                    unsigned *__restrict__ counters, int fold_g) {
  constexpr int EPV = BnVec<BF16>::EPV;
  __shared__ float rows[NT / 64][2 * BN_MAXC];
  __shared__ float tot[2 * BN_MAXC];
  __shared__ int flag;
  const int segs = c / EPV;
  const int seg = threadIdx.x % segs, rl = threadIdx.x / segs, rp = NT / segs;
  float s[EPV], q[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) s[j] = q[j] = 0.f;
  const int64_t step = (int64_t)gridDim.x * rp;
  int64_t r = (int64_t)blockIdx.x * rp + rl;
  const uint4 *xv = (const uint4 *)x;
  for (; r + (U - 1) * step < n; r += U * step) {  // U independent 16-byte loads in flight per lane
    uint4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) raw[u] = xv[(r + u * step) * segs + seg];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[EPV];
      bn_unpack<BF16>(raw[u], v);
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        s[j] += v[j];
        q[j] = fmaf(v[j], v[j], q[j]);
      }
    }
  }
  for (; r < n; r += step) {
    float v[EPV];
    bn_unpack<BF16>(xv[r * segs + seg], v);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      s[j] += v[j];
      q[j] = fmaf(v[j], v[j], q[j]);
    }
  }
  bn_block_reduce<EPV, NT>(s, q, c, segs, seg, rows, tot, &flag, stats, counters, fold_g);
}

template <bool BF16>
__global__ void __launch_bounds__(256)
    bn_apply_kernel(const void *__restrict__ x, int n, int c, const float *__restrict__ gamma,
                    const float *__restrict__ beta, float *running_mean, float *running_var,
                    float momentum, float eps, int training, int relu,
                    const float *__restrict__ stats, int copies, void *__restrict__ y, float *save_mean,
                    float *save_invstd, void *__restrict__ y_lo, const void *__restrict__ res) {
  // res (optional): a second [n, c] matrix added BEFORE the ReLU -- y = relu(bn(x) + res), the tail of a residual
  // block (SparseBasicBlock, det3d/models/backbones/scn.py:84-87: out += identity; relu)
  constexpr int EPV = BnVec<BF16>::EPV;
  __shared__ float scale[BN_MAXC], shift[BN_MAXC];
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float mean, var;
    if (training == 2) {
      // cross-rank statistics (DistributedSyncBN, det3d/ops/syncbn/syncbn.py:39-52): the caller has all-reduced the
      // per-rank moments and hands in the finished mean | variance; the running statistics are the caller's too
      mean = stats[ch];
      var = stats[c + ch];
    } else if (training) {
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < copies; ++k) {
        s1 += stats[(size_t)k * 2 * c + ch];
        s2 += stats[(size_t)k * 2 * c + c + ch];
      }
      mean = s1 / (float)n;
      var = fmaxf(s2 / (float)n - mean * mean, 0.f);  // biased, as F.batch_norm
    } else {
      mean = running_mean[ch];
      var = running_var[ch];
    }
    float invstd = rsqrtf(var + eps);
    float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
    scale[ch] = g * invstd;
    shift[ch] = b - mean * g * invstd;
    if (blockIdx.x == 0) {
      if (save_mean) save_mean[ch] = mean;
      if (save_invstd) save_invstd[ch] = invstd;
      if (training == 1 && running_mean) {
        float unbiased = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
      }
    }
  }
  __syncthreads();
  const int segs = c / EPV;
  const int seg = threadIdx.x % segs, rl = threadIdx.x / segs, rp = 256 / segs;
  float sc[EPV], sh[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    sc[j] = scale[seg * EPV + j];
    sh[j] = shift[seg * EPV + j];
  }
  const int64_t step = (int64_t)gridDim.x * rp;
  int64_t r = (int64_t)blockIdx.x * rp + rl;
  for (; r + 3 * step < n; r += 4 * step) {
    float v[4][EPV];
#pragma unroll
    for (int u = 0; u < 4; ++u) BnVec<BF16>::load(x, (r + u * step) * segs + seg, v[u]);
    if (res) {
      float a[4][EPV];
#pragma unroll
      for (int u = 0; u < 4; ++u) BnVec<BF16>::load(res, (r + u * step) * segs + seg, a[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < EPV; ++j) v[u][j] = fmaf(v[u][j], sc[j], sh[j]) + a[u][j];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < EPV; ++j) v[u][j] = fmaf(v[u][j], sc[j], sh[j]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) v[u][j] = relu ? fmaxf(v[u][j], 0.f) : v[u][j];
      bn_store<BF16>(y, y_lo, (r + u * step) * segs + seg, v[u]);
    }
  }
  for (; r < n; r += step) {
    float v[EPV], a[EPV];
    BnVec<BF16>::load(x, r * segs + seg, v);
    if (res) BnVec<BF16>::load(res, r * segs + seg, a);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float z = fmaf(v[j], sc[j], sh[j]);
      if (res) z += a[j];
      v[j] = relu ? fmaxf(z, 0.f) : z;
    }
    bn_store<BF16>(y, y_lo, r * segs + seg, v);
  }
}

// dz = dy * [z > 0] ; dbeta = sum dz ; dgamma = sum dz * xhat
template <bool BF16, int NT, int U>
__global__ void __launch_bounds__(NT)
    bn_bwd_reduce_kernel(const void *__restrict__ x, const void *__restrict__ dy, int n, int c,
                         const float *__restrict__ gamma, const float *__restrict__ beta,
                         const float *__restrict__ mean, const float *__restrict__ invstd, int relu,
                         float *__restrict__ partial, unsigned *__restrict__ counters, int fold_g,
                         const void *__restrict__ res) {
  // res (optional): the matrix that was added in front of the ReLU (bn_apply_kernel): the mask is [z + res > 0]
  constexpr int EPV = BnVec<BF16>::EPV;
  constexpr int UR = U / 2;  // rows in flight: each row is two loads (x, dy)
  __shared__ float rows[NT / 64][2 * BN_MAXC];
  __shared__ float tot[2 * BN_MAXC];
  __shared__ int flag;
  const int segs = c / EPV;
  const int seg = threadIdx.x % segs, rl = threadIdx.x / segs, rp = NT / segs;
  // z = x * zs + zb (the ReLU mask), xhat = x * is - mu * is
  float is[EPV], mi[EPV], zs[EPV], zb[EPV], sg[EPV], sb[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    int ch = seg * EPV + j;
    const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
    is[j] = invstd[ch];
    mi[j] = -mean[ch] * is[j];
    zs[j] = g * is[j];
    zb[j] = fmaf(mi[j], g, b);
    sg[j] = sb[j] = 0.f;
  }
  const int64_t step = (int64_t)gridDim.x * rp;
  int64_t r = (int64_t)blockIdx.x * rp + rl;
  const uint4 *xv = (const uint4 *)x, *dv = (const uint4 *)dy, *av = (const uint4 *)res;
  for (; r + (UR - 1) * step < n; r += UR * step) {
    uint4 rx[UR], rd[UR], ra[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      rx[u] = xv[(r + u * step) * segs + seg];
      rd[u] = dv[(r + u * step) * segs + seg];
      if (res) ra[u] = av[(r + u * step) * segs + seg];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      float v[EPV], d[EPV], a[EPV];
      bn_unpack<BF16>(rx[u], v);
      bn_unpack<BF16>(rd[u], d);
      if (res) bn_unpack<BF16>(ra[u], a);
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        float z = fmaf(v[j], zs[j], zb[j]);
        if (res) z += a[j];
        float dz = (relu && z <= 0.f) ? 0.f : d[j];
        sb[j] += dz;
        sg[j] = fmaf(dz, fmaf(v[j], is[j], mi[j]), sg[j]);
      }
    }
  }
  for (; r < n; r += step) {
    float v[EPV], d[EPV], a[EPV];
    bn_unpack<BF16>(xv[r * segs + seg], v);
    bn_unpack<BF16>(dv[r * segs + seg], d);
    if (res) bn_unpack<BF16>(av[r * segs + seg], a);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float z = fmaf(v[j], zs[j], zb[j]);
      if (res) z += a[j];
      float dz = (relu && z <= 0.f) ? 0.f : d[j];
      sb[j] += dz;
      sg[j] = fmaf(dz, fmaf(v[j], is[j], mi[j]), sg[j]);
    }
  }
  bn_block_reduce<EPV, NT>(sg, sb, c, segs, seg, rows, tot, &flag, partial, counters, fold_g);  // [dgamma | dbeta]
}

// dx = gamma * invstd * (dz - dbeta/N - xhat * dgamma/N) = A dz + P x + Q per channel
template <bool BF16, int UR>
__global__ void __launch_bounds__(256)
    bn_bwd_dx_kernel(const void *__restrict__ x, const void *__restrict__ dy, int n, int c,
                     const float *__restrict__ gamma, const float *__restrict__ beta,
                     const float *__restrict__ mean, const float *__restrict__ invstd, int relu,
                     const float *__restrict__ partial, int copies, float *__restrict__ dgamma,
                     float *__restrict__ dbeta, void *__restrict__ dx, void *__restrict__ dx_lo,
                     const void *__restrict__ res, void *__restrict__ dres) {
  // res / dres (optional, together): the matrix added in front of the ReLU and its gradient dz = dy * [z + res > 0]
  constexpr int EPV = BnVec<BF16>::EPV;
  __shared__ float tot[2 * BN_MAXC];
  for (int i = threadIdx.x; i < 2 * c; i += 256) {  // fold the replicated partial sums
    float t = 0.f;
    for (int k = 0; k < copies; ++k) t += partial[(size_t)k * 2 * c + i];
    tot[i] = t;
    if (blockIdx.x == 0) {
      if (i < c) dgamma[i] = t;
      else dbeta[i - c] = t;
    }
  }
  __syncthreads();
  const int segs = c / EPV;
  const int seg = threadIdx.x % segs, rl = threadIdx.x / segs, rp = 256 / segs;
  float zs[EPV], zb[EPV], A[EPV], P[EPV], Q[EPV];
  const float inv_n = 1.f / (float)n;
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    int ch = seg * EPV + j;
    const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
    const float is = invstd[ch], mi = -mean[ch] * is;
    const float c1 = tot[c + ch] * inv_n, c2 = tot[ch] * inv_n;
    zs[j] = g * is;
    zb[j] = fmaf(mi, g, b);
    A[j] = g * is;
    P[j] = -A[j] * c2 * is;
    Q[j] = -A[j] * fmaf(c2, mi, c1);
  }
  const int64_t step = (int64_t)gridDim.x * rp;
  int64_t r = (int64_t)blockIdx.x * rp + rl;
  const uint4 *xv = (const uint4 *)x, *dv = (const uint4 *)dy, *av = (const uint4 *)res;
  for (; r + (UR - 1) * step < n; r += UR * step) {
    uint4 rx[UR], rd[UR], ra[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      rx[u] = xv[(r + u * step) * segs + seg];
      rd[u] = dv[(r + u * step) * segs + seg];
      if (res) ra[u] = av[(r + u * step) * segs + seg];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      float v[EPV], d[EPV], a[EPV];
      bn_unpack<BF16>(rx[u], v);
      bn_unpack<BF16>(rd[u], d);
      if (res) bn_unpack<BF16>(ra[u], a);
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        float z = fmaf(v[j], zs[j], zb[j]);
        if (res) z += a[j];
        float dz = (relu && z <= 0.f) ? 0.f : d[j];
        v[j] = fmaf(A[j], dz, fmaf(P[j], v[j], Q[j]));
        d[j] = dz;
      }
      bn_store<BF16>(dx, dx_lo, (r + u * step) * segs + seg, v);
      if (dres) bn_store<BF16>(dres, nullptr, (r + u * step) * segs + seg, d);
    }
  }
  for (; r < n; r += step) {
    float v[EPV], d[EPV], a[EPV];
    bn_unpack<BF16>(xv[r * segs + seg], v);
    bn_unpack<BF16>(dv[r * segs + seg], d);
    if (res) bn_unpack<BF16>(av[r * segs + seg], a);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float z = fmaf(v[j], zs[j], zb[j]);
      if (res) z += a[j];
      float dz = (relu && z <= 0.f) ? 0.f : d[j];
      v[j] = fmaf(A[j], dz, fmaf(P[j], v[j], Q[j]));
      d[j] = dz;
    }
    bn_store<BF16>(dx, dx_lo, r * segs + seg, v);
    if (dres) bn_store<BF16>(dres, nullptr, r * segs + seg, d);
  }
}

static int bn_check(int n, int c, int dtype, const char *who) {
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "%s: dtype %d", who, dtype);
  const int epv = dtype == D3D_BF16 ? 8 : 4;
  D3D_REQUIRE(n >= 1, "%s: n=%d (empty batches are the caller's business)", who, n);
  D3D_REQUIRE(c >= epv && c <= BN_MAXC && c % epv == 0 && 256 % (c / epv) == 0,
              "%s: c=%d unsupported (need c %% %d == 0, c/%d | 256, c <= %d)", who, c, epv, epv,
              BN_MAXC);
  return D3D_OK;
}

// reduction sweeps (stats, bwd_reduce): block size, loads in flight and grid.  Small inputs keep 256-thread blocks
// (latency: many blocks); inputs of >= 2^23 elements pick a row of this table (d3d_set_tuning knob 3, D3D_BN_VARIANT;
// measured by tools/bench_bn.py on the RPN's [140800, 128] and [140800, 256] maps)
struct BnVariant { int nt, u, cap; };
static const BnVariant BN_VARIANTS[] = {
    {1024, 4, 256}, {256, 4, 2048}, {256, 8, 1024}, {1024, 8, 256}, {512, 8, 512},
    {256, 8, 2048}, {512, 4, 1024}, {1024, 8, 512}, {256, 4, 1024}, {512, 8, 256},
};
#define BN_NUM_VARIANTS ((int)(sizeof(BN_VARIANTS) / sizeof(BN_VARIANTS[0])))
// MI355X, [140800, 128] bf16 (profiles/round2_bench_bn.json): variant 9 = 13.5 / 28.8 us (stats / bwd_reduce incl. the
// accumulator fill launch), variant 0 = 16.9 / 33.3, 2048 x 256-thread blocks (variant 1) = 22.3 / 39.5
// grid cap of the apply / bwd_dx / affine sweeps (knob 6, D3D_BN_STREAM_BLOCKS): 512 blocks x 4 rows in flight is
// 30.1 us for bwd_dx on the 36 MB map, 2048 blocks 40.6 us (same file); inputs under ~8 MB never reach the cap
static bool bn_big(int n, int c) { return (int64_t)n * c >= ((int64_t)1 << 23); }
static BnVariant bn_variant_for(int n, int c) {
  if (!bn_big(n, c)) return BnVariant{256, 4, 0};
  int v = g_bn_variant;
  if (v < 0 || v >= BN_NUM_VARIANTS) v = 0;
  return BN_VARIANTS[v];
}
#define BN_REDUCE_MAX_BLOCKS 512  // slabs the last block folds (d3d_fold.h): 32 groups of 16 at most
static int bn_reduce_grid(const BnVariant &v, int n, int c, int dtype, int small_grid) {
  if (v.cap == 0) return small_grid > BN_REDUCE_MAX_BLOCKS ? BN_REDUCE_MAX_BLOCKS : small_grid;
  const int rp = v.nt / (c / (dtype == D3D_BF16 ? 8 : 4));
  int nb = ((n + rp - 1) / rp + v.u - 1) / v.u;
  if (nb > v.cap) nb = v.cap;
  if (nb > BN_REDUCE_MAX_BLOCKS) nb = BN_REDUCE_MAX_BLOCKS;
  return nb < 1 ? 1 : nb;
}
#define BN_LAUNCH_NU(KERNEL, NT, U, ...)                                                                    \
  do {                                                                                                      \
    if (dtype == D3D_BF16) hipLaunchKernelGGL((KERNEL<true, NT, U>), dim3(nbr), dim3(NT), 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<false, NT, U>), dim3(nbr), dim3(NT), 0, stream, __VA_ARGS__);           \
  } while (0)
#define BN_LAUNCH_REDUCE(KERNEL, ...)                                                                       \
  do {                                                                                                      \
    const BnVariant bv = bn_variant_for(n, c);                                                              \
    const int nbr = bn_reduce_grid(bv, n, c, dtype, nb);                                                    \
    const D3dFoldPlan fp = d3d_fold_plan(nbr, 2 * c);                                                       \
    unsigned *counters = d3d_fold_counters(d3d_fold_counter_words(fp));                                     \
    if (!counters) return D3D_EHIP;                                                                         \
    const int fold_g = fp.g;                                                                                \
    if (bv.nt == 1024 && bv.u == 4) BN_LAUNCH_NU(KERNEL, 1024, 4, __VA_ARGS__);                             \
    else if (bv.nt == 1024) BN_LAUNCH_NU(KERNEL, 1024, 8, __VA_ARGS__);                                     \
    else if (bv.nt == 512 && bv.u == 4) BN_LAUNCH_NU(KERNEL, 512, 4, __VA_ARGS__);                          \
    else if (bv.nt == 512) BN_LAUNCH_NU(KERNEL, 512, 8, __VA_ARGS__);                                       \
    else if (bv.u == 4) BN_LAUNCH_NU(KERNEL, 256, 4, __VA_ARGS__);                                          \
    else BN_LAUNCH_NU(KERNEL, 256, 8, __VA_ARGS__);                                                         \
  } while (0)

static int bn_grid(int n, int c, int dtype) {
  const int rp = 256 / (c / (dtype == D3D_BF16 ? 8 : 4));
  int nb = (n + rp - 1) / rp;
  // A streaming pass wants many wavefronts with several loads in flight each (the sweeps are unrolled
  // 4x), but every block ends with one global atomic per channel, and same-address atomics serialise at
  // L2 (1125 blocks on one accumulator cost 23 us on an 18 MB tensor,
  // profiles/round1_kernel_stats_v2.txt).  So: >= 4 row tiles per block, up to D3D_BN_BLOCKS (default
  // 2048) blocks, and D3D_BN_COPIES replicated accumulators that the consumer kernel folds.
  static int cap = 0;
  if (!cap) {
    const char *e = getenv("D3D_BN_BLOCKS");
    cap = e ? atoi(e) : 2048;
    if (cap < 1) cap = 2048;
  }
  nb = (nb + 3) / 4;
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : nb;
}
// grid of the non-reducing sweeps (apply, bwd_dx, channel affine)
static int bn_stream_grid(int n, int c, int dtype) {
  int nb = bn_grid(n, c, dtype);
  if (g_bn_stream_blocks > 0 && nb > g_bn_stream_blocks) nb = g_bn_stream_blocks;
  return nb;
}

// floats of the workspace d3d_bn_stats / d3d_bn_bwd_reduce need: [2c] result | per-block slabs | group slabs
extern "C" size_t d3d_bn_reduce_ws_floats(int n, int c, int dtype) {
  if ((dtype != D3D_F32 && dtype != D3D_BF16) || n < 1 || c < 1) return 0;
  const int epv = dtype == D3D_BF16 ? 8 : 4;
  if (c % epv || c > BN_MAXC || 256 % (c / epv)) return 0;
  const BnVariant bv = bn_variant_for(n, c);
  const int nbr = bn_reduce_grid(bv, n, c, dtype, bn_grid(n, c, dtype));
  return d3d_fold_ws_floats(nbr, 2 * c);
}

// ---- the two halves of each direction as separate entry points: the conv epilogues (d3d_sparse_conv_fwd_ex)
// can produce the reductions, and cross-rank BatchNorm puts an all-reduce between the halves
extern "C" int d3d_bn_stats(const void *x, int n, int c, int dtype, float *stats_ws, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_bn_stats");
  if (rc) return rc;
  D3D_REQUIRE(x && stats_ws, "d3d_bn_stats: null pointer");
  const int nb = bn_grid(n, c, dtype);
  BN_LAUNCH_REDUCE(bn_stats_kernel, x, n, c, stats_ws, counters, fold_g);
  D3D_LAUNCH_TRY("bn_stats_kernel");
  return D3D_OK;
}

static int bn_apply_impl(const void *x, int n, int c, int dtype, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float momentum, float eps, int training,
                         int relu, const float *stats_ws, int copies, void *y, void *y_lo, float *save_mean,
                         float *save_invstd, d3d_stream_t stream_, const void *res = nullptr);
// y = relu?(bn(x) + residual): the tail of a residual block in one sweep
extern "C" int d3d_bn_apply_residual(const void *x, const void *residual, int n, int c, int dtype, const float *gamma,
                                     const float *beta, float *running_mean, float *running_var, float momentum,
                                     float eps, int training, int relu, const float *stats_ws, int copies, void *y,
                                     float *save_mean, float *save_invstd, d3d_stream_t stream_) {
  D3D_REQUIRE(residual, "d3d_bn_apply_residual: null residual");
  return bn_apply_impl(x, n, c, dtype, gamma, beta, running_mean, running_var, momentum, eps, training, relu, stats_ws,
                       copies, y, nullptr, save_mean, save_invstd, stream_, residual);
}
extern "C" int d3d_bn_apply(const void *x, int n, int c, int dtype, const float *gamma, const float *beta,
                            float *running_mean, float *running_var, float momentum, float eps, int training,
                            int relu, const float *stats_ws, int copies, void *y, float *save_mean,
                            float *save_invstd, d3d_stream_t stream_) {
  return bn_apply_impl(x, n, c, dtype, gamma, beta, running_mean, running_var, momentum, eps, training, relu, stats_ws,
                       copies, y, nullptr, save_mean, save_invstd, stream_);
}
// f32 rows in, the split format of the dense conv kernels out (y_hi / y_lo: bf16 [n, c] planes)
extern "C" int d3d_bn_apply_split(const float *x, int n, int c, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, float momentum, float eps, int training,
                                  int relu, const float *stats_ws, int copies, void *y_hi, void *y_lo,
                                  float *save_mean, float *save_invstd, d3d_stream_t stream_) {
  D3D_REQUIRE(y_lo, "d3d_bn_apply_split: null y_lo");
  return bn_apply_impl(x, n, c, D3D_F32, gamma, beta, running_mean, running_var, momentum, eps, training, relu, stats_ws,
                       copies, y_hi, y_lo, save_mean, save_invstd, stream_);
}
static int bn_apply_impl(const void *x, int n, int c, int dtype, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float momentum, float eps, int training,
                         int relu, const float *stats_ws, int copies, void *y, void *y_lo, float *save_mean,
                         float *save_invstd, d3d_stream_t stream_, const void *res) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_bn_apply");
  if (rc) return rc;
  D3D_REQUIRE(copies >= 1 && copies <= 256, "d3d_bn_apply: copies=%d", copies);
  D3D_REQUIRE(training >= 0 && training <= 2, "d3d_bn_apply: training=%d (0 running statistics, 1 batch sums in stats_ws, "
              "2 finished mean | variance in stats_ws)", training);
  D3D_REQUIRE(x && y, "d3d_bn_apply: null pointer");
  D3D_REQUIRE(training ? stats_ws != nullptr : (running_mean && running_var),
              "d3d_bn_apply: training needs stats_ws, eval needs running stats");
  const int nb = bn_stream_grid(n, c, dtype);
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(nb), dim3(256), 0, stream, x, n, c, gamma, beta,
                       running_mean, running_var, momentum, eps, training, relu, stats_ws, copies, y,
                       save_mean, save_invstd, y_lo, res);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(nb), dim3(256), 0, stream, x, n, c, gamma, beta,
                       running_mean, running_var, momentum, eps, training, relu, stats_ws, copies, y,
                       save_mean, save_invstd, y_lo, res);
  D3D_LAUNCH_TRY("bn_apply_kernel");
  return D3D_OK;
}

static int bn_bwd_reduce_impl(const void *x, const void *res, const void *dy, int n, int c, int dtype,
                              const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                              int relu, float *partial_ws, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_bn_bwd_reduce");
  if (rc) return rc;
  D3D_REQUIRE(x && dy && save_mean && save_invstd && partial_ws, "d3d_bn_bwd_reduce: null pointer");
  const int nb = bn_grid(n, c, dtype);
  BN_LAUNCH_REDUCE(bn_bwd_reduce_kernel, x, dy, n, c, gamma, beta, save_mean, save_invstd, relu, partial_ws,
                   counters, fold_g, res);
  D3D_LAUNCH_TRY("bn_bwd_reduce_kernel");
  return D3D_OK;
}
extern "C" int d3d_bn_bwd_reduce(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                                 const float *beta, const float *save_mean, const float *save_invstd, int relu,
                                 float *partial_ws, d3d_stream_t stream_) {
  return bn_bwd_reduce_impl(x, nullptr, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, stream_);
}
// backward of y = relu?(bn(x) + residual): the ReLU mask is [bn(x) + residual > 0]
extern "C" int d3d_bn_bwd_reduce_residual(const void *x, const void *residual, const void *dy, int n, int c, int dtype,
                                          const float *gamma, const float *beta, const float *save_mean,
                                          const float *save_invstd, int relu, float *partial_ws,
                                          d3d_stream_t stream_) {
  D3D_REQUIRE(residual, "d3d_bn_bwd_reduce_residual: null residual");
  return bn_bwd_reduce_impl(x, residual, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, stream_);
}

static int bn_bwd_dx_impl(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                          const float *beta, const float *save_mean, const float *save_invstd, int relu,
                          const float *partial_ws, int copies, void *dx, void *dx_lo, float *dgamma, float *dbeta,
                          d3d_stream_t stream_, const void *res = nullptr, void *dres = nullptr);
// ... and its data gradients: dx as d3d_bn_bwd_dx, dresidual = dy * [bn(x) + residual > 0]
extern "C" int d3d_bn_bwd_dx_residual(const void *x, const void *residual, const void *dy, int n, int c, int dtype,
                                      const float *gamma, const float *beta, const float *save_mean,
                                      const float *save_invstd, int relu, const float *partial_ws, int copies,
                                      void *dx, void *dresidual, float *dgamma, float *dbeta, d3d_stream_t stream_) {
  D3D_REQUIRE(residual && dresidual, "d3d_bn_bwd_dx_residual: null residual / dresidual");
  return bn_bwd_dx_impl(x, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, copies, dx, nullptr,
                        dgamma, dbeta, stream_, residual, dresidual);
}
extern "C" int d3d_bn_bwd_dx(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                             const float *beta, const float *save_mean, const float *save_invstd, int relu,
                             const float *partial_ws, int copies, void *dx, float *dgamma, float *dbeta,
                             d3d_stream_t stream_) {
  return bn_bwd_dx_impl(x, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, copies, dx, nullptr,
                        dgamma, dbeta, stream_);
}
// f32 x / dy in, dx in the split format (bf16 planes) out
extern "C" int d3d_bn_bwd_dx_split(const float *x, const float *dy, int n, int c, const float *gamma,
                                   const float *beta, const float *save_mean, const float *save_invstd, int relu,
                                   const float *partial_ws, int copies, void *dx_hi, void *dx_lo, float *dgamma,
                                   float *dbeta, d3d_stream_t stream_) {
  D3D_REQUIRE(dx_lo, "d3d_bn_bwd_dx_split: null dx_lo");
  return bn_bwd_dx_impl(x, dy, n, c, D3D_F32, gamma, beta, save_mean, save_invstd, relu, partial_ws, copies, dx_hi,
                        dx_lo, dgamma, dbeta, stream_);
}
static int bn_bwd_dx_impl(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                          const float *beta, const float *save_mean, const float *save_invstd, int relu,
                          const float *partial_ws, int copies, void *dx, void *dx_lo, float *dgamma, float *dbeta,
                          d3d_stream_t stream_, const void *res, void *dres) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_bn_bwd_dx");
  if (rc) return rc;
  D3D_REQUIRE(copies >= 1 && copies <= 256, "d3d_bn_bwd_dx: copies=%d", copies);
  D3D_REQUIRE(x && dy && dx && dgamma && dbeta && save_mean && save_invstd && partial_ws,
              "d3d_bn_bwd_dx: null pointer");
  const int nb = bn_stream_grid(n, c, dtype);
#define BN_DX(BF, UR)                                                                                        \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<BF, UR>), dim3(nb), dim3(256), 0, stream, x, dy, n, c, gamma, beta,   \
                     save_mean, save_invstd, relu, partial_ws, copies, dgamma, dbeta, dx, dx_lo, res, dres)
  const int ur = g_bn_dx_rows >= 4 ? 4 : (g_bn_dx_rows <= 1 ? 1 : 2);
  if (dtype == D3D_BF16) {
    if (ur == 4) BN_DX(true, 4); else if (ur == 2) BN_DX(true, 2); else BN_DX(true, 1);
  } else {
    if (ur == 4) BN_DX(false, 4); else if (ur == 2) BN_DX(false, 2); else BN_DX(false, 1);
  }
#undef BN_DX
  D3D_LAUNCH_TRY("bn_bwd_dx_kernel");
  return D3D_OK;
}

// ---- per-channel affine sweeps for cross-rank BatchNorm (det3d/ops/syncbn): the statistics travel through an
// all-reduce between the reduction and these passes, so the coefficients arrive as [c] vectors
//   fwd: y  = relu?(x * scale + shift)                                  (syncbn_cuda_kernel.cu:166-198)
//   bwd: dx = ca * dz + cb * x + cc,  dz = dy * [x * scale + shift > 0]  (:595-640 with the ReLU mask folded in)
template <bool BF16, bool BWD>
__global__ void __launch_bounds__(256)
    bn_affine_kernel(const void *__restrict__ x, const void *__restrict__ dy, int n, int c,
                     const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                     const float *__restrict__ ca, const float *__restrict__ cb, const float *__restrict__ cc,
                     void *__restrict__ out) {
  constexpr int EPV = BnVec<BF16>::EPV;
  const int segs = c / EPV;
  const int seg = threadIdx.x % segs, rl = threadIdx.x / segs, rp = 256 / segs;
  float sc[EPV], sh[EPV], a[EPV], b[EPV], d[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = seg * EPV + j;
    sc[j] = scale[ch];
    sh[j] = shift[ch];
    if (BWD) {
      a[j] = ca[ch];
      b[j] = cb[ch];
      d[j] = cc[ch];
    }
  }
  const int64_t step = (int64_t)gridDim.x * rp;
  for (int64_t r = (int64_t)blockIdx.x * rp + rl; r < n; r += step) {
    float v[EPV], g[EPV];
    BnVec<BF16>::load(x, r * segs + seg, v);
    if (BWD) BnVec<BF16>::load(dy, r * segs + seg, g);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const float z = fmaf(v[j], sc[j], sh[j]);
      if (BWD) {
        const float dz = (relu && z <= 0.f) ? 0.f : g[j];
        v[j] = fmaf(a[j], dz, fmaf(b[j], v[j], d[j]));
      } else {
        v[j] = relu ? fmaxf(z, 0.f) : z;
      }
    }
    BnVec<BF16>::store(out, r * segs + seg, v);
  }
}

extern "C" int d3d_channel_affine(const void *x, int n, int c, int dtype, const float *scale, const float *shift,
                                  int relu, void *y, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_channel_affine");
  if (rc) return rc;
  D3D_REQUIRE(x && y && scale && shift, "d3d_channel_affine: null pointer");
  const int nb = bn_stream_grid(n, c, dtype);
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL((bn_affine_kernel<true, false>), dim3(nb), dim3(256), 0, stream, x, nullptr, n, c, scale,
                       shift, relu, nullptr, nullptr, nullptr, y);
  else
    hipLaunchKernelGGL((bn_affine_kernel<false, false>), dim3(nb), dim3(256), 0, stream, x, nullptr, n, c, scale,
                       shift, relu, nullptr, nullptr, nullptr, y);
  D3D_LAUNCH_TRY("bn_affine_kernel");
  return D3D_OK;
}

extern "C" int d3d_channel_affine_bwd(const void *x, const void *dy, int n, int c, int dtype, const float *scale,
                                      const float *shift, int relu, const float *coef_dz, const float *coef_x,
                                      const float *coef_1, void *dx, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = bn_check(n, c, dtype, "d3d_channel_affine_bwd");
  if (rc) return rc;
  D3D_REQUIRE(x && dy && dx && scale && shift && coef_dz && coef_x && coef_1, "d3d_channel_affine_bwd: null pointer");
  const int nb = bn_stream_grid(n, c, dtype);
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL((bn_affine_kernel<true, true>), dim3(nb), dim3(256), 0, stream, x, dy, n, c, scale, shift,
                       relu, coef_dz, coef_x, coef_1, dx);
  else
    hipLaunchKernelGGL((bn_affine_kernel<false, true>), dim3(nb), dim3(256), 0, stream, x, dy, n, c, scale, shift,
                       relu, coef_dz, coef_x, coef_1, dx);
  D3D_LAUNCH_TRY("bn_affine_kernel");
  return D3D_OK;
}

extern "C" int d3d_bn_relu_fwd(const void *x, int n, int c, int dtype, const float *gamma,
                               const float *beta, float *running_mean, float *running_var,
                               float momentum, float eps, int training, int relu, void *y,
                               float *save_mean, float *save_invstd, float *stats_ws,
                               d3d_stream_t stream_) {
  // (training = 2, finished moments handed in by the caller, exists for d3d_bn_apply and its forms only: here the
  // statistics sweep below would overwrite them with local sums)
  D3D_REQUIRE(training == 0 || training == 1, "d3d_bn_relu_fwd: training=%d (0 or 1; finished cross-rank moments go to "
              "d3d_bn_apply with training = 2)", training);
  D3D_REQUIRE(training ? stats_ws != nullptr : (running_mean && running_var),
              "d3d_bn_relu_fwd: training needs stats_ws, eval needs running stats");
  if (training) {
    int rc = d3d_bn_stats(x, n, c, dtype, stats_ws, stream_);
    if (rc) return rc;
  }
  return d3d_bn_apply(x, n, c, dtype, gamma, beta, running_mean, running_var, momentum, eps, training, relu,
                      stats_ws, 1, y, save_mean, save_invstd, stream_);
}

extern "C" int d3d_bn_relu_bwd(const void *x, const void *dy, int n, int c, int dtype,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd, int relu, void *dx, float *dgamma,
                               float *dbeta, float *partial_ws, d3d_stream_t stream_) {
  int rc = d3d_bn_bwd_reduce(x, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, stream_);
  if (rc) return rc;
  return d3d_bn_bwd_dx(x, dy, n, c, dtype, gamma, beta, save_mean, save_invstd, relu, partial_ws, 1,
                       dx, dgamma, dbeta, stream_);
}
