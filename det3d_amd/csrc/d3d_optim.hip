// The optimizer step of the training hot path in two launches: global-norm gradient clipping, true weight decay
// and Adam over every parameter tensor of the model.
//
// Reference: det3d/solver/fastai_optim.py (OptimWrapper.step: `p.data.mul_(1 - wd * lr)` for every parameter when
// true_wd, then torch.optim.Adam.step with the one-cycle momentum as beta1), gradient clipping
// det3d/torchie/trainer/hooks/optimizer.py (clip_grad_norm_, max_norm 35 in the shipped configs).  In PyTorch that is
// ~12 multi-tensor launches (norms, norm of norms, clamp, scale, decay, two fused-Adam launches) with a host-side
// tensor-list walk in front of each: at 84 parameter tensors the device sat idle for 0.65 ms of a 7.7 ms step around
// them (profiles/round3_kernel_stats_fp32.txt, gap table).  Here:
//
//   adam_norm_kernel    one block per chunk of CHUNK elements of one tensor: sum of squares of the (scaled) gradient,
//                       added in a fixed order -> partial[chunk]
//   adam_update_kernel  every block adds ALL partials in the same fixed order (a few thousand floats out of L2), so
//                       every block holds the identical total without a cross-block rendezvous; clip coefficient,
//                       decay, moments, update for its chunk; block 0 publishes the norm
//
// Deterministic (no atomics), no host synchronisation, no scratch beyond the partials.  The gradient tensors are new
// allocations every step (autograd), so their addresses travel as kernel arguments; everything else about the
// tensors is static and sits in a device table the caller uploads once.
#include <math.h>

#include "d3d_common.h"

#define ADAM_THREADS 256

struct AdamGrads {
  const float *g[D3D_ADAM_MAX_TENSORS];
};

// fixed-order block sum: lanes by xor butterfly, waves in wave order; every thread returns the total
__device__ __forceinline__ float adam_block_sum(float s, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = s;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < ADAM_THREADS / 64; ++w) t += red[w];
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(ADAM_THREADS)
    adam_norm_kernel(const int64_t *__restrict__ tensors, const int32_t *__restrict__ chunks, int chunk0, int tensor0,
                     int chunk_elems, AdamGrads grads, float grad_scale, float *__restrict__ partial) {
  __shared__ float red[ADAM_THREADS / 64];
  const int c = chunk0 + (int)blockIdx.x;
  const int t = chunks[2 * c], off = chunks[2 * c + 1];
  const int64_t numel = tensors[4 * t + 3];
  const int n = (int)(numel - off < chunk_elems ? numel - off : chunk_elems);
  const float *g = grads.g[t - tensor0] + off;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += ADAM_THREADS) {
    const float v = g[i] * grad_scale;
    s = fmaf(v, v, s);
  }
  s = adam_block_sum(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}

__global__ void __launch_bounds__(ADAM_THREADS)
    adam_update_kernel(const int64_t *__restrict__ tensors, const int32_t *__restrict__ chunks, int chunk0, int tensor0,
                       int chunk_elems, int n_chunks_total, AdamGrads grads, D3dAdamHyper hp, float step_size,
                       float bc2_sqrt, const float *__restrict__ partial, float *__restrict__ grad_norm) {
  __shared__ float red[ADAM_THREADS / 64];
  float s = 0.f;
  for (int j = threadIdx.x; j < n_chunks_total; j += ADAM_THREADS) s += partial[j];
  const float total = sqrtf(adam_block_sum(s, red));
  if (blockIdx.x == 0 && threadIdx.x == 0 && chunk0 == 0) grad_norm[0] = total;
  float coef = hp.grad_scale;
  if (hp.max_norm > 0.f) {
    const float k = hp.max_norm / (total + 1e-6f);  // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max = 1)
    coef *= k < 1.f ? k : 1.f;
  }
  const int c = chunk0 + (int)blockIdx.x;
  const int t = chunks[2 * c], off = chunks[2 * c + 1];
  const int64_t numel = tensors[4 * t + 3];
  const int n = (int)(numel - off < chunk_elems ? numel - off : chunk_elems);
  float *p = (float *)tensors[4 * t + 0] + off;
  float *m = (float *)tensors[4 * t + 1] + off;
  float *v = (float *)tensors[4 * t + 2] + off;
  const float *g = grads.g[t - tensor0] + off;
  const float w1 = 1.f - hp.beta1, w2 = 1.f - hp.beta2;
  for (int i = threadIdx.x; i < n; i += ADAM_THREADS) {
    const float gi = g[i] * coef;
    const float mi = fmaf(gi - m[i], w1, m[i]);            // lerp(m, g, 1 - beta1)
    const float vi = fmaf(w2 * gi, gi, hp.beta2 * v[i]);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + hp.eps;
    p[i] = fmaf(-step_size, mi / denom, p[i] * hp.decay);  // true weight decay first, then the Adam update
  }
}

extern "C" int d3d_adam_step(const int64_t *tensors, int n_tensors, const int32_t *chunks, int n_chunks,
                             int chunk_elems, const int32_t *tensor_chunk_start, const uint64_t *grads_host,
                             const D3dAdamHyper *hyper, float *partial, float *grad_norm, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(tensors && chunks && tensor_chunk_start && grads_host && hyper && partial && grad_norm,
              "d3d_adam_step: null pointer");
  D3D_REQUIRE(n_tensors >= 1 && n_chunks >= n_tensors && chunk_elems >= ADAM_THREADS,
              "d3d_adam_step: n_tensors=%d n_chunks=%d chunk_elems=%d", n_tensors, n_chunks, chunk_elems);
  const D3dAdamHyper hp = *hyper;
  D3D_REQUIRE(hp.step >= 1 && hp.beta1 >= 0.f && hp.beta1 < 1.f && hp.beta2 >= 0.f && hp.beta2 < 1.f && hp.eps > 0.f,
              "d3d_adam_step: step=%d beta1=%g beta2=%g eps=%g", hp.step, hp.beta1, hp.beta2, hp.eps);
  // bias corrections in double on the host, like torch.optim.Adam
  const double bc1 = 1.0 - pow((double)hp.beta1, (double)hp.step);
  const double bc2 = 1.0 - pow((double)hp.beta2, (double)hp.step);
  const float step_size = (float)((double)hp.lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  // gradient addresses ride in the kernel arguments, D3D_ADAM_MAX_TENSORS tensors per launch
  for (int pass = 0; pass < 2; ++pass) {
    for (int t0 = 0; t0 < n_tensors; t0 += D3D_ADAM_MAX_TENSORS) {
      const int t1 = t0 + D3D_ADAM_MAX_TENSORS < n_tensors ? t0 + D3D_ADAM_MAX_TENSORS : n_tensors;
      AdamGrads G;
      for (int t = t0; t < t1; ++t) {
        D3D_REQUIRE(grads_host[t] != 0, "d3d_adam_step: tensor %d has no gradient", t);
        G.g[t - t0] = (const float *)(uintptr_t)grads_host[t];
      }
      const int c0 = tensor_chunk_start[t0], c1 = tensor_chunk_start[t1];
      D3D_REQUIRE(c0 >= 0 && c1 > c0 && c1 <= n_chunks, "d3d_adam_step: chunk range %d..%d of %d", c0, c1, n_chunks);
      if (pass == 0) {
        hipLaunchKernelGGL(adam_norm_kernel, dim3(c1 - c0), dim3(ADAM_THREADS), 0, stream, tensors, chunks, c0, t0,
                           chunk_elems, G, hp.grad_scale, partial);
        D3D_LAUNCH_TRY("adam_norm_kernel");
      } else {
        hipLaunchKernelGGL(adam_update_kernel, dim3(c1 - c0), dim3(ADAM_THREADS), 0, stream, tensors, chunks, c0, t0,
                           chunk_elems, n_chunks, G, hp, step_size, bc2_sqrt, partial, grad_norm);
        D3D_LAUNCH_TRY("adam_update_kernel");
      }
    }
  }
  return D3D_OK;
}

// ------------------------------------------------------------------------------------------------ d3d_pack_all
// The conv kernels read their weights in packed operand forms (sparse: [kvol][cout][cin] / [kvol][cin][cout] with
// mirrored offsets, bf16 or hi / lo planes; dense: [tap][cout][cin] / [mirrored tap][cin][cout]).  An optimizer step
// changes every weight, so every layer repacked in front of its first launch of the next step: 21 small launches inside
// the host-bound forward phase of the KITTI step.  d3d_pack_all rewrites ALL packed forms in one launch behind the
// optimizer: descriptors travel as kernel arguments (<= D3D_PACK_MAX per launch), a block finds its tensor by its first
// block index.  Same element arithmetic as spconv_pack_weights_kernel / conv2d_pack_kernel.
#define PACK_ELEMS_PER_BLOCK 2048
struct PackDescDev {
  const float *w;
  void *fwd, *fwd_lo, *dgrad, *dgrad_lo;
  int kind, a, b, c;      // sparse (0): kvol, cin, cout; dense (1): cout, cin, taps
  int flip, mode_fwd, mode_dgrad;
  int block_begin;
};
struct PackAllArgs {
  PackDescDev d[D3D_PACK_MAX];
  int n;
};

__device__ __forceinline__ uint16_t pk_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// Output-major (round 6, second session): a block writes PACK_ELEMS_PER_BLOCK consecutive elements of ONE packed operand
// (descriptor, layout), a thread eight of them -- 16-byte stores per plane; the eight sources are read at the stride the
// transposition gives them (the weights are 20 MB and L2 / Infinity-Cache resident).  The first version walked the SOURCE
// and scattered four 2-byte stores per element: 36 us per step for 5.3 M weights.  Same element arithmetic, same bytes.
// Needs the fastest dimension of every layout to be a multiple of 8 (cin and cout of every layer the packers serve are
// multiples of 16) -- a tensor that is not takes pack_all_generic_kernel, the source-major first version, below.
__device__ __forceinline__ void pk_store8(void *dst, void *dst_lo, int mode, size_t o, size_t lo_off, const float (&v)[8]) {
  if (mode == 0) {  // plain f32 copy (channel pairs without an MFMA tiling)
    float4 *d = (float4 *)((float *)dst + o);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint16_t h0 = pk_bf16(v[2 * j]), h1 = pk_bf16(v[2 * j + 1]);
    h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    const uint16_t l0 = pk_bf16(v[2 * j] - __uint_as_float((uint32_t)h0 << 16));
    const uint16_t l1 = pk_bf16(v[2 * j + 1] - __uint_as_float((uint32_t)h1 << 16));
    l[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
  }
  *(uint4 *)((uint16_t *)dst + o) = make_uint4(h[0], h[1], h[2], h[3]);
  if (dst_lo) *(uint4 *)((uint16_t *)dst_lo + o + lo_off) = make_uint4(l[0], l[1], l[2], l[3]);
}

__global__ void __launch_bounds__(256) pack_all_kernel(PackAllArgs A) {
  int ti = 0;
  for (int i = 1; i < A.n; ++i) ti = (int)blockIdx.x >= A.d[i].block_begin ? i : ti;  // wave-uniform
  const PackDescDev &D = A.d[ti];
  const int total = D.a * D.b * D.c;
  const int per_layout = (total + PACK_ELEMS_PER_BLOCK - 1) / PACK_ELEMS_PER_BLOCK;
  int bl = (int)blockIdx.x - D.block_begin;
  // blocks [0, per_layout): the forward operand when there is one, the rest: the data-gradient operand
  const bool fwd = D.fwd && bl < per_layout;
  if (D.fwd && !fwd) bl -= per_layout;
  const int o = bl * PACK_ELEMS_PER_BLOCK + (int)threadIdx.x * 8;
  if (o >= total) return;
  float v[8];
  if (D.kind == 0) {
    const int kvol = D.a, cin = D.b, cout = D.c;
    if (fwd) {  // [k][co][ci .. ci + 7]  <-  w[k][ci + j][co]
      const int ci = o % cin, co = (o / cin) % cout, k = o / (cin * cout);
      const float *src = D.w + ((size_t)k * cin + ci) * cout + co;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * cout];
      if (D.mode_fwd == 2)  // planes interleaved per offset: [k][plane][co][ci]
        pk_store8(D.fwd, D.fwd, 2, ((size_t)(2 * k) * cout + co) * cin + ci, (size_t)cout * cin, v);
      else
        pk_store8(D.fwd, nullptr, D.mode_fwd, (size_t)o, 0, v);
    } else {    // [kd][ci][co .. co + 7]  <-  w[k][ci][co + j], kd = flipped k
      const int co = o % cout, ci = (o / cout) % cin, kd = o / (cin * cout);
      const int k = D.flip ? kvol - 1 - kd : kd;
      const float *src = D.w + ((size_t)k * cin + ci) * cout + co;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j];
      if (D.mode_dgrad == 2)
        pk_store8(D.dgrad, D.dgrad, 2, ((size_t)(2 * kd) * cin + ci) * cout + co, (size_t)cin * cout, v);
      else
        pk_store8(D.dgrad, nullptr, D.mode_dgrad, (size_t)o, 0, v);
    }
  } else {
    const int cout = D.a, cin = D.b, taps = D.c;
    if (fwd) {  // [tap][co][ci .. ci + 7]  <-  w[co][ci + j][tap]
      const int ci = o % cin, co = (o / cin) % cout, tap = o / (cin * cout);
      const float *src = D.w + ((size_t)co * cin + ci) * taps + tap;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * taps];
      pk_store8(D.fwd, D.fwd_lo, 1, (size_t)o, 0, v);
    } else {    // [mirrored tap][ci][co .. co + 7]  <-  w[co + j][ci][tap]
      const int co = o % cout, ci = (o / cout) % cin, mt = o / (cin * cout);
      const int tap = taps - 1 - mt;
      const float *src = D.w + ((size_t)co * cin + ci) * taps + tap;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * cin * taps];
      pk_store8(D.dgrad, D.dgrad_lo, 1, (size_t)o, 0, v);
    }
  }
}

// the first version: one SOURCE element per thread and step, any shape
__global__ void __launch_bounds__(256) pack_all_generic_kernel(PackAllArgs A) {
  int ti = 0;
  for (int i = 1; i < A.n; ++i) ti = (int)blockIdx.x >= A.d[i].block_begin ? i : ti;  // wave-uniform
  const PackDescDev &D = A.d[ti];
  const int total = D.a * D.b * D.c;
  const int first = ((int)blockIdx.x - D.block_begin) * PACK_ELEMS_PER_BLOCK;
  for (int t = first + (int)threadIdx.x; t < first + PACK_ELEMS_PER_BLOCK && t < total; t += 256) {
    const float v = D.w[t];
    const uint16_t h = pk_bf16(v);
    const uint16_t l = pk_bf16(v - __uint_as_float((uint32_t)h << 16));
    if (D.kind == 0) {
      const int kvol = D.a, cin = D.b, cout = D.c;
      const int per_k = cin * cout;
      const int k = t / per_k, rem = t - k * per_k;
      const int ci = rem / cout, co = rem - ci * cout;
      const int kd = D.flip ? kvol - 1 - k : k;
      if (D.fwd) {
        if (D.mode_fwd == 0) ((float *)D.fwd)[(k * cout + co) * cin + ci] = v;
        else if (D.mode_fwd == 1) ((uint16_t *)D.fwd)[(k * cout + co) * cin + ci] = h;
        else {
          ((uint16_t *)D.fwd)[((k * 2 + 0) * cout + co) * cin + ci] = h;
          ((uint16_t *)D.fwd)[((k * 2 + 1) * cout + co) * cin + ci] = l;
        }
      }
      if (D.dgrad) {
        if (D.mode_dgrad == 0) ((float *)D.dgrad)[(kd * cin + ci) * cout + co] = v;
        else if (D.mode_dgrad == 1) ((uint16_t *)D.dgrad)[(kd * cin + ci) * cout + co] = h;
        else {
          ((uint16_t *)D.dgrad)[((kd * 2 + 0) * cin + ci) * cout + co] = h;
          ((uint16_t *)D.dgrad)[((kd * 2 + 1) * cin + ci) * cout + co] = l;
        }
      }
    } else {
      const int cout = D.a, cin = D.b, taps = D.c;
      const int tap = t % taps, ci = (t / taps) % cin, co = t / (taps * cin);
      const int fi = (tap * cout + co) * cin + ci;
      const int di = ((taps - 1 - tap) * cin + ci) * cout + co;
      if (D.fwd) ((uint16_t *)D.fwd)[fi] = h;
      if (D.fwd_lo) ((uint16_t *)D.fwd_lo)[fi] = l;
      if (D.dgrad) ((uint16_t *)D.dgrad)[di] = h;
      if (D.dgrad_lo) ((uint16_t *)D.dgrad_lo)[di] = l;
    }
  }
}

extern "C" int d3d_sparse_conv_has_epilogue(int cin, int cout, int dtype);

extern "C" int d3d_pack_all(const D3dPackDesc *descs, int n, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(descs && n >= 0, "d3d_pack_all: arguments");
  // two passes over the descriptors: those the output-major kernel takes (channel counts in multiples of 8), then the rest
  for (int pass = 0; pass < 2; ++pass) {
    PackAllArgs A;
    A.n = 0;
    int blocks = 0;
    auto flush = [&]() -> int {
      if (blocks > 0) {
        if (pass == 0) hipLaunchKernelGGL(pack_all_kernel, dim3(blocks), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL(pack_all_generic_kernel, dim3(blocks), dim3(256), 0, stream, A);
        D3D_LAUNCH_TRY("pack_all_kernel");
      }
      A.n = 0;
      blocks = 0;
      return D3D_OK;
    };
    for (int i = 0; i < n; ++i) {
      const D3dPackDesc &s = descs[i];
      D3D_REQUIRE(s.weight && (s.fwd || s.dgrad) && s.a >= 1 && s.b >= 1 && s.c >= 1 && (s.kind == 0 || s.kind == 1),
                  "d3d_pack_all: descriptor %d", i);
      D3D_REQUIRE((int64_t)s.a * s.b * s.c < ((int64_t)1 << 30), "d3d_pack_all: tensor %d too large", i);
      const int fast_f = s.b, fast_d = s.kind == 0 ? s.c : s.a;  // cin | cout: the fastest dimensions of the two layouts
      const bool vec8 = fast_f % 8 == 0 && fast_d % 8 == 0;
      if (vec8 != (pass == 0)) continue;
      PackDescDev &d = A.d[A.n];
      d.w = s.weight; d.fwd = s.fwd; d.fwd_lo = s.fwd_lo; d.dgrad = s.dgrad; d.dgrad_lo = s.dgrad_lo;
      d.kind = s.kind; d.a = s.a; d.b = s.b; d.c = s.c; d.flip = s.flip;
      d.mode_fwd = d.mode_dgrad = 1;
      if (s.kind == 0) {  // as d3d_pack_conv_weights: f32 features -> hi / lo planes where an MFMA tiling exists
        D3D_REQUIRE(s.dtype == D3D_F32 || s.dtype == D3D_BF16, "d3d_pack_all: dtype %d", s.dtype);
        d.mode_fwd = s.dtype == D3D_BF16 ? 1 : (d3d_sparse_conv_has_epilogue(s.b, s.c, s.dtype) ? 2 : 0);
        d.mode_dgrad = s.dtype == D3D_BF16 ? 1 : (d3d_sparse_conv_has_epilogue(s.c, s.b, s.dtype) ? 2 : 0);
      }
      d.block_begin = blocks;
      const int per_layout = (s.a * s.b * s.c + PACK_ELEMS_PER_BLOCK - 1) / PACK_ELEMS_PER_BLOCK;
      blocks += pass == 0 ? per_layout * ((s.fwd ? 1 : 0) + (s.dgrad ? 1 : 0)) : per_layout;
      if (++A.n == D3D_PACK_MAX)
        if (int rc = flush()) return rc;
    }
    if (int rc = flush()) return rc;
  }
  return D3D_OK;
}
