// Deterministic cross-block reduction of per-channel sums (BatchNorm statistics, BatchNorm-backward sums).
//
// Round 2 summed these with f32 atomics (LDS, then replicated global accumulators): the last bits depended on
// arrival order, 14 train-mode BN + ReLU layers amplified that, and test tolerances had to be calibrated against
// the step's own run-to-run noise.  Here every block hands in ONE partial vector, summed in a fixed order inside
// the block, and the vectors are folded in a fixed order across blocks: the result is a pure function of
// (input, grid) -- two identical launches are bit-equal.
//
// Layout of the caller's workspace `ws` (f32): [nval] result | [nb][nval] block slabs | [ng][nval] group slabs.
// Fold: blocks are grouped G consecutive block ids at a time (G = 16; 32 above 256 blocks, 64 above 2048).  The LAST block
// of a group to arrive (ticket counter) sums the group's slabs in ascending block order into the group slab;
// the last GROUP to finish sums the group slabs in ascending order into ws[0..nval).  Which block does the
// summing depends on timing, WHAT is summed in which order does not.  The hand-off is recipe R1 of
// cdna_hip_programming.md 6 Guideline 16: write-through (sc1) slab stores by every writing lane, vmcnt(0) in every
// storing wave + workgroup barrier, then ONE lane's relaxed agent-scope ticket (no release fence: a release would
// write back every dirty line of the XCD's L2 -- i.e. the conv's own output tile -- once per block: measured +15 us
// on a 44 us launch); the folding block issues one agent-scope acquire fence, then reads the slabs with sc1 loads.
//
// Ticket counters live in the ring of zeroed words of the calling thread's context (d3d_fold_counters, d3d_core.hip;
// d3d_context.h: one ring per context and device, so two contexts never share a word): a counter is
// in use for the duration of ONE kernel and is reset by the block that draws the last ticket, so the ring never
// needs re-zeroing; the ring is long enough (2^16 words) that two launches in flight never share a word.
#pragma once
#include "d3d_common.h"

// The slab hand-off below (write-through stores + vmcnt(0) + relaxed ticket, no release fence) relies on gfx942 / gfx950
// behaviour: sc1 stores leave the XCD's L2 before the counter increment is visible.  Any other target must use the
// release / acquire form of cdna_hip_programming.md 6 Guideline 16.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "d3d_fold.h: the fence-free slab hand-off is only valid on gfx942 / gfx950"
#endif

#define D3D_FOLD_MAX_BLOCKS 8192

struct D3dFoldPlan {
  int nb;      // blocks of the producing launch
  int g;       // blocks per group
  int ng;      // groups
  int nval;    // values per slab
};

static inline D3dFoldPlan d3d_fold_plan(int nb, int nval) {
  D3dFoldPlan p;
  p.nb = nb;
  p.g = nb <= 256 ? 16 : (nb <= 2048 ? 32 : 64);
  p.ng = (nb + p.g - 1) / p.g;
  p.nval = nval;
  return p;
}
// floats the caller's workspace must hold for a launch of `nb` blocks
static inline size_t d3d_fold_ws_floats(int nb, int nval) {
  const D3dFoldPlan p = d3d_fold_plan(nb, nval);
  return (size_t)nval * (1 + (size_t)p.nb + (size_t)p.ng);
}
// device pointer to `count` zeroed ticket words for ONE launch on the current device (nullptr + error set on failure)
unsigned *d3d_fold_counters(int count);
static inline int d3d_fold_counter_words(const D3dFoldPlan &p) { return p.ng + 1; }

typedef __attribute__((address_space(1))) float d3d_gf32;
typedef __attribute__((address_space(1))) unsigned d3d_gu32;

__device__ __forceinline__ void d3d_store_sc1(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float d3d_load_sc1(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every thread of the block calls this after the block's `nval` partial sums sit in LDS `vals` (visible to all
// threads, i.e. after a barrier).  `flag` is one LDS word.  NT = threads of the block.
// Returns true in the ONE block that drew the last ticket: the result is then complete in ws[0..nval) (written by
// this block's threads i = threadIdx.x, threadIdx.x + NT, ...); every other block returns false.
// `b` of `nb`: this block's position in the fold (d3d_fold_block: its blockIdx.x of gridDim.x; a launch whose blocks
// fold in several independent sets -- conv2d_halo_kernel: one set per 128-channel tile -- passes its own numbering)
template <int NT>
__device__ __forceinline__ bool d3d_fold_block_at(const float *vals, int nval, float *__restrict__ ws,
                                                  unsigned *__restrict__ counters, int g, int *flag, int b, int nb) {
  const int ng = (nb + g - 1) / g;
  float *slabs = ws + nval;
  float *gslabs = slabs + (size_t)nb * nval;
  const int grp = b / g;
  const int gfirst = grp * g;
  const int gcount = (gfirst + g <= nb) ? g : nb - gfirst;
  // ---- level 0: this block's slab
  for (int i = threadIdx.x; i < nval; i += NT) d3d_store_sc1(&slabs[(size_t)b * nval + i], vals[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&counters[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)(gcount - 1));
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&counters[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return false;
  // ---- level 1: the group's slabs, ascending block order (all loads of a thread in flight at once)
  float *dst = ng == 1 ? ws : gslabs + (size_t)grp * nval;
  for (int i = threadIdx.x; i < nval; i += NT) {
    float s = 0.f;
    for (int j0 = 0; j0 < gcount; j0 += 32) {
      float v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j)
        v[j] = d3d_load_sc1(&slabs[(size_t)(gfirst + (j0 + j < gcount ? j0 + j : 0)) * nval + i]);
#pragma unroll
      for (int j = 0; j < 32; ++j) s += j0 + j < gcount ? v[j] : 0.f;
    }
    if (ng == 1) dst[i] = s;
    else d3d_store_sc1(&dst[i], s);
  }
  if (ng == 1) return true;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&counters[ng], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)(ng - 1));
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&counters[ng], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return false;
  // ---- level 2: the group slabs, ascending group order
  for (int i = threadIdx.x; i < nval; i += NT) {
    float s = 0.f;
    for (int j0 = 0; j0 < ng; j0 += 32) {
      float v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = d3d_load_sc1(&gslabs[(size_t)(j0 + j < ng ? j0 + j : 0) * nval + i]);
#pragma unroll
      for (int j = 0; j < 32; ++j) s += j0 + j < ng ? v[j] : 0.f;
    }
    ws[i] = s;
  }
  return true;
}

template <int NT>
__device__ __forceinline__ bool d3d_fold_block(const float *vals, int nval, float *__restrict__ ws,
                                               unsigned *__restrict__ counters, int g, int *flag) {
  return d3d_fold_block_at<NT>(vals, nval, ws, counters, g, flag, (int)blockIdx.x, (int)gridDim.x);
}
