// Shared host/device helpers for libd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/d3d.h"
#include "d3d_context.h"

int d3d_set_error(int code, const char *fmt, ...);

#define D3D_HIP_TRY(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return d3d_set_error(D3D_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                            \
  } while (0)

#define D3D_LAUNCH_TRY(name)                                                             \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess)                                                                \
      return d3d_set_error(D3D_EHIP, "launch of %s failed: %s (%s:%d)", name,             \
                           hipGetErrorString(_e), __FILE__, __LINE__);                   \
  } while (0)

#define D3D_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return d3d_set_error(D3D_EINVAL, __VA_ARGS__);  \
  } while (0)

static inline size_t d3d_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct D3dArena {
  char *base;
  size_t size;
  size_t off;
  D3dArena(void *p, size_t n) : base((char *)p), size(n), off(0) {}
  template <class T> T *take(size_t count) {
    size_t bytes = d3d_align_up(count * sizeof(T));
    if (off + bytes > size) return nullptr;
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
};

static inline int d3d_next_pow2(int64_t x) {
  int64_t p = 1;
  while (p < x) p <<= 1;
  return (int)p;
}

// ---------------------------------------------------------------- device side
#define D3D_WAVE 64

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain EVERY outstanding global
// load first (s_waitcnt vmcnt(0) before s_barrier), which serialises software-pipelined gathers; here the
// in-flight global loads stay in flight and hipcc's own counted vmcnt(N) waits at their first use still apply.
__device__ __forceinline__ void d3d_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Raw buffer descriptor over `p` for gathers whose INVALID lanes must read zeros without a branch and without a
// select on the loaded data (either would make hipcc wait for the load right where it is issued): invalid lanes
// pass D3D_OOB_OFFSET, the hardware's range check (offset >= num_records) returns 0 for them.  num_records is
// fixed at 2 GB - 16: callers guarantee their tensors are smaller (checked on the host side).
typedef __attribute__((ext_vector_type(4))) unsigned int d3d_u32x4;
#define D3D_OOB_OFFSET 0xFFFFFFF0u
#define D3D_RSRC_BYTES 0x7FFFFFF0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t d3d_make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, /*stride*/ 0, D3D_RSRC_BYTES, 0x00020000);
}
__device__ __forceinline__ uint4 d3d_buffer_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  d3d_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 d3d_buffer_load8(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  __attribute__((ext_vector_type(2))) unsigned int v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_offset, 0, 0);
  return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ int d3d_buffer_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_offset, 0, 0);
}
#define D3D_EMPTY_KEY 0xFFFFFFFFu

__device__ __forceinline__ uint32_t d3d_hash_u32(uint32_t k, uint32_t mask) {
  // murmur3 finalizer: keys are linear indices of a regular (z,y,x) lattice, and half of all
  // probes are for ABSENT neighbours that walk to the next empty slot -- a weak mix clusters
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  k *= 0xc2b2ae35u;
  k ^= k >> 16;
  return k & mask;
}

// insert `key` into an open-addressing table of (mask+1) slots; returns slot.
__device__ __forceinline__ uint32_t d3d_hash_insert(uint32_t *keys, uint32_t mask, uint32_t key) {
  uint32_t s = d3d_hash_u32(key, mask);
  while (true) {
    uint32_t cur = keys[s];
    if (cur == key) return s;
    if (cur == D3D_EMPTY_KEY) {
      uint32_t old = atomicCAS(&keys[s], D3D_EMPTY_KEY, key);
      if (old == D3D_EMPTY_KEY || old == key) return s;
    }
    s = (s + 1) & mask;
  }
}

// ---- packed (key, value) table: one 8-byte entry per slot = one load per probe
#define D3D_EMPTY_KV 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ void d3d_kv_insert(unsigned long long *tab, uint32_t mask, uint32_t key,
                                              uint32_t val) {
  const unsigned long long entry = (unsigned long long)key | ((unsigned long long)val << 32);
  uint32_t s = d3d_hash_u32(key, mask);
  while (true) {
    unsigned long long cur = tab[s];
    if ((uint32_t)cur == key && cur != D3D_EMPTY_KV) return;  // duplicate coordinate: first wins
    if (cur == D3D_EMPTY_KV) {
      unsigned long long old = atomicCAS(&tab[s], D3D_EMPTY_KV, entry);
      if (old == D3D_EMPTY_KV || (uint32_t)old == key) return;
    }
    s = (s + 1) & mask;
  }
}

// value stored for `key`, or -1
__device__ __forceinline__ int d3d_kv_find(const unsigned long long *tab, uint32_t mask,
                                           uint32_t key) {
  uint32_t s = d3d_hash_u32(key, mask);
  while (true) {
    unsigned long long cur = tab[s];
    if (cur == D3D_EMPTY_KV) return -1;
    if ((uint32_t)cur == key) return (int)(cur >> 32);
    s = (s + 1) & mask;
  }
}

// returns slot of `key` or 0xFFFFFFFF
__device__ __forceinline__ uint32_t d3d_hash_find(const uint32_t *keys, uint32_t mask,
                                                  uint32_t key) {
  uint32_t s = d3d_hash_u32(key, mask);
  while (true) {
    uint32_t cur = keys[s];
    if (cur == key) return s;
    if (cur == D3D_EMPTY_KEY) return D3D_EMPTY_KEY;
    s = (s + 1) & mask;
  }
}
