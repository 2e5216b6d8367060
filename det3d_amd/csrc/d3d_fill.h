// Byte-pattern fills as ordinary kernel dispatches.
//
// hipMemsetAsync goes through ROCclr's blit path (`__amd_rocclr_fillBufferAligned`); between two kernels
// on the same stream it costs a 13-16 us bubble on the MI355X (profiles/round1_gaps_v5_memset.txt: ~90 fills
// per training step = 1.1-1.6 ms of idle GPU under the profiler, and the latency is bimodal).  A plain kernel keeps the
// queue in dispatch mode (2-3 us), and up to four regions with different byte values go in ONE launch
// (the voxelizer's hash keys / first indices / counters / slots, a rulebook's table + counters, ...).
#pragma once
#include "d3d_common.h"

struct D3dFillSeg {
  void *p;
  uint32_t byte;  // memset semantics: the low byte, replicated
  size_t bytes;
};
struct D3dFillArgs {
  D3dFillSeg s[4];
  int n;
};

static __global__ void __launch_bounds__(256) d3d_fill_kernel(D3dFillArgs a) {
  for (int k = 0; k < a.n; ++k) {
    uint8_t *p = (uint8_t *)a.s[k].p;
    const size_t bytes = a.s[k].bytes;
    const uint32_t b = a.s[k].byte & 0xffu, v = b | (b << 8) | (b << 16) | (b << 24);
    size_t head = (16 - ((uintptr_t)p & 15)) & 15;
    if (head > bytes) head = bytes;
    const size_t nvec = (bytes - head) / 16, tail0 = head + nvec * 16;
    uint4 *pv = (uint4 *)(p + head);
    const uint4 vv = make_uint4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x)
      pv[i] = vv;
    if (blockIdx.x == 0) {
      for (size_t i = threadIdx.x; i < head; i += blockDim.x) p[i] = (uint8_t)b;
      for (size_t i = tail0 + threadIdx.x; i < bytes; i += blockDim.x) p[i] = (uint8_t)b;
    }
  }
}

// returns a hipError_t-compatible int (0 = ok) so that call sites can wrap it in D3D_HIP_TRY
static inline hipError_t d3d_fill_async(hipStream_t stream, const D3dFillSeg *segs, int n) {
  D3dFillArgs a;
  size_t most = 0;
  a.n = 0;
  for (int i = 0; i < n && a.n < 4; ++i) {
    if (!segs[i].p || !segs[i].bytes) continue;
    a.s[a.n++] = segs[i];
    if (segs[i].bytes > most) most = segs[i].bytes;
  }
  if (!a.n) return hipSuccess;
  size_t nb = (most / 16 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(d3d_fill_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a);
  return hipGetLastError();
}

static inline hipError_t d3d_fill_async(hipStream_t stream, void *p, int byte, size_t bytes) {
  D3dFillSeg s = {p, (uint32_t)byte, bytes};
  return d3d_fill_async(stream, &s, 1);
}
