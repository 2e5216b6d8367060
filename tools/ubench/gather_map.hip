// Micro-benchmark (round 5): what does the lane -> address mapping of a row gather cost on MI355X?
// The sparse forward kernel (csrc/d3d_spconv.hip) gathers f32 rows of 256 B straight into MFMA B-operand registers:
// lane (lr = lane & 15, lg = lane >> 4) reads 16 B of row idx[site lr] -- the four lanes of a quad read four DIFFERENT
// rows.  Variants, same rows, same bytes, same number of dwordx4 instructions (8 per wave and offset for 32 sites):
//   A  operand mapping (today):     lane (lr, lg): row lr, bytes ks*128 + lg*32 + half*16
//   B  quad-contiguous 64 B:        lane l: row l >> 2, bytes chunk*64 + (l & 3)*16          (16 rows per instruction)
//   D  operand lanes, 64 B per row:  lane (lr, lg): row lr, bytes q*64 + lg*16 (a K-permutation of the MFMA operand)
//   C  row-contiguous 256 B:        lane l: row l >> 4, bytes (l & 15)*16                    (4 rows per instruction)
// No MFMA, no LDS: the loaded dwords are folded with v_xor so that nothing can be dropped.  Neighbour table: sites in
// cell order, offset k = (dz, dy, dx) -> site + dz*2900 + dy*7 + dx + jitter, valid with probability `fill`.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_map.hip -o tools/ubench/gather_map && tools/ubench/gather_map
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define OOB 0xFFFFFFF0u

__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}

template <int MAP, int DEPTH, int NMFMA = 0, bool AG = false, bool NOGATHER = false>
__global__ void __launch_bounds__(512, 4) gather_kernel(const float *feat, const int32_t *nbr, int n, int kvol,
                                                        uint32_t *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 8 + wave) * 32;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)feat, 0, 0x7FFFFFF0, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  // NMFMA independent-of-the-gather MFMAs per offset on 8 accumulators (operands: loop-invariant registers), as the
  // forward kernel issues them (48 per offset); AG: accumulators in AGPRs (inline asm), else wherever hipcc puts them
  typedef __attribute__((ext_vector_type(4))) float f4;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
  f4 macc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) macc[i] = f4{0.f, 0.f, 0.f, 0.f};
  const bf8 opa = __builtin_bit_cast(bf8, u32x4{0x3f803f80u + (uint32_t)lane, 0x3f803f80u, 0x3c003c00u, 0x3f803f80u});
  const bf8 opb = __builtin_bit_cast(bf8, u32x4{0x3f803f80u, 0x3c003c00u + (uint32_t)wave, 0x3f803f80u, 0x3f803f80u});
  // per instruction j (0..7): which of the wave's 32 sites this lane fetches, and the byte offset inside the row
  int ls[8];
  uint32_t lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (MAP == 0) {
      const int m = j >> 2, ks = (j >> 1) & 1, half = j & 1;
      ls[j] = m * 16 + (lane & 15);
      lo[j] = (uint32_t)(ks * 128 + (lane >> 4) * 32 + half * 16);
    } else if (MAP == 3) {  // D: operand lanes, but the four k-groups of a row read 64 CONTIGUOUS bytes (k-slots permuted)
      const int m = j >> 2, q = j & 3;
      ls[j] = m * 16 + (lane & 15);
      lo[j] = (uint32_t)(q * 64 + (lane >> 4) * 16);
    } else if (MAP == 1) {
      ls[j] = (j & 1) * 16 + (lane >> 2);
      lo[j] = (uint32_t)((j >> 1) * 64 + (lane & 3) * 16);
    } else {
      ls[j] = j * 4 + (lane >> 4);
      lo[j] = (uint32_t)((lane & 15) * 16);
    }
  }
  // as in the forward kernel: the neighbour indices of offset k are loaded DEPTH offsets before its rows are requested,
  // the rows DEPTH offsets before they are used -- one exposed latency per offset, not two
  u32x4 fr[DEPTH][8];
  int ix[DEPTH][8];
  auto load_idx = [&](int k, int (&x)[8]) {
    k = k < kvol ? k : kvol - 1;
    const int32_t *nk = nbr + (int64_t)k * n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = row0 + ls[j];
      x[j] = nk[s < n ? s : n - 1];
    }
  };
  auto issue = [&](const int (&x)[8], u32x4 (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = row0 + ls[j] < n && x[j] >= 0;
      f[j] = bload(rs, ok ? (uint32_t)x[j] * 256u + lo[j] : OOB);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_idx(d, ix[d]);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    issue(ix[d], fr[d]);
    load_idx(d + DEPTH, ix[d]);
  }
  for (int kb = 0; kb < kvol; kb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= fr[d][j];
#pragma unroll
      for (int q = 0; q < NMFMA; ++q) {
        if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(macc[q & 7]) : "v"(opa), "v"(opb));
        else macc[q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(opa, opb, macc[q & 7], 0, 0, 0);
      }
      if constexpr (!NOGATHER) issue(ix[d], fr[d]);
      load_idx(kb + d + 2 * DEPTH, ix[d]);
    }
  }
  float ms = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ms += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ (NMFMA ? __float_as_uint(ms) : 0u);
}

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <int MAP, int DEPTH, int NMFMA = 0, bool AG = false, bool NOGATHER = false>
static void run(const char *name, const float *feat, const int32_t *nbr, int n, int kvol, uint32_t *out, double bytes) {
  const int nb = (n + 255) / 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather_kernel<MAP, DEPTH, NMFMA, AG, NOGATHER>), dim3(nb), dim3(512), 0, 0, feat, nbr, n, kvol, out);
  hipEventRecord(e0, 0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<MAP, DEPTH, NMFMA, AG, NOGATHER>), dim3(nb), dim3(512), 0, 0, feat, nbr, n, kvol, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000 / reps;
  uint32_t h = 0;
  std::vector<uint32_t> ho((size_t)nb * 512);
  hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  for (uint32_t v : ho) h ^= v;
  printf("  %-34s depth %d  %7.1f us  %7.0f GB/s of valid rows   xor %08x\n", name, DEPTH, us, bytes / us / 1e3, h);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 126683, kvol = 27;
  const double fill = argc > 2 ? atof(argv[2]) : 0.49;
  std::vector<int32_t> nbr((size_t)kvol * n);
  size_t valid = 0;
  for (int k = 0; k < kvol; ++k) {
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    for (int s = 0; s < n; ++s) {
      int t = s + dz * 2900 + dy * 7 + dx + (int)(rnd() % 7) - 3;
      const bool ok = (rnd() % 10000) < fill * 10000 && t >= 0 && t < n;
      nbr[(size_t)k * n + s] = ok ? t : -1;
      valid += ok;
    }
  }
  float *feat;
  int32_t *dn;
  uint32_t *out;
  hipMalloc(&feat, (size_t)n * 256);
  hipMalloc(&dn, nbr.size() * 4);
  hipMalloc(&out, (size_t)((n + 255) / 256) * 512 * 4);
  std::vector<float> hf((size_t)n * 64);
  for (auto &v : hf) v = (float)(rnd() % 1000) * 0.001f;
  hipMemcpy(feat, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice);
  const double bytes = (double)valid * 256;
  printf("gather of f32 rows (256 B): n = %d sites, %zu valid pairs (%.2f per site), %.1f MB of valid rows\n", n, valid,
         (double)valid / n, bytes / 1e6);
  run<0, 1>("A operand mapping (today)", feat, dn, n, kvol, out, bytes);
  run<1, 1>("B quad-contiguous 64 B", feat, dn, n, kvol, out, bytes);
  run<2, 1>("C row-contiguous 256 B", feat, dn, n, kvol, out, bytes);
  run<3, 1>("D operand lanes, 64 B sectors", feat, dn, n, kvol, out, bytes);
  run<3, 1, 48, false>("D + 48 MFMA / offset", feat, dn, n, kvol, out, bytes);
  run<0, 1, 48, false, true>("48 MFMA / offset, no gather", feat, dn, n, kvol, out, bytes);
  run<0, 1, 48, true, true>("48 MFMA / offset (AGPR), no gather", feat, dn, n, kvol, out, bytes);
  run<0, 1, 48, false>("A + 48 MFMA / offset", feat, dn, n, kvol, out, bytes);
  run<0, 1, 48, true>("A + 48 MFMA / offset (AGPR acc)", feat, dn, n, kvol, out, bytes);
  run<1, 1, 48, false>("B + 48 MFMA / offset", feat, dn, n, kvol, out, bytes);
  run<1, 1, 48, true>("B + 48 MFMA / offset (AGPR acc)", feat, dn, n, kvol, out, bytes);
  run<0, 2, 48, false>("A + 48 MFMA / offset", feat, dn, n, kvol, out, bytes);
  run<0, 2, 48, true>("A + 48 MFMA / offset (AGPR acc)", feat, dn, n, kvol, out, bytes);
  run<0, 2>("A operand mapping (today)", feat, dn, n, kvol, out, bytes);
  run<1, 2>("B quad-contiguous 64 B", feat, dn, n, kvol, out, bytes);
  run<2, 2>("C row-contiguous 256 B", feat, dn, n, kvol, out, bytes);
  return 0;
}
