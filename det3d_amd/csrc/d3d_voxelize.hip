// Batched hard voxelizer for gfx950: a deterministic parallel restatement of the
// reference's serial first-come loop (det3d/ops/point_cloud/point_cloud_ops.py:7-55,
// point_cloud_ops.h:12-71), bit-exact in voxel ids, per-voxel point order and the
// max_voxels early `break`:
//   1. hash every in-range point by its (b,z,y,x) cell; atomicMin keeps each
//      cell's first point index                         (vox_insert)
//   2. a point is its cell's "leader" iff it is that first index; an exclusive
//      scan over leader flags in point order IS the serial voxel numbering
//   3. the leader whose per-frame rank == max_voxels is where the serial loop
//      breaks: its index is the frame's cut, every later point is dropped
//   4. each surviving point bubbles its index into its voxel's T-slot list with
//      atomicMin, which leaves the T smallest indices in ascending order
//      (= the first T points in input order)             (vox_assign)
//   5. gather rows / fused mean                          (vox_fill, vox_mean)
// All passes are HBM/L2-bound integer work: one coalesced read of the points
// per pass, 4-byte atomics into tables that fit L2.
#include "d3d_scan.h"
#include "d3d_fill.h"

#define VOX_SLOT_EMPTY 0x7f7f7f7f  // memset(0x7f) pattern, larger than any point index
#define VOX_MAX_BATCH 1024

struct VoxParams {
  const float *points;
  int64_t n;
  int stride;
  const int32_t *frame_off;
  int batch;
  float vs[3];
  float lo[3];
  int grid[3];  // x, y, z
  int T;
  int M;
};

__device__ __forceinline__ int vox_find_frame(const int32_t *off, int batch, int64_t i) {
  int b = 0;
  while (b + 1 < batch && i >= off[b + 1]) ++b;
  return b;
}

// reference arithmetic: c = floor((p - lo) / vs) in f32 with a true division
// (point_cloud_ops.py:36); reject c < 0 or c >= grid.  NaN is rejected too.
__device__ __forceinline__ bool vox_cell(const VoxParams &P, const float *p, int &x, int &y,
                                         int &z) {
  float cx = floorf((p[0] - P.lo[0]) / P.vs[0]);
  if (!(cx >= 0.f && cx < (float)P.grid[0])) return false;
  float cy = floorf((p[1] - P.lo[1]) / P.vs[1]);
  if (!(cy >= 0.f && cy < (float)P.grid[1])) return false;
  float cz = floorf((p[2] - P.lo[2]) / P.vs[2]);
  if (!(cz >= 0.f && cz < (float)P.grid[2])) return false;
  x = (int)cx;
  y = (int)cy;
  z = (int)cz;
  return true;
}

__global__ void __launch_bounds__(256)
    vox_insert_kernel(VoxParams P, uint32_t *hkeys, uint32_t hmask, int32_t *hfirst,
                      uint32_t *pslot) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const float *p = P.points + i * P.stride;
  int x, y, z;
  if (!vox_cell(P, p, x, y, z)) {
    pslot[i] = D3D_EMPTY_KEY;
    return;
  }
  int b = vox_find_frame(P.frame_off, P.batch, i);
  uint32_t key = (uint32_t)((((int64_t)b * P.grid[2] + z) * P.grid[1] + y) * P.grid[0] + x);
  uint32_t s = d3d_hash_insert(hkeys, hmask, key);
  atomicMin(&hfirst[s], (int)i);
  pslot[i] = s;
}

struct VoxLeaderFlag {
  const uint32_t *pslot;
  const int32_t *hfirst;
  __device__ int operator()(int64_t i) const {
    uint32_t s = pslot[i];
    return (s != D3D_EMPTY_KEY && hfirst[s] == (int)i) ? 1 : 0;
  }
};

struct VoxRankSink {
  const uint32_t *pslot;
  int32_t *hrank;      // per hash slot: global (whole-batch) voxel rank
  int32_t *rank_base;  // per frame: rank at the frame's first point
  const int32_t *frame_off;
  int batch;
  __device__ void operator()(int64_t i, int v, int excl) const {
    if (v) hrank[pslot[i]] = excl;
    for (int b = 0; b < batch; ++b)
      if (i == frame_off[b]) rank_base[b] = excl;
  }
};

// one thread: per-frame voxel counts, packed output bases, cut initialisation
__global__ void vox_meta_kernel(VoxParams P, const int *scan_total, int32_t *rank_base,
                                int32_t *outbase, int32_t *cut, int32_t *num_voxels) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int total = *scan_total;
  for (int b = 0; b < P.batch; ++b)
    if (P.frame_off[b] >= P.n) rank_base[b] = total;
  rank_base[P.batch] = total;
  int acc = 0;
  for (int b = 0; b < P.batch; ++b) {
    int v = rank_base[b + 1] - rank_base[b];
    int nv = v < P.M ? v : P.M;
    num_voxels[b] = nv;
    outbase[b] = acc;
    acc += nv;
    cut[b] = P.frame_off[b + 1];
  }
  outbase[P.batch] = acc;
  num_voxels[P.batch] = acc;
}

__global__ void __launch_bounds__(256)
    vox_leader_kernel(VoxParams P, const uint32_t *pslot, const int32_t *hfirst,
                      const int32_t *hrank, const int32_t *rank_base, const int32_t *outbase,
                      int32_t *cut, int32_t *coors) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  uint32_t s = pslot[i];
  if (s == D3D_EMPTY_KEY || hfirst[s] != (int)i) return;
  int b = vox_find_frame(P.frame_off, P.batch, i);
  int local = hrank[s] - rank_base[b];
  if (local == P.M) cut[b] = (int)i;  // the serial loop breaks here (point_cloud_ops.py:46-47)
  if (local < P.M) {
    int x, y, z;
    vox_cell(P, P.points + i * P.stride, x, y, z);
    int4 c = make_int4(b, z, y, x);
    reinterpret_cast<int4 *>(coors)[outbase[b] + local] = c;
  }
}

__global__ void __launch_bounds__(256)
    vox_assign_kernel(VoxParams P, const uint32_t *pslot, const int32_t *hrank,
                      const int32_t *rank_base, const int32_t *outbase, const int32_t *cut,
                      int32_t *count, int32_t *slots) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  uint32_t s = pslot[i];
  if (s == D3D_EMPTY_KEY) return;
  int b = vox_find_frame(P.frame_off, P.batch, i);
  if ((int)i >= cut[b]) return;
  int row = outbase[b] + (hrank[s] - rank_base[b]);
  atomicAdd(&count[row], 1);
  int32_t *sl = slots + (int64_t)row * P.T;
  int v = (int)i;
  for (int t = 0; t < P.T; ++t) {
    int old = atomicMin(&sl[t], v);
    if (old == VOX_SLOT_EMPTY) break;
    v = old > v ? old : v;  // the larger index moves on to the next slot
  }
}

__global__ void __launch_bounds__(256)
    vox_fill_kernel(VoxParams P, const int32_t *outbase, const int32_t *count,
                    const int32_t *slots, int32_t *num_points, float *voxels) {
  const int64_t total = (int64_t)outbase[P.batch] * P.T;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = g / P.T;
    int t = (int)(g - row * P.T);
    int n = count[row];
    n = n < P.T ? n : P.T;
    if (t == 0) num_points[row] = n;
    if (voxels) {
      float *dst = voxels + g * P.stride;
      if (t < n) {
        const float *src = P.points + (int64_t)slots[g] * P.stride;
        for (int c = 0; c < P.stride; ++c) dst[c] = src[c];
      } else {
        for (int c = 0; c < P.stride; ++c) dst[c] = 0.f;
      }
    }
  }
}

// mean VFE (voxel_encoder.py:206-211): sum over the voxel's points in input
// order, f32, then one division by the point count.
__global__ void __launch_bounds__(256)
    vox_mean_kernel(VoxParams P, const int32_t *outbase, const int32_t *count,
                    const int32_t *slots, int n_ch, float *mean) {
  const int64_t total = (int64_t)outbase[P.batch] * n_ch;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = g / n_ch;
    int c = (int)(g - row * n_ch);
    int n = count[row];
    n = n < P.T ? n : P.T;
    float s = 0.f;
    for (int t = 0; t < n; ++t) s += P.points[(int64_t)slots[row * P.T + t] * P.stride + c];
    mean[g] = s / (float)n;
  }
}

struct VoxLayout {
  int hsize;
  size_t n_tiles;
};

static VoxLayout vox_layout(int64_t n_points) {
  VoxLayout L;
  L.hsize = d3d_next_pow2(n_points * 2 < 1024 ? 1024 : n_points * 2);
  L.n_tiles = (size_t)d3d_scan_num_tiles(n_points) + 1;
  return L;
}

extern "C" size_t d3d_voxelize_workspace_bytes(int64_t n_points, int batch, int max_points,
                                               int max_voxels) {
  VoxLayout L = vox_layout(n_points);
  size_t rows = (size_t)batch * max_voxels;
  size_t b = 0;
  b += d3d_align_up((size_t)L.hsize * 4) * 3;          // hkeys, hfirst, hrank
  b += d3d_align_up((size_t)(n_points + 1) * 4);       // pslot
  b += d3d_align_up(L.n_tiles * 4);                    // scan tiles
  b += d3d_align_up((size_t)(batch + 1) * 4) * 3;      // rank_base, outbase, cut
  b += d3d_align_up(rows * 4);                         // count
  b += d3d_align_up(rows * (size_t)max_points * 4);    // slots
  return b + 256;
}

extern "C" int d3d_voxelize(const float *points, int64_t n_points, int point_stride,
                            const int32_t *frame_offsets, int batch, const float *voxel_size,
                            const float *range, const int32_t *grid, int max_points,
                            int max_voxels, int32_t *coors, int32_t *num_points, float *voxels,
                            float *mean_feats, int n_mean_ch, int32_t *num_voxels,
                            void *workspace, size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(batch >= 1 && batch <= VOX_MAX_BATCH, "d3d_voxelize: batch=%d out of range", batch);
  D3D_REQUIRE(n_points >= 0 && n_points < (int64_t)VOX_SLOT_EMPTY, "d3d_voxelize: n_points");
  D3D_REQUIRE(point_stride >= 3, "d3d_voxelize: point_stride=%d < 3", point_stride);
  D3D_REQUIRE(max_points >= 1 && max_voxels >= 1, "d3d_voxelize: max_points/max_voxels");
  D3D_REQUIRE(n_mean_ch >= 0 && n_mean_ch <= point_stride, "d3d_voxelize: n_mean_ch");
  D3D_REQUIRE(grid[0] > 0 && grid[1] > 0 && grid[2] > 0, "d3d_voxelize: empty grid");
  D3D_REQUIRE(coors && num_points && num_voxels && frame_offsets, "d3d_voxelize: null output");
  const double cells = (double)batch * grid[0] * grid[1] * grid[2];
  if (cells >= 4294967295.0)
    return d3d_set_error(D3D_ECAPACITY, "d3d_voxelize: batch*grid = %.0f cells exceeds u32 keys",
                         cells);
  if (workspace_bytes < d3d_voxelize_workspace_bytes(n_points, batch, max_points, max_voxels))
    return d3d_set_error(D3D_ECAPACITY, "d3d_voxelize: workspace too small");

  VoxLayout L = vox_layout(n_points);
  const size_t rows = (size_t)batch * max_voxels;
  D3dArena A(workspace, workspace_bytes);
  uint32_t *hkeys = A.take<uint32_t>(L.hsize);
  int32_t *hfirst = A.take<int32_t>(L.hsize);
  int32_t *hrank = A.take<int32_t>(L.hsize);
  uint32_t *pslot = A.take<uint32_t>(n_points + 1);
  int *tiles = A.take<int>(L.n_tiles);
  int32_t *rank_base = A.take<int32_t>(batch + 1);
  int32_t *outbase = A.take<int32_t>(batch + 1);
  int32_t *cut = A.take<int32_t>(batch + 1);
  int32_t *count = A.take<int32_t>(rows);
  int32_t *slots = A.take<int32_t>(rows * max_points);
  if (!slots) return d3d_set_error(D3D_ECAPACITY, "d3d_voxelize: workspace carve failed");

  VoxParams P;
  P.points = points;
  P.n = n_points;
  P.stride = point_stride;
  P.frame_off = frame_offsets;
  P.batch = batch;
  for (int j = 0; j < 3; ++j) {
    P.vs[j] = voxel_size[j];
    P.lo[j] = range[j];
    P.grid[j] = grid[j];
  }
  P.T = max_points;
  P.M = max_voxels;

  {
    const D3dFillSeg segs[4] = {{hkeys, 0xFF, (size_t)L.hsize * 4},
                                {hfirst, 0x7f, (size_t)L.hsize * 4},
                                {count, 0, rows * 4},
                                {slots, 0x7f, rows * (size_t)max_points * 4}};
    D3D_HIP_TRY(d3d_fill_async(stream, segs, 4));  // one launch instead of four hipMemsetAsync
  }

  const int nb = (int)((n_points + 255) / 256);
  if (nb > 0) {
    hipLaunchKernelGGL(vox_insert_kernel, dim3(nb), dim3(256), 0, stream, P, hkeys,
                       (uint32_t)(L.hsize - 1), hfirst, pslot);
    D3D_LAUNCH_TRY("vox_insert_kernel");
  }
  VoxLeaderFlag F{pslot, hfirst};
  VoxRankSink G{pslot, hrank, rank_base, frame_offsets, batch};
  int rc = d3d_exclusive_scan(F, G, n_points, tiles, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(vox_meta_kernel, dim3(1), dim3(64), 0, stream, P,
                     tiles + d3d_scan_num_tiles(n_points), rank_base, outbase, cut, num_voxels);
  D3D_LAUNCH_TRY("vox_meta_kernel");
  if (nb > 0) {
    hipLaunchKernelGGL(vox_leader_kernel, dim3(nb), dim3(256), 0, stream, P, pslot, hfirst, hrank,
                       rank_base, outbase, cut, coors);
    D3D_LAUNCH_TRY("vox_leader_kernel");
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, stream, P, pslot, hrank,
                       rank_base, outbase, cut, count, slots);
    D3D_LAUNCH_TRY("vox_assign_kernel");
  }
  int64_t cap_items = (int64_t)rows * max_points;
  int fb = (int)((cap_items + 255) / 256);
  if (fb > 4096) fb = 4096;
  if (fb < 1) fb = 1;
  hipLaunchKernelGGL(vox_fill_kernel, dim3(fb), dim3(256), 0, stream, P, outbase, count, slots,
                     num_points, voxels);
  D3D_LAUNCH_TRY("vox_fill_kernel");
  if (mean_feats && n_mean_ch > 0) {
    int64_t items = (int64_t)rows * n_mean_ch;
    int mb = (int)((items + 255) / 256);
    if (mb > 2048) mb = 2048;
    if (mb < 1) mb = 1;
    hipLaunchKernelGGL(vox_mean_kernel, dim3(mb), dim3(256), 0, stream, P, outbase, count, slots,
                       n_mean_ch, mean_feats);
    D3D_LAUNCH_TRY("vox_mean_kernel");
  }
  return D3D_OK;
}
