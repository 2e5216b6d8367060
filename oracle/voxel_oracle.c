/*
 * TEST INFRASTRUCTURE -- CPU oracle, not product code.  Only tests/, the smoke
 * check and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of the reference hard voxelizer
 *   det3d/ops/point_cloud/point_cloud_ops.py:7-55  (_points_to_voxel_reverse_kernel)
 *   det3d/ops/point_cloud/point_cloud_ops.py:112-184 (points_to_voxel wrapper:
 *       dense -1 grid, zeroed outputs, slicing to voxel_num)
 * pinned against (a) the reference's own C++ twin point_cloud_ops.h:12-71
 * compiled into oracle/_ref and (b) golden vectors produced by running the
 * reference Python loop itself (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* grid = round((hi - lo) / vs) in float32, as point_cloud_ops.py:26-29 */
void d3d_oracle_grid_size(const float *vs, const float *range, int *grid) {
  for (int j = 0; j < 3; ++j) {
    float g = (range[3 + j] - range[j]) / vs[j];
    grid[j] = (int)rintf(g); /* np.round: half to even, same as rintf */
  }
}

/*
 * points [n, stride] f32; outputs sized for max_voxels rows and zero/(-1)
 * initialised here.  coors are (z, y, x).  Returns voxel_num, or -1 if the
 * dense grid cannot be allocated.
 */
int d3d_oracle_points_to_voxel(const float *points, int64_t n, int stride, const float *vs,
                               const float *range, int max_points, int max_voxels,
                               float *voxels, int32_t *coors, int32_t *num_points_per_voxel) {
  int grid[3];
  d3d_oracle_grid_size(vs, range, grid);
  const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  /* the reference allocates and fills this dense map on every call (:150) */
  int32_t *coor_to_voxelidx = (int32_t *)malloc(cells * sizeof(int32_t));
  if (!coor_to_voxelidx) return -1;
  memset(coor_to_voxelidx, 0xFF, cells * sizeof(int32_t));
  memset(num_points_per_voxel, 0, (size_t)max_voxels * sizeof(int32_t));
  memset(voxels, 0, (size_t)max_voxels * max_points * stride * sizeof(float));
  memset(coors, 0, (size_t)max_voxels * 3 * sizeof(int32_t));

  int voxel_num = 0;
  int coor[3];
  for (int64_t i = 0; i < n; ++i) {
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      /* float32 subtract, float32 true division, floor (:36) */
      float c = floorf((points[i * stride + j] - range[j]) / vs[j]);
      if (c < 0 || c >= (float)grid[j]) {
        failed = 1;
        break;
      }
      coor[2 - j] = (int)c;
    }
    if (failed) continue;
    size_t lin = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int voxelidx = coor_to_voxelidx[lin];
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (voxel_num >= max_voxels) break; /* drops every later point (:46-47) */
      voxel_num += 1;
      coor_to_voxelidx[lin] = voxelidx;
      coors[voxelidx * 3 + 0] = coor[0];
      coors[voxelidx * 3 + 1] = coor[1];
      coors[voxelidx * 3 + 2] = coor[2];
    }
    int num = num_points_per_voxel[voxelidx];
    if (num < max_points) {
      memcpy(voxels + ((size_t)voxelidx * max_points + num) * stride, points + i * stride,
             (size_t)stride * sizeof(float));
      num_points_per_voxel[voxelidx] += 1;
    }
  }
  free(coor_to_voxelidx);
  return voxel_num;
}
