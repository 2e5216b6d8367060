// TEST INFRASTRUCTURE.  Compiles the DEVICE FUNCTIONS of the reference's iou3d CUDA file for the host:
//   det3d/ops/iou3d/src/iou3d_kernel.cu:13-221 (Point, cross, intersection, box_overlap, iou_bev)
//   det3d/ops/iou3d/src/iou3d_kernel.cu:295-304 (iou_normal)
// The file is plain C++ apart from the CUDA qualifiers, which the macros below define away; the
// <<<...>>> launchers at its end (:354-) do not parse as C++, so oracle/build.py hands this shim a
// temporary prefix of the file (everything before the first launcher, written to a temp directory,
// never into the repo) as IOU3D_DEV_INC.  The greedy keep loop restates iou3d.cpp:101-116.
// Output: oracle/_ref/d3d_ref_iou3d*.so (git-ignored, travels with gpurun).
#include <math.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

using std::max;
using std::min;
#define __device__
#define __global__ static
#define __shared__ static
namespace {
struct Dim3Shim {
  int x, y, z;
};
Dim3Shim blockIdx, threadIdx, blockDim;
inline void __syncthreads() {}
#include IOU3D_DEV_INC
}  // namespace

namespace py = pybind11;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;

template <float (*F)(const float *, const float *)>
static py::array_t<float> pairwise(farr a, farr b) {
  const int m = (int)a.shape(0), n = (int)b.shape(0);
  py::array_t<float> out({m, n});
  float *o = out.mutable_data();
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) o[i * n + j] = F(a.data() + 5 * i, b.data() + 5 * j);
  return out;
}

// boxes sorted by descending score; suppress j > i when iou(i, j) > thresh (kernel :284, host loop
// iou3d.cpp:101-116 -- the bitmask there is only a layout of the same pairwise predicate)
template <float (*F)(const float *, const float *)>
static std::vector<int> greedy(farr boxes, float thresh) {
  const int n = (int)boxes.shape(0);
  std::vector<char> gone(n, 0);
  std::vector<int> keep;
  for (int i = 0; i < n; ++i) {
    if (gone[i]) continue;
    keep.push_back(i);
    for (int j = i + 1; j < n; ++j)
      if (!gone[j] && F(boxes.data() + 5 * i, boxes.data() + 5 * j) > thresh) gone[j] = 1;
  }
  return keep;
}

PYBIND11_MODULE(d3d_ref_iou3d, m) {
  m.doc() = "compiled Det3D iou3d device functions (oracle cross-check only)";
  m.def("boxes_overlap_bev", &pairwise<box_overlap>);
  m.def("boxes_iou_bev", &pairwise<iou_bev>);
  m.def("boxes_iou_normal", &pairwise<iou_normal>);
  m.def("nms", &greedy<iou_bev>);
  m.def("nms_normal", &greedy<iou_normal>);
}
