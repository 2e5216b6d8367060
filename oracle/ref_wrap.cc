// TEST INFRASTRUCTURE.  Thin pybind11 shim that compiles the REFERENCE's own C++
// sources where they lie under /root/reference (include path given by
// oracle/build.py; nothing is copied into this repo):
//   det3d/ops/point_cloud/point_cloud_ops.h:12-71        points_to_voxel<float,3>
//   det3d/ops/box2d_iou_rotated/src/box_iou_rotated_utils.h:311-338
//                                                        single_box_iou_rotated<T>
// Output: oracle/_ref/d3d_ref*.so (git-ignored, travels with gpurun).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include "det3d/ops/point_cloud/point_cloud_ops.h"
#include "det3d/ops/box2d_iou_rotated/src/box_iou_rotated_utils.h"

namespace py = pybind11;

// boxes: [x_ctr, y_ctr, w, h, angle_degrees]
static double ref_box_iou_rotated_f64(py::array_t<double> a, py::array_t<double> b) {
  return detectron2::single_box_iou_rotated<double>(a.data(), b.data());
}
static float ref_box_iou_rotated_f32(py::array_t<float> a, py::array_t<float> b) {
  return detectron2::single_box_iou_rotated<float>(a.data(), b.data());
}

PYBIND11_MODULE(d3d_ref, m) {
  m.doc() = "compiled Det3D reference sources (oracle cross-check only)";
  m.def("points_to_voxel_3d_f32", &points_to_voxel<float, 3>, "points"_a, "voxels"_a, "coors"_a,
        "num_points_per_voxel"_a, "coor_to_voxelidx"_a, "voxel_size"_a, "coors_range"_a,
        "max_points"_a, "max_voxels"_a);
  m.def("box_iou_rotated_f64", &ref_box_iou_rotated_f64);
  m.def("box_iou_rotated_f32", &ref_box_iou_rotated_f32);
}
