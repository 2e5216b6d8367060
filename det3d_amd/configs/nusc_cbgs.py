# nuScenes CBGS (BASELINE.json configs[3]): VoxelFeatureExtractorV3 -> SpMiddleResNetFHD -> RPN(2 stages)
# -> MultiGroupHead with 6 task groups / 10 classes, 9-dim boxes with vector angle encoding, values of
# examples/cbgs/configs/nusc_all_vfev3_spmiddleresnetfhd_rpn2_mghead_syncbn.py (model side only).
import logging

from det3d.builder import build_box_coder
from det3d.utils.config_tool import get_downsample_factor
from det3d_amd.configs._anchors import expand, iou_assigner

norm_cfg = None
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
VOXEL = [0.1, 0.1, 0.2]

# class -> (w, l, h), anchor z, matched / unmatched IoU; grouped into the 6 CBGS heads
_CLS = dict(
    car=((1.97, 4.63, 1.74), -0.95, 0.6, 0.45), truck=((2.51, 6.93, 2.84), -0.40, 0.55, 0.4),
    construction_vehicle=((2.85, 6.37, 3.19), -0.225, 0.5, 0.35), bus=((2.94, 10.5, 3.47), -0.085, 0.55, 0.4),
    trailer=((2.90, 12.29, 3.87), 0.115, 0.5, 0.35), barrier=((2.53, 0.50, 0.98), -1.33, 0.55, 0.4),
    motorcycle=((0.77, 2.11, 1.47), -1.085, 0.5, 0.3), bicycle=((0.60, 1.70, 1.28), -1.18, 0.5, 0.35),
    pedestrian=((0.67, 0.73, 1.77), -0.935, 0.6, 0.4), traffic_cone=((0.41, 0.41, 1.07), -1.285, 0.6, 0.4))
_GROUPS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"],
           ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"]]
tasks, class_names, _gens = expand(_GROUPS, _CLS, 51.2, velocities=[0, 0])

box_coder = dict(type="ground_box3d_coder", n_dim=9, linear_dim=False, encode_angle_vector=True)
target_assigner = iou_assigner(tasks, _gens)

model = dict(
    type="VoxelNet", pretrained=None,
    reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5, norm_cfg=norm_cfg),
    backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8, norm_cfg=norm_cfg),
    neck=dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
              us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256,
              norm_cfg=norm_cfg, logger=logging.getLogger("RPN")),
    bbox_head=dict(
        type="MultiGroupHead", mode="3d", in_channels=512, norm_cfg=norm_cfg, tasks=tasks,
        weights=[1], box_coder=build_box_coder(box_coder), encode_background_as_zeros=True,
        use_sigmoid_score=True, encode_rad_error_by_sin=False, loss_aux=None,
        loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=2.0),
        loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
        loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, codewise=True, loss_weight=0.25,
                       code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0])))

assigner = dict(box_coder=box_coder, target_assigner=target_assigner,
                out_size_factor=get_downsample_factor(model), debug=False)
train_cfg = dict(assigner=assigner)
test_cfg = dict(
    nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
             nms_post_max_size=83, nms_iou_threshold=0.2),
    score_threshold=0.1, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500)

voxel_generator = dict(range=PC_RANGE, voxel_size=VOXEL, max_points_in_voxel=10, max_voxel_num=60000)
data = dict(samples_per_gpu=16, workers_per_gpu=8)
optimizer = dict(TYPE="adam", VALUE=dict(amsgrad=0.0, wd=0.01), FIXED_WD=True, MOVING_AVERAGE=False)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(type="one_cycle", lr_max=0.001, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)
