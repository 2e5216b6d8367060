# KITTI-car PointPillars (BASELINE.json configs[0], the CPU-runnable plumbing case):
# PillarFeatureNet -> PointPillarsScatter -> RPN(3 blocks) -> MultiGroupHead, values of
# examples/point_pillars/configs/kitti_point_pillars_mghead_syncbn.py, model-side parts only.
import logging

from det3d.builder import build_box_coder
from det3d.utils.config_tool import get_downsample_factor

norm_cfg = None
tasks = [dict(num_class=1, class_names=["Car"])]
PC_RANGE = [0, -39.68, -3, 69.12, 39.68, 1]
VOXEL = [0.16, 0.16, 4]
box_coder = dict(type="ground_box3d_coder", n_dim=7, linear_dim=False, encode_angle_vector=False)

model = dict(
    type="PointPillars", pretrained=None,
    reader=dict(type="PillarFeatureNet", num_filters=[64], with_distance=False, voxel_size=VOXEL,
                pc_range=PC_RANGE, norm_cfg=norm_cfg),
    backbone=dict(type="PointPillarsScatter", ds_factor=1, norm_cfg=norm_cfg),
    neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[64, 128, 256],
              us_layer_strides=[1, 2, 4], us_num_filters=[128, 128, 128], num_input_features=64,
              norm_cfg=norm_cfg, logger=logging.getLogger("RPN")),
    bbox_head=dict(
        type="MultiGroupHead", mode="3d", in_channels=384, norm_cfg=norm_cfg, tasks=tasks, weights=[1],
        box_coder=build_box_coder(box_coder), encode_background_as_zeros=True, use_sigmoid_score=True,
        encode_rad_error_by_sin=True, direction_offset=0.0,
        loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=1.0),
        loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
        loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, codewise=True, loss_weight=2.0,
                       code_weights=[1.0] * 7),
        loss_aux=dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier",
                      loss_weight=0.2)))

assigner = dict(box_coder=box_coder, out_size_factor=get_downsample_factor(model), debug=False,
                target_assigner=dict(type="iou", tasks=tasks, anchor_generators=[dict(
                    type="anchor_generator_range", class_name="Car", sizes=[1.6, 3.9, 1.56],
                    anchor_ranges=[0, -39.68, -1.0, 69.12, 39.68, -1.0], rotations=[0, 1.57],
                    matched_threshold=0.6, unmatched_threshold=0.45)]))
train_cfg = dict(assigner=assigner)
test_cfg = dict(nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                         nms_post_max_size=300, nms_iou_threshold=0.5),
                score_threshold=0.05, post_center_limit_range=[0, -40.0, -5.0, 70.4, 40.0, 5.0],
                max_per_img=100)
voxel_generator = dict(range=PC_RANGE, voxel_size=VOXEL, max_points_in_voxel=100, max_voxel_num=12000)
data = dict(samples_per_gpu=2, workers_per_gpu=2)
optimizer = dict(TYPE="adam", VALUE=dict(amsgrad=0.0, wd=0.01), FIXED_WD=True, MOVING_AVERAGE=False)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)
