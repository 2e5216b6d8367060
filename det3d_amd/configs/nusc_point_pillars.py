# nuScenes PointPillars (BASELINE.json configs[2]): PillarFeatureNet(64) -> PointPillarsScatter -> RPN(3 stages)
# -> MultiGroupHead, 6 task groups / 10 classes, 9-dim boxes; values of
# examples/point_pillars/configs/nusc_all_point_pillars_mghead_syncbn.py (model side only).
import logging

from det3d.builder import build_box_coder
from det3d.utils.config_tool import get_downsample_factor
from det3d_amd.configs._anchors import expand, iou_assigner
from det3d_amd.configs.nusc_cbgs import _CLS, _GROUPS, PC_RANGE, test_cfg  # same classes, anchors and NMS

norm_cfg = None
VOXEL = [0.2, 0.2, 8]
tasks, class_names, _gens = expand(_GROUPS, _CLS, 51.2, velocities=[0, 0])
box_coder = dict(type="ground_box3d_coder", n_dim=9, linear_dim=False, encode_angle_vector=True)
target_assigner = iou_assigner(tasks, _gens)
_UP = [128, 128, 128]

model = dict(
    type="PointPillars", pretrained=None,
    reader=dict(type="PillarFeatureNet", num_filters=[64], num_input_features=5, with_distance=False,
                voxel_size=tuple(VOXEL), pc_range=tuple(PC_RANGE), norm_cfg=norm_cfg),
    backbone=dict(type="PointPillarsScatter", ds_factor=1, norm_cfg=norm_cfg),
    neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[64, 128, 256],
              us_layer_strides=[0.5, 1, 2], us_num_filters=_UP, num_input_features=64, norm_cfg=norm_cfg,
              logger=logging.getLogger("RPN")),
    bbox_head=dict(
        type="MultiGroupHead", mode="3d", in_channels=sum(_UP), norm_cfg=norm_cfg, tasks=tasks, weights=[1],
        box_coder=build_box_coder(box_coder), encode_background_as_zeros=True, use_sigmoid_score=True,
        encode_rad_error_by_sin=False, loss_aux=None, direction_offset=0,
        loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=2.0),
        loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
        loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, codewise=True, loss_weight=0.25,
                       code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0])))

assigner = dict(box_coder=box_coder, target_assigner=target_assigner,
                out_size_factor=get_downsample_factor(model), debug=False)
train_cfg = dict(assigner=assigner)
voxel_generator = dict(range=PC_RANGE, voxel_size=VOXEL, max_points_in_voxel=20, max_voxel_num=30000)
data = dict(samples_per_gpu=4, workers_per_gpu=8)
optimizer = dict(TYPE="adam", VALUE=dict(amsgrad=0.0, wd=0.01), FIXED_WD=True, MOVING_AVERAGE=False)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(type="one_cycle", lr_max=0.002, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)
