# KITTI-car SECOND (BASELINE.json configs[1]): VoxelFeatureExtractorV3 -> SpMiddleFHD -> RPN ->
# MultiGroupHead.  Written against the reference's config API (registry `type=` dicts,
# build_box_coder / get_downsample_factor called at import) with the values of
# examples/second/configs/kitti_car_vfev3_spmiddlefhd_rpn1_mghead_syncbn.py; only the parts the
# hot path reads are kept (no dataset / db-sampler / augmentation blocks -- those stay with the
# reference's data pipeline).  The unmodified reference file loads through Config.fromfile too.
import logging

from det3d.builder import build_box_coder
from det3d.utils.config_tool import get_downsample_factor

norm_cfg = None  # BN / BN1d defaults, exactly like the shipped "_syncbn" config
CLASSES = ["Car"]
tasks = [dict(num_class=len(CLASSES), class_names=CLASSES)]
PC_RANGE = [0, -40.0, -3.0, 70.4, 40.0, 1.0]

box_coder = dict(type="ground_box3d_coder", n_dim=7, linear_dim=False, encode_angle_vector=False)

target_assigner = dict(
    type="iou", tasks=tasks, sample_positive_fraction=-1, sample_size=512, pos_area_threshold=-1,
    region_similarity_calculator=dict(type="nearest_iou_similarity"),
    anchor_generators=[dict(type="anchor_generator_range", class_name="Car", sizes=[1.6, 3.9, 1.56],
                            anchor_ranges=[0, -40.0, -1.0, 70.4, 40.0, -1.0], rotations=[0, 1.57],
                            matched_threshold=0.6, unmatched_threshold=0.45)])

RPN_WIDTH = 128
model = dict(
    type="VoxelNet", pretrained=None,
    reader=dict(type="VoxelFeatureExtractorV3", num_input_features=4, norm_cfg=norm_cfg),
    backbone=dict(type="SpMiddleFHD", num_input_features=4, ds_factor=8, norm_cfg=norm_cfg),
    neck=dict(type="RPN", layer_nums=[5], ds_layer_strides=[1], ds_num_filters=[RPN_WIDTH],
              us_layer_strides=[1], us_num_filters=[RPN_WIDTH], num_input_features=128,
              norm_cfg=norm_cfg, logger=logging.getLogger("RPN")),
    bbox_head=dict(
        type="MultiGroupHead", mode="3d", in_channels=RPN_WIDTH, norm_cfg=norm_cfg, tasks=tasks,
        weights=[1], box_coder=build_box_coder(box_coder), encode_background_as_zeros=True,
        use_sigmoid_score=True, encode_rad_error_by_sin=True, direction_offset=0.0,
        loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=1.0),
        loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
        loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, codewise=True, loss_weight=2.0,
                       code_weights=[1.0] * 7),
        loss_aux=dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier",
                      loss_weight=0.2)))

assigner = dict(box_coder=box_coder, target_assigner=target_assigner,
                out_size_factor=get_downsample_factor(model), debug=False)
train_cfg = dict(assigner=assigner)
test_cfg = dict(
    nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
             nms_post_max_size=100, nms_iou_threshold=0.01),
    score_threshold=0.3, post_center_limit_range=[0, -40.0, -5.0, 70.4, 40.0, 5.0], max_per_img=100)

voxel_generator = dict(range=PC_RANGE, voxel_size=[0.05, 0.05, 0.1], max_points_in_voxel=5,
                       max_voxel_num=20000)
data = dict(samples_per_gpu=4, workers_per_gpu=2)

optimizer = dict(TYPE="adam", VALUE=dict(amsgrad=0.0, wd=0.01), FIXED_WD=True, MOVING_AVERAGE=False)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)
total_epochs = 60
dist_params = dict(backend="nccl", init_method="env://")
