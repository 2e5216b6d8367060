"""Make the UNMODIFIED reference configs and scn.py-style code importable:

    from det3d.builder import build_box_coder            (every config, line 4)
    from det3d.utils.config_tool import get_downsample_factor
    from det3d.torchie import Config
    import spconv; from spconv import SparseConv3d, SubMConv3d

`install()` registers thin alias modules in sys.modules -- unless a real `det3d` /
`spconv` is importable, in which case nothing is touched.
"""
import importlib.util
import sys
import types

_DONE = False


def _alias(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__d3d_amd_alias__ = True
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def _real_module_available(name):
    m = sys.modules.get(name)
    if m is not None:
        return not getattr(m, "__d3d_amd_alias__", False)
    try:
        return importlib.util.find_spec(name) is not None
    except (ValueError, ImportError):
        return False


def install(force=False):
    global _DONE
    if _DONE and not force:
        return
    from . import box_ops, builder, checkpoint, collate, config, iou3d_utils, point_cloud_ops, registry
    from . import spconv as sp

    if force or not _real_module_available("det3d"):
        _alias("det3d").__path__ = []
        _alias("det3d.builder", build_box_coder=builder.build_box_coder)
        _alias("det3d.utils").__path__ = []
        _alias("det3d.utils.config_tool", get_downsample_factor=builder.get_downsample_factor)
        _alias("det3d.utils.registry", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
        _alias("det3d.torchie", Config=config.Config, ConfigDict=config.ConfigDict).__path__ = []
        _alias("det3d.torchie.trainer", load_checkpoint=checkpoint.load_checkpoint,
               save_checkpoint=checkpoint.save_checkpoint, load_state_dict=checkpoint.load_state_dict)
        _alias("det3d.torchie.parallel", collate_kitti=collate.collate_kitti)
        _alias("det3d.models", build_detector=builder.build_detector, build_reader=builder.build_reader,
               build_backbone=builder.build_backbone, build_neck=builder.build_neck,
               build_head=builder.build_head, build_loss=builder.build_loss,
               READERS=registry.READERS, BACKBONES=registry.BACKBONES, NECKS=registry.NECKS,
               HEADS=registry.HEADS, LOSSES=registry.LOSSES, DETECTORS=registry.DETECTORS).__path__ = []
        _alias("det3d.models.registry", READERS=registry.READERS, BACKBONES=registry.BACKBONES,
               NECKS=registry.NECKS, HEADS=registry.HEADS, LOSSES=registry.LOSSES,
               DETECTORS=registry.DETECTORS)
        _alias("det3d.core").__path__ = []
        _alias("det3d.core.bbox").__path__ = []
        _alias("det3d.core.bbox.box_torch_ops", **{k: getattr(box_ops, k) for k in (
            "second_box_encode", "second_box_decode", "rotate_nms", "nms", "center_to_corner_box2d",
            "corner_to_standup_nd")})
        sys.modules["det3d.core"].box_torch_ops = sys.modules["det3d.core.bbox.box_torch_ops"]
        _alias("det3d.core.input").__path__ = []
        _alias("det3d.core.input.voxel_generator", VoxelGenerator=point_cloud_ops.VoxelGenerator)
        _alias("det3d.ops").__path__ = []
        _alias("det3d.ops.point_cloud").__path__ = []
        _alias("det3d.ops.point_cloud.point_cloud_ops", points_to_voxel=point_cloud_ops.points_to_voxel)
        _alias("det3d.ops.iou3d").__path__ = []
        _alias("det3d.ops.iou3d.iou3d_utils", **{k: getattr(iou3d_utils, k) for k in (
            "boxes_iou_bev", "boxes_overlap_bev", "boxes_iou3d_gpu", "nms_gpu", "nms_normal_gpu")})
        from . import gt_sampler, loading, pipelines

        _alias("det3d.datasets").__path__ = []
        _alias("det3d.datasets.registry", PIPELINES=registry.PIPELINES)
        _alias("det3d.datasets.pipelines", Compose=pipelines.Compose, Preprocess=pipelines.Preprocess,
               Voxelization=pipelines.Voxelization, AssignTarget=pipelines.AssignTarget, Reformat=pipelines.Reformat,
               LoadPointCloudFromFile=loading.LoadPointCloudFromFile,
               LoadPointCloudAnnotations=pipelines.LoadPointCloudAnnotations).__path__ = []
        _alias("det3d.core.sampler").__path__ = []
        _alias("det3d.core.sampler.sample_ops", DataBaseSamplerV2=gt_sampler.DeviceGTSampler)
        _alias("det3d.datasets.utils").__path__ = []
        _alias("det3d.datasets.utils.kitti_object_eval_python").__path__ = []
        _alias("det3d.datasets.utils.kitti_object_eval_python.rotate_iou",
               rotate_iou_gpu_eval=box_ops.rotate_iou_gpu_eval)
    if force or not _real_module_available("spconv"):
        _alias("spconv", SparseConvTensor=sp.SparseConvTensor, SparseSequential=sp.SparseSequential,
               SparseModule=sp.SparseModule, SubMConv3d=sp.SubMConv3d, SparseConv3d=sp.SparseConv3d,
               ToDense=sp.ToDense)
    _DONE = True
