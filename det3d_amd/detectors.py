"""Detectors: det3d/models/detectors/{single_stage.py:8-74, voxelnet.py:6-52,
point_pillars.py:6-54}.  Same `forward(example, return_loss)` contract and `example` keys
(SURVEY.md 8b, B3b).  One MI355X addition: if `example` carries `voxel_mean` (the fused
VFE output of d3d_voxelize) VoxelNet feeds those [M, C] rows to the reader instead of the
padded [M, T, C] voxels, which then never have to be materialised.
"""
from torch import nn

from . import backbones, bbox_heads, losses, necks, readers  # noqa: F401  (fill the registries)
from . import builder, graph_tail
from .registry import DETECTORS


class SingleStageDetector(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.bbox_head = builder.build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    def _train_tail(self, example, bev):
        """training: the static-shape tail (neck, head, loss) replays as two hipGraphs (graph_tail.py); eagerly
        when it cannot be captured"""
        losses = graph_tail.run_tail(self, example, bev)
        if losses is not None:
            return losses
        return self._finish(example, self.neck(bev) if self.with_neck else bev, True)

    def _finish(self, example, x, return_loss):
        preds = self.bbox_head(x)
        if return_loss:
            return self.bbox_head.loss(example, preds)
        return self.bbox_head.predict(example, preds, self.test_cfg)


@DETECTORS.register_module
class VoxelNet(SingleStageDetector):
    def _backbone(self, data):
        input_features = self.reader(data["features"], data["num_voxels"])
        if data.get("plan") is not None:  # rulebooks built ahead of time (prefetch.GeometryPrefetcher)
            return self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"],
                                 plan=data["plan"])
        return self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"])

    def extract_feat(self, data):
        x = self._backbone(data)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward(self, example, return_loss=True, **kwargs):
        feats = example["voxel_mean"] if "voxel_mean" in example else example["voxels"]
        data = dict(features=feats, num_voxels=example["num_points"], coors=example["coordinates"],
                    batch_size=len(example["num_voxels"]), input_shape=example["shape"][0],
                    plan=example.get("sparse_plan"))
        if return_loss and self.training and graph_tail.GRAPH_TAIL:
            return self._train_tail(example, self._backbone(data))
        return self._finish(example, self.extract_feat(data), return_loss)


@DETECTORS.register_module
class PointPillars(SingleStageDetector):
    def extract_feat(self, data):
        input_features = self.reader(data["features"], data["num_voxels"], data["coors"])
        x = self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"])
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward(self, example, return_loss=True, **kwargs):
        data = dict(features=example["voxels"], num_voxels=example["num_points"],
                    coors=example["coordinates"], batch_size=len(example["num_voxels"]),
                    input_shape=example["shape"][0])
        if return_loss and self.training and graph_tail.GRAPH_TAIL:
            bev = self.backbone(self.reader(data["features"], data["num_voxels"], data["coors"]), data["coors"],
                                data["batch_size"], data["input_shape"])
            return self._train_tail(example, bev)
        return self._finish(example, self.extract_feat(data), return_loss)
