"""iou3d on the MI355X behind the reference's Python API (det3d/ops/iou3d/iou3d_utils.py).

Boxes are BEV rectangles (x1, y1, x2, y2, ry) as CUDA tensors.  Same names, arguments and return
conventions as the reference module (`boxes_iou_bev`, `nms_gpu`, `nms_normal_gpu`), plus
`boxes_overlap_bev` (the extension entry point `iou3d_cuda.boxes_overlap_bev_gpu`, iou3d.cpp:30-48) and
`boxes_iou3d_gpu` (commented out in the reference, :24-55, restored here on top of the overlap kernel).
All work stays on the device; the only synchronisation is the 4-byte read of the kept count.
"""
import torch

from . import _lib


def _pairwise(boxes_a, boxes_b, mode, name):
    if not (boxes_a.is_cuda and boxes_b.is_cuda):
        raise _lib.D3dError("%s: boxes must be CUDA tensors; det3d_amd iou3d runs on the GPU only" % name)
    a = boxes_a.float().contiguous()
    b = boxes_b.float().contiguous()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 5 or b.shape[1] != 5:
        raise ValueError("%s: expected [M,5] and [N,5] boxes (x1, y1, x2, y2, ry)" % name)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L = _lib.lib()
    rc = L.d3d_iou3d_pairwise(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], mode, out.data_ptr(),
                              _lib.stream_ptr(a.device))
    _lib.check(rc, "d3d_iou3d_pairwise")
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(M,5), (N,5) -> (M,N) intersection areas."""
    return _pairwise(boxes_a, boxes_b, _lib.D3D_IOU3D_OVERLAP, "boxes_overlap_bev")


def boxes_iou_bev(boxes_a, boxes_b):
    """(M,5), (N,5) -> ans_iou (M,N)   (iou3d_utils.py:7-22)."""
    return _pairwise(boxes_a, boxes_b, _lib.D3D_IOU3D_IOU_BEV, "boxes_iou_bev")


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7), (M,7) camera-frame boxes [x, y, z, h, w, l, ry] -> 3D IoU (N,M): BEV overlap of the
    (x - w/2, z - l/2, x + w/2, z + l/2, ry) footprints times the overlap of [y - h, y]."""
    def bev(b):
        hw, hl = b[:, 4] / 2, b[:, 5] / 2
        return torch.stack([b[:, 0] - hw, b[:, 2] - hl, b[:, 0] + hw, b[:, 2] + hl, b[:, 6]], 1)

    overlaps_bev = boxes_overlap_bev(bev(boxes_a), bev(boxes_b))
    a_min, a_max = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_min, b_max = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)


def _nms(boxes, scores, thresh, normal, name):
    if not boxes.is_cuda:
        raise _lib.D3dError("%s: boxes must be CUDA tensors; det3d_amd iou3d runs on the GPU only" % name)
    order = scores.sort(0, descending=True)[1]
    n = boxes.shape[0]
    if n == 0:
        return order
    b = boxes[order].float().contiguous()
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.int32, device=b.device)
    n_keep = torch.empty((1,), dtype=torch.int32, device=b.device)
    wsb = L.d3d_iou3d_nms_workspace_bytes(n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=b.device)
    rc = L.d3d_iou3d_nms(b.data_ptr(), n, float(thresh), int(normal), keep.data_ptr(), n_keep.data_ptr(),
                         ws.data_ptr(), wsb, _lib.stream_ptr(b.device))
    _lib.check(rc, "d3d_iou3d_nms")
    return order[keep[: int(n_keep.item())].long()].contiguous()


def nms_gpu(boxes, scores, thresh):
    """boxes (N,5) [x1, y1, x2, y2, ry], scores (N) -> kept indices into `boxes`, by descending
    score (iou3d_utils.py:58-73); rotated IoU, strict `>`."""
    return _nms(boxes, scores, thresh, False, "nms_gpu")


def nms_normal_gpu(boxes, scores, thresh):
    """Same with the axis-aligned IoU that ignores ry (iou3d_utils.py:76-91)."""
    return _nms(boxes, scores, thresh, True, "nms_normal_gpu")
