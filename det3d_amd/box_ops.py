"""Host side of decode + NMS.  Mirrors det3d/core/bbox/box_torch_ops.py:
  second_box_encode/second_box_decode (:27-148), rotate_nms (:528-549), nms (:506-525),
  center_to_corner_box2d / corner_to_standup_nd -- same signatures and return types.
NMS runs on the device through the C ABI (d3d_rotate_nms / d3d_nms): the reference's
.cpu().numpy() round trip + boost polygon loop is gone, the only host sync left is
reading the kept count to size the returned LongTensor.
"""
import numpy as np
import torch

from . import _lib


def second_box_encode(boxes, anchors, encode_angle_to_vector=False, smooth_dim=False, norm_velo=False):
    """box_torch_ops.py:27-77"""
    box_ndim = anchors.shape[-1]
    if box_ndim == 7:
        xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
        xg, yg, zg, wg, lg, hg, rg = torch.split(boxes, 1, dim=-1)
    else:
        xa, ya, za, wa, la, ha, vxa, vya, ra = torch.split(anchors, 1, dim=-1)
        xg, yg, zg, wg, lg, hg, vxg, vyg, rg = torch.split(boxes, 1, dim=-1)
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xt = (xg - xa) / diagonal
    yt = (yg - ya) / diagonal
    zt = (zg - za) / ha
    if smooth_dim:
        lt, wt, ht = lg / la - 1, wg / wa - 1, hg / ha - 1
    else:
        lt, wt, ht = torch.log(lg / la), torch.log(wg / wa), torch.log(hg / ha)
    ret = [xt, yt, zt, wt, lt, ht]
    if box_ndim > 7:
        if norm_velo:
            ret.extend([(vxg - vxa) / diagonal, (vyg - vya) / diagonal])
        else:
            ret.extend([vxg - vxa, vyg - vya])
    if encode_angle_to_vector:
        ret.extend([torch.cos(rg) - torch.cos(ra), torch.sin(rg) - torch.sin(ra)])
    else:
        ret.append(rg - ra)
    return torch.cat(ret, dim=-1)


def second_box_decode(box_encodings, anchors, encode_angle_to_vector=False, bin_loss=False,
                      smooth_dim=False, norm_velo=False):
    """box_torch_ops.py:80-148 -- anchors (+) deltas -> boxes [.., x y z w l h (vx vy) r]."""
    box_ndim = anchors.shape[-1]
    if box_ndim == 9:
        xa, ya, za, wa, la, ha, vxa, vya, ra = torch.split(anchors, 1, dim=-1)
        if encode_angle_to_vector:
            xt, yt, zt, wt, lt, ht, vxt, vyt, rtx, rty = torch.split(box_encodings, 1, dim=-1)
        else:
            xt, yt, zt, wt, lt, ht, vxt, vyt, rt = torch.split(box_encodings, 1, dim=-1)
    elif box_ndim == 7:
        xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
        if encode_angle_to_vector:
            xt, yt, zt, wt, lt, ht, rtx, rty = torch.split(box_encodings, 1, dim=-1)
        else:
            xt, yt, zt, wt, lt, ht, rt = torch.split(box_encodings, 1, dim=-1)
    else:
        raise ValueError("anchors must have 7 or 9 columns, got %d" % box_ndim)
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xg = xt * diagonal + xa
    yg = yt * diagonal + ya
    zg = zt * ha + za
    ret = [xg, yg, zg]
    if smooth_dim:
        lg, wg, hg = (lt + 1) * la, (wt + 1) * wa, (ht + 1) * ha
    else:
        lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
    ret.extend([wg, lg, hg])
    if encode_angle_to_vector:
        rg = torch.atan2(rty + torch.sin(ra), rtx + torch.cos(ra))
    else:
        rg = rt + ra
    if box_ndim > 7:
        if norm_velo:
            ret.extend([vxt * diagonal + vxa, vyt * diagonal + vya])
        else:
            ret.extend([vxt + vxa, vyt + vya])
    ret.append(rg)
    return torch.cat(ret, dim=-1)


def center_to_corner_box2d(centers, dims, angles=None, origin=0.5):
    """box_torch_ops.py / box_np_ops.py:477-497 (clockwise for positive angle)."""
    nrm = torch.tensor([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=dims.dtype, device=dims.device) - origin
    corners = dims.view(-1, 1, 2) * nrm.view(1, 4, 2)
    if angles is not None:
        s, c = torch.sin(angles), torch.cos(angles)
        x, y = corners[..., 0], corners[..., 1]
        corners = torch.stack([x * c[:, None] + y * s[:, None], -x * s[:, None] + y * c[:, None]], -1)
    return corners + centers.view(-1, 1, 2)


def corner_to_standup_nd(boxes_corner):
    return torch.cat([boxes_corner.min(dim=1)[0], boxes_corner.max(dim=1)[0]], dim=-1)


def _device_nms(fn_name, boxes, thresh, max_keep):
    """boxes: CUDA float32 [n,k] sorted by descending score.  Returns int32 CUDA tensor of
    kept positions (length = kept count; one 4-byte D2H to learn it)."""
    L = _lib.lib()
    n = boxes.shape[0]
    dev = boxes.device
    keep = torch.empty((max(max_keep, 1),), dtype=torch.int32, device=dev)
    n_keep = torch.empty((1,), dtype=torch.int32, device=dev)
    wsb = L.d3d_rotate_nms_workspace_bytes(n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    rc = getattr(L, fn_name)(boxes.data_ptr(), n, float(thresh), int(max_keep), keep.data_ptr(),
                             n_keep.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, fn_name)
    return keep[: int(n_keep.item())]


def _nms_entry(fn_name, bboxes, scores, pre_max_size, post_max_size, iou_threshold):
    if not bboxes.is_cuda:
        raise _lib.D3dError("%s: boxes are on %s; det3d_amd NMS runs on the GPU only" % (fn_name, bboxes.device))
    n = scores.shape[0]
    if n == 0:
        return torch.zeros([0]).long().to(bboxes.device)
    k = min(n, pre_max_size) if pre_max_size is not None else n
    # the reference sorts twice (topk, then argsort()[::-1] in rotate_nms_cc); one sorted
    # topk is the same order for distinct scores
    scores, indices = torch.topk(scores, k=k)
    boxes = bboxes[indices].float().contiguous()
    post = post_max_size if post_max_size is not None else k
    keep = _device_nms(fn_name, boxes, iou_threshold, min(post, k))
    if keep.shape[0] == 0:
        return torch.zeros([0]).long().to(bboxes.device)
    return indices[keep.long()]


def rotate_nms(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """box_torch_ops.py:528-549.  rbboxes [N,5] (x,y,w,l,r) -> LongTensor of kept indices
    (into the input, in descending score order) on the input's device."""
    return _nms_entry("d3d_rotate_nms", rbboxes, scores, pre_max_size, post_max_size, iou_threshold)


def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """box_torch_ops.py:506-525.  bboxes [N,4] (x1,y1,x2,y2), +1 pixel IoU, strict `>`."""
    return _nms_entry("d3d_nms", bboxes, scores, pre_max_size, post_max_size, iou_threshold)


def decode_filter_topk(box_preds, cls_preds, dir_preds, anchors, encode_angle_vector, score_threshold, pre_max_size):
    """d3d_decode_filter_topk: the front half of MultiGroupHead.get_task_detections for ALL frames of one task group
    (mg_head.py:984-1008 + second_box_decode) in two launches.  box_preds [B, A, code], cls_preds [B, A, nc],
    dir_preds [B, A, 2] or None (float32 / bfloat16), anchors [A, nd] or [B, A, nd] -> (boxes [B, K, nd] f32 decoded,
    scores [B, K] f32 descending, labels [B, K] i32, dir_labels [B, K] i32 or None, count [B] i32), K = pre_max_size;
    rows past count[b] are undefined."""
    if not box_preds.is_cuda:
        raise _lib.D3dError("decode_filter_topk runs on the GPU only")
    L = _lib.lib()
    dt = box_preds.dtype
    if dt not in (torch.float32, torch.bfloat16):
        box_preds, dt = box_preds.float(), torch.float32
    box_preds = box_preds.contiguous()
    cls_preds = cls_preds.to(dt).contiguous()
    dir_preds = dir_preds.to(dt).contiguous() if dir_preds is not None else None
    anchors = anchors.float().contiguous()
    b, a, code = box_preds.shape
    nc = cls_preds.shape[-1]
    nd = anchors.shape[-1]
    k = int(pre_max_size)
    dev = box_preds.device
    boxes = torch.empty((b, k, nd), dtype=torch.float32, device=dev)
    scores = torch.empty((b, k), dtype=torch.float32, device=dev)
    labels = torch.empty((b, k), dtype=torch.int32, device=dev)
    dirs = torch.empty((b, k), dtype=torch.int32, device=dev) if dir_preds is not None else None
    count = torch.empty((b,), dtype=torch.int32, device=dev)
    wsb = L.d3d_decode_filter_topk_workspace_bytes(b, a)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    rc = L.d3d_decode_filter_topk(box_preds.data_ptr(), cls_preds.data_ptr(), _lib.ptr(dir_preds),
                                  _lib.D3D_BF16 if dt == torch.bfloat16 else _lib.D3D_F32, anchors.data_ptr(),
                                  int(anchors.dim() == 3), b, a, code, nd, nc, int(bool(encode_angle_vector)),
                                  float(score_threshold), k, boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                  _lib.ptr(dirs), count.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_decode_filter_topk")
    return boxes, scores, labels, dirs, count


def rotate_nms_batched(boxes, counts, iou_threshold, post_max_size):
    """d3d_rotate_nms_batched: boxes [S, K, nd] f32 (score-sorted lists, BEV box = columns 0, 1, 3, 4, -1), counts [S]
    i32 -> keep [S, post_max_size] i32 (indices into each list, -1 padded), n_keep [S] i32; three launches for all
    lists, no host sync."""
    if not boxes.is_cuda:
        raise _lib.D3dError("rotate_nms_batched runs on the GPU only")
    L = _lib.lib()
    boxes = boxes.float().contiguous()
    counts = counts.int().contiguous()
    s, k, nd = boxes.shape
    dev = boxes.device
    keep = torch.empty((s, int(post_max_size)), dtype=torch.int32, device=dev)
    nkeep = torch.empty((s,), dtype=torch.int32, device=dev)
    wsb = L.d3d_rotate_nms_batched_workspace_bytes(s, k)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    rc = L.d3d_rotate_nms_batched(boxes.data_ptr(), nd, counts.data_ptr(), s, k, float(iou_threshold),
                                  int(post_max_size), keep.data_ptr(), nkeep.data_ptr(), ws.data_ptr(), wsb,
                                  _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_rotate_nms_batched")
    return keep, nkeep


def rotate_iou(boxes_a, boxes_b):
    """pairwise rotated BEV IoU [M,N] of (x,y,w,l,r) boxes (d3d_rotate_iou)."""
    if not boxes_a.is_cuda:
        raise _lib.D3dError("rotate_iou runs on the GPU only")
    L = _lib.lib()
    a = boxes_a.float().contiguous()
    b = boxes_b.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    rc = L.d3d_rotate_iou(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(),
                          _lib.stream_ptr(a.device))
    _lib.check(rc, "d3d_rotate_iou")
    return out


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """Evaluation-side rotated IoU, signature of kitti_object_eval_python/rotate_iou.py:264-341:
    boxes [N,5], query_boxes [K,5] as (cx, cy, dx, dy, r) -> [N,K]; criterion -1: IoU, 0: / area(query),
    1: / area(box), 2: raw intersection area.  numpy in -> numpy out (same dtype) like the reference;
    CUDA tensors in -> CUDA float32 tensor out (no host round trip)."""
    import numpy as np

    as_numpy = not torch.is_tensor(boxes)
    dev = torch.device("cuda", device_id) if as_numpy else boxes.device
    if dev.type != "cuda":
        raise _lib.D3dError("rotate_iou_gpu_eval runs on the GPU only")
    b = torch.as_tensor(np.asarray(boxes, dtype=np.float32) if as_numpy else boxes).to(dev).float().contiguous()
    q = torch.as_tensor(np.asarray(query_boxes, dtype=np.float32) if as_numpy else query_boxes).to(dev).float().contiguous()
    out = torch.zeros((b.shape[0], q.shape[0]), dtype=torch.float32, device=dev)
    if out.numel():
        rc = _lib.lib().d3d_rotate_iou_eval(b.data_ptr(), b.shape[0], q.data_ptr(), q.shape[0], int(criterion),
                                            out.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(rc, "d3d_rotate_iou_eval")
    return out.cpu().numpy().astype(np.asarray(boxes).dtype) if as_numpy else out
