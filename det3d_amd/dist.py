"""Frame-level data parallelism: one process per GPU, full replica per rank, one gradient
exchange per iteration.  Replaces the reference's DistributedDataParallel wrap
(det3d/torchie/apis/train.py:282-289) PLUS its redundant second flat all-reduce
(det3d/core/utils/dist_utils.py:8-57) with a single flat-bucket all-reduce: the whole
gradient is 7-30 MB, so over xGMI the step is latency-bound and one RCCL call on one
contiguous buffer beats DDP's 25 MB bucketing + the extra pass.
backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_dist(backend=None):
    """env:// rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), as
    tools/train.py:73-110 does.  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # D3D_DIST_BACKEND=gloo lets several ranks share one GPU (flow test without RCCL)
            backend = os.environ.get("D3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_params(module, src=0):
    """same starting point on every rank (DDP does this at wrap time)."""
    if world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    # every rank must issue the broadcasts in the SAME order: iterate dtypes sorted by name (a set
    # of torch.dtype iterates in id() order, which differs between processes -> collective mismatch)
    for dtype in sorted({t.dtype for t in tensors}, key=str):
        group = [t for t in tensors if t.dtype == dtype]
        flat = torch._utils._flatten_dense_tensors(group)
        dist.broadcast(flat, src)
        for t, f in zip(group, torch._utils._unflatten_dense_tensors(flat, group)):
            t.copy_(f)


class FlatGradBucket:
    """Gradient exchange + clipping on ONE contiguous buffer.  After backward the per-parameter
    grads are concatenated once (1 launch), all-reduced once (RCCL), the norm / clip run on the
    flat buffer, and every p.grad is re-pointed at its slice -- so the optimizer sees ordinary
    .grad tensors and no per-parameter kernels are launched (the first version pre-bound
    p.grad to views, which made autograd issue one accumulate-add per parameter: 84 launches)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert len({p.dtype for p in self.params}) == 1, "single dtype bucket"
        self.flat = None

    def zero(self):
        for p in self.params:
            p.grad = None

    def gather(self):
        grads = []
        for p in self.params:
            g = p.grad
            if g is None:
                g = torch.zeros_like(p)
            grads.append(g.reshape(-1))
        self.flat = torch.cat(grads)
        off = 0
        for p in self.params:
            p.grad = self.flat[off: off + p.numel()].view(p.shape)
            off += p.numel()
        return self.flat

    def allreduce_mean(self):
        """mean over ranks (= allreduce_grads + div_(world_size), dist_utils.py:23-28)"""
        if self.flat is None:
            self.gather()
        w = world_size()
        if w == 1:
            return
        dist.all_reduce(self.flat)
        self.flat.div_(w)

    def clip_norm_(self, max_norm):
        if self.flat is None:
            self.gather()
        total = torch.linalg.vector_norm(self.flat)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


def all_gather_objects(obj):
    """validation-time gather of per-rank detections (torchie/trainer/utils.py:99-147)."""
    w = world_size()
    if w == 1:
        return [obj]
    out = [None] * w
    dist.all_gather_object(out, obj)
    return out


def barrier():
    if world_size() > 1:
        dist.barrier()
