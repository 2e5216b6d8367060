"""The static-shape tail of the training step -- RPN neck, multi-group head, loss -- as two hipGraphs.

From the dense BEV map onwards every tensor of the step has the same shape in every iteration (the map is
[B, C*D, H, W]; anchors, labels and regression targets are per-anchor arrays), while the sparse stack in
front of it is data dependent (voxel counts, rulebook sizes).  Eagerly the tail is ~170 of the step's ~350 kernel
launches, each costing the host 15-20 us of dispatch; with the forward and the backward of the tail captured once
(`torch.cuda.make_graphed_callables`: one hipGraph each, replayed with two launches per step) the host issues
only the sparse stack and the optimiser, and stays ahead of the GPU.

What is captured: `neck(bev)` (det3d/models/necks/rpn.py:102-117), `bbox_head(x)` (mg_head.py:553-560) and the
fused loss of every task group (mg_head.py:562-668, here d3d_head_loss), plus their backward.  The parameters'
gradients come out of the backward graph as ordinary autograd outputs, so gradient hooks (overlapped all-reduce
buckets) fire as in the eager step.  Not captured: anything with a collective inside (DistributedSyncBN with
world_size > 1) or a task group the fused loss does not cover -- `GraphedTail.build` returns None and the
detector runs the tail eagerly.

OFF by default (D3D_GRAPH_TAIL=1 turns it on).  Measured on MI355X with the geometry prefetch in place
(profiles/round2_bench_graph_ab.txt): the eager tail runs the KITTI step at 668 frames/s on that box, the
graphed tail -- which has to be captured with find mode OFF, see build() -- at 643, and its private
pool kept growing (24 GB reserved after 400 steps).  The graphs pay off only while the host is the bottleneck (662 ->
697 frames/s before the step became GPU-bound; that 697 was measured with find-mode solvers inside the graph and is
INVALID: their gradients were inf).  With more than one rank it stays off unless D3D_GRAPH_TAIL_DIST=1: capturing while a
process group with a watchdog thread is alive, and gradient hooks that then run on the capture stream, have not been
exercised with RCCL across GPUs, and a failed capture is not something to discover in a multi-GPU job.
"""
import os

import torch
from torch import nn

GRAPH_TAIL = os.environ.get("D3D_GRAPH_TAIL", "0") == "1"
GRAPH_TAIL_DIST = os.environ.get("D3D_GRAPH_TAIL_DIST", "0") == "1"


class _Tail(nn.Module):
    """neck -> head -> per-task loss vectors; tensor-only signature (what graph capture needs)."""

    def __init__(self, neck, bbox_head, with_anchors):
        super().__init__()
        self.neck = neck
        self.bbox_head = bbox_head
        self.with_anchors = with_anchors

    def forward(self, bev, *flat):
        x = self.neck(bev) if self.neck is not None else bev
        preds = self.bbox_head(x)
        n = len(preds)
        example = {"labels": list(flat[:n]), "reg_targets": list(flat[n: 2 * n]),
                   "anchors": list(flat[2 * n: 3 * n]) if self.with_anchors else None}
        batch = bev.shape[0]
        return tuple(self.bbox_head.loss_vector(example, p, t, batch) for t, p in enumerate(preds))


def _has_collective(module):
    from .syncbn import DistributedSyncBN
    from . import dist as d3d_dist

    return d3d_dist.exchanging() and any(isinstance(m, DistributedSyncBN) for m in module.modules())


class GraphedTail:
    """Owns the captured graphs for ONE signature (shapes / dtypes of the BEV map and the target arrays)."""

    def __init__(self, graphed, signature, n_tasks, with_anchors):
        self.graphed = graphed
        self.signature = signature
        self.n_tasks = n_tasks
        self.with_anchors = with_anchors

    @staticmethod
    def flat_inputs(example, with_anchors):
        flat = [t.contiguous() for t in example["labels"]]
        flat += [t.float().contiguous() for t in example["reg_targets"]]
        if with_anchors:
            flat += [t.float().contiguous() for t in example["anchors"]]
        return flat

    @staticmethod
    def signature_of(bev, flat, autocast_dtype):
        return (tuple(bev.shape), bev.dtype, bev.stride(), autocast_dtype) + \
            tuple((tuple(t.shape), t.dtype) for t in flat)

    @classmethod
    def build(cls, detector, example, bev, autocast_dtype):
        """Capture for the shapes of this (example, bev); returns None when the tail cannot be captured."""
        head = detector.bbox_head
        neck = detector.neck if detector.with_neck else None
        if not (bev.is_cuda and bev.requires_grad and hasattr(head, "loss_vector")):
            return None
        if _has_collective(head) or (neck is not None and _has_collective(neck)):
            return None
        with_anchors = bool(getattr(head, "use_direction_classifier", False))
        flat = cls.flat_inputs(example, with_anchors)
        if not head.loss_vector_ok(bev, flat[: len(example["labels"])]):
            return None
        tail = _Tail(neck, head, with_anchors)
        # (The parameters' gradient accumulators are created inside make_graphed_callables, i.e. on its capture
        # stream; autograd then reports a stream mismatch once and hops streams when it accumulates.  Creating and
        # holding them on the compute stream beforehand is worse: the capture would have to synchronise with the
        # default stream, which invalidates it -- measured: hipStreamEndCapture crashes.)
        sample = (bev.detach().clone().requires_grad_(True),) + tuple(t.clone() for t in flat)
        ctx = torch.autocast(device_type="cuda", dtype=autocast_dtype, cache_enabled=False) \
            if autocast_dtype is not None else torch.autocast(device_type="cuda", enabled=False)
        # the warm-up iterations inside make_graphed_callables run real forward passes: keep the BatchNorm running
        # statistics (and counters) of the eager model unchanged by restoring them afterwards
        buffers = {k: v.clone() for k, v in tail.state_dict().items() if k.split(".")[-1] in
                   ("running_mean", "running_var", "num_batches_tracked")}
        # MIOpen find mode (cudnn.benchmark) must be off while warming up and capturing: the solvers it picks for
        # the weight gradients (split-K igemm with atomics) rely on zero-initialisation that is not part of the
        # captured work, so every replay accumulated onto the previous one -- gradient norm inf after a few
        # steps (round-2 diagnosis).  Immediate-mode solvers replay correctly.
        from . import dense_conv

        # packed conv weights: a capture packs into its own buffers with its own launch (dense_conv._packed), so that a
        # replay convolves with the weights of ITS step; entries of an earlier capture must not satisfy this one
        dense_conv._CAPTURE_PACK.clear()
        # No cyclic garbage collection while the stream is capturing: a collection that starts inside the capture can
        # finalise objects of EARLIER work that own device resources (streams, events, library contexts), and releasing
        # those is not a capturable operation -- seen once as an abort of the test process with "Garbage-collecting" on top
        # of the capture's stack (round 6, in a full-suite run; the test passes on its own).  Collect first, then hold.
        import gc

        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            with ctx, torch.backends.cudnn.flags(enabled=True, benchmark=False,
                                                 deterministic=torch.backends.cudnn.deterministic):
                graphed = torch.cuda.make_graphed_callables(tail, sample, num_warmup_iters=3, allow_unused_input=True)
        finally:
            if gc_was_on:
                gc.enable()
            dense_conv._CAPTURE_PACK.clear()  # the graphs' private pool owns the buffers from here on
        with torch.no_grad():
            sd = tail.state_dict()
            for k, v in buffers.items():
                sd[k].copy_(v)
        return cls(graphed, cls.signature_of(bev, flat, autocast_dtype), len(example["labels"]), with_anchors)

    def __call__(self, bev, flat):
        return self.graphed(bev, *flat)


def run_tail(detector, example, bev):
    """neck + head + loss of `detector` on the dense map `bev` through the captured graphs (capturing on first
    use and whenever the signature changes); None when this tail / these inputs cannot be graphed."""
    if not (GRAPH_TAIL and torch.is_grad_enabled() and detector.training and bev.is_cuda):
        return None
    if getattr(detector, "_graph_tail_failed", False):
        return None
    if not GRAPH_TAIL_DIST:
        from . import dist as d3d_dist

        if d3d_dist.exchanging():
            return None
    autocast_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
    head = detector.bbox_head
    if not hasattr(head, "loss_vector"):
        return None
    with_anchors = bool(getattr(head, "use_direction_classifier", False))
    flat = GraphedTail.flat_inputs(example, with_anchors)
    sig = GraphedTail.signature_of(bev, flat, autocast_dtype)
    cache = detector.__dict__.setdefault("_graph_tails", {})
    gt = cache.get(sig)
    if gt is None:
        if len(cache) >= 4:  # shapes keep changing (ragged batches): graphs would be re-captured forever
            return None
        try:
            # a captured graph bakes in the addresses of everything allocated during capture: the step arena (arena.py)
            # hands out addresses that depend on the sparse stack's sizes of THIS batch, so the capture (and the warm-up
            # runs in front of it) allocates from the caching allocator's graph pool instead
            from . import arena

            with arena.bound(None):
                gt = GraphedTail.build(detector, example, bev, autocast_dtype)
        except Exception as e:  # capture is an optimisation: report loudly, continue eagerly
            import warnings

            warnings.warn("det3d_amd: hipGraph capture of the RPN/head/loss tail failed (%s: %s); running it eagerly"
                          % (type(e).__name__, e))
            detector.__dict__["_graph_tail_failed"] = True
            return None
        if gt is None:
            detector.__dict__["_graph_tail_failed"] = True
            return None
        cache[sig] = gt
    with torch.autocast(device_type="cuda", enabled=False):
        vecs = gt(bev, flat)
    return head.loss_from_vectors(vecs)
