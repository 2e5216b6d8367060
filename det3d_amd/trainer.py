"""One optimizer step of the reference training loop, without its generic hook machinery:
   forward -> sum(losses["loss"]) -> backward -> gradient mean over ranks -> clip(35) ->
   true weight decay + Adam(0.9, 0.99) -> one-cycle lr / momentum
References: det3d/torchie/trainer/trainer.py:347-415 (batch_processor_inline, train loop),
det3d/core/utils/dist_utils.py:45-57 (DistOptimizerHook), hooks/optimizer.py:10-13 (clip),
det3d/solver/fastai_optim.py:155-171 (p *= 1 - wd*lr, then Adam with weight_decay 0),
det3d/solver/learning_schedules_fastai.py:68-95 (OneCycle, cosine annealing).
"""
import contextlib
import gc
import math

import torch

from . import dist as d3d_dist


def freeze_gc():
    """Call once after the warm-up steps of a training loop.  One step creates a few thousand short-lived
    Python containers, so CPython runs a full generational collection every few steps, and a full collection
    walks every tracked object of the process (torch + numpy: ~10^6 objects, 40-80 ms) -- measured on MI355X
    as a bimodal 8.5 / 12.5 ms step (bench.py --no-gc-freeze).  gc.freeze() moves everything alive now into
    the permanent generation; the collector stays enabled and later collections only scan what the loop
    itself creates."""
    gc.collect()
    gc.freeze()


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    def __init__(self, total_step, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
        self.total_step = max(int(total_step), 1)
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, tuple(moms), div_factor, pct_start
        self.low_lr = lr_max / div_factor

    def at(self, step):
        """(lr, beta1) for global step `step` -- LRSchedulerStep.step semantics: phase starts
        are int(total * pct); the last phase whose start <= step wins."""
        a1 = int(self.total_step * self.pct_start)
        step = min(step, self.total_step)
        if step >= a1 and self.total_step > a1:
            pct = (step - a1) / (self.total_step - a1)
            return (annealing_cos(self.lr_max, self.low_lr / 1e4, pct),
                    annealing_cos(self.moms[1], self.moms[0], pct))
        pct = step / max(a1, 1)
        return annealing_cos(self.low_lr, self.lr_max, pct), annealing_cos(self.moms[0], self.moms[1], pct)


class TrainStep:
    def __init__(self, model, total_steps, lr_max=3e-3, wd=0.01, moms=(0.95, 0.85), div_factor=10.0,
                 pct_start=0.4, grad_clip=35.0, autocast_dtype=None, step_arena=None):
        """step_arena: the data-dependent tensors of the sparse forward / backward (activations, their gradients,
        workspaces) come from the device's step arena, rewound at the start of every call (arena.py) -- a steady-state
        step then allocates nothing from the device.  Needs forward, backward and optimizer inside this call on one
        stream, which is what __call__ does."""
        self.model = model
        if step_arena is None:
            from . import runtime

            step_arena = runtime.get("step_arena")
        self.step_arena = bool(step_arena)
        self.sched = OneCycle(total_steps, lr_max, moms, div_factor, pct_start)
        self.wd = wd
        self.grad_clip = grad_clip
        self.autocast_dtype = autocast_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        # one rank: a single concatenation after backward; several ranks: 3 buckets whose all-reduces start from
        # autograd hooks and overlap the sparse backbone's backward (D3D_DP_BUCKETS overrides the count)
        import os

        nb = int(os.environ.get("D3D_DP_BUCKETS", "3"))
        self.overlap = d3d_dist.exchanging() and nb > 0
        if self.overlap:
            self.bucket = d3d_dist.OverlappedGradBuckets(self.params, nb)
        elif d3d_dist.exchanging():
            self.bucket = d3d_dist.FlatGradBucket(self.params)   # D3D_DP_BUCKETS=0: one flat all-reduce after backward
        else:
            self.bucket = d3d_dist.LocalGradClip(self.params)
        kw = dict(lr=self.sched.low_lr, betas=(moms[0], 0.99), weight_decay=0.0)
        dev = self.params[0].device
        # parameters on the GPU: clip + decay + Adam are the two launches of d3d_adam_step (fused_adam.py); the torch
        # optimizer below is the CPU plumbing path (BASELINE configs[0], the gloo tests)
        self.fused = None
        if dev.type == "cuda" and os.environ.get("D3D_FUSED_ADAM", "1") != "0":  # 0: the PyTorch launch sequence (A/B)
            from .fused_adam import FusedAdam

            self.fused = FusedAdam(self.params, beta2=0.99)
            self.opt = None
            from .repack import Repacker

            self.repacker = Repacker(self.params)  # every packed conv operand in one launch behind the optimizer
        else:
            try:
                self.opt = torch.optim.Adam(self.params, fused=dev.type == "cuda", **kw)
            except (RuntimeError, TypeError):
                self.opt = torch.optim.Adam(self.params, foreach=True, **kw)
        self.global_step = 0
        self.last = {}
        # diagnosis (bench.py --gpus N, per-rank medians): a list that receives one tuple of HIP events per step --
        # (step start, backward enqueued = in front of the wait for the gradient exchange, exchange waited for, step
        # end), all recorded on the stream the step runs on
        self.probe = None

    # ---- resume (the reference keeps epoch / iter / optimizer in the checkpoint: trainer.py save_checkpoint,
    # resume :470-499); here the schedule position is the global step
    def state_dict(self):
        return dict(global_step=self.global_step, optimizer=(self.fused or self.opt).state_dict())

    def load_state_dict(self, state):
        self.global_step = int(state["global_step"])
        (self.fused or self.opt).load_state_dict(state["optimizer"])

    def _ctx(self):
        if self.autocast_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast(device_type="cuda", dtype=self.autocast_dtype)

    def __call__(self, example, after_forward=None):
        """one optimizer step on `example`.  after_forward: called between the forward and the backward enqueue -- the
        place to start host work for a LATER step (the next geometry build on its worker thread): the forward phase is
        the host-bound part of the step, the backward's large dense kernels leave the host time to spare."""
        from . import arena

        A = None
        dev = self.params[0].device
        if self.step_arena and dev.type == "cuda":
            A = arena.step_arena(dev).begin()
        with arena.bound(A):
            return self._step(example, after_forward)

    def _step(self, example, after_forward):
        lr, mom = self.sched.at(self.global_step)
        ev = None
        if self.probe is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        self.bucket.zero()
        with self._ctx():
            losses = self.model(example, return_loss=True)
        loss = losses["loss"][0] if len(losses["loss"]) == 1 else sum(losses["loss"])  # (sum() starts from 0: one launch)
        if after_forward is not None:
            after_forward()
        loss.backward()
        if self.fused is not None:
            # the exchanged sums (or the local gradients) as they are: mean over ranks, clipping, true weight decay
            # and Adam happen inside the two launches
            if ev is not None:
                ev[1].record()
            if self.overlap:
                self.bucket.finish_sums()  # exchanges were started by the hooks during backward
                scale = 1.0 / d3d_dist.world_size()
            else:
                self.bucket.gather()
                self.bucket.allreduce_mean()
                scale = 1.0
            if ev is not None:
                ev[2].record()
            gnorm = self.fused.step(lr, mom, decay=1.0 - self.wd * lr, max_norm=self.grad_clip, grad_scale=scale)
            self.repacker.run()
        else:
            for g in self.opt.param_groups:
                g["lr"] = lr
                g["betas"] = (mom, g["betas"][1])
            if self.overlap:
                gnorm = self.bucket.finish(self.grad_clip)
            else:
                self.bucket.gather()  # one cat; p.grad become slices of the flat buffer
                self.bucket.allreduce_mean()
                gnorm = self.bucket.clip_norm_(self.grad_clip)
            with torch.no_grad():
                torch._foreach_mul_(self.params, 1.0 - self.wd * lr)  # true weight decay, bn_wd=True
            self.opt.step()
        self.global_step += 1
        if ev is not None:
            if self.fused is None:
                ev[1] = ev[2] = None  # (the unfused optimizer path is not instrumented)
            ev[3].record()
            self.probe.append(tuple(ev))
        self.last = dict(loss=loss.detach(), grad_norm=gnorm, lr=lr, losses=losses)
        return loss.detach()
