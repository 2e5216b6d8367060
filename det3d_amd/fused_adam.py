"""Host side of `d3d_adam_step` (csrc/d3d_optim.hip): gradient clipping + true weight decay + Adam over all parameter
tensors in two launches.

Reference: OptimWrapper.step, det3d/solver/fastai_optim.py:155-171 (`p.data.mul_(1 - wd * lr)` for every parameter,
then torch.optim.Adam with weight_decay 0 and the one-cycle momentum written into betas[0] each step) and
clip_grad_norm_ (det3d/torchie/trainer/hooks/optimizer.py:10-13).  The state is the Adam state of PyTorch under the same
names, so a checkpoint written here loads into torch.optim.Adam and back (`state_dict` / `load_state_dict` use its
layout: per parameter index `step`, `exp_avg`, `exp_avg_sq`).

Static per model: the table of (parameter, exp_avg, exp_avg_sq, numel) and the chunk list, uploaded once.  Per step the
host only collects the 84-200 gradient addresses (they are fresh autograd allocations) and makes one library call.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def plan_chunks(numels, min_chunk=4096, max_chunks=4096):
    """chunk size (a multiple of 256 elements) and the chunk list of `d3d_adam_step`: every tensor is cut into chunks of
    that size, tensor by tensor -> (chunk_elems, [(tensor, first element)], first chunk of every tensor + total)"""
    total = sum(numels)
    chunk = max(min_chunk, -(-total // max_chunks))
    chunk = (chunk + 255) // 256 * 256
    chunks, start = [], [0]
    for t, n in enumerate(numels):
        chunks += [(t, o) for o in range(0, max(n, 1), chunk)]
        start.append(len(chunks))
    return chunk, chunks, start


class FusedAdam:
    MIN_CHUNK = 4096      # elements per block of the two kernels
    MAX_CHUNKS = 4096     # the update kernel's blocks each re-add all chunk partials: keep that list short

    def __init__(self, params, beta2=0.99, eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise _lib.D3dError("FusedAdam: no parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda or not p.is_contiguous():
                raise _lib.D3dError("FusedAdam: parameters must be contiguous float32 tensors on one GPU "
                                    "(got %s %s on %s)" % (tuple(p.shape), p.dtype, p.device))
        self.device = dev
        self.beta2, self.eps = float(beta2), float(eps)
        self.step_count = 0
        self.numel = [p.numel() for p in self.params]
        off, offs = 0, []
        for n in self.numel:
            offs.append(off)
            off += (n + 3) // 4 * 4       # 16-byte aligned moment slices
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._m = [self.exp_avg[o: o + n] for o, n in zip(offs, self.numel)]
        self._v = [self.exp_avg_sq[o: o + n] for o, n in zip(offs, self.numel)]
        self.chunk, chunks, start = plan_chunks(self.numel, self.MIN_CHUNK, self.MAX_CHUNKS)
        self.n_chunks = len(chunks)
        self._chunk_start = np.asarray(start, np.int32)
        self._chunks = torch.from_numpy(np.asarray(chunks, np.int32).reshape(-1, 2)).to(dev)
        self._partial = torch.empty(self.n_chunks, dtype=torch.float32, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._grads = np.zeros(len(self.params), np.uint64)
        self._table_ptrs = None
        self._upload_table()

    def _upload_table(self):
        ptrs = [p.data_ptr() for p in self.params]
        rows = [(pp, m.data_ptr(), v.data_ptr(), n) for pp, m, v, n in zip(ptrs, self._m, self._v, self.numel)]
        self._table = torch.from_numpy(np.asarray(rows, np.int64)).to(self.device)
        self._table_ptrs = ptrs

    def step(self, lr, beta1, decay=1.0, max_norm=None, grad_scale=1.0):
        """one update from the `.grad` of every parameter; returns the gradient norm before clipping (device, 0-dim)"""
        L = _lib.lib()
        g = self._grads
        for i, p in enumerate(self.params):
            gr = p.grad
            if gr is None or gr.dtype != torch.float32 or gr.numel() != self.numel[i] or gr.shape != p.shape:
                raise _lib.D3dError("FusedAdam.step: parameter %d needs a float32 gradient of its shape" % i)
            if not gr.is_contiguous():   # e.g. a channels-last weight gradient from a library convolution
                gr = p.grad = gr.contiguous()
            g[i] = gr.data_ptr()
            if p.data_ptr() != self._table_ptrs[i]:   # storage replaced (.to(), load through .data =): re-point the table
                self._upload_table()
        self.step_count += 1
        self.last_lr, self.last_beta1 = float(lr), float(beta1)  # what state_dict() reports in its param_group
        hp = _lib.AdamHyper(lr=lr, beta1=beta1, beta2=self.beta2, eps=self.eps, decay=decay,
                            max_norm=float(max_norm) if max_norm is not None else 0.0, grad_scale=grad_scale,
                            step=self.step_count)
        rc = L.d3d_adam_step(self._table.data_ptr(), len(self.params), self._chunks.data_ptr(), self.n_chunks, self.chunk,
                             self._chunk_start.ctypes.data, g.ctypes.data, ctypes.byref(hp), self._partial.data_ptr(),
                             self._norm.data_ptr(), _lib.stream_ptr(self.device))
        _lib.check(rc, "d3d_adam_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters, or everything keyed on
        # `_version` -- the packed-weight caches of the conv layers, autograd's saved-tensor checks -- keeps serving
        # the values of the previous step
        torch.autograd.graph.increment_version(self.params)
        return self._norm[0].clone()  # (`_norm` is rewritten by the next step: hand out a value, not a view of it)

    # ---- torch.optim.Adam's checkpoint layout
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(self._m, self._v)):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m.clone().view(self.params[i].shape),
                                exp_avg_sq=v.clone().view(self.params[i].shape))
        group = dict(lr=getattr(self, "last_lr", 0.0), betas=(getattr(self, "last_beta1", 0.9), self.beta2), eps=self.eps, weight_decay=0.0, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=True, decoupled_weight_decay=False,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        st = sd["state"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        for i in range(len(self.params)):
            s = st.get(i, st.get(str(i)))
            if s is None:
                continue
            self._m[i].copy_(s["exp_avg"].reshape(-1))
            self._v[i].copy_(s["exp_avg_sq"].reshape(-1))
            self.step_count = max(self.step_count, int(float(s["step"])))
