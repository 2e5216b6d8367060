"""Every packed conv operand rewritten in ONE launch behind the optimizer step (`d3d_pack_all`, csrc/d3d_optim.hip).

The sparse and dense conv layers cache their packed weights per parameter version (spconv._PACK_CACHE,
dense_conv._PACK); an optimizer step bumps every version, so the next forward used to repack layer by layer -- 21 small
launches inside the host-bound forward phase of the KITTI step (14 sparse + 7 dense), each in front of its layer's first
kernel.  `Repacker.run()` is called by TrainStep right after `d3d_adam_step`: it walks the caches once, hands every
cached (parameter, packed buffers) pair to one launch that rewrites the SAME buffers, and stamps the entries with the
parameters' new versions -- the next forward finds them fresh.  The backward of the step that just ended is complete on
the stream by then, so nothing still reads the old bytes.  D3D_PACK_ALL=0 keeps the per-layer repacks.

PRECONDITION (enforced): the rewrite is in place, so no autograd graph built before `run()` may run its backward after
it -- its nodes hold the very buffers that now contain the NEW weights.  Every node that keeps packed operands
(spconv._SparseConvFn, sparse_stack, sparse_block, dense_stack) takes `stamp()` in its forward and calls `check()` in
its backward, which raises instead of silently differentiating with the wrong weights.  TrainStep (backward, then
optimizer step, then `run()`) never trips it; a loop that steps the optimizer between forward and backward, or keeps a
graph alive across a step, must set D3D_PACK_ALL=0.
"""
import ctypes
import os

import torch

from . import _lib

ENABLED = os.environ.get("D3D_PACK_ALL", "1") != "0"
_EPOCH = {}  # id(parameter) -> number of in-place rewrites of ITS packed operands so far


def stamp(weights=()):
    """what a node remembers in its forward: the rewrite counts of the parameters whose packed operands it keeps.  (One
    process-global counter made every pending node fail when ANOTHER model's Repacker ran between its forward and
    backward -- a teacher / EMA copy, a second optimizer: ADVICE round 5.)"""
    return tuple((id(w), _EPOCH.get(id(w), 0)) for w in weights)


def check(stamped, who):
    if any(_EPOCH.get(k, 0) != v for k, v in stamped):
        raise RuntimeError("%s: backward of a graph built before Repacker.run() rewrote the packed conv operands in place "
                           "(optimizer step between forward and backward, or a graph kept across steps); run the "
                           "backward first or set D3D_PACK_ALL=0" % who)


class Repacker:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad and p.is_cuda and p.dtype == torch.float32 and p.dim() >= 4]
        self._key = None
        self._table = None
        self._n = 0

    def _entries(self):
        from . import dense_conv, spconv

        out = []
        for p in self.params:
            e = spconv._PACK_CACHE.get(id(p))
            if e is not None and p.dim() == 5 and e[0]() is p and e[7] == (p.data_ptr(), p.device):
                cin, cout = int(p.shape[-2]), int(p.shape[-1])
                if e[4] == cin and p.is_contiguous():  # (the stem's zero-padded input channels keep their own pack)
                    out.append(("s", p, e, None))
            if p.dim() == 4 and p.is_contiguous():
                for mode in ("split", "bf16"):
                    d = dense_conv._PACK.get((id(p), mode))
                    if d is not None and d[0]() is p and d[2] == p.data_ptr():
                        out.append(("d", p, d, mode))
        return out

    def run(self):
        """call right after the optimizer step (parameters and their `_version` already moved)"""
        if not ENABLED or not self.params:
            return 0
        from . import dense_conv, spconv

        ents = self._entries()
        if not ents:
            return 0
        key = []
        for kind, p, e, mode in ents:
            if kind == "s":
                key.append((id(p), p.data_ptr(), e[5].data_ptr(), e[6].data_ptr() if e[6] is not None else 0, e[2], e[3]))
            else:
                f, d = e[3]["fwd"], e[3]["dgrad"]
                key.append((id(p), p.data_ptr(), f[0].data_ptr(), d[0].data_ptr(), mode, 0))
        key = tuple(key)
        if key != self._key:
            arr = (_lib.PackDesc * len(ents))()
            for i, (kind, p, e, mode) in enumerate(ents):
                if kind == "s":
                    cin, cout = int(p.shape[-2]), int(p.shape[-1])
                    arr[i] = _lib.PackDesc(weight=p.data_ptr(), fwd=e[5].data_ptr(), fwd_lo=None,
                                           dgrad=e[6].data_ptr() if e[6] is not None else None, dgrad_lo=None, kind=0,
                                           a=p.numel() // (cin * cout), b=cin, c=cout, flip=int(e[3]), dtype=spconv._DT[e[2]])
                else:
                    f, d = e[3]["fwd"], e[3]["dgrad"]
                    arr[i] = _lib.PackDesc(weight=p.data_ptr(), fwd=f[0].data_ptr(), fwd_lo=_lib.ptr(f[1]),
                                           dgrad=d[0].data_ptr(), dgrad_lo=_lib.ptr(d[1]), kind=1, a=int(p.shape[0]),
                                           b=int(p.shape[1]), c=int(p.shape[2] * p.shape[3]), flip=0, dtype=0)
            self._key, self._table, self._n = key, arr, len(ents)
        for kind, p, e, mode in ents:
            _EPOCH[id(p)] = _EPOCH.get(id(p), 0) + 1
        dev = ents[0][1].device
        rc = _lib.lib().d3d_pack_all(ctypes.cast(self._table, ctypes.c_void_p), self._n, _lib.stream_ptr(dev))
        _lib.check(rc, "d3d_pack_all")
        for kind, p, e, mode in ents:  # the entries now hold the operands of the parameters' CURRENT version
            if kind == "s":
                spconv._PACK_CACHE[id(p)] = (e[0], p._version) + tuple(e[2:])
            else:
                dense_conv._PACK[(id(p), mode)] = (e[0], p._version, e[2], e[3])
        return self._n
