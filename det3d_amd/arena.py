"""Step-lifetime device memory: bump arenas for every tensor whose size follows the DATA of a batch.

The reference trains for 20 epochs in one process (det3d/torchie/trainer/trainer.py:34-57); its per-step tensors come from
PyTorch's caching allocator, which serves exact sizes and splits larger blocks.  On this path the sizes of the sparse
stack's tensors change with every batch (active sites per level, rulebook capacities that are multiples of them: a
20k-voxel KITTI batch asks for a 450 MB neighbour table whose size differs by some KB from the previous batch's), so the
allocator kept splitting, fragmenting and growing -- 0.4-5 hipMalloc per step after hundreds of steps, 15 -> 29 GB
reserved, and a 20-step timing window that moved by 10 % with the number of hipMallocs inside it.

Here those tensors come from two kinds of arenas instead -- one flat buffer each, handed out front to back, nothing
freed individually:

  * the STEP arena of a device: rewound by `TrainStep` at the start of every optimizer step.  Everything a step's sparse
    forward / backward allocates through `empty()` lives until the next rewind; all of it is used on the step's stream or
    on a stream joined into it before the step ends (the weight-gradient stream), so stream order alone makes the
    rewind safe.
  * the GEOMETRY ring of a `prefetch.GeometryPrefetcher`: the rulebooks of batch k are built one or two steps ahead on a
    second stream, into slot k % 3; the build waits (on the device) for the step that read the slot's previous tenant.

An arena that runs out hands out ordinary `torch.empty` tensors and remembers the shortfall; the next rewind regrows the
buffer to 1.25 x the largest generation seen.  After the first few batches a step allocates nothing from the device.

Binding is per thread (`bound(arena)`); autograd nodes capture `current()` in their forward and allocate from that arena
in their backward, which PyTorch runs on its own thread.  Outside any binding `empty()` IS `torch.empty` -- tests, the
per-layer API and inference never see an arena.
"""
import contextlib
import threading

import torch

_ESZ = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.int32: 4, torch.uint8: 1, torch.int64: 8,
        torch.float64: 8, torch.int8: 1, torch.int16: 2, torch.bool: 1}
_tls = threading.local()
ALIGN = 256


class ArenaError(RuntimeError):
    pass


class BumpArena:
    """one flat device buffer, handed out front to back; `begin()` rewinds it"""

    GRANULE = 64 << 20

    def __init__(self, device, name="step", initial_bytes=0, headroom=1.25):
        self.device = torch.device(device)
        self.name = name
        self.headroom = headroom
        self.cap = 0
        self.buf = None
        self.storage = None
        self.off = 0          # bump pointer of this generation (keeps counting past `cap`: the size it would have needed)
        self.high = 0         # largest generation so far
        self.generation = 0
        self.spills = 0       # torch.empty fallbacks since construction
        self.grows = 0
        if initial_bytes:
            self._grow(initial_bytes)

    def _grow(self, nbytes):
        cap = (int(nbytes) + self.GRANULE - 1) // self.GRANULE * self.GRANULE
        # (the old buffer stays alive for as long as tensors of earlier generations reference its storage)
        self.buf = torch.empty((cap,), dtype=torch.uint8, device=self.device)
        self.storage = self.buf.untyped_storage()
        self.base = self.buf.storage_offset()
        self.cap = cap
        self.grows += 1

    def begin(self):
        """rewind.  Tensors handed out before this call must not be read or written by work enqueued after it."""
        self.high = max(self.high, self.off)
        if self.high > self.cap:
            self._grow(self.high * self.headroom)
        self.off = 0
        self.generation += 1
        return self

    def empty(self, shape, dtype):
        if isinstance(shape, int):
            shape = (shape,)
        n = 1
        for s in shape:
            n *= int(s)
        es = _ESZ[dtype]
        off = self.off
        end = off + ((n * es + ALIGN - 1) & ~(ALIGN - 1))
        self.off = end
        if end > self.cap:
            self.spills += 1
            return torch.empty(shape, dtype=dtype, device=self.device)
        stride, acc = [], 1
        for s in reversed(shape):
            stride.append(acc)
            acc *= int(s)
        stride.reverse()
        # an independent tensor on the arena's storage (its own version counter: not a view of `buf` for autograd)
        return torch.empty((0,), dtype=dtype, device=self.device).set_(self.storage, (self.base + off) // es, tuple(shape), tuple(stride))

    def stats(self):
        return dict(name=self.name, capacity_MB=self.cap >> 20, used_MB=self.off >> 20, high_MB=max(self.high, self.off) >> 20,
                    generation=self.generation, spills=self.spills, grows=self.grows)


_STEP = {}  # device index -> BumpArena


def step_arena(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    a = _STEP.get(idx)
    if a is None:
        a = _STEP[idx] = BumpArena(torch.device("cuda", idx), "step")
    return a


def current():
    """the arena bound to this thread, or None"""
    return getattr(_tls, "arena", None)


@contextlib.contextmanager
def bound(arena):
    """allocate `empty()` tensors of this thread from `arena` (None: from the caching allocator) inside the block"""
    prev = getattr(_tls, "arena", None)
    _tls.arena = arena
    try:
        yield arena
    finally:
        _tls.arena = prev


def bind(arena):
    """non-scoped form of `bound` for a thread's whole life (the geometry worker); returns the previous binding"""
    prev = getattr(_tls, "arena", None)
    _tls.arena = arena
    return prev


def empty(shape, dtype, device, arena=None):
    """torch.empty, from `arena` / the thread's arena if there is one.  Callers only route tensors here that die with
    the step (activations, gradients of activations, workspaces, rulebooks) -- never parameter gradients or anything
    returned to user code beyond the step."""
    a = arena if arena is not None else getattr(_tls, "arena", None)
    if a is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return a.empty(shape, dtype)


def empty_like(t, arena=None):
    return empty(tuple(t.shape), t.dtype, t.device, arena)


class Lease:
    """what an autograd node keeps of the arena its forward ran under: the backward allocates from the same arena and
    refuses to run once the arena was rewound (its saved activations are gone)."""

    __slots__ = ("arena", "generation")

    def __init__(self):
        self.arena = getattr(_tls, "arena", None)
        self.generation = self.arena.generation if self.arena is not None else -1

    def check(self, what):
        a = self.arena
        if a is not None and a.generation != self.generation:
            raise ArenaError("%s: backward after the %s arena was rewound (generation %d -> %d): the activations of that "
                             "forward pass are gone -- run forward and backward inside one TrainStep, or outside any arena"
                             % (what, a.name, self.generation, a.generation))
        return a
