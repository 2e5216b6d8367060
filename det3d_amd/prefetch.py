"""Input-side geometry of batch i+1 on a second HIP stream while batch i trains.

The training step has two places where the host must read device-side counts: the voxel counts of the batch
(point_cloud_ops.VoxelBatch.sync) and the output counts of the strided sparse convolutions
(spconv.SparseSequential.plan).  Done inline, each of them drains the GPU at the start of a step, and the sparse
forward that follows (short kernels, one Python autograd node each) is issued into an idle device.  Everything in
that prologue depends only on the INPUT batch -- device voxelizer + fused mean VFE, anchor target assignment,
every rulebook of the sparse backbone -- so it can run one batch ahead:

    pf = GeometryPrefetcher(make_example, detector)
    ex = pf.prepare(batch_0)
    for i in range(steps):
        loss = train_step(ex)            # enqueues forward + backward + optimiser of batch i (asynchronous)
        ex = pf.prepare(batch_{i+1})     # side stream: its two host syncs wait for the side stream only

The reference gets the same overlap from its DataLoader worker processes (voxelisation and target assignment on
CPU cores, torchie/trainer/trainer.py:34-57 moves the finished example to the GPU); here the work is on the GPU,
so the overlap is between two HIP streams of one process.

Memory: tensors produced on the side stream are handed to the compute stream with `record_stream`, so the caching
allocator does not recycle them while the compute stream still reads them.

`submit()` / `Pending.result()` move the preparation onto a worker THREAD as well: the two host syncs of a build then
block that thread (the waits release the GIL), not the thread that enqueues the training step:

    nxt = pf.submit(batch_1)
    for i in range(steps):
        loss = train_step(ex)            # the worker builds batch i+1 meanwhile
        ex = nxt.result()                # joins the side stream into the compute stream
        nxt = pf.submit(batch_{i+2})

Memory, round 6: the data-dependent tensors of a build (neighbour tables sized by capacities of several hundred MB, row
plans, coordinates of every level) come from a RING of bump arenas (arena.py), slot = build number mod `ring`: no
allocator traffic in steady state.  The contract that makes the ring safe: builds are consumed in the order they were
requested, and the example of build q is dead (its step fully enqueued on the consuming stream) by the time build
q + ring - 1 is REQUESTED -- true for both loops above (there, at most two builds are outstanding).  Each request records
an event on the consuming stream; build q waits on the device for the event of request q - 1 before it writes slot
q mod ring, whose previous tenant was read by a step enqueued before that event.  `ring=0` restores plain allocations.

Measured on the KITTI step: the blocking part of `prepare` varied between 1.6 and 4.1 ms per step from box to box,
enough to make a 7.6 ms step host-bound (gpurun_out/r3final3); on the worker it is off the critical path.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import torch


def _tensors(obj, seen):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v, seen)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (type, torch.nn.Module)):
        if id(obj) in seen:
            return
        seen.add(id(obj))
        yield from _tensors(vars(obj), seen)


_RESERVED = {}  # device index -> stream created by reserve_stream


def _new_stream(dev):
    # small kernels: by default they may cut in front of the step's (D3D_PREFETCH_PRIORITY=0: equal priority)
    from . import runtime

    return torch.cuda.Stream(dev, priority=int(runtime.get("prefetch_priority")))


def reserve_stream(dev):
    """Create the geometry stream NOW and submit one trivial launch on it, so that the runtime binds it to a hardware
    queue before other libraries create their streams.  dist.init_dist calls this in front of init_process_group:
    measured on MI355X (profiles/round4_multirank.txt), the same `Stream(priority=-1)` created AFTER the RCCL process
    group ran its ~75 small kernels per step at 40-90 us each instead of 5-15 us -- no longer cutting in front of the
    step's big kernels -- and the step waited 1.2 ms for its next batch's rulebooks (7.5 -> 8.8 ms)."""
    dev = torch.device(dev)
    if not torch.cuda.is_available() or os.environ.get("D3D_PREFETCH_RESERVE", "1") == "0":
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _RESERVED:
        st = _new_stream(torch.device("cuda", idx))
        with torch.cuda.stream(st):
            torch.zeros(1, device=torch.device("cuda", idx))
        st.synchronize()
        _RESERVED[idx] = st
        # the same for the weight-gradient stream of the backward pass (sparse_stack.WGRAD_STREAM): created lazily it
        # would come after the communication library's streams
        from . import sparse_stack

        if sparse_stack.WGRAD_STREAM:
            ws = sparse_stack.wgrad_stream(torch.device("cuda", idx))
            with torch.cuda.stream(ws):
                torch.zeros(1, device=torch.device("cuda", idx))
            ws.synchronize()
    return _RESERVED[idx]


_GATE = {}  # device index -> event recorded on the training stream behind the most recent sparse forward


def mark_sparse_forward(dev):
    """Called by the sparse backbones at the end of their training forward.  With runtime `geometry_gate` = "sparse_fwd"
    a build requested afterwards starts on the device only when that forward has FINISHED: the geometry kernels (hash
    probes, atomics, scans -- latency-bound) then run next to the dense RPN's MFMA-bound kernels instead of next to the
    gather-bound sparse forward of the same step, which they slow down (profiles/round6_geometry_gate.txt)."""
    from . import runtime

    if runtime.get("geometry_gate") != "sparse_fwd":
        return
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _GATE[idx] = torch.cuda.current_stream(dev).record_event()


def _stream_for(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return _RESERVED.get(idx) or _new_stream(dev)


class GeometryPrefetcher:
    def __init__(self, make_example, detector=None, device=None, enabled=True, ring=None):
        """make_example(batch) -> example dict (runs the device voxelizer etc. on the CURRENT stream);
        detector: if its backbone has `plan_geometry`, the rulebooks are built here too and travel in
        example["sparse_plan"].  ring: slots of the geometry arena ring (module docstring); 0 = caching allocator."""
        self.make_example = make_example
        if ring is None:
            from . import runtime

            ring = runtime.get("geometry_ring")
        if int(ring) in (1, 2):
            # build q waits for the event of request q - 1, which covers the step that consumed build q - 3 (two builds are
            # outstanding in the threaded loop): a slot may only come round again after three builds
            raise ValueError("GeometryPrefetcher: ring must be 0 (caching allocator) or >= 3, got %r" % (ring,))
        self.ring = int(ring) if (enabled and torch.cuda.is_available()) else 0
        self._slots = {}
        self._seq = 0
        self._mark = None  # event on the consuming stream, recorded at the previous request
        self.backbone = getattr(detector, "backbone", None) if detector is not None else None
        if not hasattr(self.backbone, "plan_geometry"):
            self.backbone = None
        self.enabled = enabled and torch.cuda.is_available()
        self.device = device
        self.stream = None

    def prepare(self, batch, ready_event=None):
        """ready_event: recorded by whoever produced `batch` (e.g. the H2D copy); None = the tensors of `batch` are
        already complete (resident inputs)."""
        if not self.enabled:
            return self._build(batch)
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        if self.stream is None:
            self.stream = _stream_for(dev)
        main = torch.cuda.current_stream(dev)
        slot, mark = self._next_slot(dev, main)
        gate = _GATE.get(dev.index)
        if ready_event is not None:
            self.stream.wait_event(ready_event)
        if mark is not None:
            self.stream.wait_event(mark)
        if gate is not None:
            self.stream.wait_event(gate)
        with torch.cuda.stream(self.stream):
            ex = self._build(batch, slot)
            done = torch.cuda.Event()
            done.record(self.stream)
        main.wait_event(done)
        for t in _tensors(ex, set()):
            if t.is_cuda:
                t.record_stream(main)
        return ex

    # ---- the same on a worker thread
    def submit(self, batch, ready_event=None):
        """start `prepare(batch)` on the worker thread; returns a handle whose result() hands the example to the calling
        thread's current stream.  One build at a time, in submission order."""
        if not self.enabled:
            return Pending(self, None, self._build(batch), None)
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        if self.stream is None:
            self.stream = _stream_for(dev)
        if getattr(self, "_pool", None) is None:
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="d3d-geometry")

        slot, mark = self._next_slot(dev, torch.cuda.current_stream(dev))
        gate = _GATE.get(dev.index)  # as of this request: the sparse forward of the step enqueued last

        def work():
            torch.cuda.set_device(dev)  # current device and current stream are per thread
            if ready_event is not None:
                self.stream.wait_event(ready_event)
            if mark is not None:
                self.stream.wait_event(mark)
            if gate is not None:
                self.stream.wait_event(gate)
            with torch.cuda.stream(self.stream):
                ex = self._build(batch, slot)
                done = torch.cuda.Event()
                done.record(self.stream)
            return ex, done

        return Pending(self, self._pool.submit(work), None, dev)

    def close(self):
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=True)

    def _next_slot(self, dev, main):
        """(arena slot of this request or None, event the build must wait for or None)"""
        if not self.ring:
            return None, None
        from . import arena

        q, self._seq = self._seq, self._seq + 1
        slot = self._slots.get(q % self.ring)
        if slot is None:
            slot = self._slots[q % self.ring] = arena.BumpArena(dev, "geometry[%d]" % (q % self.ring))
        mark, self._mark = self._mark, main.record_event()
        return slot, mark

    def arena_stats(self):
        return [self._slots[k].stats() for k in sorted(self._slots)]

    def _build(self, batch, slot=None):
        from . import arena

        if slot is not None:
            slot.begin()  # (a regrown buffer is allocated on the geometry stream; record_stream hands it to the consumer)
        with arena.bound(slot):
            ex = self.make_example(batch)
            if self.backbone is not None and "coordinates" in ex and "sparse_plan" not in ex:
                ex["sparse_plan"] = self.backbone.plan_geometry(ex["coordinates"], len(ex["num_voxels"]), ex["shape"][0])
        if slot is not None and isinstance(ex.get("sparse_plan"), dict):
            # the rulebooks live in this ring slot until build q + ring rewinds it: the backbone refuses a plan whose slot
            # has moved on (an example kept longer than the ring allows) instead of reading recycled tables
            ex["sparse_plan"]["ring_ticket"] = (slot, slot.generation)
        return ex


class Pending:
    """an example being prepared by GeometryPrefetcher.submit"""

    def __init__(self, owner, future, example, device):
        self.owner, self.future, self.example, self.device = owner, future, example, device

    def result(self):
        if self.future is None:
            return self.example
        ex, done = self.future.result()   # re-raises what the build raised
        self.future = None
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        for t in _tensors(ex, set()):
            if t.is_cuda:
                t.record_stream(main)
        self.example = ex
        return ex
