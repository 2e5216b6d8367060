"""det3d_amd/arena.py: the bump arenas behind the step's data-dependent tensors (host logic, no GPU)."""
import threading

import pytest
import torch

from det3d_amd import arena


def test_bump_arena_hands_out_disjoint_aligned_independent_tensors():
    a = arena.BumpArena("cpu", "t", initial_bytes=1 << 20)
    a.begin()
    x = a.empty((100, 7), torch.float32)
    y = a.empty(33, torch.int32)
    z = a.empty((5, 3, 2), torch.bfloat16)
    for t, shape, dt in ((x, (100, 7), torch.float32), (y, (33,), torch.int32), (z, (5, 3, 2), torch.bfloat16)):
        assert tuple(t.shape) == shape and t.dtype == dt and t.is_contiguous() and (t.data_ptr() - a.buf.data_ptr()) % arena.ALIGN == 0
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in (x, y, z))
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(2))
    base = a.buf.data_ptr()
    assert all(base <= lo and hi <= base + a.cap for lo, hi in spans)
    # independent version counters: an in-place op on one tensor is invisible to autograd's check on another
    x.zero_()
    v = y._version
    x.add_(1)
    assert y._version == v
    assert a.spills == 0


def test_overflow_spills_to_torch_empty_and_regrows_at_the_next_rewind():
    a = arena.BumpArena("cpu", "t", initial_bytes=1)  # one granule
    a.begin()
    cap0 = a.cap
    big = a.empty((cap0 // 4 + 10,), torch.float32)  # does not fit
    assert a.spills == 1 and not (a.buf.data_ptr() <= big.data_ptr() < a.buf.data_ptr() + a.cap)
    small = a.empty((16,), torch.float32)
    assert a.spills == 2  # the bump pointer keeps counting what the generation would have needed
    g = a.generation
    a.begin()
    assert a.generation == g + 1 and a.cap >= (cap0 + 40) * 1.25 - arena.BumpArena.GRANULE and a.cap > cap0
    again = a.empty((cap0 // 4 + 10,), torch.float32)
    assert a.buf.data_ptr() <= again.data_ptr() < a.buf.data_ptr() + a.cap and a.spills == 2
    del small


def test_binding_is_per_thread_and_unbound_means_torch_empty():
    assert arena.current() is None
    t = arena.empty((3, 3), torch.float32, torch.device("cpu"))
    assert tuple(t.shape) == (3, 3)
    a = arena.BumpArena("cpu", "t", initial_bytes=1 << 20).begin()
    seen = {}

    def other():
        seen["cur"] = arena.current()

    with arena.bound(a):
        assert arena.current() is a
        u = arena.empty((4,), torch.float32, torch.device("cpu"))
        assert a.buf.data_ptr() <= u.data_ptr() < a.buf.data_ptr() + a.cap
        th = threading.Thread(target=other)
        th.start()
        th.join()
        with arena.bound(None):
            assert arena.current() is None
        assert arena.current() is a
    assert arena.current() is None and seen["cur"] is None


def test_lease_refuses_a_backward_after_the_rewind():
    a = arena.BumpArena("cpu", "step", initial_bytes=1 << 20).begin()
    with arena.bound(a):
        lease = arena.Lease()
    assert lease.check("node") is a
    a.begin()
    with pytest.raises(arena.ArenaError, match="rewound"):
        lease.check("node")
    assert arena.Lease().check("node") is None  # no arena bound: nothing to check
