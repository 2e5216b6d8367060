"""prefetch.GeometryPrefetcher without a GPU: both entry points (`prepare`, `submit` -> `Pending.result`) degrade to the
inline build, call the builder exactly once per batch and attach the backbone's plan the same way."""
from det3d_amd.prefetch import GeometryPrefetcher, Pending, _tensors


class _Backbone:
    def __init__(self):
        self.calls = []

    def plan_geometry(self, coords, batch, shape):
        self.calls.append((coords, batch, shape))
        return dict(coors=coords, indice_dict={})


class _Detector:
    def __init__(self):
        self.backbone = _Backbone()


def test_inline_fallbacks_build_once_and_attach_the_plan():
    import torch

    built = []

    def make(i):
        built.append(i)
        return dict(coordinates=torch.zeros(3, 4, dtype=torch.int32), num_voxels=[3, 0], shape=[(8, 8, 2)], tag=i)

    det = _Detector()
    pf = GeometryPrefetcher(make, det, device=None, enabled=True)
    if torch.cuda.is_available():
        return  # the GPU behaviour is tests/test_graph_tail_gpu.py::test_prefetched_geometry_is_the_inline_geometry
    assert pf.enabled is False
    a = pf.prepare(0)
    h = pf.submit(1)
    assert isinstance(h, Pending) and built == [0, 1]
    b = h.result()
    assert h.result() is b                       # idempotent
    assert a["tag"] == 0 and b["tag"] == 1
    assert "sparse_plan" in a and "sparse_plan" in b and len(det.backbone.calls) == 2
    assert det.backbone.calls[0][1:] == (2, (8, 8, 2))
    pf.close()                                   # nothing to join
    # an example that already carries a plan is left alone
    pf2 = GeometryPrefetcher(lambda i: dict(make(i), sparse_plan="given"), det, enabled=True)
    assert pf2.prepare(5)["sparse_plan"] == "given" and len(det.backbone.calls) == 2


def test_tensor_walk_finds_nested_tensors_once():
    import torch

    class Box:
        def __init__(self, t):
            self.t = t
            self.me = self       # a cycle

    t1, t2 = torch.zeros(1), torch.ones(2)
    found = list(_tensors(dict(a=[t1, (t2,)], b=Box(t1), c="x", d=None), set()))
    assert sum(f is t1 for f in found) == 2 and sum(f is t2 for f in found) == 1
