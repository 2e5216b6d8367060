"""Randomised cross-check of det3d_amd.synthetic.assign_targets_task / task_anchors (the numpy restatement the GPU
tests use as their checker at full size) against the REFERENCE's own TargetAssigner.generate_anchors + assign_v2
(det3d/core/anchor/target_assigner.py:68-200), run as plain Python through ref_loader.  Needs /root/reference;
run in its own process (ref_loader installs stub modules):   python tests/golden/crosscheck_assign.py [n_cases]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_loader  # noqa: E402


def main(n_cases):
    ref_loader.install()
    ta = ref_loader.ref("det3d.core.anchor.target_assigner")
    ag = ref_loader.ref("det3d.core.anchor.anchor_generator")
    bc = ref_loader.ref("det3d.core.bbox.box_coders")
    rs = ref_loader.ref("det3d.core.bbox.region_similarity")
    from det3d_amd import synthetic

    _mg = np.meshgrid
    np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
    rng = np.random.default_rng(2024)
    try:
        for case in range(n_cases):
            nd = int(rng.choice([7, 9]))
            vec = bool(rng.integers(0, 2))
            k = int(rng.integers(1, 4))
            fmap = [1, int(rng.integers(6, 20)), int(rng.integers(6, 20))]
            xy = float(rng.choice([40.0, 51.2]))
            gens = []
            for c in range(k):
                hi = float(rng.uniform(0.45, 0.65))
                gens.append(dict(sizes=[float(v) for v in rng.uniform(0.6, 6.0, 3)], rotations=[0, 1.57],
                                 anchor_ranges=[-xy, -xy, float(rng.uniform(-1.5, 0.2))] * 1 + [xy, xy, 0.0],
                                 matched_threshold=hi, unmatched_threshold=hi - float(rng.uniform(0.05, 0.25)),
                                 class_name="c%d" % c))
                gens[-1]["anchor_ranges"][5] = gens[-1]["anchor_ranges"][2]
                if nd == 9:
                    gens[-1]["velocities"] = [0, 0]
            generators = [ag.AnchorGeneratorRange(sizes=g["sizes"], anchor_ranges=g["anchor_ranges"], rotations=g["rotations"],
                                                  velocities=g.get("velocities"), match_threshold=g["matched_threshold"],
                                                  unmatch_threshold=g["unmatched_threshold"], class_name=g["class_name"])
                          for g in gens]
            assigner = ta.TargetAssigner(box_coder=bc.GroundBox3dCoder(vec_encode=vec, n_dim=nd), anchor_generators=generators,
                                         region_similarity_calculator=rs.NearestIouSimilarity(), positive_fraction=None,
                                         sample_size=512)
            ref_anchors = assigner.generate_anchors(fmap)["anchors"].reshape(-1, nd)
            adict = assigner.generate_anchors_dict(fmap)
            anchors, kk, r, hi, lo = synthetic.task_anchors(fmap, gens)
            assert kk == k and np.array_equal(anchors, ref_anchors), case
            n = int(rng.integers(0, 12))
            cls = rng.integers(1, k + 1, n).astype(np.int32)
            gt = np.zeros((n, nd), np.float32)
            gt[:, :2] = rng.uniform(-0.9 * xy, 0.9 * xy, (n, 2))
            for i in range(n):
                g = gens[cls[i] - 1]
                gt[i, 2] = g["anchor_ranges"][2]
                gt[i, 3:6] = np.array(g["sizes"]) * rng.uniform(0.7, 1.4, 3)
            gt[:, -1] = rng.uniform(-3.14, 3.14, n)
            if nd == 9:
                gt[:, 6:8] = rng.normal(0, 3, (n, 2))
                gt[rng.random(n) < 0.2, 6:8] = np.nan
            if n and rng.random() < 0.5:  # a gt exactly on an anchor centre / copies of an anchor (ties)
                j = int(rng.integers(0, len(anchors)))
                gt[0, :2] = anchors[j, :2]
                if rng.random() < 0.5:
                    gt[0] = anchors[j]
                    cls[0] = (j // r) % k + 1
            names = np.array(["c%d" % (c - 1) for c in cls])
            ret = assigner.assign_v2(adict, gt.copy(), None, gt_classes=cls, gt_names=names)
            lab, tgt, w = synthetic.assign_targets_task(anchors, k, r, hi, lo, gt, cls, encode_angle_vector=vec)
            assert np.array_equal(lab, ret["labels"]), (case, int((lab != ret["labels"]).sum()))
            assert np.array_equal(w, ret["bbox_outside_weights"]), case
            assert np.allclose(tgt, ret["bbox_targets"], atol=1e-6), case
    finally:
        np.meshgrid = _mg
    print("crosscheck_assign: %d random task groups identical to the reference" % n_cases)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
