"""eager vs eager vs graphed tail: per-parameter gradient deviations (diagnostic for tests/test_graph_tail_gpu.py)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_graph_tail_gpu as T  # noqa: E402

for dtype in (None, torch.bfloat16):
    cfg, eager = T._model("kitti_car_second.py")
    eager2 = copy.deepcopy(eager)
    graphed = copy.deepcopy(eager)
    examples = T._kitti_examples(2)
    a = T._run(eager, examples, False, dtype)
    a2 = T._run(eager2, examples, False, dtype)
    b = T._run(graphed, examples, True, dtype)
    print("dtype", dtype, "losses", [x[0] for x in a], [x[0] for x in a2], [x[0] for x in b])
    for step, ((la, _, ga), (la2, _, ga2), (lb, _, gb)) in enumerate(zip(a, a2, b)):
        for k in ga:
            scale = float(ga[k].abs().max()) + 1e-12
            noise = float((ga[k] - ga2[k]).abs().max()) / scale
            diff = float((ga[k] - gb[k]).abs().max()) / scale
            flag = "  <<<" if diff > 4 * noise + 0.01 else ""
            print("step %d %-44s scale %9.3g  eager-eager %.2e  graph-eager %.2e%s" % (step, k, scale, noise, diff, flag))
