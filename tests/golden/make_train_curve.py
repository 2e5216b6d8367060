"""CPU-oracle leg of the 300-step training comparison:

    python tests/golden/make_train_curve.py [steps]      (CPU, ~20-40 minutes for 300 steps; no GPU, no /root/reference)

oracle/model_ref.DetectorRef (pinned to the reference's VoxelNet by tests/test_oracle_model_cpu.py) in float32 on the
CPU through trainer.TrainStep -> tests/golden/train_curve_f32.pt: the loss of every step, the gradient norm of every
step, and the raw head outputs on the held-out batch after the last step (every 4th row / column).
tests/test_precision_gpu.py runs the same schedule on the HIP path in split arithmetic and in exact f32 and compares the
three curves and the three sets of final predictions."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import train_run_spec as T  # noqa: E402


def main():
    from det3d_amd import synthetic
    from oracle import model_ref as M

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else T.STEPS
    cfg = T.config()
    ds = T.dataset()
    ref = M.DetectorRef(cfg)
    ref.load_state_dict(T.initial_state(cfg))
    ref.train()
    torch.set_num_threads(int(os.environ.get("D3D_CPU_THREADS", "16")))
    step = T.train_step_for(ref, cfg)
    step.sched.total_step = T.STEPS  # (a shortened run keeps the 300-step schedule)
    batches = [M.cpu_example(ds.host_batch(i, T.BATCH), synthetic.KITTI_CAR) for i in range(T.N_BATCHES)]
    losses, gnorms = [], []
    t0 = time.time()
    for i in range(steps):
        losses.append(float(step(batches[i % T.N_BATCHES])))
        gnorms.append(float(step.last["grad_norm"]))
        if i % 10 == 0:
            print("step %d loss %.5f gnorm %.4f  (%.0f s)" % (i, losses[-1], gnorms[-1], time.time() - t0), flush=True)
    ref.eval()
    with torch.no_grad():
        ex = M.cpu_example(ds.host_batch(T.HELD_OUT, T.BATCH), synthetic.KITTI_CAR)
        x = ref.features(ex)
        preds = T.subsample(ref.bbox_head.tasks[0](x))
    out = dict(steps=steps, losses=losses, grad_norms=gnorms, held_out_preds=preds, batch=T.BATCH, points=T.POINTS,
               torch=torch.__version__)
    name = "train_curve_f32.pt" if steps == T.STEPS else "train_curve_f32_%d.pt" % steps
    torch.save(out, os.path.join(HERE, name))
    print("wrote", name, "final loss", losses[-1])


if __name__ == "__main__":
    main()
