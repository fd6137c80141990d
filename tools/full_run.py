"""BASELINE.json configs[2] end to end: the full 800-iteration run (300 world-model epochs, then 500
joint world-model + CVAE epochs) on the 10 x 1000-step synthetic loco demo, batch 256, TE/MD/WM 4x1024,
through the trainer class the CLI uses.  Prints wall-clock and the loss curve at a few points.
`--twice`: run it a second time from the same seeds and require bit-identical parameters, Adam moments and epoch
losses (31 960 optimizer steps, ~0.5 M launches: a soak for ordering bugs between launches -- write-through
stores, deferred Adam workgroups, the gather riding in the trailing launch -- which would show as run-to-run
differences)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import make_trainer, synth_demo

data = synth_demo(0, 10, 1000, 197, 45)


def run():
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(data, 256, "cuda", m_world=300)
    losses = [tr.train()["mean_train_loss"]]     # warm-up epoch (module load, first gather)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    curve = {}
    for it in range(2, 801):
        r = tr.train()
        losses.append(r["mean_train_loss"])
        if it in (2, 100, 300, 301, 400, 600, 800):
            curve[it] = round(r["mean_train_loss"], 6)
    torch.cuda.synchronize()
    return tr, curve, time.perf_counter() - t0, losses


tr, curve, dt, losses = run()
n = len(tr.train_loader.dataset)
print("799 epochs (299 world + 500 joint) x %d samples: %.2f s wall  (%.0f samples/s incl. per-epoch host sync)"
      % (n, dt, 799 * n / dt))
print("loss at iteration:", curve)
print("Adam steps: WM %d, TE/MD %d; lr now %.3e" % (tr.optimizer.net_steps[2], tr.optimizer.net_steps[0], tr.optimizer.lr))
if "--twice" in sys.argv:
    arenas = [t.clone() for t in (tr.engine.params, tr.engine.exp_avg, tr.engine.exp_avg_sq)]
    tr2, _, dt2, losses2 = run()
    same = all(torch.equal(a, b) for a, b in zip(arenas, (tr2.engine.params, tr2.engine.exp_avg, tr2.engine.exp_avg_sq)))
    print("second run: %.2f s; parameters and Adam moments bit-identical: %s; all 800 epoch losses identical: %s"
          % (dt2, same, losses == losses2))
    if not (same and losses == losses2):
        sys.exit(1)
