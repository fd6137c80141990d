"""BASELINE.json configs[2] end to end: the full 800-iteration run (300 world-model epochs, then 500
joint world-model + CVAE epochs) on the 10 x 1000-step synthetic loco demo, batch 256, TE/MD/WM 4x1024,
through the trainer class the CLI uses.  Prints wall-clock and the loss curve at a few points."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import make_trainer, synth_demo

data = synth_demo(0, 10, 1000, 197, 45)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(data, 256, "cuda", m_world=300)
tr.train()                                   # warm-up epoch (module load, first gather)
torch.cuda.synchronize()
t0 = time.perf_counter()
curve = {}
for it in range(2, 801):
    r = tr.train()
    if it in (2, 100, 300, 301, 400, 600, 800):
        curve[it] = round(r["mean_train_loss"], 6)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(tr.train_loader.dataset)
print("799 epochs (299 world + 500 joint) x %d samples: %.2f s wall  (%.0f samples/s incl. per-epoch host sync)"
      % (n, dt, 799 * n / dt))
print("loss at iteration:", curve)
print("Adam steps: WM %d, TE/MD %d; lr now %.3e" % (tr.optimizer.net_steps[2], tr.optimizer.net_steps[0], tr.optimizer.lr))
