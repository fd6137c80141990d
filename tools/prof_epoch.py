"""Host cost of the trainer's epoch loop (torch_models.TrainModel.run_epoch) per optimizer step, world and joint
phase, against the GPU time of a step: is the Python loop ahead of the GPU?"""
import contextlib, io, os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import make_trainer, synth_demo
data = synth_demo(0, 10, 1000, 197, 45)
SMALL = "--small" in sys.argv          # 2 x 512 stacks (about the trainer's own defaults) instead of BASELINE's 4 x 1024
for m_world, name in ((1000, "world"), (0, "joint")):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(data, 256, "cuda", m_world=m_world, **(dict(width=512, depth=2) if SMALL else {}))
    for _ in range(3):
        tr.train()
    torch.cuda.synchronize()
    n = len(tr.train_loader)
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (10 * n) * 1e6
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.train()
    pr.disable()
    print("== %s phase: %.1f us wall per optimizer step over whole epochs (%d steps per epoch, one host sync per epoch)" % (name, wall, n))
    st = pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime")
    st.print_stats(12)
