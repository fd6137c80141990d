"""Rollout-path latency: PhysicsVAE forward at control-loop batch sizes (rmt:742-771)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import refpath as R
from util import make_trainer
for name, arch in (("default 256x2/512x3/1024x2", R.make_arch(197, 45)),
                   ("4x1024", R.make_arch(197, 45, te=(1024, 4), md=(1024, 4), wm=(1024, 4)))):
    data = R.synth_demo(0, 2, 50, 197, 45)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(arch, data, 32, device="cuda")
    eng = tr.engine
    for rows in (1, 4, 32):
        obs = torch.randn(rows, 394, device="cuda")
        for want_s2 in (False, True):
            out = None
            for _ in range(20):
                out = eng.infer(obs, want_s2=want_s2, out=out)
            torch.cuda.synchronize()
            # back-to-back calls: wall clock = max(host cost of a call, device time of a call)
            t0 = time.perf_counter()
            n = 300
            for _ in range(n):
                out = eng.infer(obs, want_s2=want_s2, out=out)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e6
            # one call at a time: issue -> result on the device (what the control loop waits for)
            lat = []
            for _ in range(50):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                out = eng.infer(obs, want_s2=want_s2, out=out)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t2) * 1e6)
            lat.sort()
            print("%-28s rows %2d  %s : %6.1f us / call back to back (host %5.1f), %6.1f us single-call latency"
                  % (name, rows, "TE+MD+WM" if want_s2 else "TE+MD   ", wall, (t1 - t0) / n * 1e6, lat[len(lat) // 2]))
