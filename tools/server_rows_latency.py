"""Host observation(s) -> host action(s) through the resident rollout server at 1-4 rows per request, against the launch path
(`infer` on pinned buffers) on the same rows; default stacks and the bench's 4x1024 stacks, with and without a helper stack.
    python tools/server_rows_latency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
from oracle import refpath as R          # noqa: E402  (weights / demo generator only: a measuring tool, not the product)
from util import make_trainer            # noqa: E402


def median_us(fn, n=400, warm=50):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(t)), float(np.percentile(t, 90))


for name, arch in (("default 256x2/512x3", R.make_arch(197, 45)),
                   ("4x1024", R.make_arch(197, 45, te=(1024, 4), md=(1024, 4), wm=(1024, 4)))):
    data = R.synth_demo(0, 2, 40, 197, 45, kind="dynamics")
    tr = make_trainer(arch, data, 8, device="cuda")
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    eng = tr.engine
    X, _ = R.build_windows(data)
    obs = torch.from_numpy(np.asarray(X)).float()[:, 0, :]
    for rows in (1, 2, 3, 4):
        o = obs[:rows].contiguous()
        pin_in, pin_out = torch.zeros(rows, 394).pin_memory(), torch.zeros(rows, 45).pin_memory()
        dev_in = torch.zeros(rows, 394, device="cuda")

        def launch():
            pin_in.copy_(o)
            dev_in.copy_(pin_in, non_blocking=True)
            a, _, _ = eng.infer(dev_in, noise=True, seed=1, offset=7, want_s2=False)
            pin_out.copy_(a, non_blocking=True)
            torch.cuda.synchronize()
        lm, lp = median_us(launch)
        eng.rollout_server_start(idle_ms=2000.0, lifetime_s=60.0)
        try:
            on = o.numpy()
            if rows == 1:
                sm, sp = median_us(lambda: eng.rollout_server_infer(on[0], noise=True, seed=1, offset=7))
            else:
                sm, sp = median_us(lambda: eng.rollout_server_infer_rows(on, noise=True, seed=1, offset=7))
            scope = eng.rollout_server_scope()
        finally:
            eng.rollout_server_stop()
        print("%-22s rows %d  server (%s, through Python) %6.1f us median %6.1f p90 | launch path, pinned in / out %6.1f us median %6.1f p90"
              % (name, rows, scope, sm, sp, lm, lp), flush=True)
