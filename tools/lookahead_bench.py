"""Time optimizer steps of the multi-step unroll (lookahead L, tpv:367-428) at the benchmark
sizes (B=256, TE/MD/WM 4x1024, Db=197, Da=45) and compare with L independent lookahead-1 steps.
Usage (GPU box): python tools/lookahead_bench.py [--lookahead 2] [--steps 200]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lookahead", type=int, default=2)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    from synth_demo import make_trainer, synth_demo
    data = synth_demo(0, 10, 1000, 197, 45)
    out = {}
    for L in (1, a.lookahead):
        with contextlib.redirect_stdout(io.StringIO()):
            tr = make_trainer(data, a.batch, "cuda", extra={"lookahead": L})
        eng = tr.engine
        ds = tr.train_loader.dataset
        eng.bind_dataset(*ds.device_arrays(eng.device))
        full = len(ds) // a.batch
        loss = torch.zeros(5, device="cuda")
        for name in ("world", "joint"):
            w = name == "world"
            tr.model.set_learnable_task_encoder(not w)
            tr.model.set_learnable_motor_decoder(not w)
            tr.model.set_learnable_world_model(w)
            tr.read_loss_fn_coeff(world=w)
            phase, nets = tr.phase()

            def run(n, start):
                for i in range(n):
                    g = (start + i) % full
                    sp = tr.step_params(nets, a.batch, True)
                    sp.rng_seed, sp.rng_offset = 7, (start + i) * 65536
                    eng.train_step(phase, g * a.batch, a.batch, sp, loss_out=loss)
            run(20, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(a.steps, 20)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            out["L%d_%s_us_per_step" % (L, name)] = dt * 1e6
            out["L%d_%s_loss" % (L, name)] = float(loss[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
