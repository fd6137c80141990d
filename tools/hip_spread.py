"""The HIP path against ITSELF over BASELINE.json configs[2] at full length (800 epochs = 300 world + 500 joint,
B = 256, 4x1024, StepLR(50, 0.7)): the setup of tools/full_run_parity.py (same data, same eps stream, same initial
weights as the oracle runs), repeated with every initial weight moved by one unit in the last place (a random sign
per element, another seed per run).  Together with tools/oracle_spread.py (the oracle against itself, same
perturbation) this is the scale for "HIP vs oracle" in profiles/rNN_full_run_parity.json: if the runs of one
implementation spread as widely as the two implementations differ, the difference is the trajectory's
sensitivity to rounding, not a bias of either side.

    python tools/hip_spread.py [--runs 4] > profiles/rNN_hip_spread.json
A measurement tool (imports oracle/ only to regenerate the same inputs as the checker's runs)."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import refpath as R  # noqa: E402
from util import make_trainer  # noqa: E402

TERMS = ("total", "loss_a", "loss_kl", "loss_s", "loss_cyc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=800)
    ap.add_argument("--world", type=int, default=300)
    a = ap.parse_args()
    arch = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    data = R.synth_demo(0, 10, 1000, 197, 45, kind="dynamics")
    runs, secs = [], []
    for r in range(a.runs):
        sd = R.init_state_dict(arch, seed=1)
        if r > 0:                                   # run 0: the unperturbed initial state (= full_run_parity's HIP run)
            rng = np.random.default_rng(12345 + r - 1)       # (run 1: the perturbation tools/oracle_spread.py --run ulp uses)
            for k, v in sd.items():
                if k.endswith("weight"):
                    sign = torch.from_numpy(rng.integers(0, 2, size=tuple(v.shape)).astype(np.float32) * 2 - 1)
                    sd[k] = (v.double() * (1.0 + sign.double() * 2.0 ** -23)).float()
        with contextlib.redirect_stdout(io.StringIO()):
            tr = make_trainer(arch, data, 256, m_world=a.world, device="cuda", lr_step=50, eps_fn=R.eps_stream(2, 32))
        tr.model.load_state_dict(sd)
        t0, out = time.perf_counter(), []
        for _ in range(a.epochs):
            res = tr.train()
            out.append([res["mean_train_loss"]] + list(tr.last_loss_terms[1:]))
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
        runs.append(out)

    def rel(x, y):
        return abs(x - y) / max(abs(y), 1e-12)
    n = a.epochs
    rep = {"what": "HIP path vs itself: same data / eps / schedule, initial weights one ulp apart (run 0 unperturbed)",
           "runs": a.runs, "seconds_per_run": secs, "epochs": {}, "vs_run0": {}}
    for e in sorted({1, 2, 10, 50, 100, a.world, a.world + 1, a.world + 10, a.world + 100, n - 100, n}):
        if 1 <= e <= n:
            rep["epochs"][str(e)] = {"total_per_run": [runs[r][e - 1][0] for r in range(a.runs)]}
    active = {True: ("total", "loss_s"), False: ("total", "loss_a", "loss_kl", "loss_cyc")}
    last = range(max(a.world, n - 20), n)
    for r in range(1, a.runs):
        worst = {}
        for t, name in enumerate(TERMS):
            w = 0.0
            for e in range(n):
                if name in active[e < a.world] and abs(runs[0][e][t]) > 1e-4:
                    w = max(w, rel(runs[r][e][t], runs[0][e][t]))
            worst[name] = w
        rep["vs_run0"]["run%d" % r] = {
            "max_rel_diff": worst,
            "mean_rel_diff_last_20_epochs": {name: sum(rel(runs[r][e][t], runs[0][e][t]) for e in last) / len(last)
                                             for t, name in enumerate(TERMS) if name in active[False]},
            "signed_rel_diff_total_last_20_epochs": sum((runs[r][e][0] - runs[0][e][0]) / runs[0][e][0] for e in last) / len(last)}
    rep["mean_total_last_20_epochs_per_run"] = [sum(runs[r][e][0] for e in last) / len(last) for r in range(a.runs)]
    rep["world_loss_at_switch_per_run"] = [runs[r][a.world - 1][0] for r in range(a.runs)]
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
