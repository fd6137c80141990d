"""BASELINE.json configs[2] at FULL length against the oracle: max_iter 800 (300 world-model epochs, then
500 joint world-model + CVAE epochs) on the 10 x 1000-step synthetic loco demo (learnable dynamics),
batch 256, TE/MD/WM 4x1024, StepLR(50, 0.7) -- the HIP path and the CPU restatement of the reference loop
(oracle/refpath.RefTrainer: the checker) start from the same weights and consume the same eps stream.
31 960 optimizer steps each; the CPU side takes ~12-15 minutes at 16 threads, the GPU side seconds.

Two fp32 implementations of a 32 k-step Adam trajectory do not stay bit-close (summation order, ReLU kinks,
Adam's g/(|g|+eps) conditioning), so the report is per-epoch relative differences of every loss term plus,
for scale, the oracle's own run-to-run spread under a different eps seed over the last joint epochs.

    python tools/full_run_parity.py [--epochs 800 --world 300 --spread-epochs 40] > report.json
This is a measurement tool, not a test and not the product path (it imports oracle/ as the checker)."""
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
import torch  # noqa: E402

from oracle import refpath as R  # noqa: E402
from util import make_trainer  # noqa: E402

TERMS = ("total", "loss_a", "loss_kl", "loss_s", "loss_cyc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=800)
    ap.add_argument("--world", type=int, default=300)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--spread-epochs", type=int, default=0,
                    help="also continue the oracle's joint phase from epoch `world` under a different eps seed "
                         "for this many epochs (run-to-run spread of the terms; 0 = skip)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    arch = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    data = R.synth_demo(0, 10, 1000, 197, 45, kind="dynamics")
    sd = R.init_state_dict(arch, seed=1)
    X, Y = R.build_windows(data)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(arch, data, 256, m_world=a.world, device="cuda", lr_step=50, eps_fn=R.eps_stream(2, 32))
    tr.model.load_state_dict(sd)
    t0 = time.perf_counter()
    ours = []
    for _ in range(a.epochs):
        r = tr.train()
        ours.append([r["mean_train_loss"]] + list(tr.last_loss_terms[1:]))
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    ref = R.RefTrainer(arch, sd, X, Y, 256, max_iter_world_model=a.world, lr_step=50, eps_fn=R.eps_stream(2, 32))
    t0 = time.perf_counter()
    theirs, mid_sd = [], None
    for e in range(a.epochs):
        if e == a.world and a.spread_epochs:
            mid_sd = {k: v.detach().clone() for k, v in ref.model.state_dict().items()}
        r = ref.step()
        theirs.append([r["mean_train_loss"]] + [ref.last_terms[k] for k in TERMS[1:]])
        if (e + 1) % 50 == 0:
            print("oracle epoch %d / %d  (%.0f s)" % (e + 1, a.epochs, time.perf_counter() - t0), file=sys.stderr, flush=True)
    t_cpu = time.perf_counter() - t0

    def rel(a_, b_):
        return abs(a_ - b_) / max(abs(b_), 1e-12)

    report = {"config": "BASELINE configs[2]: %d epochs (%d world + %d joint), 10x1000 demo, B=256, 4x1024, StepLR(50,0.7), same init, same eps"
                        % (a.epochs, a.world, a.epochs - a.world),
              "optimizer_steps": a.epochs * len(tr.train_loader), "gpu_seconds": t_gpu, "cpu_seconds": t_cpu, "cpu_threads": a.threads,
              "epochs": {}, "max_rel_diff": {}}
    active = {True: ("total", "loss_s"), False: ("total", "loss_a", "loss_kl", "loss_cyc")}
    for t, name in enumerate(TERMS):
        worst = 0.0
        for e in range(a.epochs):
            if name in active[e < a.world] and abs(theirs[e][t]) > 1e-4:      # (terms at the fp32 noise floor are skipped)
                worst = max(worst, rel(ours[e][t], theirs[e][t]))
        report["max_rel_diff"][name] = worst
    for e in sorted({1, 2, 10, 50, 100, a.world, a.world + 1, a.world + 10, a.world + 100, a.epochs - 100, a.epochs}):
        if 1 <= e <= a.epochs:
            report["epochs"][str(e)] = {"hip": dict(zip(TERMS, ours[e - 1])), "oracle": dict(zip(TERMS, theirs[e - 1]))}
    last = range(max(a.world, a.epochs - 20), a.epochs)
    report["mean_rel_diff_last_20_epochs"] = {
        name: sum(rel(ours[e][t], theirs[e][t]) for e in last) / len(last) for t, name in enumerate(TERMS)
        if name in active[False]}
    if mid_sd is not None:
        alt = R.RefTrainer(arch, mid_sd, X, Y, 256, max_iter_world_model=0, lr_step=50, eps_fn=R.eps_stream(777, 32))
        for g in alt.opt.param_groups:                      # same lr as the main run has at that epoch
            g["lr"] = R.lr_for_epoch(a.world + 1)
        spread = []
        for e in range(a.spread_epochs):
            r = alt.step()
            spread.append([r["mean_train_loss"]] + [alt.last_terms[k] for k in TERMS[1:]])
        k = a.spread_epochs - 1
        report["oracle_other_eps_seed_vs_oracle_at_joint_epoch_%d" % a.spread_epochs] = {
            name: rel(spread[k][t], theirs[a.world + k][t]) for t, name in enumerate(TERMS) if name in active[False]}
        report["hip_vs_oracle_at_the_same_epoch"] = {
            name: rel(ours[a.world + k][t], theirs[a.world + k][t]) for t, name in enumerate(TERMS) if name in active[False]}
        report["note_spread"] = ("the other-seed run restarts Adam's moments at the phase switch exactly like the main run "
                                 "(lazy state) but from the oracle's own world-phase weights")
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
