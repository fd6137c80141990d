"""How far do two fp32 runs of the SAME training trajectory drift apart when nothing differs but rounding?
The CPU restatement of the reference loop (oracle/refpath.RefTrainer: the checker, stock torch CPU ops) runs
BASELINE.json configs[2] at full length twice -- same data, same eps stream, same initial weights, except that in
the second run every weight of the initial state is moved by one unit in the last place (a random sign per element:
w * (1 +- 2^-23)).  A 32 k-step Adam trajectory amplifies that (ReLU kinks flip, Adam's g / (sqrt(v) + eps) is
ill-conditioned where |g| is small), so the per-epoch relative difference of the loss terms between the two runs is
the scale against which "HIP path vs oracle" (tools/full_run_parity.py, profiles/r03_full_run_parity.json) has to
be read: an implementation cannot be held closer to the oracle than the oracle is to itself.

    python tools/oracle_spread.py --run base --out /tmp/base.json [--epochs 800 --world 300 --threads 4]
    python tools/oracle_spread.py --run ulp  --out /tmp/ulp.json
    python tools/oracle_spread.py --run ulp  --seed 12346 --threads 16 --out /tmp/ulp2.json     (another sibling)
    python tools/oracle_spread.py --compare /tmp/base.json /tmp/ulp.json > profiles/rNN_oracle_spread.json
    python tools/oracle_spread.py --summary /tmp/base.json /tmp/ulp*.json > profiles/rNN_oracle_spread.json
--summary lists every run (threads, perturbation seed, final terms, mean of the last 20 epochs), the spread inside each
thread count and the offset BETWEEN thread counts (torch's CPU GEMM splits its reductions by thread count, so two
thread counts are two different fp32 implementations of the same loop).
CPU only (no GPU, no library): a measurement tool around the checker, not the product path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import refpath as R  # noqa: E402

TERMS = ("total", "loss_a", "loss_kl", "loss_s", "loss_cyc")


def run(a):
    torch.set_num_threads(a.threads)
    arch = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    data = R.synth_demo(0, 10, 1000, 197, 45, kind="dynamics")
    sd = R.init_state_dict(arch, seed=1)
    if a.run == "ulp":
        rng = np.random.default_rng(a.seed)
        for k, v in sd.items():
            if k.endswith("weight"):
                sign = torch.from_numpy(rng.integers(0, 2, size=tuple(v.shape)).astype(np.float32) * 2 - 1)
                sd[k] = (v.double() * (1.0 + sign.double() * 2.0 ** -23)).float()
    X, Y = R.build_windows(data)
    ref = R.RefTrainer(arch, sd, X, Y, 256, max_iter_world_model=a.world, lr_step=50, eps_fn=R.eps_stream(2, 32))
    out, t0 = [], time.perf_counter()
    for e in range(a.epochs):
        r = ref.step()
        out.append([r["mean_train_loss"]] + [ref.last_terms[k] for k in TERMS[1:]])
        if (e + 1) % 25 == 0:
            print("%s epoch %d / %d  (%.0f s)" % (a.run, e + 1, a.epochs, time.perf_counter() - t0), file=sys.stderr, flush=True)
            json.dump({"run": a.run, "seed": a.seed if a.run == "ulp" else None, "epochs": a.epochs, "world": a.world,
                       "threads": a.threads, "terms": out}, open(a.out, "w"))
    json.dump({"run": a.run, "seed": a.seed if a.run == "ulp" else None, "epochs": a.epochs, "world": a.world,
               "threads": a.threads, "terms": out, "seconds": time.perf_counter() - t0, "host_cpus": os.cpu_count()},
              open(a.out, "w"))


def compare(pa, pb):
    A, B = json.load(open(pa)), json.load(open(pb))
    n, world = min(len(A["terms"]), len(B["terms"])), A["world"]
    a, b = A["terms"], B["terms"]

    def rel(x, y):
        return abs(x - y) / max(abs(y), 1e-12)
    active = {True: ("total", "loss_s"), False: ("total", "loss_a", "loss_kl", "loss_cyc")}
    rep = {"what": "oracle vs oracle: same data / eps / schedule, initial weights one ulp apart (tools/oracle_spread.py)",
           "epochs_compared": n, "world_epochs": world, "threads": [A["threads"], B["threads"]], "max_rel_diff": {}, "epochs": {}}
    for t, name in enumerate(TERMS):
        worst = 0.0
        for e in range(n):
            if name in active[e < world] and abs(b[e][t]) > 1e-4:
                worst = max(worst, rel(a[e][t], b[e][t]))
        rep["max_rel_diff"][name] = worst
    for e in sorted({1, 2, 10, 50, 100, world, world + 1, world + 10, world + 100, n - 100, n}):
        if 1 <= e <= n:
            rep["epochs"][str(e)] = {"base": dict(zip(TERMS, a[e - 1])), "one_ulp": dict(zip(TERMS, b[e - 1])),
                                     "rel_diff_total": rel(a[e - 1][0], b[e - 1][0])}
    last = range(max(world, n - 20), n)
    if len(last):
        rep["mean_rel_diff_last_20_epochs"] = {name: sum(rel(a[e][t], b[e][t]) for e in last) / len(last)
                                               for t, name in enumerate(TERMS) if name in active[False]}
    print(json.dumps(rep, indent=1))


def summary(paths):
    """Every complete run side by side: what the oracle's own trajectory spreads to at the end of configs[2]."""
    runs = []
    for p in paths:
        d = json.load(open(p))
        if len(d["terms"]) < d["epochs"]:
            continue
        t = d["terms"]
        last = t[-20:]
        runs.append({"file": os.path.basename(p), "run": d["run"], "perturbation_seed": d.get("seed"), "threads": d["threads"],
                     "host_cpus": d.get("host_cpus"), "seconds": d.get("seconds"),
                     "world_loss_at_switch": t[d["world"] - 1][3],
                     "final": dict(zip(TERMS, t[-1])),
                     "mean_last_20": {n: sum(r[i] for r in last) / len(last) for i, n in enumerate(TERMS)}})
    by_threads = {}
    for r in runs:
        by_threads.setdefault(r["threads"], []).append(r)

    def stats(vals):
        m = sum(vals) / len(vals)
        sd = (sum((v - m) ** 2 for v in vals) / max(len(vals) - 1, 1)) ** 0.5
        return {"n": len(vals), "mean": m, "std": sd, "min": min(vals), "max": max(vals), "rel_range": (max(vals) - min(vals)) / m}
    rep = {"what": "oracle (oracle/refpath.RefTrainer) against itself over BASELINE configs[2] at full length: same data / eps / "
                   "schedule; siblings differ by a one-ulp perturbation of the initial weights and by torch's thread count",
           "runs": runs, "final_total_by_threads": {}, "mean_last_20_total_by_threads": {}}
    for th, rs in sorted(by_threads.items()):
        rep["final_total_by_threads"][str(th)] = stats([r["final"]["total"] for r in rs])
        rep["mean_last_20_total_by_threads"][str(th)] = stats([r["mean_last_20"]["total"] for r in rs])
    rep["final_total_all_runs"] = stats([r["final"]["total"] for r in runs])
    rep["mean_last_20_total_all_runs"] = stats([r["mean_last_20"]["total"] for r in runs])
    ths = sorted(by_threads)
    if len(ths) >= 2:
        m = rep["mean_last_20_total_by_threads"]
        lo, hi = str(ths[0]), str(ths[-1])
        rep["thread_count_effect_on_mean_last_20_total"] = {
            "threads": [ths[0], ths[-1]], "rel_offset": (m[hi]["mean"] - m[lo]["mean"]) / m[lo]["mean"],
            "pooled_std_rel": ((m[hi]["std"] ** 2 + m[lo]["std"] ** 2) / 2) ** 0.5 / m[lo]["mean"]}
    print(json.dumps(rep, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", choices=["base", "ulp"])
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--summary", nargs="+")
    ap.add_argument("--seed", type=int, default=12345, help="seed of the one-ulp perturbation (--run ulp)")
    ap.add_argument("--epochs", type=int, default=800)
    ap.add_argument("--world", type=int, default=300)
    ap.add_argument("--threads", type=int, default=4)
    a = ap.parse_args()
    if a.summary:
        summary(a.summary)
    elif a.compare:
        compare(*a.compare)
    else:
        run(a)


if __name__ == "__main__":
    main()
