"""bench.py -- training samples/s of the PhysicsVAE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--phase joint|world] [--config c2|c5]

Launched plainly with --gpus N > 1 it starts the N ranks itself (one process per GPU, RCCL over
xGMI; on a box with fewer GPUs than ranks the ranks share a GPU over gloo -- a functional check
of the same code path, reported as such).  Launched by `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` it is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment).

One "step" = one optimizer step over one minibatch of synthetic demonstrations: minibatch
gather from the HBM-resident demo set -> MLP forward -> losses -> backward -> Adam.
Workload = BASELINE.json configs[1..2] sizes: synthetic loco demo (10 episodes x 1000 steps,
dim_state_body 197, dim_action 45), batch 256 per GPU, TE/MD/WM = 4x1024.  The headline `value`
is the JOINT phase (world-model + CVAE step: encoder -> sampler -> decoder -> frozen world model,
ELBO + cycle loss, TE and MD learn), the step BASELINE.json's metric names; the world-model-only
phase of configs[1] is reported beside it as `world_value` with its own `world_roofline`.
N > 1: data-parallel, 256 rows per GPU (global batch N*256, weak scaling), gradient SUM
all-reduce over RCCL, replicated Adam.

Timing protocol: W warm-up steps, then R = 3 timed regions of max(K, 200) steps each, every
region bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks per region;
`value` / `ms_per_step` are the MEDIAN region (>= 200 steps, median of 3).

`roofline` (dominant kernel of the timed phase): algorithmic flops per launch / average launch
duration.  Two clocks are reported: HIP events stamped at the kernel's own start / end on the
launch stream in an instrumented pass (`avg_launch_us`), and -- at N = 1 -- rocprofv3
--kernel-trace of this same command run as a child process (`avg_launch_us_rocprof`); `frac` uses
the rocprofv3 duration when it is available (the larger of the two).  `traffic` = HBM bytes per
launch from separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md HBM
section), collected live by the same child runs.  `cpu_baseline` is the oracle's restatement of
the reference loop (oracle/refpath.py: the checker, timed here, never the product path) on this
host's cores: a short thread-count sweep, then one warm-up epoch and three timed epochs at the
best count.
"""
import argparse
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk
MIN_TIMED_STEPS, REPEATS = 200, 3
METRIC = "train samples/sec (world-model+VAE step), loco demo, batch 256, 1/2/4/8 GPU"

# profiler categories of the library (include/pvae.h) and the kernels rocprofv3 files under them
CAT_NAMES = {
    0: "forward layer (gemm_splitk_ws_kernel<P_ROW>: 32x32 / 64x32 output tiles)",
    5: "narrow forward layer (gemm_splitk_reg16_kernel<P_ROW>: 16x16 tiles; output layers, fused-loss layer)",
    1: "input gradient alone (gemm_splitk_ws_kernel<P_COL> / gemm_splitk_reg16_kernel<P_COL>)",
    2: "trailing weight gradient (wgrad_pair_kernel / gemm_wgrad_reg_kernel)",
    3: "bwd_pair_kernel (input gradient || weight gradient of one layer, + deferred Adam of the layer before)",
}
CAT_MATCH = {
    0: ("gemm_splitk_ws_kernel<true", "gemm_splitk_ws64_kernel<true", "gemm_splitk_ws_pro_kernel", "gemm_splitk_reg_kernel<true"),
    5: ("gemm_splitk_reg16_kernel<true",),
    1: ("gemm_splitk_ws_kernel<false", "gemm_splitk_ws64_kernel<false", "gemm_splitk_reg16_kernel<false",
        "gemm_splitk_reg_kernel<false"),
    2: ("wgrad_pair_kernel", "gemm_wgrad_reg_kernel"),
    3: ("bwd_pair_kernel", "bwd_pair64_kernel"),        # (64x32 input-gradient tiles at >= 512 rows)
}


def algorithmic_flops_per_sample(Db, Da, Z, W, d):
    """F(I,W,d,O) = I*W + (d-1)*W^2 + W*O MACs; flops = 2*MAC."""
    def F(i, o):
        return i * W + (d - 1) * W * W + W * o
    f_te, f_md, f_wm = F(2 * Db, 2 * Z), F(Db + Z, Da), F(Db + Da, Db)
    world = 3 * f_wm - (Db + Da) * W
    hid = (d - 1) * W * W
    joint = (f_te + f_md + f_wm) + (hid + W * Db + Da * W) + f_md + (hid + W * Da + Z * W) + f_te + (hid + W * 2 * Z)
    return 2 * world, 2 * joint


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--phase", choices=["world", "joint"], default="joint",
                    help="phase timed as `value` (default joint: the world-model + VAE step)")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default 256; 512 for --config c5)")
    ap.add_argument("--config", choices=["c2", "c5"], default="c2",
                    help="c2 = BASELINE configs[1..3] sizes (default); c5 = configs[4]: 1e6 transitions, "
                         "dim_state_body 400, dim_action 90, 512 rows per GPU")
    ap.add_argument("--exchange", choices=["inline", "bucketed", "sharded", "p2p", "p2p_push"], default=None,
                    help="N > 1: how the gradient is exchanged (the trainer's dp_exchange): RCCL all-reduce in line / in 6 MiB "
                         "overlapped buckets / RCCL reduce-scatter + sharded Adam + all-gather / the direct all-pairs exchange "
                         "over peer-mapped arenas (no RCCL); default: the library's own schedule")
    ap.add_argument("--no-sweep", action="store_true", help="N > 1: skip the back-to-back comparison of all exchange forms")
    ap.add_argument("--no-autotune", action="store_true",
                    help="N > 1 without --exchange: keep the library's default schedule instead of choosing the exchange form by "
                         "a short calibration before the warm-up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary-phase, roofline and rocprofv3 passes")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child runs (kernel trace + PMC)")
    ap.add_argument("--inner", action="store_true",
                    help="(internal) the child run that rocprofv3 wraps: one timed region of exactly --steps steps "
                         "of --phase, nothing else")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no rendezvous in the environment
# ---------------------------------------------------------------------------------------------
def self_launch(a):
    import torch
    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    shared = ndev < a.gpus
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if shared:
            # fewer GPUs than ranks (a 1-GPU box): the ranks share devices and exchange over gloo -- RCCL
            # refuses two ranks on one device.  Same sharding / reduction / Adam code, not a scaling number.
            env.update(PVAE_LOCAL_DEVICE=str(r % max(ndev, 1)), PVAE_DIST_BACKEND="gloo", PVAE_BENCH_SHARED_GPU="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


# ---------------------------------------------------------------------------------------------
# rocprofv3 child runs (N = 1): kernel durations + HBM traffic + MFMA busy for the timed phase
# ---------------------------------------------------------------------------------------------
def _csv_rows(path):
    import csv
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def _find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def _cat_of(name):
    for c, pats in CAT_MATCH.items():
        if any(p in name for p in pats):
            return c
    return None


def rocprof_passes(a, phase, budget_s=150):
    """-> {cat: {"avg_us", "calls", "hbm_fetch_bytes", "hbm_write_bytes", "mfma_busy"}} or {"error": ...}.
    One `rocprofv3 --kernel-trace --stats` pass and three `--pmc` passes (never combined with the
    hip/hsa trace domains) over `python bench.py --inner --phase <phase>`."""
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="pvae_rocprof_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--inner", "--phase", phase, "--config", a.config,
             "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-extra"]
    if a.batch:
        child += ["--batch", str(a.batch)]
    env = dict(os.environ, TMPDIR="/tmp")
    passes = [("trace", ["--kernel-trace", "--stats"]),
              ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"]),
              ("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"]),
              ("mfma", ["--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "--kernel-trace"])]
    out, t_end, notes = {}, time.time() + budget_s, []
    for tag, flags in passes:
        left = t_end - time.time()
        if left < 20:
            notes.append("%s pass skipped (time budget)" % tag)
            continue
        d = os.path.join(tmp, tag)
        try:
            subprocess.run([exe] + flags + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child, cwd="/tmp",
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left, check=True)
        except Exception as exc:                                   # noqa: BLE001
            notes.append("%s pass failed: %s" % (tag, type(exc).__name__))
            continue
        if tag == "trace":
            f = _find(d, "kernel_stats.csv")
            if not f:
                notes.append("trace pass wrote no kernel_stats.csv")
                continue
            for r in _csv_rows(f):
                c = _cat_of(r["Name"])
                if c is None:
                    continue
                e = out.setdefault(c, {"calls": 0, "total_ns": 0.0})
                e["calls"] += int(r["Calls"])
                e["total_ns"] += float(r["TotalDurationNs"])
        else:
            f = _find(d, "counter_collection.csv")
            if not f:
                notes.append("%s pass wrote no counter_collection.csv" % tag)
                continue
            acc = {}
            for r in _csv_rows(f):
                c = _cat_of(r["Kernel_Name"])
                if c is None:
                    continue
                k = (c, r["Counter_Name"])
                s = acc.setdefault(k, [0.0, 0])
                s[0] += float(r["Counter_Value"])
                s[1] += 1
            for (c, name), (tot, n) in acc.items():
                out.setdefault(c, {})[name] = tot / max(n, 1)
    shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for c, e in out.items():
        r = {}
        if e.get("calls"):
            r["avg_us"] = e["total_ns"] / e["calls"] / 1e3
            r["calls"] = e["calls"]
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) reports half the bytes of wide coalesced reads on gfx950
        if "FETCH_SIZE" in e:
            r["hbm_fetch_bytes"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            r["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and r.get("avg_us"):
            # busy cycles summed over the 1024 SIMDs; launch duration at the 2.4 GHz peak clock
            r["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (r["avg_us"] * 1e-6 * 2.4e9)
        res[c] = r
    if notes:
        res["notes"] = notes
    return res


# ---------------------------------------------------------------------------------------------
# CPU baseline: thread sweep of the oracle's restatement of the reference loop
# ---------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def p2p_timeout_ms():
    """How long a peer-mapped wait may take in this run before it gives up: a fraction of a second with one rank per GPU
    (a peer that never answers must not cost the library's default 20 s per wait) -- but ranks that SHARE a GPU are
    time-sliced, one step of 8 such ranks takes ~230 ms and a wait can sit out several slices (the round-4 line and one of
    round 5's lost their `exchange_sweep` entries to a 250 ms limit there)."""
    return "5000" if os.environ.get("PVAE_BENCH_SHARED_GPU") == "1" else "250"


def physical_cores():
    """(physical cores, sockets) of this host from /proc/cpuinfo's (physical id, core id) pairs; (None, None) if unreadable."""
    try:
        cores, phys, cur = set(), set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (t.strip() for t in line.split(":", 1))
                cur[k] = v
            elif not line.strip() and cur:
                if "physical id" in cur and "core id" in cur:
                    cores.add((cur["physical id"], cur["core id"]))
                    phys.add(cur["physical id"])
                cur = {}
        return (len(cores) or None), (len(phys) or None)
    except OSError:
        return None, None


def cpu_baseline(a, sd, Db, Da, Z, W, D, phase, synth_demo):
    """the oracle's restatement of the reference loop (per-sample Dataset + DataLoader collate, full forward
    incl. value branch, torch.optim.Adam, per-batch .item()) on this host's cores, same dataset shape / batch / stacks:
    a short sweep picks the thread count, then ONE warm-up epoch and >= 3 timed epochs at that count (bounded: ~10 k
    samples per epoch, ~1 s each at the best count on the boxes seen so far).  Also the 1-thread and default-thread rates
    of the sweep.  `cores` = threads used for `value`; the host's physical core count is reported beside it."""
    import torch
    from oracle import refpath as R        # the checker, timed as the CPU baseline -- this leg only
    arch = R.make_arch(Db, Da, latent=Z, te=(W, D), md=(W, D), wm=(W, D))
    data = synth_demo(0, 10, 1000, Db, Da)                       # the configs[1-2] demo set (9 990 windows)
    X, Y = R.build_windows(data)
    default_threads = torch.get_num_threads()
    sweep = sorted({t for t in (1, 8, 16, 32, 64, default_threads) if 1 <= t <= max(default_threads, 1)})
    per_setting_s = 2.0
    results = {}
    t_all = time.perf_counter()
    M = 10 ** 9 if phase == "world" else 0

    def trainer():
        return R.RefTrainer(arch, sd, X, Y, a.batch, max_iter_world_model=M)
    for t in sweep:
        torch.set_num_threads(t)
        trc = trainer()
        trc.step(max_batches=1)                                   # warm-up (allocator, thread pool)
        n, t0 = 0, time.perf_counter()
        while True:
            trc.step(max_batches=2)
            n += 2
            dt = time.perf_counter() - t0
            if dt > per_setting_s or n >= 40:
                break
        results[t] = n * a.batch / dt
    best = max(results, key=results.get)
    # the measurement proper: whole epochs at the best thread count, bounded to ~25 s
    torch.set_num_threads(best)
    trc = trainer()
    t0 = time.perf_counter()
    trc.step()                                                    # warm-up epoch (not timed)
    warm_s = time.perf_counter() - t0
    epochs = 3 if warm_s * 3 < 25.0 else max(1, int(25.0 / max(warm_s, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(epochs):
        trc.step()
    dt = time.perf_counter() - t0
    value = epochs * len(X) / dt
    torch.set_num_threads(default_threads)
    ncores, nsock = physical_cores()
    return {"value": value, "unit": "samples/s", "cores": best, "threads_best": best,
            "physical_cores": ncores, "sockets": nsock, "host_cpus": os.cpu_count(),
            "epochs_timed": epochs, "warmup_epochs": 1, "samples_per_epoch": len(X), "seconds_timed": dt,
            "value_1thread": results.get(1), "value_default_threads": results.get(default_threads),
            "default_threads": default_threads, "sweep": {str(k): v for k, v in results.items()},
            "kind": "port", "cpu_model": cpu_model(),
            "sample": "%s phase, batch %d, the 10x1000 synthetic demo of the same dims (%d windows per epoch), "
                      "oracle/refpath.RefTrainer (stock torch CPU ops in the reference's op order: per-sample Dataset + "
                      "collate, full forward incl. value branch, torch.optim.Adam, per-batch .item()); thread sweep of "
                      "%.0f s / <= 40 minibatches per count, then 1 warm-up epoch + %d timed epochs at %d threads "
                      "(SURVEY.md 8d); %.1f s in total"
                      % (phase, a.batch, len(X), per_setting_s, epochs, best, time.perf_counter() - t_all)}


# ---------------------------------------------------------------------------------------------
def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist
    from physicsvae_amd import _lib, parallel
    rank, world, local = parallel.init_from_env()
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local

    from synth_demo import make_trainer, synth_demo      # tools/: inputs only; oracle/ is imported by the
                                                         # cpu_baseline leg and nowhere else

    import contextlib
    import io
    Z, W, D = 32, 1024, 4
    if a.config == "c2":
        Db, Da = 197, 45
        a.batch = a.batch or 256
        data = synth_demo(0, 10, 1000, Db, Da)
        workload = "BASELINE configs[1-2]: synthetic loco demo 10x1000, dim_state_body 197, dim_action 45"
    else:
        Db, Da = 400, 90
        a.batch = a.batch or 512
        data = synth_demo(0, 1, 4, Db, Da)                       # placeholder file; the real set is built below
        workload = "BASELINE configs[4]: synthetic demo 1000x1001 (1e6 transitions), dim_state_body 400, dim_action 90"
    torch.manual_seed(1)                       # normc initialisation of the model's own constructor
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(data, a.batch, dev, width=W, depth=D, latent=Z,
                          extra={"dp_exchange": a.exchange} if a.exchange else None)
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}     # for the CPU leg
    eng, dp = tr.engine, tr.dp
    ds = tr.train_loader.dataset
    if a.config == "c5":
        # 1000 episodes x 1001 steps in the packed layout the gather kernel reads (states stored
        # once: 1.6 GB + 0.36 GB fp32); generated on the device, values ~N(0,1) / clipped actions
        import numpy as np
        from physicsvae_amd.train_physics_vae import WindowDataset
        gen = torch.Generator(device=dev).manual_seed(0)
        E, T = 1000, 1001
        states = torch.randn(E * T, Db, generator=gen, device=dev)
        actions = torch.randn(E * T, Da, generator=gen, device=dev).clamp_(-3, 3)
        rows_idx = (torch.arange(E, device=dev)[:, None] * T + torch.arange(T - 1, device=dev)[None, :]).reshape(-1)
        ds = WindowDataset(np.zeros((2, Db), np.float32), np.zeros((2, Da), np.float32), np.zeros(1, np.int32))
        ds._dev = (states, actions, rows_idx.to(torch.int32))
        ds.window_row = np.empty(E * (T - 1), dtype=np.int8)     # length only (host copy not needed)
        tr.train_loader.dataset = ds
    eng.bind_dataset(*ds.device_arrays(eng.device))
    n_win = len(ds)
    lib = _lib.load()

    def set_phase(name):
        world_phase = name == "world"
        tr.model.set_learnable_task_encoder(not world_phase)
        tr.model.set_learnable_motor_decoder(not world_phase)
        tr.model.set_learnable_world_model(world_phase)
        tr.read_loss_fn_coeff(world=world_phase)
        return tr.phase()

    loss_buf = torch.zeros(5, dtype=torch.float32, device=dev)

    def run_steps(phase, nets, n, start):
        """n optimizer steps, cycling through the epoch's minibatch schedule (full batches only,
        so every timed step processes exactly batch*world samples)."""
        full = n_win // (a.batch * dp.world)
        rows_done = 0
        for i in range(n):
            g = (start + i) % full
            first, rows, grows = dp.shard(g, n_win, a.batch)
            sp = tr.step_params(nets, grows, True)
            sp.rng_seed, sp.rng_offset = 7, (start + i) * 65536 + dp.rank * 64
            nfirst, nrows, _ = dp.shard((start + i + 1) % full, n_win, a.batch)
            if not dp.collective:
                eng.train_step(phase, first, rows, sp, loss_out=loss_buf,
                               next_span=(nfirst, nrows) if tr.prefetch_gather else None)
            else:
                tr.dp_step(phase, nets, first, rows, sp, None, loss_buf, next_span=(nfirst, nrows))
            rows_done += grows
        return rows_done

    def region(phase, nets, steps, start):
        if dp.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples = run_steps(phase, nets, steps, start)
        torch.cuda.synchronize()
        if dp.world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dp.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return samples, dt

    def timed(name, steps, warmup, repeats):
        """-> (samples/s, ms/step, last loss, per-region samples/s): `repeats` regions of `steps` steps."""
        phase, nets = set_phase(name)
        run_steps(phase, nets, warmup, 0)
        rates, mss, start = [], [], warmup
        for _ in range(repeats):
            samples, dt = region(phase, nets, steps, start)
            start += steps
            rates.append(samples / dt)
            mss.append(dt / steps * 1e3)
        return statistics.median(rates), statistics.median(mss), float(loss_buf[0].item()), rates

    if a.inner:                                # the command rocprofv3 wraps: one region, one line, done
        v, ms, loss, _ = timed(a.phase, a.steps, a.warmup, 1)
        print(json.dumps({"inner": True, "phase": a.phase, "value": v, "ms_per_step": ms, "last_loss": loss}), flush=True)
        return

    autotune = None
    if dp.collective and dp.world > 1 and a.exchange is None and not a.no_autotune:
        # The exchange form is chosen by measurement, before the warm-up: every form this build and this machine offer
        # runs the same 6 + 24 steps from the same state (snapshotted, restored), the slowest rank's time counts, and a
        # form whose replicas do not stay bit-identical is out (parallel.DataParallel.autotune_exchange).
        os.environ.setdefault("PVAE_P2P_TIMEOUT_MS", p2p_timeout_ms())
        ph, nets_ = set_phase(a.phase)
        chosen, report = dp.autotune_exchange(eng, lambda n: run_steps(ph, nets_, n, 0))
        autotune = {"chosen": chosen or "library default (no candidate qualified)", "candidates": report}
    timed_steps = max(a.steps, MIN_TIMED_STEPS)
    value, ms_per_step, last_loss, rates = timed(a.phase, timed_steps, a.warmup, REPEATS)
    replicas_ok = dp.replicas_identical(eng) if dp.collective and dp.world > 1 else None

    def reference_steps(K=3):
        """N ranks against ONE process: K optimizer steps of the timed phase from a common state (this run's current
        parameters, moments zeroed, Adam counters 1..K, draws supplied) through the data-parallel step as timed above,
        and the same K global minibatches through a second engine on rank 0 that holds the whole global batch.  The
        loss of step k+1 is a function of the parameters step k produced, so a rank that reads STALE parameters of
        slices its peers own (replicas would still be bit-identical) shows up here and nowhere else.  Agreement is to
        fp32 summation order: the ranks sum N shard gradients where the single process contracts over all rows (two ranks
        on one GPU measure 7e-8 on the losses and 3e-6 on the update; the bounds are 1e-5 / 1e-3)."""
        from physicsvae_amd.engine import HipEngine
        phase, nets = set_phase(a.phase)
        snap = [t.clone() for t in (eng.params, eng.exp_avg, eng.exp_avg_sq)]
        counts = dict(tr.optimizer.net_steps)
        eng.invalidate_staging()
        eng.exp_avg.zero_()
        eng.exp_avg_sq.zero_()
        B = a.batch * dp.world
        gen = torch.Generator(device="cpu").manual_seed(1234)
        eps_all = torch.randn(K, B, Z, generator=gen).to(dev)                 # identical on every rank
        losses = torch.zeros(K, 5, dtype=torch.float32, device=dev)
        sps = []
        for i in range(K):
            first, rows, grows = dp.shard(i, n_win, a.batch)
            sp = tr.step_params(nets, grows, True)
            for n_ in range(len(sp.adam_t)):
                sp.adam_t[n_] = i + 1
            sps.append(sp)
            lo = first - dp.global_first(i, a.batch)
            tr.dp_step(phase, nets, first, rows, sp, eps_all[i, lo:lo + rows].unsqueeze(0).contiguous(), losses[i], next_span=None)
        torch.cuda.synchronize()
        dp.all_reduce(losses)
        res = None
        if rank == 0:
            try:                                   # (rank-local work: whatever happens here, rank 0 reaches the barrier below)
                e1 = HipEngine(eng.arch, B, device=dev)
                e1.params.copy_(snap[0])
                e1.exp_avg.zero_()
                e1.exp_avg_sq.zero_()
                e1.bind_dataset(*ds.device_arrays(e1.device))
                l1 = torch.zeros(K, 5, dtype=torch.float32, device=dev)
                for i in range(K):
                    e1.train_step(phase, dp.global_first(i, a.batch), B, sps[i], eps=eps_all[i].unsqueeze(0).contiguous(),
                                  loss_out=l1[i], next_span=None)
                torch.cuda.synchronize()
                ld = ((losses[:, 0] - l1[:, 0]).abs() / l1[:, 0].abs().clamp_min(1e-30)).max().item()
                d_dp = (eng.segment(eng.params, nets) - eng.segment(snap[0], nets)).double()
                d_1 = (e1.segment(e1.params, nets) - eng.segment(snap[0], nets)).double()
                ud = float((d_dp - d_1).norm() / d_1.norm().clamp_min(1e-30))
                res = {"steps": K, "global_batch": B, "losses_n_ranks": losses[:, 0].tolist(),
                       "losses_one_process": l1[:, 0].tolist(), "max_rel_loss_diff": ld, "update_rel_l2_diff": ud,
                       "tolerance": {"loss": 1e-5, "update": 1e-3},
                       "matches": bool(ld < 1e-5 and ud < 1e-3 and float(d_1.norm()) > 0.0)}
                del e1
            except Exception as exc:                               # noqa: BLE001
                res = {"error": str(exc)[:300], "matches": None}
        for dst, src in zip((eng.params, eng.exp_avg, eng.exp_avg_sq), snap):
            dst.copy_(src)
        tr.optimizer.net_steps.clear()
        tr.optimizer.net_steps.update(counts)
        eng.invalidate_staging()
        torch.cuda.synchronize()
        dist.barrier()
        return res

    single_ref = None
    if dp.collective and dp.world > 1:
        try:
            single_ref = reference_steps()
        except Exception as exc:                                   # noqa: BLE001
            single_ref = {"error": str(exc)[:300], "matches": None}
    comm_rank, comm_ranks = eng.comm_info()
    shared_gpu = os.environ.get("PVAE_BENCH_SHARED_GPU") == "1"
    mode_now = a.exchange or (autotune["chosen"] if autotune and autotune["chosen"] in parallel.EXCHANGE_FORMS else None)
    if not dp.collective:
        transport = "none (single rank: Adam inside the backward launches, deferred one launch behind each weight gradient)"
    elif mode_now in ("p2p", "p2p_push"):
        transport = ("in-library peer-mapped exchange (hipIpc-mapped arenas, no RCCL; %s form): one launch per stack -- rank-order "
                     "reduce-scatter by the slice owners, Adam on the owned slice, parameters pushed to every peer"
                     % ("push" if mode_now == "p2p_push" else "pull"))
    elif eng.has_comm:
        transport = {None: "in-library RCCL all-reduce + flat Adam; default schedule: in line on the compute stream in the "
                           "world phase, 6 MiB buckets overlapped on the exchange stream in the joint phase",
                     "inline": "in-library RCCL all-reduce per stack, in line on the compute stream, + flat Adam",
                     "bucketed": "in-library RCCL all-reduce in 6 MiB buckets overlapped on the exchange stream + per-bucket Adam",
                     "sharded": "in-library RCCL reduce-scatter -> Adam on the owned 1/N slice -> all-gather of the parameters",
                     }[mode_now]
    else:
        transport = "torch.distributed (%s) bucketed async all-reduce + per-bucket Adam" % dist.get_backend()
    out = {
        "metric": METRIC,
        "value": value, "unit": "samples/s", "n_gpus": a.gpus,
        # steps = what `ms_per_step` / `value` were timed over: ONE region of that many optimizer steps (the median of
        # `timing.regions` such regions); --steps below MIN_TIMED_STEPS is raised to it (`steps_requested` echoes the flag)
        "steps": timed_steps, "steps_requested": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s, batch %d/GPU, TE/MD/WM 4x1024, %s phase" % (workload, a.batch, a.phase),
                   "phase": a.phase, "global_batch": a.batch * a.gpus, "parallelism": "dp%d" % a.gpus,
                   "exchange": transport,
                   # first layers: "staged" = the gather launch materialises the input / target panels (the default: it rides in
                   # the previous step's last launch); "direct" = PVAE_DIRECT=1, the first layers read the demonstration set
                   # in place (bit-identical, slower at this size: docs/experiments.md, round 5)
                   "first_layers": "direct" if os.environ.get("PVAE_DIRECT", "0") not in ("", "0") else "staged"},
        "timing": {"timed_steps_per_region": timed_steps, "regions": REPEATS, "statistic": "median",
                   "region_values": rates},
        "rccl_ranks": comm_ranks, "rccl_rank": comm_rank,
        "p2p_ranks": eng.p2p_status(sync=False)[1],
        "exchange_mode": a.exchange or (autotune["chosen"] if autotune else "default"),
        "exchange_autotune": autotune, "replicas_identical": replicas_ok,
        "matches_single_process": (single_ref or {}).get("matches"), "single_process_reference": single_ref,
        "ranks_share_a_gpu": shared_gpu,
        "last_loss": last_loss,
    }
    # context for every fraction below: the clock the chip sustains under fp32-MFMA load on THESE weights
    ghz, peak_now = C.c_double(), C.c_double()
    scratch = torch.zeros(1024, dtype=torch.float32, device=dev)
    _lib.check(lib.pvae_mfma_clock_probe(eng.params.data_ptr(), eng.params.numel(), scratch.data_ptr(), C.byref(ghz),
                                         C.byref(peak_now), torch.cuda.current_stream().cuda_stream), "clock probe")
    out["mfma_clock_under_load"] = {"ghz": ghz.value, "fp32_mfma_peak_at_that_clock_tflops": peak_now.value,
                                    "note": "all SIMDs issuing v_mfma_f32_16x16x4_f32 back to back on the parameter "
                                            "arena's values; roofline.peak stays the 2.4 GHz spec figure"}
    fl_world, fl_joint = algorithmic_flops_per_sample(Db, Da, Z, W, D)
    fl = {"world": fl_world, "joint": fl_joint}
    out["step_mfma_frac"] = value / a.gpus * fl[a.phase] / (PEAK_F32_MFMA_TFLOPS * 1e12)
    out["algorithmic_mflop_per_sample"] = {"world": fl_world / 1e6, "joint": fl_joint / 1e6}

    def read_cats():
        cats = {}
        for c in (0, 1, 2, 3, 4, 5):
            ms, cnt, fls = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(lib.pvae_profile_read(c, C.byref(ms), C.byref(cnt), C.byref(fls)))
            if cnt.value:
                cats[c] = dict(total_ms=ms.value, launches=cnt.value, avg_us=ms.value / cnt.value * 1e3,
                               per_launch=fls.value / cnt.value)
        return cats

    def roofline_block(name, ms_step, rp):
        """Instrumented pass of phase `name` (HIP events around every contraction launch and the RCCL
        collective) + the rocprofv3 figures `rp` of the same phase -> (roofline, kernels, collective)."""
        phase, nets = set_phase(name)
        n_prof = 50
        torch.cuda.synchronize()
        lib.pvae_profile_enable(1)
        run_steps(phase, nets, n_prof, 0)
        torch.cuda.synchronize()
        lib.pvae_profile_enable(0)
        cats = read_cats()
        coll = cats.pop(4, None)
        if not cats:
            return None, None, coll
        # the dominant kernel: largest total on the profiler's clock (rocprofv3 kernel-trace durations of the child run:
        # what the committed summary under profiles/ shows and what `avg_launch_us_rocprof` must agree with); HIP-event
        # totals decide only when no profiler figures exist.  The shares of every category are in `kernels`.
        def total_us(c):
            r_ = (rp or {}).get(c, {})
            return (r_["avg_us"] if "avg_us" in r_ else cats[c]["avg_us"]) * cats[c]["launches"]
        dom = max(cats, key=total_us)
        d = cats[dom]
        r = (rp or {}).get(dom, {})
        us_ev, us_rp = d["avg_us"], r.get("avg_us")
        us = max(us_ev, us_rp) if us_rp else us_ev
        tflops = d["per_launch"] / (us * 1e-6) / 1e12
        traffic = None
        if "hbm_fetch_bytes" in r and "hbm_write_bytes" in r:
            traffic = r["hbm_fetch_bytes"] + r["hbm_write_bytes"]
        roof = {"bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                "traffic_unit": "HBM bytes/launch: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, child runs of this command",
                "kernel": CAT_NAMES[dom], "phase": name,
                "clock": "rocprofv3 --kernel-trace (child run)" if us_rp and us_rp >= us_ev else "HIP events at kernel start/end",
                "avg_launch_us": us_ev, "avg_launch_us_rocprof": us_rp,
                "frac_hip_events": d["per_launch"] / (us_ev * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "frac_rocprof": (d["per_launch"] / (us_rp * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS) if us_rp else None,
                "mfma_busy": r.get("mfma_busy"), "frac_of_sustained_clock_peak": tflops / peak_now.value,
                "algorithmic_gflop_per_launch": d["per_launch"] / 1e9,
                "launches_per_step": d["launches"] / n_prof}
        kernels = {}
        for c, v in cats.items():
            k = {"launches_per_step": v["launches"] / n_prof, "avg_us": v["avg_us"],
                 "algo_gflop_per_launch": v["per_launch"] / 1e9,
                 "tflops": v["per_launch"] / (v["avg_us"] * 1e-6) / 1e12}
            for key in ("avg_us", "hbm_fetch_bytes", "hbm_write_bytes", "mfma_busy"):
                if key in (rp or {}).get(c, {}):
                    k["rocprof_" + key] = rp[c][key]
            k["share_of_contraction_time"] = total_us(c) / sum(total_us(x) for x in cats)
            kernels[CAT_NAMES[c]] = k
        kernels["gemm_time_share_of_step"] = sum(v["total_ms"] for v in cats.values()) / n_prof / ms_step
        return roof, kernels, coll

    if not a.no_extra:
        other = "joint" if a.phase == "world" else "world"
        v2, ms2, _, _ = timed(other, MIN_TIMED_STEPS, max(a.warmup // 2, 5), REPEATS)
        out[other + "_value"] = v2
        out[other + "_ms_per_step"] = ms2
        out[other + "_step_mfma_frac"] = v2 / a.gpus * fl[other] / (PEAK_F32_MFMA_TFLOPS * 1e12)
        rp_main = rp_other = None
        if a.gpus == 1 and not a.no_rocprof:
            t0 = time.time()
            rp_main = rocprof_passes(a, a.phase)
            rp_other = rocprof_passes(a, other, budget_s=60)       # kernel trace (+ what fits)
            out["rocprof_child_runs_s"] = time.time() - t0
            for tag, rp in ((a.phase, rp_main), (other, rp_other)):
                if rp.get("error") or rp.get("notes"):
                    out.setdefault("rocprof_notes", {})[tag] = rp.get("error") or rp.get("notes")
        roof, kernels, coll = roofline_block(a.phase, ms_per_step, rp_main)
        if roof:
            out["roofline"], out["kernels"] = roof, kernels
        roof2, kernels2, _ = roofline_block(other, ms2, rp_other)
        if roof2:
            out[other + "_roofline"], out[other + "_kernels"] = roof2, kernels2
        if coll:
            n_prof = 50
            out["allreduce_us_per_step"] = coll["total_ms"] / n_prof * 1e3
            out["allreduce_calls_per_step"] = coll["launches"] / n_prof
            out["allreduce_bytes_per_step"] = coll["per_launch"] * coll["launches"] / n_prof
        elif dp.collective:
            out["allreduce_us_per_step"] = None                   # torch.distributed transport: not instrumented

    if not a.no_extra:
        # the minibatch gather on its own (HBM-bound by nature; SURVEY.md 8d: (2 Db + Da) * 4 bytes read and the
        # same written per sample).  In the training step it rides in the last launch (prefetch), so this
        # stand-alone launch only ever runs for the first minibatch of a loader: reported for the roofline.
        n_g = 200
        first, rows, _ = dp.shard(0, n_win, a.batch)
        for _ in range(10):
            eng.gather(first, rows)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n_g):
            f_i, r_i, _ = dp.shard(i % max(n_win // (a.batch * dp.world), 1), n_win, a.batch)
            eng.gather(f_i, r_i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n_g
        gbytes = 2.0 * rows * (2 * Db + Da) * 4
        out["gather_roofline"] = {"bound": "hbm", "achieved": gbytes / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": gbytes / (us * 1e-6) / 1e9 / 8000.0, "kernel": "stage_batch_kernel (stand-alone gather)",
                                  "avg_launch_us": us, "algorithmic_bytes_per_launch": gbytes, "rows": rows,
                                  "note": "launch-latency bound at this size (%.2f MB per launch); inside the step the "
                                          "gather of the next minibatch rides in the trailing launch" % (gbytes / 1e6)}
        eng.invalidate_staging()
        # The same kernel where its launch is NOT the cost: 8192 windows per launch (an engine with token-sized stacks, the
        # same resident demonstration set) -- what the gather reaches of the HBM roofline when it is bandwidth-bound.
        try:
            from physicsvae_amd.engine import Arch, HipEngine
            big_rows = min(8192, n_win)
            ge = HipEngine(Arch(Db, Da, Z, (64, 1), (64, 1), (64, 1)), big_rows, device=dev)
            ge.bind_dataset(*ds.device_arrays(ge.device))
            for _ in range(5):
                ge.gather(0, big_rows)
            torch.cuda.synchronize()
            e0.record()
            for i in range(50):
                ge.gather((i * 97) % max(n_win - big_rows, 1), big_rows)
            e1.record()
            torch.cuda.synchronize()
            us_b = e0.elapsed_time(e1) * 1e3 / 50
            gb = 2.0 * big_rows * (2 * Db + Da) * 4
            # what the launch really moves: every window is written into FIVE padded panels (encoder [s1|s2], decoder [s1|z],
            # world model [s1|a], the two targets: each stack's first layer contracts over a dense 64-float-aligned panel of
            # its own) -- 2.9x the algorithmic write bytes at these dims, which caps `frac` at ~0.4 whatever the kernel does
            pad = lambda n: (n + 63) // 64 * 64                                                     # noqa: E731
            real = big_rows * 4.0 * ((2 * Db + Da) + pad(2 * Db) + pad(Db + Z) + pad(Db + Da) + pad(Db) + pad(Da))
            out["gather_roofline"]["bandwidth_bound_case"] = {
                "rows": big_rows, "algorithmic_bytes_per_launch": gb, "avg_launch_us": us_b,
                "achieved": gb / (us_b * 1e-6) / 1e9, "unit": "GB/s", "frac": gb / (us_b * 1e-6) / 1e9 / 8000.0,
                "panel_bytes_per_launch": real, "achieved_panel_traffic": real / (us_b * 1e-6) / 1e9,
                "frac_panel_traffic": real / (us_b * 1e-6) / 1e9 / 8000.0,
                "note": "achieved / frac: SURVEY.md 8d's algorithmic bytes ((2 Db + Da) * 4 read + the same written per window); "
                        "panel traffic: the bytes the five padded input / target panels of a window really take (DESIGN.md section 4)"}
            del ge
        except Exception as exc:                                   # noqa: BLE001
            out["gather_roofline"]["bandwidth_bound_case"] = {"error": str(exc)[:200]}

    if dp.collective and not a.no_extra and not a.no_sweep:
        # Every exchange form this build has, timed back to back in THIS run (one region each), so that one multi-GPU
        # lease yields the whole comparison: RCCL all-reduce in line / bucketed + overlapped / RCCL reduce-scatter +
        # sharded Adam + all-gather / the peer-mapped all-pairs exchange, and the step with the exchange switched off
        # ("local": every rank applies Adam to its own gradient -- what an exchange ADDS is the difference to it).
        # The modes are switched at run time; "local" lets the replicas diverge, so it is timed last.  `value` above
        # is untouched by this block.
        sweep = {}
        if eng.has_p2p:
            out["p2p_timeouts"] = eng.p2p_status()[2]

        def one(mode):
            entry = {}
            try:
                if mode in ("p2p", "p2p_push"):
                    # (a peer that never answers must cost this run a fraction of a second per wait, not the
                    #  library's default 20 s: the time-out is read when the peers are mapped)
                    os.environ.setdefault("PVAE_P2P_TIMEOUT_MS", p2p_timeout_ms())
                    if not dp.attach_p2p(eng, mode):               # collective: every rank agrees on the outcome
                        return {"skipped": "peer-mapped exchange could not be set up (see stderr)"}
                elif mode == "local":
                    if not eng.in_library_exchange:
                        return {"skipped": "no in-library exchange"}
                    eng.comm_mode("local")
                else:
                    if not eng.has_comm:
                        return {"skipped": "no RCCL communicator (ranks share a GPU over gloo)"}
                    eng.comm_mode("sharded" if mode == "sharded" else "allreduce")
                    eng.comm_config(6.0 if mode == "bucketed" else 0.0)
                phase, nets = set_phase(a.phase)
                run_steps(phase, nets, 10, 0)
                if mode in ("p2p", "p2p_push"):                    # all ranks agree on whether any wait gave up
                    bad = torch.tensor([eng.p2p_status()[2]], dtype=torch.int64, device=dev)
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                    if int(bad.item()):
                        eng.comm_mode("allreduce" if eng.has_comm else "local")
                        return {"error": "%d waits for a peer gave up within the first 10 steps" % int(bad.item())}
                lib.pvae_profile_enable(1)                         # HIP events around every collective / exchange launch
                run_steps(phase, nets, 40, 10)
                torch.cuda.synchronize()
                lib.pvae_profile_enable(0)
                coll = read_cats().get(4)
                v, ms, _, _ = timed(a.phase, MIN_TIMED_STEPS, 5, 1)
                entry = {"value": v, "ms_per_step": ms}
                if coll:
                    entry["exchange_launches_per_step"] = coll["launches"] / 40
                    entry["us_per_step_inside_exchange_launches"] = coll["total_ms"] / 40 * 1e3
                if mode in ("p2p", "p2p_push"):
                    entry["p2p_ranks"], entry["timeouts"] = eng.p2p_status()[1:]
            except Exception as exc:                               # noqa: BLE001  (a rank-local failure: say so, go on)
                entry = {"error": str(exc)[:300]}
            return entry

        for mode in ("inline", "bucketed", "sharded", "p2p", "p2p_push", "local"):
            sweep[mode] = one(mode)
        base = sweep["local"].get("ms_per_step")
        for mode, e in sweep.items():
            if base and mode != "local" and "ms_per_step" in e:
                e["exchange_exposed_us_per_step"] = (e["ms_per_step"] - base) * 1e3
        out["exchange_sweep"] = sweep
        if base:
            out["step_without_exchange_us"] = base * 1e3
            out["exchange_exposed_us_per_step"] = (ms_per_step - base) * 1e3
    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, sd, Db, Da, Z, W, D, a.phase, synth_demo)
    if dist.is_initialized():
        torch.cuda.synchronize()
        dist.barrier()
        eng.comm_destroy()
        dist.destroy_process_group()
    # RCCL writes an init banner through C stdio, which a pipe would otherwise deliver AFTER
    # Python's output: drain it first so that the JSON object is the last line on stdout.
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
