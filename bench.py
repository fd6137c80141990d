"""bench.py -- training samples/s of the PhysicsVAE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--phase world|joint]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimizer step over one minibatch of synthetic demonstrations: minibatch
gather from the HBM-resident demo set -> MLP forward -> losses -> backward -> Adam.
Workload = BASELINE.json configs[1]: synthetic loco demo (10 episodes x 1000 steps,
dim_state_body 197, dim_action 45), batch 256 per GPU, TE/MD/WM = 4x1024, world-model-only
phase.  (`--phase joint` times the joint world-model + CVAE step of configs[2]; the default
run also reports it as `joint_value`.)  N > 1: data-parallel, 256 rows per GPU (global batch
N*256, weak scaling), gradient SUM all-reduce over RCCL, replicated Adam.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch
stream in an instrumented pass after the timed region; `cpu_baseline` is the oracle's
restatement of the reference loop (oracle/refpath.py: the checker, timed here, never the
product path) on this host's cores over a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk


def algorithmic_flops_per_sample(Db, Da, Z, W, d):
    """F(I,W,d,O) = I*W + (d-1)*W^2 + W*O MACs; flops = 2*MAC."""
    def F(i, o):
        return i * W + (d - 1) * W * W + W * o
    f_te, f_md, f_wm = F(2 * Db, 2 * Z), F(Db + Z, Da), F(Db + Da, Db)
    world = 3 * f_wm - (Db + Da) * W
    hid = (d - 1) * W * W
    joint = (f_te + f_md + f_wm) + (hid + W * Db + Da * W) + f_md + (hid + W * Da + Z * W) + f_te + (hid + W * 2 * Z)
    return 2 * world, 2 * joint


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--phase", choices=["world", "joint"], default="world")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default 256; 512 for --config c5)")
    ap.add_argument("--config", choices=["c2", "c5"], default="c2",
                    help="c2 = BASELINE configs[1..3] sizes (default); c5 = configs[4]: 1e6 transitions, "
                         "dim_state_body 400, dim_action 90, 512 rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary-phase and roofline passes")
    a = ap.parse_args()

    from physicsvae_amd import _lib, parallel
    rank, world, local = parallel.init_from_env()
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local

    from synth_demo import make_trainer, synth_demo      # tools/: inputs only; oracle/ is imported by the
                                                         # cpu_baseline leg below and nowhere else

    import contextlib
    import io
    Z, W, D = 32, 1024, 4
    if a.config == "c2":
        Db, Da = 197, 45
        a.batch = a.batch or 256
        data = synth_demo(0, 10, 1000, Db, Da)
        workload = "BASELINE configs[1]: synthetic loco demo 10x1000, dim_state_body 197, dim_action 45"
    else:
        Db, Da = 400, 90
        a.batch = a.batch or 512
        data = synth_demo(0, 1, 4, Db, Da)                       # placeholder file; the real set is built below
        workload = "BASELINE configs[4]: synthetic demo 1000x1001 (1e6 transitions), dim_state_body 400, dim_action 90"
    torch.manual_seed(1)                       # normc initialisation of the model's own constructor
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(data, a.batch, dev, width=W, depth=D, latent=Z)
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
    steps_per_epoch = dp.global_steps(n_win, a.batch)
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

    def timed(name, steps, warmup):
        phase, nets = set_phase(name)
        run_steps(phase, nets, warmup, 0)
        if dp.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples = run_steps(phase, nets, steps, warmup)
        torch.cuda.synchronize()
        if dp.world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dp.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return samples / dt, dt / steps * 1e3, float(loss_buf[0].item())

    value, ms_per_step, last_loss = timed(a.phase, a.steps, a.warmup)
    out = {
        "metric": "train samples/sec (world-model+VAE step), loco demo, batch 256, 1/2/4/8 GPU",
        "value": value, "unit": "samples/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s, batch %d/GPU, TE/MD/WM 4x1024, %s phase" % (workload, a.batch, a.phase),
                   "phase": a.phase, "global_batch": a.batch * a.gpus,
                   "parallelism": "dp%d" % a.gpus,
                   "optimizer": "Adam inside the backward launches (deferred one launch behind each weight gradient)" if not dp.collective else
                   ("in-library RCCL all-reduce per net (same stream) + Adam" if eng.has_comm else
                    "torch.distributed bucketed async all-reduce + per-bucket Adam")},
        "last_loss": last_loss,
    }
    fl_world, fl_joint = algorithmic_flops_per_sample(Db, Da, Z, W, D)
    fl = fl_world if a.phase == "world" else fl_joint
    out["step_mfma_frac"] = value / a.gpus * fl / (PEAK_F32_MFMA_TFLOPS * 1e12)

    if not a.no_extra:
        other = "joint" if a.phase == "world" else "world"
        v2, ms2, _ = timed(other, max(a.steps // 4, 20), max(a.warmup // 4, 5))
        out[other + "_value"] = v2
        out[other + "_ms_per_step"] = ms2
        # ---- roofline: instrumented pass (HIP events on the launch stream around every
        # contraction launch), same workload/phase as `value`
        phase, nets = set_phase(a.phase)
        torch.cuda.synchronize()
        lib.pvae_profile_enable(1)
        n_prof = 50
        run_steps(phase, nets, n_prof, 0) if dp.world == 1 else None
        if dp.world > 1:
            for i in range(n_prof):
                first, rows, grows = dp.shard(i % (n_win // (a.batch * dp.world)), n_win, a.batch)
                sp = tr.step_params(nets, grows, True)
                eng.gather(first, rows)
                eng.forward_backward(phase, rows, sp, fused_adam=False, loss_out=loss_buf)
        torch.cuda.synchronize()
        lib.pvae_profile_enable(0)
        adam = "+Adam" if dp.world == 1 else ""
        names = {0: "gemm_splitk_ws_kernel<P_ROW> / gemm_splitk_reg16_kernel (forward layer)",
                 1: "gemm_splitk_ws_kernel<P_COL> (input gradient, 32x32 tile)",
                 2: "wgrad_pair_kernel / gemm_wgrad_reg_kernel (trailing weight gradient%s + deferred Adam of the layer before)" % adam,
                 3: "bwd_pair_kernel (input gradient || weight gradient of one layer%s)" %
                    (", gradient stored; Adam of the previous layer in extra workgroups" if dp.world == 1 else "")}
        cats = {}
        for c in (0, 1, 2, 3):
            ms, cnt, fls = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(lib.pvae_profile_read(c, C.byref(ms), C.byref(cnt), C.byref(fls)))
            if cnt.value:
                cats[c] = dict(kernel=names[c], total_ms=ms.value, launches=cnt.value,
                               avg_us=ms.value / cnt.value * 1e3,
                               algo_gflop_per_launch=fls.value / cnt.value / 1e9,
                               tflops=fls.value / (ms.value * 1e-3) / 1e12)
        dom = max(cats, key=lambda c: cats[c]["total_ms"])
        d = cats[dom]
        # HBM traffic per launch of that kernel: PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE,
        # MI355X_MICROARCH.md HBM section) of this same command, summarised under profiles/
        traffic, traffic_src = None, None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[::-1]:
                summ = json.load(open(f))
                key = {3: "bwd_pair_kernel<EpiMask,EpiGradStore>", 2: "wgrad_pair_kernel<EpiGradAdam>",
                       0: "gemm_splitk_ws_kernel<P_ROW,EpiBiasAct>", 1: "gemm_splitk_ws_kernel<P_COL,EpiMask>"}[dom]
                if key in summ and "hbm_traffic_MB" in summ[key] and dp.world == 1 and a.phase == "world" and a.config == "c2":
                    traffic, traffic_src = summ[key]["hbm_traffic_MB"] * 1e6, os.path.relpath(f, ROOT)
                    break
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "achieved": d["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": d["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "traffic_unit": "bytes/launch (HBM, rocprofv3 PMC)", "traffic_source": traffic_src,
                           "kernel": d["kernel"], "avg_launch_us": d["avg_us"],
                           "algorithmic_gflop_per_launch": d["algo_gflop_per_launch"],
                           "launches_per_step": d["launches"] / n_prof}
        out["kernels"] = {cats[c]["kernel"]: {k: v for k, v in cats[c].items() if k != "kernel"} for c in cats}
        out["gemm_time_share_of_step"] = sum(c["total_ms"] for c in cats.values()) / n_prof / ms_per_step

    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
        from oracle import refpath as R        # the checker, timed as the CPU baseline -- this leg only
        arch = R.make_arch(Db, Da, latent=Z, te=(W, D), md=(W, D), wm=(W, D))
        if a.config != "c2":
            data = synth_demo(0, 10, 1000, Db, Da)                 # bounded CPU sample of the same shape
        X, Y = R.build_windows(data)
        n_b = 3 * 39 if a.config == "c2" else 2 * (len(X) // a.batch)
        t0 = time.perf_counter()
        trc = R.RefTrainer(arch, sd, X, Y, a.batch, max_iter_world_model=(10 ** 9 if a.phase == "world" else 0))
        trc.step(max_batches=2)
        t1 = time.perf_counter()
        done, per = 0, min(39, len(X) // a.batch)
        while done < n_b:
            trc.step(max_batches=min(per, n_b - done))
            done += per
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n_b * a.batch / dt, "unit": "samples/s",
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "%d minibatches of %d (passes over the full minibatches of a 10x1000 "
                                         "synthetic demo of the same dims), %s phase, oracle/refpath.RefTrainer (stock "
                                         "torch CPU ops in the reference's op order), %.1f s" % (n_b, a.batch, a.phase, dt),
                               "host_cpus": os.cpu_count()}
    if dist.is_initialized():
        torch.cuda.synchronize()
        dist.barrier()
        eng.comm_destroy()
        dist.destroy_process_group()
    # RCCL writes an init banner through C stdio, which a pipe would otherwise deliver AFTER
    # Python's output: drain it first so that the JSON object is the last line on stdout.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
