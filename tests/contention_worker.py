"""One of N processes that share a GPU and each run the SAME training steps over and over (tests/test_gpu_contention.py).

    python contention_worker.py <repo root> <out prefix> <rank> <n procs> <repetitions>

Per configuration and phase: K = 3 optimizer steps of the single-GPU step from one state, repeated R times; after every
repetition the parameter arena, both Adam moment arenas and every step's five loss terms are compared BIT FOR BIT with
repetition 0's (a wrong tile anywhere in a step reaches them; the workspace itself is not compared: which of the two sets of
staging panels a step ends on is bookkeeping).
Nothing is exchanged between the processes -- they only contend for the chip (CUs, LDS ports, L2, the fabric), which is what
opened round 5's write-after-read window in the LDS-DMA ring of the forward kernel (one restaged slot read late: ~1 wrong tile
per 100 hidden-layer launches with four processes, never alone).  A file barrier lines the processes up in front of every
configuration so that they really overlap."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

root, out, rank, nproc, R = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tools"))
from physicsvae_amd.train_physics_vae import WindowDataset                           # noqa: E402
from synth_demo import make_trainer, synth_demo                                      # noqa: E402  (inputs only)

dev = "cuda:0"
K, Z, W, D, T = 3, 32, 1024, 4, 1001


def barrier(tag):
    open("%s.at.%s.%d" % (out, tag, rank), "w").close()
    t0 = time.time()
    while not all(os.path.exists("%s.at.%s.%d" % (out, tag, r)) for r in range(nproc)):
        if time.time() - t0 > 300:
            raise RuntimeError("a peer never reached %s" % tag)
        time.sleep(0.002)


# (name, dim_body, dim_action, rows, direct first layers) -> which LDS-DMA ring the hidden forward / input-gradient launches run on
CONFIGS = [("c3", 197, 45, 256, False),        # 32x32 tiles: six-slot super-step ring; the decoder's `_pro_` kernel (4 slots)
           ("c3_direct", 197, 45, 256, True),  # ... the `*_gather_kernel` forms of the first layers
           ("c5", 400, 90, 512, False),        # 64x32 tiles (ws64, PT = 32)
           ("r1024", 200, 48, 1024, False)]    # 64x64 tiles (ws64, PT = 64)
only = os.environ.get("PVAE_CONTENTION_ONLY")
report = {}
for name, Db, Da, rows, direct in CONFIGS:
    if only and name not in only.split(","):
        continue
    E = (K * rows + T - 2) // (T - 1) + 1
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(synth_demo(0, 1, 4, Db, Da), rows, dev, width=W, depth=D, latent=Z)
    eng = tr.engine
    gen = torch.Generator(device=dev).manual_seed(0)
    states = torch.randn(E * T + 4, Db, generator=gen, device=dev)[: E * T]       # (slack behind the last row: direct steps)
    actions = torch.randn(E * T + 4, Da, generator=gen, device=dev)[: E * T].clamp_(-3, 3)
    rows_idx = (torch.arange(E, device=dev)[:, None] * T + torch.arange(T - 1, device=dev)[None, :]).reshape(-1)
    ds = WindowDataset(np.zeros((2, Db), np.float32), np.zeros((2, Da), np.float32), np.zeros(1, np.int32))
    ds._dev = (states, actions, rows_idx.to(torch.int32))
    ds.window_row = np.empty(E * (T - 1), dtype=np.int8)
    tr.train_loader.dataset = ds
    eng.bind_dataset(*ds.device_arrays(eng.device))
    if direct:
        eng.set_direct(True)
    start = eng.params.clone()
    for phase_name in ("world", "joint"):
        wp = phase_name == "world"
        tr.model.set_learnable_task_encoder(not wp)
        tr.model.set_learnable_motor_decoder(not wp)
        tr.model.set_learnable_world_model(wp)
        tr.read_loss_fn_coeff(world=wp)
        phase, nets = tr.phase()
        losses = torch.zeros(K, 5, dtype=torch.float32, device=dev)
        ref, bad, took_direct = None, [], None
        torch.cuda.synchronize()
        barrier("%s.%s" % (name, phase_name))
        for rep in range(R):
            eng.params.copy_(start)
            eng.params_changed()
            eng.invalidate_staging()
            eng.exp_avg.zero_()
            eng.exp_avg_sq.zero_()
            for i in range(K):
                sp = tr.step_params(nets, rows, True)
                for n_ in range(len(sp.adam_t)):
                    sp.adam_t[n_] = i + 1
                sp.rng_seed, sp.rng_offset = 7, i * 65536
                if took_direct is None:
                    took_direct = bool(eng.direct_active(phase, rows, sp))
                nxt = ((i + 1) * rows, rows) if i + 1 < K else None
                eng.train_step(phase, i * rows, rows, sp, loss_out=losses[i], next_span=nxt)
            got = (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), losses.clone())
            if ref is None:
                ref = got
            elif not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(got, ref)):
                bad.append((rep, [int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(got, ref)]))
        torch.cuda.synchronize()
        report["%s/%s" % (name, phase_name)] = {"repetitions": R, "deviating": bad, "direct": took_direct,
                                                "trained": bool(float(ref[3][0, 0]) != float(ref[3][K - 1, 0]))}
    del tr, eng
    torch.cuda.empty_cache()
torch.save(report, "%s.%d" % (out, rank))
print("DONE", rank)
