"""Is the peer-mapped data-parallel step deterministic run to run?  N ranks (sharing GPUs when the box has fewer), the
BASELINE configs[3] workload, K optimizer steps from the SAME state repeated R times; after every step the parameter
arena is compared bit for bit with the first repetition's.  A deviation is localised: which step, how many elements,
which owner slice of which layer.

    python tools/p2p_race_hunt.py [p2p|p2p_push] [R] [world|joint]      (self-launches PVAE_HUNT_RANKS = 8 ranks)
"""
import contextlib
import io
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def launch():
    import torch
    world = int(os.environ.get("PVAE_HUNT_RANKS", "8"))
    ndev = max(torch.cuda.device_count(), 1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0",
               PVAE_DP_EXCHANGE=sys.argv[1] if len(sys.argv) > 1 else "p2p")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r % ndev), PVAE_LOCAL_DEVICE=str(r % ndev)),
                              stdout=None) for r in range(world)]
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main():
    import numpy as np
    import torch
    from physicsvae_amd import parallel
    rank, world, local = parallel.init_from_env(backend="gloo" if torch.cuda.device_count() < int(os.environ["WORLD_SIZE"]) else None)
    import torch.distributed as dist
    from physicsvae_amd.train_physics_vae import WindowDataset
    from synth_demo import make_trainer, synth_demo
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    form = os.environ["PVAE_DP_EXCHANGE"]
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    phase_name = sys.argv[3] if len(sys.argv) > 3 else "world"
    Z, W, D, K = 32, 1024, 4, 3
    Db, Da, per_gpu, E, T = 197, 45, 256, 8, 1001
    B = per_gpu * world
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = make_trainer(synth_demo(0, 1, 4, Db, Da), per_gpu, dev, width=W, depth=D, latent=Z, extra={"dp_exchange": form})
    eng, dp = tr.engine, tr.dp
    if os.environ.get("PVAE_HUNT_MODE") == "local":     # the peers stay mapped, but every rank applies ITS OWN gradient: no peer traffic
        eng.comm_mode("local")
    gen = torch.Generator(device=dev).manual_seed(0)
    states = torch.randn(E * T, Db, generator=gen, device=dev)
    actions = torch.randn(E * T, Da, generator=gen, device=dev).clamp_(-3, 3)
    rows_idx = (torch.arange(E, device=dev)[:, None] * T + torch.arange(T - 1, device=dev)[None, :]).reshape(-1)
    ds = WindowDataset(np.zeros((2, Db), np.float32), np.zeros((2, Da), np.float32), np.zeros(1, np.int32))
    ds._dev = (states, actions, rows_idx.to(torch.int32))
    ds.window_row = np.empty(E * (T - 1), dtype=np.int8)
    tr.train_loader.dataset = ds
    eng.bind_dataset(*ds.device_arrays(eng.device))
    n_win = len(ds)
    wp = phase_name == "world"
    tr.model.set_learnable_task_encoder(not wp)
    tr.model.set_learnable_motor_decoder(not wp)
    tr.model.set_learnable_world_model(wp)
    tr.read_loss_fn_coeff(world=wp)
    phase, nets = tr.phase()
    start = eng.params.clone()
    eps_all = torch.randn(K, B, Z, generator=torch.Generator(device="cpu").manual_seed(1234)).to(dev)
    losses = torch.zeros(K, 5, dtype=torch.float32, device=dev)
    junk = torch.ones(64 << 20, device=dev) if os.environ.get("PVAE_HUNT_FLUSH") == "1" else None
    ref = refg = None                                   # [K] parameter / gradient arenas of repetition 0
    bad = 0
    for rep in range(R):
        eng.params.copy_(start)
        eng.invalidate_staging()
        eng.exp_avg.zero_()
        eng.exp_avg_sq.zero_()
        torch.cuda.synchronize()
        dist.barrier()
        got, gotg, gotw = [], [], []
        t_rep = time.perf_counter()
        for i in range(K):
            first, rows, grows = dp.shard(i, n_win, per_gpu)
            sp = tr.step_params(nets, grows, True)
            for n_ in range(len(sp.adam_t)):
                sp.adam_t[n_] = i + 1
            lo = first - dp.global_first(i, per_gpu)
            tr.dp_step(phase, nets, first, rows, sp, eps_all[i, lo:lo + rows].unsqueeze(0).contiguous(), losses[i], next_span=None)
            if os.environ.get("PVAE_HUNT_SYNC") == "1":
                torch.cuda.synchronize()
                dist.barrier()
            if os.environ.get("PVAE_HUNT_FLUSH") == "1":    # stream 256 MB through every XCD's L2: nothing cached before survives
                junk_sum = junk.sum()
            got.append(eng.params.clone())
            gotg.append(eng.grads.clone())              # (the exchange leaves this rank's own gradient where the backward pass put it)
            gotw.append(eng.workspace.clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_rep
        to = eng.p2p_status()[2] if eng.has_p2p else 0
        if to or dt > 1.0 or (rank == 0 and rep % 5 == 0):
            print("rank %d rep %d: %.2f s, %d waits gave up so far" % (rank, rep, dt, to), flush=True)
        if ref is None:
            ref, refg, refw = got, gotg, gotw
            continue
        for i in range(K):                      # which panel of the workspace is the first (in dataflow order) to differ?
            if torch.equal(gotw[i].view(torch.int32), refw[i].view(torch.int32)):
                continue
            from physicsvae_amd import _lib as L_
            import ctypes as C_
            net = L_.NET_WM if wp else L_.NET_TE
            lays = [l for l in eng.layers if l["net"] == net]
            order = [("in", 0)] + [("act", j) for j in range(len(lays))] + [("dz", j) for j in reversed(range(len(lays)))]
            kinds = {"in": 0, "d_in": 1, "act": 2, "dz": 3}
            rep_ = []
            for kind, j in order:
                off = int(eng.lib.pvae_workspace_offset(C_.byref(eng.cfg), kinds[kind], net, j))
                width = lays[0]["ld"] if kind == "in" else lays[j]["n_out_pad"]
                a_ = gotw[i][off: off + 256 * width].view(256, width)
                b_ = refw[i][off: off + 256 * width].view(256, width)
                ne_ = (a_.view(torch.int32) != b_.view(torch.int32))
                if int(ne_.sum()):
                    rows_ = ne_.any(dim=1).nonzero().flatten()
                    cols_ = ne_.any(dim=0).nonzero().flatten()
                    rep_.append("%s[%d]: %d floats, rows %d..%d (%d rows), cols %d..%d (%d cols), max abs %.2e" %
                                (kind, j, int(ne_.sum()), int(rows_[0]), int(rows_[-1]), rows_.numel(), int(cols_[0]), int(cols_[-1]),
                                 cols_.numel(), float((a_ - b_).abs().max())))
            print("rank %d rep %d step %d workspace: %s" % (rank, rep, i + 1, " | ".join(rep_) if rep_ else "(differences outside this net's panels)"), flush=True)
            break
        for i in range(K):
            ng = int((gotg[i].view(torch.int32) != refg[i].view(torch.int32)).sum())
            if ng:
                dg = (gotg[i] - refg[i]).double()
                wne = (gotw[i].view(torch.int32) != refw[i].view(torch.int32)).nonzero().flatten()
                pprev = "n/a" if i == 0 else int((got[i - 1].view(torch.int32) != ref[i - 1].view(torch.int32)).sum())
                print("rank %d rep %d step %d: this rank's OWN gradient differs from repetition 0 in %d elements (rel L2 %.3e, max abs %.3e); "
                      "parameters before the step differ in %s elements; workspace differs in %d floats, first at %s of %d"
                      % (rank, rep, i + 1, ng, float(dg.norm() / refg[i].double().norm()), float(dg.abs().max()), pprev, wne.numel(),
                         int(wne[0]) if wne.numel() else None, gotw[i].numel()), flush=True)
                break
        for i in range(K):
            ne = (got[i].view(torch.int32) != ref[i].view(torch.int32))
            n = int(ne.sum())
            if n:
                bad += 1
                idx = ne.nonzero().flatten()
                where = []
                for info in eng.layers:
                    lo_, hi_ = info["w_offset"], info["b_offset"] + info["n_out_pad"]
                    m = int(((idx >= lo_) & (idx < hi_)).sum())
                    if m:
                        # owner slices of the single in-line bucket of this net (p2p: n4 split over the ranks)
                        noff, ncnt = eng.segments[info["net"]]
                        S4 = ((ncnt // 4) + world - 1) // world
                        sel = idx[(idx >= lo_) & (idx < hi_)]
                        owners = sorted(set((((sel - noff) // 4) // S4).tolist()))
                        where.append("net %d layer %d: %d elements, owner slices %s, first %d last %d" %
                                     (info["net"], info["index"], m, owners, int(sel[0]), int(sel[-1])))
                print("rank %d rep %d step %d: %d PARAMETER elements differ from repetition 0 (max abs diff %.3e)%s" %
                      (rank, rep, i + 1, n, float((got[i] - ref[i]).abs().max()), ("\n    " + "\n    ".join(where)) if rank == 0 else ""), flush=True)
                break
    flag = torch.tensor([bad], dtype=torch.int64, device=dev)
    dist.all_reduce(flag)
    same = dp.replicas_identical(eng)                   # (collective)
    if rank == 0:
        print("%s, %s phase, %d ranks, %d repetitions of %d steps: %d deviating repetition(s) summed over the ranks; replicas identical: %s"
              % (form, phase_name, world, R, K, int(flag.item()), same), flush=True)
    dist.barrier()


if __name__ == "__main__":
    if "RANK" not in os.environ:
        sys.exit(launch())
    main()
