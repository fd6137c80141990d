"""One rank of the full-size data-parallel check (tests/test_gpu_dp_full_size.py): BASELINE configs[3] / configs[4]'s
workload on however many GPUs the box has (fewer GPUs than ranks: the ranks share devices over gloo).

    python dp_full_size_worker.py <repo root> <out prefix> <c3|c5>        (PVAE_DP_FORMS = default,p2p,p2p_push: the
    exchange forms to run one after the other in this one set of processes -- one rendezvous, one demonstration set, one
    single-process reference per phase for all of them)

N ranks against ONE process (tm:131-161: one optimizer step per global minibatch, the loss an unweighted mean over its
rows): K = 3 optimizer steps of each phase from a common state -- moments zeroed, Adam counters 1..K, draws supplied --
through the data-parallel step, and the same K global minibatches through a second engine on rank 0 that holds the whole
global batch.  Step k+1's loss is a function of the parameters step k produced, so a rank that reads stale parameters of
slices its peers own shows up here while the replicas stay bit-identical."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

root, out, config = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tools"))
from physicsvae_amd import parallel                                                  # noqa: E402

rank, world, local = parallel.init_from_env(backend="gloo" if torch.cuda.device_count() < int(os.environ["WORLD_SIZE"]) else None)
import torch.distributed as dist                                                     # noqa: E402
from physicsvae_amd.engine import HipEngine                                          # noqa: E402
from physicsvae_amd.train_physics_vae import WindowDataset                           # noqa: E402
from synth_demo import make_trainer, synth_demo                                      # noqa: E402  (inputs only)

torch.cuda.set_device(local)
dev = "cuda:%d" % local
Z, W, D, K = 32, 1024, 4, 3
Db, Da, per_gpu, E = {"c3": (197, 45, 256, 8), "c5": (400, 90, 512, 13)}[config]
T = 1001
B = per_gpu * world
forms = (os.environ.get("PVAE_DP_FORMS") or os.environ.get("PVAE_DP_EXCHANGE") or "default").split(",")

torch.manual_seed(1)                                   # identical initial weights on every rank
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(synth_demo(0, 1, 4, Db, Da), per_gpu, dev, width=W, depth=D, latent=Z, extra={"dp_exchange": "default"})
eng, dp = tr.engine, tr.dp
# the demonstration set in the packed layout the gather reads, generated on the device (same seed on every rank)
gen = torch.Generator(device=dev).manual_seed(0)
states = torch.randn(E * T, Db, generator=gen, device=dev)
actions = torch.randn(E * T, Da, generator=gen, device=dev).clamp_(-3, 3)
rows_idx = (torch.arange(E, device=dev)[:, None] * T + torch.arange(T - 1, device=dev)[None, :]).reshape(-1)
ds = WindowDataset(np.zeros((2, Db), np.float32), np.zeros((2, Da), np.float32), np.zeros(1, np.int32))
ds._dev = (states, actions, rows_idx.to(torch.int32))
ds.window_row = np.empty(E * (T - 1), dtype=np.int8)   # length only
tr.train_loader.dataset = ds
eng.bind_dataset(*ds.device_arrays(eng.device))
n_win = len(ds)
assert n_win >= K * B


def set_phase(world_phase):
    tr.model.set_learnable_task_encoder(not world_phase)
    tr.model.set_learnable_motor_decoder(not world_phase)
    tr.model.set_learnable_world_model(world_phase)
    tr.read_loss_fn_coeff(world=world_phase)
    return tr.phase()


def set_form(form):
    """Every rank switches to the same form (collective for the peer-mapped ones)."""
    if form in ("p2p", "p2p_push"):
        assert dp.attach_p2p(eng, form), "peer-mapped exchange could not be set up"
        eng.comm_config(0.0)
        assert eng.has_p2p and eng.p2p_status()[:2] == (rank, world)
    else:
        if eng.has_p2p:
            torch.cuda.synchronize(); dist.barrier()
            eng.p2p_close()
            dist.barrier()
        if torch.cuda.device_count() < world:
            assert not eng.in_library_exchange             # gloo: torch.distributed carries the exchange
    tr.dp_exchange = None if form == "default" else form
    tr.dp_sharded = form in ("sharded", "p2p", "p2p_push") and bool(eng.in_library_exchange)


start = eng.params.clone()
gen_cpu = torch.Generator(device="cpu").manual_seed(1234)
eps_of = {name: torch.randn(K, B, Z, generator=gen_cpu).to(dev) for name in ("world", "joint")}      # identical on every rank
one = {}                                               # per phase: the single process with the global batch (rank 0, once)
out_all = {}
for form in forms:
    set_form(form)
    res = {"form": form, "ranks": world, "global_batch": B, "in_library": bool(eng.in_library_exchange)}
    for name in ("world", "joint"):
        phase, nets = set_phase(name == "world")
        eng.params.copy_(start)
        eng.params_changed()
        eng.invalidate_staging()
        eng.exp_avg.zero_()
        eng.exp_avg_sq.zero_()
        torch.cuda.synchronize()
        dist.barrier()
        eps_all = eps_of[name]
        losses = torch.zeros(K, 5, dtype=torch.float32, device=dev)
        sps = []
        for i in range(K):
            first, rows, grows = dp.shard(i, n_win, per_gpu)
            assert rows == per_gpu and grows == B
            sp = tr.step_params(nets, grows, True)
            for n_ in range(len(sp.adam_t)):
                sp.adam_t[n_] = i + 1
            sps.append(sp)
            lo = first - dp.global_first(i, per_gpu)
            tr.dp_step(phase, nets, first, rows, sp, eps_all[i, lo:lo + rows].unsqueeze(0).contiguous(), losses[i], next_span=None)
        torch.cuda.synchronize()
        dp.all_reduce(losses)
        entry = {"replicas_identical": bool(dp.replicas_identical(eng)),
                 "params_checksum": int(eng.params.view(torch.int32).to(torch.int64).sum()),
                 "timeouts": int(dp.p2p_timeouts(eng)) if eng.has_p2p else 0,
                 "losses_n_ranks": losses[:, 0].tolist()}
        if rank == 0:
            if name not in one:
                e1 = HipEngine(eng.arch, B, device=dev)
                e1.params.copy_(start)
                e1.exp_avg.zero_()
                e1.exp_avg_sq.zero_()
                e1.bind_dataset(*ds.device_arrays(e1.device))
                l1 = torch.zeros(K, 5, dtype=torch.float32, device=dev)
                for i in range(K):
                    e1.train_step(phase, dp.global_first(i, per_gpu), B, sps[i], eps=eps_all[i].unsqueeze(0).contiguous(),
                                  loss_out=l1[i], next_span=None)
                torch.cuda.synchronize()
                one[name] = (l1.clone(), (e1.segment(e1.params, nets) - eng.segment(start, nets)).double())
                del e1
            l1, d_1 = one[name]
            d_dp = (eng.segment(eng.params, nets) - eng.segment(start, nets)).double()
            frozen = [n for n in eng.segments if n not in nets and eng.segments[n][1] > 0]
            entry.update({
                "losses_one_process": l1[:, 0].tolist(),
                "max_rel_loss_diff": float(((losses[:, 0] - l1[:, 0]).abs() / l1[:, 0].abs().clamp_min(1e-30)).max()),
                "max_rel_term_diff": float(((losses[:, 1:] - l1[:, 1:]).abs() / l1[:, 1:].abs().clamp_min(1e-6)).max()),
                "update_norm": float(d_1.norm()),
                "update_rel_l2_diff": float((d_dp - d_1).norm() / d_1.norm().clamp_min(1e-30)),
                # elements whose update is off by more than a tenth of the largest update: Adam turns a gradient entry within
                # rounding of zero into a +-lr step whose sign the summation order decides -- a handful of such entries is
                # conditioning, thousands would be a stale read
                "update_flip_fraction": float(((d_dp - d_1).abs() > 0.1 * d_1.abs().max()).double().mean()),
                "frozen_untouched": all(torch.equal(eng.segment(eng.params, [n]), eng.segment(start, [n])) for n in frozen)})
        res[name] = entry
        dist.barrier()
    out_all[form] = res
torch.save(out_all, out + ".%d" % rank)
dist.barrier()
print("DONE", rank)
