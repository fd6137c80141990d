"""Random interleavings of the training entry points.  Engine A goes through the paths the trainer
uses (one-call steps with Adam inside the backward launches, the next minibatch's gather riding in
the last launch, whatever the step left pending); engine B spells every step out (gather, forward +
backward with the gradient stored, flat Adam).  Evaluation forwards, rollout calls, explicit
minibatches and dataset re-binding are thrown in between.  Whatever the order, both must hold the
same parameters and moments bit for bit: the fused / deferred / prefetched machinery carries no
state that an unexpected call sequence can corrupt."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(16))
def test_random_call_sequences_leave_identical_state(seed):
    rng = np.random.default_rng(500 + seed)
    act = ("relu", "relu", "tanh", "sigmoid", "elu")[seed % 5]          # the trainer's "act_fn"
    wd = 0.01 if seed % 3 == 0 else 0.0                                 # ... and "weight_decay"
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 3), wm=(128, 2), act=act)
    data = R.synth_demo(seed, 3, 120, 23, 7, kind="dynamics")
    data2 = R.synth_demo(seed + 50, 2, 90, 23, 7, kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed + 1), seed + 3)
    trs = [make_trainer(arch, data, 64, device="cuda") for _ in range(2)]
    sets = []
    for d in (data, data2):
        ds = make_trainer(arch, d, 64, device="cuda").train_loader.dataset
        sets.append((ds.device_arrays(trs[0].engine.device), len(ds)))
    for tr in trs:
        tr.model.load_state_dict(sd)
        tr.engine.exp_avg.zero_(); tr.engine.exp_avg_sq.zero_()
        tr.engine.bind_dataset(*sets[0][0])
    A, B = trs[0].engine, trs[1].engine
    n_win = sets[0][1]
    t_adam = [0, 0, 0]
    out = torch.zeros(5, device="cuda")
    step = 0
    for op_i in range(60):
        op = rng.choice(["train", "train", "train", "train_pf", "train_pf", "eval", "infer", "explicit", "rebind"])
        world = bool(rng.integers(0, 2))
        phase = _lib.PHASE_WORLD if world else _lib.PHASE_JOINT
        nets = [_lib.NET_WM] if world else [_lib.NET_TE, _lib.NET_MD]
        co = R.phase_coeffs(world)
        rows = int(rng.choice([64, 64, 32, 33, 7]))
        first = int(rng.integers(0, n_win - rows))

        def params():
            for n in nets:
                t_adam[n] += 1
            return make_step_params(lr=1e-3, adam_t=tuple(max(t, 1) for t in t_adam), a_rec=co["a_rec_coeff"],
                                    kl=co["vae_kl_coeff"], s_rec=co["s_rec_coeff"], cyc=co["vae_cycle_coeff"],
                                    global_rows=rows, seed=11, offset=step * 65536, weight_decay=wd)
        if op in ("train", "train_pf"):
            sp = params()
            nxt = (int(rng.integers(0, n_win - rows)), rows) if op == "train_pf" else None
            A.train_step(phase, first, rows, sp, loss_out=out, next_span=nxt)
            if nxt is not None and rng.integers(0, 2):           # ... and sometimes the prefetched batch IS the next one
                step += 1
                sp2 = params()
                A.train_step(phase, nxt[0], rows, sp2, loss_out=out, next_span=None)
                for e, s, f in ((B, sp, first), (B, sp2, nxt[0])):
                    e.gather(f, rows)
                    e.forward_backward(phase, rows, s, fused_adam=False)
                    e.adam(nets, s)
            else:
                B.gather(first, rows)
                B.forward_backward(phase, rows, sp, fused_adam=False)
                B.adam(nets, sp)
            step += 1
        elif op == "eval":
            sp = make_step_params(lr=1e-3, a_rec=co["a_rec_coeff"], kl=co["vae_kl_coeff"], s_rec=co["s_rec_coeff"],
                                  cyc=co["vae_cycle_coeff"], global_rows=rows, seed=3, offset=op_i)
            for e in (A, B):
                e.gather(first, rows)
                e.forward_backward(phase, rows, sp, backward=False, loss_out=out)
        elif op == "infer":
            obs = torch.randn(int(rng.choice([1, 4, 9])), 46, device="cuda")
            ra = A.infer(obs, noise=False)
            rb = B.infer(obs, noise=False)
            assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
        elif op == "explicit":
            sp = params()
            x = torch.randn(rows, 1, 46)
            y = torch.randn(rows, 1, 7)
            A.set_batch(x, y)
            A.forward_backward(phase, rows, sp, fused_adam=True)
            B.set_batch(x, y)
            B.forward_backward(phase, rows, sp, fused_adam=False)
            B.adam(nets, sp)
            step += 1
        else:
            k = int(rng.integers(0, 2))
            for e in (A, B):
                e.bind_dataset(*sets[k][0])
            n_win = sets[k][1]
        assert torch.equal(A.params, B.params), (seed, op_i, op)
    assert torch.equal(A.exp_avg, B.exp_avg) and torch.equal(A.exp_avg_sq, B.exp_avg_sq)
    assert not torch.equal(A.params.cpu(), trs[0].engine.params.new_zeros(A.params.shape).cpu())
