"""Minibatch gather at its edges (window byte offsets across 2^31, cond="rel" windows: tpv:133-156, tm:52-56) and
the opt-in bit-exact resume from trainer_state.pt (ours; upstream restores weights only, tm:215-216)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import arch_from_meta, make_trainer, max_err_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_windows_across_the_2GiB_byte_boundary():
    """A demonstration set of 1.5e6 state rows x 400 floats = 2.4 GB: byte offsets of the rows a
    window reads cross 2^31 (row 1 342 177).  The panels the gather kernel fills for windows at the
    start, straddling the boundary and at the very end equal the oracle's definition of a window
    (x = [s_t | s_{t+1}], y = a_t; tpv:133-156) bit for bit."""
    from physicsvae_amd.engine import Arch, HipEngine
    Db, Da, B = 400, 90, 512
    rows_total = 1_500_000
    eng = HipEngine(Arch(Db, Da, 32, (64, 1), (64, 1), (64, 1)), B, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    states = torch.randn(rows_total, Db, generator=gen, device=DEV)
    actions = torch.randn(rows_total, Da, generator=gen, device=DEV)
    assert states.numel() * 4 > 2 ** 31
    # windows: every row but the last of each 1000-row "episode" (episode boundaries are skipped)
    idx = torch.arange(rows_total, device=DEV)
    window_row = idx[(idx % 1000) != 999].to(torch.int32)
    eng.bind_dataset(states, actions, window_row)
    n = window_row.numel()
    boundary_row = 2 ** 31 // (Db * 4)                   # first row whose bytes start beyond 2^31
    first_over = int(torch.searchsorted(window_row, torch.tensor(boundary_row, device=DEV, dtype=torch.int32)))
    for first, rows in ((0, B), (first_over - B // 2, B), (n - B, B), (n - 37, 37)):
        eng.gather(first, rows)
        torch.cuda.synchronize()
        r = window_row[first: first + rows].long()
        te_in = eng.panel("in", _lib.NET_TE)[:rows]
        wm_in = eng.panel("in", _lib.NET_WM)[:rows]
        md_in = eng.panel("in", _lib.NET_MD)[:rows]
        assert torch.equal(te_in[:, :Db], states[r]) and torch.equal(te_in[:, Db: 2 * Db], states[r + 1])
        assert torch.equal(wm_in[:, :Db], states[r]) and torch.equal(wm_in[:, Db: Db + Da], actions[r])
        assert torch.equal(md_in[:, :Db], states[r])
        assert torch.equal(eng.panel("s2")[:rows, :Db], states[r + 1])
        assert torch.equal(eng.panel("act_t")[:rows, :Da], actions[r])
        assert float(te_in[:, 2 * Db:].abs().max()) == 0.0            # pad columns are zeros
        if rows < B:
            assert float(eng.panel("in", _lib.NET_TE)[rows: (rows + 31) // 32 * 32].abs().max()) == 0.0


def test_gather_of_cond_rel_windows_is_bit_exact(golden, tmp_path):
    """cond = "rel" datasets on the device: the gather kernel reads the second half of x and the target
    s2 from the row-aligned `next_states` array; the panels equal the dataset's windows bit for bit and
    a training epoch over them runs (the world model then learns state DIFFERENCES)."""
    from physicsvae_amd import train_physics_vae as T
    g = golden("ingest_rel_tiny")
    arch = arch_from_meta(g["meta"])
    Db, Da = arch["Db"], arch["Da"]
    data = R.synth_demo(3, 3, 12, Db, Da, kind="iid", quantum=0.0)
    pkl = str(tmp_path / "a.pkl")
    R.write_demo(pkl, data)
    tr = make_trainer(arch, data, 8, m_world=1, device=DEV)
    ds = T.load_dataset_for_PhysicsVAE([pkl], cond="rel")
    tr.train_loader.dataset = ds
    eng = tr.engine
    eng.bind_dataset(*ds.device_arrays(eng.device))
    for first, rows in ((0, 8), (25, 8), (32, 1)):
        eng.gather(first, rows)
        torch.cuda.synchronize()
        xs = torch.stack([ds[i][0][0] for i in range(first, first + rows)]).to(DEV)      # [rows, 2Db]
        ys = torch.stack([ds[i][1][0] for i in range(first, first + rows)]).to(DEV)
        assert torch.equal(eng.panel("in", _lib.NET_TE)[:rows, : 2 * Db], xs)
        assert torch.equal(eng.panel("s2")[:rows, :Db], xs[:, Db:])
        assert torch.equal(eng.panel("in", _lib.NET_WM)[:rows, :Db], xs[:, :Db])
        assert torch.equal(eng.panel("in", _lib.NET_WM)[:rows, Db: Db + Da], ys)
    r1, r2 = tr.train(), tr.train()
    assert np.isfinite(r1["mean_train_loss"]) and np.isfinite(r2["mean_train_loss"])
    # against the oracle on the same windows
    X, Y = R.build_windows(data, cond="rel")
    x, y = next(iter(R.make_loader(X, Y, 8)))
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    want = R.loss_and_grads(arch, sd, x, y, None, world=True)
    c = R.phase_coeffs(True)
    from physicsvae_amd.engine import make_step_params
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=8)
    eng.gather(0, 8)
    got = eng.forward_backward(_lib.PHASE_WORLD, 8, sp, backward=False).cpu()
    assert float(got[0]) == pytest.approx(float(want["total"]), rel=1e-5)


@pytest.mark.parametrize("stop_after", [3, 1, 2])
def test_trainer_state_resume_is_bit_exact(golden, tmp_path, stop_after):
    """`save_trainer_state` / `resume_trainer_state` (ours; upstream resumes weights only, tm:215-216):
    a run interrupted after `stop_after` epochs (world -> joint switch at 2: after it, before it, exactly at
    it) and resumed from its checkpoint directory continues bit for bit like the uninterrupted run: weights,
    Adam moments, step counts, StepLR position, eps stream position, and the PHASE."""
    g = golden("train_tiny")
    arch = arch_from_meta(g["meta"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    kw = dict(m_world=2, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, arch["Z"]))
    full = make_trainer(arch, data, batch, extra={"save_trainer_state": True}, **kw)
    full.model.load_state_dict(sd)
    for _ in range(stop_after):
        full.train()
    ck = full.save_checkpoint(str(tmp_path))
    assert os.path.exists(tmp_path / "trainer_state.pt")
    learnable_at_save = sorted(full.model.learnable_nets())
    want = [full.train()["mean_train_loss"] for _ in range(3)]
    res = make_trainer(arch, data, batch, extra={"resume_trainer_state": True}, **kw)
    res.restore(ck)
    assert res.iter == stop_after
    # the restore itself re-entered the phase the state was saved in (joint when saved after the switch)
    assert sorted(res.model.learnable_nets()) == learnable_at_save
    assert (res.a_rec_coeff > 0) == (stop_after > 2)
    got = [res.train()["mean_train_loss"] for _ in range(3)]
    assert got == want
    assert torch.equal(res.engine.params, full.engine.params)
    assert torch.equal(res.engine.exp_avg, full.engine.exp_avg) and torch.equal(res.engine.exp_avg_sq, full.engine.exp_avg_sq)
    assert res.optimizer.net_steps == full.optimizer.net_steps and res.optimizer.lr == full.optimizer.lr
