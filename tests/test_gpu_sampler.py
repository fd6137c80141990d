"""The reparameterisation sampler (rmt:734-740): folded into the decoder's first-layer launch vs its own launch, and the
statistics of the Philox stream that stands in for torch.randn_like."""
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


@pytest.mark.parametrize("name,rows", [("single_c2", 256), ("single_c2", 200), ("single_default", 192), ("single_default", 129),
                                       ("single_c1", 64), ("single_tiny", 8)])       # (the last two: 16x16-tile layers, no fold)
def test_sampler_folded_into_the_decoder_launch_equals_the_sampler_launch(golden, monkeypatch, name, rows):
    """PVAE_FOLD_SAMPLER=1: every workgroup of the decoder's first-layer launch forms z = mu + eps exp(logvar / 2)
    for its own 32 rows and patches it over the z columns of its input tile in LDS; column tile 0 stores z, the draws
    and the KL partial of its row block (rmt:734-740, tpv:384-389).  Against the default schedule (sampler launch):
    z, the recorded draws and the decoder's output bit for bit -- with supplied draws and with Philox draws --, the
    KL term to summation order, every gradient to fp32 noise; the oracle check runs on the folded path too."""
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, max(rows, 1))))
    x, y = x[:rows], y[:rows]
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    eps = R.eps_stream(2, arch["Z"])(0, (rows, arch["Z"]))
    trs = []
    for fold in ("0", "1"):
        monkeypatch.setenv("PVAE_FOLD_SAMPLER", fold)
        tr = make_trainer(arch, data, max(rows, 8), device=DEV)
        tr.model.load_state_dict(sd)
        trs.append(tr)
    monkeypatch.delenv("PVAE_FOLD_SAMPLER")
    c = R.phase_coeffs(False)
    nets = [_lib.NET_TE, _lib.NET_MD]
    for e in (eps, None):                                    # supplied draws, Philox draws
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=rows, seed=77, offset=5)
        got = []
        for tr in trs:
            eng = tr.engine
            eng.grads.zero_()
            eng.set_batch(x, y)
            loss = eng.forward_backward(_lib.PHASE_JOINT, rows, sp, eps=e, fused_adam=False).clone()
            got.append((loss.cpu(), eng.read("z", rows).cpu(), eng.read("eps", rows).cpu(), eng.read("a_hat", rows).cpu(),
                        eng.segment(eng.grads, nets).clone().cpu()))
        (l0, z0, e0, a0, g0), (l1, z1, e1, a1, g1) = got
        assert torch.equal(z0, z1) and torch.equal(e0, e1) and torch.equal(a0, a1)
        assert torch.allclose(l0, l1, rtol=2e-6, atol=1e-9)
        assert max_err_scaled(g1, g0) < 2e-6
        if e is not None:
            assert torch.equal(e1, eps)
    # the folded path against the oracle (samples off the ReLU kinks)
    keep = R.relu_kink_margin(arch, sd, x, y, eps, False) > 4e-6
    xs, ys, es = x[keep], y[keep], eps[keep]
    n = xs.shape[0]
    if n > 4:
        want = R.loss_and_grads(arch, sd, xs, ys, es, False)
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=n)
        eng = trs[1].engine
        eng.set_batch(xs, ys)
        loss = eng.forward_backward(_lib.PHASE_JOINT, n, sp, eps=es, fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        assert float(loss[2]) == pytest.approx(float(want["loss_kl"]), rel=1e-5, abs=1e-9)
        gv = eng.named_views(eng.grads)
        for k, gr in want["grads"].items():
            if k.startswith("_world_model") or k.startswith("_value_branch"):
                continue
            assert max_err_scaled(gv[k].cpu(), gr) < 1e-4, k


def test_philox_draws_four_per_call_are_standard_normal_and_independent_across_columns():
    """The Philox stream after round 3 (one call = the four draws of columns 4g .. 4g + 3 of a row: two Box-Muller
    pairs on hardware log / sqrt / sin / cos): moments of a standard normal, no correlation between the columns of a
    group (cos / sin partners, the two pairs of a call) nor between neighbouring rows, keyed by (seed, offset)."""
    from physicsvae_amd.engine import Arch, HipEngine
    Z, rows = 32, 256
    eng = HipEngine(Arch(16, 4, Z, (64, 1), (64, 1), (64, 1)), rows, device=DEV)
    ml = torch.zeros(rows, 2 * Z, device=DEV)              # mu = 0, logvar = 0: z IS the draw
    draws = [eng.reparam(ml, noise=True, seed=9, offset=o).double().cpu() for o in range(64)]
    e = torch.stack(draws)                                  # [64, rows, Z]
    flat = e.reshape(-1)
    assert abs(float(flat.mean())) < 5e-3 and abs(float(flat.std()) - 1.0) < 5e-3
    assert abs(float((flat ** 3).mean())) < 2e-2 and abs(float((flat ** 4).mean()) - 3.0) < 7e-2
    assert float(flat.abs().max()) < 7.0 and float((flat.abs() > 3).double().mean()) == pytest.approx(0.0027, abs=6e-4)
    cols = e.reshape(-1, Z)
    cc = np.corrcoef(cols.numpy().T)
    assert float(np.abs(cc - np.eye(Z)).max()) < 0.04       # (16 384 samples per column: sigma = 0.008, 496 pairs) incl. the (cos, sin) partners
    r2 = cols[:, 0::2] ** 2 + cols[:, 1::2] ** 2            # pair radii^2 ~ chi^2_2: mean 2
    assert float(r2.mean()) == pytest.approx(2.0, abs=0.02)
    rr = np.corrcoef(e[:, :-1].reshape(-1).numpy(), e[:, 1:].reshape(-1).numpy())[0, 1]
    assert abs(float(rr)) < 7e-3                            # neighbouring rows
    again = eng.reparam(ml, noise=True, seed=9, offset=3).double().cpu()
    assert torch.equal(again, draws[3]) and not torch.equal(draws[3], draws[4])
    other = eng.reparam(ml, noise=True, seed=10, offset=3).double().cpu()
    assert not torch.equal(other, draws[3])
