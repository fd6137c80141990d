"""Randomised-shape sweep of the HIP path against the oracle: stack widths that are not multiples
of any tile, depths 1..4 that differ per stack (a second sweep: widths and activations that differ per LAYER), tiny and odd observation / action / latent sizes,
minibatches from 1 row up, lookahead 1..3, MSE and L1.  Every case: loss terms rel 1e-5, every
gradient 1e-4 (samples on a ReLU kink excluded, see oracle.refpath.relu_kink_margin), the pad
entries of the arena stay zero, and three optimizer steps with Adam fused into the weight-gradient
launches equal gradient-store + flat Adam bit for bit.  Seeds are fixed: the sweep is deterministic."""
import os

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer, max_err_scaled, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    Db = int(r.integers(2, 72))
    Da = int(r.integers(1, 34))
    Z = int(r.integers(1, 24))
    nets = [(int(r.choice([5, 17, 31, 64, 100, 129, 200, 257, 320])), int(r.integers(1, 5))) for _ in range(3)]
    L = int(r.choice([1, 1, 1, 2, 3]))
    rows = int(r.choice([1, 2, 7, 31, 32, 33, 64, 97, 130]))
    loss = str(r.choice(["MSE", "MSE", "L1"]))
    return dict(Db=Db, Da=Da, Z=Z, te=nets[0], md=nets[1], wm=nets[2], L=L, rows=rows, loss=loss)


def _case_stacks(seed):
    """As _case, with every stack given layer by layer: its own width and activation per hidden layer -- the lists
    FC accepts (rmt:234-270) but gen_layers cannot emit."""
    r = np.random.default_rng(7000 + seed)
    c = _case(seed)
    widths = [5, 17, 31, 64, 100, 129, 200, 257, 320]
    acts = ["relu", "relu", "tanh", "sigmoid", "elu", "linear"]
    for key in ("te", "md", "wm"):
        c[key] = [(int(r.choice(widths)), str(r.choice(acts))) for _ in range(int(r.integers(1, 5)))]
    return c


@pytest.mark.parametrize("seed", range(24))
def test_random_per_layer_stacks_match_oracle(seed):
    _check(_case_stacks(seed), seed)


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_match_oracle(seed):
    _check(_case(seed), seed)


def _check(c, seed):
    L, rows, lk = c["L"], c["rows"], c["loss"]
    arch = R.make_arch(c["Db"], c["Da"], latent=c["Z"], te=c["te"], md=c["md"], wm=c["wm"])
    n_steps = rows + L + 3
    data = R.synth_demo(seed, 2, n_steps, c["Db"], c["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    x, y = next(iter(R.make_loader(X, Y, rows)))
    assert x.shape[0] == rows
    sd = R.perturb_biases(R.init_state_dict(arch, seed=seed + 1), seed=seed + 3)
    es = R.eps_stream(seed + 2, c["Z"])
    eps = torch.stack([es(t, (rows, c["Z"])) for t in range(L)])
    tr = make_trainer(arch, data, rows, device="cuda", extra={"lookahead": L, "loss": lk})
    tr.model.load_state_dict(sd)
    eng = tr.engine
    for world in (True, False):
        phase = _lib.PHASE_WORLD if world else _lib.PHASE_JOINT
        co = R.phase_coeffs(world)
        keep = R.relu_kink_margin(arch, sd, x, y, eps, world) > 4e-6
        if int(keep.sum()) == 0:
            continue
        xk, yk, ek = x[keep], y[keep], eps[:, keep]
        n = xk.shape[0]
        want = R.loss_and_grads(arch, sd, xk, yk, ek if L > 1 else ek[0], world, loss=lk)
        sp = make_step_params(lr=5e-4, a_rec=co["a_rec_coeff"], kl=co["vae_kl_coeff"], s_rec=co["s_rec_coeff"],
                              cyc=co["vae_cycle_coeff"], global_rows=n, loss=lk)
        eng.set_batch(xk, yk)
        eng.grads.fill_(float("nan"))
        e_in = (ek if L > 1 else ek[0]) if (not world or L > 1) else None
        loss = eng.forward_backward(phase, n, sp, eps=e_in, fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5, abs=1e-9), (c, world)
        for i, k in enumerate(("loss_a", "loss_kl", "loss_s", "loss_cyc")):
            assert float(loss[1 + i]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-9), (c, world, k)
        gv = eng.named_views(eng.grads)
        for k, gr in want["grads"].items():
            ours = gv[k].cpu()
            assert torch.isfinite(ours).all(), (c, world, k)
            assert max_err_scaled(ours, gr) < 1e-4, (c, world, k)
            assert rel_err(ours, gr) < 1e-4, (c, world, k)
        nets = [_lib.NET_WM] if world else [_lib.NET_TE, _lib.NET_MD]
        seg = eng.segment(eng.grads, nets)
        real = sum(gv[k].abs().double().sum().item() for k in want["grads"])
        assert seg.abs().double().sum().item() == pytest.approx(real, rel=1e-9), (c, world)   # pads stay zero

        def run(fused):
            tr.model.load_state_dict(sd)
            eng.exp_avg.zero_()
            eng.exp_avg_sq.zero_()
            for t in (1, 2, 3):
                spt = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=co["a_rec_coeff"], kl=co["vae_kl_coeff"],
                                       s_rec=co["s_rec_coeff"], cyc=co["vae_cycle_coeff"], global_rows=n, loss=lk)
                eng.set_batch(xk, yk)
                eng.forward_backward(phase, n, spt, eps=e_in, fused_adam=fused)
                if not fused:
                    eng.adam(nets, spt)
            return eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()
        p0 = eng.params.clone()
        a = run(True)
        eng.params.copy_(p0)
        b = run(False)
        for u, v in zip(a, b):
            assert torch.equal(u, v), (c, world)
        assert not torch.equal(a[0], p0)
        tr.model.load_state_dict(sd)


@pytest.mark.parametrize("switches", [{"PVAE_WGRAD32": "0", "PVAE_DGRAD16": "0"}, {"PVAE_WGRAD32": "2"}, {"PVAE_SAME_LAYER": "0"},
                                      {"PVAE_DEFER_ADAM": "0"}], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_schedule_and_geometry_switches_keep_parity(switches):
    """The narrow-layer geometries (32x32 weight-gradient tiles, 16x16 first-layer input-gradient
    tiles) are chosen per problem at launch time; PVAE_WGRAD32=0 / PVAE_DGRAD16=0 force the wide
    tiles everywhere and PVAE_WGRAD32=2 the narrow weight-gradient tiles everywhere.
    PVAE_SAME_LAYER=0 pairs wgrad_i with dgrad_{i-1} (the schedule used when the update cannot be
    deferred), PVAE_DEFER_ADAM=0 keeps Adam in every weight-gradient epilogue.  The geometry switches are process-wide
    options of the library (pvae_set_option(NULL, ...)), the schedule switches are read when a context is created: both are
    set here for the duration of the sweep above (24 of the 40 random shapes), in this process, and put back: every variant meets
    the same oracle tolerances and the same fused == flat Adam identity."""
    import os
    lib = _lib.load()
    defaults = {"PVAE_WGRAD32": 1, "PVAE_DGRAD16": 1}
    saved = {k: os.environ.get(k) for k in switches}
    try:
        for k, v in switches.items():
            if k in _lib.PROCESS_OPTIONS:
                key, conv = _lib.PROCESS_OPTIONS[k]
                assert lib.pvae_set_option(None, key.encode(), conv(v)) >= 0
            else:
                assert k in _lib.CONTEXT_OPTIONS
                os.environ[k] = v                      # (apply_context_options reads it when the trainer creates its context)
        for seed in range(24):                             # (the first 24 of the 40 shapes: every kernel family and both lookaheads)
            _check(_case(seed), seed)
    finally:
        for k, v in saved.items():
            if k in _lib.PROCESS_OPTIONS:
                key, _ = _lib.PROCESS_OPTIONS[k]
                lib.pvae_set_option(None, key.encode(), defaults[k])
            elif v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_64x32_tiles_equal_32x32_tiles_bit_for_bit(tmp_path):
    """At 512 rows and more the forward / stand-alone input-gradient launches use 64x32 output tiles
    (splitk_ws64_body: a quarter less LDS-DMA traffic per flop).  Every output element is still the sum of the same
    four k-quarters in the same order, so the two tilings must agree to the bit: the same contractions in two
    processes (PVAE_WS64 is read when the library loads), forward with bias + ReLU and input gradient with mask,
    at 512 and 1024 rows, K = 1024 and 448."""
    import subprocess
    import sys
    script = r'''
import sys, torch
sys.path.insert(0, %r)
from physicsvae_amd.engine import gemm_probe
outs = []
for m, n, k in ((512, 1024, 1024), (1024, 512, 448), (512, 1024, 256)):
    g = torch.Generator().manual_seed(m + n + k)
    x, w, b = torch.randn(m, k, generator=g).cuda(), torch.randn(n, k, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    dz, act = torch.randn(m, n, generator=g).cuda(), torch.randn(m, k, generator=g).cuda()
    o = torch.full((m, n), float("nan"), device="cuda")
    gemm_probe(0, x, w, o, bias_or_mask=b, relu=True, m=m, n=n, k=k)
    d = torch.full((m, k), float("nan"), device="cuda")
    gemm_probe(1, dz, w, d, bias_or_mask=act, m=m, n=n, k=k)
    outs += [o.cpu(), d.cpu()]
torch.save(outs, sys.argv[1])
''' % ROOT
    files = []
    for mode in ("0", "1"):
        f = str(tmp_path / ("out%s.pt" % mode))
        env = dict(os.environ, PVAE_WS64=mode)
        subprocess.run([sys.executable, "-c", script, f], check=True, env=env, timeout=300)
        files.append(torch.load(f))
    for a, b in zip(*files):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_64x64_tiles_equal_64x32_and_32x32_tiles_bit_for_bit(tmp_path):
    """At 1024 rows and more the forward / stand-alone input-gradient launches use 64x64 output tiles whenever those still
    give every CU a workgroup (splitk_ws64_body<PT = 64>: 16 flop per DMA byte where 64x32 has 10.7).  Same four k-quarters,
    same order: all three tilings agree to the bit -- forward with bias + ReLU and input gradient with mask, 1024 ... 4096
    rows, K = 1024 / 448 / 256, N = 1024 / 512 (the switches are read when the library loads: one process each)."""
    import subprocess
    import sys
    script = r'''
import sys, torch
sys.path.insert(0, %r)
from physicsvae_amd.engine import gemm_probe
outs = []
for m, n, k in ((1024, 1024, 1024), (2048, 512, 448), (1024, 1024, 256), (4096, 1024, 1024), (1088, 1024, 192)):
    g = torch.Generator().manual_seed(m + n + k)
    x, w, b = torch.randn(m, k, generator=g).cuda(), torch.randn(n, k, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    dz, act = torch.randn(m, n, generator=g).cuda(), torch.randn(m, k, generator=g).cuda()
    o = torch.full((m, n), float("nan"), device="cuda")
    gemm_probe(0, x, w, o, bias_or_mask=b, relu=True, m=m, n=n, k=k)
    outs.append(o.cpu())
    d = torch.full((m, k), float("nan"), device="cuda")
    gemm_probe(1, dz, w, d, bias_or_mask=act, m=m, n=n, k=k)
    outs.append(d.cpu())
torch.save(outs, sys.argv[1])
''' % ROOT
    files = []
    for env_add in ({"PVAE_WS64": "0"}, {"PVAE_WS6464": "0"}, {}):
        f = str(tmp_path / ("out%d.pt" % len(files)))
        subprocess.run([sys.executable, "-c", script, f], check=True, env=dict(os.environ, **env_add), timeout=300)
        files.append(torch.load(f))
    for a, b, c in zip(*files):
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c)


def test_whole_step_at_1024_rows_on_64x64_tiles_matches_the_oracle():
    """One minibatch of exactly 1024 kink-free rows through 1024-wide stacks -- every hidden forward layer and every
    stand-alone input gradient of the frozen world model on 64x64 tiles, the pairs on 64x32 -- against the oracle: losses
    1e-5, every gradient 1e-4 (both phases)."""
    arch = R.make_arch(23, 7, latent=8, te=(1024, 2), md=(1024, 3), wm=(1024, 2))
    data = R.synth_demo(5, 2, 700, 23, 7, kind="dynamics")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, 1300)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=2), seed=4)
    eps = R.eps_stream(3, 8)(0, (x.shape[0], 8))
    tr = make_trainer(arch, data, 1024, device="cuda")
    tr.model.load_state_dict(sd)
    eng = tr.engine
    for world in (True, False):
        keep = torch.nonzero(R.relu_kink_margin(arch, sd, x, y, eps, world) > 4e-6)[:, 0]
        assert keep.numel() >= 1024
        keep = keep[:1024]
        xk, yk, ek = x[keep], y[keep], eps[keep]
        co = R.phase_coeffs(world)
        want = R.loss_and_grads(arch, sd, xk, yk, ek, world)
        sp = make_step_params(lr=5e-4, a_rec=co["a_rec_coeff"], kl=co["vae_kl_coeff"], s_rec=co["s_rec_coeff"],
                              cyc=co["vae_cycle_coeff"], global_rows=1024)
        eng.set_batch(xk, yk)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, 1024, sp, eps=None if world else ek,
                                    fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5, abs=1e-9), world
        for i, k in enumerate(("loss_a", "loss_kl", "loss_s", "loss_cyc")):
            assert float(loss[1 + i]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-9), (world, k)
        gv = eng.named_views(eng.grads)
        for k, gr in want["grads"].items():
            ours = gv[k].cpu()
            assert torch.isfinite(ours).all(), (world, k)
            assert max_err_scaled(ours, gr) < 1e-4 and rel_err(ours, gr) < 1e-4, (world, k)


def test_pair_launch_with_64x32_input_gradient_tiles_is_bit_identical(tmp_path):
    """The fused backward pairs at 512 rows and more run their input-gradient half on 64x32 tiles (splitk_reg64_body,
    bwd_pair64_kernel; PVAE_PAIR64=0 keeps 32x32).  Same k-quarters per wave, same order of the four partial sums: a
    whole backward pass (world and joint phase, 512 and 576 rows, 1024-wide stacks) leaves bit-identical gradients,
    losses and -- through the fused step -- parameters and moments."""
    import subprocess
    import sys
    script = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer
arch = R.make_arch(23, 7, latent=8, te=(1024, 2), md=(1024, 3), wm=(1024, 2))
data = R.synth_demo(0, 2, 400, 23, 7, kind="dynamics")
outs = []
for rows in (512, 576):
    tr = make_trainer(arch, data, rows, device="cuda")
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, 1), 3))
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eps = R.eps_stream(2, 8)(0, (rows, 8))
    for phase, world in ((_lib.PHASE_WORLD, True), (_lib.PHASE_JOINT, False)):
        c = R.phase_coeffs(world)
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=rows)
        eng.gather(0, rows)
        eng.grads.zero_()
        loss = eng.forward_backward(phase, rows, sp, eps=eps, fused_adam=False).clone()
        outs += [loss.cpu(), eng.grads.clone().cpu()]
        out = torch.zeros(5, device="cuda")
        for t in (1, 2):
            spt = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                   s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=rows)
            eng.train_step(phase, 0, rows, spt, eps=eps, loss_out=out)
        outs += [eng.params.clone().cpu(), eng.exp_avg.clone().cpu()]
torch.save(outs, sys.argv[1])
''' % (ROOT, os.path.join(ROOT, "tests"))
    files = []
    for mode in ("0", "1"):
        f = str(tmp_path / ("pair%s.pt" % mode))
        env = dict(os.environ, PVAE_PAIR64=mode)
        subprocess.run([sys.executable, "-c", script, f], check=True, env=env, timeout=600, stdout=subprocess.DEVNULL)
        files.append(torch.load(f))
    assert len(files[0]) == 16
    for a, b in zip(*files):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert float(files[0][1].abs().sum()) > 0
