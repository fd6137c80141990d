"""The call-persistent rollout server (include/pvae.h pvae_rollout_server_*): one kernel resident on
one XCD with the encoder's and decoder's weights in LDS, a mailbox in pinned host memory, zero launches per call.
What it replaces is PhysicsVAE.forward at B = 1 (rmt:742-771; callers envs/rllib_env_imitation.py:215-266), so every
check is against the per-layer launch path `pvae_infer`, bit for bit.

Every test stops the server before it returns (a resident kernel makes device-wide synchronisations wait for it) and
uses a short idle time-out, so that even a failing assertion cannot leave the GPU occupied for long."""
import contextlib
import time

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


@contextlib.contextmanager
def served(eng, idle_ms=200.0, lifetime_s=30.0, scope="auto"):
    eng.rollout_server_start(idle_ms=idle_ms, lifetime_s=lifetime_s, scope=scope)
    try:
        yield eng
    finally:
        eng.rollout_server_stop()


def _default_trainer(batch=8, seed=1):
    arch = R.make_arch(197, 45)                    # DEFAULT_CONFIG stacks: TE 2x256, MD 3x512 (3.4 MB of weights)
    data = R.synth_demo(0, 2, 40, 197, 45, kind="dynamics")
    tr = make_trainer(arch, data, batch, device=DEV)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=seed), seed=3))
    X, _ = R.build_windows(data)
    obs = torch.from_numpy(np.asarray(X)).float()[:, 0, :]        # [N, 2 Db]
    return arch, tr, obs


@pytest.mark.parametrize("mailbox", ["auto", "host"])
@pytest.mark.parametrize("noise", [False, True])
def test_server_equals_the_launch_path_bit_for_bit(noise, mailbox, monkeypatch):
    """Both homes of the request block: device memory the host writes through the BAR (what a large-BAR machine gets by
    itself) and pinned host memory the kernel pulls from."""
    if mailbox != "auto":
        monkeypatch.setenv("PVAE_SERVER_MAILBOX", mailbox)
    arch, tr, obs = _default_trainer()
    eng = tr.engine
    with served(eng):
        serving, _, lds = eng.rollout_server_status()
        assert serving and 64 * 1024 < lds <= 156 * 1024 and eng.rollout_server_scope() == "xcd"
        mode = eng.rollout_server_mailbox()
        assert mode in ("host", "device") and (mailbox != "host" or mode == "host")
        for i in range(12):
            o = obs[i]
            a, ml, z = (x.copy() for x in eng.rollout_server_infer(o.numpy(), noise=noise, seed=11, offset=1000 + i))
            want_a, _, want_z = eng.infer(o[None].to(DEV), noise=noise, seed=11, offset=1000 + i, want_s2=False)
            assert np.array_equal(a, want_a.cpu().numpy()[0]), i
            assert np.array_equal(z, want_z.cpu().numpy()[0]), i
            assert np.array_equal(ml[: arch["Z"]], eng.read("mu", 1).cpu().numpy()[0]), i
            assert np.array_equal(ml[arch["Z"]:], eng.read("logvar", 1).cpu().numpy()[0]), i
            if not noise:
                assert np.array_equal(z, ml[: arch["Z"]])             # z = mu (latent_prior_noise False)
        assert eng.rollout_server_status()[1] == 12
    assert eng.rollout_server_status()[0] is False
    # and the action is the model's: against the oracle's forward at 2e-5 (as every rollout test)
    sd = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
    m = R.RefModel(arch)
    m.load_state_dict(sd)
    m.eps_source = lambda shape: torch.zeros(shape)
    with torch.no_grad():
        logits = m(obs[:1])
    with served(eng):
        a, _, _ = eng.rollout_server_infer(obs[0].numpy(), noise=False)
        assert np.abs(a - logits[0, : arch["Da"]].numpy()).max() <= 2e-5 * max(1.0, float(logits.abs().max()))


def test_server_follows_the_weights_on_reload_and_after_a_restart():
    arch, tr, obs = _default_trainer()
    eng = tr.engine
    o = obs[3].numpy()
    with served(eng, idle_ms=3000.0):
        before = eng.rollout_server_infer(o, noise=False)[0].copy()
        # an optimizer step through the library moves the decoder: the next request waits for the step's stream and
        # re-reads the weights by itself (no synchronisation, no reload by the caller)
        sp = make_step_params(lr=5e-3, global_rows=8)
        eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
        tr.model.set_learnable_task_encoder(True)
        tr.model.set_learnable_motor_decoder(True)
        tr.model.set_learnable_world_model(False)
        eng.train_step(_lib.PHASE_JOINT, 0, 8, sp)
        after = eng.rollout_server_infer(o, noise=False)[0].copy()
        want = eng.infer(torch.from_numpy(o)[None].to(DEV), noise=False, want_s2=False)[0].cpu().numpy()[0]
        assert not np.array_equal(want, before)
        assert np.array_equal(after, want)
        assert np.array_equal(eng.rollout_server_infer(o, noise=False)[0], want)            # and they stay
        # a write the library cannot see (here: a direct write into the arena) is stale until announced
        with torch.no_grad():
            eng.params.mul_(0.5)
        torch.cuda.current_stream().synchronize()
        want2 = eng.infer(torch.from_numpy(o)[None].to(DEV), noise=False, want_s2=False)[0].cpu().numpy()[0]
        assert not np.array_equal(want2, want)
        assert np.array_equal(eng.rollout_server_infer(o, noise=False)[0], want)            # still the old weights
        assert np.array_equal(eng.rollout_server_infer(o, noise=False, reload=True)[0], want2)
        with torch.no_grad():
            eng.params.mul_(2.0)                                                            # (exact: back to the trained weights)
        eng.params_changed(torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(eng.rollout_server_infer(o, noise=False)[0], want)
        assert np.array_equal(eng.rollout_server_decode(np.concatenate([o[: arch["Db"]], np.zeros(arch["Z"], np.float32)])),
                              eng.net_forward(_lib.NET_MD, torch.cat([torch.from_numpy(o[: arch["Db"]]), torch.zeros(arch["Z"])])[None].to(DEV))
                              [:, : arch["Da"]].cpu().numpy()[0])
    # idle time-out: the kernel leaves by itself, the next call brings it back (weights read from the arena again)
    with served(eng, idle_ms=30.0):
        assert np.array_equal(eng.rollout_server_infer(o, noise=False)[0], want)
        time.sleep(0.3)
        assert eng.rollout_server_status()[0] is False
        assert np.array_equal(eng.rollout_server_infer(o, noise=False)[0], want)
        assert eng.rollout_server_status()[0] is True
        served_before = eng.rollout_server_status()[1]
        for i in range(5):
            eng.rollout_server_infer(obs[i].numpy(), noise=True, seed=3, offset=i)
        assert eng.rollout_server_status()[1] == served_before + 5


def test_module_forward_served_equals_module_forward_launched():
    """PhysicsVAE.forward (rmt:742-771) with a one-row CPU observation, served by the resident kernel, against the same
    module's launch path: logits [a_hat | log_std], z, mu / logvar bit-identical; the lazily evaluated prediction and
    value estimate agree; `load_state_dict` reaches the resident copy of the weights."""
    arch, tr, obs = _default_trainer()
    m = tr.model
    m.eval()
    m.set_exploration_std(0.2)

    def call(o):
        with torch.no_grad():
            logits, _ = m.forward({"obs_flat": o}, [], None)
            return (logits.cpu().clone(), m.task_encoder_variable().cpu().clone(), m._cur_task_encoder_mu.cpu().clone(),
                    m._cur_task_encoder_logvar.cpu().clone(), m._cur_future_state.cpu().clone(), m.value_function().cpu().clone())
    for noise in (False, True):
        m.latent_prior_noise = noise
        m._st._rng_calls = 100
        want = [call(obs[i:i + 1].to(DEV)) for i in range(4)]
        m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
        try:
            m._st._rng_calls = 100
            got = [call(obs[i:i + 1]) for i in range(4)]
        finally:
            m.stop_rollout_server()
        for w, g in zip(want, got):
            for a, b in zip(w, g):
                assert torch.equal(a, b)
    # new weights through the module API reach the kernel's LDS copy
    sd2 = R.perturb_biases(R.init_state_dict(arch, seed=9), seed=4)
    m.latent_prior_noise = False
    m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        first = call(obs[:1])[0]
        m.load_state_dict(sd2)
        second = call(obs[:1])[0]
    finally:
        m.stop_rollout_server()
    assert not torch.equal(first, second)
    assert torch.equal(second, call(obs[:1].to(DEV))[0])
    # ... also when the FIRST request after the load is a decoder-only one
    Db, Z = arch["Db"], arch["Z"]
    s1, z = obs[:1, :Db], torch.full((1, Z), 0.25)
    m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        with torch.no_grad():
            d1 = m.forward_decoder(s1, z)[0].clone()
            m.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
            d2 = m.forward_decoder(s1, z)[0].clone()
    finally:
        m.stop_rollout_server()
    with torch.no_grad():
        want = m.forward_decoder(s1.to(DEV), z.to(DEV))[0].cpu()
    assert not torch.equal(d1, d2) and torch.equal(d2, want)


@pytest.mark.parametrize("noise", [False, True])
def test_served_forward_honours_rollout_predicts_state(noise):
    """`rollout_predicts_state` on the SERVED path (rmt:758: upstream evaluates forward_world with every forward): True --
    the prediction is there when forward returns; "lazy" -- on first read; False -- never.  In each case it is the launch
    path's prediction for the same observation and draws, bit for bit."""
    arch, tr, obs = _default_trainer()
    m = tr.model
    m.eval()
    m.latent_prior_noise = noise
    m._st._rng_calls = 40
    with torch.no_grad():
        m.forward({"obs_flat": obs[:1].to(DEV)}, [], None)
        want = m._cur_future_state.cpu().clone()
    m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        with torch.no_grad():
            for setting in (True, "lazy", False):
                m.rollout_predicts_state = setting
                m._st._rng_calls = 40
                m.forward({"obs_flat": obs[:1]}, [], None)
                if setting is True:
                    assert m._st._cur_future_state is not None              # evaluated with the forward, as upstream
                    assert torch.equal(m._st._cur_future_state.cpu(), want)
                elif setting == "lazy":
                    assert m._st._cur_future_state is None
                    assert torch.equal(m._cur_future_state.cpu(), want)
                else:
                    assert m._cur_future_state is None
    finally:
        m.rollout_predicts_state = "lazy"
        m.stop_rollout_server()


@pytest.mark.parametrize("mailbox", ["auto", "host"])
def test_decoder_only_requests_equal_forward_decoder(mailbox, monkeypatch):
    """The "pass_through" rollout (envs/rllib_env_imitation.py:233-258): the caller draws z itself and calls
    `forward_decoder(z_body, z_task)` (rmt:822-837).  Served: the encoder's layers are skipped, the action equals the
    launch path's `net_forward(NET_MD, [s1 | z])` bit for bit -- at the engine and at the module surface, interleaved with
    full requests."""
    if mailbox != "auto":
        monkeypatch.setenv("PVAE_SERVER_MAILBOX", mailbox)
    arch, tr, obs = _default_trainer()
    eng, m = tr.engine, tr.model
    Db, Z = arch["Db"], arch["Z"]
    g = torch.Generator().manual_seed(3)
    m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        for i in range(8):
            s1, z = obs[i, :Db], torch.randn(Z, generator=g)
            x = torch.cat([s1, z])[None]
            want = eng.net_forward(_lib.NET_MD, x.to(DEV))[:, : arch["Da"]].cpu()
            got = eng.rollout_server_decode(x.numpy()).copy()
            assert np.array_equal(got, want.numpy()[0]), i
            with torch.no_grad():
                logits, _ = m.forward_decoder(s1[None], z[None])
            assert logits.device.type == "cpu" and torch.equal(logits[:, : arch["Da"]], want)
            a = eng.rollout_server_infer(obs[i].numpy(), noise=False)[0].copy()        # a full request in between
            assert np.array_equal(a, eng.infer(obs[i][None].to(DEV), noise=False, want_s2=False)[0].cpu().numpy()[0])
    finally:
        m.stop_rollout_server()
    with torch.no_grad():
        logits_launch, _ = m.forward_decoder(obs[7:8, :Db].to(DEV), z[None].to(DEV))     # the launch path, same module call
    assert torch.equal(logits_launch.cpu(), logits)


def test_two_engines_serve_side_by_side():
    """Two models in one process, each with a resident server: they take different XCDs and answer independently."""
    arch, tr1, obs = _default_trainer(seed=1)
    _, tr2, _ = _default_trainer(seed=7)
    e1, e2 = tr1.engine, tr2.engine
    with served(e1, idle_ms=2000.0), served(e2, idle_ms=2000.0):
        for i in range(6):
            o = obs[i]
            a1 = e1.rollout_server_infer(o.numpy(), noise=True, seed=2, offset=i)[0].copy()
            a2 = e2.rollout_server_infer(o.numpy(), noise=True, seed=2, offset=i)[0].copy()
            assert np.array_equal(a1, e1.infer(o[None].to(DEV), noise=True, seed=2, offset=i, want_s2=False)[0].cpu().numpy()[0])
            assert np.array_equal(a2, e2.infer(o[None].to(DEV), noise=True, seed=2, offset=i, want_s2=False)[0].cpu().numpy()[0])
            assert not np.array_equal(a1, a2)


def test_server_coexists_with_launches_on_the_compute_stream():
    """Training steps and per-layer rollout launches run while the server is resident (its stream has a hardware queue
    of its own; the other XCDs' CUs are free), and requests are answered in between."""
    arch, tr, obs = _default_trainer()
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    sp = make_step_params(lr=5e-4, global_rows=8, s_rec=1.0, a_rec=0.0, kl=0.0, cyc=0.0)
    with served(eng, idle_ms=500.0):
        t0 = time.perf_counter()
        for i in range(20):
            eng.train_step(_lib.PHASE_WORLD, 0, 8, sp)                  # world phase: encoder / decoder untouched
            a = eng.rollout_server_infer(obs[i].numpy(), noise=False)[0].copy()
            want = eng.infer(obs[i][None].to(DEV), noise=False, want_s2=False)[0]
            torch.cuda.current_stream().synchronize()
            assert np.array_equal(a, want.cpu().numpy()[0]), i
        assert time.perf_counter() - t0 < 5.0                           # nothing waited for an idle time-out


def _big_trainer():
    arch = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    data = R.synth_demo(0, 2, 20, 197, 45, kind="iid")
    tr = make_trainer(arch, data, 8, device=DEV)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=2), seed=5))
    X, _ = R.build_windows(data)
    return arch, tr, torch.from_numpy(np.asarray(X)).float()[:, 0, :]


def test_server_refuses_what_does_not_fit_and_the_launch_path_stays():
    """4x1024 stacks need 459 KB per workgroup on ONE XCD: refused by name when that scope is forced; stacks that fit
    nowhere (4x2048: 390 KB per workgroup even over all 256 CUs) are refused in every scope.  The launch path is untouched."""
    arch, tr, obs = _big_trainer()
    eng = tr.engine
    with pytest.raises(RuntimeError, match="LDS"):
        eng.rollout_server_start(scope="xcd")
    assert eng.rollout_server_status()[0] is False
    with pytest.raises(RuntimeError, match="not started"):
        eng.rollout_server_infer(np.zeros(394, np.float32))
    with pytest.raises(RuntimeError, match="not started"):
        eng.rollout_server_selfbench(np.zeros(394, np.float32), n=3)
    assert torch.isfinite(eng.infer(obs[:1].to(DEV), noise=False)[0]).all()
    huge = R.make_arch(197, 45, latent=32, te=(2048, 4), md=(2048, 4), wm=(256, 2))
    tr2 = make_trainer(huge, R.synth_demo(0, 2, 20, 197, 45, kind="iid"), 8, device=DEV)
    with pytest.raises(RuntimeError, match="all 256 CUs"):
        tr2.engine.rollout_server_start()


@pytest.mark.parametrize("which", ["4x1024", "default"])
@pytest.mark.parametrize("noise", [False, True])
def test_chip_wide_server_equals_the_launch_path_bit_for_bit(which, noise):
    """Stacks too big for one XCD's LDS (4x1024: 28 MB of weights) are dealt out over all 256 CUs, 1/256 of every layer's
    features each; the hand-over words then cross XCDs (agent-scope stores / loads: 0.5-0.64 us per hop against 0.43-0.53
    inside an XCD, tools/xcd_pingpong.hip).  Same action, mu / logvar, z as the launch path, bit for bit -- also for the
    default stacks when the chip-wide scope is asked for."""
    arch, tr, obs = _big_trainer() if which == "4x1024" else _default_trainer()
    eng = tr.engine
    with served(eng, scope="auto" if which == "4x1024" else "chip"):
        assert eng.rollout_server_scope() == "chip" and eng.rollout_server_status()[2] <= 156 * 1024
        for i in range(10):
            o = obs[i]
            a, ml, z = (x.copy() for x in eng.rollout_server_infer(o.numpy(), noise=noise, seed=5, offset=77 + i))
            want_a, _, want_z = eng.infer(o[None].to(DEV), noise=noise, seed=5, offset=77 + i, want_s2=False)
            assert np.array_equal(a, want_a.cpu().numpy()[0]), i
            assert np.array_equal(z, want_z.cpu().numpy()[0]), i
            assert np.array_equal(ml[: arch["Z"]], eng.read("mu", 1).cpu().numpy()[0]), i
            assert np.array_equal(ml[arch["Z"]:], eng.read("logvar", 1).cpu().numpy()[0]), i
        us = np.sort(eng.rollout_server_selfbench(obs[0].numpy(), n=300))
        print("%s, chip-wide server: %.1f us median inside the library call" % (which, us[len(us) // 2]))
    assert eng.rollout_server_status()[0] is False


def test_server_latency_is_below_the_launch_path():
    """Host observation -> host action at B = 1: the server against `infer_host` (7 launches, kernels reading / writing
    pinned host memory).  Medians over 300 calls each; the server must be clearly faster (the figure itself is
    measured by tools/infer_latency.py and committed under profiles/)."""
    arch, tr, obs = _default_trainer()
    eng = tr.engine
    o = obs[0].numpy()
    for _ in range(20):
        eng.infer_host(o[None], noise=True, seed=1, offset=5)
    t_launch = []
    for i in range(300):
        t0 = time.perf_counter()
        eng.infer_host(o[None], noise=True, seed=1, offset=i)
        t_launch.append(time.perf_counter() - t0)
    with served(eng):
        for _ in range(20):
            eng.rollout_server_infer(o, noise=True, seed=1, offset=5)
        t_srv = []
        for i in range(300):
            t0 = time.perf_counter()
            eng.rollout_server_infer(o, noise=True, seed=1, offset=i)
            t_srv.append(time.perf_counter() - t0)
    med_l, med_s = np.median(t_launch) * 1e6, np.median(t_srv) * 1e6
    print("host -> host: launches %.1f us, server %.1f us" % (med_l, med_s))
    assert med_s < 0.9 * med_l, (med_l, med_s)
    with served(eng):
        us = np.sort(eng.rollout_server_selfbench(o, n=500))
        print("inside the library call: %.1f us median" % us[len(us) // 2])
        assert us[len(us) // 2] < med_s + 1.0


@pytest.mark.parametrize("noise", [False, True])
def test_server_serves_a_helper_model_bit_for_bit(noise):
    """`motor_decoder_helper_enable` (rmt:670-680, 833-835): the helper's layers follow the decoder's in the resident kernel
    (their first layer reads the decoder's input again), workgroup 0 adds range * h with helper_add_kernel's expression.
    Same action as the launch path `pvae_infer` (which adds the term in the library), bit for bit -- also for decoder-only
    requests -- and the module's served forward equals its launched forward."""
    from test_gpu_helper_training import _trainer, _weights
    base = R.make_arch(197, 45)
    h, sd = _weights(base)
    data = R.synth_demo(0, 2, 40, 197, 45, kind="dynamics")
    tr = _trainer(base, data, 8, device=DEV)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    X, _ = R.build_windows(data)
    obs = torch.from_numpy(np.asarray(X)).float()[:, 0, :]
    ref = R.RefModel(h)
    ref.load_state_dict(sd)
    ref.latent_prior_noise = False
    with served(eng):
        assert eng.rollout_server_status()[0]
        for i in range(8):
            o = obs[i]
            a, ml, z = (x.copy() for x in eng.rollout_server_infer(o.numpy(), noise=noise, seed=11, offset=1000 + i))
            want_a, _, want_z = eng.infer(o[None].to(DEV), noise=noise, seed=11, offset=1000 + i, want_s2=False)
            assert np.array_equal(a, want_a[0].cpu().numpy()) and np.array_equal(z, want_z[0].cpu().numpy())
            if not noise:
                with torch.no_grad():
                    lg = ref(o[None])
                assert float(np.abs(a - lg[0, :45].numpy()).max()) < 2e-5 * max(1.0, float(lg.abs().max()))
            # decoder only ("pass_through"): the caller supplies z
            sz = torch.cat([o[:197], torch.from_numpy(z)])
            a2 = eng.rollout_server_decode(sz.numpy()).copy()
            assert np.array_equal(a2, a)
            # ... and against forward_decoder on the LAUNCH path (pvae_net_forward(MD) + range * helper added in torch): the
            # kernel forms decoder + range * h with one fma, so the two agree to an ulp of the action, not to the bit
            # (include/pvae.h pvae_rollout_server_decode)
            with torch.no_grad():
                lg2, _ = tr.model.forward_decoder(o[None, :197].to(DEV), torch.from_numpy(z)[None].to(DEV))
            want2 = lg2[0, :45].cpu().numpy()
            assert float(np.abs(a2 - want2).max()) <= 4e-7 * max(1.0, float(np.abs(want2).max()))
    # the module surface
    tr.model.eval()
    tr.model.latent_prior_noise = noise
    tr.model._st._rng_calls = 100
    with torch.no_grad():
        launched = [tr.model.forward({"obs_flat": obs[i][None].to(DEV)}, [], None)[0].cpu() for i in range(3)]
    tr.model._st._rng_calls = 100
    tr.model.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        for i in range(3):
            lg, _ = tr.model.forward({"obs_flat": obs[i][None]}, [], None)
            assert torch.equal(lg, launched[i])
    finally:
        tr.model.stop_rollout_server()


@pytest.mark.parametrize("helper", [False, True])
@pytest.mark.parametrize("noise", [False, True])
def test_server_serves_two_to_four_rows_in_one_request(noise, helper):
    """pvae_rollout_server_infer_rows: 1 <= rows <= 4 observations per request (rmt:742-771 serves any batch); every weight
    fragment read from LDS feeds all rows (2 or 4 accumulators per feature; 3 rows run the 4-row body).  Action, mu / logvar
    and z of every row equal `pvae_infer` on the same rows (row r draws Philox row r), bit for bit -- with and without a
    helper stack -- and the single-row request in between still equals its own launch path."""
    if helper:
        from test_gpu_helper_training import _trainer, _weights
        base = R.make_arch(197, 45)
        h, sd = _weights(base)
        data = R.synth_demo(0, 2, 40, 197, 45, kind="dynamics")
        tr = _trainer(base, data, 8, device=DEV)
        tr.model.load_state_dict(sd)
        X, _ = R.build_windows(data)
        obs = torch.from_numpy(np.asarray(X)).float()[:, 0, :]
    else:
        _, tr, obs = _default_trainer()
    eng = tr.engine
    Z = eng.arch.Z
    with served(eng):
        for i, rows in enumerate((2, 3, 4, 1, 4, 2)):
            o = obs[i: i + rows]
            a, ml, z = eng.rollout_server_infer_rows(o.numpy(), noise=noise, seed=11, offset=2000 + i)
            want_a, _, want_z = eng.infer(o.to(DEV), noise=noise, seed=11, offset=2000 + i, want_s2=False)
            assert a.shape == (rows, 45) and np.array_equal(a, want_a.cpu().numpy())
            assert np.array_equal(z, want_z.cpu().numpy())
            assert np.array_equal(ml[:, :Z], eng.read("mu", rows).cpu().numpy())
            a1, _, z1 = (x.copy() for x in eng.rollout_server_infer(o[0].numpy(), noise=noise, seed=11, offset=2000 + i))
            assert np.array_equal(a1, a[0]) and np.array_equal(z1, z[0])       # (row 0 of a batch == the single-row request)
        with pytest.raises(RuntimeError):
            eng.rollout_server_infer_rows(obs[:5].numpy())
    # the module surface: 3 CPU rows, served == launched
    m = tr.model
    m.eval()
    m.latent_prior_noise = noise
    m._st._rng_calls = 50
    with torch.no_grad():
        want = m.forward({"obs_flat": obs[:3].to(DEV)}, [], None)[0].cpu()
        want_z, want_s2 = m.task_encoder_variable().cpu().clone(), m._cur_future_state.cpu().clone()
    m._st._rng_calls = 50
    m.start_rollout_server(idle_ms=2000.0, lifetime_s=30.0)
    try:
        with torch.no_grad():
            got = m.forward({"obs_flat": obs[:3]}, [], None)[0]
            assert torch.equal(got, want) and torch.equal(m.task_encoder_variable().cpu(), want_z)
            assert torch.equal(m._cur_future_state.cpu(), want_s2)
    finally:
        m.stop_rollout_server()
