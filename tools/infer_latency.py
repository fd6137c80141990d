"""Rollout-path latency: PhysicsVAE forward at control-loop batch sizes (rmt:742-771)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import synth_demo, write_demo
from physicsvae_amd import train_physics_vae as T
import tempfile
td = tempfile.mkdtemp(prefix="pvae_infer_")
write_demo(os.path.join(td, "demo.pkl"), synth_demo(0, 2, 50, 197, 45))
QUICK = "--quick" in sys.argv                      # default sizes, one row: the control loop's case only
for name, sizes in ((("default 256x2/512x3/1024x2", []),) if QUICK else (("default 256x2/512x3/1024x2", []),          # the trainer's own defaults (tpv:264-280)
                    ("4x1024", [a for p in ("TE", "MD", "world_model") for a in ("--%s_width" % p, "1024", "--%s_depth" % p, "4")]))):
    T.args = T.arg_parser().parse_args(["--data_train", os.path.join(td, "demo.pkl"), "--batch_size", "32"] + sizes)
    cfg = T.get_trainer_config(T.args)
    cfg["model"]["custom_model_config"]["device"] = "cuda"
    with contextlib.redirect_stdout(io.StringIO()):
        tr = T.TrainModel(cfg)
    eng = tr.engine
    for rows in ((1,) if QUICK else (1, 4, 32)):
        obs = torch.randn(rows, 394, device="cuda")
        for want_s2 in (False, True):
            out = None
            for _ in range(20):
                out = eng.infer(obs, want_s2=want_s2, out=out)
            torch.cuda.synchronize()
            # back-to-back calls: wall clock = max(host cost of a call, device time of a call)
            t0 = time.perf_counter()
            n = 300
            for _ in range(n):
                out = eng.infer(obs, want_s2=want_s2, out=out)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e6
            # one call at a time: issue -> result on the device (what the control loop waits for)
            lat = []
            for _ in range(50):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                out = eng.infer(obs, want_s2=want_s2, out=out)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t2) * 1e6)
            lat.sort()
            # the same forward as one replayable HIP graph (static input/output tensors)
            gi = eng.graphed_infer(rows, want_s2=want_s2, noise=False)
            for _ in range(20):
                gi(obs)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(n):
                gi(obs)
            torch.cuda.synchronize()
            gwall = (time.perf_counter() - t3) / n * 1e6
            glat = []
            for _ in range(50):
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                gi(obs)
                torch.cuda.synchronize()
                glat.append((time.perf_counter() - t4) * 1e6)
            glat.sort()
            print("%-28s rows %2d  %s : %6.1f us / call back to back (host %5.1f), %6.1f us single-call latency | "
                  "as a HIP graph: %6.1f us back to back, %6.1f us single call"
                  % (name, rows, "TE+MD+WM" if want_s2 else "TE+MD   ", wall, (t1 - t0) / n * 1e6, lat[len(lat) // 2],
                     gwall, glat[len(glat) // 2]))
    # host observation -> host action, what the control loop actually waits for (the env runs on the CPU): the plain way
    # (copy the observation up, forward, copy the action down) against infer_host (the kernels read the observation from
    # and write the action to pinned host memory themselves; no copy launches, no stream synchronisation)
    obs_h = torch.randn(1, 394)
    for label, fn in (("obs.cuda() -> infer -> a_hat.cpu()", lambda: eng.infer(obs_h.to("cuda", non_blocking=False), want_s2=False)[0].cpu()),
                      ("infer_host (pinned in / out)      ", lambda: eng.infer_host(obs_h))):
        for _ in range(30):
            fn()
        lat = []
        for _ in range(200):
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            r = fn()
            lat.append((time.perf_counter() - t6) * 1e6)
        lat.sort()
        print("%-28s rows  1  host obs -> host action, %s : %6.1f us median, %6.1f us p90"
              % (name, label, lat[len(lat) // 2], lat[int(len(lat) * 0.9)]))
    a_dev = eng.infer(obs_h.cuda(), noise=False, want_s2=False)[0].cpu()
    a_host = eng.infer_host(obs_h, noise=False)
    assert torch.equal(a_dev, a_host), "infer_host disagrees with infer"
    # the call-persistent rollout server (include/pvae.h pvae_rollout_server_*): a kernel resident on one XCD with the
    # encoder's and decoder's weights in LDS answers from a mailbox in pinned host memory -- no launch per call
    for scope in (("auto",) if QUICK else ("auto", "chip")):
        try:
            eng.rollout_server_start(idle_ms=200.0, lifetime_s=60.0, scope=scope)
        except RuntimeError as exc:
            print("%-28s rows  1  rollout server: %s" % (name, str(exc).split(":")[-1].strip()[:160]))
            continue
        if scope == "chip" and eng.rollout_server_scope() != "chip":
            eng.rollout_server_stop()
            continue
        try:
            o = obs_h.numpy()[0]
            where = "%s, LDS %d KB per workgroup, request block in %s memory" % (
                {"xcd": "32 workgroups on one XCD", "chip": "256 workgroups over the chip"}[eng.rollout_server_scope()],
                eng.rollout_server_status()[2] // 1024, eng.rollout_server_mailbox())
            for i in range(50):
                eng.rollout_server_infer(o, noise=True, seed=0, offset=i)
            lat = []
            for i in range(500):
                t7 = time.perf_counter()
                eng.rollout_server_infer(o, noise=True, seed=0, offset=i)
                lat.append((time.perf_counter() - t7) * 1e6)
            lat.sort()
            a_srv = eng.rollout_server_infer(o, noise=False)[0].copy()
            us = sorted(eng.rollout_server_selfbench(o, n=2000))
            print("%-28s rows  1  host obs -> host action, rollout server (%s)" % (name, where))
            print("%-28s          timed inside the library call (a compiled host's view) : %6.1f us median, %6.1f us p90, %6.1f us min"
                  % (name, us[len(us) // 2], us[int(len(us) * 0.9)], us[0]))
            print("%-28s          through Python (ctypes)                               : %6.1f us median, %6.1f us p90, %6.1f us min"
                  % (name, lat[len(lat) // 2], lat[int(len(lat) * 0.9)], lat[0]))
            tl, mhz = eng.rollout_server_timeline()
            print("%-28s          on the device, us after workgroup 0 saw the request: %s | reply issued %.1f | shader clock %.0f MHz"
                  % (name, "  ".join("L%d in %.1f out %.1f" % (l, tl[1 + 2 * l], tl[2 + 2 * l]) for l in range((len(tl) - 2) // 2)), tl[-1], mhz))
            # a 30 Hz control loop: 33 ms of host work between two calls (the kernel stays resident: idle time-out 200 ms)
            lat = []
            for i in range(20):
                time.sleep(0.033)
                t7 = time.perf_counter()
                eng.rollout_server_infer(o, noise=True, seed=0, offset=i)
                lat.append((time.perf_counter() - t7) * 1e6)
            lat.sort()
            print("%-28s          at 30 Hz (33 ms between calls), through Python         : %6.1f us median" % (name, lat[len(lat) // 2]))
        finally:
            eng.rollout_server_stop()
        assert (a_srv == a_dev.numpy()[0]).all(), "rollout server disagrees with infer"
    # the module surface RLlib drives (rmt:742-771 + value_function): forward alone, and forward + the lazily
    # evaluated value branch (plain torch Linear layers: it takes no part in the supervised path)
    m = tr.model
    m.latent_prior_noise = False
    obs1 = torch.randn(1, 394, device="cuda")
    for with_value in (False, True):
        lat = []
        with torch.no_grad():                      # as RLlib's sampler calls it
            for _ in range(20):
                m.forward({"obs_flat": obs1}, [], None)
                if with_value:
                    m.value_function()
            for _ in range(100):
                torch.cuda.synchronize()
                t5 = time.perf_counter()
                logits, _ = m.forward({"obs_flat": obs1}, [], None)
                if with_value:
                    v = m.value_function()
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t5) * 1e6)
        lat.sort()
        print("%-28s rows  1  PhysicsVAE.forward%s : %6.1f us single-call latency"
              % (name, " + value_function()" if with_value else "                   ", lat[len(lat) // 2]))
    # the same module call served by the resident kernel: a one-row CPU observation in, CPU logits out
    try:
        m.start_rollout_server(idle_ms=200.0, lifetime_s=60.0)
    except RuntimeError:
        pass
    else:
        try:
            obs_c = obs1.cpu()
            lat = []
            with torch.no_grad():
                for _ in range(50):
                    m.forward({"obs_flat": obs_c}, [], None)
                for _ in range(300):
                    t5 = time.perf_counter()
                    logits, _ = m.forward({"obs_flat": obs_c}, [], None)
                    lat.append((time.perf_counter() - t5) * 1e6)
            lat.sort()
            print("%-28s rows  1  PhysicsVAE.forward, CPU observation, served by the resident kernel : %6.1f us median"
                  % (name, lat[len(lat) // 2]))
        finally:
            m.stop_rollout_server()
