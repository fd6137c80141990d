"""Soak of the resident rollout server: N requests of mixed kinds (1-4 rows, decoder-only, weight reloads, with and without
noise) while training steps run on the compute stream of the same process; every answer is compared bit for bit with the
launch path's answer to the same request, computed beforehand.  Prints mismatches / time-outs (expected: none).
    python tools/server_soak.py [n_requests]"""
import os
import sys
import time

import numpy as np
import torch

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
from oracle import refpath as R          # noqa: E402  (weights / demo generator: a tool, not the product)
from physicsvae_amd import _lib          # noqa: E402
from physicsvae_amd.engine import make_step_params   # noqa: E402
from util import make_trainer            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
arch = R.make_arch(197, 45)
data = R.synth_demo(0, 4, 300, 197, 45, kind="dynamics")
tr = make_trainer(arch, data, 256, device="cuda")
sds = [R.perturb_biases(R.init_state_dict(arch, seed=s), seed=3) for s in (1, 2)]
eng = tr.engine
X, _ = R.build_windows(data)
obs = np.asarray(X)[:64, 0, :].astype(np.float32)
Db, Z = 197, 32
# expected answers per (weights, kind, first row, rows, noise): from the launch path
want = {}
for w, sd in enumerate(sds):
    tr.model.load_state_dict(sd)
    for rows in (1, 2, 3, 4):
        for i in range(0, 16):
            for noise in (0, 1):
                a, _, z = eng.infer(torch.from_numpy(obs[i: i + rows]).cuda(), noise=bool(noise), seed=5, offset=100 + i, want_s2=False)
                want[(w, rows, i, noise)] = (a.cpu().numpy().copy(), z.cpu().numpy().copy())
tr.model.load_state_dict(sds[0])
cur = 0
rng = np.random.default_rng(0)
eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
scratch = torch.zeros(5, device="cuda")
# a second engine trains meanwhile (its launches share the GPU with the resident kernel)
tr2 = make_trainer(R.make_arch(197, 45, te=(512, 2), md=(512, 2), wm=(512, 2)), data, 256, device="cuda")
e2 = tr2.engine
e2.bind_dataset(*tr2.train_loader.dataset.device_arrays(e2.device))
bad = timeouts = 0
kinds = {"rows1": 0, "rows2-4": 0, "decode": 0, "reload": 0}
eng.rollout_server_start(idle_ms=50.0, lifetime_s=600.0)
t0 = time.time()
try:
    for n in range(N):
        if n % 64 == 0:                              # keep the compute stream busy
            sp = make_step_params(lr=1e-4, adam_t=(n // 64 + 1,) * 5, a_rec=1.0, kl=1.0, s_rec=0.0, cyc=1e-3, global_rows=256)
            e2.train_step(_lib.PHASE_JOINT, (n // 64 * 256) % 512, 256, sp, loss_out=scratch)
        if n % 5000 == 4999:                         # new weights through the module API: the next request reloads
            cur ^= 1
            tr.model.load_state_dict(sds[cur])
            kinds["reload"] += 1
        if n % 20011 == 20010:
            time.sleep(0.08)                         # longer than the idle time-out: the kernel leaves and comes back
        i, noise = int(rng.integers(0, 16)), int(rng.integers(0, 2))
        r = rng.random()
        try:
            if r < 0.6:
                a, _, z = eng.rollout_server_infer(obs[i], noise=bool(noise), seed=5, offset=100 + i)
                ok = np.array_equal(a, want[(cur, 1, i, noise)][0][0]) and np.array_equal(z, want[(cur, 1, i, noise)][1][0])
                kinds["rows1"] += 1
            elif r < 0.9:
                rows = int(rng.integers(2, 5))
                a, _, z = eng.rollout_server_infer_rows(obs[i: i + rows], noise=bool(noise), seed=5, offset=100 + i)
                ok = np.array_equal(a, want[(cur, rows, i, noise)][0]) and np.array_equal(z, want[(cur, rows, i, noise)][1])
                kinds["rows2-4"] += 1
            else:
                zz = want[(cur, 1, i, noise)][1][0]
                a = eng.rollout_server_decode(np.concatenate([obs[i, :Db], zz]))
                ok = np.array_equal(a, want[(cur, 1, i, noise)][0][0])
                kinds["decode"] += 1
        except RuntimeError as exc:
            timeouts += 1
            ok = True
            print("request %d: %s" % (n, str(exc)[:120]), flush=True)
        bad += 0 if ok else 1
finally:
    eng.rollout_server_stop()
torch.cuda.synchronize()
print("%d requests in %.1f s (%s): %d answers differ from the launch path, %d time-outs / errors"
      % (N, time.time() - t0, ", ".join("%s %d" % kv for kv in kinds.items()), bad, timeouts))
