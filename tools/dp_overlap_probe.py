"""One-rank cost of the exchange machinery of pvae_dp_train_step at the bench sizes."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import make_trainer as mk, synth_demo
from physicsvae_amd.engine import make_step_params
data = synth_demo(0, 4, 400, 197, 45)
with contextlib.redirect_stdout(io.StringIO()):
    tr = mk(data, 256, "cuda")
eng = tr.engine
eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
eng.comm_init(0, 1, eng.comm_unique_id())
out = torch.zeros(5, device="cuda")
if os.environ.get("PROBE_SIDE_STREAM"):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
for name in ("world", "joint"):
    w = name == "world"
    tr.model.set_learnable_task_encoder(not w); tr.model.set_learnable_motor_decoder(not w)
    tr.model.set_learnable_world_model(w)
    tr.read_loss_fn_coeff(world=w)
    phase, nets = tr.phase()
    for mb in (None, 0.0, 100.0, 6.0):
        if mb is not None:
            eng.comm_config(mb, 0)
        n = 200
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                sp = make_step_params(lr=1e-3, adam_t=(i + 1, i + 1, i + 1), a_rec=tr.a_rec_coeff, kl=tr.vae_kl_coeff,
                                      s_rec=tr.s_rec_coeff, cyc=tr.vae_cycle_coeff, global_rows=256, seed=7, offset=i * 65536)
                first = 256 * (i % 5)
                if mb is None:
                    eng.train_step(phase, first, 256, sp, loss_out=out, next_span=(256 * ((i + 1) % 5), 256))
                else:
                    eng.dp_train_step(phase, first, 256, sp, loss_out=out, next_span=(256 * ((i + 1) % 5), 256))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("%s bucket %-6s: %.1f us/step (host enqueue %.1f)" % (name, "fused" if mb is None else mb, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6), flush=True)
eng.comm_destroy()
