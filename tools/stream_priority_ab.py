"""Does the stream a step is issued on change its period?  The joint step of bench.py (--inner) on torch's default
stream (the legacy null stream), on a plain side stream and on a high-priority side stream."""
import io, contextlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from synth_demo import make_trainer, synth_demo
data = synth_demo(0, 10, 1000, 197, 45)
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(data, 256, "cuda")
eng = tr.engine
eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
loss = torch.zeros(5, device="cuda")
tr.model.set_learnable_task_encoder(True); tr.model.set_learnable_motor_decoder(True); tr.model.set_learnable_world_model(False)
tr.read_loss_fn_coeff(world=False)
phase, nets = tr.phase()
sp = tr.step_params(nets, 256, True)
def run(n):
    for i in range(n):
        eng.train_step(phase, (i * 256) % 9000, 256, sp, loss_out=loss, next_span=(((i + 1) * 256) % 9000, 256))
for name, st in (("default stream", None), ("side stream", torch.cuda.Stream()), ("high-priority side stream", torch.cuda.Stream(priority=-1)),
                 ("default stream", None)):
    ctx = torch.cuda.stream(st) if st is not None else contextlib.nullcontext()
    with ctx:
        run(60); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(400); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
    print("%-28s %.2f us per joint step" % (name, best))
