"""Is the step host-launch-bound?  Replay one captured step as a HIP graph vs eager launches."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from physicsvae_amd import _lib
from synth_demo import make_trainer, synth_demo
data = synth_demo(0, 10, 1000, 197, 45)
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(data, 256, "cuda")
eng = tr.engine
eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
loss = torch.zeros(5, device="cuda")
for name in ("world", "joint"):
    w = name == "world"
    tr.model.set_learnable_task_encoder(not w); tr.model.set_learnable_motor_decoder(not w); tr.model.set_learnable_world_model(w)
    tr.read_loss_fn_coeff(world=w)
    phase, nets = tr.phase()
    sp = tr.step_params(nets, 256, True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20):
            eng.train_step(phase, 0, 256, sp, loss_out=loss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            eng.train_step(phase, 0, 256, sp, loss_out=loss)
        t1 = time.perf_counter()           # host enqueue time only
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s eager: enqueue %.1f us/step, total %.1f us/step" % (name, (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.train_step(phase, 0, 256, sp, loss_out=loss)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            g.replay()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s graph: total %.1f us/step" % (name, (t2 - t0) / 300 * 1e6))
