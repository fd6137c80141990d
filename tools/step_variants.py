"""Step-time decomposition: fused Adam vs gradient store (no Adam) vs store + flat Adam vs staged DP path."""
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
def timeit(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name in ("world", "joint"):
    w = name == "world"
    tr.model.set_learnable_task_encoder(not w); tr.model.set_learnable_motor_decoder(not w); tr.model.set_learnable_world_model(w)
    tr.read_loss_fn_coeff(world=w)
    phase, nets = tr.phase()
    sp = tr.step_params(nets, 256, True)
    def fused(): eng.train_step(phase, 0, 256, sp, loss_out=loss)
    def store():
        eng.gather(0, 256); eng.forward_backward(phase, 256, sp, fused_adam=False, loss_out=loss)
    def store_adam():
        store(); eng.adam(nets, sp)
    def fwd_only():
        eng.gather(0, 256); eng.forward_backward(phase, 256, sp, backward=False, loss_out=loss)
    def staged():
        eng.gather(0, 256); eng.forward_seed(phase, 256, sp)
        k, n, segs = 0, 1, []
        while k < n:
            seg, net, n = eng.backward_stage(phase, 256, sp, k, loss_out=loss); k += 1
            if seg: segs.append((net,) + seg)
        for net, off, cnt in segs: eng.adam_segment(net, off, cnt, sp)
    print("%s: fused %.1f | store (no Adam) %.1f | store + flat Adam %.1f | staged DP path (no comm) %.1f | forward+loss only %.1f  us/step" % (
        name, timeit(fused), timeit(store), timeit(store_adam), timeit(staged), timeit(fwd_only)))
