"""Host-side cost of one optimizer step (ctypes call + plan + launches) vs the GPU step time:
short bursts measure the pure host cost, a long run the back-pressured steady state.
Usage (GPU box): python tools/host_rate.py"""
import sys, os, time, io, contextlib, torch
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tools")]
from synth_demo import make_trainer, synth_demo
data = synth_demo(0,10,1000,197,45)
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(data, 256, "cuda")
eng=tr.engine; ds=tr.train_loader.dataset
eng.bind_dataset(*ds.device_arrays(eng.device))
loss=torch.zeros(5,device="cuda")
for name in ("world","joint"):
    w = name=="world"
    tr.model.set_learnable_task_encoder(not w); tr.model.set_learnable_motor_decoder(not w); tr.model.set_learnable_world_model(w); tr.read_loss_fn_coeff(world=w)
    phase,nets=tr.phase()
    def run(n):
        for i in range(n):
            g=i%38
            sp=tr.step_params(nets,256,True)
            eng.train_step(phase,g*256,256,sp,loss_out=loss,next_span=((g+1)%38*256,256))
    run(50); torch.cuda.synchronize()
    best = 1e9
    for rep in range(10):                      # short bursts: the launch queue never fills, so this is pure host cost
        t0 = time.perf_counter(); run(16); t1 = time.perf_counter(); torch.cuda.synchronize()
        best = min(best, (t1 - t0) / 16 * 1e6)
    t0=time.perf_counter(); run(2000); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(name, "host cost us/step %.1f (16-step bursts), steady-state enqueue %.1f, wall us/step %.1f"%(best,(t1-t0)/2000*1e6,(t2-t0)/2000*1e6))
