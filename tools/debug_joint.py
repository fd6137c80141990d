import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer, max_err_scaled, rel_err
import contextlib, io
arch = R.make_arch(197,45,latent=32,te=(1024,4),md=(1024,4),wm=(1024,4))
data = R.synth_demo(0, 2, 300, 197, 45)
X,Y = R.build_windows(data)
x,y = next(iter(R.make_loader(X,Y,256)))
sd = R.perturb_biases(R.init_state_dict(arch,1),3)
eps = R.eps_stream(2,32)(0,(256,32))
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(arch, data, 256, device="cuda")
tr.model.load_state_dict(sd)
eng = tr.engine
# oracle with hooks to capture dz per layer
model = R.RefModel(arch); model.load_state_dict(sd); model.eps_source = lambda s: eps
model.set_learnable("_world_model", False)
pre = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); pre[name] = out
    return f
for net in ("_task_encoder","_motor_decoder","_world_model"):
    for i, slim in enumerate(getattr(model, net)._model):
        slim._model[0].register_forward_hook(hook((net,i)))
tot, terms = R.compute_loss(model, x, y, R.phase_coeffs(False))
tot.backward()
want = {k:p.grad for k,p in model.named_parameters() if p.grad is not None}
sp = make_step_params(lr=5e-4, global_rows=256)
eng.set_batch(x,y)
eng.grads.fill_(float("nan"))
loss = eng.forward_backward(_lib.PHASE_JOINT, 256, sp, eps=eps, fused_adam=False).cpu()
print("loss", loss.tolist(), float(tot))
gv = eng.named_views(eng.grads)
for k,g in want.items():
    print("%-45s max %.2e rel %.2e" % (k, max_err_scaled(gv[k].cpu(), g), rel_err(gv[k].cpu(), g)))
names = {"_task_encoder":0,"_motor_decoder":1,"_world_model":2}
for (net,i),t in pre.items():
    if t.grad is None: continue
    ours = eng.panel("dz", names[net], i)[:256, :t.shape[1]].cpu()
    err = (ours - t.grad).abs()
    rows_bad = (err.max(dim=1).values > 1e-4 * t.grad.abs().max()).nonzero().flatten().tolist()
    print("dz %-16s L%d max %.2e rel %.2e bad rows %s" % (net, i, max_err_scaled(ours, t.grad), rel_err(ours, t.grad), rows_bad[:20]))
    a_ours = eng.panel("act", names[net], i)[:256, :t.shape[1]].cpu()
    a_ref = torch.relu(t.detach()) if i < 4 else t.detach()
    print("   act max %.2e" % max_err_scaled(a_ours, a_ref))
