"""First layers that read the demonstration set where it lies (tpv:365-377, rmt:822-844: the torch.cat sites as
address arithmetic in the kernels' loaders) against the staging launch that materialises the five input / target panels:
the SAME bits -- losses, every parameter, both Adam moments -- over several optimizer steps of both phases, for every
kernel family a gathered first layer can run on:

  32x32 tiles + Pro patch   dim_body 197 (rows 4-byte aligned, a chunk straddles the [s_t | .] boundary), 256 / 250 rows
  64x32 tiles, chunk select  dim_body 400 at 512 rows      64x64 tiles  at 1024 rows
  the fallback               dim_body 197 at 512 rows (no Pro patch on 64-row tiles): the step stages, and says so

plus the data-parallel step, windows that cross episode boundaries and the weight pad columns staying exactly zero."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(Db, Da, width, depth, rows, n_ep=4, T=700):
    arch = R.make_arch(Db, Da, latent=32, te=(width, depth), md=(width, depth), wm=(width, depth))
    data = R.synth_demo(0, 2, 8, Db, Da)                       # (placeholder file: the set below is generated on the device)
    tr = make_trainer(arch, data, rows, device=DEV)
    eng = tr.engine
    gen = torch.Generator(device=DEV).manual_seed(5)
    R_ = n_ep * T

    def roomy(n, w):
        buf = torch.zeros(n * w + 16, device=DEV)
        v = buf[: n * w].view(n, w)
        v.copy_(torch.randn(n, w, generator=gen, device=DEV))
        return v
    states, actions = roomy(R_, Db), roomy(R_, Da).clamp_(-3, 3)
    idx = (torch.arange(n_ep, device=DEV)[:, None] * T + torch.arange(T - 1, device=DEV)[None, :]).reshape(-1).to(torch.int32)
    eng.bind_dataset(states, actions, idx)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, 1), 3))
    return arch, tr, eng, (states, actions, idx)


def _steps(tr, eng, phase_world, rows, direct, K=3, first0=0, dp=False):
    eng.set_direct(direct)
    tr.model.set_learnable_task_encoder(not phase_world)
    tr.model.set_learnable_motor_decoder(not phase_world)
    tr.model.set_learnable_world_model(phase_world)
    phase = _lib.PHASE_WORLD if phase_world else _lib.PHASE_JOINT
    c = R.phase_coeffs(phase_world)
    eng.exp_avg.zero_(); eng.exp_avg_sq.zero_(); eng.invalidate_staging()
    g = torch.Generator().manual_seed(9)
    eps = torch.randn(K, 1, rows, eng.arch.Z, generator=g).to(DEV)
    out = torch.zeros(K, 5, device=DEV)
    active = []
    for i in range(K):
        sp = make_step_params(lr=5e-4, adam_t=(i + 1, i + 1, i + 1), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                              s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=rows)
        active.append(eng.direct_active(phase, rows, sp, fused=not dp))
        first = first0 + i * rows
        if dp:
            eng.dp_train_step(phase, first, rows, sp, eps=eps[i].contiguous(), loss_out=out[i], next_span=(first + rows, rows))
        else:
            eng.train_step(phase, first, rows, sp, eps=eps[i].contiguous(), loss_out=out[i], next_span=(first + rows, rows))
    torch.cuda.synchronize()
    return (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()), active


def _both(tr, eng, world, rows, **kw):
    start = eng.params.clone()
    got = {}
    for direct in (False, True):
        eng.params.copy_(start)
        got[direct], act = _steps(tr, eng, world, rows, direct, **kw)
        assert all(a == direct for a in act) or not direct, act
        got[(direct, "active")] = act
    eng.params.copy_(start)
    eng.set_direct(True)
    return got


@pytest.mark.parametrize("rows", [256, 250])
@pytest.mark.parametrize("world", [True, False])
def test_direct_first_layers_equal_the_staged_panels_bit_for_bit_on_32x32_tiles(world, rows):
    """dim_body 197 / dim_action 45 (the BASELINE dims: unaligned rows, the [s_t | z] and [s_t | a] boundaries inside a
    16-byte chunk), 2x512 stacks, 256 rows and a ragged 250: the wave-specialised 32x32 kernel with the sampler prologue
    (decoder) / the column patch (world model), the gathered weight gradients of layer 0, both targets read in place."""
    arch, tr, eng, _ = _trainer(197, 45, 512, 2, rows)
    got = _both(tr, eng, world, rows, first0=650)               # (the three minibatches cross an episode boundary at window 699)
    assert all(got[(True, "active")]), "the step did not take the direct path"
    for a, b in zip(got[False], got[True]):
        assert torch.equal(a, b)
    assert float(got[True][3][0, 0]) != float(got[True][3][1, 0])
    # pad columns of every first-layer weight block are still exactly zero (a gathered operand must not leak into them)
    for info in eng.layers:
        if info["index"] == 0:
            blk = got[True][0][info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]].view(info["n_out_pad"], info["ld"])
            for pad in (blk[:, info["n_in"]:], blk[info["n_out"]:]):
                assert pad.numel() == 0 or float(pad.abs().max()) == 0.0


@pytest.mark.parametrize("rows", [512, 1024])
@pytest.mark.parametrize("world", [True, False])
def test_direct_first_layers_on_64_row_tiles(world, rows):
    """dim_body 400 / dim_action 90 (BASELINE configs[4]'s dims: 16-byte-aligned rows, no straddling chunk), 1x1024 stacks:
    64x32 tiles at 512 rows, 64x64 at 1024 -- the second column block ([. | z], [. | a], [. | a_hat]) is chunk-selected by
    the loaders, no patch."""
    arch, tr, eng, _ = _trainer(400, 90, 1024, 1, rows, n_ep=6)
    got = _both(tr, eng, world, rows, first0=100)
    assert all(got[(True, "active")])
    for a, b in zip(got[False], got[True]):
        assert torch.equal(a, b)


def test_steps_that_do_not_qualify_stage_as_before():
    """dim_body 197 at 512 rows: the 64-row-tile kernels have no Pro patch for the straddling chunk -> the step stages its
    panels (and still trains).  (A dataset allocation without 16 readable bytes behind its last row does the same:
    pvae_bind_dataset asks hipMemGetAddressRange; WindowDataset.device_arrays allocates with that room.)"""
    arch, tr, eng, (states, actions, idx) = _trainer(197, 45, 1024, 1, 512)
    res, active = _steps(tr, eng, True, 512, True)
    assert not any(active) and torch.isfinite(res[3]).all()


@pytest.mark.parametrize("world", [True, False])
def test_data_parallel_step_takes_the_direct_path_too(world):
    """pvae_dp_train_step (gradient store + exchange + Adam; here one rank through the peer-mapped launch): direct == staged."""
    arch, tr, eng, _ = _trainer(197, 45, 512, 2, 256)
    eng.p2p_open(0, 1, [eng.p2p_export()])
    eng.comm_mode("p2p")
    got = _both(tr, eng, world, 256, dp=True)
    assert all(got[(True, "active")])
    for a, b in zip(got[False], got[True]):
        assert torch.equal(a, b)
    eng.p2p_close()
