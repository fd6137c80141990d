"""`shuffle_data: True` on the HIP path (tm:166-175, 181: the reference's DataLoader shuffles when the key is spelt
right; upstream's own config says "suffle_data", tpv:260).  A shuffled pass is a permuted window -> row table bound for
the pass; everything behind it -- gather, prefetch, the step -- is the sequential path's."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from util import make_trainer, max_err_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_shuffled_training_matches_the_reference_capture(golden):
    """The reference's trainer with shuffle_data under torch.manual_seed(seed) (tests/golden/shuffle_tiny.npz): the HIP
    trainer under the same seed binds, pass by pass, the table of the order the reference's dataset was asked in, and
    reproduces its epoch losses across the phase switch (1e-3) and its final weights."""
    g = golden("shuffle_tiny")
    n_ep, n_steps, batch, m_world, n_epochs, seed = [int(v) for v in g["meta"][9:15]]
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    data = R.synth_demo(0, n_ep, n_steps, 7, 3, kind="dynamics")
    tr = make_trainer(arch, data, batch, m_world=m_world, device=DEV, eps_fn=R.eps_stream(2, 4), lr_step=2,
                      extra={"shuffle_data": True})
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    base = tr.train_loader.dataset.device_arrays(DEV)[2].clone()
    torch.manual_seed(seed)
    losses = []
    for e in range(n_epochs):
        losses.append(tr.train()["mean_train_loss"])
        order = torch.from_numpy(g["order_epoch%d" % e]).to(DEV)
        assert torch.equal(tr.engine.dataset[2], base[order]), "epoch %d ran in another order than the reference's" % e
    np.testing.assert_allclose(losses, g["epoch_losses"], rtol=1e-3)
    for k, v in tr.model.state_dict().items():
        if ("final::" + k) in g.files and v.numel():
            assert max_err_scaled(v.cpu(), g["final::" + k]) < 5e-3, k
    assert torch.equal(tr.train_loader.dataset.device_arrays(DEV)[2], base)           # the dataset's own table is untouched


def test_shuffled_epochs_with_prefetch_equal_explicit_gathers():
    """256-row minibatches at the BASELINE dims, shuffled: the step whose gather rode in the previous step's last launch
    (prefetch across minibatches; the batch prefetched across the EPOCH boundary came from the old table and is dropped)
    equals a trainer that gathers every minibatch explicitly, bit for bit, over three epochs."""
    arch = R.make_arch(197, 45, latent=32, te=(256, 2), md=(256, 2), wm=(256, 2))
    data = R.synth_demo(0, 3, 300, 197, 45, kind="dynamics")
    sds = []
    for prefetch in (True, False):
        tr = make_trainer(arch, data, 256, m_world=1, device=DEV, extra={"shuffle_data": True})
        tr.prefetch_gather = prefetch
        tr.model.load_state_dict(R.init_state_dict(arch, seed=1))
        torch.manual_seed(7)
        losses = [tr.train()["mean_train_loss"] for _ in range(3)]
        sds.append(({k: v.clone() for k, v in tr.model.state_dict().items()}, losses))
    assert sds[0][1] == sds[1][1]
    for k in sds[0][0]:
        assert torch.equal(sds[0][0][k], sds[1][0][k]), k


@pytest.mark.parametrize("lookahead", [1, 2])
def test_shuffled_runs_track_the_oracle_trainer(lookahead):
    """Shuffled passes against the oracle's trainer (whose loader IS torch's shuffled DataLoader) under the same
    torch.manual_seed, at wider dims, with lookahead 1 and 2 (the permuted table holds the windows' first rows; an unrolled
    sample reads rows + t): per-epoch losses across the phase switch within 1e-3."""
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
    data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")
    X, Y = R.build_windows(data, lookahead=lookahead)
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    ref = R.RefTrainer(arch, sd, X, Y, 16, 2, lr_step=2, eps_fn=R.eps_stream(2, 8), shuffle=True)
    torch.manual_seed(99)
    want = [ref.step()["mean_train_loss"] for _ in range(4)]
    tr = make_trainer(arch, data, 16, m_world=2, device=DEV, eps_fn=R.eps_stream(2, 8), lr_step=2,
                      extra={"shuffle_data": True, "lookahead": lookahead})
    tr.model.load_state_dict(sd)
    torch.manual_seed(99)
    got = [tr.train()["mean_train_loss"] for _ in range(4)]
    np.testing.assert_allclose(got, want, rtol=1e-3)
