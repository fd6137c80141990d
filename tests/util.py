"""Shared helpers for the tests (oracle = checker only)."""
import os
import tempfile

import torch

from oracle import refpath as R
from physicsvae_amd import train_physics_vae as T
from physicsvae_amd.tune import grid_search as tune_grid


def arch_from_meta(meta):
    """`meta` of a fixture, or the fixture itself (then an edited "act_fn" recorded in it is honoured)."""
    act, stacks = "relu", None
    if hasattr(meta, "files"):
        act = str(meta["act_fn"]) if "act_fn" in meta.files else "relu"
        stacks = meta["stacks"] if "stacks" in meta.files else None          # stacks recorded layer by layer
        meta = meta["meta"]
    Db, Da, Z, tw, td, mw, md_, ww, wd = [int(v) for v in meta[:9]]
    arch = R.make_arch(Db, Da, latent=Z, te=(tw, td), md=(mw, md_), wm=(ww, wd), act=act)
    return R.with_stacks(arch, stacks) if stacks is not None else arch


def general(spec):
    return len(spec) > 0 and isinstance(spec[0], (tuple, list))


def first_width(spec):
    return spec[0][0] if general(spec) else spec[0]


def depth_of(spec):
    return len(spec) if general(spec) else spec[1]


def make_trainer(arch, data, batch, m_world=2, device=None, eps_fn=None, lr_step=50, extra=None):
    """physicsvae_amd TrainModel over a synthetic demo dict (written to a temp pickle)."""
    td = tempfile.mkdtemp(prefix="pvae_test_")
    pkl = os.path.join(td, "demo.pkl")
    R.write_demo(pkl, data)
    argv = ["--data_train", pkl, "--batch_size", str(batch), "--max_iter_world_model", str(m_world),
            "--max_iter", str(max(m_world, 100)),
            "--latent_dim", str(arch["Z"]),
            "--TE_width", str(first_width(arch["te"])), "--TE_depth", str(depth_of(arch["te"])),
            "--MD_width", str(first_width(arch["md"])), "--MD_depth", str(depth_of(arch["md"])),
            "--world_model_width", str(first_width(arch["wm"])), "--world_model_depth", str(depth_of(arch["wm"]))]
    if arch.get("prior", R.PRIORS[0]) not in (R.PRIORS[0], False):
        argv += ["--prior", arch["prior"]]
    T.args = T.arg_parser().parse_args(argv)
    cfg = T.get_trainer_config(T.args)
    if arch.get("prior", R.PRIORS[0]) is False:           # (not reachable from the CLI, upstream neither: a dict edit)
        cfg["latent_prior_type"] = tune_grid([False])
    cfg["act_fn"] = arch.get("act", "relu")               # tpv:262, a dict edit as well
    for key, prefix in (("te", "TE"), ("md", "MD"), ("wm", "world_model")):     # stacks given layer by layer
        if general(arch[key]):
            cfg[prefix + "_layers"] = R.fc_layer_list(arch[key], cfg["act_fn"])
    if arch.get("prior") == R.PRIORS[1] and tuple(arch["pr"]) != tuple(arch["te"]):
        cfg["model"]["custom_model_config"]["latent_prior_layers"] = T.gen_layers(arch["pr"][0], arch["pr"][1])
    cfg["lr_schedule_params"] = {"step_size": lr_step, "gamma": 0.7}
    for key, name in (("te_inputs", "task_encoder_inputs"), ("md_inputs", "motor_decoder_inputs")):   # rmt:470, 485: dict edits
        if key in arch:
            cfg["model"]["custom_model_config"][name] = list(arch[key])
    if device is not None:
        cfg["model"]["custom_model_config"]["device"] = device
    if eps_fn is not None:
        cfg["eps_fn"] = eps_fn
    cfg.update(extra or {})
    tr = T.TrainModel(cfg)
    tr._tmpdir = td
    return tr


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err_scaled(a, b):
    """max |a-b| / (max|b| + tiny): elementwise check robust to near-zero entries."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
