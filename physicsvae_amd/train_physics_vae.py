"""The reference's train_physics_vae.py surface on the MI355X-native hot path.

Same CLI flags (tpv:30-55), same dataset schema (tpv:57-92), same trainer-config keys
(tpv:247-286), same two-phase schedule and loss (tpv:313-435), same five checkpoint files
(tpv:440-467) -- driven by HIP kernels instead of a PyTorch autograd graph, and without a
ray / gym dependency.  Extra flags (ours): --TE_width/--TE_depth/--MD_*/--world_model_* to
pick the MLP sizes from the command line, --seed.

    python -m physicsvae_amd.train_physics_vae --data_train demo.pkl \\
        --max_iter 800 --max_iter_world_model 300
"""
import argparse
import copy
import os
import pickle

import numpy as np
import torch

from . import torch_models, tune
from .engine import make_step_params
from .model import PhysicsVAE, fc_spec
from .spaces import Box

args = None          # module-global, as in the reference (read by TrainModel.load_dataset, tpv:339)


def arg_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--max_iter_world_model", type=int, default=0)
    p.add_argument("--max_iter", type=int, default=100)
    p.add_argument("--num_cpus", type=int, default=1)
    p.add_argument("--num_gpus", type=int, default=0)
    p.add_argument("--data_train", action="append", required=True, type=str, default=None)
    p.add_argument("--data_test", action="append", type=str, default=None)
    p.add_argument("--num_data", type=int, default=None)
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--lr", type=float, default=0.0005)
    p.add_argument("--lr_schedule", type=str, default="step")
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--checkpoint_freq", type=int, default=100)
    p.add_argument("--checkpoint", type=str, default=None)
    p.add_argument("--cluster", action="store_true")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--name", type=str, default=None)
    p.add_argument("--local_dir", type=str, default="~/ray_results")
    p.add_argument("--world_model", type=str, default=None)
    p.add_argument("--latent_dim", type=int, default=32)
    # NB (reference quirk kept): action='append' on a list default
    # APPENDS to the default, so `--vae_kl_coeff 0.1` means a sweep over [1.0, 0.1]: two trials,
    # which tune.run executes one after the other (tune.expand_grid).
    p.add_argument("--vae_kl_coeff", type=float, action="append", default=[1.0])
    p.add_argument("--vae_cycle_coeff", type=float, action="append", default=[1e-3])
    p.add_argument("--latent_prior_type", type=str, action="append",
                   default=["normal_zero_mean_one_std"])
    # ours: the same three settings as SINGLE values (one trial, no sweep over the default)
    p.add_argument("--kl_coeff", type=float, default=None, help="single vae_kl_coeff (replaces the sweep list)")
    p.add_argument("--cycle_coeff", type=float, default=None, help="single vae_cycle_coeff")
    p.add_argument("--prior", type=str, default=None, help="single latent_prior_type")
    # ours: MLP sizes are dict-only in the reference (tpv:263-280)
    p.add_argument("--TE_width", type=int, default=256)
    p.add_argument("--TE_depth", type=int, default=2)
    p.add_argument("--MD_width", type=int, default=512)
    p.add_argument("--MD_depth", type=int, default=3)
    p.add_argument("--world_model_width", type=int, default=1024)
    p.add_argument("--world_model_depth", type=int, default=2)
    p.add_argument("--seed", type=int, default=0)
    # ours: two more dict-only keys of the reference (tpv:262 "act_fn", tpv:253 "weight_decay")
    p.add_argument("--act_fn", type=str, default="relu", choices=["relu", "tanh", "sigmoid", "elu"],
                   help="hidden activation of the task encoder, motor decoder and world model")
    p.add_argument("--weight_decay", type=float, default=0.0, help="Adam's L2 term")
    p.add_argument("--lookahead", type=int, default=1,
                   help="steps unrolled through the world model per sample (config key tpv:277)")
    return p


META_KEYS = ("iter_per_episode", "dim_state", "dim_state_body", "dim_state_task", "dim_action", "exp_std")


def merge_dataset(files):
    """Concatenate the episode lists of several demo pickles; later files must agree with
    the first on the meta fields (tpv:94-114)."""
    merged = None
    for path in files:
        with open(path, "rb") as f:
            data = pickle.load(f)
        print(path, "is loaded")
        if merged is None:
            merged = data
            continue
        for key in META_KEYS:
            assert merged[key] == data[key], "dataset meta mismatch on %r" % key
        merged["episodes"] = merged["episodes"] + data["episodes"]
    return merged


class WindowDataset(torch_models.DatasetBase):
    """(s_t, s_{t+1}, a_t) windows over demonstration episodes, stored ONCE.

    The reference materialises X[N, L, 2*Db] / Y[N, L, Da] in float64 with every state row
    duplicated (tpv:133-156: 6.4 GB for 1e6 transitions at Db=400).  Here the episodes are
    packed as float32 `states[R, Db]`, `actions[R, Da]` plus `window_row[N]` (row of s_t; s_{t+1}
    is the next row) -- the layout the gather kernel reads from HBM.  `__getitem__`, `X`, `Y`
    reproduce the reference's tensors on demand."""

    def __init__(self, states, actions, window_row, lookahead=1, next_states=None):
        self.states = np.ascontiguousarray(states, dtype=np.float32)
        # cond "rel" (tpv:149-150): row r holds fp32(s_{r+1} - s_r), formed in float64 like the reference
        self.next_states = None if next_states is None else np.ascontiguousarray(next_states, dtype=np.float32)
        self.actions = np.ascontiguousarray(actions, dtype=np.float32)
        self.window_row = np.ascontiguousarray(window_row, dtype=np.int32)
        self.lookahead = int(lookahead)          # steps per window: rows r .. r+L of one episode
        self.normalize_x = self.normalize_y = False        # tpv:163-164
        self._dev = None

    def __len__(self):
        return len(self.window_row)

    def _steps(self, rows):
        return np.asarray(rows)[..., None] + np.arange(self.lookahead)

    def _second(self, r):
        return self.states[r + 1] if self.next_states is None else self.next_states[r]

    def __getitem__(self, index):
        r = self._steps(int(self.window_row[index]))                      # [L]
        x = np.concatenate([self.states[r], self._second(r)], axis=1)     # tpv:141-154
        return torch.from_numpy(x.copy()), torch.from_numpy(self.actions[r].copy())

    @property
    def X(self):
        r = self._steps(self.window_row)                                   # [N, L]
        return np.concatenate([self.states[r], self._second(r)], axis=2).astype(np.float64)

    @property
    def Y(self):
        return self.actions[self._steps(self.window_row)].astype(np.float64)

    def with_lookahead(self, lookahead):
        """Windows of `lookahead` steps over the same rows: keeps the starts whose next
        lookahead-1 rows are window starts too (i.e. the span stays inside one episode)."""
        lookahead = int(lookahead)
        if lookahead == self.lookahead:
            return self
        if self.lookahead != 1:
            raise ValueError("can only widen a lookahead-1 window list")
        starts = np.asarray(self.window_row, dtype=np.int64)
        have = np.zeros(len(self.states) + lookahead, dtype=bool)
        have[starts] = True
        keep = np.ones(len(starts), dtype=bool)
        for j in range(1, lookahead):
            keep &= have[starts + j]
        ds = WindowDataset(self.states, self.actions, starts[keep].astype(np.int32), lookahead, self.next_states)
        ds.meta = getattr(self, "meta", {})
        return ds

    def device_arrays(self, device):
        d = torch.device(device)
        have = None if self._dev is None else self._dev[0].device
        if have is None or have.type != d.type or (d.index is not None and d.index != have.index):
            def roomy(a):
                # (16 readable bytes behind the last row: the first layers fetch the rows in 16-byte chunks where they lie,
                #  include/pvae.h pvae_set_direct; without it the steps fall back to the staging launch)
                a = torch.from_numpy(a)
                buf = torch.empty(a.numel() + 16, dtype=a.dtype, device=device)
                buf[a.numel():].zero_()
                out = buf[: a.numel()].view(a.shape)
                out.copy_(a)
                return out
            self._dev = (roomy(self.states), roomy(self.actions), torch.from_numpy(self.window_row).to(device))
            if self.next_states is not None:
                self._dev = self._dev + (torch.from_numpy(self.next_states).to(device),)
        return self._dev


PACK_MAGIC = b"PVAEDEMO1\n"


def save_packed(dataset, path, meta=None):
    """Write a WindowDataset as ONE mmap-able file: magic, a JSON header line (dims, counts,
    byte offsets, the pickle's meta fields), then the raw little-endian arrays, each 64-byte
    aligned: states fp32 [R][Db], actions fp32 [R][Da], window_row int32 [N].  This is the
    layout the gather kernel reads from HBM, so loading is a straight copy (no float64 blow-up:
    config 5's 6.4 GB of X/Y becomes 1.96 GB)."""
    import json
    if dataset.next_states is not None:
        raise NotImplementedError("packed files hold cond='abs' windows (states stored once)")
    arrays = [("states", dataset.states), ("actions", dataset.actions), ("window_row", dataset.window_row)]
    header = {"version": 1, "dim_state_body": int(dataset.states.shape[1]),
              "dim_action": int(dataset.actions.shape[1]), "n_rows": int(dataset.states.shape[0]),
              "n_windows": int(len(dataset.window_row)), "lookahead": int(dataset.lookahead),
              "meta": meta or {}}
    pos = 4096                                   # header block is padded to 4 KB
    for name, arr in arrays:
        pos = (pos + 63) // 64 * 64
        header[name] = {"offset": pos, "dtype": str(arr.dtype), "shape": list(arr.shape)}
        pos += arr.nbytes
    blob = PACK_MAGIC + json.dumps(header).encode() + b"\n"
    assert len(blob) <= 4096, "header too large"
    with open(path, "wb") as f:
        f.write(blob.ljust(4096, b"\0"))
        for name, arr in arrays:
            f.seek(header[name]["offset"])
            f.write(np.ascontiguousarray(arr).tobytes())
    return path


def load_packed(path):
    """Memory-map a file written by save_packed (or tools/pack_demo.py) as a WindowDataset."""
    import json
    with open(path, "rb") as f:
        head = f.read(4096)
    if not head.startswith(PACK_MAGIC):
        raise ValueError("%s is not a packed PhysicsVAE demonstration file" % path)
    header = json.loads(head[len(PACK_MAGIC):].split(b"\n", 1)[0])
    arrs = {}
    for name in ("states", "actions", "window_row"):
        h = header[name]
        arrs[name] = np.memmap(path, mode="r", dtype=np.dtype(h["dtype"]), offset=h["offset"], shape=tuple(h["shape"]))
    ds = WindowDataset(arrs["states"], arrs["actions"], arrs["window_row"], header.get("lookahead", 1))
    ds.meta = header.get("meta", {})
    return ds


def load_dataset_for_PhysicsVAE(files, num_samples=None, lookahead=1, cond="abs", use_a_gt=False):
    """tpv:117-164.  Windows are emitted episode by episode, i ascending; `num_samples` caps the
    total at exactly that many (tpv:137-138).  Files ending in .pvd are packed demonstration
    files (save_packed); anything else is the reference's pickle."""
    assert files and len(files) > 0
    assert lookahead >= 1
    if cond not in ("abs", "rel"):
        raise NotImplementedError(cond)                                       # tpv:151-152
    if all(str(f).endswith(".pvd") for f in files):
        if cond != "abs":
            raise NotImplementedError("packed files hold cond='abs' windows; cond='rel' needs the float64 pickle")
        parts = [load_packed(f).with_lookahead(lookahead) for f in files]
        for p_ in parts[1:]:                     # same compatibility rule as merge_dataset
            assert p_.states.shape[1] == parts[0].states.shape[1] and p_.actions.shape[1] == parts[0].actions.shape[1]
            for key in META_KEYS:
                assert parts[0].meta.get(key) == p_.meta.get(key), "dataset meta mismatch on %r" % key
        if len(parts) == 1 and num_samples is None:
            ds = parts[0]
        else:
            offs = np.cumsum([0] + [len(p_.states) for p_ in parts[:-1]])
            rows = np.concatenate([np.asarray(p_.window_row, dtype=np.int64) + o for p_, o in zip(parts, offs)])
            if num_samples is not None:
                rows = rows[:num_samples]
            ds = WindowDataset(np.concatenate([p_.states for p_ in parts]),
                               np.concatenate([p_.actions for p_ in parts]), rows.astype(np.int32), lookahead)
        print("Packed demonstrations:", files, "windows:", len(ds))
        return ds
    data = merge_dataset(files)
    episodes = data["episodes"]
    states, actions, rows, deltas = [], [], [], []
    base = 0
    for ep in episodes:
        T = len(ep["time"])
        assert T >= lookahead
        sb = np.asarray(ep["state_body"], dtype=np.float64)
        ac = np.asarray(ep["action_gt" if use_a_gt else "action"], dtype=np.float64)
        n = T - lookahead
        if num_samples is not None:
            n = max(0, min(n, num_samples - len(rows)))
        states.append(sb.astype(np.float32))
        actions.append(ac.astype(np.float32))
        if cond == "rel":                        # tpv:149-150, in float64 as upstream; the last row has no successor
            deltas.append(np.concatenate([sb[1:] - sb[:-1], np.zeros((1, sb.shape[1]))]).astype(np.float32))
        rows.extend(range(base, base + n))
        base += T
    ds = WindowDataset(np.concatenate(states), np.concatenate(actions), np.asarray(rows, dtype=np.int32), lookahead,
                       np.concatenate(deltas) if cond == "rel" else None)
    ds.meta = {k: data.get(k) for k in META_KEYS}
    print("------------------Data Loaded------------------")
    print("File:", files)
    print("Num Episodes:", len(episodes))
    print("Num Transitions (Tuples):", len(ds))
    print("-----------------------------------------------")
    return ds


def create_model(config):
    mc = config["model"]
    cmc = mc["custom_model_config"]
    return PhysicsVAE(obs_space=cmc["observation_space"], action_space=cmc["action_space"],
                      num_outputs=2 * cmc["action_space"].shape[0], model_config=mc,
                      name="physics_vae")


MODEL_CONFIG = copy.deepcopy(PhysicsVAE.DEFAULT_CONFIG)


def gen_layers(width, depth, out_size="output", act_hidden="relu", act_out="linear", add_softmax=False):
    """tpv:180-192."""
    if add_softmax:
        raise NotImplementedError("softmax heads are not used by the PhysicsVAE trainer")
    return fc_spec(width, depth, out_size=out_size, act_hidden=act_hidden, act_out=act_out)


def inspect_dataset(path):
    if str(path).endswith(".pvd"):
        ds = load_packed(path)
        db, da = ds.states.shape[1], ds.actions.shape[1]
        return 2 * db, db, db, da
    with open(path, "rb") as f:
        ep0 = pickle.load(f)["episodes"][0]
    db, da = len(ep0["state_body"][0]), len(ep0["action"][0])
    return 2 * db, db, db, da


def get_trainer_config(a):
    """tpv:194-288.  s_t = (sb_t, sb_{t+1}): the model's observation is two body states."""
    assert a.max_iter_world_model <= a.max_iter
    dim_state, dim_body, dim_task, dim_action = inspect_dataset(a.data_train[0])

    def box(n, scale):
        return Box(low=-scale * np.ones(n), high=scale * np.ones(n), dtype=np.float64)

    # argparse appends to the very list object given as default: work on copies
    kl_list = [a.kl_coeff] if getattr(a, "kl_coeff", None) is not None else list(a.vae_kl_coeff)
    cyc_list = [a.cycle_coeff] if getattr(a, "cycle_coeff", None) is not None else list(a.vae_cycle_coeff)
    prior_list = [a.prior] if getattr(a, "prior", None) is not None else list(a.latent_prior_type)
    cmc = copy.deepcopy(MODEL_CONFIG)
    cmc.update(observation_space=box(dim_state, 1000.0), observation_space_body=box(dim_body, 1000.0),
               observation_space_task=box(dim_task, 1000.0), action_space=box(dim_action, 3.0),
               world_model_load_weights=a.world_model, max_batch=a.batch_size)
    return {
        "max_iter_world_model": a.max_iter_world_model,
        "model": {"custom_model": "physics_vae", "custom_model_config": cmc},
        "lr": a.lr,
        "lr_schedule_params": {"step_size": 50, "gamma": 0.70},
        "lr_schedule": a.lr_schedule,
        "weight_decay": getattr(a, "weight_decay", 0.0),
        "dataset_train": a.data_train,
        "dataset_test": a.data_test,
        "use_gpu": False,
        "loss": "MSE",
        "loss_test": "MSE",
        "batch_size": a.batch_size,
        "suffle_data": True,            # sic -- the key the reference sets; nothing reads it
        "latent_dim": a.latent_dim,
        "latent_prior_type": tune.grid_search(prior_list),
        "act_fn": getattr(a, "act_fn", "relu"),
        "MD_width": tune.grid_search([getattr(a, "MD_width", 512)]),
        "MD_depth": tune.grid_search([getattr(a, "MD_depth", 3)]),
        "TE_width": tune.grid_search([getattr(a, "TE_width", 256)]),
        "TE_depth": tune.grid_search([getattr(a, "TE_depth", 2)]),
        "lookahead": getattr(a, "lookahead", 1),       # tpv:277 hard-wires 1; --lookahead exposes it
        "world_model_width": tune.grid_search([getattr(a, "world_model_width", 1024)]),
        "world_model_depth": tune.grid_search([getattr(a, "world_model_depth", 2)]),
        "vae_kl_coeff": tune.grid_search(kl_list),
        "motor_decoder_a_rec_coeff": 1.0,
        "world_model_s_rec_coeff": 0.0,
        "vae_cycle_coeff": tune.grid_search(cyc_list),
        "seed": getattr(a, "seed", 0),
    }


def update_model_config(trainer_config):
    """tpv:290-311: expand width/depth into layer-spec lists inside custom_model_config.  Ours: a trainer-config
    key `TE_layers` / `MD_layers` / `world_model_layers` holding a full layer list (the dicts FC.__init__ reads,
    rmt:234-270: per-layer hidden_size / activation / init_weight) takes the place of gen_layers' uniform stack --
    the model accepts such lists upstream too, only the trainer's helper cannot emit them."""
    cmc = trainer_config["model"]["custom_model_config"]
    act = trainer_config.get("act_fn")
    cmc["task_encoder_output_dim"] = trainer_config.get("latent_dim")
    cmc["latent_prior_type"] = trainer_config.get("latent_prior_type")
    for key, prefix in (("task_encoder_layers", "TE"), ("motor_decoder_layers", "MD"),
                        ("world_model_layers", "world_model")):
        cmc[key] = trainer_config.get(prefix + "_layers") or gen_layers(
            width=trainer_config.get(prefix + "_width"), depth=trainer_config.get(prefix + "_depth"), act_hidden=act)
    cmc["max_batch"] = trainer_config.get("batch_size", cmc.get("max_batch", 256))
    cmc["lookahead"] = trainer_config.get("lookahead", 1) or 1     # sizes the unroll workspace


class TrainModel(torch_models.TrainModel):
    """tpv:313-467."""

    def setup(self, config):
        update_model_config(config)
        self.config = config
        self.max_iter_world_model = config.get("max_iter_world_model")
        self.latent_prior_type = config.get("latent_prior_type")
        self.lookahead = config.get("lookahead")
        super().setup(config)
        self.model.set_learnable_task_encoder(False)
        self.model.set_learnable_motor_decoder(False)
        self.model.set_learnable_world_model(True)
        self.model.set_learnable_latent_prior(False)      # (ours) trains with the encoder, see model.py
        self.read_loss_fn_coeff(world=True)

    def read_loss_fn_coeff(self, world):
        c = self.config
        self.vae_kl_coeff = 0.0 if world else c.get("vae_kl_coeff")
        self.a_rec_coeff = 0.0 if world else c.get("motor_decoder_a_rec_coeff")
        self.s_rec_coeff = 1.0 if world else c.get("world_model_s_rec_coeff")
        self.vae_cycle_coeff = 0.0 if world else c.get("vae_cycle_coeff")

    def load_dataset(self, file):
        num = getattr(args, "num_data", None) if args is not None else None
        return load_dataset_for_PhysicsVAE(file, num_samples=num, lookahead=self.lookahead)

    def _enter_joint_phase(self):                 # tpv:342-348
        self.model.set_learnable_task_encoder(True)
        self.model.set_learnable_motor_decoder(True)
        self.model.set_learnable_world_model(False)
        self.model.set_learnable_latent_prior(True)
        self.read_loss_fn_coeff(world=False)

    def step(self):
        # the flip is tested BEFORE the increment: epochs 1..M are world, M+1.. joint (tpv:342)
        if self.iter == self.max_iter_world_model:
            self._enter_joint_phase()
        return super().step()

    def load_trainer_state(self, path):
        """(ours) Besides what the base class restores, the PHASE: `setup` has put the trainer into the world
        phase and `step` flips only when iter == max_iter_world_model, so a state saved after the switch
        (iter > M) must re-enter the joint phase here -- otherwise the resumed run would train the world model
        forever on joint-phase Adam counters."""
        state = super().load_trainer_state(path)
        m = self.max_iter_world_model
        if m is not None and self.iter > m:
            self._enter_joint_phase()
        saved = state.get("learnable_nets")
        if saved is not None and sorted(saved) != sorted(self.model.learnable_nets()):
            raise RuntimeError("trainer_state.pt: learnable stacks %s do not match the phase its epoch counter "
                               "implies (%s); was max_iter_world_model changed?" % (saved, self.model.learnable_nets()))
        return state

    def create_model(self, config):
        return create_model(config)

    def step_params(self, nets, global_rows, train):
        t = self.optimizer.next_counts(nets) if train else [1, 1, 1, 1, 0]
        return make_step_params(lr=self.optimizer.lr, adam_t=t, a_rec=self.a_rec_coeff,
                                kl=self.vae_kl_coeff, s_rec=self.s_rec_coeff, cyc=self.vae_cycle_coeff,
                                global_rows=global_rows, loss=self.loss_name,
                                weight_decay=self.optimizer.param_groups[0]["weight_decay"])

    # explicit-batch entry points with the reference's signatures (tpv:356-435) -------------
    def compute_model(self, x, eps=None):
        logits, _ = self.model(input_dict={"obs": x, "obs_flat": x}, state=None, seq_lens=None)
        return logits[..., : logits.shape[1] // 2]

    def compute_loss(self, y, x, eps=None):
        """Loss of one caller-supplied minibatch (x [B,L,2Db], y [B,L,Da]), forward only.
        Returns the 0-dim total (device tensor); per-term values in `self.last_loss_terms`."""
        phase, nets = self.phase()
        rows = self.engine.set_batch(x, y)
        sp = make_step_params(lr=self.optimizer.lr, a_rec=self.a_rec_coeff, kl=self.vae_kl_coeff,
                              s_rec=self.s_rec_coeff, cyc=self.vae_cycle_coeff, global_rows=rows,
                              seed=self.rng_seed, offset=self.global_batch * 65536, loss=self.loss_name)
        out = torch.zeros(5, dtype=torch.float32, device=self.engine.device)
        self.engine.forward_backward(phase, rows, sp, eps=eps, backward=False, loss_out=out)
        self.last_loss_terms = out
        return out[0]

    def save_checkpoint(self, checkpoint_dir):
        path = super().save_checkpoint(checkpoint_dir)
        m = self.model
        for fname, fn in (("model.pt", m.save_weights), ("task_encoder.pt", m.save_weights_task_encoder),
                          ("motor_decoder.pt", m.save_weights_motor_decoder),
                          ("world_model.pt", m.save_weights_world_model)):
            target = os.path.join(checkpoint_dir, fname)
            fn(target)
            print("Saved:", target)
        if m._latent_prior is not None:               # tpv:462-466: the sixth file of a model with a learned prior mean
            target = os.path.join(checkpoint_dir, "latent_prior.pt")
            m.save_weights_latent_prior(target)
            print("Saved:", target)
        return path


def main(argv=None):
    global args
    args = arg_parser().parse_args(argv)
    from . import parallel
    parallel.init_from_env()
    torch.manual_seed(args.seed)
    trainer_config = get_trainer_config(args)
    if args.checkpoint is None:
        analysis = tune.run(TrainModel, stop={"training_iteration": args.max_iter},
                            checkpoint_freq=args.checkpoint_freq, checkpoint_at_end=True,
                            config=trainer_config, local_dir=args.local_dir, name=args.name, resume=args.resume)
        best = analysis.get_best_logdir(metric="training_iteration", mode="max")          # tpv:503-506
        checkpoint = analysis.get_best_checkpoint(logdir=best)
        best_config = next(t.config for t in analysis.trials if t.logdir == best)
    else:
        checkpoint = args.checkpoint
        best_config = tune.expand_grid(trainer_config)[0]
    if args.output is not None:
        # the reference's --output branch instantiates the abstract base and cannot work
        # (SURVEY.md App. C-3); here it exports the full state_dict as intended
        trainer = TrainModel(best_config)
        trainer.restore(checkpoint)
        torch.save(trainer.model.portable_state_dict(), args.output)
        print("Model Saved:", args.output)


if __name__ == "__main__":
    main()
