"""Synthetic expert-demonstration files in the reference's pickle schema (tpv:57-92; writer
envs/rllib_env_imitation.py:63-87, 140-144), for benchmarks and measurement tools: N(0,1) body
states, N(0,1) actions clipped to +-3 (the action-space bound of tpv:217, 230-233).  Nothing here
is on the product path or in the parity checker (the tests generate their inputs in oracle/).

    python tools/synth_demo.py out.pkl [--episodes 10 --steps 1000 --dim_body 197 --dim_action 45]
"""
import argparse
import pickle

import numpy as np


def synth_demo(seed, n_episodes, n_steps, dim_body, dim_action, quantum=1024.0):
    rng = np.random.default_rng(seed)
    episodes = []
    for _ in range(n_episodes):
        act = np.clip(rng.standard_normal((n_steps, dim_action)), -3.0, 3.0)
        sb = rng.standard_normal((n_steps, dim_body))
        if quantum:                                # exactly representable in fp32 on every platform
            sb = np.round(sb * quantum) / quantum
            act = np.round(act * quantum) / quantum
        nxt = np.minimum(np.arange(n_steps) + 1, n_steps - 1)
        episodes.append({
            "time": [float(t) / 30.0 for t in range(n_steps)],
            "state": [np.concatenate([sb[t], sb[nxt[t]]]) for t in range(n_steps)],
            "state_body": [sb[t].copy() for t in range(n_steps)],
            "state_task": [sb[nxt[t]].copy() for t in range(n_steps)],
            "action": [act[t].copy() for t in range(n_steps)],
            "reward": [0.0] * n_steps,
        })
    return {"dim_action": dim_action, "dim_state": 2 * dim_body, "dim_state_body": dim_body,
            "dim_state_task": dim_body, "exp_std": 0.05, "iter_per_episode": 1, "episodes": episodes}


def write_demo(path, data):
    with open(path, "wb") as f:
        pickle.dump(data, f)


def make_trainer(data, batch, device, width=1024, depth=4, latent=32, m_world=10 ** 9, extra=None):
    """physicsvae_amd TrainModel over a synthetic demo dict (written to a temporary pickle)."""
    import os
    import tempfile

    from physicsvae_amd import train_physics_vae as T
    td = tempfile.mkdtemp(prefix="pvae_bench_")
    pkl = os.path.join(td, "demo.pkl")
    write_demo(pkl, data)
    argv = ["--data_train", pkl, "--batch_size", str(batch), "--max_iter_world_model", str(min(m_world, 10 ** 9)),
            "--max_iter", str(10 ** 9), "--latent_dim", str(latent)]
    for p in ("TE", "MD", "world_model"):
        argv += ["--%s_width" % p, str(width), "--%s_depth" % p, str(depth)]
    T.args = T.arg_parser().parse_args(argv)
    cfg = T.get_trainer_config(T.args)
    cfg["model"]["custom_model_config"]["device"] = device
    cfg.update(extra or {})
    tr = T.TrainModel(cfg)
    tr._tmpdir = td
    return tr


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--dim_body", type=int, default=197)
    ap.add_argument("--dim_action", type=int, default=45)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    write_demo(a.out, synth_demo(a.seed, a.episodes, a.steps, a.dim_body, a.dim_action))
    print("wrote", a.out)
