"""The stand-alone minibatch gather (stage_batch_kernel) on its own: microseconds per launch and bytes per second at the
configs[2] and configs[4] dims, 256 / 512 and 8192 windows per launch, windows taken from a 1e6-transition demonstration set
that lives in HBM (1000 episodes x 1001 steps).  Algorithmic bytes = (2 Db + Da) * 4 read + the same written per window
(SURVEY.md 8d); "panel bytes" = what the launch really writes (every panel row of every consumer, pad columns included)
plus what it reads.   python tools/gather_bench.py      (PVAE_LIB_PATH=ab_libs/libG0.so: an A/B build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from physicsvae_amd.engine import Arch, HipEngine  # noqa: E402
from physicsvae_amd.train_physics_vae import WindowDataset  # noqa: E402


def pad64(n):
    return (n + 63) // 64 * 64


def main():
    dev = "cuda"
    for Db, Da in ((197, 45), (400, 90)):
        gen = torch.Generator(device=dev).manual_seed(0)
        E, T = 1000, 1001
        states = torch.randn(E * T, Db, generator=gen, device=dev)
        actions = torch.randn(E * T, Da, generator=gen, device=dev).clamp_(-3, 3)
        rows_idx = (torch.arange(E, device=dev)[:, None] * T + torch.arange(T - 1, device=dev)[None, :]).reshape(-1).to(torch.int32)
        n_win = rows_idx.numel()
        Z = 32
        panel_floats = pad64(2 * Db) + pad64(Db + Z) + pad64(Db + Da) + pad64(Db) + pad64(Da)
        for rows in (256, 512, 8192):
            eng = HipEngine(Arch(Db, Da, Z, (64, 1), (64, 1), (64, 1)), rows, device=dev)
            eng.bind_dataset(states, actions, rows_idx)
            for _ in range(5):
                eng.gather(0, rows)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 100
            e0.record()
            for i in range(n):
                eng.gather((i * 9973 * 97) % (n_win - rows), rows)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            alg = 2.0 * rows * (2 * Db + Da) * 4
            real = rows * ((2 * Db + Da) * 4 + panel_floats * 4)
            print("Db %3d Da %2d  %5d windows: %6.2f us / launch   algorithmic %6.2f MB = %5.2f TB/s (%4.1f %% of 8 TB/s)   "
                  "panel traffic %6.2f MB = %5.2f TB/s" % (Db, Da, rows, us, alg / 1e6, alg / us / 1e6, alg / us / 1e6 / 8 * 100,
                                                          real / 1e6, real / us / 1e6))
            del eng


if __name__ == "__main__":
    main()
