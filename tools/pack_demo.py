"""pickle -> packed demonstration file (one pass; see physicsvae_amd.train_physics_vae.save_packed).
    python tools/pack_demo.py demo_a.pkl [demo_b.pkl ...] -o demo.pvd [--num_data N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsvae_amd import train_physics_vae as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--num_data", type=int, default=None)
    a = ap.parse_args()
    ds = T.load_dataset_for_PhysicsVAE(a.files, num_samples=a.num_data)
    T.save_packed(ds, a.output, meta=ds.meta)
    print("wrote %s: %d rows, %d windows, %.1f MB" % (a.output, len(ds.states), len(ds), os.path.getsize(a.output) / 1e6))


if __name__ == "__main__":
    main()
