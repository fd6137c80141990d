"""Per-shape timing of the three contraction kernels through pvae_gemm_probe (HIP events
on the launch stream, back-to-back launches).  Run on the GPU box:
    python tools/gemm_bench.py [--iters 200]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from physicsvae_amd.engine import gemm_probe  # noqa: E402

SHAPES = {  # (M rows, N out, K in) of the 4x1024 stacks at batch 256
    "first  242->1024": (256, 1024, 256),
    "hidden 1024->1024": (256, 1024, 1024),
    "last   1024->197": (256, 256, 1024),
    "last   1024->64": (256, 64, 1024),
    "te in  394->1024": (256, 1024, 448),
    "hidden b512": (512, 1024, 1024),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rows", type=int, nargs="*", default=[], help="also time the 1024 x 1024 hidden layer at these row counts")
    a = ap.parse_args()
    for r in a.rows:
        SHAPES["hidden b%d" % r] = (r, 1024, 1024)
    dev = "cuda"
    print("%-20s %-8s %10s %10s" % ("shape", "kind", "us", "TFLOP/s"))
    for name, (m, n, k) in SHAPES.items():
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        dz = torch.randn(m, n, device=dev)
        b = torch.randn(n, device=dev)
        out_f = torch.empty(m, n, device=dev)
        out_d = torch.empty(m, k, device=dev)
        out_w = torch.empty(n, k, device=dev)
        runs = {"forward": lambda: gemm_probe(0, x, w, out_f, bias_or_mask=b, relu=True, m=m, n=n, k=k),
                "dgrad": lambda: gemm_probe(1, dz, w, out_d, bias_or_mask=x, m=m, n=n, k=k),
                "wgrad": lambda: gemm_probe(2, dz, x, out_w, m=m, n=n, k=k)}
        for kind, fn in runs.items():
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print("%-20s %-8s %10.2f %10.1f" % (name, kind, us, 2.0 * m * n * k / us / 1e6))


if __name__ == "__main__":
    main()
