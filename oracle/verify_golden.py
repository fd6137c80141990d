"""oracle/verify_golden.py -- is every committed fixture still what the reference produces?

Test infrastructure, dev container only (it imports /root/reference through oracle/gen_golden.py).  Regenerates every
fixture of tests/golden/ into a scratch directory with the committed generator and compares it with the committed file
array by array, byte for byte (names, dtypes, shapes, contents).  A generator change that silently alters a capture, or
a fixture that was edited by hand, shows up here.  `tests/test_oracle_golden.py::test_fixtures_regenerate_byte_for_byte`
runs it where /root/reference exists and skips elsewhere (the GPU box has no reference).

    python oracle/verify_golden.py [--only NAME ...] [--jobs N]
"""
import argparse
import concurrent.futures
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("PVAE_REFERENCE", "/root/reference")


def fixture_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def compare(a_path, b_path):
    """-> list of differences between two .npz files (empty: identical arrays)."""
    a, b = np.load(a_path, allow_pickle=False), np.load(b_path, allow_pickle=False)
    diffs = []
    if sorted(a.files) != sorted(b.files):
        diffs.append("keys differ: only committed %s, only regenerated %s" %
                     (sorted(set(a.files) - set(b.files))[:5], sorted(set(b.files) - set(a.files))[:5]))
    for k in sorted(set(a.files) & set(b.files)):
        x, y = a[k], b[k]
        if x.dtype != y.dtype or x.shape != y.shape:
            diffs.append("%s: %s%s committed, %s%s regenerated" % (k, x.dtype, x.shape, y.dtype, y.shape))
        elif x.tobytes() != y.tobytes():
            diffs.append("%s: contents differ" % k)
    return diffs


def regenerate(name, out_dir):
    """One fixture through the committed generator, written to out_dir (gen_golden.OUT redirected)."""
    code = ("import sys; sys.argv = ['gen_golden.py', '--only', %r]; sys.path.insert(0, %r); import gen_golden as G; "
            "G.OUT = %r; G.main()" % (name, HERE, out_dir))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    if res.returncode != 0:
        return "generator failed: " + (res.stderr or res.stdout)[-800:]
    if not os.path.exists(os.path.join(out_dir, name + ".npz")):
        return "the generator knows no fixture of this name"
    return None


def verify(names=None, jobs=None):
    """-> {name: [differences]} for the fixtures that do NOT regenerate byte for byte."""
    if not os.path.isdir(REF):
        raise RuntimeError("the reference is not present at %s" % REF)
    names = list(names or fixture_names())
    bad = {}
    with tempfile.TemporaryDirectory(prefix="pvae_verify_") as td:
        def one(n):
            err = regenerate(n, td)
            return n, ([err] if err else compare(os.path.join(GOLDEN, n + ".npz"), os.path.join(td, n + ".npz")))
        with concurrent.futures.ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as pool:
            for n, d in pool.map(one, names):
                if d:
                    bad[n] = d
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    todo = a.only or fixture_names()
    bad = verify(todo, a.jobs)
    for n in todo:
        print("%-28s %s" % (n, "DIFFERS: " + "; ".join(bad[n][:4]) if n in bad else "byte-identical"))
    sys.exit(1 if bad else 0)
