import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: a GPU test that takes a minute or more (full-size training runs)")


def pytest_sessionstart(session):
    """Make sure libpvae_gfx950.so exists and is not older than its sources (hipcc cross-compiles
    without a GPU, ~40 s).  A missing compiler is only an error if there is no library at all."""
    from physicsvae_amd import build
    try:
        build.build(force=False)
    except Exception as exc:                                       # noqa: BLE001
        if not os.path.exists(build.LIB):
            raise pytest.UsageError("libpvae_gfx950.so is missing and could not be built: %s" % exc)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
