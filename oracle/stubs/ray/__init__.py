"""Dev-container-only stand-in for the `ray` package (ray[rllib]==1.11.0 is not
installable offline).  It exists so that `oracle/gen_golden.py` can import the
reference's three hot-path modules unmodified and record golden vectors.  Nothing
here ships in the product path; nothing here is derived from ray's source."""
from . import tune  # noqa: F401


def init(*args, **kwargs):
    return None


def shutdown(*args, **kwargs):
    return None
