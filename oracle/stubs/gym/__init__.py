"""Dev-container-only stand-in for `gym` (only spaces.Box is touched)."""
from . import spaces  # noqa: F401
