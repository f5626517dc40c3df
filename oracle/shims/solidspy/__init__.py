"""Stand-in for solidspy>=1.0.16 (not installed; PARITY UNPINNED)."""
from . import uelutil  # noqa: F401
