"""Stand-in for absl (only `absl.logging`, which nerfactor/util/logging.py imports)."""
from . import app, flags, logging  # noqa: F401
