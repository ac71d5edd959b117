"""Stand-in for absl (only `absl.logging`, which nerfactor/util/logging.py imports)."""
from . import logging  # noqa: F401
