"""Minimal ``utils`` (reference utils/__init__.py: helper + config_eff re-exports used by train.py:33 / eval.py:16)."""
from .config_eff import EFFICIENTDET  # noqa: F401
from .helper import get_state_dict  # noqa: F401
