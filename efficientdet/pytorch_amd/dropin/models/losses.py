"""Import-path shim for ``models.losses`` (reference models/losses.py:29-152; imported by train.py:31)."""
from efficientdet.pytorch_amd.efficientdet import FocalLoss  # noqa: F401
