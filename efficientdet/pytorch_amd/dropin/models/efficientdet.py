"""Import-path shim for ``models.efficientdet`` (reference models/efficientdet.py:10-100): the MI355X-native module."""
from efficientdet.pytorch_amd.config import MODEL_MAP  # noqa: F401
from efficientdet.pytorch_amd.efficientdet import Anchors, EfficientDet, FocalLoss  # noqa: F401
