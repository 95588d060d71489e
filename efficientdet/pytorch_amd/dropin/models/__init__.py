"""models/__init__.py:1 of the reference: ``from .efficientdet import EfficientDet``."""
from .efficientdet import EfficientDet  # noqa: F401
