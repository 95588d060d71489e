"""utils/config_eff.py:1-41 of the reference: the EFFICIENTDET scale table."""
from efficientdet.pytorch_amd.config import EFFICIENTDET  # noqa: F401
