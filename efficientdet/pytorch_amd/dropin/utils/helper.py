"""utils/helper.py:25-30 of the reference (also unwraps DistributedDataParallel, which the reference forgets)."""
from efficientdet.pytorch_amd.checkpoint import get_state_dict  # noqa: F401
