"""MI355X-native EfficientDet forward/backward path (hand-written HIP for gfx950 behind the
reference's nn.Module surface).  See DESIGN.md / INTEGRATION.md.

    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET     # replaces models.efficientdet / utils.config_eff
    (or: put efficientdet/pytorch_amd/dropin on the path and keep `from models.efficientdet import EfficientDet`)
"""
from .checkpoint import get_state_dict  # noqa: F401
from .config import EFFICIENTDET, MODEL_MAP  # noqa: F401
from .efficientdet import EfficientDet, PackedImages  # noqa: F401
from .synthetic import synthetic_batch  # noqa: F401
