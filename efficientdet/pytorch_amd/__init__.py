"""MI355X-native EfficientDet forward/backward path (hand-written HIP for gfx950 behind the
reference's nn.Module surface).  See DESIGN.md / INTEGRATION.md.

    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET     # replaces models.efficientdet / utils.config_eff
"""
from .config import EFFICIENTDET, MODEL_MAP  # noqa: F401
from .efficientdet import EfficientDet  # noqa: F401
