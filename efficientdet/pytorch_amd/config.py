"""Model scale tables and EfficientNet architecture arithmetic (host logic, pure Python).

Mirrors, by behaviour, the reference's utils/config_eff.py:1-41 (EFFICIENTDET), models/efficientdet.py:10-19
(MODEL_MAP) and the width/depth scaling of models/utils.py:55-76,171-286 -- including this reference's edited
block table whose stages 5 and 7 are stride 2 (models/utils.py:264-269), so the seven stage outputs sit at
strides 2..128 and the neck consumes the last five.
"""
import math
from collections import namedtuple

EFFICIENTDET = {
    'efficientdet-d0': {'input_size': 512, 'backbone': 'B0', 'W_bifpn': 64, 'D_bifpn': 2, 'D_class': 3},
    'efficientdet-d1': {'input_size': 640, 'backbone': 'B1', 'W_bifpn': 88, 'D_bifpn': 3, 'D_class': 3},
    'efficientdet-d2': {'input_size': 768, 'backbone': 'B2', 'W_bifpn': 112, 'D_bifpn': 4, 'D_class': 3},
    'efficientdet-d3': {'input_size': 896, 'backbone': 'B3', 'W_bifpn': 160, 'D_bifpn': 5, 'D_class': 4},
    'efficientdet-d4': {'input_size': 1024, 'backbone': 'B4', 'W_bifpn': 224, 'D_bifpn': 6, 'D_class': 4},
    'efficientdet-d5': {'input_size': 1280, 'backbone': 'B5', 'W_bifpn': 288, 'D_bifpn': 7, 'D_class': 4},
    'efficientdet-d6': {'input_size': 1408, 'backbone': 'B6', 'W_bifpn': 384, 'D_bifpn': 8, 'D_class': 5},
    'efficientdet-d7': {'input_size': 1636, 'backbone': 'B6', 'W_bifpn': 384, 'D_bifpn': 8, 'D_class': 5},
}

MODEL_MAP = {
    'efficientdet-d0': 'efficientnet-b0', 'efficientdet-d1': 'efficientnet-b1', 'efficientdet-d2': 'efficientnet-b2',
    'efficientdet-d3': 'efficientnet-b3', 'efficientdet-d4': 'efficientnet-b4', 'efficientdet-d5': 'efficientnet-b5',
    'efficientdet-d6': 'efficientnet-b6', 'efficientdet-d7': 'efficientnet-b6',
}

# name -> (width multiplier, depth multiplier, native resolution)
BACKBONES = {
    'efficientnet-b0': (1.0, 1.0, 224), 'efficientnet-b1': (1.0, 1.1, 240), 'efficientnet-b2': (1.1, 1.2, 260),
    'efficientnet-b3': (1.2, 1.4, 300), 'efficientnet-b4': (1.4, 1.8, 380), 'efficientnet-b5': (1.6, 2.2, 456),
    'efficientnet-b6': (1.8, 2.6, 528), 'efficientnet-b7': (2.0, 3.1, 600),
}

BN_EPS = 1e-3
DROP_CONNECT_RATE = 0.2

Stage = namedtuple('Stage', 'repeat kernel stride expand cin cout')
STAGES = (Stage(1, 3, 1, 1, 32, 16), Stage(2, 3, 2, 6, 16, 24), Stage(2, 5, 2, 6, 24, 40), Stage(3, 3, 2, 6, 40, 80),
          Stage(3, 5, 2, 6, 80, 112), Stage(4, 5, 2, 6, 112, 192), Stage(1, 3, 2, 6, 192, 320))

Block = namedtuple('Block', 'k stride expand cin cout cexp cse pad skip stage_end')


def scale_width(channels, mult, divisor=8):
    """Channel rounding to multiples of 8, never dropping more than 10 %."""
    if not mult:
        return channels
    scaled = channels * mult
    rounded = max(divisor, int(scaled + divisor / 2) // divisor * divisor)
    if rounded < 0.9 * scaled:
        rounded += divisor
    return int(rounded)


def scale_depth(repeats, mult):
    return int(math.ceil(mult * repeats)) if mult else repeats


def tf_same_pad(native_size, k, stride):
    """(lo, hi) zero padding, computed ONCE from the backbone's native resolution and applied at every
    feature size (the reference's static-same-padding quirk)."""
    out = math.ceil(native_size / stride)
    total = max((out - 1) * stride + k - native_size, 0)
    return total // 2, total - total // 2


def conv_out(size, k, stride, pad):
    return (size + pad[0] + pad[1] - k) // stride + 1


def backbone_plan(backbone):
    """-> (stem_channels, stem_pad, [Block...], head_channels, native_size)."""
    if backbone not in BACKBONES:
        raise ValueError('model_name should be one of: ' + ', '.join(sorted(BACKBONES)))
    width, depth, native = BACKBONES[backbone]
    blocks = []
    for st in STAGES:
        cin, cout, rep = scale_width(st.cin, width), scale_width(st.cout, width), scale_depth(st.repeat, depth)
        for j in range(rep):
            bi = cin if j == 0 else cout
            bs = st.stride if j == 0 else 1
            blocks.append(Block(k=st.kernel, stride=bs, expand=st.expand, cin=bi, cout=cout, cexp=bi * st.expand,
                                cse=max(1, int(bi * 0.25)), pad=tf_same_pad(native, st.kernel, bs),
                                # only the repeat blocks of a stage take the identity skip in the reference
                                skip=(j > 0 and bi == cout), stage_end=(j == rep - 1)))
    return scale_width(32, width), tf_same_pad(native, 3, 2), blocks, scale_width(1280, width), native
