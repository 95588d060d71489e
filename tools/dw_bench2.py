"""The 16 depthwise launches of an EfficientNet-B0 train step (batch 32 @ 512, fp32 storage, z-only expand storage) one by one:
forward (Swish on the staged tile, z + pooled sums out), data gradient (Swish' of the expand pre-activation fused), weight
gradient -- microseconds and algorithmic TB/s each.  DW_ONLY=fwd,dgrad,wgrad selects."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

dev, dt, B = 'cuda', torch.float32, int(os.environ.get('DW_B', 32))
only = os.environ.get('DW_ONLY', 'fwd,dgrad,wgrad').split(',')
BLOCKS = [(32, 256, 3, 1, 1), (96, 256, 3, 2, 6), (144, 128, 3, 1, 6), (144, 128, 5, 2, 6), (240, 64, 5, 1, 6), (240, 64, 3, 2, 6),
          (480, 32, 3, 1, 6), (480, 32, 5, 1, 6), (672, 32, 5, 1, 6), (672, 32, 5, 2, 6), (1152, 16, 5, 1, 6), (1152, 16, 3, 1, 6),
          (672, 16, 5, 2, 6), (1152, 8, 5, 1, 6), (1152, 8, 3, 2, 6)]          # (the last three: this reference's B0 strides -- 8x8 / 4x4 maps)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
for (C, H, k, s, e) in BLOCKS:
    x = Map.of(torch.randn(B, H, H, C, device=dev))
    w = torch.randn(k * k, C, device=dev) * 0.1
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    Ho = H // s
    plo = (k - 1) // 2 if s == 1 else (k - 2) // 2
    act = ops.ACT_SWISH if e != 1 else ops.ACT_NONE
    line = 'C%-4d %3d^2 k%d s%d ' % (C, H, k, s)
    if 'fwd' in only:
        us = timeit(lambda: ops.dwconv_fwd(x, w, sc, sh, k, s, plo, plo, Ho, Ho, save_z=True, pool=True, save_y=False, in_act=act))
        tot['fwd'] += us
        line += ' fwd %6.1f us %4.2f TB/s ' % (us, 4 * B * C * (H * H + Ho * Ho) / us / 1e6)
    dz = Map.of(torch.randn(B, Ho, Ho, C, device=dev))
    if 'dgrad' in only:
        us = timeit(lambda: ops.dwconv_dgrad(dz, w, sc, x if e != 1 else None, H, H, k, s, plo, plo))
        tot['dgrad'] += us
        line += ' dgrad %6.1f us %4.2f TB/s ' % (us, 4 * B * C * ((2 if e != 1 else 1) * H * H + Ho * Ho) / us / 1e6)
    if 'wgrad' in only:
        us = timeit(lambda: ops.dwconv_wgrad(x, dz, k, s, plo, plo, in_act=act))
        tot['wgrad'] += us
        line += ' wgrad %6.1f us %4.2f TB/s' % (us, 4 * B * C * (H * H + Ho * Ho) / us / 1e6)
    print(line, flush=True)
print('sum (15 shapes) ', {k_: round(v, 1) for k_, v in tot.items()})
