"""Is the depthwise forward / the expand conv bound by its WRITES?  Times dw fwd with and without the pre-activation copy
(z) and the 1x1 expand conv with and without z, on the three largest EfficientNet-B0 maps at batch 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

dev, dt, B = 'cuda', torch.bfloat16, 32


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


for (C, H, k, s) in [(32, 256, 3, 1), (96, 256, 3, 2), (144, 128, 3, 1), (144, 128, 5, 2), (240, 64, 5, 1)]:
    x = Map.of(torch.randn(B, H, H, C, device=dev).to(dt))
    w = torch.randn(k * k, C, device=dev) * 0.1
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    Ho = H // s
    plo = (k - 1) // 2 if s == 1 else (k - 2) // 2
    pool = torch.zeros(B, C, device=dev)
    for save_z in (True, False):
        us = timeit(lambda: ops.dwconv_fwd(x, w, sc, sh, k, s, plo, plo, Ho, Ho, save_z=save_z, pool=True))
        by = 2 * B * C * (H * H + Ho * Ho * (2 if save_z else 1))
        print('dw fwd C%-4d %3d^2 k%d s%d z=%d  %7.1f us  %5.2f TB/s' % (C, H, k, s, save_z, us, by / us / 1e6))
    dz = Map.of(torch.randn(B, Ho, Ho, C, device=dev).to(dt))
    us = timeit(lambda: ops.dwconv_dgrad(dz, w, sc, x, H, H, k, s, plo, plo))
    print('dw dgrad C%-4d %3d^2 k%d s%d      %7.1f us  %5.2f TB/s' % (C, H, k, s, us, 2 * B * C * (2 * H * H + Ho * Ho) / us / 1e6))
    if not os.environ.get('DW_BENCH_NO_WGRAD'):
        us = timeit(lambda: ops.dwconv_wgrad(x, dz, k, s, plo, plo))
        print('dw wgrad C%-4d %3d^2 k%d s%d      %7.1f us  %5.2f TB/s' % (C, H, k, s, us, 2 * B * C * (H * H + Ho * Ho) / us / 1e6))

for (Cin, Cout, H) in [(16, 96, 256), (24, 144, 128), (40, 240, 64)]:
    x = Map.of(torch.randn(B, H, H, Cin, device=dev).to(dt))
    wp = ops.pack_weight(torch.randn(Cout, Cin, 1, 1, device=dev) * 0.1, dt)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    y = Map.new(B, H, H, Cout, dt, dev); z = Map.new(B, H, H, Cout, dt, dev)
    for save_z in (True, False):
        us = timeit(lambda: ops.conv2d(x, wp, y, Cin=Cin, Cout=Cout, KH=1, KW=1, scale=sc, shift=sh, act=ops.ACT_SWISH,
                                       zs=z if save_z else None))
        by = 2 * B * H * H * (Cin + Cout * (2 if save_z else 1))
        print('expand %3d->%-4d %3d^2 z=%d  %7.1f us  %5.2f TB/s' % (Cin, Cout, H, save_z, us, by / us / 1e6))
