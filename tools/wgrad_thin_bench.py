"""The high-resolution pointwise weight gradients of a D0 train step (B = 32 @ 512), one by one: microseconds and algorithmic TB/s.
Run twice (EFFDET_WGRAD_THIN=0 / 1) to compare conv_wgrad_thin_kernel with the tiled kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

ops.set_f32_arith(os.environ.get('WG_ARITH', 'bf16x3'))
dev, B = 'cuda', 32
SHAPES = [(256, 32, 16, True), (256, 16, 96, False), (128, 96, 24, True), (128, 24, 144, False), (128, 144, 24, True), (64, 144, 40, True),
          (64, 40, 240, False)]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (H, Cin, Cout, img) in SHAPES:
    x = Map.of(torch.randn(B, H, H, Cin, device=dev)); dz = Map.of(torch.randn(B, H, H, Cout, device=dev))
    kid = ops.conv2d_wgrad_kernel_id(x, dz, Cin=Cin, Cout=Cout, KH=1, KW=1)
    try:
        us = timeit(lambda: ops.conv2d_wgrad(x, dz, Cin=Cin, Cout=Cout, KH=1, KW=1, image_splits=img))
    except RuntimeError as e:
        print('%3d^2 %3d->%-3d failed: %s' % (H, Cin, Cout, str(e)[:60])); continue
    G, _ = ops.conv2d_wgrad(x, dz, Cin=Cin, Cout=Cout, KH=1, KW=1, image_splits=img)
    print('%3d^2 %3d->%-3d image_splits=%d kernel %d slabs %4d  %7.1f us  %5.2f TB/s' % (H, Cin, Cout, img, kid, G.shape[0], us, 4.0 * B * H * H * (Cin + Cout) / us / 1e6), flush=True)
