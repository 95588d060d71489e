"""Depthwise data + weight gradient: the fused kernel (effdet_dwconv_bwd) against the two launches it replaces, D0 B = 32 @512 shapes of the
k = 3 blocks: 0 (s1, 32 ch, 256^2), 1 (s2, 96 ch, 256^2 -> 128^2), 2 (s1, 144 ch, 128^2), 5 (s2, 240 ch, 64^2 -> 32^2), 6/7 (s1, 480 ch, 32^2),
15 (s1, 1152 ch, 16^2)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402
from efficientdet.pytorch_amd.config import tf_same_pad  # noqa: E402

B = 32


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


for (k, s, C, H) in [(3, 1, 32, 256), (3, 2, 96, 256), (3, 1, 144, 128), (3, 2, 240, 64), (3, 1, 480, 32), (3, 1, 1152, 16),
                     (5, 2, 144, 128), (5, 1, 240, 64), (5, 1, 480, 32), (5, 1, 672, 32), (5, 2, 672, 32), (5, 1, 1152, 16)]:
    pad = tf_same_pad(H, k, s)
    Ho = (H + pad[0] + pad[1] - k) // s + 1
    ze = Map.of(torch.randn(B, H, H, C, device='cuda'))
    dzd = Map.of(torch.randn(B, Ho, Ho, C, device='cuda'))
    wk = torch.randn(k * k, C, device='cuda'); sc = torch.rand(C, device='cuda') + 0.5
    fw = lambda: ops.dwconv_wgrad(ze, dzd, k, s, pad[0], pad[0], in_act=ops.ACT_SWISH)
    fd = lambda: ops.dwconv_dgrad(dzd, wk, sc, ze, H, H, k, s, pad[0], pad[0])
    ff = lambda: ops.dwconv_bwd(dzd, wk, sc, ze, k, s, pad[0], pad[0])
    tw, td, ts, tf = timeit(fw), timeit(fd), timeit(lambda: (fw(), fd())), timeit(ff)
    gb = 4.0 * B * C * (Ho * Ho + 2 * H * H) / 1e9
    print('k%d s%d C%-4d %3d^2: wgrad %6.1f us  dgrad %6.1f us  both %6.1f  fused %6.1f us (%.2f TB/s of %.2f GB)' % (k, s, C, H, tw, td, ts, tf, gb / tf * 1e3, gb),
          flush=True)
