"""Would running the depthwise weight gradient and data gradient of an MBConv block TOGETHER pay?  Both read dz_d and z_e; launched on two
streams (eager) they can share those reads in L2 / the Infinity Cache and fill each other's idle CUs.  Shapes: D0 B = 32 @512 blocks
1 (k3 s2, 96 ch, 256^2 -> 128^2), 2 (k3 s1, 144 ch, 128^2), 4 (k5 s1, 240 ch, 64^2), 9 (k5 s1, 672 ch, 32^2), 13 (k5 s1, 1152 ch, 16^2)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402
from efficientdet.pytorch_amd.config import tf_same_pad  # noqa: E402

B = 32
side = torch.cuda.Stream()


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


for (k, s, C, H) in [(3, 2, 96, 256), (3, 1, 144, 128), (5, 1, 240, 64), (5, 1, 672, 32), (5, 1, 1152, 16)]:
    pad = tf_same_pad(H, k, s)
    Ho = (H + pad[0] + pad[1] - k) // s + 1
    ze = Map.of(torch.randn(B, H, H, C, device='cuda'))
    dzd = Map.of(torch.randn(B, Ho, Ho, C, device='cuda'))
    wk = torch.randn(k * k, C, device='cuda'); sc = torch.rand(C, device='cuda') + 0.5
    fw = lambda: ops.dwconv_wgrad(ze, dzd, k, s, pad[0], pad[0], in_act=ops.ACT_SWISH)
    fd = lambda: ops.dwconv_dgrad(dzd, wk, sc, ze, H, H, k, s, pad[0], pad[0])

    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fw()
        fd()
        main.wait_stream(side)
    tw, td, ts, tb = timeit(fw), timeit(fd), timeit(lambda: (fw(), fd())), timeit(both)
    print('k%d s%d C%-4d %3d^2: wgrad %6.1f us  dgrad %6.1f us  one stream %6.1f  two streams %6.1f' % (k, s, C, H, tw, td, ts, tb), flush=True)
