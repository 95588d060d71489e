"""Fused expand -> depthwise forward (inference) against the expand conv + depthwise pair, on the MBConv shapes of D0 at B = 32 @512.
    python tools/fused_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops                      # noqa: E402
from efficientdet.pytorch_amd.ops import Map                  # noqa: E402

dev, B = 'cuda', 32


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


shapes = [(16, 256, 3, 2, (0, 1)), (24, 128, 3, 1, (1, 1)), (24, 128, 5, 2, (1, 2)), (40, 64, 5, 1, (2, 2)), (40, 64, 3, 2, (0, 1))]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (Cin, H, k, s, (plo, phi)) in shapes:
    Cexp = 6 * Cin
    x = Map.of(torch.randn(B, H, H, Cin, device=dev))
    we = torch.randn(Cexp, Cin, 1, 1, device=dev) / Cin ** 0.5
    s0 = torch.rand(Cexp, device=dev) + 0.5; t0 = torch.randn(Cexp, device=dev) * 0.1
    wd = torch.randn(Cexp, 1, k, k, device=dev) * 0.3
    wk = ops.dw_pack_weight(wd)
    s1 = torch.rand(Cexp, device=dev) + 0.5; t1 = torch.randn(Cexp, device=dev) * 0.1
    Ho = (H + plo + phi - k) // s + 1
    wp = ops.pack_weight(we, torch.float32)
    xe = Map.new(B, H, H, Cexp, torch.float32, dev)

    def pair():
        ops.conv2d(x, wp, xe, Cin=Cin, Cout=Cexp, KH=1, KW=1, scale=s0, shift=t0, act=ops.ACT_SWISH)
        return ops.dwconv_fwd(xe, wk, s1, t1, k, s, plo, plo, Ho, Ho, save_z=False, pool=True)
    tp = timeit(pair)
    tf = timeit(lambda: ops.expand_dw_fwd(x, we, s0, t0, wk, s1, t1, k, s, plo, plo, Ho, Ho))
    by = 4 * B * (H * H * Cin * ((Cexp + 31) // 32) + Ho * Ho * Cexp)
    print('Cin %2d %3d^2 k%d s%d: expand + depthwise %7.1f us | fused %7.1f us (%.2f TB/s of its own bytes, %.0f MB)' % (Cin, H, k, s, tp, tf, by / tf / 1e6, by / 1e6))
