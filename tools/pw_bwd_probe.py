"""Expand-conv backward: the fused kernel (effdet_pw_bwd) against the two launches it replaces, D0 B = 32 @512 blocks 1 (16 -> 96, 256^2),
2 / 3 (24 -> 144, 128^2)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

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


for (Ci, H, skip) in [(16, 256, False), (24, 128, True), (24, 128, False), (32, 128, True)]:
    Ce = 6 * Ci
    dz = Map.of(torch.randn(B, H, H, Ce, device='cuda')); x = Map.of(torch.randn(B, H, H, Ci, device='cuda'))
    res = Map.of(torch.randn(B, H, H, Ci, device='cuda')) if skip else None
    we = torch.randn(Ce, Ci, 1, 1, device='cuda'); s0 = torch.rand(Ce, device='cuda') + 0.5
    wp = ops.pack_weight(we, torch.float32, mode=1, scale=s0)
    dxs = Map.new(B, H, H, Ci, torch.float32, 'cuda')
    fw = lambda: ops.conv2d_wgrad(x, dz, Cin=Ci, Cout=Ce, KH=1, KW=1)
    fd = lambda: ops.conv2d(dz, wp, dxs, Cin=Ce, Cout=Ci, KH=1, KW=1, res=res, res_mode=ops.RES_ADD if skip else ops.RES_NONE)
    ff = lambda: ops.pw_bwd(dz, x, we, s0, res)
    tw, td, tf = timeit(fw), timeit(fd), timeit(ff)
    gb = 4.0 * B * H * H * (Ce + (3 if skip else 2) * Ci) / 1e9
    fl = 4.0 * B * H * H * Ce * Ci / 1e12
    print('Cin%d -> %d %3d^2 skip=%d: wgrad %6.1f us  dgrad %6.1f us  fused %6.1f us (%.2f TB/s of %.2f GB, %.1f TFLOP/s)' % (
        Ci, Ce, H, skip, tw, td, tf, gb / tf * 1e3, gb, fl / tf * 1e6), flush=True)
