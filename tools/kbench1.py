"""Micro-benchmark of one conv shape (single level): fwd (BN+swish epilogue, y and z written), dgrad, wgrad."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32); ap.add_argument('--H', type=int, default=128)
ap.add_argument('--cin', type=int, default=24); ap.add_argument('--cout', type=int, default=144)
ap.add_argument('--k', type=int, default=1); ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--z', type=int, default=1)
a = ap.parse_args()
dt, dev = torch.bfloat16, 'cuda'
x = Map.new(a.B, a.H, a.H, a.cin, dt, dev); x.t.normal_()
y = Map.new(a.B, a.H, a.H, a.cout, dt, dev); z = Map.new(a.B, a.H, a.H, a.cout, dt, dev) if a.z else None
w = torch.randn(a.cout, a.cin, a.k, a.k, device=dev) * 0.05
sc = torch.ones(a.cout, device=dev); sh = torch.zeros(a.cout, device=dev)
wp = ops.pack_weight(w, dt); wd = ops.pack_weight(w, dt, mode=1, scale=sc)
M = a.B * a.H * a.H
pad = a.k // 2


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


bx, by = M * a.cin * 2, M * a.cout * 2
ms = timeit(lambda: ops.conv2d(x, wp, y, Cin=a.cin, Cout=a.cout, KH=a.k, KW=a.k, pad_t=pad, pad_l=pad, scale=sc, shift=sh, act=ops.ACT_SWISH, zs=z))
print('fwd   %.3f ms  %.0f GB/s (algorithmic %d MB)' % (ms, (bx + by * (2 if a.z else 1)) / ms / 1e6, (bx + by * (2 if a.z else 1)) / 1e6))
ms = timeit(lambda: ops.conv2d(y, wd, x, Cin=a.cout, Cout=a.cin, KH=a.k, KW=a.k, pad_t=pad, pad_l=pad))
print('dgrad %.3f ms  %.0f GB/s' % (ms, (bx + by) / ms / 1e6))
db = torch.zeros(a.cout, device=dev)
ms = timeit(lambda: ops.conv2d_wgrad(x, y, None, db, Cin=a.cin, Cout=a.cout, KH=a.k, KW=a.k, pad_t=pad, pad_l=pad))
print('wgrad %.3f ms  %.0f GB/s' % (ms, (bx + by) / ms / 1e6))
