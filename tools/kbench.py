"""Micro-benchmark of the MFMA conv kernels on the head-tower shape (5 pyramid levels, 256->256 3x3, batch 32).
Used under rocprofv3 (--kernel-trace / --pmc) to attribute time inside the dominant kernels."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import functional as Fn, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--which', default='fwd,wgrad')
ap.add_argument('--arith', default='f32', help="fp32 storage: 'f32' or 'bf16x3' products")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
ops.set_f32_arith(a.arith)
dev = 'cuda'
sizes = [(int(v), int(v)) for v in os.environ.get('KB_SIZES', '64,32,16,8,4').split(',')]
_, x = Fn.pyramid_alloc(a.B, sizes, a.cin, dt, dev)
_, y = Fn.pyramid_alloc(a.B, sizes, a.cout, dt, dev)
x[0].t.copy_(torch.randn(x[0].t.numel(), device=dev).to(dt)); y[0].t.copy_(torch.randn(y[0].t.numel(), device=dev).to(dt))
w = torch.randn(a.cout, a.cin, 3, 3, device=dev) * 0.02
b = torch.zeros(a.cout, device=dev)
wp = ops.pack_weight(w, dt)
M = sum(a.B * h * ww for h, ww in sizes)
flops = 2.0 * M * 9 * a.cin * a.cout


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


if 'fwd' in a.which:
    ms = timeit(lambda: ops.conv2d(x, wp, y, Cin=a.cin, Cout=a.cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b, act=ops.ACT_RELU))
    print('fwd   %.3f ms  %.1f TFLOP/s' % (ms, flops / ms / 1e9))
if 'wgrad' in a.which:
    G = torch.zeros(a.cout, 9, a.cin, device=dev); db = torch.zeros(a.cout, device=dev)
    ms = timeit(lambda: ops.conv2d_wgrad(x, y, G, db, Cin=a.cin, Cout=a.cout, KH=3, KW=3, pad_t=1, pad_l=1))
    print('wgrad %.3f ms  %.1f TFLOP/s' % (ms, flops / ms / 1e9))
