"""Micro-benchmark of the bf16x3 head kernels on the RetinaHead shapes (5 pyramid levels of a 512^2 input, batch 32): plain fp32
storage with register splits (EFFDET_F32_BF16X3) against the split activation layout (EFFDET_F32_SPLIT), forward conv, data
gradient (ReLU-mask residual) and weight gradient.  Operands mimic the train step: activations are ReLU outputs (half zeros),
gradients are masked the same way (the matrix pipe's clock depends on the data)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import functional as Fn, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--which', default='fwd,dgrad,wgrad')
ap.add_argument('--shapes', default='256:256,256:720,64:256')
a = ap.parse_args()
dev = 'cuda'
dt = torch.float32
ops.set_f32_arith('bf16x3')
sizes = [(int(v), int(v)) for v in os.environ.get('KB_SIZES', '64,32,16,8,4').split(',')]
M = sum(a.B * h * w for h, w in sizes)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


def pyr(C, relu, split):
    flat, maps = Fn.pyramid_alloc(a.B, sizes, C, dt, dev)
    v = torch.randn(flat.numel(), device=dev)
    if relu:
        v = torch.relu(v)
    flat.copy_(ops.to_split(v) if split else v)
    return maps


for shp in a.shapes.split(','):
    cin, cout = (int(v) for v in shp.split(':'))
    ldo = (cout + 63) // 64 * 64
    flops = 2.0 * M * 9 * cin * cout
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    for split in (False, True):
        tag = 'split' if split else 'plain'
        x = pyr(cin, True, split)
        if 'fwd' in a.which and cout % 32 == 0:
            y = pyr(cout, False, False)
            wp = ops.pack_weight(w, dt, x3=True)
            ms = timeit(lambda: ops.conv2d(x, wp, y, Cin=cin, Cout=cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b, act=ops.ACT_RELU, split=split))
            print('%-5s fwd   %d->%d  %.3f ms  %.1f TFLOP/s' % (tag, cin, cout, ms, flops / ms / 1e9))
        if 'dgrad' in a.which and cin % 32 == 0 and cout % 32 == 0:
            dz = pyr(cout, True, split)
            dx = pyr(cin, False, False)
            wd = ops.pack_weight(w, dt, mode=1, x3=True)
            ms = timeit(lambda: ops.conv2d(dz, wd, dx, Cin=cout, Cout=cin, KH=3, KW=3, pad_t=1, pad_l=1, res=x, res_mode=ops.RES_RELU_MASK, split=split))
            print('%-5s dgrad %d->%d  %.3f ms  %.1f TFLOP/s' % (tag, cout, cin, ms, flops / ms / 1e9))
        if 'wgrad' in a.which:
            dz = pyr(ldo, True, split)
            ms = timeit(lambda: ops.conv2d_wgrad(x, dz, Cin=cin, Cout=cout, KH=3, KW=3, pad_t=1, pad_l=1, split=split))
            print('%-5s wgrad %d->%d  %.3f ms  %.1f TFLOP/s' % (tag, cin, cout, ms, flops / ms / 1e9))
        del x
        torch.cuda.empty_cache()
