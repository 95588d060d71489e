"""Micro-benchmark of the RetinaHead FORWARD convs (5 pyramid levels of a 512^2 input, batch 32) in the three arithmetics that can serve
them: exact fp32, the bf16x3 split layout, and the f16x3 H-split form (with / without the bf16 split copy the training forward leaves
for the gradient kernels).  Operands mimic the step: ReLU outputs (half zeros)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import functional as Fn, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--shapes', default='64:256,256:256,256:720,256:36')
ap.add_argument('--modes', default='f32,split,h,h+ys')
a = ap.parse_args()
dev, dt = 'cuda', torch.float32
sizes = [(int(v), int(v)) for v in os.environ.get('KB_SIZES', '64,32,16,8,4').split(',')]
M = sum(a.B * h * w for h, w in sizes)
L, C = ops.L, ops.C


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


def pyr(Cc, fill=None, kind='plain'):
    flat, maps = Fn.pyramid_alloc(a.B, sizes, Cc, dt, dev)
    if fill is not None:
        v = torch.relu(torch.randn(flat.numel(), device=dev))
        if kind == 'split':
            v = ops.to_split(v)
        elif kind == 'h':
            o = torch.empty_like(v)
            L.check(L.lib().effdet_to_split2(L.ptr(v), None, L.ptr(o), C.c_longlong(v.numel()), None, L.stream_ptr()), 'to_split2')
            v = o
        flat.copy_(v)
    return maps


for shp in a.shapes.split(','):
    cin, cout = (int(v) for v in shp.split(':'))
    final = cout % 32 != 0 or cout == 720
    flops = 2.0 * M * 9 * cin * cout
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    if final:
        out = torch.empty((a.B, M // a.B * 9, cout // 9), device=dev)
        y = Fn.head_out_maps(out, a.B, sizes, cout // 9)
    else:
        y = pyr(cout)
    ys = None if final else pyr(cout)
    kw = dict(Cin=cin, Cout=cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b, act=ops.ACT_SIGMOID if cout == 720 else (ops.ACT_NONE if final else ops.ACT_RELU),
              out_f32=final)
    for mode in a.modes.split(','):
        if mode == 'f32':
            ops.set_f32_arith('f32')
            x = pyr(cin, 1)
            wp = ops.pack_weight(w, dt)
            fn = lambda: ops.conv2d(x, wp, y, ysplit=ys, **kw)
        elif mode == 'split':
            ops.set_f32_arith('bf16x3')
            x = pyr(cin, 1, 'split')
            wp = ops.pack_weight(w, dt, x3=True)
            fn = lambda: ops.conv2d(x, wp, y, split=True, **kw)
        else:
            ops.set_f32_arith('f32')
            x = pyr(cin, 1, 'h')
            wp = ops.pack_weight(w, dt, h3=True)
            yy = ys if mode == 'h+ys' else None
            if final and mode == 'h+ys':
                continue
            fn = lambda: ops.conv2d(x, wp, y, hsplit=True, ysplit=yy, **kw)
        ms = timeit(fn)
        print('%-6s fwd %d->%d  %.3f ms  %.1f TFLOP/s' % (mode, cin, cout, ms, flops / ms / 1e9), flush=True)
        del x
    torch.cuda.empty_cache()
