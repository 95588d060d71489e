"""A/B micro-benchmark of the implicit-GEMM variants on the RetinaHead shapes (5 pyramid levels, batch 32 @512):
    python tools/kbench2.py [--reps 20]
prints TFLOP/s per (shape, variant); variant 0 = the 128x128 / 16x16x32 kernel, 44|42|24|22 = big-tile 32x32x16 shapes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import functional as Fn, ops, _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--variants', default='0,442,242')
ap.add_argument('--kord', type=int, default=0)
ap.add_argument('--only', type=int, default=-1, help='run only shape #i')
ap.add_argument('--f32', default='', help="fp32 storage: 'f32' or 'bf16x3' arithmetic")
ap.add_argument('--zeros', action='store_true', help='zero-filled operands (DVFS / data-toggling probe)')
a = ap.parse_args()
dt, dev = (torch.float32 if a.f32 else torch.bfloat16), 'cuda'
if a.f32:
    ops.set_f32_arith(a.f32)
ops.tuning_set(3, a.kord)
sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
M = sum(a.B * h * w for h, w in sizes)


def bench(cin, cout, act, res, out_f32=False):
    _, x = Fn.pyramid_alloc(a.B, sizes, cin, dt, dev)
    x[0].t.copy_((torch.zeros if a.zeros else torch.randn)(x[0].t.numel(), device=dev).to(dt))
    if out_f32:
        buf = torch.empty((a.B, M // a.B * 9, cout // 9), dtype=torch.float32, device=dev)
        y = Fn.head_out_maps(buf, a.B, sizes, cout // 9)
    else:
        _, y = Fn.pyramid_alloc(a.B, sizes, cout, dt, dev)
    r = None
    if res:
        _, r = Fn.pyramid_alloc(a.B, sizes, cout, dt, dev)
        r[0].t.copy_(torch.randn(r[0].t.numel(), device=dev).to(dt))
    w = (torch.zeros if a.zeros else torch.randn)(cout, cin, 3, 3, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    wp = ops.pack_weight(w, dt)
    flops = 2.0 * M * 9 * cin * cout
    out = {}
    for v in [int(t) for t in a.variants.split(',')]:
        ops.tuning_set(L.TUNE_IGEMM_BIG, v)
        fn = lambda: ops.conv2d(x, wp, y, Cin=cin, Cout=cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b, act=act, res=r,
                                res_mode=ops.RES_RELU_MASK if res else ops.RES_NONE, out_f32=out_f32)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        out[v] = flops / ms / 1e9
    return out


SHAPES = [('tower fwd 256->256 relu', (256, 256, ops.ACT_RELU, False)), ('tower dgrad 256->256 relu-mask', (256, 256, ops.ACT_NONE, True)),
                   ('tower0 fwd 64->256 relu', (64, 256, ops.ACT_RELU, False)), ('retina_cls fwd 256->720 sigmoid f32', (256, 720, ops.ACT_SIGMOID, False, True)),
                   ('dcls dgrad 768->256 relu-mask', (768, 256, ops.ACT_NONE, True)), ('tower0 dgrad 256->64', (256, 64, ops.ACT_NONE, False))]
for i, (name, args) in enumerate(SHAPES):
    if a.only >= 0 and i != a.only:
        continue
    r = bench(*args)
    print('%-38s ' % name + '  '.join('v%-2d %7.1f' % (k, v) for k, v in r.items()) + '  TFLOP/s', flush=True)

