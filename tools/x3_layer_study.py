"""Which layers carry the bf16x3 forward error?  (VERDICT r4 #1a.)  CPU study on the ORACLE (test infrastructure; nothing here ships):
every dense conv that the GPU path would run with bf16 hi + lo operand splits (ops._mma_dtype_code: K % 32 == 0, K >= 256, Cout >= 32) is
emulated as conv(hi x, hi w) + conv(hi x, lo w) + conv(lo x, hi w) in fp32, per REGION of the network, and the outputs are compared with
the all-exact run under the parity rule of tests/gpu_util.assert_close (|a - b| / max(|b|, 1e-2 max|b|)).  Also counted: ReLU decisions of
the head towers that fall the other way (what moves gradients in the deep families).

    python tools/x3_layer_study.py [d0_512_eval | d4_1024_eval | d4_256_eval ...]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O     # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else 'd0_512_eval'
g = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'), allow_pickle=False)
net, nc = str(g['network']), int(g['num_classes'])
sd = O.golden_state_dict(g)
img, _ = O.synthetic_batch(1, int(g['S']), seed=1, num_classes=nc)
torch.set_num_threads(os.cpu_count() or 8)
name_of = {id(v): k for k, v in sd.items()}
real_conv = F.conv2d
ACTIVE = set()          # regions emulated in bf16x3
PRE = {}                # tower pre-activations (name -> tensor) of the current run


def region(name):
    if name.startswith('backbone.'):
        return 'backbone'
    if name.startswith('neck.'):
        return 'neck'
    if 'cls_convs' in name:
        return 'cls_tower'
    if 'reg_convs' in name:
        return 'reg_tower'
    if 'retina_cls' in name:
        return 'retina_cls'
    if 'retina_reg' in name:
        return 'retina_reg'
    return 'other'


def hi(t):
    return t.bfloat16().float()


def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    name = name_of.get(id(w), '?')
    K = w.shape[1] * w.shape[2] * w.shape[3]
    x3 = region(name) in ACTIVE and groups == 1 and K % 32 == 0 and K >= 256 and w.shape[0] >= 32
    if x3:
        xh, wh = hi(x), hi(w)
        xl, wl = hi(x - xh), hi(w - wh)
        y = real_conv(xh, wh, None, stride, padding, dilation, groups) + real_conv(xh, wl, None, stride, padding, dilation, groups) + \
            real_conv(xl, wh, None, stride, padding, dilation, groups)
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
    else:
        y = real_conv(x, w, b, stride, padding, dilation, groups)
    if 'cls_convs' in name or 'reg_convs' in name:
        PRE.setdefault(name, []).append(y)
    return y


O.F.conv2d = conv


def run(active):
    ACTIVE.clear(); ACTIVE.update(active); PRE.clear()
    with torch.no_grad():
        cls, reg, _, taps = O.forward_raw(sd, net, nc, img, taps=True)
    D = max(int(k.split('_')[0][5:]) for k in taps if k.startswith('bifpn'))
    neck = [taps[f'bifpn{D}_p{l}'] for l in range(5)]
    return cls, reg, neck, {k: [t.clone() for t in v] for k, v in PRE.items()}


def erel(a, b):
    floor = 1e-2 * float(b.abs().max())
    return float(((a - b).abs() / torch.clamp(b.abs(), min=floor)).max())


ref = run(())
print('%s  (%s, %d classes, 1 x %d^2); errors = max |a - b| / max(|b|, 1e-2 max|b|) against the all-exact run; gate 1e-3' % (case, net, nc, int(g['S'])))
print('%-46s %10s %10s %10s %14s' % ('bf16x3 in', 'cls', 'reg', 'neck taps', 'ReLU flips'))
ALL = ('backbone', 'neck', 'cls_tower', 'reg_tower', 'retina_cls', 'retina_reg')
for label, act in [('everything (the all-bf16x3 mode)', ALL),
                   ('backbone + neck only', ('backbone', 'neck')),
                   ('the whole head only', ('cls_tower', 'reg_tower', 'retina_cls', 'retina_reg')),
                   ('cls tower + retina_cls only', ('cls_tower', 'retina_cls')),
                   ('reg tower + retina_reg only', ('reg_tower', 'retina_reg')),
                   ('the two output convs only', ('retina_cls', 'retina_reg')),
                   ('retina_cls only', ('retina_cls',)),
                   ('everything but the reg tower + retina_reg', ('backbone', 'neck', 'cls_tower', 'retina_cls'))]:
    cls, reg, neck, pre = run(act)
    flips = tot = 0
    for k, lst in pre.items():
        for a, b in zip(lst, ref[3][k]):
            flips += int(((a > 0) != (b > 0)).sum()); tot += a.numel()
    print('%-46s %10.2e %10.2e %10.2e %8d / %.1e' % (label, erel(cls, ref[0]), erel(reg, ref[1]), max(erel(a, b) for a, b in zip(neck, ref[2])), flips, tot), flush=True)
