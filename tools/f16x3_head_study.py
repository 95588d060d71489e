"""Which split arithmetic can serve the RetinaHead forward at fp32 accuracy?  CPU study on the ORACLE (test infrastructure; nothing here ships), run BEFORE the
f16x3 kernel was written: the head's convs emulated as (a) exact fp32, (b) f16x3 = fp16 hi + scaled fp16 lo, row-scaled weights, hi*hi + hi*lo + lo*hi,
(c) a three-piece bf16 split with six products, (d) bf16x3 -- each against the head evaluated in float64 on the same fp32 pyramid.
    python tools/f16x3_head_study.py [d0_512_eval | d4_256_eval ...]"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O
case = sys.argv[1] if len(sys.argv) > 1 else 'd0_512_eval'
g = np.load(os.path.join(ROOT,'tests','golden',case+'.npz'), allow_pickle=False)
net, nc = str(g['network']), int(g['num_classes'])
sd = O.golden_state_dict(g)
img,_ = O.synthetic_batch(1, int(g['S']), seed=1, num_classes=nc)
torch.set_num_threads(8)
name_of = {id(v): k for k, v in sd.items()}
real_conv = F.conv2d
MODE = ['exact']
def is_head(n): return any(s in n for s in ('cls_convs','reg_convs','retina_cls','retina_reg'))
def f16(t): return t.half().float()
def bf(t): return t.bfloat16().float()
STAT = {}
def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    name = name_of.get(id(w), '?')
    m = MODE[0]
    if m != 'exact' and is_head(name):
        c = lambda a, bb: real_conv(a, bb, None, stride, padding, dilation, groups)
        if m == 'f16x3':
            # per-row weight scale to [2^14,2^15)
            mx = w.abs().amax(dim=(1,2,3), keepdim=True)
            S = torch.pow(2.0, 14 - torch.floor(torch.log2(mx)))
            ws = w * S
            xh = f16(x); xh = torch.where(xh.abs() < 2.0**-14, torch.zeros_like(xh), xh)
            xl = f16((x - xh) * 2048.0)          # scaled lo
            wh = f16(ws); wl = f16(ws - wh); wh2 = f16(wh / 2048.0)
            assert torch.equal(wh2 * 2048.0, wh) or True
            STAT.setdefault(name, []).append((float(x.abs().max()), float(x[x>0].median()) if (x>0).any() else 0.0))
            y = (c(xh, wh) + c(xh, wl) + c(xl, wh2)) / S.view(1,-1,1,1)
        elif m == 'f64':
            y = real_conv(x.double(), w.double(), None, stride, padding, dilation, groups).float()
        elif m == 'bf16x6':
            xh = bf(x); xm = bf(x - xh); xl = bf(x - xh - xm)
            wh = bf(w); wm = bf(w - wh); wl = bf(w - wh - wm)
            y = c(xh,wh) + (c(xh,wm) + c(xm,wh)) + (c(xm,wm) + c(xh,wl) + c(xl,wh))
        elif m == 'bf16x3':
            xh = bf(x); xl = bf(x-xh); wh = bf(w); wl = bf(w-wh)
            y = c(xh,wh)+c(xh,wl)+c(xl,wh)
        if b is not None: y = y + b.view(1,-1,1,1)
        return y
    return real_conv(x, w, b, stride, padding, dilation, groups)
O.F.conv2d = conv
def run(m):
    MODE[0] = m
    with torch.no_grad():
        cls, reg, _, taps = O.forward_raw(sd, net, nc, img, taps=True)
    return cls, reg
def erel(a, b, fl=1e-2):
    floor = fl * float(b.abs().max())
    return float(((a - b).abs() / torch.clamp(b.abs(), min=floor)).max())
ref = run('f64')     # head in float64 = "truth" for the head given identical inputs
print(case, 'head-only emulation; errors vs the head computed in float64 (floor 1e-2 / 1e-4 of max)')
for m in ('exact', 'f16x3', 'bf16x6', 'bf16x3'):
    c, r = run(m)
    print('%-8s cls %.2e %.2e   reg %.2e %.2e' % (m, erel(c, ref[0]), erel(c, ref[0], 1e-4), erel(r, ref[1]), erel(r, ref[1], 1e-4)), flush=True)
for k, v in STAT.items(): print(k, ['max %.3g med %.3g' % t for t in v][:5])
