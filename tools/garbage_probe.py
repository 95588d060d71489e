"""Uninitialised-read detector, finite flavour: every torch.empty / empty_like / Map.new inside the package is filled with a
large finite constant (comparisons and ReLU masks ignore NaN, not 1e6) and one train step is compared with a clean one.
python tools/garbage_probe.py H W [only=<substring of the calling function>]"""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops
from efficientdet.pytorch_amd import functional as Fn
from oracle import effdet_oracle as O
H, W = int(sys.argv[1]), int(sys.argv[2])
ONLY = [a[5:] for a in sys.argv if a.startswith('only=')]
net, nc, B = 'efficientdet-d0', 12, 2
c = EFFICIENTDET[net]
sd = O.make_state_dict(net, nc, seed=4)
g = torch.Generator().manual_seed(H * 7 + W)
img = torch.randn(B, 3, H, W, generator=g).cuda()
ann = torch.full((B, 4, 5), -1.0)
ann[0, 0] = torch.tensor([10., 12., 90., 100., 3.]); ann[0, 1] = torch.tensor([W - 70., H - 64., W - 5., H - 9., 7.])
ann[1, 0] = torch.tensor([W / 2 - 30., 20., W / 2 + 34., 110., 0.])
ann = ann.cuda()
FILL = [False]
HITS = {}
_empty, _empty_like = torch.empty, torch.empty_like


def _caller():
    for fr in traceback.extract_stack()[-4::-1]:
        if 'pytorch_amd' in fr.filename:
            return '%s:%s' % (os.path.basename(fr.filename), fr.name)
    return '?'


def _maybe_fill(t):
    if FILL[0] and t.is_cuda and t.numel():
        who = _caller()
        if not ONLY or any(o in who for o in ONLY):
            HITS[who] = HITS.get(who, 0) + 1
            if t.dtype.is_floating_point:
                t.fill_(1.0e6)
            else:
                t.view(torch.uint8).fill_(0x55)
    return t


torch.empty = lambda *a, **k: _maybe_fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _maybe_fill(_empty_like(*a, **k))


def run():
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], compute_dtype=torch.float32)
    m.load_state_dict(sd); m.backbone.drop_connect_rate = 0.0
    m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, (float(cl.detach()), float(rl.detach()))


clean, lc = run()
FILL[0] = True
dirty, ld = run()
FILL[0] = False
print('losses clean', lc, 'garbage-filled', ld)
bad = []
for k in clean:
    a, b = clean[k].double(), dirty[k].double()
    d = float('inf') if not bool(torch.isfinite(b).all()) else float((a - b).norm()) / (float(a.norm()) + 1e-30)
    if d > 1e-3:
        bad.append((d, k))
bad.sort(reverse=True)
print('%d of %d gradient tensors off by > 1e-3 with garbage-filled torch.empty:' % (len(bad), len(clean)), [(k, '%.1e' % d) for d, k in bad[:8]])
print('filled allocation sites:', sorted(HITS.items(), key=lambda kv: -kv[1])[:40])
