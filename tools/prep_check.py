"""Diagnostic: twin models, batched parameter prep vs single launches, per-step max gradient deviation."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import effdet_oracle as O
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET

def build(batched):
    c = EFFICIENTDET['efficientdet-d0']
    m = EfficientDet(20, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], compute_dtype=torch.float32)
    m.load_state_dict(O.make_state_dict('efficientdet-d0', 20, seed=1)); m.backbone.drop_connect_rate = 0.0
    m.batched_prep = batched
    m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
    return m

img, ann = O.synthetic_batch(3, 128, seed=9, num_classes=20); img, ann = img.cuda(), ann.cuda()
for pair in ((False, False), (True, False)):
    ms = [build(b) for b in pair]
    opts = [torch.optim.SGD(m.live_parameters(), lr=1e-3) for m in ms]
    names = [n for n, p in ms[0].named_parameters()]
    for it in range(3):
        gs = []
        for m, o in zip(ms, opts):
            o.zero_grad(); cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward()
            gs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}); o.step()
        worst = max(((float((gs[0][n] - gs[1][n]).abs().max()) / (float(gs[1][n].abs().max()) + 1e-30), n) for n in gs[0]))
        print(pair, 'step', it, 'worst rel dev %.3g at %s' % worst)
