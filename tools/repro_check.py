"""Run the same forward many times and report run-to-run differences per stage (race / uninitialised-read hunt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET  # noqa: E402
from oracle import effdet_oracle as O  # noqa: E402

dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.bfloat16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
net, nc = 'efficientdet-d0', 20
c = EFFICIENTDET[net]
m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=False, compute_dtype=dt)
m.load_state_dict(O.make_state_dict(net, nc, seed=0)); m = m.cuda().eval()
img, _ = O.synthetic_batch(2, S, seed=1, num_classes=nc)
img = img.cuda()


def run():
    with torch.no_grad():
        feats = m._backbone(img)
        p = m._neck(feats[-5:])
    cls, reg, _ = m.forward_raw(img)
    return [f.float().clone() for f in feats] + [t.float().clone() for t in p] + [cls.clone(), reg.clone()]


ref = run()
names = ['stage%d' % i for i in range(7)] + ['neck%d' % i for i in range(5)] + ['cls', 'reg']
worst = {n: 0.0 for n in names}
bad = 0
for it in range(80):
    # churn the allocator so that buffers move around between runs
    junk = [torch.full((1 + (it * 7919) % 100000,), float('nan'), device='cuda') for _ in range(3)]
    out = run()
    del junk
    for n, a, b in zip(names, out, ref):
        d = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        if not (d == d):
            d = float('inf')
        worst[n] = max(worst[n], d)
        if d > 1e-4:
            bad += 1
            print('iter', it, n, 'rel diff', d)
print('worst run-to-run rel diff per stage:', {k: '%.2e' % v for k, v in worst.items()}, 'bad', bad)
