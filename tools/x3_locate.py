"""GPU: WHERE does the bf16x3 arithmetic leave the exact-fp32 one?  Both HIP modes on a train golden's inputs; per parameter tensor the
relative VECTOR difference of the gradients, printed in backward order (head -> BiFPN stacks, last first -> backbone blocks, last
first), plus the forward outputs and neck features.
    python tools/x3_locate.py d3_128_train [d2_128_train ...]"""
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O                                   # noqa: E402  (inputs / weights only)
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET          # noqa: E402


def run(g, arith):
    net, nc = str(g['network']), int(g['num_classes']); c = EFFICIENTDET[net]
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32, f32_arith=arith)
    m.load_state_dict(O.golden_state_dict(g)); m.backbone.drop_connect_rate = 0.0
    m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
    img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
    img = img.cuda()
    with torch.no_grad():
        cls, reg, _ = m.forward_raw(img)
        feats = [f.clone() for f in m.extract_feat(img)]
    cl, rl = m([img, torch.from_numpy(g['annots']).cuda()]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
    return dict(cls=cls, reg=reg, feats=feats, loss=(float(cl), float(rl)), grads={k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None})


def order(k):
    if k.startswith('bbox_head'):
        return (0, 0, k)
    mm = re.match(r'neck\.stack_bifpn_convs\.(\d+)\.', k)
    if mm:
        return (1, -int(mm.group(1)), k)
    if k.startswith('neck'):
        return (2, 0, k)
    mm = re.match(r'backbone\._blocks\.(\d+)\.', k)
    if mm:
        return (3, -int(mm.group(1)), k)
    return (4, 0, k)


for case in sys.argv[1:] or ['d3_128_train']:
    g = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'), allow_pickle=False)
    a, b = run(g, 'f32'), run(g, 'bf16x3')
    rel = lambda x, y: float((x.double() - y.double()).norm() / max(float(y.double().norm()), 1e-300))
    print('%s: losses f32 %s bf16x3 %s | forward cls %.2e reg %.2e | neck features %s' % (
        case, a['loss'], b['loss'], rel(b['cls'], a['cls']), rel(b['reg'], a['reg']), ' '.join('%.1e' % rel(y, x) for x, y in zip(a['feats'], b['feats']))))
    groups = {}
    for k in sorted(a['grads'], key=order):
        o = order(k)
        name = 'head' if o[0] == 0 else 'bifpn stack %d' % -o[1] if o[0] == 1 else 'neck other' if o[0] == 2 else 'block %d' % -o[1] if o[0] == 3 else 'stem'
        groups.setdefault(name, []).append((rel(b['grads'][k], a['grads'][k]), k))
    for name, rows in groups.items():
        rows.sort(reverse=True)
        print('  %-16s median %.1e  worst %.1e %s' % (name, sorted(r for r, _ in rows)[len(rows) // 2], rows[0][0], rows[0][1]))
        if os.environ.get('X3_DETAIL') and re.search(os.environ['X3_DETAIL'], name):
            for r, k in sorted(rows, key=lambda t: t[1]):
                d = (b['grads'][k] - a['grads'][k])
                # a ReLU mask falling the other way at ONE (pixel, channel) changes ONE output-channel row of a weight gradient /
                # one entry of a bias gradient: the share of the squared difference held by the largest row says flip vs noise
                rows2 = (d.reshape(d.shape[0], -1) ** 2).sum(1) if d.dim() >= 1 and d.shape[0] > 1 else d.reshape(1, -1).pow(2).sum(1)
                share = float(rows2.max() / max(float(rows2.sum()), 1e-300))
                print('      %-64s %.1e  (norm %.2e; largest output-channel row holds %.3f of the squared difference, row %d of %d)' % (
                    k, r, float(a['grads'][k].norm()), share, int(rows2.argmax()), rows2.numel()))
