import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import effdet_oracle as O
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
case = sys.argv[1]
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', case + '.npz'), allow_pickle=False)
net, nc = str(g['network']), int(g['num_classes']); c = EFFICIENTDET[net]
gmax = max(float(g[k][2]) for k in g.files if k.startswith('grad_') and k.endswith('_summary'))
for arith in ('f32', 'bf16x3'):
    for rep in range(3):
        m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32, f32_arith=arith)
        m.load_state_dict(O.make_state_dict(net, nc, seed=int(g['seed']))); m.backbone.drop_connect_rate = 0.0
        m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
        img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
        cl, rl = m([img.cuda(), torch.from_numpy(g['annots']).cuda()]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
        rows = []
        for k, p in m.named_parameters():
            if p.grad is None: continue
            ref = g['grad_' + k + '_summary']; l2 = float(p.grad.double().norm())
            rows.append((abs(l2 - ref[2]) / max(ref[2], 1e-12), k, ref[2]))
        rows.sort(reverse=True)
        print(case, arith, rep, ' | '.join('%s %.2e (norm %.2e = %.1e of max)' % (k.replace('backbone._blocks.', 'b'), r, n, n / gmax) for r, k, n in rows[:3]))
