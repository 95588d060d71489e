"""GPU: per-parameter gradient-norm deviations of the HIP path from the real reference's goldens AND from the fp64 oracle
(tests/golden/<case>_f64norms.npz, tools/grad_truth.py), both parity modes, fused / unfused squeeze-excite backward.
    python tools/grad_outliers.py [case ...]      -> gpurun_out/grad_outliers.txt, gpurun_out/grad_deviations.npz (every tensor)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O                                   # noqa: E402  (checker only)
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET          # noqa: E402
from efficientdet.pytorch_amd import functional as Fn                    # noqa: E402

cases = sys.argv[1:] or ['d0_128_train', 'd1_128_train', 'd0_512_train']
lines = []
dump = {}       # case|arith -> (names, relative gradient-norm deviations from the reference golden)


def say(s):
    print(s); lines.append(s)


for case in cases:
    g = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'), allow_pickle=False)
    f64p = os.path.join(ROOT, 'tests', 'golden', case + '_f64norms.npz')
    t64 = None
    if os.path.exists(f64p):
        z = np.load(f64p, allow_pickle=False); t64 = dict(zip([str(n) for n in z['names']], z['norms']))
    net, nc = str(g['network']), int(g['num_classes']); c = EFFICIENTDET[net]
    gmax = max(float(g[k][2]) for k in g.files if k.startswith('grad_') and k.endswith('_summary'))
    for arith in ('f32', 'bf16x3'):
        for fused in (True, False):
            Fn.SE_FUSED = fused
            m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32, f32_arith=arith)
            m.load_state_dict(O.golden_state_dict(g)); m.backbone.drop_connect_rate = 0.0
            m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
            img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
            cl, rl = m([img.cuda(), torch.from_numpy(g['annots']).cuda()]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
            rows = []
            for k, p in m.named_parameters():
                if p.grad is None:
                    continue
                ref = float(g['grad_' + k + '_summary'][2]); l2 = float(p.grad.double().norm())
                r64 = abs(l2 - t64[k]) / max(t64[k], 1e-300) if t64 and k in t64 else float('nan')
                rows.append((abs(l2 - ref) / max(ref, 1e-12), k, ref, r64))
            if fused:
                dump['%s|%s' % (case, arith)] = (np.array([r[1] for r in rows]), np.array([r[0] for r in rows]))
            rows.sort(reverse=True)
            say('%s %s se_fused=%d: tensors > 1e-3: %d, > 3e-4: %d, > 1e-4: %d' % (case, arith, fused, sum(r[0] > 1e-3 for r in rows),
                                                                             sum(r[0] > 3e-4 for r in rows), sum(r[0] > 1e-4 for r in rows)))
            for r, k, n, r64 in rows[:5]:
                say('    %-58s vs reference %.2e  vs fp64 oracle %.2e  (norm %.2e = %.1e of the largest)' % (k, r, r64, n, n / gmax))
Fn.SE_FUSED = True
d = os.path.join(ROOT, 'gpurun_out')
if os.path.isdir(d):
    open(os.path.join(d, 'grad_outliers.txt'), 'w').write('\n'.join(lines) + '\n')
    np.savez(os.path.join(d, 'grad_deviations.npz'), **{k + '|names': v[0] for k, v in dump.items()}, **{k + '|rel': v[1] for k, v in dump.items()})
