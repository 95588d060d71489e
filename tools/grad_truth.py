"""Which side of a gradient-norm comparison is further from the truth?  

For a train golden (tests/golden/<case>.npz: losses + per-parameter gradient norms recorded from the REAL reference in fp32) this
runs the oracle -- the pinned torch-CPU restatement of the reference -- in fp64 on the same inputs and prints, per parameter
tensor, the relative deviation of the reference's fp32 norm from the fp64 norm.  A tensor whose REFERENCE value is already
1e-3 away from the fp64 truth cannot be held to 1e-3 against that reference by any implementation; tests/test_gpu_model.py
uses the committed table (tests/golden/<case>_f64norms.npz) to gate the HIP path against the fp64 truth for exactly those tensors.

    python tools/grad_truth.py d1_128_train [--write]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O      # noqa: E402


def f64_norms(case):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'), allow_pickle=False)
    net, nc = str(g['network']), int(g['num_classes'])
    sd = O.golden_state_dict(g)
    dead = set(str(x) for x in g['dead_params'])
    params = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k and k not in dead}
    live = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    live.update(params)
    img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
    ann = torch.from_numpy(g['annots'])
    cl, rl = O.train_losses(live, net, nc, img.double(), ann.double())
    (cl.mean() + rl.mean()).backward()
    out = {k: float(p.grad.norm()) for k, p in params.items()}
    return g, out, float(cl), float(rl)


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'd1_128_train'
    g, n64, cl, rl = f64_norms(case)
    print('%s: fp64 oracle losses %.9g %.9g | reference fp32 %.9g %.9g' % (case, cl, rl, float(g['cls_loss'][0]), float(g['reg_loss'][0])))
    gmax = max(n64.values())
    rows = []
    for k, t in n64.items():
        ref = float(g['grad_' + k + '_summary'][2])
        rows.append((abs(ref - t) / max(t, 1e-300), k, ref, t))
    rows.sort(reverse=True)
    print('largest norm %.6g; reference-fp32 vs fp64-oracle gradient norms, worst 12 of %d:' % (gmax, len(rows)))
    for r, k, ref, t in rows[:12]:
        print('  %.3e  %-60s ref %.6e  f64 %.6e  (%.1e of the largest norm)' % (r, k, ref, t, t / gmax))
    print('tensors with reference deviation > 1e-4: %d, > 1e-3: %d' % (sum(r[0] > 1e-4 for r in rows), sum(r[0] > 1e-3 for r in rows)))
    if '--write' in sys.argv:
        path = os.path.join(ROOT, 'tests', 'golden', case + '_f64norms.npz')
        np.savez(path, names=np.array(list(n64.keys())), norms=np.array(list(n64.values()), dtype=np.float64),
                 cls_loss=np.float64(cl), reg_loss=np.float64(rl))
        print('wrote', path)


if __name__ == '__main__':
    main()
