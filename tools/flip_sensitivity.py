"""Is a gradient-norm gate of 1e-3 meaningful for EVERY parameter tensor?  (blocks.3._se_reduce of d1_128_train.)

The oracle (pinned torch-CPU fp32 restatement of the reference) is run on the golden's inputs with the IMAGE scaled by
(1 + eps), eps ~ 1e-7: every smooth quantity moves by ~1e-7, so any parameter whose gradient norm moves by >> 1e-6 does so
through a DISCONTINUITY of the network (a head ReLU whose pre-activation is ~1e-7 of scale, a BiFPN 2x2 max-pool tie, the
smooth-L1 switch at |d| = 1/9) falling the other way.  Prints, per perturbation, the tensors whose norm moved by > 1e-5 relative to
the unperturbed run, and writes the per-tensor MAXIMUM over the perturbations to tests/golden/<case>_flipsens.npz -- the measured
instability of the REFERENCE arithmetic itself, which tests/test_gpu_model.py adds to the 1e-3 gate of exactly those tensors.
    python tools/flip_sensitivity.py d1_128_train [n=12] [--write] [--eps=3e-7] [--noise]
--eps: the perturbation bound; --noise: an independent factor (1 + eps u), u ~ U(-1, 1), per image ELEMENT instead of one scale for
the image (what an arithmetic with a per-operand rounding of eps does to the first layer); neither is written to the golden."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O      # noqa: E402


def norms(g, scale):
    net, nc = str(g['network']), int(g['num_classes'])
    dead = set(str(x) for x in g['dead_params'])
    sd = O.golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k and k not in dead}
    live = dict(sd); live.update(params)
    img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
    cl, rl = O.train_losses(live, net, nc, img * scale, torch.from_numpy(g['annots']))       # scale: a float or a tensor like img
    (cl.mean() + rl.mean()).backward()
    return {k: float(p.grad.double().norm()) for k, p in params.items()}


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'd1_128_train'
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 12
    g = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'), allow_pickle=False)
    torch.set_num_threads(16)
    base = norms(g, 1.0)
    worst = {k: 0.0 for k in base}
    rng = np.random.RandomState(0)
    bound = ([float(a[6:]) for a in sys.argv if a.startswith('--eps=')] or [3e-7])[0]
    noise = '--noise' in sys.argv
    assert not ('--write' in sys.argv and (noise or bound != 3e-7)), 'the golden holds the 3e-7 scaling only'
    for i in range(n):
        eps = float(rng.uniform(-bound, bound))
        if noise:
            B, S = int(g['B']), int(g['S'])
            gen = torch.Generator().manual_seed(100 + i)
            cur = norms(g, 1.0 + bound * (2.0 * torch.rand(B, 3, S, S, generator=gen) - 1.0))
        else:
            cur = norms(g, 1.0 + eps)
        moved = sorted(((abs(cur[k] - base[k]) / max(base[k], 1e-300), k) for k in base), reverse=True)
        for r, k in moved:
            worst[k] = max(worst[k], r)
        print('%s image x (1 %+.1e): %3d tensors moved > 1e-5 | %s' % (case, eps, sum(r > 1e-5 for r, _ in moved),
              ' | '.join('%s %.1e' % (k.replace('backbone._blocks.', 'b'), r) for r, k in moved[:4])), flush=True)
    top = sorted(((v, k) for k, v in worst.items()), reverse=True)
    print('max over %d perturbations, worst 8: %s' % (n, ' | '.join('%s %.2e' % (k.replace('backbone._blocks.', 'b'), v) for v, k in top[:8])))
    if '--write' in sys.argv:
        p = os.path.join(ROOT, 'tests', 'golden', case + '_flipsens.npz')
        np.savez(p, names=np.array(list(worst.keys())), sens=np.array(list(worst.values()), dtype=np.float64), n=np.int64(n))
        print('wrote', p)


if __name__ == '__main__':
    main()
