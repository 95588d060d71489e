"""tools/flip_sensitivity.py for the inputs of tests/test_gpu_model.py::test_non_square_input_vs_oracle (H x W = 128x256, 384x128):
the pinned fp32 oracle with the image scaled by (1 + eps), |eps| <= 1.2e-6 (60 draws); per parameter tensor the largest relative L2
movement of its GRADIENT, written to tests/golden/nonsquare_<H>x<W>_flipsens.npz.  (384x128: eps = -1.2e-6 moves the stem weight
gradient by 1.2e-2, _bn0 by 3.9e-3, block 0's _bn1 by 3.7e-3 -- digit for digit the deviation of the HIP path from the unperturbed
oracle: one discontinuity, crossed by the reference itself one part in a million away.)
    python tools/flip_sensitivity_nonsquare.py 384 128 [n=60]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O      # noqa: E402


def inputs(H, W, B=2):
    """The test's own inputs (kept in one place: the test imports this)."""
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g)
    ann = torch.full((B, 4, 5), -1.0)
    ann[0, 0] = torch.tensor([10., 12., 90., 100., 3.]); ann[0, 1] = torch.tensor([W - 70., H - 64., W - 5., H - 9., 7.])
    ann[1, 0] = torch.tensor([W / 2 - 30., 20., W / 2 + 34., 110., 0.])
    return img, ann


def main():
    H, W = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    net, nc = 'efficientdet-d0', 12
    sd = O.make_state_dict(net, nc, seed=4)
    img, ann = inputs(H, W)
    pn = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k and not k.startswith(('backbone._conv_head', 'backbone._bn1', 'backbone._fc'))]

    def grads(scale):
        ps = {k: sd[k].clone().requires_grad_(True) for k in pn}
        sd2 = dict(sd); sd2.update(ps)
        cl, rl = O.train_losses(sd2, net, nc, img * scale, ann)
        (cl.mean() + rl.mean()).backward()
        return {k: p.grad.double() for k, p in ps.items()}
    torch.set_num_threads(16)
    base = grads(1.0)
    worst = {k: 0.0 for k in base}
    rng = np.random.RandomState(1)
    for i in range(n):
        eps = float(rng.uniform(-1.2e-6, 1.2e-6))
        cur = grads(1.0 + eps)
        rows = sorted(((float((cur[k] - base[k]).norm()) / max(float(base[k].norm()), 1e-30), k) for k in base), reverse=True)
        for r, k in rows:
            worst[k] = max(worst[k], r)
        print('%dx%d %+.1e: %s' % (H, W, eps, ' | '.join('%s %.1e' % (k.replace('backbone._blocks.', 'b'), r) for r, k in rows[:3])), flush=True)
    p = os.path.join(ROOT, 'tests', 'golden', 'nonsquare_%dx%d_flipsens.npz' % (H, W))
    np.savez(p, names=np.array(list(worst.keys())), sens=np.array(list(worst.values()), dtype=np.float64), n=np.int64(n))
    top = sorted(((v, k) for k, v in worst.items()), reverse=True)[:6]
    print('wrote', p, '| worst:', ' | '.join('%s %.2e' % (k, v) for v, k in top))


if __name__ == '__main__':
    main()
