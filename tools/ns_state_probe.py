"""Is the HIP path's result a function of its inputs only?  D0 train pass at 384x128 (the case whose stem gradient sits 1.2e-2 from
the CPU oracle's in tests/test_gpu_model.py::test_non_square_input_vs_oracle) in a fresh process, again, after a pass at another
shape, after 200 MB of allocator churn, and on a second model instance: all gradients must be BITWISE equal to the first pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET      # noqa: E402
from oracle import effdet_oracle as O                                 # noqa: E402 (weights + nothing else)

net, nc, B = 'efficientdet-d0', 12, 2
c = EFFICIENTDET[net]
sd = O.make_state_dict(net, nc, seed=4)


def batch(H, W):
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g)
    ann = torch.full((B, 4, 5), -1.0)
    ann[0, 0] = torch.tensor([10., 12., 90., 100., 3.]); ann[0, 1] = torch.tensor([W - 70., H - 64., W - 5., H - 9., 7.])
    ann[1, 0] = torch.tensor([W / 2 - 30., 20., W / 2 + 34., 110., 0.])
    return img.cuda(), ann.cuda()


def model():
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32)
    m.load_state_dict(sd); m.backbone.drop_connect_rate = 0.0
    m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
    return m


def grads(m, H, W):
    for p in m.parameters():
        p.grad = None
    img, ann = batch(H, W)
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, float(cl), float(rl)


def cmp(tag, a, b):
    bad = [k for k in a[0] if not torch.equal(a[0][k], b[0][k])]
    worst = max((float((a[0][k] - b[0][k]).double().norm() / a[0][k].double().norm().clamp_min(1e-30)), k) for k in a[0])
    print('%-46s losses equal %s, %3d / %d gradient tensors differ, worst rel %.2e (%s)' % (tag, (a[1], a[2]) == (b[1], b[2]), len(bad), len(a[0]), worst[0], worst[1]))


m = model()
first = grads(m, 384, 128)
cmp('same model, second pass:', first, grads(m, 384, 128))
grads(m, 128, 256)
cmp('after a pass at 128x256:', first, grads(m, 384, 128))
junk = [torch.randn(50_000_000, device='cuda') for _ in range(4)]; del junk
cmp('after allocator churn:', first, grads(m, 384, 128))
cmp('a second model instance:', first, grads(model(), 384, 128))
