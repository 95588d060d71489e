"""State-leak probe: gradients of case B run first, then again after case A ran in the same process.
python tools/seq_probe.py [noprep] [noeval]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
from oracle import effdet_oracle as O
net, nc, B = 'efficientdet-d0', 12, 2
c = EFFICIENTDET[net]
sd = O.make_state_dict(net, nc, seed=4)
NOPREP, NOEVAL = 'noprep' in sys.argv, 'noeval' in sys.argv


def run(H, W):
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g).cuda()
    ann = torch.full((B, 4, 5), -1.0)
    ann[0, 0] = torch.tensor([10., 12., 90., 100., 3.]); ann[0, 1] = torch.tensor([W - 70., H - 64., W - 5., H - 9., 7.])
    ann[1, 0] = torch.tensor([W / 2 - 30., 20., W / 2 + 34., 110., 0.])
    ann = ann.cuda()
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], compute_dtype=torch.float32)
    m.load_state_dict(sd); m.backbone.drop_connect_rate = 0.0
    m = m.cuda()
    if NOPREP:
        m.batched_prep = False
    if not NOEVAL:
        m.eval(); m.is_training = False
        with torch.no_grad():
            m.forward_raw(img)
    m.train(); m.is_training = True; m.freeze_bn()
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
    print('   loss %dx%d: %.9g %.9g' % (H, W, float(cl.detach()), float(rl.detach())))
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


from efficientdet.pytorch_amd import functional as Fn, ops as _ops
LOG = []
_orig = Fn.mbconv_bwd


def _logged(sv, dy):
    dx, g = _orig(sv, dy)
    torch.cuda.synchronize()
    LOG[-1].append((sv['blk'].cin, sv['blk'].cexp, float(dy.tensor().double().norm()), float(dx.tensor().double().norm()),
                    {k: float(v.double().norm()) for k, v in g.items()}))
    return dx, g


Fn.mbconv_bwd = _logged
LOG.append([])
b1 = run(384, 128)
LOG.append([])
a = run(128, 256)
LOG.append([])
if 'flush' in sys.argv:
    torch.cuda.empty_cache()
if 'nanfill' in sys.argv:
    torch.cuda.empty_cache()
    junk = [torch.full((n,), float('nan'), device='cuda') for n in [1 << k for k in range(8, 27)] * 6]
    del junk
b2 = run(384, 128)
for i, (r1, r2) in enumerate(zip(LOG[0], LOG[2])):
    dgs = {k: abs(r1[4][k] - r2[4][k]) / (r1[4][k] + 1e-30) for k in r1[4]}
    kmax = max(dgs, key=dgs.get)
    print('bwd call %2d (cin %d cexp %d): dy rel diff %.1e  dx rel diff %.1e  worst param %s %.1e' % (i, r1[0], r1[1], abs(r1[2] - r2[2]) / r1[2], abs(r1[3] - r2[3]) / r1[3], kmax, dgs[kmax]))
bad = sorted(((float((b1[k].double() - b2[k].double()).norm()) / (float(b1[k].double().norm()) + 1e-30), k) for k in b1), reverse=True)
print('noprep' if NOPREP else 'prep', 'noeval' if NOEVAL else 'eval-first', [(k.replace('backbone._blocks.', 'b'), '%.1e' % d) for d, k in bad[:8]])
