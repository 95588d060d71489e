"""Uninitialised-read detector: poison the caching allocator's free blocks with NaN bit patterns, then run one train step and
report which outputs / gradients are non-finite or differ from a clean run.  python tools/poison_probe.py H W [dtype]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
from oracle import effdet_oracle as O            # (weights / inputs generator only)
H, W = int(sys.argv[1]), int(sys.argv[2])
dtype = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == 'bf16') else torch.float32
net, nc, B = 'efficientdet-d0', 12, 2
c = EFFICIENTDET[net]
sd = O.make_state_dict(net, nc, seed=4)
g = torch.Generator().manual_seed(H * 7 + W)
img = torch.randn(B, 3, H, W, generator=g).cuda()
ann = torch.full((B, 4, 5), -1.0)
ann[0, 0] = torch.tensor([10., 12., 90., 100., 3.]); ann[1, 0] = torch.tensor([W / 2 - 30., 20., W / 2 + 34., 110., 0.])
ann = ann.cuda()


def run():
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], compute_dtype=dtype)
    m.load_state_dict(sd); m.backbone.drop_connect_rate = 0.0
    m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, (float(cl.detach()), float(rl.detach()))


def poison(gb=6):
    t = torch.empty(gb * (1 << 28), dtype=torch.int32, device='cuda'); t.fill_(0x7fc07fc0)      # NaN as fp32 and as bf16 pairs
    del t                                                                                        # stays cached: later allocations reuse it


clean, lc = run()
torch.cuda.empty_cache()
poison()
dirty, ld = run()
print('losses clean', lc, 'poisoned', ld)
bad = []
for k in clean:
    a, b = clean[k].double(), dirty[k].double()
    if not bool(torch.isfinite(b).all()):
        bad.append((float('inf'), k))
    else:
        d = float((a - b).norm()) / (float(a.norm()) + 1e-30)
        if d > 1e-5:
            bad.append((d, k))
bad.sort(reverse=True)
print('%d of %d gradient tensors differ / are non-finite after poisoning:' % (len(bad), len(clean)))
for d, k in bad[:40]:
    print('  %-60s %s' % (k, d))
