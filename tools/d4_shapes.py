"""configs[4] forward by launch shape (default: the headline forward arithmetic): every MFMA launch of one eager D4 B=8 @1024 forward, grouped by kernel symbol and shape."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops
net = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d4'
arith = sys.argv[2] if len(sys.argv) > 2 else 'f32_hf16x3_bwd_bf16x3'
B, S = (8, 1024) if net.endswith('d4') else (32, 512)
cfg = EFFICIENTDET[net]
torch.manual_seed(0)
m = EfficientDet(80, network=net, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], D_class=cfg['D_class'], is_training=False, f32_arith=arith).cuda().eval()
img = torch.randn(B, 3, S, S, device='cuda')
with torch.no_grad():
    for _ in range(2):
        m.forward_raw(img)
    ops.PROFILE = ops.LaunchProfile()
    m.forward_raw(img)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, flops, e0, e1, note, nb in ops.PROFILE.records:
    d = agg.setdefault((name, note), [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += flops
tot = sum(v[1] for v in agg.values())
print('%s B=%d @%d %s, one eager forward: %.2f ms in %d timed launches' % (net, B, S, arith, tot, sum(v[0] for v in agg.values())))
for (name, note), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    rate = fl / (ms * 1e-3) / 1e12
    print('%-38s %-40s x%-3d %8.3f ms  %7.1f %s' % (name, note, n, ms, rate if not note.startswith('BYTES') else rate * 1e3, 'TFLOP/s' if not note.startswith('BYTES') else 'GB/s'))
