"""Per-launch breakdown (HIP events) of the conv / depthwise launches of one eval forward: python tools/eval_profile.py NET B S"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops
net, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
c = EFFICIENTDET[net]
m = EfficientDet(80, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=False, compute_dtype=torch.bfloat16).cuda().eval()
img = torch.randn(B, 3, S, S, device='cuda')
with torch.no_grad():
    for _ in range(2):
        m.forward_raw(img)
    ops.PROFILE = ops.LaunchProfile()
    m.forward_raw(img); torch.cuda.synchronize()
rec = [(n, f, e0.elapsed_time(e1), note) for n, f, e0, e1, note in ops.PROFILE.records]
ops.PROFILE = None
print('total profiled ms %.2f over %d launches' % (sum(r[2] for r in rec), len(rec)))
agg = {}
for n, f, ms, note in rec:
    a = agg.setdefault((n, note), [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += f
for (n, note), (cnt, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOPN', '36'))]:
    unit = 'GB/s' if note.startswith('BYTES') else 'TF/s'
    print('%-30s %-36s x%-2d %7.3f ms  %7.1f %s' % (n, note, cnt, ms, f / ms / (1e6 if unit == 'GB/s' else 1e9), unit))
