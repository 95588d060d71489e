"""Phase timers of nms_round_kernel (library built with -DEFFDET_NMS_PROF): python tools/nms_prof.py NETWORK B S"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, _lib as L
net, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
c = EFFICIENTDET[net]
torch.manual_seed(0)
m = EfficientDet(80, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=False, compute_dtype=torch.bfloat16).cuda().eval()
img = torch.randn(B, 3, S, S, device='cuda')
with torch.no_grad():
    m.detect(img); torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.lib().effdet_nms_prof(out)            # reset
    m.detect(img); torch.cuda.synchronize()
    L.lib().effdet_nms_prof(out)
v = [int(x) for x in out]
names = ['a-prime scan', 'compaction', 'bit-matrix', 'greedy resolve', 'append']
tot = sum(v[:5])
for n, x in zip(names, v[:5]):
    print('%-16s %10d ticks  %5.1f %%' % (n, x, 100.0 * x / max(tot, 1)))
print('tiles %d  survivors/tile %.1f' % (v[6], v[5] / max(v[6], 1)))
