"""python tools/graph_probe_detect.py NETWORK B S DTYPE -- does GraphedDetect capture + replay at this size?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, synthetic_batch
from efficientdet.pytorch_amd.graph import GraphedDetect
net, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dt = torch.float32 if sys.argv[4] == 'f32' else torch.bfloat16
c = EFFICIENTDET[net]
torch.manual_seed(0)
m = EfficientDet(80, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=False, compute_dtype=dt).cuda().eval()
img = synthetic_batch(B, S, seed=1)[0].cuda()
e = m.detect(img)
print('eager ok', len(e[0][0]), flush=True)
g = GraphedDetect(m, img)
print('captured', flush=True)
for _ in range(3):
    r = g()
torch.cuda.synchronize()
print('OK', sys.argv[1:], len(r[0][0]), flush=True)
