"""Does whole-step capture work for a configuration?  python tools/graph_probe.py DTYPE NC B S [torchopt]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ddp, synthetic_batch
from efficientdet.pytorch_amd.graph import GraphedTrainStep
from efficientdet.pytorch_amd.optim import ClipAdamW
dt = torch.float32 if sys.argv[1] == 'f32' else torch.bfloat16
nc, B, S = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
c = EFFICIENTDET['efficientdet-d0']
torch.manual_seed(0)
m = EfficientDet(nc, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], compute_dtype=dt).cuda()
m.train(); m.freeze_bn(); ddp.freeze_dead_parameters(m)
ps = [p for p in m.parameters() if p.requires_grad]
opt = ClipAdamW(ps, lr=1e-4, max_norm=0.1)
img, ann = synthetic_batch(B, S, seed=1, num_classes=nc)
g = GraphedTrainStep(m, opt, img.cuda(), ann.cuda())
for _ in range(3):
    cl, rl = g()
torch.cuda.synchronize()
print('OK', sys.argv[1:], float(cl), float(rl))
