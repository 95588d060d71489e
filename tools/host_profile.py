"""Where does the HOST time of an eager train step go?  cProfile over a few eager D0 steps (B = 32 @ 512, bf16x3): top functions by
own time and by cumulative time.  (N > 1 runs eager under DDP: the step is host-bound as soon as this exceeds the GPU time.)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops, synthetic_batch  # noqa: E402
from efficientdet.pytorch_amd.optim import ClipAdamW  # noqa: E402

B, S = int(os.environ.get('HP_B', 32)), int(os.environ.get('HP_S', 512))
c = EFFICIENTDET['efficientdet-d0']
torch.manual_seed(0)
m = EfficientDet(80, network='efficientdet-d0', W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32,
                 f32_arith=os.environ.get('HP_ARITH', 'f32_hf16x3_bwd_bf16x3')).cuda()
m.train(); m.is_training = True; m.freeze_bn()
opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
img, ann = synthetic_batch(B, S, seed=1, num_classes=80)
img, ann = img.cuda(), ann.cuda()


def step():
    opt.zero_grad(set_to_none=True)
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host ms/step (issue only) %.2f' % ((t1 - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    print('==== by', key)
    pstats.Stats(pr).sort_stats(key).print_stats(28)
