"""Where do the device-to-device copies of a train step come from?  One eager D0 step under torch.profiler with python stacks;
prints every Memcpy / aten::copy_ / aten::clone grouped by the innermost frames of this repository."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops, synthetic_batch  # noqa: E402
from efficientdet.pytorch_amd.optim import ClipAdamW  # noqa: E402

B, S = int(os.environ.get('CT_B', 8)), int(os.environ.get('CT_S', 512))
ops.set_f32_arith('bf16x3')
c = EFFICIENTDET['efficientdet-d0']
torch.manual_seed(0)
m = EfficientDet(80, network='efficientdet-d0', W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32).cuda()
m.train(); m.is_training = True; m.freeze_bn()
opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
img, ann = synthetic_batch(B, S, seed=1, num_classes=80)
img, ann = img.cuda(), ann.cuda()


def step():
    opt.zero_grad(set_to_none=True)
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::add_', 'aten::add', 'aten::mul', 'aten::zero_', 'aten::fill_', 'aten::sum', 'aten::mean'):
        fr = [s for s in (e.stack or []) if 'efficientdet' in s or 'bench' in s or 'autograd' in s][:3]
        cnt[(e.name, str(e.input_shapes)[:60], ' <- '.join(f.split('/')[-1][:60] for f in fr))] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
kn = collections.Counter(e.name[:80] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print('--- device activities'); print(sum(kn.values()))
for k, v in kn.most_common(60):
    print(v, k)
