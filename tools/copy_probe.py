"""Which aten ops sit between the HIP launches of one eval forward / one train step (torch.profiler, CPU-side op list)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
from efficientdet.pytorch_amd.synthetic import synthetic_batch
from torch.profiler import profile, ProfilerActivity
mode = sys.argv[1] if len(sys.argv) > 1 else 'eval'
c = EFFICIENTDET['efficientdet-d0']
m = EfficientDet(80, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=(mode == 'train'), compute_dtype=torch.bfloat16).cuda()
img, ann = synthetic_batch(8, 512, seed=1, num_classes=80)
img, ann = img.cuda(), ann.cuda()
if mode == 'train':
    m.train(); m.freeze_bn()
    run = lambda: sum(x.mean() for x in m([img, ann])).backward()
else:
    m.eval()
    run = lambda: m.forward_raw(img)
ctx = torch.enable_grad() if mode == 'train' else torch.no_grad()
with ctx:
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        run(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
