"""Per-launch breakdown (HIP events) of the MFMA conv launches of one D0 train step, sorted by time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops, ddp  # noqa: E402
from efficientdet.pytorch_amd.synthetic import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = sys.argv[2] if len(sys.argv) > 2 else 'bf16'          # bf16 | f32 | f32_bf16x3
cfg = EFFICIENTDET['efficientdet-d0']
m = EfficientDet(80, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], compute_dtype=torch.bfloat16 if MODE == 'bf16' else torch.float32,
                 f32_arith='bf16x3' if MODE == 'f32_bf16x3' else 'f32').cuda()
m.train(); ddp.freeze_dead_parameters(m)
img, ann = synthetic_batch(B, 512, seed=1)
img, ann = img.cuda(), ann.cuda()
for _ in range(2):
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward()
ops.PROFILE = ops.LaunchProfile()
cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward()
torch.cuda.synchronize()
rec = [(n, f, e0.elapsed_time(e1), note) for n, f, e0, e1, note in ops.PROFILE.records]
ops.PROFILE = None
tot = sum(r[2] for r in rec)
print('total conv ms %.2f over %d launches' % (tot, len(rec)))
agg = {}
for n, f, ms, note in rec:
    k = (n, note); a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += f
for (n, note), (c, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOPN', '32'))]:
    unit = 'GB/s' if note.startswith('BYTES') else 'TF/s'
    print('%-28s %-34s x%-2d %7.3f ms  %7.1f %s' % (n, note, c, ms, f / ms / (1e6 if unit == 'GB/s' else 1e9), unit))
