"""Is the oracle ('port') a fair stand-in for the reference on the CPU?  BUILD CONTAINER ONLY (needs /root/reference): the real reference
(imported through oracle/make_golden.py's shims) and the oracle timed on the same weights and inputs, same thread count --
configs[0] (B = 1 forward to (cls, reg, anchors)) and SURVEY 8d's B = 4 forward + FocalLoss + backward at 512 x 512."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O      # noqa: E402
from oracle import make_golden as MG       # noqa: E402

torch.set_num_threads(int(os.environ.get('THREADS', '8')))
net, nc = 'efficientdet-d0', 80
E = MG.import_reference()
sd = O.make_state_dict(net, nc, seed=0)
ref = MG.build_ref(E, net, nc, sd, is_training=False)
ref.eval()
img1, _ = O.synthetic_batch(1, 512, seed=1, num_classes=nc)
img4, ann4 = O.synthetic_batch(4, 512, seed=1, num_classes=nc)


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)


with torch.no_grad():
    def ref_fwd():
        f = ref.extract_feat(img1); o = ref.bbox_head(f)
        return torch.cat(list(o[0]), 1), torch.cat(list(o[1]), 1), ref.anchors(img1)
    t_ref = best(ref_fwd)
    t_orc = best(lambda: O.forward_raw(sd, net, nc, img1))
print('configs[0] B=1 forward, %d threads: reference %.1f ms, oracle %.1f ms' % (torch.get_num_threads(), t_ref * 1e3, t_orc * 1e3))
reft = MG.build_ref(E, net, nc, sd, is_training=True)
reft.train(); reft.is_training = True; reft.freeze_bn()
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k
          and not k.startswith(('backbone._conv_head', 'backbone._bn1', 'backbone._fc'))}
live = dict(sd); live.update(params)


def ref_step():
    reft.zero_grad()
    cl, rl = reft([img4, ann4]); (cl.mean() + rl.mean()).backward()


def orc_step():
    for p in params.values():
        p.grad = None
    cl, rl = O.train_losses(live, net, nc, img4, ann4); (cl.mean() + rl.mean()).backward()


t_ref = best(ref_step, 2); t_orc = best(orc_step, 2)
print('B=4 forward + loss + backward, %d threads: reference %.2f img/s, oracle %.2f img/s' % (torch.get_num_threads(), 4 / t_ref, 4 / t_orc))
