"""Host time to ISSUE one eager train step (D0 B = 32 @ 512, bf16x3) alone and next to K other host processes that are issuing
kernel launches of their own (what 8 ranks sharing one host do to each other under eager DDP): K dummy processes each run a
throttled loop of tiny launches (~15 k launches / s, the rate of an eager step) on the same GPU.  VERDICT r3 #2.
    python tools/host_contention.py [K=7]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 7
DUMMY = r"""
import time, torch
x = torch.zeros(64, device='cuda'); torch.cuda.synchronize()
t_end = time.time() + float(%r)
while time.time() < t_end:
    t0 = time.perf_counter()
    for _ in range(400):
        x.add_(1.0)                      # ~400 launches, the count of one eager step
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dt < 0.027: time.sleep(0.027 - dt)     # ... per 27 ms, the step time
"""


def main():
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, synthetic_batch
    from efficientdet.pytorch_amd.optim import ClipAdamW
    c = EFFICIENTDET['efficientdet-d0']
    torch.manual_seed(0)
    m = EfficientDet(80, network='efficientdet-d0', W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32,
                     f32_arith='bf16x3').cuda()
    m.train(); m.is_training = True; m.freeze_bn()
    opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
    img, ann = synthetic_batch(32, 512, seed=1, num_classes=80)
    img, ann = img.cuda(), ann.cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        cl, rl = m([img, ann])
        (cl.mean() + rl.mean()).backward()
        opt.step()

    def measure(tag, n=10):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%-34s host %.2f ms/step to issue, wall %.2f ms/step' % (tag, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
    measure('alone (1 host process):')
    procs = [subprocess.Popen([sys.executable, '-c', DUMMY % 60.0]) for _ in range(K)]
    time.sleep(20)                       # the dummies import torch and create their contexts
    measure('with %d concurrent launch loops:' % K)
    for p in procs:
        p.kill()
    for p in procs:
        p.wait()
    measure('alone again:')


if __name__ == '__main__':
    main()
