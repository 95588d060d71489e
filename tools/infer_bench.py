"""configs[1] / configs[4]: eval forward + decode + on-device NMS on random-init weights (NMS worst case).
python tools/infer_bench.py [--network efficientdet-d0 --batch 32 --size 512 --reps 10]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET

ap = argparse.ArgumentParser()
ap.add_argument('--network', default='efficientdet-d0'); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=512); ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
cfg = EFFICIENTDET[a.network]
torch.manual_seed(0)
m = EfficientDet(80, network=a.network, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], D_class=cfg['D_class'], is_training=False,
                 compute_dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32).cuda().eval()
img = torch.randn(a.batch, 3, a.size, a.size, device='cuda')


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(a.reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.reps * 1e3, r


with torch.no_grad():
    tf, _ = timeit(lambda: m.forward_raw(img))
    td, dets = timeit(lambda: m.detect(img))
print('%s B=%d @%d %s: forward %.3f ms/img, forward+decode+NMS %.3f ms/img, kept[0]=%d'
      % (a.network, a.batch, a.size, a.dtype, tf / a.batch, td / a.batch, dets[0][0].numel()))
