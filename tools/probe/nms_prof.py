import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, _lib as L
cfg = EFFICIENTDET['efficientdet-d0']; torch.manual_seed(0)
m = EfficientDet(80, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], is_training=False, compute_dtype=torch.bfloat16).cuda().eval()
img = torch.randn(32, 3, 512, 512, device='cuda')
with torch.no_grad():
    m.detect(img); torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)(); L.lib().effdet_nms_prof(out)
    m.detect(img); torch.cuda.synchronize()
    L.lib().effdet_nms_prof(out)
names = ['a_prime', 'compact', 'bitmatrix', 'fixedpoint', 'append', 'sum_S', 'subtiles', 'fp_iters']
print({n: int(v) for n, v in zip(names, out)})
