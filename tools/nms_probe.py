"""Convergence of the spatial-hash NMS sweeps on the benchmark's worst case (random-init D0: every anchor a candidate)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops, _lib as L
B, S = int(sys.argv[1]), int(sys.argv[2]); net = sys.argv[3] if len(sys.argv) > 3 else 'efficientdet-d0'
c = EFFICIENTDET[net]
torch.manual_seed(0)
m = EfficientDet(80, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], is_training=False, compute_dtype=torch.bfloat16).cuda().eval()
img = torch.randn(B, 3, S, S, device='cuda')
with torch.no_grad():
    cls, reg, anc = m.forward_raw(img)
    boxes, score, label = ops.decode_score(anc, reg, cls, S, S)
A = score.shape[1]
nbytes = int(L.lib().effdet_nms_workspace_bytes(B, C.c_longlong(A)))
ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
idx = torch.empty((B, A), dtype=torch.int32, device='cuda'); cnt = torch.empty(B, dtype=torch.int32, device='cuda')
for rep in range(3):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    L.check(L.lib().effdet_nms(L.ptr(boxes), L.ptr(score), C.c_float(0.01), C.c_float(0.5), L.ptr(idx), L.ptr(cnt), L.ptr(ws), C.c_longlong(nbytes), B, C.c_longlong(A), L.stream_ptr()), 'nms')
    e1.record(); torch.cuda.synchronize()
print('nms %.3f ms for B=%d A=%d; kept[0]=%d' % (e0.elapsed_time(e1), B, A, int(cnt[0])))
ITERS = 24
tail = ((B * (ITERS + 1) * 4 + 255) // 256) * 256
und = ws[nbytes - tail:nbytes - tail + B * (ITERS + 1) * 4].view(torch.int32).view(B, ITERS + 1).cpu()
print('undecided after sweep t (image 0):', und[0].tolist())
print('undecided after sweep t (sum over images):', und.sum(0).tolist())
