"""configs[1] / configs[4]: eval forward + decode + on-device NMS on random-init weights (every anchor a candidate: the NMS worst case).
Prints the forward alone, eager detect, the ONE-graph detect (graph.GraphedDetect) and the post-processing alone (decode + NMS + gather).
    python tools/infer_bench.py [--network efficientdet-d0 --batch 32 --size 512 --reps 10 --dtype f32_hf16x3|f32_bf16x3|f32|bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops       # noqa: E402
from efficientdet.pytorch_amd.graph import GraphedDetect                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--network', default='efficientdet-d0'); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=512); ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--dtype', default='f32_bf16x3'); ap.add_argument('--no-graph', action='store_true')
a = ap.parse_args()
cfg = EFFICIENTDET[a.network]
torch.manual_seed(0)
m = EfficientDet(80, network=a.network, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], D_class=cfg['D_class'], is_training=False,
                 compute_dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32,
                 f32_arith={'f32_bf16x3': 'bf16x3', 'f32_hf16x3': 'f32_hf16x3_bwd_bf16x3'}.get(a.dtype, 'f32')).cuda().eval()
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
    tf, (cls, reg, anc) = timeit(lambda: m.forward_raw(img))
    td, dets = timeit(lambda: m.detect(img))

    def post():
        boxes, score, label = ops.decode_score(anc, reg, cls, a.size, a.size)
        idx, count = ops.nms(boxes, score, float(m.threshold), float(m.iou_threshold))
        return ops.gather_dets(boxes, score, label, idx, count)
    tp, _ = timeit(post)
    tg = float('nan')
    if not a.no_graph and not os.environ.get('EFFDET_NMS_V1'):
        gd = GraphedDetect(m, img)
        tg, gdets = timeit(gd)
        assert all(torch.equal(x[0], y[0]) and torch.equal(x[2], y[2]) for x, y in zip(dets, gdets)), 'graph replay != eager'
print('%s B=%d @%d %s: forward %.4f ms/img | eager detect %.4f | one-graph detect %.4f | decode+NMS+gather alone %.4f ms/img (%.3f ms/batch) | kept[0]=%d'
      % (a.network, a.batch, a.size, a.dtype, tf / a.batch, td / a.batch, tg / a.batch, tp / a.batch, tp, dets[0][0].numel()))
