"""Would running the high-resolution MBConv blocks depth-first over batch CHUNKS pay?  (The 6x-expanded maps of blocks 0-3 are
268 / 805 / 302 / 302 MB at B = 32 in fp32 -- none fits the 256 MB Infinity Cache between its producer and its consumer; at
B = 8 they are 67 / 201 / 75 / 75 MB.)  Times stem + blocks 0..N-1 forward + backward through functional.py at B = 32 in one
piece vs. depth-first over chunks of 16 / 8 / 4 images (same kernels, same total work; the chunked form runs the per-node tail
jobs once per chunk, so this slightly UNDER-states what a fused implementation would get).
    python tools/chunk_probe.py [nblocks=4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops         # noqa: E402
from efficientdet.pytorch_amd import functional as Fn                        # noqa: E402
from efficientdet.pytorch_amd.ops import Map                                 # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = 'efficientdet-d0'
c = EFFICIENTDET[net]
m = EfficientDet(80, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32, f32_arith='bf16x3').cuda()
ops.set_f32_arith('bf16x3')
bb = m.backbone
dt = torch.float32
B, S = 32, 512
img = torch.randn(B, 3, S, S, device='cuda')


def params(i):
    mm, blk = bb._blocks[i], bb.plan[i]
    P = {'bn1.running_mean': mm._bn1.running_mean, 'bn1.running_var': mm._bn1.running_var, 'bn2.running_mean': mm._bn2.running_mean,
         'bn2.running_var': mm._bn2.running_var, 'dw.weight': mm._depthwise_conv.weight, 'bn1.weight': mm._bn1.weight, 'bn1.bias': mm._bn1.bias,
         'se_reduce.weight': mm._se_reduce.weight, 'se_reduce.bias': mm._se_reduce.bias, 'se_expand.weight': mm._se_expand.weight,
         'se_expand.bias': mm._se_expand.bias, 'project.weight': mm._project_conv.weight, 'bn2.weight': mm._bn2.weight, 'bn2.bias': mm._bn2.bias}
    if blk.expand != 1:
        P.update({'bn0.running_mean': mm._bn0.running_mean, 'bn0.running_var': mm._bn0.running_var, 'expand.weight': mm._expand_conv.weight,
                  'bn0.weight': mm._bn0.weight, 'bn0.bias': mm._bn0.bias})
    return P


PS = [params(i) for i in range(NB)]


PREPS = {}


def run(chunk):
    prep = PREPS.setdefault(chunk, ops.ParamPrep('bf16x3'))       # batched parameter repacks, as in the model (one launch per run)
    ops.set_prep(prep); prep.begin_step(img.device)
    with torch.no_grad():
        for c0 in range(0, B, chunk):
            im = img[c0:c0 + chunk]
            bn = bb._bn0
            z, ssv = Fn.stem_fwd(im, bb._conv_stem.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bb.stem_pad, dt, True, z_only=True)
            x, saved = z, []
            for i in range(NB):
                rs = torch.ones(chunk, device='cuda') if bb.plan[i].skip else None
                x, sv = Fn.mbconv_fwd(x, bb.plan[i], PS[i], dt, True, rs, xpre=z if i == 0 else None, in_act=ops.ACT_SWISH if i == 0 else ops.ACT_NONE)
                saved.append(sv)
            dy = Map.new(x.B, x.H, x.W, x.C, dt, 'cuda', zero=True)
            for i in range(NB - 1, -1, -1):
                with ops.unpack_batch():
                    dy, g = Fn.mbconv_bwd(saved[i], dy)
            Fn.stem_bwd(ssv, dy, dy_is_dz=True)


for chunk in (32, 16, 8, 4, 32, 8):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run(chunk)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    if PREPS[chunk].dirty:
        PREPS[chunk]._build()
    g = torch.cuda.CUDAGraph()                                      # replayed as a hipGraph: the chunked forms must not be host-bound
    with torch.cuda.graph(g):
        run(chunk)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print('stem + blocks 0..%d fwd+bwd, B=32 in chunks of %2d: %.3f ms (hipGraph replay)' % (NB - 1, chunk, e0.elapsed_time(e1) / 10))
    del g
