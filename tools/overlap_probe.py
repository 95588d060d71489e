"""Can MFMA-bound leaf work (the head's weight gradients) hide the latency-bound trunk kernels when both run as two branches of one
hipGraph?  A = 10 split-layout weight gradients of the RetinaHead shape (B = 32 @512: ~5.4 ms), B = the D0 trunk forward (stem + 16 MBConv
blocks + BiFPN: ~150 small / HBM-bound launches).  Timed as graph replays: A alone, B alone, A then B on one stream, A || B on two."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, functional as Fn, ops  # noqa: E402

dev = 'cuda'
B = 32
sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
cfg = EFFICIENTDET['efficientdet-d0']
torch.manual_seed(0)
m = EfficientDet(80, network='efficientdet-d0', W_bifpn=64, D_bifpn=2, D_class=3, is_training=False, f32_arith='f32').cuda().eval()
img = torch.randn(B, 3, 512, 512, device=dev)


def pyr(C):
    flat, maps = Fn.pyramid_alloc(B, sizes, C, torch.float32, dev)
    flat.copy_(ops.to_split(torch.relu(torch.randn(flat.numel(), device=dev))))
    return maps


x, dz = pyr(256), pyr(256)


def A():
    ops.set_f32_arith('bf16x3')
    for _ in range(10):
        ops.conv2d_wgrad(x, dz, Cin=256, Cout=256, KH=3, KW=3, pad_t=1, pad_l=1, split=True)
    ops.set_f32_arith('f32')


def Bf():
    with torch.no_grad():
        m.extract_feat(img)


side = torch.cuda.Stream()


def seq():
    A(); Bf()


def par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        A()
    Bf()
    main.wait_stream(side)


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g


def timeit(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (('A  (10 head weight gradients)', A), ('B  (trunk forward)', Bf), ('A ; B one stream', seq), ('A || B two branches', par)):
    print('%-32s %.3f ms' % (name, timeit(graphed(fn))), flush=True)
