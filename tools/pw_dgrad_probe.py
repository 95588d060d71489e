"""Project-conv data gradient with its squeeze-excite / Swish' epilogue: the streaming MFMA kernel (effdet_pw_dgrad_se) against the launch it
replaces (skinny VALU kernel for Co 16 / 24, implicit GEMM for Co 40), D0 B = 32 @512 blocks 0-4."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops  # noqa: E402
from efficientdet.pytorch_amd.ops import Map  # noqa: E402

B = 32
torch.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (Co, Ce, H) in [(16, 32, 256), (24, 96, 128), (24, 144, 128), (40, 144, 64), (40, 240, 64)]:
    dy = Map.of(torch.randn(B, H, H, Co, device='cuda')); zd = Map.of(torch.randn(B, H, H, Ce, device='cuda'))
    wp = torch.randn(Co, Ce, 1, 1, device='cuda'); s2 = torch.rand(Co, device='cuda') + 0.5
    gate = torch.rand(B, Ce, device='cuda'); dpool = torch.randn(B, Ce, device='cuda') * 1e-3; rs = torch.rand(B, device='cuda') + 0.5
    wpk = ops.pack_weight(wp, torch.float32, mode=1, scale=s2)
    dzd = Map.new(B, H, H, Ce, torch.float32, 'cuda')
    f_old = lambda: ops.conv2d(dy, wpk, dzd, Cin=Co, Cout=Ce, KH=1, KW=1, rowscale=rs, bc_scale=gate, bc_shift=dpool, res=zd, res_mode=ops.RES_SWISH_GRAD)
    f_new = lambda: ops.pw_dgrad_se(dy, wp, s2, rs, gate, dpool, zd)
    to, tn = timeit(f_old), timeit(f_new)
    gb = 4.0 * B * H * H * (Co + 2 * Ce) / 1e9
    print('project dgrad %d -> %3d %3d^2: replaced %6.1f us (%.2f TB/s)  streaming %6.1f us (%.2f TB/s of %.2f GB)' % (Co, Ce, H, to, gb / to * 1e3, tn, gb / tn * 1e3, gb), flush=True)
