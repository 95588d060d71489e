"""Run-to-run determinism + correctness stress of the fp32 weight-gradient kernels on small single-level shapes."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientdet.pytorch_amd import ops
from efficientdet.pytorch_amd.ops import Map
bad = 0
for (B, H, W, Cin, Cout) in [(2, 4, 4, 64, 64), (2, 2, 2, 64, 64), (2, 8, 8, 64, 64), (2, 16, 16, 64, 64), (2, 4, 4, 64, 256), (3, 4, 2, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g); dz = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    (F.conv2d(x, w, None, 1, 1) * dz).sum().backward()
    xm = Map.of(x.permute(0, 2, 3, 1).contiguous().cuda()); zm = Map.of(dz.permute(0, 2, 3, 1).contiguous().cuda())
    outs = []
    for rep in range(40):
        junk = torch.randn(1 << 20, device='cuda')          # perturb allocator / cache state between runs
        db = torch.zeros(Cout, device='cuda')
        G, _dbp = ops.conv2d_wgrad(xm, zm, None, db, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1)
        dw = torch.empty(Cout, Cin, 3, 3, device='cuda'); ops.unpack_wgrad(G, dw)
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    err = max(float((o - w.grad).abs().max()) for o in outs)
    same = all(torch.equal(outs[0], o) for o in outs)
    print((B, H, W, Cin, Cout), 'max err vs torch %.3e' % err, 'bitwise stable' if same else 'RUN-TO-RUN DIFFERENCES', flush=True)
    bad += (not same) or err > 1e-3
print('BAD' if bad else 'ALL OK')
