"""GPU parity: BiFPN fast-normalised fusion nodes forward/backward (incl. raw weight gradients)."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close
from tests.test_gpu_backbone_ops import nhwc, nchw, q_, TOL, DT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_fuse_node(dtype, mode):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(10 + mode)
    q = q_(dtype)
    B, H, W, C = 2, 6, 8, 64
    rows, cols, col = (3, 3, 1) if mode == 1 else (2, 5, 2)
    wraw = (0.2 + torch.rand(rows, cols, generator=g))
    if mode == 0:
        wraw[1, 3] = -0.4
    wraw.requires_grad_(True)
    a = q(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
    bshape = (B, C, H // 2, W // 2) if mode == 0 else (B, C, 2 * H, 2 * W)
    b = q(torch.randn(bshape, generator=g)).requires_grad_(True)
    c = q(torch.randn(B, C, H, W, generator=g)).requires_grad_(True) if mode == 1 else None
    eps = 1e-4
    wn = F.relu(wraw); wn = wn / (wn.sum(0) + eps)
    if mode == 0:
        out = (wn[0, col] * a + wn[1, col] * F.interpolate(b, scale_factor=2, mode='nearest')) / (wn[0, col] + wn[1, col] + eps)
    elif mode == 1:
        out = (wn[0, col] * a + wn[1, col] * F.max_pool2d(b, 2) + wn[2, col] * c) / (wn[0, col] + wn[1, col] + wn[2, col] + eps)
    else:
        out = (wn[0, col] * a + wn[1, col] * F.max_pool2d(b, 2)) / (wn[0, col] + wn[1, col] + eps)
    dout = q(torch.randn(out.shape, generator=g))
    out.backward(dout)
    dev = 'cuda'
    am, bm = nhwc(a.detach(), dtype), nhwc(b.detach(), dtype)
    cm = nhwc(c.detach(), dtype) if c is not None else None
    wd = wraw.detach().to(dev)
    om = ops.bifpn_fuse_fwd(am, bm, cm, wd, col, mode)
    torch.cuda.synchronize()
    assert_close(nchw(om), out.detach(), TOL[dtype], 'fuse fwd')
    da = Map.new(am.B, am.H, am.W, C, dtype, dev); db = Map.new(bm.B, bm.H, bm.W, C, dtype, dev)
    dc = Map.new(am.B, am.H, am.W, C, dtype, dev) if mode == 1 else None
    dn = torch.zeros(ops.fuse_dn_floats(cols), device=dev)
    ops.bifpn_fuse_bwd(nhwc(dout, dtype), am, bm, cm, da, db, dc, False, False, False, wd, dn, col, mode)
    dw = torch.zeros(rows, cols, device=dev)
    ops.bifpn_weight_bwd(wd, dn, dw)
    torch.cuda.synchronize()
    assert_close(nchw(da), a.grad, TOL[dtype], 'fuse da'); assert_close(nchw(db), b.grad, TOL[dtype], 'fuse db')
    if mode == 1:
        assert_close(nchw(dc), c.grad, TOL[dtype], 'fuse dc')
    assert_close(dw.cpu(), wraw.grad, 3e-2 if dtype == torch.bfloat16 else 2e-3, 'fuse dw')
    # accumulate flags
    da2 = Map.of(da.tensor().clone())
    ops.bifpn_fuse_bwd(nhwc(dout, dtype), am, bm, cm, da2, db, dc, True, False, False, wd, dn, col, mode)
    assert_close(nchw(da2), q(2 * a.grad), 2 * TOL[dtype], 'fuse da accumulate')
