"""GPU parity: depthwise conv (fwd/dgrad/wgrad), squeeze-excite (fwd/bwd), BN fold, elementwise helpers."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close, assert_close_scale

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}


def q_(dtype):
    return (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)


def nhwc(t, dtype):
    from efficientdet.pytorch_amd.ops import Map
    return Map.of(t.permute(0, 2, 3, 1).contiguous().to('cuda', dtype))


def nchw(m):
    return m.tensor().float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cfg', [(2, 16, 16, 32, 3, 1, (1, 1)), (2, 17, 17, 96, 3, 2, (0, 1)), (1, 16, 16, 144, 5, 2, (1, 2)),
                                 (2, 8, 8, 240, 5, 1, (2, 2)), (3, 4, 4, 1152, 3, 2, (0, 1)), (2, 9, 7, 672, 5, 1, (2, 2)),
                                 (2, 2, 2, 1152, 3, 2, (0, 1)), (2, 40, 24, 48, 5, 1, (2, 2)), (2, 33, 33, 40, 5, 2, (1, 2)),
                                 (4, 64, 64, 64, 3, 1, (1, 1))])
def test_dwconv_fwd_bwd(dtype, cfg):
    from efficientdet.pytorch_amd import ops
    B, H, W, C, k, s, (plo, phi) = cfg
    g = torch.Generator().manual_seed(1)
    q = q_(dtype)
    x = q(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(C, 1, k, k, generator=g) * (2.0 / (k * k)) ** 0.5).requires_grad_(True)
    scale = 0.5 + torch.rand(C, generator=g); shift = torch.randn(C, generator=g) * 0.2
    c = F.conv2d(F.pad(x, [plo, phi, plo, phi]), w, None, s, 0, 1, C)
    z = c * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = z * torch.sigmoid(z)
    Ho, Wo = y.shape[2:]
    dev = 'cuda'
    wk = ops.dw_pack_weight(w.detach().to(dev))
    assert_close(wk.cpu(), w.detach().reshape(C, k * k).t(), 1e-7, 'dw pack')
    xm = nhwc(x.detach(), dtype)
    ym, zm, pp = ops.dwconv_fwd(xm, wk, scale.to(dev), shift.to(dev), k, s, plo, plo, Ho, Wo, save_z=True, pool=True)
    torch.cuda.synchronize()
    assert_close(nchw(zm), z, TOL[dtype], 'dw z'); assert_close(nchw(ym), y, TOL[dtype], 'dw y')
    # the squeeze-excite pool arrives as per-(image, tile group) partial rows [B][G][C]: no atomics, two launches bitwise equal
    assert pp.dim() == 3 and pp.shape[0] == B and pp.shape[2] == C
    assert_close(pp.sum(dim=1).cpu(), q(y.detach()).sum(dim=(2, 3)), 5 * TOL[dtype], 'se pool')
    assert torch.equal(ops.dwconv_fwd(xm, wk, scale.to(dev), shift.to(dev), k, s, plo, plo, Ho, Wo, save_z=True, pool=True)[2], pp)
    # z-only storage (training): same stored z, no y, and the pooled sum is that of Swish(stored z)
    y2, z2, pp2 = ops.dwconv_fwd(xm, wk, scale.to(dev), shift.to(dev), k, s, plo, plo, Ho, Wo, save_z=True, pool=True, save_y=False)
    assert y2 is None and torch.equal(z2.tensor(), zm.tensor())
    zs = nchw(z2)
    assert_close(pp2.sum(dim=1).cpu(), (zs * torch.sigmoid(zs)).sum(dim=(2, 3)), 5 * TOL[dtype], 'se pool (z-only)')
    # backward wrt the pre-activation z: dz given
    dz = q(torch.randn(z.shape, generator=g))
    z.backward(dz)
    dzm = nhwc(dz, dtype)
    dxm = ops.dwconv_dgrad(dzm, wk, scale.to(dev), None, H, W, k, s, plo, plo)
    gk, dsum = ops.dwconv_wgrad(xm, dzm, k, s, plo, plo)
    gk2, dsum2 = ops.dwconv_wgrad(xm, dzm, k, s, plo, plo)
    assert torch.equal(gk, gk2) and torch.equal(dsum, dsum2)              # slab reduction in a fixed order: bitwise reproducible
    wsum = torch.empty(C, device=dev)
    dw = ops.dw_unpack_wgrad(gk, scale.to(dev), w.detach().to(dev), wsum)
    # the frozen-BN form, launched directly and as a job of the node's tail batch: same bits
    mean, inv = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    single = ops.dw_unpack_wgrad_bn(gk, scale.to(dev), w.detach().to(dev), dsum, mean, inv)
    with ops.unpack_batch():
        batched = ops.dw_unpack_wgrad_bn(gk, scale.to(dev), w.detach().to(dev), dsum, mean, inv)
    torch.cuda.synchronize()
    for a, b in zip(single, batched):
        assert torch.equal(a, b)
    assert torch.equal(single[0], dw)
    assert_close(nchw(dxm), x.grad, TOL[dtype], 'dw dgrad')
    assert_close(dw.cpu(), w.grad, 5 * TOL[dtype], 'dw wgrad')
    assert_close(dsum.cpu(), dz.sum(dim=(0, 2, 3)), 5 * TOL[dtype], 'dw dsum')
    assert_close(wsum.cpu(), (w.detach() * (w.grad / scale.view(-1, 1, 1, 1))).sum(dim=(1, 2, 3)), 1e-2, 'dw wsum')
    # z-only storage of the producing expand conv (training): handed the PRE-activation, forward and weight gradient Swish their
    # staged tiles themselves == the same kernels on the activated tensor
    pre = q(torch.randn(B, C, H, W, generator=g))
    am, pm = nhwc(q(pre * torch.sigmoid(pre)), dtype), nhwc(pre, dtype)
    _, za, pa = ops.dwconv_fwd(am, wk, scale.to(dev), shift.to(dev), k, s, plo, plo, Ho, Wo, save_z=True, pool=True, save_y=False)
    _, zb, pb = ops.dwconv_fwd(pm, wk, scale.to(dev), shift.to(dev), k, s, plo, plo, Ho, Wo, save_z=True, pool=True, save_y=False,
                               in_act=ops.ACT_SWISH)
    assert_close_scale(nchw(zb), nchw(za), 2 * TOL[dtype], 'dw z from the pre-activation')
    assert_close_scale(pb.sum(dim=1).cpu(), pa.sum(dim=1).cpu(), 2 * TOL[dtype], 'se pool from the pre-activation')
    ga, da = ops.dwconv_wgrad(am, dzm, k, s, plo, plo)
    gb, dbb = ops.dwconv_wgrad(pm, dzm, k, s, plo, plo, in_act=ops.ACT_SWISH)
    assert_close_scale(gb.cpu(), ga.cpu(), 2 * TOL[dtype], 'dw wgrad from the pre-activation'); assert torch.equal(dbb, da)
    # fused swish'(zprev) epilogue
    zp = q(torch.randn(B, C, H, W, generator=g))
    dxm2 = ops.dwconv_dgrad(dzm, wk, scale.to(dev), nhwc(zp, dtype), H, W, k, s, plo, plo)
    sg = torch.sigmoid(zp)
    assert_close(nchw(dxm2), x.grad * (sg * (1 + zp * (1 - sg))), TOL[dtype], 'dw dgrad*swish')


@pytest.mark.parametrize('cfg', [(2, 16, 16, 32, 3, 1, (1, 1)), (2, 17, 17, 96, 3, 2, (0, 1)), (2, 33, 21, 144, 3, 1, (1, 1)), (2, 32, 32, 96, 3, 2, (0, 1)),
                                 (3, 40, 24, 16, 3, 2, (0, 1)), (2, 31, 31, 480, 3, 2, (1, 1)), (2, 64, 48, 144, 3, 1, (1, 1)), (4, 128, 128, 32, 3, 1, (1, 1)),
                                 (2, 8, 8, 1152, 3, 2, (0, 1)),
                                 (2, 64, 64, 144, 5, 2, (1, 2)), (1, 65, 67, 40, 5, 2, (2, 2)), (2, 32, 32, 240, 5, 1, (2, 2)), (2, 40, 28, 48, 5, 1, (2, 2)),
                                 (1, 32, 33, 672, 5, 1, (2, 2)), (2, 64, 64, 96, 5, 2, (1, 2)), (1, 16, 16, 1152, 5, 1, (2, 2)), (2, 32, 32, 672, 5, 2, (1, 2))])
def test_dwconv_fused_data_and_weight_gradient(cfg):
    """effdet_dwconv_bwd (one pass over dz and the stored pre-activation) vs torch autograd of conv(swish(zprev)) -- models/efficientnet.py:85-88 --
    and vs the two separate kernels it replaces; two launches bitwise equal (slab reduction in a fixed order)."""
    from efficientdet.pytorch_amd import ops
    B, H, W, C, k, s, (plo, phi) = cfg
    g = torch.Generator().manual_seed(5)
    zp = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(C, 1, k, k, generator=g) * (2.0 / (k * k)) ** 0.5).requires_grad_(True)
    scale = 0.5 + torch.rand(C, generator=g)
    x = zp * torch.sigmoid(zp)
    z = F.conv2d(F.pad(x, [plo, phi, plo, phi]), w, None, s, 0, 1, C) * scale.view(1, -1, 1, 1)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    dev = 'cuda'
    wk = ops.dw_pack_weight(w.detach().to(dev))
    zpm, dzm = nhwc(zp.detach(), torch.float32), nhwc(dz, torch.float32)
    out = ops.dwconv_bwd(dzm, wk, scale.to(dev), zpm, k, s, plo, plo)
    if k == 5 and H * W < (1024 if s == 1 else 4096):
        assert out is None, 'k = 5 below 32 x 32 (stride 2: 64 x 64) stays on the two separate kernels'
        return
    assert out is not None, 'the fused kernel must serve every other fp32 geometry from 8 x 8 up'
    dxm, gk, dsum = out
    out2 = ops.dwconv_bwd(dzm, wk, scale.to(dev), zpm, k, s, plo, plo)
    assert torch.equal(out2[0].tensor(), dxm.tensor()) and torch.equal(out2[1], gk) and torch.equal(out2[2], dsum)
    dw = ops.dw_unpack_wgrad(gk, scale.to(dev), w.detach().to(dev), torch.empty(C, device=dev))
    assert_close(nchw(dxm), zp.grad, 2e-4, 'fused dw dgrad*swish\'')
    assert_close(dw.cpu(), w.grad, 1e-3, 'fused dw wgrad')
    assert_close(dsum.cpu(), dz.sum(dim=(0, 2, 3)), 1e-3, 'fused dw dsum')
    # the separate kernels (same arithmetic per element for the data gradient: bitwise; the weight gradient sums in another order)
    dx_sep = ops.dwconv_dgrad(dzm, wk, scale.to(dev), zpm, H, W, k, s, plo, plo)
    g_sep, ds_sep = ops.dwconv_wgrad(zpm, dzm, k, s, plo, plo, in_act=ops.ACT_SWISH)
    assert torch.equal(dx_sep.tensor(), dxm.tensor())
    assert_close_scale(gk.cpu(), g_sep.cpu(), 2e-5, 'fused vs separate dw wgrad'); assert_close_scale(dsum.cpu(), ds_sep.cpu(), 2e-5, 'fused vs separate dsum')


@pytest.mark.parametrize('cfg', [(2, 256, 256, 16, True), (2, 256, 256, 16, False), (8, 128, 128, 24, True), (3, 217, 203, 24, False), (8, 128, 128, 32, True)])
def test_expand_conv_fused_data_and_weight_gradient(cfg):
    """effdet_pw_bwd (both gradients of the MBConv expand conv from one read of the expanded gradient; models/efficientnet.py:82-84 backward) vs a
    float64 matmul, vs the two launches it replaces, through the unpack job that sums its slabs; two launches bitwise equal."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    B, H, W, Ci, skip = cfg
    Ce = 6 * Ci
    g = torch.Generator().manual_seed(11)
    dev = 'cuda'
    dz = torch.randn(B, H, W, Ce, generator=g).to(dev); x = torch.randn(B, H, W, Ci, generator=g).to(dev)
    we = (torch.randn(Ce, Ci, 1, 1, generator=g) / Ci ** 0.5).to(dev); s0 = (0.5 + torch.rand(Ce, generator=g)).to(dev)
    res = torch.randn(B, H, W, Ci, generator=g).to(dev) if skip else None
    out = ops.pw_bwd(Map.of(dz), Map.of(x), we, s0, Map.of(res) if skip else None)
    assert out is not None
    dx, slabs, parts = out
    out2 = ops.pw_bwd(Map.of(dz), Map.of(x), we, s0, Map.of(res) if skip else None)
    assert torch.equal(out2[0].tensor(), dx.tensor()) and torch.equal(out2[1], slabs) and torch.equal(out2[2], parts)
    mean, inv = torch.randn(Ce, device=dev), torch.rand(Ce, device=dev) + 0.5
    dw, dgam, dbeta = ops.unpack_wgrad_bn(slabs, we, s0, parts, mean, inv)
    torch.cuda.synchronize()
    dz2, x2 = dz.double().view(-1, Ce), x.double().view(-1, Ci)
    ws = (we.view(Ce, Ci) * s0.view(-1, 1)).double()                       # fl(w * s) as the kernel forms it, then exact
    dx_ref = dz2 @ ws + (res.double().view(-1, Ci) if skip else 0)
    G_ref = dz2.t() @ x2
    assert_close(dx.tensor().view(-1, Ci).cpu(), dx_ref.float().cpu(), 1e-4, 'fused expand dgrad')
    assert_close(dw.view(Ce, Ci).cpu(), (G_ref * s0.double().view(-1, 1)).float().cpu(), 1e-4, 'fused expand wgrad')
    assert_close(dbeta.cpu(), dz2.sum(0).float().cpu(), 1e-4, 'fused expand dsum')
    # the two separate launches
    G0, d0 = ops.conv2d_wgrad(Map.of(x), Map.of(dz), Cin=Ci, Cout=Ce, KH=1, KW=1)
    dw_s, dgam_s, dbeta_s = ops.unpack_wgrad_bn(G0, we, s0, d0, mean, inv)
    dx_s = Map.new(B, H, W, Ci, torch.float32, dev)
    ops.conv2d(Map.of(dz), ops.pack_weight(we, torch.float32, mode=1, scale=s0), dx_s, Cin=Ce, Cout=Ci, KH=1, KW=1,
               res=Map.of(res) if skip else None, res_mode=ops.RES_ADD if skip else ops.RES_NONE)
    assert_close_scale(dx.tensor().cpu(), dx_s.tensor().cpu(), 2e-6, 'fused vs separate expand dgrad')
    assert_close_scale(dw.cpu(), dw_s.cpu(), 2e-6, 'fused vs separate expand wgrad')
    assert_close_scale(dgam.cpu(), dgam_s.cpu(), 2e-5, 'fused vs separate dgamma'); assert_close_scale(dbeta.cpu(), dbeta_s.cpu(), 2e-6, 'fused vs separate dbeta')


@pytest.mark.parametrize('cfg', [(2, 256, 256, 16, 96, True), (8, 128, 128, 24, 96, True), (5, 128, 128, 24, 144, False), (17, 64, 64, 40, 144, True),
                                 (16, 64, 67, 40, 240, True), (3, 201, 117, 24, 144, True)])
def test_project_conv_data_gradient_with_se_epilogue(cfg):
    """effdet_pw_dgrad_se (project-conv data gradient + squeeze-excite backward + Swish' in one streaming MFMA kernel; models/efficientnet.py:89-104
    backward) vs float64, vs the implicit-GEMM / skinny launch it replaces; two launches bitwise equal."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    B, H, W, Co, Ce, use_rs = cfg
    g = torch.Generator().manual_seed(13)
    dev = 'cuda'
    dy = torch.randn(B, H, W, Co, generator=g).to(dev); zd = torch.randn(B, H, W, Ce, generator=g).to(dev)
    wp = (torch.randn(Co, Ce, 1, 1, generator=g) / Ce ** 0.5).to(dev); s2 = (0.5 + torch.rand(Co, generator=g)).to(dev)
    gate = torch.rand(B, Ce, generator=g).to(dev); dpool = (torch.randn(B, Ce, generator=g) * 1e-2).to(dev)
    rs = (0.5 + torch.rand(B, generator=g)).to(dev) if use_rs else None
    out = ops.pw_dgrad_se(Map.of(dy), wp, s2, rs, gate, dpool, Map.of(zd))
    assert out is not None
    out2 = ops.pw_dgrad_se(Map.of(dy), wp, s2, rs, gate, dpool, Map.of(zd))
    assert torch.equal(out.tensor(), out2.tensor())
    ws = (wp.view(Co, Ce) * s2.view(-1, 1)).double()
    v = (dy.double().view(B, -1, Co) @ ws) * (rs.double().view(B, 1, 1) if use_rs else 1.0)
    v = v * gate.double().view(B, 1, Ce) + dpool.double().view(B, 1, Ce)
    z = zd.double().view(B, -1, Ce); sg = torch.sigmoid(z)
    ref = (v * (sg * (1 + z * (1 - sg)))).float().view(B, H, W, Ce)
    assert_close(out.tensor().cpu(), ref.cpu(), 2e-4, 'project dgrad + SE epilogue')
    sep = Map.new(B, H, W, Ce, torch.float32, dev)
    ops.conv2d(Map.of(dy), ops.pack_weight(wp, torch.float32, mode=1, scale=s2), sep, Cin=Co, Cout=Ce, KH=1, KW=1, rowscale=rs,
               bc_scale=gate, bc_shift=dpool, res=Map.of(zd), res_mode=ops.RES_SWISH_GRAD)
    assert_close_scale(out.tensor().cpu(), sep.tensor().cpu(), 5e-6, 'streaming kernel vs the launch it replaces')


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cfg', [(2, 8, 8, 96, 4), (3, 5, 7, 240, 10), (2, 2, 2, 1152, 48), (2, 1, 1, 32, 8), (5, 3, 3, 144, 6), (2, 4, 4, 672, 28)])
def test_squeeze_excite_fwd_bwd(dtype, cfg):
    """pool -> gate MLP -> channel scale, and the whole backward chain, vs torch autograd."""
    from efficientdet.pytorch_amd import ops
    B, H, W, C, Cse = cfg
    g = torch.Generator().manual_seed(2)
    q = q_(dtype)
    z = q(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)     # depthwise pre-activation
    w1 = (torch.randn(Cse, C, 1, 1, generator=g) / C ** 0.5).requires_grad_(True)
    b1 = (torch.randn(Cse, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn(C, Cse, 1, 1, generator=g) / Cse ** 0.5).requires_grad_(True)
    b2 = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    x = z * torch.sigmoid(z)
    xq = x.detach()
    if dtype == torch.bfloat16:        # the kernels see the bf16-rounded activation; mirror with a straight-through round
        x = x + (q(xq) - xq)
    sq = F.adaptive_avg_pool2d(x, 1)
    mid = F.conv2d(sq, w1, b1)
    gate = torch.sigmoid(F.conv2d(mid * torch.sigmoid(mid), w2, b2))
    y = gate * x
    dy = q(torch.randn(y.shape, generator=g))
    y.backward(dy)
    dev = 'cuda'
    xm = nhwc(x.detach(), dtype); zm = nhwc(z.detach(), dtype)
    pool = x.detach().sum(dim=(2, 3)).to(dev)
    inv = 1.0 / (H * W)
    w1d, b1d, w2d, b2d = (t.detach().to(dev) for t in (w1, b1, w2, b2))
    gd, midd, pool_out = ops.se_gate_fwd(pool, w1d.view(Cse, C), b1d, w2d.view(C, Cse), b2d, inv, save_mid=True)
    assert torch.equal(pool_out, pool)                                   # G = 1: the pooled sum is passed through
    # the same pool handed over as 3 partial rows per image: summed in order inside the gate kernel
    parts = torch.stack([0.25 * pool, 0.5 * pool, 0.25 * pool], dim=1).contiguous()
    gd3, _, pool3 = ops.se_gate_fwd(parts, w1d.view(Cse, C), b1d, w2d.view(C, Cse), b2d, inv, save_mid=True)
    assert_close(pool3.cpu(), pool.cpu(), 1e-6, 'se pool from partial rows'); assert_close(gd3.cpu(), gd.cpu(), 1e-5, 'se gate from partial rows')
    ymm = ops.channel_scale(xm, gd)
    torch.cuda.synchronize()
    assert_close(gd.cpu(), gate.detach().view(B, C), 1e-4, 'se gate')
    assert_close(nchw(ymm), y.detach(), TOL[dtype], 'se scale')
    # the same two ops fed with the pre-activation (z-only storage): Swish is recomputed inside
    assert_close(nchw(ops.channel_scale(zm, gd, ops.ACT_SWISH)), y.detach(), TOL[dtype], 'se scale (from z)')
    dym = nhwc(dy, dtype)
    dgp = ops.se_dgate(dym, xm)                                          # [B][slabs][C] partial rows, no atomics
    assert torch.equal(dgp, ops.se_dgate(dym, xm))
    dg = dgp.sum(dim=1)
    assert_close(dg.cpu(), (dy * x.detach()).sum(dim=(2, 3)), 5 * TOL[dtype], 'se dgate')
    dg_z = ops.se_dgate(dym, zm, ops.ACT_SWISH).sum(dim=1)
    # bf16: Swish(z) here is unrounded while xm holds its bf16 rounding -> a 2^-9 perturbation of every term of a sum with
    # cancellation: compare at tensor scale
    assert_close_scale(dg_z.cpu(), dg.cpu(), 1e-4 if dtype == torch.float32 else 1e-2, 'se dgate (from z)')
    dpool, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgp, gd, midd, pool, w1d.view(Cse, C), b1d, w2d.view(C, Cse), inv)
    dzm = ops.se_bwd_apply(dym, gd, dpool, zm)
    # the same call inside a tail batch: the parameter gradients leave with effdet_backward_tail at the end of the block, same bits
    with ops.unpack_batch():
        dpool_t, dw1_t, db1_t, dw2_t, db2_t = ops.se_gate_bwd(dgp, gd, midd, pool, w1d.view(Cse, C), b1d, w2d.view(C, Cse), inv)
    torch.cuda.synchronize()
    for a, b in ((dpool, dpool_t), (dw1, dw1_t), (db1, db1_t), (dw2, dw2_t), (db2, db2_t)):
        assert torch.equal(a, b)
    t = 5 * TOL[dtype]
    assert_close(dw1.cpu(), w1.grad.view(Cse, C), t, 'se dw1'); assert_close(db1.cpu(), b1.grad, t, 'se db1')
    assert_close(dw2.cpu(), w2.grad.view(C, Cse), t, 'se dw2'); assert_close(db2.cpu(), b2.grad, t, 'se db2')
    assert_close(nchw(dzm), z.grad, t, 'se dz')


def test_bn_fold_and_param_grad():
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(3)
    C = 100
    gamma = (0.5 + torch.rand(C, generator=g)).requires_grad_(True); beta = torch.randn(C, generator=g).requires_grad_(True)
    mean = torch.randn(C, generator=g); var = 0.5 + torch.rand(C, generator=g)
    sc, sh, inv = ops.bn_fold(*(t.detach().cuda() for t in (gamma, beta, mean, var)))
    invr = 1 / torch.sqrt(var + 1e-3)
    assert_close(sc.cpu(), gamma.detach() * invr, 1e-6, 'scale'); assert_close(sh.cpu(), beta.detach() - mean * gamma.detach() * invr, 1e-5, 'shift')
    # y = bn(c), c = conv out [M, C]; dgamma = sum dz*(c-mean)*invstd with sum_m dz*c given as wsum
    c = torch.randn(50, C, generator=g); dz = torch.randn(50, C, generator=g)
    y = (c - mean) * invr * gamma + beta
    y.backward(dz)
    wsum = (dz * c).sum(0).cuda(); dsum = dz.sum(0).cuda()
    dg, db = ops.bn_param_grad(wsum, dsum, mean.cuda(), inv)
    assert_close(dg.cpu(), gamma.grad, 1e-4, 'dgamma'); assert_close(db.cpu(), beta.grad, 1e-5, 'dbeta')


@pytest.mark.parametrize('dtype', DT)
def test_elementwise_helpers(dtype):
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(4)
    q = q_(dtype)
    B, H, W, C = 3, 5, 6, 40
    dy = q(torch.randn(B, C, H, W, generator=g)); aux = q(torch.randn(B, C, H, W, generator=g)); rs = torch.rand(B, generator=g) + 0.5
    out = ops.act_bwd(nhwc(dy, dtype), nhwc(aux, dtype), ops.ACT_RELU)
    assert_close(nchw(out), dy * (aux > 0), TOL[dtype], 'relu bwd')
    sg = torch.sigmoid(aux)
    out = ops.act_bwd(nhwc(dy, dtype), nhwc(aux, dtype), ops.ACT_SWISH, rowscale=rs.cuda())
    assert_close(nchw(out), dy * sg * (1 + aux * (1 - sg)) * rs.view(-1, 1, 1, 1), TOL[dtype], 'swish bwd')
    out = ops.act_bwd(nhwc(dy, dtype), None, ops.ACT_NONE, rowscale=rs.cuda())
    assert_close(nchw(out), dy * rs.view(-1, 1, 1, 1), TOL[dtype], 'rowscale bwd')
    a = nhwc(dy, dtype); ops.add_inplace(a, nhwc(aux, dtype))
    assert_close(nchw(a), q(dy + aux), TOL[dtype], 'add')
    cs = torch.zeros(C, device='cuda')
    ops.colsum(nhwc(dy, dtype).tensor().view(-1, C), cs)
    assert_close(cs.cpu(), dy.sum(dim=(0, 2, 3)), 5 * TOL[dtype], 'colsum')
    x = torch.randn(2, 3, 9, 7, generator=g)
    m = ops.nchw_to_nhwc(x.cuda(), dtype, cpad=8)
    assert_close(m.tensor()[..., :3].float().cpu().permute(0, 3, 1, 2), q(x), 1e-6, 'nchw->nhwc')
    assert float(m.tensor()[..., 3:].float().abs().max()) == 0.0
    assert_close(ops.nhwc_to_nchw(nhwc(x, dtype)).cpu(), q(x), 1e-6, 'nhwc->nchw')


@pytest.mark.parametrize('B,C,Cse', [(3, 96, 4), (8, 2688, 112), (16, 672, 28), (20, 144, 6), (40, 1152, 48)])
def test_se_gate_both_forms(B, C, Cse):
    """Small batches take the 8-workgroups-per-image, launch-per-layer form (Cse >= 8), the rest one workgroup per image:
    both == the torch MLP, incl. the saved pre-Swish squeeze activations."""
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C)
    pool = torch.randn(B, C, generator=g) * 50.0
    w1 = torch.randn(Cse, C, generator=g) / C ** 0.5; b1 = torch.randn(Cse, generator=g) * 0.1
    w2 = torch.randn(C, Cse, generator=g) / Cse ** 0.5; b2 = torch.randn(C, generator=g) * 0.1
    inv = 1.0 / 49.0
    mid = (pool * inv) @ w1.t() + b1
    gate = torch.sigmoid((mid * torch.sigmoid(mid)) @ w2.t() + b2)
    gd, midd, _ = ops.se_gate_fwd(pool.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), inv, save_mid=True)
    torch.cuda.synchronize()
    assert_close(midd.cpu(), mid, 1e-4, 'se mid'); assert_close(gd.cpu(), gate, 1e-4, 'se gate')


@pytest.mark.parametrize('cfg', [(2, 32, 32, 16, 3, 2, (0, 1)), (2, 32, 24, 24, 3, 1, (1, 1)), (1, 33, 31, 24, 5, 2, (1, 2)),
                                 (2, 16, 40, 40, 5, 1, (2, 2)), (3, 17, 9, 32, 3, 1, (1, 1)), (2, 64, 64, 16, 3, 2, (0, 1)),
                                 (1, 48, 48, 40, 3, 2, (0, 1))])
def test_fused_expand_depthwise_forward(cfg):
    """effdet_mbconv_expand_dw_fwd (inference): expand 1x1 + BN + Swish -> depthwise k x k (TF-'same' pad of the EXPANDED map: the
    expand of a padding pixel is swish(shift), not zero -- the case a fused loader gets wrong first) + BN + Swish, and the
    squeeze-excite partial sums, vs torch on the same inputs (models/efficientnet.py:82-88); odd sizes, every (k, stride, Cin)."""
    from efficientdet.pytorch_amd import ops
    B, H, W, Cin, k, s, (plo, phi) = cfg
    Cexp = 6 * Cin
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Cin, H, W, generator=g)
    we = torch.randn(Cexp, Cin, 1, 1, generator=g) / Cin ** 0.5
    s0 = 0.5 + torch.rand(Cexp, generator=g); t0 = torch.randn(Cexp, generator=g) * 0.5        # large shifts: padding pixels would show
    wd = torch.randn(Cexp, 1, k, k, generator=g) * (2.0 / (k * k)) ** 0.5
    s1 = 0.5 + torch.rand(Cexp, generator=g); t1 = torch.randn(Cexp, generator=g) * 0.2
    e = F.conv2d(x, we) * s0.view(1, -1, 1, 1) + t0.view(1, -1, 1, 1)
    e = e * torch.sigmoid(e)
    z = F.conv2d(F.pad(e, [plo, phi, plo, phi]), wd, None, s, 0, 1, Cexp) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1)
    y = z * torch.sigmoid(z)
    Ho, Wo = y.shape[2:]
    dev = 'cuda'
    xm = nhwc(x, torch.float32)
    wk = ops.dw_pack_weight(wd.to(dev))
    ym, pp = ops.expand_dw_fwd(xm, we.to(dev), s0.to(dev), t0.to(dev), wk, s1.to(dev), t1.to(dev), k, s, plo, plo, Ho, Wo)
    torch.cuda.synchronize()
    assert_close(nchw(ym), y, TOL[torch.float32], 'fused expand + depthwise y %s' % (cfg,))
    assert pp.shape[0] == B and pp.shape[2] == Cexp
    assert_close(pp.sum(dim=1).cpu(), y.sum(dim=(2, 3)), 5 * TOL[torch.float32], 'se pool')
    ym2, pp2 = ops.expand_dw_fwd(xm, we.to(dev), s0.to(dev), t0.to(dev), wk, s1.to(dev), t1.to(dev), k, s, plo, plo, Ho, Wo)
    assert torch.equal(ym2.tensor(), ym.tensor()) and torch.equal(pp2, pp)                       # bitwise reproducible
