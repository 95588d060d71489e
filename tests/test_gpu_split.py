"""GPU parity of the SPLIT activation layout (EFFDET_F32_SPLIT: every 32 channels as [32 x bf16 hi | 32 x bf16 lo], 4 bytes per
element) that the RetinaHead uses in the bf16x3 arithmetic: conversion, implicit-GEMM conv with split input / split or fp32
output / ReLU-mask residual, the three-product weight-gradient kernel, and the loss kernels' split gradient rows -- each against
torch-CPU fp32 on the same seeded inputs (gate 1e-3 element-relative with the usual floor: products carry ~16 mantissa bits,
measured error ~4e-6 of scale)."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close, assert_close_scale

pytestmark = pytest.mark.gpu
TOL = 1e-3


def ops_split_ok(S):
    from efficientdet.pytorch_amd import ops
    return ops.wgrad_split_supported(2, [((S // 8) >> i, (S // 8) >> i) for i in range(5)], 256, 256, 256)


def from_split(t):
    """[..., C] tensor holding the split layout -> fp32 values (hi + lo), on the host."""
    C = t.shape[-1]
    raw = t.detach().cpu().contiguous().view(torch.bfloat16).view(-1, C // 32, 2, 32).float()
    return (raw[:, :, 0] + raw[:, :, 1]).reshape(t.shape)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def test_to_split_roundtrip_and_precision():
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 96, generator=g) * torch.logspace(-6, 6, 96).view(1, 1, 1, -1)
    s = ops.to_split(x.cuda())
    back = from_split(s)
    assert float(((back - x).abs() / x.abs().clamp_min(1e-30)).max()) <= 2.0 ** -15      # hi + lo carries >= 16 mantissa bits
    hi = s.cpu().view(torch.bfloat16).view(-1, 3, 2, 32)[:, :, 0].float().reshape(x.shape)
    assert torch.equal(hi, x.bfloat16().float())                                          # hi = round-to-nearest-even bf16


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, Cout, act, res(relu mask), out_f32, rowscale
    (2, 16, 16, 64, 256, 1, False, False, False),          # tower layer 0
    (1, 8, 8, 256, 256, 1, False, False, False),           # tower
    (2, 8, 8, 256, 256, 0, True, False, True),             # tower data gradient: ReLU mask of the forward activation + row scale
    (1, 8, 8, 256, 720, 3, False, True, False),            # retina_cls + sigmoid, plain fp32 out
    (1, 8, 8, 256, 36, 0, False, True, False),             # retina_reg
    (2, 4, 4, 768, 256, 0, True, False, False),            # d(cls logits) data gradient (padded 768-channel rows)
    (3, 8, 16, 256, 64, 0, False, True, False),            # tower-0 data gradient back to the neck (64-wide tile, fp32 out)
    (1, 1, 1, 64, 256, 1, False, False, False),            # 1x1 map
    (2, 2, 128, 64, 128, 1, False, False, False),          # partial tiles, H != W
    (3, 24, 24, 256, 256, 1, False, False, False),         # several 256-pixel tiles incl. a partial one (persistent form: tile walk)
    (2, 20, 12, 128, 320, 0, True, False, True),           # partial channel tile (320 = 256 + 64), ReLU mask + row scale
])
@pytest.mark.parametrize('pers', [0, 1])
def test_conv_split_layout(cfg, pers):
    """pers = 1: eligible shapes (Cout >= 192, K >= 1152) through the persistent 256 x 256 / 32x32x16 form of the kernel."""
    from efficientdet.pytorch_amd import ops, _lib as L
    from efficientdet.pytorch_amd.ops import Map
    B, H, W, Cin, Cout, act, res, out_f32, rowscale = cfg
    if pers and not (Cout >= 192 and 9 * Cin >= 1152):
        pytest.skip('shape not served by the persistent form')
    old = (ops.tuning_set(L.TUNE_SPLIT_PERS, pers), ops.tuning_set(L.TUNE_IGEMM_BIG_MIN_M, 0), ops.tuning_set(L.TUNE_SPLIT_KORD, 0))
    try:
        for kord in (0, 1):            # K walk: tap-major / channel-group-major
            ops.tuning_set(L.TUNE_SPLIT_KORD, kord)
            _conv_split_case(cfg, pers)
    finally:
        ops.tuning_set(L.TUNE_SPLIT_PERS, old[0]); ops.tuning_set(L.TUNE_IGEMM_BIG_MIN_M, old[1]); ops.tuning_set(L.TUNE_SPLIT_KORD, old[2])


def _conv_split_case(cfg, pers):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    B, H, W, Cin, Cout, act, res, out_f32, rowscale = cfg
    g = torch.Generator().manual_seed(sum(cfg[:5]))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    ref = F.conv2d(x, w, shift, padding=1)
    if act == 1: ref = F.relu(ref)
    elif act == 3: ref = torch.sigmoid(ref)
    rs = None
    if rowscale:
        rs = torch.rand(B, generator=g) + 0.5
        ref = ref * rs.view(-1, 1, 1, 1)
    r = None
    if res:
        r = F.relu(torch.randn(B, Cout, H, W, generator=g))
        r[0, :, 0, 0] = 0.0                                    # exact zeros must mask (hi == 0)
        ref = torch.where(r > 0, ref, torch.zeros_like(ref))
    ops.set_f32_arith('bf16x3')
    try:
        xm = Map.of(ops.to_split(_nhwc(x)))
        wp = ops.pack_weight(w.cuda(), torch.float32, x3=True)
        ym = Map.new(B, H, W, Cout, torch.float32, 'cuda')
        rm = Map.of(ops.to_split(_nhwc(r))) if res else None
        kw = dict(Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=shift.cuda(), act=act, res=rm,
                  res_mode=ops.RES_RELU_MASK if res else ops.RES_NONE, rowscale=rs.cuda() if rowscale else None,
                  out_f32=out_f32, split=True)
        if pers:                       # the launch really is the persistent kernel
            ops.PROFILE = ops.LaunchProfile()
        try:
            ops.conv2d(xm, wp, ym, **kw)
            torch.cuda.synchronize()
            if pers:
                assert ops.PROFILE.records[0][0] == 'conv_igemm_pers_kernel<split,bf16x3>', ops.PROFILE.records[0][0]
        finally:
            ops.PROFILE = None
    finally:
        ops.set_f32_arith('f32')
    got = ym.tensor().cpu() if out_f32 else from_split(ym.tensor())
    assert_close(got.permute(0, 3, 1, 2), ref, TOL, 'split conv %s' % (cfg,))


def test_conv_split_grouped_levels_and_accumulate():
    """Five pyramid levels in ONE launch through flat level-major buffers (the head's layout), then the plain-fp32 output form with
    RES_ADD (second tower accumulating onto the first one's data gradient)."""
    from efficientdet.pytorch_amd import ops, functional as Fn
    g = torch.Generator().manual_seed(5)
    B, sizes, Cin, Cout = 2, [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)], 256, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xs = [torch.randn(B, Cin, h, ww, generator=g) for (h, ww) in sizes]
    base = [torch.randn(B, Cout, h, ww, generator=g) for (h, ww) in sizes]
    _, xmaps = Fn.pyramid_alloc(B, sizes, Cin, torch.float32, 'cuda')
    _, ymaps = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, 'cuda')
    for x, m, y0, ym in zip(xs, xmaps, base, ymaps):
        Fn.level_tensor(m).copy_(ops.to_split(_nhwc(x)))
        Fn.level_tensor(ym).copy_(_nhwc(y0))
    ops.set_f32_arith('bf16x3')
    try:
        ops.conv2d(xmaps, ops.pack_weight(w.cuda(), torch.float32, x3=True), ymaps, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1,
                   res=ymaps, res_mode=ops.RES_ADD, split=True, out_f32=True)
        torch.cuda.synchronize()
    finally:
        ops.set_f32_arith('f32')
    for x, y0, ym in zip(xs, base, ymaps):
        assert_close(Fn.level_tensor(ym).cpu().permute(0, 3, 1, 2), F.conv2d(x, w, None, padding=1) + y0, TOL, 'level %s' % (tuple(x.shape),))


@pytest.mark.parametrize('cfg', [
    # B, sizes, Cin, Cout, lddz
    (2, [(16, 16), (8, 8), (4, 4)], 256, 256, 256),        # tower weight gradient, three levels in one launch
    (2, [(16, 16), (8, 8), (4, 4), (2, 4)], 64, 256, 256), # first tower layer (64-channel pyramid), incl. an 8-pixel level
    (1, [(8, 8)], 256, 36, 64),                            # retina_reg: 36 output channels in a 64-channel padded row
    (1, [(8, 16)], 256, 720, 768),                         # retina_cls: 720 in 768
])
def test_wgrad_split_layout(cfg):
    from efficientdet.pytorch_amd import ops, functional as Fn
    B, sizes, Cin, Cout, lddz = cfg
    g = torch.Generator().manual_seed(Cin + Cout)
    xs = [torch.randn(B, Cin, h, w, generator=g) for (h, w) in sizes]
    dzs = [torch.randn(B, Cout, h, w, generator=g) * (torch.rand(B, Cout, h, w, generator=g) > 0.5) for (h, w) in sizes]
    wt = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    bt = torch.zeros(Cout, requires_grad=True)
    for x, dz in zip(xs, dzs):
        (F.conv2d(x, wt, bt, padding=1) * dz).sum().backward()
    assert ops.wgrad_split_supported(B, sizes, Cin, Cout, lddz)
    _, xmaps = Fn.pyramid_alloc(B, sizes, Cin, torch.float32, 'cuda')
    _, zmaps = Fn.pyramid_alloc(B, sizes, lddz, torch.float32, 'cuda')
    for x, m, dz, zm in zip(xs, xmaps, dzs, zmaps):
        Fn.level_tensor(m).copy_(ops.to_split(_nhwc(x)))
        pad = torch.zeros(B, dz.shape[2], dz.shape[3], lddz)
        pad[..., :Cout] = dz.permute(0, 2, 3, 1)
        Fn.level_tensor(zm).copy_(ops.to_split(pad.cuda()))
    G, dbp = ops.conv2d_wgrad(xmaps, zmaps, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, split=True)
    dw = torch.empty(Cout, Cin, 3, 3, device='cuda')
    db = ops.unpack_wgrad(G, dw, dbias_part=dbp)
    G2, dbp2 = ops.conv2d_wgrad(xmaps, zmaps, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, split=True)
    torch.cuda.synchronize()
    assert torch.equal(G, G2) and torch.equal(dbp, dbp2)                          # bitwise reproducible
    assert_close_scale(dw.cpu(), wt.grad, 1e-4, 'split wgrad dw %s' % (cfg,))     # sums of thousands of products: tensor scale
    assert_close_scale(db.cpu(), bt.grad, 1e-4, 'split wgrad db')


def test_wgrad_split_refuses_levels_it_cannot_take():
    from efficientdet.pytorch_amd import ops
    assert not ops.wgrad_split_supported(2, [(16, 32), (1, 2)], 256, 256, 256)    # a 1 x 2 level has no whole 8-pixel runs
    assert ops.wgrad_split_supported(32, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 256, 256, 256)


def test_loss_kernels_write_the_split_layout():
    """focal_loss_fwd_grad / focal_loss_bwd_reg with split outputs == their plain-fp32 outputs (to the layout's 2^-16)."""
    from efficientdet.pytorch_amd import ops
    from oracle import effdet_oracle as O
    B, S, nc = 2, 128, 8
    anc = ops.anchors(S, S, 'cuda')
    A = anc.shape[1]
    g = torch.Generator().manual_seed(3)
    cls = torch.rand(B, A, nc, generator=g).clamp(1e-3, 1 - 1e-3).cuda()
    reg = (torch.randn(B, A, 4, generator=g) * 0.3).cuda()
    ann = O.synthetic_batch(B, S, seed=2, num_classes=nc)[1].cuda()
    dld = (9 * nc + 63) // 64 * 64
    l0, ws0, d0 = ops.focal_loss_fwd_grad(cls, reg, anc, ann, torch.float32, dld)
    l1, ws1, d1 = ops.focal_loss_fwd_grad(cls, reg, anc, ann, torch.float32, dld, split=True)
    assert torch.equal(l0, l1)
    assert_close_scale(from_split(d1), d0.cpu(), 2.0 ** -15, 'dcls split')
    gs = torch.tensor([0.7, 1.3], device='cuda')
    r0 = ops.focal_loss_bwd_reg(reg, anc, ann, gs, ws0, torch.float32)                       # [B, A, 4]
    r1 = ops.focal_loss_bwd_reg(reg, anc, ann, gs, ws0, torch.float32, reg_ld=64, split=True)   # [B, A/9, 64] split
    r2 = ops.focal_loss_bwd_reg(reg, anc, ann, gs, ws0, torch.float32, reg_ld=40)            # [B, A/9, 40] plain, pad zeroed
    torch.cuda.synchronize()
    want = r0.cpu().view(B, A // 9, 36)
    assert torch.equal(r2.cpu()[..., :36], want) and not bool(r2[..., 36:].any())
    back = from_split(r1)
    assert_close_scale(back[..., :36], want, 2.0 ** -15, 'dreg split')
    assert not bool(back[..., 36:].any())


@pytest.mark.parametrize('S', [512, 1024])       # (pyramids whose coarsest level still has whole 8-pixel runs: >= 4 x 4, i.e. inputs >= 512)
def test_model_with_split_head_matches_plain_head(S):
    """Whole model, bf16x3 arithmetic: the split-layout head == the plain-fp32-storage head (register splits) on losses and all
    274 gradients -- the two differ only in WHEN the same hi / lo values are formed."""
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, functional as Fn
    from oracle import effdet_oracle as O
    net, nc = 'efficientdet-d0', 20
    c = EFFICIENTDET[net]
    img, ann = O.synthetic_batch(2, S, seed=3, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    res = {}
    old = Fn.HEAD_SPLIT
    assert Fn.HEAD_SPLIT and ops_split_ok(S)
    try:
        for flag in (True, False):
            Fn.HEAD_SPLIT = flag
            m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'],
                             compute_dtype=torch.float32, f32_arith='bf16x3')
            m.load_state_dict(O.make_state_dict(net, nc, seed=0)); m.backbone.drop_connect_rate = 0.0
            m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
            cl, rl = m([img, ann])
            (cl.mean() + rl.mean()).backward()
            torch.cuda.synchronize()
            res[flag] = (float(cl), float(rl), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
            m.eval(); m.is_training = False
            with torch.no_grad():
                res[flag] += (m.forward_raw(img)[:2],)
    finally:
        Fn.HEAD_SPLIT = old
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 1e-4 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-4 * abs(b[1]), (a[:2], b[:2])
    assert_close_scale(a[3][0].cpu(), b[3][0].cpu(), 1e-4, 'cls'); assert_close_scale(a[3][1].cpu(), b[3][1].cpu(), 1e-4, 'reg')
    for k in b[2]:
        d = float((a[2][k].double() - b[2][k].double()).norm()); n = float(b[2][k].double().norm())
        assert d <= 2e-3 * n + 1e-12, (k, d, n)


def test_head_towers_on_two_streams_compute_the_same_bits():
    """functional.HEAD_TWO_STREAMS (A/B knob, off: measured 37.95 vs 37.62 ms per D0 B = 32 step -- the other tower's workgroups do not fill a
    launch's draining tail): the regression tower forked onto a side stream and joined before the towers' sum must give the same losses
    and the same gradients, bit for bit, in the headline arithmetic."""
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, functional as Fn
    from oracle import effdet_oracle as O
    net, nc = 'efficientdet-d0', 8
    c = EFFICIENTDET[net]
    sd = O.make_state_dict(net, nc, seed=3)
    img, ann = O.synthetic_batch(2, 512, seed=2, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    outs = []
    old = Fn.HEAD_TWO_STREAMS
    try:
        for two in (False, True):
            Fn.HEAD_TWO_STREAMS = two
            m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32,
                             f32_arith='f32_bwd_bf16x3')
            m.load_state_dict(sd); m.backbone.drop_connect_rate = 0.0
            m = m.cuda(); m.train(); m.is_training = True; m.freeze_bn()
            cl, rl = m([img, ann])
            (cl.mean() + rl.mean()).backward()
            torch.cuda.synchronize()
            outs.append((cl.detach().clone(), rl.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        Fn.HEAD_TWO_STREAMS = old
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[2].keys() == b[2].keys() and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
