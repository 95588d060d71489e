"""GPU parity of the f16x3 forward arithmetic (EFFDET_F32_HSPLIT): activations as [32 x f16 hi | 32 x f16 lo * 2^11] per 32 channels,
weights as row-scaled f16 hi | lo pairs, 3 x v_mfma_f32_16x16x32_f16 per product (fp32 accumulate) -- the RetinaHead's forward
convs (models/retinahead.py:109-129) in the headline mode.  The claim under test is "fp32-equivalent": against a float64 convolution
of the same fp32 inputs the f16x3 result may be at most 2x as far away as the exact-fp32 MFMA kernel's (measured ~1.0-1.2x), over eight
orders of magnitude of activation scale and with weight rows of very different magnitudes.  Layout conversions are checked bit for bit
against their definition."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _clear_range_watch():
    """Several cases here feed values beyond fp16's range on purpose: leave the device's watch word clean for whoever runs next."""
    yield
    from efficientdet.pytorch_amd import ops
    for t in ops._range_flags.values():
        t.zero_()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def from_hsplit(t):
    """[..., C] tensor holding the H-split layout -> (values hi + lo' / 2048 in float64, hi, lo') on the host."""
    C = t.shape[-1]
    raw = t.detach().cpu().contiguous().view(torch.float16).view(-1, C // 32, 2, 32)
    hi, lo = raw[:, :, 0].double().reshape(t.shape), raw[:, :, 1].double().reshape(t.shape)
    return hi + lo / 2048.0, hi, lo


def from_split(t):
    C = t.shape[-1]
    raw = t.detach().cpu().contiguous().view(torch.bfloat16).view(-1, C // 32, 2, 32).float()
    return (raw[:, :, 0] + raw[:, :, 1]).reshape(t.shape)


def to_split2(x, bf=True):
    from efficientdet.pytorch_amd import ops
    L, C = ops.L, ops.C
    s = torch.empty_like(x) if bf else None
    h = torch.empty_like(x)
    L.check(L.lib().effdet_to_split2(L.ptr(x), L.ptr(s), L.ptr(h), C.c_longlong(x.numel()), L.ptr(ops.range_flag(x.device)), L.stream_ptr()), 'effdet_to_split2')
    return s, h


def test_to_split2_is_the_definition():
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 96, generator=g) * torch.logspace(-9, 4.5, 96).view(1, 1, 1, -1)
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 2.0 ** -14, 2.0 ** -14 * 0.999, 65504.0, 6.0e-8, 1.0, -3.0e-5])
    s, h = to_split2(x.cuda())
    assert torch.equal(s.cpu().view(torch.int32), ops.to_split(x.cuda()).cpu().view(torch.int32))       # the bf16 half of the pass == effdet_to_split
    val, hi, lo = from_hsplit(h)
    # hi = RNE fp16 (denormals kept: the fp16 MFMA of gfx950 honours them); lo' = RNE fp16 of the exact remainder * 2^11
    hi_ref = x.half()
    assert torch.equal(hi.half(), hi_ref)
    lo_ref = ((x.double() - hi_ref.double()) * 2048.0).float().half()
    assert torch.equal(lo.half(), lo_ref)                                    # (as values: the sign of a zero remainder is not part of the format)
    # 22 significand bits wherever hi is a normal fp16 number; below that an absolute error of at most 2^-36 (both halves denormal-spaced)
    normal = (x.abs() >= 2.0 ** -14) & (x.abs() < 65504)
    rel = ((val - x.double()).abs() / x.double().abs().clamp_min(1e-300))
    assert float(rel[normal].max()) <= 2.0 ** -21.9
    assert float((val - x.double()).abs()[x.abs() < 2.0 ** -14].max()) <= 2.0 ** -36
    # out of range is LOUD: inf, not a clamped value
    big = from_hsplit(to_split2(torch.full((1, 1, 1, 32), 7.0e4).cuda(), bf=False)[1])[0]
    assert not bool(torch.isfinite(big).any())


def unpack_h3(wp, Cout, K):
    """pack_weight(..., h3=True) buffer -> (hi, lo as float64 [Cout][K], inverse row scales [Cout])."""
    raw = wp.detach().cpu().contiguous()
    body = raw[:Cout * K].view(torch.float16).view(Cout, K // 32, 2, 32).double()
    inv = raw[Cout * K:].double()
    return (body[:, :, 0].reshape(Cout, K), body[:, :, 1].reshape(Cout, K), inv)


@pytest.mark.parametrize('prep', [False, True])
def test_pack_weight_h3_rows(prep):
    """Row-scaled fp16 hi | lo weights, standalone launch and as a job of the batched per-step preparation."""
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(1)
    Cout, Cin = 72, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * torch.logspace(-8, 3, Cout).view(-1, 1, 1, 1)
    w[5] = 0.0                                                                               # an all-zero row: S = 1, no NaN
    w[6, :, :, :] *= torch.logspace(0, -7, Cin).view(-1, 1, 1)                               # one row spanning 7 decades
    wd = w.cuda()
    if prep:
        P = ops.ParamPrep('f32')
        ops.set_prep(P)
        try:
            P.begin_step(wd.device)
            ops.pack_weight(wd, torch.float32, h3=True)          # records
            P.begin_step(wd.device)                              # builds + replays the table
            wp = ops.pack_weight(wd, torch.float32, h3=True).clone()
            assert P.replay
        finally:
            ops.set_prep(None)
    else:
        wp = ops.pack_weight(wd, torch.float32, h3=True)
    torch.cuda.synchronize()
    K = 9 * Cin
    hi, lo, inv = unpack_h3(wp, Cout, K)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, K).double()            # [Cout][tap][Cin]
    S = 1.0 / inv
    m = wk.abs().amax(dim=1)
    nz = m > 0
    assert bool(((m * S)[nz] >= 2.0 ** 14).all()) and bool(((m * S)[nz] < 2.0 ** 15).all())
    assert torch.equal(torch.log2(S), torch.log2(S).round())                                 # powers of two
    assert float(S[5]) == 1.0 and float(hi[5].abs().max()) == 0.0
    ws = wk * S.view(-1, 1)
    assert torch.equal(hi.half(), ws.float().half())                                         # RNE fp16 of the scaled value
    assert torch.equal(lo.half(), (ws - hi).float().half())
    big = ws.abs() >= 2.0 ** -3                                                              # (lo is a normal fp16 number there)
    assert float((((hi + lo) - ws).abs() / ws.abs().clamp_min(1e-300))[big].max()) <= 2.0 ** -21.9
    assert float(((hi + lo) - ws).abs()[~big].max()) <= 2.0 ** -25                           # absolute floor elsewhere: 2^-39 of the row maximum


def _conv_ref64(x, w, b, act):
    ref = F.conv2d(x.double(), w.double(), b.double() if b is not None else None, padding=1)
    if act == 1: ref = F.relu(ref)
    elif act == 3: ref = torch.sigmoid(ref)
    return ref


def _run_h(x, w, shift, act, out_f32, ysplit=False, kord=None):
    from efficientdet.pytorch_amd import ops, _lib as L
    from efficientdet.pytorch_amd.ops import Map
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    old = ops.tuning_set(L.TUNE_SPLIT_KORD, kord) if kord is not None else None
    try:
        xm = Map.of(to_split2(_nhwc(x), bf=False)[1])
        wp = ops.pack_weight(w.cuda(), torch.float32, h3=True)
        ym = Map.new(B, H, W, Cout, torch.float32, 'cuda')
        ys = Map.new(B, H, W, Cout, torch.float32, 'cuda') if ysplit else None
        ops.conv2d(xm, wp, ym, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=shift.cuda() if shift is not None else None,
                   act=act, out_f32=out_f32, hsplit=True, ysplit=ys)
        torch.cuda.synchronize()
    finally:
        if old is not None:
            ops.tuning_set(L.TUNE_SPLIT_KORD, old)
    return ym.tensor(), (ys.tensor() if ysplit else None)


def _run_exact(x, w, shift, act):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    ops.set_f32_arith('f32')
    ym = Map.new(B, H, W, Cout, torch.float32, 'cuda')
    ops.conv2d(Map.of(_nhwc(x)), ops.pack_weight(w.cuda(), torch.float32), ym, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1,
               shift=shift.cuda() if shift is not None else None, act=act)
    torch.cuda.synchronize()
    return ym.tensor().cpu()


def _err(got, ref64):
    return float((got.double() - ref64).abs().max() / ref64.abs().max())


CASES = [
    # B, H, W, Cin, Cout, act, out_f32
    (2, 16, 16, 64, 256, 1, False),           # tower layer 0 (D0: 64-channel pyramid)
    (1, 8, 8, 256, 256, 1, False),            # tower
    (1, 8, 8, 256, 720, 3, True),             # retina_cls + sigmoid, plain fp32 out
    (1, 8, 8, 256, 36, 0, True),              # retina_reg (64-wide tile)
    (1, 1, 1, 64, 256, 1, False),             # 1x1 map
    (2, 2, 128, 64, 128, 1, False),           # partial tiles, H != W
    (3, 24, 24, 256, 256, 1, False),          # many tiles incl. a partial one
    (2, 20, 12, 224, 320, 0, False),          # D4's 224-channel pyramid, partial channel tile (320 = 256 + 64)
    (1, 12, 12, 32, 64, 1, False),            # the shortest K the pack accepts (288), 64-wide tile with an H-split output
]


@pytest.mark.parametrize('cfg', CASES)
@pytest.mark.parametrize('kord', [0, 1])
def test_conv_f16x3_is_fp32_equivalent(cfg, kord):
    B, H, W, Cin, Cout, act, out_f32 = cfg
    g = torch.Generator().manual_seed(sum(cfg[:5]) + 7)
    x = F.relu(torch.randn(B, Cin, H, W, generator=g)) * 1.7
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    ref = _conv_ref64(x, w, shift, act)
    y, ys = _run_h(x, w, shift, act, out_f32, ysplit=not out_f32, kord=kord)
    got = y.cpu().double() if out_f32 else from_hsplit(y)[0]
    e_h = _err(got.permute(0, 3, 1, 2), ref)
    e_x = _err(_run_exact(x, w, shift, act).permute(0, 3, 1, 2), ref)
    # the H-split OUTPUT carries 22 bits (2^-22 = 2.4e-7 of an element, <= that of the tensor scale): allow it on top
    assert e_h <= 2.0 * e_x + (0.0 if out_f32 else 2.5e-7), (cfg, kord, e_h, e_x)
    assert e_h <= 3e-6, (cfg, e_h)                               # of the tensor's scale: what an fp32 accumulation over K <= 2304 leaves (measured 0.3-1.4e-6)
    assert_close(got.permute(0, 3, 1, 2).float(), ref.float(), 1e-4, 'f16x3 conv %s' % (cfg,))
    if ys is not None:
        # the second output: the SAME fp32 values in the bf16 split layout (16 bits), zeros of the ReLU exactly zero in both
        vb = from_split(ys).double()
        assert float((vb - got).abs().max()) <= 2.0 ** -15 * float(got.abs().max())
        assert torch.equal(vb == 0, got == 0)


@pytest.mark.parametrize('xscale', [1e-5, 1e-3, 1.0, 1e2, 3e3])
def test_conv_f16x3_over_activation_scales(xscale):
    """The scaled lo half keeps 22 bits over fp16's whole normal range and the matrix pipe honours a denormal hi: the same gate from 1e-5
    to 3e3 typical activation magnitude (an unscaled lo half would be a denormal for every |x| < 0.125).  Weight rows differ by 12 orders of
    magnitude (row scales)."""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 2, 8, 8, 256, 128
    x = F.relu(torch.randn(B, Cin, H, W, generator=g)) * xscale
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5 * torch.logspace(-6, 6, Cout).view(-1, 1, 1, 1)
    ref = _conv_ref64(x, w, None, 0)
    y, _ = _run_h(x, w, None, 0, True)
    got = y.cpu().double().permute(0, 3, 1, 2)
    ex = _run_exact(x, w, None, 0).double().permute(0, 3, 1, 2)
    # per output channel (the rows have very different scales): error relative to the channel's own scale
    sc = ref.abs().amax(dim=(0, 2, 3), keepdim=True)
    e_h = float(((got - ref).abs() / sc).max()); e_x = float(((ex - ref).abs() / sc).max())
    assert e_h <= 2.0 * e_x, (xscale, e_h, e_x)


def test_conv_f16x3_tiny_activations_degrade_gracefully():
    """Far below fp16's normal range (typical |x| = 1e-6: hi has 4-5 bits left, the remainder is denormal-spaced): a few 1e-5, never garbage
    (profiles/r06_f16x3_denormal_probe.txt)."""
    g = torch.Generator().manual_seed(12)
    x = F.relu(torch.randn(1, 64, 8, 8, generator=g)) * 1e-6
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    ref = _conv_ref64(x, w, None, 0)
    y, _ = _run_h(x, w, None, 0, True)
    assert _err(y.cpu().permute(0, 3, 1, 2), ref) <= 1e-4


def test_conv_f16x3_grouped_levels():
    """Five pyramid levels in ONE launch through the head's flat level-major buffers."""
    from efficientdet.pytorch_amd import ops, functional as Fn
    g = torch.Generator().manual_seed(5)
    B, sizes, Cin, Cout = 2, [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)], 64, 256
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    xs = [F.relu(torch.randn(B, Cin, h, ww, generator=g)) for (h, ww) in sizes]
    pm = [ops.Map.of(_nhwc(x)) for x in xs]            # (the neck hands the head one tensor per level)
    _, xh = Fn._pyramid_to_split(pm, B, sizes, Cin, torch.float32, 'cuda', bf=False, h=True)
    _, ym = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, 'cuda')
    _, ysm = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, 'cuda')
    ops.conv2d(xh, ops.pack_weight(w.cuda(), torch.float32, h3=True), ym, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b.cuda(),
               act=ops.ACT_RELU, hsplit=True, ysplit=ysm)
    torch.cuda.synchronize()
    for x, m, ms in zip(xs, ym, ysm):
        ref = F.relu(F.conv2d(x, w, b, padding=1))
        assert_close(from_hsplit(Fn.level_tensor(m))[0].float().permute(0, 3, 1, 2), ref, 1e-4, 'level %s' % (tuple(x.shape),))
        assert_close(from_split(Fn.level_tensor(ms)).permute(0, 3, 1, 2), ref, 1e-4, 'split copy %s' % (tuple(x.shape),))


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_bifpn_fusion_writes_the_h_split_operand(mode):
    """effdet_bifpn_fuse_fwd2: the H-split output is exactly the H-split of the plain output (same values, one pass), with and without
    the plain copy -- the operand of the node's f16x3 conv (models/bifpn.py:189-202)."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(3 + mode)
    B, H, W, C = 2, 8, 12, 64
    a = Map.of(torch.randn(B, H, W, C, generator=g).cuda())
    hb, wb = (H // 2, W // 2) if mode == 0 else (2 * H, 2 * W)
    b = Map.of(torch.randn(B, hb, wb, C, generator=g).cuda())
    c = Map.of(torch.randn(B, H, W, C, generator=g).cuda()) if mode == 1 else None
    wraw = torch.rand(3 if mode == 1 else 2, 5, generator=g).cuda()
    plain = ops.bifpn_fuse_fwd(a, b, c, wraw, 2, mode)
    both, hs = ops.bifpn_fuse_fwd(a, b, c, wraw, 2, mode, plain=True, hsplit=True)
    none, hs2 = ops.bifpn_fuse_fwd(a, b, c, wraw, 2, mode, plain=False, hsplit=True)
    torch.cuda.synchronize()
    assert none is None and torch.equal(both.tensor(), plain.tensor()) and torch.equal(hs.tensor().view(torch.int32), hs2.tensor().view(torch.int32))
    want = to_split2(plain.tensor().contiguous(), bf=False)[1]
    assert torch.equal(hs.tensor().view(torch.int32), want.view(torch.int32))


def test_two_independent_convs_in_one_launch_equal_the_separate_launches():
    """effdet_conv_t.seg_w / seg_shift: the same layer of the head's two towers (own inputs, own packed weights, own bias) as ONE grouped
    launch of 10 segments -- bit for bit the two 5-segment launches (same tiles, same K walk), H-split output and bf16 split copy alike."""
    from efficientdet.pytorch_amd import ops, functional as Fn
    g = torch.Generator().manual_seed(21)
    B, sizes, Cin, Cout = 2, [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)], 64, 256
    ws = [torch.randn(Cout, Cin, 3, 3, generator=g).cuda() / 24.0 for _ in range(2)]
    bs = [torch.randn(Cout, generator=g).cuda() * 0.1 for _ in range(2)]
    wps = [ops.pack_weight(w, torch.float32, h3=True) for w in ws]
    xs = []
    for _ in range(2):
        pm = [ops.Map.of(_nhwc(F.relu(torch.randn(B, Cin, h, w, generator=g)))) for (h, w) in sizes]
        xs.append(Fn._pyramid_to_split(pm, B, sizes, Cin, torch.float32, 'cuda', bf=False, h=True)[1])
    kw = dict(Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, act=ops.ACT_RELU, hsplit=True)
    fy, ya, yb = Fn.pyramid_alloc_pair(B, sizes, Cout, torch.float32, 'cuda')
    fs, sa, sb = Fn.pyramid_alloc_pair(B, sizes, Cout, torch.float32, 'cuda')
    ops.conv2d(xs[0] + xs[1], wps[0], ya + yb, shift=bs[0], ysplit=sa + sb, seg_w=[wps[0]] * 5 + [wps[1]] * 5, seg_shift=[bs[0]] * 5 + [bs[1]] * 5, **kw)
    gy, za, zb = Fn.pyramid_alloc_pair(B, sizes, Cout, torch.float32, 'cuda')
    gs, ta, tb = Fn.pyramid_alloc_pair(B, sizes, Cout, torch.float32, 'cuda')
    ops.conv2d(xs[0], wps[0], za, shift=bs[0], ysplit=ta, **kw)
    ops.conv2d(xs[1], wps[1], zb, shift=bs[1], ysplit=tb, **kw)
    torch.cuda.synchronize()
    assert torch.equal(fy.view(torch.int32), gy.view(torch.int32)) and torch.equal(fs.view(torch.int32), gs.view(torch.int32))
    assert not torch.equal(Fn.level_tensor(ya[0]).view(torch.int32), Fn.level_tensor(yb[0]).view(torch.int32))      # (the towers differ)


def test_out_of_range_activations_are_an_error_not_a_plausible_score():
    """|x| >= 65520 cannot be held by the fp16 split: the producers set the caller's watch word (an inf logit would come out of the sigmoid as
    a plausible 1.0), and the host-side check turns it into an exception and clears it.  Exactly at the edge: 65504 passes, 65520 does not."""
    from efficientdet.pytorch_amd import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    ops.range_flag(dev).zero_()
    ok = torch.full((1, 2, 2, 32), 65504.0, device='cuda')
    to_split2(ok, bf=False)
    ops.check_range_flag(dev)                                   # in range: no complaint
    to_split2(torch.full((1, 2, 2, 32), 65520.0, device='cuda'), bf=False)
    with pytest.raises(FloatingPointError):
        ops.check_range_flag(dev)
    ops.check_range_flag(dev)                                   # ... and the flag was cleared
    # a conv whose OUTPUT leaves the range (H-split epilogue), and NaN inputs
    x = torch.full((1, 64, 4, 4), 100.0)
    w = torch.full((64, 64, 3, 3), 2.0)
    y, _ = _run_h(x, w, None, 0, False)                         # 9 * 64 * 200 = 115 200 per output
    with pytest.raises(FloatingPointError):
        ops.check_range_flag(dev)
    assert not bool(torch.isfinite(from_hsplit(y)[0][0, 1:3, 1:3]).any())     # (the corners see 4 taps: 51 200, still in range)
    y, _ = _run_h(x, w, None, 0, True)                          # plain fp32 output: representable, correct, no flag
    ops.check_range_flag(dev)
    assert float(y.max()) == 115200.0
    to_split2(torch.full((1, 1, 1, 32), float('nan'), device='cuda'), bf=False)
    with pytest.raises(FloatingPointError):
        ops.check_range_flag(dev)


def test_f16x3_descriptor_is_refused_where_it_does_not_apply():
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    x = Map.new(1, 4, 4, 64, torch.float32, 'cuda')
    y = Map.new(1, 4, 4, 64, torch.float32, 'cuda')
    w = ops.pack_weight(torch.randn(64, 64, 3, 3).cuda(), torch.float32, h3=True)
    with pytest.raises(RuntimeError):        # a residual op is not part of the forward form
        ops.conv2d(x, w, y, Cin=64, Cout=64, KH=3, KW=3, pad_t=1, pad_l=1, res=y, res_mode=ops.RES_ADD, hsplit=True, out_f32=True)
    with pytest.raises(RuntimeError):        # K % 32 != 0 / K < 256: no f16x3 rows
        ops.pack_weight(torch.randn(64, 24, 1, 1).cuda(), torch.float32, h3=True)
