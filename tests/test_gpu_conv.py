"""GPU parity: MFMA implicit-GEMM conv (C ABI effdet_conv2d) vs torch-CPU fp32 F.conv2d."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import assert_close

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}
TOL_X3 = 1e-3      # fp32 storage, bf16x3 products: the SURVEY §8d gate itself (measured error ~4e-6 of tensor scale)


def _tol(dtype):
    from efficientdet.pytorch_amd import ops
    return TOL_X3 if (dtype == torch.float32 and ops.F32_ARITH == 'bf16x3') else TOL[dtype]


def _run(dtype, B, H, W, Cin, Cout, k, stride=1, pad=(1, 1, 1, 1), act=0, bn=False, res=False, save_z=False,
         rowscale=False, out_f32=False, seed=0):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = (0.5 + torch.rand(Cout, generator=g)) if bn else None
    shift = torch.randn(Cout, generator=g) * 0.3
    pl, pr, pt, pb = pad
    if dtype == torch.bfloat16:     # compare against the same rounded operands
        xq = x.bfloat16().float(); wq = w.bfloat16().float()
    else:
        xq, wq = x, w
    ref = F.conv2d(F.pad(xq, [pl, pr, pt, pb]), wq, None, stride)
    Ho, Wo = ref.shape[2], ref.shape[3]
    if bn:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    zref = ref.clone()
    if act == 1: ref = F.relu(ref)
    elif act == 2: ref = ref * torch.sigmoid(ref)
    elif act == 3: ref = torch.sigmoid(ref)
    rs = None
    if rowscale:
        rs = torch.rand(B, generator=g) + 0.5
        ref = ref * rs.view(-1, 1, 1, 1)
    r = None
    if res:
        r = torch.randn(B, Cout, Ho, Wo, generator=g)
        rq = r.bfloat16().float() if dtype == torch.bfloat16 else r
        ref = ref + rq
    dev = 'cuda'
    xm = Map.of(x.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
    wp = ops.pack_weight(w.to(dev), dtype)
    ym = Map.new(B, Ho, Wo, Cout, torch.float32 if out_f32 else dtype, dev)
    zm = Map.new(B, Ho, Wo, Cout, dtype, dev) if save_z else None
    rm = Map.of(r.permute(0, 2, 3, 1).contiguous().to(dev, dtype)) if res else None
    ops.conv2d(xm, wp, ym, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=stride, pad_t=pt, pad_l=pl,
               scale=scale.to(dev) if bn else None, shift=shift.to(dev), act=act,
               res=rm, res_mode=ops.RES_ADD if res else ops.RES_NONE,
               rowscale=rs.to(dev) if rowscale else None, zs=zm, out_f32=out_f32)
    torch.cuda.synchronize()
    got = ym.tensor().float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref, _tol(dtype), 'conv y %s' % ((dtype, B, H, W, Cin, Cout, k, stride),))
    if save_z:
        assert_close(zm.tensor().float().cpu().permute(0, 3, 1, 2), zref, _tol(dtype), 'conv z')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv3x3_head_shapes(dtype):
    _run(dtype, 2, 16, 16, 64, 256, 3, act=1)                       # tower layer 0 (K=576)
    _run(dtype, 1, 8, 8, 256, 256, 3, act=1)                        # tower (K=2304)
    _run(dtype, 1, 8, 8, 256, 720, 3, act=3, out_f32=True)          # retina_cls + sigmoid, fp32 out
    _run(dtype, 1, 8, 8, 256, 36, 3, out_f32=True)                  # retina_reg
    _run(dtype, 2, 4, 4, 64, 64, 3)                                 # BiFPN conv on a tiny level
    _run(dtype, 1, 1, 1, 64, 64, 3)                                 # 1x1 map (D0@128: top level)
    # more 3x3 'same' shapes: partial tiles, H != W, borders, fused epilogues
    _run(dtype, 3, 32, 32, 64, 128, 3, act=1, res=True)
    _run(dtype, 3, 4, 4, 256, 128, 3, act=2, bn=True, save_z=True)
    _run(dtype, 1, 64, 64, 64, 192, 3)
    _run(dtype, 2, 8, 16, 128, 144, 3, act=1)
    _run(dtype, 1, 2, 128, 64, 128, 3)


@pytest.fixture
def big_igemm():
    """Route eligible bf16 convs (Cin % 64 == 0, Cout >= 128) through the big-tile 32x32x16 kernel regardless of size."""
    from efficientdet.pytorch_amd import ops, _lib as L
    old_min = ops.tuning_set(L.TUNE_IGEMM_BIG_MIN_M, 0)
    old = ops.tuning_set(L.TUNE_IGEMM_BIG, 0)

    def use(variant):
        ops.tuning_set(L.TUNE_IGEMM_BIG, variant)
    yield use
    ops.tuning_set(L.TUNE_IGEMM_BIG, old); ops.tuning_set(L.TUNE_IGEMM_BIG_MIN_M, old_min)


@pytest.mark.parametrize('variant', [442, 242, 243, 423])
def test_conv_big_tile_variants(big_igemm, variant):
    """The big-tile kernel (v_mfma_f32_32x32x16_bf16, scalar tap walk through the SGPR offset) == F.conv2d on the shapes it
    serves: head towers / retina_cls (+ their data gradients), partial M and N tiles, borders, every fused epilogue."""
    big_igemm(variant)
    dt = torch.bfloat16
    _run(dt, 2, 16, 16, 64, 256, 3, act=1)                          # tower layer 0 (one K-step per tap)
    _run(dt, 1, 8, 8, 256, 256, 3, act=1)                           # tower (4 K-steps per tap)
    _run(dt, 1, 8, 8, 256, 720, 3, act=3, out_f32=True)             # retina_cls + sigmoid, fp32 out, 3 n-tiles (last partial)
    _run(dt, 3, 13, 9, 64, 200, 3, act=1, res=True)                 # ragged m tiles, H != W, partial n tile
    _run(dt, 2, 20, 20, 768, 256, 3)                                # the d(cls logits) data gradient (Cin padded to 768)
    _run(dt, 3, 4, 4, 256, 128, 3, act=2, bn=True, save_z=True)
    _run(dt, 1, 64, 64, 64, 192, 3, rowscale=True)
    _run(dt, 2, 24, 24, 128, 144, 1, pad=(0, 0, 0, 0), act=2, bn=True, save_z=True)    # pointwise
    _run(dt, 1, 17, 17, 64, 128, 3, stride=2, pad=(0, 1, 0, 1), act=2, bn=True)        # TF-same stride 2 (asymmetric pad)
    _run(dt, 2, 12, 12, 64, 160, 5, pad=(2, 2, 2, 2))                                  # 25 taps
    _run(dt, 2, 1, 1, 64, 256, 3)                                                      # 1x1 map: 8 of 9 taps fall outside
    _run(dt, 1, 2, 128, 64, 130, 3)                                                    # Cout % 4 != 0 tail


@pytest.mark.parametrize('variant', [442, 243])
def test_conv_big_tile_grouped_pyramid(big_igemm, variant):
    big_igemm(variant)
    _grouped(torch.bfloat16, Cin=64, nc=20)
    _grouped(torch.bfloat16, Cin=256, nc=80, B=3)


@pytest.fixture
def bf16x3():
    """fp32 storage with bf16x3 products (operands split into bf16 hi + lo, 3 bf16 MFMAs, fp32 accumulate)."""
    from efficientdet.pytorch_amd import ops
    old = ops.set_f32_arith('bf16x3')
    yield
    ops.set_f32_arith(old)


def test_conv_bf16x3(bf16x3):
    """The split kernel == F.conv2d at the fp32 tolerance on everything the rule routes to it (K % 32 == 0, K >= 256,
    Cout >= 32: head towers / heads / their data gradients, BiFPN 3x3, wide pointwise convs) -- and the same calls on
    shapes the rule leaves on the plain fp32 kernel still work with the mode switched on."""
    from efficientdet.pytorch_amd import ops, _lib as L
    import ctypes as C
    dt = torch.float32
    assert ops._mma_dtype_code(dt, 576, 256) == L.F32_BF16X3 and ops._mma_dtype_code(dt, 324, 256) == L.F32
    assert ops._mma_dtype_code(dt, 2304, 16) == L.F32 and ops._mma_dtype_code(torch.bfloat16, 576, 256) == L.BF16
    _run(dt, 2, 16, 16, 64, 256, 3, act=1)                          # tower layer 0
    _run(dt, 1, 8, 8, 256, 256, 3, act=1)                           # tower
    _run(dt, 1, 8, 8, 256, 720, 3, act=3, out_f32=True)             # retina_cls + sigmoid (3 n-tiles, last partial)
    _run(dt, 1, 8, 8, 256, 36, 3, out_f32=True)                     # retina_reg (64-wide tile)
    _run(dt, 3, 13, 9, 64, 200, 3, act=1, res=True)                 # ragged m tiles, H != W, partial n tile
    _run(dt, 2, 20, 20, 768, 256, 3)                                # d(cls logits) data gradient
    _run(dt, 3, 4, 4, 256, 128, 3, act=2, bn=True, save_z=True)
    _run(dt, 1, 64, 64, 64, 192, 3, rowscale=True)
    _run(dt, 2, 24, 24, 1152, 192, 1, pad=(0, 0, 0, 0), act=2, bn=True, save_z=True)   # wide pointwise
    _run(dt, 1, 17, 17, 64, 128, 3, stride=2, pad=(0, 1, 0, 1), act=2, bn=True)        # TF-same stride 2
    _run(dt, 2, 12, 12, 64, 160, 5, pad=(2, 2, 2, 2))                                  # 25 taps
    _run(dt, 2, 1, 1, 64, 256, 3)                                                      # 1x1 map
    _run(dt, 1, 2, 128, 64, 130, 3)                                                    # Cout % 4 != 0 tail
    _run(dt, 2, 8, 8, 96, 24, 1, pad=(0, 0, 0, 0), bn=True, res=True, rowscale=True)   # not eligible: plain kernel
    _run(dt, 3, 13, 9, 40, 54, 3)                                                      # K = 360: not a multiple of 32
    _grouped(dt, Cin=64, nc=20)
    _grouped(dt, Cin=256, nc=80, B=3)


def test_dgrad_bf16x3(bf16x3):
    test_dgrad_via_flipped_weights(torch.float32)


def test_wgrad_bf16x3(bf16x3):
    """Weight gradient with bf16x3 products: every shape class of the fp32 tests (DMA kernel levels take the split form,
    the rest the exact register-transpose kernel), incl. bias sums, split-K and the grouped pyramid."""
    test_wgrad_shapes(torch.float32)
    test_wgrad_grouped_pyramid(torch.float32)


def test_bf16x3_error_level(bf16x3):
    """What the split costs: on a K = 2304 reduction the result stays within 2e-5 of fp64 (scale-relative) -- the exact
    fp32 kernel is at ~1e-6, single bf16 at ~4e-3."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 256, 16, 16, generator=g); w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    xm = Map.of(x.permute(0, 2, 3, 1).contiguous().cuda())
    errs = {}
    for mode in ('f32', 'bf16x3'):
        ops.set_f32_arith(mode)
        ym = Map.new(2, 16, 16, 256, torch.float32, 'cuda')
        ops.conv2d(xm, ops.pack_weight(w.cuda(), torch.float32), ym, Cin=256, Cout=256, KH=3, KW=3, pad_t=1, pad_l=1)
        torch.cuda.synchronize()
        errs[mode] = float((ym.tensor().cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max())
    print('scale-relative max error vs fp64:', errs)
    assert errs['f32'] < 5e-6 and errs['bf16x3'] < 2e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv1x1_backbone_shapes(dtype):
    for (cin, cout) in [(16, 96), (96, 24), (144, 24), (24, 144), (240, 40), (40, 240), (480, 80), (672, 112),
                        (1152, 192), (1152, 320), (320, 64), (112, 64), (32, 16)]:
        _run(dtype, 2, 8, 8, cin, cout, 1, pad=(0, 0, 0, 0), act=2, bn=True, save_z=True)
    _run(dtype, 2, 8, 8, 96, 24, 1, pad=(0, 0, 0, 0), bn=True, res=True, rowscale=True)   # project + skip + drop_connect


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_odd_geometry(dtype):
    _run(dtype, 3, 13, 9, 40, 54, 3, act=0)                                   # ragged tiles, Cout % 4 != 0
    _run(dtype, 1, 17, 17, 24, 32, 3, stride=2, pad=(0, 1, 0, 1), act=2, bn=True)   # TF-same stride 2
    _run(dtype, 2, 12, 12, 16, 16, 5, pad=(2, 2, 2, 2))                       # 5x5, several taps per K-step
    _run(dtype, 1, 40, 40, 64, 64, 3, act=1)                                  # multiple m-tiles


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_grouped_pyramid(dtype):
    """One launch over 5 levels with shared weights, outputs written into the [B, A, C] head layout."""
    _grouped(dtype)


def _grouped(dtype, Cin=64, nc=20, B=2):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(3)
    Cout = 9 * nc
    sizes = [16, 8, 4, 2, 1]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    bias = torch.randn(Cout, generator=g) * 0.1
    xs = [torch.randn(B, Cin, s, s, generator=g) for s in sizes]
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    refs = [torch.sigmoid(F.conv2d(q(x), q(w), bias, 1, 1)).permute(0, 2, 3, 1).reshape(B, -1, nc) for x in xs]
    ref = torch.cat(refs, 1)
    A = ref.shape[1]
    dev = 'cuda'
    tot = sum(B * s * s * Cin for s in sizes)
    flat = torch.empty(tot, dtype=dtype, device=dev)
    xm, off = [], 0
    for x, s in zip(xs, sizes):
        n = B * s * s * Cin
        flat[off:off + n] = x.permute(0, 2, 3, 1).reshape(-1).to(dev, dtype)
        xm.append(Map(flat, B, s, s, Cin, off=off)); off += n
    out = torch.zeros(B, A, nc, dtype=torch.float32, device=dev)
    ym, aoff = [], 0
    for s in sizes:
        ym.append(Map(out, B, s, s, Cout, ld=Cout, bstride=A * nc, off=aoff * nc)); aoff += s * s * 9
    ops.conv2d(xm, ops.pack_weight(w.to(dev), dtype), ym, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1,
               shift=bias.to(dev), act=ops.ACT_SIGMOID, out_f32=True)
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, _tol(dtype), 'grouped head output')


def _run_wgrad(dtype, B, H, W, Cin, Cout, k, stride=1, pad=(1, 1, 1, 1), seed=0, bias=True):
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    pl, pr, pt, pb = pad
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    xp = F.pad(q(x), [pl, pr, pt, pb])
    y = F.conv2d(xp, w, None, stride)
    dz = torch.randn(y.shape, generator=g)
    y.backward(q(dz))
    ref = w.grad                                     # [Cout][Cin][k][k]
    dev = 'cuda'
    xm = Map.of(x.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
    dzm = Map.of(dz.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
    gp = torch.zeros(Cout, k * k, Cin, dtype=torch.float32, device=dev)
    db = torch.zeros(Cout, dtype=torch.float32, device=dev) if bias else None
    ops.conv2d_wgrad(xm, dzm, gp, db, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=stride, pad_t=pt, pad_l=pl)
    dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=dev)
    ops.unpack_wgrad(gp, dw)
    torch.cuda.synchronize()
    tol = (_tol(dtype) if _tol(dtype) == TOL_X3 else 3e-4) if dtype == torch.float32 else 1e-2
    assert_close(dw.cpu(), ref, tol, 'wgrad %s' % ((dtype, B, H, W, Cin, Cout, k, stride),))
    if bias:
        assert_close(db.cpu(), q(dz).sum(dim=(0, 2, 3)), tol, 'dbias')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_shapes(dtype):
    _run_wgrad(dtype, 2, 16, 16, 64, 256, 3)
    _run_wgrad(dtype, 1, 8, 8, 256, 256, 3)
    _run_wgrad(dtype, 2, 8, 8, 256, 180, 3)
    _run_wgrad(dtype, 2, 4, 4, 256, 36, 3)
    _run_wgrad(dtype, 3, 13, 9, 40, 54, 3)
    _run_wgrad(dtype, 2, 12, 12, 96, 24, 1, pad=(0, 0, 0, 0))
    _run_wgrad(dtype, 2, 12, 12, 16, 96, 1, pad=(0, 0, 0, 0))
    _run_wgrad(dtype, 1, 17, 17, 24, 32, 3, stride=2, pad=(0, 1, 0, 1))
    _run_wgrad(dtype, 2, 32, 32, 4 if dtype == torch.float32 else 8, 32, 3, stride=2, pad=(0, 1, 0, 1))       # the stem's geometry (bf16x3: the DMA kernel's stride-2 form)
    _run_wgrad(dtype, 3, 16, 24, 8, 40, 3, stride=2, pad=(0, 1, 0, 1))
    _run_wgrad(dtype, 4, 40, 40, 64, 64, 3)          # several K-splits
    _run_wgrad(dtype, 2, 1, 1, 64, 64, 3)
    # bf16: shapes that take the DMA + LDS-transpose-read kernel with 8-pixel pieces made of whole image rows
    _run_wgrad(dtype, 2, 4, 4, 256, 40, 3)
    _run_wgrad(dtype, 3, 2, 4, 64, 64, 3)
    _run_wgrad(dtype, 1, 6, 4, 32, 48, 3)
    _run_wgrad(dtype, 2, 8, 16, 64, 64, 3)
    _run_wgrad(dtype, 5, 4, 2, 64, 64, 3)
    _run_wgrad(dtype, 3, 24, 24, 24, 144, 1, pad=(0, 0, 0, 0))
    # short rows x many split-K slabs: the sliced slab reduction of the unpack kernel
    _run_wgrad(dtype, 8, 64, 64, 16, 96, 1, pad=(0, 0, 0, 0))
    _run_wgrad(dtype, 4, 64, 64, 144, 24, 1, pad=(0, 0, 0, 0))


def test_thin_pointwise_wgrad_kernel():
    """conv_wgrad_thin_kernel (fp32 storage, exact fp32 MFMA in both arithmetic modes): the backbone's high-resolution 1x1 shapes,
    partial 16-channel tiles, a pixel count that is not a whole stage, per-image slabs -- against torch and bitwise run to run."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    dev = 'cuda'
    g = torch.Generator().manual_seed(21)
    for (B, H, W, Cin, Cout) in [(8, 64, 64, 16, 96), (8, 64, 64, 32, 16), (8, 64, 64, 24, 144), (8, 64, 64, 144, 24), (9, 60, 64, 144, 40),
                                 (8, 64, 64, 96, 24), (3, 128, 128, 16, 16)]:
        x = torch.randn(B, H, W, Cin, generator=g).to(dev); dz = torch.randn(B, H, W, Cout, generator=g).to(dev)
        xm, zm = Map.of(x), Map.of(dz)
        assert ops.conv2d_wgrad_kernel_id(xm, zm, Cin=Cin, Cout=Cout, KH=1, KW=1) == 1
        ref = torch.einsum('bhwn,bhwc->nc', dz.double(), x.double())
        G, dbp = ops.conv2d_wgrad(xm, zm, Cin=Cin, Cout=Cout, KH=1, KW=1)
        G2, dbp2 = ops.conv2d_wgrad(xm, zm, Cin=Cin, Cout=Cout, KH=1, KW=1)
        assert torch.equal(G, G2) and torch.equal(dbp, dbp2)
        got = G.double().sum(0).view(Cout, Cin)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-5 * scale, (B, H, W, Cin, Cout)
        assert float((dbp.double().sum(0) - dz.double().sum(dim=(0, 1, 2))).abs().max()) <= 2e-5 * float(dz.double().sum(dim=(0, 1, 2)).abs().max() + 1)
        if (H * W) % 32 == 0:
            Gi, _ = ops.conv2d_wgrad(xm, zm, Cin=Cin, Cout=Cout, KH=1, KW=1, image_splits=True)       # slab s = image s // q
            q = Gi.shape[0] // B
            per = Gi.double().view(B, q, Cout, Cin).sum(1)
            refi = torch.einsum('bhwn,bhwc->bnc', dz.double(), x.double())
            assert float((per - refi).abs().max()) <= 2e-5 * float(refi.abs().max())
    # the stem's form: 3x3 stride 2 on a 4-channel NHWC image (3 real channels), bottom / right padding, gathered im2col rows
    for (B, H, Cout) in [(8, 128, 32), (3, 224, 24)]:
        img = torch.randn(B, 3, H, H, generator=g)
        x4 = torch.zeros(B, H, H, 4); x4[..., :3] = img.permute(0, 2, 3, 1)
        Ho = H // 2
        dz = torch.randn(B, Cout, Ho, Ho, generator=g)
        wt = torch.zeros(Cout, 3, 3, 3, requires_grad=True)
        (F.conv2d(F.pad(img, [0, 1, 0, 1]), wt, None, 2) * dz).sum().backward()
        xm, zm = Map.of(x4.to(dev)), Map.of(dz.permute(0, 2, 3, 1).contiguous().to(dev))
        assert ops.conv2d_wgrad_kernel_id(xm, zm, Cin=4, Cout=Cout, KH=3, KW=3, stride=2) == 1
        G, dbp = ops.conv2d_wgrad(xm, zm, Cin=4, Cout=Cout, KH=3, KW=3, stride=2)
        G2, _ = ops.conv2d_wgrad(xm, zm, Cin=4, Cout=Cout, KH=3, KW=3, stride=2)
        assert torch.equal(G, G2)
        got = G.double().sum(0)[:, :, :3].permute(0, 2, 1).reshape(Cout, 3, 3, 3).cpu()       # [Cout][tap][c] -> OIHW
        assert float((got - wt.grad.double()).abs().max()) <= 2e-5 * float(wt.grad.abs().max()), (B, H, Cout)
        assert float((G.double().sum(0)[:, :, 3]).abs().max()) == 0.0                              # the padding channel
        assert float((dbp.double().sum(0).cpu() - dz.double().sum(dim=(0, 2, 3))).abs().max()) <= 2e-5 * float(dz.double().sum(dim=(0, 2, 3)).abs().max() + 1)
    # not this kernel: few pixels, wide rows, 3x3
    xs, zs = Map.of(torch.zeros(2, 16, 16, 16, device=dev)), Map.of(torch.zeros(2, 16, 16, 96, device=dev))
    assert ops.conv2d_wgrad_kernel_id(xs, zs, Cin=16, Cout=96, KH=1, KW=1) == 0
    xw, zw = Map.of(torch.zeros(8, 64, 64, 240, device=dev)), Map.of(torch.zeros(8, 64, 64, 40, device=dev))
    assert ops.conv2d_wgrad_kernel_id(xw, zw, Cin=240, Cout=40, KH=1, KW=1) == 0
    x3_, z3_ = Map.of(torch.zeros(8, 64, 64, 64, device=dev)), Map.of(torch.zeros(8, 64, 64, 64, device=dev))
    assert ops.conv2d_wgrad_kernel_id(x3_, z3_, Cin=64, Cout=64, KH=3, KW=3, pad_t=1, pad_l=1) == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_grouped_pyramid(dtype):
    """One wgrad launch over 5 pyramid levels that share a weight (levels split between the two bf16 kernels)."""
    from efficientdet.pytorch_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout = 3, 64, 72
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    xs = [torch.randn(B, Cin, h, w, generator=g) for h, w in sizes]
    dzs = [torch.randn(B, Cout, h, w, generator=g) for h, w in sizes]
    wt = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    sum((F.conv2d(q(x), wt, None, 1, 1) * q(dz)).sum() for x, dz in zip(xs, dzs)).backward()
    dev = 'cuda'
    _, xm = Fn.pyramid_alloc(B, sizes, Cin, dtype, dev)
    _, zm = Fn.pyramid_alloc(B, sizes, Cout, dtype, dev)
    for m, t in zip(xm + zm, xs + dzs):
        Fn.level_tensor(m).copy_(t.permute(0, 2, 3, 1).to(dev, dtype))
    gp = torch.zeros(Cout, 9, Cin, dtype=torch.float32, device=dev)
    db = torch.zeros(Cout, dtype=torch.float32, device=dev)
    ops.conv2d_wgrad(xm, zm, gp, db, Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev)
    ops.unpack_wgrad(gp, dw)
    torch.cuda.synchronize()
    tol = (_tol(dtype) if _tol(dtype) == TOL_X3 else 3e-4) if dtype == torch.float32 else 1e-2
    assert_close(dw.cpu(), wt.grad, tol, 'pyramid wgrad')
    assert_close(db.cpu(), sum(q(dz).sum(dim=(0, 2, 3)) for dz in dzs), tol, 'pyramid dbias')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_independent_problems_in_one_launch(dtype):
    """group=True: five maps in SEPARATE allocations, each with its own weight gradient, share one weight-gradient launch; every
    problem's slab range unpacks to what a launch of its own gives (the BiFPN nodes' form), and to torch's gradient."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(13)
    B, Cc = 4, 64
    sizes = [(32, 32), (8, 8), (16, 16), (4, 4), (2, 2)]
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    dev = 'cuda'
    xs = [torch.randn(B, Cc, h, w, generator=g) for h, w in sizes]
    dzs = [torch.randn(B, Cc, h, w, generator=g) for h, w in sizes]
    pad = [torch.empty(1000 * (i + 1), device=dev) for i in range(5)]              # keep the allocations apart
    xm = [Map.of(t.permute(0, 2, 3, 1).contiguous().to(dev, dtype)) for t in xs]
    zm = [Map.of(t.permute(0, 2, 3, 1).contiguous().to(dev, dtype)) for t in dzs]
    outs = ops.conv2d_wgrad(xm, zm, Cin=Cc, Cout=Cc, KH=3, KW=3, pad_t=1, pad_l=1, group=True)
    assert len(outs) == 5 and len(pad) == 5
    tol = (_tol(dtype) if _tol(dtype) == TOL_X3 else 3e-4) if dtype == torch.float32 else 1e-2
    for i, (G, dbp) in enumerate(outs):
        dw = torch.empty(Cc, Cc, 3, 3, device=dev); db = ops.unpack_wgrad(G, dw, dbias_part=dbp)
        G1, dbp1 = ops.conv2d_wgrad(xm[i], zm[i], Cin=Cc, Cout=Cc, KH=3, KW=3, pad_t=1, pad_l=1)
        dw1 = torch.empty(Cc, Cc, 3, 3, device=dev); db1 = ops.unpack_wgrad(G1, dw1, dbias_part=dbp1)
        wt = torch.zeros(Cc, Cc, 3, 3, requires_grad=True)
        (F.conv2d(q(xs[i]), wt, None, 1, 1) * q(dzs[i])).sum().backward()
        torch.cuda.synchronize()
        assert_close(dw.cpu(), wt.grad, tol, 'grouped wgrad %d' % i)
        assert_close(dw.cpu(), dw1.cpu(), 1e-4 if dtype == torch.float32 else 1e-2, 'grouped vs single %d' % i)    # (other split-K boundaries)
        assert_close(db.cpu(), q(dzs[i]).sum(dim=(0, 2, 3)), tol, 'grouped dbias %d' % i)
        assert_close(db.cpu(), db1.cpu(), 1e-4, 'grouped vs single dbias %d' % i)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dgrad_via_flipped_weights(dtype):
    """Data gradient of a stride-1 conv = forward kernel on the mode-1 packed (flipped, transposed) weights."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 10, 12, 64, 96
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    scale = 0.5 + torch.rand(Cout, generator=g)
    y = F.conv2d(x, q(w * scale.view(-1, 1, 1, 1)), None, 1, 1)
    dz = torch.randn(y.shape, generator=g)
    y.backward(q(dz))
    dev = 'cuda'
    wd = ops.pack_weight(w.to(dev), dtype, mode=1, scale=scale.to(dev))
    dzm = Map.of(dz.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
    dxm = Map.new(B, H, W, Cin, dtype, dev)
    ops.conv2d(dzm, wd, dxm, Cin=Cout, Cout=Cin, KH=3, KW=3, pad_t=1, pad_l=1)
    torch.cuda.synchronize()
    assert_close(dxm.tensor().float().cpu().permute(0, 3, 1, 2), x.grad, _tol(dtype), 'dgrad')


@pytest.mark.parametrize('Cin,Cout', [(16, 96), (24, 144), (16, 32), (24, 96)])
def test_conv_pw_skinny_kernel(Cin, Cout):
    """The VALU 'skinny' pointwise kernel (fp32, Cin 16 / 24, >= 64 k pixels: MBConv expand convs of the high-resolution blocks and
    the project convs' data gradients) == torch on every epilogue it serves: BN + Swish with the saved pre-activation, z-only
    (act NONE), and the fused squeeze-excite backward form (row scale, per-(image, channel) affine, Swish' of a residual)."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    B, H, W = 5, 128, 104                                       # 66 560 pixels, not a multiple of the per-workgroup pixel count
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g); shift = torch.randn(Cout, generator=g) * 0.3
    c = (x.reshape(-1, Cin) @ w.view(Cout, Cin).t()).view(B, H, W, Cout)
    z = c * scale + shift
    dev = 'cuda'
    xm = Map.of(x.to(dev)); wp = ops.pack_weight(w.to(dev), torch.float32)

    def run(**kw):
        ym = Map.new(B, H, W, Cout, torch.float32, dev)
        ops.PROFILE = ops.LaunchProfile()
        try:
            ops.conv2d(xm, wp, ym, Cin=Cin, Cout=Cout, KH=1, KW=1, **kw)
            torch.cuda.synchronize()
            assert ops.PROFILE.records[0][0] == 'conv_pw_f32_kernel', ops.PROFILE.records[0][0]
        finally:
            ops.PROFILE = None
        return ym.tensor().cpu()
    # (a) inference / save-y training form: BN affine + Swish, pre-activation saved
    zm = Map.new(B, H, W, Cout, torch.float32, dev)
    y = run(scale=scale.to(dev), shift=shift.to(dev), act=ops.ACT_SWISH, zs=zm)
    assert_close(y, z * torch.sigmoid(z), 2e-4, 'pw swish'); assert_close(zm.tensor().cpu(), z, 2e-4, 'pw z')
    # (b) z-only training form
    assert_close(run(scale=scale.to(dev), shift=shift.to(dev), act=ops.ACT_NONE), z, 2e-4, 'pw z-only')
    # (c) fused squeeze-excite backward: (rs_b * conv * gate[b][n] + dpool[b][n]) * swish'(res)
    rs = torch.rand(B, generator=g) + 0.5
    gate = torch.rand(B, Cout, generator=g); dpool = torch.randn(B, Cout, generator=g) * 0.1
    res = torch.randn(B, H, W, Cout, generator=g)
    sg = torch.sigmoid(res)
    ref = (c * rs.view(B, 1, 1, 1) * gate.view(B, 1, 1, Cout) + dpool.view(B, 1, 1, Cout)) * (sg * (1 + res * (1 - sg)))
    got = run(rowscale=rs.to(dev), bc_scale=gate.to(dev), bc_shift=dpool.to(dev), res=Map.of(res.to(dev)), res_mode=ops.RES_SWISH_GRAD)
    assert_close(got, ref, 2e-4, 'pw fused se backward')
    # (d) identity-skip residual add
    assert_close(run(res=Map.of(res.to(dev)), res_mode=ops.RES_ADD), c + res, 2e-4, 'pw res add')


def test_batched_unpack_is_bitwise_the_single_launches():
    """effdet_unpack_conv_wgrad_batch: 30 jobs of the three forms (bias rows, frozen-BN parameter gradients, per-image slab
    scale; short and long rows, 1..128 slabs) in two launches == the same jobs launched one by one, bit for bit."""
    from efficientdet.pytorch_amd import ops
    torch.manual_seed(5)
    dev = 'cuda'
    shapes = [(64, 64, 3, 1), (64, 64, 3, 6), (256, 256, 3, 8), (24, 96, 1, 32), (96, 16, 1, 40), (40, 144, 1, 64), (36, 256, 3, 5),
              (88, 40, 1, 3), (720, 256, 3, 2), (16, 32, 1, 128)] * 3
    jobs = []
    for i, (co, ci, k, ns) in enumerate(shapes):
        cp = (ci + 3) // 4 * 4
        g = torch.randn(ns, co, k * k, cp, device=dev)
        w = torch.randn(co, ci, k, k, device=dev)
        part = torch.randn(ns, co, device=dev)
        kind = i % 3
        rs = (torch.rand(ns // 2 if ns % 2 == 0 and kind == 2 else 1, device=dev) + 0.5) if kind != 0 else None
        jobs.append((kind, g, w, part, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev), torch.rand(co, device=dev) + 0.5, rs, cp))

    def run():
        outs = []
        for kind, g, w, part, scale, mean, inv, rs, cp in jobs:
            if kind == 1:
                outs.append(ops.unpack_wgrad_bn(g, w, scale, part, mean, inv, cin_pad=cp, slab_scale=rs))
            else:
                dw = torch.empty_like(w)
                db = ops.unpack_wgrad(g, dw, dbias_part=part, cin_pad=cp, slab_scale=rs)
                outs.append((dw, db))
        return outs
    single = run()
    with ops.unpack_batch():
        batched = run()
    torch.cuda.synchronize()
    for a, b in zip(single, batched):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    # the first job against plain torch
    kind, g, w, part, *_ = jobs[0]
    ref = g.sum(0)[:, :, :w.shape[1]].permute(0, 2, 1).reshape(w.shape[0], w.shape[1], 3, 3)
    assert torch.allclose(single[0][0], ref, atol=1e-5) and torch.allclose(single[0][1], part.sum(0), atol=1e-5)


def test_integration_md_stub_runs():
    """The ctypes stub PRINTED in INTEGRATION.md section 2 (what a reference maintainer would paste), extracted from the document
    and executed as is: one RetinaHead tower conv (models/module.py:495-501, 3x3 + bias + ReLU) vs F.conv2d on the same
    bf16-rounded operands."""
    from tests.test_abi import integration_md_stub
    ns = {}
    exec(integration_md_stub(), ns)
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 16, 24, 64, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.3
    y = ns['conv3x3_bias_relu'](x.permute(0, 2, 3, 1).contiguous().to('cuda', torch.bfloat16), w.cuda(), b.cuda())
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, padding=1))
    assert y.shape == (B, H, W, Cout) and y.dtype == torch.bfloat16
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL[torch.bfloat16], 'INTEGRATION.md stub')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('bf16x3', [False, True])
@pytest.mark.parametrize('B,H,W,Cin,Cout', [(3, 16, 16, 96, 24), (2, 16, 8, 480, 80), (4, 16, 16, 1152, 320)])
def test_conv_per_image_weights_fold_the_se_gate(dtype, bf16x3, B, H, W, Cin, Cout):
    """effdet_conv_t.w_image_stride + effdet_scale_pack_weight: y_b = BN(W diag(gate_b) x_b) (+ residual) == the reference's
    project conv applied to x * gate (models/efficientnet.py:86-95), for every tile-width / arithmetic the MBConv blocks reach."""
    from efficientdet.pytorch_amd import ops
    from efficientdet.pytorch_amd.ops import Map
    if bf16x3 and dtype != torch.float32:
        pytest.skip('bf16x3 is an arithmetic of fp32 storage')
    old = ops.set_f32_arith('bf16x3' if bf16x3 else 'f32')
    try:
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, Cin, H, W, generator=g)
        gate = torch.rand(B, Cin, generator=g)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
        scale = 0.5 + torch.rand(Cout, generator=g); shift = torch.randn(Cout, generator=g) * 0.3
        r = torch.randn(B, Cout, H, W, generator=g)
        q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
        wb = q(w.view(1, Cout, Cin) * gate.view(B, 1, Cin))                       # what the pack stores (bf16 mode: rounded once)
        ref = torch.einsum('bnk,bkhw->bnhw', wb, q(x)) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + q(r)
        dev = 'cuda'
        xm = Map.of(x.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
        rm = Map.of(r.permute(0, 2, 3, 1).contiguous().to(dev, dtype))
        wpb, stride = ops.scale_pack_weight(w.to(dev), gate.to(dev), dtype)
        ym = Map.new(B, H, W, Cout, dtype, dev)
        ops.conv2d(xm, wpb, ym, Cin=Cin, Cout=Cout, KH=1, KW=1, scale=scale.to(dev), shift=shift.to(dev), res=rm, res_mode=ops.RES_ADD,
                   w_image_stride=stride)
        torch.cuda.synchronize()
        assert_close(ym.tensor().float().cpu().permute(0, 3, 1, 2), ref, _tol(dtype), 'per-image weights %s' % ((dtype, bf16x3, B, H, W, Cin, Cout),))
        # images that are not whole 128-pixel tiles are refused, not mis-addressed
        x2 = Map.of(torch.zeros(B, 8, 8, Cin, device=dev, dtype=dtype)); y2 = Map.new(B, 8, 8, Cout, dtype, dev)
        with pytest.raises(RuntimeError):
            ops.conv2d(x2, wpb, y2, Cin=Cin, Cout=Cout, KH=1, KW=1, w_image_stride=stride)
    finally:
        ops.set_f32_arith(old)


def test_conv_second_output_in_the_split_layout():
    """effdet_conv_t.y_split: an exact-fp32 conv writes its output a second time as [32 x bf16 hi | 32 x bf16 lo] groups -- the first
    output is untouched (bitwise the conv without it) and the second equals effdet_to_split of the first, bit for bit; grouped over
    pyramid levels like the RetinaHead tower convs that use it ('f32_bwd_bf16x3': exact forward, split-layout gradient kernels)."""
    from efficientdet.pytorch_amd import functional as Fn, ops
    ops.set_f32_arith('f32')          # (the process-wide arithmetic is whatever the last model's backward left: this test is about exact fp32)
    torch.manual_seed(3)
    B, Cin, Cout = 2, 64, 256
    sizes = [(16, 16), (8, 8), (4, 4)]
    dev = 'cuda'
    _, xs = Fn.pyramid_alloc(B, sizes, Cin, torch.float32, dev)
    for m in xs:
        Fn.level_tensor(m).normal_()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    wp = ops.pack_weight(w, torch.float32)
    f0, y0 = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, dev)
    f1, y1 = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, dev)
    fs, ys = Fn.pyramid_alloc(B, sizes, Cout, torch.float32, dev)
    kw = dict(Cin=Cin, Cout=Cout, KH=3, KW=3, pad_t=1, pad_l=1, shift=b, act=ops.ACT_RELU)
    ops.conv2d(xs, wp, y0, **kw)
    ops.conv2d(xs, wp, y1, ysplit=ys, **kw)
    torch.cuda.synchronize()
    assert torch.equal(f0, f1)
    assert torch.equal(fs.view(torch.int32), ops.to_split(f0).view(torch.int32))
    # refused where the split copy cannot be addressed like y (Cout not whole 32-channel groups) or the arithmetic is not exact fp32
    w2 = torch.randn(48, Cin, 3, 3, device=dev)
    _, y2 = Fn.pyramid_alloc(B, sizes, 48, torch.float32, dev); _, s2 = Fn.pyramid_alloc(B, sizes, 48, torch.float32, dev)
    with pytest.raises(RuntimeError):
        ops.conv2d(xs, ops.pack_weight(w2, torch.float32), y2, Cin=Cin, Cout=48, KH=3, KW=3, pad_t=1, pad_l=1, ysplit=s2)
