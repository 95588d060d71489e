"""CPU tier: architecture arithmetic, state_dict layout (checkpoint ABI) and the no-CPU-fallback contract."""
import os

import pytest
import torch

from oracle import effdet_oracle as O


@pytest.mark.parametrize('net', ['efficientdet-d%d' % i for i in range(8)])
def test_backbone_plan_matches_oracle(net):
    from efficientdet.pytorch_amd.config import MODEL_MAP, backbone_plan, EFFICIENTDET
    stem_c, stem_pad, blocks, head_c, native = backbone_plan(MODEL_MAP[net])
    ostem, oblocks, ohead, onative = O.backbone_blocks(net)
    assert (stem_c, stem_pad, head_c, native) == (ostem['cout'], ostem['pad'], ohead, onative)
    assert len(blocks) == len(oblocks)
    for b, o in zip(blocks, oblocks):
        assert (b.k, b.stride, b.expand, b.cin, b.cout, b.cexp, b.cse, b.pad, b.skip, b.stage_end) == \
            (o['k'], o['s'], o['e'], o['cin'], o['cout'], o['cexp'], o['cse'], o['pad'], o['skip'], o['stage_end'])
    assert EFFICIENTDET[net] == {**O.EFFICIENTDET[net], 'backbone': EFFICIENTDET[net]['backbone']}


def test_d0_shapes_and_strides():
    from efficientdet.pytorch_amd.config import backbone_plan, conv_out
    _, pad, blocks, _, _ = backbone_plan('efficientnet-b0')
    s = conv_out(512, 3, 2, pad)
    sizes = []
    for b in blocks:
        s = conv_out(s, b.k, b.stride, b.pad)
        if b.stage_end:
            sizes.append((b.cout, s))
    assert sizes == [(16, 256), (24, 128), (40, 64), (80, 32), (112, 16), (192, 8), (320, 4)]


@pytest.mark.parametrize('net,nc', [('efficientdet-d0', 80), ('efficientdet-d2', 20), ('efficientdet-d4', 4)])
def test_state_dict_layout_is_the_reference_layout(net, nc):
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
    c = EFFICIENTDET[net]
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'])
    sd = O.make_state_dict(net, nc)          # key order / shapes verified against the real reference (make_golden.py)
    got = m.state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(tuple(got[k].shape) == tuple(sd[k].shape) for k in sd)
    m.load_state_dict(sd)
    if net == 'efficientdet-d0':
        assert len(sd) == 426 and sum(p.numel() for p in m.parameters()) == 11_505_854
        assert sum(p.numel() for p in m.live_parameters()) == 11_505_854 - 1_693_160


def test_constructor_surface_and_errors():
    from efficientdet.pytorch_amd import EfficientDet, MODEL_MAP
    import inspect
    sig = inspect.signature(EfficientDet.__init__)
    names = list(sig.parameters)[1:9]
    assert names == ['num_classes', 'network', 'D_bifpn', 'W_bifpn', 'D_class', 'is_training', 'threshold', 'iou_threshold']
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d['network'], d['D_bifpn'], d['W_bifpn'], d['D_class'], d['is_training'], d['threshold'], d['iou_threshold']) == \
        ('efficientdet-d0', 3, 88, 3, True, 0.01, 0.5)
    assert set(MODEL_MAP) == {'efficientdet-d%d' % i for i in range(8)}
    with pytest.raises(KeyError):
        EfficientDet(3, network='efficientdet-d9')
    m = EfficientDet(3, W_bifpn=64, D_bifpn=1)
    for attr in ('backbone', 'neck', 'bbox_head', 'anchors', 'criterion', 'threshold', 'iou_threshold', 'is_training',
                 'freeze_bn', 'extract_feat'):
        assert hasattr(m, attr)
    # re-init like models/efficientdet.py:47-53: BN gamma 1 / beta 0, conv ~ N(0, sqrt(2/(k*k*Cout)))
    assert float(m.backbone._bn0.weight.min()) == 1.0 and float(m.backbone._bn0.bias.abs().max()) == 0.0
    w = m.bbox_head.cls_convs[1].conv.weight
    assert abs(float(w.std()) - (2.0 / (9 * 256)) ** 0.5) < 2e-3


def test_refuses_cpu_tensors():
    from efficientdet.pytorch_amd import EfficientDet
    m = EfficientDet(3, W_bifpn=64, D_bifpn=1, is_training=False)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 3, 128, 128))
    m.is_training = True
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m([torch.zeros(1, 3, 128, 128), torch.zeros(1, 2, 5)])


def test_conv_kernel_selection_rule():
    """effdet_conv2d_kernel (no device work): the persistent 256x256 / 32x32x16 kernel serves the long-K head convs of the
    benchmark shape, everything else stays on the 128x128 implicit-GEMM kernel (measured rule, csrc/conv_igemm.hip)."""
    import ctypes as C
    from efficientdet.pytorch_amd import _lib as L

    def kid(dtype, B, cin, cout, res_mode=0, sizes=((64, 64), (32, 32), (16, 16), (8, 8), (4, 4)), k=3):
        d = L.ConvDesc()
        d.x = d.w = d.y = 0x1000
        d.res = 0x1000 if res_mode else None
        d.dtype, d.B, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l = dtype, B, cin, cout, k, k, 1, k // 2, k // 2
        d.ldx, d.ldy, d.res_mode, d.nseg = cin, cout, res_mode, len(sizes)
        off_i = off_o = 0
        for i, (h, w) in enumerate(sizes):
            s = d.seg[i]
            s.H = s.Ho = h; s.W = s.Wo = w
            s.in_off, s.in_bstride, s.out_off, s.out_bstride = off_i, h * w * cin, off_o, h * w * cout
            off_i += B * h * w * cin; off_o += B * h * w * cout
        return int(L.lib().effdet_conv2d_kernel(C.byref(d)))
    assert kid(L.BF16, 32, 256, 256) == 10 + 442                      # tower forward
    assert kid(L.BF16, 32, 256, 720) == 10 + 442                      # retina_cls forward
    assert kid(L.BF16, 32, 768, 256, res_mode=L.RES_RELU_MASK) == 10 + 442      # d(cls logits) data gradient (K = 6912)
    assert kid(L.BF16, 32, 256, 256, res_mode=L.RES_RELU_MASK) == 0   # tower data gradient: residual behind a short K loop
    assert kid(L.BF16, 32, 64, 256) == 0                              # first tower layer: K = 576
    assert kid(L.BF16, 32, 256, 64) == 1 and kid(L.BF16, 32, 256, 36) == 1
    assert kid(L.F32, 32, 256, 256) == 0                              # the parity dtype never takes the bf16 kernel
    assert kid(L.BF16, 2, 256, 256) == 0                              # small launches: not enough tiles
    assert kid(L.BF16, 32, 224, 224) == 0                             # D4 head: Cin % 64 != 0
    assert kid(7, 32, 256, 256) == -1                                 # EFFDET_EINVAL
    # maximum sizes: a tensor the 32-bit buffer offsets cannot span is refused loudly (EFFDET_EUNSUPPORTED), never wrapped
    assert kid(L.F32, 64, 144, 24, sizes=((512, 512),), k=1) == -3     # 64 x 512 x 512 x 144 fp32 = 9.7 GB input
    assert kid(L.BF16, 16, 144, 24, sizes=((512, 512),), k=1) >= 0     # 1.2 GB: fine
    # fp32 storage with bf16x3 products: its own kernels (ids 4..7), K % 32 == 0 required
    assert kid(L.F32_BF16X3, 32, 256, 256) == 4 and kid(L.F32_BF16X3, 32, 256, 64) == 5 and kid(L.F32_BF16X3, 32, 256, 36) == 5
    assert kid(L.F32_BF16X3, 32, 64, 32) == 6
    assert kid(L.F32_BF16X3, 32, 36, 256) == -3                       # K = 324: EFFDET_EUNSUPPORTED (the caller's rule never asks)
    assert kid(L.F32_BF16X3, 32, 256, 16) == 7                        # works; the Python rule just never asks (HBM-bound)
    # the f16x3 forward form (H-split activations, three-piece f16 weights): its own kernel, 128- or 64-channel block tile
    assert kid(L.F32_HSPLIT, 32, 256, 256) == 30 and kid(L.F32_HSPLIT, 32, 64, 256) == 30 and kid(L.F32_HSPLIT, 32, 256, 64) == 31
    assert kid(L.F32_HSPLIT, 32, 256, 36) == -3                        # H-split output rows are whole 32-channel groups (retina_reg: out_f32)
    assert kid(L.F32_HSPLIT, 32, 256, 256, res_mode=L.RES_RELU_MASK) == -3     # forward convs only: no residual op


def test_bf16x3_rule_and_enum_values_match_header():
    """ops._mma_dtype_code decides the weight pack AND the launch, so one rule; the enum values are the header's."""
    import re
    import torch
    from efficientdet.pytorch_amd import _lib as L, ops
    h = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'effdet_hip.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    vals = {k: int(v) for k, v in re.findall(r'(EFFDET_(?:F32|BF16|F32_BF16X3|F32_SPLIT|F32_HSPLIT))\s*=\s*(\d+)', h)}
    assert vals == {'EFFDET_F32': L.F32, 'EFFDET_BF16': L.BF16, 'EFFDET_F32_BF16X3': L.F32_BF16X3, 'EFFDET_F32_SPLIT': L.F32_SPLIT,
                    'EFFDET_F32_HSPLIT': L.F32_HSPLIT}
    old = ops.set_f32_arith('bf16x3')
    try:
        f, b = torch.float32, torch.bfloat16
        assert ops._mma_dtype_code(f, 2304, 256) == L.F32_BF16X3 and ops._mma_dtype_code(f, 576, 64) == L.F32_BF16X3
        assert ops._mma_dtype_code(f, 324, 256) == L.F32          # K % 32 != 0
        assert ops._mma_dtype_code(f, 96, 576) == L.F32           # short K: HBM-bound pointwise conv keeps the plain kernel
        assert ops._mma_dtype_code(f, 2304, 16) == L.F32
        assert ops._mma_dtype_code(b, 2304, 256) == L.BF16
        assert ops._mma_dtype_code(f) == L.F32_BF16X3             # weight gradient: no packed operand, no K rule
    finally:
        ops.set_f32_arith(old)
    assert ops._mma_dtype_code(torch.float32, 2304, 256) == L.F32
    import pytest
    with pytest.raises(ValueError):
        ops.set_f32_arith('fp8')


def test_model_arithmetic_names_a_forward_backward_pair():
    """ops.MODEL_ARITH: a model's `f32_arith` names the arithmetic of the launches issued in its forward (ops.F32_ARITH) and the one its
    autograd nodes switch to in backward (ops.F32_ARITH_BWD).  'f32_bwd_bf16x3' (the bench headline) = exact forward, bf16x3 gradients;
    a ParamPrep carries the NAME, so set_prep() restores the forward half and the nodes' ctx.arith the backward half."""
    import pytest
    from efficientdet.pytorch_amd import ops, EfficientDet
    try:
        assert ops.MODEL_ARITH == {'f32': ('f32', 'f32', 'f32'), 'bf16x3': ('bf16x3', 'bf16x3', 'f32'), 'f32_bwd_bf16x3': ('f32', 'bf16x3', 'f32'),
                                   'f32_hf16x3_bwd_bf16x3': ('f32', 'bf16x3', 'f16x3')}
        ops.set_model_arith('f32_hf16x3_bwd_bf16x3')             # round 6 headline: + the RetinaHead's forward convs in the f16x3 form
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD, ops.F32_ARITH_HEAD) == ('f32', 'bf16x3', 'f16x3')
        ops.set_model_arith('f32_bwd_bf16x3')
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD, ops.F32_ARITH_HEAD) == ('f32', 'bf16x3', 'f32')
        with ops.backward_scope(None, 'bf16x3'):                 # an autograd node's backward: its arithmetic inside, the caller's back at exit
            assert (ops.F32_ARITH, ops.F32_ARITH_BWD) == ('bf16x3', 'bf16x3')
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD, ops.F32_ARITH_HEAD) == ('f32', 'bf16x3', 'f32')
        ops.set_f32_arith(ops.F32_ARITH_BWD)                     # what a node's backward does with its ctx.arith
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD) == ('bf16x3', 'bf16x3')
        ops.set_prep(ops.ParamPrep('f32_bwd_bf16x3'))            # the next forward of the model: back to the exact half
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD) == ('f32', 'bf16x3')
        ops.set_prep(ops.ParamPrep('bf16x3'))
        assert (ops.F32_ARITH, ops.F32_ARITH_BWD) == ('bf16x3', 'bf16x3')
        with pytest.raises(ValueError):
            ops.set_model_arith('bf16x6')
        with pytest.raises(ValueError):
            EfficientDet(4, W_bifpn=64, D_bifpn=2, f32_arith='fp8')
        m = EfficientDet(4, W_bifpn=64, D_bifpn=2, f32_arith='f32_bwd_bf16x3')
        assert m.f32_arith == 'f32_bwd_bf16x3'
        import copy
        assert copy.deepcopy(m).f32_arith == 'f32_bwd_bf16x3'
    finally:
        ops.set_prep(None)
        ops.set_f32_arith('f32')


def _wgrad_desc(dtype, B, cin, cout, sizes, k=1, stride=1, pad=0, ldx=None, lddz=None, image_splits=0, want_bias=True, separate=False):
    """effdet_wgrad_t over fake device pointers (planning entry points do no device work).  separate: every level in its own
    'allocation' (offsets from a common base, 4 KiB apart beyond the tensor), as the grouped BiFPN launch passes them."""
    from efficientdet.pytorch_amd import _lib as L
    d = L.WgradDesc()
    d.x, d.dz, d.dw = 0x10000000, 0x50000000, None
    d.dbias = 1 if want_bias else None
    d.dtype, d.B, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l = dtype, B, cin, cout, k, k, stride, pad, pad
    d.ldx, d.lddz, d.nseg, d.image_splits = ldx or cin, lddz or cout, len(sizes), image_splits
    ox = oz = 0
    for i, (h, w) in enumerate(sizes):
        s = d.seg[i]
        ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
        s.H, s.W, s.Ho, s.Wo = h, w, ho, wo
        s.in_off, s.in_bstride, s.out_off, s.out_bstride = ox, h * w * d.ldx, oz, ho * wo * d.lddz
        ox += B * h * w * d.ldx + (1024 * (i + 1) if separate else 0)
        oz += B * ho * wo * d.lddz + (1024 * (i + 1) if separate else 0)
    return d


def test_weight_gradient_planning_entry_points():
    """effdet_conv2d_wgrad_kernel / _splits / _seg_slabs / _workspace_bytes (host planning only): which launches take the thin
    pointwise kernel (and the stem's form of it), that the per-segment slab ranges tile [0, splits) without overlap -- the contract
    the grouped BiFPN weight gradients unpack by -- and that per-image splits land on image boundaries."""
    import ctypes as C
    from efficientdet.pytorch_amd import _lib as L
    lib = L.lib()
    kid = lambda d: int(lib.effdet_conv2d_wgrad_kernel(C.byref(d)))
    # the six high-resolution 1x1 weight gradients of a D0 step (B = 32 @ 512) + the stem: thin kernel, in both fp32-storage modes
    for dt in (L.F32, L.F32_BF16X3):
        for (hw, cin, cout) in [(256, 32, 16), (256, 16, 96), (128, 96, 24), (128, 24, 144), (128, 144, 24), (64, 144, 40)]:
            assert kid(_wgrad_desc(dt, 32, cin, cout, [(hw, hw)])) == 1, (dt, hw, cin, cout)
        assert kid(_wgrad_desc(dt, 32, 4, 32, [(512, 512)], k=3, stride=2)) == 1            # the stem
    assert kid(_wgrad_desc(L.F32, 32, 40, 240, [(64, 64)])) == 0          # 280 channels: tile kernels
    assert kid(_wgrad_desc(L.F32, 32, 40, 144, [(64, 64)])) == 0          # 9 x 3 tiles: no such instantiation
    assert kid(_wgrad_desc(L.F32, 2, 16, 96, [(64, 64)])) == 0            # 8192 pixels: too few
    assert kid(_wgrad_desc(L.F32, 32, 16, 96, [(128, 128)], ldx=32)) == 0  # strided rows
    assert kid(_wgrad_desc(L.F32, 32, 16, 96, [(64, 64), (32, 32)])) == 0  # two levels
    assert kid(_wgrad_desc(L.BF16, 32, 16, 96, [(256, 256)])) == 0        # bf16 storage
    assert kid(_wgrad_desc(L.F32, 32, 8, 32, [(512, 512)], k=3, stride=2)) == 0     # not the stem's 4-channel image
    assert kid(_wgrad_desc(L.F32_SPLIT, 32, 256, 256, [(64, 64), (32, 32)], k=3, pad=1)) == 2
    # per-image splits on the thin kernel: B * q slabs, q | image pixels
    d = _wgrad_desc(L.F32_BF16X3, 32, 96, 24, [(128, 128)], image_splits=1)
    splits = int(lib.effdet_conv2d_wgrad_splits(C.byref(d)))
    assert splits >= 32 and splits % 32 == 0 and (128 * 128) % (splits // 32) == 0
    assert int(lib.effdet_conv2d_wgrad_workspace_bytes(C.byref(d))) == splits * 24 * (96 + 1) * 4
    # grouped independent problems (the BiFPN nodes): 5 levels in separate allocations, slab ranges partition [0, splits)
    for dt, sizes in ((L.F32_BF16X3, [(64, 64), (16, 16), (8, 8), (8, 8), (4, 4)]), (L.F32, [(32, 32), (32, 32), (16, 16)]),
                      (L.BF16, [(64, 64), (5, 7), (8, 8), (3, 3), (4, 4)])):            # (bf16: levels split between two kernels)
        d = _wgrad_desc(dt, 32, 64, 64, sizes, k=3, pad=1, separate=True)
        n = len(sizes)
        first, count = (C.c_int * n)(), (C.c_int * n)()
        tot = int(lib.effdet_conv2d_wgrad_seg_slabs(C.byref(d), first, count))
        assert tot == int(lib.effdet_conv2d_wgrad_splits(C.byref(d))) and tot >= n
        covered = sorted((first[i], first[i] + count[i]) for i in range(n))
        assert covered[0][0] == 0 and covered[-1][1] == tot and all(count[i] >= 1 for i in range(n))
        assert all(covered[i][1] == covered[i + 1][0] for i in range(n - 1)), covered
    assert int(lib.effdet_conv2d_wgrad_seg_slabs(None, None, None)) == -1


def test_tail_batch_context_records_and_joins(monkeypatch):
    """ops.unpack_batch (host logic of effdet_backward_tail): jobs issued inside are recorded, an inner block joins the outer one,
    the launch happens once at the outermost exit, and an exception inside drops the batch instead of launching half a node."""
    from efficientdet.pytorch_amd import ops, _lib as L
    launched = []

    def fake_flush(self):
        launched.append([j.kind for j in self.jobs]); self.jobs, self.keep, self.bytes = [], [], 0
    monkeypatch.setattr(ops._TailBatch, 'flush', fake_flush)
    job = L.UnpackJob()
    with ops.unpack_batch():
        assert ops._cur_batch() is not None
        ops._cur_batch().add(L.TAIL_UNPACK, job, 'keep')
        with ops.unpack_batch():                                  # e.g. bifpn_module_bwd inside _NeckFn.backward
            ops._cur_batch().add(L.TAIL_SE_PARAMS, L.SeParamJob())
        assert launched == []                                      # the inner exit does not launch
        ops._cur_batch().add(L.TAIL_DW_UNPACK, L.DwUnpackJob())
    assert launched == [[L.TAIL_UNPACK, L.TAIL_SE_PARAMS, L.TAIL_DW_UNPACK]] and ops._cur_batch() is None
    with pytest.raises(RuntimeError):
        with ops.unpack_batch():
            ops._cur_batch().add(L.TAIL_UNPACK, job)
            raise RuntimeError('backward failed')
    assert len(launched) == 1 and ops._cur_batch() is None
    monkeypatch.setattr(ops, 'UNPACK_BATCHED', False)              # EFFDET_UNPACK_BATCH=0: the context is a no-op, jobs launch one by one
    with ops.unpack_batch():
        assert ops._cur_batch() is None


def test_tail_batch_is_thread_local_and_bounds_pinned_bytes(monkeypatch):
    """Replica backwards of a multi-device nn.DataParallel run concurrently on per-device autograd threads: a thread must never
    see (or append to) another thread's open batch.  And the deferred jobs pin their workspaces only up to a byte budget."""
    import threading
    import torch
    from efficientdet.pytorch_amd import ops, _lib as L
    launched = []

    def fake_flush(self):
        launched.append((threading.get_ident(), len(self.jobs))); self.jobs, self.keep, self.bytes = [], [], 0
    monkeypatch.setattr(ops._TailBatch, 'flush', fake_flush)
    seen, gate_a, gate_b = {}, threading.Event(), threading.Event()

    def other():
        gate_a.wait(10)
        seen['other_before'] = ops._cur_batch()                   # the main thread's block is open right now
        with ops.unpack_batch():
            seen['other_inside'] = ops._cur_batch()
            ops._cur_batch().add(L.TAIL_UNPACK, L.UnpackJob())
        gate_b.set()
    th = threading.Thread(target=other); th.start()
    with ops.unpack_batch():
        mine = ops._cur_batch()
        mine.add(L.TAIL_UNPACK, L.UnpackJob())
        gate_a.set(); assert gate_b.wait(10)
        assert len(mine.jobs) == 1                                 # the other thread's job did not land here
    th.join()
    assert seen['other_before'] is None and seen['other_inside'] is not None and seen['other_inside'] is not mine
    assert sorted(n for _, n in launched) == [1, 1] and len({t for t, _ in launched}) == 2
    launched.clear()
    monkeypatch.setattr(ops, 'TAIL_KEEP_BYTES', 1000)
    with ops.unpack_batch():
        ops._cur_batch().add(L.TAIL_UNPACK, L.UnpackJob(), torch.empty(100))       # 400 B pinned
        ops._cur_batch().add(L.TAIL_UNPACK, L.UnpackJob(), torch.empty(200))       # 1200 B > budget: both leave now
        assert [n for _, n in launched] == [2] and ops._cur_batch().keep == []
        ops._cur_batch().add(L.TAIL_UNPACK, L.UnpackJob(), torch.empty(10))
    assert [n for _, n in launched] == [2, 1]


def test_capture_probe_port_and_arguments():
    """ddp.probe_port: next to the job's MASTER_PORT, never on it, inside the port range; a multi-rank probe without a shared port is refused."""
    from efficientdet.pytorch_amd import ddp
    for p in (29500, 1024, 64990, 65535):
        q = ddp.probe_port(str(p))
        assert q != p and 1024 <= q < 65536 and abs(q - p) == 101
    with pytest.raises(ValueError):
        ddp.rccl_graph_probe(0, rank=1, world_size=2)


def test_nccl_init_keeps_a_watchdog_query_error_from_killing_the_job(monkeypatch):
    """ddp.init_process_group_from_env('nccl', for_capture=True): the watchdog's spurious hipErrorCapturedEvent (captured DDP step) must
    not be re-thrown into std::terminate, and a retired watchdog must not trip the heartbeat monitor; explicit user settings win.
    Plain eager DDP (the default) keeps torch's failure detection: neither variable is touched."""
    import torch.distributed as dist
    from efficientdet.pytorch_amd import ddp
    calls = []
    monkeypatch.setattr(dist, 'is_initialized', lambda: False)
    monkeypatch.setattr(dist, 'init_process_group', lambda **kw: calls.append(kw))
    for k in ('TORCH_NCCL_RETHROW_CUDA_ERRORS', 'TORCH_NCCL_ENABLE_MONITORING', 'MASTER_ADDR', 'MASTER_PORT'):
        monkeypatch.delenv(k, raising=False)
    ddp.init_process_group_from_env('nccl')
    assert 'TORCH_NCCL_RETHROW_CUDA_ERRORS' not in os.environ and 'TORCH_NCCL_ENABLE_MONITORING' not in os.environ
    ddp.init_process_group_from_env('nccl', for_capture=True)
    assert os.environ['TORCH_NCCL_RETHROW_CUDA_ERRORS'] == '0' and os.environ['TORCH_NCCL_ENABLE_MONITORING'] == '0'
    assert calls == [dict(backend='nccl', init_method='env://')] * 2 and os.environ['MASTER_ADDR'] == '127.0.0.1'
    monkeypatch.setenv('TORCH_NCCL_RETHROW_CUDA_ERRORS', '1')
    ddp.init_process_group_from_env('nccl', for_capture=True)
    assert os.environ['TORCH_NCCL_RETHROW_CUDA_ERRORS'] == '1'
    for k in ('TORCH_NCCL_RETHROW_CUDA_ERRORS', 'TORCH_NCCL_ENABLE_MONITORING'):
        monkeypatch.delenv(k, raising=False)
    ddp.init_process_group_from_env('gloo', for_capture=True)
    assert 'TORCH_NCCL_RETHROW_CUDA_ERRORS' not in os.environ
