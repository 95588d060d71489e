"""``EfficientDet`` -- drop-in for the reference's ``models.efficientdet.EfficientDet`` (models/efficientdet.py:22-100)
whose forward/backward runs on hand-written HIP kernels for MI355X (gfx950).

Same constructor arguments, same ``forward`` contract (train: ``model([images, annotations])`` ->
``(cls_loss[1], reg_loss[1])``; eval: ``model(img)`` -> ``[scores, labels, boxes]``), same public
attributes and the same 426-key ``state_dict`` layout (D0), so the reference's train.py / eval.py call
patterns work unchanged.  The nn.Conv2d / nn.BatchNorm2d children are PARAMETER CONTAINERS only: they are
never called; ``functional.py`` drives libeffdet_hip.so with their tensors.  There is no CPU fallback:
the model refuses to run without the HIP library and a GPU.
"""
import math
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .config import (BN_EPS, DROP_CONNECT_RATE, EFFICIENTDET, MODEL_MAP, backbone_plan)  # noqa: F401
from .ops import Map


# --------------------------------------------------------------------------- parameter containers
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter container: executed through libeffdet_hip.so, not called directly')


class _MBConvParams(_Holder):
    """state_dict keys of models/efficientnet.py:28-73 (MBConvBlock.__init__)."""

    def __init__(self, blk):
        super().__init__()
        if blk.expand != 1:
            self._expand_conv = nn.Conv2d(blk.cin, blk.cexp, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(blk.cexp, momentum=0.01, eps=BN_EPS)
        self._depthwise_conv = nn.Conv2d(blk.cexp, blk.cexp, blk.k, groups=blk.cexp, bias=False)
        self._bn1 = nn.BatchNorm2d(blk.cexp, momentum=0.01, eps=BN_EPS)
        self._se_reduce = nn.Conv2d(blk.cexp, blk.cse, 1)
        self._se_expand = nn.Conv2d(blk.cse, blk.cexp, 1)
        self._project_conv = nn.Conv2d(blk.cexp, blk.cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(blk.cout, momentum=0.01, eps=BN_EPS)


class _Backbone(_Holder):
    """state_dict keys of models/efficientnet.py:122-182 (incl. the never-executed _conv_head/_bn1/_fc)."""

    def __init__(self, name):
        super().__init__()
        stem_c, stem_pad, blocks, head_c, native = backbone_plan(name)
        self.model_name = name                                  # 'efficientnet-bN' (checkpoint.load_pretrained_backbone)
        self.plan = blocks
        self.stem_pad = stem_pad
        self._conv_stem = nn.Conv2d(3, stem_c, 3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(stem_c, momentum=0.01, eps=BN_EPS)
        self._blocks = nn.ModuleList([_MBConvParams(b) for b in blocks])
        self._conv_head = nn.Conv2d(blocks[-1].cout, head_c, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(head_c, momentum=0.01, eps=BN_EPS)
        self._fc = nn.Linear(head_c, 1000)
        self.drop_connect_rate = DROP_CONNECT_RATE

    def get_list_features(self):
        return [b.cout for b in self.plan if b.stage_end]


class _ConvModule(_Holder):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)


class _BiFPNModule(_Holder):
    """models/bifpn.py:133-164."""

    def __init__(self, channels, levels=5):
        super().__init__()
        self.w1 = nn.Parameter(torch.full((2, levels), 0.5))
        self.w2 = nn.Parameter(torch.full((3, levels - 2), 0.5))
        self.bifpn_convs = nn.ModuleList([nn.Sequential(_ConvModule(channels, channels, 3)) for _ in range(2 * (levels - 1))])


class _Neck(_Holder):
    """models/bifpn.py:10-94."""

    def __init__(self, in_channels, out_channels, stack):
        super().__init__()
        self.lateral_convs = nn.ModuleList([_ConvModule(c, out_channels, 1) for c in in_channels])
        self.stack_bifpn_convs = nn.ModuleList([_BiFPNModule(out_channels, len(in_channels)) for _ in range(stack)])


class _Head(_Holder):
    """models/retinahead.py:35-98."""

    def __init__(self, num_classes, in_channels, feat=256):
        super().__init__()
        self.num_classes = num_classes
        self.cls_convs = nn.ModuleList([_ConvModule(in_channels if i == 0 else feat, feat, 3) for i in range(4)])
        self.reg_convs = nn.ModuleList([_ConvModule(in_channels if i == 0 else feat, feat, 3) for i in range(4)])
        self.retina_cls = nn.Conv2d(feat, 9 * num_classes, 3, padding=1)
        self.retina_reg = nn.Conv2d(feat, 36, 3, padding=1)


class Anchors(nn.Module):
    """models/module.py:145-180 on device (float64 arithmetic, bit-exact), cached per input shape."""

    def __init__(self):
        super().__init__()
        self._cache = {}

    def forward(self, image):
        key = (int(image.shape[2]), int(image.shape[3]), str(image.device))
        if key not in self._cache:
            self._cache[key] = ops.anchors(key[0], key[1], image.device)
        return self._cache[key]


class FocalLoss(nn.Module):
    """models/losses.py:29-152 as two HIP passes; forward only (training uses the fused head+loss node)."""

    def forward(self, classifications, regressions, anchors, annotations):
        losses, _ = ops.focal_loss_fwd(classifications.contiguous(), regressions.contiguous(), anchors.contiguous(),
                                       annotations.contiguous().float())
        return losses[0:1], losses[1:2]


class PackedImages:
    """A batch already in the stem conv's input layout (NHWC, compute dtype, channels zero-padded to one 16-byte chunk),
    as produced on the device by data.DeviceCollater -- accepted wherever the model takes an NCHW fp32 image batch, and
    skips the NCHW -> NHWC repack.  Quacks like the [B,3,H,W] tensor it stands for (shape / device / is_cuda / float())."""

    def __init__(self, nhwc_map):
        self.map = nhwc_map
        self.shape = torch.Size((nhwc_map.B, 3, nhwc_map.H, nhwc_map.W))
        self.device = nhwc_map.t.device
        self.is_cuda = nhwc_map.t.is_cuda
        self.dtype = nhwc_map.t.dtype

    def float(self):
        return self

    def contiguous(self):
        return self


# --------------------------------------------------------------------------- autograd nodes
def _t(m):
    return m.tensor()


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, w, gamma, beta, mean, var, pad, dtype, train, link=None):
        ctx.prep, ctx.arith = ops.get_prep(), ops.F32_ARITH_BWD
        if isinstance(img, PackedImages) and (img.map.dtype != dtype or img.map.C != Fn.chunk_elems(dtype)):
            raise RuntimeError('PackedImages were packed for %s / %d channels, the model computes in %s'
                               % (img.map.dtype, img.map.C, dtype))
        y, saved = Fn.stem_fwd(img, w, gamma, beta, mean, var, pad, dtype, train, z_only=link is not None)
        ctx.saved = saved if train else None
        # link to the one consumer (block 0): if its backward applied this Swish' itself (see _MBConvFn) it says so here
        ctx.link = link if train else None
        if link is not None:
            link['z'] = saved[1] if train else None
        return _t(y)

    @staticmethod
    def backward(ctx, dy):
        with ops.backward_scope(ctx.prep, ctx.arith):      # this model's parameter arena + BACKWARD arithmetic; the arithmetic found at entry is put back at exit
            return _StemFn._bwd(ctx, dy)

    @staticmethod
    def _bwd(ctx, dy):
        fused = bool(ctx.link and ctx.link.pop('dz_done', False))
        dw, dg, db = Fn.stem_bwd(ctx.saved, Map.of(dy.contiguous()), dy_is_dz=fused)
        ctx.saved = None
        return None, dw, dg, db, None, None, None, None, None, None


_MB_KEYS = ('expand.weight', 'bn0.weight', 'bn0.bias', 'dw.weight', 'bn1.weight', 'bn1.bias', 'se_reduce.weight',
            'se_reduce.bias', 'se_expand.weight', 'se_expand.bias', 'project.weight', 'bn2.weight', 'bn2.bias')


STEM_LINK = os.environ.get('EFFDET_STEM_LINK', '1') == '1'        # A/B switch: the stem's Swish' inside block 0's depthwise data gradient


class _MBConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, dtype, rowscale, buffers, train, *params):
        ctx.prep, ctx.arith = ops.get_prep(), ops.F32_ARITH_BWD
        P = dict(buffers)
        keys = [k for k in _MB_KEYS if not (blk.expand == 1 and k in ('expand.weight', 'bn0.weight', 'bn0.bias'))]
        P.update(dict(zip(keys, params)))
        # 'stem_link' (block 0 only, expand == 1, no skip): the stem's pre-activation -- this block's depthwise data gradient then
        # returns d(loss)/d(z_stem) and tells the stem node so (one full pass over the 256 x 256 x 32 map less per step)
        link = buffers.get('stem_link') if (train and blk.expand == 1 and not blk.skip and STEM_LINK) else None
        ctx.link = link
        xm = Map.of(x)
        if link is not None:
            # the stem stored its PRE-activation only (stem_fwd z_only) and `x` must BE that tensor: the depthwise kernels Swish
            # it on the fly (in_act) and the backward hands d(z_stem) back.  Anything else between the stem and block 0 (a copy, a
            # cast) would make this block consume un-Swished activations silently -- refuse instead.
            if link.get('z') is None or link['z'].t.data_ptr() != x.data_ptr():
                raise RuntimeError('stem link broken: block 0 was not handed the stem\'s pre-activation tensor itself '
                                   '(EFFDET_STEM_LINK=0 disables the fused stem Swish)')
            xm = link['z']
        y, saved = Fn.mbconv_fwd(xm, blk, P, dtype, train, rowscale, xpre=link['z'] if link else None,
                                 in_act=ops.ACT_SWISH if link is not None else ops.ACT_NONE)
        ctx.saved, ctx.keys = (saved if train else None), keys
        return _t(y)

    @staticmethod
    def backward(ctx, dy):
        with ops.backward_scope(ctx.prep, ctx.arith):      # this model's parameter arena + BACKWARD arithmetic; the arithmetic found at entry is put back at exit
            return _MBConvFn._bwd(ctx, dy)

    @staticmethod
    def _bwd(ctx, dy):
        with ops.unpack_batch():                                   # the node's weight-gradient unpacks leave as one launch
            dx, g = Fn.mbconv_bwd(ctx.saved, Map.of(dy.contiguous()))
        if ctx.saved['blk'].expand == 1 and ctx.saved['blk'].skip:
            ops.add_inplace(dx, Map.of(dy.contiguous()))
        keys = ctx.keys
        if ctx.link is not None:
            ctx.link['dz_done'] = True; ctx.link.pop('z', None)
        ctx.saved = None
        return (_t(dx), None, None, None, None, None) + tuple(g[k] for k in keys)


class _NeckFn(torch.autograd.Function):
    """laterals + all stacked BiFPN modules as one node (5 feature maps in, 5 out)."""

    @staticmethod
    def forward(ctx, dtype, nlev, stack, train, *args):
        ctx.prep, ctx.arith = ops.get_prep(), ops.F32_ARITH_BWD
        feats = [Map.of(t) for t in args[:nlev]]
        rest = args[nlev:]
        lw, lb = rest[0:nlev], rest[nlev:2 * nlev]
        rest = rest[2 * nlev:]
        W = lw[0].shape[0]
        p = Fn.lateral_fwd(feats, lw, lb, W, dtype)
        saved_mods = []
        for s in range(stack):
            w1, w2 = rest[0], rest[1]
            cw, cb = rest[2:10], rest[10:18]
            rest = rest[18:]
            p, sv = Fn.bifpn_module_fwd(p, w1, w2, cw, cb, dtype, train)
            saved_mods.append(sv)
        ctx.saved = (feats, lw, saved_mods, dtype, nlev, stack) if train else None
        return tuple(_t(m) for m in p)

    @staticmethod
    def backward(ctx, *douts):
        with ops.backward_scope(ctx.prep, ctx.arith):      # this model's parameter arena + BACKWARD arithmetic; the arithmetic found at entry is put back at exit
            return _NeckFn._bwd(ctx, *douts)

    @staticmethod
    def _bwd(ctx, *douts):
        feats, lw, saved_mods, dtype, nlev, stack = ctx.saved
        d = [Map.of(t.contiguous()) for t in douts]
        mod_grads = []
        with ops.unpack_batch():                                   # 8 unpacks per module + 5 laterals: two launches for the node
            for k, sv in enumerate(reversed(saved_mods)):
                d, dw1, dw2, dcw, dcb = Fn.bifpn_module_bwd(sv, d, dtype, own=k > 0)
                mod_grads.append((dw1, dw2) + tuple(dcw) + tuple(dcb))
            dfs, dlw, dlb = Fn.lateral_bwd(feats, lw, d, dtype)
        ctx.saved = None
        out = (None, None, None, None) + tuple(_t(m) for m in dfs) + tuple(dlw) + tuple(dlb)
        for mg in reversed(mod_grads):
            out += mg
        return out


_HEAD_KEYS = [f'{t}_convs.{i}.{k}' for t in ('cls', 'reg') for i in range(4) for k in ('weight', 'bias')] + \
             ['retina_cls.weight', 'retina_cls.bias', 'retina_reg.weight', 'retina_reg.bias']


class _HeadFn(torch.autograd.Function):
    """RetinaHead alone as a differentiable node: (classification [B,A,nc] probabilities, regression [B,A,4]) of
    models/efficientdet.py:64-66 for callers that bring their own criterion (the training forward uses the fused
    head+loss node below, which never materialises d(probabilities))."""

    @staticmethod
    def forward(ctx, dtype, num_classes, train, *args):
        ctx.prep, ctx.arith = ops.get_prep(), ops.F32_ARITH_BWD
        p = [Map.of(t) for t in args[:5]]
        HP = dict(zip(_HEAD_KEYS, args[5:]))
        cls, reg, saved = Fn.head_fwd(p, HP, num_classes, dtype, train)
        ctx.saved = (saved, cls, dtype) if train else None
        return cls, reg

    @staticmethod
    def backward(ctx, dcls, dreg):
        with ops.backward_scope(ctx.prep, ctx.arith):      # this model's parameter arena + BACKWARD arithmetic; the arithmetic found at entry is put back at exit
            return _HeadFn._bwd(ctx, dcls, dreg)

    @staticmethod
    def _bwd(ctx, dcls, dreg):
        saved, cls, dtype = ctx.saved
        dlogit, dr = ops.head_out_bwd(dcls.contiguous().float(), cls, dreg.contiguous().float(), dtype)
        with ops.unpack_batch():
            dp, g = Fn.head_bwd(saved, dlogit, dr, dtype)
        ctx.saved = None
        return (None, None, None) + tuple(Fn.level_tensor(m) for m in dp) + tuple(g[k] for k in _HEAD_KEYS)


LOSS_FWD_GRAD = os.environ.get('EFFDET_LOSS_FWD_GRAD', '1') == '1'    # A/B switch: class-loss gradient written in forward


class _HeadLossFn(torch.autograd.Function):
    """RetinaHead + focal / smooth-L1 loss as ONE node: the loss kernel hands the head's data-gradient convs
    d(logit) and d(reg) directly in the activation dtype (no fp32 gradient tensor round trip)."""

    @staticmethod
    def forward(ctx, dtype, num_classes, anchors, annots, train, *args):
        ctx.prep, ctx.arith = ops.get_prep(), ops.F32_ARITH_BWD
        p = [Map.of(t) for t in args[:5]]
        HP = dict(zip(_HEAD_KEYS, args[5:]))
        cls, reg, saved = Fn.head_fwd(p, HP, num_classes, dtype, train)
        nc = num_classes
        if train and nc % 4 == 0 and LOSS_FWD_GRAD:
            # ONE pass over the 15.7 MB/image of probabilities: losses + d(logits) for an upstream gradient of one, already in
            # the pixel-major, 64-channel-padded rows the head's gradient convs read; cls itself is not kept for backward
            dld = (9 * nc + 63) // 64 * 64
            losses, ws, dpix = ops.focal_loss_fwd_grad(cls, reg, anchors, annots, dtype, dld, split=saved[5])    # (split-layout head: see functional.head_uses_split)
            ctx.saved = (saved, None, reg, anchors, annots, ws, dtype, dpix, dld)
        else:
            losses, ws = ops.focal_loss_fwd(cls, reg, anchors, annots)
            ctx.saved = (saved, cls, reg, anchors, annots, ws, dtype, None, 0) if train else None
        return losses[0:1].clone(), losses[1:2].clone()

    @staticmethod
    def backward(ctx, gcls, greg):
        with ops.backward_scope(ctx.prep, ctx.arith):      # this model's parameter arena + BACKWARD arithmetic; the arithmetic found at entry is put back at exit
            return _HeadLossFn._bwd(ctx, gcls, greg)

    @staticmethod
    def _bwd(ctx, gcls, greg):
        saved, cls, reg, anchors, annots, ws, dtype, dpix, dld = ctx.saved
        gscale = torch.cat([gcls.reshape(1), greg.reshape(1)]).float().contiguous()
        if dpix is not None:
            split = saved[5]
            rld = 64 if split else 0                # split layout: d(reg) pixel-major, 36 -> 64 channels (two [hi|lo] groups)
            dreg = ops.focal_loss_bwd_reg(reg, anchors, annots, gscale, ws, dtype, reg_ld=rld, split=split)
            with ops.unpack_batch():                               # the head's 10 weight-gradient unpacks: one launch
                dp, g = Fn.head_bwd(saved, dpix, dreg, dtype, dcls_ld=dld, cls_gscale=gscale[0:1], dreg_ld=rld, in_split=split)
        else:
            nc = cls.shape[2]
            if nc % 4 == 0:      # d(logits) straight into the pixel-major, 64-channel-padded rows the head's gradient convs read
                dld = (9 * nc + 63) // 64 * 64
                dcls, dreg = ops.focal_loss_bwd_pix(cls, reg, anchors, annots, gscale, ws, dtype, dld)
            else:
                dld = 0
                dcls, dreg = ops.focal_loss_bwd(cls, reg, anchors, annots, gscale, ws, dtype)
            with ops.unpack_batch():
                dp, g = Fn.head_bwd(saved, dcls, dreg, dtype, dcls_ld=dld)
        ctx.saved = None
        return (None, None, None, None, None) + tuple(Fn.level_tensor(m) for m in dp) + tuple(g[k] for k in _HEAD_KEYS)


# --------------------------------------------------------------------------- the model
class EfficientDet(nn.Module):
    def __init__(self, num_classes, network='efficientdet-d0', D_bifpn=3, W_bifpn=88, D_class=3, is_training=True,
                 threshold=0.01, iou_threshold=0.5, compute_dtype=torch.float32, f32_arith='f32'):
        super().__init__()
        self.backbone = _Backbone(MODEL_MAP[network])          # KeyError on a bad name, like the reference
        self.is_training = is_training
        self.neck = _Neck(self.backbone.get_list_features()[-5:], W_bifpn, D_bifpn)
        self.bbox_head = _Head(num_classes, W_bifpn)            # D_class is accepted and ignored (reference Q3)
        self.anchors = Anchors()
        self.threshold = threshold
        self.iou_threshold = iou_threshold
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype
        # MFMA arithmetic on fp32 storage: 'f32' = exact fp32 products (v_mfma_f32_16x16x4_f32), 'bf16x3' = operands split into
        # bf16 hi + lo, three bf16 MFMAs per product (~16 mantissa bits; ~2x the fp32 matrix-pipe rate), 'f32_bwd_bf16x3' = every
        # forward value in exact fp32 (bit for bit the 'f32' forward), only the gradient convolutions in the bf16x3 form.
        # Ignored for bf16.
        if f32_arith not in ops.MODEL_ARITH:
            raise ValueError('f32_arith must be one of %s' % (sorted(ops.MODEL_ARITH),))
        self.f32_arith = f32_arith
        self._prep = {}                                         # (compute dtype, device) -> ops.ParamPrep (batched per-step repacks)
        self._dc = {}                                           # drop_connect generator state (seed, step counter, per-device keep table)
        self.batched_prep = True                                # False: every repack is its own launch (debug / A-B)
        for m in self.modules():                                # models/efficientdet.py:47-53
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self.freeze_bn()
        self.criterion = FocalLoss()

    # ---- reference surface ----
    def freeze_bn(self):
        """BatchNorm is always the frozen per-channel affine here (models/efficientdet.py:88-92)."""
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def live_parameters(self):
        """Parameters that receive gradients (everything except the 5 never-executed backbone tensors)."""
        dead = {id(p) for p in list(self.backbone._conv_head.parameters()) + list(self.backbone._bn1.parameters()) +
                list(self.backbone._fc.parameters())}
        return [p for p in self.parameters() if id(p) not in dead]

    def _check(self, img):
        if not img.is_cuda:
            raise RuntimeError('efficientdet.pytorch_amd runs on MI355X only: inputs must be on a GPU (no CPU fallback)')

    def _apply(self, fn, *a, **k):
        self._prep = {}                                         # parameter storage may move: drop the recorded job tables
        return super()._apply(fn, *a, **k)

    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save(model): the recorded job tables hold raw device addresses of THIS module's
        parameters and arenas -- a copy must record its own (a replay against the original's addresses would write into
        foreign or freed memory)."""
        st = self.__dict__.copy()
        st['_prep'] = {}
        st['_dc'] = {k: v for k, v in self._dc.items() if k in ('seed', 'step')}
        return st

    def _drop_connect_rowscales(self, B, device):
        """{block index: rowscale [B] fp32} for this training step: floor(keep + u)/keep per identity-skip block
        (models/efficientnet.py:199-203 rate schedule, models/utils.py:79-90), all blocks from ONE HIP launch
        (Philox4x32-10 keyed by (seed, step)).  backbone.drop_masks = {block: 0/1 tensor [B]} overrides the generator
        (parity tests inject the reference's own Bernoulli draws)."""
        bb = self.backbone
        nblk = len(bb.plan)
        idx = [i for i, blk in enumerate(bb.plan) if blk.skip and bb.drop_connect_rate * float(i) / nblk]
        if not idx:
            return {}
        keep = [1.0 - bb.drop_connect_rate * float(i) / nblk for i in idx]
        inject = getattr(bb, 'drop_masks', None)
        if inject is not None:
            return {i: (inject[i].to(device=device, dtype=torch.float32) / kp).contiguous() for i, kp in zip(idx, keep) if i in inject}
        dc = self._dc
        if 'seed' not in dc:
            # one draw from torch's default generator: torch.manual_seed() controls the stream; ranks decorrelate by rank
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed ^= (torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15
            dc['seed'], dc['step'] = seed & (2 ** 64 - 1), 0
        tk = ('keep', str(device), bb.drop_connect_rate)
        if tk not in dc:
            dc[tk] = torch.tensor(keep, dtype=torch.float32, device=device)
        sk = ('step_dev', str(device))
        if sk not in dc:                                         # the step counter lives on the device: graph replays advance it too
            dc[sk] = torch.tensor([dc['step']], dtype=torch.int64, device=device)
        rows = ops.drop_connect_scales(dc[tk], B, dc['seed'], 0, dc[sk])
        dc['step'] += 1
        return {i: rows[j] for j, i in enumerate(idx)}

    def _backbone(self, img):
        bb, dt = self.backbone, self.compute_dtype
        # every forward path starts here: replay (or start recording) this model's batched parameter preparation
        key = (dt, img.device, self.f32_arith)
        ops.set_model_arith(self.f32_arith)
        if ops.F32_ARITH_HEAD == 'f16x3' and not torch.cuda.is_current_stream_capturing():
            ops.range_flag(img.device)            # (the out-of-range watch word exists before anything could be captured)
        if not self.batched_prep or getattr(self, '_is_replica', False):
            # (replicas of nn.DataParallel are rebuilt every forward with fresh parameter tensors: nothing to record against)
            ops.set_prep(None)
        else:
            if key not in self._prep:
                self._prep[key] = ops.ParamPrep(self.f32_arith)
            ops.set_prep(self._prep[key])
            self._prep[key].begin_step(img.device)
        bn = bb._bn0
        train = torch.is_grad_enabled()
        link = {} if (train and STEM_LINK and bb.plan[0].expand == 1 and not bb.plan[0].skip) else None
        x = _StemFn.apply(img, bb._conv_stem.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bb.stem_pad, dt, train, link)
        feats = []
        rowscales = self._drop_connect_rowscales(int(img.shape[0]), img.device) if (self.training and bb.drop_connect_rate) else {}
        for i, (blk, m) in enumerate(zip(bb.plan, bb._blocks)):
            buffers = {'bn1.running_mean': m._bn1.running_mean, 'bn1.running_var': m._bn1.running_var,
                       'bn2.running_mean': m._bn2.running_mean, 'bn2.running_var': m._bn2.running_var}
            params = []
            if blk.expand != 1:
                buffers.update({'bn0.running_mean': m._bn0.running_mean, 'bn0.running_var': m._bn0.running_var})
                params += [m._expand_conv.weight, m._bn0.weight, m._bn0.bias]
            params += [m._depthwise_conv.weight, m._bn1.weight, m._bn1.bias, m._se_reduce.weight, m._se_reduce.bias,
                       m._se_expand.weight, m._se_expand.bias, m._project_conv.weight, m._bn2.weight, m._bn2.bias]
            if i == 0 and link is not None:
                buffers['stem_link'] = link
            rowscale = rowscales.get(i)                         # models/efficientnet.py:199-203, models/utils.py:79-90
            x = _MBConvFn.apply(x, blk, dt, rowscale, buffers, train, *params)
            if blk.stage_end:
                feats.append(x)
        return feats

    def _neck(self, feats):
        nk = self.neck
        args = list(feats) + [c.conv.weight for c in nk.lateral_convs] + [c.conv.bias for c in nk.lateral_convs]
        for mod in nk.stack_bifpn_convs:
            args += [mod.w1, mod.w2] + [s[0].conv.weight for s in mod.bifpn_convs] + [s[0].conv.bias for s in mod.bifpn_convs]
        return _NeckFn.apply(self.compute_dtype, len(feats), len(nk.stack_bifpn_convs), torch.is_grad_enabled(), *args)

    def _head_params(self):
        h = self.bbox_head
        ps = []
        for tower in (h.cls_convs, h.reg_convs):
            for c in tower:
                ps += [c.conv.weight, c.conv.bias]
        return ps + [h.retina_cls.weight, h.retina_cls.bias, h.retina_reg.weight, h.retina_reg.bias]

    def extract_feat(self, img):
        """Backbone + neck features as NCHW fp32 tensors (the reference's public helper, :94-100)."""
        self._check(img)
        with torch.no_grad():
            p = self._neck(self._backbone(img)[-5:])
            return tuple(ops.nhwc_to_nchw(Map.of(t)) for t in p)

    def forward_raw(self, img):
        """(classification [B,A,nc] probabilities, regression [B,A,4], anchors [1,A,4]) of models/efficientdet.py:64-66.
        Differentiable when gradients are enabled (backbone, neck and head nodes all record), so a caller may apply its
        own criterion to the triple the way the reference's forward does (:67)."""
        self._check(img)
        p = self._neck(self._backbone(img.float())[-5:])
        cls, reg = _HeadFn.apply(self.compute_dtype, self.num_classes, torch.is_grad_enabled(), *p, *self._head_params())
        return cls, reg, self.anchors(img)

    def detect(self, img):
        """Eval post-processing for EVERY image of the batch (the reference handles image 0 only).
        -> list of (scores[K], labels[K] int64, boxes[K,4]) per image, score-descending."""
        with torch.no_grad():
            cls, reg, anc = self.forward_raw(img)
        H, W = int(img.shape[2]), int(img.shape[3])
        boxes, score, label = ops.decode_score(anc, reg, cls, H, W)
        idx, count = ops.nms(boxes, score, float(self.threshold), float(self.iou_threshold))
        s, l, b = ops.gather_dets(boxes, score, label, idx, count)
        counts = count.tolist()                                  # the one device->host sync (the reference syncs too)
        if ops.MODEL_ARITH[self.f32_arith][2] == 'f16x3' and not torch.cuda.is_current_stream_capturing():
            ops.check_range_flag(s.device)                       # (a sigmoid turns an inf logit into a plausible score: make overflow an error)
        return [(s[i, :n], l[i, :n], b[i, :n]) for i, n in enumerate(counts)]

    def forward(self, inputs):
        if self.is_training:
            inputs, annotations = inputs
            self._check(inputs)
            p = self._neck(self._backbone(inputs.float())[-5:])
            anc = self.anchors(inputs)
            return _HeadLossFn.apply(self.compute_dtype, self.num_classes, anc, annotations.float().contiguous(),
                                     torch.is_grad_enabled(), *p, *self._head_params())
        dets = self.detect(inputs)
        s, l, b = dets[0]
        if s.numel() == 0:
            print('No boxes to NMS')
            return [torch.zeros(0), torch.zeros(0), torch.zeros(0, 4)]
        return [s, l, b]
