"""CPU oracle for the EfficientDet forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file.  The product package
(``efficientdet.pytorch_amd``) never imports it and has no CPU fallback.

What it is: a functional, torch-CPU fp32 restatement of the reference algorithm
(toandaominh1997/EfficientDet.Pytorch), written from the reference's behaviour,
each function citing the reference file:line it follows (paths relative to the
reference root).  It is *pinned* against the real reference executed in the build
container: ``oracle/make_golden.py`` imports the reference itself (with the three
shims of SURVEY.md §8c) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.

NMS arithmetic lives in a third-party dependency (torchvision, unpinned:
requirements.txt:4) that is absent from the reference tree, so ``nms_greedy``
restates torchvision.ops.nms' published semantics -> "parity unpinned" for NMS
(the only call site is models/efficientdet.py:82-83).

Everything is a pure function of (state_dict, inputs); parameters use the
reference's state_dict key layout so the same dict feeds the HIP path.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# architecture arithmetic (models/utils.py:55-76, 171-302; models/efficientnet.py:122-182)
# --------------------------------------------------------------------------

# models/efficientdet.py:10-19
MODEL_MAP = {f'efficientdet-d{i}': f'efficientnet-b{min(i, 6)}' for i in range(8)}

# models/utils.py:171-184  (width, depth, native resolution)
_EFFNET_COEF = {
    'efficientnet-b0': (1.0, 1.0, 224), 'efficientnet-b1': (1.0, 1.1, 240),
    'efficientnet-b2': (1.1, 1.2, 260), 'efficientnet-b3': (1.2, 1.4, 300),
    'efficientnet-b4': (1.4, 1.8, 380), 'efficientnet-b5': (1.6, 2.2, 456),
    'efficientnet-b6': (1.8, 2.6, 528), 'efficientnet-b7': (2.0, 3.1, 600),
}

# models/utils.py:264-269 -- note stages 5 and 7 are stride 2 in this reference
# (repeat, kernel, stride, expand, in, out)
_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 2, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 2, 6, 192, 320)]

# utils/config_eff.py:1-41
EFFICIENTDET = {
    'efficientdet-d0': dict(input_size=512, W_bifpn=64, D_bifpn=2, D_class=3),
    'efficientdet-d1': dict(input_size=640, W_bifpn=88, D_bifpn=3, D_class=3),
    'efficientdet-d2': dict(input_size=768, W_bifpn=112, D_bifpn=4, D_class=3),
    'efficientdet-d3': dict(input_size=896, W_bifpn=160, D_bifpn=5, D_class=4),
    'efficientdet-d4': dict(input_size=1024, W_bifpn=224, D_bifpn=6, D_class=4),
    'efficientdet-d5': dict(input_size=1280, W_bifpn=288, D_bifpn=7, D_class=4),
    'efficientdet-d6': dict(input_size=1408, W_bifpn=384, D_bifpn=8, D_class=5),
    'efficientdet-d7': dict(input_size=1636, W_bifpn=384, D_bifpn=8, D_class=5),
}


def round_filters(filters, width, divisor=8):
    """models/utils.py:55-68."""
    if not width:
        return filters
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth):
    """models/utils.py:71-76."""
    return int(math.ceil(depth * repeats)) if depth else repeats


def same_pad(native, k, s):
    """Static TF-'same' padding computed ONCE from the backbone's native resolution
    (models/utils.py:126-155).  Returns (lo, hi) for one axis."""
    o = math.ceil(native / s)
    p = max((o - 1) * s + (k - 1) + 1 - native, 0)
    return p // 2, p - p // 2


def backbone_blocks(network):
    """Expanded MBConv block list: models/efficientnet.py:146-170 + models/utils.py:187-257."""
    width, depth, native = _EFFNET_COEF[MODEL_MAP[network]]
    blocks = []
    for (r, k, s, e, ci, co) in _STAGES:
        ci, co, r = round_filters(ci, width), round_filters(co, width), round_repeats(r, depth)
        for j in range(r):
            first = (j == 0)
            cin = ci if first else co
            stride = s if first else 1
            blocks.append(dict(
                k=k, s=stride, e=e, cin=cin, cout=co, cexp=cin * e,
                cse=max(1, int(cin * 0.25)),                       # efficientnet.py:60-61
                pad=same_pad(native, k, stride),
                # efficientnet.py:98-99: ``stride == 1`` is False for the list-valued stride of the
                # first block of a stage, so only repeat blocks (int stride 1) take the skip.
                skip=(not first) and cin == co,
                stage_end=(j == r - 1)))
    stem = dict(cout=round_filters(32, width), pad=same_pad(native, 3, 2))
    head_c = round_filters(1280, width)
    return stem, blocks, head_c, native


# --------------------------------------------------------------------------
# deterministic random-init state_dict in the reference's key layout
# --------------------------------------------------------------------------

def make_state_dict(network='efficientdet-d0', num_classes=80, W_bifpn=None, D_bifpn=None,
                    seed=0, randomize_bn=True, bn2_gain=1.0):
    """Deterministic weights shared by oracle, reference (via load_state_dict) and HIP path.
    Conv weights follow the reference's own init N(0, sqrt(2/(k*k*Cout)))
    (models/efficientdet.py:47-53); BN statistics/affine, biases and fusion weights are randomised
    mildly so every term of the arithmetic is exercised (the reference's defaults of 0/1 would
    hide bugs).  Key set == reference ``EfficientDet(...).state_dict()`` (checked in make_golden).
    ``bn2_gain`` scales the affine weight of every MBConv project BN: the reference's fan-out init makes the signal grow from block
    to block, and over the 45 blocks of B6 the loss lands at ~50 where the reference's own gradients move by 1e-3 under a 3e-7
    input scaling (tools/flip_sensitivity.py) -- the D6 fixtures use 0.5 so that they test kernels, not chaos."""
    cfg = EFFICIENTDET[network]
    W = cfg['W_bifpn'] if W_bifpn is None else W_bifpn
    D = cfg['D_bifpn'] if D_bifpn is None else D_bifpn
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, co, ci, k, bias=False):
        std = math.sqrt(2.0 / (k * k * co))
        sd[name + '.weight'] = torch.randn(co, ci, k, k, generator=g) * std
        if bias:
            sd[name + '.bias'] = torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        if randomize_bn:
            sd[name + '.weight'] = 0.75 + 0.5 * torch.rand(c, generator=g)
            sd[name + '.bias'] = torch.randn(c, generator=g) * 0.1
            sd[name + '.running_mean'] = torch.randn(c, generator=g) * 0.1
            sd[name + '.running_var'] = 0.5 + torch.rand(c, generator=g)
        else:
            sd[name + '.weight'] = torch.ones(c)
            sd[name + '.bias'] = torch.zeros(c)
            sd[name + '.running_mean'] = torch.zeros(c)
            sd[name + '.running_var'] = torch.ones(c)
        sd[name + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)

    stem, blocks, head_c, _ = backbone_blocks(network)
    conv('backbone._conv_stem', stem['cout'], 3, 3)
    bn('backbone._bn0', stem['cout'])
    for i, b in enumerate(blocks):
        p = f'backbone._blocks.{i}.'
        if b['e'] != 1:
            conv(p + '_expand_conv', b['cexp'], b['cin'], 1)
            bn(p + '_bn0', b['cexp'])
        # depthwise fan-in is only k*k: use fan-in scaling so the signal survives 16 blocks
        sd[p + '_depthwise_conv.weight'] = torch.randn(b['cexp'], 1, b['k'], b['k'], generator=g) * \
            math.sqrt(2.0 / (b['k'] * b['k']))
        bn(p + '_bn1', b['cexp'])
        conv(p + '_se_reduce', b['cse'], b['cexp'], 1, bias=True)
        conv(p + '_se_expand', b['cexp'], b['cse'], 1, bias=True)
        conv(p + '_project_conv', b['cout'], b['cexp'], 1)
        bn(p + '_bn2', b['cout'])
        if bn2_gain != 1.0:
            sd[p + '_bn2.weight'] = sd[p + '_bn2.weight'] * bn2_gain
    # dead tensors that exist in the reference state_dict but are never executed
    conv('backbone._conv_head', head_c, blocks[-1]['cout'], 1)
    bn('backbone._bn1', head_c)
    sd['backbone._fc.weight'] = torch.randn(1000, head_c, generator=g) * 0.01
    sd['backbone._fc.bias'] = torch.zeros(1000)

    feats = [b['cout'] for b in blocks if b['stage_end']][-5:]
    for l, c in enumerate(feats):
        conv(f'neck.lateral_convs.{l}.conv', W, c, 1, bias=True)
    for s in range(D):
        q = f'neck.stack_bifpn_convs.{s}.'
        w1 = 0.2 + 0.8 * torch.rand(2, 5, generator=g)
        w2 = 0.2 + 0.8 * torch.rand(3, 3, generator=g)
        w1[1, 2] = -0.3          # one negative entry exercises the ReLU clamp (bifpn.py:177)
        sd[q + 'w1'], sd[q + 'w2'] = w1, w2
        for c in range(8):
            conv(q + f'bifpn_convs.{c}.0.conv', W, W, 3, bias=True)
    for t in range(4):
        conv(f'bbox_head.cls_convs.{t}.conv', 256, W if t == 0 else 256, 3, bias=True)
    for t in range(4):
        conv(f'bbox_head.reg_convs.{t}.conv', 256, W if t == 0 else 256, 3, bias=True)
    conv('bbox_head.retina_cls', 9 * num_classes, 256, 3, bias=True)
    conv('bbox_head.retina_reg', 36, 256, 3, bias=True)
    return sd


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------

BN_EPS = 1e-3          # models/utils.py:273  (batch_norm_epsilon)


def _bn(sd, name, x):
    """Frozen (eval-mode) BatchNorm -- models/efficientdet.py:88-92, SURVEY Q6."""
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], False, 0.0, BN_EPS)


def _swish(x):
    """models/utils.py:31-52."""
    return x * torch.sigmoid(x)


def _conv_same(x, w, bias, stride, pad, groups=1):
    """ZeroPad2d(static) then conv with padding 0 -- models/utils.py:151-155."""
    lo, hi = pad
    if lo or hi:
        x = F.pad(x, [lo, hi, lo, hi])
    return F.conv2d(x, w, bias, stride, 0, 1, groups)


def backbone_forward(sd, network, img, drop_masks=None, keep=None):
    """models/efficientnet.py:190-209 (+ MBConvBlock.forward :75-105).  Returns the 7 stage outputs.
    ``drop_masks``: optional {block_idx: tensor[B] of 0/1} Bernoulli keep masks for drop_connect
    (models/utils.py:79-90); None = drop_connect disabled (eval, or parity runs)."""
    stem, blocks, _, _ = backbone_blocks(network)
    x = _swish(_bn(sd, 'backbone._bn0', _conv_same(img, sd['backbone._conv_stem.weight'], None, 2, stem['pad'])))
    outs, taps = [], OrderedDict(stem=x)
    for i, b in enumerate(blocks):
        p = f'backbone._blocks.{i}.'
        inp = x
        if b['e'] != 1:
            x = _swish(_bn(sd, p + '_bn0', F.conv2d(x, sd[p + '_expand_conv.weight'])))
        x = _swish(_bn(sd, p + '_bn1', _conv_same(x, sd[p + '_depthwise_conv.weight'], None,
                                                    b['s'], b['pad'], groups=b['cexp'])))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = F.conv2d(_swish(F.conv2d(sq, sd[p + '_se_reduce.weight'], sd[p + '_se_reduce.bias'])),
                      sd[p + '_se_expand.weight'], sd[p + '_se_expand.bias'])
        x = torch.sigmoid(sq) * x
        x = _bn(sd, p + '_bn2', F.conv2d(x, sd[p + '_project_conv.weight']))
        if b['skip']:
            if drop_masks is not None and i in drop_masks:
                kp = keep[i]
                x = x / kp * drop_masks[i].view(-1, 1, 1, 1).to(x.dtype)
            x = x + inp
        taps[f'block{i}'] = x
        if b['stage_end']:
            outs.append(x)
    return outs, taps


def bifpn_forward(sd, feats, D):
    """models/bifpn.py:96-129 and BiFPNModule.forward :172-203 (double normalisation, SURVEY Q2;
    top-down results overwrite the list the bottom-up pass reads, originals kept in a clone)."""
    eps = 1e-4
    p = [F.conv2d(f, sd[f'neck.lateral_convs.{l}.conv.weight'], sd[f'neck.lateral_convs.{l}.conv.bias'])
         for l, f in enumerate(feats)]
    taps = OrderedDict((f'lateral{l}', t) for l, t in enumerate(p))
    L = len(p)
    for s in range(D):
        q = f'neck.stack_bifpn_convs.{s}.'

        def cv(c, t):
            return F.conv2d(t, sd[q + f'bifpn_convs.{c}.0.conv.weight'], sd[q + f'bifpn_convs.{c}.0.conv.bias'], 1, 1)
        w1 = F.relu(sd[q + 'w1']); w1 = w1 / (w1.sum(0) + eps)       # bifpn.py:177-178 (out-of-place, Q10)
        w2 = F.relu(sd[q + 'w2']); w2 = w2 / (w2.sum(0) + eps)       # bifpn.py:179-180
        orig = [t.clone() for t in p]
        c = 0
        for i in range(L - 1, 0, -1):                                  # bifpn.py:188-192
            p[i - 1] = (w1[0, i - 1] * p[i - 1] + w1[1, i - 1] * F.interpolate(p[i], scale_factor=2, mode='nearest')) \
                / (w1[0, i - 1] + w1[1, i - 1] + eps)
            p[i - 1] = cv(c, p[i - 1]); c += 1
        for i in range(0, L - 2):                                      # bifpn.py:194-198
            p[i + 1] = (w2[0, i] * p[i + 1] + w2[1, i] * F.max_pool2d(p[i], 2) + w2[2, i] * orig[i + 1]) \
                / (w2[0, i] + w2[1, i] + w2[2, i] + eps)
            p[i + 1] = cv(c, p[i + 1]); c += 1
        p[L - 1] = (w1[0, L - 1] * p[L - 1] + w1[1, L - 1] * F.max_pool2d(p[L - 2], 2)) \
            / (w1[0, L - 1] + w1[1, L - 1] + eps)                      # bifpn.py:200-202
        p[L - 1] = cv(c, p[L - 1])
        for l, t in enumerate(p):
            taps[f'bifpn{s}_p{l}'] = t
    return p, taps


def head_forward(sd, feats, num_classes):
    """models/retinahead.py:109-132 (shared weights over levels; sigmoid on cls, Q4; NHWC flatten)."""
    cls_out, reg_out = [], []
    for f in feats:
        c = r = f
        for t in range(4):
            c = F.relu(F.conv2d(c, sd[f'bbox_head.cls_convs.{t}.conv.weight'], sd[f'bbox_head.cls_convs.{t}.conv.bias'], 1, 1))
            r = F.relu(F.conv2d(r, sd[f'bbox_head.reg_convs.{t}.conv.weight'], sd[f'bbox_head.reg_convs.{t}.conv.bias'], 1, 1))
        c = torch.sigmoid(F.conv2d(c, sd['bbox_head.retina_cls.weight'], sd['bbox_head.retina_cls.bias'], 1, 1))
        r = F.conv2d(r, sd['bbox_head.retina_reg.weight'], sd['bbox_head.retina_reg.bias'], 1, 1)
        B = f.shape[0]
        cls_out.append(c.permute(0, 2, 3, 1).reshape(B, -1, num_classes))
        reg_out.append(r.permute(0, 2, 3, 1).reshape(B, -1, 4))
    return torch.cat(cls_out, 1), torch.cat(reg_out, 1)          # models/efficientdet.py:64-65


def anchors_for_image(H, W):
    """models/module.py:145-214,252-273: float64 NumPy arithmetic, cast to float32 at the end (Q14).
    Order: level, y, x, (ratio-major, scale-minor)."""
    ratios = np.array([0.5, 1, 2]); scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])
    out = []
    for lvl in (3, 4, 5, 6, 7):
        stride, base = 2 ** lvl, 2 ** (lvl + 2)
        a = np.zeros((9, 4))
        a[:, 2:] = base * np.tile(scales, (2, 3)).T
        areas = a[:, 2] * a[:, 3]
        a[:, 2] = np.sqrt(areas / np.repeat(ratios, 3))
        a[:, 3] = a[:, 2] * np.repeat(ratios, 3)
        a[:, 0::2] -= np.tile(a[:, 2] * 0.5, (2, 1)).T
        a[:, 1::2] -= np.tile(a[:, 3] * 0.5, (2, 1)).T
        fh, fw = (H + stride - 1) // stride, (W + stride - 1) // stride
        sx = (np.arange(0, fw) + 0.5) * stride
        sy = (np.arange(0, fh) + 0.5) * stride
        sx, sy = np.meshgrid(sx, sy)
        shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
        out.append((a.reshape(1, 9, 4) + shifts.reshape(1, -1, 4).transpose(1, 0, 2)).reshape(-1, 4))
    return torch.from_numpy(np.concatenate(out, 0)[None].astype(np.float32))


def decode_clip(anchors, regression, H, W):
    """models/module.py:24-49 (BBoxTransform) then :57-67 (ClipBoxes)."""
    a = anchors
    w = a[:, :, 2] - a[:, :, 0]; h = a[:, :, 3] - a[:, :, 1]
    cx = a[:, :, 0] + 0.5 * w; cy = a[:, :, 1] + 0.5 * h
    std = torch.from_numpy(np.array([0.1, 0.1, 0.2, 0.2]).astype(np.float32))
    dx = regression[:, :, 0] * std[0]; dy = regression[:, :, 1] * std[1]
    dw = regression[:, :, 2] * std[2]; dh = regression[:, :, 3] * std[3]
    pcx = cx + dx * w; pcy = cy + dy * h
    pw = torch.exp(dw) * w; ph = torch.exp(dh) * h
    b = torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=2)
    b[:, :, 0] = torch.clamp(b[:, :, 0], min=0); b[:, :, 1] = torch.clamp(b[:, :, 1], min=0)
    b[:, :, 2] = torch.clamp(b[:, :, 2], max=W); b[:, :, 3] = torch.clamp(b[:, :, 3], max=H)
    return b


def nms_greedy(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics (third-party, unpinned -- 'parity unpinned'): stable sort by
    descending score, suppress when IoU > thr, area = (x2-x1)*(y2-y1) (no +1), returns int64
    indices into the candidate list in score order.  NumPy-vectorised per kept box."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    order = np.argsort(-s, kind='stable')
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(b), dtype=bool)
    keep = []
    for i in order:
        if dead[i]:
            continue
        keep.append(i)
        iw = np.maximum(np.minimum(x2[i], x2) - np.maximum(x1[i], x1), np.float32(0))
        ih = np.maximum(np.minimum(y2[i], y2) - np.maximum(y1[i], y1), np.float32(0))
        inter = iw * ih
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (area[i] + area - inter)
        dead |= iou > np.float32(iou_threshold)
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def detect_single(classification, boxes, threshold, iou_threshold):
    """models/efficientdet.py:72-86 for ONE image (the reference only handles image 0, Q8)."""
    scores = classification.max(dim=1)[0]
    mask = scores > threshold
    if int(mask.sum()) == 0:
        return torch.zeros(0), torch.zeros(0, dtype=torch.long), torch.zeros(0, 4)
    cand_cls, cand_box, cand_s = classification[mask], boxes[mask], scores[mask]
    keep = nms_greedy(cand_box, cand_s, iou_threshold)
    ns, nc = cand_cls[keep].max(dim=1)
    return ns, nc, cand_box[keep]


def calc_iou(a, b):
    """models/losses.py:6-26."""
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = torch.min(a[:, 2].unsqueeze(1), b[:, 2]) - torch.max(a[:, 0].unsqueeze(1), b[:, 0])
    ih = torch.min(a[:, 3].unsqueeze(1), b[:, 3]) - torch.max(a[:, 1].unsqueeze(1), b[:, 1])
    iw = iw.clamp(min=0); ih = ih.clamp(min=0)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).unsqueeze(1) + area - iw * ih
    return iw * ih / ua.clamp(min=1e-8)


def focal_loss(classifications, regressions, anchors, annotations):
    """models/losses.py:32-152, restated per image; returns ([1], [1]) like the reference (:152)."""
    alpha, gamma = 0.25, 2.0
    anchor = anchors[0]
    aw = anchor[:, 2] - anchor[:, 0]; ah = anchor[:, 3] - anchor[:, 1]
    acx = anchor[:, 0] + 0.5 * aw; acy = anchor[:, 1] + 0.5 * ah
    cls_losses, reg_losses = [], []
    for j in range(classifications.shape[0]):
        ann = annotations[j]; ann = ann[ann[:, 4] != -1]
        if ann.shape[0] == 0:                                           # losses.py:54-58
            cls_losses.append(torch.tensor(0.)); reg_losses.append(torch.tensor(0.)); continue
        p = classifications[j].clamp(1e-4, 1.0 - 1e-4)
        iou_max, iou_arg = calc_iou(anchor, ann[:, :4]).max(dim=1)
        tgt = torch.full_like(p, -1.0)
        tgt[iou_max < 0.4] = 0
        pos = iou_max >= 0.5
        npos = pos.sum()
        assigned = ann[iou_arg]
        tgt[pos] = 0
        tgt[pos, assigned[pos, 4].long()] = 1
        af = torch.where(tgt == 1., torch.full_like(p, alpha), torch.full_like(p, 1 - alpha))
        fw = af * torch.where(tgt == 1., 1. - p, p).pow(gamma)
        bce = -(tgt * torch.log(p) + (1.0 - tgt) * torch.log(1.0 - p))
        cl = torch.where(tgt != -1.0, fw * bce, torch.zeros_like(p))
        cls_losses.append(cl.sum() / npos.float().clamp(min=1.0))
        if int(npos) > 0:                                               # losses.py:108-148
            asg = assigned[pos]
            gw = asg[:, 2] - asg[:, 0]; gh = asg[:, 3] - asg[:, 1]
            gcx = asg[:, 0] + 0.5 * gw; gcy = asg[:, 1] + 0.5 * gh
            gw = gw.clamp(min=1); gh = gh.clamp(min=1)
            t = torch.stack(((gcx - acx[pos]) / aw[pos], (gcy - acy[pos]) / ah[pos],
                             torch.log(gw / aw[pos]), torch.log(gh / ah[pos]))).t()
            t = t / torch.tensor([[0.1, 0.1, 0.2, 0.2]])
            d = (t - regressions[j][pos]).abs()
            reg_losses.append(torch.where(d <= 1.0 / 9.0, 0.5 * 9.0 * d.pow(2), d - 0.5 / 9.0).mean())
        else:
            reg_losses.append(torch.tensor(0.))
    return torch.stack(cls_losses).mean(dim=0, keepdim=True), torch.stack(reg_losses).mean(dim=0, keepdim=True)


# --------------------------------------------------------------------------
# whole-model entry points
# --------------------------------------------------------------------------

def forward_raw(sd, network, num_classes, img, D_bifpn=None, drop_masks=None, keep=None, taps=False):
    """(classification, regression, anchors) of models/efficientdet.py:62-66."""
    D = EFFICIENTDET[network]['D_bifpn'] if D_bifpn is None else D_bifpn
    feats, t1 = backbone_forward(sd, network, img, drop_masks, keep)
    p, t2 = bifpn_forward(sd, feats[-5:], D)
    cls, reg = head_forward(sd, p, num_classes)
    anc = anchors_for_image(img.shape[2], img.shape[3])
    if taps:
        t1.update(t2)
        return cls, reg, anc, t1
    return cls, reg, anc


def detect(sd, network, num_classes, img, threshold=0.01, iou_threshold=0.5, D_bifpn=None):
    """Eval forward for every image of the batch = the reference run B times at batch 1 (Q8)."""
    cls, reg, anc = forward_raw(sd, network, num_classes, img, D_bifpn)
    boxes = decode_clip(anc, reg, img.shape[2], img.shape[3])
    return [detect_single(cls[b], boxes[b], threshold, iou_threshold) for b in range(img.shape[0])]


def train_losses(sd, network, num_classes, img, annots, D_bifpn=None, drop_masks=None, keep=None):
    cls, reg, anc = forward_raw(sd, network, num_classes, img, D_bifpn, drop_masks, keep)
    return focal_loss(cls, reg, anc, annots)


def golden_state_dict(g):
    """The state dict a tests/golden/*.npz fixture was produced with (network, num_classes, seed and, when present, bn2_gain)."""
    return make_state_dict(str(g['network']), int(g['num_classes']), seed=int(g['seed']),
                           bn2_gain=float(g['bn2_gain']) if 'bn2_gain' in getattr(g, 'files', g) else 1.0)


def synthetic_batch(B, S, seed=1, max_boxes=8, num_classes=80):
    """SURVEY §8(d) synthetic inputs: randn images, COCO-shape targets [B,8,5], pad rows = -1."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, S, S, generator=g)
    ann = torch.full((B, max_boxes, 5), -1.0)
    for b in range(B):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        x1 = torch.rand(n, generator=g) * 0.78 * S; y1 = torch.rand(n, generator=g) * 0.78 * S
        w = 16 + torch.rand(n, generator=g) * 0.4 * S; h = 16 + torch.rand(n, generator=g) * 0.4 * S
        ann[b, :n, 0], ann[b, :n, 1] = x1, y1
        ann[b, :n, 2] = (x1 + w).clamp(max=S - 1); ann[b, :n, 3] = (y1 + h).clamp(max=S - 1)
        ann[b, :n, 4] = torch.randint(0, num_classes, (n,), generator=g).float()
    return img, ann
