"""Generate tests/golden/*.npz by executing the REAL reference (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY -- runs in the build container (the reference does not exist on the GPU
box).  Shims, none of which touches /root/reference:
  1. torchvision is not installed -> stub ``torchvision.ops.nms`` with the greedy restatement;
  2. ``load_pretrained_weights`` -> no-op (no network; weights come from make_state_dict);
  3. ``Tensor.cuda`` -> identity (models/losses.py hard-codes .cuda()) and an out-of-place rewrite
     of models/bifpn.py:178,180 (the in-place ``w /=`` breaks autograd on torch >= 1.5).

Usage:  python oracle/make_golden.py            (writes tests/golden/)
"""
import hashlib
import inspect
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import effdet_oracle as O  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    sys.path.insert(0, REF)
    tv, ops = types.ModuleType('torchvision'), types.ModuleType('torchvision.ops')
    ops.nms = lambda boxes, scores, iou_threshold: O.nms_greedy(boxes, scores, iou_threshold)
    tv.ops = ops
    sys.modules['torchvision'], sys.modules['torchvision.ops'] = tv, ops          # shim 1
    import models.utils as mu
    import models.efficientnet as me
    mu.load_pretrained_weights = me.load_pretrained_weights = lambda *a, **k: None  # shim 2
    from models.efficientdet import EfficientDet
    import models.bifpn as mb
    torch.Tensor.cuda = lambda self, *a, **k: self                                 # shim 3a
    src = inspect.getsource(mb.BiFPNModule.forward) \
        .replace("w1 /= torch.sum(w1, dim=0) + self.eps", "w1 = w1 / (torch.sum(w1, dim=0) + self.eps)") \
        .replace("w2 /= torch.sum(w2, dim=0) + self.eps", "w2 = w2 / (torch.sum(w2, dim=0) + self.eps)")
    ns = {}
    exec(textwrap.dedent(src), mb.__dict__, ns)
    mb.BiFPNModule.forward = ns['forward']                                         # shim 3b
    return EfficientDet


def build_ref(EfficientDet, network, num_classes, sd, **kw):
    c = O.EFFICIENTDET[network]
    m = EfficientDet(num_classes=num_classes, network=network, W_bifpn=c['W_bifpn'],
                     D_bifpn=c['D_bifpn'], D_class=c['D_class'], **kw)
    ref_keys = list(m.state_dict().keys())
    assert ref_keys == list(sd.keys()), 'state_dict key layout differs from the reference'
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict(sd)
    # parity runs: no drop_connect
    m.backbone._global_params = m.backbone._global_params._replace(drop_connect_rate=0.0)
    return m


def sample(t, n=256):
    """Deterministic strided sample of a tensor (fixture stays small)."""
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy().astype(np.float32)


def summary(t):
    f = t.detach().double().reshape(-1)
    return np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().sqrt().item()], dtype=np.float64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def taps_from_reference(m, img):
    """Stage-boundary activations of the real reference via forward hooks."""
    taps = {}
    hooks = [m.backbone._bn0.register_forward_hook(lambda mod, i, o: None)]
    for i, blk in enumerate(m.backbone._blocks):
        hooks.append(blk.register_forward_hook(lambda mod, i_, o, k=f'block{i}': taps.__setitem__(k, o.detach().clone())))
    for l, lc in enumerate(m.neck.lateral_convs):
        hooks.append(lc.register_forward_hook(lambda mod, i_, o, k=f'lateral{l}': taps.__setitem__(k, o.detach().clone())))
    for s, bm in enumerate(m.neck.stack_bifpn_convs):
        def h(mod, i_, o, s=s):
            for l, t in enumerate(o):
                taps[f'bifpn{s}_p{l}'] = t.detach().clone()
        hooks.append(bm.register_forward_hook(h))
    return taps, hooks


def case_eval(EfficientDet, name, network, num_classes, B, S, threshold, full=True, seed=0, dets=True, bn2_gain=1.0):
    sd = O.make_state_dict(network, num_classes, seed=seed, bn2_gain=bn2_gain)
    m = build_ref(EfficientDet, network, num_classes, sd, is_training=False, threshold=threshold)
    m.eval()
    img, _ = O.synthetic_batch(B, S, seed=1, num_classes=num_classes)
    taps, hooks = taps_from_reference(m, img)
    with torch.no_grad():
        x = m.extract_feat(img)
        outs = m.bbox_head(x)
        cls = torch.cat(list(outs[0]), 1); reg = torch.cat(list(outs[1]), 1)
        anc = m.anchors(img)
    for h_ in hooks:
        h_.remove()
    with torch.no_grad():                                # the reference only post-processes image 0 (Q8)
        dets = [m(img[b:b + 1]) for b in range(B)] if dets else []
    d = dict(network=network, num_classes=num_classes, B=B, S=S, seed=seed, threshold=threshold, bn2_gain=bn2_gain,
             anchors_sha256=sha(anc.numpy()), anchors_first=anc[0, :4].numpy(), anchors_last=anc[0, -4:].numpy(),
             cls_summary=summary(cls), reg_summary=summary(reg), cls_sample=sample(cls, 4096), reg_sample=sample(reg, 4096))
    if full:
        d['cls'] = cls.numpy(); d['reg'] = reg.numpy(); d['anchors'] = anc.numpy()
    for k, v in taps.items():
        d['tap_' + k + '_sample'] = sample(v, 512)
        d['tap_' + k + '_summary'] = summary(v)
    for b, (s_, c_, bx) in enumerate(dets):
        d[f'det{b}_scores'] = s_.numpy(); d[f'det{b}_labels'] = c_.numpy(); d[f'det{b}_boxes'] = bx.numpy()
    if not dets:                                         # (D4 @1024: 196k candidates x python greedy NMS -- forward only)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
        print(name, 'cls', tuple(cls.shape), 'forward only')
        return
    # NMS candidates of image 0 exactly as the reference builds them (models/efficientdet.py:69-83)
    with torch.no_grad():
        boxes = m.clipBoxes(m.regressBoxes(anc, reg[:1]), img[:1])
        scores = cls[:1].max(dim=2, keepdim=True)[0]
        mask = (scores > threshold)[0, :, 0]
        cb, cs = boxes[0, mask], scores[0, mask, 0]
        keep = O.nms_greedy(cb, cs, 0.5)
    d['nms_boxes'] = cb.numpy(); d['nms_scores'] = cs.numpy(); d['nms_keep'] = keep.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'cls', tuple(cls.shape), 'cands', int(mask.sum()), 'kept', [len(x[0]) for x in dets])


def case_dets_separated(EfficientDet, name, network, num_classes, B, S, gain=1.0, seed=0, want=32):
    """Eval detections with WELL-SEPARATED scores: at the default thresholds a random-init model has thousands of candidates in a
    narrow score band (gaps ~1e-7: the other eval goldens can only compare counts and leading scores).  Here the threshold
    is chosen (per image set, recorded in the fixture) so that only the ~`want` best candidates per image remain and every
    score gap exceeds 4e-6 and the margin to the threshold 5e-5 -- well above fp32 conv rounding (~1e-7 on a probability) -- so the COMPLETE
    lists (scores, labels, boxes, per image) of the real reference must be reproduced."""
    sd = O.make_state_dict(network, num_classes, seed=seed)
    sd['bbox_head.retina_cls.weight'] = sd['bbox_head.retina_cls.weight'] * gain
    img, _ = O.synthetic_batch(B, S, seed=1, num_classes=num_classes)
    m = build_ref(EfficientDet, network, num_classes, sd, is_training=False, threshold=0.5)
    m.eval()
    with torch.no_grad():
        outs = m.bbox_head(m.extract_feat(img))
        sc = torch.cat(list(outs[0]), 1).max(dim=2)[0]                      # [B, A] best-class probability
    srt = torch.sort(sc.reshape(-1), descending=True)[0]
    thr = None
    for k in range(want * B, 8, -1):                                         # a threshold in the widest gap below the k-th score
        lo, hi = float(srt[k]), float(srt[k - 1])
        if hi - lo > 2e-4:
            thr = round((lo + hi) / 2, 5)
            if min(abs(thr - lo), abs(hi - thr)) > 5e-5:
                break
    assert thr is not None
    m.threshold = thr
    with torch.no_grad():
        dets = [m(img[b:b + 1]) for b in range(B)]
    d = dict(network=network, num_classes=num_classes, B=B, S=S, seed=seed, threshold=thr, gain=gain)
    gaps = []
    for b, (s_, c_, bx) in enumerate(dets):
        cand = torch.sort(sc[b][sc[b] > thr])[0]
        if len(cand) > 1:
            gaps.append(float((cand[1:] - cand[:-1]).min()))
        d[f'det{b}_scores'] = s_.numpy(); d[f'det{b}_labels'] = c_.numpy(); d[f'det{b}_boxes'] = bx.numpy()
        d[f'det{b}_ncand'] = int(len(cand))
    assert min(gaps) > 4e-6, gaps
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'threshold', thr, 'candidates', [int(d[f'det{b}_ncand']) for b in range(B)], 'kept', [len(x[0]) for x in dets],
          'smallest candidate score gap', min(gaps))


def case_dets_dense(EfficientDet, name, network, num_classes, B, S, gain=0.5, seed=2, want=400):
    """Complete eval detections with ~100 KEPT boxes per image (round 4; the 'separated' cases keep 6-32): the threshold sits in a
    gap of > 1e-4 just below the ~`want`-th best candidate per image, so the candidate SET cannot hinge on conv rounding; among
    ~400 candidates in a 0.05-wide score band many neighbours are closer than fp32 conv rounding, so the consumer compares the lists
    as SETS of (label, box, score) -- the order of two near-tied boxes may differ, their membership may not."""
    sd = O.make_state_dict(network, num_classes, seed=seed)
    sd['bbox_head.retina_cls.weight'] = sd['bbox_head.retina_cls.weight'] * gain
    img, _ = O.synthetic_batch(B, S, seed=1, num_classes=num_classes)
    m = build_ref(EfficientDet, network, num_classes, sd, is_training=False, threshold=0.5)
    m.eval()
    with torch.no_grad():
        outs = m.bbox_head(m.extract_feat(img))
        sc = torch.cat(list(outs[0]), 1).max(dim=2)[0]
    srt = torch.sort(sc.reshape(-1), descending=True)[0]
    thr = None
    for k in range(want * B, want * B // 2, -1):
        lo, hi = float(srt[k]), float(srt[k - 1])
        if hi - lo > 1e-4:
            thr = (lo + hi) / 2
            break
    assert thr is not None
    m.threshold = thr
    with torch.no_grad():
        dets = [m(img[b:b + 1]) for b in range(B)]
    d = dict(network=network, num_classes=num_classes, B=B, S=S, seed=seed, threshold=np.float64(thr), gain=gain)
    for b, (s_, c_, bx) in enumerate(dets):
        d[f'det{b}_scores'] = s_.numpy(); d[f'det{b}_labels'] = c_.numpy(); d[f'det{b}_boxes'] = bx.numpy()
        d[f'det{b}_ncand'] = int((sc[b] > thr).sum())
    assert min(len(x[0]) for x in dets) >= 50
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'threshold', thr, 'candidates', [int(d[f'det{b}_ncand']) for b in range(B)], 'kept', [len(x[0]) for x in dets])


def case_dets_many(EfficientDet, name, network, num_classes, S, gain=1.0, seeds=range(39, 200), want=4500, min_kept=1000, max_ties=1):
    """A complete detection list of the bench's OWN size class (round 6): >= `min_kept` kept boxes out of ~`want` candidates for one
    image.  With thousands of candidates some pairs are always closer than conv rounding, and where such a pair OVERLAPS (IoU > 0.5) the
    kept SET -- not just the order -- hinges on it; neighbouring anchors share most of their receptive field, so such pairs are common
    (10-60 per instance over 60 seeds).  The case searches the weight seed for an instance with at most `max_ties` overlapping candidate
    pairs within 5e-6 (recorded as near_tie_pairs; seeds 2..38 were tried first: 9-158 pairs each) and a threshold in a gap > 2e-5: the kept set is
    then a function of the network up to those recorded pairs, and the consumer compares the lists as sets with that allowance."""
    img, _ = O.synthetic_batch(1, S, seed=1, num_classes=num_classes)
    for seed in seeds:
        sd = O.make_state_dict(network, num_classes, seed=seed)
        sd['bbox_head.retina_cls.weight'] = sd['bbox_head.retina_cls.weight'] * gain
        m = build_ref(EfficientDet, network, num_classes, sd, is_training=False, threshold=0.5)
        m.eval()
        with torch.no_grad():
            cls, reg, anc = m.bbox_head(m.extract_feat(img))[0], None, None
            sc = torch.cat(list(cls), 1).max(dim=2)[0][0]
        srt = torch.sort(sc, descending=True)[0]
        thr = None
        for k in range(want, want // 2, -1):
            lo, hi = float(srt[k]), float(srt[k - 1])
            if hi - lo > 2e-5:
                thr = (lo + hi) / 2
                break
        if thr is None:
            continue
        m.threshold = thr
        with torch.no_grad():
            s_, c_, bx = m(img)
            # the candidates as the reference forms them (models/efficientdet.py:64-81), to measure the overlapping near-ties
            feats = m.extract_feat(img)
            outs = m.bbox_head(feats)
            classification = torch.cat([o for o in outs[0]], dim=1); regression = torch.cat([o for o in outs[1]], dim=1)
            boxes = m.clipBoxes(m.regressBoxes(m.anchors(img), regression), img)
            keep = classification.max(dim=2, keepdim=True)[0][0, :, 0] > thr
            cb, cs = boxes[0][keep], classification[0][keep].max(dim=1)[0]
        order = torch.argsort(cs, descending=True); cb, cs = cb[order], cs[order]
        iou = O.calc_iou(cb, cb)
        near = (cs[:, None] - cs[None, :]).abs() < 5e-6
        near.fill_diagonal_(False)
        bad = int(((iou > 0.5) & near).sum()) // 2
        print(name, 'seed', seed, 'thr %.6f' % thr, 'candidates', int(keep.sum()), 'kept', len(s_), 'overlapping pairs within 5e-6:', bad, flush=True)
        if bad <= max_ties and len(s_) >= min_kept:
            d = dict(network=network, num_classes=num_classes, B=1, S=S, seed=seed, threshold=np.float64(thr), gain=gain, near_tie_pairs=bad,
                     det0_scores=s_.numpy(), det0_labels=c_.numpy(), det0_boxes=bx.numpy(), det0_ncand=int(keep.sum()))
            np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
            return
    raise RuntimeError('no seed gave a list free of overlapping near-ties')


def case_train(EfficientDet, name, network, num_classes, B, S, seed=0, empty_last=True, drop_connect=0.0, nsample=64, bn2_gain=1.0):
    sd = O.make_state_dict(network, num_classes, seed=seed, bn2_gain=bn2_gain)
    m = build_ref(EfficientDet, network, num_classes, sd, is_training=True)
    m.train(); m.is_training = True; m.freeze_bn()
    img, ann = O.synthetic_batch(B, S, seed=1, num_classes=num_classes)
    if empty_last:
        ann[-1, :, :] = -1.0        # last image has no annotations -> zero-loss branch (losses.py:54-58)
    draws = []
    if drop_connect:
        # drop_connect ACTIVE (models/utils.py:79-90): the reference draws torch.rand([B,1,1,1]) once per MBConv block with an
        # identity skip (models/efficientnet.py:98-101), block order; record every draw so that the SAME Bernoulli masks
        # can be injected into the oracle (drop_masks=) and into the HIP model (backbone.drop_masks)
        m.backbone._global_params = m.backbone._global_params._replace(drop_connect_rate=drop_connect)
        real_rand = torch.rand

        def rec_rand(*a, **k):
            r = real_rand(*a, **k)
            draws.append(r.reshape(-1).clone())
            return r
        torch.manual_seed(1234)
        torch.rand = rec_rand
    try:
        cl, rl = m([img, ann])
    finally:
        if drop_connect:
            torch.rand = real_rand
    (cl.mean() + rl.mean()).backward()
    d = dict(network=network, num_classes=num_classes, B=B, S=S, seed=seed,
             cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), annots=ann.numpy(), grad_nsample=nsample, bn2_gain=bn2_gain)
    if drop_connect:
        _, blocks, _, _ = O.backbone_blocks(network)
        skip_idx = [i for i, b in enumerate(blocks) if b['skip']]
        assert len(draws) == len(skip_idx), (len(draws), len(skip_idx))
        nb = len(blocks)
        keep = np.array([1.0 - drop_connect * float(i) / nb for i in skip_idx], dtype=np.float64)
        masks = np.stack([torch.floor(torch.tensor(k, dtype=torch.float32) + r).numpy() for k, r in zip(keep, draws)])
        d.update(drop_connect_rate=drop_connect, drop_blocks=np.array(skip_idx), drop_keep=keep, drop_masks=masks.astype(np.float32))
        print(name, 'drop masks: kept fraction', float(masks.mean()), 'blocks', len(skip_idx))
    dead = []
    for k, p in m.named_parameters():
        if p.grad is None:
            dead.append(k); continue
        d['grad_' + k + '_sample'] = sample(p.grad, nsample)
        d['grad_' + k + '_summary'] = summary(p.grad)
    d['dead_params'] = np.array(dead)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'losses', cl.item(), rl.item(), 'dead', dead)


def case_train_dropconnect(EfficientDet, name, network, num_classes, B, S):
    case_train(EfficientDet, name, network, num_classes, B, S, drop_connect=0.5)     # rate 0.5: plenty of dropped rows at B=4


def case_anchors(EfficientDet):
    from models.module import Anchors
    a = Anchors()
    d = {}
    for (H, W) in [(128, 128), (512, 512), (1024, 1024), (256, 384), (640, 512)]:
        anc = a(torch.zeros(1, 3, H, W)).numpy()
        d[f'sha_{H}x{W}'] = sha(anc); d[f'n_{H}x{W}'] = anc.shape[1]
        d[f'head_{H}x{W}'] = anc[0, :18]; d[f'tail_{H}x{W}'] = anc[0, -18:]
    np.savez_compressed(os.path.join(OUT, 'anchors.npz'), **d)
    print('anchors', {k: v for k, v in d.items() if k.startswith('n_')})


def reference_source_objects(relpath, names, ns):
    """The classes / functions `names` of one reference source file, compiled FROM THE REFERENCE'S OWN TEXT (ast: only those top-level
    definitions -- the file's imports, albumentations / cv2 / pycocotools among them, are not executed) into namespace `ns`."""
    import ast
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    nodes = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert sorted(n.name for n in nodes) == sorted(names), (relpath, [n.name for n in nodes])
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, 'exec'), ns)
    return ns


def case_pipeline_chain(_E=None):
    """Normalizer -> Augmenter -> Resizer -> collater (datasets/augmentation.py:69-150) run from the reference's source
    on seeded uint8 images the way datasets/voc0712.py:107-116 feeds them (RGB, float32 / 255).  `cv2.resize` -- OpenCV is absent -- is the
    ONE stub: the restated INTER_LINEAR rule of oracle/pipeline_oracle.py, so the interpolation itself stays "parity unpinned"; everything
    around it (float64 normalisation, flip + box mirror, scale rule with its int() truncation, zero canvas, annotation scale, -1
    padding, NCHW permute) is the reference's own arithmetic."""
    from oracle import pipeline_oracle as PO
    cv2 = types.ModuleType('cv2')
    cv2.resize = lambda image, size: PO.resize_bilinear(image, size[0], size[1])
    ns = reference_source_objects('datasets/augmentation.py', ['Normalizer', 'Augmenter', 'Resizer', 'collater'],
                                  {'np': np, 'torch': torch, 'cv2': cv2})
    S = 96
    rng = np.random.RandomState(20260930)
    shapes = [(50, 100), (97, 64), (96, 96), (33, 80), (120, 45), (64, 64)]
    nann = [3, 1, 0, 5, 2, 4]
    d = {'S': S, 'n': len(shapes)}
    samples, flips = [], []
    for i, ((h, w), na) in enumerate(zip(shapes, nann)):
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        x1 = rng.uniform(0, w * 0.6, na); y1 = rng.uniform(0, h * 0.6, na)
        ann = np.stack([x1, y1, x1 + rng.uniform(4, w * 0.4, na), y1 + rng.uniform(4, h * 0.4, na), rng.randint(0, 20, na).astype(np.float64)], 1) \
            if na else np.zeros((0, 5))
        d[f'img{i}'], d[f'annot{i}'] = img, ann.astype(np.float32)
        sample = {'img': img.astype(np.float32) / 255., 'annot': ann.astype(np.float32).astype(np.float64).copy()}      # voc0712.py:109
        sample = ns['Normalizer']()(sample)
        np.random.seed(1000 + i); flip = bool(np.random.rand() < 0.5)                 # what Augmenter's own draw will be
        np.random.seed(1000 + i); sample = ns['Augmenter']()(sample)
        flips.append(flip)
        sample = ns['Resizer']()(sample, common_size=S)
        samples.append(sample)
    imgs, annots = ns['collater'](samples)
    assert imgs.shape == (len(shapes), 3, S, S) and any(flips) and not all(flips)
    d['flips'] = np.array(flips)
    d['out_imgs'] = imgs.float().numpy()                      # train.py feeds the model .float() of this float64 batch
    d['out_annots'] = annots.numpy()
    d['out_scales'] = np.array([s['scale'] for s in samples], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'pipeline_chain.npz'), **d)
    print('pipeline_chain', imgs.shape, annots.shape, flips)


def case_eval_consumer(_E=None):
    """eval.py:76-136 `_get_detections` and eval.py:260-338 `evaluate_coco` run from the reference's source on a stub
    dataset / model that replays seeded (scores, labels, boxes) lists -- thresholding, the top-100 order, the per-class split, the
    boxes / scale division and the xywh COCO rows are the reference's own arithmetic.  pycocotools is absent: COCOeval / loadRes are no-op
    stubs (they run AFTER the rows were written to <set_name>_bbox_results.json, which is what the fixture keeps)."""
    import json
    import tempfile
    torch.Tensor.cuda = lambda self, *a, **k: self
    rng = np.random.RandomState(777)
    NC = 6
    counts, scales = [150, 40, 0, 12], [0.5, 1.25, 2.0, 0.37]
    dets = []
    for n in counts:
        sc = np.sort(rng.permutation(np.linspace(0.001, 0.999, 997))[:n].astype(np.float32))[::-1].copy()       # distinct, descending (NMS order)
        if n == 40:
            sc = (sc * 0.049).astype(np.float32)                                                                # an image with nothing above 0.05
        x1 = rng.uniform(0, 300, n); y1 = rng.uniform(0, 300, n)
        bx = np.stack([x1, y1, x1 + rng.uniform(2, 200, n), y1 + rng.uniform(2, 200, n)], 1).astype(np.float32)
        dets.append((sc, rng.randint(0, NC, n).astype(np.int64), bx))

    class DS:
        image_ids = [101, 102, 103, 104]
        set_name = 'golden_tmp'

        def __len__(self): return len(counts)
        def num_classes(self): return NC
        def label_to_coco_label(self, l): return 10 + 3 * l
        def __getitem__(self, i): return {'img': torch.zeros(4, 4, 3), 'scale': scales[i], 'index': i}

        class coco:
            @staticmethod
            def loadRes(path): return None

    class Model:
        def __init__(self): self.i = 0
        def eval(self): pass
        def train(self): pass

        def __call__(self, x):
            s, l, b = dets[self.i % len(dets)]; self.i += 1
            return torch.from_numpy(s.copy()), torch.from_numpy(l.copy()), torch.from_numpy(b.copy())

    class COCOeval:
        def __init__(self, *a): self.params = types.SimpleNamespace(imgIds=None)
        def evaluate(self): pass
        def accumulate(self): pass
        def summarize(self): pass
    ns = reference_source_objects('eval.py', ['_get_detections', 'evaluate_coco'], {'np': np, 'torch': torch, 'json': json, 'COCOeval': COCOeval})
    all_det = ns['_get_detections'](DS(), Model(), score_threshold=0.05, max_detections=100)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            ns['evaluate_coco'](DS(), Model(), threshold=0.05)
            rows = json.load(open('golden_tmp_bbox_results.json'))
        finally:
            os.chdir(cwd)
    d = {'num_classes': NC, 'scales': np.array(scales, dtype=np.float64), 'score_threshold': 0.05, 'max_detections': 100,
         'image_ids': np.array(DS.image_ids), 'coco_label_a': 10, 'coco_label_b': 3}
    for i, (s, l, b) in enumerate(dets):
        d[f'in{i}_scores'], d[f'in{i}_labels'], d[f'in{i}_boxes'] = s, l, b
        for c in range(NC):
            d[f'det{i}_class{c}'] = np.asarray(all_det[i][c], dtype=np.float64).reshape(-1, 5)
    d['coco_image_id'] = np.array([r['image_id'] for r in rows]); d['coco_category_id'] = np.array([r['category_id'] for r in rows])
    d['coco_score'] = np.array([r['score'] for r in rows], dtype=np.float64); d['coco_bbox'] = np.array([r['bbox'] for r in rows], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'eval_consumer.npz'), **d)
    print('eval_consumer: per-image kept', [sum(len(all_det[i][c]) for c in range(NC)) for i in range(len(counts))], 'coco rows', len(rows))


CASES = {
    'anchors': lambda E: case_anchors(E),
    'd0_128_eval': lambda E: case_eval(E, 'd0_128_eval', 'efficientdet-d0', 20, 2, 128, threshold=0.5),
    'd0_128_train': lambda E: case_train(E, 'd0_128_train', 'efficientdet-d0', 20, 3, 128),
    'd0_512_eval': lambda E: case_eval(E, 'd0_512_eval', 'efficientdet-d0', 80, 1, 512, threshold=0.6, full=False),
    'd4_256_eval': lambda E: case_eval(E, 'd4_256_eval', 'efficientdet-d4', 4, 1, 256, threshold=0.5, full=False),
    # forward (eval) fixtures for every other model family (d7 = d6's architecture): head outputs + stage-boundary taps, forward only
    'd1_128_eval': lambda E: case_eval(E, 'd1_128_eval', 'efficientdet-d1', 6, 1, 128, threshold=0.5, full=False, dets=False),
    'd2_128_eval': lambda E: case_eval(E, 'd2_128_eval', 'efficientdet-d2', 5, 1, 128, threshold=0.5, full=False, dets=False),
    'd3_128_eval': lambda E: case_eval(E, 'd3_128_eval', 'efficientdet-d3', 4, 1, 128, threshold=0.5, full=False, dets=False),
    'd5_128_eval': lambda E: case_eval(E, 'd5_128_eval', 'efficientdet-d5', 3, 1, 128, threshold=0.5, full=False, dets=False),
    'd6_128_eval': lambda E: case_eval(E, 'd6_128_eval', 'efficientdet-d6', 3, 1, 128, threshold=0.5, full=False, dets=False, bn2_gain=0.5),
    'd1_128_train': lambda E: case_train(E, 'd1_128_train', 'efficientdet-d1', 6, 2, 128),
    # the BASELINE.json geometries themselves: configs[2] = D0 train @512 with every parameter gradient,
    # configs[4] = D4 @1024 (80 classes, the COCO shape both are quoted on)
    # every other model family of the reference's table (train.py / models/efficientdet.py MODEL_MAP): d7 is d6's architecture at
    # another input size, so d6 covers both
    'd2_128_train': lambda E: case_train(E, 'd2_128_train', 'efficientdet-d2', 5, 2, 128, nsample=16),
    'd3_128_train': lambda E: case_train(E, 'd3_128_train', 'efficientdet-d3', 4, 2, 128, nsample=16),
    'd4_128_train': lambda E: case_train(E, 'd4_128_train', 'efficientdet-d4', 4, 2, 128, nsample=16),
    'd5_128_train': lambda E: case_train(E, 'd5_128_train', 'efficientdet-d5', 3, 2, 128, nsample=16),
    'd6_128_train': lambda E: case_train(E, 'd6_128_train', 'efficientdet-d6', 3, 2, 128, nsample=16, bn2_gain=0.5),
    'd0_512_train': lambda E: case_train(E, 'd0_512_train', 'efficientdet-d0', 80, 2, 512, empty_last=False),
    'd4_1024_eval': lambda E: case_eval(E, 'd4_1024_eval', 'efficientdet-d4', 80, 1, 1024, threshold=0.6, full=False, dets=False),
    'd0_128_dets_separated': lambda E: case_dets_separated(E, 'd0_128_dets_separated', 'efficientdet-d0', 20, 2, 128),
    # the same at a BASELINE geometry (configs[1]: D0 @512, 80 classes): complete detection lists, not 'count within 1 %'
    'd0_512_dets_separated': lambda E: case_dets_separated(E, 'd0_512_dets_separated', 'efficientdet-d0', 80, 2, 512, gain=0.5, seed=2),
    # ... and with ~100 kept boxes per image (complete lists compared as sets: near-tied neighbours may swap places)
    'd0_512_dets_dense': lambda E: case_dets_dense(E, 'd0_512_dets_dense', 'efficientdet-d0', 80, 2, 512),
    # configs[4]'s geometry end to end (round 5): D4 @1024, 80 classes -- the real reference's COMPLETE lists (196 416 anchors per image in,
    # scores / labels / boxes in NMS order out), gated in both parity modes like d0_512_dets_separated
    'd4_1024_dets_separated': lambda E: case_dets_separated(E, 'd4_1024_dets_separated', 'efficientdet-d4', 80, 2, 1024, gain=0.5, seed=3),
    'd4_1024_dets_dense': lambda E: case_dets_dense(E, 'd4_1024_dets_dense', 'efficientdet-d4', 80, 1, 1024),
    'd0_128_dropconnect': lambda E: case_train_dropconnect(E, 'd0_128_dropconnect', 'efficientdet-d0', 20, 4, 128),
    # the callers either side of the path, executed from the reference's own source text
    # a complete list of the bench's own size class: >= 1000 kept boxes of ~4500 candidates (D0 @512, 80 classes), free of overlapping near-ties
    'd0_512_dets_many': lambda E: case_dets_many(E, 'd0_512_dets_many', 'efficientdet-d0', 80, 512),
    'pipeline_chain': case_pipeline_chain,
    'eval_consumer': case_eval_consumer,
}


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    E = import_reference()
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name](E)
