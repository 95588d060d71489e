"""CPU tier for the boundary steps (drop_connect generator, input pipeline, eval consumer, checkpoint format, import shims):
the oracles against published known answers / the reference's own arithmetic, and the host logic of the package."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import effdet_oracle as O
from oracle import pipeline_oracle as PO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Random123 v1.09 kat_vectors, "philox4x32 10" rows
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize('ctr,key,want', KAT)
def test_philox_oracle_and_host_twin_known_answers(ctr, key, want):
    from efficientdet.pytorch_amd import ops
    assert PO.philox4x32_10(ctr, key) == want
    assert ops.philox_host(ctr, key) == want          # the C-ABI host twin of the device generator (no GPU work)


def test_drop_connect_oracle_mask_statistics():
    keep = [1.0 - 0.2 * i / 16 for i in (2, 4, 6, 7, 9, 10, 12, 13, 14)]
    rows = PO.drop_connect_scales(keep, 512, seed=1234, step=7)
    for kp, r in zip(keep, rows):
        assert all(v == 0.0 or abs(v - 1.0 / kp) < 1e-6 for v in np.unique(r).tolist())
        assert abs(float((r > 0).mean()) - kp) < 0.06           # Bernoulli(keep), 512 draws
    assert not np.array_equal(rows, PO.drop_connect_scales(keep, 512, seed=1234, step=8))


def test_synthetic_batch_matches_the_parity_generator():
    from efficientdet.pytorch_amd.synthetic import synthetic_batch
    for args in [(3, 64, 5, 8, 20), (2, 128, 1, 8, 80)]:
        a, b = synthetic_batch(*args), O.synthetic_batch(*args)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_oracle_reproduces_reference_drop_connect_golden(golden_dir):
    """The real reference with drop_connect ACTIVE (rate 0.5, its own torch.rand draws recorded) == oracle with the same
    Bernoulli masks injected (models/utils.py:79-90, models/efficientnet.py:98-101)."""
    g = np.load(os.path.join(golden_dir, 'd0_128_dropconnect.npz'), allow_pickle=False)
    net, nc = str(g['network']), int(g['num_classes'])
    sd = O.golden_state_dict(g)
    img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
    masks = {int(i): torch.from_numpy(m) for i, m in zip(g['drop_blocks'], g['drop_masks'])}
    keep = {int(i): float(k) for i, k in zip(g['drop_blocks'], g['drop_keep'])}
    assert 0.3 < float(g['drop_masks'].mean()) < 1.0 and float(g['drop_masks'].min()) == 0.0     # rows really were dropped
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k}
    live = dict(sd); live.update(params)
    cl, rl = O.train_losses(live, net, nc, img, torch.from_numpy(g['annots']), drop_masks=masks, keep=keep)
    (cl.mean() + rl.mean()).backward()
    assert abs(float(cl) - float(g['cls_loss'])) <= 2e-5 * abs(float(g['cls_loss']))
    assert abs(float(rl) - float(g['reg_loss'])) <= 2e-5 * abs(float(g['reg_loss']))
    for k in ('backbone._blocks.2._project_conv.weight', 'backbone._blocks.9._depthwise_conv.weight', 'backbone._conv_stem.weight'):
        ref = g['grad_' + k + '_summary']
        l2 = float(params[k].grad.double().pow(2).sum().sqrt())
        assert abs(l2 - ref[2]) <= 5e-3 * ref[2], (k, l2, ref[2])
    # and the no-mask run differs (the masks matter)
    cl0, _ = O.train_losses(sd, net, nc, img, torch.from_numpy(g['annots']))
    assert abs(float(cl0) - float(g['cls_loss'])) > 1e-3 * abs(float(g['cls_loss']))


def test_preprocess_oracle_identity_geometry():
    """S x S input: the resize is the identity, output == (img/255 - mean)/std exactly; portrait / landscape pad geometry."""
    rng = np.random.RandomState(0)
    im = rng.randint(0, 256, (64, 64, 3), dtype=np.uint8)
    out, ann, sc = PO.preprocess_reference(im, np.array([[1., 2., 30., 40., 3.]]), common_size=64)
    assert sc == 1.0 and np.allclose(out, (im / 255.0 - PO.MEAN) / PO.STD, atol=1e-6)
    assert np.allclose(ann, [[1, 2, 30, 40, 3]])
    im = rng.randint(0, 256, (50, 100, 3), dtype=np.uint8)
    out, ann, sc = PO.preprocess_reference(im, np.array([[10., 5., 90., 45., 1.]]), common_size=64, flip=True)
    assert sc == 0.64 and out.shape == (64, 64, 3) and np.all(out[32:] == 0) and np.any(out[31] != 0)
    assert np.allclose(ann[0, :4], np.array([100 - 90, 5, 100 - 10, 45]) * 0.64)
    up = PO.resize_bilinear(np.arange(4, dtype=np.float32).reshape(2, 2, 1), 4, 4)[..., 0]
    assert np.allclose(up[0], [0, 0.25, 0.75, 1.0]) and np.allclose(up[:, 0], [0, 0.5, 1.5, 2.0])   # half-pixel centres + edge clamp


def test_resize_rule_agrees_with_an_independent_bilinear_implementation():
    """`cv2.resize` (absent) stays "parity unpinned"; what CAN be checked here: the restated INTER_LINEAR rule (half-pixel centres, edge
    clamp, no antialiasing) against torch's independent implementation of the same documented rule,
    F.interpolate(mode='bilinear', align_corners=False), up- and down-sampling, non-integer ratios."""
    import torch.nn.functional as F
    rng = np.random.RandomState(5)
    for (h, w, rh, rw) in [(50, 100, 48, 96), (97, 64, 96, 63), (33, 80, 39, 96), (120, 45, 96, 36), (17, 23, 64, 64)]:
        img = rng.rand(h, w, 3).astype(np.float32)
        got = PO.resize_bilinear(img, rw, rh)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(rh, rw), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 2e-5, (h, w, rh, rw, float(np.abs(got - ref).max()))     # (fp32 vs fp64 source coordinates: ~1e-5 of a unit pixel difference)


def test_finalize_oracle_formats():
    s = np.array([0.9, 0.8, 0.5, 0.04], dtype=np.float32); l = np.array([1, 0, 1, 2]); b = np.arange(16, dtype=np.float32).reshape(4, 4)
    d = PO.finalize_reference(s, l, b, scale=2.0, score_threshold=0.05, max_detections=2)
    assert d.shape == (2, 6) and np.allclose(d[0], [0, 0.5, 1, 1.5, 0.9, 1]) and np.allclose(d[1, 4:], [0.8, 0])
    r = PO.coco_results_reference(s, l, b, 2.0, image_id=7)
    assert len(r) == 3 and r[0]['bbox'] == [0.0, 0.5, 1.0, 1.0] and r[2]['category_id'] == 1 and r[0]['image_id'] == 7


def test_pipeline_oracle_is_pinned_on_the_reference_source_chain(golden_dir):
    """tests/golden/pipeline_chain.npz: Normalizer -> Augmenter -> Resizer -> collater executed FROM THE REFERENCE'S SOURCE TEXT
    (oracle/make_golden.py::case_pipeline_chain; only cv2.resize is a stub = the restated INTER_LINEAR rule, which therefore stays
    "parity unpinned").  The oracle's chain must reproduce it bit for bit: normalisation in float64, flip + box mirror, the int()
    truncated resize geometry, zero canvas, annotation scale, -1 padding to the longest list, NCHW."""
    g = np.load(os.path.join(golden_dir, 'pipeline_chain.npz'), allow_pickle=False)
    n, S = int(g['n']), int(g['S'])
    samples = [{'img': g[f'img{i}'], 'annot': g[f'annot{i}']} for i in range(n)]
    imgs, ann, scales = PO.collate_reference(samples, S, g['flips'])
    assert imgs.shape == g['out_imgs'].shape and np.array_equal(imgs, g['out_imgs'])
    assert ann.shape == g['out_annots'].shape and np.array_equal(ann, g['out_annots'])
    assert np.array_equal(scales, g['out_scales'])
    assert g['flips'].any() and not g['flips'].all() and (g['out_annots'][2] == -1).all()        # both branches, an empty annotation list


def test_eval_oracle_is_pinned_on_the_reference_source_loops(golden_dir):
    """tests/golden/eval_consumer.npz: eval.py:76-136 `_get_detections` and eval.py:260-338 `evaluate_coco` executed from the reference's
    source text on replayed detection lists (an image above the top-100 cut, one with nothing above the threshold, an empty one, a short
    one).  finalize_reference / coco_results_reference must reproduce every row."""
    g = np.load(os.path.join(golden_dir, 'eval_consumer.npz'), allow_pickle=False)
    NC, thr, mx = int(g['num_classes']), float(g['score_threshold']), int(g['max_detections'])
    rows = []
    for i, scale in enumerate(g['scales']):
        s, l, b = g[f'in{i}_scores'], g[f'in{i}_labels'], g[f'in{i}_boxes']
        d = PO.finalize_reference(s, l, b, scale, thr, mx)
        for c in range(NC):
            want = g[f'det{i}_class{c}']
            got = d[d[:, -1] == c, :-1] if len(d) else np.zeros((0, 5))
            assert got.shape == want.shape and np.array_equal(got.astype(np.float64), want), (i, c)
        rows += PO.coco_results_reference(s, l, b, scale, int(g['image_ids'][i]), thr,
                                          lambda c: int(g['coco_label_a']) + int(g['coco_label_b']) * c)
    assert len(rows) == len(g['coco_score']) > 100
    assert [r['image_id'] for r in rows] == g['coco_image_id'].tolist() and [r['category_id'] for r in rows] == g['coco_category_id'].tolist()
    assert np.array_equal(np.array([r['score'] for r in rows]), g['coco_score'])
    assert np.array_equal(np.array([r['bbox'] for r in rows], dtype=np.float64), g['coco_bbox'])


def test_checkpoint_roundtrip_prefix_and_pretrained(tmp_path):
    from efficientdet.pytorch_amd import EfficientDet, checkpoint as ck
    m = EfficientDet(4, network='efficientdet-d0', W_bifpn=64, D_bifpn=2)
    dp = torch.nn.DataParallel(m)
    assert list(ck.get_state_dict(dp).keys()) == list(m.state_dict().keys())        # utils/helper.py:25-30
    path = str(tmp_path / 'VOC' / 'efficientdet-d0' / 'checkpoint_3.pth')
    ck.save_checkpoint(path, dp, epoch=3, args={'num_class': 4, 'network': 'efficientdet-d0'})
    m2 = EfficientDet(4, network='efficientdet-d0', W_bifpn=64, D_bifpn=2)
    c = ck.load_checkpoint(path, m2)
    assert c['epoch'] == 3 and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # a DDP-style checkpoint ('module.' prefix, what the reference writes from its distributed path) loads too
    torch.save({'epoch': 0, 'parser': None, 'state_dict': {'module.' + k: v for k, v in m.state_dict().items()}}, path)
    ck.load_checkpoint(path, m2)
    # offline pretrained backbone: an 'efficientnet-b0' state_dict (keys without the 'backbone.' prefix) from a local file
    bsd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in m.backbone.state_dict().items()}
    torch.save(bsd, str(tmp_path / ck.PRETRAINED_FILES['efficientnet-b0']))
    assert ck.load_pretrained_backbone(m2, directory=str(tmp_path)) == 'efficientnet-b0'
    assert torch.equal(m2.backbone._blocks[3]._project_conv.weight, bsd['_blocks.3._project_conv.weight'])
    assert not torch.equal(m2.backbone._fc.weight, bsd['_fc.weight'])               # load_fc=False drops the classifier
    with pytest.raises(FileNotFoundError):
        ck.load_pretrained_backbone(EfficientDet(4, network='efficientdet-d1', W_bifpn=88, D_bifpn=3), directory=str(tmp_path))
    bad = dict(bsd); bad.pop('_blocks.0._bn1.weight')
    with pytest.raises(AssertionError):
        ck.load_pretrained_backbone(m2, source=bad)
    # reference_semantics=True: the reference's __init__ redraws every Conv2d weight and resets every BatchNorm
    # weight / bias AFTER from_pretrained (models/efficientdet.py:33, 47-53): only running statistics and conv biases survive
    m3 = EfficientDet(4, network='efficientdet-d0', W_bifpn=64, D_bifpn=2)
    ck.load_pretrained_backbone(m3, directory=str(tmp_path), reference_semantics=True)
    b3 = m3.backbone
    assert torch.equal(b3._blocks[3]._bn1.running_mean, bsd['_blocks.3._bn1.running_mean'])
    assert torch.equal(b3._blocks[3]._se_reduce.bias, bsd['_blocks.3._se_reduce.bias'])
    assert not torch.equal(b3._blocks[3]._project_conv.weight, bsd['_blocks.3._project_conv.weight'])
    assert bool((b3._blocks[3]._bn2.weight == 1).all()) and bool((b3._blocks[3]._bn2.bias == 0).all())


def test_dropin_import_paths():
    """train.py:30-33 / eval.py:15-16 / demo.py:5 import lines resolve to the MI355X module through the shim package."""
    code = ("from models.efficientdet import EfficientDet\nfrom models.losses import FocalLoss\nfrom models import EfficientDet as E2\n"
            "from utils import EFFICIENTDET, get_state_dict\nimport efficientdet.pytorch_amd as P\n"
            "assert EfficientDet is P.EfficientDet is E2 and EFFICIENTDET is P.EFFICIENTDET\nprint('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'efficientdet', 'pytorch_amd', 'dropin') + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=str(os.path.dirname(ROOT)))
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr


def test_model_copies_do_not_share_recorded_device_tables():
    import copy
    import pickle
    from efficientdet.pytorch_amd import EfficientDet, ops
    m = EfficientDet(4, network='efficientdet-d0', W_bifpn=64, D_bifpn=2)
    m._prep[('x', 'y')] = ops.ParamPrep()
    m._dc.update(seed=5, step=3)
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert c._prep == {} and c._dc == {'seed': 5, 'step': 3}
        assert torch.equal(c.bbox_head.retina_cls.weight, m.bbox_head.retina_cls.weight)
    assert len(m._prep) == 1
