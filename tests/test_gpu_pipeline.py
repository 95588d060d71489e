"""GPU parity of the boundary kernels (csrc/pipeline.hip) and of the rows they close: drop_connect (a7), device input
pipeline (f2), batched eval consumer (f3), differentiable (cls, reg, anchors) triple (a2), optimizer checkpointing (f4)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import effdet_oracle as O
from oracle import pipeline_oracle as PO
from tests.gpu_util import assert_close, assert_close_scale

pytestmark = pytest.mark.gpu


def _model(net, nc, dtype, seed=0, **kw):
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
    c = EFFICIENTDET[net]
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=dtype, **kw)
    m.load_state_dict(O.make_state_dict(net, nc, seed=seed))
    return m.cuda()


# ------------------------------------------------------------------------------------------------ drop_connect (a7)
def test_drop_connect_device_stream_is_bit_exact():
    """Integer parity: the HIP generator == the Philox4x32-10 oracle (pinned on Random123's known answers), word for word."""
    from efficientdet.pytorch_amd import ops
    keep = [1.0 - 0.2 * i / 16 for i in (2, 4, 6, 7, 9, 10, 12, 13, 14)]
    kd = torch.tensor(keep, dtype=torch.float32, device='cuda')
    for seed, step, B in [(1234, 0, 32), (2 ** 63 + 12345, 2 ** 33 + 7, 5), (0, 1, 1)]:
        got = ops.drop_connect_scales(kd, B, seed, step).cpu().numpy()
        assert np.array_equal(got, PO.drop_connect_scales(keep, B, seed, step)), (seed, step, B)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_drop_connect_injected_masks_vs_real_reference(golden_dir, dtype):
    """The REAL reference ran with drop_connect active (rate 0.5) and its Bernoulli draws were recorded; the HIP model with
    the same masks injected through the rowscale path (conv epilogue forward, act_bwd backward) must reproduce its losses
    and every parameter gradient (models/utils.py:79-90, models/efficientnet.py:98-101)."""
    g = np.load(os.path.join(golden_dir, 'd0_128_dropconnect.npz'), allow_pickle=False)
    net, nc = str(g['network']), int(g['num_classes'])
    m = _model(net, nc, dtype, seed=int(g['seed']))
    m.backbone.drop_connect_rate = float(g['drop_connect_rate'])
    m.backbone.drop_masks = {int(i): torch.from_numpy(r) for i, r in zip(g['drop_blocks'], g['drop_masks'])}
    m.train(); m.is_training = True; m.freeze_bn()
    img, _ = O.synthetic_batch(int(g['B']), int(g['S']), seed=1, num_classes=nc)
    cl, rl = m([img.cuda(), torch.from_numpy(g['annots']).cuda()])
    (cl.mean() + rl.mean()).backward()
    torch.cuda.synchronize()
    ltol, gtol = (1e-3, 1e-3) if dtype == torch.float32 else (3e-2, 0.15)
    assert_close(cl.detach().cpu(), torch.from_numpy(g['cls_loss']), ltol, 'cls loss')
    assert_close(rl.detach().cpu(), torch.from_numpy(g['reg_loss']), ltol, 'reg loss')
    dead = set(str(x) for x in g['dead_params'])
    for k, p in m.named_parameters():
        if k in dead:
            continue
        ref = g['grad_' + k + '_summary']
        l2 = float(p.grad.double().pow(2).sum().sqrt())
        assert abs(l2 - ref[2]) <= gtol * ref[2] + 1e-7, (k, l2, ref[2])
    # without the injected masks the generator path runs (different masks -> different loss), and eval ignores the rate
    m.backbone.drop_masks = None
    torch.manual_seed(3)
    cl2, _ = m([img.cuda(), torch.from_numpy(g['annots']).cuda()])
    assert abs(float(cl2) - float(g['cls_loss'])) > 1e-4 * abs(float(g['cls_loss']))
    cl3, _ = m([img.cuda(), torch.from_numpy(g['annots']).cuda()])
    assert float(cl3) != float(cl2)                       # the step counter advances the stream


def test_drop_connect_generator_statistics_and_seeding():
    m = _model('efficientdet-d0', 4, torch.bfloat16)
    m.train()
    torch.manual_seed(11)
    a = m._drop_connect_rowscales(256, torch.device('cuda'))
    b = m._drop_connect_rowscales(256, torch.device('cuda'))
    assert sorted(a) == [2, 4, 6, 7, 9, 10, 12, 13, 14]          # the identity-skip blocks of B0 (block 0.. of each stage skipped)
    for i, r in a.items():
        kp = 1.0 - 0.2 * i / 16
        assert abs(float((r > 0).float().mean()) - kp) < 0.08 and not torch.equal(r, b[i])
    m2 = _model('efficientdet-d0', 4, torch.bfloat16); m2.train()
    torch.manual_seed(11)
    a2 = m2._drop_connect_rowscales(256, torch.device('cuda'))
    assert all(torch.equal(a[i], a2[i]) for i in a)              # torch.manual_seed controls the stream


# ------------------------------------------------------------------------------------------------ input pipeline (f2)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_device_collater_vs_reference_chain(dtype):
    from efficientdet.pytorch_amd.data import DeviceCollater
    from efficientdet.pytorch_amd.synthetic import synthetic_raw_images
    from efficientdet.pytorch_amd import ops
    imgs, annots = synthetic_raw_images(5, seed=3, min_side=40, max_side=150)
    imgs.append(np.random.RandomState(1).randint(0, 256, (96, 96, 3), dtype=np.uint8)); annots.append(np.zeros((0, 5), np.float32))
    samples = [{'img': i, 'annot': a} for i, a in zip(imgs, annots)]
    col = DeviceCollater(common_size=96, dtype=dtype, flip_x=0.5, seed=4)
    flips = np.random.RandomState(4).rand(len(samples)) < 0.5
    assert flips.any() and not flips.all()
    packed, ann, scale = col(samples)
    ref_img, ref_ann, ref_scale = PO.collate_reference(samples, 96, flips)
    got = ops.nhwc_to_nchw(packed.map).cpu().numpy()
    assert got.shape == (6, packed.map.C, 96, 96) and np.all(got[:, 3:] == 0)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2          # bf16 storage of values up to ~2.6
    assert np.abs(got[:, :3] - ref_img).max() <= tol * max(1.0, np.abs(ref_img).max()), np.abs(got[:, :3] - ref_img).max()
    np.testing.assert_allclose(scale.cpu().numpy(), ref_scale, rtol=1e-6)
    np.testing.assert_allclose(ann.cpu().numpy(), ref_ann, rtol=1e-5, atol=1e-4)
    # the 96x96 image is not resized: exact normalisation, bit-for-bit in fp32
    if dtype == torch.float32 and not flips[5]:
        want = ((imgs[5].astype(np.float32) * np.float32(1 / 255.0) - np.array(PO.MEAN[0, 0], np.float32)) *
                (np.float32(1) / np.array(PO.STD[0, 0], np.float32))).transpose(2, 0, 1)
        np.testing.assert_allclose(got[5, :3], want, rtol=0, atol=1e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_device_collater_vs_the_reference_source_golden(golden_dir, dtype):
    """The same launch against tests/golden/pipeline_chain.npz -- the chain executed from the reference's own source text
    (datasets/augmentation.py:69-150; cv2.resize alone is the restated rule): images within fp32 rounding of the float64 host arithmetic,
    annotation rows and scales to fp32 rounding, the Augmenter's flip decisions injected."""
    import types
    from efficientdet.pytorch_amd.data import DeviceCollater
    from efficientdet.pytorch_amd import ops
    g = np.load(os.path.join(golden_dir, 'pipeline_chain.npz'), allow_pickle=False)
    n, S = int(g['n']), int(g['S'])
    samples = [{'img': g[f'img{i}'], 'annot': g[f'annot{i}']} for i in range(n)]
    col = DeviceCollater(common_size=S, dtype=dtype, flip_x=0.5, seed=0)
    col.rng = types.SimpleNamespace(rand=lambda B: np.where(g['flips'], 0.0, 1.0))          # the reference run's own draws
    packed, ann, scale = col(samples)
    got = ops.nhwc_to_nchw(packed.map).cpu().numpy()
    ref = g['out_imgs']
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert got.shape[0] == n and np.abs(got[:, :3] - ref).max() <= tol * max(1.0, np.abs(ref).max()), np.abs(got[:, :3] - ref).max()
    assert np.all(got[:, 3:] == 0)
    np.testing.assert_allclose(scale.cpu().numpy(), g['out_scales'], rtol=1e-6)
    np.testing.assert_allclose(ann.cpu().numpy(), g['out_annots'], rtol=1e-5, atol=1e-4)


def test_eval_consumer_vs_the_reference_source_golden(golden_dir):
    """evaluate.finalize / all_detections_rows / coco_results against tests/golden/eval_consumer.npz -- eval.py:76-136 and :260-338
    executed from the reference's own source text on replayed detection lists: every row bit for bit (same fp32 divisions)."""
    from efficientdet.pytorch_amd import evaluate as EV
    g = np.load(os.path.join(golden_dir, 'eval_consumer.npz'), allow_pickle=False)
    NC, thr, mx = int(g['num_classes']), float(g['score_threshold']), int(g['max_detections'])
    B = len(g['scales'])
    A = max(len(g[f'in{i}_scores']) for i in range(B))
    s = torch.zeros(B, A); l = torch.zeros(B, A, dtype=torch.int64); b = torch.zeros(B, A, 4); cnt = torch.zeros(B, dtype=torch.int32)
    for i in range(B):
        k = len(g[f'in{i}_scores']); cnt[i] = k
        s[i, :k] = torch.from_numpy(g[f'in{i}_scores']); l[i, :k] = torch.from_numpy(g[f'in{i}_labels']); b[i, :k] = torch.from_numpy(g[f'in{i}_boxes'])
    scales = [float(v) for v in g['scales']]                                    # python floats, as the reference's Resizer returns them
    dets, counts = EV.finalize(s.cuda(), l.cuda(), b.cuda(), cnt.cuda(), scales, score_threshold=thr, max_detections=mx)
    rows = EV.all_detections_rows(dets, counts, NC)
    for i in range(B):
        for c in range(NC):
            want = g[f'det{i}_class{c}']
            assert rows[i][c].shape == want.shape and np.array_equal(np.asarray(rows[i][c], dtype=np.float64), want), (i, c)
    dx, cx = EV.finalize(s.cuda(), l.cuda(), b.cuda(), cnt.cuda(), scales, score_threshold=thr, xywh=True)
    a_, b_ = int(g['coco_label_a']), int(g['coco_label_b'])
    res = EV.coco_results(dx, cx, image_ids=[int(v) for v in g['image_ids']], label_to_coco_label=lambda c: a_ + b_ * c)
    assert len(res) == len(g['coco_score']) > 100
    assert [r['image_id'] for r in res] == g['coco_image_id'].tolist() and [r['category_id'] for r in res] == g['coco_category_id'].tolist()
    assert np.array_equal(np.array([r['score'] for r in res]), g['coco_score'])
    assert np.array_equal(np.array([r['bbox'] for r in res], dtype=np.float64), g['coco_bbox'])


def test_packed_images_feed_the_model_like_nchw():
    """model(PackedImages) == model(NCHW tensor holding the same values): the stem reads the packed batch directly."""
    from efficientdet.pytorch_amd import ops, PackedImages
    m = _model('efficientdet-d0', 6, torch.float32, is_training=False)
    m.eval()
    img, _ = O.synthetic_batch(2, 128, seed=3, num_classes=6)
    img = img.cuda()
    packed = PackedImages(ops.nchw_to_nhwc(img, torch.float32, cpad=4))
    c1, r1, a1 = m.forward_raw(img)
    c2, r2, a2 = m.forward_raw(packed)
    assert torch.equal(a1, a2)
    # (the same kernels on the same values -- the path has no float atomics: bitwise equal)
    assert torch.equal(c2, c1) and torch.equal(r2, r1)
    mb = _model('efficientdet-d0', 6, torch.bfloat16, is_training=False)
    with pytest.raises(RuntimeError):
        mb.forward_raw(packed)                               # fp32 pack into a bf16 model: refused, not reinterpreted


# ------------------------------------------------------------------------------------------------ eval consumer (f3)
def test_batched_eval_consumer_vs_reference_loop():
    from efficientdet.pytorch_amd import evaluate as EV
    m = _model('efficientdet-d0', 8, torch.float32, is_training=False, threshold=0.4)
    m.eval()
    img, _ = O.synthetic_batch(3, 128, seed=2, num_classes=8)
    scales = np.array([0.5, 1.25, 2.0], dtype=np.float32)
    # ONE forward pass feeds both the device consumer and the reference loop (two passes differ by atomics-order noise)
    with torch.no_grad():
        cls, reg, anc = m.forward_raw(img.cuda())
    s_, l_, b_, cnt = EV.postprocess(m, cls, reg, anc, 128, 128)
    per_image = [(s_[i, :n], l_[i, :n], b_[i, :n]) for i, n in enumerate(cnt.tolist())]
    assert min(len(s) for s, _, _ in per_image) > 100          # random init: plenty above the threshold
    thr = float(np.median(per_image[0][0].cpu().numpy()[:100]))         # a threshold that cuts inside the top-100
    dets, counts = EV.finalize(s_, l_, b_, cnt, scales, score_threshold=thr, max_detections=100)
    for b, (s, l, bx) in enumerate(per_image):
        ref = PO.finalize_reference(s.cpu().numpy(), l.cpu().numpy(), bx.cpu().numpy(), scales[b], thr, 100)
        assert counts[b] == len(ref) and 0 < counts[b] <= 100
        np.testing.assert_array_equal(dets[b, :counts[b]], ref)             # same divisions, same rows: bit-exact
        assert np.all(dets[b, counts[b]:, 5] == -1)
    rows = EV.all_detections_rows(dets, counts, 8)
    assert len(rows) == 3 and len(rows[0]) == 8 and sum(len(r) for r in rows[0]) == counts[0] and rows[0][0].shape[1] == 5
    dx, cx = EV.finalize(s_, l_, b_, cnt, scales, score_threshold=thr, max_detections=100, xywh=True)
    d2, c2 = EV.detections_batched(m, img.cuda(), scales, score_threshold=thr, max_detections=100)      # the one-call form
    assert d2.shape == dets.shape and abs(int(c2.sum()) - int(counts.sum())) <= 3
    res = EV.coco_results(dx, cx, image_ids=[10, 11, 12], label_to_coco_label=lambda c: c + 1)
    s, l, bx = per_image[1]
    ref = PO.coco_results_reference(s.cpu().numpy()[:100], l.cpu().numpy()[:100], bx.cpu().numpy()[:100], scales[1], 11, thr, lambda c: c + 1)
    got = [r for r in res if r['image_id'] == 11]
    assert got == ref


# ------------------------------------------------------------------------------------------------ differentiable triple (a2)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_forward_raw_is_differentiable(dtype):
    """(classification, regression, anchors) of models/efficientdet.py:64-66 under autograd: a caller-side criterion on the
    triple gives the same parameter gradients as the oracle's autograd through the same criterion."""
    net, nc = 'efficientdet-d0', 6
    m = _model(net, nc, dtype)
    m.backbone.drop_connect_rate = 0.0
    m.train(); m.freeze_bn()
    img, _ = O.synthetic_batch(2, 128, seed=4, num_classes=nc)
    gen = torch.Generator().manual_seed(0)
    cls, reg, anc = m.forward_raw(img.cuda())
    wc = torch.randn(cls.shape, generator=gen); wr = torch.randn(reg.shape, generator=gen)
    assert cls.requires_grad and reg.requires_grad and not anc.requires_grad
    ((cls * wc.cuda()).sum() + (reg * wr.cuda()).sum()).backward()
    sd = O.make_state_dict(net, nc, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k}
    live = dict(sd); live.update(params)
    rc, rr, _ = O.forward_raw(live, net, nc, img)
    ((rc * wc).sum() + (rr * wr).sum()).backward()
    # fp32: 5e-3 of the tensor's scale -- the BiFPN max-pool routing and the ReLU masks are discontinuous in the forward values,
    # so the 1e-7 summation-order difference to the CPU oracle can move one gradient entry by a fixed amount; bf16: see below
    tol = 5e-3 if dtype == torch.float32 else 0.35
    assert_close_scale(cls.detach().cpu(), rc.detach(), 1e-3 if dtype == torch.float32 else 4e-2, 'cls')
    gmax = max(float(v.grad.norm()) for v in params.values() if v.grad is not None)
    for k, p in m.named_parameters():
        if params[k].grad is None:
            assert p.grad is None, k
            continue
        if dtype == torch.float32 or k.startswith('bbox_head'):
            assert_close_scale(p.grad.cpu(), params[k].grad, tol, k)
            if dtype == torch.float32:
                d = float((p.grad.cpu().double() - params[k].grad.double()).norm())
                assert d <= 5e-3 * float(params[k].grad.double().norm()) or d <= 5e-5 * gmax, (k, d)
        else:
            # bf16 through ~100 random-weight layers: single entries of the backbone / neck gradients are noisy (BN scales are
            # sums of cancelling terms), the direction is what a training step uses: cosine similarity per tensor
            a, b = p.grad.double().cpu().reshape(-1), params[k].grad.double().reshape(-1)
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            assert cos > (0.75 if a.numel() > 64 else 0.5), (k, cos)


# ------------------------------------------------------------------------------------------------ optimizer checkpoint (f4) + async race
def _toy_params(n=7, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(33,), (64, 3, 3, 3), (5, 17), (4096 + 13,), (1,), (256, 64), (8, 8, 8)][:n]
    return [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]


def test_clip_adamw_state_dict_roundtrip_matches_torch():
    """save -> load -> step: the resumed optimizer continues exactly like torch.optim.AdamW + clip_grad_norm_ that never stopped."""
    from efficientdet.pytorch_amd.optim import ClipAdamW
    ps = _toy_params(); rs = [p.detach().clone().requires_grad_(True) for p in ps]
    a = ClipAdamW(ps, lr=1e-2, max_norm=0.5); r = torch.optim.AdamW(rs, lr=1e-2)
    gen = torch.Generator().manual_seed(1)

    def step(opt, params, ref):
        grads = [torch.randn(p.shape, generator=gen) for p in params]
        for p, q, g in zip(params, ref, grads):
            p.grad = g.cuda().clone(); q.grad = g.cuda().clone()
        opt.step()
        torch.nn.utils.clip_grad_norm_(ref, 0.5); r.step()
    for _ in range(3):
        step(a, ps, rs)
    sd = copy.deepcopy(a.state_dict())
    assert [float(s['step']) for s in sd['state'].values()] == [3.0] * len(ps)          # the DEVICE counters, not a stale host copy
    ps2 = [p.detach().clone().requires_grad_(True) for p in ps]
    b = ClipAdamW(ps2, lr=1e-2, max_norm=0.5)
    b.load_state_dict(sd)
    for _ in range(2):
        step(b, ps2, rs)
    for p, q in zip(ps2, rs):
        assert_close(p.detach().cpu(), q.detach().cpu(), 2e-5, 'resumed parameters')
    # torch.optim.AdamW's own state_dict loads too (same keys)
    c = ClipAdamW([p.detach().clone().requires_grad_(True) for p in ps], lr=1e-2, max_norm=0.5)
    c.load_state_dict(r.state_dict())
    assert float(c.state_dict()['state'][0]['step']) == 5.0


def test_clip_adamw_no_sync_between_steps():
    """The host may run steps ahead of the stream (bench.py never synchronises): gradient tensors are fresh every step, so
    the pointer table changes every step; the pinned staging ring must not be rewritten under an in-flight upload."""
    from efficientdet.pytorch_amd.optim import ClipAdamW
    ps = _toy_params(); rs = [p.detach().clone().requires_grad_(True) for p in ps]
    a = ClipAdamW(ps, lr=1e-2, max_norm=0.0); r = torch.optim.AdamW(rs, lr=1e-2)
    gen = torch.Generator().manual_seed(2)
    gl = [[torch.randn(p.shape, generator=gen).cuda() for p in ps] for _ in range(12)]
    big = torch.randn(4096, 4096, device='cuda')
    torch.cuda.synchronize()
    keep = []
    for it in range(12):
        _ = big @ big                                        # keeps the stream busy so the host gets ahead
        for p, g in zip(ps, gl[it]):
            p.grad = g.clone(); keep.append(p.grad)          # distinct addresses every step
        a.step()
    for it in range(12):
        for q, g in zip(rs, gl[it]):
            q.grad = g.clone()
        r.step()
    torch.cuda.synchronize()
    for p, q in zip(ps, rs):
        assert_close(p.detach().cpu(), q.detach().cpu(), 2e-5, 'parameters after 12 unsynchronised steps')


def test_deepcopy_model_runs_independently():
    m = _model('efficientdet-d0', 4, torch.float32, is_training=False); m.eval()
    img, _ = O.synthetic_batch(1, 128, seed=1, num_classes=4)
    c1, _, _ = m.forward_raw(img.cuda()); c1b, _, _ = m.forward_raw(img.cuda())      # second call replays the recorded table
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(1.5)
    del m; torch.cuda.empty_cache()
    c2, _, _ = m2.forward_raw(img.cuda()); c2b, _, _ = m2.forward_raw(img.cuda())
    assert bool(torch.isfinite(c2).all()) and not torch.allclose(c2, c1)
    assert_close(c2b.cpu(), c2.cpu(), 1e-5, 'replayed copy')


# ------------------------------------------------------------------------------------------------ whole-step hipGraph
def test_graphed_train_step_tracks_eager():
    """One captured hipGraph per step == the eager step: same losses step by step (drop_connect ACTIVE: the device-side step
    counter draws the same masks in both), parameters after 6 steps within AdamW's sign noise."""
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    from efficientdet.pytorch_amd.optim import ClipAdamW
    net, nc, lr = 'efficientdet-d0', 8, 1e-4
    img, ann = O.synthetic_batch(4, 128, seed=6, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    runs = {}
    for mode in ('eager', 'graph'):
        torch.manual_seed(21)
        m = _model(net, nc, torch.float32)
        m.train(); m.is_training = True; m.freeze_bn()
        from efficientdet.pytorch_amd import ddp
        ddp.freeze_dead_parameters(m)
        opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=lr, max_norm=0.1)
        losses = []

        def step():
            opt.zero_grad(set_to_none=True)
            cl, rl = m([img, ann])
            (cl.mean() + rl.mean()).backward()
            opt.step()
            return cl, rl
        for _ in range(2):
            cl, rl = step(); losses.append(float(cl) + float(rl))
        del cl, rl             # (a live loss keeps the eager autograd graph -- and its default-stream AccumulateGrad nodes -- alive)
        if mode == 'graph':
            g = GraphedTrainStep(m, opt, img, ann, warmup=1)         # one more (untracked-loss) step on the capture side stream
            losses.append(losses[-1])
            for _ in range(3):
                cl, rl = g(); losses.append(float(cl) + float(rl))
        else:
            for _ in range(4):
                cl, rl = step(); losses.append(float(cl) + float(rl))
        torch.cuda.synchronize()
        runs[mode] = (losses, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
    le, lg = runs['eager'][0], runs['graph'][0]
    assert all(np.isfinite(le)) and all(np.isfinite(lg))
    for i, (a, b) in enumerate(zip(le, lg)):
        if i != 2:                                                  # (index 2 = the warm-up step inside the constructor)
            assert abs(a - b) <= 1e-5 * abs(a), (le, lg)           # same kernels, same masks, same order: equal up to float() printing
    assert len(set(lg[3:])) == 3                                    # replays do real, different steps
    d = (runs['eager'][1] - runs['graph'][1]).abs()
    assert float(d.mean()) < 0.5 * lr and float(d.max()) <= 6 * 2 * lr + 1e-6, (float(d.mean()), float(d.max()))


@pytest.mark.parametrize('arith', ['f32', 'f32_bwd_bf16x3', 'f32_hf16x3_bwd_bf16x3', 'bf16x3'])
@pytest.mark.parametrize('geom', [(4, 128, 20), (2, 256, 8)])
def test_a_graph_replay_is_the_eager_step(arith, geom):
    """graph.replay_vs_eager: ONE eager step and ONE replay from the same restored state (parameters, Adam moments + counters, drop_connect
    counter; same batch, same masks) must produce the same parameters and the same losses -- measured against the step's own update, not
    against AdamW's per-step travel.  (Round 5: the looser 'parameters within 2 lr per step' gates above stayed green while replays of the
    captured step trained on ~1e-7 of their gradients -- the focal-loss workspace was reset by a hipMemsetAsync node that did not hold
    inside the graph, so num_pos read a recycled float bit pattern.)"""
    from efficientdet.pytorch_amd.graph import GraphedTrainStep, replay_vs_eager
    from efficientdet.pytorch_amd.optim import ClipAdamW
    from efficientdet.pytorch_amd import ddp
    B, S, nc = geom
    img, ann = O.synthetic_batch(B, S, seed=6, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    torch.manual_seed(5)
    m = _model('efficientdet-d0', nc, torch.float32, f32_arith=arith)
    m.train(); m.is_training = True; m.freeze_bn()
    ddp.freeze_dead_parameters(m)
    opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); opt.step()
    del cl, rl
    g = GraphedTrainStep(m, opt, img, ann, warmup=2)
    g()
    r = replay_vs_eager(g)
    print('replay vs eager (%s, B=%d @%d): %s' % (arith, B, S, r))
    assert r['finite'] and r['update_norm'] > 0
    assert r['eager_vs_eager'] == 0.0 and r['replay_vs_replay'] == 0.0, r          # no float atomics: both are pure functions of the state
    assert r['replay_vs_eager'] <= 1e-6, r                                          # the same kernels on the same values
    for a, b in zip(r['losses_replay'], r['losses_eager']):
        assert abs(a - b) <= 1e-6 * abs(b), r


def test_eager_step_right_after_a_replay_reads_its_own_gradients():
    """Order eager, replay, eager (bench.py's fall-back after a failed self-check; a second replay_vs_eager): the replay's memcpy node
    re-installs the GRAPH's gradient-pointer table on the device, and after zero_grad(set_to_none=True) the eager gradients come back at
    the addresses of the last eager step -- ClipAdamW must upload its pointers again instead of trusting its cache.  Twin run: the same
    three steps with the middle one eager must give the same parameters bit for bit."""
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    from efficientdet.pytorch_amd.optim import ClipAdamW
    from efficientdet.pytorch_amd import ddp
    img, ann = O.synthetic_batch(2, 128, seed=6, num_classes=8)
    img, ann = img.cuda(), ann.cuda()
    res = {}
    for mode in ('eager', 'mixed'):
        m = _model('efficientdet-d0', 8, torch.float32, f32_arith='f32_bwd_bf16x3')
        m.train(); m.is_training = True; m.freeze_bn(); m.backbone.drop_connect_rate = 0.0
        ddp.freeze_dead_parameters(m)
        opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, max_norm=0.1)

        def eager():
            opt.zero_grad(set_to_none=True)
            cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); opt.step()
        eager(); eager()
        if mode == 'mixed':
            g = GraphedTrainStep(m, opt, img, ann, warmup=1)      # (1 warm-up step inside) ...
            g()                                                   # ... + one replay
        else:
            eager(); eager()
        eager(); eager()                                          # eager steps right behind the replay
        torch.cuda.synchronize()
        res[mode] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    assert bool(torch.isfinite(res['mixed']).all())
    assert torch.equal(res['eager'], res['mixed']), float((res['eager'] - res['mixed']).abs().max())


def test_graphed_detect_matches_eager():
    from efficientdet.pytorch_amd.graph import GraphedDetect
    m = _model('efficientdet-d0', 8, torch.float32, is_training=False, threshold=0.3)
    m.eval()
    img, _ = O.synthetic_batch(3, 128, seed=8, num_classes=8)
    img = img.cuda()
    eager = m.detect(img)
    gd = GraphedDetect(m, img)
    for rep in range(2):
        got = gd()
        assert len(got) == 3
        for (s, l, b), (es, el, eb) in zip(got, eager):
            assert torch.equal(s, es) and torch.equal(l, el) and torch.equal(b, eb)      # a replay runs the same kernels: bitwise equal
            assert l.dtype == torch.int64 and b.shape[1] == 4
    img2, _ = O.synthetic_batch(3, 128, seed=9, num_classes=8)
    gd.images.copy_(img2.cuda())
    got2 = gd()
    e2 = m.detect(img2.cuda())
    assert torch.equal(got2[0][0], e2[0][0]) and not torch.equal(got2[0][0][:5], eager[0][0][:5])


def test_inference_skips_param_prep_only_while_weights_are_unchanged():
    """Eval forwards reuse the packed parameter copies while no source tensor was written (Tensor._version); any in-place
    update (optimizer step, load_state_dict, .mul_) must show up in the very next forward."""
    m = _model('efficientdet-d0', 20, torch.float32, is_training=False)
    m.eval(); m.is_training = False
    img = O.synthetic_batch(2, 128, seed=3, num_classes=20)[0].cuda()
    with torch.no_grad():
        c0, r0, _ = m.forward_raw(img)             # records
        c1, r1, _ = m.forward_raw(img)             # first replay (launch)
        prep = next(iter(m._prep.values()))
        assert prep.replay and prep._fresh is not None
        fresh = prep._fresh
        c2, r2, _ = m.forward_raw(img)             # skipped launch
        assert prep._fresh == fresh
        assert torch.equal(c2, c1) and torch.equal(r2, r1)          # skip == replay, bit for bit (no float atomics anywhere)
        assert torch.equal(c1, c0) and torch.equal(r1, r0)          # replay == record
        m.bbox_head.retina_reg.weight.mul_(2.0); m.bbox_head.retina_reg.bias.mul_(2.0)
        c3, r3, _ = m.forward_raw(img)
        assert prep._fresh != fresh
        assert_close(c3.cpu(), c1.cpu(), 1e-3, 'cls untouched')
        assert_close(r3.cpu(), (2.0 * r1).cpu(), 1e-3, 'reg doubled')
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd['bbox_head.retina_reg.weight'] *= 0.5; sd['bbox_head.retina_reg.bias'] *= 0.5
        m.load_state_dict(sd)
        c4, r4, _ = m.forward_raw(img)
        assert_close(r4.cpu(), r1.cpu(), 1e-3, 'load_state_dict seen')
    # a training-mode forward in between always refreshes and resets the marker
    m.train(); m.is_training = True; m.freeze_bn()
    ann = O.synthetic_batch(2, 128, seed=3, num_classes=20)[1].cuda()
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward()
    assert prep._fresh is None


def test_interleaved_models_keep_their_own_f32_arithmetic():
    """f32_arith is process-wide state inside ops; every forward sets it, every backward node runs under the value its forward
    recorded and puts back what it found (ops.backward_scope) -- so two models with different arithmetic can interleave forward and backward passes (batched
    parameter prep on or off).  Checked on the kernels actually launched (the launch profile names them)."""
    from efficientdet.pytorch_amd import ops
    img, ann = O.synthetic_batch(2, 128, seed=13, num_classes=8)
    img, ann = img.cuda(), ann.cuda()

    def make(arith, batched):
        m = _model('efficientdet-d0', 8, torch.float32, f32_arith=arith)
        m.backbone.drop_connect_rate = 0.0
        m.batched_prep = batched
        m.train(); m.is_training = True; m.freeze_bn()
        return m

    def profiled(fn):
        ops.PROFILE = ops.LaunchProfile()
        try:
            fn(); torch.cuda.synchronize()
            return [r[0] for r in ops.PROFILE.records if r[0].startswith('conv_')]
        finally:
            ops.PROFILE = None
    try:
        for batched in (True, False):
            ma, mb = make('bf16x3', batched), make('f32', batched)
            la = ma([img, ann]); lb = mb([img, ann])                 # forward A, forward B: the global is now 'f32' ...
            assert ops.F32_ARITH == 'f32'
            ka = profiled(lambda: (la[0].mean() + la[1].mean()).backward())      # ... backward A must still run bf16x3 kernels
            assert ops.F32_ARITH == 'f32'                                        # ... and hand the arithmetic it found back (ops.backward_scope)
            kb = profiled(lambda: (lb[0].mean() + lb[1].mean()).backward())
            assert ops.F32_ARITH == 'f32'
            assert sum('bf16x3' in k for k in ka) >= 30, ka          # head / BiFPN data gradients + every DMA-eligible weight gradient
            assert not any('bf16x3' in k for k in kb), kb
            for p in list(ma.parameters()) + list(mb.parameters()):
                assert p.grad is None or bool(torch.isfinite(p.grad).all())
    finally:
        ops.set_f32_arith('f32')


# ------------------------------------------------------------------------------------------------ boundary claims as tests
def _train_model(nc=8, dtype=torch.float32, seed=0, **kw):
    m = _model('efficientdet-d0', nc, dtype, **kw)
    m.backbone.drop_connect_rate = 0.0
    m.train(); m.is_training = True; m.freeze_bn()
    return m


def test_eval_after_raw_pointer_updates_sees_the_new_weights():
    """The inference-side 'packed copies are still fresh' shortcut looks at Tensor._version, which raw-pointer writers never
    move: an eager ClipAdamW.step() and a hipGraph replay of the whole train step both have to invalidate it."""
    from efficientdet.pytorch_amd import ddp
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    from efficientdet.pytorch_amd.optim import ClipAdamW
    nc = 8
    m = _train_model(nc)
    ddp.freeze_dead_parameters(m)
    opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, max_norm=0.0)
    img, ann = O.synthetic_batch(2, 128, seed=3, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()

    def eval_out():
        m.eval(); m.is_training = False
        with torch.no_grad():
            c, r, _ = m.forward_raw(img)
        m.train(); m.is_training = True; m.freeze_bn()
        return c.clone(), r.clone()

    def fresh_twin_out():              # a second model loaded from the CURRENT state_dict: packs everything anew
        t = _model('efficientdet-d0', nc, torch.float32, is_training=False)
        t.load_state_dict(m.state_dict()); t = t.cuda().eval()
        with torch.no_grad():
            c, r, _ = t.forward_raw(img)
        return c, r
    e0 = eval_out(); e0b = eval_out()
    assert torch.equal(e0[1], e0b[1])
    # (1) eager ClipAdamW step between two eval forwards
    opt.zero_grad(set_to_none=True)
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); opt.step()
    del cl, rl
    e1 = eval_out(); e1b = eval_out()          # the second one takes the 'still fresh' shortcut
    assert not torch.equal(e1[1], e0[1]) and torch.equal(e1[1], e1b[1])
    t1 = fresh_twin_out()
    assert torch.equal(e1[0], t1[0]) and torch.equal(e1[1], t1[1])
    # (2) eval, graph replays, eval: the arena was last packed BEFORE the final replay's optimizer update
    g = GraphedTrainStep(m, opt, img, ann, warmup=1)
    e2 = eval_out()
    g(); g()
    e3 = eval_out()
    assert not torch.equal(e3[1], e2[1])
    t3 = fresh_twin_out()
    assert torch.equal(e3[0], t3[0]) and torch.equal(e3[1], t3[1])


def test_graphed_step_follows_the_lr_schedule():
    """ClipAdamW's hyper-parameters live in a device buffer that GraphedTrainStep refreshes before each replay: lr = 0 freezes
    the weights, restoring it moves them again (train.py:133,269 drives ReduceLROnPlateau).  A stock optimizer, whose
    kernels bake the values in at capture, must fail loudly instead of silently ignoring the change."""
    from efficientdet.pytorch_amd import ddp
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    from efficientdet.pytorch_amd.optim import ClipAdamW
    nc = 8
    m = _train_model(nc)
    ddp.freeze_dead_parameters(m)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = ClipAdamW(params, lr=1e-3, max_norm=0.1, weight_decay=0.0)
    img, ann = O.synthetic_batch(2, 128, seed=3, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    g = GraphedTrainStep(m, opt, img, ann, warmup=1)
    snap = lambda: torch.cat([p.detach().reshape(-1) for p in params]).clone()
    p0 = snap(); g(); p1 = snap()
    assert float((p1 - p0).abs().max()) > 1e-4
    opt.param_groups[0]['lr'] = 0.0
    g(); p2 = snap()
    assert torch.equal(p2, p1)                                   # lr = 0 (and no weight decay): nothing moves
    opt.param_groups[0]['lr'] = 1e-3
    g(); p3 = snap()
    assert float((p3 - p2).abs().max()) > 1e-4
    # eager steps after the capture use the same device buffer
    opt.param_groups[0]['lr'] = 0.0
    opt.zero_grad(set_to_none=True)
    cl, rl = m([img, ann]); (cl.mean() + rl.mean()).backward(); opt.step()
    assert torch.equal(snap(), p3)
    # an optimizer without a device-side hyper-parameter buffer (stock torch optimizers bake lr into the captured kernels):
    # a change after capture must raise, not be silently ignored
    class Stock:
        param_groups = [{'params': [], 'lr': 1e-3, 'betas': (0.9, 0.999)}]
    g.optimizer = Stock(); g._hyper0 = g._hyper_sig()
    g()
    Stock.param_groups[0]['lr'] = 5e-4
    with pytest.raises(RuntimeError, match='hyper-parameters changed'):
        g()


def test_param_prep_re_records_when_a_parameter_moves():
    """The recorded job table holds device addresses: p.data = <new tensor> (or any other move of a source) must be noticed
    (pointer validation on every eager step) and the table re-recorded -- never a replay against stale memory."""
    nc = 8
    m = _model('efficientdet-d0', nc, torch.float32, is_training=False); m.eval()
    img = O.synthetic_batch(2, 128, seed=3, num_classes=nc)[0].cuda()
    with torch.no_grad():
        m.forward_raw(img); c1, r1, _ = m.forward_raw(img)
        prep = next(iter(m._prep.values()))
        assert prep.replay
        w = m.bbox_head.retina_reg.weight
        old = w.data
        w.data = (2.0 * old).clone()                          # new storage; the old tensor stays alive (and unchanged)
        m.bbox_head.retina_reg.bias.data = (2.0 * m.bbox_head.retina_reg.bias.data).clone()
        c2, r2, _ = m.forward_raw(img)                        # detects the move, records again
        c3, r3, _ = m.forward_raw(img)                        # replays the new table
    assert torch.equal(c2, c1) and torch.equal(c3, c1)
    assert_close(r2.cpu(), (2.0 * r1).cpu(), 1e-5, 'reg doubled after the move'); assert torch.equal(r3, r2)
    n1 = len(prep.jobs)
    for _ in range(3):
        with torch.no_grad():
            m.forward_raw(img)
    assert len(prep.jobs) == n1                               # bounded: no growth from step to step


def test_data_parallel_single_device_train_step_equals_plain():
    """train.py:262-265 (the reference's default wrap): nn.DataParallel(model) on the one visible GPU.  Same losses and gradients
    as the bare module, bit for bit, step after step, with the parameter-preparation table bounded."""
    nc = 8
    img, ann = O.synthetic_batch(4, 128, seed=6, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    res = {}
    for wrapped in (False, True):
        m = _train_model(nc)
        net = torch.nn.DataParallel(m, device_ids=[0]) if wrapped else m
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
        out = []
        for it in range(3):
            opt.zero_grad()
            cl, rl = net([img, ann])
            (cl.mean() + rl.mean()).backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)
            opt.step()
            out.append((float(cl), float(rl)))
        res[wrapped] = (out, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone(), len(next(iter(m._prep.values())).jobs))
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert res[True][2] == res[False][2]


def test_gradient_accumulation_over_two_backward_passes():
    """train.py:114-118 with grad_accumulation_steps = 2: two backward passes into the same .grad == the sum of the two
    single-pass gradients (through the custom autograd nodes), and ClipAdamW.step() on the accumulated gradients == on their sum."""
    from efficientdet.pytorch_amd import ddp
    from efficientdet.pytorch_amd.optim import ClipAdamW
    nc = 8
    m = _train_model(nc)
    ddp.freeze_dead_parameters(m)
    params = [p for p in m.parameters() if p.requires_grad]
    ia, aa = O.synthetic_batch(2, 128, seed=6, num_classes=nc)
    ib, ab = O.synthetic_batch(2, 128, seed=7, num_classes=nc)
    batches = [(ia.cuda(), aa.cuda()), (ib.cuda(), ab.cuda())]
    single = []
    for img, ann in batches:
        for p in params:
            p.grad = None
        cl, rl = m([img, ann]); ((cl.mean() + rl.mean()) / 2).backward()
        single.append([p.grad.clone() for p in params])
    for p in params:
        p.grad = None
    for img, ann in batches:                                   # accumulate: no zero_grad in between
        cl, rl = m([img, ann]); ((cl.mean() + rl.mean()) / 2).backward()
    for p, g0, g1 in zip(params, single[0], single[1]):
        assert torch.equal(p.grad, g0 + g1)
    twin = copy.deepcopy(m)
    tparams = [p for p in twin.parameters() if p.requires_grad]
    for q, g0, g1 in zip(tparams, single[0], single[1]):
        q.grad = g0 + g1
    o1 = ClipAdamW(params, lr=1e-3, max_norm=0.1); o2 = ClipAdamW(tparams, lr=1e-3, max_norm=0.1)
    o1.step(); o2.step()
    for p, q in zip(params, tparams):
        assert torch.equal(p.detach(), q.detach())
