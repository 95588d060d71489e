"""GPU parity at the BASELINE geometries themselves (configs[1] D0 B=32 @512, configs[2] the same batch in training,
configs[4] D4 B=8 @1024) through size-independent properties -- the goldens pin the arithmetic at sizes the CPU oracle
finishes in seconds; these tests pin that the FULL-SIZE launches (thousands of tiles, XCD remap, grouped 5-level segments,
the persistent head kernel, split-K weight gradients over 174 592 pixels, 24 / 96 NMS rounds) compute the same function:

  * batch independence: image i of the big batch == the same image run alone (the small-size path the goldens cover);
  * loss / gradient linearity: the batch loss is the mean of the per-image losses and every parameter gradient the mean of
    the per-image gradients (models/losses.py:32-152 normalises per image, then .mean over the batch);
  * NMS against its DEFINITION, evaluated independently with torch ops on the device: the kept list is in candidate order,
    pairwise IoU <= thr inside it, and every dropped candidate overlaps an EARLIER kept box by > thr -- the three facts
    that characterise the greedy result uniquely (induction on the rank), so no restatement of the algorithm is involved.
"""
import pytest
import torch

from oracle import effdet_oracle as O

pytestmark = pytest.mark.gpu


def _model(net, nc, dtype, training, seed=0, f32_arith='f32', bn2_gain=1.0):
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
    c = EFFICIENTDET[net]
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], is_training=training,
                     compute_dtype=dtype, threshold=0.01, iou_threshold=0.5, f32_arith=f32_arith)
    m.load_state_dict(O.make_state_dict(net, nc, seed=seed, bn2_gain=bn2_gain))
    m.backbone.drop_connect_rate = 0.0
    m = m.cuda()
    if training:
        m.train(); m.is_training = True; m.freeze_bn()
    else:
        m.eval(); m.is_training = False
    return m


def _scale_err(a, b):
    return float((a.float() - b.float()).abs().max()) / (float(b.float().abs().max()) + 1e-30)


@pytest.fixture(autouse=True)
def _restore_arith():
    from efficientdet.pytorch_amd import ops
    yield
    ops.set_f32_arith('f32')


@pytest.mark.parametrize('net,B,S,dtype,tol', [
    ('efficientdet-d0', 32, 512, torch.float32, 1e-4),        # configs[1] / [2] geometry
    ('efficientdet-d0', 32, 512, 'f32_bf16x3', 3e-4),         # fp32 storage, bf16x3 products (B = 1 and B = 32 take other tilings)
    ('efficientdet-d0', 32, 512, torch.bfloat16, 2e-2),
    ('efficientdet-d4', 8, 1024, torch.bfloat16, 0.10),       # configs[4] geometry (the D4 bf16 gate of the golden tests:
                                                              #  B = 1 takes the 16x16x32 head kernel, B = 8 the persistent 32x32x16
                                                              #  one -- another summation order, amplified like storage rounding)
    ('efficientdet-d4', 8, 1024, torch.float32, 1e-4),        # configs[4] in the parity modes: exact fp32 ...
    ('efficientdet-d4', 8, 1024, 'f32_bf16x3', 3e-4),         # ... and fp32 storage + bf16x3 products
    ('efficientdet-d0', 32, 512, 'f32_hf16x3_bwd_bf16x3', 1e-4),      # the round-6 headline: f16x3 head + BiFPN convs (B = 1: the small BiFPN
    ('efficientdet-d4', 8, 1024, 'f32_hf16x3_bwd_bf16x3', 1e-4),      #  launches stay on the exact kernel -- same values to fp32 rounding)
])
def test_batch_independence_at_benchmark_size(net, B, S, dtype, tol):
    """fp32: identical K-reduction order per output element; what differs between B = 1 and B = 32 is the number of tile groups
    the squeeze-excite pool is summed over (1e-6); bf16: that flips bf16 roundings which the network amplifies like any storage
    rounding (gate: the golden tests' tolerance)."""
    if dtype == 'f32_bf16x3':
        m = _model(net, 80, torch.float32, False, f32_arith='bf16x3')
    elif isinstance(dtype, str):
        m = _model(net, 80, torch.float32, False, f32_arith=dtype)
    else:
        m = _model(net, 80, dtype, False)
    img = O.synthetic_batch(B, S, seed=5, num_classes=80)[0].cuda()
    with torch.no_grad():
        cls, reg, anc = m.forward_raw(img)
        worst = [0.0, 0.0]
        for i in sorted({0, B // 2 + 1, B - 1}):
            c1, r1, a1 = m.forward_raw(img[i:i + 1])
            assert torch.equal(a1, anc)
            worst[0] = max(worst[0], _scale_err(cls[i:i + 1], c1)); worst[1] = max(worst[1], _scale_err(reg[i:i + 1], r1))
    print('%s B=%d @%d %s: image-in-batch vs image-alone, max err / scale: cls %.2e reg %.2e' % (net, B, S, dtype, worst[0], worst[1]))
    assert worst[0] <= tol and worst[1] <= tol, worst


def test_loss_and_gradients_are_batch_means_at_benchmark_size():
    """configs[2] (D0, B=32 @512, 80 classes), fp32: loss(batch) == mean_i loss(image i) and grad(batch) == mean_i grad(image i)
    for all 274 live tensors -- the full-size weight-gradient launches (split-K over 174 592 head pixels, 2 M backbone
    pixels) against 32 small ones."""
    B, S = 32, 512
    m = _model('efficientdet-d0', 80, torch.float32, True)
    img, ann = O.synthetic_batch(B, S, seed=1, num_classes=80)
    img, ann = img.cuda(), ann.cuda()
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    full = {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}
    full_loss = (float(cl.detach()), float(rl.detach()))
    acc = {k: torch.zeros_like(v) for k, v in full.items()}
    lsum = [0.0, 0.0]
    for i in range(B):
        for p in m.parameters():
            p.grad = None
        c1, r1 = m([img[i:i + 1], ann[i:i + 1]])
        (c1.mean() + r1.mean()).backward()
        lsum[0] += float(c1.detach()); lsum[1] += float(r1.detach())
        for k, p in m.named_parameters():
            if p.grad is not None:
                acc[k] += p.grad.double()
    assert abs(full_loss[0] - lsum[0] / B) <= 1e-4 * abs(full_loss[0]), (full_loss, lsum)
    assert abs(full_loss[1] - lsum[1] / B) <= 1e-4 * abs(full_loss[1]), (full_loss, lsum)
    assert len(full) == 274
    worst, bad = (0.0, None), []
    for k, v in full.items():
        d = float((v - acc[k] / B).norm()); n = float(v.norm())
        rel = d / max(n, 1e-30)
        if rel > worst[0]:
            worst = (rel, k)
        # ONE relative clause for every tensor, however small its norm (round 5: the absolute escape `d <= 1e-6 * largest norm` is gone,
        # as in the golden gradient test): ReLU / max-pool ties may fall differently in the two runs, hence 2e-3 and not 1e-5
        if rel > 2e-3:
            bad.append((k, rel, d, n))
    assert not bad, bad
    print('batch loss %s vs mean of per-image losses %s; worst gradient deviation %s' % (full_loss, [x / B for x in lsum], worst))


@pytest.mark.parametrize('arith', ['f32', 'bf16x3', 'f32_bwd_bf16x3', 'f32_hf16x3_bwd_bf16x3'])
def test_largest_family_at_its_native_size(arith):
    """EfficientDet-D6 (B6 backbone: 45 MBConv blocks, BiFPN 384 x 8, 5-conv heads; D7 is the same network at another size) at
    its own 1408 x 1408, B = 2, train mode: the batch loss is the mean of the two per-image losses, every live gradient is the
    mean of the per-image gradients, everything finite -- the launches no golden reaches (176 x 176 stride-8 maps, 384-channel
    BiFPN / head convs) against the same image run alone."""
    net, S, B = 'efficientdet-d6', 1408, 2
    m = _model(net, 3, torch.float32, True, f32_arith=arith, bn2_gain=0.5)
    img, ann = O.synthetic_batch(B, S, seed=3, num_classes=3)
    img, ann = img.cuda(), ann.cuda()
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    full = {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}
    assert all(bool(torch.isfinite(v).all()) for v in full.values()) and len(full) > 700
    fl = (float(cl.detach()), float(rl.detach()))
    acc = {k: torch.zeros_like(v) for k, v in full.items()}
    ls = [0.0, 0.0]
    for i in range(B):
        for p in m.parameters():
            p.grad = None
        c1, r1 = m([img[i:i + 1], ann[i:i + 1]])
        (c1.mean() + r1.mean()).backward()
        ls[0] += float(c1.detach()) / B; ls[1] += float(r1.detach()) / B
        for k, p in m.named_parameters():
            if p.grad is not None:
                acc[k] += p.grad.double() / B
    assert abs(fl[0] - ls[0]) <= 1e-4 * abs(fl[0]) and abs(fl[1] - ls[1]) <= 1e-4 * abs(fl[1]), (fl, ls)
    worst, bad = (0.0, None), []
    # exact fp32: the golden gradient gate; bf16x3 products: B = 2 and B = 1 take other tilings (another rounding of the 2^-17 split
    # operands), so ReLU masks within 1e-5 of zero fall differently in the two runs (tests/test_gpu_model.py, profiles/r04_x3_locate.txt)
    # and every tensor upstream collects them -- measured 9.9e-3 on the stem weight (the end of the backward chain); the stated gate
    # of the deep families, doubled
    # ('f32_bwd_bf16x3': exact forward, no mask can flip -- the exact mode's gates.)  ONE relative clause per tensor, no absolute escape
    # (round 5).  What the escape had hidden, measured: exact fp32 is within 2e-3 on every tensor except the LAST of the 45-block backward
    # chain, the stem weight at 6.1e-3 (d = 0.048 on a norm of 7.89; the same value in both exact-forward modes) -- a sum over 1408 x 1408
    # pixels with heavy cancellation into which the few ReLU / max-pool ties that fall differently between the B = 2 and B = 1 launches
    # (other tilings, other summation orders) are carried; tests/test_gpu_model.py::test_non_square_input_vs_oracle measures the same
    # effect on the reference arithmetic itself (its own stem gradient moves by 1.25e-2 under a 1.2e-6 input scaling).  Stated gate for
    # the stem tensors: 1e-2.
    gtol = 2e-2 if arith == 'bf16x3' else 2e-3
    stem_tol = 2e-2 if arith == 'bf16x3' else 8e-3          # (measured 6.1e-3 in every exact-trunk mode; round-5 advisor: keep the gate near the measurement)
    for k, v in full.items():
        d = float((v - acc[k]).norm()); rel = d / max(float(v.norm()), 1e-30)
        if rel > worst[0]:
            worst = (rel, k)
        if rel > (stem_tol if k.startswith(('backbone._conv_stem', 'backbone._bn0')) else gtol):
            bad.append((k, rel, d, float(v.norm())))
    assert not bad, bad
    stem = max((float((v - acc[k]).norm()) / max(float(v.norm()), 1e-30) for k, v in full.items() if k.startswith(('backbone._conv_stem', 'backbone._bn0'))), default=0.0)
    print('D6 @1408 B=2 (%s): losses %s vs per-image mean %s; worst gradient deviation %s; stem tensors %.2e' % (arith, fl, ls, worst, stem))


def _iou_gt(a, b, thr):
    """[Na,4] x [Nb,4] -> bool [Na,Nb]: the reference arithmetic (separately rounded fp32 ops, area without +1)."""
    iw = torch.minimum(a[:, None, 2], b[None, :, 2]) - torch.maximum(a[:, None, 0], b[None, :, 0])
    ih = torch.minimum(a[:, None, 3], b[None, :, 3]) - torch.maximum(a[:, None, 1], b[None, :, 1])
    pos = (iw > 0) & (ih > 0)
    inter = iw * ih
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = (aa[:, None] + ab[None, :]) - inter
    return pos & (inter / union > thr)


def _check_greedy_definition(boxes, score, keep, thr, score_thr, chunk=4096):
    """boxes [A,4], score [A], keep = kept anchor indices in output order (all on the device)."""
    cand = torch.nonzero(score > score_thr).flatten()
    order = cand[torch.sort(score[cand], descending=True, stable=True)[1]]           # ties: lower anchor index first
    rank = torch.full((score.numel(),), -1, dtype=torch.long, device=score.device)
    rank[order] = torch.arange(order.numel(), device=score.device)
    kr = rank[keep.long()]
    assert bool((kr >= 0).all()), 'a kept index is not a candidate'
    assert bool((kr[1:] > kr[:-1]).all()), 'kept list is not in candidate (score) order'
    kb = boxes[keep.long()]
    nk = keep.numel()
    # (ii) no kept pair overlaps by more than thr
    for s in range(0, nk, chunk):
        m = _iou_gt(kb[s:s + chunk], kb, thr)
        m[torch.arange(min(chunk, nk - s), device=m.device), torch.arange(s, min(s + chunk, nk), device=m.device)] = False
        assert not bool(m.any()), 'two kept boxes overlap by more than the threshold'
    # (iii) every dropped candidate is overlapped (> thr) by a kept box that precedes it
    is_kept = torch.zeros(score.numel(), dtype=torch.bool, device=score.device); is_kept[keep.long()] = True
    dropped = order[~is_kept[order]]
    for s in range(0, dropped.numel(), chunk):
        d = dropped[s:s + chunk]
        m = _iou_gt(boxes[d], kb, thr) & (kr[None, :] < rank[d][:, None])
        assert bool(m.any(dim=1).all()), 'a dropped candidate has no earlier kept box overlapping it'
    return order.numel(), nk


@pytest.mark.parametrize('net,B,S,dtype', [('efficientdet-d0', 32, 512, torch.float32), ('efficientdet-d4', 8, 1024, torch.bfloat16),
                                           ('efficientdet-d4', 8, 1024, torch.float32)])
def test_nms_satisfies_the_greedy_definition_at_benchmark_size(net, B, S, dtype):
    """configs[1] / configs[4] on random-init weights: every anchor is a candidate (49 104 / 196 416 per image)."""
    from efficientdet.pytorch_amd import ops
    m = _model(net, 80, dtype, False)
    img = O.synthetic_batch(B, S, seed=2, num_classes=80)[0].cuda()
    with torch.no_grad():
        cls, reg, anc = m.forward_raw(img)
        boxes, score, label = ops.decode_score(anc, reg, cls, S, S)
        idx, count = ops.nms(boxes, score, 0.01, 0.5)
        torch.cuda.synchronize()
        counts = count.tolist()
        for b in sorted({0, B - 1}):
            nc_, nk = _check_greedy_definition(boxes[b], score[b], idx[b, :counts[b]], 0.5, 0.01)
            print('%s image %d: %d candidates -> %d kept, greedy definition holds' % (net, b, nc_, nk))
        # idempotence: the kept boxes alone survive a second pass unchanged (all images, one launch)
        kmax = max(counts)
        kb = torch.zeros((B, kmax, 4), device=boxes.device); ks = torch.zeros((B, kmax), device=boxes.device)
        for b in range(B):
            sel = idx[b, :counts[b]].long()
            kb[b, :counts[b]] = boxes[b, sel]; ks[b, :counts[b]] = score[b, sel]
        idx2, count2 = ops.nms(kb.contiguous(), ks.contiguous(), 0.01, 0.5)
        torch.cuda.synchronize()
        assert count2.tolist() == counts
        for b in (0, B - 1):
            assert torch.equal(idx2[b, :counts[b]].long(), torch.arange(counts[b], device=idx2.device))


def test_graph_replayed_step_tracks_eager_at_benchmark_size():
    """configs[2] as bench.py times it (bf16, B=32 @512, drop_connect off for comparability): three replays of the captured
    step == three eager steps from the same weights -- losses and the updated parameters."""
    from efficientdet.pytorch_amd.optim import ClipAdamW
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    img, ann = O.synthetic_batch(32, 512, seed=1, num_classes=80)
    img, ann = img.cuda(), ann.cuda()

    def make():
        m = _model('efficientdet-d0', 80, torch.bfloat16, True)
        ps = [p for p in m.live_parameters()]
        return m, ps, ClipAdamW(ps, lr=1e-4, max_norm=0.1)
    me, pe, oe = make()
    eager = []
    for _ in range(3):                       # (GraphedTrainStep's 2 warm-up steps are steps 1-2 of the twin; capture itself runs nothing)
        oe.zero_grad(set_to_none=True)
        cl, rl = me([img, ann]); loss = cl.mean() + rl.mean(); loss.backward(); oe.step()
        eager.append(float(loss.detach()))
    mg, pg, og = make()
    step = GraphedTrainStep(mg, og, img, ann, warmup=2)
    outs = step()
    torch.cuda.synchronize()
    lg = float((outs[0].mean() + outs[1].mean()).detach())
    # the first replay is the 3rd optimizer step of the twin: compare like with like (no float atomics: the same kernels in the
    # same order give the same bits, so the loss is EQUAL up to the float() conversions)
    assert abs(lg - eager[2]) <= 1e-5 * abs(eager[2]), (lg, eager)
    worst, worst_abs = 0.0, 0.0
    for a, b in zip(pe, pg):
        d = float((a.detach() - b.detach()).abs().max())
        worst = max(worst, d / (float(a.detach().abs().max()) + 1e-12)); worst_abs = max(worst_abs, d)
    print('graph replay loss %.5f vs eager %.5f (eager history %s); worst parameter deviation after 3 steps %.2e of scale' % (lg, eager[2], eager, worst))
    assert worst_abs <= 3 * 2 * 1e-4 + 1e-6, (worst_abs, worst)          # AdamW moves a weight by <= lr per step: sign noise <= 2 lr per step


def test_decode_score_at_benchmark_size():
    """BBoxTransform + ClipBoxes + class max / first-arg-max over all 32 x 49 104 anchors == the oracle's formulas evaluated with
    torch ops on the device (the oracle function itself is dtype / device generic)."""
    from efficientdet.pytorch_amd import ops
    m = _model('efficientdet-d0', 80, torch.float32, False)
    img = O.synthetic_batch(32, 512, seed=4, num_classes=80)[0].cuda()
    with torch.no_grad():
        cls, reg, anc = m.forward_raw(img)
        boxes, score, label = ops.decode_score(anc, reg, cls, 512, 512)
        ref = O.decode_clip(anc.cpu(), reg.cpu(), 512, 512)
        smax, amax = cls.max(dim=2)
    torch.cuda.synchronize()
    assert float((boxes.cpu() - ref).abs().max()) <= 2e-3                  # pixels, on boxes up to 512 wide (exp is v_exp_f32 here)
    assert torch.equal(score, smax)
    assert torch.equal(label.long(), amax)                                 # (torch.max returns the first maximal index on ties too)
