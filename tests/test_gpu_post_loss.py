"""GPU parity: anchors (bit-exact), decode+clip+score, NMS (identical keep lists), focal/smooth-L1 loss fwd+bwd."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import effdet_oracle as O
from tests.gpu_util import assert_close

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_anchors_bit_exact(golden_dir):
    from efficientdet.pytorch_amd import ops
    g = np.load(os.path.join(golden_dir, 'anchors.npz'))
    for (H, W) in [(128, 128), (512, 512), (1024, 1024), (256, 384), (640, 512)]:
        a = ops.anchors(H, W, 'cuda').cpu().numpy()
        assert a.shape[1] == int(g[f'n_{H}x{W}']) == ops.num_anchors(H, W)
        assert _sha(a) == str(g[f'sha_{H}x{W}']), (H, W)
        assert np.array_equal(a, O.anchors_for_image(H, W).numpy())


def test_decode_score_matches_oracle():
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(0)
    H, W, B, nc = 256, 384, 3, 20
    anc = O.anchors_for_image(H, W)
    A = anc.shape[1]
    reg = torch.randn(B, A, 4, generator=g) * 2.0
    cls = torch.rand(B, A, nc, generator=g)
    ref = O.decode_clip(anc, reg, H, W)
    boxes, score, label = ops.decode_score(anc.cuda(), reg.cuda(), cls.cuda(), H, W)
    assert_close(boxes.cpu(), ref, 1e-5, 'decode')
    ms, ml = cls.max(dim=2)
    assert torch.equal(score.cpu(), ms) and torch.equal(label.cpu().long(), ml)


def _nms_ref(boxes, score, thr, iou):
    mask = score > thr
    idx = torch.nonzero(mask).flatten()
    keep = O.nms_greedy(boxes[mask], score[mask], iou)
    return idx[keep]


@pytest.mark.parametrize('case', ['d0_128_eval', 'd0_512_eval'])
def test_nms_golden_candidates(golden_dir, case):
    """Candidates produced by the REAL reference (decouples NMS index parity from conv rounding)."""
    from efficientdet.pytorch_amd import ops
    g = np.load(os.path.join(golden_dir, case + '.npz'))
    boxes = torch.from_numpy(g['nms_boxes']); score = torch.from_numpy(g['nms_scores'])
    idx, cnt = ops.nms(boxes[None].cuda(), score[None].cuda(), 0.0, 0.5)
    n = int(cnt[0])
    assert n == len(g['nms_keep'])
    assert np.array_equal(idx[0, :n].cpu().numpy().astype(np.int64), g['nms_keep'])


def test_nms_batched_thresholds_ties_and_empty():
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(7)
    B, A = 4, 20000
    xy = torch.rand(B, A, 2, generator=g) * 480
    wh = 8 + torch.rand(B, A, 2, generator=g) * 90
    boxes = torch.cat([xy, xy + wh], 2)
    score = torch.rand(B, A, generator=g)
    score[1] = (score[1] * 16).floor() / 16          # massive exact ties -> stability of the sort
    score[2] = 0.001                                   # nothing passes the threshold
    boxes[3, :5000] = boxes[3, 0]                      # thousands of identical boxes
    idx, cnt = ops.nms(boxes.cuda(), score.cuda(), 0.05, 0.5)
    for b in range(B):
        ref = _nms_ref(boxes[b], score[b], 0.05, 0.5)
        n = int(cnt[b])
        assert n == len(ref), (b, n, len(ref))
        assert torch.equal(idx[b, :n].cpu().long(), ref), b
    sb, sl, bb = ops.gather_dets(boxes.cuda(), score.cuda(), torch.zeros(B, A, dtype=torch.int32, device='cuda'), idx, cnt)
    n0 = int(cnt[0])
    assert torch.equal(sb[0, :n0].cpu(), score[0][idx[0, :n0].cpu().long()])
    assert torch.equal(bb[0, :n0].cpu(), boxes[0][idx[0, :n0].cpu().long()])


@pytest.mark.parametrize('iou', [0.5, 0.7, 0.3])
def test_nms_adversarial_geometry(iou):
    """Spatial-hash NMS (iou >= 0.5) and the round-based form (iou < 0.5) == sequential greedy NMS on inputs built to break a
    spatial index: a 3000-box dependency chain (every box's fate hinges on its predecessor: the fixed point needs thousands of
    sweeps), degenerate / inverted / non-finite boxes, extreme aspect ratios, huge and negative coordinates, boxes whose areas
    sit exactly on octave boundaries, and nested boxes with IoU a hair either side of the threshold."""
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(3)
    rows = []
    # (0) dependency chain: unit squares of side 100 shifted by 30 px: IoU(k, k+1) = 0.538, IoU(k, k+2) = 0.25
    n = 3000
    x = torch.arange(n, dtype=torch.float32) * 30.0
    chain = torch.stack([x, torch.zeros(n), x + 100.0, torch.full((n,), 100.0)], 1)
    rows.append((chain, torch.linspace(0.99, 0.5, n)))
    # (1) mixed bag
    m = 6000
    xy = torch.rand(m, 2, generator=g) * 900 - 50                       # some negative coordinates
    wh = torch.exp(torch.rand(m, 2, generator=g) * 7.0)                 # sides 1 .. 1100, aspect ratios up to ~1000
    mixed = torch.cat([xy, xy + wh], 1)
    mixed[0:50, 2:] = mixed[0:50, :2]                                   # zero area
    mixed[50:100, 2] = mixed[50:100, 0] - 5.0                           # inverted in x (negative area)
    mixed[100:110] = float('nan'); mixed[110:115, 2] = float('inf'); mixed[115:120, 0] = -float('inf')
    mixed[120:130] = mixed[120:130] * 1.0e6                             # far away, huge
    side = 2.0 ** torch.arange(2, 12).float()                           # areas exactly 2^4 .. 2^22: octave boundaries
    for k, sd in enumerate(side):
        mixed[200 + 3 * k] = torch.tensor([300.0, 300.0, 300.0 + sd, 300.0 + sd])
        mixed[201 + 3 * k] = torch.tensor([300.0, 300.0, 300.0 + sd, 300.0 + sd * 0.5])        # exactly half the area, nested: IoU = 0.5
        mixed[202 + 3 * k] = torch.tensor([300.0, 300.0, 300.0 + sd, 300.0 + sd * 0.5000001])
    rows.append((mixed, torch.rand(m, generator=g)))
    # (2) dense anchor-like lattice with jitter (the benchmark's regime) + exact score ties
    gx, gy = torch.meshgrid(torch.arange(0, 256, 8.0), torch.arange(0, 256, 8.0), indexing='ij')
    ctr = torch.stack([gx.flatten(), gy.flatten()], 1).repeat_interleave(9, 0) + torch.rand(9216, 2, generator=g)
    sz = torch.tensor([32.0, 40.3, 50.8]).repeat_interleave(3).repeat(1024)[:, None] * torch.tensor([[1.0, 1.0], [0.7, 1.4], [1.4, 0.7]]).repeat(3072, 1)
    lat = torch.cat([ctr - sz / 2, ctr + sz / 2], 1).clamp(0, 256)
    rows.append((lat, (torch.rand(9216, generator=g) * 64).floor() / 64))
    A = max(len(r[0]) for r in rows)
    boxes = torch.zeros(len(rows), A, 4); score = torch.zeros(len(rows), A)
    for b, (bx, sc) in enumerate(rows):
        boxes[b, :len(bx)] = bx; score[b, :len(sc)] = sc
    idx, cnt = ops.nms(boxes.cuda(), score.cuda(), 0.01, iou)
    torch.cuda.synchronize()
    for b in range(len(rows)):
        ref = _nms_ref(boxes[b], score[b], 0.01, iou)
        n_ = int(cnt[b])
        assert n_ == len(ref), (b, n_, len(ref))
        assert torch.equal(idx[b, :n_].cpu().long(), ref), b
    if iou == 0.5:
        assert int(cnt[0]) == 1500                                      # every other box of the chain survives


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('nc', [20, 6])
def test_focal_loss_fwd_bwd(dtype, nc):
    from efficientdet.pytorch_amd import ops
    g = torch.Generator().manual_seed(11)
    S, B = 256, 4
    anc = O.anchors_for_image(S, S)
    A = anc.shape[1]
    _, ann = O.synthetic_batch(B, S, seed=3, num_classes=nc)
    ann[2] = -1.0                                        # an image without annotations
    logits = torch.randn(B, A, nc, generator=g) * 2.0
    logits[0, :50] = 12.0; logits[0, 50:100] = -12.0    # exercise the clamp
    cls = torch.sigmoid(logits).requires_grad_(True)
    reg = (torch.randn(B, A, 4, generator=g) * 0.5).requires_grad_(True)
    cl, rl = O.focal_loss(cls, reg, anc, ann)
    gs = torch.tensor([0.7, 1.3])
    (gs[0] * cl.sum() + gs[1] * rl.sum()).backward()
    losses, ws = ops.focal_loss_fwd(cls.detach().cuda(), reg.detach().cuda(), anc.cuda(), ann.cuda())
    assert_close(losses.cpu(), torch.cat([cl.detach(), rl.detach()]), 2e-4, 'losses')
    dcls, dreg = ops.focal_loss_bwd(cls.detach().cuda(), reg.detach().cuda(), anc.cuda(), ann.cuda(), gs.cuda(), ws, dtype)
    ref_dlogit = cls.grad * cls.detach() * (1 - cls.detach())
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    assert_close(dcls.float().cpu(), ref_dlogit, tol, 'dcls_logit')
    assert_close(dreg.float().cpu(), reg.grad, tol, 'dreg')
    if nc % 4 == 0:      # the pixel-major, channel-padded variant that feeds the head's gradient convs: same values, zero padding
        dld = (9 * nc + 63) // 64 * 64
        dpix, dreg2 = ops.focal_loss_bwd_pix(cls.detach().cuda(), reg.detach().cuda(), anc.cuda(), ann.cuda(), gs.cuda(), ws, dtype, dld)
        assert dpix.shape == (B, A // 9, dld)
        assert torch.equal(dpix[:, :, :9 * nc].reshape(B, A, nc), dcls) and torch.equal(dreg2, dreg)
        assert float(dpix[:, :, 9 * nc:].float().abs().max()) == 0.0
        # training fast path: losses + the gradient for an upstream gradient of ONE in a single pass over cls, d(reg) separately
        losses2, ws2, dpix1 = ops.focal_loss_fwd_grad(cls.detach().cuda(), reg.detach().cuda(), anc.cuda(), ann.cuda(), dtype, dld)
        assert_close(losses2.cpu(), torch.cat([cl.detach(), rl.detach()]), 2e-4, 'losses (fwd_grad)')
        assert_close(dpix1.float().cpu() * float(gs[0]), dpix.float().cpu(), 1e-5 if dtype == torch.float32 else 1e-2, 'dcls (fwd_grad)')
        assert float(dpix1[:, :, 9 * nc:].float().abs().max()) == 0.0
        dreg3 = ops.focal_loss_bwd_reg(reg.detach().cuda(), anc.cuda(), ann.cuda(), gs.cuda(), ws2, dtype)
        assert torch.equal(dreg3, dreg)


@pytest.mark.parametrize('max_norm', [0.1, 0.0])
def test_clip_adamw_matches_torch(max_norm):
    """effdet_clip_adamw_step vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the same GPU tensors
    (ragged sizes, an unaligned gradient view, a parameter without gradient), three steps."""
    from efficientdet.pytorch_amd.optim import ClipAdamW
    g = torch.Generator().manual_seed(21)
    shapes = [(256, 256, 3, 3), (17,), (4097,), (3, 5, 7), (1,), (720, 256, 3, 3), (64,), (8193,)]
    pa = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = ClipAdamW(pa, lr=1e-2, weight_decay=0.01, max_norm=max_norm)
    ob = torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01)
    for it in range(3):
        big = torch.randn(20000, generator=g).cuda() * (3.0 if it else 0.01)      # first step: norm below max_norm*...
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 4 and it == 1:
                a.grad = None; b.grad = None                                        # parameter skipped this step
                continue
            gr = torch.randn(a.shape, generator=g).cuda() * (0.02 if it == 0 else 1.0)
            if i == 1:
                a.grad = big[3:3 + 17].view(17)                                     # 12-byte-offset view: unaligned path
                b.grad = a.grad.clone()
            else:
                a.grad = gr; b.grad = gr.clone()
        if max_norm:
            ref_norm = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        oa.step(); ob.step()
        torch.cuda.synchronize()
        if max_norm:
            assert abs(float(oa.grad_norm()) - float(ref_norm)) <= 2e-6 * float(ref_norm)
        for a, b in zip(pa, pb):
            assert_close(a.detach().cpu(), b.detach().cpu(), 1e-5, 'param step %d' % it)
    st = oa.state_dict()
    assert len(st['state']) == len(shapes)


def test_clip_adamw_is_bitwise_independent_of_gradient_alignment():
    """Gradients handed over as views into a flat bucket at arbitrary 4-byte offsets (DistributedDataParallel with
    gradient_as_bucket_view) must give the same bits as separately allocated ones: the norm's summation order -- hence the clip
    coefficient, hence every updated parameter -- may not depend on pointer alignment (found by tests/test_gpu_rccl.py)."""
    from efficientdet.pytorch_amd.optim import ClipAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 32, 3, 3), (6,), (4099,), (10,), (720, 64, 3, 3), (1,), (8190,)]
    pa = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = ClipAdamW(pa, lr=1e-2, weight_decay=0.01, max_norm=0.1)
    ob = ClipAdamW(pb, lr=1e-2, weight_decay=0.01, max_norm=0.1)
    for it in range(3):
        grads = [torch.randn(s, generator=g).cuda() for s in shapes]
        n = sum(x.numel() for x in grads)
        bucket = torch.empty(n + 3, device='cuda')
        off = 1 + it                                            # 4-, 8-, 12-byte offsets: every tensor lands unaligned somewhere
        for a, b, gr in zip(pa, pb, grads):
            a.grad = gr
            v = bucket[off:off + gr.numel()].view(gr.shape); v.copy_(gr); b.grad = v
            off += gr.numel()
        oa.step(); ob.step()
        torch.cuda.synchronize()
        assert torch.equal(oa.grad_norm(), ob.grad_norm())
        for a, b in zip(pa, pb):
            assert torch.equal(a.detach(), b.detach()), ('step', it, a.shape)
    assert torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
