"""Bitwise reproducibility of the HIP path: the reference on CPU is deterministic (models/efficientnet.py:86-92,
models/losses.py:86-104 are plain sums), so two runs of the same binary on the same inputs must agree to the last bit --
every reduction of the path (squeeze-excite pool, loss statistics, conv bias / frozen-BN sum(dz), BiFPN fusion-weight
gradients, SE gate gradient, depthwise weight gradient, split-K slabs) is a fixed-order sum of per-workgroup partials; there is
no float atomicAdd left in csrc/ (tests/test_abi.py greps for it)."""
import copy

import pytest
import torch

from oracle import effdet_oracle as O

pytestmark = pytest.mark.gpu

MODES = {'f32': (torch.float32, 'f32'), 'f32_bf16x3': (torch.float32, 'bf16x3'), 'bf16': (torch.bfloat16, 'f32'),
         'f32_bwd_bf16x3': (torch.float32, 'f32_bwd_bf16x3'),
         'f32_hf16x3_bwd_bf16x3': (torch.float32, 'f32_hf16x3_bwd_bf16x3')}      # (the last: the bench headline)


def _model(net, nc, mode, seed=0, **kw):
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
    c = EFFICIENTDET[net]
    dt, arith = MODES[mode]
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=dt,
                     f32_arith=arith, **kw)
    m.load_state_dict(O.make_state_dict(net, nc, seed=seed))
    return m


def _train_pass(m, img, ann):
    for p in m.parameters():
        p.grad = None
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    torch.cuda.synchronize()
    out = {'cls_loss': cl.detach().clone(), 'reg_loss': rl.detach().clone()}
    out.update({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    return out


def _assert_bitwise(a, b, what):
    assert a.keys() == b.keys()
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, '%s: %d of %d tensors differ between two runs, e.g. %s (max abs diff %.3e)' % (
        what, len(bad), len(a), bad[:3], float((a[bad[0]].float() - b[bad[0]].float()).abs().max()))


@pytest.mark.parametrize('mode', list(MODES))
def test_two_train_passes_are_bitwise_equal_d0_128(mode):
    """Forward + loss + backward twice on the same model (drop_connect off: its step counter would advance), and once more on
    a deep copy made before the first run (different parameter / workspace addresses): losses and all 274 gradients bit for bit."""
    net, nc = 'efficientdet-d0', 20
    m = _model(net, nc, mode).cuda()
    m.backbone.drop_connect_rate = 0.0
    m.train(); m.is_training = True; m.freeze_bn()
    twin = copy.deepcopy(m)
    img, ann = O.synthetic_batch(3, 128, seed=2, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    a = _train_pass(m, img, ann)             # records the parameter-preparation table
    b = _train_pass(m, img, ann)             # replays it
    c = _train_pass(m, img, ann)
    assert len(a) == 274 + 2
    _assert_bitwise(a, b, 'record vs replay')
    _assert_bitwise(b, c, 'replay vs replay')
    filler = torch.empty(37 * 1024 * 1024, device='cuda')      # shift every later allocation
    t = _train_pass(twin, img, ann)
    del filler
    _assert_bitwise(a, t, 'model vs deep copy at other addresses')


def test_two_train_passes_are_bitwise_equal_at_benchmark_size():
    """BASELINE configs[2] itself: D0, batch 32 @ 512x512, 80 classes, fp32 -- two passes bitwise equal (every split-K / tile-group
    partial count of the real workload, e.g. 22 pool groups per image, 500+ weight-gradient slabs)."""
    net, nc = 'efficientdet-d0', 80
    m = _model(net, nc, 'f32').cuda()
    m.backbone.drop_connect_rate = 0.0
    m.train(); m.is_training = True; m.freeze_bn()
    img, ann = O.synthetic_batch(32, 512, seed=1, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    a = _train_pass(m, img, ann)
    b = _train_pass(m, img, ann)
    _assert_bitwise(a, b, 'D0 B=32 @512 fp32')
    assert all(bool(torch.isfinite(v).all()) for v in a.values())


@pytest.mark.parametrize('mode', ['f32_hf16x3_bwd_bf16x3', 'f32_bwd_bf16x3', 'f32_bf16x3', 'bf16'])
def test_two_train_passes_are_bitwise_equal_at_benchmark_size_fast_modes(mode):
    net, nc = 'efficientdet-d0', 80
    m = _model(net, nc, mode).cuda()
    m.backbone.drop_connect_rate = 0.0
    m.train(); m.is_training = True; m.freeze_bn()
    img, ann = O.synthetic_batch(32, 512, seed=1, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    a = _train_pass(m, img, ann)
    b = _train_pass(m, img, ann)
    _assert_bitwise(a, b, 'D0 B=32 @512 ' + mode)


def test_drop_connect_runs_are_reproducible_from_the_seed():
    """drop_connect ACTIVE: two models built under the same torch.manual_seed draw the same Philox seed, hence the same masks
    at the same step -> bitwise-equal losses and gradients step by step."""
    net, nc = 'efficientdet-d0', 8
    img, ann = O.synthetic_batch(4, 128, seed=5, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()
    runs = []
    for _ in range(2):
        m = _model(net, nc, 'f32').cuda()
        m.train(); m.is_training = True; m.freeze_bn()
        torch.manual_seed(77)
        runs.append([_train_pass(m, img, ann) for _ in range(2)])
    _assert_bitwise(runs[0][0], runs[1][0], 'step 0')
    _assert_bitwise(runs[0][1], runs[1][1], 'step 1')
    assert not torch.equal(runs[0][0]['cls_loss'], runs[0][1]['cls_loss'])      # different masks at step 1


@pytest.mark.parametrize('mode', list(MODES))
def test_eval_forward_and_detections_are_bitwise_equal(mode):
    net, nc = 'efficientdet-d0', 20
    m = _model(net, nc, mode, is_training=False, threshold=0.4).cuda().eval()
    img, _ = O.synthetic_batch(3, 128, seed=4, num_classes=nc)
    img = img.cuda()
    with torch.no_grad():
        c0, r0, _ = m.forward_raw(img)
        c1, r1, _ = m.forward_raw(img)
        c2, r2, _ = m.forward_raw(img)       # (the re-pack launch is skipped here: same packed weights)
    assert torch.equal(c0, c1) and torch.equal(r0, r1) and torch.equal(c1, c2) and torch.equal(r1, r2)
    d0, d1 = m.detect(img), m.detect(img)
    for (s0, l0, b0), (s1, l1, b1) in zip(d0, d1):
        assert torch.equal(s0, s1) and torch.equal(l0, l1) and torch.equal(b0, b1)
