"""Helpers shared by the -m gpu parity tests."""
import torch


def assert_close(got, ref, tol, what=''):
    """|a-b| <= tol * max(|b|, tiny) with tiny = 1e-2 * |b|_max (SURVEY §8d parity gate)."""
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    floor = 1e-2 * float(ref.abs().max()) + 1e-30
    bound = tol * torch.maximum(ref.abs(), torch.tensor(floor))
    bad = (got - ref).abs() > bound
    if bool(bad.any()) or not bool(torch.isfinite(got).all()):
        i = int(torch.nonzero(bad.reshape(-1))[0]) if bool(bad.any()) else 0
        raise AssertionError('%s: %d/%d elements exceed tol %g (first at %d: got %g ref %g; max abs err %g, ref max %g)' % (
            what, int(bad.sum()), bad.numel(), tol, i, float(got.reshape(-1)[i]), float(ref.reshape(-1)[i]),
            float((got - ref).abs().max()), float(ref.abs().max())))


def assert_close_scale(got, ref, tol, what=''):
    """Tensor-scale criterion for the bf16 throughput mode: |a-b| <= tol * max|b| for every element."""
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert bool(torch.isfinite(got).all()), what
    err = float((got - ref).abs().max()); scale = float(ref.abs().max()) + 1e-30
    assert err <= tol * scale, '%s: max abs err %g > %g * ref max %g' % (what, err, tol, scale)


def unmatched_detections(got, ref, score_tol, box_tol=1e-2):
    """Compare two detection lists (scores, labels, boxes) as SETS: every reference detection must have its own partner with the same
    label, every box coordinate within box_tol pixels and the score within score_tol (absolute).  -> (number of reference detections
    without a partner, number of extra detections).  Used where neighbouring scores are closer than conv rounding, so the ORDER of two
    near-tied boxes is not defined but their membership is."""
    import numpy as np
    gs, gl, gb = (np.asarray(t.detach().cpu() if hasattr(t, 'detach') else t) for t in got)
    rs, rl, rb = (np.asarray(t.detach().cpu() if hasattr(t, 'detach') else t) for t in ref)
    free = np.ones(len(gs), dtype=bool)
    missing = 0
    for i in range(len(rs)):
        ok = free & (gl == rl[i]) & (np.abs(gs - rs[i]) <= score_tol) & (np.abs(gb - rb[i]).max(axis=1) <= box_tol if len(gs) else free)
        j = np.nonzero(ok)[0]
        if len(j):
            free[j[np.argmin(np.abs(gs[j] - rs[i]))]] = False
        else:
            missing += 1
    return missing, int(free.sum())
