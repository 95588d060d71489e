import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tests.test_gpu_hsplit import _run_h, _run_exact, _conv_ref64, _err
for xs in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0):
    g = torch.Generator().manual_seed(12)
    x = F.relu(torch.randn(1, 64, 8, 8, generator=g)) * xs
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    ref = _conv_ref64(x, w, None, 0)
    y, _ = _run_h(x, w, None, 0, True)
    e = _run_exact(x, w, None, 0)
    print('xscale %g: f16x3 err %.3e exact err %.3e' % (xs, _err(y.cpu().permute(0,3,1,2), ref), _err(e.permute(0,3,1,2), ref)))
