"""Synthetic inputs of the benchmark: randn images and COCO-shape targets.

``synthetic_batch`` draws exactly the stream the parity tests' generator draws (tests assert the two functions agree
tensor for tensor), so bench.py's GPU leg needs nothing outside this package."""
import torch


def synthetic_batch(B, S, seed=1, max_boxes=8, num_classes=80):
    """-> (images [B,3,S,S] fp32 ~ N(0,1), annotations [B,max_boxes,5] fp32 = x1,y1,x2,y2,label; unused rows = -1
    (the layout ``collater`` builds, datasets/augmentation.py:69-91))."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, S, S, generator=g)
    ann = torch.full((B, max_boxes, 5), -1.0)
    for b in range(B):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        x1 = torch.rand(n, generator=g) * 0.78 * S
        y1 = torch.rand(n, generator=g) * 0.78 * S
        w = 16 + torch.rand(n, generator=g) * 0.4 * S
        h = 16 + torch.rand(n, generator=g) * 0.4 * S
        ann[b, :n, 0], ann[b, :n, 1] = x1, y1
        ann[b, :n, 2] = (x1 + w).clamp(max=S - 1)
        ann[b, :n, 3] = (y1 + h).clamp(max=S - 1)
        ann[b, :n, 4] = torch.randint(0, num_classes, (n,), generator=g).float()
    return img, ann


def synthetic_raw_images(B, seed=1, min_side=240, max_side=640):
    """Decoded-JPEG stand-ins for the input pipeline: B uint8 HWC RGB arrays of mixed sizes + pixel-space boxes."""
    import numpy as np
    rng = np.random.RandomState(seed)
    imgs, annots = [], []
    for _ in range(B):
        h, w = int(rng.randint(min_side, max_side + 1)), int(rng.randint(min_side, max_side + 1))
        imgs.append(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8))
        n = int(rng.randint(0, 6))
        a = np.zeros((n, 5), dtype=np.float32)
        if n:
            a[:, 0] = rng.uniform(0, 0.6 * w, n); a[:, 1] = rng.uniform(0, 0.6 * h, n)
            a[:, 2] = a[:, 0] + rng.uniform(8, 0.4 * w, n); a[:, 3] = a[:, 1] + rng.uniform(8, 0.4 * h, n)
            a[:, 4] = rng.randint(0, 80, n)
        annots.append(a)
    return imgs, annots
