"""Device-side input pipeline: the work of the reference's Normalizer -> Augmenter -> Resizer ->
collater chain (datasets/augmentation.py:69-150) as ONE HIP launch per batch.

The host only concatenates the decoded uint8 HWC images into a pinned staging buffer (one memcpy per image) and issues
one async H2D copy; bilinear resize to the common size, /255, mean/std normalisation, horizontal flip, zero padding, the
NHWC / compute-dtype / channel-padded pack the stem conv reads, and the annotation rescale all happen on the GPU
(csrc/pipeline.hip::preprocess_kernel).  The float64 512x512x3 canvas of augmentation.py:111, the NCHW permute of the
collater and the model-side NCHW -> NHWC repack disappear.  There is no CPU fallback: the collater needs the GPU."""
import numpy as np
import torch

from . import ops
from .efficientdet import PackedImages
from .functional import chunk_elems

MEAN = (0.485, 0.456, 0.406)       # datasets/augmentation.py:141-142
STD = (0.229, 0.224, 0.225)


class DeviceCollater:
    """collate_fn replacement: ``batch = collater(samples)`` -> (PackedImages, annotations [B,M,5] fp32 device, scales [B]).

    samples: list of dicts {'img': uint8 ndarray [H,W,3] RGB, 'annot': float ndarray [n,5]} -- what the reference's
    datasets yield BEFORE its transform chain (the chain runs here, on the device).  flip_x: probability of the
    Augmenter's horizontal flip (0 disables; eval pipelines have none)."""

    def __init__(self, common_size=512, dtype=torch.bfloat16, device='cuda', flip_x=0.0, seed=0):
        self.S, self.dtype, self.device = int(common_size), dtype, torch.device(device)
        self.flip_x = float(flip_x)
        self.rng = np.random.RandomState(seed)
        self._stage = [None, None]      # two pinned staging buffers: batch k+1 is assembled while batch k's copy is in flight
        self._evt = [None, None]
        self._slot = 0

    def _pinned(self, nbytes):
        s = self._slot; self._slot ^= 1
        if self._evt[s] is not None:
            self._evt[s].synchronize()
        if self._stage[s] is None or self._stage[s].numel() < nbytes:
            self._stage[s] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return s, self._stage[s]

    def __call__(self, samples):
        B = len(samples)
        imgs = [np.ascontiguousarray(s['img']) for s in samples]
        for im in imgs:
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('DeviceCollater takes decoded uint8 HWC RGB images')
        sizes = [im.size for im in imgs]
        offs = np.concatenate([[0], np.cumsum([(n + 15) // 16 * 16 for n in sizes])]).astype(np.int64)
        slot, stage = self._pinned(int(offs[-1]))
        view = stage.numpy()
        for im, o, n in zip(imgs, offs[:-1], sizes):
            view[o:o + n] = im.reshape(-1)
        hw = np.array([[im.shape[0], im.shape[1]] for im in imgs], dtype=np.int32)
        flips = (self.rng.rand(B) < self.flip_x).astype(np.uint8) if self.flip_x > 0 else None
        M = max(1, max((len(s['annot']) for s in samples), default=0))
        ann = np.full((B, M, 5), -1.0, dtype=np.float32)                    # collater's padding (augmentation.py:78-86)
        for b, s in enumerate(samples):
            a = np.asarray(s['annot'], dtype=np.float32).reshape(-1, 5)
            ann[b, :len(a)] = a
        dev = self.device
        src = stage[:int(offs[-1])].to(dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); self._evt[slot] = ev
        d_off = torch.from_numpy(offs[:-1].copy()).to(dev, non_blocking=True)
        d_hw = torch.from_numpy(hw).to(dev, non_blocking=True)
        d_flip = torch.from_numpy(flips).to(dev, non_blocking=True) if flips is not None else None
        d_ann = torch.from_numpy(ann).to(dev, non_blocking=True)
        m, scale = ops.preprocess_batch(src, d_off, d_hw, self.S, self.dtype, chunk_elems(self.dtype), MEAN, STD, d_flip, d_ann)
        return PackedImages(m), d_ann, scale
