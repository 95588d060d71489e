"""CPU restatements (numpy) of the boundary steps either side of the conv path.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product.

* philox4x32_10 / drop_connect_scales -- the counter-based generator behind the HIP drop_connect row scales, restated from
  the published algorithm (Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11; Random123 v1.09
  philox.h: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85, 10 rounds) and pinned on Random123's
  known-answer vectors (tests/test_pipeline_oracle.py).  The mask arithmetic is models/utils.py:79-90.
* preprocess_reference -- datasets/augmentation.py:94-150 (Normalizer -> Augmenter flip -> Resizer) + collater :69-91.
  **parity unpinned for the resize**: the interpolation lives in OpenCV (`cv2.resize`, third-party, unpinned in
  requirements.txt, not installed here); this restates its documented INTER_LINEAR rule (half-pixel centres,
  src = (dst + 0.5) * src/dst - 0.5, edge clamp).  Everything else (normalise, pad, flip, annotation scale) follows the
  reference line by line.
* finalize_reference -- eval.py:96-127 (_get_detections) and eval.py:279-306 (evaluate_coco) for one image.
"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c = [int(x) & MASK for x in ctr]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def drop_connect_scales(keep, B, seed, step):
    """[nslot, B] float32: floor(keep + u)/keep with u = (word0 >> 8) * 2^-24 of Philox(ctr = {b, slot, step}, key = seed)."""
    out = np.zeros((len(keep), B), dtype=np.float32)
    for s, kp in enumerate(keep):
        kp = np.float32(kp)
        for b in range(B):
            w = philox4x32_10([b, s, step & MASK, (step >> 32) & MASK], [seed & MASK, (seed >> 32) & MASK])
            u = np.float32(w[0] >> 8) * np.float32(1.0 / 16777216.0)
            out[s, b] = np.floor(np.float32(kp + u)) / kp
    return out


# ------------------------------------------------------------------------------------------------ input pipeline
MEAN = np.array([[[0.485, 0.456, 0.406]]])
STD = np.array([[[0.229, 0.224, 0.225]]])


def resize_bilinear(img, rw, rh):
    """cv2.resize(img, (rw, rh)) with the default INTER_LINEAR, float image [h, w, c]."""
    h, w = img.shape[:2]
    sx = ((np.arange(rw, dtype=np.float64) + 0.5) * (w / rw) - 0.5).astype(np.float32)
    sy = ((np.arange(rh, dtype=np.float64) + 0.5) * (h / rh) - 0.5).astype(np.float32)

    def split(s, n):
        i0 = np.floor(s).astype(np.int64); f = (s - i0).astype(np.float32)
        lo = i0 < 0; i0[lo] = 0; f[lo] = 0
        hi = i0 >= n - 1; i0[hi] = n - 1; f[hi] = 0
        return i0, np.minimum(i0 + 1, n - 1), f
    x0, x1, fx = split(sx, w)
    y0, y1, fy = split(sy, h)
    im = img.astype(np.float32)
    top = im[y0][:, x0] + fx[None, :, None] * (im[y0][:, x1] - im[y0][:, x0])
    bot = im[y1][:, x0] + fx[None, :, None] * (im[y1][:, x1] - im[y1][:, x0])
    return top + fy[:, None, None] * (bot - top)


def preprocess_reference(img_u8, annots, common_size=512, flip=False):
    """One sample through the reference's chain, in the reference's order: load (/255, datasets/coco.py load_image),
    Normalizer (:145-150), Augmenter flip (:124-138), Resizer (:94-114).  -> (image [S,S,3] float32, annots, scale)."""
    image = img_u8.astype(np.float32) / 255.0
    annots = np.array(annots, dtype=np.float64).reshape(-1, 5).copy()
    image = ((image.astype(np.float32) - MEAN) / STD)
    if flip:
        image = image[:, ::-1, :]
        cols = image.shape[1]
        x1 = annots[:, 0].copy(); x2 = annots[:, 2].copy()
        annots[:, 0] = cols - x2; annots[:, 2] = cols - x1
    height, width, _ = image.shape
    if height > width:
        scale = common_size / height; rh = common_size; rw = int(width * scale)
    else:
        scale = common_size / width; rh = int(height * scale); rw = common_size
    image = resize_bilinear(image, rw, rh)
    new_image = np.zeros((common_size, common_size, 3))
    new_image[0:rh, 0:rw] = image
    annots[:, :4] *= scale
    return new_image.astype(np.float32), annots.astype(np.float32), scale


def collate_reference(samples, common_size=512, flips=None):
    """collater (:69-91): stacked NCHW float images, annotations padded with -1 to the longest list."""
    outs = [preprocess_reference(s['img'], s['annot'], common_size, bool(flips[i]) if flips is not None else False)
            for i, s in enumerate(samples)]
    imgs = np.stack([o[0] for o in outs]).transpose(0, 3, 1, 2)
    M = max(1, max(len(o[1]) for o in outs))
    ann = -np.ones((len(outs), M, 5), dtype=np.float32)
    for i, o in enumerate(outs):
        ann[i, :len(o[1])] = o[1]
    return imgs, ann, np.array([o[2] for o in outs])


# ------------------------------------------------------------------------------------------------ eval consumer
def finalize_reference(scores, labels, boxes, scale, score_threshold=0.05, max_detections=100):
    """eval.py:104-117 for one image: -> image_detections [n, 6] = boxes/scale, score, label."""
    scores = np.asarray(scores, dtype=np.float32).copy(); labels = np.asarray(labels).copy()
    boxes = np.asarray(boxes, dtype=np.float32).copy()
    boxes /= float(scale)        # (the reference's scale is a PYTHON float, datasets/augmentation.py:100-104: the division runs in float32)
    indices = np.where(scores > score_threshold)[0]
    if indices.shape[0] == 0:
        return np.zeros((0, 6), dtype=np.float32)
    scores = scores[indices]
    scores_sort = np.argsort(-scores, kind='stable')[:max_detections]
    return np.concatenate([boxes[indices[scores_sort], :], scores[scores_sort][:, None],
                           labels[indices[scores_sort]][:, None].astype(np.float32)], axis=1)


def coco_results_reference(scores, labels, boxes, scale, image_id, threshold=0.05, label_to_coco_label=lambda c: c):
    """eval.py:279-306 for one image (scores descending)."""
    boxes = np.asarray(boxes, dtype=np.float32).copy()
    boxes /= float(scale)
    res = []
    if boxes.shape[0] > 0:
        boxes[:, 2] -= boxes[:, 0]; boxes[:, 3] -= boxes[:, 1]
        for i in range(boxes.shape[0]):
            if float(scores[i]) < threshold:
                break
            res.append({'image_id': image_id, 'category_id': label_to_coco_label(int(labels[i])), 'score': float(scores[i]),
                        'bbox': boxes[i].tolist()})
    return res
