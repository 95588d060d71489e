"""Thin Python wrappers over the C ABI (raw device tensors in, raw device tensors out).

Plumbing only: PyTorch owns the memory and the stream; every numeric op is a HIP kernel in
libeffdet_hip.so.  No function here has a CPU path.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib as L
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH, RES_ADD, RES_NONE,  # noqa: F401
                   RES_RELU_MASK, RES_SWISH_GRAD)


class LaunchProfile:
    """Optional per-launch HIP-event timing of the MFMA kernels (bench.py's live roofline measurement).
    Events are recorded on the SAME stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []          # (kernel symbol, algorithmic flops, start event, end event, shape note)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1, _, nbytes in self.records:
            d = out.setdefault(name, {'launches': 0, 'flops': 0.0, 'ms': 0.0, 'bytes': 0.0})
            d['launches'] += 1; d['flops'] += flops; d['ms'] += e0.elapsed_time(e1); d['bytes'] += nbytes
        return out

    def by_shape(self, name, top=4):
        """The launches of one kernel symbol grouped by shape note, heaviest first (a symbol mixes MFMA-bound head shapes with
        HBM-bound backbone ones; this shows them apart)."""
        torch.cuda.synchronize()
        agg = {}
        for n, flops, e0, e1, note, _ in self.records:
            if n == name:
                d = agg.setdefault(note, {'launches': 0, 'flops': 0.0, 'ms': 0.0})
                d['launches'] += 1; d['flops'] += flops; d['ms'] += e0.elapsed_time(e1)
        rows = sorted(agg.items(), key=lambda kv: -kv[1]['ms'])[:top]
        return {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)}
                for k, v in rows}


PROFILE = None     # set to a LaunchProfile() to time every conv launch


def _timed(name, flops, fn, note='', nbytes=0.0):
    """flops: algorithmic FLOPs of an MFMA launch (for the byte-counted kernels, whose notes start with 'BYTES': their algorithmic bytes);
    nbytes: algorithmic bytes of an MFMA launch (operands read once + outputs written once), next to the PMC traffic in the roofline."""
    if PROFILE is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record()
    PROFILE.records.append((name, flops, e0, e1, note, nbytes))
    return r


# Arithmetic of the MFMA kernels on fp32 STORAGE (process-wide; EfficientDet(..., f32_arith=...) sets it):
#   'f32'    v_mfma_f32_16x16x4_f32 -- exact fp32 products (the strict parity mode)
#   'bf16x3' operands split into bf16 hi + lo in registers, 3 bf16 MFMAs per product (~16 mantissa bits), fp32 accumulate
# F32_ARITH is what the launches issued NOW use; F32_ARITH_BWD is what the autograd nodes recorded by the running forward will
# switch to in their backward.  A MODEL names the pair (MODEL_ARITH): 'f32' and 'bf16x3' use one arithmetic throughout;
# 'f32_bwd_bf16x3' computes every forward value -- the (classification, regression, anchors) triple of models/efficientdet.py:64-66,
# the losses, every ReLU / max-pool / IoU decision the backward depends on -- with exact fp32 products, bit for bit what 'f32'
# computes, and takes only the GRADIENT convolutions (data + weight gradients) through the three-product bf16 form: their ~1e-5
# product error lands on gradients whose masks were decided exactly, 100x inside the 1e-3 gradient gate.
# F32_ARITH_HEAD: arithmetic of the RetinaHead's FORWARD convs when F32_ARITH is 'f32' -- 'f32' (exact), or 'f16x3': operands as
# fp16 hi + scaled fp16 lo (22 significand bits), 3 fp16 MFMAs per product into one fp32 accumulator.  Per product ~2^-22 relative --
# below the rounding noise of the fp32 accumulation itself (measured: outputs within 1.2x of the exact mode's distance to a float64
# head) -- at the fp16 matrix rate: "fp32-equivalent", not bit-identical to the exact mode.
# 'f32_hf16x3_bwd_bf16x3' = that forward head + exact fp32 everywhere else in the forward + bf16x3 gradient convs.
F32_ARITH = 'f32'
F32_ARITH_BWD = 'f32'
F32_ARITH_HEAD = 'f32'
MODEL_ARITH = {'f32': ('f32', 'f32', 'f32'), 'bf16x3': ('bf16x3', 'bf16x3', 'f32'), 'f32_bwd_bf16x3': ('f32', 'bf16x3', 'f32'),
               'f32_hf16x3_bwd_bf16x3': ('f32', 'bf16x3', 'f16x3')}


def set_f32_arith(mode, bwd=None):
    """Arithmetic of the launches from now on (and, unless `bwd` says otherwise, of the backward of nodes recorded from now on)."""
    global F32_ARITH, F32_ARITH_BWD
    if mode not in ('f32', 'bf16x3') or (bwd is not None and bwd not in ('f32', 'bf16x3')):
        raise ValueError("f32_arith must be 'f32' or 'bf16x3'")
    old, F32_ARITH = F32_ARITH, mode
    F32_ARITH_BWD = mode if bwd is None else bwd
    return old


def set_model_arith(name):
    """A model's arithmetic by name (MODEL_ARITH): forward arithmetic now, its backward arithmetic for the nodes it records."""
    if name not in MODEL_ARITH:
        raise ValueError('f32_arith must be one of %s' % (sorted(MODEL_ARITH),))
    global F32_ARITH_HEAD
    fwd, bwd, head = MODEL_ARITH[name]
    set_f32_arith(fwd, bwd)
    F32_ARITH_HEAD = head


# Out-of-fp16-range watch of the f16x3 arithmetic: one device int per GPU that every H-split producer (pyramid conversion, BiFPN fusion,
# tower conv epilogues) ORs bit 0 into when a value cannot be held (|v| >= 65520 or NaN).  Created outside stream capture (the model's
# forward asks for it before anything else), never reset by the kernels.
_range_flags = {}


def range_flag(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _range_flags.get(key)
    if t is None:
        t = _range_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def check_range_flag(device, what='f16x3'):
    """Host-side read of the watch (synchronises the current stream): raises -- and clears the flag -- if a value left fp16's range since
    the last check."""
    t = range_flag(device)
    if int(t.item()) != 0:
        t.zero_()
        raise FloatingPointError('%s: an activation of the RetinaHead / BiFPN left the range of the fp16 operand split (|x| >= 65520 or NaN); '
                                 "the outputs of this call are not valid -- use f32_arith='f32_bwd_bf16x3' or 'f32' for this model" % what)


class backward_scope:
    """with backward_scope(ctx.prep, ctx.arith): the body of an autograd node's backward -- this model's parameter arena and its BACKWARD
    arithmetic for the launches inside, and the arithmetic found at entry put back at exit, so that direct ops.conv2d / functional.* calls
    issued between a backward and the next model forward do not silently inherit the gradient arithmetic."""

    def __init__(self, prep, arith):
        self.prep, self.arith = prep, arith

    def __enter__(self):
        self.old = (F32_ARITH, F32_ARITH_BWD, F32_ARITH_HEAD)
        set_prep(self.prep); set_f32_arith(self.arith)
        return self

    def __exit__(self, et, ev, tb):
        global F32_ARITH, F32_ARITH_BWD, F32_ARITH_HEAD
        F32_ARITH, F32_ARITH_BWD, F32_ARITH_HEAD = self.old
        return False


def _mma_dtype_code(dtype, K=None, N=None):
    """dtype code of an MFMA launch.  For the dense conv (K, N given) the bf16x3 form also needs weights packed in its
    pre-split layout, so the SAME rule decides the pack (pack_weight) and the launch (conv2d): reduction length a
    multiple of 32 and long enough for the matrix pipe to be the bound."""
    code = L.dtype_code(dtype)
    if code != L.F32 or F32_ARITH != 'bf16x3':
        return code
    if K is not None and (K % 32 or K < 256 or N < 32):
        return code
    return L.F32_BF16X3


def _igemm_symbol(dtype, desc):
    """Kernel symbol the library will launch for this descriptor (profiling attribution only)."""
    kid = int(L.lib().effdet_conv2d_kernel(C.byref(desc)))
    if kid in (30, 31):
        return 'conv_igemm_kernel<hsplit,%d,f16x3>' % (128, 64)[kid - 30]
    if kid >= 10000:
        return 'conv_igemm_pers_kernel<split,bf16x3>'
    if kid == 20:
        return 'conv_pw_f32_kernel'
    if kid >= 10:
        v = str(kid - 10)
        return 'conv_igemm_pers_kernel<%s,%s,%s>' % (v[0], v[1], v[2])
    if kid in (8, 9):
        return 'conv_igemm_kernel<split,%d,bf16x3>' % (128, 64)[kid - 8]
    if 4 <= kid < 8:
        return 'conv_igemm_kernel<f32,%d,bf16x3>' % (128, 64, 32, 16)[kid - 4]
    return 'conv_igemm_kernel<%s,%d>' % ('bf16' if dtype == torch.bfloat16 else 'f32', (128, 64, 32, 16)[max(kid, 0)])


class Map:
    """An NHWC feature-map view: element (b,h,w,c) at  t.data_ptr() + (off + b*bstride + (h*W+w)*ld + c)*itemsize."""
    __slots__ = ('t', 'B', 'H', 'W', 'C', 'ld', 'bstride', 'off')

    def __init__(self, t, B, H, W, C, ld=None, bstride=None, off=0):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.ld = C if ld is None else ld
        self.bstride = H * W * self.ld if bstride is None else bstride
        self.off = off

    @staticmethod
    def new(B, H, W, C, dtype, device, zero=False):
        f = torch.zeros if zero else torch.empty
        return Map(f((B, H, W, C), dtype=dtype, device=device), B, H, W, C)

    @staticmethod
    def of(t):
        assert t.dim() == 4 and t.is_contiguous()
        return Map(t, t.shape[0], t.shape[1], t.shape[2], t.shape[3])

    @property
    def dtype(self):
        return self.t.dtype

    def addr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()

    def tensor(self):
        """The [B,H,W,C] tensor (only for plain contiguous maps)."""
        assert self.ld == self.C and self.bstride == self.H * self.W * self.C and self.off == 0
        return self.t.view(self.B, self.H, self.W, self.C)


def _segs(desc, xs, ys, base_x, base_y, isz_x, isz_y):
    desc.nseg = len(xs)
    for i, (x, y) in enumerate(zip(xs, ys)):
        s = desc.seg[i]
        s.H, s.W, s.Ho, s.Wo = x.H, x.W, y.H, y.W
        dx = x.addr() - base_x
        dy = y.addr() - base_y
        assert dx % isz_x == 0 and dy % isz_y == 0
        s.in_off, s.in_bstride = dx // isz_x, x.bstride
        s.out_off, s.out_bstride = dy // isz_y, y.bstride


# ----------------------------------------------------------------------------- batched parameter preparation
PREP_PACK0, PREP_PACK1, PREP_BNFOLD, PREP_DWPACK = 0, 1, 2, 3

# Writers that change parameters through RAW POINTERS (effdet_clip_adamw_step, every replay of a captured train step) never
# move Tensor._version, which is what the inference-side "packed copies are still fresh" test looks at: they bump this
# process-wide generation instead (ClipAdamW.step, graph.GraphedTrainStep.__call__), and the fingerprint includes it.
_PARAM_GENERATION = 0


def bump_param_generation():
    global _PARAM_GENERATION
    _PARAM_GENERATION += 1



class ParamPrep:
    """Record / replay of the per-step parameter repacks (conv weight packs, frozen-BN folds, depthwise packs).

    The first step of a model runs them one by one (~190 four-microsecond launches for D0) and RECORDS each as a job of
    effdet_prepare_params; from the second step on begin_step() replays the whole table in ONE launch at the start of
    the forward pass and pack_weight / bn_fold / dw_pack_weight return views of its output arena.  Parameter storage
    must be stable (it is under in-place optimizers, load_state_dict and DDP); the owner drops the table when the
    module is moved (nn.Module._apply).  A lookup miss falls back to the single launch and re-records."""

    def __init__(self, f32_arith='f32'):
        self.f32_arith = f32_arith          # the owner's arithmetic by name (MODEL_ARITH; see set_model_arith)
        self.jobs, self.outs, self.bn_src = {}, {}, {}
        self.table, self.dirty, self.replay = None, False, False
        self._n0 = 0                        # jobs of the first build since the last reset (bounds the table's growth)
        self._fresh = None                  # (source version sum, stream) the arena was last refreshed for -- inference only
        self.zbuf, self.zpos, self.zreq, self.zsize = None, 0, 0, 0

    # -- zero-initialised scratch of one step (SE pools, bias / BN / fusion-weight accumulators): ONE memset per step
    #    instead of ~60; sized by what the previous step asked for, a fresh buffer every step (gradients may keep views)
    def _begin_zeros(self, device):
        self.zsize = max(self.zsize, self.zreq)
        self.zbuf = torch.zeros(self.zsize, dtype=torch.float32, device=device) if self.zsize else None
        self.zpos = self.zreq = 0

    def zeros(self, n, device):
        n64 = (n + 63) // 64 * 64
        self.zreq += n64
        if self.zbuf is not None and self.zpos + n64 <= self.zbuf.numel() and self.zbuf.device == device:
            v = self.zbuf[self.zpos:self.zpos + n]
            self.zpos += n64
            return v
        return torch.zeros(n, dtype=torch.float32, device=device)

    def reset(self):
        """Forget every recorded job (the next step records afresh)."""
        self.jobs, self.outs, self.bn_src = {}, {}, {}
        self.table, self.dirty, self.replay, self._fresh, self._n0 = None, False, False, None, 0

    def _sources_moved(self):
        """A recorded source tensor no longer lives where the job table says (p.data = ..., a parameter swapped for another
        tensor, replicas of nn.DataParallel): the device-side table would read stale or foreign memory."""
        for job in self.jobs.values():
            for t, a in zip(job[1], job[7]):
                if t is not None and t.data_ptr() != a:
                    return True
        return False

    def begin_step(self, device=None):
        if device is not None:
            self._begin_zeros(device)
        capturing = torch.cuda.is_current_stream_capturing()
        if self.jobs and not capturing:
            # pointer validation (jobs are keyed on addresses): re-record from scratch when a source moved, and keep the table
            # bounded -- lookups that miss every step (tensors re-created per forward) would otherwise grow it without limit
            if self._sources_moved() or (self.table is not None and len(self.jobs) > 2 * max(self._n0, 64)):
                self.reset()
        if self.dirty:
            self._build()
        self.replay = self.table is not None
        if self.replay:
            # Inference: the packed copies in the arena stay valid while no source tensor was written (torch optimizers,
            # load_state_dict, .copy_ all bump Tensor._version; the raw-pointer writers -- ClipAdamW, graph replays -- bump
            # _PARAM_GENERATION): skip the launch (0.05 ms of a 5 ms D0 forward, 0.2 of 12 for D4).
            # Never under graph capture (a captured forward must refresh them on every replay) nor with autograd on.
            fp = None
            if not torch.is_grad_enabled() and not capturing:
                fp = (self._version_sum(), _PARAM_GENERATION, torch.cuda.current_stream().cuda_stream)
                if fp == self._fresh:
                    return
            jobs, bj, bf, nblocks = self.table[:4]
            L.check(L.lib().effdet_prepare_params(L.ptr(jobs), L.ptr(bj), L.ptr(bf), nblocks, L.stream_ptr()), 'effdet_prepare_params')
            self._fresh = fp

    def _version_sum(self):
        return sum(t._version for job in self.jobs.values() for t in job[1] if t is not None)

    def lookup(self, key):
        return self.outs.get(key) if self.replay else None

    def record(self, key, kind, srcs, dims, dtype, shape, eps=0.0, code=None, work=None):
        """work: number of 1-per-thread work items of the job when it is not the number of output elements (the f16x3 pack: rows + row scales)."""
        if key not in self.jobs:
            self.jobs[key] = (kind, srcs, dims, dtype, shape, eps, L.dtype_code(dtype) if code is None else code,
                              tuple(t.data_ptr() if t is not None else 0 for t in srcs), work)
            self.dirty = True

    def _build(self):
        import numpy as np
        keys = list(self.jobs)
        dev = self.jobs[keys[0]][1][0].device
        offs, total = [], 0
        for k in keys:
            kind, srcs, dims, dtype, shape, eps = self.jobs[k][:6]
            n = 1
            for d in shape:
                n *= d
            offs.append((total, n)); total += (n * (2 if dtype == torch.bfloat16 else 4) + 255) // 256 * 256
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        arr = (L.PrepJob * len(keys))()
        block_job, block_first, nb = [], [], 0
        self.outs, self.bn_src = {}, {}      # (drop the pointers of record-time temporaries)
        for i, k in enumerate(keys):
            kind, srcs, dims, dtype, shape, eps, code = self.jobs[k][:7]
            off, n = offs[i]
            out = arena[off:off + n * (2 if dtype == torch.bfloat16 else 4)].view(dtype).view(shape)
            j = arr[i]
            ptrs = [t.data_ptr() if t is not None else None for t in srcs] + [None] * (4 - len(srcs))
            j.a, j.b, j.c, j.d = ptrs[:4]
            j.out, j.kind, j.dtype, j.eps = out.data_ptr(), kind, code, eps
            j.n0, j.n1, j.n2, j.n3, j.n4 = (list(dims) + [0] * 5)[:5]
            blocks = ((self.jobs[k][8] or n) + 255) // 256
            block_first.append(nb); block_job += [i] * blocks; nb += blocks
            self.outs[k] = out
            if kind == PREP_BNFOLD:
                self.bn_src[out[0].data_ptr()] = (srcs[0], srcs[3], eps)
        jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        bj = torch.from_numpy(np.asarray(block_job, dtype=np.int32)).to(dev)
        bf = torch.from_numpy(np.asarray(block_first, dtype=np.int32)).to(dev)
        self.table, self.dirty, self._fresh = (jobs_dev, bj, bf, nb, arena), False, None
        if not self._n0:
            self._n0 = len(keys)


_tls = threading.local()     # .prep: the ParamPrep of the model whose forward / backward runs on this thread


def set_prep(p):
    _tls.prep = p
    if p is not None:
        set_model_arith(p.f32_arith)        # the model's forward arithmetic; its nodes' backward sets theirs (ctx.arith)


def get_prep():
    return getattr(_tls, 'prep', None)


def zeros(shape, device):
    """fp32 zeros, from the running model's per-step zero pool when there is one."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    n = 1
    for d in shape:
        n *= d
    prep = get_prep()
    device = torch.device(device)
    t = prep.zeros(n, device) if prep is not None else torch.zeros(n, dtype=torch.float32, device=device)
    return t.view(shape)


def pack_weight(w_oihw, dtype, mode=0, scale=None, cin_pad=None, x3=False, h3=False):
    """OIHW fp32 -> packed [Cout][taps][Cin_pad] (mode 0) or data-gradient operand [Cin][taps'][Cout] (mode 1).
    x3=True: always the pre-split bf16x3 operand layout (the convs on split-layout activations need it whatever the size).
    h3=True (mode 0, fp32 storage): the f16x3 forward operand (EFFDET_F32_HSPLIT) -- a flat fp32-typed buffer holding Cout rows of
    row-scaled [32 x f16 hi | 32 x f16 lo] groups followed by the Cout row scales 1 / S_n (conv2d(..., hsplit=True) finds them there)."""
    Cout, Cin, KH, KW = w_oihw.shape
    w = w_oihw.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    if cin_pad is None:
        cin_pad = Cin if mode == 0 else Cout
    shape = (Cout, KH * KW, cin_pad) if mode == 0 else (Cin, KH * KW, cin_pad)
    code = L.F32_BF16X3 if x3 else _mma_dtype_code(dtype, KH * KW * cin_pad, shape[0])
    work = None
    if h3:
        assert mode == 0 and dtype == torch.float32 and not x3
        work = Cout * 256                                   # one 256-thread workgroup per output row (it derives the row's scale)
        code, shape = L.F32_HSPLIT, (Cout * KH * KW * cin_pad + Cout,)
    PREP = get_prep()
    if PREP is not None:
        bn = PREP.bn_src.get(scale.data_ptr()) if scale is not None else None
        if scale is None or bn is not None:
            key = ('pack', w.data_ptr(), mode, dtype, cin_pad, scale is not None, code)
            hit = PREP.lookup(key)
            if hit is not None:
                return hit
            PREP.record(key, PREP_PACK1 if mode else PREP_PACK0, (w_oihw, bn[0] if bn else None, bn[1] if bn else None),
                        (Cout, Cin, KH, KW, cin_pad), dtype, shape, bn[2] if bn else 0.0, code, work)
    out = torch.empty(shape, dtype=dtype, device=w.device)
    L.check(L.lib().effdet_pack_conv_weight(L.ptr(w), L.ptr(scale), L.ptr(out), code, mode,
                                            Cout, Cin, KH, KW, cin_pad, L.stream_ptr()), 'effdet_pack_conv_weight')
    return out


def scale_pack_weight(w_oihw, gate, dtype):
    """Per-image 1x1 weights W_b = W * diag(gate_b), packed for conv2d(..., w_image_stride=...): -> (packed [B][Cout][Cin], image
    stride in bytes).  The squeeze-excite gate of an MBConv block applied through the project conv's weights instead of a pass
    over the activations (models/efficientnet.py:86-95)."""
    Cout, Cin = w_oihw.shape[0], w_oihw.shape[1]
    assert w_oihw.shape[2:] == (1, 1) and gate.shape[1] == Cin and gate.dtype == torch.float32 and gate.is_contiguous()
    B = gate.shape[0]
    code = _mma_dtype_code(dtype, Cin, Cout)
    out = torch.empty((B, Cout, Cin), dtype=dtype, device=gate.device)
    L.check(L.lib().effdet_scale_pack_weight(L.ptr(w_oihw.detach()), L.ptr(gate), L.ptr(out), code, B, Cout, Cin, L.stream_ptr()),
            'effdet_scale_pack_weight')
    return out, Cout * Cin * out.element_size()


def conv2d(xs, wp, ys, *, Cin, Cout, KH, KW, stride=1, pad_t=0, pad_l=0, scale=None, shift=None, act=ACT_NONE,
           res=None, res_mode=RES_NONE, rowscale=None, zs=None, out_f32=False, split=False, bc_scale=None, bc_shift=None,
           w_image_stride=0, ysplit=None, hsplit=False, seg_w=None, seg_shift=None):
    """Grouped implicit-GEMM conv: xs/ys (and optional zs/res) are lists of Map, one per pyramid level.
    split=True (EFFDET_F32_SPLIT): xs hold the split layout ([32 x bf16 hi | 32 x bf16 lo] per 32 channels, 4 B per element), wp
    is packed for bf16x3; ys are written split too unless out_f32 (then plain fp32; res, if any, is plain and ADDed).
    ysplit (exact-fp32 and f16x3 convs only): Maps addressed like ys that receive the output a second time in the split layout.
    hsplit=True (EFFDET_F32_HSPLIT, the f16x3 forward arithmetic): xs hold the H-split layout ([32 x f16 hi | 32 x f16 lo * 2^11] per 32
    channels), wp = pack_weight(..., h3=True); ys are written H-split too unless out_f32; no scale / res / rowscale.
    seg_w / seg_shift (lists, one entry per map, None = wp / shift): the maps are INDEPENDENT convs of one geometry with their own packed
    weights / bias rows -- the same layer of the head's two towers in one launch (<= 10 maps).  Same values as separate launches."""
    if isinstance(xs, Map):
        xs, ys = [xs], [ys]
        zs = [zs] if zs is not None else None
        res = [res] if res is not None else None
        ysplit = [ysplit] if ysplit is not None else None
    d = L.ConvDesc()
    x0, y0 = xs[0], ys[0]
    isx, isy = x0.t.element_size(), y0.t.element_size()
    base_x = min(x.addr() for x in xs)
    base_y = min(y.addr() for y in ys)
    d.x, d.w, d.y = base_x, wp.data_ptr(), base_y
    d.z = d.res = None
    if zs is not None:
        # z/res share the output addressing: same relative offsets as ys
        bz = min(z.addr() for z in zs)
        for y, z in zip(ys, zs):
            assert (z.addr() - bz) * isy == (y.addr() - base_y) * z.t.element_size() and z.ld == y.ld and z.bstride == y.bstride
        d.z = bz
    if res is not None:
        br = min(r.addr() for r in res)
        for y, r in zip(ys, res):
            assert (r.addr() - br) * isy == (y.addr() - base_y) * r.t.element_size() and r.ld == y.ld and r.bstride == y.bstride
        d.res = br
    d.y_split = None
    d.range_flag = range_flag(x0.t.device).data_ptr() if (hsplit and not out_f32) else None
    for i in range(len(xs)):
        d.seg_w[i] = seg_w[i].data_ptr() if (seg_w is not None and seg_w[i] is not None) else None
        d.seg_shift[i] = seg_shift[i].data_ptr() if (seg_shift is not None and seg_shift[i] is not None) else None
    if ysplit is not None:
        bs = min(q.addr() for q in ysplit)
        for y, q in zip(ys, ysplit):
            assert (q.addr() - bs) == (y.addr() - base_y) and q.t.element_size() == isy and q.ld == y.ld and q.bstride == y.bstride
        d.y_split = bs
    d.scale, d.shift, d.rowscale = (t.data_ptr() if t is not None else None for t in (scale, shift, rowscale))
    d.bc_scale, d.bc_shift = (t.data_ptr() if t is not None else None for t in (bc_scale, bc_shift))     # [B][Cout] fp32, after rowscale, before res
    assert not (hsplit and (split or scale is not None))
    d.dtype, d.out_f32 = (L.F32_HSPLIT if hsplit else (L.F32_SPLIT if split else _mma_dtype_code(x0.dtype, KH * KW * Cin, Cout))), int(out_f32)
    d.B, d.Cin, d.Cout, d.KH, d.KW = x0.B, Cin, Cout, KH, KW
    d.stride, d.pad_t, d.pad_l = stride, pad_t, pad_l
    d.ldx, d.ldy = x0.ld, y0.ld
    d.act, d.res_mode = act, res_mode
    d.w_image_stride = int(w_image_stride)       # bytes; image b reads its own packed weights (scale_pack_weight)
    _segs(d, xs, ys, base_x, base_y, isx, isy)
    flops = 2.0 * KH * KW * Cin * Cout * sum(y.B * y.H * y.W for y in ys)
    # algorithmic bytes: the input map, the packed weights, every output stream (y, the pre-activation copy, the split copy), the residual
    nbytes = isx * Cin * sum(x.B * x.H * x.W for x in xs) + wp.numel() * wp.element_size() + \
        (4 if out_f32 else isy) * Cout * sum(y.B * y.H * y.W for y in ys) * (1 + (zs is not None) + (ysplit is not None) + (res is not None))
    _timed(_igemm_symbol(x0.dtype, d) if PROFILE is not None else '', flops,
           lambda: L.check(L.lib().effdet_conv2d(C.byref(d), L.stream_ptr()), 'effdet_conv2d'),
           'k%d s%d Cin%d Cout%d M%d' % (KH, stride, Cin, Cout, sum(y.B * y.H * y.W for y in ys)), nbytes=float(nbytes))


def _wgrad_desc(xs, dzs, dw, dbias, Cin, Cout, KH, KW, stride, pad_t, pad_l, want_bias, split, image_splits):
    d = L.WgradDesc()
    x0, z0 = xs[0], dzs[0]
    isz = x0.t.element_size()
    base_x = min(x.addr() for x in xs)
    base_z = min(z.addr() for z in dzs)
    d.x, d.dz = base_x, base_z
    d.dw = dw.data_ptr() if dw is not None else None
    d.dbias = dbias.data_ptr() if dbias is not None else (1 if (want_bias and dw is None) else None)    # dw None: only a request flag
    d.dtype = L.F32_SPLIT if split else _mma_dtype_code(x0.dtype)     # split: both operands in the split layout (see conv2d)
    d.B, d.Cin, d.Cout, d.KH, d.KW = x0.B, Cin, Cout, KH, KW
    d.stride, d.pad_t, d.pad_l = stride, pad_t, pad_l
    d.ldx, d.lddz = x0.ld, z0.ld
    d.image_splits = int(image_splits)        # split-K boundaries on image boundaries: slabs [B*q], slab s = image s // q (one level only)
    _segs(d, xs, dzs, base_x, base_z, isz, z0.t.element_size())
    return d


def conv2d_wgrad_kernel_id(x, dz, *, Cin, Cout, KH, KW, stride=1, pad_t=0, pad_l=0, split=False):
    """Which kernel conv2d_wgrad would launch for one map pair (effdet_conv2d_wgrad_kernel): 0 tiled, 1 thin pointwise, 2 split."""
    d = _wgrad_desc([x], [dz], None, None, Cin, Cout, KH, KW, stride, pad_t, pad_l, True, split, False)
    return int(L.lib().effdet_conv2d_wgrad_kernel(C.byref(d)))


def conv2d_wgrad(xs, dzs, dw=None, dbias=None, *, Cin, Cout, KH, KW, stride=1, pad_t=0, pad_l=0, want_bias=True, split=False,
                 image_splits=False, group=False):
    """Weight gradient -> (slabs [splits][Cout][taps][Cin] fp32, bias partial rows [splits][Cout] fp32 or None): the UNREDUCED
    split-K partials for unpack_wgrad / unpack_wgrad_bn to sum, in slab order, while unpacking (no float atomics anywhere: two
    runs are bitwise equal).  With dw given (packed [Cout][taps][Cin] fp32) the library reduces itself: dw += ..., dbias += ...
    group=True (dw None, <= 5 maps): the maps are INDEPENDENT problems of the same conv geometry (different tensors, different
    weights) sharing one launch -> a list of (slabs_i, parts_i), one per map (effdet_conv2d_wgrad_seg_slabs)."""
    if isinstance(xs, Map):
        xs, dzs = [xs], [dzs]
    x0 = xs[0]
    d = _wgrad_desc(xs, dzs, dw, dbias, Cin, Cout, KH, KW, stride, pad_t, pad_l, want_bias, split, image_splits)
    flops = 2.0 * KH * KW * Cin * Cout * sum(z.B * z.H * z.W for z in dzs)
    splits = int(L.lib().effdet_conv2d_wgrad_splits(C.byref(d)))
    if splits < 1:
        raise RuntimeError('effdet_conv2d_wgrad: unsupported geometry')
    n = Cout * KH * KW * Cin
    ws = torch.empty(splits * (n + Cout), dtype=torch.float32, device=x0.t.device)
    nbytes = ws.numel() * 4
    # (bf16: DMA + LDS-transpose-read kernel, fp32: DMA + direct-operand kernel; levels neither can take use the register-transpose kernel)
    _timed('conv_wgrad_tr_kernel<8>' if x0.dtype == torch.bfloat16 else
           ('conv_wgrad_split_kernel' if split else
            ('conv_wgrad_thin_kernel' if int(L.lib().effdet_conv2d_wgrad_kernel(C.byref(d))) == 1 else
             ('conv_wgrad_f32dma_kernel<4,bf16x3>' if d.dtype == L.F32_BF16X3 else 'conv_wgrad_f32dma_kernel<8>'))), flops,
           lambda: L.check(L.lib().effdet_conv2d_wgrad(C.byref(d), L.ptr(ws), C.c_longlong(nbytes), L.stream_ptr()),
                           'effdet_conv2d_wgrad'),
           'k%d s%d Cin%d Cout%d M%d' % (KH, stride, Cin, Cout, sum(z.B * z.H * z.W for z in dzs)),
           nbytes=float(x0.t.element_size() * Cin * sum(x.B * x.H * x.W for x in xs) +
                        dzs[0].t.element_size() * Cout * sum(z.B * z.H * z.W for z in dzs) + 4.0 * splits * (n + Cout)))
    slabs = ws[:splits * n].view(splits, Cout, KH * KW, Cin)
    parts = ws[splits * n:].view(splits, Cout) if d.dbias else None
    if group:
        assert dw is None and not image_splits
        first, count = (C.c_int * len(xs))(), (C.c_int * len(xs))()
        tot = int(L.lib().effdet_conv2d_wgrad_seg_slabs(C.byref(d), first, count))
        assert tot == splits
        return [(slabs[first[i]:first[i] + count[i]], parts[first[i]:first[i] + count[i]] if parts is not None else None)
                for i in range(len(xs))]
    return slabs, parts


class _TailBatch:
    """Deferred leaf work of one backward node (effdet_backward_tail): inside ``with unpack_batch():`` unpack_wgrad /
    unpack_wgrad_bn / dw_unpack_wgrad_bn / the parameter-gradient half of se_gate_bwd only record a job (outputs are allocated
    right away, inputs stay referenced here) and the whole list goes out as one launch per 24 jobs when the block exits.
    Nothing inside the block may READ these outputs."""

    def __init__(self):
        self.jobs, self.keep, self.bytes = [], [], 0

    def add(self, kind, job, *tensors):
        t = L.TailJob()
        t.kind = kind
        if kind == L.TAIL_UNPACK:
            t.u.conv = job
        elif kind == L.TAIL_SE_PARAMS:
            t.u.se = job
        else:
            t.u.dw = job
        self.jobs.append(t); self.keep.append(tensors)
        # the deferred jobs pin their split-K slab workspaces (tens of MB per head conv): bound the lifetime, not only the launch
        # size -- past the byte budget the jobs recorded so far leave now and their inputs go back to the allocator
        self.bytes += sum(x.numel() * x.element_size() for x in tensors if isinstance(x, torch.Tensor))
        if self.bytes > TAIL_KEEP_BYTES and not getattr(_tls, 'hold_tail', False):
            self.flush()

    def flush(self):
        if self.jobs:
            arr = (L.TailJob * len(self.jobs))(*self.jobs)
            L.check(L.lib().effdet_backward_tail(arr, len(self.jobs), L.stream_ptr()), 'effdet_backward_tail')
        self.jobs, self.keep, self.bytes = [], [], 0


UNPACK_BATCHED = os.environ.get('EFFDET_UNPACK_BATCH', '1') != '0'      # A/B switch: 0 = every tail job is its own launch
TAIL_KEEP_BYTES = int(os.environ.get('EFFDET_TAIL_KEEP_MB', '2048')) << 20  # flush early once the deferred jobs pin this much workspace (256 MB split the
# D0 head's ten unpacks over three launches: 515 -> 704 us of tail time per step; the chains of few jobs do not fill the GPU)


def hold_tail_flush(on):
    """While on, the open tail batch of this thread never flushes early (its jobs were recorded under more than one stream: they may only
    be launched by the owner of the batch, after the streams have joined)."""
    _tls.hold_tail = bool(on)


def _cur_batch():
    """The open tail batch of THIS thread (thread-local like the ParamPrep pointer: replica backwards of a multi-device
    nn.DataParallel run concurrently on per-device autograd threads and must never see each other's batch)."""
    return getattr(_tls, 'unpack_batch', None)


class unpack_batch:
    """Context manager: batch the tail jobs (see _TailBatch) issued inside into one launch at exit (re-entrant: an inner block
    joins the outer one)."""

    def __enter__(self):
        self.owner = _cur_batch() is None and UNPACK_BATCHED
        if self.owner:
            _tls.unpack_batch = _TailBatch()
        return self

    def __exit__(self, et, ev, tb):
        if self.owner:
            b, _tls.unpack_batch = _tls.unpack_batch, None
            if et is None:
                b.flush()
        return False


def _unpack_job(g, dw, Cout, Cin, KH, KW, cin_pad, nslabs, slab_scale, slab_cscale=None, **ptrs):
    j = L.UnpackJob()
    j.g, j.dw_oihw, j.slab_scale = g.data_ptr(), dw.data_ptr(), (slab_scale.data_ptr() if slab_scale is not None else None)
    for k, t in ptrs.items():
        setattr(j, k, t.data_ptr() if t is not None else None)
    j.Cout, j.Cin, j.KH, j.KW, j.Cin_pad, j.nslabs = Cout, Cin, KH, KW, (Cin if cin_pad is None else cin_pad), nslabs
    j.slabs_per_scale = nslabs // slab_scale.numel() if slab_scale is not None else 1
    if slab_cscale is not None:      # [images][Cin_pad] per-input-channel factors (the SE gate): slab s belongs to image s // slabs_per_scale
        assert slab_cscale.dtype == torch.float32 and slab_cscale.is_contiguous() and slab_cscale.shape[1] == j.Cin_pad
        assert nslabs % slab_cscale.shape[0] == 0 and (slab_scale is None or slab_scale.numel() == slab_cscale.shape[0])
        j.slab_cscale = slab_cscale.data_ptr()
        j.slabs_per_scale = nslabs // slab_cscale.shape[0]
    return j


def _unpack_submit(job, *tensors):
    batch = _cur_batch()
    if batch is not None:
        batch.add(L.TAIL_UNPACK, job, *tensors)
    else:
        L.check(L.lib().effdet_unpack_conv_wgrad_batch(C.byref(job), 1, L.stream_ptr()), 'effdet_unpack_conv_wgrad_batch')


def unpack_wgrad(g, dw_oihw, scale=None, w_oihw=None, wsum=None, accumulate=False, cin_pad=None, dbias_part=None, slab_scale=None):
    """g: packed gradient [Cout][taps][Cin_pad] or unreduced slabs [splits][Cout][taps][Cin_pad] (summed here).
    dbias_part ([splits][Cout], from conv2d_wgrad): -> the bias gradient [Cout] (summed in slab order), else None.
    Inside ``with unpack_batch():`` the launch is deferred to the end of the block."""
    Cout, Cin, KH, KW = dw_oihw.shape
    nslabs = g.shape[0] if g.dim() == 4 else 1
    db = torch.empty(Cout, dtype=torch.float32, device=dw_oihw.device) if dbias_part is not None else None
    assert dbias_part is None or dbias_part.shape == (nslabs, Cout)
    assert slab_scale is None or nslabs % slab_scale.numel() == 0
    job = _unpack_job(g, dw_oihw, Cout, Cin, KH, KW, cin_pad, nslabs, slab_scale, scale=scale, w_oihw=w_oihw, wsum=wsum,
                      dsum_part=dbias_part, dbias_out=db)
    job.accumulate = int(accumulate)
    _unpack_submit(job, g, dw_oihw, scale, w_oihw, wsum, dbias_part, db, slab_scale)
    return db


def unpack_wgrad_bn(g, w_oihw, scale, dsum_part, mean, invstd, cin_pad=None, slab_scale=None, slab_cscale=None):
    """unpack_wgrad + bn_param_grad in one launch -> (dw, dgamma, dbeta); dsum_part: the [splits][Cout] rows of conv2d_wgrad.
    slab_scale [B] (per-image slabs of conv2d_wgrad(image_splits=True)): slab s is multiplied by slab_scale[s // (splits / B)].
    slab_cscale [B][Cin]: and, per input channel, by the image's squeeze-excite gate (forward on per-image weights)."""
    Cout, Cin, KH, KW = w_oihw.shape
    nslabs = g.shape[0] if g.dim() == 4 else 1
    if dsum_part.dim() == 1:
        dsum_part = dsum_part.view(1, Cout)
    assert dsum_part.shape == (nslabs, Cout) and dsum_part.is_contiguous()
    assert slab_scale is None or nslabs % slab_scale.numel() == 0
    dw = torch.empty_like(w_oihw)
    dgb = torch.empty((2, Cout), dtype=torch.float32, device=dw.device)
    wd = w_oihw.detach()
    job = _unpack_job(g, dw, Cout, Cin, KH, KW, cin_pad, nslabs, slab_scale, slab_cscale, scale=scale, w_oihw=wd, dsum_part=dsum_part, mean=mean,
                      invstd=invstd, dgamma=dgb[0], dbeta=dgb[1])
    _unpack_submit(job, g, dw, scale, wd, dsum_part, mean, invstd, dgb, slab_scale, slab_cscale)
    return dw, dgb[0], dgb[1]


def wgrad_split_supported(B, sizes, Cin, Cout, lddz, KH=3, KW=3, pad=1):
    """Can the split-operand weight-gradient kernel take these pyramid levels (same-padded stride-1 conv, flat level-major
    buffers)?  Host-side query only (no device work)."""
    d = L.WgradDesc()
    d.x, d.dz, d.dw, d.dbias = 128, 128, None, None
    d.dtype, d.B, d.Cin, d.Cout, d.KH, d.KW = L.F32_SPLIT, B, Cin, Cout, KH, KW
    d.stride, d.pad_t, d.pad_l, d.ldx, d.lddz = 1, pad, pad, Cin, lddz
    d.nseg = len(sizes)
    ox = oz = 0
    for i, (h, w) in enumerate(sizes):
        s = d.seg[i]
        s.H, s.W, s.Ho, s.Wo = h, w, h, w
        s.in_off, s.in_bstride, s.out_off, s.out_bstride = ox, h * w * Cin, oz, h * w * lddz
        ox += B * h * w * Cin; oz += B * h * w * lddz
    return int(L.lib().effdet_conv2d_wgrad_splits(C.byref(d))) >= 1


def to_split(t):
    """plain fp32 tensor (rows of whole 32-channel groups) -> the same shape in the split layout (out of place)."""
    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 32 == 0
    out = torch.empty_like(t)
    L.check(L.lib().effdet_to_split(L.ptr(t), L.ptr(out), C.c_longlong(t.numel()), L.stream_ptr()), 'effdet_to_split')
    return out


def nhwc_to_nchw(m):
    out = torch.empty((m.B, m.C, m.H, m.W), dtype=torch.float32, device=m.t.device)
    L.check(L.lib().effdet_nhwc_to_nchw_f32(L.ptr(m.tensor()), L.ptr(out), L.dtype_code(m.dtype), m.B, m.H, m.W, m.C,
                                            L.stream_ptr()), 'effdet_nhwc_to_nchw_f32')
    return out


def nchw_to_nhwc(x, dtype, cpad=None):
    B, Cc, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    cpad = Cc if cpad is None else cpad
    m = Map.new(B, H, W, cpad, dtype, x.device)
    L.check(L.lib().effdet_nchw_f32_to_nhwc(L.ptr(x), L.ptr(m.t), L.dtype_code(dtype), B, H, W, Cc, cpad, L.stream_ptr()),
            'effdet_nchw_f32_to_nhwc')
    return m


# ----------------------------------------------------------------------------- frozen BN
def bn_fold(gamma, beta, mean, var, eps=1e-3):
    C_ = gamma.numel()
    PREP = get_prep()
    if PREP is not None:
        key = ('bn', gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), eps)
        hit = PREP.lookup(key)
        if hit is not None:
            return hit[0], hit[1], hit[2]
        PREP.record(key, PREP_BNFOLD, (gamma, beta, mean, var), (C_,), torch.float32, (3, C_), eps)
    out = torch.empty((3, C_), dtype=torch.float32, device=gamma.device)     # scale | shift | invstd
    L.check(L.lib().effdet_bn_fold(L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(var), C.c_float(eps),
                                   L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), C_, L.stream_ptr()), 'effdet_bn_fold')
    if PREP is not None:
        PREP.bn_src[out[0].data_ptr()] = (gamma, var, eps)
    return out[0], out[1], out[2]


def bn_param_grad(wsum, dsum, mean, invstd):
    C_ = wsum.numel()
    dg = torch.empty(C_, dtype=torch.float32, device=wsum.device); db = torch.empty_like(dg)
    L.check(L.lib().effdet_bn_param_grad(L.ptr(wsum), L.ptr(dsum), L.ptr(mean), L.ptr(invstd), L.ptr(dg), L.ptr(db), C_,
                                         L.stream_ptr()), 'effdet_bn_param_grad')
    return dg, db


# ----------------------------------------------------------------------------- depthwise
def dw_pack_weight(w_c1kk):
    Cc, _, k, _ = w_c1kk.shape
    PREP = get_prep()
    if PREP is not None:
        key = ('dw', w_c1kk.data_ptr())
        hit = PREP.lookup(key)
        if hit is not None:
            return hit
        PREP.record(key, PREP_DWPACK, (w_c1kk,), (Cc, 0, k * k), torch.float32, (k * k, Cc))
    out = torch.empty((k * k, Cc), dtype=torch.float32, device=w_c1kk.device)
    L.check(L.lib().effdet_dw_pack_weight(L.ptr(w_c1kk.detach()), L.ptr(out), Cc, k, L.stream_ptr()), 'effdet_dw_pack_weight')
    return out


def dw_unpack_wgrad(g_kkc, scale, w_c1kk, wsum=None):
    Cc, _, k, _ = w_c1kk.shape
    dw = torch.empty_like(w_c1kk)
    L.check(L.lib().effdet_dw_unpack_wgrad(L.ptr(g_kkc), L.ptr(scale), L.ptr(w_c1kk.detach()), L.ptr(dw), L.ptr(wsum), Cc, k,
                                           L.stream_ptr()), 'effdet_dw_unpack_wgrad')
    return dw


def dw_unpack_wgrad_bn(g_kkc, scale, w_c1kk, dsum, mean, invstd):
    """dw_unpack_wgrad + bn_param_grad in one launch -> (dw, dgamma, dbeta).  Deferred inside ``with unpack_batch():``."""
    Cc, _, k, _ = w_c1kk.shape
    dw = torch.empty_like(w_c1kk)
    dgb = torch.empty((2, Cc), dtype=torch.float32, device=dw.device)
    wd = w_c1kk.detach()
    batch = _cur_batch()
    if batch is not None:
        j = L.DwUnpackJob()
        j.g_kkc, j.scale, j.w_c1kk, j.dw_c1kk, j.dsum, j.mean, j.invstd = (t.data_ptr() for t in (g_kkc, scale, wd, dw, dsum, mean, invstd))
        j.dgamma, j.dbeta, j.C, j.kk = dgb[0].data_ptr(), dgb[1].data_ptr(), Cc, k * k
        batch.add(L.TAIL_DW_UNPACK, j, g_kkc, scale, wd, dw, dsum, mean, invstd, dgb)
    else:
        L.check(L.lib().effdet_dw_unpack_wgrad_bn(L.ptr(g_kkc), L.ptr(scale), L.ptr(wd), L.ptr(dw), L.ptr(dsum), L.ptr(mean),
                                                  L.ptr(invstd), L.ptr(dgb[0]), L.ptr(dgb[1]), Cc, k, L.stream_ptr()),
                'effdet_dw_unpack_wgrad_bn')
    return dw, dgb[0], dgb[1]


def dwconv_fwd(x, w_kkc, scale, shift, k, stride, pad_t, pad_l, Ho, Wo, save_z=False, pool=False, save_y=True, in_act=ACT_NONE):
    """-> (y, z, pool_part); save_y=False (needs save_z) stores the pre-activation only: consumers recompute Swish (act=ACT_SWISH).
    pool=True: pool_part [B][G][C] fp32 = per-(image, tile group) partial sums of the Swish output for se_gate_fwd.
    in_act=ACT_SWISH: x is the pre-activation of the expand conv (z-only storage); Swish is applied to the staged tiles."""
    assert save_y or save_z
    if pool:
        G = int(L.lib().effdet_dwconv_fwd_pool_groups(L.dtype_code(x.dtype), x.B, x.C, stride, Ho, Wo))
        if G < 1:
            raise RuntimeError('effdet_dwconv_fwd_pool_groups: unsupported geometry')
        pool = torch.empty((x.B, G, x.C), dtype=torch.float32, device=x.t.device)
    else:
        pool = None
    y = Map.new(x.B, Ho, Wo, x.C, x.dtype, x.t.device) if save_y else None
    z = Map.new(x.B, Ho, Wo, x.C, x.dtype, x.t.device) if save_z else None
    nbytes = x.t.element_size() * x.B * x.C * (x.H * x.W + Ho * Wo * (int(save_y) + int(save_z)))
    _timed('dw_fwd_lds_kernel', nbytes, lambda: L.check(L.lib().effdet_dwconv_fwd(
        L.ptr(x.tensor()), L.ptr(w_kkc), L.ptr(scale), L.ptr(shift), L.ptr(y.t if y else None), L.ptr(z.t if z else None), L.ptr(pool),
        L.dtype_code(x.dtype), x.B, x.H, x.W, x.C, k, stride, pad_t, pad_l, Ho, Wo, in_act, L.stream_ptr()), 'effdet_dwconv_fwd'),
        'BYTES k%d s%d C%d %dx%d' % (k, stride, x.C, x.H, x.W))
    return y, z, pool


def expand_dw_fwd(x, w_expand, scale0, shift0, w_kkc, scale1, shift1, k, stride, pad_t, pad_l, Ho, Wo):
    """Inference: expand 1x1 + BN + Swish -> depthwise k x k + BN + Swish in ONE kernel (the expanded map stays in LDS).
    x: fp32 Map [B,H,W,Cin], Cin in {16, 24, 32, 40}; w_expand: the OIHW 1x1 weight.  -> (y Map [B,Ho,Wo,Cexp], pool_part [B][G][Cexp])."""
    Cexp, Cin = w_expand.shape[0], w_expand.shape[1]
    assert x.dtype == torch.float32 and x.C == Cin
    G = int(L.lib().effdet_mbconv_expand_dw_pool_groups(x.B, Cexp, stride, Ho, Wo))
    if G < 1:
        raise RuntimeError('effdet_mbconv_expand_dw_pool_groups: unsupported geometry')
    pool = torch.empty((x.B, G, Cexp), dtype=torch.float32, device=x.t.device)
    y = Map.new(x.B, Ho, Wo, Cexp, torch.float32, x.t.device)
    nbytes = 4 * x.B * (x.H * x.W * Cin * ((Cexp + 31) // 32) + Ho * Wo * Cexp)
    _timed('dw_fwd_fused_kernel', nbytes, lambda: L.check(L.lib().effdet_mbconv_expand_dw_fwd(
        L.ptr(x.tensor()), L.ptr(w_expand.detach()), L.ptr(scale0), L.ptr(shift0), L.ptr(w_kkc), L.ptr(scale1), L.ptr(shift1), L.ptr(y.t), L.ptr(pool),
        x.B, x.H, x.W, Cin, Cexp, k, stride, pad_t, pad_l, Ho, Wo, L.stream_ptr()), 'effdet_mbconv_expand_dw_fwd'),
        'BYTES fused k%d s%d Cin%d C%d %dx%d' % (k, stride, Cin, Cexp, x.H, x.W))
    return y, pool


def dwconv_dgrad(dz, w_kkc, scale, zprev, H, W, k, stride, pad_t, pad_l):
    dx = Map.new(dz.B, H, W, dz.C, dz.dtype, dz.t.device)
    nbytes = dz.t.element_size() * dz.B * dz.C * (dz.H * dz.W + H * W * (2 if zprev else 1))
    _timed('dw_dgrad_lds_kernel', nbytes, lambda: L.check(L.lib().effdet_dwconv_dgrad(
        L.ptr(dz.tensor()), L.ptr(w_kkc), L.ptr(scale), L.ptr(zprev.tensor() if zprev else None), L.ptr(dx.t),
        L.dtype_code(dz.dtype), dz.B, H, W, dz.C, k, stride, pad_t, pad_l, dz.H, dz.W, L.stream_ptr()), 'effdet_dwconv_dgrad'),
        'BYTES k%d s%d C%d %dx%d' % (k, stride, dz.C, H, W))
    return dx


def dwconv_wgrad(x, dz, k, stride, pad_t, pad_l, in_act=ACT_NONE):
    g = torch.empty((k * k + 1, x.C), dtype=torch.float32, device=x.t.device)       # taps | dsum (overwritten by the slab reduction)
    geo = (L.dtype_code(x.dtype), x.B, x.H, x.W, x.C, k, stride, pad_t, pad_l, dz.H, dz.W)
    nbytes = int(L.lib().effdet_dwconv_wgrad_workspace_bytes(*geo))
    if nbytes < 0:
        raise RuntimeError('effdet_dwconv_wgrad: unsupported geometry')
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.t.device)
    traffic = x.t.element_size() * x.B * x.C * (x.H * x.W + dz.H * dz.W)
    _timed('dw_wgrad_lds_kernel', traffic, lambda: L.check(L.lib().effdet_dwconv_wgrad(
        L.ptr(x.tensor()), L.ptr(dz.tensor()), L.ptr(g), L.ptr(g[k * k]), L.ptr(ws), C.c_longlong(nbytes), *geo, in_act, L.stream_ptr()),
        'effdet_dwconv_wgrad'), 'BYTES k%d s%d C%d %dx%d' % (k, stride, x.C, x.H, x.W))
    return g[:k * k], g[k * k]


def dwconv_bwd(dz, w_kkc, scale, zprev, k, stride, pad_t, pad_l):
    """Data + weight gradient of the depthwise conv in one pass (the conv's input was stored as the pre-activation ``zprev`` only).
    -> (dx, g [k*k][C], dsum [C]), or None when the fused kernel does not serve the geometry (callers run dwconv_wgrad + dwconv_dgrad)."""
    H, W = zprev.H, zprev.W
    geo = (L.dtype_code(dz.dtype), dz.B, H, W, dz.C, k, stride, pad_t, pad_l, dz.H, dz.W)
    nbytes = int(L.lib().effdet_dwconv_bwd_workspace_bytes(*geo))
    if nbytes < 0:
        raise RuntimeError('effdet_dwconv_bwd: invalid geometry')
    if nbytes == 0 or zprev.ld != zprev.C or zprev.off != 0 or dz.ld != dz.C or dz.off != 0:
        return None
    dx = Map.new(dz.B, H, W, dz.C, dz.dtype, dz.t.device)
    g = torch.empty((k * k + 1, dz.C), dtype=torch.float32, device=dz.t.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dz.t.device)
    traffic = dz.t.element_size() * dz.B * dz.C * (dz.H * dz.W + 2 * H * W)
    _timed('dw_bwd_lds_kernel', traffic, lambda: L.check(L.lib().effdet_dwconv_bwd(
        L.ptr(dz.tensor()), L.ptr(w_kkc), L.ptr(scale), L.ptr(zprev.tensor()), L.ptr(dx.t), L.ptr(g), L.ptr(g[k * k]), L.ptr(ws),
        C.c_longlong(nbytes), *geo, L.stream_ptr()), 'effdet_dwconv_bwd'), 'BYTES k%d s%d C%d %dx%d' % (k, stride, dz.C, H, W))
    return dx, g[:k * k], g[k * k]


def pw_bwd(dz, x, w_expand, scale, res=None):
    """Data + weight gradient of the MBConv expand conv in one pass over the expanded gradient ``dz`` (effdet_pw_bwd).
    -> (dx Map, slabs [S][Cexp][1][Cin], dsum parts [S][Cexp]) for unpack_wgrad_bn, or None when the fused kernel does not serve the geometry."""
    Cexp, Cin = w_expand.shape[0], w_expand.shape[1]
    M = dz.B * dz.H * dz.W
    dense = all(m is None or (m.ld == m.C and m.off == 0 and m.bstride == m.H * m.W * m.C and m.dtype == torch.float32) for m in (dz, x, res))
    S = int(L.lib().effdet_pw_bwd_slabs(C.c_longlong(M), Cin, Cexp)) if dense and dz.C == Cexp and x.C == Cin else 0
    if S < 1:
        return None
    dx = Map.new(x.B, x.H, x.W, Cin, torch.float32, x.t.device)
    ws = torch.empty(S * (Cexp * Cin + Cexp), dtype=torch.float32, device=x.t.device)
    slabs, parts = ws[:S * Cexp * Cin].view(S, Cexp, 1, Cin), ws[S * Cexp * Cin:].view(S, Cexp)
    _timed('conv_pw_bwd_kernel', 4.0 * Cin * Cexp * M, lambda: L.check(L.lib().effdet_pw_bwd(
        L.ptr(dz.tensor()), L.ptr(x.tensor()), L.ptr(w_expand.detach()), L.ptr(scale), L.ptr(res.tensor() if res is not None else None), L.ptr(dx.t),
        L.ptr(slabs), L.ptr(parts), C.c_longlong(M), Cin, Cexp, L.stream_ptr()), 'effdet_pw_bwd'), 'Cin%d Cexp%d M%d' % (Cin, Cexp, M),
        nbytes=4.0 * M * (Cexp + (3 if res is not None else 2) * Cin))
    return dx, slabs, parts


def pw_dgrad_se(dy, w_project, scale, rowscale, gate, dpool, zd):
    """Project-conv data gradient with the squeeze-excite backward + Swish' epilogue (effdet_pw_dgrad_se) -> dz_d Map, or None when the
    kernel does not serve the geometry (callers use conv2d(..., bc_scale=gate, bc_shift=dpool, res=zd, res_mode=RES_SWISH_GRAD))."""
    Cout, Cexp = w_project.shape[0], w_project.shape[1]
    M = dy.B * dy.H * dy.W
    dense = all(m.ld == m.C and m.off == 0 and m.bstride == m.H * m.W * m.C and m.dtype == torch.float32 for m in (dy, zd))
    if not dense or dy.C != Cout or zd.C != Cexp or not int(L.lib().effdet_pw_dgrad_se_supported(C.c_longlong(M), Cout, Cexp)):
        return None
    dz = Map.new(zd.B, zd.H, zd.W, Cexp, torch.float32, zd.t.device)
    _timed('conv_pw_dgrad_se_kernel', 2.0 * Cout * Cexp * M, lambda: L.check(L.lib().effdet_pw_dgrad_se(
        L.ptr(dy.tensor()), L.ptr(w_project.detach()), L.ptr(scale), L.ptr(rowscale), L.ptr(gate), L.ptr(dpool), L.ptr(zd.tensor()), L.ptr(dz.t),
        C.c_longlong(M), dy.H * dy.W, dy.B, Cout, Cexp, L.stream_ptr()), 'effdet_pw_dgrad_se'), 'Cout%d Cexp%d M%d' % (Cout, Cexp, M),
        nbytes=4.0 * M * (Cout + 2 * Cexp))
    return dz


# ----------------------------------------------------------------------------- squeeze-excite
def se_gate_fwd(pool_part, w1, b1, w2, b2, inv_hw, save_mid=False):
    """pool_part: [B][G][C] partial sums of dwconv_fwd (or a plain [B][C] pool) -> (gate, mid, pool [B][C] = the pooled SUM)."""
    if pool_part.dim() == 2:
        pool_part = pool_part.unsqueeze(1)
    B, G, Cc = pool_part.shape
    assert pool_part.is_contiguous()
    Cse = w1.shape[0]
    gate = torch.empty((B, Cc), dtype=torch.float32, device=pool_part.device)
    pool = torch.empty((B, Cc), dtype=torch.float32, device=pool_part.device)
    mid = torch.empty((B, Cse), dtype=torch.float32, device=pool_part.device) if save_mid else None
    ws = torch.empty((B, Cse), dtype=torch.float32, device=pool_part.device)
    L.check(L.lib().effdet_se_gate_fwd_split(L.ptr(pool_part), G, L.ptr(pool), L.ptr(w1.detach()), L.ptr(b1.detach()), L.ptr(w2.detach()),
                                             L.ptr(b2.detach()), L.ptr(gate), L.ptr(mid), L.ptr(ws), B, Cc, Cse, C.c_float(inv_hw),
                                             L.stream_ptr()), 'effdet_se_gate_fwd_split')
    return gate, mid, pool


def channel_scale(x, gate, act=ACT_NONE):
    """act(x) * gate[b][c]; act=ACT_SWISH when x is the depthwise pre-activation (z-only storage)."""
    y = Map.new(x.B, x.H, x.W, x.C, x.dtype, x.t.device)
    L.check(L.lib().effdet_channel_scale(L.ptr(x.tensor()), L.ptr(gate), L.ptr(y.t), act, L.dtype_code(x.dtype), x.B,
                                         C.c_longlong(x.H * x.W), x.C, L.stream_ptr()), 'effdet_channel_scale')
    return y


def se_dgate(dy, x, act=ACT_NONE):
    """-> dgate_part [B][slabs][C]: per-pixel-slab partial sums (added in slab order by se_gate_bwd)."""
    dg = torch.empty((x.B, int(L.lib().effdet_se_dgate_slabs(C.c_longlong(x.H * x.W))), x.C), dtype=torch.float32, device=x.t.device)
    L.check(L.lib().effdet_se_dgate(L.ptr(dy.tensor()), L.ptr(x.tensor()), L.ptr(dg), act, L.dtype_code(x.dtype), x.B,
                                    C.c_longlong(x.H * x.W), x.C, L.stream_ptr()), 'effdet_se_dgate')
    return dg


def se_dgate_from_wgrad(slabs, w_oc, bn_scale, rowscale, B):
    """(d loss / d gate * gate) [B][Cexp] from the project conv's per-image weight-gradient slabs [B*q][Cout][1][Cexp] (see the header)."""
    S, Co, _, Ce = slabs.shape
    assert S % B == 0
    out = torch.empty((B, Ce), dtype=torch.float32, device=slabs.device)
    L.check(L.lib().effdet_se_dgate_from_wgrad(L.ptr(slabs), L.ptr(w_oc.detach()), L.ptr(bn_scale), L.ptr(rowscale), L.ptr(out), B, S // B,
                                               Co, Ce, L.stream_ptr()), 'effdet_se_dgate_from_wgrad')
    return out


def se_gate_bwd(dgate, gate, mid, pool, w1, b1, w2, inv_hw, times_gate=False):
    """dgate: [B][slabs][C] partial rows of se_dgate (or a plain [B][C] gradient).
    -> (dpool, dw1, db1, dw2, db2); the parameter grads are fresh tensors (overwritten, not accumulated)."""
    if dgate.dim() == 2:
        dgate = dgate.unsqueeze(1)
    assert dgate.is_contiguous()
    B, Cc = pool.shape
    Cse = w1.shape[0]
    dev = pool.device
    dpool = torch.empty_like(pool)
    out = torch.empty(2 * Cc * Cse + Cc + Cse + B * (Cc + 2 * Cse), dtype=torch.float32, device=dev)
    dw1 = out[:Cse * Cc].view(Cse, Cc); o = Cse * Cc
    dw2 = out[o:o + Cc * Cse].view(Cc, Cse); o += Cc * Cse
    db1 = out[o:o + Cse]; o += Cse
    db2 = out[o:o + Cc]; o += Cc
    ws = out[o:]
    assert ws.numel() >= L.lib().effdet_se_gate_bwd_workspace_floats(B, Cc, Cse)
    batch = _cur_batch()
    defer = batch is not None                  # the parameter gradients (phase B) join the node's tail launch
    w1d, b1d, w2d = w1.detach(), b1.detach(), w2.detach()
    L.check(L.lib().effdet_se_gate_bwd(L.ptr(dgate), dgate.shape[1], int(times_gate), L.ptr(gate), L.ptr(mid), L.ptr(pool), L.ptr(w1d), L.ptr(b1d),
                                       L.ptr(w2d), L.ptr(dpool), L.ptr(None if defer else dw1), L.ptr(None if defer else db1),
                                       L.ptr(None if defer else dw2), L.ptr(None if defer else db2), L.ptr(ws),
                                       B, Cc, Cse, C.c_float(inv_hw), L.stream_ptr()), 'effdet_se_gate_bwd')
    if defer:
        j = L.SeParamJob()
        j.du, j.dmid, j.sw = ws.data_ptr(), ws.data_ptr() + 4 * B * Cc, ws.data_ptr() + 4 * B * (Cc + Cse)
        j.pool, j.dw1, j.db1, j.dw2, j.db2 = pool.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr()
        j.B, j.C, j.Cse, j.inv_hw = B, Cc, Cse, inv_hw
        batch.add(L.TAIL_SE_PARAMS, j, out, pool)
    return dpool, dw1, db1, dw2, db2


def se_bwd_apply(dy, gate, dpool, z):
    out = Map.new(z.B, z.H, z.W, z.C, z.dtype, z.t.device)
    L.check(L.lib().effdet_se_bwd_apply(L.ptr(dy.tensor()), L.ptr(gate), L.ptr(dpool), L.ptr(z.tensor()), L.ptr(out.t),
                                        L.dtype_code(z.dtype), z.B, C.c_longlong(z.H * z.W), z.C, L.stream_ptr()), 'effdet_se_bwd_apply')
    return out


def act_bwd(dy, aux, act, rowscale=None, out=None):
    out = out or Map.new(dy.B, dy.H, dy.W, dy.C, dy.dtype, dy.t.device)
    L.check(L.lib().effdet_act_bwd(L.ptr(dy.tensor()), L.ptr(aux.tensor() if aux is not None else None), L.ptr(rowscale),
                                   L.ptr(out.t), L.dtype_code(dy.dtype), act, dy.B, C.c_longlong(dy.H * dy.W * dy.C),
                                   L.stream_ptr()), 'effdet_act_bwd')
    return out


def add_inplace(y, x):
    """y += x for two same-shaped contiguous tensors / Maps."""
    ty = y.tensor() if isinstance(y, Map) else y
    tx = x.tensor() if isinstance(x, Map) else x
    assert ty.numel() == tx.numel() and ty.dtype == tx.dtype
    L.check(L.lib().effdet_add_inplace(L.ptr(ty), L.ptr(tx), L.dtype_code(ty.dtype), C.c_longlong(ty.numel()), L.stream_ptr()),
            'effdet_add_inplace')


def colsum(t2d, out):
    rows, Cc = t2d.shape
    L.check(L.lib().effdet_colsum(L.ptr(t2d), L.ptr(out), L.dtype_code(t2d.dtype), C.c_longlong(rows), Cc, Cc, L.stream_ptr()),
            'effdet_colsum')


# ----------------------------------------------------------------------------- BiFPN fusion
def bifpn_fuse_fwd(a, b, c, wraw, col, mode, plain=True, hsplit=False):
    """-> the fused map (plain Map), or with hsplit=True the pair (plain Map or None, H-split Map): the operand of an f16x3 conv."""
    out = Map.new(a.B, a.H, a.W, a.C, a.dtype, a.t.device) if plain else None
    outh = Map.new(a.B, a.H, a.W, a.C, a.dtype, a.t.device) if hsplit else None
    wr, wc = wraw.shape
    L.check(L.lib().effdet_bifpn_fuse_fwd2(L.ptr(a.tensor()), L.ptr(b.tensor()), L.ptr(c.tensor() if c is not None else None),
                                           L.ptr(out.t if plain else None), L.ptr(outh.t if hsplit else None), L.ptr(wraw.detach()), wr, wc, col, mode,
                                           L.dtype_code(a.dtype), a.B, a.H, a.W, a.C, L.ptr(range_flag(a.t.device) if hsplit else None),
                                           L.stream_ptr()), 'effdet_bifpn_fuse_fwd2')
    return (out, outh) if hsplit else out


FUSE_COL_FLOATS = 4 + 3 * 2048     # EFFDET_FUSE_COL_FLOATS (include/effdet_hip.h): per weight column, [count | per-workgroup partial triples]


def fuse_dn_floats(wcols):
    return wcols * FUSE_COL_FLOATS


def bifpn_fuse_bwd(dout, a, b, c, da, db, dc, da_acc, db_acc, dc_acc, wraw, dn, col, mode):
    wr, wc = wraw.shape
    assert dn.numel() == fuse_dn_floats(wc)
    L.check(L.lib().effdet_bifpn_fuse_bwd(L.ptr(dout.tensor()), L.ptr(a.tensor()), L.ptr(b.tensor()),
                                          L.ptr(c.tensor() if c is not None else None), L.ptr(da.tensor()), L.ptr(db.tensor()),
                                          L.ptr(dc.tensor() if dc is not None else None), int(da_acc), int(db_acc), int(dc_acc),
                                          L.ptr(wraw.detach()), L.ptr(dn), wr, wc, col, mode, L.dtype_code(a.dtype),
                                          a.B, a.H, a.W, a.C, L.stream_ptr()), 'effdet_bifpn_fuse_bwd')


def bifpn_weight_bwd(wraw, dn, dwraw):
    wr, wc = wraw.shape
    L.check(L.lib().effdet_bifpn_weight_bwd(L.ptr(wraw.detach()), L.ptr(dn), L.ptr(dwraw), wr, wc, L.stream_ptr()),
            'effdet_bifpn_weight_bwd')


# ----------------------------------------------------------------------------- anchors / decode / NMS
def num_anchors(H, W):
    return int(L.lib().effdet_num_anchors(H, W))


def anchors(H, W, device):
    A = num_anchors(H, W)
    out = torch.empty((1, A, 4), dtype=torch.float32, device=device)
    L.check(L.lib().effdet_anchors(L.ptr(out), H, W, L.stream_ptr()), 'effdet_anchors')
    return out


def decode_score(anc, reg, cls, img_h, img_w):
    B, A, nc = cls.shape
    boxes = torch.empty((B, A, 4), dtype=torch.float32, device=cls.device)
    score = torch.empty((B, A), dtype=torch.float32, device=cls.device)
    label = torch.empty((B, A), dtype=torch.int32, device=cls.device)
    # algorithmic bytes: probabilities + box deltas read once, boxes / score / label written once (anchors: one [A,4] table for the batch)
    nbytes = 4 * (B * A * (nc + 4) + A * 4 + B * A * 6)
    _timed('decode_score_kernel', nbytes, lambda: L.check(L.lib().effdet_decode_score(
        L.ptr(anc), L.ptr(reg), L.ptr(cls), L.ptr(boxes), L.ptr(score), L.ptr(label), B, C.c_longlong(A), nc, C.c_float(img_w), C.c_float(img_h),
        L.stream_ptr()), 'effdet_decode_score'), 'BYTES B%d A%d nc%d' % (B, A, nc))
    return boxes, score, label


def nms(boxes, score, threshold, iou_threshold):
    """-> (idx [B][A] int32 kept anchor indices in order, count [B] int32); all on device, no sync."""
    B, A = score.shape
    nbytes = int(L.lib().effdet_nms_workspace_bytes(B, C.c_longlong(A)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=score.device)
    idx = torch.empty((B, A), dtype=torch.int32, device=score.device)
    count = torch.empty((B,), dtype=torch.int32, device=score.device)
    # algorithmic bytes of the whole NMS call (sort + greedy rounds = ~90 launches): every candidate's box + score read once, its index
    # written once -- the O(K^2) suppression work is latency / ALU, not bytes, so the GB/s of this entry says how far from a streaming
    # pass the greedy algorithm is, not how well a kernel streams
    _timed('nms (radix sort + cross / matrix / resolve rounds)', 4 * B * A * 6, lambda: L.check(L.lib().effdet_nms(
        L.ptr(boxes), L.ptr(score), C.c_float(threshold), C.c_float(iou_threshold), L.ptr(idx), L.ptr(count),
        L.ptr(ws), C.c_longlong(nbytes), B, C.c_longlong(A), L.stream_ptr()), 'effdet_nms'), 'BYTES B%d A%d' % (B, A))
    return idx, count


def gather_dets(boxes, score, label, idx, count):
    B, A = score.shape
    os_ = torch.empty((B, A), dtype=torch.float32, device=score.device)
    ol = torch.empty((B, A), dtype=torch.int64, device=score.device)
    ob = torch.empty((B, A, 4), dtype=torch.float32, device=score.device)
    L.check(L.lib().effdet_gather_dets(L.ptr(boxes), L.ptr(score), L.ptr(label), L.ptr(idx), L.ptr(count), L.ptr(os_), L.ptr(ol),
                                       L.ptr(ob), B, C.c_longlong(A), L.stream_ptr()), 'effdet_gather_dets')
    return os_, ol, ob


# ----------------------------------------------------------------------------- loss
def focal_loss_fwd(cls, reg, anc, annots):
    B, A, nc = cls.shape
    N = annots.shape[1]
    nbytes = int(L.lib().effdet_loss_workspace_bytes(B, C.c_longlong(A), nc))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cls.device)
    losses = torch.empty(2, dtype=torch.float32, device=cls.device)
    L.check(L.lib().effdet_focal_loss_fwd(L.ptr(cls), L.ptr(reg), L.ptr(anc), L.ptr(annots), L.ptr(losses), L.ptr(ws),
                                          C.c_longlong(nbytes), B, C.c_longlong(A), nc, N, L.stream_ptr()), 'effdet_focal_loss_fwd')
    return losses, ws


def focal_loss_bwd(cls, reg, anc, annots, gscale, ws, dtype):
    B, A, nc = cls.shape
    dcls = torch.empty((B, A, nc), dtype=dtype, device=cls.device)
    dreg = torch.empty((B, A, 4), dtype=dtype, device=cls.device)
    L.check(L.lib().effdet_focal_loss_bwd(L.ptr(cls), L.ptr(reg), L.ptr(anc), L.ptr(annots), L.ptr(gscale), L.ptr(ws), L.ptr(dcls),
                                          L.ptr(dreg), L.dtype_code(dtype), B, C.c_longlong(A), nc, annots.shape[1], L.stream_ptr()),
            'effdet_focal_loss_bwd')
    return dcls, dreg


def focal_loss_bwd_pix(cls, reg, anc, annots, gscale, ws, dtype, dld):
    """focal_loss_bwd with d(cls logits) pixel-major and channel-padded: -> (dcls_pix [B, A/9, dld], dreg [B, A, 4])."""
    B, A, nc = cls.shape
    dcls = torch.empty((B, A // 9, dld), dtype=dtype, device=cls.device)
    dreg = torch.empty((B, A, 4), dtype=dtype, device=cls.device)
    L.check(L.lib().effdet_focal_loss_bwd_pix(L.ptr(cls), L.ptr(reg), L.ptr(anc), L.ptr(annots), L.ptr(gscale), L.ptr(ws), L.ptr(dcls),
                                              dld, L.ptr(dreg), L.dtype_code(dtype), B, C.c_longlong(A), nc, annots.shape[1],
                                              L.stream_ptr()), 'effdet_focal_loss_bwd_pix')
    return dcls, dreg


def focal_loss_fwd_grad(cls, reg, anc, annots, dtype, dld, split=False):
    """Training fast path: -> (losses [2], ws, dcls_pix [B, A/9, dld]) in one pass over cls; dcls_pix is the gradient wrt the
    logits for an upstream gradient of ONE (the caller scales downstream, see effdet_hip.h)."""
    B, A, nc = cls.shape
    nbytes = int(L.lib().effdet_loss_workspace_bytes(B, C.c_longlong(A), nc))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cls.device)
    losses = torch.empty(2, dtype=torch.float32, device=cls.device)
    dcls = torch.empty((B, A // 9, dld), dtype=dtype, device=cls.device)
    L.check(L.lib().effdet_focal_loss_fwd_grad(L.ptr(cls), L.ptr(reg), L.ptr(anc), L.ptr(annots), L.ptr(losses), L.ptr(ws),
                                               C.c_longlong(nbytes), L.ptr(dcls), dld, L.F32_SPLIT if split else L.dtype_code(dtype), B, C.c_longlong(A), nc,
                                               annots.shape[1], L.stream_ptr()), 'effdet_focal_loss_fwd_grad')
    return losses, ws, dcls


def focal_loss_bwd_reg(reg, anc, annots, gscale, ws, dtype, reg_ld=0, split=False):
    """-> dreg [B, A, 4], or with reg_ld pixel-major and channel-padded [B, A/9, reg_ld] (split=True: in the split layout)."""
    B, A, _ = reg.shape
    dreg = torch.empty((B, A // 9, reg_ld) if reg_ld else (B, A, 4), dtype=dtype, device=reg.device)
    L.check(L.lib().effdet_focal_loss_bwd_reg(L.ptr(reg), L.ptr(anc), L.ptr(annots), L.ptr(gscale), L.ptr(ws), L.ptr(dreg), reg_ld,
                                              L.F32_SPLIT if split else L.dtype_code(dtype), B, C.c_longlong(A), annots.shape[1],
                                              L.stream_ptr()), 'effdet_focal_loss_bwd_reg')
    return dreg


def pad_rows(src_map, cpad):
    """Level map with unaligned channel count -> fresh contiguous [B,H,W,cpad] map, zero padded."""
    m = src_map
    dst = Map.new(m.B, m.H, m.W, cpad, m.dtype, m.t.device)
    L.check(L.lib().effdet_pad_rows(L.ptr(m.t), L.ptr(dst.t), L.dtype_code(m.dtype), C.c_longlong(m.off), C.c_longlong(m.bstride),
                                    m.ld, m.B, m.H * m.W, m.C, cpad, L.stream_ptr()), 'effdet_pad_rows')
    return dst


# ----------------------------------------------------------------------------- boundary kernels (csrc/pipeline.hip)
def drop_connect_scales(keep_dev, B, seed, step, step_dev=None):
    """-> [nslot, B] fp32 rows of floor(keep + u)/keep (models/utils.py:79-90), one launch for every skip block of a step.
    step_dev (int64 device tensor [1]): the step counter lives on the device and is advanced by the kernel (hipGraph replay)."""
    n = keep_dev.numel()
    out = torch.empty((n, B), dtype=torch.float32, device=keep_dev.device)
    L.check(L.lib().effdet_drop_connect_scales(L.ptr(out), L.ptr(keep_dev), n, B, C.c_ulonglong(seed & (2 ** 64 - 1)),
                                               C.c_ulonglong(step), L.ptr(step_dev), L.stream_ptr()), 'effdet_drop_connect_scales')
    return out


def philox_host(ctr, key):
    """Host twin of the device generator (Philox4x32-10): -> 4 uint32 words."""
    c = (C.c_uint * 4)(*ctr); k = (C.c_uint * 2)(*key); o = (C.c_uint * 4)()
    f = L.lib().effdet_philox4x32_10
    f.restype = None
    f(c, k, o)
    return [int(x) for x in o]


def preprocess_batch(src_u8, src_off, src_hw, S, dtype, cpad, mean, std, flip=None, annots=None):
    """uint8 HWC images (concatenated, device) -> (Map [B,S,S,cpad] normalised / resized / padded, scale [B] fp32).
    annots [B,M,5] fp32 (device) is transformed in place (datasets/augmentation.py:69-150)."""
    B = src_hw.shape[0]
    out = Map.new(B, S, S, cpad, dtype, src_u8.device)
    scale = torch.empty(B, dtype=torch.float32, device=src_u8.device)
    m = (C.c_float * 3)(*mean); s = (C.c_float * 3)(*std)
    L.check(L.lib().effdet_preprocess_batch(L.ptr(src_u8), L.ptr(src_off), L.ptr(src_hw), L.ptr(flip), L.ptr(out.t), L.ptr(scale),
                                            L.ptr(annots), annots.shape[1] if annots is not None else 0, L.dtype_code(dtype), B, S,
                                            cpad, m, s, L.stream_ptr()), 'effdet_preprocess_batch')
    return out, scale


def finalize_dets(score, label, boxes, count, scale, score_threshold, max_det, xywh=False):
    """Batched eval consumer: -> (out [B,max_det,6] = x1,y1,x2,y2,score,label with boxes / scale, out_count [B] int32)."""
    B, A = score.shape
    out = torch.empty((B, max_det, 6), dtype=torch.float32, device=score.device)
    oc = torch.empty(B, dtype=torch.int32, device=score.device)
    L.check(L.lib().effdet_finalize_dets(L.ptr(score), L.ptr(label), L.ptr(boxes), L.ptr(count), L.ptr(scale),
                                         C.c_float(score_threshold), max_det, int(xywh), L.ptr(out), L.ptr(oc), B, C.c_longlong(A),
                                         L.stream_ptr()), 'effdet_finalize_dets')
    return out, oc


def head_out_bwd(dprob, prob, dreg, dtype):
    """(d loss / d probability, probability, d loss / d regression) fp32 -> (dlogit, dreg) in the compute dtype."""
    dl = torch.empty(prob.shape, dtype=dtype, device=prob.device)
    dr = torch.empty(dreg.shape, dtype=dtype, device=prob.device)
    L.check(L.lib().effdet_head_out_bwd(L.ptr(dprob), L.ptr(prob), L.ptr(dreg), L.ptr(dl), L.ptr(dr), L.dtype_code(dtype),
                                        C.c_longlong(prob.numel()), C.c_longlong(dreg.numel()), L.stream_ptr()), 'effdet_head_out_bwd')
    return dl, dr


def tuning_set(key, value):
    """Kernel-selection knob of the library (speed only; see effdet_tuning_set in include/effdet_hip.h) -> previous value."""
    return int(L.lib().effdet_tuning_set(int(key), int(value)))
