"""Thin Python wrappers over the C ABI (raw device tensors in, raw device tensors out).

Plumbing only: PyTorch owns the memory and the stream; every numeric op is a HIP kernel in
libeffdet_hip.so.  No function here has a CPU path.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH, RES_ADD, RES_NONE,  # noqa: F401
                   RES_RELU_MASK, RES_SWISH_GRAD)


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


def pack_weight(w_oihw, dtype, mode=0, scale=None):
    """OIHW fp32 -> packed [Cout][taps][Cin] (mode 0) or data-gradient operand [Cin][taps'][Cout] (mode 1)."""
    Cout, Cin, KH, KW = w_oihw.shape
    w = w_oihw.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    shape = (Cout, KH * KW, Cin) if mode == 0 else (Cin, KH * KW, Cout)
    out = torch.empty(shape, dtype=dtype, device=w.device)
    L.check(L.lib().effdet_pack_conv_weight(L.ptr(w), L.ptr(scale), L.ptr(out), L.dtype_code(dtype), mode,
                                            Cout, Cin, KH, KW, L.stream_ptr()), 'effdet_pack_conv_weight')
    return out


def conv2d(xs, wp, ys, *, Cin, Cout, KH, KW, stride=1, pad_t=0, pad_l=0, scale=None, shift=None, act=ACT_NONE,
           res=None, res_mode=RES_NONE, rowscale=None, zs=None, out_f32=False):
    """Grouped implicit-GEMM conv: xs/ys (and optional zs/res) are lists of Map, one per pyramid level."""
    if isinstance(xs, Map):
        xs, ys = [xs], [ys]
        zs = [zs] if zs is not None else None
        res = [res] if res is not None else None
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
    d.scale, d.shift, d.rowscale = (t.data_ptr() if t is not None else None for t in (scale, shift, rowscale))
    d.dtype, d.out_f32 = L.dtype_code(x0.dtype), int(out_f32)
    d.B, d.Cin, d.Cout, d.KH, d.KW = x0.B, Cin, Cout, KH, KW
    d.stride, d.pad_t, d.pad_l = stride, pad_t, pad_l
    d.ldx, d.ldy = x0.ld, y0.ld
    d.act, d.res_mode = act, res_mode
    _segs(d, xs, ys, base_x, base_y, isx, isy)
    L.check(L.lib().effdet_conv2d(C.byref(d), L.stream_ptr()), 'effdet_conv2d')


def conv2d_wgrad(xs, dzs, dw, dbias=None, *, Cin, Cout, KH, KW, stride=1, pad_t=0, pad_l=0):
    """dw[Cout][taps][Cin] (fp32) += dz^T * im2col(x);  dbias[Cout] += colsum(dz)."""
    if isinstance(xs, Map):
        xs, dzs = [xs], [dzs]
    d = L.WgradDesc()
    x0, z0 = xs[0], dzs[0]
    isz = x0.t.element_size()
    base_x = min(x.addr() for x in xs)
    base_z = min(z.addr() for z in dzs)
    d.x, d.dz, d.dw = base_x, base_z, dw.data_ptr()
    d.dbias = dbias.data_ptr() if dbias is not None else None
    d.dtype = L.dtype_code(x0.dtype)
    d.B, d.Cin, d.Cout, d.KH, d.KW = x0.B, Cin, Cout, KH, KW
    d.stride, d.pad_t, d.pad_l = stride, pad_t, pad_l
    d.ldx, d.lddz = x0.ld, z0.ld
    _segs(d, xs, dzs, base_x, base_z, isz, z0.t.element_size())
    L.check(L.lib().effdet_conv2d_wgrad(C.byref(d), L.stream_ptr()), 'effdet_conv2d_wgrad')


def unpack_wgrad(g, dw_oihw, scale=None, w_oihw=None, wsum=None, accumulate=False):
    Cout, Cin, KH, KW = dw_oihw.shape
    L.check(L.lib().effdet_unpack_conv_wgrad(L.ptr(g), L.ptr(scale), L.ptr(w_oihw), L.ptr(dw_oihw), L.ptr(wsum),
                                             int(accumulate), Cout, Cin, KH, KW, L.stream_ptr()),
            'effdet_unpack_conv_wgrad')


def nhwc_to_nchw(m):
    out = torch.empty((m.B, m.C, m.H, m.W), dtype=torch.float32, device=m.t.device)
    L.check(L.lib().effdet_nhwc_to_nchw_f32(L.ptr(m.tensor()), L.ptr(out), L.dtype_code(m.dtype), m.B, m.H, m.W, m.C,
                                            L.stream_ptr()), 'effdet_nhwc_to_nchw_f32')
    return out


def nchw_to_nhwc(x, dtype):
    B, Cc, H, W = x.shape
    x = x.contiguous().float()
    m = Map.new(B, H, W, Cc, dtype, x.device)
    L.check(L.lib().effdet_nchw_f32_to_nhwc(L.ptr(x), L.ptr(m.t), L.dtype_code(dtype), B, H, W, Cc, L.stream_ptr()),
            'effdet_nchw_f32_to_nhwc')
    return m
