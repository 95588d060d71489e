"""Explicit forward / backward of the EfficientDet components over the HIP kernels.

Each component has a ``*_fwd`` that returns (outputs, saved) and a ``*_bwd`` that maps output gradients
to input + parameter gradients.  PyTorch only owns memory, streams and the outer autograd graph
(efficientdet.py wraps these in autograd.Function nodes); every numeric op is a kernel of
libeffdet_hip.so.  Reference lines cited are relative to the reference repository root.

Frozen-BN backward without re-reading the conv output: for y = act(s*c + t), c = conv(x, W),
s = gamma*invstd:   dz = dy*act'(z);  dx = dgrad(dz, W*s);  G = x^T dz;  dW = s*G;
dgamma = invstd*(sum_k W*G - mean*sum dz);  dbeta = sum dz.
"""
import os

import torch

from . import ops
from .config import BN_EPS, conv_out
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH, RES_ADD, RES_NONE, RES_RELU_MASK, Map

DW_SAVE_Y = os.environ.get('EFFDET_DW_SAVE_Y', '0') == '1'      # A/B switch: also store the depthwise Swish output in training
BIFPN_WGRAD_GROUP = os.environ.get('EFFDET_BIFPN_WGRAD_GROUP', '1') != '0'     # A/B switch: grouped BiFPN weight gradients
EXPAND_Z_ONLY = os.environ.get('EFFDET_EXPAND_Z_ONLY', '1') == '1'   # training: the expand conv stores its pre-activation only
SE_FUSED = os.environ.get('EFFDET_SE_FUSED', '1') == '1'             # squeeze-excite backward fused into the project conv's gradients
DW_BWD_FUSED = os.environ.get('EFFDET_DW_BWD_FUSED', '1') == '1'       # training, fp32, k = 3: depthwise data + weight gradient in one kernel
PW_BWD_FUSED = os.environ.get('EFFDET_PW_BWD_FUSED', '1') == '1'       # training, fp32, Cin 16 / 24 / 32: expand conv data + weight gradient in one kernel
PW_DGRAD_SE = os.environ.get('EFFDET_PW_DGRAD_SE', '1') == '1'         # training, fp32, Co 16 / 24 / 40: project-conv data gradient + SE backward epilogue as a streaming MFMA kernel
FUSE_EXPAND_DW = os.environ.get('EFFDET_FUSE_EXPAND_DW', '1') == '1'  # inference, fp32 storage: expand conv inside the depthwise kernel
FUSE_CIN = tuple(int(v) for v in os.environ.get('EFFDET_FUSE_CIN', '16,24,32').split(','))     # block input widths that take it (A/B)
# f16x3 BiFPN convs from this much work per launch (output pixels x C^2; x 18 = FLOPs): below ~2 GFLOP the launch is a handful of tiles behind a
# DMA round trip and the exact kernel's deep-staged narrow tiles win (measured, D4 B = 8 @1024, 224 -> 224: M = 8192 87 -> 63 us, M = 2048
# 56 -> 61, M = 512 55 -> 61; D0 B = 32 @512, 64 -> 64: M = 131072 97 -> 59, M = 32768 35 -> 27, M = 8192 22 -> 27)
BIFPN_F16X3_MIN_WORK = float(os.environ.get('EFFDET_BIFPN_F16X3_MIN_WORK', '1.2e8'))
GATE_IN_WEIGHTS = os.environ.get('EFFDET_GATE_IN_WEIGHTS', '1') == '1'  # the SE gate folded into per-image project weights (no channel_scale pass)
# ... in training too (fp32 storage, fused SE backward): built, tested, and OFF.  The depthwise forward then stores its Swish output
# next to the pre-activation, which costs about what channel_scale's pass did: 27.65 / 27.97 ms (off) vs 27.72 / 27.62 ms (on) in round 4;
# round 6, alternating in one GPU call: 28.79 / 28.75 (off) vs 28.66 / 28.62 (on), 16 launches fewer -- but W diag(gate) is another
# rounding of the project conv's products, and with it the D1 and D6 train goldens leave their `1e-3 + s_k` gates (the single ReLU /
# max-pool ties of those fixtures fall the other way): 0.4 % of the step is not worth a parity gate
GATE_IN_WEIGHTS_TRAIN = os.environ.get('EFFDET_GATE_IN_WEIGHTS_TRAIN', '0') == '1'


def chunk_elems(dtype):
    return 8 if dtype == torch.bfloat16 else 4


class ZeroArena:
    """One zero-filled fp32 allocation carved into gradient accumulators (one memset instead of dozens)."""

    def __init__(self, nfloats, device):
        self.buf = ops.zeros(max(int(nfloats), 1), device)
        self.pos = 0

    def take(self, *shape):
        n = 1
        for s in shape:
            n *= s
        v = self.buf[self.pos:self.pos + n].view(*shape)
        self.pos += (n + 63) // 64 * 64
        assert self.pos <= self.buf.numel() + 64
        return v

    @staticmethod
    def need(*sizes):
        return sum((n + 63) // 64 * 64 for n in sizes)


def as_map(t):
    return Map.of(t)


# ------------------------------------------------------------------------------------------ stem
def stem_fwd(img, w, gamma, beta, mean, var, pad, dtype, train, z_only=False):
    """models/efficientnet.py:193  swish(bn0(conv_stem(img)))  (3x3 s2, static same pad).
    z_only (training, consumer = an expand-1 MBConv block handed `xpre`): only the PRE-activation is stored and returned -- the
    block's depthwise kernels Swish their staged tiles (one 256 x 256 x 32 write stream less)."""
    ce = chunk_elems(dtype)
    B, _, H, W = img.shape
    if hasattr(img, 'map'):       # efficientdet.PackedImages: already NHWC / compute dtype / one chunk of channels
        x = img.map
    else:
        x = ops.nchw_to_nhwc(img.contiguous().float(), dtype, cpad=ce)
    s, t, inv = ops.bn_fold(gamma, beta, mean, var, BN_EPS)
    wp = ops.pack_weight(w, dtype, cin_pad=ce)
    Cout = w.shape[0]
    Ho, Wo = conv_out(H, 3, 2, pad), conv_out(W, 3, 2, pad)
    if z_only and train:
        z = Map.new(B, Ho, Wo, Cout, dtype, img.device)
        ops.conv2d(x, wp, z, Cin=ce, Cout=Cout, KH=3, KW=3, stride=2, pad_t=pad[0], pad_l=pad[0], scale=s, shift=t, act=ACT_NONE)
        return z, (x, z, s, inv, mean, w, pad)
    y = Map.new(B, Ho, Wo, Cout, dtype, img.device)
    z = Map.new(B, Ho, Wo, Cout, dtype, img.device) if train else None
    ops.conv2d(x, wp, y, Cin=ce, Cout=Cout, KH=3, KW=3, stride=2, pad_t=pad[0], pad_l=pad[0], scale=s, shift=t,
               act=ACT_SWISH, zs=z)
    return y, (x, z, s, inv, mean, w, pad)


def stem_bwd(saved, dy, dy_is_dz=False):
    """dy_is_dz: the incoming gradient already went through the stem's Swish' (fused into block 0's depthwise data gradient)."""
    x, z, s, inv, mean, w, pad = saved
    Cout, ce = w.shape[0], x.C
    dz = dy if dy_is_dz else ops.act_bwd(dy, z, ACT_SWISH)
    G, dsum = ops.conv2d_wgrad(x, dz, Cin=ce, Cout=Cout, KH=3, KW=3, stride=2, pad_t=pad[0], pad_l=pad[0])
    dw, dg, db = ops.unpack_wgrad_bn(G, w, s, dsum, mean, inv, cin_pad=ce)     # + frozen-BN gamma/beta grads, one launch
    return dw, dg, db


# ------------------------------------------------------------------------------------------ MBConv
def mbconv_fwd(x, blk, P, dtype, train, rowscale=None, xpre=None, in_act=ACT_NONE):
    """models/efficientnet.py:75-105.  P: dict of parameter / buffer tensors of the block.
    rowscale: optional [B] fp32 = drop_connect keep-mask / keep_prob (models/utils.py:79-90).
    in_act=ACT_SWISH (expand == 1 blocks only, with xpre = x): x is the producer's PRE-activation (stem_fwd z_only) -- stated by the
    caller, never inferred."""
    assert in_act == ACT_NONE or (blk.expand == 1 and xpre is x)
    dev = x.t.device
    B, H, W = x.B, x.H, x.W
    sv = {'x': x, 'blk': blk, 'P': P, 'rowscale': rowscale, 'xpre': xpre}      # xpre: see mbconv_bwd (expand == 1 blocks)
    dw_in_act = ACT_NONE
    Ho, Wo = conv_out(H, blk.k, blk.stride, blk.pad), conv_out(W, blk.k, blk.stride, blk.pad)
    # (measured per block, D0 B = 32 @512, expand + depthwise -> fused: k3/s2 Cin 16 492 -> 335 us, k3/s1 Cin 24 267 -> 216; but
    #  k5/s2 Cin 24 224 -> 333, k5/s1 Cin 40 157 -> 168, k3/s2 Cin 40 107 -> 143: the k = 5 tiles leave one workgroup per CU and at
    #  Cin = 40 the expand's MFMA work outweighs the bytes saved -- so only the k = 3, Cin <= 32 blocks take it)
    fuse = (not train and FUSE_EXPAND_DW and blk.expand != 1 and dtype == torch.float32 and blk.k == 3 and blk.cin in FUSE_CIN
            and x.ld == x.C and x.off == 0)
    if fuse:
        s0, t0, i0 = ops.bn_fold(P['bn0.weight'], P['bn0.bias'], P['bn0.running_mean'], P['bn0.running_var'], BN_EPS)
        xe = None            # (the 6x-expanded map exists only tile by tile, in LDS: see ops.expand_dw_fwd)
    elif blk.expand != 1:
        s0, t0, i0 = ops.bn_fold(P['bn0.weight'], P['bn0.bias'], P['bn0.running_mean'], P['bn0.running_var'], BN_EPS)
        if train and EXPAND_Z_ONLY and Ho * Wo > 64:      # (<= 8x8 maps: the direct weight-gradient kernel would Swish every tap load)
            # training stores ONE tensor for the expand conv, its pre-activation: the depthwise forward / weight-gradient kernels
            # Swish their staged tiles (the two streams y_e, z_e were 35 % of the backbone's forward writes)
            ze = Map.new(B, H, W, blk.cexp, dtype, dev)
            ops.conv2d(x, ops.pack_weight(P['expand.weight'], dtype), ze, Cin=blk.cin, Cout=blk.cexp, KH=1, KW=1,
                       scale=s0, shift=t0, act=ACT_NONE)
            xe, dw_in_act = ze, ACT_SWISH
        else:
            xe = Map.new(B, H, W, blk.cexp, dtype, dev)
            ze = Map.new(B, H, W, blk.cexp, dtype, dev) if train else None
            ops.conv2d(x, ops.pack_weight(P['expand.weight'], dtype), xe, Cin=blk.cin, Cout=blk.cexp, KH=1, KW=1,
                       scale=s0, shift=t0, act=ACT_SWISH, zs=ze)
        sv.update(s0=s0, i0=i0, ze=ze)
    else:
        xe = x
        dw_in_act = in_act          # ACT_SWISH: the producer (the stem) handed its pre-activation ONLY (stem_fwd z_only)
    s1, t1, i1 = ops.bn_fold(P['bn1.weight'], P['bn1.bias'], P['bn1.running_mean'], P['bn1.running_var'], BN_EPS)
    wk = ops.dw_pack_weight(P['dw.weight'])
    # training stores the depthwise pre-activation ONLY (the step is bound by HBM write bandwidth, ~2.5 TB/s measured):
    # the gate multiply and the backward of the gate recompute Swish from it
    # gate in per-image project weights (round 4): inference always; training when the fused SE backward applies (it takes the
    # project conv's weight gradient per image, which is where the gate re-enters) -- the depthwise conv then stores BOTH its
    # pre-activation (backward) and its Swish output (the project conv's operand) and channel_scale's read + write pass is gone
    giw = GATE_IN_WEIGHTS and (Ho * Wo) % 128 == 0 and (not train or (
        GATE_IN_WEIGHTS_TRAIN and SE_FUSED and dtype == torch.float32 and (Ho * Wo) % 32 == 0))
    z_only = train and not DW_SAVE_Y and not giw
    if fuse:
        xd, pool_part = ops.expand_dw_fwd(x, P['expand.weight'], s0, t0, wk, s1, t1, blk.k, blk.stride, blk.pad[0], blk.pad[0], Ho, Wo)
        zd = None
    else:
        xd, zd, pool_part = ops.dwconv_fwd(xe, wk, s1, t1, blk.k, blk.stride, blk.pad[0], blk.pad[0], Ho, Wo, save_z=train, pool=True,
                                           save_y=not z_only, in_act=dw_in_act)
    inv_hw = 1.0 / (Ho * Wo)
    w1 = P['se_reduce.weight'].view(blk.cse, blk.cexp); w2 = P['se_expand.weight'].view(blk.cexp, blk.cse)
    gate, mid, pool = ops.se_gate_fwd(pool_part, w1, P['se_reduce.bias'], w2, P['se_expand.bias'], inv_hw, save_mid=train)
    s2, t2, i2 = ops.bn_fold(P['bn2.weight'], P['bn2.bias'], P['bn2.running_mean'], P['bn2.running_var'], BN_EPS)
    y = Map.new(B, Ho, Wo, blk.cout, dtype, dev)
    if giw:
        # y = (W diag(gate_b)) x_d -- the gate rides in per-image project weights (a few MB for the whole batch) instead of
        # a read + write pass over the expanded map (16 channel_scale launches moved 3.2 GB per D0 B = 32 forward: 5 % of it)
        xs = None
        wpb, wstride = ops.scale_pack_weight(P['project.weight'], gate, dtype)
        ops.conv2d(xd, wpb, y, Cin=blk.cexp, Cout=blk.cout, KH=1, KW=1, scale=s2, shift=t2, act=ACT_NONE,
                   rowscale=rowscale if blk.skip else None,
                   res=x if blk.skip else None, res_mode=RES_ADD if blk.skip else RES_NONE, w_image_stride=wstride)
    else:
        xs = ops.channel_scale(zd, gate, ACT_SWISH) if z_only else ops.channel_scale(xd, gate)
        ops.conv2d(xs, ops.pack_weight(P['project.weight'], dtype), y, Cin=blk.cexp, Cout=blk.cout, KH=1, KW=1,
                   scale=s2, shift=t2, act=ACT_NONE, rowscale=rowscale if blk.skip else None,
                   res=x if blk.skip else None, res_mode=RES_ADD if blk.skip else RES_NONE)
    if train:
        sv.update(xe=xe, s1=s1, i1=i1, wk=wk, xd=xd, zd=zd, pool=pool, gate=gate, mid=mid, xs=xs, s2=s2, i2=i2,
                  inv_hw=inv_hw, dw_in_act=dw_in_act, giw=giw)
    return y, sv


def mbconv_bwd(sv, dy):
    """-> (dx Map, grads dict keyed like P)."""
    blk, P, x = sv['blk'], sv['P'], sv['x']
    dtype, dev = x.dtype, x.t.device
    B, H, W = x.B, x.H, x.W
    g = {}
    Ce, Co, Ci, Cs, kk = blk.cexp, blk.cout, blk.cin, blk.cse, blk.k * blk.k
    # ---- project conv (+ drop_connect scale on the branch) + squeeze-excite backward ----
    rs = sv['rowscale'] if blk.skip else None
    wp = P['project.weight']
    w1 = P['se_reduce.weight'].view(Cs, Ce); w2 = P['se_expand.weight'].view(Ce, Cs)
    hw = dy.H * dy.W
    giw = sv.get('giw', False)
    if giw and not (SE_FUSED and hw % 32 == 0):       # (the switch was flipped between forward and backward: tests do that)
        sv['xs'] = ops.channel_scale(sv['xd'], sv['gate']); giw = False
    if SE_FUSED and hw % (64 if dtype == torch.bfloat16 else 32) == 0:
        # Fused form (no pass over the activations for the gate gradient, dxs never materialised):
        #   per-image partial weight gradients M_b = dy_b^T xs_b (split-K on image boundaries)
        #   dW = sum_b rs_b M_b (slab scale in the unpack);  (dgate*gate)[b][c] = rs_b sum_n W'[n][c] M_b[n][c]
        #   dz_d = (rs_b (dy W') gate[b][c] + dpool[b][c]) * swish'(z_d)   in the epilogue of the project conv's data gradient
        # -- se_dgate (2 tensor reads), se_bwd_apply (2 reads + 1 write) and the drop_connect act_bwd become one extra read of z_d.
        # (giw: the forward ran on W diag(gate_b) and x_d; the slabs are then M'_b = dy_b^T x_d,b -- the gate re-enters as a per-(image,
        #  channel) factor of the unpack, and sum_n W'[n][c] M'_b[n][c] is d(loss)/d(gate) itself, not yet times the gate)
        G2, dsum2 = ops.conv2d_wgrad(sv['xd'] if giw else sv['xs'], dy, Cin=Ce, Cout=Co, KH=1, KW=1, image_splits=True)
        g['project.weight'], g['bn2.weight'], g['bn2.bias'] = ops.unpack_wgrad_bn(G2, wp, sv['s2'], dsum2, P['bn2.running_mean'], sv['i2'],
                                                                                 slab_scale=rs, slab_cscale=sv['gate'] if giw else None)
        dgg = ops.se_dgate_from_wgrad(G2, wp, sv['s2'], rs, B)
        dpool, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgg, sv['gate'], sv['mid'], sv['pool'], w1, P['se_reduce.bias'], w2,
                                                    sv['inv_hw'], times_gate=not giw)
        # the high-resolution blocks (Co 16 / 24 / 40 over >= 64 k pixels): a streaming kernel of its own (round 6)
        dzd = ops.pw_dgrad_se(dy, wp, sv['s2'], rs, sv['gate'], dpool, sv['zd']) if PW_DGRAD_SE and dtype == torch.float32 else None
        if dzd is None:
            dzd = Map.new(B, dy.H, dy.W, Ce, dtype, dev)
            ops.conv2d(dy, ops.pack_weight(wp, dtype, mode=1, scale=sv['s2']), dzd, Cin=Co, Cout=Ce, KH=1, KW=1, rowscale=rs,
                       bc_scale=sv['gate'], bc_shift=dpool, res=sv['zd'], res_mode=ops.RES_SWISH_GRAD)
    else:
        dz2 = ops.act_bwd(dy, None, ACT_NONE, rowscale=rs) if rs is not None else dy
        G2, dsum2 = ops.conv2d_wgrad(sv['xs'], dz2, Cin=Ce, Cout=Co, KH=1, KW=1)
        g['project.weight'], g['bn2.weight'], g['bn2.bias'] = ops.unpack_wgrad_bn(G2, wp, sv['s2'], dsum2, P['bn2.running_mean'], sv['i2'])
        dxs = Map.new(B, dy.H, dy.W, Ce, dtype, dev)
        ops.conv2d(dz2, ops.pack_weight(wp, dtype, mode=1, scale=sv['s2']), dxs, Cin=Co, Cout=Ce, KH=1, KW=1)
        dgate = ops.se_dgate(dxs, sv['xd']) if sv['xd'] is not None else ops.se_dgate(dxs, sv['zd'], ACT_SWISH)
        dpool, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgate, sv['gate'], sv['mid'], sv['pool'], w1, P['se_reduce.bias'], w2,
                                                    sv['inv_hw'])
        dzd = ops.se_bwd_apply(dxs, sv['gate'], dpool, sv['zd'])
    g['se_reduce.weight'], g['se_reduce.bias'] = dw1.view(Cs, Ce, 1, 1), db1
    g['se_expand.weight'], g['se_expand.bias'] = dw2.view(Ce, Cs, 1, 1), db2
    # ---- depthwise ----
    # (expand == 1, i.e. block 0: the depthwise conv reads the block input itself; given the pre-activation `xpre` of the producer --
    #  the stem's z -- its Swish' is applied here, in the data gradient's epilogue, instead of a separate pass in the stem's backward)
    zprev = sv.get('ze') if blk.expand != 1 else sv.get('xpre')
    fused = None
    if DW_BWD_FUSED and sv['dw_in_act'] == ACT_SWISH and zprev is sv['xe'] and dtype == torch.float32:
        # z-only storage: the depthwise input IS swish(zprev) -- one kernel reads dz_d and zprev once for both gradients (round 6)
        fused = ops.dwconv_bwd(dzd, sv['wk'], sv['s1'], zprev, blk.k, blk.stride, blk.pad[0], blk.pad[0])
    if fused is not None:
        dze, gk, dsum1 = fused
    else:
        gk, dsum1 = ops.dwconv_wgrad(sv['xe'], dzd, blk.k, blk.stride, blk.pad[0], blk.pad[0], in_act=sv['dw_in_act'])
        dze = ops.dwconv_dgrad(dzd, sv['wk'], sv['s1'], zprev, H, W, blk.k, blk.stride, blk.pad[0], blk.pad[0])
    g['dw.weight'], g['bn1.weight'], g['bn1.bias'] = ops.dw_unpack_wgrad_bn(gk, sv['s1'], P['dw.weight'], dsum1,
                                                                             P['bn1.running_mean'], sv['i1'])
    if blk.expand == 1:
        return dze, g           # block 0: depthwise acts on the block input directly, no skip
    # ---- expand conv; the identity-skip gradient is added in the data-gradient epilogue ----
    we = P['expand.weight']
    fusedp = ops.pw_bwd(dze, x, we, sv['s0'], dy if blk.skip else None) if PW_BWD_FUSED and dtype == torch.float32 else None
    if fusedp is not None:
        # the high-resolution blocks (Cin 16 / 24 / 32): both gradients of the expand conv from ONE read of the 6x-expanded gradient (round 6)
        dx, G0, dsum0 = fusedp
    else:
        G0, dsum0 = ops.conv2d_wgrad(x, dze, Cin=Ci, Cout=Ce, KH=1, KW=1)
    g['expand.weight'], g['bn0.weight'], g['bn0.bias'] = ops.unpack_wgrad_bn(G0, we, sv['s0'], dsum0, P['bn0.running_mean'], sv['i0'])
    if fusedp is None:
        dx = Map.new(B, H, W, Ci, dtype, dev)
        ops.conv2d(dze, ops.pack_weight(we, dtype, mode=1, scale=sv['s0']), dx, Cin=Ce, Cout=Ci, KH=1, KW=1,
                   res=dy if blk.skip else None, res_mode=RES_ADD if blk.skip else RES_NONE)
    return dx, g


# ------------------------------------------------------------------------------------------ BiFPN
def lateral_fwd(feats, weights, biases, W, dtype):
    """models/bifpn.py:100-103: 1x1 conv + bias per level (different Cin / weights per level)."""
    outs = []
    for f, w, b in zip(feats, weights, biases):
        y = Map.new(f.B, f.H, f.W, W, dtype, f.t.device)
        ops.conv2d(f, ops.pack_weight(w, dtype), y, Cin=f.C, Cout=W, KH=1, KW=1, shift=b)
        outs.append(y)
    return outs


def lateral_bwd(feats, weights, douts, dtype):
    dfs, dws, dbs = [], [], []
    for f, w, dy in zip(feats, weights, douts):
        W, Cin = w.shape[0], w.shape[1]
        G, dbp = ops.conv2d_wgrad(f, dy, Cin=Cin, Cout=W, KH=1, KW=1)
        dw = torch.empty_like(w)
        db = ops.unpack_wgrad(G, dw, dbias_part=dbp)
        df = Map.new(f.B, f.H, f.W, Cin, dtype, w.device)
        ops.conv2d(dy, ops.pack_weight(w, dtype, mode=1), df, Cin=W, Cout=Cin, KH=1, KW=1)
        dfs.append(df); dws.append(dw); dbs.append(db)
    return dfs, dws, dbs


def _conv3_fwd(x, w, b, dtype, xh=None):
    """xh: the same input in the H-split layout -> the conv runs in the f16x3 arithmetic (plain fp32 output)."""
    src = x if xh is None else xh
    y = Map.new(src.B, src.H, src.W, w.shape[0], dtype, src.t.device)
    if xh is not None:
        ops.conv2d(xh, ops.pack_weight(w, dtype, h3=True), y, Cin=w.shape[1], Cout=w.shape[0], KH=3, KW=3, pad_t=1, pad_l=1, shift=b,
                   hsplit=True, out_f32=True)
    else:
        ops.conv2d(x, ops.pack_weight(w, dtype), y, Cin=w.shape[1], Cout=w.shape[0], KH=3, KW=3, pad_t=1, pad_l=1, shift=b)
    return y


def bifpn_module_fwd(p, w1, w2, cw, cb, dtype, train):
    """models/bifpn.py:172-203.  p: 5 Maps (fine -> coarse); cw/cb: the 8 conv weights / biases.
    Node order and aliasing follow the reference exactly: top-down results overwrite p[3..0], the
    bottom-up pass reads those plus the ORIGINAL inputs (its `inputs_clone`)."""
    L_ = len(p)
    t_in = list(p)
    cur = list(p)
    nodes = []            # (mode, col, wsel, a_name, b_name, c_name, out_name, fused Map)
    names = {('in', l): t_in[l] for l in range(L_)}
    c = 0
    # f16x3 (ops.F32_ARITH_HEAD, the headline mode): the node's 3x3 conv reads the fused map in the H-split layout, which the fusion kernel
    # writes itself (inference: instead of the plain map; training: next to it -- the conv's weight gradient reads the plain one)
    hs = head_uses_f16x3(cw[0].shape[0], dtype) and cw[0].shape[1] == cw[0].shape[0]

    def fuse(a, b, cc, w, col, mode):
        if not hs or a.B * a.H * a.W * a.C * a.C < BIFPN_F16X3_MIN_WORK:
            return ops.bifpn_fuse_fwd(a, b, cc, w, col, mode), None
        return ops.bifpn_fuse_fwd(a, b, cc, w, col, mode, plain=train, hsplit=True)
    for i in range(L_ - 1, 0, -1):                                  # top-down, bifpn.py:188-192
        a_n = ('in', i - 1); b_n = ('in', i) if i == L_ - 1 else ('td', i)
        f, fh = fuse(names[a_n], names[b_n], None, w1, i - 1, 0)
        out_n = ('td', i - 1); names[out_n] = _conv3_fwd(f, cw[c], cb[c], dtype, fh)
        nodes.append((0, i - 1, 1, a_n, b_n, None, out_n, f, c)); c += 1
    for i in range(0, L_ - 2):                                      # bottom-up, bifpn.py:194-198
        a_n = ('td', i + 1); b_n = ('td', 0) if i == 0 else ('bu', i); c_n = ('in', i + 1)
        f, fh = fuse(names[a_n], names[b_n], names[c_n], w2, i, 1)
        out_n = ('bu', i + 1); names[out_n] = _conv3_fwd(f, cw[c], cb[c], dtype, fh)
        nodes.append((1, i, 2, a_n, b_n, c_n, out_n, f, c)); c += 1
    a_n = ('in', L_ - 1); b_n = ('bu', L_ - 2)                      # top node, bifpn.py:200-202
    f, fh = fuse(names[a_n], names[b_n], None, w1, L_ - 1, 2)
    out_n = ('top', L_ - 1); names[out_n] = _conv3_fwd(f, cw[c], cb[c], dtype, fh)
    nodes.append((2, L_ - 1, 1, a_n, b_n, None, out_n, f, c))
    out_names = [('td', 0)] + [('bu', l) for l in range(1, L_ - 1)] + [out_n]
    outs = [names[n] for n in out_names]
    cur = outs
    saved = (nodes, names, out_names, w1, w2, cw) if train else None
    return cur, saved


def bifpn_module_bwd(saved, douts, dtype, own=False):
    """-> (d_inputs [5 Maps], dw1, dw2, dcw [8], dcb [8]).  own: `douts` are private buffers of the caller (the d_inputs of the next
    module's backward) and may be accumulated into; gradients handed over by autograd are copied first."""
    nodes, names, out_names, w1, w2, cw = saved
    dev = w1.device
    Wc = cw[0].shape[0]
    grads = {}                                   # tensor name -> Map (private, safe to accumulate into)
    for n, d in zip(out_names, douts):
        grads[n] = d if own else Map.of(d.tensor().clone())    # never write into autograd's grad_outputs
    n1, n2 = ops.fuse_dn_floats(w1.shape[1]), ops.fuse_dn_floats(w2.shape[1])
    ar = ZeroArena(ZeroArena.need(n1, n2), dev)
    dn1, dn2 = ar.take(n1), ar.take(n2)     # per-workgroup partial rows of d loss / d n_r (zeroed: a column without a launch adds nothing)
    dcw, dcb = [None] * 8, [None] * 8

    def target(name):
        like = names[name]
        if name in grads:
            return grads[name], True
        grads[name] = Map.new(like.B, like.H, like.W, like.C, dtype, dev)
        return grads[name], False

    pending = []                                                     # (M, f, dz, conv index): weight gradients, leaf work
    for (mode, col, wsel, a_n, b_n, c_n, out_n, f, ci) in reversed(nodes):
        dz = grads[out_n]                                            # conv has bias only: dz = dy (final: its consumers ran first)
        if BIFPN_WGRAD_GROUP:
            pending.append((f.B * f.H * f.W, f, dz, ci))
        else:
            G, dbp = ops.conv2d_wgrad(f, dz, Cin=Wc, Cout=Wc, KH=3, KW=3, pad_t=1, pad_l=1)
            dw = torch.empty_like(cw[ci]); db = ops.unpack_wgrad(G, dw, dbias_part=dbp)
            dcw[ci], dcb[ci] = dw, db
        df = Map.new(f.B, f.H, f.W, Wc, dtype, dev)
        ops.conv2d(dz, ops.pack_weight(cw[ci], dtype, mode=1), df, Cin=Wc, Cout=Wc, KH=3, KW=3, pad_t=1, pad_l=1)
        da, da_acc = target(a_n); dbm, db_acc = target(b_n)
        dc, dc_acc = target(c_n) if c_n is not None else (None, False)
        ops.bifpn_fuse_bwd(df, names[a_n], names[b_n], names[c_n] if c_n else None, da, dbm, dc, da_acc, db_acc, dc_acc,
                           w1 if wsel == 1 else w2, dn1 if wsel == 1 else dn2, col, mode)
    # The 8 weight gradients of the module as TWO launches of independent problems (one pyramid level each, own weights): launched
    # one by one they are 19-89 us apiece -- the 4x4 .. 16x16 levels are 5-160 workgroups walking one dependent chain -- 268 us
    # per module for 27 GFLOP.  The largest level shares its launch with the four smallest, the 32x32 / 16x16 ones share the other.
    if pending:
        pending.sort(key=lambda t: -t[0])
        chunks = [pending[:1] + pending[4:], pending[1:4]] if len(pending) > 5 else [pending]
        for ch in chunks:
            if not ch:
                continue
            outs = ops.conv2d_wgrad([t[1] for t in ch], [t[2] for t in ch], Cin=Wc, Cout=Wc, KH=3, KW=3, pad_t=1, pad_l=1, group=True)
            for (_, _, _, ci), (G, dbp) in zip(ch, outs):
                dw = torch.empty_like(cw[ci]); dcb[ci] = ops.unpack_wgrad(G, dw, dbias_part=dbp); dcw[ci] = dw
    dw1 = ops.zeros(tuple(w1.shape), w1.device); dw2 = ops.zeros(tuple(w2.shape), w2.device)
    ops.bifpn_weight_bwd(w1, dn1, dw1); ops.bifpn_weight_bwd(w2, dn2, dw2)
    dins = [grads[('in', l)] for l in range(len(out_names))]
    return dins, dw1, dw2, dcw, dcb


# ------------------------------------------------------------------------------------------ RetinaHead
def pyramid_alloc(B, sizes, C, dtype, device):
    """All levels of one activation in ONE flat buffer (level-major) so that grouped launches can
    address every pyramid tensor of the same channel count with identical relative offsets."""
    tot = sum(B * h * w * C for (h, w) in sizes)
    flat = torch.empty(tot, dtype=dtype, device=device)
    maps, off = [], 0
    for (h, w) in sizes:
        maps.append(Map(flat, B, h, w, C, off=off)); off += B * h * w * C
    return flat, maps


def pyramid_alloc_pair(B, sizes, C, dtype, device):
    """Two pyramids of one geometry (the same layer of the head's two towers) in ONE flat buffer, tower-major then level-major: every
    buffer allocated this way has the same relative offsets, which is what lets one grouped launch address outputs, split copies and
    ReLU-mask residuals of both towers from one base each.  -> (flat, maps of tower A, maps of tower B)"""
    tot = sum(B * h * w * C for (h, w) in sizes)
    flat = torch.empty(2 * tot, dtype=dtype, device=device)
    out = []
    for half in (0, 1):
        maps, off = [], half * tot
        for (h, w) in sizes:
            maps.append(Map(flat, B, h, w, C, off=off)); off += B * h * w * C
        out.append(maps)
    return flat, out[0], out[1]


def level_tensor(m):
    n = m.B * m.H * m.W * m.C
    return m.t[m.off:m.off + n].view(m.B, m.H, m.W, m.C)


def head_out_maps(buf, B, sizes, per_anchor):
    """Views of a [B, A, per_anchor] output buffer as per-level NHWC maps with 9*per_anchor channels
    (models/retinahead.py:119-127: NHWC permute + view(B, -1, C) is free in this layout)."""
    A = sum(h * w for (h, w) in sizes) * 9
    maps, aoff = [], 0
    for (h, w) in sizes:
        maps.append(Map(buf, B, h, w, 9 * per_anchor, ld=9 * per_anchor, bstride=A * per_anchor, off=aoff * per_anchor))
        aoff += h * w * 9
    return maps


# The two towers of the RetinaHead are independent chains of equal launches.  A 256 -> 256 tower conv is 2728 tiles on 512 workgroup slots =
# 5.33 rounds: the last round runs a third full, i.e. ~11 % of every launch is a draining tail (PMC: MFMA busy 0.82 = 0.89 of quantisation x
# 0.92 in the loop).  On two streams (one graph branch each under capture) the regression tower's workgroups fill the classification
# tower's tail and vice versa.  Fork / join discipline instead of record_stream: the side stream starts behind everything the main stream
# has enqueued, and the main stream waits for it before anything the side stream touched can be freed or reused.
HEAD_TWO_STREAMS = os.environ.get('EFFDET_HEAD_TWO_STREAMS', '0') == '1'
_side_streams = {}


class _Fork:
    """with _Fork(device, on) as f: ...main-stream work...; with f.side(): ...side-stream work...   (joined at exit; on=False: one stream)"""

    def __init__(self, device, on):
        self.on = on
        if on:
            key = (device.index, torch.cuda.current_stream(device).cuda_stream)
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device)
            self.main, self.s = torch.cuda.current_stream(device), _side_streams[key]

    def __enter__(self):
        if self.on:
            self.s.wait_stream(self.main)
        return self

    def side(self):
        import contextlib
        return torch.cuda.stream(self.s) if self.on else contextlib.nullcontext()

    def join(self):
        if self.on:
            self.main.wait_stream(self.s)
            self.on = False

    def __exit__(self, et, ev, tb):
        self.join()
        return False


HEAD_PAIR_TOWERS = os.environ.get('EFFDET_HEAD_PAIR_TOWERS', '1') == '1'     # A/B switch: layer t of both towers as one launch (f16x3 forward, split-layout data gradients)
HEAD_SPLIT = os.environ.get('EFFDET_HEAD_SPLIT', '1') == '1'     # A/B switch: split-layout head activations in the bf16x3 arithmetic
_split_ok = {}


def head_uses_split(B, sizes, Wc, dtype, arith=None):
    """bf16x3 arithmetic on fp32 storage: the RetinaHead's activations and gradients live in the SPLIT layout (every 32 channels
    as [32 x bf16 hi | 32 x bf16 lo], same 4 bytes per element) so that the head's convs and weight gradients -- 95 % of the
    step's FLOPs -- read ready-made MFMA operands instead of splitting fp32 values in registers in every K-step.  Needs pyramid
    levels the split weight-gradient kernel can take (whole 8-pixel runs per row); otherwise the plain-fp32 path stays.
    arith: the arithmetic asked about (default: the one the launches use now)."""
    if not (HEAD_SPLIT and dtype == torch.float32 and (arith or ops.F32_ARITH) == 'bf16x3' and Wc % 32 == 0):
        return False
    key = (B, tuple(sizes), Wc)
    if key not in _split_ok:
        _split_ok[key] = ops.wgrad_split_supported(B, sizes, 256, 256, 256) and ops.wgrad_split_supported(B, sizes, Wc, 256, 256)
    return _split_ok[key]


def _pyramid_to_split(src_maps, B, sizes, C_, dtype, dev, bf=True, h=False):
    """A pyramid activation (plain fp32, one tensor per level) -> fresh flat level-major buffers holding it in the split layout (bf) and /
    or the H-split layout of the f16x3 forward convs (h), written by ONE pass per level.  -> (split maps or None, H-split maps or None)"""
    dst = pyramid_alloc(B, sizes, C_, dtype, dev)[1] if bf else None
    dsth = pyramid_alloc(B, sizes, C_, dtype, dev)[1] if h else None
    for i, s_ in enumerate(src_maps):
        n = s_.B * s_.H * s_.W * s_.C
        if h:
            ops.L.check(ops.L.lib().effdet_to_split2(ops.L.ptr(s_.tensor()), ops.C.c_void_p(dst[i].addr() if bf else None),
                                                     ops.C.c_void_p(dsth[i].addr()), ops.C.c_longlong(n), ops.L.ptr(ops.range_flag(dev)),
                                                     ops.L.stream_ptr()), 'effdet_to_split2')
        else:
            ops.L.check(ops.L.lib().effdet_to_split(ops.L.ptr(s_.tensor()), ops.C.c_void_p(dst[i].addr()), ops.C.c_longlong(n), ops.L.stream_ptr()),
                        'effdet_to_split')
    return dst, dsth


def head_uses_f16x3(Wc, dtype):
    """The f16x3 forward head (ops.F32_ARITH_HEAD): fp32 storage, exact-fp32 arithmetic around it, whole 32-channel groups."""
    return HEAD_SPLIT and dtype == torch.float32 and ops.F32_ARITH == 'f32' and ops.F32_ARITH_HEAD == 'f16x3' and Wc % 32 == 0


def head_fwd(p, HP, num_classes, dtype, train):
    """models/retinahead.py:109-132 for all 5 levels per launch (weights are shared across levels).
    p: 5 Maps; HP: dict of head parameter tensors.  -> classification [B,A,nc] fp32 (probabilities),
    regression [B,A,4] fp32.
    Arithmetic 'f32_bwd_bf16x3' (forward exact, gradients in the three-product form): the forward below runs on plain fp32
    activations with exact-fp32 products, and every activation the BACKWARD will read -- the pyramid and the eight tower outputs,
    operands of the weight gradients and ReLU masks of the data gradients -- is stored a second time in the split layout by the
    conv that produces it (`ysplit`: the epilogue writes both forms; the plain copy dies with the next layer), so that head_bwd
    runs the split-layout gradient kernels unchanged."""
    dev = p[0].t.device
    B, Wc = p[0].B, p[0].C
    sizes = [(m.H, m.W) for m in p]
    A = sum(h * w for (h, w) in sizes) * 9
    split = head_uses_split(B, sizes, Wc, dtype)
    split_bwd = (not split) and train and ops.F32_ARITH_BWD == 'bf16x3' and head_uses_split(B, sizes, Wc, dtype, 'bf16x3')
    # f16x3 forward (fp32-equivalent three-product fp16 form): every conv below reads H-split operands and writes H-split outputs; the
    # backward is the split-layout bf16x3 one (split_bwd: its operands / masks ride along as `ysplit`).  Training geometries the split
    # gradient kernels cannot take keep the exact-fp32 head.
    hs = (not split) and head_uses_f16x3(Wc, dtype) and (split_bwd or not train)
    pin, pin_h = p, None
    if split or split_bwd or hs:    # the pyramid once in the split layout(s): operand of both towers' first conv and of their weight gradients
        pin, pin_h = _pyramid_to_split(p, B, sizes, Wc, dtype, dev, bf=split or split_bwd, h=hs)
        if pin is None:
            pin = p
    acts = {'cls': [], 'reg': []}
    cls = torch.empty((B, A, num_classes), dtype=torch.float32, device=dev)
    reg = torch.empty((B, A, 4), dtype=torch.float32, device=dev)

    def tower_fwd(tower):
        cur = pin_h if hs else (p if split_bwd else pin)
        for t in range(4):
            w, b = HP[f'{tower}_convs.{t}.weight'], HP[f'{tower}_convs.{t}.bias']
            _, nxt = pyramid_alloc(B, sizes, 256, dtype, dev)
            nxs = pyramid_alloc(B, sizes, 256, dtype, dev)[1] if split_bwd else None
            ops.conv2d(cur, ops.pack_weight(w, dtype, x3=split, h3=hs), nxt, Cin=w.shape[1], Cout=256, KH=3, KW=3, pad_t=1, pad_l=1,
                       shift=b, act=ACT_RELU, split=split, ysplit=nxs, hsplit=hs)
            acts[tower].append(nxs if split_bwd else nxt); cur = nxt       # (split_bwd: `cur` stays plain fp32 / H-split and is not kept for backward)
        if tower == 'cls':
            ops.conv2d(cur, ops.pack_weight(HP['retina_cls.weight'], dtype, x3=split, h3=hs), head_out_maps(cls, B, sizes, num_classes),
                       Cin=256, Cout=9 * num_classes, KH=3, KW=3, pad_t=1, pad_l=1, shift=HP['retina_cls.bias'],
                       act=ACT_SIGMOID, out_f32=True, split=split, hsplit=hs)
        else:
            ops.conv2d(cur, ops.pack_weight(HP['retina_reg.weight'], dtype, x3=split, h3=hs), head_out_maps(reg, B, sizes, 4),
                       Cin=256, Cout=36, KH=3, KW=3, pad_t=1, pad_l=1, shift=HP['retina_reg.bias'], out_f32=True, split=split, hsplit=hs)
    paired = hs and HEAD_PAIR_TOWERS
    if paired:
        # The two towers are independent chains of identical launches: layer t of BOTH as ONE launch (per-segment weights / bias,
        # effdet_conv_t.seg_w / seg_shift): 2 x 2728 tiles on the 512 workgroup slots = 10.66 rounds instead of twice 5.33 -- the
        # draining third of a round is paid once per layer, not twice.  Same tiles, same K walk: bit for bit the separate launches.
        cur = {'cls': pin_h, 'reg': pin_h}
        for t in range(4):
            w2 = [HP[f'{tw}_convs.{t}.weight'] for tw in ('cls', 'reg')]
            b2 = [HP[f'{tw}_convs.{t}.bias'] for tw in ('cls', 'reg')]
            wp2 = [ops.pack_weight(w, dtype, h3=True) for w in w2]
            _, ya, yb = pyramid_alloc_pair(B, sizes, 256, dtype, dev)
            sa = sb = None
            if split_bwd:
                _, sa, sb = pyramid_alloc_pair(B, sizes, 256, dtype, dev)
            L5 = len(sizes)
            ops.conv2d(cur['cls'] + cur['reg'], wp2[0], ya + yb, Cin=w2[0].shape[1], Cout=256, KH=3, KW=3, pad_t=1, pad_l=1, shift=b2[0],
                       act=ACT_RELU, hsplit=True, ysplit=(sa + sb) if split_bwd else None,
                       seg_w=[wp2[0]] * L5 + [wp2[1]] * L5, seg_shift=[b2[0]] * L5 + [b2[1]] * L5)
            acts['cls'].append(sa if split_bwd else ya); acts['reg'].append(sb if split_bwd else yb)
            cur = {'cls': ya, 'reg': yb}
        ops.conv2d(cur['cls'], ops.pack_weight(HP['retina_cls.weight'], dtype, h3=True), head_out_maps(cls, B, sizes, num_classes),
                   Cin=256, Cout=9 * num_classes, KH=3, KW=3, pad_t=1, pad_l=1, shift=HP['retina_cls.bias'],
                   act=ACT_SIGMOID, out_f32=True, hsplit=True)
        ops.conv2d(cur['reg'], ops.pack_weight(HP['retina_reg.weight'], dtype, h3=True), head_out_maps(reg, B, sizes, 4),
                   Cin=256, Cout=36, KH=3, KW=3, pad_t=1, pad_l=1, shift=HP['retina_reg.bias'], out_f32=True, hsplit=True)
    else:
        with _Fork(dev, HEAD_TWO_STREAMS and ops.PROFILE is None) as fk:
            with fk.side():
                tower_fwd('reg')
            tower_fwd('cls')
    saved = (pin, acts, sizes, HP, num_classes, split or split_bwd, paired and split_bwd) if train else None
    return cls, reg, saved


def _split_rows(maps, Cfp, dtype):
    """plain fp32 gradient maps (possibly strided views of a [B, A, per] buffer) -> fresh contiguous maps of Cfp (% 32 == 0)
    channels per pixel in the split layout."""
    out = []
    for m in maps:
        padded = ops.pad_rows(m, Cfp) if (m.C != Cfp or m.ld != Cfp or m.bstride != m.H * m.W * Cfp or m.off) else m
        out.append(Map.of(ops.to_split(padded.tensor())))
    return out


def head_bwd(saved, dcls_logit, dreg, dtype, dcls_ld=0, cls_gscale=None, dreg_ld=0, in_split=False):
    """dcls_logit [B,A,nc] (or, with dcls_ld, pixel-major and channel-padded [B,A/9,dcls_ld]: ops.focal_loss_bwd_pix),
    dreg [B,A,4] (or, with dreg_ld, pixel-major [B,A/9,dreg_ld]): gradients wrt the cls LOGITS and box deltas, in `dtype` -- when
    the head was run in the split layout (saved[5]) they are needed in the split layout too: in_split says the pixel-major forms
    already are (the loss kernels write it directly), anything else is converted here.  -> (dp: 5 Maps, grads dict keyed like HP).
    cls_gscale (fp32 tensor [1]): dcls_logit was computed for an upstream gradient of one (ops.focal_loss_fwd_grad); the
    real scalar enters here where the chain is linear: as the per-image output scale of retina_cls's data-gradient conv
    and as a factor on retina_cls's own parameter gradients."""
    p, acts, sizes, HP, nc, split = saved[:6]
    paired = len(saved) > 6 and saved[6] and split          # the forward laid both towers' activations out pairwise (pyramid_alloc_pair)
    dev = p[0].t.device
    B, Wc = p[0].B, p[0].C
    g = {}
    apix = sum(h * w for (h, w) in sizes)
    last = {}

    def tower_final(tower, dout, per, pix_ld):
        """retina_cls / retina_reg: weight gradient + data gradient back into the tower (ReLU mask of its last layer fused) -> dz maps"""
        fin = f'retina_{tower}'
        wf = HP[fin + '.weight']
        Cf = wf.shape[0]
        ce = 32 if split else chunk_elems(dtype)
        if pix_ld:
            # the loss kernel already wrote rows of pix_ld channels per pixel (zeros past 9*per): aligned 128-B K-slices
            Cfp, poff, dzmaps = pix_ld, 0, []
            for (h, w) in sizes:
                dzmaps.append(Map(dout, B, h, w, Cfp, ld=Cfp, bstride=apix * Cfp, off=poff * Cfp)); poff += h * w
            if split and not in_split:
                assert Cfp % 32 == 0
                dzmaps = _split_rows(dzmaps, Cfp, dtype)
        else:
            dzmaps = head_out_maps(dout, B, sizes, per)
            Cfp = (Cf + ce - 1) // ce * ce
            if split:
                dzmaps = _split_rows(dzmaps, Cfp, dtype)
            elif Cfp != Cf:          # 9*num_classes (or 36) channels are not whole 16-byte chunks: zero-pad the rows
                dzmaps = [ops.pad_rows(m, Cfp) for m in dzmaps]
        G, dbp = ops.conv2d_wgrad(acts[tower][3], dzmaps, Cin=256, Cout=Cf, KH=3, KW=3, pad_t=1, pad_l=1, split=split)
        # (cls_gscale: the class-loss gradient was written for an upstream gradient of one -- the upstream scalar multiplies the
        #  slabs and bias partial rows inside the unpack, and the rows of the data gradient)
        gs = cls_gscale if tower == 'cls' else None
        dw = torch.empty_like(wf); db = ops.unpack_wgrad(G, dw, dbias_part=dbp, slab_scale=gs)
        rows = gs.expand(B).contiguous() if gs is not None else None
        g[fin + '.weight'], g[fin + '.bias'] = dw, db
        # data gradient with the ReLU mask of the producing tower layer fused into the epilogue
        _, dz = pyramid_alloc(B, sizes, 256, dtype, dev)
        ops.conv2d(dzmaps, ops.pack_weight(wf, dtype, mode=1, cin_pad=Cfp, x3=split), dz, Cin=Cfp, Cout=256, KH=3, KW=3, pad_t=1,
                   pad_l=1, res=acts[tower][3], res_mode=RES_RELU_MASK, rowscale=rows, split=split)
        return dz

    def layer_wgrad(tower, t, dz):
        w = HP[f'{tower}_convs.{t}.weight']
        xin = acts[tower][t - 1] if t > 0 else p
        G, dbp = ops.conv2d_wgrad(xin, dz, Cin=w.shape[1], Cout=256, KH=3, KW=3, pad_t=1, pad_l=1, split=split)
        dw = torch.empty_like(w); db = ops.unpack_wgrad(G, dw, dbias_part=dbp)
        g[f'{tower}_convs.{t}.weight'], g[f'{tower}_convs.{t}.bias'] = dw, db

    def tower_bwd(tower, dout, per, pix_ld):
        dz = tower_final(tower, dout, per, pix_ld)
        for t in range(3, -1, -1):
            layer_wgrad(tower, t, dz)
            wd = ops.pack_weight(HP[f'{tower}_convs.{t}.weight'], dtype, mode=1, x3=split)
            if t > 0:
                _, nz = pyramid_alloc(B, sizes, 256, dtype, dev)
                ops.conv2d(dz, wd, nz, Cin=256, Cout=256, KH=3, KW=3, pad_t=1, pad_l=1, res=acts[tower][t - 1],
                           res_mode=RES_RELU_MASK, split=split)
                dz = nz
            else:
                last[tower] = (dz, wd)                  # 256 -> Wc back to the neck: after the join (the towers' sum)

    if paired:
        # both towers in lockstep: the weight gradients stay one launch per (tower, layer), the 256 -> 256 data gradients of layer t run
        # as ONE launch for both towers (head_fwd's pairing, same reason: 10.66 rounds of tiles instead of twice 5.33)
        dzs = {'reg': tower_final('reg', dreg, 4, dreg_ld), 'cls': tower_final('cls', dcls_logit, nc, dcls_ld)}
        L5 = len(sizes)
        for t in range(3, -1, -1):
            for tw in ('reg', 'cls'):
                layer_wgrad(tw, t, dzs[tw])
            wd2 = [ops.pack_weight(HP[f'{tw}_convs.{t}.weight'], dtype, mode=1, x3=split) for tw in ('cls', 'reg')]
            if t > 0:
                _, na, nb = pyramid_alloc_pair(B, sizes, 256, dtype, dev)
                ops.conv2d(dzs['cls'] + dzs['reg'], wd2[0], na + nb, Cin=256, Cout=256, KH=3, KW=3, pad_t=1, pad_l=1,
                           res=acts['cls'][t - 1] + acts['reg'][t - 1], res_mode=RES_RELU_MASK, split=split,
                           seg_w=[wd2[0]] * L5 + [wd2[1]] * L5)
                dzs = {'cls': na, 'reg': nb}
            else:
                last['cls'], last['reg'] = (dzs['cls'], wd2[0]), (dzs['reg'], wd2[1])
    else:
        # (the two towers on two streams, see HEAD_TWO_STREAMS: the deferred unpack jobs of both leave in the node's ONE tail launch, on
        #  the main stream, after the join -- hence no early flush in between)
        two = HEAD_TWO_STREAMS and ops.PROFILE is None
        ops.hold_tail_flush(two)
        try:
            with _Fork(dev, two) as fk:
                with fk.side():
                    tower_bwd('reg', dreg, 4, dreg_ld)
                tower_bwd('cls', dcls_logit, nc, dcls_ld)
        finally:
            ops.hold_tail_flush(False)
    # back to plain fp32 for the neck (out_f32 in the split form): the first tower writes, the second accumulates onto it
    _, dp_maps = pyramid_alloc(B, sizes, Wc, dtype, dev)
    dz, wd = last['cls']
    ops.conv2d(dz, wd, dp_maps, Cin=256, Cout=Wc, KH=3, KW=3, pad_t=1, pad_l=1, split=split, out_f32=split)
    dz, wd = last['reg']
    ops.conv2d(dz, wd, dp_maps, Cin=256, Cout=Wc, KH=3, KW=3, pad_t=1, pad_l=1, res=dp_maps, res_mode=RES_ADD, split=split, out_f32=split)
    return dp_maps, g
