"""ctypes binding of libeffdet_hip.so (include/effdet_hip.h).

The library is the product: there is NO CPU / eager fallback.  ``lib()`` raises if the shared
library is missing and every op raises if a call returns a non-zero status.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the .so binds to torch's already-loaded libamdhip64)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EFFDET_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libeffdet_hip.so')   # override: A/B experiment builds (tools/)
MAX_SEG = 5
MAX_CONV_SEG = 10                  # effdet_conv_t: 5 pyramid levels x 2 independent convs of one geometry (the head's two towers)
ABI_VERSION = 10                   # EFFDET_ABI_VERSION of include/effdet_hip.h this binding was written against (tests/test_abi.py)
F32, BF16, F32_BF16X3, F32_SPLIT = 0, 1, 2, 3      # F32_BF16X3: fp32 storage, bf16x3 products (conv2d / conv2d_wgrad only); F32_SPLIT: [32 hi | 32 lo] bf16 pairs
F32_HSPLIT = 4                                     # the f16x3 forward arithmetic: [32 x f16 hi | 32 x f16 lo * 2^11] activations, row-scaled f16 hi | lo weights
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID = 0, 1, 2, 3
RES_NONE, RES_ADD, RES_RELU_MASK, RES_SWISH_GRAD = 0, 1, 2, 3
TUNE_IGEMM_BIG, TUNE_IGEMM_BIG_MIN_M, TUNE_SPLIT_PERS, TUNE_IGEMM_KORD, TUNE_SPLIT_KORD = 0, 1, 2, 3, 4

_ERR = {-1: 'EFFDET_EINVAL', -2: 'EFFDET_ELAUNCH', -3: 'EFFDET_EUNSUPPORTED'}


class Seg(C.Structure):
    _fields_ = [('H', C.c_int), ('W', C.c_int), ('Ho', C.c_int), ('Wo', C.c_int),
                ('in_off', C.c_longlong), ('in_bstride', C.c_longlong),
                ('out_off', C.c_longlong), ('out_bstride', C.c_longlong)]


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p), ('res', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('rowscale', C.c_void_p),
                ('bc_scale', C.c_void_p), ('bc_shift', C.c_void_p),
                ('dtype', C.c_int), ('out_f32', C.c_int),
                ('B', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('KH', C.c_int), ('KW', C.c_int),
                ('stride', C.c_int), ('pad_t', C.c_int), ('pad_l', C.c_int),
                ('ldx', C.c_int), ('ldy', C.c_int), ('act', C.c_int), ('res_mode', C.c_int),
                ('nseg', C.c_int), ('seg', Seg * MAX_CONV_SEG), ('w_image_stride', C.c_longlong), ('y_split', C.c_void_p), ('range_flag', C.c_void_p),
                ('seg_w', C.c_void_p * MAX_CONV_SEG), ('seg_shift', C.c_void_p * MAX_CONV_SEG)]


class WgradDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('dz', C.c_void_p), ('dw', C.c_void_p), ('dbias', C.c_void_p),
                ('dtype', C.c_int),
                ('B', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('KH', C.c_int), ('KW', C.c_int),
                ('stride', C.c_int), ('pad_t', C.c_int), ('pad_l', C.c_int),
                ('ldx', C.c_int), ('lddz', C.c_int), ('nseg', C.c_int), ('image_splits', C.c_int), ('seg', Seg * MAX_SEG)]


class PrepJob(C.Structure):      # effdet_prep_job_t
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('c', C.c_void_p), ('d', C.c_void_p), ('out', C.c_void_p),
                ('kind', C.c_int), ('dtype', C.c_int), ('n0', C.c_int), ('n1', C.c_int), ('n2', C.c_int), ('n3', C.c_int),
                ('n4', C.c_int), ('eps', C.c_float)]


class UnpackJob(C.Structure):    # effdet_unpack_job_t
    _fields_ = [(n, C.c_void_p) for n in ('g', 'scale', 'w_oihw', 'dw_oihw', 'wsum', 'dsum_part', 'mean', 'invstd', 'dgamma', 'dbeta',
                                          'dbias_out', 'slab_scale', 'slab_cscale')] + \
               [(n, C.c_int) for n in ('accumulate', 'Cout', 'Cin', 'KH', 'KW', 'Cin_pad', 'nslabs', 'slabs_per_scale')]


class SeParamJob(C.Structure):   # effdet_se_param_job_t
    _fields_ = [(n, C.c_void_p) for n in ('du', 'dmid', 'sw', 'pool', 'dw1', 'db1', 'dw2', 'db2')] + \
               [('B', C.c_int), ('C', C.c_int), ('Cse', C.c_int), ('inv_hw', C.c_float)]


class DwUnpackJob(C.Structure):  # effdet_dw_unpack_job_t
    _fields_ = [(n, C.c_void_p) for n in ('g_kkc', 'scale', 'w_c1kk', 'dw_c1kk', 'wsum', 'dsum', 'mean', 'invstd', 'dgamma', 'dbeta')] + \
               [('C', C.c_int), ('kk', C.c_int)]


class _TailUnion(C.Union):
    _fields_ = [('conv', UnpackJob), ('se', SeParamJob), ('dw', DwUnpackJob)]


class TailJob(C.Structure):      # effdet_tail_job_t
    _fields_ = [('kind', C.c_int), ('u', _TailUnion)]


TAIL_UNPACK, TAIL_SE_PARAMS, TAIL_DW_UNPACK = 0, 1, 2

_lib = None

# every symbol include/effdet_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'effdet_conv2d', 'effdet_conv2d_kernel', 'effdet_tuning_set', 'effdet_conv2d_wgrad', 'effdet_conv2d_wgrad_workspace_bytes', 'effdet_conv2d_wgrad_splits', 'effdet_conv2d_wgrad_seg_slabs', 'effdet_conv2d_wgrad_kernel', 'effdet_pack_conv_weight', 'effdet_scale_pack_weight', 'effdet_unpack_conv_wgrad', 'effdet_unpack_conv_wgrad_bn', 'effdet_unpack_conv_wgrad_batch', 'effdet_backward_tail', 'effdet_dw_unpack_wgrad_bn', 'effdet_prepare_params',
    'effdet_bn_fold', 'effdet_bn_param_grad', 'effdet_dw_pack_weight', 'effdet_dw_unpack_wgrad', 'effdet_bifpn_weight_bwd',
    'effdet_dwconv_fwd', 'effdet_dwconv_fwd_pool_groups', 'effdet_mbconv_expand_dw_fwd', 'effdet_mbconv_expand_dw_pool_groups', 'effdet_dwconv_dgrad', 'effdet_dwconv_wgrad', 'effdet_dwconv_wgrad_workspace_bytes', 'effdet_dwconv_bwd', 'effdet_dwconv_bwd_workspace_bytes', 'effdet_pw_bwd', 'effdet_pw_bwd_slabs', 'effdet_pw_dgrad_se', 'effdet_pw_dgrad_se_supported',
    'effdet_se_gate_fwd', 'effdet_se_gate_fwd_split', 'effdet_channel_scale', 'effdet_se_dgate', 'effdet_se_dgate_slabs', 'effdet_se_dgate_from_wgrad', 'effdet_se_gate_bwd', 'effdet_se_gate_bwd_workspace_floats', 'effdet_se_bwd_apply',
    'effdet_act_bwd', 'effdet_add_inplace', 'effdet_colsum', 'effdet_bifpn_fuse_fwd', 'effdet_bifpn_fuse_fwd2', 'effdet_bifpn_fuse_bwd',
    'effdet_anchors', 'effdet_num_anchors', 'effdet_decode_score', 'effdet_nms_workspace_bytes', 'effdet_nms',
    'effdet_gather_dets', 'effdet_loss_workspace_bytes', 'effdet_focal_loss_fwd', 'effdet_focal_loss_bwd', 'effdet_focal_loss_bwd_pix', 'effdet_focal_loss_fwd_grad', 'effdet_focal_loss_bwd_reg',
    'effdet_clip_adamw_step', 'effdet_opt_chunk',
    'effdet_drop_connect_scales', 'effdet_philox4x32_10', 'effdet_preprocess_batch', 'effdet_finalize_dets', 'effdet_head_out_bwd',
    'effdet_nhwc_to_nchw_f32', 'effdet_nchw_f32_to_nhwc', 'effdet_pad_rows', 'effdet_to_split', 'effdet_to_split2', 'effdet_version', 'effdet_abi_version',
]


def lib():
    """Load (once) and return the HIP library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libeffdet_hip.so is missing (%s). Build it with `python -m efficientdet.pytorch_amd.build` '
                '(or __graft_entry__.build()). There is no CPU fallback for this path.' % LIB_PATH)
        cand = C.CDLL(LIB_PATH)
        # a stale build (or a foreign EFFDET_HIP_LIB) with other signatures would be called with shifted arguments: refuse it
        got = int(cand.effdet_abi_version()) if hasattr(cand, 'effdet_abi_version') else 0
        if got != ABI_VERSION:
            raise RuntimeError('%s has ABI generation %d, this binding needs %d: rebuild it (`python -m efficientdet.pytorch_amd.build`)'
                               % (LIB_PATH, got, ABI_VERSION))
        _lib = cand
        _lib.effdet_version.restype = C.c_char_p
        for name in ('effdet_num_anchors', 'effdet_nms_workspace_bytes', 'effdet_loss_workspace_bytes',
                     'effdet_conv2d_wgrad_workspace_bytes', 'effdet_dwconv_wgrad_workspace_bytes', 'effdet_dwconv_bwd_workspace_bytes',
                     'effdet_se_gate_bwd_workspace_floats'):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_longlong
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError('%s failed: %s (%d)' % (what, _ERR.get(status, 'unknown'), status))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (every launch asks: the raw-handle query is ~0.3 us, the
    torch.cuda.current_stream() object ~9 us -- 1.2 ms of host time per eager D0 train step)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError('unsupported storage dtype %s' % t)
