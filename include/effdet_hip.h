/* effdet_hip.h -- C ABI of libeffdet_hip.so: the MI355X (gfx950) EfficientDet hot path.
 *
 * The reference (toandaominh1997/EfficientDet.Pytorch) is pure Python: it has no FFI; its
 * "operator API" is the nn.Module surface of models/efficientdet.py::EfficientDet, and every
 * kernel it runs is a stock torch / torchvision op.  Each entry point below replaces the stock
 * op(s) named in its comment (reference file:line), so that a maintainer can bind them with
 * ctypes from a module that keeps the reference's constructor, forward and
 * state_dict layout.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated;
 *   - returns 0 on success, a negative EFFDET_E* code otherwise; never throws, never allocates,
 *     never synchronises; work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - thread-safe and re-entrant (no global mutable state);
 *   - activations are NHWC ("channels last"), dtype EFFDET_F32 or EFFDET_BF16 (fp32 accumulate);
 *   - tensors that feed losses / NMS (probabilities, box deltas, anchors, boxes) are always fp32.
 */
#ifndef EFFDET_HIP_H
#define EFFDET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* effdet_stream_t; /* hipStream_t */

enum { EFFDET_OK = 0, EFFDET_EINVAL = -1, EFFDET_ELAUNCH = -2, EFFDET_EUNSUPPORTED = -3 };
enum { EFFDET_F32 = 0, EFFDET_BF16 = 1,
       /* effdet_conv2d / effdet_conv2d_wgrad only: fp32 STORAGE (every pointer as for EFFDET_F32), products formed as
        * bf16x3 -- each operand value split in registers into bf16 hi + bf16 lo, hi*hi + hi*lo + lo*hi on
        * v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~16 mantissa bits per product; 3/8 of the matrix-pipe passes
        * of v_mfma_f32_16x16x4_f32).  effdet_conv2d: KH*KW*Cin % 32 == 0 and w packed by effdet_pack_conv_weight /
        * EFFDET_PREP_PACK* with this same dtype (pre-split [32 x hi | 32 x lo] groups, same byte size as fp32). */
       EFFDET_F32_BF16X3 = 2,
       /* effdet_conv2d / effdet_conv2d_wgrad / effdet_to_split only: the SPLIT activation layout of the bf16x3 arithmetic.  4
        * bytes per element like fp32 (same offsets / strides, in elements), but every 128-byte group of a pixel row -- 32
        * channels -- holds [32 x bf16 hi | 32 x bf16 lo] with value = hi + lo (+ O(2^-17 |value|)): channel n at byte
        * (n / 32) * 128 + (n % 32) * 2 (hi) and + 64 (lo).  Written by the epilogue of effdet_conv2d (y of an EFFDET_F32_SPLIT conv
        * unless out_f32), the loss kernels and effdet_to_split; read by effdet_conv2d / effdet_conv2d_wgrad as MFMA operands with
        * no splitting work in the loop.  Channel counts and row pitches are multiples of 32, rows 128-byte aligned.
        * effdet_conv2d: x split; w packed with EFFDET_F32_BF16X3; y split (Cout % 32 == 0; res, if any, is the EFFDET_RES_RELU_MASK
        * activation in the same layout) or, with out_f32, plain fp32 (res, if any, EFFDET_RES_ADD in plain fp32); no z. */
       EFFDET_F32_SPLIT = 3,
       /* effdet_conv2d / effdet_pack_conv_weight / EFFDET_PREP_PACK0 / effdet_to_split2 only: the "f16x3" arithmetic of the FORWARD
        * RetinaHead (models/retinahead.py:109-129) -- fp32-equivalent products at the fp16 matrix rate.  Activations in the H-SPLIT
        * layout: 4 bytes per element and the group geometry of EFFDET_F32_SPLIT, but the halves are IEEE fp16:
        * [32 x f16 hi | 32 x f16 lo'], hi = RNE_f16(v) (fp16 denormals included: the matrix pipe honours them), lo' = RNE_f16((v - hi) * 2^11),
        * i.e. v = hi + lo' * 2^-11 with 22 significand bits for 2^-14 <= |v| < 65504, absolute error <= 2^-36 below (|v| >= 65520 overflows
        * to inf: loud).
        * Weights packed by effdet_pack_conv_weight(dtype = EFFDET_F32_HSPLIT, mode 0): per output channel n the row w[n] * S_n (S_n = the
        * power of two that puts max |w[n]| into [2^14, 2^15)) as 128-byte groups of 32 k: [32 x f16 hi | 32 x f16 lo] (lo unscaled: the row
        * scale keeps it normal), followed by Cout floats 1 / S_n; 4 * Cout * K + 4 * Cout bytes, K = KH * KW * Cin_pad, K % 32 == 0, K >= 256.
        * effdet_conv2d: y = act(conv / S_n + shift[n]) from 3 x v_mfma_f32_16x16x32_f16 per K-step (hi*hi + lo*hi into one fp32
        * accumulator, hi*lo' into a second one that enters as * 2^-11): per product ~2^-22 relative, below the rounding noise of an fp32
        * accumulation.  scale / rowscale / bc_* / res / z
        * / w_image_stride must be unset; y is H-split (Cout % 32 == 0) or, with out_f32, plain fp32; y_split (optional, y H-split) receives
        * the values once more in the bf16 EFFDET_F32_SPLIT layout (operands / ReLU masks of the bf16x3 gradient kernels). */
       EFFDET_F32_HSPLIT = 4 };
enum { EFFDET_ACT_NONE = 0, EFFDET_ACT_RELU = 1, EFFDET_ACT_SWISH = 2, EFFDET_ACT_SIGMOID = 3 };
/* what the `res` tensor of a conv does in the epilogue */
enum { EFFDET_RES_NONE = 0, EFFDET_RES_ADD = 1, EFFDET_RES_RELU_MASK = 2, EFFDET_RES_SWISH_GRAD = 3 };

#define EFFDET_MAX_SEG 5
#define EFFDET_MAX_CONV_SEG 10   /* effdet_conv_t only: 5 pyramid levels x 2 independent convs of the same geometry (the RetinaHead's two towers) */

/* One pyramid level ("segment") of a grouped launch.  A single-tensor conv has nseg = 1.
 * Element (b, h, w, c) of the input lives at  x + in_off  + b*in_bstride  + (h*W  + w )*ldx + c,
 * element (b,ho,wo, n) of the output lives at y + out_off + b*out_bstride + (ho*Wo + wo)*ldy + n
 * (same addressing for z and res).  Offsets/strides are in ELEMENTS. */
typedef struct {
  int H, W, Ho, Wo;
  long long in_off, in_bstride, out_off, out_bstride;
} effdet_seg_t;

/* ---------------------------------------------------------------------------------------------
 * Dense convolution as an MFMA implicit GEMM (1x1 and 3x3, stride 1 or 2, asymmetric zero pad).
 * Replaces F.conv2d(groups=1) [+ ZeroPad2d] [+ frozen BatchNorm2d] [+ bias] [+ ReLU / Swish /
 * sigmoid] [+ residual add] of:
 *   models/utils.py:151-155 (Conv2dStaticSamePadding.forward), models/efficientnet.py:88-89,98,104
 *   (expand / project conv + BN + swish + skip), models/module.py:495-501 (ConvModule.forward:
 *   BiFPN lateral + 3x3 convs, RetinaHead towers), models/retinahead.py:116-123 (retina_cls +
 *   sigmoid, retina_reg).  Also used for the data gradient of stride-1 convs (flipped weights).
 *   y = act( (conv(x,w)) * scale[n] + shift[n] ) [* rowscale[b]] [* bc_scale[b][n] + bc_shift[b][n]] [res op]
 *   z (optional) receives the pre-activation value  conv*scale+shift  (saved for backward).
 * w is packed [Cout][KH*KW][Cin] in `dtype` (effdet_pack_conv_weight).
 * Requirements: Cin % (16/sizeof(dtype)) == 0, ldx % (16/sizeof(dtype)) == 0, x 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; const void* w; void* y; void* z; const void* res;
  const float* scale; const float* shift; /* per output channel, may be NULL (=> 1 / 0) */
  const float* rowscale;                  /* per image [B], may be NULL (drop_connect keep mask / keep_prob) */
  const float* bc_scale; const float* bc_shift; /* both or neither: per (image, output channel) affine [B][Cout], applied after
                                           * rowscale and before the res op:  v = v*bc_scale[b][n] + bc_shift[b][n]
                                           * (squeeze-excite backward fused into the project conv's data gradient) */
  int dtype, out_f32;                     /* out_f32: y is written as fp32 regardless of dtype */
  int B, Cin, Cout, KH, KW, stride, pad_t, pad_l;
  int ldx, ldy;                           /* channel strides (elements) of x rows and y/z/res rows */
  int act, res_mode;
  int nseg;
  effdet_seg_t seg[EFFDET_MAX_CONV_SEG];
  long long w_image_stride; /* 0: one weight tensor for every image.  != 0 (BYTES; one segment, Ho*Wo a multiple of 128, the
                             * implicit-GEMM kernels only): image b reads its OWN packed weights at w + b * w_image_stride -- how the
                             * squeeze-excite gate of an MBConv block is applied without a pass over the activations:
                             * W_b = W * diag(gate_b), packed by effdet_scale_pack_weight (models/efficientnet.py:86-95) */
  void* y_split;            /* optional (dtype EFFDET_F32, no out_f32 / res op, Cout % 32 == 0, ldy / out_off / out_bstride % 32 == 0): the
                             * output values a second time in the EFFDET_F32_SPLIT layout, addressed like y (same ldy / out_off /
                             * out_bstride relative to y_split).  An exact-fp32 forward leaves the operands of split-layout bf16x3
                             * GRADIENT kernels this way: forward values untouched, no conversion pass (models/retinahead.py:109-118) */
  int* range_flag;          /* optional (dtype EFFDET_F32_HSPLIT, y H-split): device int; bit 0 is set (integer atomicOr, only then) when an
                             * output value cannot be held by the H-split layout (|v| >= 65520 or NaN).  The caller owns, resets and reads it. */
  const void* seg_w[EFFDET_MAX_CONV_SEG];      /* optional per-segment operands (NULL entry: the descriptor's w / shift): segments of ONE launch may be */
  const float* seg_shift[EFFDET_MAX_CONV_SEG]; /* independent convs of the same geometry with their own packed weights and bias -- the same layer of
                             * the RetinaHead's classification and regression towers (models/retinahead.py:112-118) as one launch of 2 x 2728 tiles
                             * instead of two of 2728 (5.33 rounds of the 512 workgroup slots each: the draining third of a round is paid once).
                             * Same dtype / packing / byte size as w; plain implicit-GEMM kernels only (no persistent / skinny form, no
                             * w_image_stride); seg_shift needs shift != NULL.  Values are those of the separate launches, bit for bit. */
} effdet_conv_t;
int effdet_conv2d(const effdet_conv_t* p, effdet_stream_t stream);
/* Per-image 1x1 weights for effdet_conv_t.w_image_stride:  out[b][n][k] = w[n][k] * gate[b][k]  in the packed layout of `dtype`
 * (EFFDET_F32: fp32 [Cout][Cin]; EFFDET_F32_BF16X3: the pre-split [32 hi | 32 lo] groups, Cin % 32 == 0; EFFDET_BF16: bf16).
 * w: fp32 [Cout][Cin] (an OIHW 1x1 weight as is), gate: fp32 [B][Cin].  Image stride of `out` = Cout * Cin * (2 for bf16, else 4) bytes. */
int effdet_scale_pack_weight(const float* w, const float* gate, void* out, int dtype, int B, int Cout, int Cin, effdet_stream_t stream);

/* Which kernel effdet_conv2d would launch for this descriptor (no device work): a negative EFFDET_E* code, 0..3 = the
 * implicit-GEMM kernel with a 128 / 64 / 32 / 16-channel block tile, 4..7 = the same tiles in the bf16x3 form
 * (EFFDET_F32_BF16X3), >= 10 = 10 + the persistent big-tile variant. */
int effdet_conv2d_kernel(const effdet_conv_t* p);

/* Kernel-selection knobs (speed only -- every setting computes the same values; process-wide, meant for A/B runs and
 * tests).  Returns the previous value, or EFFDET_EINVAL for an unknown key.
 *   EFFDET_TUNE_IGEMM_BIG       : persistent big-tile bf16 implicit-GEMM (v_mfma_f32_32x32x16_bf16) for Cin % 64 == 0,
 *                                 Cout >= 128 convs: 0 off | 1 = 442 (256x256 tile, 16 waves, 2 LDS stages) | 242 | 243 | 423
 *                                 (waves along pixels, waves along channels, LDS stages)
 *   EFFDET_TUNE_IGEMM_BIG_MIN_M : minimum output pixels per launch for that variant */
enum { EFFDET_TUNE_IGEMM_BIG = 0, EFFDET_TUNE_IGEMM_BIG_MIN_M = 1,
       EFFDET_TUNE_SPLIT_PERS = 2 /* EFFDET_F32_SPLIT convs, persistent 256x256 32x32x16 form for Cout >= 192, long K: 0 off, 1 all, 2 (default) only without a residual epilogue, 3 only with one */,
       EFFDET_TUNE_IGEMM_KORD = 3 /* K walk of the persistent variants: 0 tap-major, 1 channel-group-major */,
       EFFDET_TUNE_SPLIT_KORD = 4 /* K walk of the EFFDET_F32_SPLIT convs: 0 tap-major, 1 channel-group-major */, EFFDET_TUNE_COUNT = 5 };
int effdet_tuning_set(int key, int value);

/* Weight gradient of the same convolution:  dw[n][tap][c] += sum_m dz[m][n] * x[pix(m)+tap][c]
 * (fp32, packed [Cout][KH*KW][Cin]; split-K partial slabs + a reduce pass, so levels / K-splits add up),
 * and optionally dbias[n] += sum_m dz[m][n].  Replaces autograd of F.conv2d w.r.t. weight/bias.
 * No float atomics: every split-K block stores its partial tile (and its partial bias row) with plain stores and the
 * partials are added in slab order, so two runs on the same inputs are BITWISE equal.
 * dbias: with dw != NULL, dbias[n] += the sum.  With dw == NULL (slabs left unreduced) any non-NULL dbias only REQUESTS the
 * per-split partial rows [splits][Cout], stored in `workspace` right behind the slabs (at float offset
 * splits*Cout*KH*KW*Cin) for effdet_unpack_conv_wgrad / _bn to sum; the dbias pointer itself is not written.
 * Segment geometry: in_* addresses x, out_* addresses dz (Ho,Wo rows).
 * dtype EFFDET_F32_BF16X3: fp32 x / dz; the pyramid levels the DMA-staged kernel can take (stride 1, 'same' taps, even
 * row pairs) form their products as bf16x3 (both operands split in registers), the others use the exact fp32 kernel. */
typedef struct {
  const void* x; const void* dz; float* dw; float* dbias;
  int dtype;
  int B, Cin, Cout, KH, KW, stride, pad_t, pad_l;
  int ldx, lddz;
  int nseg;
  int image_splits;   /* != 0 (one level only, Ho*Wo a multiple of the kernel's K-step): split-K boundaries fall on image
                       * boundaries -- effdet_conv2d_wgrad_splits(p) = B * q slabs, slab s holds the partial gradient of image s / q */
  effdet_seg_t seg[EFFDET_MAX_SEG];
} effdet_wgrad_t;
/* workspace: effdet_conv2d_wgrad_workspace_bytes(p) bytes of scratch for the split-K partial slabs + bias partial rows. */
long long effdet_conv2d_wgrad_workspace_bytes(const effdet_wgrad_t* p);
/* number of split-K slabs the launch writes; with p->dw == NULL the slabs are left unreduced in `workspace`
 * ([splits][Cout][KH*KW][Cin] fp32) for effdet_unpack_conv_wgrad(..., nslabs = splits) to sum while unpacking. */
int effdet_conv2d_wgrad_splits(const effdet_wgrad_t* p);
/* Which slabs belong to which segment: first[s], count[s] (arrays of p->nseg ints) = the slab range of segment s in the
 * `workspace` layout of effdet_conv2d_wgrad; returns the total (= effdet_conv2d_wgrad_splits) or an error.  With p->dw == NULL
 * this makes the segments INDEPENDENT problems of one launch: same conv geometry, different x / dz tensors (in_off / out_off are
 * element offsets from p->x / p->dz and may address other allocations) and different weight gradients -- each unpacked from
 * its own slab range.  The 8 nodes of a BiFPN module (64 -> 64 3x3 convs on one level each, 19-89 us apiece as single
 * launches because the small levels are one long dependent chain on 5..30 workgroups) leave as two such launches. */
int effdet_conv2d_wgrad_seg_slabs(const effdet_wgrad_t* p, int* first, int* count);
/* Which kernel serves the descriptor: 0 = the 128 x 128-tile kernels (DMA-staged / register-transpose, per level), 1 = the thin
 * pointwise kernel (one contiguous 1x1 level, Cin + Cout <= 192, >= 32768 pixels: exact fp32 MFMA on linearly staged pixel
 * rows -- also in the EFFDET_F32_BF16X3 mode), 2 = the split-layout kernel. */
int effdet_conv2d_wgrad_kernel(const effdet_wgrad_t* p);
int effdet_conv2d_wgrad(const effdet_wgrad_t* p, void* workspace, long long workspace_bytes, effdet_stream_t stream);

/* OIHW fp32 master weight -> packed [Cout][KH*KW][Kpad] (mode 0, forward; channels >= Cin are
 * zero: the stem pads its 3 image channels to one 16-byte chunk) or the data-gradient operand
 * [Cin][KH*KW flipped][Kpad] (mode 1; Kpad >= Cout, entries >= Cout zero: the head's 9*num_classes /
 * 36-channel gradient maps are padded to whole chunks), optionally multiplied by scale[cout]
 * (frozen-BN fold).  `Cin_pad` is that inner-dimension padding Kpad.  dtype = packed element type. */
int effdet_pack_conv_weight(const float* w_oihw, const float* scale, void* out, int dtype, int mode,
                            int Cout, int Cin, int KH, int KW, int Cin_pad, effdet_stream_t stream);
/* packed fp32 gradient [nslabs][Cout][KH*KW][Cin_pad] (slabs summed) -> OIHW fp32:  dw_oihw (+)= scale[cout] * g.
 * If wsum != NULL also wsum[cout] = sum_{tap,c} w_oihw * g  (needed for the frozen-BN gamma grad).
 * dbias_part / dbias_out (both or neither): the [nslabs][Cout] bias partial rows of effdet_conv2d_wgrad -> dbias_out[cout] =
 * their sum in slab order (overwritten, not accumulated). */
int effdet_unpack_conv_wgrad(const float* g, const float* scale, const float* w_oihw, float* dw_oihw,
                             float* wsum, int accumulate, int Cout, int Cin, int KH, int KW, int Cin_pad, int nslabs,
                             const float* dbias_part, float* dbias_out, const float* slab_scale, int slabs_per_scale,
                             effdet_stream_t stream);
/* slab_scale (optional, with slabs_per_scale | nslabs): slab s (and its bias partial row) is multiplied by
 * slab_scale[s / slabs_per_scale] while summing -- per-image slabs (effdet_wgrad_t.image_splits) x the drop_connect row scale of
 * the image: dW = sum_b rs[b] * M_b without a scaled copy of dz. */
/* The same unpack for a conv followed by a frozen BatchNorm, with effdet_bn_param_grad fused in (one launch instead
 * of two per BN conv): dw = scale*g, dgamma = invstd*(sum_k w*g - mean*dsum), dbeta = dsum, where dsum[cout] is the sum of
 * the [nslabs][Cout] partial rows dsum_part (see effdet_conv2d_wgrad). */
int effdet_unpack_conv_wgrad_bn(const float* g, const float* scale, const float* w_oihw, float* dw_oihw, const float* dsum_part,
                                const float* mean, const float* invstd, float* dgamma, float* dbeta, int Cout, int Cin,
                                int KH, int KW, int Cin_pad, int nslabs, const float* slab_scale, int slabs_per_scale,
                                effdet_stream_t stream);

/* The two calls above as ONE descriptor, and any number of them in one launch (the descriptors travel in the kernel arguments,
 * 24 per launch): a backward node of the model ends with 2 (MBConv) .. 34 (the BiFPN stack) of these few-KiB unpacks, each a
 * chain of dependent slab loads -- batched, the chains overlap.
 * Fields as in effdet_unpack_conv_wgrad / _bn: dgamma != NULL selects the frozen-BN form (dsum_part, mean, invstd, dbeta, w_oihw
 * required), dbias_out != NULL the bias form (dsum_part = the [nslabs][Cout] partial rows), wsum as above; unused pointers NULL. */
typedef struct {
  const float* g; const float* scale; const float* w_oihw; float* dw_oihw; float* wsum;
  const float* dsum_part; const float* mean; const float* invstd; float* dgamma; float* dbeta; float* dbias_out;
  const float* slab_scale;
  const float* slab_cscale; /* optional [nslabs / slabs_per_scale][Cin_pad]: slab s is ALSO multiplied, per input channel c, by
                             * slab_cscale[s / slabs_per_scale][c] -- the squeeze-excite gate of the image when the project conv ran on
                             * per-image weights W diag(gate_b) (effdet_conv_t.w_image_stride): its weight gradient is taken against the
                             * UN-gated activations, M'_b = dy_b^T x_b, and dW = sum_b rs[b] * M'_b * diag(gate_b) */
  int accumulate, Cout, Cin, KH, KW, Cin_pad, nslabs, slabs_per_scale;
} effdet_unpack_job_t;
int effdet_unpack_conv_wgrad_batch(const effdet_unpack_job_t* jobs, int njobs, effdet_stream_t stream);

/* The backward "tail" of a node: every piece of leaf work nothing else in the node waits for, as ONE launch --
 *   EFFDET_TAIL_UNPACK     a weight-gradient unpack (above);
 *   EFFDET_TAIL_SE_PARAMS  phase B of effdet_se_gate_bwd: dw1/db1/dw2/db2 of the squeeze-excite FCs as batch reductions over the
 *                          du [B][C] | dmid [B][Cse] | sw [B][Cse] rows phase A left in its workspace (in that order) and the
 *                          pooled sums `pool` [B][C];
 *   EFFDET_TAIL_DW_UNPACK  effdet_dw_unpack_wgrad / _bn (g_kkc [k*k][C] -> [C][1][k][k], + frozen-BN gamma/beta gradients).
 * An MBConv block's backward ends with four such launches (two conv unpacks, the SE parameters, the depthwise unpack: 8-12 us
 * of dependent-load latency each); as one launch their chains overlap (D0 train step: 48 launches fewer). */
enum { EFFDET_TAIL_UNPACK = 0, EFFDET_TAIL_SE_PARAMS = 1, EFFDET_TAIL_DW_UNPACK = 2 };
typedef struct {
  const float* du; const float* dmid; const float* sw; const float* pool;
  float* dw1; float* db1; float* dw2; float* db2;
  int B, C, Cse; float inv_hw;
} effdet_se_param_job_t;
typedef struct {
  const float* g_kkc; const float* scale; const float* w_c1kk; float* dw_c1kk; float* wsum;
  const float* dsum; const float* mean; const float* invstd; float* dgamma; float* dbeta;
  int C, kk;
} effdet_dw_unpack_job_t;
typedef struct {
  int kind;
  union { effdet_unpack_job_t conv; effdet_se_param_job_t se; effdet_dw_unpack_job_t dw; } u;
} effdet_tail_job_t;
int effdet_backward_tail(const effdet_tail_job_t* jobs, int njobs, effdet_stream_t stream);

/* Batched parameter preparation: every per-step repack of the model's parameters in ONE launch (a D0 train step
 * issued ~190 of these 4-microsecond kernels one by one: 125 weight packs, 48 BN folds, 16 depthwise packs).
 * A job is one of
 *   EFFDET_PREP_PACK0 / _PACK1   effdet_pack_conv_weight mode 0 / 1   (a: OIHW weight; scale = b/sqrt(c+eps) when
 *                                b (gamma) and c (running_var) are given -- the BN fold is recomputed inline so the
 *                                jobs of one launch are independent)
 *   EFFDET_PREP_BNFOLD           effdet_bn_fold      (a gamma, b beta, c mean, d var -> out = scale|shift|invstd, 3*n0)
 *   EFFDET_PREP_DWPACK           effdet_dw_pack_weight (a: [C][1][k][k] -> out [k*k][C] fp32;  n0 = C, n2 = k*k)
 * jobs / block_job / block_first are DEVICE arrays: workgroup i handles elements [256*(i - block_first[j]), +256) of job
 * j = block_job[i] -- except a PACK0 job with dtype EFFDET_F32_HSPLIT, which takes ONE WORKGROUP PER OUTPUT ROW (n0 workgroups: the row's
 * scale is a reduction over the row).  The table is built once per model (parameter storage is stable) and replayed every step. */
enum { EFFDET_PREP_PACK0 = 0, EFFDET_PREP_PACK1 = 1, EFFDET_PREP_BNFOLD = 2, EFFDET_PREP_DWPACK = 3 };
typedef struct {
  const float* a; const float* b; const float* c; const float* d; void* out;
  int kind, dtype;         /* dtype of `out` for the PACK jobs (EFFDET_F32 / EFFDET_BF16 / EFFDET_F32_BF16X3 / EFFDET_F32_HSPLIT) */
  int n0, n1, n2, n3, n4;  /* PACK: Cout, Cin, KH, KW, Kpad;  BNFOLD: C;  DWPACK: C, -, k*k */
  float eps;
} effdet_prep_job_t;
int effdet_prepare_params(const effdet_prep_job_t* jobs, const int* block_job, const int* block_first, int nblocks,
                          effdet_stream_t stream);

/* Frozen (eval-mode) BatchNorm2d as a per-channel affine (models/efficientdet.py:88-92, eps 1e-3):
 *   scale = gamma/sqrt(var+eps), shift = beta - mean*scale, invstd = 1/sqrt(var+eps). */
int effdet_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   float* scale, float* shift, float* invstd, int C, effdet_stream_t stream);
/* Its parameter gradients from the conv weight-gradient by-products (frozen-BN backward):
 *   dgamma = invstd*(wsum - mean*dsum), dbeta = dsum,  wsum[c] = sum_k W[c][k]*G[c][k], dsum = colsum(dz). */
int effdet_bn_param_grad(const float* wsum, const float* dsum, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, int C, effdet_stream_t stream);

/* The stem (models/efficientnet.py:193, 3x3 stride-2 conv on the image + BN + Swish) runs through
 * effdet_conv2d on an NHWC copy of the image whose 3 channels are zero-padded to one 16-byte chunk
 * (effdet_nchw_f32_to_nhwc with Cpad = 8 / 4); its weight gradient through effdet_conv2d_wgrad. */

/* ---------------------------------------------------------------------------------------------
 * Depthwise kxk conv (k = 3 or 5, stride 1 or 2, asymmetric zero pad) + frozen BN + Swish, with
 * the per-(image, channel) sum of the output (the squeeze of squeeze-excite) as a side output.
 * Replaces models/efficientnet.py:91 (_depthwise_conv/_bn1/_swish) and the adaptive_avg_pool2d
 * of :95.  w: [k*k][C] fp32.
 * pool (optional): PARTIAL sums [B][G][C] fp32, G = effdet_dwconv_fwd_pool_groups(...) -- one row per (image, group of
 * output tiles), every entry overwritten with a plain store (no zeroing, no float atomics); effdet_se_gate_fwd adds the G
 * rows of an image in a fixed order, so the pooled sum -- and everything downstream -- is bitwise reproducible.
 * y (Swish output) and z (pre-activation, for backward) are each optional but not both NULL; with
 * y == NULL the pooled sum is that of Swish(stored z), i.e. exactly what the consumers recompute.
 * in_act = EFFDET_ACT_SWISH (forward and weight gradient): x holds the PRE-activation of the producing expand conv (training
 * stores that tensor once: effdet_conv2d with act = NONE) and Swish is applied to the staged input tile; EFFDET_ACT_NONE: x is
 * the activated tensor.
 * ------------------------------------------------------------------------------------------- */
int effdet_dwconv_fwd_pool_groups(int dtype, int B, int C, int stride, int Ho, int Wo);
int effdet_dwconv_fwd(const void* x, const float* w_kkc, const float* scale, const float* shift,
                      void* y, void* z, float* pool, int dtype, int B, int H, int W, int C, int k,
                      int stride, int pad_t, int pad_l, int Ho, int Wo, int in_act, effdet_stream_t stream);
/* Fused expand (1x1 conv + frozen BN + Swish) -> depthwise (k x k, BN, Swish) FORWARD of an MBConv block for inference
 * (models/efficientnet.py:82-88): the 6x-expanded map never touches HBM.  fp32 NHWC; x [B][H][W][Cin], Cin in {16, 24, 32, 40};
 * w_expand [Cexp][Cin] (the OIHW 1x1 weight as is), scale0/shift0 the folded BN0 [Cexp]; w_dw [k*k][Cexp] (effdet_dw_pack_weight),
 * scale1/shift1 the folded BN1; y [B][Ho][Wo][Cexp] = the Swish output; pool (optional) [B][G][Cexp] per-(image, tile group)
 * partial sums for effdet_se_gate_fwd*, G = effdet_mbconv_expand_dw_pool_groups(...). */
int effdet_mbconv_expand_dw_pool_groups(int B, int Cexp, int stride, int Ho, int Wo);
int effdet_mbconv_expand_dw_fwd(const float* x, const float* w_expand, const float* scale0, const float* shift0,
                                const float* w_dw, const float* scale1, const float* shift1, float* y, float* pool,
                                int B, int H, int W, int Cin, int Cexp, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                effdet_stream_t stream);

/* data gradient: dx[b,h,w,c] = sum_taps dz[b,ho,wo,c] * w[tap][c] * scale[c];  optionally
 * multiplied by swish'(zprev) (the expand conv's saved pre-activation) in the epilogue. */
int effdet_dwconv_dgrad(const void* dz, const float* w_kkc, const float* scale, const void* zprev,
                        void* dx, int dtype, int B, int H, int W, int C, int k, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, effdet_stream_t stream);
/* weight gradient g[tap][c] = sum dz*x (unscaled), dsum[c] = sum dz (both OVERWRITTEN).  Every workgroup stores its partial
 * rows into its own slab in `workspace` (effdet_dwconv_wgrad_workspace_bytes) and a reduce pass adds the slabs in a fixed
 * order: no float atomics, two runs are bitwise equal. */
long long effdet_dwconv_wgrad_workspace_bytes(int dtype, int B, int H, int W, int C, int k, int stride, int pad_t,
                                              int pad_l, int Ho, int Wo);
int effdet_dwconv_wgrad(const void* x, const void* dz, float* g_kkc, float* dsum, void* workspace,
                        long long workspace_bytes, int dtype, int B, int H, int W, int C, int k, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, int in_act, effdet_stream_t stream);
/* Data gradient AND weight gradient of the depthwise conv in one pass over dz and zprev (models/efficientnet.py:85-88 backward; replaces
 * the two entry points above when the conv's input was stored as its producer's PRE-activation only, i.e. x = swish(zprev)):
 *   dx = (sum_taps dz * w * scale) * swish'(zprev)      g[tap][c] = sum dz * swish(zprev)(tap)      dsum[c] = sum dz
 * fp32, k = 3.  effdet_dwconv_bwd_workspace_bytes returns 0 when the fused form does not serve the geometry (use the two separate
 * entry points), < 0 on invalid arguments, else the bytes of the slab workspace (one [k*k + 1][C] row set per workgroup, added in a
 * fixed order by a reduce pass: no float atomics).  g_kkc / dsum are OVERWRITTEN. */
long long effdet_dwconv_bwd_workspace_bytes(int dtype, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                                            int Ho, int Wo);
int effdet_dwconv_bwd(const void* dz, const float* w_kkc, const float* scale, const void* zprev, void* dx, float* g_kkc, float* dsum,
                      void* workspace, long long workspace_bytes, int dtype, int B, int H, int W, int C, int k, int stride,
                      int pad_t, int pad_l, int Ho, int Wo, effdet_stream_t stream);
/* Backward of the MBConv expand conv (1x1, Cin -> Cexp = 6 Cin, frozen BN folded; models/efficientnet.py:82-84) in one pass over the
 * expanded gradient dz [M][Cexp] (M = B*H*W pixels, fp32 NHWC, dense rows) -- replaces effdet_conv2d_wgrad + effdet_conv2d for this conv:
 *   dx[M][Cin] = dz (scale[ce] * w_expand[ce][ci]) (+ res[M][Cin], optional: the identity-skip gradient)
 *   slabs[S][Cexp][Cin] = per-workgroup partial sums of dz^T x (UNSCALED),  dsum_part[S][Cexp] = partial column sums of dz
 * with S = effdet_pw_bwd_slabs(M, Cin, Cexp) (0 = the fused form does not serve this geometry: Cin in {16, 24, 32}, M >= 131072).
 * The slabs are summed in slab order by effdet_unpack_conv_wgrad_bn / effdet_backward_tail (nslabs = S): no float atomics.
 * w_expand is the OIHW 1x1 master weight as is. */
int effdet_pw_bwd_slabs(long long M, int Cin, int Cexp);
int effdet_pw_bwd(const float* dz, const float* x, const float* w_expand, const float* scale, const float* res, float* dx,
                  float* slabs, float* dsum_part, long long M, int Cin, int Cexp, effdet_stream_t stream);
/* Data gradient of the MBConv project conv (1x1, Cexp -> Cout, frozen BN2 folded) with the squeeze-excite backward and the depthwise
 * conv's Swish' in its epilogue (models/efficientnet.py:89-104 backward) -- replaces effdet_conv2d(rowscale, bc_scale, bc_shift, res,
 * EFFDET_RES_SWISH_GRAD) for the high-resolution blocks:
 *   dz[M][Cexp] = (rowscale[b] * (dy[M][Cout] (scale[co] * w_project[co][ce])) * gate[b][ce] + dpool[b][ce]) * swish'(zd[M][Cexp])
 * fp32 NHWC, dense rows, M = B * HW pixels; w_project is the OIHW 1x1 master weight [Cout][Cexp] as is; rowscale may be null.
 * effdet_pw_dgrad_se_supported: Cout in {16, 24, 40}, Cexp % 16 == 0, M >= 65536 (else use effdet_conv2d). */
int effdet_pw_dgrad_se_supported(long long M, int Cout, int Cexp);
int effdet_pw_dgrad_se(const float* dy, const float* w_project, const float* scale, const float* rowscale, const float* gate,
                       const float* dpool, const float* zd, float* dz, long long M, int HW, int B, int Cout, int Cexp,
                       effdet_stream_t stream);
/* depthwise weight layout: master [C][1][k][k] fp32 -> [k*k][C];  gradient back:
 * dw[c][t] = scale[c]*g[t][c], wsum[c] = sum_t w[c][t]*g[t][c]. */
int effdet_dw_pack_weight(const float* w_c1kk, float* out_kkc, int C, int k, effdet_stream_t stream);
int effdet_dw_unpack_wgrad(const float* g_kkc, const float* scale, const float* w_c1kk, float* dw_c1kk,
                           float* wsum, int C, int k, effdet_stream_t stream);
int effdet_dw_unpack_wgrad_bn(const float* g_kkc, const float* scale, const float* w_c1kk, float* dw_c1kk, const float* dsum,
                              const float* mean, const float* invstd, float* dgamma, float* dbeta, int C, int k,
                              effdet_stream_t stream);       /* + the BN parameter gradients, as above */

/* ---------------------------------------------------------------------------------------------
 * Squeeze-excite gate:  gate[b][c] = sigmoid(W2 * swish(W1 * (pool[b]/HW) + b1) + b2)
 * Replaces models/efficientnet.py:95-97.  w1: [Cse][C], w2: [C][Cse] (the 1x1 conv weights as
 * stored), fp32.  One workgroup per image, wave-level reductions.  mid (optional) saves the
 * pre-swish squeeze activations [B][Cse] for backward.
 * pool_part: [B][G][C] partial sums as left by effdet_dwconv_fwd (G = 1: a plain [B][C] pool); they are added in a fixed
 * order and the pooled SUM is published in pool_out [B][C] (optional; the gate backward needs it).
 * ------------------------------------------------------------------------------------------- */
int effdet_se_gate_fwd(const float* pool_part, int G, float* pool_out, const float* w1, const float* b1, const float* w2,
                       const float* b2, float* gate, float* mid, int B, int C, int Cse, float inv_hw,
                       effdet_stream_t stream);
/* The same gate for small batches (B <= 16): 8 workgroups per image, one launch per layer (a single workgroup per image is
 * bound by what one CU can pull: 34 us per block for D4 at B = 8).  ws_sw: [B][Cse] floats of scratch.  B > 16 forwards to
 * effdet_se_gate_fwd. */
int effdet_se_gate_fwd_split(const float* pool_part, int G, float* pool_out, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* gate, float* mid, float* ws_sw, int B, int C, int Cse,
                             float inv_hw, effdet_stream_t stream);
/* y = act(x) * gate[b][c]  (models/efficientnet.py:98).  act = EFFDET_ACT_NONE: x is the depthwise OUTPUT;
 * act = EFFDET_ACT_SWISH: x is the depthwise PRE-activation z (training stores z only -- the step is bound by HBM
 * write bandwidth -- and every consumer recomputes Swish from the stored value). */
int effdet_channel_scale(const void* x, const float* gate, void* y, int act, int dtype, int B, long long HW,
                         int C, effdet_stream_t stream);
/* backward of (gate, scale):  given dy (grad of act(x)*gate) and x (act as above):
 *   dgate_part[b][slab][c] = sum over the pixel slab of dy*act(x), slab < effdet_se_dgate_slabs(HW)
 * (every entry overwritten with a plain store; effdet_se_gate_bwd adds the slabs in order: no float atomics)   */
int effdet_se_dgate_slabs(long long HW);
/* The same gradient WITHOUT a pass over the activations, from the per-image partial weight gradients of the project conv
 * (effdet_conv2d_wgrad with image_splits: slabs [B*q][Cout][Cexp], unscaled):
 *   dgate_times_gate[b][c] = rowscale[b] * sum_{i<q} sum_n slabs[b*q+i][n][c] * w_oc[n][c] * bn_scale[n]  = (d loss/d gate)[b][c] * gate[b][c]
 * (w_oc: the project conv's OIHW weight = [Cout][Cexp]; rowscale optional).  Feed it to effdet_se_gate_bwd with dgate_slabs = 1,
 * dgate_times_gate = 1. */
int effdet_se_dgate_from_wgrad(const float* slabs, const float* w_oc, const float* bn_scale, const float* rowscale,
                               float* dgate_times_gate, int B, int q, int Cout, int Cexp, effdet_stream_t stream);
int effdet_se_dgate(const void* dy, const void* x, float* dgate_part, int act, int dtype, int B, long long HW, int C,
                    effdet_stream_t stream);
/* tiny FC backward: from dgate_part[b][slab][c] (partial grads wrt gate, dgate_slabs rows per image; 1 = a plain
 * [B][C] gradient), gate, mid, pool -> dpool[b][c] (grad wrt the
 * SUM pool, i.e. already multiplied by inv_hw), dw1, db1, dw2, db2 (OVERWRITTEN, fp32; batch reductions with one
 * thread per parameter: no atomics, deterministic).  workspace: effdet_se_gate_bwd_workspace_floats() fp32, left holding
 * du [B][C] | dmid [B][Cse] | sw [B][Cse].  With dw1 = db1 = dw2 = db2 = NULL only dpool is produced and the parameter
 * gradients are left to an EFFDET_TAIL_SE_PARAMS job of effdet_backward_tail over that workspace. */
long long effdet_se_gate_bwd_workspace_floats(int B, int C, int Cse);
int effdet_se_gate_bwd(const float* dgate_part, int dgate_slabs, int dgate_times_gate, const float* gate, const float* mid, const float* pool,
                       const float* w1, const float* b1, const float* w2, float* dpool, float* dw1, float* db1,
                       float* dw2, float* db2, float* workspace, int B, int C, int Cse, float inv_hw,
                       effdet_stream_t stream);
/* dx = (dy*gate[b][c] + dpool[b][c]) * swish'(z)   -- gradient wrt the depthwise pre-activation z */
int effdet_se_bwd_apply(const void* dy, const float* gate, const float* dpool, const void* z, void* dzout,
                        int dtype, int B, long long HW, int C, effdet_stream_t stream);

/* elementwise: dz = dy * act'(aux)  (act = RELU: aux = y;  SWISH: aux = z) [* rowscale[b]] */
int effdet_act_bwd(const void* dy, const void* aux, const float* rowscale, void* dz, int dtype, int act,
                   int B, long long HWC, effdet_stream_t stream);
/* y (+)= x   (gradient accumulation across branches) */
int effdet_add_inplace(void* y, const void* x, int dtype, long long n, effdet_stream_t stream);
/* per-channel column sum: out[c] += sum_rows x[row][c]  (one workgroup per 64 columns, fixed order) */
int effdet_colsum(const void* x, float* out, int dtype, long long rows, int C, int ldx,
                  effdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BiFPN fast-normalised fusion nodes (models/bifpn.py:177-202).  Weights arrive raw (w1 [2][L],
 * w2 [3][L-2]); the double normalisation relu -> /(sum+eps) -> /(wa+wb+eps) is done on device.
 *   mode 0 (top-down):   out = (a*wa + up2(b)*wb)/(wa+wb+eps)          b is at half resolution
 *   mode 1 (bottom-up3): out = (a*wa + maxpool2(b)*wb + c*wc)/(sum+eps) b is at double resolution
 *   mode 2 (bottom-up2): out = (a*wa + maxpool2(b)*wb)/(sum+eps)
 * wraw: pointer to the full raw weight matrix (fp32), wrows rows x wcols columns; col selects the node.
 * ------------------------------------------------------------------------------------------- */
int effdet_bifpn_fuse_fwd(const void* a, const void* b, const void* c, void* out, const float* wraw,
                          int wrows, int wcols, int col, int mode, int dtype, int B, int H, int W, int C,
                          effdet_stream_t stream);
/* The same node writing the fused map in up to two forms: `out` (plain, may be NULL) and `out_hsplit` (EFFDET_F32_HSPLIT, may be NULL;
 * fp32 only, C % 32 == 0, 128-byte aligned) -- the operand of the node's 3x3 conv (models/bifpn.py:189-202, ConvModule) when it runs in
 * the f16x3 arithmetic; training keeps the plain form as well (operand of the conv's weight gradient). */
int effdet_bifpn_fuse_fwd2(const void* a, const void* b, const void* c, void* out, void* out_hsplit, const float* wraw,
                           int wrows, int wcols, int col, int mode, int dtype, int B, int H, int W, int C, int* range_flag,
                           effdet_stream_t stream);
/* backward: given dout -> da, db, dc (each overwritten, or += when *_accum), and the partial sums of
 * d loss / d n_r (grad wrt the ONCE-normalised weights), one row per workgroup and NO float atomics:
 *   dn[col * EFFDET_FUSE_COL_FLOATS]                    = number of workgroups of this node's launch
 *   dn[col * EFFDET_FUSE_COL_FLOATS + 4 + 3*wg + r]     = workgroup wg's partial of d loss / d n_r
 * dn is an fp32 scratch of wcols * EFFDET_FUSE_COL_FLOATS floats, zeroed by the caller (a column whose node is never
 * launched then contributes nothing).  effdet_bifpn_weight_bwd adds each column's rows in a fixed order and maps
 * dn -> dwraw (+=) through the first normalisation: two runs are bitwise equal. */
#define EFFDET_FUSE_MAX_WG 2048
#define EFFDET_FUSE_COL_FLOATS (4 + 3 * EFFDET_FUSE_MAX_WG)
int effdet_bifpn_fuse_bwd(const void* dout, const void* a, const void* b, const void* c, void* da, void* db,
                          void* dc, int da_accum, int db_accum, int dc_accum, const float* wraw, float* dn,
                          int wrows, int wcols, int col, int mode, int dtype, int B, int H, int W, int C,
                          effdet_stream_t stream);
int effdet_bifpn_weight_bwd(const float* wraw, const float* dn, float* dwraw, int wrows, int wcols,
                            effdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Anchors (models/module.py:145-214,252-273): float64 arithmetic on device, cast to fp32;
 * bit-exact with the reference's NumPy float64 -> float32 path.  out: [A][4].
 * ------------------------------------------------------------------------------------------- */
int effdet_anchors(float* out, int H, int W, effdet_stream_t stream);
long long effdet_num_anchors(int H, int W);

/* Box decode + clip (models/module.py:24-49, 57-67) fused with the per-anchor class max and the
 * threshold test of models/efficientdet.py:72-73.  boxes [B][A][4], score [B][A], label [B][A]. */
int effdet_decode_score(const float* anchors, const float* reg, const float* cls, float* boxes,
                        float* score, int* label, int B, long long A, int num_classes, float img_w,
                        float img_h, effdet_stream_t stream);

/* Per-image threshold + stable sort by descending score + greedy class-agnostic NMS
 * (models/efficientdet.py:73-86 and torchvision.ops.nms semantics: suppress IoU > thr).
 * Everything stays on the device; no host round trip.
 * Outputs (per image, capacity A each): out_idx [B][A] (indices into the anchor list, in kept
 * order), out_count [B].  workspace: effdet_nms_workspace_bytes(B, A) bytes = per candidate (B*A of them) 2 x 4 B sort
 * keys + 2 x 4 B indices + 4 B flag + 3 x 16 B boxes (sorted, kept, hash overflow), the radix sort's digit histograms
 * (B x 256 x ceil(A / 2048) x 4 B), a round's survivor list and suppression bit-matrix (round = 2048 candidates, 4096 above 64 k
 * anchors: B x round^2 / 8 bytes), plus the kept-box spatial hash: HT = 2^k >= max(1024, A/2) slots per image x (4 B counter +
 * 8 x 16 B entries) -- 134 MB for D0 at B = 32 (A = 49 104) and for D4 @1024 at B = 8 (A = 196 416).
 * The sort is in-tree (stable LSD radix, 4 x 8 bits, per image); the call enqueues KERNELS only (no memset / memcpy nodes, launch
 * geometry a function of B and A alone), so it can be captured into a hipGraph and replayed (graph.GraphedDetect does). */
long long effdet_nms_workspace_bytes(int B, long long A);
int effdet_nms(const float* boxes, const float* score, float threshold, float iou_threshold, int* out_idx,
               int* out_count, void* workspace, long long workspace_bytes, int B, long long A,
               effdet_stream_t stream);
/* gather kept rows: scores [B][A], labels [B][A] (int64), boxes [B][A][4] for the first count[b] rows */
int effdet_gather_dets(const float* boxes, const float* score, const int* label, const int* idx,
                       const int* count, float* out_scores, long long* out_labels, float* out_boxes, int B,
                       long long A, effdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Focal + smooth-L1 loss, forward and backward in one pass (models/losses.py:32-152): per-image
 * IoU assignment, focal BCE (alpha .25, gamma 2), smooth-L1 (beta 1/9); no per-image host loop,
 * no syncs.  cls = probabilities [B][A][nc] fp32, reg [B][A][4] fp32, anchors [A][4],
 * annots [B][N][5] (pad rows label = -1).  Outputs: losses[2] (batch-mean cls, reg), and the
 * gradients wrt the LOGITS of cls (dcls_logit, written in `dtype`) and wrt reg (dreg, `dtype`),
 * already scaled by gscale[0] (cls) / gscale[1] (reg) = the upstream grads of the two losses.
 * workspace: effdet_loss_workspace_bytes(B, A, num_classes) (anchor assignment, per-image statistics and the per-workgroup
 * partial sums of both loss terms: they are added in a fixed order, so the losses are bitwise reproducible).
 * ------------------------------------------------------------------------------------------- */
long long effdet_loss_workspace_bytes(int B, long long A, int num_classes);
int effdet_focal_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* annots,
                          float* losses, void* workspace, long long workspace_bytes, int B, long long A,
                          int num_classes, int N, effdet_stream_t stream);
int effdet_focal_loss_bwd(const float* cls, const float* reg, const float* anchors, const float* annots,
                          const float* gscale, const void* workspace, void* dcls_logit, void* dreg, int dtype,
                          int B, long long A, int num_classes, int N, effdet_stream_t stream);
/* The same, with d(cls logits) written PIXEL-major with a padded channel pitch: dcls_pix[b][pixel][dld], channel =
 * anchor*num_classes + class, zeros in [9*num_classes, dld) -- directly the (cache-line aligned when dld % 64 == 0) input
 * rows of the head's data-gradient / weight-gradient convs.  Requires num_classes % 4 == 0, dld % 4 == 0, A % 9 == 0. */
int effdet_focal_loss_bwd_pix(const float* cls, const float* reg, const float* anchors, const float* annots,
                              const float* gscale, const void* workspace, void* dcls_pix, int dld, void* dreg, int dtype,
                              int B, long long A, int num_classes, int N, effdet_stream_t stream);
/* Training fast path: forward losses AND d(loss)/d(cls logits) in ONE pass over cls.  dcls_pix (layout as above) holds
 * the gradient for an upstream gradient of 1 (gscale[0] = 1); the caller applies the real upstream scalar downstream,
 * where the chain is linear (effdet_conv2d rowscale on the data gradient, a scalar on the retina_cls parameter
 * gradients).  effdet_focal_loss_bwd_reg then produces d(reg) (scaled by gscale[1]) without touching cls. */
int effdet_focal_loss_fwd_grad(const float* cls, const float* reg, const float* anchors, const float* annots,
                               float* losses, void* workspace, long long workspace_bytes, void* dcls_pix, int dld,
                               int dtype, int B, long long A, int num_classes, int N, effdet_stream_t stream);
/* dtype EFFDET_F32_SPLIT (fwd_grad / bwd_reg): the gradient rows are written in the split layout (dld / reg_ld % 32 == 0, 128-byte
 * aligned buffers) -- ready-made MFMA operands of the head's gradient convs in the bf16x3 arithmetic.
 * reg_ld != 0: d(reg) is written PIXEL-major like dcls_pix -- dreg[b][pixel][reg_ld], channel = anchor*4 + k, zeros in [36, reg_ld);
 * reg_ld == 0: [B][A][4]. */
int effdet_focal_loss_bwd_reg(const float* reg, const float* anchors, const float* annots, const float* gscale,
                              const void* workspace, void* dreg, int reg_ld, int dtype, int B, long long A, int N,
                              effdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Train-step tail: torch.nn.utils.clip_grad_norm_(params, max_norm) followed by
 * torch.optim.AdamW.step() (reference train.py:115-118) as three launches over a device-resident pointer table.
 *   params / grads / exp_avg / exp_avg_sq : device arrays of ntensors DEVICE POINTERS (as 64-bit integers) to fp32
 *                                           tensors; grads[i] == 0 skips tensor i (parameter without gradient)
 *   numel[i]                              : elements of tensor i
 *   block_tensor / block_first            : workgroup i handles elements [(i - block_first[t]) * effdet_opt_chunk(), +chunk)
 *                                           of tensor t = block_tensor[i]      (device arrays, nblocks / ntensors ints)
 *   scratch                               : 64 + nblocks floats; scratch[0] receives the total gradient norm
 *   steps                                 : ntensors ints, zero before the first call: per-tensor AdamW step counters
 *                                           (advanced on device for every tensor that has a gradient, as torch does)
 * max_norm <= 0 disables clipping.  write_grad != 0 writes the clipped gradients back like clip_grad_norm_ does (costs
 * one more store per element).
 * hyper_dev (optional, DEVICE, 6 floats {max_norm, lr, beta1, beta2, eps, weight_decay}): when given, the update kernel
 * reads the hyper-parameters from it at RUN time instead of the by-value arguments, so that a captured step (hipGraph
 * replay) follows a learning-rate schedule (train.py:133,269 drives ReduceLROnPlateau every epoch).  Whether the norm pass
 * runs at all is still decided by the by-value max_norm (> 0) at launch / capture time.
 * ------------------------------------------------------------------------------------------- */
int effdet_opt_chunk(void);
int effdet_clip_adamw_step(const unsigned long long* params, const unsigned long long* grads,
                           const unsigned long long* exp_avg, const unsigned long long* exp_avg_sq, const long long* numel,
                           const int* block_tensor, const int* block_first, int ntensors, int nblocks, float* scratch,
                           int* steps, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int write_grad, const float* hyper_dev, effdet_stream_t stream);

/* Row repack with zero channel padding: dst[b][pix][0..Cpad) = src[src_off + b*src_bstride + pix*src_ld + c]
 * for c < C, 0 beyond (makes an unaligned-channel gradient map consumable by effdet_conv2d). */
int effdet_pad_rows(const void* src, void* dst, int dtype, long long src_off, long long src_bstride, int src_ld,
                    int B, int HW, int C, int Cpad, effdet_stream_t stream);

/* plain fp32 -> split layout (EFFDET_F32_SPLIT), elementwise over n elements (n % 4 == 0; rows are whole 32-channel groups and
 * both buffers 128-byte aligned, so element i of src lands in the group of element i of dst).  Out of place. */
int effdet_to_split(const float* src, void* dst, long long n, effdet_stream_t stream);
/* The same pass writing up to two layouts of the same values: dst_split (EFFDET_F32_SPLIT, bf16 halves; may be NULL) and dst_hsplit
 * (EFFDET_F32_HSPLIT, fp16 hi + scaled lo; may be NULL) -- the BiFPN pyramid entering the RetinaHead once as the operand of the f16x3
 * forward convs and once as the operand of the bf16x3 weight gradients (models/retinahead.py:109-113).  Same requirements as above.
 * range_flag (may be NULL): see effdet_conv_t.range_flag -- set when a value does not fit the H-split layout. */
int effdet_to_split2(const float* src, void* dst_split, void* dst_hsplit, long long n, int* range_flag, effdet_stream_t stream);

/* NCHW fp32 <-> NHWC dtype conversions for the module boundary (feature maps returned by extract_feat) */
int effdet_nhwc_to_nchw_f32(const void* x, float* y, int dtype, int B, int H, int W, int C, effdet_stream_t stream);
/* Cpad >= C: channels C..Cpad-1 of the NHWC output are written as zeros */
int effdet_nchw_f32_to_nhwc(const float* x, void* y, int dtype, int B, int H, int W, int C, int Cpad, effdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Boundary kernels either side of the conv path (csrc/pipeline.hip).
 *
 * drop_connect (models/utils.py:79-90; one draw per identity-skip MBConv block, models/efficientnet.py:98-101):
 *   out[slot][b] = floor(keep_prob[slot] + u) / keep_prob[slot],  u = (x >> 8) * 2^-24 with x = word 0 of
 *   Philox4x32-10(counter = {b, slot, step_lo, step_hi}, key = {seed_lo, seed_hi}).  One launch per step for all slots;
 *   the result is the `rowscale` of the block's project conv (forward) and of its dz (backward).
 *   step_dev != NULL: the step number is read from that device word and advanced by the kernel (a captured hipGraph
 *   then draws fresh masks on every replay); NULL: `step` is used.
 *   effdet_philox4x32_10 is the HOST twin of the generator (pure function, no device work; tests pin the stream on it).
 * ------------------------------------------------------------------------------------------- */
int effdet_drop_connect_scales(float* out, const float* keep_prob, int nslot, int B, unsigned long long seed,
                               unsigned long long step, unsigned long long* step_dev, effdet_stream_t stream);
void effdet_philox4x32_10(const unsigned ctr[4], const unsigned key[2], unsigned out[4]);

/* Device-side input pipeline (datasets/augmentation.py:69-150: Normalizer -> Augmenter -> Resizer -> collater):
 *   src      : uint8 RGB HWC images of mixed sizes, concatenated; image b starts at byte src_off[b], size src_hw[2b], [2b+1]
 *   flip     : optional [B] bytes, non-zero = horizontal flip (Augmenter)
 *   out_nhwc : [B][S][S][Cpad] in `dtype`: cv2.resize(INTER_LINEAR) so that the longer side is S, (v/255 - mean)/std,
 *              zeros outside the resized region and in channels 3..Cpad-1  (what the stem conv reads: no NCHW round trip)
 *   scale_out: optional [B] resize factors (eval.py:105 divides boxes by it)
 *   annots   : optional [B][max_annots][5] fp32, transformed IN PLACE (flip, then * scale); rows with label -1 untouched
 * mean / std are HOST arrays of 3 floats. */
int effdet_preprocess_batch(const unsigned char* src, const long long* src_off, const int* src_hw, const unsigned char* flip,
                            void* out_nhwc, float* scale_out, float* annots, int max_annots, int dtype, int B, int S,
                            int Cpad, const float mean[3], const float std[3], effdet_stream_t stream);

/* Batched evaluation consumer (eval.py:96-127 and :279-306) over effdet_gather_dets' score-descending rows:
 *   out[b][k] = (x1, y1, x2, y2, score, label) / scale[b] on the boxes, for the first out_count[b] =
 *   min(max_det, #{score > score_threshold}) detections of image b; xywh != 0 emits (x, y, w, h) (MS COCO);
 *   rows beyond out_count[b] are zero with label -1.  out: [B][max_det][6] fp32. */
int effdet_finalize_dets(const float* score, const long long* label, const float* boxes, const int* count, const float* scale,
                         float score_threshold, int max_det, int xywh, float* out, int* out_count, int B, long long A,
                         effdet_stream_t stream);

/* Gradient of the head outputs (models/retinahead.py:119-127 under autograd):  dlogit = dprob * p * (1 - p) and dreg,
 * both stored in `dtype` for the head's data-gradient convs.  ncls / nreg: element counts. */
int effdet_head_out_bwd(const float* dprob, const float* prob, const float* dreg, void* dlogit, void* dreg_out, int dtype,
                        long long ncls, long long nreg, effdet_stream_t stream);

/* library identification: returns "effdet-hip gfx950 <version>" */
const char* effdet_version(void);

/* ABI generation of this header: bumped whenever an entry point's signature or a descriptor struct's layout changes.  A binding
 * compares effdet_abi_version() of the library it loaded with the EFFDET_ABI_VERSION it was written against and refuses a
 * mismatch (a stale .so called through ctypes / cgo with shifted arguments reads garbage instead of failing). */
#define EFFDET_ABI_VERSION 10
int effdet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EFFDET_HIP_H */
