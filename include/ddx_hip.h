/*
 * ddx_hip.h -- C ABI of libddx_hip.so: the MI355X (gfx950) kernels behind the EDM2-UNet denoising /
 * mel-latent hot path of dualdiffusion.
 *
 * Conventions (precedent for a native boundary in the reference:
 * src/training/module_trainers/dae_trainer_m1.py:237-267):
 *   - every entry point returns 0 (DDX_OK) or a negative ddx_status; nothing allocates, nothing
 *     synchronises, all buffers are caller-owned device memory, all work is enqueued on `stream`
 *     (a hipStream_t passed as void*);
 *   - activations are NHWC ("channels_last", the reference's physical layout,
 *     src/pipelines/dual_diffusion_pipeline.py:235) in DDX_F32 or DDX_BF16; accumulation is fp32;
 *   - descriptors are plain C structs of pointers and sizes; no torch types anywhere.
 *
 * Each entry point names the reference code (under /root/reference/src) it replaces.
 */
#ifndef DDX_HIP_H
#define DDX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ddx_stream; /* hipStream_t */

enum ddx_dtype { DDX_F32 = 0, DDX_BF16 = 1 };
enum ddx_status {
  DDX_OK = 0,
  DDX_ERR_ARG = -1,         /* malformed descriptor */
  DDX_ERR_UNSUPPORTED = -2, /* shape / dtype combination not built */
  DDX_ERR_LAUNCH = -3,      /* HIP launch failure (hipGetLastError) */
  DDX_ERR_NO_DEVICE = -4
};

/* Library identity: version string and the gfx arch the kernels were compiled for. */
const char* ddx_version(void);
const char* ddx_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Weight preparation  (modules/mp_tools.py:359-364 MPConv.forward weight path, :375-378 normalize_weights)
 *   w' = [normalize(w)] * gain / sqrt(fan_in), cast, re-laid out for the implicit-GEMM kernel:
 *   wp[g][chunk][tap][NgP][CK],  Ng = Cout/groups, NgP = roundup(Ng,32), chunk = c/CK within the group,
 *   zero padded.  gain_eff = gain * (gain_ptr ? *gain_ptr : 1).
 *   qk_head_dim > 0 re-orders output rows (head, d, {q,k}) -> (head, {q,k}, d)  (unet_edm2_b4.py:137-139).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* w;      /* master weights [Cout][Cg][ks][ks] */
  void* wp;           /* prepared weights, ddx_wprep_bytes() bytes */
  const float* gain_ptr;
  float gain;
  int32_t w_dtype, wp_dtype;
  int32_t Cout, Cg, ksize, groups;
  int32_t CK;         /* K chunk of the consumer kernel: 32 or 64 */
  int32_t normalize;  /* 1 = forced weight norm inside forward (module.training) */
  int32_t qk_head_dim;
  /* per-source input scaling folded into the weights (mp_cat feeding a conv without a non-linearity, i.e. conv_skip of
   * decoder blocks): input channels < in_split are scaled by in_scale0, the rest by in_scale1.  in_split <= 0: none. */
  int32_t in_split;
  float in_scale0, in_scale1;
  /* transpose = 1: emit the weights of the DATA-GRADIENT conv instead (backward of F.conv2d w.r.t. its input):
   *   dX = conv(dY, wp_t) with wp_t[g][c][n][tap] = w'[g*Ng+n][c][flipped tap], i.e. a conv with Cout' = groups*Cg output
   *   channels and Cg' = Ng input channels per group, consumed by ddx_mpconv2d_fwd like any other prepared weight
   *   (size ddx_wprep_bytes(groups*Cg, Ng, ksize, groups, CK, dtype)).  row_scale: [Cout] fp32 workspace (caller-owned). */
  int32_t transpose;
  float* row_scale;
  /* rows_total > 0: this weight fills rows [row_offset, row_offset + Cout) of a prepared matrix with rows_total rows
   * (ddx_wprep_bytes(rows_total, ...); zero-filled by the caller once) -- several 1x1 convs on the same input run as ONE
   * conv (attn_qk | attn_v, unet_edm2_b4.py:139-140).  groups = 1, transpose = 0. */
  int32_t row_offset, rows_total;
} ddx_wprep_desc;

size_t ddx_wprep_bytes(int32_t Cout, int32_t Cg, int32_t ksize, int32_t groups, int32_t CK, int32_t dtype);
int ddx_mpconv_wprep(const ddx_wprep_desc* d, ddx_stream stream);
/* In-place forced weight normalisation of master fp32 weights (mp_tools.py:375-378). rows = Cout. */
int ddx_normalize_weights(void* w, int32_t w_dtype, int64_t rows, int64_t fan_in, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Magnitude-preserving conv2d forward  (mp_tools.py:366-373 F.conv2d, stride 1, same padding; ksize 1|3)
 * with the neighbouring element-wise work of Block.forward (unet_edm2_b4.py:110-158) fused in:
 *   input   = up to two NHWC sources concatenated on channels with scales (mp_cat, mp_tools.py:294-301),
 *             optionally 2x nearest-upsampled or 2x2 average-pooled on the fly (mp_tools.py:71-79)
 *   prologue: bit0 = mp_silu (mp_tools.py:268), bit1 = multiply by chan_scale[b][cin]  (y*c, unet_edm2_b4.py:122)
 *             (scale applied first, then silu)
 *   epilogue: 0 = store; 1 = mp_sum(residual, y, t) (mp_tools.py:274-279); then clip to +-clip if clip > 0
 * ------------------------------------------------------------------------------------------------ */
enum { DDX_RESAMPLE_KEEP = 0, DDX_RESAMPLE_UP = 1, DDX_RESAMPLE_DOWN = 2,
       /* adjoints, ddx_resample2d only: gradient of UP (2x2 sums) and of DOWN (nearest, x 1/4) */
       DDX_RESAMPLE_UP_BWD = 3, DDX_RESAMPLE_DOWN_BWD = 4 };
enum { DDX_PRO_NONE = 0, DDX_PRO_SILU = 1, DDX_PRO_SCALE = 2, DDX_PRO_SCALE_SILU = 3 };
enum { DDX_EPI_STORE = 0, DDX_EPI_MPSUM = 1, DDX_EPI_SILU_BWD = 2 /* internal: ddx_mpconv2d_dgrad_act */,
       /* out = normalize(y, dim = channels) (mp_tools.py:42-49: y / (eps + rms_c(y)), eps = res_t) on the fp32 accumulators, then clip /
        * out2 as usual: the pixel norm that follows the skip conv of an encoder block (unet_edm2_ddec_mclt_b1.py:107-109) without its
        * own pass.  LDS-DMA kernel only, one group, all output channels in ONE channel tile (Cout <= 64; <= 32 when Cout <= 32) or in one
        * unit of a wide 1x1 layer (192 <= Cout <= 512, plain bf16 operands; the waves of a pixel row exchange their partial sums of
        * squares through LDS): DDX_ERR_UNSUPPORTED otherwise. */
       DDX_EPI_PIXELNORM = 3 };
enum { DDX_PAD_ZERO = 0, DDX_PAD_REFLECT_W = 1,
       /* flag, OR-ed in: src1 is read from the pair-swapped image (index b ^ 1; B even) -- the second depth tap of a (2,k,k)
        * MPConv3D on a stereo pair folded into the batch (modules/daes/dae_edm2_d3.py:62-84: the reflected depth row of a depth
        * of two IS the other slice), without materialising the swapped copy */
       DDX_PAD_SWAP_SRC1 = 2,
       /* flag: the input is the 4-part concatenation [src0 | src1 | src0' | src1'] (' = image b ^ 1), 2 * (C0 + C1) channels: both
        * depth taps of a (2,k,k) MPConv3D over an mp_cat operand that is never materialised */
       DDX_PAD_SWAP_PAIRED = 4 };

enum { DDX_LAYOUT_SRC0_C16 = 1, DDX_LAYOUT_SRC1_C16 = 2, DDX_LAYOUT_OUT_C16 = 4, DDX_LAYOUT_OUT2_C16 = 8 };

typedef struct {
  const void* src0;         /* NHWC [B][sH][sW][C0] */
  const void* src1;         /* NHWC [B][sH][sW][C1] or NULL */
  const float* chan_scale;  /* [B][C0+C1] fp32 or NULL */
  const void* wp;           /* prepared weights (ddx_mpconv_wprep with the same CK) */
  const void* residual;     /* NHWC [B][H][W][Cout] or NULL */
  void* out;                /* NHWC [B][H][W][Cout] */
  int32_t B, H, W;          /* OUTPUT spatial size */
  int32_t C0, C1, Cout, groups, ksize;
  int32_t CK;               /* must equal the wprep CK */
  int32_t resample, prologue, epilogue;
  float scale0, scale1;     /* mp_cat weights (1,1 when unused) */
  float res_t;              /* t of mp_sum(residual, y, t) */
  float clip;               /* <= 0: none */
  int32_t dtype;            /* activations and wp */
  int32_t force_direct;     /* kernel choice: 0 automatic, 1 scalar reference kernel, 2 register-staged MFMA, 3 LDS-DMA MFMA,
                             * 4 small-M weight-streaming MFMA (conv_sm.hip; needs wp prepared with CK = 16 -- with CK = 16 it is
                             * also the automatic choice, the layout is its own),
                             * 5 the few-input-channel kernel (conv_few.hip: 3x3 over 8 zero-padded channels, plain store + twin),
                             * 6 the mid-size 1x1 GEMM kernel (conv_gemm.hip: raw operands, one group, plain store; 128 x 128 tiles),
                             * 16 + 3 * tile + k: register-staged MFMA with tile 0..3 = 256x64, 256x32, 128x64, 128x32 (pixels x
                             * channels) and split-K 1 << k (k = 0..2); DDX_ERR_UNSUPPORTED when the combination is not built.
                             * Used by plan-time autotuning (the host times the candidates once per layer shape). */
  /* producer-side activation (so that the CONSUMER conv needs no prologue and stages its operand untouched):
   *   out_act = 1: the stored output is mp_silu(y * out_scale[b][cout]) (out_scale NULL: mp_silu(y))  -- conv_res0 feeding
   *                conv_res1 (unet_edm2_b4.py:119-122);
   *   out2 != NULL: additionally store out2 = mp_silu(out2_scale * y) of the final (post mp_sum / clip) value -- the
   *                activated twin of a block output that the next block's conv_res0 reads (unet_edm2_b4.py:119).
   *                With out_act = 0 and out_scale != NULL the channel scale belongs to the twin: out = y (raw) and
   *                out2 = mp_silu(out2_scale * out_scale[b][cout] * y) -- the training forward keeps both. */
  const float* out_scale;   /* [B][Cout] fp32 or NULL */
  void* out2;               /* NHWC [B][H][W][Cout] or NULL */
  int32_t out_act;
  float out2_scale;
  /* bit 0: DDX_PAD_ZERO (F.conv2d padding) | DDX_PAD_REFLECT_W: zero rows above / below, mirrored columns left / right -- the
   * ReflectionPad3d((k/2, k/2, 0, 0, ...)) + conv3d(padding=(0, k/2, 0)) of MPConv3D (modules/daes/dae_edm2_d3.py:62-84). */
  int32_t pad_mode;
  /* > 0: the prologue (chan_scale / mp_silu) only applies to output channels below prologue_rows; the rest read the raw
   * input (merged attn_qk | attn_v conv: qk = conv(x * c_qk), v = conv(x)).  Must be a multiple of 64. */
  int32_t prologue_rows;
  /* operand twins of the attention block (small-M kernel conv_sm.hip; where noted also other kernels):
   *   out2_linear = 1: out2 = y_final * out2_chan_scale[b][cout] (no activation) -- the scaled twin x * c_qk that attn_qk reads
   *                (unet_edm2_b4.py:131-133), written by the conv that produces x (also the register-staged MFMA kernel);
   *   src0_alt != NULL (with prologue == NONE and prologue_rows > 0): output channels below prologue_rows read src0_alt instead
   *                of src0 -- merged attn_qk | attn_v conv over [x * c_qk | x] without a per-element prologue.  Also served by the
   *                wide 1x1 units of the LDS-DMA kernel (one group, no src1, prologue_rows a multiple of 256: a unit's 256 output
   *                channels read ONE source); DDX_ERR_UNSUPPORTED on the other kernels. */
  int32_t out2_linear;
  /* Channel-blocked tensors (LDS-DMA kernel, bf16 inference plans): a flagged tensor is stored [B][C/16][H][W][16] instead of NHWC,
   * so that a K-stage of the 3x3 kernel (16 input channels of a halo row) is ONE contiguous run of (TW+2) * 32 bytes instead of
   * TW+2 granules of 32 bytes a pixel stride apart (4x fewer cache-line requests per staged byte), and an epilogue store
   * instruction writes 1 KiB contiguously.  Only tensors whose every reader is a 3x3 LDS-DMA conv qualify (conv_res0's output,
   * the activated twins): the host decides per tensor (engine.PlanBuilder).  DDX_LAYOUT_* bits; the channel count of a flagged
   * tensor must be a multiple of 16; DDX_ERR_UNSUPPORTED when the layer does not run on the LDS-DMA kernel. */
  int32_t layout;
  const float* out2_chan_scale;   /* [B][Cout] fp32 or NULL */
  const void* src0_alt;           /* NHWC like src0, or NULL */
  /* 1: `residual` is [B][H/2][W/2][Cout] and enters mp_sum nearest-neighbour 2x upsampled (pixel (h, w) reads (h/2, w/2); H, W even).
   * The residual branch of an up block is conv_skip(resample(x)) (unet_edm2_b4.py:110-117); a 1x1 conv commutes with the nearest
   * resample exactly (every output pixel is the same dot product), so the host runs the skip conv at the SOURCE size -- a quarter
   * of the matrix work and of the output bytes -- and the consumer gathers: the upsampled tensor never exists. */
  int32_t residual_up;
  /* > 0: every run of `out_head_norm` consecutive output channels of a pixel is RMS-normalised in the epilogue, in fp32 on the accumulators:
   * y / (out_head_eps + |y|_2 / sqrt(out_head_norm)) -- normalize(x, dim) of mp_tools.py:42-49 applied to the q | k | v vectors of the merged
   * attn_qk | attn_v conv (unet_edm2_b4.py:141-143), so that the attention kernel stages them as they are (pass it eps < 0).  Served for 64
   * on plain-store 1x1 layers by the mid-size GEMM kernel and by the register-staged kernel when its channel tile is a multiple of 64;
   * DDX_ERR_UNSUPPORTED otherwise (ddx_mpconv2d_path tells). */
  int32_t out_head_norm;
  float out_head_eps;
} ddx_conv_desc;

int ddx_mpconv2d_fwd(const ddx_conv_desc* d, ddx_stream stream);
/* Which kernel ddx_mpconv2d_fwd would run for this descriptor (no launch): 1 scalar, 2 register-staged MFMA, 3 LDS-DMA MFMA,
 * 4 small-M, 5 few-input-channel, 6 mid-size 1x1 GEMM; negative = error code.  The host asks before it decides the layout of a tensor (`layout` is ignored here). */
int ddx_mpconv2d_path(const ddx_conv_desc* d);
/* CK the library wants for a conv of this shape (call before wprep).  npix = B*H*W of the output (0 = unknown). */
int32_t ddx_mpconv2d_pick_ck(int32_t Cg, int32_t ksize, int32_t dtype, int64_t npix);

/* ------------------------------------------------------------------------------------------------
 * One EDM2 block body as ONE launch (inference, bf16): reference src/modules/unets/unet_edm2_b4.py:121-135
 *     y = conv_res0(src);  y = mp_silu(y * chan_scale[b][:]);  y = conv_res1(y);  out = clip(mp_sum(residual, y, res_t));
 *     out2 = mp_silu(out2_scale * out)   (optional)
 * with `src` the already activated block input (the producer-side twin mp_silu(x)), both convs 3x3, zero padded, grouped alike.  The hidden
 * tensor (conv_res0's output, `hidden` channels) never reaches HBM: csrc/conv_pair.hip.  Served for 32 -> 64 -> 32 channels per group
 * (C = 32 * groups, hidden = 64 * groups: the level-0 encoder blocks of the default UNet), NHWC bf16 tensors, weights prepared by
 * ddx_mpconv_wprep with CK = 32; ddx_mpconv_pair_supported() says whether a shape qualifies -- otherwise run the two ddx_mpconv2d_fwd launches,
 * which compute the same thing (equal up to the fp32 summation order inside a conv).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* src;           /* NHWC [B][H][W][C] bf16: activated input of conv_res0 */
  const void* wp0;           /* prepared conv_res0 weights [hidden][C/groups][3][3], CK0 = 32 */
  const void* wp1;           /* prepared conv_res1 weights [C][hidden/groups][3][3], CK1 = 32 */
  const float* chan_scale;   /* [B][hidden] fp32: c = emb_linear(emb) * gain + 1 */
  const void* residual;      /* NHWC [B][H][W][C] bf16 */
  void* out;                 /* NHWC [B][H][W][C] bf16 */
  void* out2;                /* NHWC [B][H][W][C] bf16 or NULL */
  int32_t B, H, W, C, hidden, groups, CK0, CK1, dtype;
  float res_t, clip, out2_scale;
} ddx_conv_pair_desc;

int ddx_mpconv_pair_supported(int32_t B, int32_t C, int32_t groups, int32_t hidden, int32_t dtype);
int ddx_mpconv_pair_fwd(const ddx_conv_pair_desc* d, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Data-gradient conv fused with the backward of the producer-side activation it feeds (training).
 * The forward stored a = mp_silu(y * s) (act = 1) or a = y * s (act = 0), s[b][c] = chan_scale[b][c] * scale, as the operand
 * of the conv described by `conv` (unet_edm2_b4.py:119-122, :139).  With dA = conv (src0 = dY, wp = the transposed
 * preparation) this writes in the conv's epilogue, without dA ever reaching HBM,
 *     dz = dA * mp_silu'(y * s) | dA;     out = dz * s (+ add);     dchan_scale[b][c] += scale * sum_pixels dz * y.
 * The output channels may be split after `split` channels over two tensors with their own y and scalar scale (the two
 * sources of an mp_cat: conv.out / y0 / scale0 hold channels [0, split), out1 / y1 / scale1 the rest; split % 64 == 0).
 * Served by the kernel the forward dispatch would pick for the conv: the LDS-DMA kernel or, for the small layers it leaves to the
 * register-staged kernel (levels 3 / 4 of the default UNet; split % 4 == 0), that kernel with the same epilogue.  Either way dchan_scale is
 * accumulated with one atomic per (wave, channel); `workspace` is not used any more (round 4) and the size query returns a 16-byte token.
 * ddx_mpconv2d_dgrad_act_workspace_bytes() returns 0 when the layer does not qualify for either -- run ddx_mpconv2d_fwd +
 * ddx_silu_scale_bwd_ex then.  conv.epilogue / residual / out2 / out_act / out_scale are ignored.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  ddx_conv_desc conv;
  const void* y0;            /* NHWC [B][H][W][split ? split : Cout] pre-activation tensor */
  const void* y1;            /* NHWC [B][H][W][Cout - split] or NULL */
  void* out1;                /* NHWC [B][H][W][Cout - split] or NULL */
  const void* add;           /* NHWC [B][H][W][Cout] or NULL */
  const float* chan_scale;   /* [B][Cout] fp32 or NULL */
  float* dchan_scale;        /* [B][Cout] fp32, accumulated, or NULL (needs chan_scale, split = 0) */
  float* workspace;          /* unused (kept for the ABI); may be NULL */
  int32_t split, act;
  float scale0, scale1;
} ddx_dgrad_act_desc;

size_t ddx_mpconv2d_dgrad_act_workspace_bytes(const ddx_dgrad_act_desc* d);
int ddx_mpconv2d_dgrad_act(const ddx_dgrad_act_desc* d, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the conv (training, unet_trainer.py:246 `accelerator.backward`):
 *   data gradient  : dX = ddx_mpconv2d_fwd(dY, wprep(..., transpose = 1))  -- the forward kernels on transposed weights;
 *   weight gradient: ddx_mpconv2d_wgrad below, w.r.t. the PREPARED weight w' (natural layout [Cout][Cg][ks][ks], fp32):
 *       dW'[o][c][kh][kw] = sum_{b,h,w} dY[b][h][w][o] * X[b][h+kh-p][w+kw-p][g*Cg+c]
 *   X is the conv's operand exactly as the forward staged it (activated twin / raw tensor; two NHWC sources = mp_cat with
 *   the scales folded into w'; resample UP = nearest-upsampled on the fly).  bf16 operands, fp32 accumulation and result.
 *   workspace: ddx_wgrad_workspace_bytes() bytes (per-split partial sums, summed deterministically); accumulate = 1
 *   adds to dw instead of overwriting it.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* dy;           /* NHWC [B][H][W][Cout] */
  const void* x0;           /* NHWC [B][sH][sW][C0] */
  const void* x1;           /* NHWC [B][sH][sW][C1] or NULL */
  float* dw;                /* [Cout][(C0+C1)/groups][ks][ks] fp32 */
  float* workspace;
  int32_t B, H, W;          /* size of dY (= conv output) */
  int32_t C0, C1, Cout, groups, ksize;
  int32_t resample;         /* DDX_RESAMPLE_KEEP | DDX_RESAMPLE_UP */
  int32_t dtype;            /* DDX_BF16 (matrix cores) | DDX_F32 (scalar parity kernel, no workspace needed) */
  int32_t accumulate;       /* 0: dw = gradient; 1: dw += gradient; 2: leave the split-K partial sums [parts][Cout][Cg][ks][ks] in `workspace`
                             * (parts = ddx_wgrad_parts(d)), no reduction, `dw` is not touched -- the consumer adds the slices
                             * (ddx_wpath_job.dwp_parts) */
} ddx_wgrad_desc;

size_t ddx_wgrad_workspace_bytes(const ddx_wgrad_desc* d);
/* split-K slices the launch described by `d` writes (0 when the descriptor is not served), and a bound that only depends on the
 * weight's shape (for sizing a per-layer partial-sum buffer once): ddx_wgrad_parts(d) <= ddx_wgrad_parts_max(Cout, Cg, groups, ksize) */
int32_t ddx_wgrad_parts(const ddx_wgrad_desc* d);
int32_t ddx_wgrad_parts_max(int32_t Cout, int32_t Cg, int32_t groups, int32_t ksize);
int ddx_mpconv2d_wgrad(const ddx_wgrad_desc* d, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise backward pieces of Block.forward (what autograd runs between the conv gradients):
 *   ddx_silu_scale_bwd : a = mp_silu(y * chan_scale[b][c] * scale)  ->  dy = da * a'(.) * chan_scale * scale,
 *                        dc[b][c] += sum_pixels da * a'(.) * y * scale      (dc NULL or chan_scale NULL: no per-channel factor)
 *   ddx_mpsum_clip_bwd : out = clip(mp_sum(res, y, t), +-clip)  ->  dres = a*m*dout (NULL: skipped), dy = b*m*dout,
 *                        m = |out| < clip (clip <= 0: no mask, out unused)
 *   ddx_pixelnorm_bwd  : y = x / (eps + |x|_2 / sqrt(C))  ->  dx
 *   ddx_mpconv_wprep_bwd: gradient w.r.t. the master weight and the gain parameter from the gradient w.r.t. the prepared
 *                        weight (dwp, natural layout [Cout][Cg][ks][ks] fp32 = output of ddx_mpconv2d_wgrad); `d` describes the
 *                        FORWARD preparation (normalize, gain, in_split, qk_head_dim); dgain (scalar, may be NULL) is accumulated.
 * NHWC tensors in `dtype`, dc / dw / dgain fp32.
 * ------------------------------------------------------------------------------------------------ */
int ddx_silu_scale_bwd(const void* da, const void* y, const float* chan_scale, float scale, void* dy, float* dc, int32_t B,
                       int64_t HW, int32_t C, int32_t dtype, ddx_stream stream);
/* Same with a row-strided da (da_ld elements between pixel rows: a channel slice of a wider NHWC tensor, i.e. one source of an
 * mp_cat) and an optional row-strided addend: dy = da * a'(.) * chan_scale * scale + add  (the skip conv's data gradient). */
int ddx_silu_scale_bwd_ex(const void* da, int64_t da_ld, const void* y, const float* chan_scale, float scale, const void* add,
                          int64_t add_ld, void* dy, float* dc, int32_t B, int64_t HW, int32_t C, int32_t act, int32_t dtype,
                          ddx_stream stream);
/* out = mp_silu(x * chan_scale[b][c] * scale): recomputes a conv operand in the backward pass (training keeps the raw tensors).
 * act = 0 (here and in ddx_silu_scale_bwd_ex) drops the mp_silu: the plain `x * c` operand of attn_qk (unet_edm2_b4.py:139). */
int ddx_silu_scale_fwd(const void* x, const float* chan_scale, float scale, void* out, int32_t B, int64_t HW, int32_t C,
                       int32_t act, int32_t dtype, ddx_stream stream);
/* out = a + b (+ c): gradient contributions of several consumers of one tensor (same shape, c may be NULL). */
int ddx_add3(const void* a, const void* b, const void* c, void* out, int64_t n, int32_t dtype, ddx_stream stream);
int ddx_mpsum_clip_bwd(const void* dout, const void* out, void* dres, void* dy, float t, float clip, int64_t n, int32_t dtype,
                       ddx_stream stream);
int ddx_pixelnorm_bwd(const void* dy, const void* x, void* dx, int64_t rows, int32_t C, float eps, int32_t dtype, ddx_stream stream);
int ddx_mpconv_wprep_bwd(const ddx_wprep_desc* d, const float* dwp, float* dw, float* dgain, int32_t accumulate, ddx_stream stream);
/* EDM2 training loss and its gradients (training/module_trainers/unet_trainer.py:271-282), fp32 NCHW:
 *   wl[b] = mean((denoised - target)^2) * (sigma^2 + sd^2) / (sigma sd)^2;  loss[b] = wl / exp(logvar) + logvar  (logvar NULL: wl)
 *   d_denoised = d mean_b(loss) / d denoised,  d_logvar[b] = d mean_b(loss) / d logvar[b]   (either may be NULL)
 * workspace: B floats. */
int ddx_edm2_loss(const float* denoised, const float* target, const float* sigma, const float* logvar, float sigma_data, float* loss,
                  float* d_denoised, float* d_logvar, float* workspace, int32_t B, int64_t n_per_sample, ddx_stream stream);
/* The same with `use_dynamic_sigma_data` (unet_trainer.py:263-269): sigma_data_vec [B] (or NULL: sigma_data) is the per-sample sigma_data of
 * the loss weight. */
int ddx_edm2_loss_v(const float* denoised, const float* target, const float* sigma, const float* logvar, float sigma_data,
                    const float* sigma_data_vec, float* loss, float* d_denoised, float* d_logvar, float* workspace, int32_t B,
                    int64_t n_per_sample, ddx_stream stream);
/* Magnitude-preserving dropout of a block's hidden activation in training (unet_edm2_b4.py:124-125: F.dropout(y, p) * (1 - p)^0.5), in place:
 * x[i] <- keep_i ? x[i] / sqrt(1 - p) : 0.  The keep mask is a pure function of (seed, stream_id, i) (Philox4x32-10): the backward applies the
 * same call to the gradient tensor of the same shape.  stream_id separates the draws of one step (one per block). */
int ddx_mp_dropout(void* x, int64_t n, float p, uint64_t seed, uint32_t stream_id, int32_t dtype, ddx_stream stream);
/* Backward of the x_ref blend of the UNet output, D = mp_sum(x_ref[:, :-1], D0, t = x_ref[:, -1:]) (unet_edm2_b4.py:293-294), NCHW fp32:
 * d_d0 = d D / d D0 applied to d_out, d_x_ref [B][C + 1][H][W] = the gradient w.r.t. both the reference channels and t (NULL: skipped). */
int ddx_unet_xref_mix_bwd(const float* d_out_nchw, const float* d0_nchw, const float* x_ref_nchw, float* d_d0_nchw, float* d_x_ref_nchw,
                          int32_t B, int32_t C, int32_t H, int32_t W, ddx_stream stream);
/* Backward of the preconditioning output D = c_skip x_in + c_out y (unet_edm2_b4.py:291) w.r.t. y: dy (NHWC, channels zero-padded to
 * Cpad) = c_out(sigma_b) * dD (NCHW fp32). */
int ddx_unet_output_combine_bwd(const float* d_out_nchw, const float* sigma, void* dy_nhwc, int32_t B, int32_t C, int32_t H, int32_t W,
                                int32_t Cpad, float sigma_data, int32_t dtype, ddx_stream stream);
/* All small-M linear layers that share the input x [M][K] (every emb_linear* of a UNet reads emb) in two launches:
 * dwp_j[o][k] = sum_m dc_j[m][o] x[m][g K/groups_j + k],  dx[m][.] += sum_j dc_j[m][o] w_j[o][k] row_scale_j[o]  (atomics;
 * dx NULL: skipped).  fp32 master weights.  Job table on the device. */
typedef struct {
  const float* dc;         /* [M][O] */
  const void* w;           /* fp32 [O][K/groups] */
  const float* row_scale;  /* [O] (ddx_wprep_rowscale / DDX_WPATH_ROWSCALE) */
  float* dwp;              /* [O][K/groups] gradient w.r.t. the prepared weight */
  int32_t O, groups;
} ddx_linear_bwd_job;

int ddx_linear_small_bwd_batched(const ddx_linear_bwd_job* jobs_dev, int32_t njobs, int32_t max_O, const float* x, int32_t x_stride,
                                 float* dx, int32_t M, int32_t K, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-tensor weight path (training): one launch per phase over a job table that lives on the device.
 * Replaces, per optimizer step, the per-module calls of MPConv.forward's weight branch under autograd and of
 * MPConv.normalize_weights (mp_tools.py:359-364, :375-378; trainer.py:375-381) for every layer at once.
 * Master weights are fp32.  row_prefix[j] = first workgroup (row) of job j in THIS phase, row_prefix[njobs] = total_rows;
 * a job that takes no part in a phase has zero rows there.  PREP also writes row_scale when the job has one (ROWSCALE is only
 * needed for jobs without a forward preparation).  Rows per job: Cout (NORMALIZE, PREP, ROWSCALE, BWD),
 * Cg * groups (TRANSPOSED).  BWD accumulates into *dgain with atomics: zero it first.
 * ------------------------------------------------------------------------------------------------ */
enum { DDX_WPATH_NORMALIZE = 0, DDX_WPATH_PREP = 1, DDX_WPATH_ROWSCALE = 2, DDX_WPATH_TRANSPOSED = 3, DDX_WPATH_BWD = 4 };

typedef struct {
  void* w;               /* fp32 master weight [Cout][Cg][k][k] (rewritten in place by NORMALIZE) */
  void* wp;              /* PREP: forward prepared buffer (ddx_wprep_bytes(Cout, Cg, ...), CK) */
  void* wp_t;            /* TRANSPOSED: data-gradient prepared buffer (ddx_wprep_bytes(Cg*groups, Cout/groups, ...), CK_t) */
  float* row_scale;      /* ROWSCALE writes, TRANSPOSED reads: [Cout] */
  const float* gain_ptr; /* learnable gain (device scalar) or NULL */
  const float* dwp;      /* BWD: gradient w.r.t. the prepared weight, natural [Cout][Cg][k][k] fp32 */
  float* dw;             /* BWD: gradient w.r.t. the master weight */
  float* dgain;          /* BWD: gradient w.r.t. *gain_ptr (accumulated) or NULL */
  float gain;
  int32_t Cout, Cg, ksize, groups, CK, CK_t, normalize, qk_head_dim, in_split;
  float in_scale0, in_scale1;
  /* BWD: dwp holds dwp_parts partial sums, [dwp_parts][Cout][Cg][k][k] (the split-K slices of ddx_mpconv2d_wgrad with accumulate = 2,
   * unused slices zero), which the pass adds in slice order while it reads them -- the weight-gradient GEMMs need no reduction
   * launch of their own.  0 or 1: dwp is the finished gradient. */
  int32_t dwp_parts, reserved;
} ddx_wpath_job;

int ddx_wpath_multi(const ddx_wpath_job* jobs_dev, const int32_t* row_prefix_dev, int32_t njobs, int32_t total_rows, int32_t phase,
                    int32_t wp_dtype, ddx_stream stream);

/* Per-row factor of the weight path: row_scale[o] = gain_eff / sqrt(fan_in) / (normalize ? eps + |w_o| / sqrt(fan_in) : 1), so that
 * w' = w * row_scale[o]  (mp_tools.py:359-364). */
int ddx_wprep_rowscale(const void* w, int32_t w_dtype, float* row_scale, const float* gain_ptr, float gain, int64_t rows,
                       int64_t fan_in, int32_t normalize, ddx_stream stream);
/* Backward of the small-M linear layers c = add_const + x @ w'^T (emb_linear*, unet_edm2_b4.py:121; grouped like the forward
 * ddx_linear_small_batched): dwp[o][k] = sum_m dc[m][o] x[m][g*K/groups + k]  (gradient w.r.t. the PREPARED weight, feed it to
 * ddx_mpconv_wprep_bwd), dx[m][.] += dc[m][o] * w[o][k] * row_scale[o]  (dx NULL: skipped; accumulated with atomics). */
int ddx_linear_small_bwd(const float* dc, const float* x, int32_t x_stride, const void* w, int32_t w_dtype, const float* row_scale,
                         float* dwp, float* dx, int32_t M, int32_t O, int32_t K, int32_t groups, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * RMS ("pixel") normalisation over the channel axis of NHWC rows  (mp_tools.py:42-49 with dim=1,
 * unet_edm2_b4.py:117): y = x / (eps + ||x||_2 / sqrt(C)).  rows = B*H*W.  In place allowed.
 * ------------------------------------------------------------------------------------------------ */
int ddx_pixelnorm_fwd(const void* x, void* y, int64_t rows, int32_t C, float eps, int32_t dtype, ddx_stream stream);
/* Same, additionally storing y_act = mp_silu(y) (the operand of the encoder block's conv_res0, unet_edm2_b4.py:117-119). */
int ddx_pixelnorm_act_fwd(const void* x, void* y, void* y_act, int64_t rows, int32_t C, float eps, int32_t dtype, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Self-attention over all H*W tokens  (unet_edm2_b4.py:137-148): q,k,v RMS-normalised over the head
 * dim per token (normalize(dim=2)), softmax(q.k / sqrt(d)) v.
 *   qk: NHWC [B][T][2C] with channels ordered (head, {q,k}, d)   <- wprep with qk_head_dim = d
 *   v : NHWC [B][T][C]  with channels ordered (head, d)
 *   out: NHWC [B][T][C]
 * ------------------------------------------------------------------------------------------------ */
int ddx_attn_fwd(const void* qk, const void* v, void* out, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                 float eps, int32_t dtype, ddx_stream stream);
/* Same with the producer-side activation of attn_proj's operand: out = mp_silu(o * out_scale[b][c]) (unet_edm2_b4.py:150-151). */
int ddx_attn_act_fwd(const void* qk, const void* v, void* out, const float* out_scale, int32_t B, int32_t T, int32_t heads,
                     int32_t head_dim, float eps, int32_t dtype, ddx_stream stream);
/* Same with explicit row strides (elements): qk rows are qk_ld apart, v rows v_ld -- q|k and v may be channel ranges of one
 * merged [B][T][3C] tensor written by a single conv. */
int ddx_attn_act_fwd_ld(const void* qk, int32_t qk_ld, const void* v, int32_t v_ld, void* out, const float* out_scale, int32_t B,
                        int32_t T, int32_t heads, int32_t head_dim, float eps, int32_t dtype, ddx_stream stream);
/* Axis-folded ("separable row / column") attention: the batch entries are the (image, column) pairs of N feature maps
 * [N][T][fold][channels] and the T tokens of an entry are `fold` rows apart -- attention along H for every (b, z, w) of the
 * reference's DAE_G1 block (modules/daes/dae_edm2_g1.py:209-228) without transposing the maps.  fold = 1 is ddx_attn_act_fwd_ld
 * (tokens = the H*W pixels of image n).  out_scale, if given, is indexed by image ([N][heads * head_dim]).
 * N * fold <= 65535.
 * eps < 0 (bf16): q, k and v are already RMS-normalised per head by their producer (ddx_conv_desc::out_head_norm); the kernel then only
 * applies 1 / sqrt(head_dim) to q and stages k / v untouched.  The same holds for the three entry points above. */
int ddx_attn_fold_fwd(const void* qk, int32_t qk_ld, const void* v, int32_t v_ld, void* out, const float* out_scale, int32_t N, int32_t T,
                      int32_t fold, int32_t heads, int32_t head_dim, float eps, int32_t dtype, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Small-M linear layers on raw master weights (no wprep): out[b][o] = post( sum_k x[b][k] * w'[o][k] )
 *   w' = [normalize(w[o])] * gain_eff / sqrt(K)     (mp_tools.py:359-367)
 *   grouped: row o only sees x[b][ (o / (O/groups)) * K .. +K )          (emb_linear, groups = mlp_groups)
 *   post: out = acc + add_const                                         (the "+ 1." of unet_edm2_b4.py:121)
 * A batch of jobs runs in one launch (all per-block emb_linear* of a UNet forward).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* w;          /* [O][K] master weights */
  const float* gain_ptr;  /* may be NULL */
  float* out;             /* [M][O] fp32 */
  float gain, add_const;
  int32_t O, K, groups, normalize;
} ddx_linear_job;

int ddx_linear_small_batched(const ddx_linear_job* jobs_dev, int32_t njobs, int32_t max_O, const float* x, int32_t x_stride,
                             int32_t M, int32_t w_dtype, ddx_stream stream);

/* MPFourier (mp_tools.py:316-330): out[b][c] = cos(x[b]*freqs[c] + phases[c]) * sqrt(2); x = log(sigma)/4 when
 * log_sigma_quarter != 0 (unet_edm2_b4.py:262, :238). */
int ddx_mpfourier(const float* x, const float* freqs, const float* phases, float* out, int32_t M, int32_t C,
                  int32_t log_sigma_quarter, ddx_stream stream);

/* emb = mp_silu(mp_sum(a, b, t)) (unet_edm2_b4.py:272-274) on [M][C] fp32;  with silu=0: plain mp_sum with a
 * per-row t (get_embeddings, unet_edm2_b4.py:232-235: a row-broadcast when a_rows == 1). */
int ddx_mpsum_rows(const float* a, int32_t a_rows, const float* b, const float* t_rows, float t, float* out, int32_t M,
                   int32_t C, int32_t silu, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * UNet input / output glue  (unet_edm2_b4.py:257-269,277 and :290-296)
 *   prep : x = c_in(sigma) * x_in ; channels [x(4), 1, ln_freq[h], 0...] -> NHWC [B][H][W][Cpad]
 *   final: D_x = c_skip*x_in + c_out*y ; optional inpainting mix with x_ref (mp_sum with t = x_ref[:, -1:])
 * x_in, x_ref, out are NCHW fp32 (the module's public I/O); y is NHWC `dtype`.
 * ------------------------------------------------------------------------------------------------ */
int ddx_unet_input_prep(const float* x_nchw, const float* sigma, const float* ln_freq_h, void* out_nhwc, int32_t B,
                        int32_t C, int32_t H, int32_t W, int32_t Cpad, float sigma_data, int32_t dtype, ddx_stream stream);
int ddx_unet_output_combine(const void* y_nhwc, const float* x_in_nchw, const float* sigma, const float* x_ref_nchw,
                            float* out_nchw, int32_t B, int32_t C, int32_t H, int32_t W, float sigma_data, int32_t dtype,
                            ddx_stream stream);

/* Stand-alone 2x nearest upsample / 2x2 average pool of an NHWC tensor (resample_2d, mp_tools.py:71-79) for the places
 * where it cannot ride in a conv's gather (blocks without a skip conv).  H, W = OUTPUT size; mode = DDX_RESAMPLE_UP|DOWN, or their
 * adjoints DDX_RESAMPLE_UP_BWD (output = half size) | DDX_RESAMPLE_DOWN_BWD (output = double size) for the backward pass. */
int ddx_resample2d(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode, int32_t dtype,
                   ddx_stream stream);

/* out = a*x + b*y + c*z on fp32 vectors (y, z may be NULL; in place allowed): CFG lerp, Heun average and the sample
 * update + ancestral noise of the EDM sampler step (pipelines/dual_diffusion_pipeline.py:701-737). */
int ddx_lincomb3(const float* x, float a, const float* y, float b, const float* z, float c, float* out, int64_t n,
                 ddx_stream stream);

/* The sampler step with device-resident scalars (the whole CFG + Heun step of dual_diffusion_pipeline.py:683-737 recorded as ONE
 * plan / hipGraph: its per-step numbers cannot be kernel arguments).  `step` is a device int32 counter.
 *   ddx_sampler_load: x_in[c] = x_pre[c] = sample for the nb / B copies c (CFG batch doubling; x_pre may be NULL) and
 *                     sigma_out[0..nb) = sig_table[step][which][0..nb)   (sig_table: [steps][2][nb] fp32, which = 0 | 1, nb <= 256)
 *   ddx_lincomb3_dev: out = a*x + b*y + c*z with (a, b, c) = coef[step * stride + {ia, ib, ic}] (ib / ic < 0: operand unused) and
 *                     z advanced by step * z_step_stride elements; same arithmetic, in the same order, as ddx_lincomb3
 *   ddx_step_advance: *step += 1 */
int ddx_sampler_load(const float* sample, float* x_in, float* x_pre, float* sigma_out, const float* sig_table, const int32_t* step,
                     int32_t which, int32_t B, int32_t nb, int64_t n_per_copy, ddx_stream stream);
int ddx_lincomb3_dev(const float* x, const float* y, const float* z, float* out, int64_t n, const float* coef, const int32_t* step,
                     int32_t stride, int32_t ia, int32_t ib, int32_t ic, int64_t z_step_stride, ddx_stream stream);
int ddx_step_advance(int32_t* step, ddx_stream stream);

/* Layout conversion helpers NCHW fp32 <-> NHWC dtype (module boundary). */
int ddx_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype, ddx_stream stream);
int ddx_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype, ddx_stream stream);
/* Stereo depth axis <-> image batch (DAE_G1; reference tensor_4d_to_5d / tensor_5d_to_4d, utils/dual_diffusion_utils.py:571-575):
 * NCHW fp32 [B][C*Z][H][W] (channel = c * Z + z) <-> NHWC images n = Z * b + z.  stereo_to_images writes Cpad channels per pixel:
 * the C data channels, then 1.0 when add_const (the constant channel, dae_edm2_g1.py:334-335), then zeros;
 * images_to_stereo reads the first C of `ld` channels. */
int ddx_stereo_to_images(const float* x, void* y, int32_t B, int32_t C, int32_t Z, int32_t H, int32_t W, int32_t Cpad, int32_t add_const,
                         int32_t dtype, ddx_stream stream);
int ddx_images_to_stereo(const void* x, int32_t ld, float* y, int32_t B, int32_t C, int32_t Z, int32_t H, int32_t W, int32_t dtype,
                         ddx_stream stream);
/* Same, reading the first C channels of pixels that are `ld` channels wide (outputs of convs padded to 8 rows). */
int ddx_nhwc_to_nchw_ld(const void* x, int32_t ld, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype,
                        ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Layout glue of the diffusion decoder (modules/unets/unet_edm2_ddec_mclt_b1.py:295-326).  The 5-D (B, C, 2, H, W) tensors of the
 * reference are NHWC images ordered n = 2*b + z; "pair-swapped" = the same rows stored at image n ^ 1 (the reflected depth row of
 * MPConv3D, i.e. the other stereo channel), which the depth-2 kernels read as the conv's second source.
 *   ddx_ddec_input_prep    : x [B][2][H][W], x_ref [B][2][H*ppf][W] fp32 -> out / out_swapped (may be NULL) [2B][H][W][Cpad] with channels
 *                            [x / sqrt(sd^2 + sigma^2), psd chunk 0..ppf-1, 1 (add_const), 0 ...]
 *   ddx_cat2_swap          : out = [scale_a * a | scale_b * b] on channels (mp_cat; b / out may be NULL: plain copy) and the same
 *                            rows pair-swapped into out_swapped
 *   ddx_ddec_output_combine: out[b][z][h][w] = c_skip * x_in + c_out * y[n][h][w][0]   (y has y_channels per pixel)
 * ------------------------------------------------------------------------------------------------ */
int ddx_ddec_input_prep(const float* x, const float* x_ref, const float* sigma, void* out, void* out_swapped, int32_t B, int32_t H,
                        int32_t W, int32_t ppf, int32_t Cpad, float sigma_data, int32_t add_const, int32_t dtype, ddx_stream stream);
int ddx_cat2_swap(const void* a, float scale_a, const void* b, float scale_b, void* out, void* out_swapped, int64_t images,
                  int64_t rows_per_image, int32_t C0, int32_t C1, int32_t dtype, ddx_stream stream);
/* out = [scale_a * a | scale_b * b] (mp_cat, rounded in the tensor dtype) and out_act = mp_silu(out): the operand of the
 * skip conv and the pre-activated operand of conv_res0 of a decoder block in one pass (unet_edm2_ddec_mclt_b1.py:107-116). */
int ddx_cat2_act(const void* a, float scale_a, const void* b, float scale_b, void* out, void* out_act, int64_t rows, int32_t C0, int32_t C1,
                 int32_t dtype, ddx_stream stream);
int ddx_ddec_output_combine(const void* y, int32_t y_channels, const float* x_in, const float* sigma, float* out, int32_t B,
                            int64_t per_sample, float sigma_data, int32_t dtype, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Fused mel-STFT  (modules/formats/old/spectrogram.py:176-179,217-226 == torch.stft(center, reflect, onesided) -> abs ->
 * FrequencyScale.scale (modules/formats/frequency_scale.py:127-128) -> ** abs_exponent -> (x - mean) * scale).
 *   audio [B][C][L] fp32 (C = 1 or 2) -> out [B][C][n_mel][T] fp32, T = 1 + L / hop frames.
 *   window [n_fft] (hann^32 etc., as the reference builds it), twiddle [n_fft] = (cos, -sin)(2 pi k / n_fft),
 *   mel filters as contiguous bands: filter m = band_w[m][0..band_len[m]) applied to bins band_start[m]...
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* audio;
  const float* window;
  const float* twiddle;       /* n_fft (re, im) pairs */
  const int32_t* band_start;  /* [n_mel] first STFT bin of each filter */
  const int32_t* band_len;    /* [n_mel] number of bins */
  const float* band_w;        /* [n_mel][band_stride] */
  float* out;
  int32_t B, C, L, T, n_fft, hop, n_mel, band_stride;
  float exponent, mean, scale;
} ddx_melstft_desc;

int ddx_mel_stft(const ddx_melstft_desc* d, ddx_stream stream);

/* Dual-window mel-scale spectrogram of MS_MDCT_DualFormat (reference src/modules/formats/ms_mdct_dual.py:229-257): per frame
 * mel[m] = (sum_k band_w[m][k] * (bin_scale_low[k] * |STFT_low[k]| + bin_scale_high[k] * |STFT_high[k]|)) ** exponent * scale + offset
 * with window_low / window_high already divided by their L2 norms (torchaudio normalized="window"), bin_scale_low = blend / density,
 * bin_scale_high = (1 - blend) / density ([n_fft/2 + 1] fp32 host tables), center = True reflect padding, hop-spaced frames.
 * audio [B][C][L] fp32 -> out [B][C][n_mel][T] fp32.  n_fft = 4096 is built. */
typedef struct {
  const float* audio;
  const float* window_low;
  const float* window_high;
  const float* twiddle;         /* [n_fft][2] cos / -sin table of exp(-2 pi i t / n_fft) */
  const float* bin_scale_low;
  const float* bin_scale_high;
  const int32_t* band_start;    /* [n_mel] first STFT bin of every filter */
  const int32_t* band_len;      /* [n_mel] number of bins */
  const float* band_w;          /* [n_mel][band_stride] filter values */
  float* out;
  int32_t B, C, L, T, n_fft, hop, n_mel, band_stride;
  float exponent, scale, offset;
} ddx_msmel_desc;
int ddx_ms_mel_spec(const ddx_msmel_desc* d, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * FGLA stereo phase reconstruction  (modules/formats/old/phase_recovery.py:39-129 `griffinlim`; decode half of
 * SpectrogramFormat.sample_to_raw, spectrogram.py:181-185,228-238).  n_fft = 6400 only.
 *   mel_to_amplitude : amp[r][t][m] = clip(mel[r][m][t] / scale + mean, 0) ** power      (rows r = B*C; feeds the un-mel GEMM)
 *   fgla_synth       : frames[b][t][c][n] = window[n] * irfft(angles * mags)[n]; angles = u / (|u| + 1e-16)
 *                      (u == NULL: angles = 1); mags = relu(un-mel) [B][C][T][mag_stride], stereo anneal
 *                      lerp(merged, spec, t_lerp) (t_lerp <= 0: merged), final_pass: the magnitudes themselves
 *   fgla_ola         : audio[b][c][j] = sum_t frames / sum_t window^2        (torch.istft, center trim, length hop*(T-1))
 *   fgla_analysis    : u[b][t][c][k] <- rfft(window * reflect_pad(audio) frame t)[k] - momentum * u[b][t][c][k]
 *                      (torch.stft + the in-place `angles.sub_(tprev, alpha=momentum)` whose result the reference keeps
 *                      as tprev, phase_recovery.py:110-119)
 *   fgla_iter        : fgla_analysis of one iteration followed by fgla_synth of the next, per frame in one launch: same state, same
 *                      frames as the two calls (u read once and written once; the synthesis uses the value just formed)
 * The state u is frame-major [B][T][C][u_stride] complex64 (re, im pairs; n_fft/2+1 valid bins per row, u_stride even and
 * >= n_fft/2+2 so that two bins move per 16-byte access; mag_stride likewise), zero before the first iteration.
 * ------------------------------------------------------------------------------------------------ */
int ddx_mel_to_amplitude(const float* mel, float* amp, int32_t rows, int32_t n_mel, int32_t T, float scale, float mean,
                         float power, ddx_stream stream);
int ddx_fgla_synth(const float* u, int32_t u_stride, const float* mags, const float* window, const float* twiddle, float* frames,
                   int32_t B, int32_t C, int32_t T, int32_t n_fft, int32_t mag_stride, float t_lerp, int32_t final_pass,
                   ddx_stream stream);
int ddx_fgla_ola(const float* frames, const float* window, float* audio, int32_t B, int32_t C, int32_t T, int32_t n_fft,
                 int32_t hop, ddx_stream stream);
int ddx_fgla_analysis(const float* audio, const float* window, const float* twiddle, float* u, int32_t u_stride, int32_t B,
                      int32_t C, int32_t T, int32_t L, int32_t n_fft, int32_t hop, float momentum, ddx_stream stream);
int ddx_fgla_iter(const float* audio, const float* window, const float* twiddle, float* u, int32_t u_stride, const float* mags,
                  int32_t mag_stride, float* frames, int32_t B, int32_t C, int32_t T, int32_t L, int32_t n_fft, int32_t hop,
                  float momentum, float t_lerp, int32_t final_pass, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Batched bf16 GEMM + row softmax: the pieces of the attention BACKWARD pass (unet_edm2_b4.py:137-148 under autograd;
 * see dualdiffusion_amd/training/attention_grad.py for the composition).
 *   C[b0][b1][m][n] = alpha * sum_k A(m,k) * B(k,n),  batch = nb0 x nb1 with element strides s?0 / s?1.
 *   a_kmajor = 0: A(m,k) = A[m*lda + k]   (reduction index contiguous);  1: A(m,k) = A[k*lda + m]
 *   b_kmajor = 0: B(k,n) = B[n*ldb + k]   (like a weight matrix);        1: B(k,n) = B[k*ldb + n]
 *   C row-major [m][n] with ldc, bf16 or fp32 (c_fp32).  lda, ldb and the A/B batch strides: multiples of 8; the contiguous
 *   axis of A and B must be readable (zero padded) up to the next multiple of 8.  Row kernels: rows of n values, stride ld.
 *   ddx_softmax_rows     : P = softmax(S * scale) over rows of n   (S fp32, P bf16)
 *   ddx_softmax_bwd_rows : dS = P o (dP - sum_j P dP) * scale        (P bf16, dP fp32, dS bf16)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* A;
  const void* B;
  void* C;
  int64_t lda, ldb, ldc;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  int32_t M, N, K, nb0, nb1;
  int32_t a_kmajor, b_kmajor, c_fp32;
  float alpha;
} ddx_bgemm_desc;

int ddx_bgemm_bf16(const ddx_bgemm_desc* d, ddx_stream stream);
int ddx_softmax_rows(const void* s, void* p, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream);
int ddx_softmax_bwd_rows(const void* p, const void* dp, void* ds, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream);
/* float32 parity path of the three calls above (operands, P / dS and the result fp32; scalar kernels): lets the whole backward pass
 * run in fp32 so that gradient parity is asserted at fp32 tolerance.  Same descriptor / argument meaning. */
int ddx_bgemm_f32(const ddx_bgemm_desc* d, ddx_stream stream);
int ddx_softmax_rows_f32(const void* s, void* p, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream);
int ddx_softmax_bwd_rows_f32(const void* p, const void* dp, void* ds, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-scale 2-D spectral loss, one block width per call  (training/loss/multiscale_spectral.py:213-294 `MSSLoss2D.stft2d`
 * + `mss_loss`) -- value AND gradient in one pass:
 *   loss[b] += mean_{c,blocks,kh,kw} weight[c][kh][kw] * ( loss_scale * | |S| - |T| | + phase_scale * (|Re S - Re T| + |Im S - Im T|) )
 *              (use_mse: squared distances; loss_scale = abs_loss_scale, phase_scale = phase_loss_scale, either may be 0)
 *   grad    += d(sum_b loss[b]) / d(sample)            (grad NULL: value only)
 *   S, T = rfft2(window * block, ortho) of the reflect-padded (w/2) sample / target, blocks every `step` pixels,
 *   midside 1: channels (L+R, L-R) (`use_midside_transform="stack"`), 0: (L, R).
 * sample, target, grad: [B][2][H][W] fp32; window [w][w]; weight [w][w/2+1]; twiddle [w] = (cos, -sin)(2 pi k / w) (must be a valid buffer; the line transforms carry their factors as literals since round 4);
 * loss [B] fp32.  loss and grad are ACCUMULATED: zero them before the first block width.  w in {8, 16, 32, 64}.
 * weight_ld: floats between the weight tables of the call's two channels (0: one table for both -- the static weightings).
 * stats non-NULL: STATISTICS call for frequency_weighting = "dynamic" (:252-253): no loss; stats[c][kh][kw] (2 x w x (w/2+1) floats,
 *   zeroed by the caller) += sum over (b, blocks) of |T_c[kh][kw]|; sample / weight / loss / grad are not read.  The caller divides by
 *   B * (H / step + 1) * (W / step + 1), clips and inverts it into the per-channel weight tables of the loss call.
 * use_midside_transform = "cat" is two calls per width (midside 0 and 1) with scaled loss_scale / phase_scale (host side).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* sample;
  const float* target;
  const float* window;
  const float* weight;
  const float* twiddle;
  float* loss;
  float* grad;
  int32_t B, C, H, W;
  int32_t block_width, step, midside, use_mse;
  float loss_scale;
  float phase_scale;
  int32_t weight_ld, reserved;
  float* stats;
} ddx_mss_desc;

int ddx_mss_loss_scale(const ddx_mss_desc* d, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step (training/trainer.py:1027-1063 clip_grad_norm_ + optimizer.step, :456-474 torch.optim.AdamW, EMA lerp) as
 * multi-tensor kernels over a DEVICE table of jobs, one job per parameter tensor (fp32 master weights / moments):
 *   ddx_multi_grad_norm: workspace3[0] = sum g^2, [1] = clip coefficient min(1, max_norm / (norm + 1e-6)), [2] = norm,
 *                        norm taken of grad_scale * g  (grad_scale = loss_scale / world_size after a SUM all-reduce)
 *   ddx_multi_adamw    : g' = g * grad_scale * (clip_coef ? clip_coef[0] : 1); decoupled weight decay; bias-corrected AdamW
 *                        (step = 1-based step count); ema (if non-NULL) <- lerp(ema, p, 1 - ema_beta).
 *                        clip_coef, when given, must point at TWO floats {coefficient, gradient norm} -- workspace3 + 1 of
 *                        ddx_multi_grad_norm: a non-finite norm in clip_coef[1] skips the whole step on the device.
 * max_n = largest job size (grid sizing).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  float* ema;   /* or NULL */
  int64_t n;
} ddx_optim_job;

int ddx_multi_grad_norm(const ddx_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float grad_scale, float max_norm,
                        float* workspace3, ddx_stream stream);
/* workspace3[1], [2] from workspace3[0] alone (a sharded pass sums |g|^2 of the local shard with ddx_multi_grad_norm, all-reduces the
 * one float over the ranks and then asks for the clip coefficient of the global norm) */
int ddx_clip_coef(float* workspace3, float grad_scale, float max_norm, ddx_stream stream);
int ddx_multi_adamw(const ddx_optim_job* jobs_dev, int32_t njobs, int64_t max_n, const float* clip_coef, float grad_scale, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step, float ema_beta, ddx_stream stream);

/* The whole post-backward parameter pass in ONE launch: clip * AdamW, up to DDX_MAX_EMAS EMAs updated in
 * configuration order with optional feedback into the training weights (reference src/training/ema.py:284-321:
 * ema <- lerp(ema, p, 1 - beta); p <- lerp(p, ema, 1 - feedback_beta)), then the forced weight normalisation of every row of
 * a weight-normalised tensor (mp_tools.py:375-378, trainer.py:1105-1108).  rows = output channels (1 for 0-d gains);
 * ema_beta / feedback_beta are HOST arrays of n_ema floats (feedback_beta < 0: none).  A non-finite gradient norm in
 * clip_coef[1] skips the step on the device (as ddx_multi_adamw). */
#define DDX_MAX_EMAS 4
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  float* ema[DDX_MAX_EMAS];   /* or NULL */
  int64_t n;
  int64_t rows;
  int32_t normalize;          /* 1: rows are RMS-normalised after the update */
  int32_t reserved;
} ddx_optim_job_ex;
int ddx_multi_adamw_ema_wn(const ddx_optim_job_ex* jobs_dev, int32_t njobs, int64_t max_rows, const float* clip_coef, float grad_scale,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, int32_t n_ema,
                           const float* ema_beta, const float* feedback_beta, float norm_eps, ddx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Launch plans: a recorded sequence of the calls above, replayed with one FFI call and optionally
 * as a hipGraph (the MI355X replacement for the reference's torch.compile, modules/module.py:145-149).
 * Recording: between ddx_plan_begin() and ddx_plan_end() every entry point above is recorded into
 * the plan instead of being launched (thread-local).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ddx_plan ddx_plan;
ddx_plan* ddx_plan_begin(void);
int ddx_plan_end(ddx_plan* p);
/* Two-lane recording: ops recorded after ddx_plan_fork() run on a side stream that waits for everything recorded before the
 * fork; ddx_plan_main() switches back to the main lane without synchronising; ddx_plan_join() makes the main lane wait for
 * the side lane.  Outside a recording the three calls are no-ops (everything runs in issue order). */
int ddx_plan_fork(void);
int ddx_plan_main(void);
int ddx_plan_join(void);
int ddx_plan_num_ops(const ddx_plan* p);
int ddx_plan_run(ddx_plan* p, ddx_stream stream);            /* eager replay */
int ddx_plan_graph_build(ddx_plan* p, ddx_stream stream);    /* capture into a hipGraphExec */
int ddx_plan_graph_launch(ddx_plan* p, ddx_stream stream);
/* per-op metadata (kernel family tag, algorithmic flops and bytes) and a hipEvent-timed eager replay */
int ddx_plan_op_info(const ddx_plan* p, int i, const char** tag, double* flops, double* bytes);
int ddx_plan_profile(ddx_plan* p, ddx_stream stream, int reps, float* ms_out);
/* While a plan is being recorded: append the launches of the finished plan `src` (e.g. a module's forward inside a sampler step). */
int ddx_plan_include(const ddx_plan* src);
void ddx_plan_destroy(ddx_plan* p);

/* sizeof() of the descriptor structs of this header as the library was compiled, so that a binding can check its mirrors against
 * the binary instead of against hand-counted bytes (tests/test_abi.py does): which = 0 ddx_wprep_desc, 1 ddx_conv_desc,
 * 2 ddx_dgrad_act_desc, 3 ddx_wgrad_desc, 4 ddx_linear_bwd_job, 5 ddx_wpath_job, 6 ddx_linear_job, 7 ddx_melstft_desc, 8 ddx_msmel_desc,
 * 9 ddx_bgemm_desc, 10 ddx_mss_desc, 11 ddx_optim_job, 12 ddx_optim_job_ex, 13 ddx_conv_pair_desc; -1 for any other code.  ddx_abi_offsetof_tail(which) is the
 * offset of the LAST field of the same struct (catches a mirror that is one trailing field short inside the tail padding). */
int64_t ddx_abi_sizeof(int32_t which);
int64_t ddx_abi_offsetof_tail(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* DDX_HIP_H */
