// C-ABI entry points for the magnitude-preserving conv path: weight preparation (fused weight-norm + gain + cast +
// re-layout, reference src/modules/mp_tools.py:359-364), forced weight normalisation (:375-378) and the conv forward
// dispatcher (MFMA implicit GEMM or the scalar kernel).
#include <cmath>
#include <cstdlib>

#include "conv_params.hpp"
#include "wpath_rows.hpp"

namespace ddx {

// one workgroup per (destination) output channel; row bodies in wpath_rows.hpp
template <typename TW_, typename TP>
__global__ __launch_bounds__(256) void wprep_kernel(const TW_* __restrict__ w, TP* __restrict__ wp, const float* gain_ptr,
                                                    float gain, int Cout, int Cg, int taps, int G, int CK, int normalize,
                                                    int qk_d, float eps, int in_split, float in_s0, float in_s1, int row_off, int rows_total) {
  __shared__ float scratch[4];
  wprep_row<TW_, TP>(w, wp, gain_ptr, gain, Cout, Cg, taps, G, CK, normalize, qk_d, eps, in_split, in_s0, in_s1, blockIdx.x, scratch, row_off,
                     rows_total);
}

// ---- data-gradient (transposed) preparation: per-row scale first, then one workgroup per destination row (g, c)
template <typename TW_>
__global__ __launch_bounds__(256) void wprep_rowscale_kernel(const TW_* __restrict__ w, float* __restrict__ row_scale, const float* gain_ptr,
                                                             float gain, int fan, int normalize, float eps) {
  __shared__ float scratch[4];
  wprep_rowscale_row<TW_>(w, row_scale, gain_ptr, gain, fan, normalize, eps, blockIdx.x, scratch);
}

template <typename TW_, typename TP>
__global__ __launch_bounds__(256) void wprep_transposed_kernel(const TW_* __restrict__ w, TP* __restrict__ wp, const float* __restrict__ row_scale,
                                                               int Cout, int Cg, int taps, int G, int CK, int qk_d, int in_split,
                                                               float in_s0, float in_s1) {
  wprep_transposed_row<TW_, TP>(w, wp, row_scale, Cout, Cg, taps, G, CK, qk_d, in_split, in_s0, in_s1, blockIdx.x);
}

template <typename TW_>
__global__ __launch_bounds__(256) void normalize_rows_kernel(TW_* w, int64_t fan, float eps) {
  __shared__ float scratch[4];
  normalize_row<TW_>(w, fan, eps, blockIdx.x, scratch);
}

}  // namespace ddx

using namespace ddx;

extern "C" size_t ddx_wprep_bytes(int32_t Cout, int32_t Cg, int32_t ksize, int32_t groups, int32_t CK, int32_t dtype) {
  if (groups <= 0 || CK <= 0) return 0;
  const int Ng = Cout / groups, NgP = round_up(Ng, 32), nchunk = ceil_div(Cg, CK);
  return (size_t)groups * nchunk * ksize * ksize * NgP * CK * dtype_size(dtype);
}

extern "C" int32_t ddx_mpconv2d_pick_ck(int32_t Cg, int32_t ksize, int32_t dtype, int64_t npix) {
  // small-M layers run split-K over 64-channel chunks; large ones amortise barriers over 128-channel chunks
  const bool small_m = npix > 0 && npix <= 8192;
  if (ksize == 1 && dtype == DDX_BF16 && Cg >= 128 && Cg % 128 == 0 && !small_m) return 128;
  if (ksize == 1 && Cg >= 64) return 64;
  return 32;
}

extern "C" int ddx_mpconv_wprep(const ddx_wprep_desc* dp, ddx_stream stream) {
  if (!dp || !dp->w || !dp->wp) return set_error(DDX_ERR_ARG, "wprep: null");
  const ddx_wprep_desc d = *dp;
  if (d.groups <= 0 || d.Cout % d.groups || (d.ksize != 1 && d.ksize != 3 && d.ksize != 5) || (d.CK != 16 && d.CK != 32 && d.CK != 64 && d.CK != 128))
    return set_error(DDX_ERR_ARG, "wprep: bad shape");
  if (d.qk_head_dim > 0 && (d.groups != 1 || d.Cout % (2 * d.qk_head_dim))) return set_error(DDX_ERR_ARG, "wprep: bad qk_head_dim");
  if (d.rows_total != 0 && (d.groups != 1 || d.transpose || d.row_offset < 0 || d.row_offset + d.Cout > d.rows_total))
    return set_error(DDX_ERR_ARG, "wprep: row_offset / rows_total describe a slice of a merged forward matrix (groups = 1)");
  if (d.transpose) {
    if (!d.row_scale) return set_error(DDX_ERR_ARG, "wprep: transpose needs the row_scale workspace");
    return dispatch([d](hipStream_t s) -> int {
      const int taps = d.ksize * d.ksize, Ng = d.Cout / d.groups, Cin = d.Cg * d.groups, fan = d.Cg * taps;
      const size_t bytes = ddx_wprep_bytes(Cin, Ng, d.ksize, d.groups, d.CK, d.wp_dtype);
      if (round_up(d.Cg, 32) != d.Cg || Ng % d.CK) {
        if (int rc = zero_bytes(d.wp, bytes, s)) return rc;
      }
#define DDX_WPREP_T(TWT, TPT)                                                                                              \
  do {                                                                                                                     \
    hipLaunchKernelGGL((wprep_rowscale_kernel<TWT>), dim3(d.Cout), dim3(256), 0, s, (const TWT*)d.w, d.row_scale, d.gain_ptr, \
                       d.gain, fan, d.normalize, 1e-4f);                                                                   \
    hipLaunchKernelGGL((wprep_transposed_kernel<TWT, TPT>), dim3(Cin), dim3(256), 0, s, (const TWT*)d.w, (TPT*)d.wp,        \
                       (const float*)d.row_scale, d.Cout, d.Cg, taps, d.groups, d.CK, d.qk_head_dim, d.in_split, d.in_scale0, \
                       d.in_scale1);                                                                                       \
  } while (0)
      if (d.w_dtype == DDX_F32 && d.wp_dtype == DDX_F32) DDX_WPREP_T(float, float);
      else if (d.w_dtype == DDX_F32 && d.wp_dtype == DDX_BF16) DDX_WPREP_T(float, bf16);
      else if (d.w_dtype == DDX_BF16 && d.wp_dtype == DDX_BF16) DDX_WPREP_T(bf16, bf16);
      else if (d.w_dtype == DDX_BF16 && d.wp_dtype == DDX_F32) DDX_WPREP_T(bf16, float);
      else return set_error(DDX_ERR_ARG, "wprep: dtype");
#undef DDX_WPREP_T
      return check_launch("wprep(transpose)");
    }, stream, "wprep");
  }
  return dispatch([d](hipStream_t s) -> int {
    const int taps = d.ksize * d.ksize;
    const int Ng = d.Cout / d.groups;
    const size_t bytes = ddx_wprep_bytes(d.Cout, d.Cg, d.ksize, d.groups, d.CK, d.wp_dtype);
    if (d.rows_total == 0 && (round_up(Ng, 32) != Ng || d.Cg % d.CK)) {  // (merged matrices are zero-filled by their owner)
      if (int rc = zero_bytes(d.wp, bytes, s)) return rc;
    }
#define DDX_WPREP(TWT, TPT)                                                                                         \
  hipLaunchKernelGGL((wprep_kernel<TWT, TPT>), dim3(d.Cout), dim3(256), 0, s, (const TWT*)d.w, (TPT*)d.wp, d.gain_ptr, \
                     d.gain, d.Cout, d.Cg, taps, d.groups, d.CK, d.normalize, d.qk_head_dim, 1e-4f, d.in_split, d.in_scale0,   \
                     d.in_scale1, d.row_offset, d.rows_total)
    if (d.w_dtype == DDX_F32 && d.wp_dtype == DDX_F32) DDX_WPREP(float, float);
    else if (d.w_dtype == DDX_F32 && d.wp_dtype == DDX_BF16) DDX_WPREP(float, bf16);
    else if (d.w_dtype == DDX_BF16 && d.wp_dtype == DDX_BF16) DDX_WPREP(bf16, bf16);
    else if (d.w_dtype == DDX_BF16 && d.wp_dtype == DDX_F32) DDX_WPREP(bf16, float);
    else return set_error(DDX_ERR_ARG, "wprep: dtype");
#undef DDX_WPREP
    return check_launch("wprep");
  }, stream, "wprep");
}

extern "C" int ddx_wprep_rowscale(const void* w, int32_t w_dtype, float* row_scale, const float* gain_ptr, float gain, int64_t rows,
                                  int64_t fan_in, int32_t normalize, ddx_stream stream) {
  if (!w || !row_scale || rows <= 0 || fan_in <= 0) return set_error(DDX_ERR_ARG, "wprep_rowscale: bad args");
  return dispatch([=](hipStream_t s) -> int {
    if (w_dtype == DDX_F32)
      hipLaunchKernelGGL(wprep_rowscale_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (const float*)w, row_scale, gain_ptr, gain, (int)fan_in, normalize, 1e-4f);
    else
      hipLaunchKernelGGL(wprep_rowscale_kernel<bf16>, dim3((unsigned)rows), dim3(256), 0, s, (const bf16*)w, row_scale, gain_ptr, gain, (int)fan_in, normalize, 1e-4f);
    return check_launch("wprep_rowscale");
  }, stream);
}

extern "C" int ddx_normalize_weights(void* w, int32_t w_dtype, int64_t rows, int64_t fan_in, ddx_stream stream) {
  if (!w || rows <= 0 || fan_in <= 0) return set_error(DDX_ERR_ARG, "normalize_weights: bad args");
  return dispatch([=](hipStream_t s) -> int {
    if (w_dtype == DDX_F32)
      hipLaunchKernelGGL(normalize_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (float*)w, fan_in, 1e-4f);
    else
      hipLaunchKernelGGL(normalize_rows_kernel<bf16>, dim3((unsigned)rows), dim3(256), 0, s, (bf16*)w, fan_in, 1e-4f);
    return check_launch("normalize_weights");
  }, stream);
}

// descriptor checks + the device parameter block shared by the forward entry and the fused data-gradient entry
static int conv_fill(const ddx_conv_desc& d, ConvParams* pp) {
  if (!d.src0 || !d.wp || !d.out) return set_error(DDX_ERR_ARG, "conv: null buffer");
  if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.C0 <= 0 || d.Cout <= 0 || d.groups <= 0) return set_error(DDX_ERR_ARG, "conv: bad size");
  if ((d.C1 > 0) != (d.src1 != nullptr)) return set_error(DDX_ERR_ARG, "conv: src1/C1 mismatch");
  const int Cin = (d.C0 + d.C1) * ((d.pad_mode & DDX_PAD_SWAP_PAIRED) ? 2 : 1);
  if (Cin % d.groups || d.Cout % d.groups) return set_error(DDX_ERR_ARG, "conv: channels not divisible by groups");
  if (d.ksize != 1 && d.ksize != 3 && d.ksize != 5) return set_error(DDX_ERR_UNSUPPORTED, "conv: ksize must be 1, 3 or 5");
  // (5x5: the few-channel input / output convs of DAE_G1, dae_edm2_g1.py:274,303 -- served by the scalar kernel only)
  if ((d.prologue & DDX_PRO_SCALE) && !d.chan_scale) return set_error(DDX_ERR_ARG, "conv: chan_scale missing");
  if (d.epilogue == DDX_EPI_MPSUM && !d.residual) return set_error(DDX_ERR_ARG, "conv: residual missing");
  if (d.resample == DDX_RESAMPLE_UP && ((d.H | d.W) & 1)) return set_error(DDX_ERR_ARG, "conv: upsampled size must be even");
  if (d.residual_up && (d.epilogue != DDX_EPI_MPSUM || ((d.H | d.W) & 1) || (d.residual_up & ~1)))
    return set_error(DDX_ERR_ARG, "conv: residual_up needs the mp_sum epilogue and an even output size");
  if (d.dtype != DDX_F32 && d.dtype != DDX_BF16) return set_error(DDX_ERR_ARG, "conv: dtype");

  ConvParams p{};
  p.src0 = d.src0; p.src1 = d.src1; p.cscale = d.chan_scale; p.wp = d.wp; p.res = d.residual; p.out = d.out;
  p.B = d.B; p.H = d.H; p.W = d.W;
  p.sH = d.resample == DDX_RESAMPLE_UP ? d.H / 2 : (d.resample == DDX_RESAMPLE_DOWN ? d.H * 2 : d.H);
  p.sW = d.resample == DDX_RESAMPLE_UP ? d.W / 2 : (d.resample == DDX_RESAMPLE_DOWN ? d.W * 2 : d.W);
  p.C0 = d.C0; p.C1 = d.C1; p.Cin = Cin; p.Cout = d.Cout; p.G = d.groups;
  p.Cg = Cin / d.groups; p.Ng = d.Cout / d.groups; p.NgP = round_up(p.Ng, 32);
  p.CK = d.CK; p.nchunk = ceil_div(p.Cg, d.CK);
  p.resample = d.resample; p.prologue = d.prologue; p.epilogue = d.epilogue;
  p.pro_rows = d.prologue_rows;
  p.scale0 = d.scale0; p.scale1 = d.scale1;
  const float t = d.res_t, nrm = std::sqrt((1.f - t) * (1.f - t) + t * t);
  p.res_a = (1.f - t) / nrm; p.res_b = t / nrm;
  p.res_up = d.residual_up;
  p.clip = d.clip;
  p.out_cs = d.out_scale; p.out2 = d.out2; p.out_act = d.out_act; p.out2_scale = d.out2_scale;
  p.src0_alt = d.src0_alt; p.out2_cs = d.out2_chan_scale; p.out2_linear = d.out2_linear;
  p.head_norm = d.out_head_norm; p.head_eps = d.out_head_eps;
  if (d.out_head_norm && (d.out_head_norm != 64 || d.ksize != 1 || d.groups != 1 || d.Cout % 64 || d.epilogue != DDX_EPI_STORE || d.out2 || d.out_act ||
                          d.out_scale || d.dtype != DDX_BF16 || !(d.out_head_eps >= 0.f)))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: out_head_norm is served for 64-channel heads on plain-store bf16 1x1 layers without groups");
  if (d.out2_linear && (!d.out2 || !d.out2_chan_scale)) return set_error(DDX_ERR_ARG, "conv: out2_linear needs out2 and out2_chan_scale");
  if (d.src0_alt && (d.prologue != DDX_PRO_NONE || d.prologue_rows <= 0)) return set_error(DDX_ERR_ARG, "conv: src0_alt needs prologue_rows > 0 and no prologue");
  p.layout = d.layout;
  if (d.layout) {
    if (d.layout & ~15) return set_error(DDX_ERR_ARG, "conv: layout");
    if (d.dtype != DDX_BF16 || d.ksize != 3) return set_error(DDX_ERR_UNSUPPORTED, "conv: channel-blocked tensors are served by the 3x3 bf16 LDS-DMA kernel");
    if (((d.layout & DDX_LAYOUT_SRC0_C16) && d.C0 % 16) || ((d.layout & DDX_LAYOUT_SRC1_C16) && (!d.src1 || d.C1 % 16)) ||
        ((d.layout & (DDX_LAYOUT_OUT_C16 | DDX_LAYOUT_OUT2_C16)) && d.Cout % 16) || ((d.layout & DDX_LAYOUT_OUT2_C16) && !d.out2))
      return set_error(DDX_ERR_ARG, "conv: a channel-blocked tensor needs a multiple of 16 channels");
  }
  p.reflect_w = (d.pad_mode & DDX_PAD_REFLECT_W) ? 1 : 0;
  p.swap1 = (d.pad_mode & DDX_PAD_SWAP_SRC1) ? 1 : 0;
  p.paired = (d.pad_mode & DDX_PAD_SWAP_PAIRED) ? 1 : 0;
  if (d.pad_mode & ~7) return set_error(DDX_ERR_ARG, "conv: pad_mode");
  if (p.paired && (p.swap1 || !d.src1 || (d.B & 1) || d.groups != 1 || d.chan_scale))
    return set_error(DDX_ERR_ARG, "conv: DDX_PAD_SWAP_PAIRED needs src1, an even image count, one group and no channel scale");
  if (p.reflect_w && (d.W < 2 || d.resample == DDX_RESAMPLE_DOWN)) return set_error(DDX_ERR_UNSUPPORTED, "conv: reflect padding needs W >= 2 and no fused 2x2 average");
  if (p.swap1 && (!d.src1 || (d.B & 1))) return set_error(DDX_ERR_ARG, "conv: DDX_PAD_SWAP_SRC1 needs src1 and an even image count");
  *pp = p;
  return 0;
}

static int conv_fwd_impl(const ddx_conv_desc& d, ddx_stream stream, bool query);

extern "C" int ddx_mpconv2d_fwd(const ddx_conv_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "conv: null descriptor");
  return conv_fwd_impl(*dp, stream, false);
}

extern "C" int ddx_mpconv2d_path(const ddx_conv_desc* dp) {
  if (!dp) return set_error(DDX_ERR_ARG, "conv: null descriptor");
  ddx_conv_desc d = *dp;
  d.layout = 0;
  return conv_fwd_impl(d, nullptr, true);
}

// query: return the kernel the descriptor selects (1 scalar, 2 register-staged MFMA, 3 LDS-DMA, 4 small-M, 5 few-channel, 6 1x1 GEMM) instead of launching it
static int conv_fwd_impl(const ddx_conv_desc& d, ddx_stream stream, bool query) {
  if (d.epilogue != DDX_EPI_STORE && d.epilogue != DDX_EPI_MPSUM && d.epilogue != DDX_EPI_PIXELNORM) return set_error(DDX_ERR_ARG, "conv: epilogue");
  if (d.epilogue == DDX_EPI_PIXELNORM && (d.residual || d.out_act || d.out_scale || !(d.res_t > 0.f)))
    return set_error(DDX_ERR_ARG, "conv: the pixel-norm epilogue takes eps in res_t and no residual / output activation");
  ConvParams p{};
  if (int rc = conv_fill(d, &p)) return rc;
  const int ks = d.ksize, dt = d.dtype;
  if (d.epilogue == DDX_EPI_PIXELNORM) {
    p.norm_eps = d.res_t;
    if (d.force_direct == 1 || d.force_direct == 2 || d.force_direct >= 16 || !conv_dma_supported(p, ks, dt, /*any_size=*/true))
      return set_error(DDX_ERR_UNSUPPORTED, "conv: the pixel-norm epilogue is built for the LDS-DMA kernel (one group; Cout <= 64, or a wide 1x1 layer with 192 ... 512 output channels)");
  }
  // d.force_direct selects the kernel: 0 = automatic, 1 = scalar reference kernel, 2 = register-staged MFMA kernel,
  // 3 = LDS-DMA MFMA kernel (error when the layer does not qualify)
  // >= 16: the register-staged kernel with tile / split-K configuration (force_direct - 16), see ConvParams::force_cfg
  if (d.force_direct >= 16) p.force_cfg = d.force_direct - 16 + 1;
  // CK = 16 is the weight layout of the small-M weight-streaming kernel (conv_sm.hip): the choice was made when the weights were prepared
  if (d.out_head_norm && (d.CK == 16 || d.force_direct == 1 || d.force_direct == 3 || d.force_direct == 4 || d.force_direct == 5))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: out_head_norm runs on the 1x1 GEMM kernel and the register-staged kernel only");
  if (d.CK == 16 && d.force_direct != 1 && d.force_direct != 3 && !d.layout) {
    if ((d.force_direct != 0 && d.force_direct != 4) || !conv_sm_supported(p, ks, dt))
      return set_error(DDX_ERR_UNSUPPORTED, "conv: weights prepared with CK = 16 run on the small-M kernel only, and this layer does not qualify");
    if (query) return 4;
    if (d.layout) return set_error(DDX_ERR_UNSUPPORTED, "conv: channel-blocked tensors need the LDS-DMA kernel");
    const double flops_sm = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg * ks * ks;
    const double bytes_sm = 2.0 * ((double)p.B * p.sH * p.sW * p.Cin + (double)p.B * p.H * p.W * p.Cout * (d.residual ? (d.residual_up ? 1.25 : 2.0) : 1.0) + (double)p.Cout * p.Cg * ks * ks);
    return dispatch([p, ks](hipStream_t s) -> int { return launch_conv_sm(p, ks, s); }, stream, ks == 3 ? "conv3x3_sm" : "conv1x1_sm", flops_sm, bytes_sm);
  }
  if (d.force_direct == 4) return set_error(DDX_ERR_UNSUPPORTED, "conv: the small-M kernel needs weights prepared with CK = 16");
  // the input convs (3x3 over 8 zero-padded channels, plain store + twin) have their own kernel
  if (d.force_direct == 5 && (d.epilogue != DDX_EPI_STORE || !conv_few_supported(p, ks, dt)))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: layer does not qualify for the few-input-channel kernel (3x3 over 8 padded channels, plain store + twin)");
  if ((d.force_direct == 0 || d.force_direct == 5) && d.epilogue == DDX_EPI_STORE && conv_few_supported(p, ks, dt)) {
    if (query) return 5;
    const double flops_f = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg * ks * ks;
    const double bytes_f = 2.0 * ((double)p.B * p.H * p.W * (p.Cin + (double)p.Cout * (d.out2 ? 2.0 : 1.0)));
    return dispatch([p](hipStream_t s) -> int { return launch_conv_few(p, s); }, stream, "conv3x3_few", flops_f, bytes_f);
  }
  // mid-size raw 1x1 layers (levels 2 / 3): 128 x 128 GEMM tiles on a deep LDS-DMA ring
  if (d.force_direct == 6 && !conv_gemm_supported(p, ks, dt, false)) return set_error(DDX_ERR_UNSUPPORTED, "conv: layer does not qualify for the 1x1 GEMM kernel");
  if (d.force_direct == 6 || (d.force_direct == 0 && !d.out2_linear && conv_gemm_supported(p, ks, dt, true))) {
    if (query) return 6;
    const double flops_g = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg;
    const double bytes_g = 2.0 * ((double)p.B * p.H * p.W * (p.Cin + (double)p.Cout) + (double)p.Cout * p.Cg);
    return dispatch([p](hipStream_t s) -> int { return launch_conv_gemm(p, s); }, stream, "conv1x1_gemm", flops_g, bytes_g);
  }
  const bool mfma = d.force_direct != 1 && conv_mfma_supported(p, ks, dt);
  static const bool dma_enabled = []() { const char* e = std::getenv("DDX_CONV_DMA"); return !e || e[0] != '0'; }();
  if (d.force_direct == 3 && !conv_dma_supported(p, ks, dt, /*any_size=*/true))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: layer does not qualify for the LDS-DMA kernel");
  const bool dma = d.force_direct == 3 || d.epilogue == DDX_EPI_PIXELNORM ||
                   (d.force_direct == 0 && mfma && dma_enabled && !d.out_head_norm && conv_dma_supported(p, ks, dt, false));
  if (d.out_head_norm && (!mfma || conv_mfma_tile_bn(p, ks, dt) % d.out_head_norm))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: out_head_norm needs a channel tile that is a multiple of the head size");
  if (d.force_direct >= 16 && !mfma) return set_error(DDX_ERR_UNSUPPORTED, "conv: layer does not qualify for the register-staged MFMA kernel");
  if (d.out2_linear && d.CK != 16 && (dma || !mfma))
    return set_error(DDX_ERR_UNSUPPORTED, "conv: out2_linear is served by the small-M kernel (CK = 16) and the register-staged MFMA kernel only");
  if (d.src0_alt && !dma) return set_error(DDX_ERR_UNSUPPORTED, "conv: src0_alt is served by the small-M kernel (CK = 16) and the wide 1x1 units of the LDS-DMA kernel only");
  if (query) return dma ? 3 : mfma ? 2 : 1;
  if (d.layout && (!dma || p.epilogue == DDX_EPI_PIXELNORM)) return set_error(DDX_ERR_UNSUPPORTED, "conv: channel-blocked tensors need the LDS-DMA kernel (ddx_mpconv2d_path)");
  const double flops = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg * ks * ks;
  const double es = (double)dtype_size(dt);
  const double bytes = es * ((double)p.B * p.sH * p.sW * p.Cin + (double)p.B * p.H * p.W * p.Cout * (d.residual ? (d.residual_up ? 1.25 : 2.0) : 1.0) +
                             (double)p.Cout * p.Cg * ks * ks);
  return dispatch([p, ks, dt, mfma, dma](hipStream_t s) -> int {
    if (dma) return launch_conv_dma(p, ks, s);
    return mfma ? launch_conv_mfma(p, ks, dt, s) : launch_conv_direct(p, ks, dt, s);
  }, stream, dma ? (ks == 3 ? "conv3x3_dma" : "conv1x1_dma") : mfma ? (ks == 3 ? "conv3x3_mfma" : "conv1x1_mfma") : "conv_direct", flops, bytes);
}

// ---- data-gradient conv fused with the backward of the producer-side activation (LDS-DMA kernel only)
static int dgrad_act_fill(const ddx_dgrad_act_desc& d, ConvParams* pp) {
  ddx_conv_desc c = d.conv;
  c.epilogue = DDX_EPI_STORE; c.residual = nullptr; c.residual_up = 0; c.out2 = nullptr; c.out_act = 0; c.out_scale = nullptr;
  ConvParams p{};
  if (int rc = conv_fill(c, &p)) return rc;
  if (!d.y0 || (d.split > 0 && (!d.y1 || !d.out1)) || d.split < 0 || d.split >= d.conv.Cout) return set_error(DDX_ERR_ARG, "dgrad_act: bad parts");
  if (d.dchan_scale && (!d.chan_scale || d.split > 0)) return set_error(DDX_ERR_ARG, "dgrad_act: dchan_scale needs chan_scale and one part");
  p.epilogue = DDX_EPI_SILU_BWD;
  p.res = d.y0; p.bwd_y1 = d.y1; p.bwd_out1 = d.out1; p.bwd_add = d.add; p.out_cs = d.chan_scale;
  p.bwd_dc = d.dchan_scale;
  p.bwd_split = d.split; p.bwd_act = d.act; p.bwd_s0 = d.scale0; p.bwd_s1 = d.scale1;
  *pp = p;
  return 0;
}

// DDX_EPI_SILU_BWD on the register-staged kernel: 4-channel items must lie in one part
static bool dgrad_act_on_mfma(const ConvParams& p) { return p.bwd_split % 4 == 0 && p.Ng % 4 == 0 && p.CK != 16; }

extern "C" size_t ddx_mpconv2d_dgrad_act_workspace_bytes(const ddx_dgrad_act_desc* dp) {
  if (!dp) return 0;
  ConvParams p{};
  ddx_dgrad_act_desc d = *dp;
  static float dummy;
  if (!d.conv.src0) d.conv.src0 = &dummy;
  if (!d.conv.wp) d.conv.wp = &dummy;
  if (!d.conv.out) d.conv.out = &dummy;
  if (!d.y0) d.y0 = &dummy;
  if (d.split > 0 && !d.y1) d.y1 = &dummy;
  if (d.split > 0 && !d.out1) d.out1 = &dummy;
  if (dgrad_act_fill(d, &p) != 0) return 0;
  if (d.conv.dtype != DDX_BF16 || !conv_mfma_supported(p, d.conv.ksize, d.conv.dtype)) return 0;
  if (!conv_dma_supported(p, d.conv.ksize, d.conv.dtype, false))
    return dgrad_act_on_mfma(p) ? 16 : 0;   // register-staged kernel: no workspace (per-wave atomics into dchan_scale), a token size says "fused"
  return conv_dma_bwd_ws_bytes(p, d.conv.ksize);
}

extern "C" int ddx_mpconv2d_dgrad_act(const ddx_dgrad_act_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "dgrad_act: null descriptor");
  const ddx_dgrad_act_desc d = *dp;
  ConvParams p{};
  if (int rc = dgrad_act_fill(d, &p)) return rc;
  const int ks = d.conv.ksize;
  const double flops = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg * ks * ks;
  const double bytes = 2.0 * ((double)p.B * p.sH * p.sW * p.Cin + (double)p.B * p.H * p.W * p.Cout * (d.add ? 3.0 : 2.0) + (double)p.Cout * p.Cg * ks * ks);
  if (d.conv.dtype == DDX_BF16 && conv_mfma_supported(p, ks, d.conv.dtype) && !conv_dma_supported(p, ks, d.conv.dtype, false) && dgrad_act_on_mfma(p)) {
    // the small layers (levels 3 / 4) that the forward dispatch gives to the register-staged kernel: same kernel, activation backward in its epilogue
    const int dt = d.conv.dtype;
    return dispatch([p, ks, dt](hipStream_t s) -> int { return launch_conv_mfma(p, ks, dt, s); }, stream, ks == 3 ? "conv3x3_mfma_bwd" : "conv1x1_mfma_bwd", flops, bytes);
  }
  if (d.conv.dtype != DDX_BF16 || !conv_dma_supported(p, ks, d.conv.dtype, /*any_size=*/true))
    return set_error(DDX_ERR_UNSUPPORTED, "dgrad_act: layer does not qualify for the LDS-DMA kernel (run the conv and ddx_silu_scale_bwd)");
  return dispatch([p, ks](hipStream_t s) -> int { return launch_conv_dma(p, ks, s); }, stream, ks == 3 ? "conv3x3_dma_bwd" : "conv1x1_dma_bwd", flops, bytes);
}
