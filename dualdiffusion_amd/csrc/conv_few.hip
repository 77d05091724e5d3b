// 3x3 conv over FEW input channels (8, zero-padded): the input conv of the UNet / VAE / decoders (`conv_in`: reference
// unet_edm2_b4.py:113-124, 263-277 -- x, the constant channel and the ln-frequency channel -> model_channels).
//
// The layer is a pure write stream: 1.4 MB in, 45 (+ 45 for the activated twin) MB out at B=4, 3 GFLOP.  On the general kernels it was
// staged like any other conv (32-channel K chunks: three quarters of the MFMA work multiplies zeros, one launch tile per 64 output
// channels re-stages the input) and took 40-48 us, 280 us at batch 32.  Here
//   * the whole K range of a pixel is 9 taps x 8 channels = 72 -> FIVE k-steps of v_mfma_f32_32x32x16_bf16: k-step s covers taps 2s and
//     2s + 1, so the B operand of lane (pixel, khalf) is ONE 16-byte LDS read of the halo tile at the pixel shifted by tap 2s + khalf, and
//     the A operand (weights) of a wave's output-channel fragments is loaded once per workgroup into registers (40 VGPRs);
//   * a workgroup owns an 8 x 32 pixel tile and ALL output channels; results go through an LDS tile (pixel-major, padded) so that the
//     stores are whole NHWC rows: 32 pixels x Cout channels = one contiguous run (16 KB at 256 channels) per tile row;
//   * the activated twin mp_silu(out2_scale * y) is written the same way from the same accumulators.
#include <algorithm>
#include <cstdlib>

#include "conv_params.hpp"

namespace ddx {
namespace {

constexpr int kFewTH = 8, kFewTW = 32, kFewRP = 2;          // tile; tile rows per pass through the LDS output tiles
constexpr int kFewHW = kFewTW + 2, kFewHH = kFewTH + 2;     // halo tile

template <int NFW>
__global__ __launch_bounds__(256, 2) void conv_few_kernel(const ConvParams p, const int tiles_w, const int tiles_h) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  int bx = blockIdx.x;
  const int tx = bx % tiles_w; bx /= tiles_w;
  const int ty = bx % tiles_h;
  const int b = bx / tiles_h;
  const int h0 = ty * kFewTH, w0 = tx * kFewTW;
  const int Cout = p.Cout;
  const int pstride = Cout * 2 + 16;                          // bytes per pixel row of the LDS output tiles (16-byte pad: bank spread)
  char* halo = smem;                                          // [10][34] pixels x 16 bytes
  char* otile = smem + kFewHH * kFewHW * 16;                  // [RP * 32 pixels][pstride]
  char* ttile = otile + kFewRP * kFewTW * pstride;            // twin
  const bf16* src = reinterpret_cast<const bf16*>(p.src0);
  const bf16* wp = reinterpret_cast<const bf16*>(p.wp);

  // ---- halo tile (zero outside the image)
  for (int i = tid; i < kFewHH * kFewHW; i += 256) {
    const int hy = i / kFewHW, hx = i - hy * kFewHW;
    const int gh = h0 - 1 + hy, gw = w0 - 1 + hx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) v = *reinterpret_cast<const u32x4*>(src + (((size_t)b * p.H + gh) * p.W + gw) * 8);
    *reinterpret_cast<u32x4*>(halo + i * 16) = v;
  }
  // ---- this wave's weight fragments: output channels (wave * NFW + i) * 32 + l31, k-step s = taps 2s | 2s + 1, 8 channels each
  const int nfrag = Cout / 32;
  bf16x8 wf[NFW][5];
#pragma unroll
  for (int i = 0; i < NFW; ++i) {
    const int f = wave * NFW + i;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + khalf;
      bf16x8 v = {};
      if (f < nfrag && tap < 9) v = *reinterpret_cast<const bf16x8*>(wp + ((size_t)tap * p.NgP + f * 32 + l31) * p.CK);
      wf[i][s] = v;
    }
  }
  // activation fragment addresses: tap 2s + khalf of pixel column l31 (tile row added per pass); tap 9 reads tap 8 against zero weights
  int boff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + khalf, 8);
    boff[s] = ((tap / 3) * kFewHW + l31 + tap % 3) * 16;
  }
  __syncthreads();

  bf16* out = reinterpret_cast<bf16*>(p.out);
  bf16* out2 = reinterpret_cast<bf16*>(p.out2);
  for (int r0 = 0; r0 < kFewTH; r0 += kFewRP) {
#pragma unroll
    for (int rr = 0; rr < kFewRP; ++rr) {
      const int j = r0 + rr;
      f32x16 acc[NFW];
#pragma unroll
      for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(halo + j * kFewHW * 16 + boff[s]);
#pragma unroll
        for (int i = 0; i < NFW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][s], xb, acc[i], 0, 0, 0);
      }
      // lane (pixel l31, khalf) holds channels 8 q + 4 khalf + e of each fragment: 8-byte pieces into the pixel-major LDS tiles
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        const int f = wave * NFW + i;
        if (f >= nfrag) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = f * 32 + 8 * q + 4 * khalf;
          Vec4<bf16> ov, tv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = acc[i][4 * q + e];
            ov.set(e, y);
            tv.set(e, mp_silu_f(y * p.out2_scale));
          }
          const int po = (rr * kFewTW + l31) * pstride + ch * 2;
          *reinterpret_cast<bf16x4*>(otile + po) = ov.v;
          if (out2) *reinterpret_cast<bf16x4*>(ttile + po) = tv.v;
        }
      }
    }
    __syncthreads();
    // ---- whole NHWC rows out: tile row j, 32 pixels x Cout channels contiguous in HBM
    const int vpp = Cout / 8;                                  // 16-byte vectors per pixel
    const int nvec = kFewRP * kFewTW * vpp;
    for (int i = tid; i < nvec; i += 256) {
      const int px = i / vpp, v = i - px * vpp;
      const int rr = px / kFewTW, pw = px - rr * kFewTW;
      const int gh = h0 + r0 + rr, gw = w0 + pw;
      if (gh >= p.H || gw >= p.W) continue;
      const size_t go = (((size_t)b * p.H + gh) * p.W + gw) * Cout + v * 8;
      *reinterpret_cast<u32x4*>(out + go) = *reinterpret_cast<const u32x4*>(otile + px * pstride + v * 16);
      if (out2) *reinterpret_cast<u32x4*>(out2 + go) = *reinterpret_cast<const u32x4*>(ttile + px * pstride + v * 16);
    }
    __syncthreads();
  }
}

}  // namespace

// 3x3, one group, eight (zero-padded) input channels from ONE source, plain store (+ activated twin): the input convs
bool conv_few_supported(const ConvParams& p, int ksize, int dtype) {
  if (dtype != DDX_BF16 || ksize != 3 || p.G != 1 || p.C0 != 8 || p.src1 || p.Cin != 8) return false;
  if (p.resample != DDX_RESAMPLE_KEEP || p.prologue != DDX_PRO_NONE || p.epilogue != DDX_EPI_STORE || p.scale0 != 1.0f) return false;
  if (p.out_act || p.out_cs || p.clip > 0.f || p.reflect_w || p.swap1 || p.paired || p.layout || p.out2_linear || p.src0_alt) return false;
  if (p.Cout % 32 || p.Cout < 32 || p.Cout > 256 || p.CK != 32) return false;
  return true;
}

int launch_conv_few(const ConvParams& p, hipStream_t s) {
  const int tiles_h = ceil_div(p.H, kFewTH), tiles_w = ceil_div(p.W, kFewTW);
  const int nfw = ceil_div(p.Cout / 32, 4);
  const size_t smem = (size_t)kFewHH * kFewHW * 16 + (size_t)(p.out2 ? 2 : 1) * kFewRP * kFewTW * (p.Cout * 2 + 16);
  const dim3 grid((unsigned)(p.B * tiles_h * tiles_w));
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_few_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_few_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_few)");
    attr_done = true;
  }
  if (nfw == 1) hipLaunchKernelGGL(conv_few_kernel<1>, grid, dim3(256), smem, s, p, tiles_w, tiles_h);
  else hipLaunchKernelGGL(conv_few_kernel<2>, grid, dim3(256), smem, s, p, tiles_w, tiles_h);
  return check_launch("conv_few");
}

}  // namespace ddx
