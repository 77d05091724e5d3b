// Device-side parameter block shared by the MFMA and the scalar conv kernels.
#pragma once
#include "common.hpp"

namespace ddx {

struct ConvParams {
  const void* src0;
  const void* src1;
  const float* cscale;
  const void* wp;
  const void* res;
  void* out;
  int B, H, W;        // output size
  int sH, sW;         // source size (H/2 for UP, 2H for DOWN)
  int C0, C1, Cin;    // source channel counts, Cin = C0 + C1
  int Cout, G, Cg, Ng, NgP;
  int nchunk, CK;
  int resample, prologue, epilogue;
  int pro_rows;       // > 0: the prologue only applies to output channels below it (merged attn_qk | attn_v conv: qk reads x * c_qk, v reads x)
  float scale0, scale1;
  float res_a, res_b;  // out = res * res_a + acc * res_b
  int res_up;          // the residual is [B][H/2][W/2][Cout], read nearest-upsampled (ddx_conv_desc::residual_up)
  float clip;
  const float* out_cs;  // producer-side activation scale [B][Cout] (or null)
  void* out2;           // activated twin of the output (or null)
  int out_act;
  float out2_scale;
  int reflect_w;        // columns outside the image are mirrored (ReflectionPad on W) instead of zero
  float norm_eps;       // DDX_EPI_PIXELNORM: eps of normalize()
  const void* src0_alt; // small-M kernel: source of the output-channel tiles below pro_rows (x * c twin), or null
  const float* out2_cs; // small-M kernel: per-(b, channel) scale of a LINEAR twin (out2 = y * out2_cs), with out2_linear
  int out2_linear;
  int head_norm;        // > 0: RMS-normalise every run of head_norm output channels of a pixel in the epilogue (ddx_conv_desc::out_head_norm)
  float head_eps;
  int layout;           // DDX_LAYOUT_* bits: which tensors are channel-blocked [B][C/16][H][W][16] (LDS-DMA kernel)
  int swap1;            // src1 is read from image b ^ 1 (DDX_PAD_SWAP_SRC1)
  int paired;           // input = [src0 | src1 | src0' | src1'], ' = image b ^ 1 (DDX_PAD_SWAP_PAIRED); Cin = 2 * (C0 + C1)
  // DDX_EPI_SILU_BWD (data-gradient conv fused with the backward of the producer-side activation; `res` = y of the first part)
  const void* bwd_y1;   // y of the second channel part (or null)
  void* bwd_out1;       // output of the second channel part
  const void* bwd_add;  // [B][H][W][Cout] gradient added to the result (or null)
  float* bwd_dc;        // [B][Cout] accumulated chan_scale gradient (one atomic per wave and channel)
  int bwd_split, bwd_act;
  float bwd_s0, bwd_s1;
  // spatial tiling (MFMA kernel)
  int TH, TW, tiles_h, tiles_w, arows_alloc;
  float inv_TWP;
  int group_smem;  // LDS bytes of one split-K group's staging region
  int tile_order;  // LDS-DMA kernel: 1 = pixel tiles numbered column-major inside an image, each XCD walking ONE contiguous range of them (halo rows and
                   // columns of neighbouring tiles then meet in the same L2 within a few units); 0 = row-major, tiles dealt round-robin to the XCDs
  int force_cfg;   // register-staged kernel: 0 = heuristic choice, else 1 + 3 * tile (0..3: 256x64, 256x32, 128x64, 128x32) + split-K index (1, 2, 4)
};

// index into the prepared weight tensor wp[g][chunk][tap][NgP][CK]
__host__ __device__ inline size_t wp_index(int g, int n, int tap, int c, int nchunk, int taps, int NgP, int CK) {
  return ((((size_t)g * nchunk + c / CK) * taps + tap) * NgP + n) * CK + (c % CK);
}

int launch_conv_mfma(const ConvParams& p, int ksize, int dtype, hipStream_t s);   // conv_mfma.hip
bool conv_mfma_supported(const ConvParams& p, int ksize, int dtype);
int conv_mfma_tile_bn(const ConvParams& p, int ksize, int dtype);   // output channels per tile of the configuration launch_conv_mfma will pick (0: none)
void conv_mfma_plan_tiles(ConvParams& p, int ksize, int dtype);
int launch_conv_direct(const ConvParams& p, int ksize, int dtype, hipStream_t s);  // conv_direct.hip
bool conv_dma_supported(const ConvParams& p, int ksize, int dtype, bool any_size);  // conv_dma.hip
int launch_conv_dma(const ConvParams& p, int ksize, hipStream_t s);
bool conv_sm_supported(const ConvParams& p, int ksize, int dtype);  // conv_sm.hip (small-M weight-streaming kernel, CK = 16 layout)
int launch_conv_sm(const ConvParams& p, int ksize, hipStream_t s);
bool conv_few_supported(const ConvParams& p, int ksize, int dtype);  // conv_few.hip (3x3 over 8 zero-padded input channels: the input convs)
int launch_conv_few(const ConvParams& p, hipStream_t s);
bool conv_gemm_supported(const ConvParams& p, int ksize, int dtype, bool auto_pick);  // conv_gemm.hip (mid-size raw 1x1 layers: 128 x 128 GEMM tiles, deep LDS-DMA ring)
int launch_conv_gemm(const ConvParams& p, hipStream_t s);
size_t conv_dma_bwd_ws_bytes(const ConvParams& p, int ksize);  // 16 when the DDX_EPI_SILU_BWD epilogue serves the layer, else 0

}  // namespace ddx
