// 1x1 magnitude-preserving conv as a plain GEMM for the MID-SIZE layers of the UNet (levels 2 / 3 of the default model: 1376 ... 5504
// pixels at B = 4; reference src/modules/unets/unet_edm2_b4.py:110-158: conv_skip over mp_cat, the merged attn_qk | attn_v conv).
//
// Why another kernel: between the wide LDS-DMA units (256 x 256, levels 0 / 1) and the small-M kernels these layers have too few
// pixels x channels for 256-wide units to fill 256 CUs, and on 128 x 64 register-staged tiles they move 1.5 x the bytes through
// L2 -> LDS, which is what bounds them (round 4: a CU sustains what it keeps in flight -- one 16 ... 24 KB stage per workgroup gave
// ~45 GB/s per CU).  Here:
//   * 128 pixels x 128 channels per workgroup (4 waves, 64 x 64 each: 4 fragment reads per 4 MFMAs), two workgroups per CU,
//     one unit per workgroup (grid = units: 264 for the level-3 qkv conv, 258 for the level-2 skip over mp_cat);
//   * 64-channel K stages by LDS-DMA into a ring of TWO slots (32-channel stages where the channel counts need them), the next
//     stage in flight while this one multiplies (counted vmcnt, raw barrier: a __syncthreads() would drain the queue), one
//     barrier per stage.  Measured on the level-3 qkv conv: (32,2) 18.9, (32,3) 19.8, (32,4) 19.3, (64,2) 18.6, (64,3) 26.7 us,
//     6 / 8 slots ~30 us -- deeper rings lose (occupancy 2 -> 1 workgroup per CU costs more than the extra stages buy);
//   * rows are 64 / 128 bytes, 16-byte slots XOR-swizzled on the SOURCE address (conv_dma.hip's scheme) -- conflict-free
//     ds_read_b128;
//   * raw operands only (two sources of an mp_cat with the scales folded into the weights, `src0_alt` per channel tile), plain
//     store (+ clip) epilogue through an LDS transpose (the layers served need nothing else).
#include <algorithm>
#include <cstdlib>

#include "conv_params.hpp"

namespace ddx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOob = 0x7fffff00;
constexpr int GM = 128, GN = 128;
constexpr int ES = GN + 4;                           // fp32 row stride of the epilogue tile
constexpr int smem_bytes(int gk, int ns) { return GM * ES * 4 > ns * (GM + GN) * gk * 2 ? GM * ES * 4 : ns * (GM + GN) * gk * 2; }

__device__ __forceinline__ void dma16(rsrc_t rs, int voff, int soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)std::min<size_t>(bytes, 0x7ffffff0u), 0x00020000);
}
// 16-byte slot swizzle of LDS row r (conv_dma.hip's DmaGeom::swz): 64-byte rows (r >> 2) & 3, 128-byte rows (r >> 1) & 7
template <int GK> __device__ __forceinline__ int swz(int r) { return GK == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct GemmArgs { int M, MT, NT, by_n, nst; };

// GK = channels per K stage (LDS rows of 2 GK bytes), NS = ring slots (NS - 1 stages in flight)
template <int GK, int NS>
__global__ __launch_bounds__(256, smem_bytes(GK, NS) <= 80 * 1024 ? 2 : 1) void conv_gemm_kernel(const ConvParams p, const GemmArgs a) {
  constexpr int kStage = (GM + GN) * GK * 2;
  constexpr int RB = GK * 2, LPR = RB / 16, RPW = 1024 / RB;      // bytes per row, 16-byte slots per row, rows per DMA instruction
  constexpr int PPW = GM / RPW / 4;                               // pieces per wave and operand
  constexpr int KS = GK / 16;                                     // MFMA k-steps per stage
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  // unit decode: the tiles that share an operand meet on one XCD (observed block -> XCD map id % 8: speed only)
  int mt, nt;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    if (a.by_n) { mt = slot % a.MT; nt = (slot / a.MT) * 8 + xcd; }
    else { nt = slot % a.NT; mt = (slot / a.NT) * 8 + xcd; }
    if (mt >= a.MT || nt >= a.NT) return;
  }
  const int m0 = mt * GM, n0 = nt * GN;

  const void* s0p = (p.src0_alt && n0 < p.pro_rows) ? p.src0_alt : p.src0;
  const rsrc_t rs0 = make_rsrc(s0p, (size_t)a.M * p.C0 * 2);
  const rsrc_t rs1 = p.src1 ? make_rsrc(p.src1, (size_t)a.M * p.C1 * 2) : rs0;
  const rsrc_t rsw = make_rsrc(p.wp, (size_t)p.nchunk * p.NgP * p.CK * 2);
  const int ck_shift = __builtin_ctz(p.CK);

  // ---- DMA bookkeeping: a stage is GM / RPW activation pieces + as many weight pieces of 1 KiB (RPW rows of RB bytes); wave w moves
  // pieces w, w + 4, ... of each.  Per lane: byte offsets of its row inside the tensors; the K advance is the scalar offset.
  const int lrow = lane / LPR, lslot = lane % LPR;
  int av0[PPW], av1[PPW], bv[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int r = (wave + 4 * i) * RPW + lrow;
    const int m = m0 + r;
    const int sl = (lslot ^ swz<GK>(r)) * 16;
    av0[i] = m < a.M ? m * p.C0 * 2 + sl : kOob;
    av1[i] = m < a.M ? m * p.C1 * 2 + sl : kOob;
    const int n = min(n0 + r, p.NgP - 1);   // rows past NgP only feed outputs that are never stored
    bv[i] = ((n << ck_shift) * 2) + sl;
  }
  auto issue = [&](int s) {   // stage s -> ring slot s % NS
    char* base = smem + (s % NS) * kStage;
    const int k0 = s * GK;
    const bool first = k0 < p.C0;
    const int soff_a = (first ? k0 : k0 - p.C0) * 2;
    const int soff_b = ((((k0 >> ck_shift) * p.NgP) << ck_shift) + (k0 & (p.CK - 1))) * 2;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + 4 * i;
      if (first) dma16(rs0, av0[i], soff_a, base + piece * 1024);
      else dma16(rs1, av1[i], soff_a, base + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma16(rsw, bv[i], soff_b, base + GM * RB + (wave + 4 * i) * 1024);
  };

  // ---- fragment read offsets inside a stage (bytes): weights = A operand (rows = output channels), activations = B (columns = pixels)
  int xoff[2], woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wm * 64 + j * 32 + l31;
    xoff[j] = r * RB + ((khalf ^ swz<GK>(r)) << 4);    // k-step ks adds 2 ks to the slot: XOR commutes below
    const int rw = wn * 64 + j * 32 + l31;
    woff[j] = GM * RB + rw * RB + ((khalf ^ swz<GK>(rw)) << 4);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nst = a.nst;
  static_assert(NS >= 2 && NS <= 4, "ring of two to four slots (deeper rings measured 50 % slower: the LDS-DMA queue stalls the issuing waves)");
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nst) issue(i);
  for (int s = 0; s < nst; ++s) {
    // stage s has landed when at most the younger stages' pieces (2 PPW per stage and wave) are outstanding
    const int younger = min(nst - 1 - s, NS - 2);
    if (younger >= 2) wait_vmcnt<2 * 2 * PPW>();
    else if (younger == 1) wait_vmcnt<2 * PPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is done reading stage s - 1, whose slot is refilled now
    __builtin_amdgcn_sched_barrier(0);
    if (s + NS - 1 < nst) issue(s + NS - 1);
    const char* st = smem + (s % NS) * kStage;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 wf[2], xf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf[j] = *reinterpret_cast<const bf16x8*>(st + (woff[j] ^ (ks << 5)));
        xf[j] = *reinterpret_cast<const bf16x8*>(st + (xoff[j] ^ (ks << 5)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // (nothing in flight any more) every wave is done with the ring the output tile overlays

  // ---- epilogue: accumulators (lane = one pixel, runs of 4 channels) -> LDS [pixel][channel] fp32 -> 8-byte bf16 stores on NHWC rows
  float* sE = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ml = wm * 64 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        f32x4 y4;
#pragma unroll
        for (int e = 0; e < 4; ++e) y4[e] = acc[i][j][4 * k4 + e];
        *reinterpret_cast<f32x4*>(sE + (size_t)ml * ES + wn * 64 + i * 32 + 8 * k4 + 4 * khalf) = y4;
      }
  }
  __syncthreads();
  bf16* out = reinterpret_cast<bf16*>(p.out);
#pragma unroll
  for (int it = 0; it < GM * (GN / 4) / 256; ++it) {
    const int idx = tid + it * 256;
    const int ml = idx >> 5, c4 = idx & 31;
    const int m = m0 + ml, n = n0 + c4 * 4;
    f32x4 y4 = *reinterpret_cast<const f32x4*>(sE + (size_t)ml * ES + c4 * 4);
    if (p.head_norm) {
      // 16 consecutive lanes hold the 64 channels of one head of one pixel (ddx_conv_desc::out_head_norm): y / (eps + |y| / sqrt(64))
      float ss = y4[0] * y4[0] + y4[1] * y4[1] + y4[2] * y4[2] + y4[3] * y4[3];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float sc = 1.0f / (p.head_eps + sqrtf(ss) * 0.125f);
#pragma unroll
      for (int e = 0; e < 4; ++e) y4[e] *= sc;
    }
    if (m >= a.M || n >= p.Ng) continue;
    bf16x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (bf16)(p.clip > 0.f ? fminf(fmaxf(y4[e], -p.clip), p.clip) : y4[e]);
    *reinterpret_cast<bf16x4*>(out + (size_t)m * p.Cout + n) = ov;
  }
}

template <int GK, int NS>
int launch_cfg(const ConvParams& p, GemmArgs a, int grid, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<GK, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes(GK, NS)) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_gemm)");
    attr_done = true;
  }
  a.nst = p.Cin / GK;
  hipLaunchKernelGGL((conv_gemm_kernel<GK, NS>), dim3(grid), dim3(256), smem_bytes(GK, NS), s, p, a);
  return check_launch("conv_gemm");
}

}  // namespace

// raw 1x1 layers, one group, plain store; `auto_pick`: also the size window where this kernel is the default choice
bool conv_gemm_supported(const ConvParams& p, int ksize, int dtype, bool auto_pick) {
  if (dtype != DDX_BF16 || ksize != 1 || p.G != 1) return false;
  if (p.prologue != DDX_PRO_NONE || p.scale0 != 1.0f || (p.src1 && p.scale1 != 1.0f)) return false;
  if (p.resample != DDX_RESAMPLE_KEEP || p.reflect_w || p.swap1 || p.paired || p.layout) return false;
  if (p.epilogue != DDX_EPI_STORE || p.out2 || p.out_act || p.out_cs) return false;
  if (p.C0 % 32 || (p.src1 && p.C1 % 32) || p.CK % 32 || p.Cout % 4) return false;
  if (p.src0_alt && (p.src1 || p.pro_rows <= 0 || p.pro_rows % GN)) return false;
  const long M = (long)p.B * p.H * p.W;
  if (M * std::max(p.C0, p.C1) * 2 >= 0x7fffff00l) return false;
  if (!auto_pick) return true;
  static const int knob = std::getenv("DDX_CONV_GEMM") ? atoi(std::getenv("DDX_CONV_GEMM")) : 1;
  if (!knob) return false;
  // units: from ~2/3 of the CUs to ~1.5 per CU (measured, tools/conv_bench.py --path gemm+auto with DDX_CONV_GEMM=0: level-3 qkv 22.1 ->
  // 18.6 us, level-2 skip convs 17.2 -> 12.3, 22.2 -> 15.4, 31.8 -> 28.8 us; from 688 units (level 1) the 256-wide LDS-DMA units win)
  // Round 5: and again from 600 units (levels 0 / 1 at every batch) -- with operands that stream from HBM
  // (tools/conv_bench.py --batch 32 --cold-act 3) the one-unit-per-workgroup kernel at two workgroups per CU beats the persistent 192 x 256
  // units: level-1 skip over mp_cat 362 -> 320 us, 243 -> 219, level-0 341 -> 328 at B = 32; 127 -> 118, 93 -> 85 at B = 8 (deeper rings and
  // 32-channel stages lose here too: 354 ... 393 us); at B = 4 the step reads 4.413 / 4.412 / 4.395 ms with the window opening at 2500 / 1300 / 600 units
  // (two runs each on one box; between 400 and 600 units -- level 2 at B = 8 -- the persistent units measured 44.4 against 48.0 us).
  // DDX_CONV_GEMM=2: the round-4 window only; DDX_GEMM_MIN_UNITS=n moves the upper window.
  const long units = (long)ceil_div((int)M, GM) * ceil_div(p.Ng, GN);
  static const long big = std::getenv("DDX_GEMM_MIN_UNITS") ? atol(std::getenv("DDX_GEMM_MIN_UNITS")) : 600;
  return p.Cin >= 256 && units >= 160 && (units <= 400 || (knob != 2 && units >= big));
}

int launch_conv_gemm(const ConvParams& p, hipStream_t s) {
  GemmArgs a{};
  a.M = p.B * p.H * p.W;
  a.MT = ceil_div(a.M, GM);
  a.NT = ceil_div(p.Ng, GN);
  // which tile index is dealt over the XCDs: the one whose count splits more evenly into eight
  const double imb_n = (double)ceil_div(a.NT, 8) * 8 / a.NT, imb_m = (double)ceil_div(a.MT, 8) * 8 / a.MT;
  a.by_n = imb_n <= imb_m ? 1 : 0;
  const int grid = a.by_n ? a.MT * round_up(a.NT, 8) : a.NT * round_up(a.MT, 8);
  // 64-channel stages (whole 128-byte lines) in a two-slot ring, two workgroups per CU; 32-channel stages where the layer's channel
  // counts need them.  Measured (tools/conv_bench.py --path gemm, level-3 qkv): 32-channel stages with 2 / 3 / 4 slots 18.9 / 19.8 /
  // 19.3 us, 64-channel with 2 slots 18.6, with 3 slots (one workgroup per CU) 26.7, rings of 6 / 8 slots 31.6 / 30.1 us.
  static const int force = []() { const char* e = std::getenv("DDX_GEMM_CFG"); return e ? atoi(e) : 0; }();   // (experiments: 322, 323, 324, 642, 643)
  if (force == 323) return launch_cfg<32, 3>(p, a, grid, s);
  if (force == 324) return launch_cfg<32, 4>(p, a, grid, s);
  if (force == 322) return launch_cfg<32, 2>(p, a, grid, s);
  if (force == 643 && !(p.C0 % 64 || (p.src1 && p.C1 % 64) || p.CK % 64)) return launch_cfg<64, 3>(p, a, grid, s);
  if (p.C0 % 64 || (p.src1 && p.C1 % 64) || p.CK % 64) return launch_cfg<32, 2>(p, a, grid, s);
  return launch_cfg<64, 2>(p, a, grid, s);
}

}  // namespace ddx
