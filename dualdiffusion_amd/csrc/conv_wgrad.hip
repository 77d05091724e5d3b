// Weight gradient of the magnitude-preserving conv (backward of F.conv2d w.r.t. its weight, reference
// src/modules/mp_tools.py:369 under autograd) as an LDS-DMA MFMA kernel for gfx950, bf16 operands, fp32 result:
//   dW'[g*Ng+n][c][tap] = sum_{b,h,w} dY[b,h,w,g*Ng+n] * X[b,h+dh-1,w+dw-1,g*Cg+c]        (X zero-padded)
// i.e. per group a GEMM with M = output channels, N = input channels x taps and the PIXELS as the reduction index.
// Both operands live in HBM as [pixel][channel] rows, so the reduction index is the slow one: fragments cannot be read
// with plain ds_read_b128.  The tiles are staged exactly as the forward kernel stages them (buffer_load ... lds, zero
// padding by out-of-range offsets) and read with the gfx950 transpose read ds_read_b64_tr_b16, which hands every lane 4
// consecutive pixels of its channel column from a row-major [4 pixels][16 channels] block (probed:
// tools/probe/ds_read_tr_probe.hip).  All read addresses are one per-lane base + compile-time immediates.
//   unit   = (group, 64 output channels, 32 input channels, a range of 4x32-pixel tiles [split-K]);
//   waves  = 2 (output-channel fragments) x 2 (tap subsets {0..4}, {5..8}): 5 or 4 accumulator fragments per wave;
//   stage  = one pixel tile: dY 128 x 64 (16 KB) + X halo 6x34 x 32 (13 KB), double buffered, one barrier per tile;
//   output = fp32 partial sums per K split (deterministic), summed by a second small kernel.
// The 1x1 case uses the same code with one tap (waves split 2 x 2 over 64 x 64 channels).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_params.hpp"

namespace ddx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
constexpr int kOob = 0x7fffff00;

__device__ __forceinline__ void dma16(rsrc_t rs, int voff, int soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 4 consecutive rows of this lane's channel column (see file header)
__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 frag8(const char* p, int off_lo, int off_hi) {
  const s16x4 lo = tr_read(p + off_lo), hi = tr_read(p + off_hi);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

struct WgradParams {
  const void* dy; const void* x0; const void* x1;
  float* ws;  // [ksplit][Cout][Cg][taps] fp32 partial sums
  int B, H, W, sH, sW, C0, C1, Cin, Cout, G, Cg, Ng, resample;
  int tiles_h, tiles_w, ntile_px, ksplit, n_tiles, c_tiles;
  int wide;  // 1x1 layers on 128 x 128 channel tiles (conv_wgrad1x1_wide_kernel)
};

constexpr int kTH = 4, kTW = 32, kPix = kTH * kTW;  // pixel tile = one stage

template <int KS> struct WgGeom {
  static constexpr int TAPS = KS * KS, PAD = KS / 2;
  static constexpr int BNW = 64;                       // output channels per unit
  static constexpr int BCW = KS == 3 ? 32 : 64;        // input channels per unit
  static constexpr int TWP = kTW + 2 * PAD, XROWS = (kTH + 2 * PAD) * TWP;
  static constexpr int DY_RB = BNW * 2, X_RB = BCW * 2;              // row bytes
  static constexpr int DY_PIECES = kPix * DY_RB / 1024;             // 16
  static constexpr int X_PIECES = (XROWS * X_RB + 1023) / 1024;     // 13 (3x3) / 16 (1x1)
  static constexpr int DY_BYTES = DY_PIECES * 1024, X_BYTES = X_PIECES * 1024;
  static constexpr int STAGE = DY_BYTES + X_BYTES;
  static constexpr int DI = (DY_PIECES + 3) / 4, XI = (X_PIECES + 3) / 4;
  // epilogue transpose image [64 output channels][BCW x TAPS floats (+4: the two half-waves land on different banks)]
  static constexpr int ROWF = BCW * TAPS, ROWP = ROWF + 4;
  static constexpr int EPI_BYTES = BNW * ROWP * 4;
  static constexpr int SMEM = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
  // accumulator fragments per wave: 3x3: taps {0..4} / {5..8} of one 32-channel input tile; 1x1: one 32x32 block
  static constexpr int NFRAG = KS == 3 ? 5 : 1;
};

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
  using GEO = WgGeom<KS>;
  constexpr int TAPS = GEO::TAPS, PAD = GEO::PAD, TWP = GEO::TWP, NFRAG = GEO::NFRAG;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nf = wave & 1;                       // output-channel fragment of this wave
  const int wq = wave >> 1;                      // 3x3: tap subset; 1x1: input-channel fragment
  const int tap0 = KS == 3 ? wq * 5 : 0;
  const int ntap = KS == 3 ? (wq ? 4 : 5) : 1;
  const int cf = KS == 3 ? 0 : wq;

  // ---- unit decode
  int u = blockIdx.x;
  const int ksp = u % p.ksplit; u /= p.ksplit;
  const int ct = u % p.c_tiles; u /= p.c_tiles;
  const int nt = u % p.n_tiles;
  const int g = u / p.n_tiles;
  const int n0 = nt * GEO::BNW, c0 = ct * GEO::BCW;
  const int per = (p.ntile_px + p.ksplit - 1) / p.ksplit;
  const int t_begin = ksp * per, t_end = min(t_begin + per, p.ntile_px);

  // ---- DMA bookkeeping: rows of the two images this lane moves (tile independent part)
  const int cabs = g * p.Cg + c0;
  const bool second = cabs >= p.C0;
  const rsrc_t rsx = second ? make_rsrc(p.x1, (size_t)p.B * p.sH * p.sW * p.C1 * 2) : make_rsrc(p.x0, (size_t)p.B * p.sH * p.sW * p.C0 * 2);
  const rsrc_t rsy = make_rsrc(p.dy, (size_t)p.B * p.H * p.W * p.Cout * 2);
  const int xstride2 = (second ? p.C1 : p.C0) * 2;
  const int xchan2 = (second ? cabs - p.C0 : cabs) * 2;
  const int ychan2 = (g * p.Ng + n0) * 2;
  int dth[GEO::DI], dtw[GEO::DI], dslot[GEO::DI];
#pragma unroll
  for (int i = 0; i < GEO::DI; ++i) {
    const int r = (wave + 4 * i) * (1024 / GEO::DY_RB) + lane / (GEO::DY_RB / 16);
    dth[i] = r / kTW; dtw[i] = r % kTW;
    dslot[i] = (lane % (GEO::DY_RB / 16)) * 16;
  }
  int xhh[GEO::XI], xww[GEO::XI], xslot[GEO::XI];
#pragma unroll
  for (int i = 0; i < GEO::XI; ++i) {
    const int r = (wave + 4 * i) * (1024 / GEO::X_RB) + lane / (GEO::X_RB / 16);
    xhh[i] = r < GEO::XROWS ? r / TWP - PAD : -(1 << 20);
    xww[i] = r % TWP - PAD;
    xslot[i] = (lane % (GEO::X_RB / 16)) * 16;
  }
  auto issue = [&](int t, int st) {
    char* sb = smem + st * GEO::STAGE;
    int tile = t;
    const int tx = tile % p.tiles_w; tile /= p.tiles_w;
    const int ty = tile % p.tiles_h;
    const int b = tile / p.tiles_h;
    const int h0 = ty * kTH, w0 = tx * kTW;
#pragma unroll
    for (int i = 0; i < GEO::DI; ++i) {
      const int piece = wave + 4 * i;
      if (piece < GEO::DY_PIECES) {
        const int h = h0 + dth[i], w = w0 + dtw[i];
        const bool ok = h < p.H && w < p.W;
        const int off = ok ? ((b * p.H + h) * p.W + w) * (p.Cout * 2) + dslot[i] : kOob;
        dma16(rsy, off, ychan2, sb + piece * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < GEO::XI; ++i) {
      const int piece = wave + 4 * i;
      if (piece < GEO::X_PIECES) {
        const int h = h0 + xhh[i], w = w0 + xww[i];
        const bool ok = h >= 0 && h < p.H && w >= 0 && w < p.W;
        const int pix = p.resample == DDX_RESAMPLE_UP ? (b * p.sH + (h >> 1)) * p.sW + (w >> 1) : (b * p.sH + h) * p.sW + w;
        const int off = ok ? pix * xstride2 + xslot[i] : kOob;
        dma16(rsx, off, xchan2, sb + GEO::DY_BYTES + piece * 1024);
      }
    }
  };

  // ---- fragment read bases (bytes inside a stage): lane i of a 16-lane group supplies row (i>>2), 4-channel run (i&3)
  const int gq = lane >> 4, li = lane & 15;
  const int kbase = (gq >> 1) * 8 + (li >> 2);                  // pixel offset of this lane's supplied row in a 16-pixel step
  const int a_base = kbase * GEO::DY_RB + (nf * 32 + (gq & 1) * 16 + (li & 3) * 4) * 2;
  const int b_base = GEO::DY_BYTES + kbase * GEO::X_RB + (cf * 32 + (gq & 1) * 16 + (li & 3) * 4) * 2;

  f32x16 acc[NFRAG];
#pragma unroll
  for (int f = 0; f < NFRAG; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  auto compute = [&](auto stage_tag, auto wq_tag) {
    constexpr int STG = decltype(stage_tag)::value;
    constexpr int WQ = decltype(wq_tag)::value;   // compile-time tap subset (3x3)
    constexpr int T0 = KS == 3 ? WQ * 5 : 0;
    constexpr int NT = KS == 3 ? (WQ ? 4 : 5) : 1;
    const char* sa = smem + STG * GEO::STAGE + a_base;
    const char* sx = smem + STG * GEO::STAGE + b_base;
#pragma unroll
    for (int ks = 0; ks < kPix / 16; ++ks) {
      const int th = ks / (kTW / 16), tw0 = (ks % (kTW / 16)) * 16;
      const bf16x8 af = frag8(sa, ks * 16 * GEO::DY_RB, (ks * 16 + 4) * GEO::DY_RB);
#pragma unroll
      for (int f = 0; f < NT; ++f) {
        const int tap = T0 + f;
        const int row = (th + tap / KS) * TWP + tw0 + tap % KS;  // halo row of the step's first pixel, shifted by the tap
        const bf16x8 xf = frag8(sx, row * GEO::X_RB, (row + 4) * GEO::X_RB);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf, acc[f], 0, 0, 0);
      }
    }
  };
  auto compute_w = [&](auto stage_tag) {
    if (KS == 3 && wq) compute(stage_tag, std::integral_constant<int, 1>{});
    else compute(stage_tag, std::integral_constant<int, 0>{});
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (t_begin < t_end) issue(t_begin, 0);
  for (int t = t_begin; t < t_end; t += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t_end) issue(t + 1, 1);
    compute_w(S0{});
    if (t + 1 >= t_end) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < t_end) issue(t + 2, 0);
    compute_w(S1{});
  }

  // ---- store the partial sums.  acc[f][4q+e] = dW'[n = 8q + 4*khalf + e][c = lane & 31] of tap0 + f: in the gradient's
  // natural [n][c][tap] layout one lane's 16 values are 4-byte pieces 36 B apart, so the block is transposed through LDS
  // (the stage buffers are free now) and leaves as 16-byte stores of contiguous (BCW x TAPS)-float runs per output channel.
  const int khalf = lane >> 5, l31 = lane & 31;
  float* sT = reinterpret_cast<float*>(smem);
  __syncthreads();  // every wave is done reading the last stage
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    if (f >= ntap) break;
    const int tap = tap0 + f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) sT[(nf * 32 + 8 * q + 4 * khalf + e) * GEO::ROWP + (cf * 32 + l31) * TAPS + tap] = acc[f][4 * q + e];
  }
  __syncthreads();
  float* wsp = p.ws + (size_t)ksp * p.Cout * p.Cg * TAPS;
  const int valid = min(GEO::BCW, p.Cg - c0) * TAPS;  // floats of a row that exist (multiple of 8: Cg % 8 == 0)
  constexpr int V4 = GEO::ROWF / 4;
#pragma unroll
  for (int i = 0; i < GEO::BNW * V4 / 256; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx / V4, v = idx - row * V4;
    if (n0 + row < p.Ng && 4 * v < valid)
      *reinterpret_cast<f32x4*>(wsp + ((size_t)(g * p.Ng + n0 + row) * p.Cg + c0) * TAPS + 4 * v) = *reinterpret_cast<const f32x4*>(sT + row * GEO::ROWP + 4 * v);
  }
}

// ---------------------------------------------------------------------------------------------- 1x1, large tiles
// The 64 x 64 unit of the generic kernel gives every wave ONE accumulator fragment: 4 transpose reads per MFMA (LDS-bound
// at a quarter of the matrix rate) and both operands are re-read from L2 once per 64-channel tile of the other one.
// For 1x1 layers whose channel counts are multiples of 128:  unit = 128 output x 128 input channels, each of the 4 waves
// owns a 64 x 64 quadrant (2 x 2 fragments: 2 reads per MFMA, operands re-read half as often), a stage is 64 FLAT pixels
// (no halo: the tensor is one long [pixel][channel] matrix) of both operands, 16 KB each, double buffered.
// LDS image: rows of 256 B; the 16-byte chunk c of row r sits at position c ^ ((r & 3) << 1), applied on the DMA source
// address, so the 4 rows x 32 B that a 16-lane group of ds_read_b64_tr_b16 touches fall on 32 distinct banks.
struct Wg1 {
  static constexpr int BN = 128, BC = 128, PX = 64, RB = 256;
  static constexpr int PIECES = PX * RB / 1024;             // 16 per operand
  static constexpr int OP_BYTES = PIECES * 1024, STAGE = 2 * OP_BYTES;
  static constexpr int ROWP = BC + 4, EPI_BYTES = BN * ROWP * 4;
  static constexpr int SMEM = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad1x1_wide_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;  // this wave's 64 x 64 quadrant

  int u = blockIdx.x;
  const int ksp = u % p.ksplit; u /= p.ksplit;
  const int ct = u % p.c_tiles; u /= p.c_tiles;
  const int nt = u % p.n_tiles;
  const int g = u / p.n_tiles;
  const int n0 = nt * Wg1::BN, c0 = ct * Wg1::BC;
  const int per = (p.ntile_px + p.ksplit - 1) / p.ksplit;
  const int t_begin = ksp * per, t_end = min(t_begin + per, p.ntile_px);
  const int total_px = p.B * p.H * p.W;

  const int cabs = g * p.Cg + c0;
  const bool second = cabs >= p.C0;
  const rsrc_t rsx = second ? make_rsrc(p.x1, (size_t)total_px * p.C1 * 2) : make_rsrc(p.x0, (size_t)total_px * p.C0 * 2);
  const rsrc_t rsy = make_rsrc(p.dy, (size_t)total_px * p.Cout * 2);
  const int xstride2 = (second ? p.C1 : p.C0) * 2;
  const int xchan2 = (second ? cabs - p.C0 : cabs) * 2;
  const int ychan2 = (g * p.Ng + n0) * 2;
  // DMA: piece = 4 rows; lane -> (row in piece, position in row); it fetches the logical chunk that belongs at its position
  int drow[4], dchunk2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 4 + (lane >> 4);
    drow[i] = r;
    dchunk2[i] = ((lane & 15) ^ ((r & 3) << 1)) * 16;
  }
  auto issue = [&](int t, int st) {
    char* sb = smem + st * Wg1::STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = t * Wg1::PX + drow[i];
      const bool ok = pix < total_px;
      dma16(rsy, ok ? pix * (p.Cout * 2) + dchunk2[i] : kOob, ychan2, sb + (wave + 4 * i) * 1024);
      dma16(rsx, ok ? pix * xstride2 + dchunk2[i] : kOob, xchan2, sb + Wg1::OP_BYTES + (wave + 4 * i) * 1024);
    }
  };

  // fragment read bases: lane i of a 16-lane group supplies row (i >> 2), 8 bytes (4 channels) at column group (i & 3)
  const int gq = lane >> 4, li = lane & 15;
  const int kbase = (gq >> 1) * 8 + (li >> 2);
  const int sw = ((li >> 2) & 3) << 1;
  int a_base[2], b_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = ((wn * 2 + i) * 4 + (gq & 1) * 2 + ((li & 3) >> 1)) ^ sw;
    const int cb = ((wc * 2 + i) * 4 + (gq & 1) * 2 + ((li & 3) >> 1)) ^ sw;
    a_base[i] = kbase * Wg1::RB + ca * 16 + (li & 1) * 8;
    b_base[i] = Wg1::OP_BYTES + kbase * Wg1::RB + cb * 16 + (li & 1) * 8;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](auto stage_tag) {
    constexpr int STG = decltype(stage_tag)::value;
    const char* sb = smem + STG * Wg1::STAGE;
#pragma unroll
    for (int ks = 0; ks < Wg1::PX / 16; ++ks) {
      bf16x8 af[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = frag8(sb + a_base[i], ks * 16 * Wg1::RB, (ks * 16 + 4) * Wg1::RB);
        xf[i] = frag8(sb + b_base[i], ks * 16 * Wg1::RB, (ks * 16 + 4) * Wg1::RB);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], xf[j], acc[i][j], 0, 0, 0);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (t_begin < t_end) issue(t_begin, 0);
  for (int t = t_begin; t < t_end; t += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t_end) issue(t + 1, 1);
    compute(S0{});
    if (t + 1 >= t_end) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < t_end) issue(t + 2, 0);
    compute(S1{});
  }

  // ---- partial sums through an LDS transpose (acc[i][j][4q+e] = dW'[n = 8q + 4*khalf + e][c = lane & 31] of the fragment)
  const int khalf = lane >> 5, l31 = lane & 31;
  float* sT = reinterpret_cast<float*>(smem);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sT[(wn * 64 + i * 32 + 8 * q + 4 * khalf + e) * Wg1::ROWP + wc * 64 + j * 32 + l31] = acc[i][j][4 * q + e];
  __syncthreads();
  float* wsp = p.ws + (size_t)ksp * p.Cout * p.Cg;
#pragma unroll
  for (int i = 0; i < Wg1::BN * (Wg1::BC / 4) / 256; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx / (Wg1::BC / 4), v = idx - row * (Wg1::BC / 4);
    *reinterpret_cast<f32x4*>(wsp + ((size_t)(g * p.Ng + n0 + row) * p.Cg + c0) + 4 * v) = *reinterpret_cast<const f32x4*>(sT + row * Wg1::ROWP + 4 * v);
  }
}

// the wide 1x1 kernel needs whole 128-channel tiles on both sides and, with two sources, the split on a tile boundary
static bool wgrad1x1_wide_ok(int ks, int Ng, int Cg, int C0, int C1) {
  return ks == 1 && Ng % 128 == 0 && Cg % 128 == 0 && (C1 == 0 || (C0 % Cg) % 128 == 0);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n, int ksplit, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? dw[i] : 0.f;
  for (int k = 0; k < ksplit; ++k) s += ws[(size_t)k * n + i];
  dw[i] = s;
}

// split-K factor: enough units to fill the chip (512 workgroup slots; 768 measured 4-5 % slower on the train step: more
// partial-sum traffic for no shorter critical path), but never more partial-sum traffic than ~4x the
// gradient itself (small-M layers have large weights and few pixel tiles: they run unsplit and write dW directly)
int wgrad_ksplit(int G, int Ng, int Cg, int ks, long ntile_px, bool wide = false) {
  const int bcw = wide ? 128 : (ks == 3 ? 32 : 64);
  const long base = (long)G * ceil_div(Ng, wide ? 128 : 64) * ceil_div(Cg, bcw);
  const long want_units = 512;      // (768 measured slower: more partial-sum traffic, same critical path)
  const double cap_mb = 48.0;
  long want = std::max<long>(1, want_units / base);
  const double dw_mb = (double)G * Ng * Cg * ks * ks * 4.0 / 1e6;
  if (dw_mb * want > cap_mb) want = std::max<long>(1, (long)(cap_mb / dw_mb));
  return (int)std::min<long>(want, ntile_px);
}

int launch_wgrad1x1_wide(const WgradParams& p, hipStream_t s) {
  auto kern = conv_wgrad1x1_wide_kernel;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Wg1::SMEM) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_wgrad1x1_wide)");
    attr_done = true;
  }
  const int units = p.G * p.n_tiles * p.c_tiles * p.ksplit;
  hipLaunchKernelGGL(kern, dim3(units), dim3(256), Wg1::SMEM, s, p);
  return check_launch("conv_wgrad1x1_wide");
}

template <int KS>
int launch_wgrad(const WgradParams& p, hipStream_t s) {
  using GEO = WgGeom<KS>;
  auto kern = conv_wgrad_kernel<KS>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GEO::SMEM) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_wgrad)");
    attr_done = true;
  }
  const int units = p.G * p.n_tiles * p.c_tiles * p.ksplit;
  hipLaunchKernelGGL(kern, dim3(units), dim3(256), GEO::SMEM, s, p);
  return check_launch("conv_wgrad");
}

// ---- float32 parity path: one workgroup per (output channel, input channel of the group, tap), plain fp32 products summed over all
// pixels (block tree + fixed order).  No tiling, no matrix cores: it exists so that gradient parity can be asserted at fp32 tolerance
// (tests/test_gpu_backward.py), not for speed.
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x0, const float* __restrict__ x1,
                                                             float* __restrict__ dw, int B, int H, int W, int sH, int sW, int C0, int C1, int Cout, int G,
                                                             int KS, int up, int accumulate) {
  __shared__ float scratch[4];
  const int taps = KS * KS, pad = KS / 2;
  const int Cg = (C0 + C1) / G, Ng = Cout / G;
  const int idx = blockIdx.x;                      // ((o * Cg + c) * taps + tap)
  const int tap = idx % taps, c = (idx / taps) % Cg, o = idx / (taps * Cg);
  const int g = o / Ng, cabs = g * Cg + c;
  const float* xs = cabs < C0 ? x0 : x1;
  const int Cs = cabs < C0 ? C0 : C1, cc = cabs < C0 ? cabs : cabs - C0;
  const int dh = tap / KS - pad, dwo = tap % KS - pad;
  float acc = 0.f;
  const long npx = (long)B * H * W;
  for (long pix = threadIdx.x; pix < npx; pix += 256) {
    const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    const int ih = h + dh, iw = w + dwo;
    if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
    const int shh = up ? (ih >> 1) : ih, sww = up ? (iw >> 1) : iw;
    acc += dy[pix * Cout + o] * xs[(((long)b * sH + shh) * sW + sww) * Cs + cc];
  }
  acc = block_sum_256(acc, scratch);
  if (threadIdx.x == 0) dw[idx] = accumulate ? dw[idx] + acc : acc;
}

}  // namespace
}  // namespace ddx

using namespace ddx;

static int wgrad_f32(const ddx_wgrad_desc& d, ddx_stream stream) {
  if (!d.dy || !d.x0 || !d.dw) return set_error(DDX_ERR_ARG, "wgrad: null buffer");
  if (d.ksize != 1 && d.ksize != 3) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: ksize must be 1 or 3");
  if ((d.C1 > 0) != (d.x1 != nullptr) || d.groups <= 0 || (d.C0 + d.C1) % d.groups || d.Cout % d.groups) return set_error(DDX_ERR_ARG, "wgrad: bad channels");
  if (d.resample == DDX_RESAMPLE_DOWN) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: avg-pool gather not built");
  const ddx_wgrad_desc q = d;
  const int up = d.resample == DDX_RESAMPLE_UP;
  const size_t n = (size_t)d.Cout * ((d.C0 + d.C1) / d.groups) * d.ksize * d.ksize;
  return dispatch([q, up, n](hipStream_t s) -> int {
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3((unsigned)n), dim3(256), 0, s, (const float*)q.dy, (const float*)q.x0, (const float*)q.x1, q.dw, q.B, q.H,
                       q.W, up ? q.H / 2 : q.H, up ? q.W / 2 : q.W, q.C0, q.C1, q.Cout, q.groups, q.ksize, up, q.accumulate);
    return check_launch("conv_wgrad_f32");
  }, stream, "conv_wgrad_f32");
}

static int wgrad_fill(const ddx_wgrad_desc& d, WgradParams* pp) {
  if (!d.dy || !d.x0 || !d.dw) return set_error(DDX_ERR_ARG, "wgrad: null buffer");
  if (d.dtype != DDX_BF16) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: bf16 operands only");
  if (d.ksize != 1 && d.ksize != 3) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: ksize must be 1 or 3");
  if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.groups <= 0) return set_error(DDX_ERR_ARG, "wgrad: bad size");
  if ((d.C1 > 0) != (d.x1 != nullptr)) return set_error(DDX_ERR_ARG, "wgrad: x1/C1 mismatch");
  const int Cin = d.C0 + d.C1;
  if (Cin % d.groups || d.Cout % d.groups) return set_error(DDX_ERR_ARG, "wgrad: channels not divisible by groups");
  const int bcw = d.ksize == 3 ? 32 : 64;
  if (d.C0 % 8 || (d.x1 && d.C1 % 8) || d.Cout % 8 || (Cin / d.groups) % 8 || (d.Cout / d.groups) % 8)
    return set_error(DDX_ERR_UNSUPPORTED, "wgrad: channel counts (per group) must be multiples of 8");
  // an input-channel tile (bcw channels from g*Cg + k*bcw) must lie inside ONE source
  if (d.x1 && ((d.C0 % (Cin / d.groups)) % bcw)) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: the source split must fall on an input-channel tile");
  if (d.resample == DDX_RESAMPLE_DOWN) return set_error(DDX_ERR_UNSUPPORTED, "wgrad: avg-pool gather not built");
  if (d.resample == DDX_RESAMPLE_UP && ((d.H | d.W) & 1)) return set_error(DDX_ERR_ARG, "wgrad: upsampled size must be even");
  WgradParams p{};
  p.dy = d.dy; p.x0 = d.x0; p.x1 = d.x1; p.ws = d.workspace ? d.workspace : d.dw;
  p.B = d.B; p.H = d.H; p.W = d.W;
  p.sH = d.resample == DDX_RESAMPLE_UP ? d.H / 2 : d.H; p.sW = d.resample == DDX_RESAMPLE_UP ? d.W / 2 : d.W;
  p.C0 = d.C0; p.C1 = d.C1; p.Cin = Cin; p.Cout = d.Cout; p.G = d.groups; p.Cg = Cin / d.groups; p.Ng = d.Cout / d.groups;
  p.resample = d.resample;
  if ((size_t)p.B * p.H * p.W * std::max(p.Cout, std::max(p.C0, p.C1)) * 2 >= (size_t)0x7fff0000)
    return set_error(DDX_ERR_UNSUPPORTED, "wgrad: tensor too large for 32-bit buffer offsets");
  // (measured on MI355X: 1.4-1.65x faster from 5.5k pixels up, slower at 1.4k pixels where few, short units remain)
  p.wide = d.resample == DDX_RESAMPLE_KEEP && (long)p.B * p.H * p.W >= 4096 && wgrad1x1_wide_ok(d.ksize, p.Ng, p.Cg, p.C0, p.C1) ? 1 : 0;
  if (p.wide) {
    p.tiles_h = p.tiles_w = 0; p.ntile_px = ceil_div(p.B * p.H * p.W, Wg1::PX);
    p.n_tiles = p.Ng / 128; p.c_tiles = p.Cg / 128;
  } else {
    p.tiles_h = ceil_div(p.H, kTH); p.tiles_w = ceil_div(p.W, kTW); p.ntile_px = p.B * p.tiles_h * p.tiles_w;
    p.n_tiles = ceil_div(p.Ng, 64); p.c_tiles = ceil_div(p.Cg, bcw);
  }
  p.ksplit = wgrad_ksplit(p.G, p.Ng, p.Cg, d.ksize, p.ntile_px, p.wide != 0);
  *pp = p;
  return 0;
}

extern "C" size_t ddx_wgrad_workspace_bytes(const ddx_wgrad_desc* dp) {
  if (!dp) return 0;
  if (dp->dtype == DDX_F32) return 16;   // (the fp32 parity kernel needs none; non-zero = supported)
  WgradParams p{};
  ddx_wgrad_desc d = *dp;
  static float dummy;
  if (!d.dw) d.dw = &dummy;
  if (!d.dy) d.dy = &dummy;
  if (!d.x0) d.x0 = &dummy;
  if (d.C1 > 0 && !d.x1) d.x1 = &dummy;
  if (wgrad_fill(d, &p) != 0) return 0;
  return (size_t)p.ksplit * p.Cout * p.Cg * d.ksize * d.ksize * sizeof(float);
}

extern "C" int32_t ddx_wgrad_parts(const ddx_wgrad_desc* dp) {
  if (!dp || dp->dtype != DDX_BF16) return 0;
  WgradParams p{};
  ddx_wgrad_desc d = *dp;
  static float dummy;
  if (!d.dw) d.dw = &dummy;
  if (!d.dy) d.dy = &dummy;
  if (!d.x0) d.x0 = &dummy;
  if (d.C1 > 0 && !d.x1) d.x1 = &dummy;
  if (wgrad_fill(d, &p) != 0) return 0;
  return p.ksplit;
}

extern "C" int32_t ddx_wgrad_parts_max(int32_t Cout, int32_t Cg, int32_t groups, int32_t ksize) {
  if (Cout <= 0 || Cg <= 0 || groups <= 0 || Cout % groups || (ksize != 1 && ksize != 3)) return 0;
  const int Ng = Cout / groups;
  // (the two tilings of a 1x1 layer have different unit counts: the bound covers both)
  const int a = wgrad_ksplit(groups, Ng, Cg, ksize, 1l << 40, false);
  const int b = ksize == 1 && Ng % 128 == 0 && Cg % 128 == 0 ? wgrad_ksplit(groups, Ng, Cg, ksize, 1l << 40, true) : 1;
  return std::max(a, b);
}

extern "C" int ddx_mpconv2d_wgrad(const ddx_wgrad_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "wgrad: null descriptor");
  const ddx_wgrad_desc d = *dp;
  if (d.dtype == DDX_F32) return wgrad_f32(d, stream);
  WgradParams p{};
  if (int rc = wgrad_fill(d, &p)) return rc;
  if (!d.workspace) return set_error(DDX_ERR_ARG, "wgrad: workspace missing (ddx_wgrad_workspace_bytes)");
  const int ks = d.ksize;
  const size_t n = (size_t)p.Cout * p.Cg * ks * ks;
  float* dw = d.dw;
  const int accumulate = d.accumulate;
  const double flops = 2.0 * p.B * p.H * p.W * (double)p.Cout * p.Cg * ks * ks;
  const double bytes = 2.0 * ((double)p.B * p.H * p.W * p.Cout + (double)p.B * p.sH * p.sW * p.Cin) + 4.0 * n;
  if (accumulate < 0 || accumulate > 2) return set_error(DDX_ERR_ARG, "wgrad: accumulate");
  return dispatch([p, ks, n, dw, accumulate](hipStream_t s) -> int {
    if (accumulate == 2)   // the split-K slices stay in the workspace: the consumer adds them (ddx_wpath_job.dwp_parts)
      return p.wide ? launch_wgrad1x1_wide(p, s) : (ks == 3 ? launch_wgrad<3>(p, s) : launch_wgrad<1>(p, s));
    if (p.ksplit == 1 && !accumulate) {  // unsplit: the kernel writes the gradient itself
      WgradParams q = p;
      q.ws = dw;
      return q.wide ? launch_wgrad1x1_wide(q, s) : (ks == 3 ? launch_wgrad<3>(q, s) : launch_wgrad<1>(q, s));
    }
    const int rc = p.wide ? launch_wgrad1x1_wide(p, s) : (ks == 3 ? launch_wgrad<3>(p, s) : launch_wgrad<1>(p, s));
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)p.ws, dw, n, p.ksplit, accumulate);
    return check_launch("wgrad_reduce");
  }, stream, ks == 3 ? "conv3x3_wgrad" : "conv1x1_wgrad", flops, bytes);
}
