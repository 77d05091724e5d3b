// Magnitude-preserving 1x1 conv2d forward for the SMALL-M layers of the UNet (level 4 of the default model: 86 pixels per image, 344 at
// B = 4; reference src/modules/unets/unet_edm2_b4.py:110-158 at H x W = 2 x 43: conv_skip, attn_proj): an implicit GEMM built around
// latency, not around tile throughput.
//
// What bounds these layers is the number of SERIAL memory round trips inside a launch (1-2 us of roofline work took 12-19 us: a K
// loop of dependent chunks on the register-staged kernel, a ring of four weight fragments per wave in the round-2 version of this
// file).  Round-4 structure: a workgroup is EIGHT waves that split K eight ways, and every byte the workgroup needs is requested in
// ONE burst before anything waits:
//   * epilogue operands (residual rows, channel scales) first: their latency hides behind everything else;
//   * the pixel rows of the tile (whole 128-byte lines, all input channels of a K pass) go into LDS by LDS-DMA (buffer_load ... lds,
//     zero padding by out-of-range offsets, rows padded by one 16-byte slot so that fragment reads are bank-conflict free, one
//     all-zero row); layers with more than 80 sixteen-channel chunks per pixel tile run two or three K passes;
//   * weights never touch LDS: prepared with 16-channel chunks (wp[g][c16][tap][NgP][16]) a 32-row x 16-channel MFMA A fragment is
//     ONE contiguous 1 KiB block; a wave loads ALL fragments of its K share (<= 10 steps) straight into the A operand registers,
//     BEHIND the rows in the memory queue -- a counted s_waitcnt + raw s_barrier release the matrix loop when the rows have
//     landed, the weight fragments are waited for one by one;
//   * the matrix loop is branch-free (steps past a wave's share multiply re-read weights with the zero row) with a ring of three
//     fragment register sets; the eight partial tiles are summed through LDS inside the fused epilogue (fixed order: deterministic).
// Measured (round 4, tools/sm_ablate.sh): an EMPTY launch of this shape (no loads, no MFMA, only the epilogue's dependent load ->
// store chain) costs 4.6-6.4 us; the level-4 layers run 7.2-15.6 us.  A 3x3 variant of the same structure (slab of the whole image
// in LDS, 23 weight fragments per wave) measured 8.3-23.7 us against 9.8-19.7 us on the register-staged kernel and lost 0.12 ms on the
// whole step (it also moves the merged qkv conv here: 21-25 us against 19-22): removed, DESIGN.md section 6d.
// Operands are raw (producer-side activation); `src0_alt` selects a second source per output-channel tile.
// Workgroups that read the same weights are laid on the same XCD (observed block -> XCD map b % 8: speed only).
#include <algorithm>
#include <cstdlib>

#include "conv_params.hpp"

namespace ddx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOob = 0x7fffff00;  // voffset of a lane that must read zeros
constexpr int kWaves = 8, kThreads = kWaves * 64;

__device__ __forceinline__ void dma16(rsrc_t rs, int voff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)std::min<size_t>(bytes, 0x7ffffff0u), 0x00020000);
}
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

struct SmArgs {
  int V, slots;                          // 16-byte slots per staged row (without the pad slot), slots of the staged rows
  int SC, ppw;                           // 16-channel chunks per K pass, chunks per wave
  int npass, SCtot;                      // K passes through LDS, chunks of the whole layer
  int PT, WT;                            // pixel tiles, weight (channel) tiles
  int M, HW;                             // B*H*W, H*W
  int dbg;                               // DDX_ABLATE ablation bits (timing experiments; wrong results): 1 no weight loads, 2 no operand DMA, 4 no MFMA, 8 no reduction
  float inv_V1, inv_HW, inv_W;
};

// ---------------------------------------------------------------------------------------------------------------- epilogue
// Cross-wave K reduction (every wave writes its partial tile, the store phase sums the eight in a fixed order) + fused element-wise
// tail, coalesced on NHWC rows.  pix_of(ml) = global output pixel of tile row ml, or -1.
// Epilogue operands of a thread's items, requested at kernel start so that their latency hides behind the operand burst.
template <int EI>
struct SmEpiPre {
  long eoff[EI];
  bf16x4 rres[EI];
  f32x4 ecs[EI], ecs2[EI];
};

template <int PF, int NF, typename PixFn>
__device__ __forceinline__ void sm_epilogue_prefetch(const ConvParams& p, const SmArgs& a, int g, int n0, PixFn pix_of,
                                                     SmEpiPre<(PF * 32 * NF * 8 + kThreads - 1) / kThreads>& e) {
  constexpr int BM = PF * 32, BN = NF * 32, G4 = BN / 4, EI = (BM * G4 + kThreads - 1) / kThreads;
  const int tid = threadIdx.x;
  const bf16* res = reinterpret_cast<const bf16*>(p.res);
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = tid + it * kThreads;
    const int ml = idx / G4, c4 = idx % G4;
    const int n = n0 + c4 * 4;
    const int pix = (idx < BM * G4 && n < p.Ng) ? pix_of(ml) : -1;
    e.eoff[it] = pix >= 0 ? (long)pix * p.Cout + (size_t)g * p.Ng + n : -1;
    const int b = fdiv(max(pix, 0), a.inv_HW);
    long roff = e.eoff[it] < 0 ? 0 : e.eoff[it];
    if (p.res_up) {   // residual at half size: pixel (b, h, w) reads (b, h / 2, w / 2)
      const int hw = max(pix, 0) - b * p.H * p.W, h = fdiv(hw, a.inv_W), w = hw - h * p.W;
      roff = pix >= 0 ? (long)((b * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Cout + (size_t)g * p.Ng + n : 0;
    }
    const int ebc = b * p.Cout + min(g * p.Ng + n, p.Cout - 4);
    if (p.epilogue == DDX_EPI_MPSUM) e.rres[it] = *reinterpret_cast<const bf16x4*>(res + roff);
    if (p.out_cs) e.ecs[it] = *reinterpret_cast<const f32x4*>(p.out_cs + ebc);
    if (p.out2_cs) e.ecs2[it] = *reinterpret_cast<const f32x4*>(p.out2_cs + ebc);
  }
}

// Cross-wave K reduction (every wave writes its partial tile, the store phase sums the eight in a fixed order) + fused element-wise
// tail, coalesced on NHWC rows.
template <int PF, int NF>
__device__ __forceinline__ void sm_epilogue(const ConvParams& p, const SmArgs& a, f32x16 (&acc)[NF][PF], char* smem, int q,
                                            const SmEpiPre<(PF * 32 * NF * 8 + kThreads - 1) / kThreads>& pre) {
  constexpr int BM = PF * 32, BN = NF * 32, ES = BN + 4, G4 = BN / 4, EI = (BM * G4 + kThreads - 1) / kThreads;
  const int tid = threadIdx.x, lane = tid & 63, khalf = lane >> 5, l31 = lane & 31;
  bf16* out = reinterpret_cast<bf16*>(p.out);
  __syncthreads();  // every wave is done reading the staged operand the partial tiles overlay
  float* sE = reinterpret_cast<float*>(smem) + (size_t)q * BM * ES;
  if (!(a.dbg & 8) || q == 0) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int ml = j * 32 + l31;
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          f32x4 y4;
#pragma unroll
          for (int e = 0; e < 4; ++e) y4[e] = acc[i][j][4 * k4 + e];
          *reinterpret_cast<f32x4*>(sE + (size_t)ml * ES + i * 32 + 8 * k4 + 4 * khalf) = y4;
        }
    }
  }
  __syncthreads();
  const float* sE0 = reinterpret_cast<const float*>(smem);
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = min(tid + it * kThreads, BM * G4 - 1);
    const int ml = idx / G4, c4 = idx % G4;
    f32x4 part[kWaves];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) part[w] = *reinterpret_cast<const f32x4*>(sE0 + (size_t)((a.dbg & 8) ? 0 : w) * BM * ES + (size_t)ml * ES + c4 * 4);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      y[e] = ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]));
    if (p.epilogue == DDX_EPI_MPSUM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (float)pre.rres[it][e] * p.res_a + y[e] * p.res_b;
    }
    if (p.clip > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fminf(fmaxf(y[e], -p.clip), p.clip);
    }
    if (pre.eoff[it] < 0) continue;
    if (p.out2) {
      bf16x4 tv;
      if (p.out2_linear) {  // scaled twin y * c2[b][c] (operand of attn_qk)
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)(y[e] * pre.ecs2[it][e]);
      } else if (p.out_cs && !p.out_act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)mp_silu_f(y[e] * pre.ecs[it][e] * p.out2_scale);
      } else {  // activated twin for the next block's conv_res0
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)mp_silu_f(y[e] * p.out2_scale);
      }
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.out2) + pre.eoff[it]) = tv;
    }
    if (p.out_act) {  // producer-side mp_silu(y * c)
      if (p.out_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] *= pre.ecs[it][e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = mp_silu_f(y[e]);
    }
    bf16x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (bf16)y[e];
    *reinterpret_cast<bf16x4*>(out + pre.eoff[it]) = ov;
  }
}

// work decode shared by both kernels: weight tiles wt = 8 k + xcd stay on one XCD for all their pixel tiles
__device__ __forceinline__ bool sm_decode(const SmArgs& a, int* wt, int* px) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int wtq = slot / a.PT;
  *wt = wtq * 8 + xcd;
  *px = slot - wtq * a.PT;
  return *wt < a.WT;
}

// one staged 16-byte slot: source select between the two tensors of an mp_cat (complementary lane masks when a piece straddles them)
__device__ __forceinline__ void stage_slot(const ConvParams& p, rsrc_t rs0, rsrc_t rs1, bool ok, int pix, int cabs, char* dst, bool one_src,
                                           bool only1) {
  if (one_src) {
    const int voff = ok ? (only1 ? pix * p.C1 + (cabs - p.C0) : pix * p.C0 + cabs) * 2 : kOob;
    dma16(only1 ? rs1 : rs0, voff, dst);
  } else {
    const bool first = cabs < p.C0;
    if (first || !ok) dma16(rs0, ok ? (pix * p.C0 + cabs) * 2 : kOob, dst);
    if (!first && ok) dma16(rs1, (pix * p.C1 + (cabs - p.C0)) * 2, dst);
  }
}

// ---------------------------------------------------------------------------------------------------------------- 1x1
// Tile = PF * 32 consecutive pixels of the flat pixel list x NF * 32 output channels.  Per K pass: the pass's channels of the tile's
// pixel rows go into LDS (plus one all-zero row), wave q loads the weight fragments of chunks [q ppw, (q + 1) ppw) of the pass; the
// matrix loop is branch-free: steps past a wave's share multiply (finite, re-read) weights with the zero row.
template <int PF, int NF, int NP>
__global__ __launch_bounds__(kThreads, 2) void conv_sm1_kernel(const ConvParams p, const SmArgs a) {
  constexpr int BM = PF * 32, BN = NF * 32, EI = (BM * NF * 8 + kThreads - 1) / kThreads;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  int wt, px;
  if (!sm_decode(a, &wt, &px)) return;
  const int n0 = wt * BN;           // one group
  const int m0 = px * BM;

  SmEpiPre<EI> pre;
  sm_epilogue_prefetch<PF, NF>(p, a, 0, n0, [&](int ml) { const int m = m0 + ml; return m < a.M ? m : -1; }, pre);

  // output-channel tiles below pro_rows read the alternative source (x * c twin)
  const void* s0p = (p.src0_alt && n0 < p.pro_rows) ? p.src0_alt : p.src0;
  const rsrc_t rs0 = make_rsrc(s0p, (size_t)p.B * p.sH * p.sW * p.C0 * 2);
  const rsrc_t rs1 = p.src1 ? make_rsrc(p.src1, (size_t)p.B * p.sH * p.sW * p.C1 * 2) : rs0;
  const bf16* wp = reinterpret_cast<const bf16*>(p.wp);
  const bf16* wlane[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) wlane[i] = wp + (size_t)min(n0 + i * 32 + l31, p.NgP - 1) * 16 + khalf * 8;

  f32x16 acc[NF][PF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int V = a.V, rstride = (V + 1) * 8;   // elements
  const int npieces = (a.slots + 63) >> 6;
  const bf16* sA = reinterpret_cast<const bf16*>(smem);
  for (int pass = 0; pass < a.npass; ++pass) {
    const int pc0 = pass * a.SC;                    // first chunk of the pass
    const int pcn = min(a.SC, a.SCtot - pc0);       // chunks of this pass
    if (pass > 0) __syncthreads();   // every wave is done with the previous pass's rows
    // ---- pixel rows of the tile, channels of the pass: slot sidx = r * (V + 1) + v (v == V: pad slot), 64 consecutive slots per wave
    // instruction; row BM is the zero row
    {
      const int c_lo = pc0 * 16, c_hi = (pc0 + pcn) * 16;
      const bool one_src = p.src1 == nullptr || c_hi <= p.C0 || c_lo >= p.C0;   // (workgroup-uniform)
      const bool only1 = p.src1 != nullptr && c_lo >= p.C0;
      for (int piece = q; piece < ((a.dbg & 2) ? 0 : npieces); piece += kWaves) {
        const int sidx = piece * 64 + lane;
        const int r = fdiv(sidx, a.inv_V1), v = sidx - r * (V + 1);
        const int m = m0 + r;
        const int cabs = c_lo + v * 8;
        const bool ok = r < BM && v < V && m < a.M && cabs < c_hi;
        int sp = m;
        if (p.resample == DDX_RESAMPLE_UP) {
          const int b = fdiv(m, a.inv_HW), hw = m - b * a.HW;
          const int h = fdiv(hw, a.inv_W), w = hw - h * p.W;
          sp = (b * p.sH + (h >> 1)) * p.sW + (w >> 1);
        }
        stage_slot(p, rs0, rs1, ok, sp, cabs, smem + piece * 1024, one_src, only1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's weight fragments, all in flight behind the rows
    bf16x8 wr[NP][NF];
    if (!(a.dbg & 1)) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = min(pc0 + q * a.ppw + i, a.SCtot - 1);
        const size_t woff = (size_t)c * p.NgP * 16;
#pragma unroll
        for (int n = 0; n < NF; ++n) wr[i][n] = *reinterpret_cast<const bf16x8*>(wlane[n] + woff);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int e = 0; e < 8; ++e) wr[i][n][e] = (bf16)1.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    // the rows (issued first: memory returns in order) have landed when at most the weight loads are outstanding
    // (raw barrier: __syncthreads() would drain vmcnt to 0 because LDS-DMA writes are pending LDS stores to the compiler's fence)
    if (a.dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * NF) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 4)) {
      // fragments of step i + 2 are read while step i multiplies (ring of three register sets)
      bf16x8 xf[3][PF];
      auto read_x = [&](int i, int u) {
        const int cl = q * a.ppw + i;   // chunk inside the pass
        const bool valid = i < a.ppw && cl < pcn;   // (wave-uniform)
        const int clc = valid ? cl : 0;
#pragma unroll
        for (int j = 0; j < PF; ++j) xf[u][j] = *reinterpret_cast<const bf16x8*>(sA + (size_t)(valid ? j * 32 + l31 : BM) * rstride + clc * 16 + khalf * 8);
      };
      read_x(0, 0);
      read_x(1, 1);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i + 2 < NP) read_x(i + 2, (i + 2) % 3);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int j = 0; j < PF; ++j) acc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[i][n], xf[i % 3][j], acc[n][j], 0, 0, 0);
      }
    }
  }
  sm_epilogue<PF, NF>(p, a, acc, smem, q, pre);
}

// ------------------------------------------------------------------------------------------- host side

struct SmPlan { SmArgs a; int PF, NF; size_t smem; long wgs; };
constexpr size_t kLdsMax = 156 * 1024;
constexpr int kNP1 = 10;   // weight fragments per wave and K pass

int env_int(const char* name) { const char* e = std::getenv(name); return e ? atoi(e) : 0; }

bool sm_plan(const ConvParams& p, SmPlan* out) {
  static const int dbg = env_int("DDX_ABLATE");   // timing ablations only (tools/sm_ablate.sh)
  if (p.G != 1) return false;
  SmPlan best{}; double best_cost = 1e30; bool found = false;
  for (int PF = 1; PF <= 2; ++PF) {
    for (int NF = 1; NF <= 2; ++NF) {
      if (NF == 2 && p.Ng <= 32) continue;
      SmArgs a{};
      a.dbg = dbg;
      const int BM = PF * 32, BN = NF * 32;
      const size_t red = (size_t)kWaves * BM * (BN + 4) * sizeof(float);
      a.HW = p.H * p.W; a.M = p.B * a.HW;
      a.inv_HW = 1.0f / (float)a.HW; a.inv_W = 1.0f / (float)p.W;
      a.WT = ceil_div(p.Ng, BN);
      a.SCtot = p.Cin / 16;
      // K passes: <= 8 * kNP1 chunks per pass, and the staged rows (one pad slot each) must fit LDS
      const int max_chunks_lds = (int)((kLdsMax / ((size_t)(BM + 1) * 16) - 1) / 2);
      const int max_chunks = std::min(kWaves * kNP1, max_chunks_lds);
      if (max_chunks < 8) continue;
      a.npass = ceil_div(a.SCtot, max_chunks);
      a.SC = ceil_div(a.SCtot, a.npass);
      a.ppw = ceil_div(a.SC, kWaves);
      a.V = a.SC * 2;
      a.slots = (BM + 1) * (a.V + 1);   // + the zero row
      a.inv_V1 = 1.0f / (float)(a.V + 1);
      a.PT = ceil_div(a.M, BM);
      const double util = (double)a.M / ((double)a.PT * BM);
      const size_t smem = std::max(red, (size_t)round_up(a.slots, 64) * 16);
      if (smem > kLdsMax) continue;
      const long wgs = (long)a.PT * a.WT;
      // relative cost (us): rounds of resident workgroups (one per CU) x (bytes a workgroup pulls at ~100 GB/s + matrix steps + fixed latency)
      const double rounds = std::ceil((double)wgs / 256.0);
      const double bytes = 2.0 * ((double)BN * p.Cin + (double)BM * p.Cin);
      const double mfma_us = (double)a.ppw * a.npass * PF * NF * 32.0 / 2100.0;
      double cost = rounds * (bytes / 100e3 + mfma_us + 3.0) / std::max(util, 0.1);
      // measured (MI355X, tools/conv_bench.py --cases small --path sm with the tile forced): 64-pixel x 32-channel tiles beat the 32 x 64
      // ones of equal byte count (level-4 proj 10.7 vs 14.4 us, skip over mp_cat 15.6 vs 21.7: 20 weight fragments per wave in flight
      // are slower than 10 + a second operand pass)
      if (NF == 2) cost *= 1.3;
      if (cost < best_cost) { best_cost = cost; best = SmPlan{a, PF, NF, smem, wgs}; found = true; }
    }
  }
  if (found) *out = best;
  return found;
}

template <int PF, int NF>
int launch_sm(const ConvParams& p, const SmPlan& pl, hipStream_t s) {
  static bool attr_done = false;
  auto kern = conv_sm1_kernel<PF, NF, kNP1>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_sm)");
    attr_done = true;
  }
  const int grid = pl.a.PT * round_up(pl.a.WT, 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), pl.smem, s, p, pl.a);
  return check_launch("conv_sm");
}

}  // namespace

bool conv_sm_supported(const ConvParams& p, int ksize, int dtype) {
  if (dtype != DDX_BF16 || ksize != 1 || p.CK != 16) return false;
  if (p.resample == DDX_RESAMPLE_DOWN || p.reflect_w || p.swap1 || p.paired) return false;
  if (p.prologue != DDX_PRO_NONE || p.scale0 != 1.0f || (p.src1 && p.scale1 != 1.0f)) return false;   // raw operands only
  if (p.Cg % 16 || p.C0 % 16 || (p.src1 && p.C1 % 16)) return false;
  if (p.Ng % 4 || p.Cout % 4) return false;
  if (p.epilogue != DDX_EPI_STORE && p.epilogue != DDX_EPI_MPSUM) return false;
  if (p.src0_alt && (p.pro_rows <= 0 || p.pro_rows % 64)) return false;
  if ((size_t)p.B * p.sH * p.sW * std::max(p.C0, p.C1) * 2 >= (size_t)0x7fffff00) return false;
  SmPlan pl;
  return sm_plan(p, &pl);
}

int launch_conv_sm(const ConvParams& p, int ksize, hipStream_t s) {
  SmPlan pl;
  if (ksize != 1 || !sm_plan(p, &pl)) return set_error(DDX_ERR_UNSUPPORTED, "conv_sm: no tile fits LDS");
  if (pl.PF == 1) return pl.NF == 1 ? launch_sm<1, 1>(p, pl, s) : launch_sm<1, 2>(p, pl, s);
  return pl.NF == 1 ? launch_sm<2, 1>(p, pl, s) : launch_sm<2, 2>(p, pl, s);
}

}  // namespace ddx
