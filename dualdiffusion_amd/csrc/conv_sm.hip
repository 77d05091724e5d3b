// Magnitude-preserving conv2d forward for the SMALL-M layers of the UNet (levels 3 / 4 of the default model: 344 or 86 pixels per
// image, 83 % of the weights, reference src/modules/unets/unet_edm2_b4.py:110-158 at H x W = 4 x 86 / 2 x 43): weight-streaming
// implicit GEMMs built around latency, not around tile throughput.
//
// What bounds these layers is the number of SERIAL memory round trips inside a launch (2-5 us of roofline work took 11-25 us on
// the register-staged kernel: K loop of dependent chunks, 4-step LDS split-K reduction, residual and scale loads behind the
// matrix phase).  Here every byte a workgroup needs is requested as early as the hardware lets it and nothing waits on a
// workgroup barrier inside the K loop:
//   * weights never touch LDS: prepared with 16-channel chunks (wp[g][c16][tap][NgP][16]) a 32-row x 16-channel MFMA A fragment is
//     ONE contiguous 1 KiB block, loaded straight into the A operand registers by a per-wave ring that runs D steps ahead;
//   * the 4 waves of a workgroup split K, each wave's accumulators cover the whole (32 PF pixels) x (32 NF channels) tile and
//     nothing is exchanged until ONE two-round reduction through LDS at the end;
//   * 1x1 layers (conv_sm1): the activation fragments go global -> registers too (a lane = one pixel, 16 bytes of its channel row
//     per step; a wave owns a contiguous quarter of the channels, i.e. whole 128-byte lines) -- no LDS, no barrier until the end;
//   * 3x3 layers (conv_sm3): the halo tile of ALL channels of the group is brought into LDS once by LDS-DMA (buffer_load ... lds,
//     every piece in flight at the same time, zero padding by out-of-range offsets, rows padded by one 16-byte slot so that
//     fragment reads are bank-conflict free), ONE barrier, then nine taps of matrix work per staged byte;
//   * residual rows and channel scales of the epilogue are requested before the reduction.
// Operands are raw (producer-side activation); the one per-channel prologue of these levels, attn_qk reading
// x * c, is served by a scaled twin written by the producing conv (`out2_linear`) and selected per output-channel tile (`src0_alt`).
// Workgroups that read the same weights are laid on the same XCD (observed block -> XCD map b % 8: speed only).
#include <algorithm>
#include <cstdlib>

#include "conv_params.hpp"

namespace ddx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOob = 0x7fffff00;  // voffset of a lane that must read zeros

__device__ __forceinline__ void dma16(rsrc_t rs, int voff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)std::min<size_t>(bytes, 0x7ffffff0u), 0x00020000);
}
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

struct SmArgs {
  int TH, TW, TWP, tiles_h, tiles_w, R;  // 3x3: pixel tile inside an image, staged rows (halo included)
  int V, slots;                          // 3x3: 16-byte slots per staged row (without the pad slot), slots of the slab
  int nsteps;                            // K steps per wave (3x3: ceil(9 * Cg/16 / 4); 1x1: Cin/16 / 4)
  int PT, WT, ntiles;                    // pixel tiles, weight tiles (= G * ntiles), channel tiles per group
  int M, HW;                             // 1x1: B*H*W, H*W
  float inv_TW, inv_TWP, inv_V1, inv_ks16, inv_HW, inv_W;
};

// ---------------------------------------------------------------------------------------------------------------- epilogue
// Cross-wave K reduction (waves 0, 1 write two fp32 regions, waves 2, 3 add) + fused element-wise tail, coalesced on NHWC rows.
// pix_of(ml) = global output pixel of tile row ml, or -1.
template <int PF, int NF, typename PixFn>
__device__ __forceinline__ void sm_epilogue(const ConvParams& p, const SmArgs& a, f32x16 (&acc)[NF][PF], char* smem, int q, int g, int n0,
                                            PixFn pix_of, bool slab_in_use) {
  constexpr int BM = PF * 32, BN = NF * 32, ES = BN + 4, G4 = BN / 4, EI = (BM * G4 + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, khalf = lane >> 5, l31 = lane & 31;
  bf16* out = reinterpret_cast<bf16*>(p.out);
  const bf16* res = reinterpret_cast<const bf16*>(p.res);
  long eoff[EI];
  [[maybe_unused]] long roff[EI];
  int ebc[EI];
  bf16x4 rres[EI];
  f32x4 ecs[EI], ecs2[EI];
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = tid + it * 256;
    const int ml = idx / G4, c4 = idx % G4;
    const int n = n0 + c4 * 4;
    const int pix = (idx < BM * G4 && n < p.Ng) ? pix_of(ml) : -1;
    eoff[it] = pix >= 0 ? (long)pix * p.Cout + (size_t)g * p.Ng + n : -1;
    const int b = fdiv(max(pix, 0), a.inv_HW);
    if (p.res_up) {   // residual at half size: pixel (b, h, w) reads (b, h / 2, w / 2)
      const int hw = max(pix, 0) - b * p.H * p.W, h = fdiv(hw, a.inv_W), w = hw - h * p.W;
      roff[it] = pix >= 0 ? (long)((b * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Cout + (size_t)g * p.Ng + n : 0;
    }
    ebc[it] = b * p.Cout + min(g * p.Ng + n, p.Cout - 4);
  }
  if (p.epilogue == DDX_EPI_MPSUM) {
#pragma unroll
    for (int it = 0; it < EI; ++it) rres[it] = *reinterpret_cast<const bf16x4*>(res + (p.res_up ? roff[it] : (eoff[it] < 0 ? 0 : eoff[it])));
  }
  if (p.out_cs) {
#pragma unroll
    for (int it = 0; it < EI; ++it) ecs[it] = *reinterpret_cast<const f32x4*>(p.out_cs + ebc[it]);
  }
  if (p.out2_cs) {
#pragma unroll
    for (int it = 0; it < EI; ++it) ecs2[it] = *reinterpret_cast<const f32x4*>(p.out2_cs + ebc[it]);
  }
  if (slab_in_use) __syncthreads();  // every wave is done reading the activation slab the regions overlay
  float* sE = reinterpret_cast<float*>(smem) + (size_t)(q & 1) * BM * ES;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if ((q >> 1) == round) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int ml = j * 32 + l31;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            float* dst = sE + (size_t)ml * ES + i * 32 + 8 * k4 + 4 * khalf;
            f32x4 y4;
#pragma unroll
            for (int e = 0; e < 4; ++e) y4[e] = acc[i][j][4 * k4 + e];
            if (round == 1) {
              const f32x4 o4 = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
              for (int e = 0; e < 4; ++e) y4[e] += o4[e];
            }
            *reinterpret_cast<f32x4*>(dst) = y4;
          }
      }
    }
    __syncthreads();
  }
  const float* sE0 = reinterpret_cast<const float*>(smem);
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = min(tid + it * 256, BM * G4 - 1);
    const int ml = idx / G4, c4 = idx % G4;
    const f32x4 ya = *reinterpret_cast<const f32x4*>(sE0 + (size_t)ml * ES + c4 * 4);
    const f32x4 yb = *reinterpret_cast<const f32x4*>(sE0 + (size_t)BM * ES + (size_t)ml * ES + c4 * 4);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = ya[e] + yb[e];
    if (p.epilogue == DDX_EPI_MPSUM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (float)rres[it][e] * p.res_a + y[e] * p.res_b;
    }
    if (p.clip > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fminf(fmaxf(y[e], -p.clip), p.clip);
    }
    if (eoff[it] < 0) continue;
    if (p.out2) {
      bf16x4 tv;
      if (p.out2_linear) {  // scaled twin y * c2[b][c] (operand of attn_qk)
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)(y[e] * ecs2[it][e]);
      } else if (p.out_cs && !p.out_act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)mp_silu_f(y[e] * ecs[it][e] * p.out2_scale);
      } else {  // activated twin for the next block's conv_res0
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = (bf16)mp_silu_f(y[e] * p.out2_scale);
      }
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.out2) + eoff[it]) = tv;
    }
    if (p.out_act) {  // producer-side mp_silu(y * c)
      if (p.out_cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] *= ecs[it][e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = mp_silu_f(y[e]);
    }
    bf16x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (bf16)y[e];
    *reinterpret_cast<bf16x4*>(out + eoff[it]) = ov;
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

// ---------------------------------------------------------------------------------------------------------------- 1x1
// D steps of {NF weight fragments, PF activation fragments} in flight per wave; wave q owns channels [q K/4, (q+1) K/4).
template <int PF, int NF, int D, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_sm1_kernel(const ConvParams p, const SmArgs a) {
  constexpr int BM = PF * 32, BN = NF * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  int wt, px;
  if (!sm_decode(a, &wt, &px)) return;
  const int n0 = wt * BN;           // one group
  const int m0 = px * BM;

  // output-channel tiles below pro_rows read the alternative source (x * c twin)
  const bf16* s0p = reinterpret_cast<const bf16*>((p.src0_alt && n0 < p.pro_rows) ? p.src0_alt : p.src0);
  const bf16* s1p = reinterpret_cast<const bf16*>(p.src1);
  const bf16* wp = reinterpret_cast<const bf16*>(p.wp);
  const bf16* wlane[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) wlane[i] = wp + (size_t)min(n0 + i * 32 + l31, p.NgP - 1) * 16 + khalf * 8;
  // this lane's pixels (rows past M read the tile's first pixel; their results are never stored)
  int xoff0[PF], xoff1[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    int m = m0 + j * 32 + l31;
    if (m >= a.M) m = m0;
    int sp = m;
    if (p.resample == DDX_RESAMPLE_UP) {
      const int b = fdiv(m, a.inv_HW), r = m - b * a.HW;
      const int h = fdiv(r, a.inv_W), w = r - h * p.W;
      sp = (b * p.sH + (h >> 1)) * p.sW + (w >> 1);
    }
    xoff0[j] = sp * p.C0 + khalf * 8;
    xoff1[j] = sp * p.C1 + khalf * 8 - p.C0;   // (+ channel: second source starts at channel C0)
  }
  const int ks0 = q * a.nsteps;  // first 16-channel step of this wave

  bf16x8 wr[D][NF], xr[D][PF];
  auto issue = [&](int u, int t) {   // step t of this wave into ring slot u (steps past the end re-read the last one; unused)
    const int ks = ks0 + min(t, a.nsteps - 1);
    const int c = ks * 16;
    const size_t woff = (size_t)ks * p.NgP * 16;
#pragma unroll
    for (int i = 0; i < NF; ++i) wr[u][i] = *reinterpret_cast<const bf16x8*>(wlane[i] + woff);
    if (c < p.C0) {
#pragma unroll
      for (int j = 0; j < PF; ++j) xr[u][j] = *reinterpret_cast<const bf16x8*>(s0p + xoff0[j] + c);
    } else {
#pragma unroll
      for (int j = 0; j < PF; ++j) xr[u][j] = *reinterpret_cast<const bf16x8*>(s1p + xoff1[j] + c);
    }
  };
  f32x16 acc[NF][PF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int u = 0; u < D; ++u) issue(u, u);
  for (int t0 = 0; t0 < a.nsteps; t0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (t0 + u < a.nsteps) {   // (wave-uniform; the last turn of the ring may be partial)
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int j = 0; j < PF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[u][i], xr[u][j], acc[i][j], 0, 0, 0);
      }
      issue(u, t0 + u + D);
    }
  }
  sm_epilogue<PF, NF>(p, a, acc, smem, q, 0, n0, [&](int ml) { const int m = m0 + ml; return m < a.M ? m : -1; }, false);
}

// ---------------------------------------------------------------------------------------------------------------- 3x3
template <int PF, int NF, int D, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_sm3_kernel(const ConvParams p, const SmArgs a) {
  constexpr int BN = NF * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  int wt, px;
  if (!sm_decode(a, &wt, &px)) return;
  const int g = wt / a.ntiles, n0 = (wt - g * a.ntiles) * BN;
  const int tx = px % a.tiles_w, ty = (px / a.tiles_w) % a.tiles_h, b = px / (a.tiles_w * a.tiles_h);
  const int h0 = ty * a.TH, w0 = tx * a.TW;
  const int TW = a.TW, TWP = a.TWP, MT = a.TH * TW;
  const int ks16 = p.Cg >> 4, SC = 9 * ks16;

  // ---- weight ring first: step s = tap * ks16 + c16 (s = 4 t + q for this wave)
  const bf16* wp = reinterpret_cast<const bf16*>(p.wp);
  const bf16* wlane[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) wlane[i] = wp + ((size_t)g * ks16 * 9 * p.NgP + min(n0 + i * 32 + l31, p.NgP - 1)) * 16 + khalf * 8;
  bf16x8 wr[D][NF];
  auto issue_w = [&](int u, int t) {
    const int s = min(4 * t + q, SC - 1);
    const int tap = fdiv(s, a.inv_ks16), c16 = s - tap * ks16;
    const size_t off = ((size_t)c16 * 9 + tap) * p.NgP * 16;
#pragma unroll
    for (int i = 0; i < NF; ++i) wr[u][i] = *reinterpret_cast<const bf16x8*>(wlane[i] + off);
  };
#pragma unroll
  for (int u = 0; u < D; ++u) issue_w(u, u);

  // ---- activation slab by LDS-DMA: slot sidx = r * (V + 1) + v (v == V: pad slot), 64 consecutive slots per wave instruction
  {
    const rsrc_t rs0 = make_rsrc(p.src0, (size_t)p.B * p.sH * p.sW * p.C0 * 2);
    const rsrc_t rs1 = p.src1 ? make_rsrc(p.src1, (size_t)p.B * p.sH * p.sW * p.C1 * 2) : rs0;
    const int npieces = (a.slots + 63) >> 6;
    const int cg0 = g * p.Cg;
    const bool one_src = p.src1 == nullptr || cg0 + p.Cg <= p.C0 || cg0 >= p.C0;   // (workgroup-uniform)
    const bool only1 = p.src1 != nullptr && cg0 >= p.C0;
    for (int piece = q; piece < npieces; piece += 4) {
      const int sidx = piece * 64 + lane;
      const int r = fdiv(sidx, a.inv_V1), v = sidx - r * (a.V + 1);
      const int hh = fdiv(r, a.inv_TWP), ww = r - hh * TWP;
      const int ih = h0 - 1 + hh, iw = w0 - 1 + ww;
      const bool ok = r < a.R && v < a.V && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int pix = (p.resample == DDX_RESAMPLE_UP) ? (b * p.sH + (ih >> 1)) * p.sW + (iw >> 1) : (b * p.sH + ih) * p.sW + iw;
      const int cabs = cg0 + v * 8;
      char* dst = smem + piece * 1024;
      if (one_src) {
        const int voff = ok ? (only1 ? pix * p.C1 + (cabs - p.C0) : pix * p.C0 + cabs) * 2 : kOob;
        dma16(only1 ? rs1 : rs0, voff, dst);
      } else {  // the group straddles the two sources of an mp_cat: two passes under complementary lane masks
        const bool first = cabs < p.C0;
        if (first || !ok) dma16(rs0, ok ? (pix * p.C0 + cabs) * 2 : kOob, dst);
        if (!first && ok) dma16(rs1, (pix * p.C1 + (cabs - p.C0)) * 2, dst);
      }
    }
  }

  int arow[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int ml = j * 32 + l31;
    const int th = fdiv(ml, a.inv_TW);
    arow[j] = (ml < MT) ? th * TWP + (ml - th * TW) : 0;
  }
  f32x16 acc[NF][PF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces (and its first weight fragments) have landed
  __syncthreads();

  const bf16* sA = reinterpret_cast<const bf16*>(smem);
  const int rstride = (a.V + 1) * 8;  // elements
  for (int t0 = 0; t0 < a.nsteps; t0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int s = 4 * (t0 + u) + q;
      const bool valid = s < SC;
      const int sc = valid ? s : 0;
      const int tap = fdiv(sc, a.inv_ks16), c16 = sc - tap * ks16;
      const int t3 = tap / 3;
      const int toff = t3 * TWP + (tap - 3 * t3);
      bf16x8 xf[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        xf[j] = *reinterpret_cast<const bf16x8*>(sA + (size_t)(arow[j] + toff) * rstride + c16 * 16 + khalf * 8);
        if (!valid) {   // steps past the end (at most D - 1 + 3 per wave): zero operand against the (finite) weights they re-read
#pragma unroll
          for (int e = 0; e < 8; ++e) xf[j][e] = (bf16)0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[u][i], xf[j], acc[i][j], 0, 0, 0);
      issue_w(u, t0 + u + D);
    }
  }
  sm_epilogue<PF, NF>(p, a, acc, smem, q, g, n0, [&](int ml) {
    const int th = fdiv(ml, a.inv_TW), tw = ml - th * TW;
    const int h = h0 + th, w = w0 + tw;
    return (ml < MT && h < p.H && w < p.W) ? (b * p.H + h) * p.W + w : -1;
  }, true);
}

// ------------------------------------------------------------------------------------------- host side

struct SmPlan { SmArgs a; int PF, NF, OCC, D; size_t smem; long wgs; };
constexpr size_t kLdsTwo = 78 * 1024, kLdsOne = 156 * 1024;

// pixel tile TH x TW <= BM of an H x W image: fewest tiles, then fewest staged rows
void sm_tile(int H, int W, int BM, int* TH, int* TW) {
  long best_tiles = -1; int best_rows = 0;
  for (int tw = 1; tw <= W && tw <= BM; ++tw)
    for (int th = 1; th <= H && th * tw <= BM; ++th) {
      const long tiles = (long)ceil_div(H, th) * ceil_div(W, tw);
      const int rows = (th + 2) * (tw + 2);
      if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && rows < best_rows)) { best_tiles = tiles; best_rows = rows; *TH = th; *TW = tw; }
    }
}

int env_int(const char* name) { const char* e = std::getenv(name); return e ? atoi(e) : 0; }

bool sm_plan(const ConvParams& p, int ks, SmPlan* out) {
  static const int force_pf = env_int("DDX_SM_PF"), force_nf = env_int("DDX_SM_NF");
  static const int env_d1 = env_int("DDX_SM_D1"), env_d3 = env_int("DDX_SM_D3");
  const int kD1 = env_d1 ? env_d1 : 4, kD3 = env_d3 ? env_d3 : 4;
  SmPlan best{}; double best_cost = 1e30; bool found = false;
  for (int PF = 2; PF <= 4; ++PF) {
    if (force_pf && PF != force_pf) continue;
    for (int NF = 1; NF <= 2; ++NF) {
      if (force_nf && NF != force_nf) continue;
      if (PF == 4 && NF == 1) continue;    // not built
      SmArgs a{};
      const int BM = PF * 32, BN = NF * 32;
      const size_t red = (size_t)2 * BM * (BN + 4) * sizeof(float);
      size_t smem = red;
      double util;
      a.HW = p.H * p.W; a.M = p.B * a.HW;
      a.inv_HW = 1.0f / (float)a.HW; a.inv_W = 1.0f / (float)p.W;
      a.ntiles = ceil_div(p.Ng, BN);
      a.WT = p.G * a.ntiles;
      if (ks == 1) {
        if (p.G != 1 || p.Cin % 64) continue;
        a.nsteps = p.Cin / 64;
        if (a.nsteps % 4) continue;   // a wave owns whole 64-channel lines
        a.PT = ceil_div(a.M, BM);
        util = (double)a.M / ((double)a.PT * BM);
      } else {
        sm_tile(p.H, p.W, BM, &a.TH, &a.TW);
        a.TWP = a.TW + 2;
        a.tiles_h = ceil_div(p.H, a.TH); a.tiles_w = ceil_div(p.W, a.TW);
        a.R = (a.TH + 2) * a.TWP;
        a.V = p.Cg / 8;
        a.slots = a.R * (a.V + 1);
        a.nsteps = round_up(ceil_div(9 * (p.Cg / 16), 4), kD3);
        a.PT = p.B * a.tiles_h * a.tiles_w;
        a.inv_TW = 1.0f / (float)a.TW; a.inv_TWP = 1.0f / (float)a.TWP; a.inv_V1 = 1.0f / (float)(a.V + 1); a.inv_ks16 = 1.0f / (float)(p.Cg / 16);
        smem = std::max(red, (size_t)round_up(a.slots, 64) * 16);
        if (smem > kLdsOne) continue;
        util = (double)p.H * p.W / ((double)a.tiles_h * a.tiles_w * BM);
      }
      const int D = ks == 1 ? kD1 : kD3;
      // registers: accumulators + D ring steps of (NF weight + PF activation | NF weight) fragments
      const int regs = PF * NF * 16 + D * 4 * (ks == 1 ? NF + PF : NF) + 40;
      if (regs > 480) continue;
      const int occ = (smem <= kLdsTwo && regs <= 250) ? 2 : 1;
      const long wgs = (long)a.PT * a.WT;
      // relative cost: rounds of resident workgroups x (matrix time of one workgroup + fixed latency), padding waste included
      const double rounds = std::ceil((double)wgs / (256.0 * occ));
      const double mfma_us = (double)a.nsteps * PF * NF * 32.0 / 2100.0 * (NF == 1 ? 1.4 : 1.0) * (occ == 2 ? 1.5 : 1.0);
      const double cost = rounds * (mfma_us + 3.0) / std::max(util, 0.1) * (util < 0.7 ? 1.3 : 1.0);
      if (cost < best_cost) { best_cost = cost; best = SmPlan{a, PF, NF, occ, D, smem, wgs}; found = true; }
    }
  }
  if (found) *out = best;
  return found;
}

template <typename K>
int launch_kernel(K kern, const ConvParams& p, const SmPlan& pl, hipStream_t s, bool* attr_done) {
  if (!*attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_sm)");
    *attr_done = true;
  }
  const int grid = pl.a.PT * round_up(pl.a.WT, 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), pl.smem, s, p, pl.a);
  return check_launch("conv_sm");
}

template <int KS, int PF, int NF, int OCC, int D>
int launch_sm_d(const ConvParams& p, const SmPlan& pl, hipStream_t s) {
  static bool attr_done = false;
  if constexpr (KS == 1) return launch_kernel(conv_sm1_kernel<PF, NF, D, OCC>, p, pl, s, &attr_done);
  else return launch_kernel(conv_sm3_kernel<PF, NF, D, OCC>, p, pl, s, &attr_done);
}
template <int KS, int PF, int NF, int OCC>
int launch_sm(const ConvParams& p, const SmPlan& pl, hipStream_t s) {
  if (pl.D == 4) return launch_sm_d<KS, PF, NF, OCC, 4>(p, pl, s);
  if (pl.D == 8) return launch_sm_d<KS, PF, NF, OCC, 8>(p, pl, s);
  if constexpr (KS == 3) { if (pl.D == 12) return launch_sm_d<KS, PF, NF, OCC, 12>(p, pl, s); }
  return set_error(DDX_ERR_UNSUPPORTED, "conv_sm: ring depth not built");
}

}  // namespace

bool conv_sm_supported(const ConvParams& p, int ksize, int dtype) {
  if (dtype != DDX_BF16 || (ksize != 1 && ksize != 3) || p.CK != 16) return false;
  if (p.resample == DDX_RESAMPLE_DOWN || p.reflect_w || p.swap1 || p.paired) return false;
  if (p.prologue != DDX_PRO_NONE || p.scale0 != 1.0f || (p.src1 && p.scale1 != 1.0f)) return false;   // raw operands only
  if (p.Cg % 16 || p.C0 % 16 || (p.src1 && p.C1 % 16)) return false;
  if (p.Ng % 4 || p.Cout % 4) return false;
  if (p.epilogue != DDX_EPI_STORE && p.epilogue != DDX_EPI_MPSUM) return false;
  if (p.src0_alt && (ksize != 1 || p.pro_rows <= 0 || p.pro_rows % 64)) return false;
  if ((size_t)p.B * p.sH * p.sW * std::max(p.C0, p.C1) * 2 >= (size_t)0x7fffff00) return false;
  SmPlan pl;
  return sm_plan(p, ksize, &pl);
}

int launch_conv_sm(const ConvParams& p, int ksize, hipStream_t s) {
  SmPlan pl;
  if (!sm_plan(p, ksize, &pl)) return set_error(DDX_ERR_UNSUPPORTED, "conv_sm: no tile fits LDS");
#define DDX_SM(KS_, PF_, NF_)                                                          \
  if (ksize == KS_ && pl.PF == PF_ && pl.NF == NF_)                                    \
    return pl.OCC == 2 ? launch_sm<KS_, PF_, NF_, 2>(p, pl, s) : launch_sm<KS_, PF_, NF_, 1>(p, pl, s)
  DDX_SM(3, 2, 1); DDX_SM(3, 2, 2); DDX_SM(3, 3, 1); DDX_SM(3, 3, 2); DDX_SM(3, 4, 2);
  DDX_SM(1, 2, 1); DDX_SM(1, 2, 2); DDX_SM(1, 3, 1); DDX_SM(1, 3, 2); DDX_SM(1, 4, 2);
#undef DDX_SM
  return set_error(DDX_ERR_UNSUPPORTED, "conv_sm: configuration not built");
}

}  // namespace ddx
