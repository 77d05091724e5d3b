// Prologue-free bf16 MPConv forward with LDS-DMA staging (global_load_lds_dwordx4) on gfx950.
//
// Same math and the same fused epilogue as conv_mfma.hip (reference src/modules/mp_tools.py:366-373 plus the
// element-wise tail of Block.forward, src/modules/unets/unet_edm2_b4.py:110-158), for the layers whose input is
// consumed untouched (the producer already applied mp_silu / the per-channel factors; mp_cat scales live in the
// prepared weights).  Those are all the large convs of the UNet, so their operand path is pure data movement:
//   * every stage (SK input channels of the (TH+2)x(TW+2) halo + the [tap][BN][SK] weight slice) goes HBM/L2 -> LDS
//     by DMA, no VGPR staging, no ds_write, no convert; NST stages in flight, ONE barrier per stage;
//   * LDS rows are 2*SK bytes with no padding (the DMA destination is lane-linear); bank conflicts of the
//     ds_read_b128 fragment reads are removed by XOR-swizzling the 16-byte slot of a row with bits of the row index,
//     applied on the per-lane SOURCE address of the DMA and on the read address;
//   * out-of-image halo rows read a zero page, so padding costs no branch;
//   * the epilogue transposes each wave's 32x32 accumulator fragments through a wave-private LDS patch (no
//     workgroup barrier) and stores 16 bytes per lane on NHWC rows.
// One workgroup = 4 waves = 256 output pixels x 64 output channels; two workgroups per CU overlap each other.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_params.hpp"

namespace ddx {

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOobOffset = 0x7fffff00;  // voffset of a lane that must read zeros (beyond num_records of any tensor here)

// 16 bytes per lane, global -> LDS (lane-linear destination at l).  Raw buffer addressing: the address is
// base + voff + soff and lanes with voff + soff >= num_records write zeros -- that is the conv zero padding.
// Ablation builds for timing experiments (tools/dma_ablate.sh; the results are WRONG): -DDDX_ABL_NODMA issues no DMA,
// -DDDX_ABL_NOMATRIX skips the fragment reads and MFMAs, -DDDX_ABL_NOSTORE skips the epilogue's global stores.
__device__ __forceinline__ void dma16(rsrc_t rs, int voff, int soff, void* l) {
#ifdef DDX_ABL_NODMA
  (void)rs; (void)voff; (void)soff; (void)l;
  return;
#endif
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// floor(x / d) for 0 <= x < 2^22 with inv = 1/d (uniform operands stay off the integer-division sequence)
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kBM = 256;

// Phase timing (build with -DDDX_DMA_TRACE, tools/dma_trace.sh): every wave accumulates shader-clock cycles per pipeline phase
// (DMA wait, barrier, DMA issue incl. next-unit setup, matrix phase, epilogue, per-unit setup) and adds them to g_trace at exit;
// DDX_DMA_TRACE=1 in the environment prints the per-wave means after each launch.  The s_memtime round trips inflate the
// kernel by ~20 %; the split between the phases is what the numbers are for.
#ifdef DDX_DMA_TRACE
__device__ unsigned long long g_trace[12];
#define DDX_TR_INIT long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64()
#define DDX_TR(k) do { const long long now_ = clock64(); tr[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define DDX_TR_INIT do {} while (0)
#define DDX_TR(k) do {} while (0)
#endif

// WM waves along the pixels (MF fragments of 32 pixels each) x WN waves along the output channels (NF fragments of 32 channels)
template <int KS, int SK, int NF, int WN, int WM = 4, int MF = 2> struct DmaGeom {
  static constexpr int TAPS = KS * KS, PAD = KS / 2;
  static constexpr int NW = WM * WN;
  static constexpr int BM = WM * MF * 32;    // output pixels of a unit
  static constexpr int BN = 32 * NF * WN;
  static constexpr int RB = SK * 2;          // bytes per LDS row
  static constexpr int LPR = RB / 16;        // lanes (16-byte slots) per row
  static constexpr int RPW = 1024 / RB;      // rows per DMA wave-instruction
  // halo rows a tile may stage (256 pixels: 8x32 -> 340; 512: 8x64 -> 660; 1024: 16x64 -> 1188)
  static constexpr int AROWS = KS == 3 ? (BM == 256 ? 352 : BM == 512 ? 672 : 1216) : BM;
  static constexpr int APIECES = (AROWS + RPW - 1) / RPW;
  static constexpr int BPIECES = (TAPS * BN + RPW - 1) / RPW;
  static constexpr int A_BYTES = APIECES * 1024, B_BYTES = BPIECES * 1024;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int NST = 2;  // (a 3-stage ring with counted vmcnt was measured on the 1x1 variant: no gain)
  static constexpr int AI = (APIECES + NW - 1) / NW, BI = (BPIECES + NW - 1) / NW;
  static constexpr int EPI_WAVE = 32 * 36 * 4;  // one 32 pixel x 32 channel fp32 patch, rows padded to 36 floats
  // 8-fragment waves in 4-wave workgroups: two workgroups per CU only fit when the epilogue patches reuse stage 1 (idle between
  // the last matrix phase of a unit and the second stage of the next one) -- at the price of one barrier before the epilogue
  static constexpr bool EPI_OVERLAY = NF * MF > 4 && NW == 4;
  // output channel scales of the unit's BN channels, staged by DMA with the unit's first stage (two 1-KiB DMA targets, units
  // alternate): a global load inside the epilogue would wait behind the next unit's first stage, which is in flight there
  static constexpr int CS_OFF = NST * STAGE + (EPI_OVERLAY ? 0 : NW * EPI_WAVE);
  static constexpr int SMEM = CS_OFF + 2048;
  // 16-byte slot swizzle of LDS row r (conflict-free ds_read_b128 over 32 consecutive rows)
  static __device__ __forceinline__ int swz(int r) { return LPR == 2 ? ((r >> 3) & 1) : ((r >> 2) & 3); }
};

// Persistent workgroups: gridDim.x workgroups walk the unit list (unit = pixel tile x channel tile x group) with a
// stride of gridDim.x.  The stage pipeline runs across unit boundaries (the first stage of the next unit is in flight
// while the last stage of the current one is multiplied and its epilogue runs), and the second workgroup of every CU
// starts half a unit late so that one workgroup's memory phases (epilogue stores, first-stage latency) fall into the
// other's matrix phase instead of both doing the same thing at the same time.
// EB = 1: the epilogue is the backward of a = mp_silu(y * s) instead of mp_sum / activation (DDX_EPI_SILU_BWD, see ddx_hip.h).
// DEEP = 1: activation ring of three stages and weight ring of two (see the DEEP main loop)
// WS = 1: weights stationary -- a workgroup keeps ONE (group, channel tile), its nk weight stages stay in LDS for the whole launch
// and only activations stream (layers with few input channels per group, where the weight slices are most of the staged bytes)
template <int KS, int SK, int NF, int WN, int PD, int EB = 0, int WM = 4, int MF = 2, int DEEP = 0, int WS = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? 2 : 1)) void conv_dma_kernel(const ConvParams p, const int total_units, const int ntile_n, const int per_xcd, const int tile_order) {
  using GEO = DmaGeom<KS, SK, NF, WN, WM, MF>;
  constexpr int NW = GEO::NW;
  constexpr int TAPS = GEO::TAPS, PAD = GEO::PAD, RB = GEO::RB, LPR = GEO::LPR, RPW = GEO::RPW, BN = GEO::BN;
  constexpr int NST = GEO::NST, AI = GEO::AI, BI = GEO::BI;
  constexpr int KSTEPS = SK / 16;
  constexpr bool LATE_RES = NF * MF > 4;  // too many fragments to hold every residual row during the last matrix phase

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;  // this wave's 64-pixel slab and (32*NF)-channel slab of the unit
  const int khalf = lane >> 5, l31 = lane & 31;
  const int TW = p.TW, TWP = TW + 2 * PAD;
  const int R = (p.TH + 2 * PAD) * TWP;
  const float inv_TW = 1.0f / (float)TW;
  const int ntile_px = p.B * p.tiles_h * p.tiles_w;
  const int nk = p.Cg / SK;

  const int ck_shift = __builtin_ctz(p.CK);
  struct Unit { int b, h0, w0, n0, g; };
  const float inv_px = 1.0f / (float)ntile_px, inv_nn = 1.0f / (float)ntile_n;
  const float inv_tw = 1.0f / (float)p.tiles_w, inv_th = 1.0f / (float)p.tiles_h;
  // XCD-aware order (per_xcd > 0; chosen per layer by the launcher): workgroups are dealt round-robin to the 8 XCDs, so slot
  // u belongs to XCD u % 8, which walks its own contiguous eighth of a list ordered pixel tile -> channel tile -> group.
  // All channel slices of a pixel tile (64-byte runs of the same 128-byte lines when Cg = 32) and its halo neighbours then
  // meet in ONE L2 at about the same time instead of being fetched from HBM once per XCD.
  const int gn = p.G * ntile_n;
  const float inv_gn = 1.0f / (float)gn, inv_G = 1.0f / (float)p.G;
  auto order = [&](int u) { return per_xcd ? (u & 7) * per_xcd + (u >> 3) : u; };
  // WS order: workgroup w owns combo (w >> 3) % gn = (channel tile, group) and the tiles j, j + stride, ... with
  // j = (w & 7) + 8 * (w / (8 * gn)), stride = gridDim.x / gn: every XCD (w & 7) sees all groups of its own tiles
  const int ws_c = WS ? (int)((blockIdx.x >> 3) % gn) : 0;
  const int ws_j = WS ? (int)((blockIdx.x & 7) + 8 * (blockIdx.x / (8 * gn))) : 0;
  const int ws_stride = WS ? (int)(gridDim.x / gn) : 1;
  const float inv_grid = 1.0f / (float)gridDim.x;
  auto ws_tile = [&](int u) { return ws_j + fdiv(u, inv_grid) * ws_stride; };
  auto live = [&](int u) {
    if constexpr (WS) return ws_tile(u) < ntile_px;
    else return per_xcd ? ((u >> 3) < per_xcd && order(u) < total_units) : u < total_units;
  };
  auto decode = [&](int u) {
    Unit t;
    int tile;
    if constexpr (WS) {
      tile = ws_tile(u);
      const int nt = fdiv(ws_c, inv_G);
      t.g = ws_c - nt * p.G;
      t.n0 = nt * BN;
    } else if (per_xcd) {
      const int o = order(u);
      tile = fdiv(o, inv_gn);
      const int rem = o - tile * gn;
      const int nt = fdiv(rem, inv_G);
      t.g = rem - nt * p.G;
      t.n0 = nt * BN;
    } else {
      const int r = fdiv(u, inv_px);
      tile = u - r * ntile_px;
      t.g = fdiv(r, inv_nn);
      t.n0 = (r - t.g * ntile_n) * BN;
    }
    // tile -> (image, tile row, tile column).  tile_order 0: row-major.  1: column-major inside an image (the XCD-contiguous
    // order walks consecutive tiles, so vertical neighbours -- which share 2 of the 10 halo rows -- run back to back in one L2).
    // 2: plain order, where tile T runs on XCD T % 8 at time T / 8: columns of tiles_h tiles are dealt to the XCDs as runs
    // (run r = image * tiles_w + column -> XCD r % 8), so a column's tiles are resident together in one L2.
    if (tile_order == 0) {
      const int row = fdiv(tile, inv_tw);
      t.w0 = (tile - row * p.tiles_w) * p.TW;
      t.b = fdiv(row, inv_th);
      t.h0 = (row - t.b * p.tiles_h) * p.TH;
    } else {
      int run, th;
      if (tile_order == 1) {
        run = fdiv(tile, inv_th);
        th = tile - run * p.tiles_h;
      } else {
        const int k = tile >> 3;
        const int kr = fdiv(k, inv_th);
        th = k - kr * p.tiles_h;
        run = kr * 8 + (tile & 7);
      }
      t.b = fdiv(run, inv_tw);
      t.w0 = (run - t.b * p.tiles_w) * p.TW;
      t.h0 = th * p.TH;
    }
    return t;
  };

  // ---- DMA source bookkeeping (per lane): this wave moves pieces wave, wave+4, ...
  const int lrow = lane / LPR, lslot = lane % LPR;
  // tile-independent halo coordinates of this lane's rows: a table for the 4-fragment variants; recomputed per unit (from a
  // laundered lane row, so that the compiler does not turn it back into a table) where the registers go to accumulators
  constexpr bool DMA_TABLE = MF <= 2;
  auto a_row = [&](int i, int lr, int& hh_out, int& ww_out, int& slot_out) {
    const int r = (wave + NW * i) * RPW + lr;
    const int hh = (int)(((float)r + 0.5f) * p.inv_TWP);
    hh_out = r < R ? hh - PAD : -(1 << 20);
    ww_out = r - hh * TWP - PAD;
    slot_out = (lslot ^ GEO::swz(r)) * 8;
  };
  auto b_row = [&](int i, int lr, int& tap_out, int& n_out, int& slot_out) {
    const int r = min((wave + NW * i) * RPW + lr, TAPS * BN - 1);
    tap_out = r / BN;
    n_out = r - tap_out * BN;
    slot_out = (lslot ^ GEO::swz(r)) * 8;
  };
  int ahh[DMA_TABLE ? AI : 1], aww[DMA_TABLE ? AI : 1], aslot[DMA_TABLE ? AI : 1];
  int btap[DMA_TABLE ? BI : 1], bn[DMA_TABLE ? BI : 1], bslot[DMA_TABLE ? BI : 1];
  if constexpr (DMA_TABLE) {
#pragma unroll
    for (int i = 0; i < AI; ++i) a_row(i, lrow, ahh[i], aww[i], aslot[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i) b_row(i, lrow, btap[i], bn[i], bslot[i]);
  }
  const rsrc_t rs0 = make_rsrc(p.src0, (size_t)p.B * p.sH * p.sW * p.C0 * 2);
  const rsrc_t rs1 = p.src1 ? make_rsrc(p.src1, (size_t)p.B * p.sH * p.sW * p.C1 * 2) : rs0;
  const rsrc_t rsw = make_rsrc(p.wp, (size_t)p.G * p.nchunk * TAPS * p.NgP * p.CK * 2);
  const rsrc_t rscs = make_rsrc(p.out_cs ? (const void*)p.out_cs : p.wp, p.out_cs ? (size_t)p.B * p.Cout * 4 : 0);
  constexpr bool CS_LDS = !EB && NF <= 2 && MF <= 2;   // (the 8-fragment variants have no registers to spare for it)
  const bool cs_lds = CS_LDS && p.out_cs != nullptr;
  // LDS map of the WS variant: [A stage 0 | A stage 1 | nk weight stages | epilogue patches | channel scales]
  const int ws_boff = 2 * GEO::A_BYTES, ws_eoff = ws_boff + (p.Cg / SK) * GEO::B_BYTES;
  const int cs_base = WS ? ws_eoff + NW * GEO::EPI_WAVE : (DEEP ? 3 * GEO::A_BYTES + 2 * GEO::B_BYTES + (NW * GEO::EPI_WAVE <= GEO::B_BYTES ? 0 : NW * GEO::EPI_WAVE) : GEO::CS_OFF);

  // issue cursor: (unit, stage) of the next DMA batch.  Per lane only byte offsets inside the tensors are kept; the
  // stage (input-channel) advance is a scalar offset, so one batch costs one m0 write + one buffer_load per piece.
  int iu = blockIdx.x, iq = 0, isrc = -1, iunit = 0;
  Unit it{};
  int apix[AI], avoff[AI], bvoff[BI];
  [[maybe_unused]] int aslot_u[DMA_TABLE ? 1 : AI];
  auto issue_setup = [&](int u) {
    it = decode(u);
    isrc = -1;
    int lr = lrow;
    if constexpr (!DMA_TABLE) asm volatile("" : "+v"(lr));
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int hh, ww, sl;
      if constexpr (DMA_TABLE) { hh = ahh[i]; ww = aww[i]; sl = aslot[i]; }
      else { a_row(i, lr, hh, ww, sl); aslot_u[i] = sl; }
      const int ih = it.h0 + hh;
      int iw = it.w0 + ww;
      if (p.reflect_w) iw = iw < 0 ? -iw : (iw >= p.W && iw < p.W + PAD ? 2 * (p.W - 1) - iw : iw);  // only the true border mirrors
      const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int pix = p.resample == DDX_RESAMPLE_UP ? (it.b * p.sH + (ih >> 1)) * p.sW + (iw >> 1) : (it.b * p.sH + ih) * p.sW + iw;
      apix[i] = ok ? pix : -1;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int tp, nn, sl;
      if constexpr (DMA_TABLE) { tp = btap[i]; nn = bn[i]; sl = bslot[i]; }
      else b_row(i, lr, tp, nn, sl);
      const int n = min(it.n0 + nn, p.NgP - 1);  // rows past NgP only feed outputs that are never stored
      bvoff[i] = (((tp * p.NgP + n) << ck_shift) + sl) * 2;
    }
  };
  auto issue_next = [&](auto stage) {  // stage: integral_constant (2-stage pipeline) or runtime int
    if (!live(iu)) return;
    char* sbase = smem + (int)stage * (WS ? GEO::A_BYTES : GEO::STAGE);
    int cabs = it.g * p.Cg + iq * SK;  // first channel of this stage in the (virtually concatenated) input
    const int half = p.C0 + p.C1;
    const int swapped = (p.paired && cabs >= half) ? 1 : 0;   // [src0 | src1 | src0' | src1']: second half from image b ^ 1
    if (swapped) cabs -= half;
    const int src_id = cabs >= p.C0 ? 1 : 0;
    const bool c16 = KS == 3 && ((p.layout >> src_id) & 1);   // channel-blocked source [B][C/16][sH][sW][16]: a stage is one plane
    if (src_id + 2 * swapped != isrc) {  // (re)compute the per-lane row offsets for this source's channel stride
      isrc = src_id + 2 * swapped;
      const int cs2 = (src_id ? p.C1 : p.C0) * 2;
      const int dpix = (swapped || (src_id && p.swap1)) ? ((it.b ^ 1) - it.b) * p.sH * p.sW : 0;   // pair-swapped image
      if (c16) {
        const int hw = p.sH * p.sW;
        const int ibase = (it.b * hw + dpix) * cs2 - it.b * hw * 32;   // image base of the (swapped) image minus the image part of apix
#pragma unroll
        for (int i = 0; i < AI; ++i) avoff[i] = apix[i] >= 0 ? apix[i] * 32 + ibase + (DMA_TABLE ? aslot[i] : aslot_u[i]) * 2 : kOobOffset;
      } else {
#pragma unroll
        for (int i = 0; i < AI; ++i) avoff[i] = apix[i] >= 0 ? (apix[i] + dpix) * cs2 + (DMA_TABLE ? aslot[i] : aslot_u[i]) * 2 : kOobOffset;
      }
    }
    const int cin_src = src_id ? cabs - p.C0 : cabs;
    const int soff_a = c16 ? (cin_src >> 4) * p.sH * p.sW * 32 : cin_src * 2;
    const rsrc_t rsa = src_id ? rs1 : rs0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int piece = wave + NW * i;
      if (piece < GEO::APIECES) dma16(rsa, avoff[i], soff_a, sbase + piece * 1024);
    }
    if (cs_lds && iq == 0 && wave == 0)   // (rides with the unit's first stage: landed at that stage's barrier)
      dma16(rscs, lane < BN / 4 ? lane * 16 : kOobOffset, ((it.b * p.Cout + it.g * p.Ng + it.n0) * 4), smem + cs_base + (iunit & 1) * 1024);
    if constexpr (!WS) {
      const int k0 = iq * SK;
      const int soff_b = ((((it.g * p.nchunk + (k0 >> ck_shift)) * TAPS * p.NgP) << ck_shift) + (k0 & (p.CK - 1))) * 2;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int piece = wave + NW * i;
        if (piece < GEO::BPIECES) dma16(rsw, bvoff[i], soff_b, sbase + GEO::A_BYTES + piece * 1024);
      }
    }
    if (++iq == nk) {
      iq = 0;
      ++iunit;
      iu += gridDim.x;
      if (live(iu)) issue_setup(iu);
    }
  };

  // ---- DEEP pipeline: the activation stages (HBM latency) run TWO ahead in a ring of three slots, the weight stages (L2) one
  // ahead in a ring of two.  Timing ablations (tools/dma_ablate.sh) show why: with one stage in flight per workgroup the DMA
  // side alone takes stages x latency (44 us for 64->64 x8) -- as long as the matrix side alone (50 us) -- and the two overlap
  // only to 72 us.  LDS: 3 x A + 2 x B (+ separate epilogue patches for 32-channel tiles; 64-channel tiles overlay the idle
  // weight slot) + channel scales = 71 KB, still two workgroups per CU.  The A cursor is (iu, iq, it, ...) above; the weights
  // have their own cursor.  One barrier per stage as before; the wait leaves exactly the younger activation batch in flight.
  constexpr int D_BOFF = 3 * GEO::A_BYTES;                                             // weight slots
  constexpr int D_EOFF = D_BOFF + 2 * GEO::B_BYTES;                                    // epilogue patches (32-channel tiles)
  constexpr bool D_OVERLAY = NW * GEO::EPI_WAVE <= GEO::B_BYTES;                       // patches fit a weight slot
  constexpr int D_CSOFF = D_EOFF + (D_OVERLAY ? 0 : NW * GEO::EPI_WAVE);
  int iuB = blockIdx.x, iqB = 0, iunitB = 0;
  Unit itB{};
  [[maybe_unused]] int bvoffB[BI];
  auto setup_B = [&](int u) {
    itB = decode(u);
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int tp, nn, sl;
      if constexpr (DMA_TABLE) { tp = btap[i]; nn = bn[i]; sl = bslot[i]; }
      else b_row(i, lrow, tp, nn, sl);
      const int n = min(itB.n0 + nn, p.NgP - 1);
      bvoffB[i] = (((tp * p.NgP + n) << ck_shift) + sl) * 2;
    }
  };
  auto issue_B = [&](int slot) {
    if (!live(iuB)) return;
    char* sb = smem + D_BOFF + slot * GEO::B_BYTES;
    if (cs_lds && iqB == 0 && wave == 0)
      dma16(rscs, lane < BN / 4 ? lane * 16 : kOobOffset, ((itB.b * p.Cout + itB.g * p.Ng + itB.n0) * 4), smem + cs_base + (iunitB & 1) * 1024);
    const int k0 = iqB * SK;
    const int soff_b = ((((itB.g * p.nchunk + (k0 >> ck_shift)) * TAPS * p.NgP) << ck_shift) + (k0 & (p.CK - 1))) * 2;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int piece = wave + NW * i;
      if (piece < GEO::BPIECES) dma16(rsw, bvoffB[i], soff_b, sb + piece * 1024);
    }
    if (++iqB == nk) {
      iqB = 0;
      ++iunitB;
      iuB += gridDim.x;
      if (live(iuB)) setup_B(iuB);
    }
  };
  auto issue_A = [&](int slot) -> bool {
    if (!live(iu)) return false;
    char* sbase = smem + slot * GEO::A_BYTES;
    int cabs = it.g * p.Cg + iq * SK;
    const int half = p.C0 + p.C1;
    const int swapped = (p.paired && cabs >= half) ? 1 : 0;
    if (swapped) cabs -= half;
    const int src_id = cabs >= p.C0 ? 1 : 0;
    if (src_id + 2 * swapped != isrc) {
      isrc = src_id + 2 * swapped;
      const int cs2 = (src_id ? p.C1 : p.C0) * 2;
      const int dpix = (swapped || (src_id && p.swap1)) ? ((it.b ^ 1) - it.b) * p.sH * p.sW : 0;
#pragma unroll
      for (int i = 0; i < AI; ++i) avoff[i] = apix[i] >= 0 ? (apix[i] + dpix) * cs2 + (DMA_TABLE ? aslot[i] : aslot_u[i]) * 2 : kOobOffset;
    }
    const int soff_a = (src_id ? cabs - p.C0 : cabs) * 2;
    const rsrc_t rsa = src_id ? rs1 : rs0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int piece = wave + NW * i;
      if (piece < GEO::APIECES) dma16(rsa, avoff[i], soff_a, sbase + piece * 1024);
    }
    if (++iq == nk) {
      iq = 0;
      iu += gridDim.x;
      if (live(iu)) issue_setup(iu);
    }
    return true;
  };
  // (waves 0 .. APIECES % NW - 1 move one activation piece more than the others: the younger batch this wave may leave in flight)
  auto wait_stage = [&](bool younger_a) {
    if (!younger_a) { wait_vmcnt<0>(); return; }
    constexpr int REM = GEO::APIECES % NW;
    if (REM == 0 || wave < REM) wait_vmcnt<AI>();
    else wait_vmcnt<(AI > 1 ? AI - 1 : 0)>();
  };

  // ---- fragment read addresses (bytes inside a stage); the tile geometry is the same for every unit
  // (MF > 2: only the halo row of tap 0 is kept per fragment and the tap offsets are added at the read -- 5 VALU per read
  // against 4 * 9 address registers the 8-fragment variant does not have)
  constexpr bool AOFF_TABLE = MF <= 2;
  int aoff[AOFF_TABLE ? MF : 1][AOFF_TABLE ? TAPS : 1];
  int abase[MF];
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int ml = (wm * MF + j) * 32 + l31;
    const int th = (int)(((float)ml + 0.5f) * inv_TW);
    const int tw = ml - th * TW;
    abase[j] = th * TWP + tw;
    if constexpr (AOFF_TABLE) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int r = abase[j] + (t / KS) * TWP + (t % KS);
        aoff[j][t] = r * RB + ((khalf ^ GEO::swz(r)) << 4);  // k-step ks adds (2*ks) to the slot: XOR commutes below
      }
    }
  }
  auto a_addr = [&](int j, int tap) {
    if constexpr (AOFF_TABLE) return aoff[j][tap];
    else {
      int b = abase[j];
      asm volatile("" : "+v"(b));  // (loop-invariant otherwise: the compiler would hoist all 36 addresses out of the unit loop and spill them)
      const int r = b + (tap / KS) * TWP + (tap % KS);
      return r * RB + ((khalf ^ GEO::swz(r)) << 4);
    }
  };
  // weight rows tap*BN + i*32 + l31: the swizzle only depends on l31
  const int bsw = GEO::swz(l31);
  const int boff_r = l31 * RB;

  f32x16 acc[NF][MF];
  // matrix phase of one stage: activations at sA, weight slices at sB (compile-time offsets in the two-stage pipeline: every
  // LDS address is then register + immediate)
  auto compute_at = [&](const char* sA, const char* sB) {
#ifdef DDX_ABL_NOMATRIX
    (void)sA; (void)sB;
    return;
#endif
    constexpr int SLOTS = TAPS * KSTEPS;
    // fragments are read PD slots ahead of the MFMAs that consume them (register ring of PD+1 slots), so that one wave
    // alone keeps the matrix pipe fed while the other wave of its SIMD is in a memory phase
    constexpr int FRING = PD + 1;
    bf16x8 wf[FRING][NF], xf[FRING][MF];
    auto load_frags = [&](int slot, int buf) {
      const int tap = slot / KSTEPS, ks = slot % KSTEPS;
#pragma unroll
      for (int i = 0; i < NF; ++i)
        wf[buf][i] = *reinterpret_cast<const bf16x8*>(sB + (tap * BN + (wn * NF + i) * 32) * RB + boff_r + (((2 * ks + khalf) ^ bsw) << 4));
#pragma unroll
      for (int j = 0; j < MF; ++j)
        xf[buf][j] = *reinterpret_cast<const bf16x8*>(sA + (a_addr(j, tap) ^ (ks << 5)));
    };
#pragma unroll
    for (int s0 = 0; s0 < PD && s0 < SLOTS; ++s0) load_frags(s0, s0 % FRING);
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
      const int cur = slot % FRING;
      if (slot + PD < SLOTS) load_frags(slot + PD, (slot + PD) % FRING);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads in front of the MFMAs of this slot
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][i], xf[cur][j], acc[i][j], 0, 0, 0);
    }
  };

  auto compute = [&](auto stage) {
    compute_at(smem + (int)stage * GEO::STAGE, smem + (int)stage * GEO::STAGE + GEO::A_BYTES);
  };

  static_assert(!GEO::EPI_OVERLAY || NW * GEO::EPI_WAVE <= GEO::STAGE, "epilogue patches overlay stage 1");
  float* sE = reinterpret_cast<float*>(smem + (WS ? ws_eoff : (DEEP ? D_EOFF : (GEO::EPI_OVERLAY ? GEO::STAGE : NST * GEO::STAGE))) + wave * GEO::EPI_WAVE);
  bf16* out = reinterpret_cast<bf16*>(p.out);
  const bf16* res = reinterpret_cast<const bf16*>(p.res);

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  DDX_TR_INIT;
  if (live(iu)) issue_setup(iu);
  [[maybe_unused]] bool a_ahead = false;   // DEEP: the activation batch after the next one is in flight
  [[maybe_unused]] int gs = 0, sa = 0;     // DEEP: global stage counter of the compute side (weight slot gs & 1), gs % 3
  if constexpr (DEEP) {
    static_assert(!EB && NF * MF <= 4 && WN == 1, "deep ring: forward epilogues of the 4-fragment variants");
    if (live(iuB)) setup_B(iuB);
    issue_A(0);
    issue_B(0);
    a_ahead = issue_A(1);
  } else {
    if constexpr (WS) {
      static_assert(!EB && !DEEP && NF * MF <= 4 && WN == 1, "stationary weights: forward epilogues of the 4-fragment variants");
      if (live(iu)) {   // all nk weight stages of this workgroup's (group, channel tile), once
        for (int q = 0; q < nk; ++q) {
          const int k0 = q * SK;
          const int soff_b = ((((it.g * p.nchunk + (k0 >> ck_shift)) * TAPS * p.NgP) << ck_shift) + (k0 & (p.CK - 1))) * 2;
#pragma unroll
          for (int i = 0; i < BI; ++i) {
            const int piece = wave + NW * i;
            if (piece < GEO::BPIECES) dma16(rsw, bvoff[i], soff_b, smem + ws_boff + q * GEO::B_BYTES + piece * 1024);
          }
        }
      }
    }
    issue_next(S0{});
  }
  DDX_TR(2);
  int cunit = -1;
  for (int u = blockIdx.x; live(u); u += gridDim.x) {
    ++cunit;
    [[maybe_unused]] const float* cs_l = reinterpret_cast<const float*>(smem + cs_base + (cunit & 1) * 1024);
    const Unit t = DEEP ? decode(u) : it;  // (two-stage pipeline: the issue cursor is still on this unit, it moves on during the last stage)
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // (EB) the output channels may belong to two tensors (gradients of the two mp_cat sources): this unit's part
    int ld_u = p.Cout, cofs = 0;
    const bf16* res_u = res;
    bf16* out_u = out;
    float sc_u = 1.0f;
    [[maybe_unused]] int epix[MF][2];
    [[maybe_unused]] float dcacc[NF][8];
    if constexpr (EB) {
      sc_u = p.bwd_s0;
      if (p.bwd_split > 0) {
        const bool second = t.g * p.Ng + t.n0 >= p.bwd_split;
        ld_u = second ? p.Cout - p.bwd_split : p.bwd_split;
        cofs = second ? p.bwd_split : 0;
        res_u = second ? reinterpret_cast<const bf16*>(p.bwd_y1) : res;
        out_u = second ? reinterpret_cast<bf16*>(p.bwd_out1) : out;
        sc_u = second ? p.bwd_s1 : p.bwd_s0;
      }
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dcacc[i][e] = 0.f;
    }
    // epilogue addressing: items of a 32 pixel x 32 channel patch are (pixel, 8-channel run); 128 items, 2 per lane.
    // Item idx = lane + 64 * tt is (pixel idx >> 2, run idx & 3): four lanes cover the 64-byte NHWC row of a pixel.  When the
    // main output is channel-blocked and nothing NHWC is read or written, (pixel (idx >> 1) & 31, run 2 * (idx >> 6) + (idx & 1)):
    // the 64 lanes of a store instruction write the 32-byte pieces of 32 consecutive pixels of ONE 16-channel plane = 1 KiB.
    constexpr bool C16_OUT = !EB && WN == 1 && NF <= 2 && MF <= 2;
    const bool map16 = C16_OUT && (p.layout & 4) && p.epilogue != DDX_EPI_MPSUM && (!p.out2 || (p.layout & 8));
    auto item_px = [&](int tt) { return map16 ? (lane >> 1) : (lane >> 2) + 16 * tt; };
    auto item_c8 = [&](int tt) { return map16 ? (2 * tt + (lane & 1)) * 8 : (lane & 3) * 8; };
    long eoff[MF][2];
    [[maybe_unused]] int epx[C16_OUT ? MF : 1][2];   // pixel index of the item (channel-blocked stores)
    auto epilogue_offsets = [&]() {
#pragma unroll
      for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int ml = (wm * MF + j) * 32 + item_px(tt);
          const int th = (int)(((float)ml + 0.5f) * inv_TW);
          const int tw = ml - th * TW;
          const int h = t.h0 + th, w = t.w0 + tw;
          const bool ok = h < p.H && w < p.W;
          const int pix = (t.b * p.H + h) * p.W + w;
          eoff[j][tt] = ok ? (long)((size_t)pix * ld_u + (size_t)t.g * p.Ng + t.n0 - cofs + wn * (NF * 32) + item_c8(tt)) : -1;
          if constexpr (EB) epix[j][tt] = ok ? pix : -1;
          if constexpr (C16_OUT) epx[j][tt] = pix;
        }
    };
    // element offset of channel c of the item's pixel in a channel-blocked [B][Cout/16][H][W][16] tensor
    [[maybe_unused]] const long c16_img = (long)t.b * (p.Cout / 16 - 1) * p.H * p.W * 16;
    [[maybe_unused]] auto c16_off = [&](int j, int tt, int c) {
      if constexpr (C16_OUT) return (long)epx[j][tt] * 16 + c16_img + (long)(c >> 4) * p.H * p.W * 16 + (c & 15);
      else return 0l;
    };
    if constexpr (!LATE_RES) epilogue_offsets();  // needed by the residual prefetch inside the last matrix phase
    u32x4 rres[LATE_RES ? (NF > 2 ? 2 : 1) : NF][MF][2];
    auto rslot = [](int i) { return LATE_RES ? (NF > 2 ? (i & 1) : 0) : i; };

    if constexpr (DEEP) {
      for (int q = 0; q < nk; ++q) {
        DDX_TR(5);
        wait_stage(a_ahead);             // stage gs landed (its weights were issued before the younger activation batch)
        DDX_TR(0);
        __builtin_amdgcn_s_barrier();    // ... for every wave, and everyone is done reading the slots of stage gs - 1
        DDX_TR(1);
        issue_B((gs + 1) & 1);                                   // weights one stage ahead (slot of stage gs - 1)
        a_ahead = issue_A(sa == 0 ? 2 : sa - 1);                 // activations two ahead: slot (gs + 2) % 3 = (gs - 1) % 3
        DDX_TR(2);
        if (q + 1 == nk && p.epilogue == DDX_EPI_MPSUM) {        // residual rows ride along with the last matrix phase
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                const bool ok = eoff[j][tt] >= 0 && t.n0 + (wn * NF + i) * 32 + item_c8(tt) < p.Ng;
                rres[i][j][tt] = *reinterpret_cast<const u32x4*>(res_u + (ok ? eoff[j][tt] + i * 32 : 0));
              }
        }
        compute_at(smem + sa * GEO::A_BYTES, smem + D_BOFF + (gs & 1) * GEO::B_BYTES);
        DDX_TR(3);
        ++gs;
        sa = sa == 2 ? 0 : sa + 1;
      }
    } else {
      // stages come in pairs (nk is even): even stages live in LDS stage 0, odd ones in stage 1
      for (int q = 0; q < nk; q += 2) {
        DDX_TR(5);
        wait_vmcnt<0>();
        DDX_TR(0);
        __builtin_amdgcn_s_barrier();  // this stage landed for every wave; everyone is done reading the other one
        DDX_TR(1);
        issue_next(S1{});
        DDX_TR(2);
        if constexpr (WS) compute_at(smem, smem + ws_boff + q * GEO::B_BYTES);
        else compute(S0{});
        DDX_TR(3);
        wait_vmcnt<0>();
        DDX_TR(0);
        __builtin_amdgcn_s_barrier();
        DDX_TR(1);
        issue_next(S0{});
        DDX_TR(2);
        if (!LATE_RES && q + 2 == nk && (EB || p.epilogue == DDX_EPI_MPSUM)) {  // residual (EB: y) rows ride along with the last matrix phase
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                const bool ok = eoff[j][tt] >= 0 && t.n0 + (wn * NF + i) * 32 + item_c8(tt) < p.Ng;
                rres[i][j][tt] = *reinterpret_cast<const u32x4*>(res_u + (ok ? eoff[j][tt] + i * 32 : 0));
              }
        }
        if constexpr (WS) compute_at(smem + GEO::A_BYTES, smem + ws_boff + (q + 1) * GEO::B_BYTES);
        else compute(S1{});
        DDX_TR(3);
      }
    }

    // ---------------------------------------------------------------- epilogue (wave-private, no workgroup barrier)
    if constexpr (LATE_RES) epilogue_offsets();  // (kept out of the matrix phase's register budget)
    if constexpr (GEO::EPI_OVERLAY) __builtin_amdgcn_s_barrier();  // every wave is done reading stage 1: the patches live there
    if constexpr (DEEP) {
      // 64-channel tiles: the patches overlay the weight slot of the stage just multiplied (idle until the next stage's barrier)
      if constexpr (D_OVERLAY) {
        __builtin_amdgcn_s_barrier();
        sE = reinterpret_cast<float*>(smem + D_BOFF + ((gs - 1) & 1) * GEO::B_BYTES + wave * GEO::EPI_WAVE);
      }
    }
    if constexpr (!EB && WN == 1 && NF <= 2 && MF <= 2) {
      if (p.epilogue == DDX_EPI_PIXELNORM) {
        // normalize(y, dim = channels) on the accumulators: a lane holds 16 of the 32 channels of pixel (lane & 31) per fragment,
        // lane ^ 32 the other 16; channel columns past Cout carry zero weights
        const float inv_sqrt_c = __builtin_amdgcn_rsqf((float)p.Cout);
#pragma unroll
        for (int j = 0; j < MF; ++j) {
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ss = fmaf(acc[i][j][r], acc[i][j][r], ss);
          ss += __shfl_xor(ss, 32, 64);
          const float inv = 1.0f / (p.norm_eps + sqrtf(ss) * inv_sqrt_c);
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (LATE_RES && p.epilogue == DDX_EPI_MPSUM) {  // wide tiles: no registers to prefetch all residual rows, load per column
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const bool ok = eoff[j][tt] >= 0 && t.n0 + (wn * NF + i) * 32 + item_c8(tt) < p.Ng;
            rres[rslot(i)][j][tt] = *reinterpret_cast<const u32x4*>(res + (ok ? eoff[j][tt] + i * 32 : 0));
          }
      }
#pragma unroll
      for (int j = 0; j < MF; ++j) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 y4;
#pragma unroll
          for (int e = 0; e < 4; ++e) y4[e] = acc[i][j][4 * qd + e];
          *reinterpret_cast<f32x4*>(sE + l31 * 36 + 8 * qd + 4 * khalf) = y4;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS is in-order per wave: only the compiler must not reorder
        DDX_TR(6);   // (epilogue sub-phases: 6 = accumulators -> LDS patch, 7 = patch rows -> registers + math, 8 = stores)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int c8 = item_c8(tt);
          const float* srow = sE + item_px(tt) * 36 + c8;
          const f32x4 ya = *reinterpret_cast<const f32x4*>(srow);
          const f32x4 yb = *reinterpret_cast<const f32x4*>(srow + 4);
          float y[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] = ya[e]; y[4 + e] = yb[e]; }
          if constexpr (EB) {
            // y[] = dL/da of 8 channels of one pixel; a = mp_silu(z), z = yy * s, s = chan_scale[b][c] * scale:
            //   dz = da * mp_silu'(z) (act) | da;   out = dz * s (+ add);   dc partial += dz * yy
            const int nch = t.n0 + (wn * NF + i) * 32 + c8;
            if (eoff[j][tt] < 0 || nch >= p.Ng) continue;
            Vec16<bf16> yv, ov, av;
            yv.v = __builtin_bit_cast(bf16x8, rres[i][j][tt]);
            float s8[8];
            if (p.out_cs) {
              const float* csp = p.out_cs + (size_t)t.b * p.Cout + t.g * p.Ng + nch;
              const f32x4 ca = *reinterpret_cast<const f32x4*>(csp);
              const f32x4 cb = *reinterpret_cast<const f32x4*>(csp + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { s8[e] = ca[e] * sc_u; s8[4 + e] = cb[e] * sc_u; }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) s8[e] = sc_u;
            }
            if (p.bwd_add)
              av.v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.bwd_add) + (size_t)epix[j][tt] * p.Cout + t.g * p.Ng + nch);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float yy = yv.get(e);
              float dz = y[e];
              if (p.bwd_act) {
                const float z = yy * s8[e];
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
                dz *= sg * (1.0f + z * (1.0f - sg)) * kMpSiluInv;
              }
              dcacc[i][e] += dz * yy;
              ov.set(e, dz * s8[e] + (p.bwd_add ? av.get(e) : 0.f));
            }
            *reinterpret_cast<bf16x8*>(out_u + eoff[j][tt] + i * 32) = ov.v;
            continue;
          }
          if (p.epilogue == DDX_EPI_MPSUM) {
            Vec16<bf16> rv;
            rv.v = __builtin_bit_cast(bf16x8, rres[rslot(i)][j][tt]);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = rv.get(e) * p.res_a + y[e] * p.res_b;
          }
          if (p.clip > 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fminf(fmaxf(y[e], -p.clip), p.clip);
          }
          const int nch = t.n0 + (wn * NF + i) * 32 + c8;  // channel inside the group
          if (eoff[j][tt] < 0 || nch >= p.Ng) continue;
          const long off = eoff[j][tt] + i * 32;
          if (p.out2) {
            Vec16<bf16> tv;
            if (p.out_cs && !p.out_act) {  // raw main output: the channel scale belongs to the twin (training forward)
              const float* csp = CS_LDS ? cs_l + (wn * NF + i) * 32 + c8   // (LDS copy of this unit's scales)
                                        : p.out_cs + (size_t)t.b * p.Cout + t.g * p.Ng + nch;
              const f32x4 ca = *reinterpret_cast<const f32x4*>(csp);
              const f32x4 cb = *reinterpret_cast<const f32x4*>(csp + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                tv.set(e, mp_silu_f(y[e] * ca[e] * p.out2_scale));
                tv.set(4 + e, mp_silu_f(y[4 + e] * cb[e] * p.out2_scale));
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) tv.set(e, mp_silu_f(y[e] * p.out2_scale));
            }
#ifndef DDX_ABL_NOSTORE
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out2) + ((C16_OUT && (p.layout & 8)) ? c16_off(j, tt, t.g * p.Ng + nch) : off)) = tv.v;
#else
            if (tv.v[0] == (bf16)12345.f) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out2) + off) = tv.v;   // (keeps the math alive)
#endif
          }
          if (p.out_act) {
            if (p.out_cs) {
              const float* csp = CS_LDS ? cs_l + (wn * NF + i) * 32 + c8   // (LDS copy of this unit's scales)
                                        : p.out_cs + (size_t)t.b * p.Cout + t.g * p.Ng + nch;
              const f32x4 ca = *reinterpret_cast<const f32x4*>(csp);
              const f32x4 cb = *reinterpret_cast<const f32x4*>(csp + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { y[e] *= ca[e]; y[4 + e] *= cb[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = mp_silu_f(y[e]);
          }
          Vec16<bf16> ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov.set(e, y[e]);
          DDX_TR(7);
#ifndef DDX_ABL_NOSTORE
          *reinterpret_cast<bf16x8*>(out + ((C16_OUT && (p.layout & 4)) ? c16_off(j, tt, t.g * p.Ng + nch) : off)) = ov.v;
#else
          if (ov.v[0] == (bf16)12345.f) *reinterpret_cast<bf16x8*>(out + off) = ov.v;   // (keeps the math alive)
#endif
          DDX_TR(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // patch is rewritten by the next fragment
      }
    }
    DDX_TR(4);
    if constexpr (EB) {
      if (p.bwd_ws) {
        // channel sums of this wave's 64 pixels: the 16 lanes that share (lane & 3) hold the same 8*NF channels.  Transposing
        // all-reduce: every step halves the values a lane is responsible for and adds the partner's half, so 8*NF - 1 (+1)
        // shuffles leave ONE finished channel sum per lane instead of 4 * 8 * NF shuffles for a plain butterfly.
        constexpr int NV = NF * 8;
        float v[NV];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i * 8 + e] = dcacc[i][e];
        int vid = 0;
        auto step = [&](auto n_tag, int m) {  // keep one half of the n*2 live values, add the partner's copy of it
          constexpr int n = decltype(n_tag)::value;
          const bool hi = (lane & m) != 0;
#pragma unroll
          for (int k = 0; k < n; ++k) {
            const float send = hi ? v[k] : v[k + n];
            const float keep = hi ? v[k + n] : v[k];
            v[k] = keep + __shfl_xor(send, m, 64);
          }
          vid = vid * 2 + (hi ? 1 : 0);
        };
        step(std::integral_constant<int, NV / 2>{}, 32);
        step(std::integral_constant<int, NV / 4>{}, 16);
        step(std::integral_constant<int, NV / 8>{}, 8);
        if constexpr (NV == 16) step(std::integral_constant<int, 1>{}, 4);
        else v[0] += __shfl_xor(v[0], 4, 64);  // NF = 1: the last step is a plain pair sum (both lanes hold it)
        const bool writer = NV == 16 || (lane & 4) == 0;
        if (writer) p.bwd_ws[((size_t)u * NW + wave) * (NF * 32) + (vid >> 3) * 32 + (lane & 3) * 8 + (vid & 7)] = v[0];
      }
    }
  }
#ifdef DDX_DMA_TRACE
  if (lane == 0) {
    for (int k = 0; k < 9; ++k) atomicAdd(&g_trace[k], (unsigned long long)tr[k]);
    atomicAdd(&g_trace[9], 1ull);
  }
#endif
}

// dc[b][c] += scale * sum over the (pixel tile, wave) partial rows of image b written by the EB epilogue.
// Row index = unit * NW + wave with unit = (g * ntile_n + nt) * ntile_px + b * tiles_per_image + tile.
__global__ __launch_bounds__(256) void conv_dc_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dc, int rows_per_image, int rows_per_gn,
                                                             int BN, int ntile_n, int Ng, int Cout, float scale) {
  __shared__ float red[256];
  const int gn = blockIdx.x, b = blockIdx.y;
  const int g = gn / ntile_n, nt = gn - g * ntile_n;
  const int ch = threadIdx.x % BN, sub = threadIdx.x / BN, nsub = 256 / BN;
  const float* base = ws + ((size_t)gn * rows_per_gn + (size_t)b * rows_per_image) * BN;
  float s = 0.f;
  for (int r = sub; r < rows_per_image; r += nsub) s += base[(size_t)r * BN + ch];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0) {
    for (int k = 1; k < nsub; ++k) s += red[k * BN + ch];
    const int c = nt * BN + ch;
    if (c < Ng) dc[(size_t)b * Cout + g * Ng + c] += s * scale;
  }
}

template <int KS, int SK, int NF, int WN, int EB = 0, int WM = 4, int MF = 2, int DEEP = 0, int WS = 0>
int launch_dma_t(const ConvParams& p, hipStream_t s) {
  using GEO = DmaGeom<KS, SK, NF, WN, WM, MF>;
  constexpr int DEEP_SMEM = 3 * GEO::A_BYTES + 2 * GEO::B_BYTES + (GEO::NW * GEO::EPI_WAVE <= GEO::B_BYTES ? 0 : GEO::NW * GEO::EPI_WAVE) + 2048;
  constexpr int WS_NK_MAX = (80 * 1024 - 2 * GEO::A_BYTES - GEO::NW * GEO::EPI_WAVE - 2048) / GEO::B_BYTES;   // weight stages that fit
  const int SMEM_BYTES = WS ? 2 * GEO::A_BYTES + (p.Cg / SK) * GEO::B_BYTES + GEO::NW * GEO::EPI_WAVE + 2048 : (DEEP ? DEEP_SMEM : GEO::SMEM);
  static_assert((DEEP ? DEEP_SMEM : GEO::SMEM) <= (GEO::NW == 4 ? 80 : 160) * 1024, "LDS budget");
  if (WS && (p.Cg / SK > WS_NK_MAX)) return set_error(DDX_ERR_UNSUPPORTED, "conv_dma: weights do not fit LDS");
  static_assert(!EB || (NF <= 2 && WN == 1 && WM == 4 && MF == 2), "the fused backward epilogue keeps y in the residual registers");
  auto kern = conv_dma_kernel<KS, SK, NF, WN, ((MF > 2 && NF * MF > 4) ? 0 : 1), EB, WM, MF, DEEP, WS>;  // (no fragment prefetch only where 128 accumulators leave no registers)
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WS ? 80 * 1024 : SMEM_BYTES) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_dma)");
    attr_done = true;
  }
  const int ntile_n = ceil_div(p.Ng, GEO::BN);
  const long total = (long)p.B * p.tiles_h * p.tiles_w * ntile_n * p.G;
  int grid = (int)std::min<long>(total, GEO::NW == 4 ? 512 : 256);  // persistent: every CU holds 8 waves
  if (WS) grid = 512;   // (the launcher checked: 512 % (8 * combos) == 0)
  static const int grid_knob = getenv("DDX_DMA_GRID") ? atoi(getenv("DDX_DMA_GRID")) : 0;   // experiment knob: persistent grid size
  if (grid_knob > 0) grid = (int)std::min<long>(total, grid_knob);
  // XCD-aware unit order where it was measured to cut HBM fetches: 3x3 layers whose group slice of a pixel is half a cache
  // line (Cg = 32: -39 % FETCH_SIZE) or whose unit covers a whole group's 32 output channels (-16 %).  Elsewhere the plain
  // order already keeps a pixel tile on one XCD (B * tiles divisible by 8) and the contiguous order fetched 20-100 % more.
  static const int xcd_knob = getenv("DDX_DMA_XCD") ? atoi(getenv("DDX_DMA_XCD")) : 1;
  int per_xcd = 0;
  if (!EB && KS == 3 && WN == 1 && xcd_knob && total >= 64 && (xcd_knob == 2 || p.Cg <= 32 || p.Ng <= 32)) {
    grid &= ~7;
    per_xcd = (int)((total + 7) / 8);
  }
  // vertical halo sharing (see `decode`): column-major tiles for the XCD-contiguous order; run-dealt columns for the plain
  // order when tiles and columns split evenly over the 8 XCDs
  static const int col_knob = getenv("DDX_DMA_COL") ? atoi(getenv("DDX_DMA_COL")) : 0;  // measured neutral: off
  int tile_order = 0;
  if (!EB && KS == 3 && col_knob && p.tiles_h > 1) {  // (EB: the dc reduction relies on image-major unit rows)
    const int ntile_px = p.B * p.tiles_h * p.tiles_w;
    if (per_xcd) tile_order = 1;
    else if (ntile_px % 8 == 0 && (p.B * p.tiles_w) % 8 == 0 && grid % 8 == 0) tile_order = 2;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * GEO::NW), SMEM_BYTES, s, p, (int)total, ntile_n, per_xcd, tile_order);
  if (EB && p.bwd_ws && p.bwd_dc) {
    const int tpi = p.tiles_h * p.tiles_w;
    hipLaunchKernelGGL(conv_dc_reduce_kernel, dim3(p.G * ntile_n, p.B), dim3(256), 0, s, (const float*)p.bwd_ws, p.bwd_dc, tpi * GEO::NW,
                       p.B * tpi * GEO::NW, GEO::BN, ntile_n, p.Ng, p.Cout, p.bwd_s0);
  }
#ifdef DDX_DMA_TRACE
  if (getenv("DDX_DMA_TRACE")) {
    unsigned long long h[12] = {0}, z[12] = {0};
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof(h)) == hipSuccess && h[9]) {
      const double w = (double)h[9];
      fprintf(stderr, "[dma trace] %d units, cycles per wave: dma-wait %.0f barrier %.0f dma-issue %.0f matrix %.0f epilogue %.0f (patch write %.0f, "
              "read + math %.0f, stores %.0f, rest %.0f) unit-setup %.0f\n",
              (int)total, h[0] / w, h[1] / w, h[2] / w, h[3] / w, (h[4] + h[6] + h[7] + h[8]) / w, h[6] / w, h[7] / w, h[8] / w, h[4] / w, h[5] / w);
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), z, sizeof(z));
  }
#endif
  return check_launch("conv_dma");
}

// 1x1 layers with >= 192 output channels per group run as 256 x 256 GEMM tiles (8 waves) when that still leaves
// enough units for the 256 CUs
bool dma_wide_1x1(const ConvParams& p, long pixel_tiles) {
  static const int knob = std::getenv("DDX_DMA_WIDE") ? atoi(std::getenv("DDX_DMA_WIDE")) : -1;   // experiment knob: 0 never, 1 default rule
  if (knob == 0) return false;
  return p.Ng >= 192 && pixel_tiles * ceil_div(p.Ng, 256) >= 128;
}

// TH x TW with TW a multiple of 32 (fragments never wrap tile rows) and TH*TW = 256
bool dma_tile(const ConvParams& p, int ksize, int* TH, int* TW, double* util, int bm = kBM, int max_rows_3x3 = DmaGeom<3, 16, 2, 1>::AROWS) {
  const int pad = ksize / 2;
  const int max_rows = ksize == 3 ? max_rows_3x3 : bm;
  double best = -1;
  const int tws[4] = {32, 64, 128, 256};
  for (int tw : tws) {
    const int th = bm / tw;
    if (th < 1 || (th + 2 * pad) * (tw + 2 * pad) > max_rows) continue;
    const double u = (double)p.H * p.W / ((double)ceil_div(p.H, th) * ceil_div(p.W, tw) * bm);
    if (u > best + 1e-9) { best = u; *TH = th; *TW = tw; }
  }
  *util = best;
  return best > 0;
}

}  // namespace

// channel-tile width of the fused-backward launch: tiles start at g * Ng + k * BN, the part boundary must be one of them
static int dma_bwd_bn(const ConvParams& p) {
  const int in_group = p.bwd_split % p.Ng;
  if (p.Ng > 32 && in_group % 64 == 0) return 64;
  if (in_group % 32 == 0) return 32;
  return 0;
}

bool conv_dma_supported(const ConvParams& p, int ksize, int dtype, bool any_size) {
  if (dtype != DDX_BF16 || (ksize != 1 && ksize != 3)) return false;
  if (p.prologue != DDX_PRO_NONE || p.scale0 != 1.0f || (p.src1 && p.scale1 != 1.0f)) return false;
  if (p.resample == DDX_RESAMPLE_DOWN) return false;
  const int SK = ksize == 3 ? 16 : 32;
  if (p.Cg % (2 * SK) || p.C0 % SK || (p.src1 && p.C1 % 8)) return false;  // stages are processed in pairs
  if (p.paired && p.C1 % SK) return false;                                   // (no stage may straddle two of the four parts)
  if (p.CK % SK) return false;
  if (p.Ng % 8 || p.Cout % 8) return false;
  if (p.epilogue == DDX_EPI_MPSUM && p.out_act && p.out_cs && (p.Cout % 4)) return false;
  if (p.epilogue == DDX_EPI_SILU_BWD && dma_bwd_bn(p) == 0) return false;  // every channel tile must lie in ONE part
  if (p.epilogue == DDX_EPI_PIXELNORM && (p.G != 1 || p.Cout > 64)) return false;  // all output channels of a pixel in one wave's fragments
  if ((size_t)p.B * p.sH * p.sW >= (size_t)1 << 31) return false;
  int TH, TW; double util;
  if (!dma_tile(p, ksize, &TH, &TW, &util)) return false;
  if (any_size) return true;
  // automatic choice: layers with enough units to fill the persistent grid; small-M layers stay on the register-staged
  // kernel (split-K, wider K chunks)
  const long tiles = (long)p.B * ceil_div(p.H, TH) * ceil_div(p.W, TW) * p.G;
  if (util < 0.6) return false;
  if (ksize == 1 && dma_wide_1x1(p, tiles)) return true;
  // (3x3 from 384 units: the L2 layers with 96 channels per group measure 30.0 us here vs 34.7 us register-staged)
  return tiles * ceil_div(p.Ng, p.Ng <= 32 ? 32 : 64) >= (ksize == 3 ? 384 : 512);
}

size_t conv_dma_bwd_ws_bytes(const ConvParams& p, int ksize) {
  int TH = 0, TW = 0; double util;
  if (!dma_tile(p, ksize, &TH, &TW, &util)) return 0;
  const int BN = dma_bwd_bn(p);
  if (BN == 0) return 0;
  const size_t units = (size_t)p.B * ceil_div(p.H, TH) * ceil_div(p.W, TW) * ceil_div(p.Ng, BN) * p.G;
  return units * 4 * BN * sizeof(float);
}

int launch_conv_dma(const ConvParams& p_in, int ksize, hipStream_t s) {
  ConvParams p = p_in;
  int TH = 0, TW = 0; double util;
  if (!dma_tile(p, ksize, &TH, &TW, &util)) return set_error(DDX_ERR_UNSUPPORTED, "conv_dma: no tile");
  const int pad = ksize / 2;
  p.TH = TH; p.TW = TW;
  p.tiles_h = ceil_div(p.H, TH); p.tiles_w = ceil_div(p.W, TW);
  p.arows_alloc = (TH + 2 * pad) * (TW + 2 * pad);
  p.inv_TWP = 1.0f / (float)(TW + 2 * pad);
  // channel tile: 32 (one fragment column) or 64 with 4 waves and 2 workgroups per CU; 256 for wide 1x1 layers (8 waves).
  // (A 128-channel 8-wave 3x3 variant measured within 3% of the 64-channel one and loses on ragged groups: not built.)
  if (p.epilogue == DDX_EPI_SILU_BWD) {
    const bool narrow = dma_bwd_bn(p) == 32;
    if (ksize == 3) return narrow ? launch_dma_t<3, 16, 1, 1, 1>(p, s) : launch_dma_t<3, 16, 2, 1, 1>(p, s);
    return narrow ? launch_dma_t<1, 32, 1, 1, 1>(p, s) : launch_dma_t<1, 32, 2, 1, 1>(p, s);
  }
  if (ksize == 3) {
    // 512-pixel units (4 waves x 4 pixel fragments): twice the matrix work per DMA round trip of a K-stage and the weight
    // slices staged once per 512 pixels.  With 32-channel tiles (4 fragments per wave, no register pressure) this is the
    // candidate for layers with few K-stages per unit -- measured 10-40 % slower than the 256-pixel units for both channel
    // widths, kept behind the knob.
    static const int big_knob = std::getenv("DDX_DMA_BIG") ? atoi(std::getenv("DDX_DMA_BIG")) : 0;  // experiment knob: 2 = wherever it fits
    using BIG = DmaGeom<3, 16, 2, 1, 4, 4>;
    int bth = 0, btw = 0; double butil = 0;
    if (big_knob && !p.layout && dma_tile(p, 3, &bth, &btw, &butil, BIG::BM, BIG::AROWS)) {
      const int bn = p.Ng <= 32 ? 32 : 64;
      const long units = (long)p.B * ceil_div(p.H, bth) * ceil_div(p.W, btw) * p.G * ceil_div(p.Ng, bn);
      if (big_knob == 2 && units > 0) {   // (no automatic rule: both channel widths measured slower than the 256-pixel units)
        p.TH = bth; p.TW = btw;
        p.tiles_h = ceil_div(p.H, bth); p.tiles_w = ceil_div(p.W, btw);
        p.arows_alloc = (bth + 2) * (btw + 2);
        p.inv_TWP = 1.0f / (float)(btw + 2);
        return p.Ng <= 32 ? launch_dma_t<3, 16, 1, 1, 0, 4, 4>(p, s) : launch_dma_t<3, 16, 2, 1, 0, 4, 4>(p, s);
      }
    }
  }
  // stationary weights where all K-stages of a channel tile fit beside two activation stages (Cg <= 32 with 64-channel tiles,
  // Cg <= 64 with 32-channel tiles): -7 ... -15 % on those layers.  DDX_DMA_WS=0 off, 2 = also 32-channel tiles for
  // Ng = 64 layers with Cg = 64 (experiment)
  static const int ws_knob = std::getenv("DDX_DMA_WS") ? atoi(std::getenv("DDX_DMA_WS")) : 1;
  if (ksize == 3 && ws_knob) {
    const int nk = p.Cg / 16;
    const int bn = (p.Ng <= 32 || (ws_knob == 2 && p.Ng == 64 && nk == 4)) ? 32 : 64, combos = p.G * ceil_div(p.Ng, bn);
    const long tiles = (long)p.B * p.tiles_h * p.tiles_w;
    if (combos <= 64 && 512 % (8 * combos) == 0 && tiles * combos >= 1024 && nk <= (bn == 32 ? 4 : 2))
      return bn == 32 ? launch_dma_t<3, 16, 1, 1, 0, 4, 2, 0, 1>(p, s) : launch_dma_t<3, 16, 2, 1, 0, 4, 2, 0, 1>(p, s);
  }
  static const int deep_knob = std::getenv("DDX_DMA_DEEP") ? atoi(std::getenv("DDX_DMA_DEEP")) : 0;
  if (ksize == 3 && deep_knob && !p.layout) return p.Ng <= 32 ? launch_dma_t<3, 16, 1, 1, 0, 4, 2, 1>(p, s) : launch_dma_t<3, 16, 2, 1, 0, 4, 2, 1>(p, s);
  if (ksize == 3) return p.Ng <= 32 ? launch_dma_t<3, 16, 1, 1>(p, s) : launch_dma_t<3, 16, 2, 1>(p, s);
  if (p.Ng <= 32) return launch_dma_t<1, 32, 1, 1>(p, s);
  const bool wide = dma_wide_1x1(p, (long)p.B * p.tiles_h * p.tiles_w * p.G);
  return wide ? launch_dma_t<1, 32, 4, 2>(p, s) : launch_dma_t<1, 32, 2, 1>(p, s);
}

}  // namespace ddx
