// conv_res0 -> mp_silu(y * c) -> conv_res1 -> mp_sum / clip (+ activated twin) of an EDM2 encoder block as ONE kernel, for the
// level-0 blocks of the default UNet (reference src/modules/unets/unet_edm2_b4.py:121-135: y = conv_res0(mp_silu(x)); c = emb_linear(emb) * gain + 1;
// y = mp_silu(y * c); y = conv_res1(y); x = mp_sum(x, y, t)): 32 -> 64 -> 32 channels per group, 88 064 pixels at B = 4.
//
// Why: at this level the two 3x3 convs are bound by the bytes they move, not by their MACs (26 GFLOP each; 45 MB in, 90 MB of hidden
// activations out, the same 90 MB back in, 45 MB of residual, 45 + 45 MB out: 360 MB per block, 106-115 us in two launches).  Both convs are
// grouped the same way, so a group's hidden channels of a pixel tile only need that group's input channels on the tile + 2 halo pixels:
// the hidden tensor can live in LDS and never touch HBM (~200 MB per block).
//
// Structure (one workgroup of eight waves per CU, persistent, LDS 133 KB + 512 B per image):
//   * a workgroup owns ONE group (blockIdx % groups: with the round-robin workgroup -> XCD dispatch every XCD's L2 holds one group's channel
//     slice, and tiles that share halo pixels are in flight on the same XCD at the same time) and walks that group's 8 x 32-pixel tiles;
//   * WEIGHTS LIVE IN REGISTERS for the whole launch: waves 0-3 hold conv_res0's 64 x 288 slice as 36 A fragments (144 VGPRs), waves 4-7
//     conv_res1's 32 x 576 slice (144 VGPRs).  The matrix loops read only activations from LDS -- one ds_read_b128 per TWO MFMAs in conv_res0
//     (the 4-wave LDS-DMA kernels read one fragment per MFMA), 48 reads for 72 MFMAs in conv_res1 (a hidden row serves tap row h of one tile
//     row and tap row h - 1 of the next) -- three steps ahead, every address base register + immediate (sched_group_barrier pins the order);
//   * wave specialisation, software-pipelined over the tiles: in iteration i waves 0-3 turn input tile i (12 x 36 pixels, LDS) into the hidden
//     tile (10 x 34 pixels x 64 channels, bf16, LDS, zero outside the image = conv_res1's zero padding) while waves 4-7 compute output tile
//     i - 1 from the previous hidden tile (two hidden buffers).  Each SIMD hosts one wave of either kind;
//   * all eight waves stage input tile i + 1 through registers (buffer loads with out-of-range offsets for the zero padding, issued at the top
//     of the iteration, written to LDS between the iteration's two barriers, when nobody reads the single input buffer).  LDS-DMA was measured
//     first: an instruction that gathers sixteen 64-byte pixel slices at a 512-byte stride held its wave ~1200 cycles (40 of 130 us);
//   * LDS rows are padded, not swizzled: 80 B (input) and 144 B (hidden) -- 16 consecutive rows start in 16 different 16-byte bank groups;
//   * epilogues in registers: mp_silu on pairs of channels (v_pk_mul_f32 / v_pk_add_f32, the two scaled copies of c from LDS), the output
//     through v_permlane32_swap on packed bf16 pairs (conv_dma.hip's register epilogue), residual / mp_sum / clip / twin in fp32; every
//     residual word is consumed before the first store, and the stores are issued after the second barrier (they retire under the next tile).
// Traps found on the way (all in the comments where they bit): __builtin_amdgcn_permlane32_swap on fp32 values bit-cast to unsigned dropped its
// second result (packed pairs work); a select on a loaded value, or an array merged with an undefined value after a branch, makes the compiler
// wait for the load on the spot; loop-invariant per-lane address terms are hoisted into 20-60 registers and spilled (an opaque copy of the lane
// index per tile stops that); a spill reload inside the tile loop waits on vmcnt behind the staging loads.
//
// Also built and measured (round 4), not kept:
//   * conv_res0 one 32-channel half at a time (18 dependent MFMAs) with the mp_silu epilogue of the half before it issued between the MFMAs of
//     the same wave, all 144 weight registers resident: the allocator spills 22-48 registers, 114-146 us;
//   * the same with every conv_res0 wave owning ONE half (72 weight registers, fragments alternating between two accumulators, the scaled
//     copies of c in 32 registers, all staging on these waves, two taps of conv_res1's weights parked in LDS so that nothing spills): the
//     instruction stream is as intended (MFMA, ~12 VALU, MFMA, ...) and a fragment still takes ~1 500 cycles = its 576 MFMA cycles PLUS its
//     ~800 VALU cycles: 106 us.  Matrix and vector work of a SIMD add up here whether they come from two waves or are interleaved inside
//     one; what shortens a tile is fewer VALU cycles (the packed mp_silu: 130 -> 91 us) or fewer MFMA cycles, not their arrangement.
//   * Two smaller traps of those variants: a value declared outside the role branch but defined inside it is merged with an undefined value at
//     the join and occupies registers on the other role's path (28 staging registers cost the conv_res1 waves 128 spilled weight registers);
//     a register read by a store still in flight cannot be overwritten (the compiler waits for the store), so output stores issued right
//     before the next tile's accumulators are zeroed cost the store latency per tile.
//
// Measured (tools/pair_bench.py, graph replay, MI355X): B = 4: 115 us in two launches -> 88 us; B = 32: 854 -> 637 us (with the activated twin
// 952 -> 698).  Counters (tools/pmc_pair.sh): MFMA busy 30 % of the SIMD cycles, VALU 32 % (2 M of the 12.5 M VALU instructions are the
// quarter-rate v_exp_f32 / v_rcp_f32 of mp_silu), LDS 23 % with 30 % bank-conflict cycles, waves waiting 37 % of their time.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "conv_params.hpp"

namespace ddx {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int CI = 32, CH = 64, CO = 32;          // channels per group: input, hidden, output
constexpr int TH = 8, TW = 32;                    // output tile
constexpr int HR = TH + 2, HC = TW + 2, HPIX = HR * HC;      // hidden region 10 x 34 = 340 pixels
constexpr int HFRAG = (HPIX + 31) / 32;           // 11 pixel fragments
constexpr int IR = TH + 4, IC = TW + 4, IPIX = IR * IC;      // input region 12 x 36 = 432 pixels
constexpr int ISTR = CI * 2 + 16;                 // input row stride: 80 B = 5 x 16 (same idea as HSTR below: padding instead of an XOR swizzle)
constexpr int IN_BYTES = IPIX * ISTR;             // 34 560
constexpr int HSTR = CH * 2 + 16;                 // hidden row stride: 144 B = 9 x 16 -- 16 consecutive rows land on 16 different 16-byte bank groups,
                                                  // no XOR swizzle, so every tap / k-step of conv_res1 is base register + immediate
constexpr int HID_BYTES = HPIX * HSTR;            // 48 960
constexpr int IN_VEC = IPIX * 4;                  // 16-byte vectors of an input tile
constexpr int STG = (IN_VEC + 511) / 512;         // staged vectors per thread (4): all eight waves share the staging

struct PairArgs {
  const bf16* x; const bf16* res; const bf16* w0; const bf16* w1; const float* cs;
  bf16* out; bf16* out2;
  int B, H, W, C, G, tiles_h, tiles_w, units, wgs_per_group;
  float res_a, res_b, clip, out2_scale;
  int dbg;   // DDX_ABLATE timing ablations: 1 no conv_res0 matrix loop, 2 no conv_res1 matrix loop, 4 no staging loads, 8 no output stores, 16 no hidden writes
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOob = 0x7fffff00;     // byte offset beyond num_records: the buffer load returns zeros
__device__ __forceinline__ void dma16(rsrc_t rs, int voff, void* l) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, 0, 0, 0); }
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)std::min<size_t>(bytes, 0x7ffffff0u), 0x00020000);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

struct Unit { int b, h0, w0; };
__device__ __forceinline__ Unit unit_of(const PairArgs& p, int u) {
  Unit t;
  const int per_img = p.tiles_h * p.tiles_w;
  t.b = u / per_img;
  const int r = u - t.b * per_img;
  const int ty = r / p.tiles_w;
  t.h0 = ty * TH; t.w0 = (r - ty * p.tiles_w) * TW;
  return t;
}

template <bool TWIN>
__global__ __launch_bounds__(512, 1) void conv_pair_kernel(const PairArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const s_in = smem;                               // IN_BYTES (one buffer: refilled between the two barriers of a tile)
  char* const s_hid = smem + IN_BYTES;                   // 2 x HID_BYTES
  float* const s_cs = reinterpret_cast<float*>(smem + IN_BYTES + 2 * HID_BYTES);   // [B][2][CH]: c * -log2(e) and c / 0.596
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index in an SGPR)
  const int khalf0 = lane >> 5, l310 = lane & 31;
  const int g = blockIdx.x % p.G, wg = blockIdx.x / p.G;
  // (waves w and w + 4 share a SIMD: measured against odd / even roles, 95 vs 105 us)
  const bool second = wave >= 4;                         // waves 4-7: conv_res1
  const int wq = wave & 3;
  const int nu = (p.units - wg + p.wgs_per_group - 1) / p.wgs_per_group;   // units u = wg + k * wgs_per_group of this workgroup

  // ---- one-time: channel scales of this group's hidden channels, weight fragments into registers
  for (int i = tid; i < p.B * CH; i += 512) {
    const float c = p.cs[(size_t)(i / CH) * (2 * p.C) + g * CH + (i % CH)];
    s_cs[(i / CH) * 2 * CH + (i % CH)] = c * -1.44269504088896341f;
    s_cs[(i / CH) * 2 * CH + CH + (i % CH)] = c * kMpSiluInv;
  }
  bf16x8 wr[36];
  if (!second) {   // conv_res0: wp[g][0][tap][n < 64][c < 32]; fragment (tap, s, i): rows n = 32 i + l31, k = 16 s + 8 khalf ..
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          wr[(tap * 2 + s) * 2 + i] = *reinterpret_cast<const bf16x8*>(p.w0 + ((((size_t)g * 9 + tap) * CH + 32 * i + l310) * CI + 16 * s + 8 * khalf0));
  } else {         // conv_res1: wp[g][chunk < 2][tap][n < 32][c < 32]; fragment (tap, s < 4): rows n = l31, k = 16 s + 8 khalf ..
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        wr[tap * 4 + s] = *reinterpret_cast<const bf16x8*>(p.w1 + (((((size_t)g * 2 + (s >> 1)) * 9 + tap) * CO + l310) * 32 + 16 * (s & 1) + 8 * khalf0));
  }

  // ---- staging of an input tile (all waves) through registers: vector v = (row r = v >> 2, slot v & 3) of the 12 x 36 halo tile, zero outside
  // the image; loads are issued before the tile's matrix loops and written to LDS after them.  (LDS-DMA was measured first: an instruction that
  // gathers sixteen 64-byte pixel slices at a 512-byte stride held its wave ~1200 cycles, 40 us of a 130 us launch.)
  const rsrc_t rs_x = make_rsrc(p.x, (size_t)p.B * p.H * p.W * p.C * 2);
  auto stage_load = [&](const Unit& t, bool live, u32x4 (&sv)[STG]) {
    int tv = tid;
    asm volatile("" : "+v"(tv));     // (opaque: keeps the per-vector row / column terms from being hoisted out of the tile loop into 20+ registers)
#pragma unroll
    for (int k = 0; k < STG; ++k) {
      const int v = tv + 512 * k;
      const int r = v >> 2, ir = r / IC, ic = r - ir * IC;
      const int ih = t.h0 - 2 + ir, iw = t.w0 - 2 + ic;
      const bool ok = live && v < IN_VEC && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      // buffer load: an offset beyond num_records returns zeros, so the padding needs no select on the loaded value (a select would make the
      // compiler wait for the load right here, a full global round trip at the top of every tile)
      const int voff = ok ? (int)(((((size_t)t.b * p.H + ih) * p.W + iw) * p.C + g * CI + (v & 3) * 8) * 2) : kOob;
      sv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, 0, 0);
    }
  };
  auto stage_store = [&](char* buf, const u32x4 (&sv)[STG]) {
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int k = 0; k < STG; ++k) {
      const int v = tv + 512 * k;
      const int r = v >> 2;
      if (v < IN_VEC) *reinterpret_cast<u32x4*>(buf + r * ISTR + ((v & 3) << 4)) = sv[k];
    }
  };

  if (nu > 0) {
    u32x4 sv[STG];
    stage_load(unit_of(p, wg), true, sv);
    stage_store(s_in, sv);
  }
  __syncthreads();

  for (int it = 0; it <= nu; ++it) {
    // (opaque copy of the lane index: everything derived from it is recomputed per tile -- a handful of VALU instructions -- instead of being
    // hoisted out of the loop into registers that then spill; a spill reload in here waits on vmcnt behind the staging loads in flight)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int khalf = lv >> 5, l31 = lv & 31;
    u32x4 sv[STG];
    const bool more = it + 1 < nu && !(p.dbg & 4);
    // (issued unconditionally -- all offsets out of range when there is no next tile -- so that `sv` is not merged with an undefined value
    // after a branch: the copies of such a merge would wait for the loads on the spot)
    stage_load(unit_of(p, wg + (more ? it + 1 : it) * p.wgs_per_group), more, sv);
    bool done1 = false;     // (wave-uniform: has this wave passed barrier 1 of the tile yet)
    // finished 16-byte output pieces of waves 4-7: every residual word is consumed BEFORE the first store is issued (loads and stores retire out
    // of order with respect to each other, so a load needed after a store costs vmcnt(0)), and the stores are issued AFTER barrier 2 so that
    // the other waves do not wait for their issue
    u32x4 ov[2][2], tv[2][2];
    size_t eoff[2];
    bool eok[2] = {false, false};
    if (!second) {
      if (it < nu) {
        // ================= conv_res0 on tile `it`: hidden fragments f = wq, wq + 4, wq + 8
        const Unit t = unit_of(p, wg + it * p.wgs_per_group);
        const char* in = s_in;
        char* hid = s_hid + (it & 1) * HID_BYTES;
        const float* csb = s_cs + t.b * 2 * CH;
        for (int f = wq; f < HFRAG; f += 4) {
          if (p.dbg & 1) break;
          const int pp = f * 32 + l31;                 // hidden pixel (flat index in the 10 x 34 region)
          const int pc = min(pp, HPIX - 1);
          const int hr = pc / HC, hc = pc - hr * HC;
          const int r0 = hr * IC + hc;                 // input row of tap (0, 0)
          f32x16 acc0, acc1;
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
          // 18 k-steps (tap, 16-channel half); fragments are read three steps ahead (one wave of either kind per SIMD: the ring, not the
          // other wave, has to cover the LDS latency of the dependent read -> two MFMAs chain)
          const char* ib = in + r0 * ISTR + (khalf << 4);
          auto ld = [&](int q) -> bf16x8 {
            const int tap = q >> 1, sh = q & 1;
            return *reinterpret_cast<const bf16x8*>(ib + ((tap / 3) * IC + (tap % 3)) * ISTR + (sh << 5));
          };
          bf16x8 fr[3] = {ld(0), ld(1), ld(2)};
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int q = 0; q < 18; ++q) {
            const bf16x8 cur = fr[q % 3];
            if (q + 3 < 18) fr[q % 3] = ld(q + 3);
            acc0 = mfma16(wr[q * 2 + 0], cur, acc0);
            acc1 = mfma16(wr[q * 2 + 1], cur, acc1);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (q + 3 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          // epilogue: a = mp_silu(y * c) in bf16, zero outside the image; lane holds channels 32 i + 8 jj + 4 khalf + e of pixel pp
          const int gh = t.h0 - 1 + hr, gw = t.w0 - 1 + hc;
          const bool inside = gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
          if (pp < HPIX && !(p.dbg & 16)) {
            // mp_silu(y c) = y (c / 0.596) / (1 + 2^(y c (-log2 e))) on PAIRS of channels (v_pk_mul_f32 / v_pk_add_f32; the two scaled copies
            // of c come from LDS), zero outside the image by masking the packed bf16 pair
            const unsigned keep = inside ? 0xffffffffu : 0u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const f32x4 ce = *reinterpret_cast<const f32x4*>(csb + 32 * i + 8 * jj + 4 * khalf);
                const f32x4 cm = *reinterpret_cast<const f32x4*>(csb + CH + 32 * i + 8 * jj + 4 * khalf);
                unsigned pk2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const f32x2_t y = i == 0 ? f32x2_t{acc0[4 * jj + 2 * h], acc0[4 * jj + 2 * h + 1]} : f32x2_t{acc1[4 * jj + 2 * h], acc1[4 * jj + 2 * h + 1]};
                  const f32x2_t ze = y * f32x2_t{ce[2 * h], ce[2 * h + 1]};
                  const f32x2_t den = f32x2_t{__builtin_amdgcn_exp2f(ze[0]), __builtin_amdgcn_exp2f(ze[1])} + f32x2_t{1.0f, 1.0f};
                  const f32x2_t a = (y * f32x2_t{cm[2 * h], cm[2 * h + 1]}) * f32x2_t{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                  pk2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t)) & keep;
                }
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                *reinterpret_cast<u32x2*>(hid + pp * HSTR + ((4 * i + jj) << 4) + khalf * 8) = u32x2{pk2[0], pk2[1]};
              }
          }
        }
      }
    } else {
      // ================= waves 4-7: stage tile it + 1, conv_res1 + epilogue of tile it - 1
      if (it >= 1) {
        const Unit t = unit_of(p, wg + (it - 1) * p.wgs_per_group);
        const char* hid = s_hid + ((it - 1) & 1) * HID_BYTES;
        // residual of this wave's two tile rows in the ACCUMULATOR's channel order: lane (khalf, l31) = pixel (row 2 wq + j, column l31), register
        // 4 jj + e = channel 8 jj + 4 khalf + e of the group's 32
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        u32x2 rr[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int oh = t.h0 + 2 * wq + j, ow = t.w0 + l31;
          eok[j] = oh < p.H && ow < p.W && !(p.dbg & 8);
          eoff[j] = (((size_t)t.b * p.H + (eok[j] ? oh : 0)) * p.W + (eok[j] ? ow : 0)) * p.C + g * CO;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) rr[j][jj] = *reinterpret_cast<const u32x2*>(p.res + eoff[j] + 8 * jj + 4 * khalf);
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (!(p.dbg & 2)) {
          // hidden rows 2 wq + h, h < 4: fragment (h, dx, s) is tap (h, dx) of tile row 0 (h < 3) and tap (h - 1, dx) of tile row 1 (h > 0):
          // 48 reads for 72 MFMAs, read three steps ahead
          const char* hb = hid + ((2 * wq) * HC + l31) * HSTR + (khalf << 4);
          auto ld = [&](int q) -> bf16x8 {
            const int h = q / 12, dx = (q / 4) % 3, sh = q & 3;
            return *reinterpret_cast<const bf16x8*>(hb + (h * HC + dx) * HSTR + (sh << 5));
          };
          bf16x8 fr[3] = {ld(0), ld(1), ld(2)};
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int q = 0; q < 48; ++q) {
            const int h = q / 12, dx = (q / 4) % 3, sh = q & 3;
            const bf16x8 cur = fr[q % 3];
            if (q + 3 < 48) fr[q % 3] = ld(q + 3);
            if (h < 3) acc[0] = mfma16(wr[(h * 3 + dx) * 4 + sh], cur, acc[0]);
            if (h > 0) acc[1] = mfma16(wr[((h - 1) * 3 + dx) * 4 + sh], cur, acc[1]);
            if (h > 0 && h < 3) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q + 3 < 48) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        // epilogue: x = clip(mp_sum(res, y)), twin = mp_silu(out2_scale * x) in fp32 on the accumulators; then bf16 pairs, and one
        // v_permlane32_swap per dword hands the lower half-wave channels 16 pl .. + 7 and the upper half-wave 16 pl + 8 .. + 15 of its pixel
        // (conv_dma.hip's register epilogue): two 16-byte stores per lane and output, the four pieces of a pixel's 64-byte slice back to back
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned pk[8], pt[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned w = rr[j][jj][e >> 1];
              const float rv = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
              x[e] = rv * p.res_a + acc[j][4 * jj + e] * p.res_b;
              if (p.clip > 0.f) x[e] = fminf(fmaxf(x[e], -p.clip), p.clip);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x2_t f2 = {x[2 * h], x[2 * h + 1]};
              pk[2 * jj + h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2, bf16x2_t));
              if (TWIN) {
                const f32x2_t t2 = {mp_silu_f(x[2 * h] * p.out2_scale), mp_silu_f(x[2 * h + 1] * p.out2_scale)};
                pt[2 * jj + h] = __builtin_bit_cast(unsigned, __builtin_convertvector(t2, bf16x2_t));
              }
            }
          }
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * pl], pk[4 * pl + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * pl + 1], pk[4 * pl + 3], false, false);
            ov[j][pl] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            if (TWIN) {
              const auto t0 = __builtin_amdgcn_permlane32_swap(pt[4 * pl], pt[4 * pl + 2], false, false);
              const auto t1 = __builtin_amdgcn_permlane32_swap(pt[4 * pl + 1], pt[4 * pl + 3], false, false);
              tv[j][pl] = u32x4{t0[0], t1[0], t0[1], t1[1]};
            }
          }
        }
        // barrier 1: every wave is done reading the input tile (and this tile's hidden rows are written)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage_store(s_in, sv);
        done1 = true;
      }
    }
    if (!done1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stage_store(s_in, sv);
    }
    // barrier 2: the next input tile is in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (second && it >= 1) {     // output stores of tile it - 1: left in flight under the next tile
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          if (eok[j]) *reinterpret_cast<u32x4*>(p.out + eoff[j] + 16 * pl + 8 * khalf) = ov[j][pl];
          if (TWIN && eok[j]) *reinterpret_cast<u32x4*>(p.out2 + eoff[j] + 16 * pl + 8 * khalf) = tv[j][pl];
        }
    }
  }
}

}  // namespace

bool conv_pair_supported(int B, int C, int groups, int hidden, int dtype) {
  return dtype == DDX_BF16 && groups > 0 && C == groups * CI && hidden == groups * CH && B > 0 &&
         IN_BYTES + 2 * HID_BYTES + B * CH * 8 <= 160 * 1024;
}

}  // namespace ddx

using namespace ddx;

extern "C" int ddx_mpconv_pair_supported(int32_t B, int32_t C, int32_t groups, int32_t hidden, int32_t dtype) {
  return conv_pair_supported(B, C, groups, hidden, dtype) ? 1 : 0;
}

extern "C" int ddx_mpconv_pair_fwd(const ddx_conv_pair_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "conv_pair: null descriptor");
  const ddx_conv_pair_desc d = *dp;
  if (!d.src || !d.wp0 || !d.wp1 || !d.chan_scale || !d.residual || !d.out) return set_error(DDX_ERR_ARG, "conv_pair: null buffer");
  if (d.B <= 0 || d.H <= 0 || d.W <= 0) return set_error(DDX_ERR_ARG, "conv_pair: bad size");
  if (!conv_pair_supported(d.B, d.C, d.groups, d.hidden, d.dtype))
    return set_error(DDX_ERR_UNSUPPORTED, "conv_pair: bf16, 32 -> 64 -> 32 channels per group only (run the two convs)");
  if (d.CK0 != 32 || d.CK1 != 32) return set_error(DDX_ERR_UNSUPPORTED, "conv_pair: prepared weights with 32-channel chunks only");
  PairArgs a{};
  a.x = (const bf16*)d.src; a.res = (const bf16*)d.residual; a.w0 = (const bf16*)d.wp0; a.w1 = (const bf16*)d.wp1; a.cs = d.chan_scale;
  a.out = (bf16*)d.out; a.out2 = (bf16*)d.out2;
  a.B = d.B; a.H = d.H; a.W = d.W; a.C = d.C; a.G = d.groups;
  a.tiles_h = (d.H + TH - 1) / TH; a.tiles_w = (d.W + TW - 1) / TW;
  a.units = d.B * a.tiles_h * a.tiles_w;
  // (CU count and the ablation bits are read once: this runs on every launch of an eager plan)
  static const int cus = []() {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  static const int ablate = []() { const char* e = std::getenv("DDX_ABLATE"); return e ? std::atoi(e) : 0; }();
  a.wgs_per_group = std::max(1, std::min(cus / d.groups, a.units));
  const float t = d.res_t, nrm = std::sqrt((1.f - t) * (1.f - t) + t * t);
  a.dbg = ablate;
  a.res_a = (1.f - t) / nrm; a.res_b = t / nrm; a.clip = d.clip; a.out2_scale = d.out2_scale;
  const int smem = IN_BYTES + 2 * HID_BYTES + d.B * CH * 8;
  const double px = (double)d.B * d.H * d.W;
  const double flops = 2.0 * px * (double)d.hidden * CI * 9 + 2.0 * px * (double)d.C * CH * 9;
  const double bytes = 2.0 * px * d.C * (d.out2 ? 4.0 : 3.0) + 2.0 * 2.0 * (double)d.hidden * CI * 9;
  return dispatch([a, smem](hipStream_t s) -> int {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return set_error(DDX_ERR_LAUNCH, "conv_pair: LDS attribute");
      attr = true;
    }
    if (a.out2) hipLaunchKernelGGL(conv_pair_kernel<true>, dim3(a.wgs_per_group * a.G), dim3(512), smem, s, a);
    else hipLaunchKernelGGL(conv_pair_kernel<false>, dim3(a.wgs_per_group * a.G), dim3(512), smem, s, a);
    return check_launch("conv_pair");
  }, stream, "conv3x3_pair", flops, bytes);
}
