// Magnitude-preserving conv2d forward as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces F.conv2d in MPConv.forward (reference src/modules/mp_tools.py:366-373) together with the
// element-wise neighbours of Block.forward (src/modules/unets/unet_edm2_b4.py:110-158).
//
// Work decomposition (one 256-thread workgroup = 4 waves):
//   output tile  = TH x TW pixels of one image (<= BM = WM*MF*32)  x  BN = WN*NF*32 output channels of one group
//   K loop       = chunks of CK input channels of the group; per chunk
//       stage   : the (TH+2)x(TW+2) input halo of the chunk -> LDS, ONCE (not 9x im2col), with the fused
//                 prologue (mp_cat scales, y*c, mp_silu, nearest-up / avg-pool gather) applied in fp32 on the way;
//                 the chunk of prepared weights [tap][BN][CK] -> LDS
//       compute : for each of the ks*ks taps the activation fragment is the same LDS tile read at a shifted row,
//                 v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (fp32 parity path), fp32 accumulate
//   Staging is register-staged and split (issue early / write late): all global loads of chunk c+1 are issued
//   back-to-back BEFORE the MFMA phase of chunk c and only converted + written to LDS after it, so HBM/L2 latency
//   hides under the matrix work instead of serialising one round trip per loader iteration.
//   MFMA orientation: A operand = weights (rows = output channels), B operand = activations (cols = pixels), so
//   every lane ends up with 4 consecutive output channels of ONE pixel -> vector epilogue on NHWC rows.
//   LDS rows are padded by 16 B: a row stride of CK*sizeof(T)+16 bytes makes the 16-lane groups of ds_read_b128 hit
//   16 distinct bank slots (stride 80 B -> slot = 5r mod 16, 144 B -> 9r, 272 B -> 17r; all bijective mod 16).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_params.hpp"

namespace ddx {

template <typename T> struct Mfma;
template <> struct Mfma<bf16> {
  static constexpr int KM = 16;
  using Frag = bf16x8;
  static __device__ __forceinline__ Frag load(const bf16* row_k, int khalf) { return *(const bf16x8*)(row_k + khalf * 8); }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static constexpr int KM = 2;
  using Frag = float;
  static __device__ __forceinline__ Frag load(const float* row_k, int khalf) { return row_k[khalf]; }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// halo rows a tile configuration may stage (bounds the per-thread register staging)
constexpr int arows_max(int KS, int BM) { return KS == 3 ? (BM == 256 ? 352 : 208) : BM; }

// KSP > 1: intra-workgroup split-K for small-M (latency-bound) layers.  The workgroup has KSP groups of 4 waves; group q
// stages and multiplies chunks q, q+KSP, ... in its own LDS region, and the partial accumulators are summed through the
// epilogue's LDS transpose buffer.  KSP x more loads in flight per CU and KSP x fewer serial K iterations.
template <typename T, int KS, int CK, int WM, int WN, int MF, int NF, bool DN, int KSP>
__global__ __launch_bounds__(256 * KSP, (KSP == 1 ? 2 : 1)) void conv_mfma_kernel(const ConvParams p) {
  static_assert(WM * WN == 4, "4 waves per K group");
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int EV = 16 / (int)sizeof(T);
  constexpr int V = CK / EV;
  constexpr int RPP = 256 / V;  // halo rows covered per staging pass
  constexpr int STRIDE = CK + EV;
  constexpr int BM = WM * MF * 32, BN = WN * NF * 32;
  constexpr int KM = Mfma<T>::KM;
  constexpr int AI = (arows_max(KS, BM) * V + 255) / 256;  // activation vectors per thread per chunk
  constexpr int BI = (TAPS * BN * V + 255) / 256;           // weight vectors per thread per chunk
  constexpr bool PIPE = sizeof(T) == 2;                     // prefetch across the MFMA phase (bf16 path)
  using Frag = typename Mfma<T>::Frag;
  using VT = decltype(Vec16<T>().v);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kq = (KSP == 1) ? 0 : (int)(threadIdx.x >> 8);  // K group of this wave
  T* sA = reinterpret_cast<T*>(smem + (size_t)kq * p.group_smem);
  T* sB = sA + (size_t)p.arows_alloc * STRIDE;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // tid = thread index inside the K group
  const int wm = wave / WN, wn = wave % WN;
  const int khalf = lane >> 5, l31 = lane & 31;

  // tile decode: blockIdx.x = (b, tile_y, tile_x), blockIdx.y = n tile, blockIdx.z = group
  int bx = blockIdx.x;
  const int tx = bx % p.tiles_w; bx /= p.tiles_w;
  const int ty = bx % p.tiles_h;
  const int b = bx / p.tiles_h;
  const int n0 = blockIdx.y * BN;
  const int g = blockIdx.z;
  const int h0 = ty * p.TH, w0 = tx * p.TW;
  const int TW = p.TW, TWP = TW + 2 * PAD;
  const int R = (p.TH + 2 * PAD) * TWP;
  const int MT = p.TH * TW;
  const float inv_TW = 1.0f / (float)TW;

  int arow[MF];
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int ml = (wm * MF + j) * 32 + l31;
    const int th = (int)(((float)ml + 0.5f) * inv_TW);
    const int tw = ml - th * TW;
    arow[j] = (ml < MT) ? th * TWP + tw : 0;
  }

  f32x16 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int v = tid % V;
  const int r0 = tid / V;
  const T* wp = reinterpret_cast<const T*>(p.wp);
  // (n0 is workgroup-uniform: a merged conv switches the prologue per output-channel tile)
  const int pro = (p.pro_rows > 0 && g * p.Ng + n0 >= p.pro_rows) ? DDX_PRO_NONE : p.prologue;
  const bool do_silu = (pro & DDX_PRO_SILU) != 0;

  // per-item source pixel offsets (in pixels of the source image); invalid items load pixel 0 and are zeroed at
  // commit time, so the issue phase is straight-line code (no per-item branches -> no forced vmcnt(0) at joins)
  int apix[AI];
  unsigned avalid = 0;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = r0 + i * RPP;
    const int hh = (int)(((float)r + 0.5f) * p.inv_TWP);
    const int ww = r - hh * TWP;
    const int ih = h0 - PAD + hh;
    int iw = w0 - PAD + ww;
    if (p.reflect_w) iw = iw < 0 ? -iw : (iw >= p.W && iw < p.W + PAD ? 2 * (p.W - 1) - iw : iw);  // only the true border mirrors
    const bool ok = r < R && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
    int pix;
    if (DN) pix = (b * p.sH + 2 * ih) * p.sW + 2 * iw;
    else if (p.resample == DDX_RESAMPLE_UP) pix = (b * p.sH + (ih >> 1)) * p.sW + (iw >> 1);
    else pix = (b * p.sH + ih) * p.sW + iw;
    apix[i] = ok ? pix : b * p.sH * p.sW;   // (invalid items read the image's first pixel: in bounds for the swapped source too)
    avalid |= (ok ? 1u : 0u) << i;
  }

  VT areg[AI];
  u32x4 breg[BI];
  float cs[EV];
  bool cvalid_cur = false;
  float sscale_cur = 1.0f;

  // ---- issue: the global loads of one chunk, nothing consumed (except the rare DN gather).  Split into a setup
  // step and per-item loads so that the bf16 kernel can spread the loads between the MFMA groups of the previous
  // chunk: a burst of 15 KiB per wave in front of the matrix phase keeps the wave in the (L1-throughput bound)
  // VMEM issue queue and nothing overlaps.
  const T* a_src = nullptr;
  int a_cs = 0, a_cc = 0;
  const T* w_src = nullptr;
  auto issue_setup = [&](int ch) {
    const int cg = ch * CK + v * EV;  // channel inside the group
    const bool cvalid = cg < p.Cg && ch < p.nchunk;  // (split-K groups may run past the last chunk: zero operand)
    const int cabs = cvalid ? g * p.Cg + cg : 0;   // channel of the (virtually concatenated) input
    const bool second_half = p.paired && cabs >= p.C0 + p.C1;   // [src0 | src1 | src0' | src1']
    const int cpart = second_half ? cabs - (p.C0 + p.C1) : cabs;
    const bool first = cpart < p.C0;
    a_src = reinterpret_cast<const T*>(first ? p.src0 : p.src1);
    a_cs = first ? p.C0 : p.C1;
    if (second_half || (!first && p.swap1)) a_src += (ptrdiff_t)((b ^ 1) - b) * p.sH * p.sW * a_cs;   // pair-swapped image
    a_cc = first ? cpart : cpart - p.C0;
    cvalid_cur = cvalid;
    sscale_cur = first ? p.scale0 : p.scale1;
    if (pro & DDX_PRO_SCALE) {  // raw per-(b, channel) factors; folded with the source scale at commit time
      const float* csp = p.cscale + (size_t)b * p.Cin + cabs;
#pragma unroll
      for (int e = 0; e < EV; e += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(csp + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) cs[e + q] = t4[q];
      }
    } else {
#pragma unroll
      for (int e = 0; e < EV; ++e) cs[e] = 1.0f;
    }
    w_src = wp + ((size_t)(g * p.nchunk + min(ch, p.nchunk - 1)) * TAPS * p.NgP) * CK;
  };
  auto issue_a = [&](int i) {
    const T* sp = a_src + (size_t)apix[i] * a_cs + a_cc;
    if (DN) {  // 2x2 average on the fly (the 1x1 skip conv of "down" blocks)
      Vec16<T> a0, a1, a2, a3;
      a0.v = *reinterpret_cast<const VT*>(sp);
      a1.v = *reinterpret_cast<const VT*>(sp + a_cs);
      a2.v = *reinterpret_cast<const VT*>(sp + (size_t)p.sW * a_cs);
      a3.v = *reinterpret_cast<const VT*>(sp + (size_t)p.sW * a_cs + a_cs);
#pragma unroll
      for (int e = 0; e < EV; ++e) a0.set(e, 0.25f * ((a0.get(e) + a1.get(e)) + (a2.get(e) + a3.get(e))));
      areg[i] = a0.v;
    } else {
      areg[i] = *reinterpret_cast<const VT*>(sp);
    }
  };
  auto issue_b = [&](int i) {
    const int idx = min(tid + i * 256, TAPS * BN * V - 1);
    const int v2 = idx % V;
    const int rn = idx / V;
    const int tap = rn / BN;
    const int n = min(n0 + rn - tap * BN, p.NgP - 1);  // rows past NgP only feed outputs that are never stored
    breg[i] = *reinterpret_cast<const u32x4*>(w_src + ((size_t)(tap * p.NgP + n)) * CK + v2 * EV);
  };
  auto issue = [&](int ch) {
    issue_setup(ch);
#pragma unroll
    for (int i = 0; i < AI; ++i) issue_a(i);
#pragma unroll
    for (int i = 0; i < BI; ++i) issue_b(i);
  };

  // ---- commit: fused prologue in fp32, pack, write LDS
  auto commit_a = [&](auto silu_tag) {
    constexpr bool SILU = decltype(silu_tag)::value;
    float csf[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) csf[e] = cs[e] * sscale_cur;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int r = r0 + i * RPP;
      Vec16<T> a, o;
      a.v = areg[i];
      const bool ok = cvalid_cur && ((avalid >> i) & 1u);
#pragma unroll
      for (int e = 0; e < EV; ++e) {
        float x = a.get(e) * csf[e];
        if (SILU) x = mp_silu_f(x);
        o.set(e, x);
      }
      VT ov = o.v;
      if (!ok) {
        Vec16<T> z;
#pragma unroll
        for (int e = 0; e < EV; ++e) z.set(e, 0.f);
        ov = z.v;
      }
      if (r < R) *reinterpret_cast<VT*>(sA + (size_t)r * STRIDE + v * EV) = ov;
    }
  };
  // prologue-free fast path: the staged vectors go to LDS untouched (no unpack / scale / repack VALU work)
  const bool do_raw = !DN && pro == DDX_PRO_NONE && p.scale0 == 1.0f && (p.src1 == nullptr || p.scale1 == 1.0f);
  auto commit_raw = [&]() {
    Vec16<T> z;
#pragma unroll
    for (int e = 0; e < EV; ++e) z.set(e, 0.f);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int r = r0 + i * RPP;
      const bool ok = cvalid_cur && ((avalid >> i) & 1u);
      if (r < R) *reinterpret_cast<VT*>(sA + (size_t)r * STRIDE + v * EV) = ok ? areg[i] : z.v;
    }
  };
  auto commit = [&]() {
    if (do_raw) commit_raw();
    else if (do_silu) commit_a(std::true_type{});
    else commit_a(std::false_type{});
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int idx = tid + i * 256;
      if (idx < TAPS * BN * V) *reinterpret_cast<u32x4*>(sB + (size_t)(idx / V) * STRIDE + (idx % V) * EV) = breg[i];
    }
  };

  // Keep the staged registers opaque until the MFMA phase is over: without this hipcc hoists the bf16->f32 unpacking
  // of the first loads above the matrix loop, which puts a vmcnt wait (one memory latency) in front of every chunk.
  auto pin = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) asm volatile("" : "+v"(areg[i]));
#pragma unroll
    for (int i = 0; i < BI; ++i) asm volatile("" : "+v"(breg[i]));
#pragma unroll
    for (int e = 0; e < EV; ++e) asm volatile("" : "+v"(cs[e]));
  };

  // MFMA phase of the current chunk; with `prefetch` the loads of the next chunk are issued in small groups
  // between the MFMA groups (SLOTS groups per chunk)
  constexpr int SLOTS = TAPS * (CK / KM);
  constexpr int LPS = (AI + BI + SLOTS - 1) / SLOTS;  // loads per slot
  auto compute = [&](auto prefetch_tag) {
    constexpr bool prefetch = decltype(prefetch_tag)::value;
    // fragments of slot s+1 are read while the MFMAs of slot s run (explicit double buffering), so that the
    // scheduling barrier that pins the global loads between the MFMA groups does not expose LDS latency
    Frag wf[2][NF], xf[2][MF];
    auto load_frags = [&](int slot, int buf) {
      const int tap = slot / (CK / KM), ks = slot % (CK / KM);
      const int toff = (tap / KS) * TWP + (tap % KS);
#pragma unroll
      for (int i = 0; i < NF; ++i)
        wf[buf][i] = Mfma<T>::load(sB + (size_t)(tap * BN + (wn * NF + i) * 32 + l31) * STRIDE + ks * KM, khalf);
#pragma unroll
      for (int j = 0; j < MF; ++j) xf[buf][j] = Mfma<T>::load(sA + (size_t)(arow[j] + toff) * STRIDE + ks * KM, khalf);
    };
    load_frags(0, 0);
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
      const int cur = slot & 1;
      if (slot + 1 < SLOTS) load_frags(slot + 1, cur ^ 1);
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = Mfma<T>::mma(wf[cur][i], xf[cur][j], acc[i][j]);
      if (PIPE && prefetch) {
#pragma unroll
        for (int q = 0; q < LPS; ++q) {
          const int li = slot * LPS + q;
          if (li < AI) issue_a(li);
          else if (li < AI + BI) issue_b(li - AI);
        }
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every load behind the last MFMA
      }
    }
  };

  const int nit = (p.nchunk + KSP - 1) / KSP;  // K iterations of every group (same count: shared barriers)
  issue(kq);
  commit();
  __syncthreads();
  if constexpr (PIPE) {
    // steady state: chunk ch is multiplied while chunk ch+1 is loaded (interleaved), then converted and written.
    // The last chunk is peeled so that the staged registers never meet at a control-flow join (a phi there makes
    // hipcc copy them right after the loads, i.e. wait for every load inside the matrix phase).
    for (int it = 0; it + 1 < nit; ++it) {
      issue_setup((it + 1) * KSP + kq);
      compute(std::true_type{});
      __syncthreads();  // every wave is done reading this chunk
      pin();
      commit();
      __syncthreads();
    }
    compute(std::false_type{});
  } else {
    for (int it = 0; it < nit; ++it) {
      compute(std::false_type{});
      __syncthreads();
      if (it + 1 < nit) {
        issue((it + 1) * KSP + kq);
        commit();
        __syncthreads();
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- epilogue
  // The accumulators (lane = one pixel, 4-channel runs) are transposed through LDS so that global traffic is fully
  // coalesced: 16 consecutive lanes cover the BN contiguous channels of one NHWC pixel (whole 128-byte lines for the
  // residual read and the store) instead of 32 lanes touching 32 different lines with 8 bytes each.
  constexpr int ES = BN + 4;  // fp32 row stride of the transpose buffer: 8 rows x (ES*4 B) tile all 32 banks once
  float* sE = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int kg = 0; kg < KSP; ++kg) {  // K groups add their partial sums one after the other
    if (kq == kg) {
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        const int ml = (wm * MF + j) * 32 + l31;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* dst = sE + (size_t)ml * ES + (wn * NF + i) * 32 + 8 * q + 4 * khalf;
            f32x4 y4;
            if (kg == 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) y4[e] = acc[i][j][4 * q + e];
            } else {
              y4 = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
              for (int e = 0; e < 4; ++e) y4[e] += acc[i][j][4 * q + e];
            }
            *reinterpret_cast<f32x4*>(dst) = y4;
          }
      }
    }
    __syncthreads();
  }
  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.res);
  constexpr int G4 = BN / 4;            // 4-channel groups per pixel
  constexpr int NT = 256 * KSP;         // all threads of the workgroup share the store phase
  constexpr int EI = (BM * G4 + NT - 1) / NT;  // items per thread
  const int tid_all = threadIdx.x;
  using V4 = decltype(Vec4<T>().v);
  // all residual loads are issued first (clamped addresses, no per-item branches), then combined and stored:
  // a load->wait->store chain per item would cost one memory latency per item
  long eoff[EI];
  V4 rres[EI];
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = tid_all + it * NT;
    const int ml = idx / G4, c4 = idx % G4;
    const int th = (int)(((float)ml + 0.5f) * inv_TW);
    const int tw = ml - th * TW;
    const int h = h0 + th, w = w0 + tw;
    const int n = n0 + c4 * 4;
    const bool ok = idx < BM * G4 && ml < MT && h < p.H && w < p.W && n < p.Ng;
    eoff[it] = ok ? (long)((((size_t)b * p.H + h) * p.W + w) * p.Cout + (size_t)g * p.Ng + n) : -1;
    if (p.epilogue == DDX_EPI_MPSUM) {  // (residual loads issued here: with res_up the address is the half-size pixel's)
      const long roff = p.res_up ? (long)((((size_t)b * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Cout + (size_t)g * p.Ng + n) : eoff[it];
      rres[it] = *reinterpret_cast<const V4*>(res + (ok ? roff : 0));
    }
  }
  if (p.epilogue == DDX_EPI_SILU_BWD) {
    // data gradient through the producer-side activation a = mp_silu(y * c * s) (ddx_mpconv2d_dgrad_act, the LDS-DMA kernel's EB
    // epilogue on this kernel's item map): dz = g * silu'(y c s), out = dz * c * s (+ add), dc[b][ch] += s * sum_pixels dz * y.
    // The output channels may belong to two tensors (the two mp_cat sources): a 4-channel item lies in one of them (split % 4 == 0).
    // NT % G4 == 0: every item of a thread has the same 4 channels, so the dc partial sums live in four registers.
    const int ch = g * p.Ng + n0 + (tid_all % G4) * 4;
    const bool second = p.bwd_split > 0 && ch >= p.bwd_split;
    const int ld_u = p.bwd_split > 0 ? (second ? p.Cout - p.bwd_split : p.bwd_split) : p.Cout;
    const int cu = second ? ch - p.bwd_split : ch;
    const T* y_u = second ? reinterpret_cast<const T*>(p.bwd_y1) : res;
    T* out_u = second ? reinterpret_cast<T*>(p.bwd_out1) : out;
    const T* addp = reinterpret_cast<const T*>(p.bwd_add);
    const float sc_u = second ? p.bwd_s1 : p.bwd_s0;
    const bool ch_ok = n0 + (tid_all % G4) * 4 < p.Ng;
    float s4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s4[e] = sc_u;
    if (p.out_cs && ch_ok) {
      const f32x4 c4v = *reinterpret_cast<const f32x4*>(p.out_cs + (size_t)b * p.Cout + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) s4[e] *= c4v[e];
    }
    long uoff[EI];
    V4 ry[EI], ra[EI];
#pragma unroll
    for (int it = 0; it < EI; ++it) {
      const long pix = eoff[it] < 0 ? 0 : (eoff[it] - ch) / p.Cout;      // NHWC pixel index of the item (eoff = pix * Cout + ch)
      uoff[it] = eoff[it] < 0 ? -1 : pix * ld_u + cu;
      ry[it] = *reinterpret_cast<const V4*>(y_u + (uoff[it] < 0 ? 0 : uoff[it]));
      if (addp) ra[it] = *reinterpret_cast<const V4*>(addp + (eoff[it] < 0 ? 0 : eoff[it]));
    }
    float dcs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < EI; ++it) {
      const int idx = min(tid_all + it * NT, BM * G4 - 1);
      const int ml = idx / G4, c4 = idx % G4;
      const f32x4 y4 = *reinterpret_cast<const f32x4*>(sE + (size_t)ml * ES + c4 * 4);
      if (uoff[it] < 0) continue;
      Vec4<T> yv, av, ov;
      yv.v = ry[it];
      if (addp) av.v = ra[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yy = yv.get(e);
        float dz = y4[e];
        if (p.bwd_act) {
          const float z = yy * s4[e];
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
          dz *= sg * (1.0f + z * (1.0f - sg)) * kMpSiluInv;
        }
        dcs[e] += dz * yy;
        ov.set(e, dz * s4[e] + (addp ? av.get(e) : 0.f));
      }
      *reinterpret_cast<V4*>(out_u + uoff[it]) = ov.v;
    }
    if (p.bwd_dc) {   // lanes l, l + G4, ... of a wave hold the same channels: one atomic per (wave, channel)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sacc = dcs[e];
#pragma unroll
        for (int m = G4; m < 64; m <<= 1) sacc += __shfl_xor(sacc, m, 64);
        dcs[e] = sacc;
      }
      if ((tid_all & 63) < G4 && ch_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.bwd_dc + (size_t)b * p.Cout + ch + e, dcs[e] * p.bwd_s0);
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < EI; ++it) {
    const int idx = min(tid_all + it * NT, BM * G4 - 1);
    const int ml = idx / G4, c4 = idx % G4;
    const f32x4 y4 = *reinterpret_cast<const f32x4*>(sE + (size_t)ml * ES + c4 * 4);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = y4[e];
    if (p.epilogue == DDX_EPI_MPSUM) {
      Vec4<T> rv;
      rv.v = rres[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = rv.get(e) * p.res_a + y[e] * p.res_b;
    }
    if (p.clip > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fminf(fmaxf(y[e], -p.clip), p.clip);
    }
    if (p.head_norm) {
      // BN is a multiple of 64 here (conv.hip checks): 16 consecutive lanes hold the 64 channels of one head of one pixel
      // (ddx_conv_desc::out_head_norm): y / (eps + |y| / sqrt(64)).  Every lane takes part (items past the tile read row BM * G4 - 1).
      float ss = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float sc = 1.0f / (p.head_eps + sqrtf(ss) * 0.125f);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] *= sc;
    }
    if (eoff[it] < 0) continue;
    if (p.out2) {  // activated twin for the next block's conv_res0
      Vec4<T> tv;
      if (p.out2_linear) {   // linear twin y * c2[b][c]: the operand x * c_qk of the attention block's q|k conv, written by the conv that makes x
        const int ch = g * p.Ng + n0 + (idx % G4) * 4;
        const f32x4 c4v = *reinterpret_cast<const f32x4*>(p.out2_cs + (size_t)b * p.Cout + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) tv.set(e, y[e] * c4v[e]);
      } else if (p.out_cs && !p.out_act) {  // raw main output: the channel scale belongs to the twin (training forward keeps y AND mp_silu(y * c))
        const int ch = g * p.Ng + n0 + (idx % G4) * 4;
        const f32x4 c4v = *reinterpret_cast<const f32x4*>(p.out_cs + (size_t)b * p.Cout + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) tv.set(e, mp_silu_f(y[e] * c4v[e] * p.out2_scale));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) tv.set(e, mp_silu_f(y[e] * p.out2_scale));
      }
      *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out2) + eoff[it]) = tv.v;
    }
    if (p.out_act) {  // producer-side mp_silu(y * c): the consumer conv then stages its operand untouched
      if (p.out_cs) {
        const int ch = g * p.Ng + n0 + (idx % G4) * 4;  // output channel of this item (no 64-bit modulo on the offset)
        const f32x4 c4v = *reinterpret_cast<const f32x4*>(p.out_cs + (size_t)b * p.Cout + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] *= c4v[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = mp_silu_f(y[e]);
    }
    Vec4<T> ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov.set(e, y[e]);
    *reinterpret_cast<V4*>(out + eoff[it]) = ov.v;
  }
}

// ------------------------------------------------------------------------------------------- host side

static inline int elem_vec(int dtype) { return dtype == DDX_BF16 ? 8 : 4; }

bool conv_mfma_supported(const ConvParams& p, int ksize, int dtype) {
  const int ev = elem_vec(dtype);
  if (ksize != 1 && ksize != 3) return false;
  if (p.C0 % ev || (p.src1 && (p.C1 % ev))) return false;
  if (p.Cg % ev) return false;
  if (p.Ng % 4 || p.Cout % 4) return false;
  if ((p.prologue & DDX_PRO_SCALE) && (p.Cin % 4)) return false;
  if (p.CK != 32 && p.CK != 64 && p.CK != 128) return false;
  if (ksize == 3 && (p.CK != 32 || p.resample == DDX_RESAMPLE_DOWN)) return false;
  if ((size_t)p.B * p.sH * p.sW >= (size_t)1 << 31) return false;
  return true;
}

// best TH x TW (TH*TW <= BM) for an H x W image: maximise useful fraction, then minimise halo
static void best_tile(int H, int W, int BM, int pad, int max_rows, int* TH, int* TW, double* util) {
  double best = -1, best_halo = 1e30;
  int bth = 1, btw = 1;
  for (int tw = 1; tw <= W && tw <= BM; ++tw) {
    for (int th = 1; th <= H && th * tw <= BM; ++th) {
      const int rows = (th + 2 * pad) * (tw + 2 * pad);
      if (rows > max_rows) continue;
      const long tiles = (long)ceil_div(H, th) * ceil_div(W, tw);
      // a 32-lane activation fragment that stays inside one tile row reads 32 consecutive LDS rows (bank-conflict
      // free with the 16-byte row padding); fragments that wrap tile rows are ~2-way conflicted: small penalty
      const double u = (double)H * W / ((double)tiles * BM) * ((tw % 32 == 0 || pad == 0) ? 1.0 : 0.9);
      const double halo = (double)rows / (th * tw);
      if (u > best + 1e-9 || (u > best - 1e-9 && halo < best_halo)) { best = u; best_halo = halo; bth = th; btw = tw; }
    }
  }
  *TH = bth; *TW = btw; *util = best;
}

struct Choice { int BM, BN, TH, TW, KSP; size_t smem, group_smem; };

// Tile / split-K choice.  Large-M layers: BM=256 (2 workgroups per CU overlap each other).  Small-M layers (few
// workgroups, long serial K loop): BM=128 tiles with intra-workgroup split-K so that enough loads are in flight.
static Choice choose(const ConvParams& p, int ksize, int dtype) {
  const int pad = ksize / 2, taps = ksize * ksize;
  const size_t es = dtype_size(dtype);
  const int stride = p.CK + elem_vec(dtype);
  struct Cand { int BM, BN; };
  const Cand cands[4] = {{256, 64}, {256, 32}, {128, 64}, {128, 32}};
  Choice best{}; double best_score = -1;
  // p.force_cfg (ddx_conv_desc.force_direct >= 16: plan-time autotuning, tools/conv_sweep.sh) restricts the candidates to one
  int fbm = 0, fbn = 0, fksp = 0;
  if (p.force_cfg > 0) {
    const int c = (p.force_cfg - 1) / 3, k = (p.force_cfg - 1) % 3;
    if (c > 3) return Choice{};
    fbm = cands[c].BM; fbn = cands[c].BN; fksp = 1 << k;
  }
  for (const Cand& c : cands) {
    const int BM = c.BM, BN = c.BN;
    int TH, TW; double um;
    best_tile(p.H, p.W, BM, pad, arows_max(ksize, BM), &TH, &TW, &um);
    const int rows = (TH + 2 * pad) * (TW + 2 * pad);
    const size_t group = (((size_t)rows + (size_t)taps * BN) * stride * es + 255) / 256 * 256;
    const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
    const long wgs = (long)p.B * ceil_div(p.H, TH) * ceil_div(p.W, TW) * ceil_div(p.Ng, BN) * p.G;
    const int ksps[3] = {1, 2, 4};
    for (int ksp : ksps) {
      if (fbm && (BM != fbm || BN != fbn || ksp != fksp)) continue;
      if (ksp > 1 && (BM != 128 || dtype != DDX_BF16)) continue;       // split-K variants are built for BM=128 bf16
      if (ksp > 1 && p.resample == DDX_RESAMPLE_DOWN) continue;        // the avg-pool gather is only built for KSP=1
      if (ksp > 1 && p.nchunk < 2 * ksp) continue;                      // needs >= 2 iterations per group to pay
      if (ksp == 4 && (ksize == 3 || p.CK == 128)) continue;            // register budget (128 VGPRs at 1024 threads)
      const size_t smem = std::max(group * ksp, epi);
      if (smem > 160 * 1024) continue;
      const double un = (double)p.Ng / round_up(p.Ng, BN);
      // (32-channel tiles reuse a staged activation chunk half as often; a 3x3 layer does nine taps of matrix work per chunk, so it costs it
      // less: the plan-time tuner of round 4 preferred 128 x 32 tiles without split-K by >= 4 % on five level-3 conv_res0 shapes)
      double eff = (BN == 32 ? (ksize == 3 ? 0.86 : 0.80) : 1.0) * (BM == 128 ? 0.92 : 1.0);
      if (ksp == 1 && smem > 80 * 1024) eff *= 0.85;                    // one workgroup per CU: no phase overlap
      // parallelism: waves in flight relative to what fills the chip (256 CUs x 8 waves)
      const double fill = std::min(1.0, (double)wgs * 4 * ksp / 2048.0);
      eff *= 0.35 + 0.65 * fill;
      if (ksp > 1 && wgs >= 512) eff *= 0.8;                            // enough workgroups already: plain K loop
      // 3x3 layers do nine taps of matrix work per staged chunk, which hides the next chunk's latency on its own: measured
      // (force_direct >= 16 sweep through tools/conv_bench.py, L3 / L4 shapes) split-K only pays below one workgroup per CU (160 workgroups: 15.7 -> 13.2 us)
      // and costs 25-35 % from 320 workgroups up (12.2 -> 16.3, 15.0 -> 20.9 us)
      if (ksize == 3 && ksp > 1 && wgs >= 256) eff *= 0.7;
      const double score = um * un * eff;
      if (score > best_score) { best_score = score; best = Choice{BM, BN, TH, TW, ksp, smem, group}; }
    }
  }
  return best;
}

void conv_mfma_plan_tiles(ConvParams& p, int ksize, int dtype) {
  const Choice c = choose(p, ksize, dtype);
  const int pad = ksize / 2;
  p.TH = c.TH; p.TW = c.TW;
  p.tiles_h = ceil_div(p.H, c.TH); p.tiles_w = ceil_div(p.W, c.TW);
  p.arows_alloc = (c.TH + 2 * pad) * (c.TW + 2 * pad);
  p.inv_TWP = 1.0f / (float)(c.TW + 2 * pad);
  p.group_smem = (int)c.group_smem;
}

template <typename T, int KS, int CK, int WM, int WN, int MF, int NF, bool DN, int KSP>
static int launch_cfg3(const ConvParams& p, size_t smem, hipStream_t s) {
  auto kern = conv_mfma_kernel<T, KS, CK, WM, WN, MF, NF, DN, KSP>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_mfma)");
    attr_done = true;
  }
  constexpr int BN = WN * NF * 32;
  dim3 grid(p.B * p.tiles_h * p.tiles_w, ceil_div(p.Ng, BN), p.G);
  hipLaunchKernelGGL(kern, grid, dim3(256 * KSP), smem, s, p);
  return check_launch("conv_mfma");
}

template <typename T, int KS, int CK, int WM, int WN, int MF, int NF, int KSP>
static int launch_cfg(const ConvParams& p, size_t smem, hipStream_t s) {
  if constexpr (KS == 1 && KSP == 1) {
    if (p.resample == DDX_RESAMPLE_DOWN) return launch_cfg3<T, KS, CK, WM, WN, MF, NF, true, KSP>(p, smem, s);
  }
  return launch_cfg3<T, KS, CK, WM, WN, MF, NF, false, KSP>(p, smem, s);
}

template <typename T, int KS, int CK>
static int launch_ks(const ConvParams& p, const Choice& c, hipStream_t s) {
  if (c.KSP == 1) {
    if (c.BM == 256 && c.BN == 64) return launch_cfg<T, KS, CK, 4, 1, 2, 2, 1>(p, c.smem, s);
    if (c.BM == 256 && c.BN == 32) return launch_cfg<T, KS, CK, 4, 1, 2, 1, 1>(p, c.smem, s);
    if (c.BM == 128 && c.BN == 64) return launch_cfg<T, KS, CK, 2, 2, 2, 1, 1>(p, c.smem, s);
    if (c.BM == 128 && c.BN == 32) return launch_cfg<T, KS, CK, 4, 1, 1, 1, 1>(p, c.smem, s);
  }
  if constexpr (sizeof(T) == 2) {
    if (c.KSP == 2 && c.BM == 128 && c.BN == 64) return launch_cfg<T, KS, CK, 2, 2, 2, 1, 2>(p, c.smem, s);
    if (c.KSP == 2 && c.BM == 128 && c.BN == 32) return launch_cfg<T, KS, CK, 4, 1, 1, 1, 2>(p, c.smem, s);
    if (c.KSP == 4 && c.BM == 128 && c.BN == 32) return launch_cfg<T, KS, CK, 4, 1, 1, 1, 4>(p, c.smem, s);
    if constexpr (KS == 1) {
      if (c.KSP == 4 && c.BM == 128 && c.BN == 64) return launch_cfg<T, KS, CK, 2, 2, 2, 1, 4>(p, c.smem, s);
    }
  }
  return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: no tile configuration");
}

template <typename T>
static int launch_t(const ConvParams& p, int ksize, const Choice& c, hipStream_t s) {
  if (ksize == 3 && p.CK == 32) return launch_ks<T, 3, 32>(p, c, s);
  if (ksize == 1 && p.CK == 128) return launch_ks<T, 1, 128>(p, c, s);
  if (ksize == 1 && p.CK == 64) return launch_ks<T, 1, 64>(p, c, s);
  if (ksize == 1 && p.CK == 32) return launch_ks<T, 1, 32>(p, c, s);
  return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: ksize/CK combination not built");
}

int conv_mfma_tile_bn(const ConvParams& p, int ksize, int dtype) {
  const Choice c = choose(p, ksize, dtype);
  return c.BM == 0 ? 0 : c.BN;
}

int launch_conv_mfma(const ConvParams& p_in, int ksize, int dtype, hipStream_t s) {
  ConvParams p = p_in;
  const Choice c = choose(p, ksize, dtype);
  if (c.BM == 0) return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: no tile fits LDS");
  conv_mfma_plan_tiles(p, ksize, dtype);
  if (dtype == DDX_BF16) return launch_t<bf16>(p, ksize, c, s);
  return launch_t<float>(p, ksize, c, s);
}

}  // namespace ddx
