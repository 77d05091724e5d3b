// Self-attention over the H*W tokens of an NHWC feature map, flash-style on the gfx950 matrix cores.
//
// Replaces reference src/modules/unets/unet_edm2_b4.py:137-148: q,k,v RMS-normalised over the head dimension per
// token (normalize(dim=2)), then softmax(q k^T / sqrt(d)) v.  The q/k/v normalisation and the 1/sqrt(d) scale are
// fused into the LDS staging; the score matrix never leaves registers.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   S^T tile (32 keys x 32 queries) = K_tile . Q^T   : MFMA A operand = K rows (keys), B operand = Q (queries)
//   -> every lane holds 16 keys of ONE query column: the softmax statistics are per-lane plus one cross-half shuffle.
//   O^T tile (32 dims x 32 queries) += V^T . P^T     : A operand = V^T rows (dims), B operand = P straight from the
//   S^T accumulator registers (the key order inside an MFMA k-slot is arbitrary as long as A and B agree, so no
//   cross-lane permutation of P is needed; V^T is staged transposed so the matching keys are contiguous).
#include <cstdlib>

#include "common.hpp"

namespace ddx {

template <typename T> struct AttnMma;
template <> struct AttnMma<bf16> {
  static constexpr int KM = 16, VPAD = 4;
  using Frag = bf16x8;
};
template <> struct AttnMma<float> {
  static constexpr int KM = 2, VPAD = 1;
  using Frag = float;
};

// XCD-aware tile order.  Workgroup L of a 1-D grid runs on XCD L % 8, each with its own 4 MB L2; the query tiles of one (batch entry, head)
// read the same K / V rows.  With the plain (tile, head, batch) grid those tiles land on different XCDs, every XCD touches every head
// (8.4 MB of q | k | v at level 3: more than its L2) and K / V come out of the Infinity Cache once per TILE.  Here XCD x takes the
// (batch, head) pairs p = x (mod 8) and runs their nq tiles back to back (needs heads * B to be a multiple of 8).
__device__ __forceinline__ void xcd_tile(int nq, int heads, int& qtile, int& head, int& b) {
  const int L = blockIdx.x, slot = L >> 3;
  const int pair = (slot / nq) * 8 + (L & 7);
  qtile = slot % nq;
  head = pair % heads;
  b = pair / heads;
}

// DDX_ABLATE (timing A/B; results stay right): 32 the chunked kernel where the key-split one would run, 64 the plain (tile, head, batch) grid,
// 128 Q / K fragments straight from global memory in every key-split variant
static int ablate_bits() {
  static const int bits = std::getenv("DDX_ABLATE") ? std::atoi(std::getenv("DDX_ABLATE")) : 0;
  return bits;
}

// Sum over aligned groups of N consecutive lanes (N = 16-byte vectors per token row), every lane gets the total.  DPP moves inside a
// 16-lane row (quad permutes, row_half_mirror, row_mirror: pure VALU, a few cycles) instead of a chain of ds_bpermute round trips;
// the staging of a key chunk runs two such reductions per token row.
template <int N>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (N >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if constexpr (N >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if constexpr (N >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if constexpr (N >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
  if constexpr (N >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <typename T, int D, int KC_>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qk, const T* __restrict__ v, T* __restrict__ out,
                                                       const float* __restrict__ out_cs, int B, int Tn, int heads, float eps, int qk_ld, int v_ld,
                                                       int fold, int nq_xcd) {
  constexpr int EV = 16 / (int)sizeof(T);
  constexpr int VPR = D / EV;           // 16-byte vectors per token row
  constexpr int RPP = 256 / VPR;        // rows staged per pass
  constexpr int QS = D + EV;            // Q / K row stride (elements)
  constexpr int KC = KC_;               // keys per chunk (128; 64 keeps the D = 64 bf16 kernel under 256 registers)
  constexpr int VS = KC + AttnMma<T>::VPAD;
  constexpr int KM = AttnMma<T>::KM;
  constexpr int NDT = D / 32;           // 32-row tiles of O^T
  constexpr int NKT = KC / 32;
  using Frag = typename AttnMma<T>::Frag;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Q tile, then two {K chunk, V chunk} buffers: chunk c + 1 is staged into the other buffer while chunk c is multiplied (one barrier per chunk)
  constexpr int KVS = KC * QS + (sizeof(T) == 2 ? KC * QS : D * VS);   // elements of one {K, V} buffer
  T* sQ = reinterpret_cast<T*>(smem);
  T* sKV = sQ + 128 * QS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  int qtile = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if (nq_xcd) xcd_tile(nq_xcd, heads, qtile, head, b);
  const int q0 = qtile * 128;
  // token t of batch entry b lives in row rbase + t * fold of the [rows][channels] tensors.  fold = 1: the H*W tokens of image b
  // (rbase = b * T); fold = W > 1: AXIS-FOLDED attention -- the batch entries are the (image, column) pairs of [N][H][W] maps and
  // the tokens run along H (reference modules/daes/dae_edm2_g1.py:209-228: attention over h for every (b, z, w)).
  const int b_img = b / fold;
  const size_t rbase = (size_t)b_img * Tn * fold + (size_t)(b - b_img * fold);
  const int C = heads * D;
  const float inv_sqrt_d = rsqrtf((float)D);
  const bool pre = sizeof(T) == 2 && eps < 0.f;   // see ddx_attn_fold_fwd: negative eps = q, k, v already RMS-normalised per head (bf16)

  const int sv = tid % VPR;        // vector inside the row handled by this thread while staging
  const int sr = tid / VPR;

  // Global loads of a staging phase are ALL issued before the first of them is consumed (Q tile and the first K / V chunk
  // together, later chunks while the previous one is multiplied): the loops below used to expose one memory latency per
  // 32-row pass, i.e. 8 in a row for a 128-query tile with one key chunk (the L4 / L3 attention layers are latency-bound).
  constexpr int NP = 128 / RPP;          // staging passes of the 128-query tile
  constexpr int NPK = KC / RPP;          // ... of a key chunk
  static_assert(KC % RPP == 0 && NPK >= 1, "a key chunk is a whole number of staging passes");
  using VT = decltype(Vec16<T>::v);
  VT qreg[NP], kreg[NPK], vreg[NPK];
  auto issue_kv = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int key = c0 + sr + i * RPP;
      VT z = {};
      kreg[i] = z; vreg[i] = z;
      if (key < Tn) {
        kreg[i] = *reinterpret_cast<const VT*>(qk + (rbase + (size_t)key * fold) * qk_ld + head * 2 * D + D + sv * EV);
        vreg[i] = *reinterpret_cast<const VT*>(v + (rbase + (size_t)key * fold) * v_ld + head * D + sv * EV);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = q0 + sr + i * RPP;
    VT z = {};
    qreg[i] = z;
    if (q < Tn) qreg[i] = *reinterpret_cast<const VT*>(qk + (rbase + (size_t)q * fold) * qk_ld + head * 2 * D + sv * EV);
  }
  issue_kv(0);

  // ---- stage Q (normalised, pre-scaled by 1/sqrt(D)); rows past the sequence are zero
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = sr + i * RPP;
    Vec16<T> x;
    x.v = qreg[i];
    float f[EV];
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < EV; ++e) { f[e] = x.get(e); ss += f[e] * f[e]; }
    float sc = inv_sqrt_d;                       // (eps < 0: q, k, v arrive normalised from their producer's epilogue -- only the 1 / sqrt(D))
    if (!pre) {
      ss = group_sum<VPR>(ss);
      sc = inv_sqrt_d / (eps + sqrtf(ss) * inv_sqrt_d);
    }
    Vec16<T> y;
#pragma unroll
    for (int e = 0; e < EV; ++e) y.set(e, f[e] * sc);
    *reinterpret_cast<decltype(y.v)*>(sQ + r * QS + sv * EV) = y.v;
  }
  __syncthreads();

  // Q fragments of this wave's 32 queries stay in registers
  Frag qf[D / KM];
#pragma unroll
  for (int ks = 0; ks < D / KM; ++ks) {
    const T* rp = sQ + (wave * 32 + l31) * QS + ks * KM;
    if constexpr (sizeof(T) == 2) qf[ks] = *reinterpret_cast<const bf16x8*>(rp + khalf * 8);
    else qf[ks] = rp[khalf];
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- stage K (normalised) and V^T (normalised, transposed) of one chunk from the registers loaded ahead
  auto stage_kv = [&](T* sK, T* sVt) {
    if constexpr (sizeof(T) == 2) {
      if (pre) {     // normalised operands: the rows go to LDS as they came (no conversion, no reduction)
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
          const int r = sr + i * RPP;
          *reinterpret_cast<VT*>(sK + r * QS + sv * EV) = kreg[i];
          *reinterpret_cast<VT*>(sVt + r * QS + sv * EV) = vreg[i];
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int r = sr + i * RPP;
      float fk[EV], fv[EV];
      float ssk = 0.f, ssv = 0.f;
      Vec16<T> xk, xv;
      xk.v = kreg[i]; xv.v = vreg[i];
#pragma unroll
      for (int e = 0; e < EV; ++e) {
        fk[e] = xk.get(e); ssk += fk[e] * fk[e];
        fv[e] = xv.get(e); ssv += fv[e] * fv[e];
      }
      ssk = group_sum<VPR>(ssk);
      ssv = group_sum<VPR>(ssv);
      const float sck = 1.0f / (eps + sqrtf(ssk) * inv_sqrt_d);
      const float scv = 1.0f / (eps + sqrtf(ssv) * inv_sqrt_d);
      Vec16<T> yk;
#pragma unroll
      for (int e = 0; e < EV; ++e) yk.set(e, fk[e] * sck);
      *reinterpret_cast<decltype(yk.v)*>(sK + r * QS + sv * EV) = yk.v;
      if constexpr (sizeof(T) == 2) {
        // bf16: V stays row-major [key][dim] like K (one 16-byte write); the PV operand is read with the transpose read below
        Vec16<T> yv;
#pragma unroll
        for (int e = 0; e < EV; ++e) yv.set(e, fv[e] * scv);
        *reinterpret_cast<decltype(yv.v)*>(sVt + r * QS + sv * EV) = yv.v;
      } else {
#pragma unroll
        for (int e = 0; e < EV; ++e) sVt[(sv * EV + e) * VS + r] = from_f32<T>(fv[e] * scv);
      }
    }
  };
  stage_kv(sKV, sKV + KC * QS);
  if (KC < Tn) issue_kv(KC);
  __syncthreads();

  int cur = 0;
  for (int c0 = 0; c0 < Tn; c0 += KC, cur ^= 1) {
    const T* sK = sKV + cur * KVS;
    const T* sVt = sK + KC * QS;
    // the next chunk's rows go into the other buffer (every wave left it at the barrier that ended the previous iteration), and
    // the chunk after that starts travelling
    if (c0 + KC < Tn) {
      stage_kv(sKV + (cur ^ 1) * KVS, sKV + (cur ^ 1) * KVS + KC * QS);
      if (c0 + 2 * KC < Tn) issue_kv(c0 + 2 * KC);
    }

    // ---- S^T = K . Q^T for the NKT key tiles of the chunk
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      if (c0 + kt * 32 < Tn) {
#pragma unroll
        for (int ks = 0; ks < D / KM; ++ks) {
          const T* rp = sK + (kt * 32 + l31) * QS + ks * KM;
          if constexpr (sizeof(T) == 2)
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(rp + khalf * 8), qf[ks], s[kt], 0, 0, 0);
          else
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(rp[khalf], qf[ks], s[kt], 0, 0, 0);
        }
      }
    }
    // ---- online softmax (per lane = per query column; halves hold disjoint keys)
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = c0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (key >= Tn) s[kt][r] = -1e30f;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __expf(s[kt][r] - m_new);
        s[kt][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (c0 + kt * 32 >= Tn) continue;
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 pf;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) pf[kk] = (bf16)s[kt][8 * j + kk];
          // A operand = V^T: lane (dim l31, khalf) needs keys kt*32 + 16j + 4 khalf + {0..3, 8..11} of its dim column.  ds_read_b64_tr_b16
          // hands every lane 4 consecutive ROWS of its column of a [4 rows][16 columns] block addressed by its 16-lane group (lane i of
          // the group supplies row i >> 2, 4-element run i & 3): group gq = (khalf, dim half), two reads 8 keys apart.
          const int gq = lane >> 4, li = lane & 15;
          const int kb = kt * 32 + 16 * j + 4 * (gq >> 1) + (li >> 2);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const T* vp = sVt + kb * QS + dt * 32 + (gq & 1) * 16 + (li & 3) * 4;
            typedef __attribute__((ext_vector_type(4))) short s16x4_t;
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * QS));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const int kl = kt * 32 + (st & 3) + 8 * (st >> 2) + 4 * khalf;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sVt[(dt * 32 + l31) * VS + kl], s[kt][st], oacc[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // this chunk's buffer is free, the next chunk's rows are in place
  }

  // ---- finalise: O / l, lane owns query q, 4 consecutive dims per register group
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_tot;
  const int q = q0 + wave * 32 + l31;
  if (q < Tn) {
    T* orow = out + (rbase + (size_t)q * fold) * C + head * D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        Vec4<T> ov;
        const int dd = dt * 32 + 8 * g4 + 4 * khalf;
        if (out_cs) {  // producer-side activation of attn_proj's operand: mp_silu(o * c_v)
          const f32x4 c4v = *reinterpret_cast<const f32x4*>(out_cs + (size_t)b_img * C + head * D + dd);
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.set(e, mp_silu_f(oacc[dt][4 * g4 + e] * inv_l * c4v[e]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.set(e, oacc[dt][4 * g4 + e] * inv_l);
        }
        *reinterpret_cast<decltype(ov.v)*>(orow + dd) = ov.v;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Key-split kernel for the UNet's attention layers (bf16, head_dim 64, operands normalised by the qkv conv's epilogue, at most 384 tokens).
//
// The kernel above walks the keys in chunks that its four waves stage together: at 344 tokens six chunks, each behind a barrier, on a
// workgroup whose useful work is a few hundred MFMA cycles -- the layer is a chain of latencies (8 us + 2.4 us per chunk).  Here the four
// waves of a workgroup split the KEYS: wave w owns keys [w * kpw, (w + 1) * kpw) (kpw = ceil(T / 4) <= 96) for ALL the queries of the
// workgroup's tile (NQT x 32).  Nothing is shared on the way in: the K and Q fragments are 16 contiguous bytes of a token row per lane, so
// they are loaded from global memory straight into the MFMA operand registers; V needs the key-major -> dim-major transpose and goes
// through a wave-private LDS region (ds_read_b64_tr_b16), written and read by the same wave, no barrier.  Every global load of the
// workgroup is issued before the first is consumed.  Each wave ends with an (m, l, O^T) partial of its key range; the partials meet in LDS
// behind the kernel's ONE barrier (the V region is reused: a wave has finished its P.V before it writes), and wave w' then combines
// dims [16 w', 16 w' + 16) of every query (flash-decoding's combine) and stores 32 contiguous bytes per lane.
template <int NKT, int NQT, bool STAGE>
__global__ __launch_bounds__(256) void attn_ks_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ v, bf16* __restrict__ out,
                                                      const float* __restrict__ out_cs, int Tn, int heads, int qk_ld, int v_ld, int fold,
                                                      int nq_xcd) {
  constexpr int D = 64, QS = D + 8;
  constexpr int VREG = NKT * 32 * QS * 2;                     // bytes of a wave's V rows
  constexpr int OREG = NQT * D * 32 * 4 + NQT * 32 * 8;       // ... of its (O^T, m, l) partial
  constexpr int REG = VREG > OREG ? VREG : OREG;
  __shared__ __attribute__((aligned(16))) char smem[4 * REG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  int qtile = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if (nq_xcd) xcd_tile(nq_xcd, heads, qtile, head, b);
  const int q0 = qtile * (NQT * 32);
  const int b_img = b / fold;
  const size_t rbase = (size_t)b_img * Tn * fold + (size_t)(b - b_img * fold);
  const int C = heads * D;
  const int kpw = (Tn + 3) >> 2;
  const int k0 = wave * kpw;
  const int kend = min(Tn, k0 + kpw);          // keys [k0, kend) are this wave's (possibly none)
  const int last = Tn - 1;

  // ---- every global load of the wave, all issued before the first is used (row index clamped, the masks come later): V rows, the combine's
  // channel scales and Q / K -- either as MFMA fragments straight from global memory (lane = row, 16 bytes each: an instruction touches 32 cache
  // lines; cheapest for one key tile) or (STAGE) as whole rows, 8 per instruction like V, that the wave turns into fragments through its own
  // LDS region before V takes it over: the wave's LDS instructions execute in order, so write -> read -> overwrite needs no barrier.
  bf16x8 vreg[NKT * 4], qf[NQT][4], kf[NKT][4];
  bf16x8 qreg[STAGE ? NQT * 4 : 1], kreg[STAGE ? NKT * 4 : 1];
#pragma unroll
  for (int i = 0; i < NKT * 4; ++i) {
    const int key = min(k0 + i * 8 + (lane >> 3), last);
    vreg[i] = *reinterpret_cast<const bf16x8*>(v + (rbase + (size_t)key * fold) * v_ld + head * D + (lane & 7) * 8);
  }
  if constexpr (STAGE) {
#pragma unroll
    for (int i = 0; i < NQT * 4; ++i)
      qreg[i] = *reinterpret_cast<const bf16x8*>(qk + (rbase + (size_t)min(q0 + i * 8 + (lane >> 3), last) * fold) * qk_ld + head * 2 * D + (lane & 7) * 8);
#pragma unroll
    for (int i = 0; i < NKT * 4; ++i)
      kreg[i] = *reinterpret_cast<const bf16x8*>(qk + (rbase + (size_t)min(k0 + i * 8 + (lane >> 3), last) * fold) * qk_ld + head * 2 * D + D + (lane & 7) * 8);
  } else {
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      const bf16* rp = qk + (rbase + (size_t)min(q0 + qt * 32 + l31, last) * fold) * qk_ld + head * 2 * D + khalf * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(rp + ks * 16);
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const bf16* rp = qk + (rbase + (size_t)min(k0 + kt * 32 + l31, last) * fold) * qk_ld + head * 2 * D + D + khalf * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *reinterpret_cast<const bf16x8*>(rp + ks * 16);
    }
  }
  // (the combine's channel scales travel with them: loaded where they are used they cost sixteen exposed round trips; without
  // out_scale the loads read the q row instead -- unconditional, so that nothing waits on them here -- and are ignored)
  f32x4 c4[4];
  {
    const float* csp = out_cs ? out_cs + (size_t)b_img * C + head * D + wave * 16
                              : reinterpret_cast<const float*>(qk + (rbase + (size_t)min(q0, last) * fold) * qk_ld + head * 2 * D);
#pragma unroll
    for (int e = 0; e < 4; ++e) c4[e] = *reinterpret_cast<const f32x4*>(csp + e * 4);
  }
  bf16* sV = reinterpret_cast<bf16*>(smem + wave * REG);
  if constexpr (STAGE) {
    static_assert(NQT <= NKT, "the query rows pass through the region sized for the key rows");
    bf16* wr = sV + (lane >> 3) * QS + (lane & 7) * 8;
    const bf16* rd = sV + l31 * QS + khalf * 8;
#pragma unroll
    for (int i = 0; i < NQT * 4; ++i) *reinterpret_cast<bf16x8*>(wr + i * 8 * QS) = qreg[i];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(rd + qt * 32 * QS + ks * 16);
#pragma unroll
    for (int i = 0; i < NKT * 4; ++i) *reinterpret_cast<bf16x8*>(wr + i * 8 * QS) = kreg[i];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *reinterpret_cast<const bf16x8*>(rd + kt * 32 * QS + ks * 16);
  }
#pragma unroll
  for (int i = 0; i < NKT * 4; ++i) *reinterpret_cast<bf16x8*>(sV + (i * 8 + (lane >> 3)) * QS + (lane & 7) * 8) = vreg[i];

  // ---- per query tile: S^T = K . Q^T, softmax over the wave's keys (base 2, the 1 / sqrt(D) inside the exponent), P as bf16 fragments
  constexpr float kSc = 0.125f * 1.4426950408889634f;
  bf16x8 pf[NQT][NKT][2];
  float m_w[NQT], l_w[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[qt][ks], s[kt], 0, 0, 0);
    }
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (key >= kend) s[kt][r] = -1e30f;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float pv = exp2f((s[kt][8 * j + kk] - mx) * kSc);
          psum += pv;
          pf[qt][kt][j][kk] = (bf16)pv;
        }
    psum += __shfl_xor(psum, 32, 64);
    m_w[qt] = mx;
    l_w[qt] = k0 < kend ? psum : 0.f;         // (a wave without keys: its exponentials are exp2(0) -- weight 0 in the combine)
  }

  // ---- O^T = V^T . P^T over the wave's keys (the transpose read of the kernel above, on the wave's own rows)
  f32x16 oacc[NQT][2];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qt][dt][r] = 0.f;
  {
    const int gq = lane >> 4, li = lane & 15;
    typedef __attribute__((ext_vector_type(4))) short s16x4_t;
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kb = kt * 32 + 16 * j + 4 * (gq >> 1) + (li >> 2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16* vp = sV + kb * QS + dt * 32 + (gq & 1) * 16 + (li & 3) * 4;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * QS));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) oacc[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][kt][j], oacc[qt][dt], 0, 0, 0);
        }
      }
  }

  // ---- the wave's partial into its region: O^T as [qt][dim][query] fp32 (consecutive lanes = consecutive queries), then (m, l) per query
  float* sO = reinterpret_cast<float*>(smem + wave * REG);
  float* sML = sO + NQT * D * 32;
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sO[(qt * D + dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + l31] = oacc[qt][dt][r];
    if (khalf == 0) {
      sML[qt * 64 + l31] = m_w[qt];
      sML[qt * 64 + 32 + l31] = l_w[qt];
    }
  }
  __syncthreads();

  // ---- combine: wave w' takes dims [16 w', 16 w' + 16) of query (lane) of tile qt (lanes beyond NQT * 32 queries idle)
  const int cq = lane & 31, cqt = lane >> 5;
  if (cqt < NQT) {
    float mw[4], lw[4], m = -1e30f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* ml = reinterpret_cast<const float*>(smem + w * REG) + NQT * D * 32 + cqt * 64;
      mw[w] = ml[cq]; lw[w] = ml[32 + cq];
      if (lw[w] > 0.f) m = fmaxf(m, mw[w]);
    }
    float lt = 0.f, aw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      aw[w] = lw[w] > 0.f ? exp2f((mw[w] - m) * kSc) : 0.f;
      lt += lw[w] * aw[w];
    }
    const float inv_l = 1.0f / lt;
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* po = reinterpret_cast<const float*>(smem + w * REG) + (cqt * D + wave * 16) * 32 + cq;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] += po[e * 32] * aw[w];
    }
    const int q = q0 + cqt * 32 + cq;
    if (q < Tn) {
      bf16* orow = out + (rbase + (size_t)q * fold) * C + head * D + wave * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float val = o[h * 8 + e] * inv_l;
          ov[e] = (bf16)(out_cs ? mp_silu_f(val * c4[h * 2 + (e >> 2)][e & 3]) : val);
        }
        *reinterpret_cast<bf16x8*>(orow + h * 8) = ov;
      }
    }
  }
}

template <int NKT, int NQT, bool STAGE>
static int launch_attn_ks(const void* qk, const void* v, void* out, const float* cs, int B, int Tn, int heads, hipStream_t s, int qk_ld, int v_ld,
                          int fold) {
  const int nq = (Tn + NQT * 32 - 1) / (NQT * 32);
  const bool xcd = (heads * (long)B) % 8 == 0 && !(ablate_bits() & 64);
  dim3 grid = xcd ? dim3((unsigned)(nq * heads * B)) : dim3(nq, heads, B);
  hipLaunchKernelGGL((attn_ks_kernel<NKT, NQT, STAGE>), grid, dim3(256), 0, s, (const bf16*)qk, (const bf16*)v, (bf16*)out, cs, Tn, heads, qk_ld, v_ld, fold,
                     xcd ? nq : 0);
  return check_launch("attn_ks");
}

// bf16, head_dim 64, pre-normalised operands, at most 4 x 96 keys: the key-split kernel (64-query tiles where that still fills the part).
// Measured (B = 4, 16 / 20 heads, graph of back-to-back launches, chunked -> key-split): 86 tokens 7.7 -> 4.8 us, 128: 8.3 -> 4.9, 256: 11.4 ->
// 9.5, 344: 14.9 -> 14.5, 384: 15.3 -> 15.1; B = 8 at 344 tokens 20.9 -> 21.9 (it moves 128 KB per 64 queries through the CU's load path where
// the chunked kernel moves 104 KB per 128: the K / Q fragment loads touch 32 lines per instruction, 3.1 of the 14.5 us) -- three key tiles
// per wave on more than 512 workgroups stay on the chunked kernel.  In the B = 4 step: attention family 0.247 -> 0.199 ms, step 4.488 -> 4.436.
static bool attn_ks_applies(int B, int Tn, int heads, int head_dim, float eps, int dtype) {
  if (!(dtype == DDX_BF16 && head_dim == 64 && eps < 0.f && Tn >= 4 && (Tn + 3) / 4 <= 96)) return false;
  return (Tn + 3) / 4 <= 64 || (long)((Tn + 63) / 64) * heads * B <= 512;
}

static int launch_attn_keysplit(const void* qk, const void* v, void* out, const float* cs, int B, int Tn, int heads, hipStream_t s, int qk_ld,
                                int v_ld, int fold) {
  const int nkt = ((Tn + 3) / 4 + 31) / 32;
  const bool two = (long)((Tn + 63) / 64) * heads * B >= 320;
  const bool stage = !(ablate_bits() & 128);     // (two or three key tiles per wave: Q / K as whole rows through the LDS region)
  if (nkt == 1) return two ? launch_attn_ks<1, 2, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold) : launch_attn_ks<1, 1, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold);
  if (nkt == 2) {
    if (stage) return two ? launch_attn_ks<2, 2, true>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold) : launch_attn_ks<2, 1, true>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold);
    return two ? launch_attn_ks<2, 2, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold) : launch_attn_ks<2, 1, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold);
  }
  if (stage) return two ? launch_attn_ks<3, 2, true>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold) : launch_attn_ks<3, 1, true>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold);
  return two ? launch_attn_ks<3, 2, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold) : launch_attn_ks<3, 1, false>(qk, v, out, cs, B, Tn, heads, s, qk_ld, v_ld, fold);
}

template <typename T, int D, int KC>
static int launch_attn_kc(const void* qk, const void* v, void* out, const float* cs, int B, int Tn, int heads, float eps, hipStream_t s, int qk_ld,
                          int v_ld, int fold) {
  constexpr int EV = 16 / (int)sizeof(T);
  // Q tile + K chunk + V chunk (bf16: row-major [KC][D + EV] like K, read transposed; fp32: transposed [D][KC + pad])
  const size_t smem = ((size_t)128 * (D + EV) + 2 * ((size_t)KC * (D + EV) + (sizeof(T) == 2 ? (size_t)KC * (D + EV) : (size_t)D * (KC + AttnMma<T>::VPAD)))) * sizeof(T);
  if (smem > 160 * 1024) return set_error(DDX_ERR_UNSUPPORTED, "attn: head_dim too large for this dtype");
  auto kern = attn_fwd_kernel<T, D, KC>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(attn)");
    attr_done = true;
  }
  const int nq = (Tn + 127) / 128;
  const bool xcd = nq > 1 && (heads * (long)B) % 8 == 0 && (long)nq * heads * B <= 0x7fffffff && !(ablate_bits() & 64);
  dim3 grid = xcd ? dim3((unsigned)(nq * heads * B)) : dim3(nq, heads, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const T*)qk, (const T*)v, (T*)out, cs, B, Tn, heads, eps, qk_ld, v_ld, fold, xcd ? nq : 0);
  return check_launch("attn_fwd");
}

template <typename T, int D>
static int launch_attn(const void* qk, const void* v, void* out, const float* cs, int B, int Tn, int heads, float eps, hipStream_t s, int qk_ld,
                       int v_ld, int fold = 1) {
  // keys per chunk: what two {K, V} chunk buffers + the Q tile leave room for in 160 KB of LDS (bf16 head_dim 64, the UNet: 64 -- 172
  // registers, two workgroups per CU; 128-key chunks measured 19.1 vs 18.0 us at 344 tokens, 37 vs 26 us at B = 8)
  if constexpr (sizeof(T) == 2) {
    if constexpr (D == 32) return launch_attn_kc<T, D, 128>(qk, v, out, cs, B, Tn, heads, eps, s, qk_ld, v_ld, fold);
    return launch_attn_kc<T, D, 64>(qk, v, out, cs, B, Tn, heads, eps, s, qk_ld, v_ld, fold);
  } else {
    if constexpr (D == 32) return launch_attn_kc<T, D, 128>(qk, v, out, cs, B, Tn, heads, eps, s, qk_ld, v_ld, fold);
    if constexpr (D == 64) return launch_attn_kc<T, D, 64>(qk, v, out, cs, B, Tn, heads, eps, s, qk_ld, v_ld, fold);
    return launch_attn_kc<T, D, 32>(qk, v, out, cs, B, Tn, heads, eps, s, qk_ld, v_ld, fold);
  }
}

}  // namespace ddx

using namespace ddx;

extern "C" int ddx_attn_fwd(const void* qk, const void* v, void* out, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                            float eps, int32_t dtype, ddx_stream stream) {
  return ddx_attn_act_fwd(qk, v, out, nullptr, B, T, heads, head_dim, eps, dtype, stream);
}

extern "C" int ddx_attn_act_fwd(const void* qk, const void* v, void* out, const float* out_scale, int32_t B, int32_t T, int32_t heads,
                                int32_t head_dim, float eps, int32_t dtype, ddx_stream stream) {
  return ddx_attn_act_fwd_ld(qk, 2 * heads * head_dim, v, heads * head_dim, out, out_scale, B, T, heads, head_dim, eps, dtype, stream);
}

extern "C" int ddx_attn_act_fwd_ld(const void* qk, int32_t qk_ld, const void* v, int32_t v_ld, void* out, const float* out_scale, int32_t B,
                                   int32_t T, int32_t heads, int32_t head_dim, float eps, int32_t dtype, ddx_stream stream) {
  return ddx_attn_fold_fwd(qk, qk_ld, v, v_ld, out, out_scale, B, T, 1, heads, head_dim, eps, dtype, stream);
}

extern "C" int ddx_attn_fold_fwd(const void* qk, int32_t qk_ld, const void* v, int32_t v_ld, void* out, const float* out_scale, int32_t N,
                                 int32_t T, int32_t fold, int32_t heads, int32_t head_dim, float eps, int32_t dtype, ddx_stream stream) {
  if (fold <= 0 || N <= 0 || (int64_t)N * fold > 0x7fffffff) return set_error(DDX_ERR_ARG, "attn: bad fold");
  const int32_t B = N * fold;
  if (B > 65535) return set_error(DDX_ERR_UNSUPPORTED, "attn: more than 65535 batch entries (gridDim.z)");
  if (!qk || !v || !out || B <= 0 || T <= 0 || heads <= 0) return set_error(DDX_ERR_ARG, "attn: bad args");
  if (head_dim != 32 && head_dim != 64 && head_dim != 128) return set_error(DDX_ERR_UNSUPPORTED, "attn: head_dim must be 32, 64 or 128");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (qk_ld < 2 * heads * head_dim || v_ld < heads * head_dim || qk_ld % ev || v_ld % ev) return set_error(DDX_ERR_ARG, "attn: bad row strides");
  return dispatch([=](hipStream_t s) -> int {
    if (attn_ks_applies(B, T, heads, head_dim, eps, dtype) && !(ablate_bits() & 32))
      return launch_attn_keysplit(qk, v, out, out_scale, B, T, heads, s, qk_ld, v_ld, fold);
    if (dtype == DDX_BF16) {
      if (head_dim == 32) return launch_attn<bf16, 32>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
      if (head_dim == 64) return launch_attn<bf16, 64>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
      return launch_attn<bf16, 128>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
    }
    if (head_dim == 32) return launch_attn<float, 32>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
    if (head_dim == 64) return launch_attn<float, 64>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
    return launch_attn<float, 128>(qk, v, out, out_scale, B, T, heads, eps, s, qk_ld, v_ld, fold);
  }, stream, "attention", 4.0 * B * heads * (double)T * T * head_dim,
     (double)dtype_size(dtype) * 4.0 * B * T * heads * head_dim);
}
