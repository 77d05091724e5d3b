// Row-level device functions of the weight path (reference src/modules/mp_tools.py:359-364 forward, :375-378 forced
// normalisation) shared by the single-tensor kernels (conv.hip, backward.hip) and the multi-tensor launches (wpath.hip).
// Every function is executed by ONE 256-thread workgroup for one row; `scratch` = 4 floats of LDS.  The `_w` variants below are
// executed by ONE WAVE per row (multi-tensor launches: rows are 1-15 KB, a workgroup with two block reductions per row spends its time
// in set-up and barriers), with 16-byte accesses where the row length allows it; same arithmetic, wave-strided summation order.
#pragma once
#include "conv_params.hpp"

namespace ddx {

// destination row od of the prepared weight -> source row of the master weight ((head, s, d) <- (head, d, s) for attn_qk)
__device__ __forceinline__ int wpath_src_row(int od, int qk_d) {
  if (qk_d <= 0) return od;
  const int head = od / (2 * qk_d), rem = od - head * 2 * qk_d;
  const int s = rem / qk_d, dd = rem - s * qk_d;
  return head * 2 * qk_d + dd * 2 + s;
}

template <typename TW_, typename TP>
__device__ __forceinline__ void wprep_row(const TW_* __restrict__ w, TP* __restrict__ wp, const float* gain_ptr, float gain, int Cout, int Cg,
                                          int taps, int G, int CK, int normalize, int qk_d, float eps, int in_split, float in_s0, float in_s1,
                                          int od, float* scratch, int row_off = 0, int rows_total = 0, float* row_scale = nullptr) {
  // row_off / rows_total: this weight fills rows [row_off, row_off + Cout) of a wider prepared matrix (merged 1x1 convs, G = 1)
  const int Ng = Cout / G, NgP = ((rows_total > 0 ? rows_total : Ng) + 31) / 32 * 32, nchunk = (Cg + CK - 1) / CK;
  const int g = od / Ng, n = od - g * Ng + row_off;
  const int os = wpath_src_row(od, qk_d);
  const int fan = Cg * taps;
  const TW_* wr = w + (size_t)os * fan;
  float inv = 1.f;
  if (normalize) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < fan; i += 256) { const float x = to_f32<TW_>(wr[i]); ss += x * x; }
    ss = block_sum_256(ss, scratch);
    inv = eps + sqrtf(ss) * sqrtf(1.0f / (float)fan);
  }
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  const float sc = gn / sqrtf((float)fan);
  // the same factor wprep_rowscale_row computes (source-row indexed): the multi-tensor path takes it from here instead of a
  // second pass over the weights
  if (row_scale && threadIdx.x == 0) row_scale[os] = sc / inv;
  for (int i = threadIdx.x; i < fan; i += 256) {
    const int c = i / taps, tap = i - c * taps;
    float x = to_f32<TW_>(wr[i]);
    if (normalize) x = x / inv;
    float scc = sc;
    if (in_split > 0) scc *= (g * Cg + c < in_split) ? in_s0 : in_s1;  // mp_cat scales folded into a linear consumer
    wp[wp_index(g, n, tap, c, nchunk, taps, NgP, CK)] = from_f32<TP>(x * scc);
  }
}

// row_scale[o] = gain_eff / sqrt(fan) / (normalize ? eps + |w_o| / sqrt(fan) : 1)
template <typename TW_>
__device__ __forceinline__ void wprep_rowscale_row(const TW_* __restrict__ w, float* __restrict__ row_scale, const float* gain_ptr, float gain,
                                                   int fan, int normalize, float eps, int row, float* scratch) {
  const TW_* wr = w + (size_t)row * fan;
  float inv = 1.f;
  if (normalize) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < fan; i += 256) { const float x = to_f32<TW_>(wr[i]); ss += x * x; }
    ss = block_sum_256(ss, scratch);
    inv = eps + sqrtf(ss) * sqrtf(1.0f / (float)fan);
  }
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  if (threadIdx.x == 0) row_scale[row] = gn / sqrtf((float)fan) / inv;
}

// data-gradient (transposed) preparation, destination row ci = input channel of the forward conv
template <typename TW_, typename TP>
__device__ __forceinline__ void wprep_transposed_row(const TW_* __restrict__ w, TP* __restrict__ wp, const float* __restrict__ row_scale, int Cout,
                                                     int Cg, int taps, int G, int CK, int qk_d, int in_split, float in_s0, float in_s1, int ci) {
  const int Ng = Cout / G, CgP = (Cg + 31) / 32 * 32, nchunk = (Ng + CK - 1) / CK;
  const int g = ci / Cg, c = ci - g * Cg;
  const float cscale = in_split > 0 ? (ci < in_split ? in_s0 : in_s1) : 1.0f;
  for (int i = threadIdx.x; i < Ng * taps; i += 256) {
    const int n = i / taps, tap = i - n * taps;
    const int os = wpath_src_row(g * Ng + n, qk_d);
    const float x = to_f32<TW_>(w[((size_t)os * Cg + c) * taps + tap]) * row_scale[os] * cscale;
    wp[wp_index(g, c, taps - 1 - tap, n, nchunk, taps, CgP, CK)] = from_f32<TP>(x);
  }
}

template <typename TW_>
__device__ __forceinline__ void normalize_row(TW_* w, int64_t fan, float eps, int64_t row, float* scratch) {
  TW_* wr = w + (size_t)row * fan;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < fan; i += 256) { const float x = to_f32<TW_>(wr[i]); ss += x * x; }
  ss = block_sum_256(ss, scratch);
  const float nrm = eps + sqrtf(ss) * sqrtf(1.0f / (float)fan);
  for (int64_t i = threadIdx.x; i < fan; i += 256) wr[i] = from_f32<TW_>(to_f32<TW_>(wr[i]) / nrm);
}

// gradient w.r.t. the master weight row from the gradient w.r.t. the prepared weight (natural layout), destination row od
template <typename TW_>
__device__ __forceinline__ void wprep_bwd_row(const float* __restrict__ dwp, const TW_* __restrict__ w, const float* gain_ptr, float gain,
                                              float* __restrict__ dw, float* __restrict__ dgain, int Cout, int Cg, int taps, int G, int normalize,
                                              int qk_d, float eps, int in_split, float in_s0, float in_s1, int accumulate, int od, float* scratch) {
  const int Ng = Cout / G;
  const int g = od / Ng;
  const int os = wpath_src_row(od, qk_d);
  const int fan = Cg * taps;
  const TW_* wr = w + (size_t)os * fan;
  const float* gr = dwp + (size_t)od * fan;
  float ss = 0.f, su = 0.f;
  for (int i = threadIdx.x; i < fan; i += 256) {
    const float x = to_f32<TW_>(wr[i]);
    const int c = i / taps;
    const float sc = in_split > 0 ? ((g * Cg + c < in_split) ? in_s0 : in_s1) : 1.0f;
    ss += x * x;
    su += gr[i] * sc * x;
  }
  ss = block_sum_256(ss, scratch);
  su = block_sum_256(su, scratch);
  const float rfan = sqrtf(1.0f / (float)fan);
  const float n = sqrtf(ss);
  const float nu = normalize ? eps + n * rfan : 1.0f;
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  const float s = gn * rfan;
  const float k = (normalize && n > 0.f) ? su * rfan / (nu * n) : 0.f;
  for (int i = threadIdx.x; i < fan; i += 256) {
    const float x = to_f32<TW_>(wr[i]);
    const int c = i / taps;
    const float sc = in_split > 0 ? ((g * Cg + c < in_split) ? in_s0 : in_s1) : 1.0f;
    const float v = (s / nu) * (gr[i] * sc - x * k);
    float* dst = dw + (size_t)os * fan + i;
    *dst = accumulate ? *dst + v : v;
  }
  // d(loss)/d(gain parameter): g_eff = gain * (*gain_ptr)
  if (dgain && threadIdx.x == 0) atomicAdd(dgain, gain * su * rfan / nu);
}

// ------------------------------------------------------------------------------------------------ one wave per row (fp32 master weights)
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

__device__ __forceinline__ float wave_row_sumsq(const float* __restrict__ wr, int fan, int lane) {
  float ss = 0.f;
  if ((fan & 3) == 0 && aligned16(wr)) {
    const f32x4* p = reinterpret_cast<const f32x4*>(wr);
    for (int i = lane; i < (fan >> 2); i += 64) { const f32x4 v = p[i]; ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
  } else {
    for (int i = lane; i < fan; i += 64) { const float x = wr[i]; ss += x * x; }
  }
  return wave_sum(ss);
}

__device__ __forceinline__ void normalize_row_w(float* w, int64_t fan, float eps, int64_t row, int lane) {
  float* wr = w + (size_t)row * fan;
  const float ss = wave_row_sumsq(wr, (int)fan, lane);
  const float nrm = eps + sqrtf(ss) * sqrtf(1.0f / (float)fan);
  if ((fan & 3) == 0 && aligned16(wr)) {
    f32x4* p = reinterpret_cast<f32x4*>(wr);
    for (int i = lane; i < (int)(fan >> 2); i += 64) { f32x4 v = p[i]; v[0] /= nrm; v[1] /= nrm; v[2] /= nrm; v[3] /= nrm; p[i] = v; }
  } else {
    for (int i = lane; i < (int)fan; i += 64) wr[i] = wr[i] / nrm;
  }
}

template <typename TP>
__device__ __forceinline__ void wprep_row_w(const float* __restrict__ w, TP* __restrict__ wp, const float* gain_ptr, float gain, int Cout, int Cg,
                                            int taps, int G, int CK, int normalize, int qk_d, float eps, int in_split, float in_s0, float in_s1,
                                            int od, int lane, float* row_scale) {
  const int Ng = Cout / G, NgP = (Ng + 31) / 32 * 32, nchunk = (Cg + CK - 1) / CK;
  const int g = od / Ng, n = od - g * Ng;
  const int os = wpath_src_row(od, qk_d);
  const int fan = Cg * taps;
  const float* wr = w + (size_t)os * fan;
  float inv = 1.f;
  if (normalize) inv = eps + sqrtf(wave_row_sumsq(wr, fan, lane)) * sqrtf(1.0f / (float)fan);
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  const float sc = gn / sqrtf((float)fan);
  if (row_scale && lane == 0) row_scale[os] = sc / inv;
  auto put = [&](int i, float x) {
    const int c = i / taps, tap = i - c * taps;
    if (normalize) x = x / inv;
    float scc = sc;
    if (in_split > 0) scc *= (g * Cg + c < in_split) ? in_s0 : in_s1;
    wp[wp_index(g, n, tap, c, nchunk, taps, NgP, CK)] = from_f32<TP>(x * scc);
  };
  if ((fan & 3) == 0 && aligned16(wr)) {
    const f32x4* p = reinterpret_cast<const f32x4*>(wr);
    for (int i = lane; i < (fan >> 2); i += 64) {
      const f32x4 v = p[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) put(4 * i + e, v[e]);
    }
  } else {
    for (int i = lane; i < fan; i += 64) put(i, wr[i]);
  }
}

__device__ __forceinline__ void wprep_rowscale_row_w(const float* __restrict__ w, float* __restrict__ row_scale, const float* gain_ptr, float gain,
                                                     int fan, int normalize, float eps, int row, int lane) {
  float inv = 1.f;
  if (normalize) inv = eps + sqrtf(wave_row_sumsq(w + (size_t)row * fan, fan, lane)) * sqrtf(1.0f / (float)fan);
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  if (lane == 0) row_scale[row] = gn / sqrtf((float)fan) / inv;
}

template <typename TP>
__device__ __forceinline__ void wprep_transposed_row_w(const float* __restrict__ w, TP* __restrict__ wp, const float* __restrict__ row_scale, int Cout,
                                                       int Cg, int taps, int G, int CK, int qk_d, int in_split, float in_s0, float in_s1, int ci, int lane) {
  const int Ng = Cout / G, CgP = (Cg + 31) / 32 * 32, nchunk = (Ng + CK - 1) / CK;
  const int g = ci / Cg, c = ci - g * Cg;
  const float cscale = in_split > 0 ? (ci < in_split ? in_s0 : in_s1) : 1.0f;
  for (int i = lane; i < Ng * taps; i += 64) {
    const int n = i / taps, tap = i - n * taps;
    const int os = wpath_src_row(g * Ng + n, qk_d);
    const float x = w[((size_t)os * Cg + c) * taps + tap] * row_scale[os] * cscale;
    wp[wp_index(g, c, taps - 1 - tap, n, nchunk, taps, CgP, CK)] = from_f32<TP>(x);
  }
}

// The same with the prepared-weight gradient given as `parts` split-K slices [parts][Cout][fan] (ddx_wpath_job.dwp_parts): the slices are
// added in slice order while the row is read (once: the summed row waits in `rowbuf`, fan <= kWpathRowBuf floats of LDS per wave, for the
// second pass; longer rows read the slices twice).
constexpr int kWpathRowBuf = 3072;
__device__ __forceinline__ void wprep_bwd_row_parts_w(const float* __restrict__ dwp, int parts, const float* __restrict__ w, const float* gain_ptr,
                                                      float gain, float* __restrict__ dw, float* __restrict__ dgain, int Cout, int Cg, int taps, int G,
                                                      int normalize, int qk_d, float eps, int in_split, float in_s0, float in_s1, int od, int lane,
                                                      float* rowbuf) {
  const int Ng = Cout / G;
  const int g = od / Ng;
  const int os = wpath_src_row(od, qk_d);
  const int fan = Cg * taps;
  const size_t pstride = (size_t)Cout * fan;
  const float* wr = w + (size_t)os * fan;
  const float* gr = dwp + (size_t)od * fan;
  auto cat_scale = [&](int i) { return in_split > 0 ? ((g * Cg + i / taps < in_split) ? in_s0 : in_s1) : 1.0f; };
  float* dr = dw + (size_t)os * fan;
  const bool keep = fan <= kWpathRowBuf;
  auto grad_at = [&](int i) {
    float s = gr[i];
    for (int k = 1; k < parts; ++k) s += gr[(size_t)k * pstride + i];
    return s;
  };
  float ss = 0.f, su = 0.f;
  if ((fan & 3) == 0 && aligned16(wr) && aligned16(gr) && aligned16(dr) && (pstride & 3) == 0) {
    const f32x4* pw = reinterpret_cast<const f32x4*>(wr);
    const f32x4* pg = reinterpret_cast<const f32x4*>(gr);
    const size_t ps4 = pstride >> 2;
    // (eight slices in flight per lane: the slices of a short row -- 288 floats at level 0, up to 128 slices -- are otherwise one dependent
    // round trip each; four running sums, combined in a fixed order: deterministic)
    auto grad4 = [&](int i) {
      f32x4 s0 = pg[i], s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
      int k = 1;
      for (; k + 7 < parts; k += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pg[(size_t)(k + u) * ps4 + i];
        s0 += t[0]; s1 += t[1]; s2 += t[2]; s3 += t[3];
        s0 += t[4]; s1 += t[5]; s2 += t[6]; s3 += t[7];
      }
      for (; k < parts; ++k) s0 += pg[(size_t)k * ps4 + i];
      return (s0 + s1) + (s2 + s3);
    };
    for (int i = lane; i < (fan >> 2); i += 64) {
      const f32x4 x = pw[i], gv = grad4(i);
      if (keep) reinterpret_cast<f32x4*>(rowbuf)[i] = gv;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss += x[e] * x[e]; su += gv[e] * cat_scale(4 * i + e) * x[e]; }
    }
    ss = wave_sum(ss);
    su = wave_sum(su);
    const float rfan = sqrtf(1.0f / (float)fan);
    const float n = sqrtf(ss);
    const float nu = normalize ? eps + n * rfan : 1.0f;
    float gn = gain;
    if (gain_ptr) gn *= *gain_ptr;
    const float s = gn * rfan;
    const float k = (normalize && n > 0.f) ? su * rfan / (nu * n) : 0.f;
    f32x4* pd = reinterpret_cast<f32x4*>(dr);
    for (int i = lane; i < (fan >> 2); i += 64) {
      const f32x4 x = pw[i], gv = keep ? reinterpret_cast<const f32x4*>(rowbuf)[i] : grad4(i);   // (a lane re-reads what it wrote)
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (s / nu) * (gv[e] * cat_scale(4 * i + e) - x[e] * k);
      pd[i] = v;
    }
    if (dgain && lane == 0) atomicAdd(dgain, gain * su * rfan / nu);
    return;
  }
  for (int i = lane; i < fan; i += 64) {
    const float x = wr[i], gv = grad_at(i);
    if (keep) rowbuf[i] = gv;
    ss += x * x; su += gv * cat_scale(i) * x;
  }
  ss = wave_sum(ss);
  su = wave_sum(su);
  const float rfan = sqrtf(1.0f / (float)fan);
  const float n = sqrtf(ss);
  const float nu = normalize ? eps + n * rfan : 1.0f;
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  const float s = gn * rfan;
  const float k = (normalize && n > 0.f) ? su * rfan / (nu * n) : 0.f;
  for (int i = lane; i < fan; i += 64) dr[i] = (s / nu) * ((keep ? rowbuf[i] : grad_at(i)) * cat_scale(i) - wr[i] * k);
  if (dgain && lane == 0) atomicAdd(dgain, gain * su * rfan / nu);
}

__device__ __forceinline__ void wprep_bwd_row_w(const float* __restrict__ dwp, const float* __restrict__ w, const float* gain_ptr, float gain,
                                                float* __restrict__ dw, float* __restrict__ dgain, int Cout, int Cg, int taps, int G, int normalize,
                                                int qk_d, float eps, int in_split, float in_s0, float in_s1, int od, int lane) {
  const int Ng = Cout / G;
  const int g = od / Ng;
  const int os = wpath_src_row(od, qk_d);
  const int fan = Cg * taps;
  const float* wr = w + (size_t)os * fan;
  const float* gr = dwp + (size_t)od * fan;
  auto cat_scale = [&](int i) { return in_split > 0 ? ((g * Cg + i / taps < in_split) ? in_s0 : in_s1) : 1.0f; };
  float* dr = dw + (size_t)os * fan;
  const bool vec = (fan & 3) == 0 && aligned16(wr) && aligned16(gr) && aligned16(dr);
  float ss = 0.f, su = 0.f;
  if (vec) {
    const f32x4 *pw = reinterpret_cast<const f32x4*>(wr), *pg = reinterpret_cast<const f32x4*>(gr);
    for (int i = lane; i < (fan >> 2); i += 64) {
      const f32x4 x = pw[i], gv = pg[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss += x[e] * x[e]; su += gv[e] * cat_scale(4 * i + e) * x[e]; }
    }
  } else {
    for (int i = lane; i < fan; i += 64) { const float x = wr[i]; ss += x * x; su += gr[i] * cat_scale(i) * x; }
  }
  ss = wave_sum(ss);
  su = wave_sum(su);
  const float rfan = sqrtf(1.0f / (float)fan);
  const float n = sqrtf(ss);
  const float nu = normalize ? eps + n * rfan : 1.0f;
  float gn = gain;
  if (gain_ptr) gn *= *gain_ptr;
  const float s = gn * rfan;
  const float k = (normalize && n > 0.f) ? su * rfan / (nu * n) : 0.f;
  if (vec) {
    const f32x4 *pw = reinterpret_cast<const f32x4*>(wr), *pg = reinterpret_cast<const f32x4*>(gr);
    f32x4* pd = reinterpret_cast<f32x4*>(dr);
    for (int i = lane; i < (fan >> 2); i += 64) {
      const f32x4 x = pw[i], gv = pg[i];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (s / nu) * (gv[e] * cat_scale(4 * i + e) - x[e] * k);
      pd[i] = v;
    }
  } else {
    for (int i = lane; i < fan; i += 64) dr[i] = (s / nu) * (gr[i] * cat_scale(i) - wr[i] * k);
  }
  if (dgain && lane == 0) atomicAdd(dgain, gain * su * rfan / nu);
}

}  // namespace ddx
