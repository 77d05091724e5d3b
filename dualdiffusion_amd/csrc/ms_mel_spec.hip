// Dual-window mel-scale spectrogram of the live format (reference src/modules/formats/ms_mdct_dual.py:229-257
// `MS_MDCT_DualFormat.raw_to_mel_spec`): two magnitude STFTs of the same 4096-sample frames -- Blackman-Harris^17 (frequency
// resolution for the lows) and Blackman-Harris^58 (time resolution for the highs), each normalised by its window's L2 norm --
// blended per bin by the squared relative mel density, divided by the mel density, reduced by the slaney-normalised triangular
// bank (2049 -> 256, 2..52 bins per filter) and scaled.  As in mel_stft.hip nothing but audio is read and nothing but mel values
// is written: per frame a workgroup runs TWO in-LDS radix-4 FFT-4096 (stereo packed as left + i*right in each), un-mixes the
// four magnitude spectra by conjugate symmetry, blends them in registers and applies the banded filter bank.
// The per-bin factors blend / density and (1 - blend) / density are host tables; both windows arrive pre-normalised.
#include "fft_lds.hpp"

namespace ddx {
namespace {

constexpr int kFPW = 8;
constexpr int kNT = 512;      // (per-thread register transforms, fft_lds.hpp fft4096_reg; 1024 threads x 64 registers with the staged one)
constexpr int N = 4096, NB = N / 2 + 1;

struct MsMelParams {
  const float* audio; const float* w_low; const float* w_high; const float2* tw;
  const float* bin_low; const float* bin_high;
  const int* bstart; const int* blen; const float* bw;
  float* out;
  int B, C, L, T, hop, n_mel, bstride;
  float exponent, scale, offset;
};

__device__ __forceinline__ int reflect_index(int j, int L) {
  if (j < 0) j = -j;
  if (j >= L) j = 2 * (L - 1) - j;
  return j;
}

__global__ __launch_bounds__(kNT, 4) void ms_mel_spec_kernel(const MsMelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* buf = reinterpret_cast<cf*>(smem);
  float* sOut = reinterpret_cast<float*>(buf + kFft4096RegEntries);   // [C * n_mel][kFPW]
  const int b = blockIdx.y;
  // (XCD-contiguous frame groups: workgroup id & 7 is its XCD, which takes one contiguous eighth of the groups -- neighbouring frames share
  // most of the audio they read, and a round-robin deal makes every XCD's L2 fetch all of it: 6.4 x the waveform in r04's counters)
  int gi = blockIdx.x;
  {
    const int ng = gridDim.x, base = ng >> 3, rem = ng & 7, x = gi & 7;
    gi = x * base + min(x, rem) + (gi >> 3);
  }
  const int f0 = gi * kFPW;
  const float* aL = p.audio + (size_t)b * p.C * p.L;
  const float* aR = p.C > 1 ? aL + p.L : nullptr;
  constexpr int MI = (NB + kNT - 1) / kNT;

#pragma unroll 1
  for (int fi = 0; fi < kFPW; ++fi) {
    const int tid = launder(threadIdx.x);
    const int f = f0 + fi;
    if (f >= p.T) break;  // uniform
    const int base = f * p.hop - N / 2;
    float bl[MI], br[MI];   // blended magnitudes of this thread's bins
#pragma unroll
    for (int i = 0; i < MI; ++i) bl[i] = br[i] = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const float* win = pass == 0 ? p.w_low : p.w_high;
      const float* bin = pass == 0 ? p.bin_low : p.bin_high;
      for (int n = 4 * tid; n < N; n += 4 * kNT) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(win + n);
        const int j0 = base + n;
        f32x4 l4, r4 = {0.f, 0.f, 0.f, 0.f};
        if (j0 >= 0 && j0 + 3 < p.L && (p.L & 3) == 0 && (p.hop & 3) == 0) {
          l4 = *reinterpret_cast<const f32x4*>(aL + j0);
          if (aR) r4 = *reinterpret_cast<const f32x4*>(aR + j0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = reflect_index(j0 + e, p.L);
            l4[e] = aL[j];
            if (aR) r4[e] = aR[j];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) buf[n + e] = cf{l4[e] * w4[e], r4[e] * w4[e]};
      }
      fft4096_reg<false, kNT>(buf, p.tw, tid);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int k = tid + i * kNT;
        if (k < NB) {
          const cf zk = buf[k], zn = cconj(buf[(N - k) & (N - 1)]);
          const cf sl = cadd(zk, zn), sr = csub(zk, zn);
          const float fk = 0.5f * bin[k];
          bl[i] += fk * sqrtf(sl.x * sl.x + sl.y * sl.y);
          br[i] += fk * sqrtf(sr.x * sr.x + sr.y * sr.y);
        }
      }
      __syncthreads();   // every thread has read its bins before the buffer is refilled / overwritten
    }
    float* mag = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int k = tid + i * kNT;
      if (k < NB) { mag[k] = bl[i]; mag[NB + k] = br[i]; }
    }
    __syncthreads();
    for (int o = tid; o < p.C * p.n_mel; o += kNT) {
      const int ch = o / p.n_mel, m = o - ch * p.n_mel;
      const float* mg = mag + ch * NB + p.bstart[m];
      const float* wv = p.bw + (size_t)m * p.bstride;
      float acc = 0.f;
      for (int i = 0; i < p.blen[m]; ++i) acc += mg[i] * wv[i];
      const float v = (p.exponent == 1.0f) ? acc : powf(acc, p.exponent);
      sOut[o * kFPW + fi] = v * p.scale + p.offset;
    }
    __syncthreads();
  }
  const int nf = min(kFPW, p.T - f0);
  for (int idx = threadIdx.x; idx < p.C * p.n_mel * kFPW; idx += kNT) {
    const int o = idx / kFPW, fi = idx - o * kFPW;
    if (fi < nf) p.out[((size_t)b * p.C * p.n_mel + o) * p.T + f0 + fi] = sOut[idx];
  }
}

}  // namespace
}  // namespace ddx

using namespace ddx;

extern "C" int ddx_ms_mel_spec(const ddx_msmel_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "ms_mel_spec: null descriptor");
  const ddx_msmel_desc d = *dp;
  if (!d.audio || !d.window_low || !d.window_high || !d.twiddle || !d.bin_scale_low || !d.bin_scale_high || !d.band_start || !d.band_len ||
      !d.band_w || !d.out)
    return set_error(DDX_ERR_ARG, "ms_mel_spec: null buffer");
  if (d.B <= 0 || (d.C != 1 && d.C != 2) || d.L <= 0 || d.T <= 0 || d.hop <= 0 || d.n_mel <= 0 || d.band_stride <= 0)
    return set_error(DDX_ERR_ARG, "ms_mel_spec: bad size");
  if (d.n_fft != 4096) return set_error(DDX_ERR_UNSUPPORTED, "ms_mel_spec: only n_fft = 4096 is built");
  if (d.L <= d.n_fft / 2) return set_error(DDX_ERR_ARG, "ms_mel_spec: audio shorter than the reflect padding");
  MsMelParams p{d.audio, d.window_low, d.window_high, reinterpret_cast<const float2*>(d.twiddle), d.bin_scale_low, d.bin_scale_high,
                d.band_start, d.band_len, d.band_w, d.out, d.B, d.C, d.L, d.T, d.hop, d.n_mel, d.band_stride, d.exponent, d.scale, d.offset};
  return dispatch([p](hipStream_t s) -> int {
    const size_t smem = (size_t)kFft4096RegEntries * sizeof(cf) + (size_t)p.C * p.n_mel * kFPW * sizeof(float);
    if (smem > 160 * 1024) return set_error(DDX_ERR_UNSUPPORTED, "ms_mel_spec: too many mel bands for LDS");
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ms_mel_spec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(ms_mel_spec)");
      attr_done = true;
    }
    dim3 grid((p.T + kFPW - 1) / kFPW, p.B);
    hipLaunchKernelGGL(ms_mel_spec_kernel, grid, dim3(kNT), smem, s, p);
    return check_launch("ms_mel_spec");
  }, stream, "ms_mel_spec", 2.0 * 5.0 * 4096 * 12.0 * p.T * p.B, 4.0 * ((double)p.B * p.C * p.L + (double)p.B * p.C * p.n_mel * p.T));
}
