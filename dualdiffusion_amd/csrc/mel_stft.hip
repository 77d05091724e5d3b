// Fused mel-STFT: audio -> framed, windowed FFT (n_fft 6400) -> |.| -> banded mel filter bank -> ^exponent -> affine.
//
// Replaces the encode half of the reference's SpectrogramFormat (src/modules/formats/old/spectrogram.py:176-179,217-226:
// torchaudio Spectrogram == torch.stft(center=True, reflect, onesided, hann^32 window), `.abs()`,
// FrequencyScale.scale = dense (3201 x 256) matmul (src/modules/formats/frequency_scale.py:127-128), `** 0.25`,
// `(x - sample_mean) * raw_to_sample_scale`).  What is different here:
//   * the complex STFT (282 MB per 45 s stereo sample) never exists in HBM: one workgroup transforms a frame in LDS
//     (mixed-radix Stockham, fft_lds.hpp) and reduces it to 256 mel values on the spot;
//   * stereo rides in ONE complex FFT: z = left + i*right, |X_L[k]| = |Z[k] + conj(Z[N-k])|/2, |X_R[k]| = |Z[k] - conj(Z[N-k])|/2;
//   * the mel filter bank is applied as what it is -- 256 contiguous bands of 3..80 bins (integer start/length per
//     filter, bit-exact with the reference's non-zero support) -- not as a dense matmul (130x fewer FLOPs);
//   * FPW consecutive frames per workgroup so that the (B, C, n_mel, T) output is written in 4*FPW-byte runs.
// HBM-bound by design: algorithmic bytes = audio in (each sample is re-read by n_fft/hop = 25 overlapping frames, from L2)
// + mel out.
#include "fft_lds.hpp"

namespace ddx {

constexpr int kFPW = 8;      // frames per workgroup
constexpr int kNT = 512;      // (the transform runs as per-thread register transforms: fft_lds.hpp fft6400_reg; 1024 threads x 64 registers with the staged one)

struct MelStftParams {
  const float* audio; const float* window; const float2* tw;
  const int* bstart; const int* blen; const float* bw;
  float* out;
  int B, C, L, T, hop, n_mel, bstride;
  float exponent, mean, scale;
};

__device__ __forceinline__ int reflect_index(int j, int L) {
  // torch.stft(center=True, pad_mode="reflect"): mirror without repeating the edge sample
  if (j < 0) j = -j;
  if (j >= L) j = 2 * (L - 1) - j;
  return j;
}

template <int N>
__global__ __launch_bounds__(kNT, 4) void mel_stft_kernel(const MelStftParams p) {
  // LDS: ONE transform buffer (in-place FFT, fft_lds.hpp; the magnitudes later overwrite its first half) + the output
  // staging: 51 + 16 KB, two workgroups per CU overlap each other's load / transform / filter phases
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* bufA = reinterpret_cast<cf*>(smem);
  float* sOut = reinterpret_cast<float*>(bufA + kFft6400RegEntries);   // [C * n_mel][kFPW]
  const int b = blockIdx.y;
  // (XCD-contiguous frame groups: workgroup id & 7 is its XCD, which takes one contiguous eighth of the groups -- neighbouring frames share
  // most of the audio they read, and a round-robin deal makes every XCD's L2 fetch all of it: 6.4 x the waveform in r04's counters)
  int gi = blockIdx.x;
  {
    const int ng = gridDim.x, base = ng >> 3, rem = ng & 7, x = gi & 7;
    gi = x * base + min(x, rem) + (gi >> 3);
  }
  const int f0 = gi * kFPW;
  const int tid = threadIdx.x;
  const float* aL = p.audio + (size_t)b * p.C * p.L;
  const float* aR = p.C > 1 ? aL + p.L : nullptr;
  const int NB = N / 2 + 1;

#pragma unroll 1
  for (int fi = 0; fi < kFPW; ++fi) {
    const int tid = launder(threadIdx.x);   // (fft_lds.hpp: no hoisting of per-thread index arithmetic out of the frame loop)
    const int f = f0 + fi;
    if (f >= p.T) break;  // uniform
    // ---- frame load: z[n] = w[n] * (left + i*right), reflect padded by N/2
    const int base = f * p.hop - N / 2;
    for (int n = 4 * tid; n < N; n += 4 * kNT) {   // 16-byte loads in the interior (base, n multiples of 4)
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.window + n);
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
      for (int e = 0; e < 4; ++e) bufA[n + e] = cf{l4[e] * w4[e], r4[e] * w4[e]};
    }
    fft6400_reg<false, kNT>(bufA, p.tw, tid);
    // ---- magnitudes of both channels: into registers, barrier, then over the (now consumed) spectrum as floats
    float* mag = reinterpret_cast<float*>(bufA);
    constexpr int MI = (N / 2 + 1 + kNT - 1) / kNT;
    float ml[MI], mr[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int k = tid + i * kNT;
      ml[i] = mr[i] = 0.f;
      if (k < NB) {
        const cf zk = bufA[k], zn = cconj(bufA[(N - k) % N]);
        const cf sl = cadd(zk, zn), sr = csub(zk, zn);
        ml[i] = 0.5f * sqrtf(sl.x * sl.x + sl.y * sl.y);
        mr[i] = 0.5f * sqrtf(sr.x * sr.x + sr.y * sr.y);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int k = tid + i * kNT;
      if (k < NB) { mag[k] = ml[i]; mag[NB + k] = mr[i]; }
    }
    __syncthreads();
    // ---- banded mel filter bank, exponent, affine
    for (int o = tid; o < p.C * p.n_mel; o += kNT) {
      const int ch = o / p.n_mel, m = o - ch * p.n_mel;
      const float* mg = mag + ch * NB + p.bstart[m];
      const float* wv = p.bw + (size_t)m * p.bstride;
      float acc = 0.f;
      for (int i = 0; i < p.blen[m]; ++i) acc += mg[i] * wv[i];
      float v = (p.exponent == 0.25f) ? sqrtf(sqrtf(acc)) : powf(acc, p.exponent);
      sOut[o * kFPW + fi] = (v - p.mean) * p.scale;
    }
    __syncthreads();
  }
  // ---- write the FPW frames of every (channel, mel) row as one run
  const int nf = min(kFPW, p.T - f0);
  for (int idx = tid; idx < p.C * p.n_mel * kFPW; idx += kNT) {
    const int o = idx / kFPW, fi = idx - o * kFPW;
    if (fi < nf) p.out[((size_t)b * p.C * p.n_mel + o) * p.T + f0 + fi] = sOut[idx];
  }
}

}  // namespace ddx

using namespace ddx;

extern "C" int ddx_mel_stft(const ddx_melstft_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "mel_stft: null descriptor");
  const ddx_melstft_desc d = *dp;
  if (!d.audio || !d.window || !d.twiddle || !d.band_start || !d.band_len || !d.band_w || !d.out)
    return set_error(DDX_ERR_ARG, "mel_stft: null buffer");
  if (d.B <= 0 || (d.C != 1 && d.C != 2) || d.L <= 0 || d.T <= 0 || d.hop <= 0 || d.n_mel <= 0 || d.band_stride <= 0)
    return set_error(DDX_ERR_ARG, "mel_stft: bad size");
  if (d.n_fft != 6400) return set_error(DDX_ERR_UNSUPPORTED, "mel_stft: only n_fft = 6400 is built");
  if (d.L <= d.n_fft / 2) return set_error(DDX_ERR_ARG, "mel_stft: audio shorter than the reflect padding");
  MelStftParams p{d.audio, d.window, reinterpret_cast<const float2*>(d.twiddle), d.band_start, d.band_len, d.band_w, d.out,
                  d.B, d.C, d.L, d.T, d.hop, d.n_mel, d.band_stride, d.exponent, d.mean, d.scale};
  return dispatch([p](hipStream_t s) -> int {
    constexpr int N = 6400;
    const size_t smem = (size_t)kFft6400RegEntries * sizeof(cf) + (size_t)p.C * p.n_mel * kFPW * sizeof(float);
    if (smem > 160 * 1024) return set_error(DDX_ERR_UNSUPPORTED, "mel_stft: too many mel bands for LDS");
    auto kern = mel_stft_kernel<N>;
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(mel_stft)");
      attr_done = true;
    }
    dim3 grid((p.T + kFPW - 1) / kFPW, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(kNT), smem, s, p);
    return check_launch("mel_stft");
  }, stream, "mel_stft", 5.0 * 6400 * 12.64 * p.T * p.B,
     4.0 * ((double)p.B * p.C * p.L + (double)p.B * p.C * p.n_mel * p.T));
}
