// FGLA (fast Griffin-Lim with momentum) stereo phase reconstruction: the decode half of the reference's
// SpectrogramFormat (src/modules/formats/old/phase_recovery.py:39-129 `griffinlim`, driven by spectrogram.py:181-185).
//
// Per iteration the reference does: lerp(merged, spec, t) -> angles*mags -> torch.istft -> torch.stft -> angles =
// rebuilt.sub_(tprev, alpha=m) -> angles /= (|angles| + 1e-16) -> tprev = rebuilt, i.e. ~8 full passes over (2B, 3201, T)
// complex64 tensors.  NB the subtraction is in place on the tensor that becomes tprev (phase_recovery.py:110-119), so the
// carried state is u_i = rebuilt_i - m*u_{i-1} and angles_i = u_i / (|u_i| + 1e-16): ONE full-size state tensor.
// Here one iteration is three kernels:
//   synth    : per frame, angles are re-derived on the fly from u, multiplied by the (stereo-annealed) magnitudes,
//              both channels packed as Z = X_L + i X_R, ONE inverse FFT-6400 in LDS, windowed frame written out;
//   ola      : overlap-add of the 25 frames covering each sample + division by the window envelope (torch.istft);
//   analysis : reflect-padded frame * window, forward FFT-6400, unpack both channels, u <- rebuilt - m*u in place.
// Spectra are stored frame-major [B][T][C][n_fft/2+1] complex (the reference's [freq][time] order would make every
// frame access strided).  HBM-bound: per iteration and sample ~2*282 MB + 141 MB read, 282 MB written for the state,
// plus the 282 MB frame buffer round trip.
#include "fft_lds.hpp"

namespace ddx {

// -DDDX_FGLA_TRACE (tools/fgla_trace.sh): thread 0 of every workgroup times its three phases -- operands into LDS (global loads + math),
// the in-LDS FFT-6400, results out (global stores issued) -- with s_memtime, plus the workgroup's start / end on the constant 100 MHz clock;
// the host prints per-phase cycles per workgroup and how many workgroups were in flight per CU on average (sum of workgroup lifetimes /
// kernel span / 256).  A barrier is added after the first phase so that the phases are the whole workgroup's, not thread 0's.
#ifdef DDX_FGLA_TRACE
#include <cstdio>
#include <vector>
constexpr int kFtMax = 1 << 16;
__device__ unsigned long long g_ftrace[kFtMax][6];   // per workgroup: load, fft, store cycles; start, end (100 MHz ticks); valid -- plain stores, no contention
#define DDX_FT_BEGIN() long long ft_t0 = clock64(), ft_t1 = 0, ft_t2 = 0; const unsigned long long ft_w0 = wall_clock64()
#define DDX_FT_LOADED() do { __syncthreads(); ft_t1 = clock64(); } while (0)
#define DDX_FT_FFT() do { __syncthreads(); ft_t2 = clock64(); } while (0)
#define DDX_FT_END(k) do { if (threadIdx.x == 0) { const long long t3 = clock64(); const unsigned long long w1 = wall_clock64(); \
    const int wgid = blockIdx.y * gridDim.x + blockIdx.x; if (wgid < kFtMax) { unsigned long long* r = g_ftrace[wgid]; \
    r[0] = ft_t1 - ft_t0; r[1] = ft_t2 - ft_t1; r[2] = t3 - ft_t2; r[3] = ft_w0; r[4] = w1; r[5] = 1; } } } while (0)
#else
#define DDX_FT_BEGIN() do {} while (0)
#define DDX_FT_LOADED() do {} while (0)
#define DDX_FT_FFT() do {} while (0)
#define DDX_FT_END(k) do {} while (0)
#endif

constexpr int kFN = 6400;
// Threads per frame: 512 (the 6400-point transform runs as per-thread register transforms, fft_lds.hpp fft6400_reg: 400 / 400 / 256 lines in its
// three passes) at <= 128 registers: two workgroups per CU.  History: rounds 1-3 ran the staged transform on 640 threads (1.9 workgroups in
// flight per CU: ten waves land 3 + 3 + 2 + 2 on the SIMDs), round 4 first on 512 threads / 64 registers (2.9 in flight: 2.06 -> 1.95 ms per
// iteration at B=4) and then with the register transform: 1.65 -> 1.45 ms (256 threads x two lines at three workgroups per CU: 1.73; an
// 80-register cap spills: 2.4 ms).  (-DDDX_FGLA_NT=n -DDDX_FGLA_MINWAVES=m rebuild with other counts.)
#ifndef DDX_FGLA_NT
#define DDX_FGLA_NT 512
#endif
#ifndef DDX_FGLA_MINWAVES
#define DDX_FGLA_MINWAVES 4
#endif
constexpr int kFNT = DDX_FGLA_NT;
constexpr int kFMinWaves = DDX_FGLA_MINWAVES;   // (__launch_bounds__ second argument = min waves per SIMD: 4 -> <= 128 VGPRs)

// Frame of a workgroup.  Consecutive frames share 24 / 25 of the audio they read (hop 256 of a 6400-sample window) and workgroups are dealt
// round-robin to the 8 XCDs: with t = blockIdx.x every XCD's L2 fetches the whole waveform.  XCD x (= blockIdx.x & 7) takes one contiguous
// eighth of the frames instead (round 5; the same remedy as the conv tiles and the MSS blocks).
__device__ __forceinline__ int fgla_frame_of_block(int T) {
  const int id = blockIdx.x, base = T >> 3, rem = T & 7, x = id & 7;
  return x * base + min(x, rem) + (id >> 3);
}

struct FglaSynthParams {
  const float2* u;                          // [B][T][C][ustride] state, NB valid per row (nullptr: angles = 1)
  const float* mags;                        // [B][C][T][mstride]
  const float* window; const float2* tw;
  float* frames;                            // [B][T][C][N]
  int B, C, T, mstride, ustride;
  float t_lerp;                             // t_lerp <= 0: merged magnitudes; final: the magnitudes themselves
  int final_pass, stereo_merge;
};

__global__ __launch_bounds__(kFNT, kFMinWaves) void fgla_synth_kernel(const FglaSynthParams p) {
  constexpr int N = kFN, NB = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* bufA = reinterpret_cast<cf*>(smem);
  const int t = fgla_frame_of_block(p.T), b = blockIdx.y, tid = threadIdx.x;
  DDX_FT_BEGIN();
  const size_t sbase = ((size_t)b * p.T + t) * p.C * p.ustride;
  // two bins per lane: rows of the state (ustride even) and of the magnitudes (mstride even) start 16 / 8-byte aligned,
  // so the state is read as one 16-byte vector per channel; the second bin of the last pair (k = NB) is row padding
  for (int k0 = 2 * tid; k0 < NB; k0 += 2 * kFNT) {
    float mg[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [channel][bin]
    f32x4 uu[2] = {{1.f, 0.f, 1.f, 0.f}, {1.f, 0.f, 1.f, 0.f}};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
      if (ch < p.C) {
        const float2 m2 = *reinterpret_cast<const float2*>(p.mags + (((size_t)b * p.C + ch) * p.T + t) * p.mstride + k0);
        mg[ch][0] = fmaxf(m2.x, 0.f); mg[ch][1] = fmaxf(m2.y, 0.f);            // relu of the un-mel
        if (p.u) uu[ch] = *reinterpret_cast<const f32x4*>(p.u + sbase + (size_t)ch * p.ustride + k0);
      }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + e;
      if (k >= NB) break;
      cf x[2] = {cf{0.f, 0.f}, cf{0.f, 0.f}};
      const float merged = 0.5f * (mg[0][e] + mg[1][e]);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        if (ch >= p.C) continue;
        cf ang = {1.f, 0.f};
        if (p.u) {
          const float ax = uu[ch][2 * e], ay = uu[ch][2 * e + 1];
          const float inv = 1.0f / (sqrtf(ax * ax + ay * ay) + 1e-16f);
          ang = cf{ax * inv, ay * inv};
        }
        float m = mg[ch][e];
        if (!p.final_pass && p.stereo_merge) m = p.t_lerp > 0.f ? merged + p.t_lerp * (mg[ch][e] - merged) : merged;
        x[ch] = cf{ang.x * m, ang.y * m};
        if (k == 0 || k == N / 2) x[ch].y = 0.f;  // c2r semantics: DC and Nyquist are real
      }
      // Z = X_L + i X_R ;  Z[N-k] = conj(X_L) + i conj(X_R)
      bufA[k] = cf{x[0].x - x[1].y, x[0].y + x[1].x};
      if (k > 0 && k < N / 2) bufA[N - k] = cf{x[0].x + x[1].y, -x[0].y + x[1].x};
    }
  }
  DDX_FT_LOADED();
  fft6400_reg<true, kFNT>(bufA, p.tw, tid);
  DDX_FT_FFT();
  float* fr = p.frames + ((size_t)b * p.T + t) * p.C * N;
  const float invn = 1.0f / (float)N;
  for (int n = 4 * tid; n < N; n += 4 * kFNT) {   // 16 bytes per lane: four samples of each channel
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.window + n);
    f32x4 l4, r4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const cf z = bufA[n + e];
      l4[e] = z.x * (w4[e] * invn);
      r4[e] = z.y * (w4[e] * invn);
    }
    *reinterpret_cast<f32x4*>(fr + n) = l4;
    if (p.C > 1) *reinterpret_cast<f32x4*>(fr + N + n) = r4;
  }
  DDX_FT_END(0);
}

// overlap-add + window-envelope normalisation (torch.istft, center=True, length = hop*(T-1))
__global__ __launch_bounds__(256) void fgla_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                       float* __restrict__ audio, int B, int C, int T, int hop, int Lout) {
  constexpr int N = kFN;
  // four consecutive samples per lane (hop, N/2 and Lout are multiples of 4, so they share their frame range and every
  // access is a 16-byte one)
  const size_t total4 = (size_t)B * C * Lout / 4;
  for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * 256) {
    const size_t i = i4 * 4;
    const int j = (int)(i % Lout);
    const int ch = (int)((i / Lout) % C);
    const int b = (int)(i / ((size_t)Lout * C));
    const int jp = j + N / 2;
    int t0 = (jp + 3 - N + hop) / hop;  // ceil((jp + 3 - N + 1) / hop): first frame that covers the LAST of the four samples ...
    if (jp + 3 - N + 1 <= 0) t0 = 0;    // ... and, the group being 4-aligned inside a hop, the first one too
    const int t1 = min(jp / hop, T - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, env = {0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t <= t1; ++t) {
      const int n = jp - t * hop;
      const f32x4 w = *reinterpret_cast<const f32x4*>(window + n);
      const f32x4 f = *reinterpret_cast<const f32x4*>(frames + (((size_t)b * T + t) * C + ch) * N + n);
      acc += f;
      env += w * w;
    }
    *reinterpret_cast<f32x4*>(audio + i) = acc / env;
  }
}

struct FglaAnalysisParams {
  const float* audio;  // [B][C][L]
  const float* window; const float2* tw;
  float2* u;           // [B][T][C][ustride] state (NB valid per row), updated in place: u = rebuilt - momentum * u
  int B, C, T, L, hop, ustride;
  float momentum;
};

__device__ __forceinline__ int reflect_idx(int j, int L) {
  if (j < 0) j = -j;
  if (j >= L) j = 2 * (L - 1) - j;
  return j;
}

__global__ __launch_bounds__(kFNT, kFMinWaves) void fgla_analysis_kernel(const FglaAnalysisParams p) {
  constexpr int N = kFN, NB = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* bufA = reinterpret_cast<cf*>(smem);
  const int t = fgla_frame_of_block(p.T), b = blockIdx.y, tid = threadIdx.x;
  DDX_FT_BEGIN();
  const float* aL = p.audio + (size_t)b * p.C * p.L;
  const float* aR = p.C > 1 ? aL + p.L : nullptr;
  const int base = t * p.hop - N / 2;
  for (int n = 4 * tid; n < N; n += 4 * kFNT) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.window + n);
    const int j0 = base + n;
    f32x4 l4, r4 = {0.f, 0.f, 0.f, 0.f};
    if (j0 >= 0 && j0 + 3 < p.L && (p.L & 3) == 0) {   // interior: 16-byte loads (base and n are multiples of 4)
      l4 = *reinterpret_cast<const f32x4*>(aL + j0);
      if (aR) r4 = *reinterpret_cast<const f32x4*>(aR + j0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = reflect_idx(j0 + e, p.L);
        l4[e] = aL[j];
        if (aR) r4[e] = aR[j];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) bufA[n + e] = cf{l4[e] * w4[e], r4[e] * w4[e]};
  }
  DDX_FT_LOADED();
  fft6400_reg<false, kFNT>(bufA, p.tw, tid);
  DDX_FT_FFT();
  float2* ro = p.u + ((size_t)b * p.T + t) * p.C * p.ustride;
  for (int k0 = 2 * tid; k0 < NB; k0 += 2 * kFNT) {   // two bins per lane: 16-byte read-modify-write of the state rows
    f32x4 ul = *reinterpret_cast<const f32x4*>(ro + k0), ur = {0.f, 0.f, 0.f, 0.f};
    if (p.C > 1) ur = *reinterpret_cast<const f32x4*>(ro + p.ustride + k0);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + e;
      if (k >= NB) break;    // (the second bin of the last pair is row padding: written back unchanged)
      const cf zk = bufA[k], zn = cconj(bufA[(N - k) % N]);
      const cf sl = cadd(zk, zn), sr = csub(zk, zn);
      ul[2 * e] = 0.5f * sl.x - p.momentum * ul[2 * e];            // X_L = (Z[k] + conj Z[N-k]) / 2
      ul[2 * e + 1] = 0.5f * sl.y - p.momentum * ul[2 * e + 1];
      ur[2 * e] = 0.5f * sr.y - p.momentum * ur[2 * e];            // X_R = (Z[k] - conj Z[N-k]) / (2i)
      ur[2 * e + 1] = -0.5f * sr.x - p.momentum * ur[2 * e + 1];
    }
    *reinterpret_cast<f32x4*>(ro + k0) = ul;
    if (p.C > 1) *reinterpret_cast<f32x4*>(ro + p.ustride + k0) = ur;
  }
  DDX_FT_END(1);
}

// analysis of iteration i and synthesis of iteration i + 1 of one frame in ONE workgroup.  Both are per-frame with the same (t, b) grid, and
// the synthesis of a frame reads nothing but that frame's state row (just written) and magnitudes: fused, the state is read once and written
// once per iteration instead of read twice (6.25 -> 4.7 GB per iteration at B = 4: state 3 x 1.1 GB -> 2 x, frames 2 x 1.1 GB, magnitudes),
// one launch and one LDS fill less.  With the staged transform the two forms took the same time (1.65 ms: the transforms bound it); with the
// register transform 1.62 -> 1.47 ms.  The loop over the bin pairs reads Z[k] and
// Z[N - k] of the forward transform, updates the state, and writes the next iteration's packed spectrum into the same two entries -- the
// only entries of the buffer this thread touches, so the hand-over needs no barrier.
struct FglaIterParams {
  const float* audio; const float* mags; const float* window; const float2* tw;
  float2* u; float* frames;
  int B, C, T, L, hop, ustride, mstride;
  float momentum, t_lerp;
  int final_pass, stereo_merge;
};

__global__ __launch_bounds__(kFNT, kFMinWaves) void fgla_iter_kernel(const FglaIterParams p) {
  constexpr int N = kFN, NB = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* bufA = reinterpret_cast<cf*>(smem);
  const int t = fgla_frame_of_block(p.T), b = blockIdx.y, tid = threadIdx.x;
  const float* aL = p.audio + (size_t)b * p.C * p.L;
  const float* aR = p.C > 1 ? aL + p.L : nullptr;
  const int base = t * p.hop - N / 2;
  for (int n = 4 * tid; n < N; n += 4 * kFNT) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.window + n);
    const int j0 = base + n;
    f32x4 l4, r4 = {0.f, 0.f, 0.f, 0.f};
    if (j0 >= 0 && j0 + 3 < p.L && (p.L & 3) == 0) {   // interior: 16-byte loads (base and n are multiples of 4)
      l4 = *reinterpret_cast<const f32x4*>(aL + j0);
      if (aR) r4 = *reinterpret_cast<const f32x4*>(aR + j0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = reflect_idx(j0 + e, p.L);
        l4[e] = aL[j];
        if (aR) r4[e] = aR[j];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) bufA[n + e] = cf{l4[e] * w4[e], r4[e] * w4[e]};
  }
  fft6400_reg<false, kFNT>(bufA, p.tw, tid);
  float2* ro = p.u + ((size_t)b * p.T + t) * p.C * p.ustride;
  // (the state rows and magnitudes of all rounds are requested together: four load -> compute -> store round trips in a row were the longest
  // phase of the frame once the transforms shrank)
  constexpr int ITER = (NB + 2 * kFNT - 1) / (2 * kFNT);
  f32x4 ulv[ITER], urv[ITER];
  float2 m2v[ITER][2];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int k0 = 2 * tid + it * 2 * kFNT;
    const int kc = k0 < NB ? k0 : 0;
    ulv[it] = *reinterpret_cast<const f32x4*>(ro + kc);
    urv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.C > 1) urv[it] = *reinterpret_cast<const f32x4*>(ro + p.ustride + kc);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      m2v[it][ch] = float2{0.f, 0.f};
      if (ch < p.C) m2v[it][ch] = *reinterpret_cast<const float2*>(p.mags + (((size_t)b * p.C + ch) * p.T + t) * p.mstride + kc);
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int k0 = 2 * tid + it * 2 * kFNT;
    if (k0 >= NB) continue;
    f32x4 ul = ulv[it], ur = urv[it];
    float mg[2][2];   // [channel][bin]
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) { mg[ch][0] = fmaxf(m2v[it][ch].x, 0.f); mg[ch][1] = fmaxf(m2v[it][ch].y, 0.f); }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + e;
      if (k >= NB) break;    // (the second bin of the last pair is row padding: written back unchanged)
      // ---- analysis (fgla_analysis_kernel): the state of this bin
      const cf zk = bufA[k], zn = cconj(bufA[(N - k) % N]);
      const cf sl = cadd(zk, zn), sr = csub(zk, zn);
      ul[2 * e] = 0.5f * sl.x - p.momentum * ul[2 * e];
      ul[2 * e + 1] = 0.5f * sl.y - p.momentum * ul[2 * e + 1];
      ur[2 * e] = 0.5f * sr.y - p.momentum * ur[2 * e];
      ur[2 * e + 1] = -0.5f * sr.x - p.momentum * ur[2 * e + 1];
      // ---- synthesis (fgla_synth_kernel) from the value just formed
      cf x[2] = {cf{0.f, 0.f}, cf{0.f, 0.f}};
      const float merged = 0.5f * (mg[0][e] + mg[1][e]);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        if (ch >= p.C) continue;
        const float ax = ch ? ur[2 * e] : ul[2 * e], ay = ch ? ur[2 * e + 1] : ul[2 * e + 1];
        const float inv = 1.0f / (sqrtf(ax * ax + ay * ay) + 1e-16f);
        float m = mg[ch][e];
        if (!p.final_pass && p.stereo_merge) m = p.t_lerp > 0.f ? merged + p.t_lerp * (mg[ch][e] - merged) : merged;
        x[ch] = cf{ax * inv * m, ay * inv * m};
        if (k == 0 || k == N / 2) x[ch].y = 0.f;
      }
      bufA[k] = cf{x[0].x - x[1].y, x[0].y + x[1].x};
      if (k > 0 && k < N / 2) bufA[N - k] = cf{x[0].x + x[1].y, -x[0].y + x[1].x};
    }
    *reinterpret_cast<f32x4*>(ro + k0) = ul;
    if (p.C > 1) *reinterpret_cast<f32x4*>(ro + p.ustride + k0) = ur;
  }
  fft6400_reg<true, kFNT>(bufA, p.tw, tid);
  float* fr = p.frames + ((size_t)b * p.T + t) * p.C * N;
  const float invn = 1.0f / (float)N;
  for (int n = 4 * tid; n < N; n += 4 * kFNT) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.window + n);
    f32x4 l4, r4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const cf z = bufA[n + e];
      l4[e] = z.x * (w4[e] * invn);
      r4[e] = z.y * (w4[e] * invn);
    }
    *reinterpret_cast<f32x4*>(fr + n) = l4;
    if (p.C > 1) *reinterpret_cast<f32x4*>(fr + N + n) = r4;
  }
}

// mel samples (B, C, n_mel, T) -> linear mel amplitudes laid out [B*C][T][n_mel] for the un-mel GEMM:
// amp = clip(x / scale + mean, 0) ** (1 / abs_exponent)   (reference spectrogram.py:232,183)
__global__ __launch_bounds__(256) void mel_to_amp_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int n_mel, int T,
                                                         float inv_scale, float mean, float power) {
  const size_t total = (size_t)rows * n_mel * T;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i % n_mel);
    const int t = (int)((i / n_mel) % T);
    const int r = (int)(i / ((size_t)n_mel * T));
    const float v = fmaxf(x[((size_t)r * n_mel + m) * T + t] * inv_scale + mean, 0.f);
    y[i] = (power == 4.0f) ? (v * v) * (v * v) : powf(v, power);
  }
}

}  // namespace ddx

using namespace ddx;

// trace builds: print and reset the per-phase counters of kernel k after its launch (synchronises: timing experiments only)
static void fgla_trace_report(int k, const char* name) {
#ifdef DDX_FGLA_TRACE
  (void)k;
  static std::vector<unsigned long long> h((size_t)kFtMax * 6);
  if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_ftrace), h.size() * 8) == hipSuccess) {
    double ph[3] = {0, 0, 0}, life = 0, n = 0;
    unsigned long long w0 = ~0ull, w1 = 0;
    for (int i = 0; i < kFtMax; ++i) {
      const unsigned long long* r = &h[(size_t)i * 6];
      if (!r[5]) continue;
      ph[0] += r[0]; ph[1] += r[1]; ph[2] += r[2]; life += (double)(r[4] - r[3]); n += 1;
      w0 = std::min(w0, r[3]); w1 = std::max(w1, r[4]);
    }
    if (n > 0) {
      const double span = (double)(w1 - w0);
      fprintf(stderr, "[fgla trace] %-8s %6.0f workgroups: cycles per workgroup  operands->LDS %7.0f  FFT-6400 %7.0f  results out %7.0f | lifetime %.2f us, "
              "kernel span %.1f us, %.2f workgroups in flight per CU\n", name, n, ph[0] / n, ph[1] / n, ph[2] / n, life / n / 100.0, span / 100.0,
              span > 0 ? life / span / 256.0 : 0.0);
    }
  }
  (void)hipMemset(nullptr, 0, 0);
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_ftrace)) == hipSuccess) (void)hipMemset(sym, 0, (size_t)kFtMax * 6 * 8);
#else
  (void)k; (void)name;
#endif
}

static int set_fft_smem(const void* kern, bool* done) {
  if (!*done) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(fgla)");
    *done = true;
  }
  return DDX_OK;
}

extern "C" int ddx_fgla_synth(const float* u, int32_t u_stride, const float* mags, const float* window, const float* twiddle, float* frames,
                              int32_t B, int32_t C, int32_t T, int32_t n_fft, int32_t mag_stride, float t_lerp, int32_t final_pass,
                              ddx_stream stream) {
  if (!mags || !window || !twiddle || !frames || B <= 0 || (C != 1 && C != 2) || T <= 0) return set_error(DDX_ERR_ARG, "fgla_synth: bad args");
  if (n_fft != kFN || mag_stride < kFN / 2 + 1) return set_error(DDX_ERR_UNSUPPORTED, "fgla_synth: only n_fft = 6400 is built");
  if (mag_stride < kFN / 2 + 2 || (mag_stride & 1) || (u && (u_stride < kFN / 2 + 2 || (u_stride & 1))))
    return set_error(DDX_ERR_ARG, "fgla_synth: row strides must be even and >= n_fft/2 + 2 (two bins per 16-byte access)");
  FglaSynthParams p{reinterpret_cast<const float2*>(u), mags, window, reinterpret_cast<const float2*>(twiddle), frames, B, C, T,
                    mag_stride, u_stride, t_lerp, final_pass, C == 2};
  return dispatch([p](hipStream_t s) -> int {
    static bool done = false;
    if (int rc = set_fft_smem(reinterpret_cast<const void*>(fgla_synth_kernel), &done)) return rc;
    hipLaunchKernelGGL(fgla_synth_kernel, dim3(p.T, p.B), dim3(kFNT), kFft6400RegEntries * sizeof(cf), s, p);
    fgla_trace_report(0, "synth");
    return check_launch("fgla_synth");
  }, stream, "fgla_synth");
}

extern "C" int ddx_fgla_ola(const float* frames, const float* window, float* audio, int32_t B, int32_t C, int32_t T, int32_t n_fft,
                            int32_t hop, ddx_stream stream) {
  if (!frames || !window || !audio || B <= 0 || C <= 0 || T <= 1 || hop <= 0) return set_error(DDX_ERR_ARG, "fgla_ola: bad args");
  if (n_fft != kFN) return set_error(DDX_ERR_UNSUPPORTED, "fgla_ola: only n_fft = 6400 is built");
  if (hop % 4) return set_error(DDX_ERR_UNSUPPORTED, "fgla_ola: hop must be a multiple of 4 (16-byte accesses)");
  return dispatch([=](hipStream_t s) -> int {
    const int Lout = hop * (T - 1);
    const size_t total = (size_t)B * C * Lout;
    hipLaunchKernelGGL(fgla_ola_kernel, dim3((unsigned)std::min<size_t>((total / 4 + 255) / 256, 65536)), dim3(256), 0, s, frames, window,
                       audio, B, C, T, hop, Lout);
    return check_launch("fgla_ola");
  }, stream, "fgla_ola");
}

extern "C" int ddx_fgla_analysis(const float* audio, const float* window, const float* twiddle, float* u, int32_t u_stride, int32_t B,
                                 int32_t C, int32_t T, int32_t L, int32_t n_fft, int32_t hop, float momentum, ddx_stream stream) {
  if (!audio || !window || !twiddle || !u || B <= 0 || (C != 1 && C != 2) || T <= 0 || L <= kFN / 2) return set_error(DDX_ERR_ARG, "fgla_analysis: bad args");
  if (n_fft != kFN) return set_error(DDX_ERR_UNSUPPORTED, "fgla_analysis: only n_fft = 6400 is built");
  if (u_stride < kFN / 2 + 2 || (u_stride & 1)) return set_error(DDX_ERR_ARG, "fgla_analysis: u_stride must be even and >= n_fft/2 + 2");
  FglaAnalysisParams p{audio, window, reinterpret_cast<const float2*>(twiddle), reinterpret_cast<float2*>(u), B, C, T, L, hop, u_stride, momentum};
  return dispatch([p](hipStream_t s) -> int {
    static bool done = false;
    if (int rc = set_fft_smem(reinterpret_cast<const void*>(fgla_analysis_kernel), &done)) return rc;
    hipLaunchKernelGGL(fgla_analysis_kernel, dim3(p.T, p.B), dim3(kFNT), kFft6400RegEntries * sizeof(cf), s, p);
    fgla_trace_report(1, "analysis");
    return check_launch("fgla_analysis");
  }, stream, "fgla_analysis");
}

extern "C" int ddx_fgla_iter(const float* audio, const float* window, const float* twiddle, float* u, int32_t u_stride, const float* mags,
                             int32_t mag_stride, float* frames, int32_t B, int32_t C, int32_t T, int32_t L, int32_t n_fft, int32_t hop,
                             float momentum, float t_lerp, int32_t final_pass, ddx_stream stream) {
  if (!audio || !window || !twiddle || !u || !mags || !frames || B <= 0 || (C != 1 && C != 2) || T <= 0 || L <= kFN / 2)
    return set_error(DDX_ERR_ARG, "fgla_iter: bad args");
  if (n_fft != kFN) return set_error(DDX_ERR_UNSUPPORTED, "fgla_iter: only n_fft = 6400 is built");
  if (u_stride < kFN / 2 + 2 || (u_stride & 1) || mag_stride < kFN / 2 + 2 || (mag_stride & 1))
    return set_error(DDX_ERR_ARG, "fgla_iter: row strides must be even and >= n_fft/2 + 2 (two bins per 16-byte access)");
  FglaIterParams p{audio, mags, window, reinterpret_cast<const float2*>(twiddle), reinterpret_cast<float2*>(u), frames, B, C, T, L, hop,
                   u_stride, mag_stride, momentum, t_lerp, final_pass, C == 2};
  return dispatch([p](hipStream_t s) -> int {
    static bool done = false;
    if (int rc = set_fft_smem(reinterpret_cast<const void*>(fgla_iter_kernel), &done)) return rc;
    hipLaunchKernelGGL(fgla_iter_kernel, dim3(p.T, p.B), dim3(kFNT), kFft6400RegEntries * sizeof(cf), s, p);
    return check_launch("fgla_iter");
  }, stream, "fgla_iter");
}

extern "C" int ddx_mel_to_amplitude(const float* mel, float* amp, int32_t rows, int32_t n_mel, int32_t T, float scale, float mean,
                                    float power, ddx_stream stream) {
  if (!mel || !amp || rows <= 0 || n_mel <= 0 || T <= 0 || scale == 0.f) return set_error(DDX_ERR_ARG, "mel_to_amplitude: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const size_t total = (size_t)rows * n_mel * T;
    hipLaunchKernelGGL(mel_to_amp_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65536)), dim3(256), 0, s, mel, amp, rows,
                       n_mel, T, 1.0f / scale, mean, power);
    return check_launch("mel_to_amplitude");
  }, stream, "mel_to_amplitude");
}
