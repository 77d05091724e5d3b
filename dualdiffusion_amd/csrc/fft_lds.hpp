// Mixed-radix (2, 4, 5) Stockham FFT of one complex sequence held in LDS, executed by a whole workgroup.
//
// Used by the mel-STFT and FGLA kernels: n_fft = 6400 = 4^4 * 5^2 (reference formats/old/spectrogram.py:116-128 through
// torch.stft) does not fit a pure radix-2 scheme.  Decimation-in-frequency Stockham autosort: stage (n, s, r) reads
//   a_j = x[q + s*(p + m*j)],  m = n/r,  p < m, q < s,  j < r
// and writes  y[q + s*(r*p + k)] = (sum_j a_j * W_r^{jk}) * W_n^{pk},  ping-ponging between two LDS buffers, one barrier
// per stage, natural-order output.  Twiddles W_N^t come from a table computed in double on the host (N entries).
#pragma once
#include "common.hpp"

namespace ddx {

struct cf { float x, y; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cconj(cf a) { return {a.x, -a.y}; }
// multiply by -i (forward) / +i (inverse)
template <bool INV> __device__ __forceinline__ cf cmul_mi(cf a) { return INV ? cf{-a.y, a.x} : cf{a.y, -a.x}; }

template <bool INV> __device__ __forceinline__ cf twiddle(const float2* __restrict__ tw, int idx) {
  const float2 t = tw[idx];
  return INV ? cf{t.x, -t.y} : cf{t.x, t.y};
}

// one butterfly of radix R (in registers): b[k] = sum_j a[j] * W_R^{jk}
template <int R, bool INV> struct Butterfly;
template <bool INV> struct Butterfly<2, INV> {
  static __device__ __forceinline__ void run(cf* a) {
    const cf t = a[0];
    a[0] = cadd(t, a[1]);
    a[1] = csub(t, a[1]);
  }
};
template <bool INV> struct Butterfly<4, INV> {
  static __device__ __forceinline__ void run(cf* a) {
    const cf s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
    const cf s13 = cadd(a[1], a[3]), d13 = cmul_mi<INV>(csub(a[1], a[3]));
    a[0] = cadd(s02, s13);
    a[1] = cadd(d02, d13);
    a[2] = csub(s02, s13);
    a[3] = csub(d02, d13);
  }
};
template <bool INV> struct Butterfly<5, INV> {
  static __device__ __forceinline__ void run(cf* a) {
    // W_5 = exp(-2 pi i / 5): c1 = cos(2pi/5), c2 = cos(4pi/5), s1 = sin(2pi/5), s2 = sin(4pi/5)
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const cf t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
    const cf u1 = csub(a[1], a[4]), u2 = csub(a[2], a[3]);
    const cf a0 = a[0];
    const cf m1 = {a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y};
    const cf m2 = {a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y};
    // -i * (s1*u1 + s2*u2) for the forward transform, +i for the inverse
    const cf v1 = cmul_mi<INV>(cf{s1 * u1.x + s2 * u2.x, s1 * u1.y + s2 * u2.y});
    const cf v2 = cmul_mi<INV>(cf{s2 * u1.x - s1 * u2.x, s2 * u1.y - s1 * u2.y});
    a[0] = cadd(a0, cadd(t1, t2));
    a[1] = cadd(m1, v1);
    a[4] = csub(m1, v1);
    a[2] = cadd(m2, v2);
    a[3] = csub(m2, v2);
  }
};

// one Stockham stage over the whole sequence (N points), sub-length n, stride s, radix R: x -> y
template <int N, int R, bool INV, int NT>
__device__ __forceinline__ void fft_stage(const cf* __restrict__ x, cf* __restrict__ y, int n, int s, const float2* __restrict__ tw) {
  const int m = n / R;
  const int tstep = N / n;  // W_n^{t} = W_N^{t * N/n}
  for (int idx = threadIdx.x; idx < N / R; idx += NT) {
    const int p = idx / s, q = idx - p * s;
    cf a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = x[q + s * (p + m * j)];
    Butterfly<R, INV>::run(a);
    y[q + s * (R * p)] = a[0];
    // (twiddle index p*k*tstep <= (n/R - 1)(R - 1) N/n < N: no reduction modulo N needed)
#pragma unroll
    for (int k = 1; k < R; ++k) y[q + s * (R * p + k)] = cmul(a[k], twiddle<INV>(tw, p * k * tstep));
  }
}

// Full transform of N = 6400 points (radices 4,4,4,4,5,5).  Input in `a`, result (natural order) in `a`; `b` is scratch.
// Every thread of the NT-thread workgroup must call it; ends with a barrier.
template <bool INV, int NT>
__device__ __forceinline__ void fft6400(cf* a, cf* b, const float2* __restrict__ tw) {
  constexpr int N = 6400;
  __syncthreads();
  fft_stage<N, 4, INV, NT>(a, b, 6400, 1, tw);   __syncthreads();
  fft_stage<N, 4, INV, NT>(b, a, 1600, 4, tw);   __syncthreads();
  fft_stage<N, 4, INV, NT>(a, b, 400, 16, tw);   __syncthreads();
  fft_stage<N, 4, INV, NT>(b, a, 100, 64, tw);   __syncthreads();
  fft_stage<N, 5, INV, NT>(a, b, 25, 256, tw);   __syncthreads();
  fft_stage<N, 5, INV, NT>(b, a, 5, 1280, tw);   __syncthreads();
}

// ---- single-buffer variant: every thread first pulls ALL its butterfly inputs of the stage into registers, barrier, then
// writes the outputs over the same array.  Two barriers per stage instead of one, but half the LDS (51 KB for 6400 points),
// so three workgroups fit a CU and the global-memory phases of one frame overlap the transform of another.
template <int N, int R, bool INV, int NT>
__device__ __forceinline__ void fft_stage_inplace(cf* __restrict__ x, int n, int s, const float2* __restrict__ tw, int tid) {
  const int m = n / R;
  const int tstep = N / n;
  constexpr int ROUNDS = (N / R + NT - 1) / NT;
  cf a[ROUNDS][R];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int idx = tid + r * NT;
    if (idx < N / R) {
      const int p = idx / s, q = idx - p * s;
#pragma unroll
      for (int j = 0; j < R; ++j) a[r][j] = x[q + s * (p + m * j)];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int idx = tid + r * NT;
    if (idx < N / R) {
      const int p = idx / s, q = idx - p * s;
      Butterfly<R, INV>::run(a[r]);
      x[q + s * (R * p)] = a[r][0];
      // one table read per butterfly: W^{pk} = (W^p)^k by repeated multiplication (k <= 4: ~3 ulp); the table sits in L2 and
      // every read is a dependent round trip inside a barrier-separated stage (FGLA 1.87 -> 1.81 ms per iteration)
      const cf w1 = twiddle<INV>(tw, p * tstep);
      cf wk = w1;
#pragma unroll
      for (int k = 1; k < R; ++k) {
        x[q + s * (R * p + k)] = cmul(a[r][k], wk);
        wk = cmul(wk, w1);
      }
    }
  }
  __syncthreads();
}

// `tid` = threadIdx.x; kernels that call this inside a loop over frames pass a value laundered through an empty asm so that
// the compiler does not hoist the (loop-invariant) index arithmetic of all six stages out of that loop and keep it in
// ~100 registers (which costs the occupancy the single buffer was bought for).
template <bool INV, int NT>
__device__ __forceinline__ void fft6400_inplace(cf* a, const float2* __restrict__ tw, int tid) {
  constexpr int N = 6400;
  __syncthreads();
  fft_stage_inplace<N, 4, INV, NT>(a, 6400, 1, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 1600, 4, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 400, 16, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 100, 64, tw, tid);
  fft_stage_inplace<N, 5, INV, NT>(a, 25, 256, tw, tid);
  fft_stage_inplace<N, 5, INV, NT>(a, 5, 1280, tw, tid);
}
// N = 4096 = 4^6 (the 128 ms frames of MS_MDCT_DualFormat's mel spectrogram, reference formats/ms_mdct_dual.py:110-139)
template <bool INV, int NT>
__device__ __forceinline__ void fft4096_inplace(cf* a, const float2* __restrict__ tw, int tid) {
  constexpr int N = 4096;
  __syncthreads();
  fft_stage_inplace<N, 4, INV, NT>(a, 4096, 1, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 1024, 4, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 256, 16, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 64, 64, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 16, 256, tw, tid);
  fft_stage_inplace<N, 4, INV, NT>(a, 4, 1024, tw, tid);
}
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }

}  // namespace ddx
