// Complex FFTs of one sequence held in LDS, executed by a whole workgroup: n_fft = 6400 = 16 x 16 x 25 (mel-STFT and FGLA: reference
// formats/old/spectrogram.py:116-128 through torch.stft) and 4096 = 16 x 16 x 16 (the dual-window mel spectrogram), as three passes of
// 16- / 25-point transforms that ONE thread runs in registers (radix-4 / radix-5 butterflies below).  Rounds 1-3 ran staged Stockham
// transforms (a radix-4 / 5 butterfly per thread between two barriers, twelve barriers per frame, strided writes serialising on the LDS
// banks): 24-31 k cycles per 6400-point frame against 11-12 k here.  Twiddles W_N^t of the inter-pass factors come from
// a table computed in double on the host (N entries); the inner ones are literals.
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

// ---- register variant: 6400 = 16 x 16 x 25 in three passes, every 16- / 25-point transform by ONE thread in registers.
// Index split (n = 400 a + r, k = k1 + 16 k2 and again
// inside the 400-point transforms: r = 25 b + c, k2 = k2a + 16 k2b); every access below has consecutive lanes on consecutive (or odd-stride)
// entries:
//   pass 1: 400 lines r (lane = r): 16 points a[400 j + r]; out[k1] * W_6400^(r k1) to I1 = a[r + 401 k1] (the skew by one entry per k1 makes the
//           stride between k1-neighbours odd for pass 2; read all, barrier, write)
//   pass 2: 400 lines (c, k1), lane = 16 c + k1: 16 points I1[25 b + c + 401 k1]; out[k2a] * W_400^(c k2a) to I2 = a[k1 + 16 c + 400 k2a]
//   pass 3: 256 lines (k2a, k1), lane = 16 k2a + k1: 25 points I2[k1 + 16 c + 400 k2a]; out[k2b] to a[k1 + 16 k2a + 256 k2b] = natural order
// Seven barriers instead of thirteen, three LDS round trips instead of six.  The inter-pass twiddles are one table read per line and its powers
// (mul_powers16); the inner ones are literals.  The buffer needs 6415 entries (the skew).
constexpr int kFft6400RegEntries = 6416;
__device__ __constant__ const float kW16c[10] = {1.0000000000e+00f, 9.2387953251e-01f, 7.0710678119e-01f, 3.8268343237e-01f, 0.0000000000e+00f, -3.8268343237e-01f, -7.0710678119e-01f, -9.2387953251e-01f, -1.0000000000e+00f, -9.2387953251e-01f};
__device__ __constant__ const float kW16s[10] = {0.0000000000e+00f, 3.8268343237e-01f, 7.0710678119e-01f, 9.2387953251e-01f, 1.0000000000e+00f, 9.2387953251e-01f, 7.0710678119e-01f, 3.8268343237e-01f, 0.0000000000e+00f, -3.8268343237e-01f};
__device__ __constant__ const float kW25c[17] = {1.0000000000e+00f, 9.6858316113e-01f, 8.7630668004e-01f, 7.2896862742e-01f, 5.3582679498e-01f, 3.0901699437e-01f, 6.2790519529e-02f, -1.8738131459e-01f, -4.2577929157e-01f, -6.3742398975e-01f, -8.0901699437e-01f, -9.2977648589e-01f, -9.9211470131e-01f, -9.9211470131e-01f, -9.2977648589e-01f, -8.0901699437e-01f, -6.3742398975e-01f};
__device__ __constant__ const float kW25s[17] = {0.0000000000e+00f, 2.4868988716e-01f, 4.8175367410e-01f, 6.8454710593e-01f, 8.4432792550e-01f, 9.5105651630e-01f, 9.9802672843e-01f, 9.8228725073e-01f, 9.0482705247e-01f, 7.7051324278e-01f, 5.8778525229e-01f, 3.6812455268e-01f, 1.2533323356e-01f, -1.2533323356e-01f, -3.6812455268e-01f, -5.8778525229e-01f, -7.7051324278e-01f};

template <int R, bool INV>
__device__ __forceinline__ void fft_rr_reg(cf (&x)[R * R], const float* __restrict__ wc, const float* __restrict__ ws) {
  // n = R a + r, k = k1 + R k2:  X[k1 + R k2] = sum_r W_R^(r k2) [ W_(R R)^(r k1) sum_a x[R a + r] W_R^(a k1) ]
  cf y[R][R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cf t[R];
#pragma unroll
    for (int a = 0; a < R; ++a) t[a] = x[R * a + r];
    Butterfly<R, INV>::run(t);
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) y[r][k1] = (r * k1 == 0) ? t[k1] : cmul(t[k1], cf{wc[r * k1], INV ? ws[r * k1] : -ws[r * k1]});
  }
#pragma unroll
  for (int k1 = 0; k1 < R; ++k1) {
    cf t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = y[r][k1];
    Butterfly<R, INV>::run(t);
#pragma unroll
    for (int k2 = 0; k2 < R; ++k2) x[k1 + R * k2] = t[k2];
  }
}

// x[k] *= w^k for k = 1..15, the powers by binary products (w^2, w^4, w^8 by squaring, the rest one product of two of those: at most four
// multiplications from the table value) -- ONE table read per line instead of fifteen
__device__ __forceinline__ void mul_powers16(cf (&x)[16], cf w1) {
  const cf w2 = cmul(w1, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
  const cf w3 = cmul(w2, w1), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
  x[1] = cmul(x[1], w1); x[2] = cmul(x[2], w2); x[3] = cmul(x[3], w3); x[4] = cmul(x[4], w4);
  x[5] = cmul(x[5], w5); x[6] = cmul(x[6], w6); x[7] = cmul(x[7], w7); x[8] = cmul(x[8], w8);
  x[9] = cmul(x[9], cmul(w8, w1)); x[10] = cmul(x[10], cmul(w8, w2)); x[11] = cmul(x[11], cmul(w8, w3));
  x[12] = cmul(x[12], cmul(w8, w4)); x[13] = cmul(x[13], cmul(w8, w5)); x[14] = cmul(x[14], cmul(w8, w6));
  x[15] = cmul(x[15], cmul(w8, w7));
}

template <bool INV, int NT>
__device__ __forceinline__ void fft6400_reg(cf* a, const float2* __restrict__ tw, int tid) {
  static_assert(NT >= 256, "pass 3: one line per thread");
  constexpr int ROUNDS = (400 + NT - 1) / NT;      // lines per thread in passes 1 and 2 (512 threads: 1, 256: 2)
  __syncthreads();
  {                                                                // ---- pass 1
    cf x[ROUNDS][16];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int r = tid + rd * NT;
      if (r < 400) {
        const cf w1 = twiddle<INV>(tw, r);
#pragma unroll
        for (int j = 0; j < 16; ++j) x[rd][j] = a[400 * j + r];
        fft_rr_reg<4, INV>(x[rd], kW16c, kW16s);
        mul_powers16(x[rd], w1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int r = tid + rd * NT;
      if (r < 400) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[r + 401 * k] = x[rd][k];
      }
    }
  }
  __syncthreads();
  {                                                                // ---- pass 2
    cf x[ROUNDS][16];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int L = tid + rd * NT, c = L >> 4, k1 = L & 15;
      if (L < 400) {
        const cf w1 = twiddle<INV>(tw, 16 * c);
#pragma unroll
        for (int b = 0; b < 16; ++b) x[rd][b] = a[25 * b + c + 401 * k1];
        fft_rr_reg<4, INV>(x[rd], kW16c, kW16s);
        mul_powers16(x[rd], w1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int L = tid + rd * NT, c = L >> 4, k1 = L & 15;
      if (L < 400) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k1 + 16 * c + 400 * k] = x[rd][k];
      }
    }
  }
  __syncthreads();
  {                                                                // ---- pass 3
    cf x[25];
    const int k1 = tid & 15, k2a = tid >> 4;
    if (tid < 256) {
#pragma unroll
      for (int c = 0; c < 25; ++c) x[c] = a[k1 + 16 * c + 400 * k2a];
      fft_rr_reg<5, INV>(x, kW25c, kW25s);
    }
    __syncthreads();
    if (tid < 256) {
#pragma unroll
      for (int k = 0; k < 25; ++k) a[tid + 256 * k] = x[k];
    }
  }
  __syncthreads();
}

// 4096 = 16 x 16 x 16 the same way (256 lines per pass, 256 threads or more): n = 256 a + r, k = k1 + 16 k2, r = 16 b + c, k2 = k2a + 16 k2b;
// I1 = a[r + 257 k1] (odd k1-stride), I2 = a[k1 + 16 c + 272 k2a] (272 = 16 mod 32: the two k2a rows of a half-wave fall on different banks),
// natural order out.  The buffer needs 4336 entries.
constexpr int kFft4096RegEntries = 4336;
template <bool INV, int NT>
__device__ __forceinline__ void fft4096_reg(cf* a, const float2* __restrict__ tw, int tid) {
  static_assert(NT >= 256, "one line per thread");
  __syncthreads();
  cf x[16];
  const bool act = tid < 256;
  if (act) {                                                       // ---- pass 1 (lane = r)
    const cf w1 = twiddle<INV>(tw, tid);
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = a[256 * j + tid];
    fft_rr_reg<4, INV>(x, kW16c, kW16s);
    mul_powers16(x, w1);
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[tid + 257 * k] = x[k];
  }
  __syncthreads();
  const int hi = tid >> 4, k1 = tid & 15;
  if (act) {                                                       // ---- pass 2 (lane = 16 c + k1)
    const cf w1 = twiddle<INV>(tw, 16 * hi);
#pragma unroll
    for (int b = 0; b < 16; ++b) x[b] = a[16 * b + hi + 257 * k1];
    fft_rr_reg<4, INV>(x, kW16c, kW16s);
    mul_powers16(x, w1);
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k1 + 16 * hi + 272 * k] = x[k];
  }
  __syncthreads();
  if (act) {                                                       // ---- pass 3 (lane = 16 k2a + k1)
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = a[k1 + 16 * c + 272 * hi];
    fft_rr_reg<4, INV>(x, kW16c, kW16s);
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[tid + 256 * k] = x[k];
  }
  __syncthreads();
}

// (kernels that call these inside a loop over frames pass a thread index laundered through an empty asm so that the compiler does not hoist
// the loop-invariant index arithmetic out of that loop and keep it in registers)
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }

}  // namespace ddx
