// Batched bf16 GEMM on the matrix cores with either operand stored reduction-major or reduction-minor, plus the row
// softmax kernels: the building blocks of the attention BACKWARD pass (reference unet_edm2_b4.py:137-148 under autograd:
// scaled_dot_product_attention of q/k/v normalised per head).  With T <= a few hundred tokens per image (L3: 344,
// L4: 86) the T x T score matrices are small (15 MB per layer at B=4), so the backward is expressed as five plain GEMMs over
// materialised P / dS instead of a fused flash-style kernel:
//     S = Qn Kn^T                (A k-minor,  B k-minor)       P  = softmax(S / sqrt(d))                 [rows]
//     dP = dO Vn^T               (A k-minor,  B k-minor)       dS = P o (dP - rowsum(P o dP)) / sqrt(d)  [rows]
//     dVn = P^T dO, dKn = dS^T Qn (A k-MAJOR, B k-MAJOR)       dQn = dS Kn   (A k-minor, B k-MAJOR)
// C[b0][b1][M][N] = alpha * sum_k A(m,k) B(k,n); a "k-minor" operand is addressed row*ld + k (k contiguous), a "k-major"
// one k*ld + row.  k-major tiles are staged as they lie in memory and read with ds_read_b64_tr_b16 (see conv_wgrad.hip).
// One workgroup = 64 x 64 output tile, 4 waves of 32 x 32, K step 32.  Requirements: M, N, K and all leading dimensions
// and offsets multiples of 8 elements (16-byte global loads); M, N, K themselves are free when the contiguous axis of every
// operand is PADDED to a multiple of 8 with zeros (the attention path pads its T x T matrices to T' = roundup(T, 8)).
#include <algorithm>

#include "common.hpp"

namespace ddx {
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct BgemmParams {
  const bf16* A; const bf16* B; void* C;
  long lda, ldb, ldc;
  long sA0, sA1, sB0, sB1, sC0, sC1;  // batch strides (elements): batch index = b0 * nb1 + b1
  int M, N, K, nb1;
  int a_kmajor, b_kmajor, c_fp32;
  float alpha;
};

constexpr int kBT = 64, kBK = 32;
constexpr int kMinorStride = kBK * 2 + 16;   // bytes per row of a k-minor tile  [64 rows][32 k]
constexpr int kMajorStride = kBT * 2 + 16;   // bytes per row of a k-major tile  [32 k][64 rows]
constexpr int kTileBytes = 64 * kMinorStride > 32 * kMajorStride ? 64 * kMinorStride : 32 * kMajorStride;

__device__ __forceinline__ s16x4 tr_read4(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// stage one operand tile (64 rows x 32 k) into LDS, zero-filled outside the matrix
__device__ __forceinline__ void stage_tile(char* lds, const bf16* base, long ld, int kmajor, int row0, int k0, int rows, int K) {
  const int t = threadIdx.x;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (!kmajor) {
    const int r = t >> 2, kc = (t & 3) * 8;
    if (row0 + r < rows && k0 + kc < K) v = *reinterpret_cast<const u32x4*>(base + (long)(row0 + r) * ld + k0 + kc);
    *reinterpret_cast<u32x4*>(lds + r * kMinorStride + kc * 2) = v;
  } else {
    const int k = t >> 3, rc = (t & 7) * 8;
    if (k0 + k < K && row0 + rc < rows) v = *reinterpret_cast<const u32x4*>(base + (long)(k0 + k) * ld + row0 + rc);
    *reinterpret_cast<u32x4*>(lds + k * kMajorStride + rc * 2) = v;
  }
}

// MFMA fragment (32 rows x 16 k) of a staged tile: rows r0 .. r0+31, k-step ks (0/1)
__device__ __forceinline__ bf16x8 load_frag(const char* lds, int kmajor, int r0, int ks) {
  const int lane = threadIdx.x & 63;
  if (!kmajor) {
    const int l31 = lane & 31, khalf = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(lds + (r0 + l31) * kMinorStride + (ks * 16 + khalf * 8) * 2);
  }
  const int gq = lane >> 4, li = lane & 15;
  const char* p = lds + (ks * 16 + (gq >> 1) * 8 + (li >> 2)) * kMajorStride + (r0 + (gq & 1) * 16 + (li & 3) * 4) * 2;
  const s16x4 lo = tr_read4(p), hi = tr_read4(p + 4 * kMajorStride);
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256) void bgemm_kernel(const BgemmParams p) {
  __shared__ __attribute__((aligned(16))) char sA[kTileBytes];
  __shared__ __attribute__((aligned(16))) char sB[kTileBytes];
  const int batch = blockIdx.z, b0 = batch / p.nb1, b1 = batch - b0 * p.nb1;
  const bf16* A = p.A + b0 * p.sA0 + b1 * p.sA1;
  const bf16* B = p.B + b0 * p.sB0 + b1 * p.sB1;
  const int m0 = blockIdx.y * kBT, n0 = blockIdx.x * kBT;
  const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += kBK) {
    stage_tile(sA, A, p.lda, p.a_kmajor, m0, k0, p.M, p.K);
    stage_tile(sB, B, p.ldb, p.b_kmajor, n0, k0, p.N, p.K);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 af = load_frag(sA, p.a_kmajor, wm * 32, ks);
      const bf16x8 bfr = load_frag(sB, p.b_kmajor, wn * 32, ks);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout: acc[4q+e] = C[m = 8q + 4*khalf + e][n = lane & 31]
  const int lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
  const int n = n0 + wn * 32 + l31;
  if (n >= p.N) return;
  const long cbase = b0 * p.sC0 + b1 * p.sC1;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + wm * 32 + 8 * q + 4 * khalf + e;
      if (m >= p.M) continue;
      const float v = acc[4 * q + e] * p.alpha;
      if (p.c_fp32) reinterpret_cast<float*>(p.C)[cbase + (long)m * p.ldc + n] = v;
      else reinterpret_cast<bf16*>(p.C)[cbase + (long)m * p.ldc + n] = (bf16)v;
    }
}

// one wave per row: P = softmax(S * scale)  (S fp32 from the GEMM, P bf16 = operand of the next GEMMs)
template <typename TP>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, TP* __restrict__ p, long rows, int n, long ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * ld;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, (float)sr[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += __expf((float)sr[j] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < n; j += 64) p[row * ld + j] = from_f32<TP>(__expf((float)sr[j] * scale - mx) * inv);
}

// dS = P o (dP - sum_j P dP) * scale   (dP fp32 from the GEMM, dS bf16)
template <typename TP>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const TP* __restrict__ p, const float* __restrict__ dp, TP* __restrict__ ds, long rows,
                                                               int n, long ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) dot += to_f32<TP>(p[row * ld + j]) * dp[row * ld + j];
  dot = wave_sum(dot);
  for (int j = lane; j < n; j += 64) ds[row * ld + j] = from_f32<TP>(to_f32<TP>(p[row * ld + j]) * (dp[row * ld + j] - dot) * scale);
}

// float32 parity path of the batched GEMM (same descriptor, operands and result fp32): one thread per output element, plain fp32
// fused multiply-adds in k order.  Correctness tool for the fp32 backward pass, not a fast path.
__global__ __launch_bounds__(256) void bgemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, long lda,
                                                        long ldb, long ldc, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1, int M, int N, int K,
                                                        int nb1, int a_kmajor, int b_kmajor, float alpha) {
  const int bi = blockIdx.z, b0 = bi / nb1, b1 = bi - b0 * nb1;
  const float* a = A + b0 * sA0 + b1 * sA1;
  const float* b = B + b0 * sB0 + b1 * sB1;
  float* c = C + b0 * sC0 + b1 * sC1;
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const float av = a_kmajor ? a[(long)k * lda + m] : a[(long)m * lda + k];
      const float bv = b_kmajor ? b[(long)k * ldb + n] : b[(long)n * ldb + k];
      acc = fmaf(av, bv, acc);
    }
    c[(long)m * ldc + n] = alpha * acc;
  }
}

}  // namespace
}  // namespace ddx

using namespace ddx;

extern "C" int ddx_bgemm_bf16(const ddx_bgemm_desc* dp, ddx_stream stream) {
  if (!dp || !dp->A || !dp->B || !dp->C) return set_error(DDX_ERR_ARG, "bgemm: null");
  const ddx_bgemm_desc d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nb0 <= 0 || d.nb1 <= 0) return set_error(DDX_ERR_ARG, "bgemm: bad size");
  const long chk[] = {d.lda, d.ldb, d.sA0, d.sA1, d.sB0, d.sB1};
  for (long v : chk)
    if (v % 8) return set_error(DDX_ERR_UNSUPPORTED, "bgemm: leading dimensions and batch strides must be multiples of 8 elements");
  BgemmParams p{};
  p.A = (const bf16*)d.A; p.B = (const bf16*)d.B; p.C = d.C;
  p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.sA0 = d.sA0; p.sA1 = d.sA1; p.sB0 = d.sB0; p.sB1 = d.sB1; p.sC0 = d.sC0; p.sC1 = d.sC1;
  p.M = d.M; p.N = d.N; p.K = d.K; p.nb1 = d.nb1;
  p.a_kmajor = d.a_kmajor; p.b_kmajor = d.b_kmajor; p.c_fp32 = d.c_fp32; p.alpha = d.alpha;
  const int nb = d.nb0 * d.nb1;
  return dispatch([p, nb](hipStream_t s) -> int {
    dim3 grid(ceil_div(p.N, kBT), ceil_div(p.M, kBT), nb);
    hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, s, p);
    return check_launch("bgemm");
  }, stream, "bgemm", 2.0 * nb * (double)d.M * d.N * d.K, 0.0);
}

extern "C" int ddx_softmax_rows(const void* s, void* p, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream) {
  if (!s || !p || rows <= 0 || n <= 0) return set_error(DDX_ERR_ARG, "softmax_rows: bad args");
  return dispatch([=](hipStream_t st) -> int {
    hipLaunchKernelGGL(softmax_rows_kernel<bf16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)s, (bf16*)p, (long)rows, n, (long)ld, scale);
    return check_launch("softmax_rows");
  }, stream, "softmax_rows");
}

extern "C" int ddx_softmax_rows_f32(const void* s, void* p, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream) {
  if (!s || !p || rows <= 0 || n <= 0) return set_error(DDX_ERR_ARG, "softmax_rows: bad args");
  return dispatch([=](hipStream_t st) -> int {
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)s, (float*)p, (long)rows, n, (long)ld, scale);
    return check_launch("softmax_rows_f32");
  }, stream, "softmax_rows");
}

extern "C" int ddx_softmax_bwd_rows_f32(const void* p, const void* dp, void* ds, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream) {
  if (!p || !dp || !ds || rows <= 0 || n <= 0) return set_error(DDX_ERR_ARG, "softmax_bwd_rows: bad args");
  return dispatch([=](hipStream_t st) -> int {
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)p, (const float*)dp, (float*)ds, (long)rows, n,
                       (long)ld, scale);
    return check_launch("softmax_bwd_rows_f32");
  }, stream, "softmax_bwd_rows");
}

extern "C" int ddx_bgemm_f32(const ddx_bgemm_desc* dp, ddx_stream stream) {
  if (!dp || !dp->A || !dp->B || !dp->C) return set_error(DDX_ERR_ARG, "bgemm: null");
  const ddx_bgemm_desc d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nb0 <= 0 || d.nb1 <= 0 || (long)d.nb0 * d.nb1 > 65535) return set_error(DDX_ERR_ARG, "bgemm: bad size");
  return dispatch([d](hipStream_t s) -> int {
    dim3 grid((unsigned)std::min<long>(((long)d.M * d.N + 255) / 256, 4096), 1, (unsigned)(d.nb0 * d.nb1));
    hipLaunchKernelGGL(bgemm_f32_kernel, grid, dim3(256), 0, s, (const float*)d.A, (const float*)d.B, (float*)d.C, d.lda, d.ldb, d.ldc, d.sA0, d.sA1, d.sB0, d.sB1,
                       d.sC0, d.sC1, d.M, d.N, d.K, d.nb1, d.a_kmajor, d.b_kmajor, d.alpha);
    return check_launch("bgemm_f32");
  }, stream, "bgemm_f32");
}

extern "C" int ddx_softmax_bwd_rows(const void* p, const void* dp, void* ds, int64_t rows, int32_t n, int64_t ld, float scale, ddx_stream stream) {
  if (!p || !dp || !ds || rows <= 0 || n <= 0) return set_error(DDX_ERR_ARG, "softmax_bwd_rows: bad args");
  return dispatch([=](hipStream_t st) -> int {
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<bf16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16*)p, (const float*)dp, (bf16*)ds, (long)rows, n, (long)ld, scale);
    return check_launch("softmax_bwd_rows");
  }, stream, "softmax_bwd_rows");
}
