// Shared device/host helpers for libddx_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "../../include/ddx_hip.h"

namespace ddx {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float kMpSiluInv = 1.0f / 0.596f;

// ---- dispatch: launch now, or record into the thread-local plan (plan.cpp)
using LaunchFn = std::function<int(hipStream_t)>;
int dispatch(LaunchFn&& fn, ddx_stream stream, const char* tag = "op", double flops = 0.0, double bytes = 0.0);
int set_error(int code, const char* msg);
int check_launch(const char* what);
// Zero `bytes` bytes (a multiple of 2) at a 4-byte aligned address with a KERNEL.  Not hipMemsetAsync: a memset node of a captured hipGraph
// was seen to leave the low word of every 8 bytes untouched on replay (32-byte workspace of the loss, round 4) -- stale sums of squares, garbage
// loss and logvar gradient from the second replay on, depending on which tensor had owned the block before.
int zero_bytes(void* p, size_t bytes, hipStream_t s);

// ---- scalar helpers
// mp_silu(x) = x * sigmoid(x) / 0.596 with v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence:
// 6 VALU instructions per element instead of ~16 (the fused conv prologue runs this on every staged activation).
__device__ __forceinline__ float mp_silu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
  return (x * kMpSiluInv) * __builtin_amdgcn_rcpf(1.0f + e);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---- 16-byte vectors of T: 4 x f32 or 8 x bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

// 4 consecutive elements (MFMA epilogue granularity)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec4<bf16> {
  bf16x4 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for 256-thread blocks (all threads get the result)
__device__ __forceinline__ float block_sum_256(float v, float* scratch4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch4[w] = v;
  __syncthreads();
  return scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline size_t dtype_size(int dt) { return dt == DDX_BF16 ? 2 : 4; }

}  // namespace ddx
