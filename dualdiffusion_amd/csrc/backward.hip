// Element-wise / row-wise backward kernels of the EDM2 block (the pieces autograd runs between the conv gradients of
// reference src/modules/unets/unet_edm2_b4.py:110-158 and src/modules/mp_tools.py:42-49,268-279,359-364).  All HBM-bound:
// each reads its operands once in 16-byte lanes and writes the gradient(s) once.
//   silu_scale_bwd  : a = mp_silu(y * c[b][ch] * s)           -> dy, dc          (conv_res1 / attn_proj operands, block inputs)
//   mpsum_clip_bwd  : out = clip(a*res + b*y)                 -> dres, dy        (mp_sum + clip_ at the end of a block)
//   pixelnorm_bwd   : y = x / (eps + |x| / sqrt(C))           -> dx              (normalize(x, dim=1) of encoder blocks)
//   wprep_bwd       : w' = normalize(w) * gain / sqrt(fan)    -> dw, dgain       (MPConv weight path, fp32 master weights)
#include <algorithm>

#include "common.hpp"
#include "wpath_rows.hpp"

namespace ddx {
namespace {

__device__ __forceinline__ float mp_silu_grad_f(float z) {
  // d/dz [ z * sigmoid(z) / 0.596 ] = sigmoid(z) * (1 + z * (1 - sigmoid(z))) / 0.596
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
  return sg * (1.0f + z * (1.0f - sg)) * kMpSiluInv;
}

// one workgroup = RPW rows-per-pass x nvec 16-byte vectors; each thread keeps the dc partial sums of its vector
template <typename T>
__global__ __launch_bounds__(256) void silu_scale_bwd_kernel(const T* __restrict__ da, int da_ld, const T* __restrict__ y, const float* __restrict__ cs,
                                                             float scale, const T* __restrict__ add, int add_ld, T* __restrict__ dy,
                                                             float* __restrict__ dc, int HW, int C, int rows_per_block, int act) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int vbase = blockIdx.z * 256;                 // rows wider than 256 vectors are split over blockIdx.z
  const int nvec = min(C / EV - vbase, 256);
  const int b = blockIdx.y;
  const int rsub = threadIdx.x / nvec, rstep = 256 / nvec;
  const int v = vbase + threadIdx.x % nvec;
  const bool idle = rsub >= rstep;  // (256 not a multiple of nvec: idle tail threads; they still reach the barrier below)
  float cv[EV], acc[EV];
#pragma unroll
  for (int e = 0; e < EV; ++e) { cv[e] = (cs ? cs[(size_t)b * C + v * EV + e] : 1.0f) * scale; acc[e] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block, r1 = idle ? 0 : min(r0 + rows_per_block, HW);
  for (int r = r0 + rsub; r < r1; r += rstep) {
    const size_t row = (size_t)b * HW + r;
    const size_t o = row * C + (size_t)v * EV;
    Vec16<T> g, yy, out, ad;
    g.v = *reinterpret_cast<const decltype(g.v)*>(da + row * da_ld + (size_t)v * EV);
    yy.v = *reinterpret_cast<const decltype(g.v)*>(y + o);
    if (add) ad.v = *reinterpret_cast<const decltype(g.v)*>(add + row * add_ld + (size_t)v * EV);
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const float yv = yy.get(e);
      const float dz = act ? g.get(e) * mp_silu_grad_f(yv * cv[e]) : g.get(e);
      out.set(e, dz * cv[e] + (add ? ad.get(e) : 0.f));
      acc[e] += dz * yv;
    }
    *reinterpret_cast<decltype(g.v)*>(dy + o) = out.v;
  }
  if (dc) {
    // one atomic per (block, channel): the row groups of the block are summed through LDS first (the [B][C] targets are few,
    // so every atomic that is not issued is contention saved)
    __shared__ float red[256 * 8];
#pragma unroll
    for (int e = 0; e < EV; ++e) red[threadIdx.x * EV + e] = acc[e];
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
      for (int e = 0; e < EV; ++e) {
        float sum = 0.f;
        for (int r = 0; r < rstep; ++r) sum += red[(r * nvec + (v - vbase)) * EV + e];
        atomicAdd(dc + (size_t)b * C + v * EV + e, sum * scale);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_scale_fwd_kernel(const T* __restrict__ x, const float* __restrict__ cs, float scale, T* __restrict__ out,
                                                             size_t rows_per_b, int C, size_t nvec_total, int act) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int nvec = C / EV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec_total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / nvec;
    const int v = (int)(i - row * nvec);
    const size_t b = row / rows_per_b;
    Vec16<T> xv, o;
    xv.v = *reinterpret_cast<const decltype(xv.v)*>(x + i * EV);
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const float z = xv.get(e) * (cs ? cs[b * C + v * EV + e] : 1.0f) * scale;
      o.set(e, act ? mp_silu_f(z) : z);
    }
    *reinterpret_cast<decltype(xv.v)*>(out + i * EV) = o.v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ out, size_t nvec) {
  constexpr int EV = 16 / (int)sizeof(T);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    Vec16<T> x, y, z, o;
    x.v = *reinterpret_cast<const decltype(x.v)*>(a + i * EV);
    y.v = *reinterpret_cast<const decltype(x.v)*>(b + i * EV);
    if (c) z.v = *reinterpret_cast<const decltype(x.v)*>(c + i * EV);
#pragma unroll
    for (int e = 0; e < EV; ++e) o.set(e, x.get(e) + y.get(e) + (c ? z.get(e) : 0.f));
    *reinterpret_cast<decltype(x.v)*>(out + i * EV) = o.v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mpsum_clip_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, T* __restrict__ dres,
                                                             T* __restrict__ dy, float a, float b, float clip, size_t nvec) {
  constexpr int EV = 16 / (int)sizeof(T);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    Vec16<T> g, o, r, yv;
    g.v = *reinterpret_cast<const decltype(g.v)*>(dout + i * EV);
    if (clip > 0.f) o.v = *reinterpret_cast<const decltype(g.v)*>(out + i * EV);
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const float m = (clip > 0.f && fabsf(o.get(e)) >= clip) ? 0.f : g.get(e);
      r.set(e, a * m);
      yv.set(e, b * m);
    }
    if (dres) *reinterpret_cast<decltype(g.v)*>(dres + i * EV) = r.v;
    *reinterpret_cast<decltype(g.v)*>(dy + i * EV) = yv.v;
  }
}

// one wave per row (as pixelnorm_kernel); C <= 64 * 4 * EV cached in registers, larger rows re-read
template <typename T>
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int64_t rows,
                                                            int C, float eps) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_sqrt_c = rsqrtf((float)C);
  const int nvec = C / EV;
  for (int64_t r = wave_id; r < rows; r += nwaves) {
    const T* xr = x + r * C; const T* gr = dy + r * C; T* or_ = dx + r * C;
    float ss = 0.f, sd = 0.f;
    for (int vi = lane; vi < nvec; vi += 64) {
      Vec16<T> xv, gv;
      xv.v = *reinterpret_cast<const decltype(xv.v)*>(xr + (size_t)vi * EV);
      gv.v = *reinterpret_cast<const decltype(xv.v)*>(gr + (size_t)vi * EV);
#pragma unroll
      for (int e = 0; e < EV; ++e) { const float f = xv.get(e); ss += f * f; sd += f * gv.get(e); }
    }
    ss = wave_sum(ss); sd = wave_sum(sd);
    const float n = sqrtf(ss);
    const float nu = eps + n * inv_sqrt_c;
    const float k = n > 0.f ? sd * inv_sqrt_c / (nu * nu * n) : 0.f;
    const float inv_nu = 1.0f / nu;
    for (int vi = lane; vi < nvec; vi += 64) {
      Vec16<T> xv, gv, o;
      xv.v = *reinterpret_cast<const decltype(xv.v)*>(xr + (size_t)vi * EV);
      gv.v = *reinterpret_cast<const decltype(xv.v)*>(gr + (size_t)vi * EV);
#pragma unroll
      for (int e = 0; e < EV; ++e) o.set(e, gv.get(e) * inv_nu - xv.get(e) * k);
      *reinterpret_cast<decltype(xv.v)*>(or_ + (size_t)vi * EV) = o.v;
    }
  }
}

// one workgroup per DESTINATION row of the prepared weight (same row mapping as wprep_kernel; body in wpath_rows.hpp)
template <typename TW_>
__global__ __launch_bounds__(256) void wprep_bwd_kernel(const float* __restrict__ dwp, const TW_* __restrict__ w, const float* gain_ptr, float gain,
                                                        float* __restrict__ dw, float* __restrict__ dgain, int Cout, int Cg, int taps, int G,
                                                        int normalize, int qk_d, float eps, int in_split, float in_s0, float in_s1,
                                                        int accumulate) {
  __shared__ float scratch[4];
  wprep_bwd_row<TW_>(dwp, w, gain_ptr, gain, dw, dgain, Cout, Cg, taps, G, normalize, qk_d, eps, in_split, in_s0, in_s1, accumulate, blockIdx.x,
                     scratch);
}

// Backward of the small-M linear layers (emb_linear*: c = 1 + x @ w'^T at M = batch, reference unet_edm2_b4.py:121 through
// mp_tools.py:366-367): one workgroup per output row o; dwp[o][k] = sum_m dc[m][o] x[m][g*Kg+k]; dx[m][g*Kg+k] += dc[m][o] w'[o][k]
template <typename TW_>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ dc, const float* __restrict__ x, const TW_* __restrict__ w,
                                                               const float* __restrict__ row_scale, float* __restrict__ dwp, float* __restrict__ dx,
                                                               int M, int O, int Kg, int groups, int x_stride) {
  const int o = blockIdx.x;
  const int g = o / (O / groups);
  for (int k = threadIdx.x; k < Kg; k += 256) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m) {
      const float d = dc[(size_t)m * O + o];
      acc += d * x[(size_t)m * x_stride + g * Kg + k];
    }
    dwp[(size_t)o * Kg + k] = acc;
  }
}

// input gradient of the same layer: one thread per (m, input column, slice of the group's output rows)
// (consecutive threads read consecutive weights of a row)
template <typename TW_>
__global__ __launch_bounds__(256) void linear_small_bwd_dx_kernel(const float* __restrict__ dc, const TW_* __restrict__ w, const float* __restrict__ row_scale,
                                                                  float* __restrict__ dx, int M, int O, int Kg, int groups, int x_stride) {
  const int kk = blockIdx.x * 256 + threadIdx.x;  // column of x
  const int m = blockIdx.y;
  if (kk >= Kg * groups) return;
  const int g = kk / Kg, k = kk - g * Kg, Og = O / groups;
  // the group's output rows are split over blockIdx.z: 16 partial sums per element meet in one atomic each
  const int per = (Og + gridDim.z - 1) / gridDim.z;
  const int o0 = g * Og + blockIdx.z * per, o1 = min(o0 + per, (g + 1) * Og);
  float acc = 0.f;
  for (int o = o0; o < o1; ++o) acc += dc[(size_t)m * O + o] * row_scale[o] * to_f32<TW_>(w[(size_t)o * Kg + k]);
  atomicAdd(dx + (size_t)m * x_stride + kk, acc);
}

// Batched forms over a device job table (all emb_linear* layers of a UNet share x = emb and dx = demb): blockIdx.y = job.
__global__ __launch_bounds__(256) void linear_small_bwd_multi_kernel(const ddx_linear_bwd_job* __restrict__ jobs, const float* __restrict__ x,
                                                                     int M, int K, int x_stride) {
  const ddx_linear_bwd_job jb = jobs[blockIdx.y];
  const int o = blockIdx.x;
  if (o >= jb.O) return;
  const int Kg = K / jb.groups;
  const int g = o / (jb.O / jb.groups);
  for (int k = threadIdx.x; k < Kg; k += 256) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += jb.dc[(size_t)m * jb.O + o] * x[(size_t)m * x_stride + g * Kg + k];
    jb.dwp[(size_t)o * Kg + k] = acc;
  }
}

__global__ __launch_bounds__(256) void linear_small_bwd_dx_multi_kernel(const ddx_linear_bwd_job* __restrict__ jobs, float* __restrict__ dx, int M, int K,
                                                                        int x_stride, int slices) {
  const ddx_linear_bwd_job jb = jobs[blockIdx.z / slices];
  const int slice = blockIdx.z % slices;
  const int kk = blockIdx.x * 256 + threadIdx.x;  // column of x
  const int m = blockIdx.y;
  if (kk >= K) return;
  const int Kg = K / jb.groups, Og = jb.O / jb.groups;
  const int g = kk / Kg, k = kk - g * Kg;
  const int per = (Og + slices - 1) / slices;
  const int o0 = g * Og + slice * per, o1 = min(o0 + per, (g + 1) * Og);
  const float* w = reinterpret_cast<const float*>(jb.w);
  float acc = 0.f;
  for (int o = o0; o < o1; ++o) acc += jb.dc[(size_t)m * jb.O + o] * jb.row_scale[o] * w[(size_t)o * Kg + k];
  atomicAdd(dx + (size_t)m * x_stride + kk, acc);
}

// EDM2 training loss (reference training/module_trainers/unet_trainer.py:271-282): per sample
//   wl = mean_chw((D - x)^2) * (sigma^2 + sd^2) / (sigma * sd)^2,   loss = wl / exp(logvar) + logvar,
// and the gradients of mean_b(loss): dD = (2 w / (N B exp(logvar))) (D - x), dlogvar = (1 - wl / exp(logvar)) / B.
__global__ __launch_bounds__(256) void edm2_loss_grad_kernel(const float* __restrict__ d, const float* __restrict__ x, const float* __restrict__ sigma,
                                                             const float* __restrict__ logvar, float sd, const float* __restrict__ sdv,
                                                             float* __restrict__ dd, float* __restrict__ ss, int B, size_t n) {
  __shared__ float scratch[4];
  const int b = blockIdx.y;
  const float sg = sigma[b];
  if (sdv) sd = sdv[b];      // use_dynamic_sigma_data (unet_trainer.py:263-269): the loss weight's sigma_data is per sample
  const float w = (sg * sg + sd * sd) / ((sg * sd) * (sg * sd));
  const float k = 2.0f * w / ((float)n * (float)B * __expf(logvar ? logvar[b] : 0.f));
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float e = d[b * n + i] - x[b * n + i];
    acc += e * e;
    if (dd) dd[b * n + i] = k * e;
  }
  acc = block_sum_256(acc, scratch);
  if (threadIdx.x == 0) atomicAdd(ss + b, acc);
}
__global__ void edm2_loss_finish_kernel(const float* __restrict__ ss, const float* __restrict__ sigma, const float* __restrict__ logvar, float sd,
                                        const float* __restrict__ sdv, float* __restrict__ loss, float* __restrict__ dlogvar, int B, size_t n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float sg = sigma[b];
  if (sdv) sd = sdv[b];
  const float w = (sg * sg + sd * sd) / ((sg * sd) * (sg * sd));
  const float wl = ss[b] / (float)n * w;
  const float lv = logvar ? logvar[b] : 0.f;
  loss[b] = logvar ? wl * __expf(-lv) + lv : wl;
  if (dlogvar) dlogvar[b] = (1.0f - wl * __expf(-lv)) / (float)B;
}

inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 8192); }

}  // namespace
}  // namespace ddx

using namespace ddx;

extern "C" int ddx_silu_scale_bwd_ex(const void* da, int64_t da_ld, const void* y, const float* chan_scale, float scale, const void* add,
                                     int64_t add_ld, void* dy, float* dc, int32_t B, int64_t HW, int32_t C, int32_t act, int32_t dtype,
                                     ddx_stream stream) {
  if (!da || !y || !dy || B <= 0 || HW <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "silu_scale_bwd: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (C % ev || da_ld % ev || (add && add_ld % ev)) return set_error(DDX_ERR_UNSUPPORTED, "silu_scale_bwd: C and the row strides must be multiples of the 16-byte vector");
  if (dc && !chan_scale) return set_error(DDX_ERR_ARG, "silu_scale_bwd: dc without chan_scale");
  return dispatch([=](hipStream_t s) -> int {
    // rows per workgroup: as many as possible (one dc atomic per workgroup and channel) while ~1024 workgroups remain --
    // a fixed 128 left the low-resolution levels (HW = 86 ... 1376) with 8 ... 44 workgroups, latency-bound at 60 us
    const int zb = (C / ev + 255) / 256;
    const int rows_per_block = (int)std::min<int64_t>(128, std::max<int64_t>(4, HW * B * zb / 1024));
    dim3 grid((unsigned)((HW + rows_per_block - 1) / rows_per_block), (unsigned)B, (unsigned)zb);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(silu_scale_bwd_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)da, (int)da_ld, (const bf16*)y, chan_scale, scale,
                         (const bf16*)add, (int)add_ld, (bf16*)dy, dc, (int)HW, C, rows_per_block, act);
    else
      hipLaunchKernelGGL(silu_scale_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)da, (int)da_ld, (const float*)y, chan_scale, scale,
                         (const float*)add, (int)add_ld, (float*)dy, dc, (int)HW, C, rows_per_block, act);
    return check_launch("silu_scale_bwd");
  }, stream, "silu_scale_bwd", 0.0, (add ? 4.0 : 3.0) * (double)B * HW * C * (double)dtype_size(dtype));
}

extern "C" int ddx_silu_scale_bwd(const void* da, const void* y, const float* chan_scale, float scale, void* dy, float* dc, int32_t B,
                                  int64_t HW, int32_t C, int32_t dtype, ddx_stream stream) {
  return ddx_silu_scale_bwd_ex(da, C, y, chan_scale, scale, nullptr, 0, dy, dc, B, HW, C, 1, dtype, stream);
}

extern "C" int ddx_silu_scale_fwd(const void* x, const float* chan_scale, float scale, void* out, int32_t B, int64_t HW, int32_t C,
                                  int32_t act, int32_t dtype, ddx_stream stream) {
  if (!x || !out || B <= 0 || HW <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "silu_scale_fwd: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (C % ev) return set_error(DDX_ERR_UNSUPPORTED, "silu_scale_fwd: C must be a multiple of the 16-byte vector");
  return dispatch([=](hipStream_t s) -> int {
    const size_t nvt = (size_t)B * HW * (C / ev);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(silu_scale_fwd_kernel<bf16>, dim3(grid_for(nvt)), dim3(256), 0, s, (const bf16*)x, chan_scale, scale, (bf16*)out, (size_t)HW, C, nvt, act);
    else
      hipLaunchKernelGGL(silu_scale_fwd_kernel<float>, dim3(grid_for(nvt)), dim3(256), 0, s, (const float*)x, chan_scale, scale, (float*)out, (size_t)HW, C, nvt, act);
    return check_launch("silu_scale_fwd");
  }, stream, "silu_scale_fwd", 0.0, 2.0 * (double)B * HW * C * (double)dtype_size(dtype));
}

extern "C" int ddx_mpsum_clip_bwd(const void* dout, const void* out, void* dres, void* dy, float t, float clip, int64_t n, int32_t dtype,
                                  ddx_stream stream) {
  if (!dout || !dy || n <= 0 || (clip > 0.f && !out)) return set_error(DDX_ERR_ARG, "mpsum_clip_bwd: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (n % ev) return set_error(DDX_ERR_UNSUPPORTED, "mpsum_clip_bwd: n must be a multiple of the 16-byte vector");
  const float nrm = std::sqrt((1.f - t) * (1.f - t) + t * t);
  const float a = (1.f - t) / nrm, b = t / nrm;
  return dispatch([=](hipStream_t s) -> int {
    const size_t nvec = (size_t)n / ev;
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(mpsum_clip_bwd_kernel<bf16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)dout, (const bf16*)out, (bf16*)dres, (bf16*)dy, a, b, clip, nvec);
    else
      hipLaunchKernelGGL(mpsum_clip_bwd_kernel<float>, dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)dout, (const float*)out, (float*)dres, (float*)dy, a, b, clip, nvec);
    return check_launch("mpsum_clip_bwd");
  }, stream, "mpsum_clip_bwd", 0.0, 4.0 * (double)n * (double)dtype_size(dtype));
}

extern "C" int ddx_pixelnorm_bwd(const void* dy, const void* x, void* dx, int64_t rows, int32_t C, float eps, int32_t dtype, ddx_stream stream) {
  if (!dy || !x || !dx || rows <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "pixelnorm_bwd: bad args");
  if (C % (dtype == DDX_BF16 ? 8 : 4)) return set_error(DDX_ERR_UNSUPPORTED, "pixelnorm_bwd: C must be a multiple of the 16-byte vector");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = (int)std::min<int64_t>((rows + 3) / 4, 16384);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(pixelnorm_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, rows, C, eps);
    else
      hipLaunchKernelGGL(pixelnorm_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, rows, C, eps);
    return check_launch("pixelnorm_bwd");
  }, stream, "pixelnorm_bwd", 0.0, 3.0 * (double)rows * C * (double)dtype_size(dtype));
}

extern "C" int ddx_mpconv_wprep_bwd(const ddx_wprep_desc* dp, const float* dwp, float* dw, float* dgain, int32_t accumulate, ddx_stream stream) {
  if (!dp || !dp->w || !dwp || !dw) return set_error(DDX_ERR_ARG, "wprep_bwd: null");
  const ddx_wprep_desc d = *dp;
  if (d.groups <= 0 || d.Cout % d.groups || (d.ksize != 1 && d.ksize != 3)) return set_error(DDX_ERR_ARG, "wprep_bwd: bad shape");
  if (d.transpose) return set_error(DDX_ERR_ARG, "wprep_bwd: describe the forward preparation (transpose = 0)");
  return dispatch([=](hipStream_t s) -> int {
    const int taps = d.ksize * d.ksize;
    if (d.w_dtype == DDX_F32)
      hipLaunchKernelGGL(wprep_bwd_kernel<float>, dim3(d.Cout), dim3(256), 0, s, dwp, (const float*)d.w, d.gain_ptr, d.gain, dw, dgain, d.Cout, d.Cg,
                         taps, d.groups, d.normalize, d.qk_head_dim, 1e-4f, d.in_split, d.in_scale0, d.in_scale1, accumulate);
    else
      hipLaunchKernelGGL(wprep_bwd_kernel<bf16>, dim3(d.Cout), dim3(256), 0, s, dwp, (const bf16*)d.w, d.gain_ptr, d.gain, dw, dgain, d.Cout, d.Cg,
                         taps, d.groups, d.normalize, d.qk_head_dim, 1e-4f, d.in_split, d.in_scale0, d.in_scale1, accumulate);
    return check_launch("wprep_bwd");
  }, stream, "wprep_bwd");
}

extern "C" int ddx_linear_small_bwd(const float* dc, const float* x, int32_t x_stride, const void* w, int32_t w_dtype, const float* row_scale,
                                    float* dwp, float* dx, int32_t M, int32_t O, int32_t K, int32_t groups, ddx_stream stream) {
  if (!dc || !x || !w || !row_scale || !dwp || M <= 0 || O <= 0 || K <= 0 || groups <= 0 || O % groups || K % groups)
    return set_error(DDX_ERR_ARG, "linear_small_bwd: bad args");
  return dispatch([=](hipStream_t s) -> int {
    dim3 gdx((K + 255) / 256, M, 16);
    if (w_dtype == DDX_F32) {
      hipLaunchKernelGGL(linear_small_bwd_kernel<float>, dim3(O), dim3(256), 0, s, dc, x, (const float*)w, row_scale, dwp, dx, M, O, K / groups, groups, x_stride);
      if (dx) hipLaunchKernelGGL(linear_small_bwd_dx_kernel<float>, gdx, dim3(256), 0, s, dc, (const float*)w, row_scale, dx, M, O, K / groups, groups, x_stride);
    } else {
      hipLaunchKernelGGL(linear_small_bwd_kernel<bf16>, dim3(O), dim3(256), 0, s, dc, x, (const bf16*)w, row_scale, dwp, dx, M, O, K / groups, groups, x_stride);
      if (dx) hipLaunchKernelGGL(linear_small_bwd_dx_kernel<bf16>, gdx, dim3(256), 0, s, dc, (const bf16*)w, row_scale, dx, M, O, K / groups, groups, x_stride);
    }
    return check_launch("linear_small_bwd");
  }, stream, "linear_small_bwd");
}

extern "C" int ddx_linear_small_bwd_batched(const ddx_linear_bwd_job* jobs_dev, int32_t njobs, int32_t max_O, const float* x, int32_t x_stride,
                                            float* dx, int32_t M, int32_t K, ddx_stream stream) {
  if (!jobs_dev || njobs <= 0 || max_O <= 0 || !x || M <= 0 || K <= 0) return set_error(DDX_ERR_ARG, "linear_small_bwd_batched: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(linear_small_bwd_multi_kernel, dim3(max_O, njobs), dim3(256), 0, s, jobs_dev, x, M, K, x_stride);
    if (dx) {
      constexpr int kSlices = 8;
      hipLaunchKernelGGL(linear_small_bwd_dx_multi_kernel, dim3((K + 255) / 256, M, njobs * kSlices), dim3(256), 0, s, jobs_dev, dx, M, K, x_stride, kSlices);
    }
    return check_launch("linear_small_bwd_batched");
  }, stream, "linear_small_bwd");
}

extern "C" int ddx_add3(const void* a, const void* b, const void* c, void* out, int64_t n, int32_t dtype, ddx_stream stream) {
  if (!a || !b || !out || n <= 0) return set_error(DDX_ERR_ARG, "add3: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (n % ev) return set_error(DDX_ERR_UNSUPPORTED, "add3: n must be a multiple of the 16-byte vector");
  return dispatch([=](hipStream_t s) -> int {
    const size_t nvec = (size_t)n / ev;
    if (dtype == DDX_BF16) hipLaunchKernelGGL(add_kernel<bf16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (const bf16*)c, (bf16*)out, nvec);
    else hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)a, (const float*)b, (const float*)c, (float*)out, nvec);
    return check_launch("add3");
  }, stream, "add3", 0.0, (c ? 4.0 : 3.0) * (double)n * (double)dtype_size(dtype));
}

extern "C" int ddx_edm2_loss_v(const float* denoised, const float* target, const float* sigma, const float* logvar, float sigma_data,
                               const float* sigma_data_vec, float* loss, float* d_denoised, float* d_logvar, float* workspace, int32_t B,
                               int64_t n_per_sample, ddx_stream stream) {
  if (!denoised || !target || !sigma || !loss || !workspace || B <= 0 || n_per_sample <= 0) return set_error(DDX_ERR_ARG, "edm2_loss: bad args");
  return dispatch([=](hipStream_t s) -> int {
    if (int rc = zero_bytes(workspace, sizeof(float) * B, s)) return rc;   // (a kernel, not a memset node: common.hpp)
    dim3 grid((unsigned)std::min<int64_t>((n_per_sample + 255) / 256, 512), (unsigned)B);
    hipLaunchKernelGGL(edm2_loss_grad_kernel, grid, dim3(256), 0, s, denoised, target, sigma, logvar, sigma_data, sigma_data_vec, d_denoised, workspace, B,
                       (size_t)n_per_sample);
    hipLaunchKernelGGL(edm2_loss_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const float*)workspace, sigma, logvar, sigma_data, sigma_data_vec, loss,
                       d_logvar, B, (size_t)n_per_sample);
    return check_launch("edm2_loss");
  }, stream, "edm2_loss");
}

extern "C" int ddx_edm2_loss(const float* denoised, const float* target, const float* sigma, const float* logvar, float sigma_data, float* loss,
                             float* d_denoised, float* d_logvar, float* workspace, int32_t B, int64_t n_per_sample, ddx_stream stream) {
  return ddx_edm2_loss_v(denoised, target, sigma, logvar, sigma_data, nullptr, loss, d_denoised, d_logvar, workspace, B, n_per_sample, stream);
}

// ---- magnitude-preserving dropout (unet_edm2_b4.py:124-125: F.dropout(y, p) * (1 - p)^0.5, training only) and its backward: the same
// element-wise factor keep / sqrt(1 - p).  The keep mask is never stored: Philox4x32-10 over (element index / 4, stream id) with the draw's
// 64-bit seed as key regenerates it wherever it is needed (forward on the activation, backward on its gradient).
namespace ddx {
namespace {
struct u32x4_t { unsigned x, y, z, w; };
__device__ __forceinline__ u32x4_t philox4x32_10(u32x4_t c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = u32x4_t{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
template <typename T>
__global__ __launch_bounds__(256) void mp_dropout_kernel(T* __restrict__ x, size_t n, float p, float scale, unsigned k0, unsigned k1, unsigned stream_id) {
  const size_t n4 = (n + 3) / 4;
  for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * 256) {
    const u32x4_t r = philox4x32_10(u32x4_t{(unsigned)i4, (unsigned)(i4 >> 32), stream_id, 0u}, k0, k1);
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = i4 * 4 + e;
      if (i >= n) break;
      const bool keep = (float)rr[e] * 2.3283064365386963e-10f >= p;      // uniform in [0, 1): P(keep) = 1 - p
      x[i] = from_f32<T>(keep ? to_f32<T>(x[i]) * scale : 0.f);
    }
  }
}
// backward of D = mp_sum(x_ref[:, :-1], D0, t = x_ref[:, -1:]) (unet_edm2_b4.py:293-294), all NCHW fp32:
//   n = sqrt((1 - t)^2 + t^2);  dD0 = dD t / n;  d xr = dD (1 - t) / n;  dt = sum_c dD [ (D0 - xr) / n - ((1 - t) xr + t D0) (2 t - 1) / n^3 ]
__global__ __launch_bounds__(256) void xref_mix_bwd_kernel(const float* __restrict__ dD, const float* __restrict__ D0, const float* __restrict__ x_ref,
                                                           float* __restrict__ dD0, float* __restrict__ dxref, int B, int C, size_t HW) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, hw = i - b * HW;
    const float t = x_ref[(b * (C + 1) + C) * HW + hw];
    const float n2 = (1.f - t) * (1.f - t) + t * t, n = sqrtf(n2), inv = 1.f / n;
    float dt = 0.f;
    for (int c = 0; c < C; ++c) {
      const size_t o = (b * C + c) * HW + hw, oxr = (b * (C + 1) + c) * HW + hw;
      const float g = dD[o], d0 = D0[o], xr = x_ref[oxr];
      dD0[o] = g * t * inv;
      if (dxref) dxref[oxr] = g * (1.f - t) * inv;
      dt += g * ((d0 - xr) * inv - ((1.f - t) * xr + t * d0) * (2.f * t - 1.f) * inv / n2);
    }
    if (dxref) dxref[(b * (C + 1) + C) * HW + hw] = dt;
  }
}
}  // namespace
}  // namespace ddx

extern "C" int ddx_mp_dropout(void* x, int64_t n, float p, uint64_t seed, uint32_t stream_id, int32_t dtype, ddx_stream stream) {
  if (!x || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return set_error(DDX_ERR_ARG, "mp_dropout: bad args");
  if (dtype != DDX_BF16 && dtype != DDX_F32) return set_error(DDX_ERR_ARG, "mp_dropout: dtype");
  const float scale = 1.0f / std::sqrt(1.0f - p);     // 1 / (1 - p) of the dropout times (1 - p)^0.5 of the block
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)(n + 3) / 4);
    if (dtype == DDX_BF16) hipLaunchKernelGGL(mp_dropout_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (bf16*)x, (size_t)n, p, scale, (unsigned)seed, (unsigned)(seed >> 32), stream_id);
    else hipLaunchKernelGGL(mp_dropout_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)x, (size_t)n, p, scale, (unsigned)seed, (unsigned)(seed >> 32), stream_id);
    return check_launch("mp_dropout");
  }, stream, "mp_dropout", 0.0, 2.0 * (double)n * (double)dtype_size(dtype));
}

extern "C" int ddx_unet_xref_mix_bwd(const float* d_out_nchw, const float* d0_nchw, const float* x_ref_nchw, float* d_d0_nchw, float* d_x_ref_nchw,
                                     int32_t B, int32_t C, int32_t H, int32_t W, ddx_stream stream) {
  if (!d_out_nchw || !d0_nchw || !x_ref_nchw || !d_d0_nchw || B <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "xref_mix_bwd: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(xref_mix_bwd_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, s, d_out_nchw, d0_nchw, x_ref_nchw, d_d0_nchw, d_x_ref_nchw, B, C,
                       (size_t)H * W);
    return check_launch("xref_mix_bwd");
  }, stream, "xref_mix_bwd");
}

// backward of D = c_skip * x_in + c_out * y w.r.t. y (unet_edm2_b4.py:291): dy[b][h][w][c] = c_out(sigma_b) * dD[b][c][h][w], written
// NHWC in the activation dtype with the channel axis zero-padded to Cpad (conv_out's gradient operand)
namespace ddx {
template <typename T>
__global__ __launch_bounds__(256) void output_combine_bwd_kernel(const float* __restrict__ dd, const float* __restrict__ sigma, T* __restrict__ dy, int B,
                                                                 int C, int H, int W, int Cpad, float sd) {
  const size_t total = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t hw = i % ((size_t)H * W);
    const int b = (int)(i / ((size_t)H * W));
    const float sg = sigma[b];
    const float c_out = sg * sd * rsqrtf(sg * sg + sd * sd);
    for (int c = 0; c < Cpad; ++c) dy[i * Cpad + c] = from_f32<T>(c < C ? c_out * dd[((size_t)b * C + c) * H * W + hw] : 0.f);
  }
}
}  // namespace ddx

extern "C" int ddx_unet_output_combine_bwd(const float* d_out_nchw, const float* sigma, void* dy_nhwc, int32_t B, int32_t C, int32_t H, int32_t W,
                                           int32_t Cpad, float sigma_data, int32_t dtype, ddx_stream stream) {
  if (!d_out_nchw || !sigma || !dy_nhwc || Cpad < C) return set_error(DDX_ERR_ARG, "output_combine_bwd: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 8192);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(output_combine_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, d_out_nchw, sigma, (bf16*)dy_nhwc, B, C, H, W, Cpad, sigma_data);
    else
      hipLaunchKernelGGL(output_combine_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, d_out_nchw, sigma, (float*)dy_nhwc, B, C, H, W, Cpad, sigma_data);
    return check_launch("output_combine_bwd");
  }, stream);
}
