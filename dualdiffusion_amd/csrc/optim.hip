// Optimizer step of the training loop as multi-tensor kernels over a device job table (one launch for all parameters):
// global gradient norm, clipping, AdamW, EMA.  Replaces, for the UNet's ~100 parameter tensors (293 M values),
// accelerator.clip_grad_norm_ + torch.optim.AdamW.step + the EMA lerp of reference src/training/trainer.py:1027-1063,456-474
// (and ema.py) -- ~10 passes over 1.2 GB in the reference, 2 here (norm, update).  fp32 master weights and moments.
#include <algorithm>

#include "common.hpp"

namespace ddx {
namespace {

constexpr int kChunk = 256 * 16;  // elements per workgroup pass

__global__ __launch_bounds__(256) void multi_sqnorm_kernel(const ddx_optim_job* __restrict__ jobs, float* __restrict__ out) {
  __shared__ float scratch[4];
  const ddx_optim_job j = jobs[blockIdx.y];
  float acc = 0.f;
  if ((j.n & 3) == 0 && (reinterpret_cast<size_t>(j.g) & 15) == 0) {   // 16-byte loads (the gradients of one flat bucket: always the case there)
    const f32x4* g4 = reinterpret_cast<const f32x4*>(j.g);
    const int64_t n4 = j.n >> 2, stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent 16-byte loads in flight per thread
      const f32x4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
      acc += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3] +
             c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3] + d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    for (; i < n4; i += stride) {
      const f32x4 g = g4[i];
      acc += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (int64_t)gridDim.x * 256) {
      const float g = j.g[i];
      acc += g * g;
    }
  }
  acc = block_sum_256(acc, scratch);
  if (threadIdx.x == 0 && acc != 0.f) atomicAdd(out, acc);
}

// clip coefficient exactly as torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6)), norm of the SCALED grads
__global__ void clip_coef_kernel(const float* __restrict__ sqnorm, float gscale, float max_norm, float* __restrict__ coef_norm) {
  const float norm = sqrtf(sqnorm[0]) * gscale;
  coef_norm[0] = fminf(1.0f, max_norm / (norm + 1e-6f));
  coef_norm[1] = norm;
}

__global__ __launch_bounds__(256) void multi_adamw_kernel(const ddx_optim_job* __restrict__ jobs, const float* __restrict__ coef, float gscale, float lr,
                                                          float beta1, float beta2, float eps, float weight_decay, float bias1, float bias2,
                                                          float ema_beta) {
  const ddx_optim_job j = jobs[blockIdx.y];
  // non-finite gradient norm (coef[1]): the whole step is skipped on the device, so NaN / Inf never reach p, m, v or the EMA
  // (the host reads the norm after this launch: one sync per step, reference trainer.py:1044-1060)
  if (coef && !(fabsf(coef[1]) <= 3.0e38f)) return;
  const float gs = gscale * (coef ? coef[0] : 1.0f);
  const float step = lr / bias1;
  const float inv_sqrt_bias2 = rsqrtf(bias2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (int64_t)gridDim.x * 256) {
    const float g = j.g[i] * gs;
    float p = j.p[i];
    const float m = beta1 * j.m[i] + (1.0f - beta1) * g;
    const float v = beta2 * j.v[i] + (1.0f - beta2) * g * g;
    p -= lr * weight_decay * p;                                        // decoupled weight decay (torch.optim.AdamW)
    p -= step * m / (sqrtf(v) * inv_sqrt_bias2 + eps);
    j.m[i] = m; j.v[i] = v; j.p[i] = p;
    if (j.ema) j.ema[i] = ema_beta * j.ema[i] + (1.0f - ema_beta) * p;  // torch.lerp(ema, p, 1 - beta)
  }
}

// ---- the whole post-backward parameter pass as ONE multi-tensor launch: clip * AdamW, then for every EMA (in configuration order,
// reference src/training/ema.py:284-321) ema_k <- lerp(ema_k, p, 1 - beta_k) and, for a feedback EMA, p <- lerp(p, ema_k,
// 1 - feedback_beta_k), then the forced weight normalisation of the row (mp_tools.py:375-378, trainer.py:1105-1108).  A workgroup
// owns whole rows (fan-in elements of one output channel): first sweep updates p / m / v / EMAs and sums p^2, second sweep re-reads
// the row (a few KB, L2-resident) and rescales it -- the pass `normalize_weights()` would make over the tensor anyway.
struct EmaCoef { float beta[DDX_MAX_EMAS]; float fb[DDX_MAX_EMAS]; };   // fb < 0: no feedback

__global__ __launch_bounds__(256) void multi_adamw_ema_wn_kernel(const ddx_optim_job_ex* __restrict__ jobs, const float* __restrict__ coef, float gscale,
                                                                 float lr, float beta1, float beta2, float eps, float weight_decay, float bias1,
                                                                 float bias2, int n_ema, EmaCoef ec, float norm_eps, int wave_rows) {
  __shared__ float scratch[4];
  const ddx_optim_job_ex j = jobs[blockIdx.y];
  if (coef && !(fabsf(coef[1]) <= 3.0e38f)) return;   // non-finite gradient norm: skip the step on the device
  const float gs = gscale * (coef ? coef[0] : 1.0f);
  const float step = lr / bias1;
  const float inv_sqrt_bias2 = rsqrtf(bias2);
  const int64_t fan = j.n / j.rows;
  auto update = [&](float g, float& p, float& m, float& v) {
    g *= gs;
    m = beta1 * m + (1.0f - beta1) * g;
    v = beta2 * v + (1.0f - beta2) * g * g;
    p -= lr * weight_decay * p;
    p -= step * m / (sqrtf(v) * inv_sqrt_bias2 + eps);
  };
  // Rows of up to 4096 elements (every conv / linear weight of the UNet: 288 ... 3840): ONE WAVE per row, 16-byte accesses, the updated
  // row stays in registers for the re-normalisation -- no block reduction, no barrier, no second read.  (A workgroup per row of 1-15 KB
  // spent its time in the two barriers of the reduction and in 4-byte accesses: 5.5 ms for the 293 M parameters with two EMAs.)
  constexpr int MAXV = 16;
  auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
  bool vec = wave_rows && (fan & 3) == 0 && fan <= 64 * 4 * MAXV && al16(j.g) && al16(j.p) && al16(j.m) && al16(j.v);
#pragma unroll
  for (int e = 0; e < DDX_MAX_EMAS; ++e)
    if (e < n_ema && j.ema[e] && !al16(j.ema[e])) vec = false;
  if (vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n4 = (int)(fan >> 2);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < j.rows; row += (int64_t)gridDim.x * 4) {
      const int64_t b4 = row * n4;
      const f32x4* g4 = reinterpret_cast<const f32x4*>(j.g) + b4;
      f32x4 *p4 = reinterpret_cast<f32x4*>(j.p) + b4, *m4 = reinterpret_cast<f32x4*>(j.m) + b4, *v4 = reinterpret_cast<f32x4*>(j.v) + b4;
      f32x4 pv[MAXV];
      float ss = 0.f;
      // (round 5) every operand vector of TCH consecutive 1-KiB pieces of the row is REQUESTED before the first is used: the plain loop's
      // load -> update -> store chain per piece left ~one piece per wave in flight (3.3 TB/s over the 293 M parameters with two EMAs)
      constexpr int TCH = 2;
#pragma unroll
      for (int t0 = 0; t0 < MAXV; t0 += TCH) {
        if (t0 * 64 >= n4) break;     // (wave-uniform)
        f32x4 gq[TCH], pq[TCH], mq[TCH], vq[TCH], eq[DDX_MAX_EMAS][TCH];
#pragma unroll
        for (int u = 0; u < TCH; ++u) {
          const int idx = lane + 64 * (t0 + u);
          const int ic = idx < n4 ? idx : 0;
          gq[u] = g4[ic]; pq[u] = p4[ic]; mq[u] = m4[ic]; vq[u] = v4[ic];
#pragma unroll
          for (int e = 0; e < DDX_MAX_EMAS; ++e)
            if (e < n_ema && j.ema[e]) eq[e][u] = (reinterpret_cast<const f32x4*>(j.ema[e]) + b4)[ic];
        }
#pragma unroll
        for (int u = 0; u < TCH; ++u) {
          const int t = t0 + u, idx = lane + 64 * t;
          if (idx < n4) {
            const f32x4 g = gq[u];
            f32x4 p = pq[u], m = mq[u], v = vq[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float pc = p[c], mc = m[c], vc = v[c];
              update(g[c], pc, mc, vc);
              p[c] = pc; m[c] = mc; v[c] = vc;
            }
            m4[idx] = m; v4[idx] = v;
#pragma unroll
            for (int e = 0; e < DDX_MAX_EMAS; ++e) {
              if (e < n_ema && j.ema[e]) {
                f32x4* e4 = reinterpret_cast<f32x4*>(j.ema[e]) + b4;
                f32x4 a = eq[e][u];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  a[c] = a[c] + (1.0f - ec.beta[e]) * (p[c] - a[c]);
                  if (ec.fb[e] >= 0.f) p[c] = p[c] + (1.0f - ec.fb[e]) * (a[c] - p[c]);
                }
                e4[idx] = a;
              }
            }
            pv[t] = p;
            ss += p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
          }
        }
      }
      float inv = 1.0f;
      if (j.normalize) inv = 1.0f / (norm_eps + sqrtf(wave_sum(ss)) * sqrtf(1.0f / (float)fan));
#pragma unroll
      for (int t = 0; t < MAXV; ++t) {
        const int idx = lane + 64 * t;
        if (idx < n4) {
          f32x4 p = pv[t];
          if (j.normalize) { p[0] *= inv; p[1] *= inv; p[2] *= inv; p[3] *= inv; }
          p4[idx] = p;
        }
      }
    }
    return;
  }
  for (int64_t row = blockIdx.x; row < j.rows; row += gridDim.x) {
    const int64_t base = row * fan;
    float ss = 0.f;
    for (int64_t k = threadIdx.x; k < fan; k += 256) {
      const int64_t i = base + k;
      const float g = j.g[i] * gs;
      float p = j.p[i];
      const float m = beta1 * j.m[i] + (1.0f - beta1) * g;
      const float v = beta2 * j.v[i] + (1.0f - beta2) * g * g;
      p -= lr * weight_decay * p;
      p -= step * m / (sqrtf(v) * inv_sqrt_bias2 + eps);
      j.m[i] = m; j.v[i] = v;
#pragma unroll
      for (int e = 0; e < DDX_MAX_EMAS; ++e) {
        if (e < n_ema && j.ema[e]) {
          float a = j.ema[e][i];
          a = a + (1.0f - ec.beta[e]) * (p - a);              // torch.lerp(ema, p, 1 - beta)
          j.ema[e][i] = a;
          if (ec.fb[e] >= 0.f) p = p + (1.0f - ec.fb[e]) * (a - p);   // feedback: torch.lerp(p, ema, 1 - feedback_beta)
        }
      }
      j.p[i] = p;
      ss += p * p;
    }
    if (j.normalize) {   // (workgroup-uniform)
      ss = block_sum_256(ss, scratch);
      const float inv = 1.0f / (norm_eps + sqrtf(ss) * sqrtf(1.0f / (float)fan));
      __syncthreads();
      for (int64_t k = threadIdx.x; k < fan; k += 256) j.p[base + k] *= inv;   // each thread re-reads what it wrote
    }
  }
}

}  // namespace
}  // namespace ddx

using namespace ddx;

extern "C" int ddx_multi_adamw_ema_wn(const ddx_optim_job_ex* jobs_dev, int32_t njobs, int64_t max_rows, const float* clip_coef, float grad_scale,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, int32_t n_ema,
                                      const float* ema_beta, const float* feedback_beta, float norm_eps, ddx_stream stream) {
  if (!jobs_dev || njobs <= 0 || max_rows <= 0 || step <= 0 || n_ema < 0 || n_ema > DDX_MAX_EMAS || (n_ema > 0 && (!ema_beta || !feedback_beta)))
    return set_error(DDX_ERR_ARG, "multi_adamw_ema_wn: bad args");
  const float bias1 = 1.0f - std::pow(beta1, (float)step), bias2 = 1.0f - std::pow(beta2, (float)step);
  EmaCoef ec{};
  for (int e = 0; e < DDX_MAX_EMAS; ++e) { ec.beta[e] = e < n_ema ? ema_beta[e] : 1.f; ec.fb[e] = e < n_ema ? feedback_beta[e] : -1.f; }
  const int wave_rows = 1;   // one wave per row (one workgroup per row with a block reduction measured 9.4 vs 4.05 ms; the block path serves rows > 4096)
  return dispatch([=](hipStream_t s) -> int {
    dim3 grid((unsigned)std::min<int64_t>(max_rows, 512), (unsigned)njobs);
    hipLaunchKernelGGL(multi_adamw_ema_wn_kernel, grid, dim3(256), 0, s, jobs_dev, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay, bias1,
                       bias2, n_ema, ec, norm_eps, wave_rows);
    return check_launch("multi_adamw_ema_wn");
  }, stream, "adamw_ema_wn");
}

extern "C" int ddx_multi_grad_norm(const ddx_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float grad_scale, float max_norm,
                                   float* workspace3, ddx_stream stream) {
  if (!jobs_dev || njobs <= 0 || max_n <= 0 || !workspace3) return set_error(DDX_ERR_ARG, "multi_grad_norm: bad args");
  return dispatch([=](hipStream_t s) -> int {
    if (int rc = zero_bytes(workspace3, sizeof(float), s)) return rc;
    // (every workgroup ends in ONE atomicAdd on the same float: few, long-running workgroups -- 64 per tensor, four 16-byte loads in
    // flight per thread -- instead of 256 per tensor whose atomics serialise behind each other)
    dim3 grid((unsigned)std::min<int64_t>((max_n + kChunk - 1) / kChunk, 64), (unsigned)njobs);
    hipLaunchKernelGGL(multi_sqnorm_kernel, grid, dim3(256), 0, s, jobs_dev, workspace3);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, (const float*)workspace3, grad_scale, max_norm, workspace3 + 1);
    return check_launch("multi_grad_norm");
  }, stream, "grad_norm");
}

extern "C" int ddx_clip_coef(float* workspace3, float grad_scale, float max_norm, ddx_stream stream) {
  if (!workspace3) return set_error(DDX_ERR_ARG, "clip_coef: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, (const float*)workspace3, grad_scale, max_norm, workspace3 + 1);
    return check_launch("clip_coef");
  }, stream, "grad_norm");
}

extern "C" int ddx_multi_adamw(const ddx_optim_job* jobs_dev, int32_t njobs, int64_t max_n, const float* clip_coef, float grad_scale, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step, float ema_beta, ddx_stream stream) {
  if (!jobs_dev || njobs <= 0 || max_n <= 0 || step <= 0) return set_error(DDX_ERR_ARG, "multi_adamw: bad args");
  const float bias1 = 1.0f - std::pow(beta1, (float)step), bias2 = 1.0f - std::pow(beta2, (float)step);
  return dispatch([=](hipStream_t s) -> int {
    dim3 grid((unsigned)std::min<int64_t>((max_n + kChunk - 1) / kChunk, 256), (unsigned)njobs);
    hipLaunchKernelGGL(multi_adamw_kernel, grid, dim3(256), 0, s, jobs_dev, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay, bias1, bias2,
                       ema_beta);
    return check_launch("multi_adamw");
  }, stream, "adamw");
}
