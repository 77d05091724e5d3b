// HBM-bound glue kernels of the UNet forward: pixel-norm, input/output preconditioning, layout conversion, the
// noise-embedding front end and the small-M linear layers.  All are bandwidth- or latency-bound: coalesced 16-byte
// accesses, wave shuffles for the reductions, no LDS staging needed.
#include <algorithm>
#include "common.hpp"

namespace ddx {

// ---------------------------------------------------------------------------------------------- pixel norm
// reference: normalize(x, dim=1) (src/modules/mp_tools.py:42-49) as used at unet_edm2_b4.py:117.
// NHWC rows are contiguous: one wave per row, 16-byte lanes, two passes over registers (row cached when C small).
template <typename T>
__global__ __launch_bounds__(256) void pixelnorm_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ y2, int64_t rows, int C, float eps) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_sqrt_c = rsqrtf((float)C);
  // rows of at most 32 vectors (C <= 256 bf16 / 128 fp32): several rows per wave, one per group of LPR lanes, so that all
  // 64 lanes move data (the L0 pixel norms, C = 256 bf16, used half of them: 2.7 TB/s)
  if (C % EV == 0 && C / EV <= 32) {
    const int nvec = C / EV;
    int LPR = 1;
    while (LPR < nvec) LPR <<= 1;
    const int G = 64 / LPR, grp = lane / LPR, sub = lane % LPR;
    for (int64_t r0 = wave_id * G; r0 < rows; r0 += nwaves * G) {
      const int64_t r = r0 + grp;
      const bool act = r < rows && sub < nvec;
      Vec16<T> c;
      float ss = 0.f;
      if (act) {
        c.v = *reinterpret_cast<const decltype(c.v)*>(x + r * C + (size_t)sub * EV);
#pragma unroll
        for (int e = 0; e < EV; ++e) { const float f = c.get(e); ss += f * f; }
      }
      for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float nrm = eps + sqrtf(ss) * inv_sqrt_c;
      if (act) {
        Vec16<T> o, o2;
#pragma unroll
        for (int e = 0; e < EV; ++e) { const float q = c.get(e) / nrm; o.set(e, q); o2.set(e, mp_silu_f(q)); }
        *reinterpret_cast<decltype(o.v)*>(y + r * C + (size_t)sub * EV) = o.v;
        if (y2) *reinterpret_cast<decltype(o.v)*>(y2 + r * C + (size_t)sub * EV) = o2.v;
      }
    }
    return;
  }
  for (int64_t r = wave_id; r < rows; r += nwaves) {
    const T* xr = x + r * C;
    T* yr = y + r * C;
    T* y2r = y2 ? y2 + r * C : nullptr;
    float ss = 0.f;
    if (C % EV == 0) {
      constexpr int MAXV = 4;  // up to 4 vectors per lane cached in registers (C <= 64*4*EV)
      Vec16<T> cache[MAXV];
      const int nvec = C / EV;
      const bool cached = nvec <= 64 * MAXV;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + k * 64;
        if (cached && vi < nvec) {
          cache[k].v = *reinterpret_cast<const decltype(cache[k].v)*>(xr + (size_t)vi * EV);
#pragma unroll
          for (int e = 0; e < EV; ++e) { const float f = cache[k].get(e); ss += f * f; }
        }
      }
      if (!cached)
        for (int vi = lane; vi < nvec; vi += 64) {
          Vec16<T> t; t.v = *reinterpret_cast<const decltype(t.v)*>(xr + (size_t)vi * EV);
#pragma unroll
          for (int e = 0; e < EV; ++e) { const float f = t.get(e); ss += f * f; }
        }
      ss = wave_sum(ss);
      const float nrm = eps + sqrtf(ss) * inv_sqrt_c;
      if (cached) {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
          const int vi = lane + k * 64;
          if (vi < nvec) {
            Vec16<T> o, o2;
#pragma unroll
            for (int e = 0; e < EV; ++e) { const float q = cache[k].get(e) / nrm; o.set(e, q); o2.set(e, mp_silu_f(q)); }
            *reinterpret_cast<decltype(o.v)*>(yr + (size_t)vi * EV) = o.v;
            if (y2r) *reinterpret_cast<decltype(o.v)*>(y2r + (size_t)vi * EV) = o2.v;
          }
        }
      } else {
        for (int vi = lane; vi < nvec; vi += 64) {
          Vec16<T> t, o, o2; t.v = *reinterpret_cast<const decltype(t.v)*>(xr + (size_t)vi * EV);
#pragma unroll
          for (int e = 0; e < EV; ++e) { const float q = t.get(e) / nrm; o.set(e, q); o2.set(e, mp_silu_f(q)); }
          *reinterpret_cast<decltype(o.v)*>(yr + (size_t)vi * EV) = o.v;
          if (y2r) *reinterpret_cast<decltype(o.v)*>(y2r + (size_t)vi * EV) = o2.v;
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) { const float f = to_f32<T>(xr[c]); ss += f * f; }
      ss = wave_sum(ss);
      const float nrm = eps + sqrtf(ss) * inv_sqrt_c;
      for (int c = lane; c < C; c += 64) {
        const float q = to_f32<T>(xr[c]) / nrm;
        yr[c] = from_f32<T>(q);
        if (y2r) y2r[c] = from_f32<T>(mp_silu_f(q));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- UNet input prep
// reference unet_edm2_b4.py:257-269,277: x = c_in * x_in ; cat(x, ones, ln_freqs).  Output NHWC with Cpad channels.
template <typename T>
__global__ __launch_bounds__(256) void input_prep_kernel(const float* __restrict__ x, const float* __restrict__ sigma,
                                                         const float* __restrict__ lnf, T* __restrict__ out, int B, int C,
                                                         int H, int W, int Cpad, float sigma_data) {
  const size_t npix = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int b = (int)(i / ((size_t)W * H));
    const float sg = sigma[b];
    const float c_in = 1.0f / sqrtf(sigma_data * sigma_data + sg * sg);
    T* o = out + i * Cpad;
    for (int c = 0; c < C; ++c) o[c] = from_f32<T>(c_in * x[(((size_t)b * C + c) * H + h) * W + w]);
    o[C] = from_f32<T>(1.0f);
    o[C + 1] = from_f32<T>(lnf[h]);
    for (int c = C + 2; c < Cpad; ++c) o[c] = from_f32<T>(0.f);
  }
}

// reference unet_edm2_b4.py:290-296: D_x = c_skip * x_in + c_out * x.float(); optional x_ref inpainting mix.
template <typename T>
__global__ __launch_bounds__(256) void output_combine_kernel(const T* __restrict__ y, const float* __restrict__ x_in,
                                                             const float* __restrict__ sigma, const float* __restrict__ x_ref,
                                                             float* __restrict__ out, int B, int C, int H, int W, float sd) {
  const size_t n = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int c = (int)((i / ((size_t)W * H)) % C);
    const int b = (int)(i / ((size_t)W * H * C));
    const float sg = sigma[b];
    const float c_skip = sd * sd / (sg * sg + sd * sd);
    const float c_out = sg * sd / sqrtf(sg * sg + sd * sd);
    float d = c_skip * x_in[i] + c_out * to_f32<T>(y[(((size_t)b * H + h) * W + w) * C + c]);
    if (x_ref) {
      const float t = x_ref[(((size_t)b * (C + 1) + C) * H + h) * W + w];
      const float a = x_ref[(((size_t)b * (C + 1) + c) * H + h) * W + w];
      d = (a + (d - a) * t) / sqrtf((1.f - t) * (1.f - t) + t * t);
    }
    out[i] = d;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int H, int W) {
  const size_t n = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    y[i] = from_f32<T>(x[(((size_t)b * C + c) * H + h) * W + w]);
  }
}

// Stereo depth axis <-> image batch (DAE_G1, reference tensor_4d_to_5d / tensor_5d_to_4d, utils/dual_diffusion_utils.py:571-575):
// NCHW fp32 [B][C*Z][H][W] (channel = c * Z + z)  <->  NHWC images n = Z * b + z with Cpad channels: [c < C: data | c == C:
// `fill` (the constant channel, dae_edm2_g1.py:334-335) when add_const | zeros].
template <typename T>
__global__ __launch_bounds__(256) void stereo_to_images_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int Z, int H, int W, int Cpad,
                                                               int add_const) {
  const size_t n = (size_t)B * Z * H * W * Cpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const size_t pix = i / Cpad;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int img = (int)(pix / ((size_t)W * H));
    const int b = img / Z, z = img - b * Z;
    float v = 0.f;
    if (c < C) v = x[(((size_t)b * C * Z + (size_t)c * Z + z) * H + h) * W + w];
    else if (c == C && add_const) v = 1.0f;
    y[i] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void images_to_stereo_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int Z, int H, int W, int ld) {
  const size_t n = (size_t)B * C * Z * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int cz = (int)((i / ((size_t)W * H)) % (C * Z));
    const int b = (int)(i / ((size_t)W * H * C * Z));
    const int c = cz / Z, z = cz - c * Z;
    y[i] = to_f32<T>(x[((((size_t)b * Z + z) * H + h) * W + w) * ld + c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W, int ld) {
  const size_t n = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int c = (int)((i / ((size_t)W * H)) % C);
    const int b = (int)(i / ((size_t)W * H * C));
    y[i] = to_f32<T>(x[(((size_t)b * H + h) * W + w) * ld + c]);
  }
}

// ---------------------------------------------------------------------------------------------- embeddings
// reference MPFourier.forward (mp_tools.py:324-330), fp32.
__global__ void mpfourier_kernel(const float* __restrict__ x, const float* __restrict__ freqs, const float* __restrict__ phases,
                                 float* __restrict__ out, int M, int C, int logq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int b = i / C, c = i - b * C;
  float v = x[b];
  if (logq) v = logf(v) / 4.0f;
  out[i] = cosf(v * freqs[c] + phases[c]) * 1.41421356237309515f;
}

// reference mp_sum (mp_tools.py:274-279) [+ mp_silu (:268)] on [M][C] fp32 rows.
__global__ void mpsum_rows_kernel(const float* __restrict__ a, int a_rows, const float* __restrict__ b, const float* __restrict__ t_rows,
                                  float t, float* __restrict__ out, int M, int C, int silu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int r = i / C, c = i - r * C;
  const float tt = t_rows ? t_rows[r] : t;
  const float av = a[(a_rows == 1 ? 0 : r) * C + c], bv = b[i];
  float v = (av + (bv - av) * tt) / sqrtf((1.f - tt) * (1.f - tt) + tt * tt);
  if (silu) v = mp_silu_f(v);
  out[i] = v;
}

// Small-M linear layers on raw master weights: one wave per output row o of a job, looping over the M activations
// rows (M <= 16 per pass).  Weight rows are streamed once with 16-byte lanes; the squared norm for the fused
// weight-norm comes from the same pass.  reference mp_tools.py:359-367.
template <typename TW_>
__global__ __launch_bounds__(256) void linear_small_kernel(const ddx_linear_job* __restrict__ jobs, const float* __restrict__ x,
                                                           int x_stride, int M, float eps) {
  // 16 lanes per output row (16 rows per workgroup), 16-byte weight loads when K allows
  constexpr int EV = 16 / (int)sizeof(TW_);
  const ddx_linear_job jb = jobs[blockIdx.y];
  // the grid is sized for the widest job of the batch: workgroups past this job's rows leave at once (they used to stream row 0
  // through the whole K loop -- most of the batched emb_linear launch's 56 us)
  if (blockIdx.x * 16 >= jb.O) return;
  const int sub = threadIdx.x & 15;
  const int o = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool active = o < jb.O;
  const int oc = active ? o : 0;
  const TW_* wr = reinterpret_cast<const TW_*>(jb.w) + (size_t)oc * jb.K;
  const int og = jb.O / jb.groups;
  const float* xg = x + (size_t)(oc / og) * jb.K;  // grouped: channel block of this output row
  constexpr int MB = 8;
  const bool vec = (jb.K % EV) == 0 && (x_stride % 4) == 0;
  for (int m0 = 0; m0 < M; m0 += MB) {
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    float ss = 0.f;
    if (vec) {
#pragma unroll 3
      for (int k = sub * EV; k < jb.K; k += 16 * EV) {
        Vec16<TW_> wv;
        wv.v = *reinterpret_cast<const decltype(wv.v)*>(wr + k);
#pragma unroll
        for (int e = 0; e < EV; ++e) { const float f = wv.get(e); ss += f * f; }
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (m0 + m < M) {
            const float* xp = xg + (size_t)(m0 + m) * x_stride + k;
#pragma unroll
            for (int e = 0; e < EV; e += 4) {
              const f32x4 x4 = *reinterpret_cast<const f32x4*>(xp + e);
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[m] += wv.get(e + q) * x4[q];
            }
          }
      }
    } else {
      for (int k = sub; k < jb.K; k += 16) {
        const float wv = to_f32<TW_>(wr[k]);
        ss += wv * wv;
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (m0 + m < M) acc[m] += wv * xg[(size_t)(m0 + m) * x_stride + k];
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    float sc = jb.gain;
    if (jb.gain_ptr) sc *= *jb.gain_ptr;
    sc *= rsqrtf((float)jb.K);
    if (jb.normalize) sc /= (eps + sqrtf(ss) * rsqrtf((float)jb.K));
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float r = acc[m];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
      if (active && sub == 0 && m0 + m < M) jb.out[(size_t)(m0 + m) * jb.O + o] = r * sc + jb.add_const;
    }
  }
}

// 2x nearest upsample / 2x2 average pool of an NHWC tensor (reference resample_2d, mp_tools.py:71-79); H, W = OUTPUT size.
template <typename T>
__global__ __launch_bounds__(256) void resample2d_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int mode) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int nvec = C / EV;  // host guarantees C % EV == 0
  const size_t total = (size_t)B * H * W * nvec;
  const bool nearest = mode == DDX_RESAMPLE_UP || mode == DDX_RESAMPLE_DOWN_BWD;   // read one source pixel of the half-size image
  const float gain = mode == DDX_RESAMPLE_UP ? 1.0f : (mode == DDX_RESAMPLE_UP_BWD ? 1.0f : 0.25f);
  const int sH = nearest ? H / 2 : H * 2, sW = nearest ? W / 2 : W * 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    size_t pix = i / nvec;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int b = (int)(pix / H);
    Vec16<T> o;
    if (nearest) {
      o.v = *reinterpret_cast<const decltype(o.v)*>(x + (((size_t)b * sH + (h >> 1)) * sW + (w >> 1)) * C + v * EV);
      if (mode == DDX_RESAMPLE_DOWN_BWD) {
#pragma unroll
        for (int e = 0; e < EV; ++e) o.set(e, gain * o.get(e));
      }
    } else {
      const T* sp = x + (((size_t)b * sH + 2 * h) * sW + 2 * w) * C + v * EV;
      Vec16<T> a0, a1, a2, a3;
      a0.v = *reinterpret_cast<const decltype(o.v)*>(sp);
      a1.v = *reinterpret_cast<const decltype(o.v)*>(sp + C);
      a2.v = *reinterpret_cast<const decltype(o.v)*>(sp + (size_t)sW * C);
      a3.v = *reinterpret_cast<const decltype(o.v)*>(sp + (size_t)sW * C + C);
#pragma unroll
      for (int e = 0; e < EV; ++e) o.set(e, gain * ((a0.get(e) + a1.get(e)) + (a2.get(e) + a3.get(e))));
    }
    *reinterpret_cast<decltype(o.v)*>(y + i * EV) = o.v;
  }
}

// out = a*x + b*y + c*z on fp32 vectors (y, z optional): the element-wise algebra of the EDM sampler step
// (CFG lerp, Heun average, sample update + ancestral noise; reference dual_diffusion_pipeline.py:701-737).
__global__ __launch_bounds__(256) void lincomb3_kernel(const float* __restrict__ x, float a, const float* __restrict__ y, float b,
                                                       const float* __restrict__ z, float c, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = a * x[i];
    if (y) v += b * y[i];
    if (z) v += c * z[i];
    out[i] = v;
  }
}

static inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 8192); }

// ---- zero_bytes (see common.hpp)
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, size_t n16, size_t nwords, int tail2) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) reinterpret_cast<u32x4*>(p)[i] = u32x4{0u, 0u, 0u, 0u};
  for (size_t i = n16 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
  if (tail2 && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint16_t*>(p + nwords)[0] = 0;
}
int zero_bytes(void* p, size_t bytes, hipStream_t s) {
  if (!p || bytes == 0) return DDX_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 1)) return set_error(DDX_ERR_ARG, "zero_bytes: alignment");
  const size_t nwords = bytes / 4;
  const size_t n16 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : nwords / 4;
  const int blocks = (int)std::min<size_t>((std::max<size_t>(n16, nwords - n16 * 4) + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), n16, nwords, (int)((bytes & 2) != 0));
  return check_launch("zero_bytes");
}

}  // namespace ddx

using namespace ddx;

extern "C" int ddx_pixelnorm_act_fwd(const void* x, void* y, void* y_act, int64_t rows, int32_t C, float eps, int32_t dtype,
                                     ddx_stream stream) {
  if (!x || !y || rows <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "pixelnorm: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = (int)std::min<int64_t>((rows + 3) / 4, 16384);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(pixelnorm_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, (bf16*)y, (bf16*)y_act, rows, C, eps);
    else
      hipLaunchKernelGGL(pixelnorm_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, (float*)y_act, rows, C, eps);
    return check_launch("pixelnorm");
  }, stream, "pixelnorm", 0.0, (y_act ? 3.0 : 2.0) * (double)rows * C * (double)dtype_size(dtype));
}

extern "C" int ddx_pixelnorm_fwd(const void* x, void* y, int64_t rows, int32_t C, float eps, int32_t dtype, ddx_stream stream) {
  return ddx_pixelnorm_act_fwd(x, y, nullptr, rows, C, eps, dtype, stream);
}

extern "C" int ddx_unet_input_prep(const float* x_nchw, const float* sigma, const float* ln_freq_h, void* out_nhwc, int32_t B,
                                   int32_t C, int32_t H, int32_t W, int32_t Cpad, float sigma_data, int32_t dtype,
                                   ddx_stream stream) {
  if (!x_nchw || !sigma || !ln_freq_h || !out_nhwc || Cpad < C + 2) return set_error(DDX_ERR_ARG, "input_prep: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * H * W);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(input_prep_kernel<bf16>, dim3(blocks), dim3(256), 0, s, x_nchw, sigma, ln_freq_h, (bf16*)out_nhwc, B, C, H, W, Cpad, sigma_data);
    else
      hipLaunchKernelGGL(input_prep_kernel<float>, dim3(blocks), dim3(256), 0, s, x_nchw, sigma, ln_freq_h, (float*)out_nhwc, B, C, H, W, Cpad, sigma_data);
    return check_launch("input_prep");
  }, stream);
}

extern "C" int ddx_unet_output_combine(const void* y_nhwc, const float* x_in_nchw, const float* sigma, const float* x_ref_nchw,
                                       float* out_nchw, int32_t B, int32_t C, int32_t H, int32_t W, float sigma_data,
                                       int32_t dtype, ddx_stream stream) {
  if (!y_nhwc || !x_in_nchw || !sigma || !out_nchw) return set_error(DDX_ERR_ARG, "output_combine: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * C * H * W);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(output_combine_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)y_nhwc, x_in_nchw, sigma, x_ref_nchw, out_nchw, B, C, H, W, sigma_data);
    else
      hipLaunchKernelGGL(output_combine_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)y_nhwc, x_in_nchw, sigma, x_ref_nchw, out_nchw, B, C, H, W, sigma_data);
    return check_launch("output_combine");
  }, stream);
}

extern "C" int ddx_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype, ddx_stream stream) {
  if (!x || !y) return set_error(DDX_ERR_ARG, "nchw_to_nhwc: null");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * C * H * W);
    if (dtype == DDX_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(blocks), dim3(256), 0, s, x, (bf16*)y, B, C, H, W);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, s, x, (float*)y, B, C, H, W);
    return check_launch("nchw_to_nhwc");
  }, stream);
}

extern "C" int ddx_stereo_to_images(const float* x, void* y, int32_t B, int32_t C, int32_t Z, int32_t H, int32_t W, int32_t Cpad, int32_t add_const,
                                    int32_t dtype, ddx_stream stream) {
  if (!x || !y || B <= 0 || C <= 0 || Z <= 0 || Cpad < C + (add_const ? 1 : 0)) return set_error(DDX_ERR_ARG, "stereo_to_images: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * Z * H * W * Cpad);
    if (dtype == DDX_BF16) hipLaunchKernelGGL(stereo_to_images_kernel<bf16>, dim3(blocks), dim3(256), 0, s, x, (bf16*)y, B, C, Z, H, W, Cpad, add_const);
    else hipLaunchKernelGGL(stereo_to_images_kernel<float>, dim3(blocks), dim3(256), 0, s, x, (float*)y, B, C, Z, H, W, Cpad, add_const);
    return check_launch("stereo_to_images");
  }, stream);
}

extern "C" int ddx_images_to_stereo(const void* x, int32_t ld, float* y, int32_t B, int32_t C, int32_t Z, int32_t H, int32_t W, int32_t dtype,
                                    ddx_stream stream) {
  if (!x || !y || B <= 0 || C <= 0 || Z <= 0 || ld < C) return set_error(DDX_ERR_ARG, "images_to_stereo: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * C * Z * H * W);
    if (dtype == DDX_BF16) hipLaunchKernelGGL(images_to_stereo_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, y, B, C, Z, H, W, ld);
    else hipLaunchKernelGGL(images_to_stereo_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, y, B, C, Z, H, W, ld);
    return check_launch("images_to_stereo");
  }, stream);
}

extern "C" int ddx_nhwc_to_nchw_ld(const void* x, int32_t ld, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype,
                                   ddx_stream stream) {
  if (!x || !y || ld < C) return set_error(DDX_ERR_ARG, "nhwc_to_nchw: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * C * H * W);
    if (dtype == DDX_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, y, B, C, H, W, ld);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, y, B, C, H, W, ld);
    return check_launch("nhwc_to_nchw");
  }, stream);
}

extern "C" int ddx_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtype, ddx_stream stream) {
  return ddx_nhwc_to_nchw_ld(x, C, y, B, C, H, W, dtype, stream);
}

extern "C" int ddx_mpfourier(const float* x, const float* freqs, const float* phases, float* out, int32_t M, int32_t C,
                             int32_t log_sigma_quarter, ddx_stream stream) {
  if (!x || !freqs || !phases || !out || M <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "mpfourier: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(mpfourier_kernel, dim3((M * C + 255) / 256), dim3(256), 0, s, x, freqs, phases, out, M, C, log_sigma_quarter);
    return check_launch("mpfourier");
  }, stream);
}

extern "C" int ddx_mpsum_rows(const float* a, int32_t a_rows, const float* b, const float* t_rows, float t, float* out, int32_t M,
                              int32_t C, int32_t silu, ddx_stream stream) {
  if (!a || !b || !out || M <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "mpsum_rows: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(mpsum_rows_kernel, dim3((M * C + 255) / 256), dim3(256), 0, s, a, a_rows, b, t_rows, t, out, M, C, silu);
    return check_launch("mpsum_rows");
  }, stream);
}

extern "C" int ddx_resample2d(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode, int32_t dtype,
                              ddx_stream stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return set_error(DDX_ERR_ARG, "resample2d: bad args");
  if (mode < DDX_RESAMPLE_UP || mode > DDX_RESAMPLE_DOWN_BWD) return set_error(DDX_ERR_ARG, "resample2d: bad mode");
  if ((mode == DDX_RESAMPLE_UP || mode == DDX_RESAMPLE_DOWN_BWD) && ((H | W) & 1)) return set_error(DDX_ERR_ARG, "resample2d: upsampled size must be even");
  if (C % (dtype == DDX_BF16 ? 8 : 4)) return set_error(DDX_ERR_UNSUPPORTED, "resample2d: channels must fill 16-byte vectors");
  return dispatch([=](hipStream_t s) -> int {
    const size_t total = (size_t)B * H * W * (C / (dtype == DDX_BF16 ? 8 : 4));
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(resample2d_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, B, H, W, C, mode);
    else
      hipLaunchKernelGGL(resample2d_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C, mode);
    return check_launch("resample2d");
  }, stream, "resample2d", 0.0, (double)dtype_size(dtype) * B * H * W * C * ((mode == DDX_RESAMPLE_UP || mode == DDX_RESAMPLE_DOWN_BWD) ? 1.25 : 5.0));
}

extern "C" int ddx_lincomb3(const float* x, float a, const float* y, float b, const float* z, float c, float* out, int64_t n,
                            ddx_stream stream) {
  if (!x || !out || n <= 0) return set_error(DDX_ERR_ARG, "lincomb3: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(lincomb3_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, s, x, a, y, b, z, c, out, (size_t)n);
    return check_launch("lincomb3");
  }, stream, "lincomb3", 0.0, 4.0 * (double)n * (2 + (y ? 1 : 0) + (z ? 1 : 0)));
}

// ---- sampler step with device-resident scalars: the whole Heun / CFG step is one recorded plan (one hipGraph), so its per-step
// numbers (sigma rows, lerp weights, noise gain, which noise tensor) are read from device tables indexed by a device step counter.
__global__ __launch_bounds__(256) void sampler_load_kernel(const float* __restrict__ sample, float* __restrict__ x_in, float* __restrict__ x_pre,
                                                           float* __restrict__ sigma_out, const float* __restrict__ sig_table, const int* __restrict__ step,
                                                           int which, int B, int nb, size_t n_per_copy) {
  const int st = *step;
  if (blockIdx.x == 0 && threadIdx.x < nb) sigma_out[threadIdx.x] = sig_table[((size_t)st * 2 + which) * nb + threadIdx.x];
  const int copies = nb / B;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_per_copy; i += (size_t)gridDim.x * 256) {
    const float v = sample[i];
    for (int c = 0; c < copies; ++c) {
      x_in[(size_t)c * n_per_copy + i] = v;
      if (x_pre) x_pre[(size_t)c * n_per_copy + i] = v;
    }
  }
}

// out = a*x + b*y + c*z with (a, b, c) = coef[step * stride + {ia, ib, ic}] (a negative index: operand absent) and z advanced by
// step * z_step_stride elements (the step's own noise tensor): the same expression, in the same order, as lincomb3_kernel
__global__ __launch_bounds__(256) void lincomb3_dev_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                           float* __restrict__ out, size_t n, const float* __restrict__ coef, const int* __restrict__ step,
                                                           int stride, int ia, int ib, int ic, size_t z_step_stride) {
  const int st = *step;
  const float a = coef[st * stride + ia];
  const float b = ib >= 0 ? coef[st * stride + ib] : 0.f;
  const float c = ic >= 0 ? coef[st * stride + ic] : 0.f;
  const float* zz = z ? z + (size_t)st * z_step_stride : nullptr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = a * x[i];
    if (y && ib >= 0) v += b * y[i];
    if (zz && ic >= 0) v += c * zz[i];
    out[i] = v;
  }
}

__global__ void step_advance_kernel(int* step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1;
}

extern "C" int ddx_sampler_load(const float* sample, float* x_in, float* x_pre, float* sigma_out, const float* sig_table, const int32_t* step,
                                int32_t which, int32_t B, int32_t nb, int64_t n_per_copy, ddx_stream stream) {
  if (!sample || !x_in || !sigma_out || !sig_table || !step || B <= 0 || nb < B || nb % B || nb > 256 || n_per_copy <= 0 || (which != 0 && which != 1))
    return set_error(DDX_ERR_ARG, "sampler_load: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(sampler_load_kernel, dim3(grid_for((size_t)n_per_copy)), dim3(256), 0, s, sample, x_in, x_pre, sigma_out, sig_table, step, which, B, nb,
                       (size_t)n_per_copy);
    return check_launch("sampler_load");
  }, stream, "sampler_load", 0.0, 4.0 * (double)n_per_copy * (1 + (nb / B) * (x_pre ? 2 : 1)));
}

extern "C" int ddx_lincomb3_dev(const float* x, const float* y, const float* z, float* out, int64_t n, const float* coef, const int32_t* step,
                                int32_t stride, int32_t ia, int32_t ib, int32_t ic, int64_t z_step_stride, ddx_stream stream) {
  if (!x || !out || !coef || !step || n <= 0 || stride <= 0 || ia < 0 || ia >= stride || ib >= stride || ic >= stride || z_step_stride < 0)
    return set_error(DDX_ERR_ARG, "lincomb3_dev: bad args");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(lincomb3_dev_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, s, x, y, z, out, (size_t)n, coef, step, stride, ia, ib, ic, (size_t)z_step_stride);
    return check_launch("lincomb3_dev");
  }, stream, "lincomb3", 0.0, 4.0 * (double)n * (2 + ((y && ib >= 0) ? 1 : 0) + ((z && ic >= 0) ? 1 : 0)));
}

extern "C" int ddx_step_advance(int32_t* step, ddx_stream stream) {
  if (!step) return set_error(DDX_ERR_ARG, "step_advance: null counter");
  return dispatch([=](hipStream_t s) -> int {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, step);
    return check_launch("step_advance");
  }, stream, "op", 0.0, 8.0);
}

extern "C" int ddx_linear_small_batched(const ddx_linear_job* jobs_dev, int32_t njobs, int32_t max_O, const float* x, int32_t x_stride,
                                        int32_t M, int32_t w_dtype, ddx_stream stream) {
  if (!jobs_dev || njobs <= 0 || max_O <= 0 || !x || M <= 0) return set_error(DDX_ERR_ARG, "linear_small: bad args");
  return dispatch([=](hipStream_t s) -> int {
    dim3 grid((max_O + 15) / 16, njobs);
    if (w_dtype == DDX_BF16)
      hipLaunchKernelGGL(linear_small_kernel<bf16>, grid, dim3(256), 0, s, jobs_dev, x, x_stride, M, 1e-4f);
    else
      hipLaunchKernelGGL(linear_small_kernel<float>, grid, dim3(256), 0, s, jobs_dev, x, x_stride, M, 1e-4f);
    return check_launch("linear_small");
  }, stream, "linear_small");
}

// ------------------------------------------------------------------------------------------------ diffusion decoder glue
// (reference src/modules/unets/unet_edm2_ddec_mclt_b1.py:295-326 around the block stack; images are ordered n = 2*b + z)
namespace ddx {

// channels [c_in * x, psd chunk 0 .. ppf-1, 1, 0-padding] of image n; also written pair-swapped (image n ^ 1) for the
// depth-2 kernels of conv_in
template <typename T>
__global__ __launch_bounds__(256) void ddec_input_prep_kernel(const float* __restrict__ x, const float* __restrict__ xref, const float* __restrict__ sigma,
                                                              T* __restrict__ out, T* __restrict__ out_sw, int B, int H, int W, int ppf, int Cpad,
                                                              float sd, int add_const) {
  // one thread per pixel: the fp32 planes are read along W (coalesced per channel), the NHWC row leaves as whole 16-byte vectors
  // (consecutive threads write consecutive rows: one contiguous run per wave)
  constexpr int EV = 16 / (int)sizeof(T);
  const int nv = Cpad / EV;
  const size_t total = (size_t)B * 2 * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    size_t r = i / W;
    const int h = (int)(r % H); r /= H;
    const int z = (int)(r & 1), b = (int)(r >> 1);
    const float c_in = rsqrtf(sd * sd + sigma[b] * sigma[b]);
    const float* xr = xref + (((size_t)b * 2 + z) * H * ppf + (size_t)h * ppf) * W + w;
    T* o = out + i * Cpad;
    T* os = out_sw ? out_sw + ((((size_t)b * 2 + (1 - z)) * H + h) * W + w) * Cpad : nullptr;
    for (int v = 0; v < nv; ++v) {
      Vec16<T> ov;
#pragma unroll
      for (int e = 0; e < EV; ++e) {
        const int c = v * EV + e;
        float val = 0.f;
        if (c == 0) val = c_in * x[i];
        else if (c <= ppf) val = xr[(size_t)(c - 1) * W];
        else if (c == ppf + 1 && add_const) val = 1.f;
        ov.set(e, val);
      }
      *reinterpret_cast<decltype(ov.v)*>(o + v * EV) = ov.v;
      if (os) *reinterpret_cast<decltype(ov.v)*>(os + v * EV) = ov.v;
    }
  }
}

// out = [sa * a | sb * b] on channels (mp_cat), out_act = mp_silu(out)
template <typename T>
__global__ __launch_bounds__(256) void cat2_act_kernel(const T* __restrict__ a, float sa, const T* __restrict__ b, float sb, T* __restrict__ out,
                                                       T* __restrict__ out_act, size_t nrows, int C0, int C1) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int nv0 = C0 / EV, nv = (C0 + C1) / EV;
  const size_t total = nrows * nv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / nv;
    const int v = (int)(i - row * nv);
    Vec16<T> x, o, oa;
    float s;
    if (v < nv0) { x.v = *reinterpret_cast<const decltype(x.v)*>(a + row * C0 + (size_t)v * EV); s = sa; }
    else { x.v = *reinterpret_cast<const decltype(x.v)*>(b + row * C1 + (size_t)(v - nv0) * EV); s = sb; }
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const float c = to_f32<T>(from_f32<T>(x.get(e) * s));   // wa * a rounded in the tensor dtype, as mp_cat does
      o.set(e, c);
      oa.set(e, mp_silu_f(c));
    }
    *reinterpret_cast<decltype(o.v)*>(out + row * (C0 + C1) + (size_t)v * EV) = o.v;
    *reinterpret_cast<decltype(o.v)*>(out_act + row * (C0 + C1) + (size_t)v * EV) = oa.v;
  }
}

// out = [sa * a | sb * b] on channels (mp_cat, b optional), and the same rows written to the pair-swapped image
template <typename T>
__global__ __launch_bounds__(256) void cat2_swap_kernel(const T* __restrict__ a, float sa, const T* __restrict__ b, float sb, T* __restrict__ out,
                                                        T* __restrict__ out_sw, size_t rows_per_image, size_t nrows, int C0, int C1) {
  constexpr int EV = 16 / (int)sizeof(T);
  const int nv0 = C0 / EV, nv = (C0 + C1) / EV;
  const size_t total = nrows * nv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / nv;
    const int v = (int)(i - row * nv);
    Vec16<T> x, o;
    float s;
    if (v < nv0) { x.v = *reinterpret_cast<const decltype(x.v)*>(a + row * C0 + (size_t)v * EV); s = sa; }
    else { x.v = *reinterpret_cast<const decltype(x.v)*>(b + row * C1 + (size_t)(v - nv0) * EV); s = sb; }
#pragma unroll
    for (int e = 0; e < EV; ++e) o.set(e, to_f32<T>(from_f32<T>(x.get(e) * s)));   // wa * a rounded in the tensor dtype, as mp_cat does
    const size_t img = row / rows_per_image;
    const size_t row_sw = (img ^ 1) * rows_per_image + (row - img * rows_per_image);
    if (out) *reinterpret_cast<decltype(x.v)*>(out + row * (C0 + C1) + (size_t)v * EV) = o.v;
    *reinterpret_cast<decltype(x.v)*>(out_sw + row_sw * (C0 + C1) + (size_t)v * EV) = o.v;
  }
}

// D[b][z][h][w] = c_skip * x_in + c_out * y[n = 2b+z][h][w][0]
template <typename T>
__global__ __launch_bounds__(256) void ddec_output_combine_kernel(const T* __restrict__ y, int ystride, const float* __restrict__ x_in,
                                                                  const float* __restrict__ sigma, float* __restrict__ out, size_t per_b, size_t total,
                                                                  float sd) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float sg = sigma[i / per_b];
    const float den = sg * sg + sd * sd;
    out[i] = (sd * sd / den) * x_in[i] + (sg * sd * rsqrtf(den)) * to_f32<T>(y[i * ystride]);
  }
}

}  // namespace ddx

extern "C" int ddx_ddec_input_prep(const float* x, const float* x_ref, const float* sigma, void* out, void* out_swapped, int32_t B, int32_t H,
                                   int32_t W, int32_t ppf, int32_t Cpad, float sigma_data, int32_t add_const, int32_t dtype, ddx_stream stream) {
  if (!x || !x_ref || !sigma || !out || Cpad < 1 + ppf + (add_const ? 1 : 0) || Cpad % (dtype == DDX_BF16 ? 8 : 4)) return set_error(DDX_ERR_ARG, "ddec_input_prep: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)B * 2 * H * W);
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(ddec_input_prep_kernel<bf16>, dim3(blocks), dim3(256), 0, s, x, x_ref, sigma, (bf16*)out, (bf16*)out_swapped, B, H, W, ppf, Cpad, sigma_data, add_const);
    else
      hipLaunchKernelGGL(ddec_input_prep_kernel<float>, dim3(blocks), dim3(256), 0, s, x, x_ref, sigma, (float*)out, (float*)out_swapped, B, H, W, ppf, Cpad, sigma_data, add_const);
    return check_launch("ddec_input_prep");
  }, stream);
}

extern "C" int ddx_cat2_swap(const void* a, float scale_a, const void* b, float scale_b, void* out, void* out_swapped, int64_t images,
                             int64_t rows_per_image, int32_t C0, int32_t C1, int32_t dtype, ddx_stream stream) {
  if (!a || !out_swapped || images <= 0 || (images & 1) || rows_per_image <= 0 || C0 <= 0 || (C1 > 0) != (b != nullptr))
    return set_error(DDX_ERR_ARG, "cat2_swap: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (C0 % ev || C1 % ev) return set_error(DDX_ERR_UNSUPPORTED, "cat2_swap: channel counts must fill 16-byte vectors");
  return dispatch([=](hipStream_t s) -> int {
    const size_t nrows = (size_t)images * rows_per_image;
    const int blocks = grid_for(nrows * ((C0 + C1) / ev));
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(cat2_swap_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)a, scale_a, (const bf16*)b, scale_b, (bf16*)out, (bf16*)out_swapped, (size_t)rows_per_image, nrows, C0, C1);
    else
      hipLaunchKernelGGL(cat2_swap_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)a, scale_a, (const float*)b, scale_b, (float*)out, (float*)out_swapped, (size_t)rows_per_image, nrows, C0, C1);
    return check_launch("cat2_swap");
  }, stream, "cat2_swap", 0.0, (double)images * rows_per_image * (C0 + C1) * (double)dtype_size(dtype) * (out ? 3.0 : 2.0));
}

extern "C" int ddx_cat2_act(const void* a, float scale_a, const void* b, float scale_b, void* out, void* out_act, int64_t rows, int32_t C0,
                            int32_t C1, int32_t dtype, ddx_stream stream) {
  if (!a || !b || !out || !out_act || rows <= 0 || C0 <= 0 || C1 <= 0) return set_error(DDX_ERR_ARG, "cat2_act: bad args");
  const int ev = dtype == DDX_BF16 ? 8 : 4;
  if (C0 % ev || C1 % ev) return set_error(DDX_ERR_UNSUPPORTED, "cat2_act: channel counts must fill 16-byte vectors");
  return dispatch([=](hipStream_t s) -> int {
    const int blocks = grid_for((size_t)rows * ((C0 + C1) / ev));
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(cat2_act_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)a, scale_a, (const bf16*)b, scale_b, (bf16*)out, (bf16*)out_act, (size_t)rows, C0, C1);
    else
      hipLaunchKernelGGL(cat2_act_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)a, scale_a, (const float*)b, scale_b, (float*)out, (float*)out_act, (size_t)rows, C0, C1);
    return check_launch("cat2_act");
  }, stream, "cat2_act", 0.0, (double)rows * (C0 + C1) * (double)dtype_size(dtype) * 3.0);
}

extern "C" int ddx_ddec_output_combine(const void* y, int32_t y_channels, const float* x_in, const float* sigma, float* out, int32_t B,
                                       int64_t per_sample, float sigma_data, int32_t dtype, ddx_stream stream) {
  if (!y || !x_in || !sigma || !out || B <= 0 || per_sample <= 0) return set_error(DDX_ERR_ARG, "ddec_output_combine: bad args");
  return dispatch([=](hipStream_t s) -> int {
    const size_t total = (size_t)B * per_sample;
    if (dtype == DDX_BF16)
      hipLaunchKernelGGL(ddec_output_combine_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)y, y_channels, x_in, sigma, out, (size_t)per_sample, total, sigma_data);
    else
      hipLaunchKernelGGL(ddec_output_combine_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)y, y_channels, x_in, sigma, out, (size_t)per_sample, total, sigma_data);
    return check_launch("ddec_output_combine");
  }, stream);
}
