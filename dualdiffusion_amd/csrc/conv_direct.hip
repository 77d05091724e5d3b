// Scalar conv2d forward: one thread per output element, any channel counts.  Same descriptor, prepared-weight
// layout and fused prologue/epilogue as conv_mfma.hip.  Used for channel counts the vector path cannot take
// (BASELINE config 1: 4 channels per group) and as an on-device cross-check of the MFMA kernel in the tests.
#include "conv_params.hpp"

namespace ddx {

template <typename T>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p, int KS) {
  const int pad = KS / 2, taps = KS * KS;
  const size_t total = (size_t)p.B * p.H * p.W * p.Cout;
  const T* wp = reinterpret_cast<const T*>(p.wp);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int o = (int)(idx % p.Cout);
    size_t pix = idx / p.Cout;
    const int w = (int)(pix % p.W); pix /= p.W;
    const int h = (int)(pix % p.H);
    const int bo = (int)(pix / p.H);
    const int g = o / p.Ng, n = o - g * p.Ng;
    float acc = 0.f;
    for (int tap = 0; tap < taps; ++tap) {
      const int ih = h + tap / KS - pad;
      int iw = w + tap % KS - pad;
      if (p.reflect_w) iw = iw < 0 ? -iw : (iw >= p.W ? 2 * (p.W - 1) - iw : iw);
      if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
      for (int c = 0; c < p.Cg; ++c) {
        const int cabs = g * p.Cg + c;
        const bool second_half = p.paired && cabs >= p.C0 + p.C1;
        const int cpart = second_half ? cabs - (p.C0 + p.C1) : cabs;
        const bool first = cpart < p.C0;
        const T* src = reinterpret_cast<const T*>(first ? p.src0 : p.src1);
        const int Cs = first ? p.C0 : p.C1;
        const int cc = first ? cpart : cpart - p.C0;
        const int b = (second_half || (!first && p.swap1)) ? (bo ^ 1) : bo;
        float x;
        if (p.resample == DDX_RESAMPLE_DOWN) {
          const size_t base = (((size_t)b * p.sH + 2 * ih) * p.sW + 2 * iw) * Cs + cc;
          x = 0.25f * ((to_f32<T>(src[base]) + to_f32<T>(src[base + Cs])) +
                       (to_f32<T>(src[base + (size_t)p.sW * Cs]) + to_f32<T>(src[base + (size_t)p.sW * Cs + Cs])));
        } else {
          const int sh = (p.resample == DDX_RESAMPLE_UP) ? (ih >> 1) : ih;
          const int sw = (p.resample == DDX_RESAMPLE_UP) ? (iw >> 1) : iw;
          x = to_f32<T>(src[(((size_t)b * p.sH + sh) * p.sW + sw) * Cs + cc]);
        }
        x *= first ? p.scale0 : p.scale1;
        const int pro = (p.pro_rows > 0 && o >= p.pro_rows) ? DDX_PRO_NONE : p.prologue;
        if (pro & DDX_PRO_SCALE) x *= p.cscale[(size_t)bo * p.Cin + cabs];
        if (pro & DDX_PRO_SILU) x = mp_silu_f(x);
        x = to_f32<T>(from_f32<T>(x));  // the MFMA path rounds the operand to T in LDS
        acc += x * to_f32<T>(wp[wp_index(g, n, tap, c, p.nchunk, taps, p.NgP, p.CK)]);
      }
    }
    if (p.epilogue == DDX_EPI_MPSUM) {
      const size_t ridx = p.res_up ? (((size_t)bo * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Cout + o : idx;
      acc = to_f32<T>(reinterpret_cast<const T*>(p.res)[ridx]) * p.res_a + acc * p.res_b;
    }
    if (p.clip > 0.f) acc = fminf(fmaxf(acc, -p.clip), p.clip);
    if (p.out2) {  // twin: with a channel scale and a raw main output the scale belongs to the twin (training forward)
      const float tc = (p.out_cs && !p.out_act) ? p.out_cs[(size_t)bo * p.Cout + o] : 1.0f;
      reinterpret_cast<T*>(p.out2)[idx] = from_f32<T>(mp_silu_f(acc * tc * p.out2_scale));
    }
    if (p.out_act) acc = mp_silu_f(p.out_cs ? acc * p.out_cs[(size_t)bo * p.Cout + o] : acc);
    reinterpret_cast<T*>(p.out)[idx] = from_f32<T>(acc);
  }
}

int launch_conv_direct(const ConvParams& p, int ksize, int dtype, hipStream_t s) {
  const size_t total = (size_t)p.B * p.H * p.W * p.Cout;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 65536);
  if (dtype == DDX_BF16)
    hipLaunchKernelGGL(conv_direct_kernel<bf16>, dim3(blocks), dim3(256), 0, s, p, ksize);
  else
    hipLaunchKernelGGL(conv_direct_kernel<float>, dim3(blocks), dim3(256), 0, s, p, ksize);
  return check_launch("conv_direct");
}

}  // namespace ddx
