// Multi-scale 2-D spectral loss, value and gradient in one pass.
//
// Replaces MSSLoss2D.mss_loss + its autograd backward (reference src/training/loss/multiscale_spectral.py:213-294):
//   per block width w, step s: reflect-pad w/2 -> unfold into w x w blocks (64x overlap at s = w/8) -> x window ->
//   rfft2(ortho) -> mid/side stack -> weighted L1/MSE between magnitudes -> mean.
// The reference materialises the unfolded tensor (~6 GB per 45 s sample and scale) and runs cuFFT on tiny blocks, twice
// (forward and backward).  Here a workgroup keeps 4096 block pixels in LDS (1 block of 64x64 ... 64 blocks of 8x8):
//   * both stereo channels ride in ONE complex 2-D FFT (z = L + iR, un-mixed by conjugate symmetry),
//   * the loss terms AND their gradient with respect to the sample are formed on the spot:
//       dL/dx = (1/w) Re IDFT2( G ),  G_k = g_k X_k/|X_k| on the rfft2 half-spectrum, zero elsewhere,
//     made real by Hermitian symmetrisation so that the two channels again share one inverse FFT,
//   * the windowed gradient is scattered back with float atomics through the adjoint of the reflect padding.
// HBM traffic: the two input images (L2/MALL resident, re-read by the overlapping blocks) + the gradient image.
#include "fft_lds.hpp"

namespace ddx {

struct MssParams {
  const float* sample; const float* target; const float* window; const float* weight; const float2* tw;
  float* loss; float* grad;
  int B, H, Wd, step, nbh, nbw, midside, use_mse;
  float scale;  // abs_loss_scale / (channels * nbh * nbw * w * (w/2+1))
};

constexpr int kMssNT = 1024;  // 16 waves per workgroup, two workgroups (2 x 64 KB of LDS) per CU
constexpr int kMssPts = 4096;

__device__ __forceinline__ int reflect_pad_index(int j, int n) {
  if (j < 0) j = -j;
  if (j >= n) j = 2 * (n - 1) - j;
  return j;
}

// One Stockham stage (radix R) over all lines of NARR [NBLK][W][W] arrays, along rows (ROWS) or columns, IN PLACE: every
// thread pulls the inputs of its butterflies into registers, barrier, then writes the outputs over the same array (two
// barriers per stage, but no scratch array: 64 KB for sample + target instead of 96 KB, i.e. two workgroups per CU whose
// load / transform / scatter phases overlap).  Sample and target go through one pass (shared index arithmetic / twiddles).
template <int W, int R, bool INV, bool ROWS, int NARR>
__device__ __forceinline__ void mss_fft_stage(cf* __restrict__ x0, cf* __restrict__ x1, int n, int s, const float2* __restrict__ tw) {
  constexpr int BPL = W / R;  // butterflies per line
  constexpr int ITS = kMssPts / R / kMssNT;
  constexpr int ES = ROWS ? 1 : W;
  const int m = n / R;
  const int tstep = W / n;
  cf a[NARR][ITS][R];
  int base[ITS], pp[ITS], qq[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int t = threadIdx.x + it * kMssNT;
    const int line = t / BPL, u = t - line * BPL;
    pp[it] = u / s; qq[it] = u - pp[it] * s;
    // rows: line = blk*W + r, element stride 1.  columns: line = blk*W + c, element stride W.
    base[it] = ROWS ? line * W : (line / W) * (W * W) + (line % W);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int o = base[it] + ES * (qq[it] + s * (pp[it] + m * j));
      a[0][it][j] = x0[o];
      if (NARR > 1) a[NARR - 1][it][j] = x1[o];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int p = pp[it], q = qq[it];
#pragma unroll
    for (int r = 0; r < NARR; ++r) Butterfly<R, INV>::run(a[r][it]);
    const int o0 = base[it] + ES * (q + s * (R * p));
    x0[o0] = a[0][it][0];
    if (NARR > 1) x1[o0] = a[NARR - 1][it][0];
#pragma unroll
    for (int k = 1; k < R; ++k) {
      const cf w = twiddle<INV>(tw, p * k * tstep);   // p*k*tstep <= (n/R - 1)(R - 1) W/n < W
      const int o = base[it] + ES * (q + s * (R * p + k));
      x0[o] = cmul(a[0][it][k], w);
      if (NARR > 1) x1[o] = cmul(a[NARR - 1][it][k], w);
    }
  }
  __syncthreads();
}

// 1-D transforms of length W along one axis, in place
template <int W, bool INV, bool ROWS, int NARR>
__device__ __forceinline__ void mss_fft_axis(cf* x0, cf* x1, const float2* tw) {
  int n = W, s = 1;
  while (n > 1) {
    if (n % 4 == 0) { mss_fft_stage<W, 4, INV, ROWS, NARR>(x0, x1, n, s, tw); n /= 4; s *= 4; }
    else { mss_fft_stage<W, 2, INV, ROWS, NARR>(x0, x1, n, s, tw); n /= 2; s *= 2; }
  }
}
template <int W, bool INV, int NARR>
__device__ __forceinline__ void mss_fft2d(cf* x0, cf* x1, const float2* tw) {
  mss_fft_axis<W, INV, true, NARR>(x0, x1, tw);
  mss_fft_axis<W, INV, false, NARR>(x0, x1, tw);
}

template <int W>
__global__ __launch_bounds__(kMssNT, 8) void mss_loss_kernel(const MssParams p) {
  constexpr int NBLK = kMssPts / (W * W);
  constexpr int HB = W / 2 + 1;                                   // rfft2 half-spectrum width
  constexpr int NHALF = NBLK * W * HB;
  constexpr int NITEM = (NHALF + kMssNT - 1) / kMssNT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* A = reinterpret_cast<cf*>(smem);
  cf* Bt = A + kMssPts;
  float2* stw = reinterpret_cast<float2*>(Bt + kMssPts);           // W twiddles
  __shared__ float red[kMssNT / 64];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, by = blockIdx.y, bx0 = blockIdx.x * NBLK;
  const size_t plane = (size_t)p.H * p.Wd;
  const float* sL = p.sample + (size_t)b * 2 * plane; const float* sR = sL + plane;
  const float* tL = p.target + (size_t)b * 2 * plane; const float* tR = tL + plane;
  if (tid < W) stw[tid] = p.tw[tid];

  // ---- load: windowed, reflect-padded blocks; z = left + i*right
#pragma unroll
  for (int it = 0; it < kMssPts / kMssNT; ++it) {
    const int idx = tid + it * kMssNT;
    const int blk = idx / (W * W), r = (idx / W) % W, c = idx % W;
    const int bx = bx0 + blk;
    cf zs{0.f, 0.f}, zt{0.f, 0.f};
    if (bx < p.nbw) {
      const int gy = reflect_pad_index(by * p.step - W / 2 + r, p.H);
      const int gx = reflect_pad_index(bx * p.step - W / 2 + c, p.Wd);
      const float w = p.window[r * W + c];
      const size_t o = (size_t)gy * p.Wd + gx;
      zs = cf{sL[o] * w, sR[o] * w};
      zt = cf{tL[o] * w, tR[o] * w};
    }
    A[idx] = zs;
    Bt[idx] = zt;
  }
  __syncthreads();
  mss_fft2d<W, false, 2>(A, Bt, stw);

  // ---- loss terms and spectral gradient on the half spectrum
  const float inv_w = 1.0f / (float)W;
  cf gl[NITEM], gr[NITEM];
  float lsum = 0.f;
#pragma unroll
  for (int it = 0; it < NITEM; ++it) {
    const int idx = tid + it * kMssNT;
    gl[it] = cf{0.f, 0.f}; gr[it] = cf{0.f, 0.f};
    if (idx >= NHALF) continue;
    const int blk = idx / (W * HB), rem = idx - blk * (W * HB);
    const int kh = rem / HB, kw = rem - kh * HB;
    if (bx0 + blk >= p.nbw) continue;
    const int o = blk * W * W + kh * W + kw;
    const int om = blk * W * W + ((W - kh) % W) * W + ((W - kw) % W);
    auto unmix = [&](const cf* Z, cf& c0, cf& c1) {
      const cf z = Z[o], zc = cconj(Z[om]);
      const cf sum = cadd(z, zc), dif = csub(z, zc);
      const cf xl{0.5f * inv_w * sum.x, 0.5f * inv_w * sum.y};       // F(left)[k] / w
      const cf xr{0.5f * inv_w * dif.y, -0.5f * inv_w * dif.x};      // F(right)[k] / w = -i (z - zc) / 2w
      if (p.midside) { c0 = cadd(xl, xr); c1 = csub(xl, xr); } else { c0 = xl; c1 = xr; }
    };
    cf s0, s1, t0, t1;
    unmix(A, s0, s1);
    unmix(Bt, t0, t1);
    const float wgt = p.weight[kh * HB + kw];
    auto term = [&](cf s, cf t, cf& g) {
      const float as = sqrtf(s.x * s.x + s.y * s.y), at = sqrtf(t.x * t.x + t.y * t.y);
      const float d = as - at;
      lsum += wgt * (p.use_mse ? d * d : fabsf(d));
      const float gd = p.use_mse ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      const float f = as > 0.f ? p.scale * wgt * gd / as : 0.f;
      g = cf{f * s.x, f * s.y};
    };
    cf g0, g1;
    term(s0, t0, g0);
    term(s1, t1, g1);
    if (p.midside) { gl[it] = cadd(g0, g1); gr[it] = csub(g0, g1); } else { gl[it] = g0; gr[it] = g1; }
  }
  // per-sample loss: block reduce + one atomic
  lsum = wave_sum(lsum);
  if ((tid & 63) == 0) red[tid >> 6] = lsum;
  __syncthreads();  // also: every thread is done reading the spectra
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < kMssNT / 64; ++w) tot += red[w];
    atomicAdd(p.loss + b, tot * p.scale);
  }
  if (!p.grad) return;

  // ---- gradient: G on the half spectrum -> Hermitian-symmetric packed spectrum -> inverse FFT -> window -> scatter
#pragma unroll
  for (int it = 0; it < NITEM; ++it) {
    const int idx = tid + it * kMssNT;
    if (idx >= NHALF) continue;
    const int blk = idx / (W * HB), rem = idx - blk * (W * HB);
    const int kh = rem / HB, kw = rem - kh * HB;
    const int o = blk * W * W + kh * W + kw;
    A[o] = gl[it];
    Bt[o] = gr[it];
  }
  __syncthreads();
  cf pks[kMssPts / kMssNT];
#pragma unroll
  for (int it = 0; it < kMssPts / kMssNT; ++it) {
    const int idx = tid + it * kMssNT;
    const int blk = idx / (W * W), kh = (idx / W) % W, kw = idx % W;
    const int mh = (W - kh) % W, mw = (W - kw) % W;
    cf pk{0.f, 0.f};
    if (kw < HB) {  // (G_L + i G_R) / 2
      const cf a = A[idx], c = Bt[idx];
      pk.x += 0.5f * (a.x - c.y); pk.y += 0.5f * (a.y + c.x);
    }
    if (mw < HB) {  // (conj(G_L[-k]) + i conj(G_R[-k])) / 2
      const int om = blk * W * W + mh * W + mw;
      const cf a = A[om], c = Bt[om];
      pk.x += 0.5f * (a.x + c.y); pk.y += 0.5f * (c.x - a.y);
    }
    pks[it] = pk;
  }
  __syncthreads();  // every thread has read its G entries: the packed spectrum replaces them in A
#pragma unroll
  for (int it = 0; it < kMssPts / kMssNT; ++it) A[tid + it * kMssNT] = pks[it];
  __syncthreads();
  mss_fft2d<W, true, 1>(A, nullptr, stw);
  float* gL = p.grad + (size_t)b * 2 * plane; float* gR = gL + plane;
#pragma unroll
  for (int it = 0; it < kMssPts / kMssNT; ++it) {
    const int idx = tid + it * kMssNT;
    const int blk = idx / (W * W), r = (idx / W) % W, c = idx % W;
    const int bx = bx0 + blk;
    if (bx >= p.nbw) continue;
    const int gy = reflect_pad_index(by * p.step - W / 2 + r, p.H);
    const int gx = reflect_pad_index(bx * p.step - W / 2 + c, p.Wd);
    const float w = p.window[r * W + c] * inv_w;
    const size_t o = (size_t)gy * p.Wd + gx;
    const cf v = A[idx];
    unsafeAtomicAdd(gL + o, v.x * w);
    unsafeAtomicAdd(gR + o, v.y * w);
  }
}

template <int W>
static int launch_mss(const MssParams& p, hipStream_t s) {
  constexpr int NBLK = kMssPts / (W * W);
  const size_t smem = 2 * kMssPts * sizeof(cf) + W * sizeof(float2);
  auto kern = mss_loss_kernel<W>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(mss_loss)");
    attr_done = true;
  }
  dim3 grid(ceil_div(p.nbw, NBLK), p.nbh, p.B);
  hipLaunchKernelGGL(kern, grid, dim3(kMssNT), smem, s, p);
  return check_launch("mss_loss");
}

}  // namespace ddx

using namespace ddx;

extern "C" int ddx_mss_loss_scale(const ddx_mss_desc* dp, ddx_stream stream) {
  if (!dp) return set_error(DDX_ERR_ARG, "mss_loss: null descriptor");
  const ddx_mss_desc d = *dp;
  if (!d.sample || !d.target || !d.window || !d.weight || !d.twiddle || !d.loss) return set_error(DDX_ERR_ARG, "mss_loss: null buffer");
  if (d.C != 2) return set_error(DDX_ERR_UNSUPPORTED, "mss_loss: stereo (C = 2) only");
  const int w = d.block_width;
  if (w != 8 && w != 16 && w != 32 && w != 64) return set_error(DDX_ERR_UNSUPPORTED, "mss_loss: block width must be 8, 16, 32 or 64");
  if (d.B <= 0 || d.step <= 0 || w / 2 >= d.H || w / 2 >= d.W) return set_error(DDX_ERR_ARG, "mss_loss: bad size");
  MssParams p{};
  p.sample = d.sample; p.target = d.target; p.window = d.window; p.weight = d.weight;
  p.tw = reinterpret_cast<const float2*>(d.twiddle);
  p.loss = d.loss; p.grad = d.grad;
  p.B = d.B; p.H = d.H; p.Wd = d.W; p.step = d.step;
  p.nbh = d.H / d.step + 1; p.nbw = d.W / d.step + 1;  // unfold count of the (H + w)-padded axis
  p.midside = d.midside; p.use_mse = d.use_mse;
  p.scale = d.loss_scale / ((float)2 * p.nbh * p.nbw * w * (w / 2 + 1));
  const double blocks = (double)p.B * p.nbh * p.nbw;
  const double flops = blocks * 3.0 * 2.0 * w * (5.0 * w * log2((double)w));  // three complex 2-D FFTs per block
  const double bytes = (double)p.B * 2 * d.H * d.W * 4 * (d.grad ? 3 : 2);
  return dispatch([p, w](hipStream_t s) -> int {
    switch (w) {
      case 8: return launch_mss<8>(p, s);
      case 16: return launch_mss<16>(p, s);
      case 32: return launch_mss<32>(p, s);
      default: return launch_mss<64>(p, s);
    }
  }, stream, "mss_loss", flops, bytes);
}
