// Magnitude-preserving conv2d forward as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces F.conv2d in MPConv.forward (reference src/modules/mp_tools.py:366-373) together with the
// element-wise neighbours of Block.forward (src/modules/unets/unet_edm2_b4.py:110-158).
//
// Work decomposition (one 256-thread workgroup = 4 waves):
//   output tile  = TH x TW pixels of one image (<= BM = WM*MF*32)  x  BN = WN*NF*32 output channels of one group
//   K loop       = chunks of CK input channels of the group; per chunk
//       stage   : the (TH+2)x(TW+2) input halo of the chunk -> LDS, ONCE (not 9x im2col), with the fused
//                 prologue (mp_cat scales, y*c, mp_silu, nearest-up / avg-pool gather) applied in fp32 on the way
//                 the chunk of prepared weights [tap][BN][CK] -> LDS
//       compute : for each of the ks*ks taps the activation fragment is the same LDS tile read at a shifted row,
//                 v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (fp32 parity path), fp32 accumulate
//   MFMA orientation: A operand = weights (rows = output channels), B operand = activations (cols = pixels), so
//   every lane ends up with 4 consecutive output channels of ONE pixel -> vector epilogue on NHWC rows.
//   LDS rows are padded by 16 B: a row stride of CK*sizeof(T)+16 bytes makes the 16-lane groups of ds_read_b128 hit
//   16 distinct bank slots (stride 80 B -> slot = 5r mod 16, stride 144 B -> 9r mod 16; both bijective).
#include "conv_params.hpp"

namespace ddx {

template <typename T> struct Mfma;
template <> struct Mfma<bf16> {
  static constexpr int KM = 16;
  using Frag = bf16x8;
  static __device__ __forceinline__ Frag load(const bf16* row_k, int khalf) { return *(const bf16x8*)(row_k + khalf * 8); }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static constexpr int KM = 2;
  using Frag = float;
  static __device__ __forceinline__ Frag load(const float* row_k, int khalf) { return row_k[khalf]; }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

template <typename T, int KS, int CK, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int EV = 16 / (int)sizeof(T);
  constexpr int V = CK / EV;
  constexpr int STRIDE = CK + EV;
  constexpr int BN = WN * NF * 32;
  constexpr int KM = Mfma<T>::KM;
  using Frag = typename Mfma<T>::Frag;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sA = reinterpret_cast<T*>(smem);
  T* sB = sA + (size_t)p.arows_alloc * STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int khalf = lane >> 5, l31 = lane & 31;

  // tile decode: blockIdx.x = (b, tile_y, tile_x), blockIdx.y = n tile, blockIdx.z = group
  int bx = blockIdx.x;
  const int tx = bx % p.tiles_w; bx /= p.tiles_w;
  const int ty = bx % p.tiles_h;
  const int b = bx / p.tiles_h;
  const int n0 = blockIdx.y * BN;
  const int g = blockIdx.z;
  const int h0 = ty * p.TH, w0 = tx * p.TW;
  const int TW = p.TW, TWP = TW + 2 * PAD;
  const int R = (p.TH + 2 * PAD) * TWP;
  const int MT = p.TH * TW;
  const float inv_TW = 1.0f / (float)TW;

  int arow[MF];
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int ml = (wm * MF + j) * 32 + l31;
    const int th = (int)(((float)ml + 0.5f) * inv_TW);
    const int tw = ml - th * TW;
    arow[j] = (ml < MT) ? th * TWP + tw : 0;
  }

  f32x16 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int v = tid % V;
  const T* wp = reinterpret_cast<const T*>(p.wp);

  for (int ch = 0; ch < p.nchunk; ++ch) {
    // ------------------------------------------------------------ stage the activation halo (fused prologue)
    {
      const int cg = ch * CK + v * EV;  // channel inside the group
      const bool cvalid = cg < p.Cg;
      const int cabs = g * p.Cg + cg;   // channel of the (virtually concatenated) input
      const bool first = cabs < p.C0;
      const T* src = reinterpret_cast<const T*>(first ? p.src0 : p.src1);
      const int Cs = first ? p.C0 : p.C1;
      const int cc = first ? cabs : cabs - p.C0;
      const float sscale = first ? p.scale0 : p.scale1;
      float cs[EV];
#pragma unroll
      for (int e = 0; e < EV; ++e) cs[e] = sscale;
      if ((p.prologue & DDX_PRO_SCALE) && cvalid) {
        const float* csp = p.cscale + (size_t)b * p.Cin + cabs;
#pragma unroll
        for (int e = 0; e < EV; e += 4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(csp + e);
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[e + q] *= t4[q];
        }
      }
      const bool do_silu = (p.prologue & DDX_PRO_SILU) != 0;
      for (int r = tid / V; r < R; r += 256 / V) {
        const int hh = (int)(((float)r + 0.5f) * p.inv_TWP);
        const int ww = r - hh * TWP;
        const int ih = h0 - PAD + hh, iw = w0 - PAD + ww;
        Vec16<T> o;
        if (cvalid && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
          float f[EV];
          if (p.resample == DDX_RESAMPLE_DOWN) {
            const size_t base = (((size_t)b * p.sH + 2 * ih) * p.sW + 2 * iw) * Cs + cc;
            Vec16<T> a0, a1, a2, a3;
            a0.v = *reinterpret_cast<const decltype(a0.v)*>(src + base);
            a1.v = *reinterpret_cast<const decltype(a0.v)*>(src + base + Cs);
            a2.v = *reinterpret_cast<const decltype(a0.v)*>(src + base + (size_t)p.sW * Cs);
            a3.v = *reinterpret_cast<const decltype(a0.v)*>(src + base + (size_t)p.sW * Cs + Cs);
#pragma unroll
            for (int e = 0; e < EV; ++e) f[e] = 0.25f * ((a0.get(e) + a1.get(e)) + (a2.get(e) + a3.get(e)));
          } else {
            const int sh = (p.resample == DDX_RESAMPLE_UP) ? (ih >> 1) : ih;
            const int sw = (p.resample == DDX_RESAMPLE_UP) ? (iw >> 1) : iw;
            Vec16<T> a0;
            a0.v = *reinterpret_cast<const decltype(a0.v)*>(src + (((size_t)b * p.sH + sh) * p.sW + sw) * Cs + cc);
#pragma unroll
            for (int e = 0; e < EV; ++e) f[e] = a0.get(e);
          }
#pragma unroll
          for (int e = 0; e < EV; ++e) {
            float x = f[e] * cs[e];
            if (do_silu) x = mp_silu_f(x);
            o.set(e, x);
          }
        } else {
#pragma unroll
          for (int e = 0; e < EV; ++e) o.set(e, 0.f);
        }
        *reinterpret_cast<decltype(o.v)*>(sA + (size_t)r * STRIDE + v * EV) = o.v;
      }
    }
    // ------------------------------------------------------------ stage the weights of this chunk
    {
      const T* wsrc = wp + ((size_t)(g * p.nchunk + ch) * TAPS * p.NgP) * CK;
      for (int idx = tid; idx < TAPS * BN * V; idx += 256) {
        const int v2 = idx % V;
        const int rn = idx / V;
        const int tap = rn / BN;
        const int n = rn - tap * BN;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (n0 + n < p.NgP) val = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(tap * p.NgP + n0 + n)) * CK + v2 * EV);
        *reinterpret_cast<uint4*>(sB + (size_t)rn * STRIDE + v2 * EV) = val;
      }
    }
    __syncthreads();
    // ------------------------------------------------------------ tensor-core part
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (tap / KS) * TWP + (tap % KS);
#pragma unroll
      for (int ks = 0; ks < CK / KM; ++ks) {
        Frag wf[NF], xf[MF];
#pragma unroll
        for (int i = 0; i < NF; ++i)
          wf[i] = Mfma<T>::load(sB + (size_t)(tap * BN + (wn * NF + i) * 32 + l31) * STRIDE + ks * KM, khalf);
#pragma unroll
        for (int j = 0; j < MF; ++j) xf[j] = Mfma<T>::load(sA + (size_t)(arow[j] + toff) * STRIDE + ks * KM, khalf);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int j = 0; j < MF; ++j) acc[i][j] = Mfma<T>::mma(wf[i], xf[j], acc[i][j]);
      }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue: lane owns pixel (col), 4-channel runs
  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int ml = (wm * MF + j) * 32 + l31;
    const int th = (int)(((float)ml + 0.5f) * inv_TW);
    const int tw = ml - th * TW;
    const int h = h0 + th, w = w0 + tw;
    if (ml >= MT || h >= p.H || w >= p.W) continue;
    const size_t pix = (((size_t)b * p.H + h) * p.W + w) * p.Cout + (size_t)g * p.Ng;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + (wn * NF + i) * 32 + 8 * q + 4 * khalf;
        if (n >= p.Ng) continue;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = acc[i][j][4 * q + e];
        if (p.epilogue == DDX_EPI_MPSUM) {
          Vec4<T> rv;
          rv.v = *reinterpret_cast<const decltype(rv.v)*>(res + pix + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = rv.get(e) * p.res_a + y[e] * p.res_b;
        }
        if (p.clip > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fminf(fmaxf(y[e], -p.clip), p.clip);
        }
        Vec4<T> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.set(e, y[e]);
        *reinterpret_cast<decltype(ov.v)*>(out + pix + n) = ov.v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- host side

struct TileCfg { int WM, WN, MF, NF; };

static inline int elem_vec(int dtype) { return dtype == DDX_BF16 ? 8 : 4; }

bool conv_mfma_supported(const ConvParams& p, int ksize, int dtype) {
  const int ev = elem_vec(dtype);
  if (ksize != 1 && ksize != 3) return false;
  if (p.C0 % ev || (p.src1 && (p.C1 % ev))) return false;
  if (p.Cg % ev) return false;
  if (p.Ng % 4 || p.Cout % 4) return false;
  if ((p.prologue & DDX_PRO_SCALE) && (p.Cin % 4)) return false;
  if (p.CK != 32 && p.CK != 64) return false;
  return true;
}

// best TH x TW (TH*TW <= BM) for an H x W image: maximise useful fraction, then minimise halo
static void best_tile(int H, int W, int BM, int pad, int max_rows, int* TH, int* TW, double* util) {
  double best = -1, best_halo = 1e30;
  int bth = 1, btw = 1;
  for (int tw = 1; tw <= W && tw <= BM; ++tw) {
    for (int th = 1; th <= H && th * tw <= BM; ++th) {
      const int rows = (th + 2 * pad) * (tw + 2 * pad);
      if (rows > max_rows) continue;
      const long tiles = (long)ceil_div(H, th) * ceil_div(W, tw);
      const double u = (double)H * W / ((double)tiles * BM);
      const double halo = (double)rows / (th * tw);
      if (u > best + 1e-9 || (u > best - 1e-9 && halo < best_halo)) { best = u; best_halo = halo; bth = th; btw = tw; }
    }
  }
  *TH = bth; *TW = btw; *util = best;
}

struct Choice { int BM, BN, TH, TW; size_t smem; };

static Choice choose(const ConvParams& p, int ksize, int dtype) {
  const int pad = ksize / 2, taps = ksize * ksize;
  const size_t es = dtype_size(dtype);
  const int stride = p.CK + elem_vec(dtype);
  const int bms[2] = {256, 128};
  Choice best{}; double best_score = -1;
  for (int bi = 0; bi < 2; ++bi) {
    const int BM = bms[bi];
    const int bns256[3] = {96, 64, 32};
    const int bns128[1] = {64};
    const int* bns = BM == 256 ? bns256 : bns128;
    const int nb = BM == 256 ? 3 : 1;
    for (int ni = 0; ni < nb; ++ni) {
      const int BN = bns[ni];
      int TH, TW; double um;
      best_tile(p.H, p.W, BM, pad, 400, &TH, &TW, &um);
      const int rows = (TH + 2 * pad) * (TW + 2 * pad);
      const size_t smem = ((size_t)rows + (size_t)taps * BN) * stride * es;
      if (smem > 160 * 1024) continue;
      const double un = (double)p.Ng / round_up(p.Ng, BN);
      // efficiency prior: wider N tiles amortise LDS reads; two workgroups per CU hide the staging
      double eff = (BN == 32 ? 0.80 : 1.0) * (smem <= 80 * 1024 ? 1.0 : 0.85) * (BM == 128 ? 0.92 : 1.0);
      const long wgs = (long)p.B * ceil_div(p.H, TH) * ceil_div(p.W, TW) * ceil_div(p.Ng, BN) * p.G;
      if (wgs < 256) eff *= (0.5 + 0.5 * (double)wgs / 256.0);
      const double score = um * un * eff;
      if (score > best_score) { best_score = score; best = Choice{BM, BN, TH, TW, smem}; }
    }
  }
  return best;
}

void conv_mfma_plan_tiles(ConvParams& p, int ksize, int dtype) {
  const Choice c = choose(p, ksize, dtype);
  const int pad = ksize / 2;
  p.TH = c.TH; p.TW = c.TW;
  p.tiles_h = ceil_div(p.H, c.TH); p.tiles_w = ceil_div(p.W, c.TW);
  p.arows_alloc = (c.TH + 2 * pad) * (c.TW + 2 * pad);
  p.inv_TWP = 1.0f / (float)(c.TW + 2 * pad);
}

template <typename T, int KS, int CK, int WM, int WN, int MF, int NF>
static int launch_cfg(const ConvParams& p, size_t smem, hipStream_t s) {
  auto kern = conv_mfma_kernel<T, KS, CK, WM, WN, MF, NF>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(DDX_ERR_LAUNCH, "hipFuncSetAttribute(conv_mfma)");
    attr_done = true;
  }
  constexpr int BN = WN * NF * 32;
  dim3 grid(p.B * p.tiles_h * p.tiles_w, ceil_div(p.Ng, BN), p.G);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  return check_launch("conv_mfma");
}

template <typename T, int KS, int CK>
static int launch_ks(const ConvParams& p, const Choice& c, hipStream_t s) {
  if (c.BM == 256 && c.BN == 96) return launch_cfg<T, KS, CK, 4, 1, 2, 3>(p, c.smem, s);
  if (c.BM == 256 && c.BN == 64) return launch_cfg<T, KS, CK, 4, 1, 2, 2>(p, c.smem, s);
  if (c.BM == 256 && c.BN == 32) return launch_cfg<T, KS, CK, 4, 1, 2, 1>(p, c.smem, s);
  if (c.BM == 128 && c.BN == 64) return launch_cfg<T, KS, CK, 2, 2, 2, 1>(p, c.smem, s);
  return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: no tile configuration");
}

template <typename T>
static int launch_t(const ConvParams& p, int ksize, const Choice& c, hipStream_t s) {
  if (ksize == 3 && p.CK == 32) return launch_ks<T, 3, 32>(p, c, s);
  if (ksize == 1 && p.CK == 64) return launch_ks<T, 1, 64>(p, c, s);
  if (ksize == 1 && p.CK == 32) return launch_ks<T, 1, 32>(p, c, s);
  return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: ksize/CK combination not built");
}

int launch_conv_mfma(const ConvParams& p_in, int ksize, int dtype, hipStream_t s) {
  ConvParams p = p_in;
  const Choice c = choose(p, ksize, dtype);
  if (c.BM == 0) return set_error(DDX_ERR_UNSUPPORTED, "conv_mfma: no tile fits LDS");
  conv_mfma_plan_tiles(p, ksize, dtype);
  if (dtype == DDX_BF16) return launch_t<bf16>(p, ksize, c, s);
  return launch_t<float>(p, ksize, c, s);
}

}  // namespace ddx
