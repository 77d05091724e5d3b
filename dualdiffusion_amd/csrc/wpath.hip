// Multi-tensor weight path for training: ONE launch per phase over a device-resident job table instead of one launch per
// layer and phase.  The default UNet has 151 conv + 69 linear weights; per step each needs the forward preparation, the
// row scale + transposed preparation for the data gradient, the backward through the weight path and the forced
// normalisation (reference src/modules/mp_tools.py:359-364, :375-378; training/trainer.py:375-381) -- about 1100
// launches of 5-13 us whose payload is 1-3 us each.  The row bodies are the single-tensor ones (wpath_rows.hpp).
#include "wpath_rows.hpp"

namespace ddx {
namespace {

// One WAVE per row, four rows per workgroup (row = 4 * blockIdx.x + wave): the rows of the 220 weight tensors are 1-15 KB each, and a
// 256-thread workgroup per row spent most of its time in the table search, two block reductions and 4-byte accesses.
template <int PHASE, typename TP>
__global__ __launch_bounds__(256) void wpath_multi_kernel(const ddx_wpath_job* __restrict__ jobs, const int32_t* __restrict__ prefix, int njobs, int total_rows) {
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (b >= total_rows) return;
  int lo = 0, hi = njobs;  // prefix[lo] <= b < prefix[hi]; zero-row jobs share their successor's prefix and are skipped
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= b) lo = mid; else hi = mid;
  }
  const ddx_wpath_job J = jobs[lo];
  const int row = b - prefix[lo];
  const int taps = J.ksize * J.ksize;
  const float* w = reinterpret_cast<const float*>(J.w);
  constexpr float eps = 1e-4f;
  if constexpr (PHASE == DDX_WPATH_NORMALIZE) {
    normalize_row_w(reinterpret_cast<float*>(J.w), (int64_t)J.Cg * taps, eps, row, lane);
  } else if constexpr (PHASE == DDX_WPATH_PREP) {
    wprep_row_w<TP>(w, reinterpret_cast<TP*>(J.wp), J.gain_ptr, J.gain, J.Cout, J.Cg, taps, J.groups, J.CK, J.normalize, J.qk_head_dim, eps,
                    J.in_split, J.in_scale0, J.in_scale1, row, lane, J.row_scale);
  } else if constexpr (PHASE == DDX_WPATH_ROWSCALE) {
    wprep_rowscale_row_w(w, J.row_scale, J.gain_ptr, J.gain, J.Cg * taps, J.normalize, eps, row, lane);
  } else if constexpr (PHASE == DDX_WPATH_TRANSPOSED) {
    wprep_transposed_row_w<TP>(w, reinterpret_cast<TP*>(J.wp_t), J.row_scale, J.Cout, J.Cg, taps, J.groups, J.CK_t, J.qk_head_dim, J.in_split,
                               J.in_scale0, J.in_scale1, row, lane);
  } else {
    __shared__ __attribute__((aligned(16))) float rowbuf[4][kWpathRowBuf];
    if (J.dwp_parts > 1)
      wprep_bwd_row_parts_w(J.dwp, J.dwp_parts, w, J.gain_ptr, J.gain, J.dw, J.dgain, J.Cout, J.Cg, taps, J.groups, J.normalize, J.qk_head_dim, eps,
                            J.in_split, J.in_scale0, J.in_scale1, row, lane, rowbuf[threadIdx.x >> 6]);
    else
      wprep_bwd_row_w(J.dwp, w, J.gain_ptr, J.gain, J.dw, J.dgain, J.Cout, J.Cg, taps, J.groups, J.normalize, J.qk_head_dim, eps, J.in_split,
                      J.in_scale0, J.in_scale1, row, lane);
  }
}

// Data-gradient (transposed) preparation through an LDS tile: the destination rows are the INPUT channels of the forward conv, so a
// row gathers one 4*taps-byte piece from every source row.  A workgroup takes 8 consecutive destination rows (input channels c0..c0+7 of
// one group): for 64 output channels at a time it reads the 8*taps contiguous floats of each source row (coalesced), transposes in
// LDS and writes, per (channel, tap), 64 consecutive output channels of the prepared layout (coalesced).  Rows that do not form such
// a block (a job whose channel count is not a multiple of 8) take the per-wave gather.
template <typename TP>
__global__ __launch_bounds__(256) void wpath_transposed_kernel(const ddx_wpath_job* __restrict__ jobs, const int32_t* __restrict__ prefix, int njobs, int total_rows) {
  constexpr int NB = 64, CB = 8, MAXT = 9;
  __shared__ float tile[NB][CB * MAXT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * CB;
  auto find = [&](int b) {
    int lo = 0, hi = njobs;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
  };
  const int j0 = find(r0);
  const ddx_wpath_job J = jobs[j0];
  const int row0 = r0 - prefix[j0];
  const int taps = J.ksize * J.ksize, Cin = J.Cg * J.groups;
  const float* w = reinterpret_cast<const float*>(J.w);
  const bool tiled = (row0 % CB) == 0 && (J.Cg % CB) == 0 && row0 + CB <= Cin && taps <= MAXT && r0 + CB <= total_rows;
  if (!tiled) {
    for (int rr = wave; rr < CB; rr += 4) {
      const int b = r0 + rr;
      if (b >= total_rows) continue;
      const int jb = find(b);
      const ddx_wpath_job K = jobs[jb];
      wprep_transposed_row_w<TP>(reinterpret_cast<const float*>(K.w), reinterpret_cast<TP*>(K.wp_t), K.row_scale, K.Cout, K.Cg, K.ksize * K.ksize, K.groups,
                                 K.CK_t, K.qk_head_dim, K.in_split, K.in_scale0, K.in_scale1, b - prefix[jb], lane);
    }
    return;
  }
  const int Ng = J.Cout / J.groups, CgP = (J.Cg + 31) / 32 * 32, CK = J.CK_t, nchunk = (Ng + CK - 1) / CK;
  const int g = row0 / J.Cg, c0 = row0 - g * J.Cg;
  const int span = CB * taps;                 // contiguous floats of one source row
  TP* wp = reinterpret_cast<TP*>(J.wp_t);
  for (int n0 = 0; n0 < Ng; n0 += NB) {
    for (int idx = tid; idx < NB * span; idx += 256) {
      const int nl = idx / span, k = idx - nl * span;
      float v = 0.f;
      if (n0 + nl < Ng) {
        const int os = wpath_src_row(g * Ng + n0 + nl, J.qk_head_dim);
        const int ci = row0 + k / taps;
        const float cscale = J.in_split > 0 ? (ci < J.in_split ? J.in_scale0 : J.in_scale1) : 1.0f;
        v = w[((size_t)os * J.Cg + c0) * taps + k] * J.row_scale[os] * cscale;
      }
      tile[nl][k] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < NB * span; idx += 256) {
      const int nl = idx & (NB - 1), ct = idx >> 6;
      const int cl = ct / taps, tap = ct - cl * taps;
      if (n0 + nl < Ng) wp[wp_index(g, c0 + cl, taps - 1 - tap, n0 + nl, nchunk, taps, CgP, CK)] = from_f32<TP>(tile[nl][ct]);
    }
    __syncthreads();
  }
}

template <int PHASE>
int launch_phase(const ddx_wpath_job* jobs, const int32_t* prefix, int njobs, int total_rows, int wp_dtype, hipStream_t s) {
  if (wp_dtype == DDX_BF16) hipLaunchKernelGGL((wpath_multi_kernel<PHASE, bf16>), dim3((total_rows + 3) / 4), dim3(256), 0, s, jobs, prefix, njobs, total_rows);
  else hipLaunchKernelGGL((wpath_multi_kernel<PHASE, float>), dim3((total_rows + 3) / 4), dim3(256), 0, s, jobs, prefix, njobs, total_rows);
  return check_launch("wpath_multi");
}

}  // namespace
}  // namespace ddx

using namespace ddx;

extern "C" int ddx_wpath_multi(const ddx_wpath_job* jobs_dev, const int32_t* row_prefix_dev, int32_t njobs, int32_t total_rows, int32_t phase,
                               int32_t wp_dtype, ddx_stream stream) {
  if (!jobs_dev || !row_prefix_dev || njobs <= 0 || total_rows < 0) return set_error(DDX_ERR_ARG, "wpath_multi: bad args");
  if (phase < DDX_WPATH_NORMALIZE || phase > DDX_WPATH_BWD) return set_error(DDX_ERR_ARG, "wpath_multi: bad phase");
  if (wp_dtype != DDX_BF16 && wp_dtype != DDX_F32) return set_error(DDX_ERR_ARG, "wpath_multi: bad dtype");
  if (total_rows == 0) return DDX_OK;
  return dispatch([=](hipStream_t s) -> int {
    switch (phase) {
      case DDX_WPATH_NORMALIZE: return launch_phase<DDX_WPATH_NORMALIZE>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
      case DDX_WPATH_PREP: return launch_phase<DDX_WPATH_PREP>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
      case DDX_WPATH_ROWSCALE: return launch_phase<DDX_WPATH_ROWSCALE>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
      case DDX_WPATH_TRANSPOSED:
        if (wp_dtype == DDX_BF16) hipLaunchKernelGGL(wpath_transposed_kernel<bf16>, dim3((total_rows + 7) / 8), dim3(256), 0, s, jobs_dev, row_prefix_dev, njobs, total_rows);
        else hipLaunchKernelGGL(wpath_transposed_kernel<float>, dim3((total_rows + 7) / 8), dim3(256), 0, s, jobs_dev, row_prefix_dev, njobs, total_rows);
        return check_launch("wpath_multi(transposed)");
      default: return launch_phase<DDX_WPATH_BWD>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
    }
  }, stream, "wpath_multi");
}
