// Multi-tensor weight path for training: ONE launch per phase over a device-resident job table instead of one launch per
// layer and phase.  The default UNet has 151 conv + 69 linear weights; per step each needs the forward preparation, the
// row scale + transposed preparation for the data gradient, the backward through the weight path and the forced
// normalisation (reference src/modules/mp_tools.py:359-364, :375-378; training/trainer.py:375-381) -- about 1100
// launches of 5-13 us whose payload is 1-3 us each.  The row bodies are the single-tensor ones (wpath_rows.hpp).
#include "wpath_rows.hpp"

namespace ddx {
namespace {

template <int PHASE, typename TP>
__global__ __launch_bounds__(256) void wpath_multi_kernel(const ddx_wpath_job* __restrict__ jobs, const int32_t* __restrict__ prefix, int njobs) {
  __shared__ float scratch[4];
  const int b = blockIdx.x;
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
    normalize_row<float>(reinterpret_cast<float*>(J.w), (int64_t)J.Cg * taps, eps, row, scratch);
  } else if constexpr (PHASE == DDX_WPATH_PREP) {
    wprep_row<float, TP>(w, reinterpret_cast<TP*>(J.wp), J.gain_ptr, J.gain, J.Cout, J.Cg, taps, J.groups, J.CK, J.normalize, J.qk_head_dim, eps,
                         J.in_split, J.in_scale0, J.in_scale1, row, scratch, 0, 0, J.row_scale);
  } else if constexpr (PHASE == DDX_WPATH_ROWSCALE) {
    wprep_rowscale_row<float>(w, J.row_scale, J.gain_ptr, J.gain, J.Cg * taps, J.normalize, eps, row, scratch);
  } else if constexpr (PHASE == DDX_WPATH_TRANSPOSED) {
    wprep_transposed_row<float, TP>(w, reinterpret_cast<TP*>(J.wp_t), J.row_scale, J.Cout, J.Cg, taps, J.groups, J.CK_t, J.qk_head_dim, J.in_split,
                                    J.in_scale0, J.in_scale1, row);
  } else {
    wprep_bwd_row<float>(J.dwp, w, J.gain_ptr, J.gain, J.dw, J.dgain, J.Cout, J.Cg, taps, J.groups, J.normalize, J.qk_head_dim, eps, J.in_split,
                         J.in_scale0, J.in_scale1, 0, row, scratch);
  }
}

template <int PHASE>
int launch_phase(const ddx_wpath_job* jobs, const int32_t* prefix, int njobs, int total_rows, int wp_dtype, hipStream_t s) {
  if (wp_dtype == DDX_BF16) hipLaunchKernelGGL((wpath_multi_kernel<PHASE, bf16>), dim3(total_rows), dim3(256), 0, s, jobs, prefix, njobs);
  else hipLaunchKernelGGL((wpath_multi_kernel<PHASE, float>), dim3(total_rows), dim3(256), 0, s, jobs, prefix, njobs);
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
      case DDX_WPATH_TRANSPOSED: return launch_phase<DDX_WPATH_TRANSPOSED>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
      default: return launch_phase<DDX_WPATH_BWD>(jobs_dev, row_prefix_dev, njobs, total_rows, wp_dtype, s);
    }
  }, stream, "wpath_multi");
}
