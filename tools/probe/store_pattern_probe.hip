// How fast does a CU retire 1-KiB store instructions, by lane -> address pattern?  The conv epilogues write 32 KB per 256-pixel unit
// and their stores were traced at 1000-2000 cycles each.  Every wave writes NI x 1 KiB, all waves of the chip together a `total`-byte
// buffer once per launch (streaming, disjoint 1-KiB pieces), 256 workgroups x 512 threads.
//   pattern 0: lane l -> piece + 16 l                                  (lane-linear, what a plain NHWC row store does)
//   pattern 1: lane l -> piece + 32 (l & 31) + 16 (l >> 5)             (register epilogue after v_permlane32_swap: same 1 KiB, halves interleaved)
//   pattern 2: 8 bytes per lane, lane l -> piece + 8 l, two instructions per KiB
//   pattern 3: as 0 with non-temporal stores                           pattern 4: as 1 with non-temporal stores
//   pattern 5: 16 bytes per lane, 32-byte runs 512 bytes apart (an NHWC row of 256 channels, 16-channel slice per pixel)
//   pattern 6: four such instructions back to back writing ADJACENT 32-byte runs (128 bytes = one cache line per pixel, rows 1 KiB apart):
//              what a register epilogue with an NHWC output of 512 channels would do for a 64-channel tile      pattern 7: two (64 bytes per pixel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* buf, size_t total, int ni) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t nwave = (size_t)gridDim.x * 8, gw = (size_t)blockIdx.x * 8 + wave;
  u32x4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  for (int i = 0; i < ni; ++i) {
    // piece index: consecutive iterations of a wave are far apart, consecutive waves adjacent (as tiles of a conv are)
    const size_t piece = ((size_t)i * nwave + gw) * 1024 % total;
    char* pp = buf + piece;
    if constexpr (PAT == 0) *reinterpret_cast<u32x4*>(pp + lane * 16) = v;
    if constexpr (PAT == 1) *reinterpret_cast<u32x4*>(pp + (lane & 31) * 32 + (lane >> 5) * 16) = v;
    if constexpr (PAT == 2) {
      *reinterpret_cast<u32x2*>(pp + lane * 8) = u32x2{v[0], v[1]};
      *reinterpret_cast<u32x2*>(pp + 512 + lane * 8) = u32x2{v[2], v[3]};
    }
    if constexpr (PAT == 3) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(pp + lane * 16));
    if constexpr (PAT == 4) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(pp + (lane & 31) * 32 + (lane >> 5) * 16));
    if constexpr (PAT == 5) {
      const size_t base = (((size_t)i * nwave + gw) * 32 * 512) % (total - 32 * 512);
      *reinterpret_cast<u32x4*>(buf + base + (lane >> 1) * 512 + (lane & 1) * 16) = v;
    }
    if constexpr (PAT == 6 || PAT == 7) {
      constexpr int NR = PAT == 6 ? 4 : 2;
      const size_t base = (((size_t)i * nwave + gw) * 32 * 1024) % (total - 32 * 1024);
#pragma unroll
      for (int r = 0; r < NR; ++r) *reinterpret_cast<u32x4*>(buf + base + (lane >> 1) * 1024 + r * 32 + (lane & 1) * 16) = v;
    }
    v[0] += 1;
  }
}

template <int PAT> int run(char* buf, size_t total, const char* what) {
  const int ni = (int)(total / 1024 / (256 * 8)) / (PAT == 6 ? 4 : PAT == 7 ? 2 : 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(store_kernel<PAT>, dim3(256), dim3(512), 0, 0, buf, total, ni);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_kernel<PAT>, dim3(256), dim3(512), 0, 0, buf, total, ni);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)ni * 256 * 8 * 1024 * (PAT == 6 ? 4 : PAT == 7 ? 2 : 1);
  printf("pattern %d (%s): %.1f us per %.0f MB = %.2f TB/s; %.0f cycles (2.4 GHz) per store instruction and wave\n", PAT, what, us, bytes / 1e6, bytes / us / 1e6,
         us * 2400.0 / ni / (PAT == 2 ? 2 : 1));
  return 0;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 90) << 20;
  char* buf;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 0, total));
  if (run<0>(buf, total, "lane-linear 16 B")) return 1;
  if (run<1>(buf, total, "half-interleaved 16 B")) return 1;
  if (run<2>(buf, total, "lane-linear 8 B x2")) return 1;
  if (run<3>(buf, total, "lane-linear 16 B nt")) return 1;
  if (run<4>(buf, total, "half-interleaved 16 B nt")) return 1;
  if (run<5>(buf, total, "32-byte runs, 512-byte stride")) return 1;
  if (run<6>(buf, total, "4 adjacent 32-byte runs per pixel row (128 B), rows 1 KiB apart")) return 1;
  if (run<7>(buf, total, "2 adjacent 32-byte runs per pixel row (64 B), rows 1 KiB apart")) return 1;
  return 0;
}
