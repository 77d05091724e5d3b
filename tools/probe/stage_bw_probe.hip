// How fast can a CU stage operands into LDS, by path?  The 3x3 LDS-DMA conv kernel stages ~29 KB per K-stage and workgroup (18 KB of
// weight slices that hit L2, 11 KB of activations that stream) and its DMA half alone runs at 6.5-7 TB/s chip-wide.  Is that the
// LDS-DMA path's ceiling, and do plain global loads (-> VGPR -> ds_write_b128) of the L2-resident part go faster?
//   mode 0: weights + activations by LDS-DMA            mode 1: weights by global_load -> ds_write_b128, activations by LDS-DMA
//   mode 2: both by global_load -> ds_write_b128        mode 3: weights by global_load, no LDS write (load rate alone)
// Persistent grid of 2 workgroups x 256 CUs, 4 waves each; every stage: issue, s_waitcnt vmcnt(0), barrier (as the kernel does),
// two LDS stage buffers alternating.  WB / AB = weight / activation bytes per stage (multiples of 4 KiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void dma16(rsrc_t rs, int voff, int soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
}

// GATHER = 1: the activation pieces are 32-byte granules at a 1-KiB pixel stride (NHWC with 512 channels, 16 channels per K-stage): a piece
// touches 32 different 128-byte lines and the four consecutive stages of a unit ask for the four quarters of the same lines
template <int MODE, int WP, int AP, int GATHER = 0>   // WP / AP: 1-KiB pieces per wave and stage (weights / activations)
__global__ __launch_bounds__(256, 2) void stage_kernel(const char* wbuf, size_t wbytes, const char* abuf, size_t abytes, int nstage, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbuf), 0, (int)wbytes, 0x00020000);
  const rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abuf), 0, (int)(abytes > 0x7fffffffu ? 0x7fffffffu : abytes), 0x00020000);
  constexpr int STAGE = (WP + AP) * 4 * 1024;
  float acc = 0.f;
  // weights: every workgroup walks the same small table (L2 / L1 hits); activations: disjoint streaming regions
  const size_t a_per_wg = (size_t)nstage * AP * 4096 * (GATHER ? 8 : 1);
  const size_t a_base = ((size_t)blockIdx.x * a_per_wg) % (abytes - a_per_wg);
  for (int s = 0; s < nstage; ++s) {
    char* sb = smem + (s & 1) * STAGE;
    const int woff = (int)(((size_t)s * WP * 4096 + (size_t)(blockIdx.x & 7) * 65536) % (wbytes - WP * 4096));
    u32x4 wr[WP > 0 ? WP : 1], ar[AP > 0 ? AP : 1];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int piece = wave + 4 * i;
      if constexpr (MODE == 0) dma16(rsw, lane * 16, woff + piece * 1024, sb + piece * 1024);
      else wr[i] = *reinterpret_cast<const u32x4*>(wbuf + woff + piece * 1024 + lane * 16);
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int piece = wave + 4 * i;
      size_t off = a_base + (size_t)s * AP * 4096 + piece * 1024;
      int lo = lane * 16;
      if constexpr (GATHER) {   // unit s / 4 owns AP * 4 * 32 pixels of 1 KiB; stage s % 4 reads bytes [32 * (s % 4), +32) of each
        off = a_base + (size_t)(s >> 2) * AP * 4 * 32 * 1024 + (size_t)piece * 32 * 1024 + (s & 3) * 32;
        lo = (lane >> 1) * 1024 + (lane & 1) * 16;
      }
      if constexpr (MODE <= 1) dma16(rsa, lo, (int)off, sb + (WP * 4 + piece) * 1024);
      else ar[i] = *reinterpret_cast<const u32x4*>(abuf + off + lo);
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < WP; ++i) *reinterpret_cast<u32x4*>(sb + (wave + 4 * i) * 1024 + lane * 16) = wr[i];
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < AP; ++i) *reinterpret_cast<u32x4*>(sb + (WP * 4 + wave + 4 * i) * 1024 + lane * 16) = ar[i];
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < WP; ++i) acc += __builtin_bit_cast(float, wr[i][0] ^ wr[i][1] ^ wr[i][2] ^ wr[i][3]);
#pragma unroll
      for (int i = 0; i < AP; ++i) acc += __builtin_bit_cast(float, ar[i][0] ^ ar[i][3]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // touch the staged bytes so that nothing is dead (one 16-byte read per lane)
    acc += *reinterpret_cast<const float*>(sb + ((lane * 16 + s * 64) & (STAGE - 1)));
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int MODE, int WP, int AP, int GATHER = 0>
int run(const char* name, const char* wbuf, size_t wbytes, const char* abuf, size_t abytes, float* sink, hipStream_t st) {
  const int nstage = 64, grid = 512;
  const int lds = 2 * (WP + AP) * 4096;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<MODE, WP, AP, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stage_kernel<MODE, WP, AP, GATHER>), dim3(grid), dim3(256), lds, st, wbuf, wbytes, abuf, abytes, nstage, sink);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stage_kernel<MODE, WP, AP, GATHER>), dim3(grid), dim3(256), lds, st, wbuf, wbytes, abuf, abytes, nstage, sink);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double bytes = (double)grid * nstage * (WP + AP) * 4096;
  printf("%-44s W %2d KB A %2d KB per stage: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, WP * 4, AP * 4, us, bytes / us * 1e-6,
         bytes / us * 1e-6 * 1e12 / 256 / 2.4e9);
  return 0;
}

int main() {
  const size_t wbytes = 4u << 20, abytes = 1024u << 20;
  char *wbuf, *abuf; float* sink;
  CK(hipMalloc(&wbuf, wbytes)); CK(hipMalloc(&abuf, abytes)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(wbuf, 1, wbytes)); CK(hipMemset(abuf, 2, abytes));
  hipStream_t st; CK(hipStreamCreate(&st));
#define RUN(M, W, A, NAME) if (run<M, W, A>(NAME, wbuf, wbytes, abuf, abytes, sink, st)) return 1
#define RUNG(M, W, A, NAME) if (run<M, W, A, 1>(NAME, wbuf, wbytes, abuf, abytes, sink, st)) return 1
  RUN(0, 5, 3, "0: all LDS-DMA (conv-like mix)");
  RUN(1, 5, 3, "1: weights VGPR+ds_write, activations DMA");
  RUN(2, 5, 3, "2: all VGPR+ds_write");
  RUN(3, 5, 3, "3: all global_load, no LDS write");
  RUN(0, 8, 0, "0: weights only (L2 hits), LDS-DMA");
  RUN(1, 8, 0, "1: weights only (L2 hits), VGPR+ds_write");
  RUN(3, 8, 0, "3: weights only (L2 hits), global_load only");
  RUN(0, 0, 8, "0: activations only (streaming), LDS-DMA");
  RUN(2, 0, 8, "2: activations only (streaming), VGPR+ds_write");
  RUN(3, 0, 8, "3: activations only (streaming), global_load only");
  RUNG(0, 5, 3, "0: conv-like mix, activations GATHERED 32 B");
  RUNG(0, 0, 8, "0: activations only, GATHERED 32 B, LDS-DMA");
  RUNG(3, 0, 8, "3: activations only, GATHERED 32 B, global_load");
  RUN(0, 2, 1, "0: small stages, all LDS-DMA");
  RUN(1, 2, 1, "1: small stages, weights VGPR+ds_write");
  return 0;
}
