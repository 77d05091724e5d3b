// What does a grid-wide barrier cost inside ONE persistent kernel on this part?  (Planning number for a launch-free block of small-M layers:
// a layer boundary becomes "every workgroup has stored its outputs" -> barrier -> "every workgroup may read them".)
// Sense-reversing counter barrier in global memory: one atomicAdd per workgroup + spin on a generation word; release / acquire fences so
// that data written before the barrier is visible after it across the 8 XCDs (device-scope atomics, buffer_wbl2 / inv via __threadfence).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                   // release: this workgroup's stores
    // (device-scope atomic loads: a plain load could spin on a stale line of this XCD's L2 forever)
    const unsigned g = __hip_atomic_load((unsigned*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (atomicAdd(count, 1u) == nwg - 1) {             // last arriver: reset and open the next generation
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      atomicAdd((unsigned*)gen, 1u);
    } else {
      int spins = 0;                                   // (bounded: a probe must not be able to hang the box)
      while (__hip_atomic_load((unsigned*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();                                   // acquire
  }
  __syncthreads();
}

__global__ void barrier_kernel(unsigned* count, unsigned* gen, int iters, float* data, int work) {
  float v = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < work; ++k) v = v * 1.0001f + 0.5f;          // `work` dependent FMAs between barriers
    if (threadIdx.x == 0) data[blockIdx.x] = v;                      // something to publish
    grid_barrier(count, gen, gridDim.x);
    v += data[(blockIdx.x + 1) % gridDim.x];                         // read a neighbour's value written before the barrier
  }
  if (v == 123.456f) data[0] = v;
}

int main() {
  unsigned *count, *gen; float* data;
  CK(hipMalloc(&count, 4)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&data, 1 << 16));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int nwg : {64, 128, 256, 512}) {
    for (int threads : {64, 256}) {
      for (int work : {0, 2000}) {
        CK(hipMemset(count, 0, 4)); CK(hipMemset(gen, 0, 4));
        const int iters = 200;
        void* args[] = {&count, &gen, (void*)&iters, &data, (void*)&work};
        // cooperative launch: every workgroup must be resident (the barrier would deadlock otherwise)
        CK(hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(nwg), dim3(threads), args, 0, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        CK(hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(nwg), dim3(threads), args, 0, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3d workgroups x %3d threads, %4d FMAs between barriers: %.2f us per iteration\n", nwg, threads, work, ms * 1e3 / iters);
      }
    }
  }
  return 0;
}
