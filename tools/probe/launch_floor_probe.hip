// How long does a kernel that does (almost) nothing take inside a hipGraph chain, as a function of workgroup size, LDS
// allocation and grid?  (Is the 10-15 us of the small-M conv layers dispatch cost of 1024-thread / 100 KB workgroups?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void null_kernel(float* out, int spin) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float v = lds[(threadIdx.x + 1) % blockDim.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;   // dependent chain: ~4 cycles per iteration
  if (v == 123.456f) out[blockIdx.x] = v;
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Cfg { int grid, block, lds, spin; };
  const Cfg cfgs[] = {{160, 1024, 128 * 1024, 0}, {160, 1024, 0, 0}, {160, 256, 32 * 1024, 0}, {160, 256, 0, 0}, {640, 256, 32 * 1024, 0},
                      {256, 256, 0, 0}, {160, 1024, 128 * 1024, 2500}, {160, 256, 32 * 1024, 2500}, {640, 256, 32 * 1024, 2500}, {1, 64, 0, 0}};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(null_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (const Cfg& c : cfgs) {
    const int N = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(null_kernel, dim3(c.grid), dim3(c.block), c.lds, s, out, c.spin);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid %4d block %4d lds %6d spin %5d : %.2f us per kernel\n", c.grid, c.block, c.lds, c.spin, ms * 1e3 / (reps * N));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
