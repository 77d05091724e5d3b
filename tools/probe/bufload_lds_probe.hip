#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const char* src, char* dst, int n, int so) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n, 0x00020000);
  int voff = (lane & 1) ? (wave * 64 + lane) * 16 : 0x7fffff00;
  *(uint4*)(smem + threadIdx.x * 16) = make_uint4(0xAAAAAAAA, 0xAAAAAAAA, 0xAAAAAAAA, 0xAAAAAAAA);
  __syncthreads();
  char* lbase = smem + wave * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lbase, 16, voff, so, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  *(uint4*)(dst + threadIdx.x * 16) = *(uint4*)(smem + threadIdx.x * 16);
}
int main() {
  char *s, *d; const int N = 256 * 16;
  hipMalloc(&s, N); hipMalloc(&d, N);
  unsigned char h[N]; for (int i = 0; i < N; ++i) h[i] = (i / 16) & 0xff;
  hipMemcpy(s, h, N, hipMemcpyHostToDevice); hipMemset(d, 0xEE, N);
  unsigned char o[N];
  for (int so : {0, 32, N}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, N / 2, so);
    hipMemcpy(o, d, N, hipMemcpyDeviceToHost);
    printf("soffset=%d num_records=%d:", so, N / 2);
    for (int i = 0; i < 6; ++i) printf(" [%d]=%d", i, o[i * 16]);
    for (int i = 126; i < 134; ++i) printf(" [%d]=%d", i, o[i * 16]);
    printf("\n");
  }
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
