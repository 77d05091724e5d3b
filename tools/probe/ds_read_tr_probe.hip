#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const short* src, short* dst) {
  __shared__ __attribute__((aligned(16))) short smem[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x;
  // row stride 64 elements (128 B): 16-lane group reads rows 0..3, cols (g*16).. of a [4][64] tile
  const int i = lane & 15, g = lane >> 4;
  const short* p = smem + (i >> 2) * 64 + g * 16 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = v[j];
}
int main() {
  short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)i;   // element value = row*64 + col
  short *s, *d; hipMalloc(&s, 8192); hipMalloc(&d, 512);
  hipMemcpy(s, h, 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d);
  short o[256]; hipMemcpy(o, d, 512, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 5, 15, 16, 17, 33, 63}) printf("lane %2d: %4d %4d %4d %4d   (expect col %d rows 0..3: %d %d %d %d)\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3], l, l, 64 + l, 128 + l, 192 + l);
  return 0;
}
