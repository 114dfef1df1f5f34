#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x % 5 == 0) voff = 0x80000000u;     // out of range -> expect zeros
  if (threadIdx.x == 63) voff = nbytes - 8;        // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + 256), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  float *src, *out; float h[512], hs[300];
  for (int i = 0; i < 300; ++i) hs[i] = i + 1;
  hipMalloc(&src, sizeof(hs)); hipMalloc(&out, sizeof(h));
  hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
  k<<<1, 64>>>(src, 256 * 4, out);
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("first half untouched: %g %g\n", h[0], h[255]);
  for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[256 + 4 * l], h[256 + 4 * l + 1], h[256 + 4 * l + 2], h[256 + 4 * l + 3]);
  printf("lane 63: %g %g %g %g\n", h[256 + 252], h[256 + 253], h[256 + 254], h[256 + 255]);
  return 0;
}
