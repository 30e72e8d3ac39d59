// Lane mapping of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index; every lane reads from
// byte address 8*lane (element 4*lane); print what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = 8 * lane;                       // lane-linear
  else addr = (lane & 3) * 64 * 2 + (lane >> 2) * 8;    // lane l: row (l&3) of 64-element rows, col group (l>>2)
  unsigned long long v;
  addr += (unsigned)(size_t)lds;  // LDS byte address of the array
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)(v >> (16 * e));
}
int main() {
  unsigned short* d;
  hipError_t e0 = hipMalloc(&d, 64 * 4 * 2); printf("malloc %s\n", hipGetErrorString(e0));
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipError_t e1 = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    printf("launch %s sync %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    hipError_t e3 = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); printf("copy %s\n", hipGetErrorString(e3));
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
