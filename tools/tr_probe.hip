// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950 (prints, per lane, the 4 values it receives).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(const short* x, short* y, int mode) {
  __shared__ __attribute__((aligned(16))) short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = x[i];
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  int off;
  if (mode == 0) off = (g * 4 + (i >> 2)) * 64 + (i & 3) * 4;        // lane i supplies row i>>2, col quarter i&3 of a 4x16 block
  else if (mode == 1) off = (g * 4) * 64 + i * 4;                       // contiguous 64 elements per group (canonical form)
  else off = (g * 4 + (i & 3)) * 64 + (i >> 2) * 4;                     // alternative: row = i&3
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(s + off));
  for (int j = 0; j < 4; ++j) y[l * 4 + j] = v[j];
}
int main() {
  short hx[4096], hy[256];
  for (int i = 0; i < 4096; ++i) hx[i] = (short)i;   // value = row*64 + col
  short *dx, *dy;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dy, sizeof(hy));
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, mode);
    hipMemcpy(hy, dy, sizeof(hy), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", hy[l * 4 + j] / 64, hy[l * 4 + j] % 64);
      printf("\n");
    }
  }
  return 0;
}
