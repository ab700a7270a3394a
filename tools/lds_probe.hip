// LDS read-rate probe (gfx950): bytes per clock and CU of ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16, lane-linear addresses and the
// address pattern of the TN weight-gradient kernel's transposed fragment reads (csrc/gemm_tn.hip: 512-byte rows, 32-byte chunk ^ ((row & 3) << 1)).
// Question it answers: is the transposed read served at the full 128 B/clk, i.e. is the TN kernel's LDS budget the NT kernel's?
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/bin/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(size_t)(p))      // p: 32-bit LDS byte address

// OP 0: b128 lane-linear   1: b64 lane-linear   2: b64_tr, TN pattern   3: plain b64, TN pattern   4: b64_tr lane-linear (8 B per lane)
// 5: b128 with the NT kernel's fragment pattern (row = lane & 31, 16-byte chunk (lane >> 5) ^ swizzle, 128-byte rows)
template <int OP>
__global__ __launch_bounds__(1024) void k(int* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
  __syncthreads();
  int off;
  if (OP == 0) off = lane * 16;
  else if (OP == 1 || OP == 4) off = lane * 8;
  else if (OP == 5) off = (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
  else {
    const int fi = lane & 15, cg = (lane >> 4) & 1, kh = lane >> 5;
    const int rowl = kh * 8 + (fi >> 2), sx = (fi >> 2) << 1;
    off = rowl * 512 + (((cg) ^ sx) << 5) + (fi & 3) * 8;
  }
  // 4 x 16 KB regions: waves of one SIMD group read their own region
  unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (wave & 3) * 16384 + off;
  i32x4 acc = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(base));          // the addresses are loop-invariant: keep the compiler from hoisting the plain reads
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned p = base + (OP == 0 ? j * 1024 : (OP == 1 || OP == 4) ? j * 512 : OP == 5 ? j * 4096 % 16384 + (j >> 2) * 64 : (j & 3) * 64 + (j >> 2) * 8192);
      if (OP == 0 || OP == 5) { i32x4 v = *LDSP(i32x4, p); acc ^= v; }
      else if (OP == 1 || OP == 3) { i32x2 v = *LDSP(i32x2, p); acc[0] ^= v[0]; acc[1] ^= v[1]; }
      else { s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s16x4, p)); i32x2 w = __builtin_bit_cast(i32x2, v); acc[0] ^= w[0]; acc[1] ^= w[1]; }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int OP>
void run(const char* name, int bytes_per_lane) {
  int* out; long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 4000;
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 65536, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double bytes_cu = (double)iters * 8 * 64 * bytes_per_lane * waves;
    printf("%-34s waves/CU=%2d: %6.1f B per tick and CU (readcyclecounter), %6.1f GB/s per CU by wall clock (%.3f ms), %5.2f ticks per wave-instruction\n", name, waves,
           bytes_cu / (double)mx, bytes_cu / (ms * 1e-3) / 1e9, ms, (double)mx / (iters * 8.0));
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("ds_read_b128 lane-linear", 16);
  run<5>("ds_read_b128 NT fragment pattern", 16);
  run<1>("ds_read_b64 lane-linear", 8);
  run<3>("ds_read_b64 TN pattern", 8);
  run<2>("ds_read_b64_tr_b16 TN pattern", 8);
  run<4>("ds_read_b64_tr_b16 lane-linear", 8);
  return 0;
}
