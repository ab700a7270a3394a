// Per-CU store throughput probe (gfx950): how fast can ONE compute unit push 16-byte-per-lane stores towards HBM, and does the rate
// depend on how many other CUs store at the same time?  Pattern = the GEMM epilogue's: each wave-instruction writes 8 rows x 128 B
// (or 16 rows x 64 B with -DHALF) of a row-major [M, ld] bf16 matrix, one 256-row x 256-column tile after the other per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/bin/store_probe && tools/bin/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool HALF>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ld_bytes, int tiles_per_wg, int tilesN, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const long long t0 = clock64();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int id = blockIdx.x + t * gridDim.x;
    const int m0 = (id / tilesN) * 256 + wm * 128, n0 = (id % tilesN) * 512 + wn * 128;   // byte columns
    char* base = out + (size_t)m0 * ld_bytes + n0;
    u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {       // 16 wave-instructions of 1 KB = the wave's 128 x 64 bf16 sub-tile
      char* p = HALF ? base + (size_t)((i >> 1) * 16 + (lane >> 2)) * ld_bytes + (i & 1) * 64 + (lane & 3) * 16
                     : base + (size_t)(i * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16;
      if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
  const int M = 294912, N = 2048, ld = N * 2, tilesN = N / 256, tiles = (M / 256) * tilesN;
  char* out; long long* cyc;
  hipMalloc(&out, (size_t)M * ld); hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("output %d x %d bf16 = %.2f GB, %d tiles of 128 KB\n", M, N, (double)M * ld / 1e9, tiles);
  for (int half = 0; half < 2; ++half)
    for (int nt = 1; nt >= 0; --nt)
      for (int wgs : {256, 128, 64, 32, 8}) {
        const int per = 36;                                  // tiles per workgroup (the GEMM's count at 256 workgroups)
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (half) { if (nt) hipLaunchKernelGGL((store_kernel<true, true>), dim3(wgs), dim3(512), 0, 0, out, ld, per, tilesN, cyc);
                      else hipLaunchKernelGGL((store_kernel<false, true>), dim3(wgs), dim3(512), 0, 0, out, ld, per, tilesN, cyc); }
          else { if (nt) hipLaunchKernelGGL((store_kernel<true, false>), dim3(wgs), dim3(512), 0, 0, out, ld, per, tilesN, cyc);
                 else hipLaunchKernelGGL((store_kernel<false, false>), dim3(wgs), dim3(512), 0, 0, out, ld, per, tilesN, cyc); }
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = (double)wgs * per * 131072.0;
        printf("%s %s stores, %3d workgroups (1 per CU): %7.1f us  %6.2f TB/s total  %6.1f GB/s per CU\n", half ? "16x64B " : "8x128B ",
               nt ? "nt   " : "plain", wgs, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
      }
  return 0;
}
