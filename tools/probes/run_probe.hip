// HBM read rate by contiguous run length: every wave instruction reads 64 x 16 B as (1024 / R) runs of R bytes, runs `stride` bytes apart,
// the way the window-attention kernels gather one head's rows (R = 64 / 128 B) out of token rows of 3C x 2 bytes.  Which runs are in
// flight at the same time is what DRAM sees: mode 0 = a wave walks ONE column of runs down the rows (neighbouring columns belong to other
// workgroups: the per-head gathers), mode 1 = the waves of a workgroup take neighbouring columns of the same rows.
// build: hipcc --offload-arch=gfx950 -O3 -o run_probe run_probe.hip ; run: ./run_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int R>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, unsigned* sink, long rows, int row_bytes, int ncol, int mode, int store, char* dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = R / 16;                       // lanes per run
  constexpr int RPI = 64 / LPR;                     // runs (rows) per instruction
  // work item = (column of runs, block of rows); a wave instruction covers RPI consecutive rows of one column
  const long nblk = rows / RPI;
  const long total = nblk * ncol;
  u32x4 acc = {0, 0, 0, 0};
  const long nw = (long)gridDim.x * 8;
  const long wid = (long)blockIdx.x * 8 + wave;
  for (long it = wid; it < total; it += nw) {
    long col, blk;
    if (mode == 0) { col = it % ncol; blk = it / ncol; col = (col * 37) % ncol; /* neighbouring columns far apart in time */ blk = (blk + col * 977) % nblk; }
    else { col = it % ncol; blk = it / ncol; }      // consecutive waves: neighbouring columns of the same rows
    const long row = blk * RPI + lane / LPR;
    const char* a = src + row * row_bytes + col * R + (lane % LPR) * 16;
    u32x4 v = *reinterpret_cast<const u32x4*>(a);
    acc += v;
    if (store) *reinterpret_cast<u32x4*>(dst + row * (long)(row_bytes / 3) + (col * R + (lane % LPR) * 16) % (row_bytes / 3)) = v;
  }
  if (acc[0] == 0x12345678u && acc[1] == 3) sink[0] = acc[2] + acc[3];
}

int main() {
  const int row_bytes = 3072;                       // stage-2 token row (3 x 512 channels)
  const long rows = 294912;                         // 512 images x 576 tokens
  char* src; unsigned* sink; char* dst;
  hipMalloc(&src, rows * row_bytes); hipMalloc(&sink, 64); hipMalloc(&dst, rows * row_bytes / 3);
  hipMemset(src, 1, rows * row_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int store = 0; store < 2; ++store)
  for (int mode = 0; mode < 2; ++mode) {
    for (int R : {64, 128, 256, 512, 1024}) {
      const int ncol = row_bytes / R;
      auto launch = [&]() {
        switch (R) {
          case 64: hipLaunchKernelGGL(probe<64>, dim3(1024), dim3(512), 0, 0, src, sink, rows, row_bytes, ncol, mode, store, dst); break;
          case 128: hipLaunchKernelGGL(probe<128>, dim3(1024), dim3(512), 0, 0, src, sink, rows, row_bytes, ncol, mode, store, dst); break;
          case 256: hipLaunchKernelGGL(probe<256>, dim3(1024), dim3(512), 0, 0, src, sink, rows, row_bytes, ncol, mode, store, dst); break;
          case 512: hipLaunchKernelGGL(probe<512>, dim3(1024), dim3(512), 0, 0, src, sink, rows, row_bytes, ncol, mode, store, dst); break;
          default: hipLaunchKernelGGL(probe<1024>, dim3(1024), dim3(512), 0, 0, src, sink, rows, row_bytes, ncol, mode, store, dst); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)rows * row_bytes * (store ? 4.0 / 3 : 1.0);
      printf("store %d mode %d R %4d: %7.1f us  %.2f TB/s\n", store, mode, R, ms * 200, bytes * 5 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
