// Issue-rate probe for the VALU ops the attention softmax is made of (gfx950): cycles per wave-instruction on one SIMD with
// 1, 2 and 3 resident waves.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { asm volatile(REP8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 1) { asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 2) { asm volatile(REP8("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)); }
    if (OP == 3) { asm volatile(REP8("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 4) { asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 5) { asm volatile(REP8("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 6) { asm volatile(REP8("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)); }
    if (OP == 7) { asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 8) { asm volatile(REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n") : "+v"(p0), "+v"(p1), "+v"(a0), "+v"(a1) : : "vcc"); }
    if (OP == 10) { asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc"); }
    if (OP == 11) { asm volatile(REP8("v_bfi_b32 %0, %1, %0, %2\n v_bfi_b32 %1, %2, %1, %3\n v_bfi_b32 %2, %3, %2, %0\n v_bfi_b32 %3, %0, %3, %1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 12) { asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21"); }
    if (OP == 13) { asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 14) { asm volatile(REP8("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 9) { asm volatile(REP8("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP>
void run(const char* name, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  const int iters = 2000;
  for (int waves : {1, 4, 8, 12}) {
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    // s_memtime ticks at a constant 100 MHz-class clock on some parts: also report per-instruction in raw ticks
    printf("%-18s waves/CU=%2d (per SIMD %.2f): %.2f ticks per wave-instruction, %.2f ticks per instruction-slot on a SIMD\n", name, waves, waves / 4.0,
           (double)mx / ((double)iters * per_iter), (double)mx / ((double)iters * per_iter) / (waves < 4 ? 1 : waves / 4.0));
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32", 32); run<1>("v_exp_f32", 32); run<2>("v_pk_fma_f32", 32); run<3>("v_max3_f32", 32); run<4>("v_cvt_pk_bf16_f32", 32);
  run<5>("v_add_f32", 32); run<6>("v_pk_add_f32", 32); run<7>("v_cndmask (stale vcc)", 32);   /* reads a VCC nothing ever wrote: 20 cycles, an artefact -- see v_cmp+v_cndmask */ run<8>("v_mad_u64_u32", 16); run<9>("v_mul_lo_u32", 32);
  run<10>("v_cmp+v_cndmask", 32); run<11>("v_bfi_b32", 32); run<12>("v_cndmask_e64 sgpr", 32); run<13>("v_rcp_f32", 32); run<14>("v_med3_f32", 32);
  return 0;
}
