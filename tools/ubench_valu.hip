// ubench_valu.hip -- issue rates of the instruction kinds p2g's scatter is made of (per SIMD, wave64, gfx950):
// plain v_fmac_f32, v_fmac_f32_dpp (row_shr), v_cvt_f64_f32, v_pk_fma_f32, with 1..8 wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(int iters, float *out, long long *cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 0.5f;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0)
      asm volatile("v_fmac_f32 %0, %0, %8\n v_fmac_f32 %1, %1, %8\n v_fmac_f32 %2, %2, %8\n v_fmac_f32 %3, %3, %8\n"
                   "v_fmac_f32 %4, %4, %8\n v_fmac_f32 %5, %5, %8\n v_fmac_f32 %6, %6, %8\n v_fmac_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    else if (KIND == 1)
      asm volatile("v_fmac_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_fmac_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_fmac_f32_dpp %4, %4, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %5, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                   "v_fmac_f32_dpp %6, %6, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %7, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    else if (KIND == 2)
      asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                   "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4\n"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    else if (KIND == 3) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m};
      asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                   "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (KIND == 4)  // one dependent chain (latency of back-to-back dependent FMAs)
      asm volatile("v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n"
                   "v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n" : "+v"(a0) : "v"(m));
    else if (KIND == 5)  // dependent DPP chain with the required wait states
      asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n" : "+v"(a0) : "v"(m));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *out; long long *cyc;
  const int NB = 256 * 8;
  CK(hipMalloc(&out, (size_t)NB * 1024 * sizeof(float)));
  CK(hipMalloc(&cyc, NB * sizeof(long long)));
  const char *names[] = {"v_fmac_f32 (8 independent)", "v_fmac_f32_dpp (8 independent)", "v_cvt_f64_f32 (8, 4 dst)", "v_pk_fma_f32 (8 = 16 fma)", "v_fmac_f32 dependent chain", "v_fmac_f32_dpp dependent + s_nop 1"};
  const int iters = 4000;
  for (int kind = 0; kind < 6; ++kind)
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
      // one workgroup per CU of 4 * waves_per_simd wavefronts (<= 16 waves = 1024 threads); 256 workgroups
      int threads = 256 * waves_per_simd; if (threads > 1024) threads = 1024;
      int wgs = 256 * (256 * waves_per_simd / threads);
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        switch (kind) {
          case 0: hipLaunchKernelGGL(k<0>, wgs, threads, 0, 0, iters, out, cyc); break;
          case 1: hipLaunchKernelGGL(k<1>, wgs, threads, 0, 0, iters, out, cyc); break;
          case 2: hipLaunchKernelGGL(k<2>, wgs, threads, 0, 0, iters, out, cyc); break;
          case 3: hipLaunchKernelGGL(k<3>, wgs, threads, 0, 0, iters, out, cyc); break;
          case 4: hipLaunchKernelGGL(k<4>, wgs, threads, 0, 0, iters, out, cyc); break;
          default: hipLaunchKernelGGL(k<5>, wgs, threads, 0, 0, iters, out, cyc); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long c0; CK(hipMemcpy(&c0, cyc, sizeof c0, hipMemcpyDeviceToHost));
      double instr_per_wave = 8.0 * iters;
      printf("%-38s %d waves/SIMD: %.3f ms, %.2f clock64 ticks per instruction per wave, %.2f ns per instruction per SIMD\n", names[kind], waves_per_simd, ms,
             (double)c0 / instr_per_wave, ms * 1e6 / (instr_per_wave * waves_per_simd));
    }
  return 0;
}
