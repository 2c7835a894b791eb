// ubench_gridbar.hip -- what a grid-wide barrier inside a persistent kernel costs against a kernel boundary (SURVEY 8(e) /
// "a cooperative persistent kernel ... with grid-wide barriers").  All workgroups of the persistent kernel
// are co-resident (launched <= slots); the barrier is the usual arrive counter + generation flag in device memory
// (agent scope), one thread per workgroup spinning.  Compared with: the same number of dependent (empty-bodied) kernels
// launched back to back on one stream.  The body between barriers touches memory like a tiny phase would (one load + store
// per thread), so both forms pay the same release / acquire.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gridbar.hip -o /tmp/ubench_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned n_wg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
}
__global__ void k_persistent(float *buf, int phases, unsigned *count, unsigned *gen) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int p = 0; p < phases; ++p) {
    buf[i] = buf[(i * 7 + 13) % ((size_t)gridDim.x * blockDim.x)] + 1.0f;   // reads what another workgroup wrote last phase
    grid_barrier(count, gen, gridDim.x);
  }
}
__global__ void k_phase(float *buf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  buf[i] = buf[(i * 7 + 13) % ((size_t)gridDim.x * blockDim.x)] + 1.0f;
}
int main() {
  unsigned *ctr; float *buf;
  CK(hipMalloc(&ctr, 2 * sizeof(unsigned))); CK(hipMemset(ctr, 0, 2 * sizeof(unsigned)));
  CK(hipMalloc(&buf, (size_t)2048 * 256 * sizeof(float))); CK(hipMemset(buf, 0, (size_t)2048 * 256 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int phases = 600;
  for (int wgs : {125, 256, 512, 1024, 1280}) {
    float ms_p = 0, ms_l = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_persistent, wgs, 256, 0, 0, buf, phases, ctr, ctr + 1);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_p, e0, e1));
      CK(hipEventRecord(e0));
      for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, wgs, 256, 0, 0, buf);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_l, e0, e1));
    }
    printf("%4d workgroups x 256 threads: grid barrier %.2f us per phase, kernel boundary %.2f us per phase\n", wgs, ms_p * 1e3 / phases, ms_l * 1e3 / phases);
  }
  return 0;
}
