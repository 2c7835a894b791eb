// ubench_lds_u64.hip -- LDS atomic add rates by operand type and active lanes: is ds_add_u64 as cheap as ds_add_f64 (then two
// 32-bit fixed-point channels per 64-bit add would halve p2g's LDS atomics)?
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds_u64.hip -o gpurun_out/ubench_lds_u64
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
__global__ void k(int stride, int iters, T *out) {
  __shared__ T tile[3072];
  for (int t = threadIdx.x; t < 3072; t += blockDim.x) tile[t] = 0;
  __syncthreads();
  int lane = threadIdx.x & 63;
  bool active = (lane % stride) == 0;
  for (int it = 0; it < iters; ++it) {
    int a = (int)(((unsigned)(lane * 2654435761u + it * 40503u) >> 7) % 3072u);
    if (active) atomicAdd(&tile[a], (T)1);
  }
  __syncthreads();
  if (threadIdx.x < 8) out[blockIdx.x * 8 + threadIdx.x] = tile[threadIdx.x];
}

template <typename T>
int run(const char *name) {
  T *out; CK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 1024, tpb = 256, iters = 4000;
  for (int stride : {1, 2, 4, 8, 16, 64}) {
    float ms;
    hipLaunchKernelGGL(k<T>, blocks, tpb, 0, 0, stride, 10, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<T>, blocks, tpb, 0, 0, stride, iters, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)blocks * tpb / 64 * iters;
    printf("%-8s active lanes %2d/64: %7.3f ms  %6.2f clk per wave-instr per CU (@2.4GHz)\n", name, 64 / stride, ms,
           ms * 1e-3 * 2.4e9 * 256 / winstr);
  }
  CK(hipFree(out));
  return 0;
}

int main() {
  if (run<double>("f64")) return 1;
  if (run<unsigned long long>("u64")) return 1;
  if (run<unsigned>("u32")) return 1;
  if (run<float>("f32")) return 1;
  return 0;
}
