// ubench_atomics.hip -- microbenchmarks that decided the p2g design:
// LDS fp32 / u32 atomic-add rates under different lane->address patterns, and global fp32 atomic-add rates.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_atomics.hip -o gpurun_out/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pattern: 0 = lane-distinct consecutive addresses, 1 = all lanes same address, 2 = groups of 12 lanes share an
// address (cell-sorted particles), 3 = pseudo-random addresses in a 3072-float tile, 4 = stride 32 (same bank)
__device__ __forceinline__ int addr_of(int pattern, int lane, int it) {
  switch (pattern) {
    case 0: return (lane + it * 64) % 3072;
    case 1: return it % 3072;
    case 2: return ((lane / 12) * 7 + it * 13) % 3072;
    case 3: return (int)(((unsigned)(lane * 2654435761u + it * 40503u) >> 7) % 3072u);
    default: return (lane * 32 + it) % 3072;
  }
}

template <typename T, bool RTN>
__global__ void k_lds(int pattern, int iters, T *out) {
  __shared__ T tile[3072];
  for (int t = threadIdx.x; t < 3072; t += blockDim.x) tile[t] = 0;
  __syncthreads();
  int lane = threadIdx.x & 63;
  T acc = 0;
  for (int it = 0; it < iters; ++it) {
    int a = addr_of(pattern, lane, it);
    if (RTN) acc += atomicAdd(&tile[a], (T)1);
    else atomicAdd(&tile[a], (T)1);
  }
  __syncthreads();
  if (threadIdx.x < 8) out[blockIdx.x * 8 + threadIdx.x] = tile[threadIdx.x] + acc;
}

// plain LDS read-modify-write (no atomic) as the speed-of-light reference
__global__ void k_lds_plain(int pattern, int iters, float *out) {
  __shared__ float tile[3072];
  for (int t = threadIdx.x; t < 3072; t += blockDim.x) tile[t] = 0;
  __syncthreads();
  int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    int a = addr_of(pattern, lane, it);
    tile[a] = tile[a] + 1.0f;
  }
  __syncthreads();
  if (threadIdx.x < 8) out[blockIdx.x * 8 + threadIdx.x] = tile[threadIdx.x];
}

// global: pattern 0 = each wave adds to 64 consecutive floats (coalesced), 1 = scattered (one line per lane),
// 2 = 16-lane runs (64 B segments)
__global__ void k_global(int pattern, int iters, float *buf, size_t n) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int lane = threadIdx.x & 63;
  size_t wave = gid >> 6;
  for (int it = 0; it < iters; ++it) {
    size_t a;
    if (pattern == 0) a = ((wave * 131 + it * 7919) * 64 + lane) % n;
    else if (pattern == 1) a = ((gid * 2654435761ull + (size_t)it * 97) * 64) % n;
    else a = (((wave * 4 + (lane >> 4)) * 2654435761ull + it * 131) * 16 + (lane & 15)) % n;
    atomicAdd(buf + a, 1.0f);
  }
}

int main() {
  float *out, *buf;
  unsigned *outu;
  size_t n = 8u << 20;  // 32 MB of floats: larger than L2, inside MALL
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&outu, 1 << 20)); CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 4, tpb = 256, iters = 2000;
  const char *pn[] = {"distinct", "same-addr", "12-lane groups", "random", "same-bank"};
  auto report = [&](const char *name, int p, float ms) {
    double lane_ops = (double)blocks * tpb * iters;
    printf("%-22s %-16s %8.3f ms  %8.1f G lane-ops/s  %6.2f clk/wave-instr/CU(@2.4GHz)\n", name, pn[p], ms,
           lane_ops / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / (lane_ops / 64));
  };
  for (int p = 0; p < 5; ++p) {
    float ms;
    hipLaunchKernelGGL((k_lds<float, false>), blocks, tpb, 0, 0, p, 10, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<float, false>), blocks, tpb, 0, 0, p, iters, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds add f32", p, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<float, true>), blocks, tpb, 0, 0, p, iters, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds add_rtn f32", p, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<unsigned, false>), blocks, tpb, 0, 0, p, iters, outu); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds add u32", p, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<double, false>), blocks, tpb, 0, 0, p, iters, (double*)outu); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds add f64", p, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<unsigned long long, false>), blocks, tpb, 0, 0, p, iters, (unsigned long long*)outu); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds add u64", p, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_plain, blocks, tpb, 0, 0, p, iters, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("lds plain rmw", p, ms);
  }
  const char *gn[] = {"coalesced 256B", "scattered", "64B runs"};
  for (int p = 0; p < 3; ++p) {
    float ms;
    int git = 200;
    hipLaunchKernelGGL(k_global, blocks, tpb, 0, 0, p, 5, buf, n); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_global, blocks, tpb, 0, 0, p, git, buf, n); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)blocks * tpb * git;
    printf("global atomicAdd f32   %-16s %8.3f ms  %8.1f G lane-ops/s\n", gn[p], ms, lane_ops / ms * 1e-6);
  }
  return 0;
}
