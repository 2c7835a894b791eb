// ubench_fetch.hip -- calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on known byte counts in THIS code's access patterns.
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half the bytes of a 16 B/lane coalesced streaming read on gfx950 and
// says other widths are uncalibrated.  The solver's particle arrays are component-major fp32: 4 B/lane coalesced loads (one
// 256 B line per wavefront and component), its grid reads are 256 B lines of one channel of one block, its corner forces are
// 12 B/lane gathers.  Each kernel below moves a KNOWN number of bytes (buffers far larger than the 256 MiB Infinity Cache, every
// byte touched exactly once); tools/gpu/r04_fetch_calib.sh runs them under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
// and tools/summarize_fetch_calib.py turns the counter / known-bytes ratios into profiles/r04_traffic_calibration.json, which
// tools/summarize_prof.py applies instead of a blanket x2.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o gpurun_out/ubench_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 4 B / lane, coalesced, component-major: the particle SoA pattern (COMPS components of n items each)
template <int COMPS>
__global__ void k_read_soa4(const float *p, size_t n, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (i < n) {
#pragma unroll
    for (int c = 0; c < COMPS; ++c) s += p[(size_t)c * n + i];
  }
  if (s == 123456.789f) out[0] = s;  // (never true: keeps the loads)
}
// 16 B / lane coalesced (the guide's calibrated case, as the cross-check)
__global__ void k_read_16(const float4 *p, size_t n4, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (i < n4) { float4 v = p[i]; s = v.x + v.y + v.z + v.w; }
  if (s == 123456.789f) out[0] = s;
}
// 12 B / lane at a random permutation of items (the corner-force gather); every item read exactly once
__global__ void k_read_gather12(const float *p, const int *perm, size_t n, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (i < n) { const float *q = p + 3 * (size_t)perm[i]; s = q[0] + q[1] + q[2]; }
  if (s == 123456.789f) out[0] = s;
}
// 4 B / lane coalesced stores
template <int COMPS>
__global__ void k_write_soa4(float *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
#pragma unroll
    for (int c = 0; c < COMPS; ++c) p[(size_t)c * n + i] = (float)c;
  }
}
// fp32 atomic adds, 64 consecutive floats per wavefront (the tile flush of p2g)
__global__ void k_atomic4(float *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(p + i, 1.0f);
}
__global__ void k_perm(int *perm, size_t n, unsigned mul) {  // a bijection of [0, n) for n a power of two: i -> i * odd mod n, then bit-mixed
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = (int)(((unsigned long long)i * mul) & (n - 1));
}

int main(int argc, char **argv) {
  const size_t n = (size_t)1 << 26;  // items per component: 256 MiB per component
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  float *buf, *out;
  int *perm;
  CK(hipMalloc(&buf, n * 16 * sizeof(float)));  // 4 GiB
  CK(hipMalloc(&out, 256));
  CK(hipMalloc(&perm, n * sizeof(int)));
  CK(hipMemset(buf, 0, n * 16 * sizeof(float)));
  const unsigned tpb = 256, grid = (unsigned)(n / tpb);
  hipLaunchKernelGGL(k_perm, grid, tpb, 0, 0, perm, n, 2654435761u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char *name, double bytes, float ms) {
    printf("%-22s known %.1f MB  %.3f ms  %.0f GB/s\n", name, bytes / 1e6, ms, bytes / (ms * 1e-3) / 1e9);
  };
  for (int r = 0; r < reps; ++r) {
    float ms;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read_soa4<16>, grid, tpb, 0, 0, buf, n, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_read_soa4<16>", 16.0 * 4 * n, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read_soa4<1>, grid, tpb, 0, 0, buf, n, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_read_soa4<1>", 4.0 * n, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read_16, (unsigned)(n * 4 / tpb), tpb, 0, 0, (const float4 *)buf, n * 4, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_read_16", 16.0 * 4 * n, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read_gather12, grid, tpb, 0, 0, buf, perm, n, out); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_read_gather12", 12.0 * n + 4.0 * n, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_write_soa4<16>, grid, tpb, 0, 0, buf, n); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_write_soa4<16>", 16.0 * 4 * n, ms);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_atomic4, grid, tpb, 0, 0, buf, n); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("k_atomic4", 4.0 * n, ms);
  }
  return 0;
}
