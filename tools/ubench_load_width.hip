// ubench_load_width.hip -- do p2g / g2p-shaped launches pay per memory INSTRUCTION or per byte?  2,704 workgroups of 256 threads with p2g's
// footprint (90 VGPRs, 30 KB of LDS: five per CU); every lane reads (and, second kernel, writes) N floats of its particle
//   soa : N 4-byte accesses, component-major arrays (the layout of csrc/fast_device.hpp: Soa)
//   v4  : N / 4 16-byte accesses, float4-major arrays (an AoSoA-4 layout: same bytes, a quarter of the requests)
// behind one dependent record load, like the head of a chunk workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_load_width.hip -o gpurun_out/ubench_load_width
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N, bool V4, bool STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(90))) void k(const int *rec, float *data, float *out, int n) {
  __shared__ double tile[3752];
  int base = rec[blockIdx.x] + threadIdx.x;
  float acc = 0.0f;
  if (!STORE) {
    if (V4) {
      const float4 *d4 = reinterpret_cast<const float4 *>(data);
#pragma unroll
      for (int c = 0; c < N / 4; ++c) { float4 v = d4[(size_t)c * n + base]; acc += v.x + v.y + v.z + v.w; }
    } else {
#pragma unroll
      for (int c = 0; c < N; ++c) acc += data[(size_t)c * n + base];
    }
  } else {
    float s = (float)base;
    if (V4) {
      float4 *d4 = reinterpret_cast<float4 *>(data);
#pragma unroll
      for (int c = 0; c < N / 4; ++c) d4[(size_t)c * n + base] = make_float4(s, s + 1, s + 2, s + c);
    } else {
#pragma unroll
      for (int c = 0; c < N; ++c) data[(size_t)c * n + base] = s + c;
    }
  }
  if (threadIdx.x == 999) tile[0] = 1.0;
  if (acc == 123.456f) out[blockIdx.x] = acc + (float)tile[0];
}

template <int N, bool V4, bool STORE>
int run(const int *rec, float *data, float *out, int n, int nwg, const char *what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL((k<N, V4, STORE>), nwg, 256, 0, 0, rec, data, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  double us = 1e3 * best / 50, mb = (double)nwg * 256 * N * 4 / 1e6;
  printf("%-6s %2d floats per lane as %-18s %5d workgroups: %6.2f us per launch, %6.1f MB, %5.2f TB/s\n", STORE ? "store" : "load", N,
         what, nwg, us, mb, mb / us / 1e6 * 1e6 / 1e6);
  return 0;
}

int main() {
  const int NMAX = 5408, n = NMAX * 256;
  int *rec; float *data, *out;
  CK(hipMalloc(&rec, NMAX * sizeof(int))); CK(hipMalloc(&data, (size_t)32 * n * sizeof(float))); CK(hipMalloc(&out, NMAX * sizeof(float)));
  int *h = new int[NMAX]; for (int i = 0; i < NMAX; ++i) h[i] = i * 256;
  CK(hipMemcpy(rec, h, NMAX * sizeof(int), hipMemcpyHostToDevice)); CK(hipMemset(data, 0, (size_t)32 * n * sizeof(float)));
  for (int nwg : {1280, 2704, 5408}) {
    if (run<16, false, false>(rec, data, out, n, nwg, "16 x dword")) return 1;
    if (run<16, true, false>(rec, data, out, n, nwg, "4 x dwordx4")) return 1;
    if (run<28, false, false>(rec, data, out, n, nwg, "28 x dword")) return 1;
    if (run<28, true, false>(rec, data, out, n, nwg, "7 x dwordx4")) return 1;
    if (run<16, false, true>(rec, data, out, n, nwg, "16 x dword")) return 1;
    if (run<16, true, true>(rec, data, out, n, nwg, "4 x dwordx4")) return 1;
  }
  return 0;
}
