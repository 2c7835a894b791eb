// ubench_dispatch.hip -- what does it cost to DISPATCH the workgroups of a p2g-shaped launch?  Empty workgroups of 256 threads with p2g's
// resource footprint (90 VGPRs, 30 KB of LDS: five per CU) against footprint-free ones, 1,280 .. 10,800 per launch, plus a variant whose
// workgroups each do one dependent pair of global loads (record -> data) like the head of a chunk workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o gpurun_out/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_DOUBLES, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(90))) void k(const int *rec, const float *data, float *out, int n) {
  __shared__ double tile[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  float acc = 0.0f;
  if (KIND >= 1) {  // record -> 16 component loads per lane (x, m, C, v of p2g)
    int base = rec[blockIdx.x];
    for (int c = 0; c < 16; ++c) acc += data[(size_t)c * n + base + threadIdx.x];
  }
  if (KIND >= 2) {  // + clear the tile, barrier, read it back (the skeleton of a chunk workgroup)
    for (int t = threadIdx.x; t < LDS_DOUBLES; t += 256) tile[t] = 0.0;
    __syncthreads();
    acc += (float)tile[threadIdx.x % (LDS_DOUBLES > 0 ? LDS_DOUBLES : 1)];
  }
  if (LDS_DOUBLES > 0 && KIND < 2 && threadIdx.x == 999) tile[0] = 1.0;
  if (acc == 123.456f) out[blockIdx.x] = acc + (float)tile[0];
}

int main() {
  const int NMAX = 10800, n = NMAX * 256;
  int *rec; float *data, *out;
  CK(hipMalloc(&rec, NMAX * sizeof(int))); CK(hipMalloc(&data, (size_t)16 * n * sizeof(float))); CK(hipMalloc(&out, NMAX * sizeof(float)));
  int *h = new int[NMAX]; for (int i = 0; i < NMAX; ++i) h[i] = i * 256;
  CK(hipMemcpy(rec, h, NMAX * sizeof(int), hipMemcpyHostToDevice)); CK(hipMemset(data, 0, (size_t)16 * n * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int sizes[] = {1280, 2560, 2704, 5400, 10800};
  for (int kind = 0; kind < 3; ++kind)
    for (int lds = 0; lds < 2; ++lds)
      for (int nwg : sizes) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0));
          for (int it = 0; it < 50; ++it) {
            if (kind == 0) { if (lds) hipLaunchKernelGGL((k<3752, 0>), nwg, 256, 0, 0, rec, data, out, n); else hipLaunchKernelGGL((k<0, 0>), nwg, 256, 0, 0, rec, data, out, n); }
            if (kind == 1) { if (lds) hipLaunchKernelGGL((k<3752, 1>), nwg, 256, 0, 0, rec, data, out, n); else hipLaunchKernelGGL((k<0, 1>), nwg, 256, 0, 0, rec, data, out, n); }
            if (kind == 2) { if (lds) hipLaunchKernelGGL((k<3752, 2>), nwg, 256, 0, 0, rec, data, out, n); else hipLaunchKernelGGL((k<1536, 2>), nwg, 256, 0, 0, rec, data, out, n); }
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        printf("kind %d (%s) lds %s  %5d workgroups: %.2f us per launch (back to back, 50 launches)\n", kind,
               kind == 0 ? "empty" : kind == 1 ? "record -> 16 loads per lane" : "loads + tile clear + barrier", lds ? "30 KB" : (kind == 2 ? "12 KB" : "none"), nwg, 1e3 * best / 50);
      }
  return 0;
}
