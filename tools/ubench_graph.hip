// Launch-bound sequences of small kernels: stream launches vs one hipGraph.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o /tmp/ubench_graph && /tmp/ubench_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_small(float *p, int n, float a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0f;
}
int main() {
  const int N = 24, n = 1 << 16;   // 24 dependent kernels of ~2 us each
  float *buf;
  CK(hipMalloc(&buf, n * sizeof(float)));
  CK(hipMemset(buf, 0, n * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto seq = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, n / 256, 256, 0, s, buf, n, 0.5f); };
  for (int w = 0; w < 5; ++w) seq();
  CK(hipStreamSynchronize(s));
  float best_s = 1e9f, best_g = 1e9f;
  double host_s = 0, host_g = 0;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipEventRecord(e0, s));
    auto t0 = std::chrono::steady_clock::now();
    seq();
    host_s += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best_s = ms < best_s ? ms : best_s;
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  seq();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipEventRecord(e0, s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    host_g += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best_g = ms < best_g ? ms : best_g;
  }
  printf("%d small dependent kernels: stream launches %.1f us on the GPU (host %.1f us to enqueue), one hipGraphLaunch %.1f us (host %.1f us)\n",
         N, best_s * 1e3, host_s / 20, best_g * 1e3, host_g / 20);
  return 0;
}
