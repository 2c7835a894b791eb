// common.hip -- small kernels shared by both back ends: pre-p2g particle operations
// (mpm_solver.py:1058-1417), selection masks (mpm_utils.py:1198-1248), counters.
#include "ctx.hpp"
#include "mpm_math.hpp"

namespace mpm {

namespace {
constexpr int TPB = 256;
inline unsigned nblk(size_t n) { return n ? (unsigned)((n + TPB - 1) / TPB) : 1u; }  // never an empty grid: kernels bound-check

// v / x / mass are AoS [n*3], [n*3], [n] in the caller's particle order
__global__ void k_pre(PreOp op, float *v, const float *x, const float *mass, int n, float dt) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int mk = op.mask[p];
  V3 pv = load_v3(v + 3 * (size_t)p);
  if (op.type == PRE_IMPULSE) {
    if (mk != 1) return;
    float m = mass[p];
    pv = pv + dt * v3(op.force[0] / m, op.force[1] / m, op.force[2] / m);
  } else if (op.type == PRE_IMPULSE_MASK) {
    if (mk < 1) return;
    pv = pv + dt * v3(op.force[0], op.force[1], op.force[2]);
  } else if (op.type == PRE_VEL_SET) {
    if (mk != 1) return;
    pv = v3(op.velocity[0], op.velocity[1], op.velocity[2]);
  } else {  // PRE_VEL_ROTATE, mpm_solver.py:1225-1255
    if (mk != 1) return;
    V3 nrm = v3(op.normal[0], op.normal[1], op.normal[2]);
    V3 a1 = v3(op.axis1[0], op.axis1[1], op.axis1[2]), a2 = v3(op.axis2[0], op.axis2[1], op.axis2[2]);
    V3 off = load_v3(x + 3 * (size_t)p) - v3(op.point[0], op.point[1], op.point[2]);
    float hd = length(off - dot(off, nrm) * nrm);
    float theta = acosf(fminf(fmaxf(dot(off, a1) / hd, -1.f), 1.f));  // wp.acos clamps its argument
    if (!(dot(off, a2) > 0.0f)) theta = -theta;
    float s1 = -hd * sinf(theta) * op.rotation_scale, s2 = hd * cosf(theta) * op.rotation_scale;
    pv = s1 * a1 + s2 * a2 + op.translation_scale * nrm;
  }
  store_v3(v + 3 * (size_t)p, pv);
}

__global__ void k_select_box(const float *x, int n, V3 point, V3 size, int32_t *mask) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  V3 off = load_v3(x + 3 * (size_t)p) - point;
  mask[p] = (fabsf(off.x) < size.x && fabsf(off.y) < size.y && fabsf(off.z) < size.z) ? 1 : 0;
}

__global__ void k_select_cyl(const float *x, int n, V3 point, V3 nrm, float hh, float radius, int32_t *mask) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  V3 off = load_v3(x + 3 * (size_t)p) - point;
  float vd = fabsf(dot(off, nrm));
  float hd = length(off - dot(off, nrm) * nrm);
  mask[p] = (vd < hh && hd < radius) ? 1 : 0;
}

__global__ void k_count(const float *a, size_t n, float thresh, int *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = (i < n && a[i] > thresh) ? 1 : 0;
  unsigned long long b = __ballot(c);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, __popcll(b));
}
}  // namespace

// reference order: all impulses first (mpm_solver.py:260), then all velocity modifiers (:269)
int launch_pre_ops(mpmhip_ctx *c, float dt, float *v, const float *x, const float *mass, int n) {
  float t = (float)c->time;
  for (int pass = 0; pass < 2; ++pass)
    for (auto &op : c->pre) {
      bool imp = op.type == PRE_IMPULSE || op.type == PRE_IMPULSE_MASK;
      if (imp != (pass == 0)) continue;
      if (!(t >= op.start_time && t < op.end_time)) continue;
      hipLaunchKernelGGL(k_pre, nblk(n), TPB, 0, c->stream, op, v, x, mass, n, dt);
    }
  return MPMHIP_OK;
}

int launch_select_box(mpmhip_ctx *c, const float *x, const float point[3], const float size[3], int32_t *mask) {
  int n = c->cfg.n_particles;
  hipLaunchKernelGGL(k_select_box, nblk(n), TPB, 0, c->stream, x, n, v3(point[0], point[1], point[2]),
                     v3(size[0], size[1], size[2]), mask);
  return MPMHIP_OK;
}

int launch_select_cylinder(mpmhip_ctx *c, const float *x, const float point[3], const float normal[3],
                           float half_height, float radius, int32_t *mask) {
  int n = c->cfg.n_particles;
  hipLaunchKernelGGL(k_select_cyl, nblk(n), TPB, 0, c->stream, x, n, v3(point[0], point[1], point[2]),
                     v3(normal[0], normal[1], normal[2]), half_height, radius, mask);
  return MPMHIP_OK;
}

int count_nonzero(mpmhip_ctx *c, const float *a, size_t n, float thresh, int *out) {
  int *d = nullptr;
  MPM_HIP_CHECK(c, hipMalloc(&d, sizeof(int)));
  MPM_HIP_CHECK(c, hipMemsetAsync(d, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_count, nblk(n), TPB, 0, c->stream, a, n, thresh, d);
  MPM_HIP_CHECK(c, hipMemcpyAsync(out, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  MPM_HIP_CHECK(c, hipFree(d));
  return MPMHIP_OK;
}

}  // namespace mpm
