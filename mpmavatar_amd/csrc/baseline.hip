// baseline.hip -- MPMHIP_MODE_BASELINE: reference-structured kernels.
//
// One kernel per reference launch, in the reference's order (mpm_solver.py:229-536), operating
// directly on the caller's AoS arrays with a dense G^3 grid and per-particle global atomics.
// This mode exists (a) as the on-GPU A/B partner of the fast path (same inputs, different
// algorithmic structure) and (b) to price what the reference's structure costs on MI355X:
// eight dense grid sweeps and 108 scattered fp32 atomics per particle.  It is NOT the product path.
#include "bc.hpp"
#include "ctx.hpp"
#include "mpm_math.hpp"

namespace mpm {

namespace {

constexpr int TPB = 256;
inline unsigned nblk(size_t n) { return n ? (unsigned)((n + TPB - 1) / TPB) : 1u; }  // never an empty grid: kernels bound-check

struct GridDesc {
  int G;
  float dx, inv_dx, grid_lim;
};

__device__ __forceinline__ size_t gidx(int G, int ix, int iy, int iz) {
  return ((size_t)ix * G + iy) * G + iz;
}

// compute_stress_from_F_trial, mpm_utils.py:1017-1105
__global__ void k_stress(mpmhip_state_ptrs st, mpmhip_model_ptrs md, mpmhip_model_scalars sc, int n_nv, int n_e,
                         float dt) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_nv) return;
  if (st.particle_selection[p] != 0) return;
  M3 stress = m3_zero();
  if (p < n_e) {
    M3 d = load_m3(st.particle_d + 9 * (size_t)p);
    QR3 q = qr_cloth(d);
    float gamma = md.gamma[p], kappa = md.kappa[p];
    float r02, r12, r22;
    V3 d3 = anisotropy_return_mapping(q, gamma, kappa, sc.friction_coeff, r02, r12, r22);
    float *dp = st.particle_d + 9 * (size_t)p;
    dp[2] = d3.x; dp[5] = d3.y; dp[8] = d3.z;
    V3 f1, f2, f3;
    kirchhoff_anisotropy(q, r02, r12, r22, d3, load_v3(st.particle_R_inv + 3 * (size_t)p), st.particle_vol[p],
                         md.mu[p], md.lam[p], gamma, kappa, stress, f1, f2, f3);
    const float *fc = st.faces + 3 * (size_t)p;
    int v1 = (int)fc[0], v2 = (int)fc[1], v3i = (int)fc[2];
    float *vf = st.vertex_force;
    atomicAdd(vf + 3 * v1 + 0, f1.x); atomicAdd(vf + 3 * v1 + 1, f1.y); atomicAdd(vf + 3 * v1 + 2, f1.z);
    atomicAdd(vf + 3 * v2 + 0, f2.x); atomicAdd(vf + 3 * v2 + 1, f2.y); atomicAdd(vf + 3 * v2 + 2, f2.z);
    atomicAdd(vf + 3 * v3i + 0, f3.x); atomicAdd(vf + 3 * v3i + 1, f3.y); atomicAdd(vf + 3 * v3i + 2, f3.z);
  } else {
    M3 Ft = load_m3(st.particle_F_trial + 9 * (size_t)p), F;
    float mu = md.mu[p], lam = md.lam[p], ys = md.yield_stress[p];
    int m = sc.material;
    TradParams tp{m, sc.alpha, sc.hardening, sc.xi, sc.plastic_viscosity, sc.softening};
    traditional_update(Ft, tp, mu, lam, ys, dt, F, stress);
    if (m == 1 || m == 5) md.yield_stress[p] = ys;
    if (m == 5) { md.mu[p] = mu; md.lam[p] = lam; }
    store_m3(st.particle_F + 9 * (size_t)p, F);
  }
  store_m3(st.particle_stress + 9 * (size_t)p, stress);
}

// p2g_apic_with_stress, mpm_utils.py:484-557
__global__ void k_p2g(mpmhip_state_ptrs st, float rpic, GridDesc gd, int n_p, int n_nv, int n_e, float dt,
                      float *grid_m, float *grid_v_in) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_p) return;
  if (st.particle_selection[p] != 0) return;
  bool is_vert = p >= n_nv;
  V3 vforce = v3(0, 0, 0);
  M3 S = m3_zero();
  if (is_vert) vforce = load_v3(st.vertex_force + 3 * (size_t)(p - n_nv));
  else {
    S = load_m3(st.particle_stress + 9 * (size_t)p);
    if (p >= n_e) S = st.particle_vol[p] * S;
  }
  Stencil s = make_stencil(load_v3(st.particle_x + 3 * (size_t)p), gd.inv_dx);
  M3 C = load_m3(st.particle_C + 9 * (size_t)p);
  C = (1.0f - rpic) * C + (rpic / 2.0f) * (C - transpose(C));
  if (rpic < -0.001f) C = m3_zero();
  float mass = st.particle_mass[p];
  V3 v = load_v3(st.particle_v + 3 * (size_t)p);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x), wy = sel3(j, s.w0.y, s.w1.y, s.w2.y),
              wz = sel3(k, s.w0.z, s.w1.z, s.w2.z);
        float dwx = sel3(i, s.dw0.x, s.dw1.x, s.dw2.x), dwy = sel3(j, s.dw0.y, s.dw1.y, s.dw2.y),
              dwz = sel3(k, s.dw0.z, s.dw1.z, s.dw2.z);
        float weight = wx * wy * wz;
        V3 dweight = gd.inv_dx * v3(dwx * wy * wz, wx * dwy * wz, wx * wy * dwz);
        V3 dpos = gd.dx * v3((float)i - s.fx.x, (float)j - s.fx.y, (float)k - s.fx.z);
        V3 force = is_vert ? weight * vforce : -1.0f * (S * dweight);
        V3 add = (weight * mass) * (v + C * dpos) + dt * force;
        size_t g = gidx(gd.G, s.bx + i, s.by + j, s.bz + k);
        atomicAdd(grid_v_in + 3 * g + 0, add.x);
        atomicAdd(grid_v_in + 3 * g + 1, add.y);
        atomicAdd(grid_v_in + 3 * g + 2, add.z);
        atomicAdd(grid_m + g, weight * mass);
      }
}

// grid_normalization_and_gravity (+ add_damping_via_grid), mpm_utils.py:561-572, 1162-1174
__global__ void k_grid_norm(const float *grid_m, const float *grid_v_in, float *grid_v_out, size_t n, float dt,
                            float gx, float gy, float gz, float damping_scale) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float m = grid_m[g];
  V3 vo = load_v3(grid_v_out + 3 * g);
  if (m > 1e-15f) {
    float inv = 1.0f / m;
    vo = v3(grid_v_in[3 * g] * inv + dt * gx, grid_v_in[3 * g + 1] * inv + dt * gy, grid_v_in[3 * g + 2] * inv + dt * gz);
  }
  if (damping_scale < 1.0f) vo = vo - (1.0f - damping_scale) * vo;
  store_v3(grid_v_out + 3 * g, vo);
}

__device__ __forceinline__ bool splat_ok(int G, const Stencil &s) {
  return s.bx >= 0 && s.bx < G - 3 && s.by >= 0 && s.by < G - 3 && s.bz >= 0 && s.bz < G - 3;
}

// compute_mesh, mpm_solver.py:829-880
__global__ void k_face_splat(const float *pts, const float *vel, float adv, const int32_t *idx, int n_f, GridDesc gd,
                             float *weight, float *v_in, float *normal) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_f) return;
  int i0 = idx[3 * f], i1 = idx[3 * f + 1], i2 = idx[3 * f + 2];
  V3 p0 = mesh_point(pts, vel, adv, i0), p1 = mesh_point(pts, vel, adv, i1), p2 = mesh_point(pts, vel, adv, i2);
  V3 fp = v3((p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f);
  V3 u0 = load_v3(vel + 3 * i0), u1 = load_v3(vel + 3 * i1), u2 = load_v3(vel + 3 * i2);
  V3 fv = v3((u0.x + u1.x + u2.x) / 3.0f, (u0.y + u1.y + u2.y) / 3.0f, (u0.z + u1.z + u2.z) / 3.0f);
  V3 fn = normalize(cross(p1 - p0, p2 - p0));
  Stencil s = make_stencil(fp, gd.inv_dx);
  if (!splat_ok(gd.G, s)) return;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float w = sel3(i, s.w0.x, s.w1.x, s.w2.x) * sel3(j, s.w0.y, s.w1.y, s.w2.y) * sel3(k, s.w0.z, s.w1.z, s.w2.z);
        size_t g = gidx(gd.G, s.bx + i, s.by + j, s.bz + k);
        atomicAdd(v_in + 3 * g, w * fv.x); atomicAdd(v_in + 3 * g + 1, w * fv.y); atomicAdd(v_in + 3 * g + 2, w * fv.z);
        atomicAdd(normal + 3 * g, w * fn.x); atomicAdd(normal + 3 * g + 1, w * fn.y); atomicAdd(normal + 3 * g + 2, w * fn.z);
        atomicAdd(weight + g, w);
      }
}

// normalize_grid + collide, mpm_solver.py:882-917
__global__ void k_collide(float *grid_v_out, const float *weight, const float *v_in, const float *normal, size_t n,
                          float friction) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float w = weight[g];
  if (w > 1e-15f) {
    float inv = 1.0f / w;
    V3 vm = v3(v_in[3 * g] * inv, v_in[3 * g + 1] * inv, v_in[3 * g + 2] * inv);
    V3 v = collide_node(load_v3(grid_v_out + 3 * g), vm, load_v3(normal + 3 * g), friction);
    store_v3(grid_v_out + 3 * g, v);
  }
}

// add_velocity_{traditional,verts,faces}, mpm_solver.py:677-788: particle q = off + tid gets vel[tid]
__global__ void k_mover_splat(const float *x, const float *vel, int n, int off, GridDesc gd, float *weight,
                              float *velocity) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Stencil s = make_stencil(load_v3(x + 3 * (size_t)(t + off)), gd.inv_dx);
  if (!splat_ok(gd.G, s)) return;
  V3 pv = load_v3(vel + 3 * (size_t)t);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float w = sel3(i, s.w0.x, s.w1.x, s.w2.x) * sel3(j, s.w0.y, s.w1.y, s.w2.y) * sel3(k, s.w0.z, s.w1.z, s.w2.z);
        size_t g = gidx(gd.G, s.bx + i, s.by + j, s.bz + k);
        atomicAdd(velocity + 3 * g, w * pv.x); atomicAdd(velocity + 3 * g + 1, w * pv.y);
        atomicAdd(velocity + 3 * g + 2, w * pv.z);
        atomicAdd(weight + g, w);
      }
}

// normalize_grid of the mover: overwrite, mpm_solver.py:790-799
__global__ void k_mover_apply(float *grid_v_out, const float *weight, const float *velocity, size_t n) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float w = weight[g];
  if (w > 1e-15f) {
    float inv = 1.0f / w;
    store_v3(grid_v_out + 3 * g, v3(velocity[3 * g] * inv, velocity[3 * g + 1] * inv, velocity[3 * g + 2] * inv));
  }
}

__global__ void k_bc(float *grid_v_out, BC bc, GridDesc gd, float time, float dt) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = (size_t)gd.G * gd.G * gd.G;
  if (g >= n) return;
  int gz = (int)(g % gd.G), gy = (int)((g / gd.G) % gd.G), gx = (int)(g / ((size_t)gd.G * gd.G));
  V3 v = load_v3(grid_v_out + 3 * g);
  if (apply_bc(bc, v, gx, gy, gz, gd.G, gd.dx, time, dt, g)) store_v3(grid_v_out + 3 * g, v);
}

// shared gather of g2p_v / g2p_e, mpm_utils.py:726-763
__device__ __forceinline__ void g2p_gather(const float *grid_v_out, V3 x, GridDesc gd, V3 &nv, M3 &nC, M3 &nF) {
  Stencil s = make_stencil(x, gd.inv_dx);
  nv = v3(0, 0, 0);
  nC = m3_zero();
  nF = m3_zero();
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x), wy = sel3(j, s.w0.y, s.w1.y, s.w2.y),
              wz = sel3(k, s.w0.z, s.w1.z, s.w2.z);
        float dwx = sel3(i, s.dw0.x, s.dw1.x, s.dw2.x), dwy = sel3(j, s.dw0.y, s.dw1.y, s.dw2.y),
              dwz = sel3(k, s.dw0.z, s.dw1.z, s.dw2.z);
        float weight = wx * wy * wz;
        V3 dweight = gd.inv_dx * v3(dwx * wy * wz, wx * dwy * wz, wx * wy * dwz);
        V3 dpos = v3((float)i - s.fx.x, (float)j - s.fx.y, (float)k - s.fx.z);
        V3 gv = load_v3(grid_v_out + 3 * gidx(gd.G, s.bx + i, s.by + j, s.bz + k));
        nv = nv + weight * gv;
        nC = nC + (weight * gd.inv_dx * 4.0f) * outer(gv, dpos);
        nF = nF + outer(gv, dweight);
      }
}

// g2p_v, mpm_utils.py:716-786
__global__ void k_g2p_v(mpmhip_state_ptrs st, const float *grid_v_out, GridDesc gd, int n_p, int n_nv, int n_e,
                        float dt) {
  int q = blockIdx.x * blockDim.x + threadIdx.x + n_e;
  if (q >= n_p) return;
  if (st.particle_selection[q] != 0) return;
  V3 x = load_v3(st.particle_x + 3 * (size_t)q), nv;
  M3 nC, nF;
  g2p_gather(grid_v_out, x, gd, nv, nC, nF);
  store_v3(st.particle_v + 3 * (size_t)q, nv);
  float dxl = 1.0f / gd.inv_dx, a_min = dxl * 2.0f, a_max = gd.grid_lim - dxl * 2.0f;
  V3 nx = x + dt * nv;
  nx = v3(fminf(fmaxf(nx.x, a_min), a_max), fminf(fmaxf(nx.y, a_min), a_max), fminf(fmaxf(nx.z, a_min), a_max));
  store_v3(st.particle_x + 3 * (size_t)q, nx);
  store_m3(st.particle_C + 9 * (size_t)q, nC);
  if (q < n_nv) {
    M3 Fn = deform_update(nF, dt, load_m3(st.particle_F + 9 * (size_t)q));
    store_m3(st.particle_F_trial + 9 * (size_t)q, Fn);
  }
}

// g2p_e, mpm_utils.py:788-857 (must run after k_g2p_v: reads the already-updated vertices)
__global__ void k_g2p_e(mpmhip_state_ptrs st, const float *grid_v_out, GridDesc gd, int n_nv, int n_e, float dt) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_e) return;
  if (st.particle_selection[p] != 0) return;
  V3 x = load_v3(st.particle_x + 3 * (size_t)p), nv;
  M3 nC, nF;
  g2p_gather(grid_v_out, x, gd, nv, nC, nF);
  const float *fc = st.faces + 3 * (size_t)p;
  size_t v1 = (size_t)((int)fc[0] + n_nv), v2 = (size_t)((int)fc[1] + n_nv), v3i = (size_t)((int)fc[2] + n_nv);
  V3 x1 = load_v3(st.particle_x + 3 * v1), x2 = load_v3(st.particle_x + 3 * v2), x3 = load_v3(st.particle_x + 3 * v3i);
  V3 u1 = load_v3(st.particle_v + 3 * v1), u2 = load_v3(st.particle_v + 3 * v2), u3 = load_v3(st.particle_v + 3 * v3i);
  store_v3(st.particle_v + 3 * (size_t)p, v3((u1.x + u2.x + u3.x) / 3.0f, (u1.y + u2.y + u3.y) / 3.0f, (u1.z + u2.z + u3.z) / 3.0f));
  store_v3(st.particle_x + 3 * (size_t)p, v3((x1.x + x2.x + x3.x) / 3.0f, (x1.y + x2.y + x3.y) / 3.0f, (x1.z + x2.z + x3.z) / 3.0f));
  store_m3(st.particle_C + 9 * (size_t)p, nC);
  float *dp = st.particle_d + 9 * (size_t)p;
  V3 d3 = v3(dp[2], dp[5], dp[8]);
  V3 d3n = (m3_identity() + dt * nF) * d3;
  store_m3(dp, m3_cols(x2 - x1, x3 - x1, d3n));
}

}  // namespace

int baseline_init(mpmhip_ctx *c) {
  size_t n = G3(c);
  MPM_HIP_CHECK(c, hipMalloc(&c->grid_m, n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMalloc(&c->grid_v_in, 3 * n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMalloc(&c->grid_v_out, 3 * n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_m, 0, n * sizeof(float), c->stream));
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_v_in, 0, 3 * n * sizeof(float), c->stream));
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_v_out, 0, 3 * n * sizeof(float), c->stream));
  return MPMHIP_OK;
}

int baseline_add_collider_storage(mpmhip_ctx *c, MeshCollider &mc) {
  size_t n = G3(c);
  MPM_HIP_CHECK(c, hipMalloc(&mc.weight, n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMalloc(&mc.v_in, 3 * n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMalloc(&mc.normal, 3 * n * sizeof(float)));
  return MPMHIP_OK;
}
int baseline_add_mover_storage(mpmhip_ctx *c, Mover &mv) {
  size_t n = G3(c);
  MPM_HIP_CHECK(c, hipMalloc(&mv.weight, n * sizeof(float)));
  MPM_HIP_CHECK(c, hipMalloc(&mv.velocity, 3 * n * sizeof(float)));
  return MPMHIP_OK;
}

int baseline_step(mpmhip_ctx *c, const StepArgs &a) {
  hipStream_t s = c->stream;
  const int n_p = c->cfg.n_particles, n_e = c->cfg.n_elements, n_v = c->cfg.n_vertices, n_nv = c->n_nv;
  const size_t n = G3(c);
  GridDesc gd{c->cfg.n_grid, c->dx, c->inv_dx, c->cfg.grid_lim};
  const float dt = a.dt;
  // zero_grid + set_vec3_to_zero(vertex_force)            mpm_solver.py:244-256
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_m, 0, n * sizeof(float), s));
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_v_in, 0, 3 * n * sizeof(float), s));
  MPM_HIP_CHECK(c, hipMemsetAsync(c->grid_v_out, 0, 3 * n * sizeof(float), s));
  if (n_v) MPM_HIP_CHECK(c, hipMemsetAsync(c->st.vertex_force, 0, (size_t)n_v * 3 * sizeof(float), s));
  // pre-p2g particle operations                           mpm_solver.py:260-279
  if (!c->pre.empty()) {
    int rc = launch_pre_ops(c, dt, c->st.particle_v, c->st.particle_x, c->st.particle_mass, n_p);
    if (rc) return rc;
  }
  {
    ScopedPhase ph(c, "compute_stress_from_F_trial");
    if (n_nv) hipLaunchKernelGGL(k_stress, nblk(n_nv), TPB, 0, s, c->st, c->md, c->sc, n_nv, n_e, dt);
  }
  {
    ScopedPhase ph(c, "p2g");
    hipLaunchKernelGGL(k_p2g, nblk(n_p), TPB, 0, s, c->st, c->sc.rpic_damping, gd, n_p, n_nv, n_e, dt, c->grid_m,
                       c->grid_v_in);
  }
  {
    ScopedPhase ph(c, "grid_update");
    hipLaunchKernelGGL(k_grid_norm, nblk(n), TPB, 0, s, c->grid_m, c->grid_v_in, c->grid_v_out, n, dt, c->sc.g[0],
                       c->sc.g[1], c->sc.g[2], c->sc.grid_v_damping_scale);
  }
  {
    ScopedPhase ph(c, "apply_Mesh_Collision_on_grid");
    for (auto &mc : c->colliders) {
      MPM_HIP_CHECK(c, hipMemsetAsync(mc.weight, 0, n * sizeof(float), s));
      MPM_HIP_CHECK(c, hipMemsetAsync(mc.v_in, 0, 3 * n * sizeof(float), s));
      MPM_HIP_CHECK(c, hipMemsetAsync(mc.normal, 0, 3 * n * sizeof(float), s));
      if (c->num_mesh_f)
        hipLaunchKernelGGL(k_face_splat, nblk(c->num_mesh_f), TPB, 0, s, c->cur_pts, c->cur_vel, c->cur_f, c->mesh_idx,
                           c->num_mesh_f, gd, mc.weight, mc.v_in, mc.normal);
      hipLaunchKernelGGL(k_collide, nblk(n), TPB, 0, s, c->grid_v_out, mc.weight, mc.v_in, mc.normal, n, mc.friction);
    }
  }
  if (a.joint_v_v && a.joint_f_v) {
    ScopedPhase ph(c, "apply_Particle_Moving_on_grid");
    for (auto &mv : c->movers) {
      MPM_HIP_CHECK(c, hipMemsetAsync(mv.weight, 0, n * sizeof(float), s));
      MPM_HIP_CHECK(c, hipMemsetAsync(mv.velocity, 0, 3 * n * sizeof(float), s));
      if (a.joint_t_v && a.n_joint_t > 0)
        hipLaunchKernelGGL(k_mover_splat, nblk(a.n_joint_t), TPB, 0, s, c->st.particle_x, a.joint_t_v, a.n_joint_t,
                           n_nv - a.n_joint_t, gd, mv.weight, mv.velocity);
      if (c->cfg.num_joint_v > 0)
        hipLaunchKernelGGL(k_mover_splat, nblk(c->cfg.num_joint_v), TPB, 0, s, c->st.particle_x, a.joint_v_v,
                           c->cfg.num_joint_v, n_nv, gd, mv.weight, mv.velocity);
      if (c->cfg.num_joint_f > 0)
        hipLaunchKernelGGL(k_mover_splat, nblk(c->cfg.num_joint_f), TPB, 0, s, c->st.particle_x, a.joint_f_v,
                           c->cfg.num_joint_f, 0, gd, mv.weight, mv.velocity);
      hipLaunchKernelGGL(k_mover_apply, nblk(n), TPB, 0, s, c->grid_v_out, mv.weight, mv.velocity, n);
    }
  }
  {
    ScopedPhase ph(c, "apply_BC_on_grid");
    for (auto &bc : c->bcs) {
      hipLaunchKernelGGL(k_bc, nblk(n), TPB, 0, s, c->grid_v_out, bc, gd, (float)c->time, dt);
      bc_host_modify(bc, (float)c->time, dt);
    }
  }
  {
    ScopedPhase ph(c, "g2p_v");
    if (n_p - n_e > 0)
      hipLaunchKernelGGL(k_g2p_v, nblk(n_p - n_e), TPB, 0, s, c->st, c->grid_v_out, gd, n_p, n_nv, n_e, dt);
  }
  {
    ScopedPhase ph(c, "g2p_e");
    if (n_e) hipLaunchKernelGGL(k_g2p_e, nblk(n_e), TPB, 0, s, c->st, c->grid_v_out, gd, n_nv, n_e, dt);
  }
  MPM_HIP_CHECK(c, hipGetLastError());
  return MPMHIP_OK;
}

}  // namespace mpm
