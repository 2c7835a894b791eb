// The drop-in boundary without Python: a jelly cube dropped inside a bounding box, driven through include/mpmhip.h only
// (the calls a ctypes / cffi / C++ binding of warp_mpm/mpm_solver.py would make, in the reference's order:
// MPMWARP(...) -> set_parameters_dict -> from_torch -> set_E_nu / prepare_mu_lam -> add_bounding_box -> p2g2p loop ->
// read-back).  Device memory comes straight from the HIP runtime; there are no torch types anywhere.
//   hipcc -O2 -Iinclude examples/c_abi_demo.cpp -o c_abi_demo -Lmpmavatar_amd/lib -lmpmhip -Wl,-rpath,$PWD/mpmavatar_amd/lib
//   ./c_abi_demo [substeps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mpmhip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MPM_OK(x) do { int r_ = (x); if (r_ != MPMHIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, mpmhip_last_error(ctx)); return 3; } } while (0)

template <class T>
static T *upload(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char **argv) {
  const int substeps = argc > 1 ? atoi(argv[1]) : 200;
  if (mpmhip_device_count() < 1) { fprintf(stderr, "no HIP device: libmpmhip has no CPU path\n"); return 1; }
  // 12^3 traditional particles, spacing 0.03, min corner (0.8, 1.0, 0.8); grid 32^3 over [0,2]^3
  const int n = 12, n_p = n * n * n, G = 32;
  const float sp = 0.03f, dt = 1e-4f, E = 100.0f, nu = 0.3f, density = 1.0f;
  std::vector<float> x(3 * n_p), v(3 * n_p, 0.0f), C(9 * n_p, 0.0f), F(9 * n_p, 0.0f), vol(n_p, sp * sp * sp), mass(n_p);
  for (int i = 0, p = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k, ++p) { x[3 * p] = 0.8f + sp * i; x[3 * p + 1] = 1.0f + sp * j; x[3 * p + 2] = 0.8f + sp * k; }
  for (int p = 0; p < n_p; ++p) { F[9 * p] = F[9 * p + 4] = F[9 * p + 8] = 1.0f; mass[p] = density * vol[p]; }
  const float mu0 = E / (2.0f * (1.0f + nu)), lam0 = E * nu / ((1.0f + nu) * (1.0f - 2.0f * nu));  // prepare_mu_lam
  std::vector<float> mu(n_p, mu0), lam(n_p, lam0), zero(n_p, 0.0f), stress(9 * n_p, 0.0f);
  std::vector<int32_t> sel(n_p, 0);

  mpmhip_ctx *ctx = nullptr;
  mpmhip_config cfg{};
  cfg.n_particles = n_p; cfg.n_elements = 0; cfg.n_vertices = 0; cfg.n_grid = G; cfg.grid_lim = 2.0f;
  cfg.device = 0; cfg.mode = MPMHIP_MODE_FAST; cfg.rebin_interval = 0; cfg.own_stream = 1;
  MPM_OK(mpmhip_create(&cfg, &ctx));

  mpmhip_state_ptrs st{};
  st.particle_x = upload(x); st.particle_v = upload(v); st.particle_C = upload(C);
  st.particle_F = upload(F); st.particle_F_trial = upload(F); st.particle_stress = upload(stress);
  st.particle_d = upload(std::vector<float>(1)); st.particle_R_inv = upload(std::vector<float>(1));
  st.faces = upload(std::vector<float>(1)); st.vertex_force = upload(std::vector<float>(1));
  st.particle_vol = upload(vol); st.particle_mass = upload(mass); st.particle_selection = upload(sel);
  mpmhip_model_ptrs md{upload(mu), upload(lam), upload(zero), upload(zero), upload(zero)};
  mpmhip_model_scalars sc{};
  sc.material = 0; sc.g[1] = -9.8f; sc.rpic_damping = 0.0f; sc.grid_v_damping_scale = 1.1f; sc.softening = 0.1f;
  MPM_OK(mpmhip_set_model_scalars(ctx, &sc));
  MPM_OK(mpmhip_bind_state(ctx, &st));
  MPM_OK(mpmhip_bind_model(ctx, &md));
  MPM_OK(mpmhip_add_bounding_box(ctx, 0.0f, 999.0f));

  MPM_OK(mpmhip_steps(ctx, dt, substeps, nullptr, nullptr, nullptr, 0, nullptr, nullptr));
  MPM_OK(mpmhip_pull_state(ctx));
  MPM_OK(mpmhip_synchronize(ctx));
  std::vector<float> xo(3 * n_p), vo(3 * n_p);
  HIP_OK(hipMemcpy(xo.data(), st.particle_x, xo.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(vo.data(), st.particle_v, vo.size() * sizeof(float), hipMemcpyDeviceToHost));
  double cy = 0, vy = 0, cy0 = 0;
  for (int p = 0; p < n_p; ++p) { cy += xo[3 * p + 1]; vy += vo[3 * p + 1]; cy0 += x[3 * p + 1]; }
  cy /= n_p; vy /= n_p; cy0 /= n_p;
  const double t = (double)substeps * dt;
  // free fall of the centre of mass (symplectic Euler: v after n steps = -g n dt, y drop = g dt^2 n (n + 1) / 2)
  const double vy_ref = -9.8 * t, drop_ref = 0.5 * 9.8 * dt * dt * (double)substeps * (substeps + 1);
  mpmhip_stats s{};
  MPM_OK(mpmhip_get_stats(ctx, &s));
  printf("substeps %lld  re-sorts %lld  active blocks %d  mean vy %.6f (free fall %.6f)  drop %.6e (free fall %.6e)  dropped %d\n",
         (long long)s.substeps, (long long)s.rebins, s.n_active_blocks, vy, vy_ref, cy0 - cy, drop_ref, s.n_dropped);
  const bool ok = std::fabs(vy - vy_ref) < 1e-4 * std::fabs(vy_ref) + 1e-6 && std::fabs((cy0 - cy) - drop_ref) < 1e-3 * drop_ref + 1e-6 &&
                  s.n_dropped == 0 && s.substeps == substeps;
  mpmhip_destroy(ctx);
  printf(ok ? "C ABI demo: OK\n" : "C ABI demo: MISMATCH\n");
  return ok ? 0 : 4;
}
