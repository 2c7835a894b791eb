// The product's per-particle device math (mpmavatar_amd/csrc/mpm_math.hpp) compiled for the host and exposed with the
// signatures of the oracle's test hooks (oracle/mpm_oracle.c: orc_hook_element / orc_hook_traditional).  The bodies below
// are the bodies of k_stress_elem / traditional_update's callers in csrc/fast.hip, minus the memory traffic.
#include "mpm_math.hpp"
#include <cstring>
using namespace mpm;

extern "C" void hm_element(const float *d, const float *R_inv, float vol, float mu, float lam, float gamma, float kappa,
                           float friction_coeff, float *new_d, float *stress, float *f1, float *f2, float *f3) {
  QR3 q = qr_cloth(load_m3(d));
  float r02, r12, r22;
  V3 d3 = anisotropy_return_mapping(q, gamma, kappa, friction_coeff, r02, r12, r22);
  float nd[9] = {d[0], d[1], d3.x, d[3], d[4], d3.y, d[6], d[7], d3.z};
  memcpy(new_d, nd, sizeof nd);
  M3 S;
  V3 a, b, c;
  kirchhoff_anisotropy(q, r02, r12, r22, d3, v3(R_inv[0], R_inv[1], R_inv[2]), vol, mu, lam, gamma, kappa, S, a, b, c);
  store_m3(stress, S); store_v3(f1, a); store_v3(f2, b); store_v3(f3, c);
}

extern "C" void hm_traditional(const float *F_trial, int material, float alpha, float hardening, float xi,
                               float plastic_viscosity, float softening, float dt, float *mu, float *lam, float *ys,
                               float *F, float *stress) {
  TradParams tp{material, alpha, hardening, xi, plastic_viscosity, softening};
  M3 Fo, S;
  float m = *mu, l = *lam, y = *ys;
  traditional_update(load_m3(F_trial), tp, m, l, y, dt, Fo, S);
  if (material == 1 || material == 5) *ys = y;      // what k_stress_trad / p2g_finish store
  if (material == 5) { *mu = m; *lam = l; }
  store_m3(F, Fo); store_m3(stress, S);
}

// Gram-Schmidt form of the sign-fixed QR that rounds 1-2 shipped (q2 = q0 x q1): kept here, test-side only, as the
// witness for the bias it introduces at the r22 = 1 discontinuity of the return mapping
static QR3 qr_gram_schmidt(const M3 &d) {
  V3 d0 = col0(d), d1 = col1(d), d2 = col2(d);
  QR3 o;
  o.r00 = sqrtf((d0.x * d0.x + d0.y * d0.y) + d0.z * d0.z);
  float i0 = 1.0f / o.r00;
  o.q0 = v3(i0 * d0.x, i0 * d0.y, i0 * d0.z);
  o.r01 = (o.q0.x * d1.x + o.q0.y * d1.y) + o.q0.z * d1.z;
  float ax = o.r01 * o.q0.x, ay = o.r01 * o.q0.y, az = o.r01 * o.q0.z;
  V3 u1 = v3(d1.x - ax, d1.y - ay, d1.z - az);
  o.r11 = sqrtf((u1.x * u1.x + u1.y * u1.y) + u1.z * u1.z);
  float i1 = 1.0f / o.r11;
  o.q1 = v3(i1 * u1.x, i1 * u1.y, i1 * u1.z);
  float c0a = o.q0.y * o.q1.z, c0b = o.q0.z * o.q1.y, c1a = o.q0.z * o.q1.x, c1b = o.q0.x * o.q1.z;
  float c2a = o.q0.x * o.q1.y, c2b = o.q0.y * o.q1.x;
  o.q2 = v3(c0a - c0b, c1a - c1b, c2a - c2b);
  o.r02 = (o.q0.x * d2.x + o.q0.y * d2.y) + o.q0.z * d2.z;
  o.r12 = (o.q1.x * d2.x + o.q1.y * d2.y) + o.q1.z * d2.z;
  o.r22 = (o.q2.x * d2.x + o.q2.y * d2.y) + o.q2.z * d2.z;
  return o;
}
extern "C" void hm_element_gram_schmidt(const float *d, const float *R_inv, float vol, float mu, float lam, float gamma,
                                        float kappa, float friction_coeff, float *new_d, float *stress, float *f1, float *f2,
                                        float *f3) {
  QR3 q = qr_gram_schmidt(load_m3(d));
  float r02, r12, r22;
  V3 d3 = anisotropy_return_mapping(q, gamma, kappa, friction_coeff, r02, r12, r22);
  float nd[9] = {d[0], d[1], d3.x, d[3], d[4], d3.y, d[6], d[7], d3.z};
  memcpy(new_d, nd, sizeof nd);
  M3 S;
  V3 a, b, c;
  kirchhoff_anisotropy(q, r02, r12, r22, d3, v3(R_inv[0], R_inv[1], R_inv[2]), vol, mu, lam, gamma, kappa, S, a, b, c);
  store_m3(stress, S); store_v3(f1, a); store_v3(f2, b); store_v3(f3, c);
}

extern "C" void hm_qr(const float *d, float *Q, float *R) {  // Q, R row-major, as orc_qr_signfixed returns them
  QR3 q = qr_cloth(load_m3(d));
  store_m3(Q, m3_cols(q.q0, q.q1, q.q2));
  float r[9] = {q.r00, q.r01, q.r02, 0.f, q.r11, q.r12, 0.f, 0.f, q.r22};
  memcpy(R, r, sizeof r);
}
