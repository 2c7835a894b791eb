// mpm_math.hpp -- per-thread fp32 3x3 algebra and constitutive models for gfx950.
//
// Everything here is a pure per-particle map held in VGPRs (no scratch arrays indexed
// dynamically).  Reference semantics: /root/reference/warp_mpm/mpm_utils.py:8-399.
// Cloth path: the sign-fixed QR of the reference (R00,R11 >= 0, det Q = +1; mpm_utils.py:109-123) by Givens
// rotations, operation for operation what the CPU oracle does (see qr_cloth); the rotation U V^T of the
// padded 2x2 block (mpm_utils.py:133-141) is the closed-form 2x2 polar rotation instead of an svd3.
#pragma once
#include <hip/hip_runtime.h>

namespace mpm {

struct V3 {
  float x, y, z;
};
struct M3 {  // row-major
  float a00, a01, a02, a10, a11, a12, a20, a21, a22;
};

__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
// wp.normalize: zero vector stays zero
__device__ __forceinline__ V3 normalize(V3 a) {
  float l = length(a);
  float inv = l > 0.0f ? 1.0f / l : 0.0f;
  return V3{a.x * inv, a.y * inv, a.z * inv};
}

__device__ __forceinline__ M3 m3_zero() { return M3{0, 0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ M3 m3_identity() { return M3{1, 0, 0, 0, 1, 0, 0, 0, 1}; }
__device__ __forceinline__ M3 m3_diag(float a, float b, float c) { return M3{a, 0, 0, 0, b, 0, 0, 0, c}; }
__device__ __forceinline__ M3 m3_cols(V3 c0, V3 c1, V3 c2) {
  return M3{c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z};
}
__device__ __forceinline__ V3 col0(const M3 &m) { return V3{m.a00, m.a10, m.a20}; }
__device__ __forceinline__ V3 col1(const M3 &m) { return V3{m.a01, m.a11, m.a21}; }
__device__ __forceinline__ V3 col2(const M3 &m) { return V3{m.a02, m.a12, m.a22}; }
__device__ __forceinline__ M3 operator+(const M3 &a, const M3 &b) {
  return M3{a.a00 + b.a00, a.a01 + b.a01, a.a02 + b.a02, a.a10 + b.a10, a.a11 + b.a11,
            a.a12 + b.a12, a.a20 + b.a20, a.a21 + b.a21, a.a22 + b.a22};
}
__device__ __forceinline__ M3 operator-(const M3 &a, const M3 &b) {
  return M3{a.a00 - b.a00, a.a01 - b.a01, a.a02 - b.a02, a.a10 - b.a10, a.a11 - b.a11,
            a.a12 - b.a12, a.a20 - b.a20, a.a21 - b.a21, a.a22 - b.a22};
}
__device__ __forceinline__ M3 operator*(float s, const M3 &a) {
  return M3{s * a.a00, s * a.a01, s * a.a02, s * a.a10, s * a.a11, s * a.a12, s * a.a20, s * a.a21, s * a.a22};
}
__device__ __forceinline__ M3 operator*(const M3 &a, const M3 &b) {
  return M3{a.a00 * b.a00 + a.a01 * b.a10 + a.a02 * b.a20, a.a00 * b.a01 + a.a01 * b.a11 + a.a02 * b.a21,
            a.a00 * b.a02 + a.a01 * b.a12 + a.a02 * b.a22, a.a10 * b.a00 + a.a11 * b.a10 + a.a12 * b.a20,
            a.a10 * b.a01 + a.a11 * b.a11 + a.a12 * b.a21, a.a10 * b.a02 + a.a11 * b.a12 + a.a12 * b.a22,
            a.a20 * b.a00 + a.a21 * b.a10 + a.a22 * b.a20, a.a20 * b.a01 + a.a21 * b.a11 + a.a22 * b.a21,
            a.a20 * b.a02 + a.a21 * b.a12 + a.a22 * b.a22};
}
// F_trial = (I + dt grad_v) F of a traditional particle (g2p_v, mpm_utils.py:780-786) with every product and every sum rounded on its
// own, as the reference's mat33 arithmetic is in the fixtures that pin this code.  Everywhere else hipcc may contract a * b + c into one
// FMA (-ffp-contract=fast-honor-pragmas); here it may not: F stays within ~1e-3 of a rotation, 1 + dt g rounds at 6e-8, and whether the
// product dt g is rounded before that sum decides the last bit of F -- which the elastic stress multiplies by E.  This one expression
// is what kept the spinning jelly cube of the reference's sequence fixture from the per-particle 1e-4 (2.2e-4 / 4.4e-4 with the
// contraction, 5.3e-5 / 4.1e-5 without; every other contraction of the library switched off on top of it: 5.3e-5;
// profiles/r06_experiments.md 2).  27 multiplies + 21 adds instead of 27 FMAs + 3 adds per traditional particle and substep.
__device__ __forceinline__ M3 deform_update(const M3 &G, float dt, const M3 &F) {
#pragma clang fp contract(off)
  float g00 = 1.0f + dt * G.a00, g01 = dt * G.a01, g02 = dt * G.a02;
  float g10 = dt * G.a10, g11 = 1.0f + dt * G.a11, g12 = dt * G.a12;
  float g20 = dt * G.a20, g21 = dt * G.a21, g22 = 1.0f + dt * G.a22;
  return M3{g00 * F.a00 + g01 * F.a10 + g02 * F.a20, g00 * F.a01 + g01 * F.a11 + g02 * F.a21, g00 * F.a02 + g01 * F.a12 + g02 * F.a22,
            g10 * F.a00 + g11 * F.a10 + g12 * F.a20, g10 * F.a01 + g11 * F.a11 + g12 * F.a21, g10 * F.a02 + g11 * F.a12 + g12 * F.a22,
            g20 * F.a00 + g21 * F.a10 + g22 * F.a20, g20 * F.a01 + g21 * F.a11 + g22 * F.a21, g20 * F.a02 + g21 * F.a12 + g22 * F.a22};
}
__device__ __forceinline__ V3 operator*(const M3 &a, V3 v) {
  return V3{a.a00 * v.x + a.a01 * v.y + a.a02 * v.z, a.a10 * v.x + a.a11 * v.y + a.a12 * v.z,
            a.a20 * v.x + a.a21 * v.y + a.a22 * v.z};
}
__device__ __forceinline__ M3 transpose(const M3 &a) {
  return M3{a.a00, a.a10, a.a20, a.a01, a.a11, a.a21, a.a02, a.a12, a.a22};
}
__device__ __forceinline__ float det(const M3 &a) {
  return a.a00 * (a.a11 * a.a22 - a.a12 * a.a21) - a.a01 * (a.a10 * a.a22 - a.a12 * a.a20) +
         a.a02 * (a.a10 * a.a21 - a.a11 * a.a20);
}
__device__ __forceinline__ M3 outer(V3 a, V3 b) {
  return M3{a.x * b.x, a.x * b.y, a.x * b.z, a.y * b.x, a.y * b.y, a.y * b.z, a.z * b.x, a.z * b.y, a.z * b.z};
}
__device__ __forceinline__ M3 load_m3(const float *p) {
  return M3{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
}
__device__ __forceinline__ void store_m3(float *p, const M3 &m) {
  p[0] = m.a00; p[1] = m.a01; p[2] = m.a02; p[3] = m.a10; p[4] = m.a11; p[5] = m.a12;
  p[6] = m.a20; p[7] = m.a21; p[8] = m.a22;
}
__device__ __forceinline__ V3 load_v3(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void store_v3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// ---------------------------------------------------------------------------------
// 3x3 SVD  A = U diag(s) V^T  by one-sided Jacobi on the columns of A (fully unrolled,
// register resident).  det V = +1 and det U = +1 with the sign of det A folded into
// s.z, |s| sorted descending -- the conventions of the McAdams routine behind wp.svd3.
// Reference consumers: mpm_utils.py:217,265,322,369,1077.
// ---------------------------------------------------------------------------------
// returns true while the pair is not yet orthogonal to fp32 resolution (drives the sweep loop's early exit)
__device__ __forceinline__ bool jacobi_pair(V3 &bp, V3 &bq, V3 &wp, V3 &wq) {
  float app = dot(bp, bp), aqq = dot(bq, bq), apq = dot(bp, bq);
  float lim = sqrtf(app * aqq);
  bool rot = fabsf(apq) > 1e-9f * lim;
  // v_rcp_f32 / v_rsq_f32 (1 ulp) instead of IEEE division sequences: the rotation only has to be orthonormal,
  // which c = rsq(1 + t^2), s = c t guarantees to rounding, and Jacobi iterations are self-correcting
  float tau = (aqq - app) * __builtin_amdgcn_rcpf(2.0f * (rot ? apq : 1.0f));
  float t = copysignf(1.0f, tau) * __builtin_amdgcn_rcpf(fabsf(tau) + sqrtf(1.0f + tau * tau));
  float c = __builtin_amdgcn_rsqf(1.0f + t * t);
  float s = c * t;
  c = rot ? c : 1.0f;
  s = rot ? s : 0.0f;
  V3 nbp = c * bp - s * bq, nbq = s * bp + c * bq;
  V3 nwp = c * wp - s * wq, nwq = s * wp + c * wq;
  bp = nbp; bq = nbq; wp = nwp; wq = nwq;
  return fabsf(apq) > 4e-7f * lim;
}

// conditional swap, component by component (a select on the V3 aggregate is lowered through scratch memory)
__device__ __forceinline__ void cswapf(bool c, float &a, float &b) {
  float t = a;
  a = c ? b : a;
  b = c ? t : b;
}
__device__ __forceinline__ void cswap(bool c, V3 &a, V3 &b) {
  cswapf(c, a.x, b.x);
  cswapf(c, a.y, b.y);
  cswapf(c, a.z, b.z);
}

__device__ __forceinline__ void svd3(const M3 &A, M3 &U, V3 &sig, M3 &V) {
  V3 b0 = col0(A), b1 = col1(A), b2 = col2(A);
  V3 w0 = v3(1, 0, 0), w1 = v3(0, 1, 0), w2 = v3(0, 0, 1);
  // cyclic sweeps; quadratic convergence: 2-3 sweeps for the near-rotations of elastic particles.  The exit test
  // is wave-uniform (one more sweep costs less than a divergent loop).
#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
    bool big = jacobi_pair(b0, b1, w0, w1);
    big |= jacobi_pair(b0, b2, w0, w2);
    big |= jacobi_pair(b1, b2, w1, w2);
    if (!__any(big)) break;
  }
  float n0 = dot(b0, b0), n1 = dot(b1, b1), n2 = dot(b2, b2);
  // sort by squared norm, descending (3-element network); count swaps to restore det V = +1
  bool s01 = n1 > n0;
  cswap(s01, b0, b1); cswap(s01, w0, w1);
  { float t = n0; n0 = s01 ? n1 : n0; n1 = s01 ? t : n1; }
  bool s02 = n2 > n0;
  cswap(s02, b0, b2); cswap(s02, w0, w2);
  { float t = n0; n0 = s02 ? n2 : n0; n2 = s02 ? t : n2; }
  bool s12 = n2 > n1;
  cswap(s12, b1, b2); cswap(s12, w1, w2);
  { float t = n1; n1 = s12 ? n2 : n1; n2 = s12 ? t : n2; }
  bool odd = (s01 != s02) != s12;
  if (odd) { b2 = -1.0f * b2; w2 = -1.0f * w2; }
  float s0 = sqrtf(n0), s1 = sqrtf(n1), s2 = sqrtf(n2);
  V3 u0 = s0 > 1e-20f ? (1.0f / s0) * b0 : v3(1, 0, 0);
  V3 u1;
  if (s1 > 1e-10f && s1 > 1e-7f * s0) {
    u1 = (1.0f / s1) * b1;
  } else {  // rank <= 1: any unit vector orthogonal to u0
    V3 a = fabsf(u0.x) <= fabsf(u0.y) && fabsf(u0.x) <= fabsf(u0.z) ? v3(1, 0, 0)
           : (fabsf(u0.y) <= fabsf(u0.z) ? v3(0, 1, 0) : v3(0, 0, 1));
    u1 = normalize(a - dot(a, u0) * u0);
  }
  V3 u2 = cross(u0, u1);
  if (dot(u2, b2) < 0.0f) s2 = -s2;
  U = m3_cols(u0, u1, u2);
  V = m3_cols(w0, w1, w2);
  sig = v3(s0, s1, s2);
}

// ---------------------------------------------------------------------------------
// isotropic Kirchhoff stresses, mpm_utils.py:8-84
// ---------------------------------------------------------------------------------
__device__ __forceinline__ M3 kirchhoff_FCR(const M3 &F, const M3 &U, const M3 &V, float J, float mu, float lam) {
  M3 R = U * transpose(V);
  M3 S = (2.0f * mu) * ((F - R) * transpose(F));
  float p = lam * J * (J - 1.0f);
  S.a00 += p; S.a11 += p; S.a22 += p;
  return S;
}
__device__ __forceinline__ M3 u_diag_vt_ft(const M3 &U, V3 t, const M3 &V, const M3 &F) {
  return U * m3_diag(t.x, t.y, t.z) * transpose(V) * transpose(F);
}
__device__ __forceinline__ M3 kirchhoff_StVK(const M3 &F, const M3 &U, const M3 &V, V3 sig, float mu, float lam) {
  V3 s = v3(fmaxf(sig.x, 0.01f), fmaxf(sig.y, 0.01f), fmaxf(sig.z, 0.01f));
  V3 e = v3(logf(s.x), logf(s.y), logf(s.z));
  float sum = e.x + e.y + e.z;
  V3 tau = v3(2.0f * mu * e.x + lam * sum, 2.0f * mu * e.y + lam * sum, 2.0f * mu * e.z + lam * sum);
  return u_diag_vt_ft(U, tau, V, F);
}
__device__ __forceinline__ M3 kirchhoff_drucker_prager(const M3 &F, const M3 &U, const M3 &V, V3 sig, float mu,
                                                       float lam) {
  float lx = logf(sig.x), ly = logf(sig.y), lz = logf(sig.z);  // no abs: NaN for inverted particles (quirk Q10)
  float sum = lx + ly + lz;
  V3 c = v3(2.0f * mu * lx * (1.0f / sig.x) + lam * sum * (1.0f / sig.x),
            2.0f * mu * ly * (1.0f / sig.y) + lam * sum * (1.0f / sig.y),
            2.0f * mu * lz * (1.0f / sig.z) + lam * sum * (1.0f / sig.z));
  return u_diag_vt_ft(U, c, V, F);
}

// ---------------------------------------------------------------------------------
// plastic return mappings, mpm_utils.py:212-399, in principal space.  Each takes the singular values of F_trial
// and returns true when the elastic deformation gradient changes, with its singular values in s_new
// (F = U diag(s_new) V^T).  Working on (U, s, V) lets the stress evaluation reuse the decomposition: the
// reference's second svd3 of the mapped F (mpm_utils.py:1077) would return exactly (U, s_new, V).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool von_mises_map(V3 so, float &ys, float &mu, float &lam, float hardening, float xi,
                                              float softening, bool damage, V3 &s_new) {
  V3 s = v3(fmaxf(so.x, 0.01f), fmaxf(so.y, 0.01f), fmaxf(so.z, 0.01f));
  V3 eps = v3(logf(s.x), logf(s.y), logf(s.z));
  float tr = eps.x + eps.y + eps.z, temp = tr / 3.0f;
  V3 tau = v3(2.0f * mu * eps.x + lam * tr, 2.0f * mu * eps.y + lam * tr, 2.0f * mu * eps.z + lam * tr);
  float st = tau.x + tau.y + tau.z;
  V3 cond = v3(tau.x - st / 3.0f, tau.y - st / 3.0f, tau.z - st / 3.0f);
  if (!(length(cond) > ys)) return false;
  if (damage && ys <= 0.0f) return false;
  V3 eh = v3(eps.x - temp, eps.y - temp, eps.z - temp);
  float ehn = length(eh) + 1e-6f;
  float dg = ehn - ys / (2.0f * mu);
  V3 corr = (dg / ehn) * eh;
  eps = eps - corr;
  if (damage) {
    ys = ys - softening * length(corr);
    if (ys <= 0.0f) { mu = 0.0f; lam = 0.0f; }
  }
  s_new = v3(expf(eps.x), expf(eps.y), expf(eps.z));
  if (hardening == 1.0f) ys = ys + 2.0f * mu * xi * dg;
  return true;
}

__device__ __forceinline__ bool viscoplastic_map(V3 so, float ys, float mu, float plastic_viscosity, float dt, V3 &s_new) {
  V3 s = v3(fmaxf(so.x, 0.01f), fmaxf(so.y, 0.01f), fmaxf(so.z, 0.01f));
  V3 b = v3(s.x * s.x, s.y * s.y, s.z * s.z);
  V3 eps = v3(logf(s.x), logf(s.y), logf(s.z));
  float tr = eps.x + eps.y + eps.z;
  V3 eh = v3(eps.x - tr / 3.0f, eps.y - tr / 3.0f, eps.z - tr / 3.0f);
  V3 st = (2.0f * mu) * eh;
  float stn = length(st);
  float y = stn - sqrtf(2.0f / 3.0f) * ys;
  if (!(y > 0.0f)) return false;
  float mu_hat = mu * (b.x + b.y + b.z) / 3.0f;
  float snn = stn - y / (1.0f + plastic_viscosity / (2.0f * mu_hat * dt));
  V3 sn = (snn / stn) * st;
  float k = 1.0f / (2.0f * mu);
  s_new = v3(expf(k * sn.x + tr / 3.0f), expf(k * sn.y + tr / 3.0f), expf(k * sn.z + tr / 3.0f));
  return true;
}

__device__ __forceinline__ bool sand_map(V3 sg, float mu, float lam, float alpha, V3 &s_new) {
  V3 eps = v3(logf(fmaxf(fabsf(sg.x), 1e-14f)), logf(fmaxf(fabsf(sg.y), 1e-14f)), logf(fmaxf(fabsf(sg.z), 1e-14f)));
  float tr = eps.x + eps.y + eps.z;
  V3 eh = v3(eps.x - tr / 3.0f, eps.y - tr / 3.0f, eps.z - tr / 3.0f);
  float ehn = length(eh);
  float dg = ehn + (3.0f * lam + 2.0f * mu) / (2.0f * mu) * tr * alpha;
  if (!(dg > 0.0f)) return false;                     // elastic (also the NaN case)
  if (tr > 0.0f) { s_new = v3(1, 1, 1); return true; }  // expansion: F = U V^T
  float k = dg / ehn;
  s_new = v3(expf(eps.x - eh.x * k), expf(eps.y - eh.y * k), expf(eps.z - eh.z * k));
  return true;
}

struct TradParams {  // MPMModelStruct scalars the traditional branch reads
  int material;
  float alpha, hardening, xi, plastic_viscosity, softening;
};

// compute_stress_from_F_trial for one traditional particle (mpm_utils.py:1047-1103) with a single SVD:
// F_trial -> (F, stress); mu / lam / ys are updated in place for the damage / hardening models.
__device__ __forceinline__ void traditional_update(const M3 &Ft, const TradParams &tp, float &mu, float &lam, float &ys,
                                                   float dt, M3 &F, M3 &stress) {
  int m = tp.material;
  F = Ft;
  stress = m3_zero();
  if (!(m == 0 || m == 1 || m == 2 || m == 3 || m == 5)) return;  // snow / neo-hookean / cloth: F <- F_trial, stress 0 (quirk Q4)
  M3 U, V;
  V3 sig;
  svd3(Ft, U, sig, V);
  V3 sn = sig;
  bool changed = false;
  if (m == 1 || m == 5) changed = von_mises_map(sig, ys, mu, lam, tp.hardening, tp.xi, tp.softening, m == 5, sn);
  else if (m == 2) changed = sand_map(sig, mu, lam, tp.alpha, sn);
  else if (m == 3) changed = viscoplastic_map(sig, ys, mu, tp.plastic_viscosity, dt, sn);
  if (changed) {
    F = U * m3_diag(sn.x, sn.y, sn.z) * transpose(V);
    sig = sn;
  }
  float J = det(F);
  if (m == 0 || m == 5) stress = kirchhoff_FCR(F, U, V, J, mu, lam);
  else if (m == 2) stress = kirchhoff_drucker_prager(F, U, V, sig, mu, lam);
  else stress = kirchhoff_StVK(F, U, V, sig, mu, lam);
  stress = 0.5f * (stress + transpose(stress));
}

// ---------------------------------------------------------------------------------
// anisotropic cloth: sign-fixed QR (Gram-Schmidt), return mapping (mpm_utils.py:179-209)
// and Kirchhoff stress + vertex forces (mpm_utils.py:101-177).
// ---------------------------------------------------------------------------------
struct QR3 {
  V3 q0, q1, q2;
  float r00, r01, r02, r11, r12, r22;
};

// The return mapping below branches on r22 > 1 and on a friction threshold, and a flat garment sits exactly on the
// first: these two functions are compiled without FMA contraction so that every kernel that inlines them (fused and
// stand-alone element finalize, both back ends) takes the same branch on the same input, whatever the surrounding
// code lets the compiler fuse (needs -ffp-contract=fast-honor-pragmas, see build.py).
//
// QR by three Givens rotations (zeroing d10, d20, d21 in that order; det Q = +1) followed by the reference's two sign
// flips (mpm_utils.py:109-123,181-195) -- the algorithm behind wp.qr3, in the operation order of the test suite's CPU
// restatement of it (checked bit for bit by tests/test_hip_math_on_host.py), so that r22 carries the same rounding here and there.
// It matters: a cloth at rest has r22 = 1 exactly, the return mapping branches on r22 > 1, and WHICH way an fp32 QR
// rounds r22 there is a property of the algorithm.  Rounds 1-2 used Gram-Schmidt with q2 = q0 x q1 (same Q, R in exact
// arithmetic): |q2| = 1 +- ulp is not re-normalised, r22 = q2 . d2 inherits that bias, and the run left the
// reference-produced cloth sequences 2.7x further (2.6e-3 / 1.6e-3 in v on ref_seq_sheet / ref_seq_garment after 80
// substeps) than the oracle did (9.7e-4 / 5.8e-4) -- reproduced and bisected on the CPU by swapping this one function
// (profiles/r03_cloth_qr_bisect.md).  Qt = G3 G2 G1 starts as the identity, so its products with 0 and 1 are written out.
__device__ __forceinline__ QR3 qr_cloth(const M3 &d) {
#pragma clang fp contract(off)
  QR3 o;
  // G1: rows 0, 1 from (d00, d10)
  float n1 = sqrtf(d.a00 * d.a00 + d.a10 * d.a10);
  float c1 = n1 == 0.0f ? 1.0f : d.a00 / n1, s1 = n1 == 0.0f ? 0.0f : d.a10 / n1;
  float t00 = c1 * d.a00 + s1 * d.a10, t01 = c1 * d.a01 + s1 * d.a11, t02 = c1 * d.a02 + s1 * d.a12;
  float t11 = (-s1) * d.a01 + c1 * d.a11, t12 = (-s1) * d.a02 + c1 * d.a12;
  // G2: rows 0, 2 from (t00, d20)
  float n2 = sqrtf(t00 * t00 + d.a20 * d.a20);
  float c2 = n2 == 0.0f ? 1.0f : t00 / n2, s2 = n2 == 0.0f ? 0.0f : d.a20 / n2;
  float u00 = c2 * t00 + s2 * d.a20, u01 = c2 * t01 + s2 * d.a21, u02 = c2 * t02 + s2 * d.a22;
  float u21 = (-s2) * t01 + c2 * d.a21, u22 = (-s2) * t02 + c2 * d.a22;
  // G3: rows 1, 2 from (t11, u21)
  float n3 = sqrtf(t11 * t11 + u21 * u21);
  float c3 = n3 == 0.0f ? 1.0f : t11 / n3, s3 = n3 == 0.0f ? 0.0f : u21 / n3;
  float w11 = c3 * t11 + s3 * u21, w12 = c3 * t12 + s3 * u22;
  float w22 = (-s3) * t12 + c3 * u22;
  // rows of Qt = columns of Q
  float a0x = c2 * c1, a0y = c2 * s1, a0z = s2;                 // row 0 after G2
  float a2x = (-s2) * c1, a2y = (-s2) * s1, a2z = c2;           // row 2 after G2
  float b1x = c3 * (-s1) + s3 * a2x, b1y = c3 * c1 + s3 * a2y, b1z = s3 * a2z;      // row 1 after G3
  float b2x = (-s3) * (-s1) + c3 * a2x, b2y = (-s3) * c1 + c3 * a2y, b2z = c3 * a2z;  // row 2 after G3
  o.q0 = v3(a0x, a0y, a0z); o.q1 = v3(b1x, b1y, b1z); o.q2 = v3(b2x, b2y, b2z);
  o.r00 = u00; o.r01 = u01; o.r02 = u02; o.r11 = w11; o.r12 = w12; o.r22 = w22;
  if (o.r00 < 0.0f) {  // mpm_utils.py:112-114: columns 0, 2 of Q and rows 0, 2 of R change sign
    o.q0 = v3(-o.q0.x, -o.q0.y, -o.q0.z); o.q2 = v3(-o.q2.x, -o.q2.y, -o.q2.z);
    o.r00 = -o.r00; o.r01 = -o.r01; o.r02 = -o.r02; o.r22 = -o.r22;
  }
  if (o.r11 < 0.0f) {  // :118-120: columns 1, 2 of Q and rows 1, 2 of R
    o.q1 = v3(-o.q1.x, -o.q1.y, -o.q1.z); o.q2 = v3(-o.q2.x, -o.q2.y, -o.q2.z);
    o.r11 = -o.r11; o.r12 = -o.r12; o.r22 = -o.r22;
  }
  return o;
}

// returns the new third director d3 (columns d1,d2 are unchanged)
__device__ __forceinline__ V3 anisotropy_return_mapping(const QR3 &q, float gamma, float kappa, float friction_coeff,
                                                        float &r02, float &r12, float &r22) {
#pragma clang fp contract(off)
  r02 = q.r02; r12 = q.r12; r22 = q.r22;
  if (q.r22 > 1.0f) {
    r22 = 1.0f;
  } else {
    float fn = kappa * (1.0f - q.r22) * (1.0f - q.r22);
    float ff = gamma * sqrtf(q.r02 * q.r02 + q.r12 * q.r12);
    if (ff > friction_coeff * fn) {
      r02 = q.r02 * friction_coeff * fn / ff;
      r12 = q.r12 * friction_coeff * fn / ff;
    }
  }
  float ax = r02 * q.q0.x, ay = r02 * q.q0.y, az = r02 * q.q0.z;
  float bx = r12 * q.q1.x, by = r12 * q.q1.y, bz = r12 * q.q1.z;
  float cx = r22 * q.q2.x, cy = r22 * q.q2.y, cz = r22 * q.q2.z;
  return v3((ax + bx) + cx, (ay + by) + cy, (az + bz) + cz);
}

// Given the QR of the *mapped* d (q0,q1,q2 unchanged by the return mapping because d1,d2 are;
// r02,r12,r22 are the mapped values), compute stress = vol * P3 (x) d3 and the three vertex forces.
__device__ __forceinline__ void kirchhoff_anisotropy(const QR3 &q, float r02, float r12, float r22, V3 d3, V3 Rinv,
                                                     float vol, float mu, float lam, float gamma, float kappa,
                                                     M3 &stress, V3 &f1, V3 &f2, V3 &f3) {
  float iD11 = Rinv.x, iD12 = Rinv.y, iD22 = Rinv.z;
  float F11 = q.r00 * iD11;
  float F12 = q.r00 * iD12 + q.r01 * iD22;
  float F22 = q.r11 * iD22;
  // polar rotation of [[F11,F12],[0,F22]]: angle atan2(-F12, F11+F22)
  float hx = F11 + F22, hy = -F12;
  float hn = sqrtf(hx * hx + hy * hy);
  float c = hx / hn, s = hy / hn;
  float J = F11 * F22;
  float lj = lam * (J - 1.0f), m2 = 2.0f * mu;
  // K2 = 2mu(F2 - Rot) + lam(J-1) [[F22,0],[-F12,F11]]; Rot = [[c,-s],[s,c]]
  float K00 = m2 * (F11 - c) + lj * F22;
  float K01 = m2 * (F12 + s);
  float K11 = m2 * (F22 - c) + lj * F11;
  float dr13 = gamma * r02, dr23 = gamma * r12;
  float dr33 = (r22 > 1.0f) ? 0.0f : -kappa * (1.0f - r22) * (1.0f - r22);
  // K3 = dr * RiDT, RiDT = [[F11,0,0],[F12,F22,0],[r02,r12,r22]]; only the upper triangle is used
  float k00 = K00 * F11 + K01 * F12 + dr13 * r02;
  float k01 = K01 * F22 + dr13 * r12;
  float k02 = dr13 * r22;
  float k11 = K11 * F22 + dr23 * r12;
  float k12 = dr23 * r22;
  float k22 = dr33 * r22;
  M3 K3s = M3{k00, k01, k02, k01, k11, k12, k02, k12, k22};
  // inverse of lower-triangular RiDT (mpm_utils.py:87-99)
  float invdet = 1.0f / (F11 * F22 * r22);
  M3 Rinvm = invdet * M3{F22 * r22, 0.0f, 0.0f, -F12 * r22, F11 * r22, 0.0f, F12 * r12 - r02 * F22, -F11 * r12, F11 * F22};
  M3 Q = m3_cols(q.q0, q.q1, q.q2);
  M3 P = Q * K3s * Rinvm;
  V3 P1 = col0(P), P2 = col1(P), P3 = col2(P);
  f2 = (-vol) * (iD11 * P1 + iD12 * P2);
  f3 = (-vol * iD22) * P2;
  f1 = -1.0f * (f2 + f3);
  stress = vol * outer(P3, d3);
}

// ---------------------------------------------------------------------------------
// quadratic B-spline stencil shared by every transfer kernel (mpm_utils.py:499-526)
// ---------------------------------------------------------------------------------
struct Stencil {
  int bx, by, bz;
  V3 fx;
  V3 w0, w1, w2;     // w[node] as vec over axes
  V3 dw0, dw1, dw2;  // dw[node] as vec over axes
};

__device__ __forceinline__ Stencil make_stencil(V3 x, float inv_dx) {
  Stencil s;
  V3 g = inv_dx * x;
  s.bx = (int)(g.x - 0.5f);
  s.by = (int)(g.y - 0.5f);
  s.bz = (int)(g.z - 0.5f);
  s.fx = v3(g.x - (float)s.bx, g.y - (float)s.by, g.z - (float)s.bz);
  V3 wa = v3(1.5f - s.fx.x, 1.5f - s.fx.y, 1.5f - s.fx.z);
  V3 wb = v3(s.fx.x - 1.0f, s.fx.y - 1.0f, s.fx.z - 1.0f);
  V3 wc = v3(s.fx.x - 0.5f, s.fx.y - 0.5f, s.fx.z - 0.5f);
  s.w0 = v3(wa.x * wa.x * 0.5f, wa.y * wa.y * 0.5f, wa.z * wa.z * 0.5f);
  s.w1 = v3(0.75f - wb.x * wb.x, 0.75f - wb.y * wb.y, 0.75f - wb.z * wb.z);
  s.w2 = v3(wc.x * wc.x * 0.5f, wc.y * wc.y * 0.5f, wc.z * wc.z * 0.5f);
  s.dw0 = v3(s.fx.x - 1.5f, s.fx.y - 1.5f, s.fx.z - 1.5f);
  s.dw1 = v3(-2.0f * wb.x, -2.0f * wb.y, -2.0f * wb.z);
  s.dw2 = wc;
  return s;
}
__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }
// weight of stencil node i (compile-time 0..2) at fractional offset f, and its derivative: the values make_stencil
// stores, recomputed where a kernel would otherwise keep nine of them in registers for a whole loop nest
__device__ __forceinline__ float bspline_w(int i, float f) {
  return i == 0 ? 0.5f * (1.5f - f) * (1.5f - f) : (i == 1 ? 0.75f - (f - 1.0f) * (f - 1.0f) : 0.5f * (f - 0.5f) * (f - 0.5f));
}
__device__ __forceinline__ float bspline_dw(int i, float f) { return i == 0 ? f - 1.5f : (i == 1 ? 2.0f - 2.0f * f : f - 0.5f); }

// body-mesh vertex i at this substep: x + adv*v with the multiply and the add rounded separately (the caller's
// torch expression mesh_x + f*mesh_v, train_material_params.py:623)
__device__ __forceinline__ V3 mesh_point(const float *pts, const float *vel, float adv, int i) {
#pragma clang fp contract(off)
  V3 p = load_v3(pts + 3 * i);
  if (adv == 0.0f) return p;
  V3 u = load_v3(vel + 3 * i);
  float ax = adv * u.x, ay = adv * u.y, az = adv * u.z;
  return v3(p.x + ax, p.y + ay, p.z + az);
}

// collider response on one node (mpm_solver.py:898-917): v is grid_v_out, vm the splatted body
// velocity, nsum the splatted (unnormalised) normal.
__device__ __forceinline__ V3 collide_node(V3 v, V3 vm, V3 nsum, float friction) {
  V3 vrel = v - vm;
  V3 n = normalize(nsum);
  float nc = dot(vrel, n);
  V3 vproj = vrel - fminf(nc, 0.0f) * n;
  float lp = length(vproj);
  V3 vfric = vproj;
  if (nc < 0.0f && lp > 1e-20f) vfric = fmaxf(0.0f, lp + nc * friction) * normalize(vproj);
  return vfric + vm;
}

}  // namespace mpm
