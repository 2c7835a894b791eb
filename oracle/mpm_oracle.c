/*
 * oracle/mpm_oracle.c -- TEST INFRASTRUCTURE ONLY (see mpm_oracle.h header).
 *
 * Serial fp32 CPU restatement of the reference substep MPMWARP.p2g2p
 * (/root/reference/warp_mpm/mpm_solver.py:229-536) and the kernels it launches
 * (warp_mpm/mpm_utils.py).  Every function cites the reference lines it follows.
 * Kernel = serial `for tid` loop in reference thread order; inner stencil loops in
 * reference order (i outer, j, k inner).  Pinned by fixtures the reference's own kernel
 * bodies produced over a NumPy stand-in of the warp module (mpm_oracle.h, PINNING).
 *
 * Compile:  gcc -O2 -ffp-contract=off -fPIC -shared  (serial oracle)
 *           gcc -O2 -fopenmp -DORC_OMP ...           (multi-core CPU baseline)
 *
 * Third-party arithmetic restated here because it lives in warp-lang 0.10.1
 * (requirements.txt:36), not in the reference tree: wp.svd3, wp.qr3, wp.normalize,
 * wp.mesh_eval_face_normal, wp.int truncation, wp.atomic_add.
 */
#include "mpm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_OMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(static)")
#define ORC_ATOMIC _Pragma("omp atomic")
#else
#define ORC_PARALLEL_FOR
#define ORC_ATOMIC
#endif

/* ------------------------------------------------------------------ */
/* small fp32 linear algebra (row-major mat33)                         */
/* ------------------------------------------------------------------ */
static void m_mul(const float *A, const float *B, float *C) {
  float T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      T[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] +
                     A[r * 3 + 2] * B[2 * 3 + c];
  memcpy(C, T, sizeof T);
}
static void m_T(const float *A, float *B) {
  float T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[c * 3 + r];
  memcpy(B, T, sizeof T);
}
static float m_det(const float *A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}
static void m_vec(const float *A, const float *x, float *y) {
  float t0 = A[0] * x[0] + A[1] * x[1] + A[2] * x[2];
  float t1 = A[3] * x[0] + A[4] * x[1] + A[5] * x[2];
  float t2 = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
  y[0] = t0; y[1] = t1; y[2] = t2;
}
static void m_diag(float a, float b, float c, float *D) {
  memset(D, 0, 9 * sizeof(float));
  D[0] = a; D[4] = b; D[8] = c;
}
static float v_len(const float *a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
/* wp.normalize: v/|v|, zero vector stays zero (warp/native/vec.h) */
static void v_normalize(const float *a, float *o) {
  float l = v_len(a);
  if (l > 0.0f) { o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l; }
  else { o[0] = o[1] = o[2] = 0.0f; }
}
static float f_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------ */
/* wp.svd3 restatement: A = U diag(sig) V^T.                           */
/* One-sided (Hestenes) Jacobi.  Conventions chosen to match the       */
/* McAdams-style routine Warp ships: det U = det V = +1, |sig| sorted   */
/* descending, the sign of det A carried by the smallest singular value.*/
/* All reference call sites use convention-independent combinations     */
/* (U V^T, U f(S) V^T) when det A > 0.                 */
/* ------------------------------------------------------------------ */
void orc_svd3(const float *A, float *U, float *sig, float *V) {
  float B[9], W[9];
  memcpy(B, A, sizeof B); /* B = A * W, columns get orthogonalised */
  m_diag(1.f, 1.f, 1.f, W);
  for (int sweep = 0; sweep < 12; ++sweep) {
    float off = 0.f;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        float app = 0.f, aqq = 0.f, apq = 0.f;
        for (int r = 0; r < 3; ++r) {
          app += B[r * 3 + p] * B[r * 3 + p];
          aqq += B[r * 3 + q] * B[r * 3 + q];
          apq += B[r * 3 + p] * B[r * 3 + q];
        }
        if (fabsf(apq) <= 1e-9f * sqrtf(app * aqq) || apq == 0.f) continue;
        off += fabsf(apq);
        float tau = (aqq - app) / (2.f * apq);
        float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
        float c = 1.f / sqrtf(1.f + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          float bp = B[r * 3 + p], bq = B[r * 3 + q];
          B[r * 3 + p] = c * bp - s * bq;
          B[r * 3 + q] = s * bp + c * bq;
          float wp_ = W[r * 3 + p], wq = W[r * 3 + q];
          W[r * 3 + p] = c * wp_ - s * wq;
          W[r * 3 + q] = s * wp_ + c * wq;
        }
      }
    if (off == 0.f) break;
  }
  float n[3];
  int idx[3] = {0, 1, 2};
  for (int c = 0; c < 3; ++c)
    n[c] = sqrtf(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]);
  /* sort descending */
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (n[idx[b]] > n[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
  float Uc[3][3], Vc[3][3], s3[3];
  for (int c = 0; c < 3; ++c) {
    int k = idx[c];
    s3[c] = n[k];
    for (int r = 0; r < 3; ++r) { Vc[c][r] = W[r * 3 + k]; Uc[c][r] = B[r * 3 + k]; }
  }
  /* make V a rotation (the sort may have introduced a reflection) */
  {
    float Vm[9];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Vm[r * 3 + c] = Vc[c][r];
    if (m_det(Vm) < 0.f)
      for (int r = 0; r < 3; ++r) { Vc[2][r] = -Vc[2][r]; Uc[2][r] = -Uc[2][r]; }
  }
  /* U columns */
  float tiny = 1e-20f;
  if (s3[0] > tiny) for (int r = 0; r < 3; ++r) Uc[0][r] /= s3[0];
  else { Uc[0][0] = 1.f; Uc[0][1] = 0.f; Uc[0][2] = 0.f; }
  if (s3[1] > tiny * 1e10f && s3[1] > 1e-7f * s3[0]) {
    for (int r = 0; r < 3; ++r) Uc[1][r] /= s3[1];
  } else { /* rank <= 1: any unit vector orthogonal to u0 */
    float a[3] = {0.f, 0.f, 0.f};
    int m = 0;
    if (fabsf(Uc[0][1]) < fabsf(Uc[0][m])) m = 1;
    if (fabsf(Uc[0][2]) < fabsf(Uc[0][m])) m = 2;
    a[m] = 1.f;
    float dp = a[0] * Uc[0][0] + a[1] * Uc[0][1] + a[2] * Uc[0][2];
    for (int r = 0; r < 3; ++r) a[r] -= dp * Uc[0][r];
    v_normalize(a, Uc[1]);
  }
  {
    /* third column: u0 x u1 gives det U = +1; sign of sigma_3 absorbs det A */
    float cx = Uc[0][1] * Uc[1][2] - Uc[0][2] * Uc[1][1];
    float cy = Uc[0][2] * Uc[1][0] - Uc[0][0] * Uc[1][2];
    float cz = Uc[0][0] * Uc[1][1] - Uc[0][1] * Uc[1][0];
    float dp = cx * Uc[2][0] + cy * Uc[2][1] + cz * Uc[2][2]; /* Uc[2] still unnormalised */
    if (dp < 0.f) s3[2] = -s3[2];
    Uc[2][0] = cx; Uc[2][1] = cy; Uc[2][2] = cz;
  }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) { U[r * 3 + c] = Uc[c][r]; V[r * 3 + c] = Vc[c][r]; }
  sig[0] = s3[0]; sig[1] = s3[1]; sig[2] = s3[2];
}

/* ------------------------------------------------------------------ */
/* wp.qr3 (Givens, det Q = +1) followed by the reference's two sign    */
/* flips so that R00 >= 0 and R11 >= 0 (mpm_utils.py:109-123,181-195). */
/* ------------------------------------------------------------------ */
static void givens(float a, float b, float *c, float *s) {
  float r = sqrtf(a * a + b * b);
  if (r == 0.f) { *c = 1.f; *s = 0.f; }
  else { *c = a / r; *s = b / r; }
}
static void rot_rows(float *M, int i, int j, float c, float s) { /* rows i,j <- G * rows */
  for (int k = 0; k < 3; ++k) {
    float a = M[i * 3 + k], b = M[j * 3 + k];
    M[i * 3 + k] = c * a + s * b;
    M[j * 3 + k] = -s * a + c * b;
  }
}
void orc_qr_signfixed(const float *d, float *Q, float *R) {
  float R0[9], Qt[9], c, s;
  memcpy(R0, d, sizeof R0);
  m_diag(1.f, 1.f, 1.f, Qt); /* Qt accumulates G3 G2 G1 = Q^T */
  givens(R0[0], R0[3], &c, &s); rot_rows(R0, 0, 1, c, s); rot_rows(Qt, 0, 1, c, s);
  givens(R0[0], R0[6], &c, &s); rot_rows(R0, 0, 2, c, s); rot_rows(Qt, 0, 2, c, s);
  givens(R0[4], R0[7], &c, &s); rot_rows(R0, 1, 2, c, s); rot_rows(Qt, 1, 2, c, s);
  R0[3] = 0.f; R0[6] = 0.f; R0[7] = 0.f;
  float Q0[9];
  m_T(Qt, Q0);
  float Q1[9], R1[9];
  if (R0[0] < 0.f) { /* mpm_utils.py:112-114 */
    float q[9] = {-Q0[0], Q0[1], -Q0[2], -Q0[3], Q0[4], -Q0[5], -Q0[6], Q0[7], -Q0[8]};
    float r[9] = {-R0[0], -R0[1], -R0[2], 0.f, R0[4], R0[5], 0.f, 0.f, -R0[8]};
    memcpy(Q1, q, sizeof q); memcpy(R1, r, sizeof r);
  } else { memcpy(Q1, Q0, sizeof Q0); memcpy(R1, R0, sizeof R0); }
  if (R1[4] < 0.f) { /* mpm_utils.py:118-120 */
    float q[9] = {Q1[0], -Q1[1], -Q1[2], Q1[3], -Q1[4], -Q1[5], Q1[6], -Q1[7], -Q1[8]};
    float r[9] = {R1[0], R1[1], R1[2], 0.f, -R1[4], -R1[5], 0.f, 0.f, -R1[8]};
    memcpy(Q, q, sizeof q); memcpy(R, r, sizeof r);
  } else { memcpy(Q, Q1, sizeof Q1); memcpy(R, R1, sizeof R1); }
}

/* anisotropy_return_mapping, mpm_utils.py:179-209 */
void orc_anisotropy_return_mapping(const float *d, float gamma, float kappa,
                                   float friction_coeff, float *new_d) {
  float Q[9], R2[9], R[9];
  orc_qr_signfixed(d, Q, R2);
  memcpy(R, R2, sizeof R);
  if (R2[8] > 1.0f) { /* :196-197 */
    R[6] = 0.f; R[7] = 0.f; R[8] = 1.0f;
  } else {
    float fn = kappa * (1.0f - R2[8]) * (1.0f - R2[8]);
    float ff = gamma * sqrtf(R2[2] * R2[2] + R2[5] * R2[5]);
    if (ff > friction_coeff * fn) { /* :201-202 */
      R[2] = R2[2] * friction_coeff * fn / ff;
      R[5] = R2[5] * friction_coeff * fn / ff;
    }
  }
  float r3[3] = {R[2], R[5], R[8]}, d3[3];
  m_vec(Q, r3, d3); /* :206 */
  float nd[9] = {d[0], d[1], d3[0], d[3], d[4], d3[1], d[6], d[7], d3[2]};
  memcpy(new_d, nd, sizeof nd);
}

/* inverse_lower_triangle, mpm_utils.py:87-99 */
static void inverse_lower_triangle(const float *M, float *O) {
  float M11 = M[0], M21 = M[3], M22 = M[4], M31 = M[6], M32 = M[7], M33 = M[8];
  float invdet = 1.0f / (M11 * M22 * M33);
  float o[9] = {M22 * M33, 0.f, 0.f, -M21 * M33, M11 * M33, 0.f,
                M21 * M32 - M31 * M22, -M11 * M32, M11 * M22};
  for (int i = 0; i < 9; ++i) O[i] = invdet * o[i];
}

/* kirchoff_stress_Anisotropy, mpm_utils.py:101-177 (vertex forces returned, caller scatters) */
void orc_kirchhoff_anisotropy(const float *R_inv, const float *d, float vol, float mu,
                              float lam, float gamma, float kappa, float *stress_out,
                              float *f1, float *f2, float *f3) {
  float iD11 = R_inv[0], iD12 = R_inv[1], iD22 = R_inv[2];
  float Q[9], R[9];
  orc_qr_signfixed(d, Q, R);
  float F11 = R[0] * iD11;
  float F12 = R[0] * iD12 + R[1] * iD22;
  float F22 = R[4] * iD22;
  float RiDT[9] = {F11, 0.f, 0.f, F12, F22, 0.f, R[2], R[5], R[8]};
  float iFTJ[4] = {F22, 0.f, -F12, F11};
  float F3[9] = {F11, F12, 0.f, 0.f, F22, 0.f, 0.f, 0.f, 0.f};
  float U3[9], V3[9], sig3[3];
  orc_svd3(F3, U3, sig3, V3); /* :137 */
  /* Rot = U2 * V2^T, :138-141 */
  float Rot[4] = {U3[0] * V3[0] + U3[1] * V3[1], U3[0] * V3[3] + U3[1] * V3[4],
                  U3[3] * V3[0] + U3[4] * V3[1], U3[3] * V3[3] + U3[4] * V3[4]};
  float J = F11 * F22;
  float F2[4] = {F11, F12, 0.f, F22};
  float K2[4];
  for (int i = 0; i < 4; ++i) K2[i] = 2.0f * mu * (F2[i] - Rot[i]) + lam * (J - 1.0f) * iFTJ[i];
  float dr33 = (R[8] > 1.0f) ? 0.0f : -kappa * (1.0f - R[8]) * (1.0f - R[8]);
  float dr[9] = {K2[0], K2[1], gamma * R[2], 0.f, K2[3], gamma * R[5], 0.f, 0.f, dr33};
  float K3[9];
  m_mul(dr, RiDT, K3);
  float K3s[9] = {K3[0], K3[1], K3[2], K3[1], K3[4], K3[5], K3[2], K3[5], K3[8]};
  float RiDTinv[9], P[9], T[9];
  inverse_lower_triangle(RiDT, RiDTinv);
  m_mul(Q, K3s, T);
  m_mul(T, RiDTinv, P);
  float P1[3] = {P[0], P[3], P[6]}, P2[3] = {P[1], P[4], P[7]}, P3[3] = {P[2], P[5], P[8]};
  float d3[3] = {d[2], d[5], d[8]};
  for (int i = 0; i < 3; ++i) {
    f2[i] = -vol * (iD11 * P1[i] + iD12 * P2[i]);
    f3[i] = -vol * iD22 * P2[i];
    f1[i] = -(f2[i] + f3[i]);
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) stress_out[r * 3 + c] = vol * (P3[r] * d3[c]);
}

/* ------------------------------------------------------------------ */
/* isotropic constitutive models, mpm_utils.py:8-84                    */
/* ------------------------------------------------------------------ */
static void stress_FCR(const float *F, const float *U, const float *V, float J, float mu,
                       float lam, float *S) { /* :8-15 */
  float Vt[9], R[9], Ft[9], D[9], T[9];
  m_T(V, Vt); m_mul(U, Vt, R); m_T(F, Ft);
  for (int i = 0; i < 9; ++i) D[i] = F[i] - R[i];
  m_mul(D, Ft, T);
  for (int i = 0; i < 9; ++i) S[i] = 2.0f * mu * T[i];
  float p = lam * J * (J - 1.0f);
  S[0] += p; S[4] += p; S[8] += p;
}
static void usvf(const float *U, const float *tau, const float *V, const float *F, float *S) {
  float D[9], Vt[9], Ft[9], T[9];
  m_diag(tau[0], tau[1], tau[2], D); m_T(V, Vt); m_T(F, Ft);
  m_mul(U, D, T); m_mul(T, Vt, T); m_mul(T, Ft, S);
}
static void stress_StVK(const float *F, const float *U, const float *V, const float *sig_in,
                        float mu, float lam, float *S) { /* :50-66 */
  float sig[3] = {fmaxf(sig_in[0], 0.01f), fmaxf(sig_in[1], 0.01f), fmaxf(sig_in[2], 0.01f)};
  float eps[3] = {logf(sig[0]), logf(sig[1]), logf(sig[2])};
  float sum = logf(sig[0]) + logf(sig[1]) + logf(sig[2]);
  float tau[3];
  for (int i = 0; i < 3; ++i) tau[i] = 2.0f * mu * eps[i] + lam * sum * 1.0f;
  usvf(U, tau, V, F, S);
}
static void stress_drucker_prager(const float *F, const float *U, const float *V,
                                  const float *sig, float mu, float lam, float *S) { /* :69-84 */
  float sum = logf(sig[0]) + logf(sig[1]) + logf(sig[2]);
  float c[3];
  for (int i = 0; i < 3; ++i)
    c[i] = 2.0f * mu * logf(sig[i]) * (1.0f / sig[i]) + lam * sum * (1.0f / sig[i]);
  usvf(U, c, V, F, S);
}

/* return mappings, mpm_utils.py:212-399 */
static void udv(const float *U, const float *e, const float *V, float *O) {
  float D[9], Vt[9], T[9];
  m_diag(e[0], e[1], e[2], D); m_T(V, Vt); m_mul(U, D, T); m_mul(T, Vt, O);
}
static void von_mises_return_mapping(orc_sim *s, const float *Ft, int p, int damage, float *Fo) {
  float U[9], V[9], so[3];
  orc_svd3(Ft, U, so, V);
  float sig[3] = {fmaxf(so[0], 0.01f), fmaxf(so[1], 0.01f), fmaxf(so[2], 0.01f)};
  float eps[3] = {logf(sig[0]), logf(sig[1]), logf(sig[2])};
  float temp = (eps[0] + eps[1] + eps[2]) / 3.0f;
  float mu = s->mu[p], lam = s->lam[p];
  float tau[3];
  for (int i = 0; i < 3; ++i) tau[i] = 2.0f * mu * eps[i] + lam * (eps[0] + eps[1] + eps[2]) * 1.0f;
  float sum_tau = tau[0] + tau[1] + tau[2];
  float cond[3] = {tau[0] - sum_tau / 3.0f, tau[1] - sum_tau / 3.0f, tau[2] - sum_tau / 3.0f};
  if (v_len(cond) > s->yield_stress[p]) {
    if (damage && s->yield_stress[p] <= 0.f) { memcpy(Fo, Ft, 9 * sizeof(float)); return; } /* :281 */
    float eh[3] = {eps[0] - temp, eps[1] - temp, eps[2] - temp};
    float ehn = v_len(eh) + 1e-6f;
    float dg = ehn - s->yield_stress[p] / (2.0f * mu);
    for (int i = 0; i < 3; ++i) eps[i] = eps[i] - (dg / ehn) * eh[i];
    if (damage) { /* :287-292 */
      float t[3] = {(dg / ehn) * eh[0], (dg / ehn) * eh[1], (dg / ehn) * eh[2]};
      s->yield_stress[p] = s->yield_stress[p] - s->softening * v_len(t);
      if (s->yield_stress[p] <= 0.f) { s->mu[p] = 0.0f; s->lam[p] = 0.0f; }
    }
    float e[3] = {expf(eps[0]), expf(eps[1]), expf(eps[2])};
    udv(U, e, V, Fo);
    if (s->hardening == 1.0f) /* :249-252, :305-308 (reads the possibly-zeroed mu) */
      s->yield_stress[p] = s->yield_stress[p] + 2.0f * s->mu[p] * s->xi * dg;
  } else {
    memcpy(Fo, Ft, 9 * sizeof(float));
  }
}
static void viscoplasticity_return_mapping(orc_sim *s, const float *Ft, int p, float dt, float *Fo) {
  float U[9], V[9], so[3]; /* :315-359 */
  orc_svd3(Ft, U, so, V);
  float sig[3] = {fmaxf(so[0], 0.01f), fmaxf(so[1], 0.01f), fmaxf(so[2], 0.01f)};
  float b[3] = {sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2]};
  float eps[3] = {logf(sig[0]), logf(sig[1]), logf(sig[2])};
  float tr = eps[0] + eps[1] + eps[2];
  float eh[3] = {eps[0] - tr / 3.0f, eps[1] - tr / 3.0f, eps[2] - tr / 3.0f};
  float mu = s->mu[p];
  float st[3] = {2.0f * mu * eh[0], 2.0f * mu * eh[1], 2.0f * mu * eh[2]};
  float stn = v_len(st);
  float y = stn - sqrtf(2.0f / 3.0f) * s->yield_stress[p];
  if (y > 0.f) {
    float mu_hat = mu * (b[0] + b[1] + b[2]) / 3.0f;
    float snn = stn - y / (1.0f + s->plastic_viscosity / (2.0f * mu_hat * dt));
    float en[3];
    for (int i = 0; i < 3; ++i) en[i] = 1.0f / (2.0f * mu) * ((snn / stn) * st[i]) + tr / 3.0f;
    float e[3] = {expf(en[0]), expf(en[1]), expf(en[2])};
    udv(U, e, V, Fo);
  } else {
    memcpy(Fo, Ft, 9 * sizeof(float));
  }
}
static void sand_return_mapping(orc_sim *s, const float *Ft, int p, float *Fo) { /* :362-399 */
  float U[9], V[9], sig[3];
  orc_svd3(Ft, U, sig, V);
  float eps[3];
  for (int i = 0; i < 3; ++i) eps[i] = logf(fmaxf(fabsf(sig[i]), 1e-14f));
  float tr = eps[0] + eps[1] + eps[2];
  float eh[3] = {eps[0] - tr / 3.0f, eps[1] - tr / 3.0f, eps[2] - tr / 3.0f};
  float ehn = v_len(eh);
  float mu = s->mu[p], lam = s->lam[p];
  float dg = ehn + (3.0f * lam + 2.0f * mu) / (2.0f * mu) * tr * s->alpha;
  if (dg <= 0.f) { memcpy(Fo, Ft, 9 * sizeof(float)); }
  if (dg > 0.f && tr > 0.f) { float Vt[9]; m_T(V, Vt); m_mul(U, Vt, Fo); }
  if (dg > 0.f && tr <= 0.f) {
    float e[3];
    for (int i = 0; i < 3; ++i) e[i] = expf(eps[i] - eh[i] * (dg / ehn));
    udv(U, e, V, Fo);
  }
  if (!(dg <= 0.f) && !(dg > 0.f)) memcpy(Fo, Ft, 9 * sizeof(float)); /* NaN: F_elastic undefined upstream */
}

/* ------------------------------------------------------------------ */
/* stencil, mpm_utils.py:499-514 (identical in every transfer kernel)  */
/* w[axis*3+node], dw[axis*3+node]  (wp.mat33(v0,v1,v2) takes columns) */
/* ------------------------------------------------------------------ */
void orc_stencil(const float *x, float inv_dx, int *base, float *w, float *dw) {
  for (int a = 0; a < 3; ++a) {
    float gp = x[a] * inv_dx;
    base[a] = (int)(gp - 0.5f); /* wp.int: truncation toward zero */
    float fx = gp - (float)base[a];
    float wa = 1.5f - fx, wb = fx - 1.0f, wc = fx - 0.5f;
    w[a * 3 + 0] = wa * wa * 0.5f;
    w[a * 3 + 1] = 0.0f - wb * wb + 0.75f;
    w[a * 3 + 2] = wc * wc * 0.5f;
    if (dw) {
      dw[a * 3 + 0] = fx - 1.5f;
      dw[a * 3 + 1] = -2.0f * (fx - 1.0f);
      dw[a * 3 + 2] = fx - 0.5f;
    }
  }
}
static void stencil_fx(const float *x, float inv_dx, float *fx) {
  for (int a = 0; a < 3; ++a) {
    float gp = x[a] * inv_dx;
    int b = (int)(gp - 0.5f);
    fx[a] = gp - (float)b;
  }
}
static inline size_t gidx(const orc_sim *s, int ix, int iy, int iz) {
  size_t G = (size_t)s->n_grid;
  return ((size_t)ix * G + (size_t)iy) * G + (size_t)iz;
}
static inline void add3(float *dst, const float *a) {
  ORC_ATOMIC
  dst[0] += a[0];
  ORC_ATOMIC
  dst[1] += a[1];
  ORC_ATOMIC
  dst[2] += a[2];
}

/* The nodes a grid-wide pass visits: the whole grid, or the active box of this substep (mpm_oracle.h, ACTIVE BOX). */
typedef struct { int lo[3], hi[3]; } orc_box;
static orc_box box_of(const orc_sim *s) {
  orc_box b;
  for (int a = 0; a < 3; ++a) {
    b.lo[a] = s->box_mode ? s->box_lo[a] : 0;
    b.hi[a] = s->box_mode ? s->box_hi[a] : s->n_grid - 1;
  }
  return b;
}
static inline int in_box(const orc_box *b, int ix, int iy, int iz) {
  return ix >= b->lo[0] && ix <= b->hi[0] && iy >= b->lo[1] && iy <= b->hi[1] && iz >= b->lo[2] && iz <= b->hi[2];
}
/* for every node g of the box, rows along z: ORC_BOX_FOR(s, bx) { ... g ... } ORC_BOX_END */
#define ORC_BOX_FOR(s, bx)                                                              \
  ORC_PARALLEL_FOR                                                                      \
  for (int ix_ = (bx).lo[0]; ix_ <= (bx).hi[0]; ++ix_)                                  \
    for (int iy_ = (bx).lo[1]; iy_ <= (bx).hi[1]; ++iy_) {                              \
      size_t g0_ = gidx((s), ix_, iy_, (bx).lo[2]), g1_ = gidx((s), ix_, iy_, (bx).hi[2]); \
      for (size_t g = g0_; g <= g1_; ++g)
#define ORC_BOX_END }
static void box_clear(const orc_sim *s, const orc_box *b, float *a, int comps) {
  if (b->lo[0] == 0 && b->lo[1] == 0 && b->lo[2] == 0 && b->hi[0] == s->n_grid - 1 && b->hi[1] == s->n_grid - 1 && b->hi[2] == s->n_grid - 1) {
    memset(a, 0, (size_t)s->n_grid * s->n_grid * s->n_grid * comps * sizeof(float));
    return;
  }
  size_t len = (size_t)(b->hi[2] - b->lo[2] + 1) * comps * sizeof(float);
  ORC_PARALLEL_FOR
  for (int ix = b->lo[0]; ix <= b->hi[0]; ++ix)
    for (int iy = b->lo[1]; iy <= b->hi[1]; ++iy) memset(a + gidx(s, ix, iy, b->lo[2]) * comps, 0, len);
}
/* bounding box of every particle's stencil (base .. base + 2 per axis, orc_stencil), clamped to the grid */
static void box_from_particles(orc_sim *s) {
  int lo0 = s->n_grid, lo1 = s->n_grid, lo2 = s->n_grid, hi0 = -1, hi1 = -1, hi2 = -1;
  for (int p = 0; p < s->n_particles; ++p) {
    int b0 = (int)(s->x[p * 3] * s->inv_dx - 0.5f), b1 = (int)(s->x[p * 3 + 1] * s->inv_dx - 0.5f), b2 = (int)(s->x[p * 3 + 2] * s->inv_dx - 0.5f);
    if (b0 < lo0) lo0 = b0;
    if (b1 < lo1) lo1 = b1;
    if (b2 < lo2) lo2 = b2;
    if (b0 > hi0) hi0 = b0;
    if (b1 > hi1) hi1 = b1;
    if (b2 > hi2) hi2 = b2;
  }
  int lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0 + 2, hi1 + 2, hi2 + 2};
  for (int a = 0; a < 3; ++a) {
    s->box_lo[a] = lo[a] < 0 ? 0 : lo[a];
    s->box_hi[a] = hi[a] > s->n_grid - 1 ? s->n_grid - 1 : hi[a];
    if (s->n_particles == 0) { s->box_lo[a] = 0; s->box_hi[a] = 0; }
  }
}

/* zero_grid, mpm_utils.py:411-417 */
void orc_zero_grid(orc_sim *s) {
  orc_box bx = box_of(s);
  box_clear(s, &bx, s->grid_m, 1);
  box_clear(s, &bx, s->grid_v_in, 3);
  box_clear(s, &bx, s->grid_v_out, 3);
}

/* Test hooks (tests/test_hip_math_on_host.py): when set, the per-particle constitutive update is taken from the caller
   instead of the restatement below -- the test compiles the PRODUCT's device math header for the host and runs it inside
   this oracle's substep, so that the header's arithmetic can be checked against the reference-produced fixtures without
   a GPU.  NULL (the default) = the oracle's own restatement; nothing but that test sets them. */
void (*orc_hook_element)(const float *d, const float *R_inv, float vol, float mu, float lam, float gamma, float kappa,
                         float friction_coeff, float *new_d, float *stress, float *f1, float *f2, float *f3) = 0;
void (*orc_hook_traditional)(const float *F_trial, int material, float alpha, float hardening, float xi,
                             float plastic_viscosity, float softening, float dt, float *mu, float *lam,
                             float *yield_stress, float *F, float *stress) = 0;

/* compute_stress_from_F_trial, mpm_utils.py:1017-1105; launch dim n_nv (mpm_solver.py:327-332) */
void orc_compute_stress_from_F_trial(orc_sim *s, float dt) {
  int n_nv = s->n_particles - s->n_vertices;
  ORC_PARALLEL_FOR
  for (int p = 0; p < n_nv; ++p) {
    if (s->selection[p] != 0) continue;
    float stress[9] = {0};
    if (p < s->n_elements) { /* particle_elements[p] == 1 */
      float nd[9];
      float f1[3], f2[3], f3[3];
      if (orc_hook_element) {
        orc_hook_element(&s->d[p * 9], &s->R_inv[p * 3], s->vol[p], s->mu[p], s->lam[p], s->gamma[p], s->kappa[p],
                         s->friction_coeff, nd, stress, f1, f2, f3);
        memcpy(&s->d[p * 9], nd, sizeof nd);
      } else {
        orc_anisotropy_return_mapping(&s->d[p * 9], s->gamma[p], s->kappa[p], s->friction_coeff, nd);
        memcpy(&s->d[p * 9], nd, sizeof nd);
        orc_kirchhoff_anisotropy(&s->R_inv[p * 3], &s->d[p * 9], s->vol[p], s->mu[p], s->lam[p],
                                 s->gamma[p], s->kappa[p], stress, f1, f2, f3);
      }
      int v1 = (int)s->faces[p * 3], v2 = (int)s->faces[p * 3 + 1], v3 = (int)s->faces[p * 3 + 2];
      add3(&s->vertex_force[v1 * 3], f1); /* :173-175 */
      add3(&s->vertex_force[v2 * 3], f2);
      add3(&s->vertex_force[v3 * 3], f3);
    } else if (orc_hook_traditional) {
      float F[9];
      orc_hook_traditional(&s->F_trial[p * 9], s->material, s->alpha, s->hardening, s->xi, s->plastic_viscosity,
                           s->softening, dt, &s->mu[p], &s->lam[p], &s->yield_stress[p], F, stress);
      memcpy(&s->F[p * 9], F, sizeof F);
    } else { /* particle_traditional[p] == 1 */
      float F[9];
      const float *Ft = &s->F_trial[p * 9];
      switch (s->material) { /* :1053-1070 */
        case 1: von_mises_return_mapping(s, Ft, p, 0, F); break;
        case 2: sand_return_mapping(s, Ft, p, F); break;
        case 3: viscoplasticity_return_mapping(s, Ft, p, dt, F); break;
        case 5: von_mises_return_mapping(s, Ft, p, 1, F); break;
        default: memcpy(F, Ft, sizeof F);
      }
      memcpy(&s->F[p * 9], F, sizeof F);
      float J = m_det(F), U[9], V[9], sig[3];
      orc_svd3(F, U, sig, V);
      int m = s->material;
      if (m == 0 || m == 5) stress_FCR(F, U, V, J, s->mu[p], s->lam[p], stress);
      if (m == 1) stress_StVK(F, U, V, sig, s->mu[p], s->lam[p], stress);
      if (m == 2) stress_drucker_prager(F, U, V, sig, s->mu[p], s->lam[p], stress);
      if (m == 3) stress_StVK(F, U, V, sig, s->mu[p], s->lam[p], stress);
      float St[9];
      m_T(stress, St);
      for (int i = 0; i < 9; ++i) stress[i] = (stress[i] + St[i]) / 2.0f; /* :1103 */
    }
    memcpy(&s->stress[p * 9], stress, sizeof stress);
  }
}

/* p2g_apic_with_stress, mpm_utils.py:484-557; launch dim n_particles, offset n_nv */
void orc_p2g(orc_sim *s, float dt) {
  int n_nv = s->n_particles - s->n_vertices;
  float rpic = s->rpic_damping;
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->n_particles; ++p) {
    if (s->selection[p] != 0) continue;
    int is_vert = p >= n_nv, is_trad = (p >= s->n_elements && p < n_nv);
    float vforce[3] = {0, 0, 0}, stress[9] = {0};
    if (is_vert) memcpy(vforce, &s->vertex_force[(p - n_nv) * 3], sizeof vforce);
    else if (is_trad) for (int i = 0; i < 9; ++i) stress[i] = s->vol[p] * s->stress[p * 9 + i];
    else memcpy(stress, &s->stress[p * 9], sizeof stress);
    int base[3];
    float w[9], dw[9], fx[3];
    orc_stencil(&s->x[p * 3], s->inv_dx, base, w, dw);
    stencil_fx(&s->x[p * 3], s->inv_dx, fx);
    const float *Cp = &s->C[p * 9];
    float C[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) /* :530-532 */
        C[r * 3 + c] = (1.0f - rpic) * Cp[r * 3 + c] + rpic / 2.0f * (Cp[r * 3 + c] - Cp[c * 3 + r]);
    if (rpic < -0.001f) memset(C, 0, sizeof C);
    float mass = s->mass[p];
    const float *vp = &s->v[p * 3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
          float dpos[3] = {((float)i - fx[0]) * s->dx, ((float)j - fx[1]) * s->dx,
                           ((float)k - fx[2]) * s->dx};
          float weight = w[0 + i] * w[3 + j] * w[6 + k];
          float dweight[3] = {dw[0 + i] * w[3 + j] * w[6 + k] * s->inv_dx,
                              w[0 + i] * dw[3 + j] * w[6 + k] * s->inv_dx,
                              w[0 + i] * w[3 + j] * dw[6 + k] * s->inv_dx};
          float force[3];
          if (is_vert) { for (int a = 0; a < 3; ++a) force[a] = weight * vforce[a]; }
          else { float t[3]; m_vec(stress, dweight, t); for (int a = 0; a < 3; ++a) force[a] = -t[a]; }
          float Cd[3];
          m_vec(C, dpos, Cd);
          float add[3];
          for (int a = 0; a < 3; ++a) add[a] = weight * mass * (vp[a] + Cd[a]) + dt * force[a];
          size_t g = gidx(s, base[0] + i, base[1] + j, base[2] + k);
          add3(&s->grid_v_in[g * 3], add);
          float wm = weight * mass;
          ORC_ATOMIC
          s->grid_m[g] += wm;
        }
  }
}

/* grid_normalization_and_gravity, mpm_utils.py:561-572 */
void orc_grid_normalization_and_gravity(orc_sim *s, float dt) {
  orc_box bx = box_of(s);
  ORC_BOX_FOR(s, bx) {
    if (s->grid_m[g] > 1e-15f) {
      float inv = 1.0f / s->grid_m[g];
      for (int a = 0; a < 3; ++a)
        s->grid_v_out[g * 3 + a] = s->grid_v_in[g * 3 + a] * inv + dt * s->g[a];
    }
  }
  ORC_BOX_END
}

/* add_damping_via_grid, mpm_utils.py:1162-1174 */
void orc_add_damping_via_grid(orc_sim *s, float scale) {
  orc_box bx = box_of(s);
  ORC_BOX_FOR(s, bx) {
    for (int a = 0; a < 3; ++a) s->grid_v_out[g * 3 + a] -= (1.0f - scale) * s->grid_v_out[g * 3 + a];
  }
  ORC_BOX_END
}

static int in_splat_bounds(const orc_sim *s, const int *b) { /* mpm_solver.py:692,730,767,858 */
  int G = s->n_grid;
  return b[0] >= 0 && b[0] < G - 3 && b[1] >= 0 && b[1] < G - 3 && b[2] >= 0 && b[2] < G - 3;
}

/* mesh collider k: zero_grid, compute_mesh, normalize_grid, collide; mpm_solver.py:819-917 */
void orc_mesh_collide(orc_sim *s, int k) {
  orc_mesh_collider *mc = &s->mesh_colliders[k];
  orc_box bx = box_of(s);
  box_clear(s, &bx, mc->weight, 1);
  box_clear(s, &bx, mc->v_in, 3);
  box_clear(s, &bx, mc->v_out, 3);
  box_clear(s, &bx, mc->normal, 3);
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->num_mesh_f; ++p) { /* compute_mesh :829-880 */
    int i0 = s->mesh_indices[3 * p], i1 = s->mesh_indices[3 * p + 1], i2 = s->mesh_indices[3 * p + 2];
    const float *p0 = &s->mesh_points[i0 * 3], *p1 = &s->mesh_points[i1 * 3], *p2 = &s->mesh_points[i2 * 3];
    const float *u0 = &s->mesh_velocities[i0 * 3], *u1 = &s->mesh_velocities[i1 * 3], *u2 = &s->mesh_velocities[i2 * 3];
    float fp[3], fv[3], e1[3], e2[3], cr[3], fn[3];
    for (int a = 0; a < 3; ++a) {
      fp[a] = (p0[a] + p1[a] + p2[a]) / 3.0f;
      fv[a] = (u0[a] + u1[a] + u2[a]) / 3.0f;
      e1[a] = p1[a] - p0[a];
      e2[a] = p2[a] - p0[a];
    }
    cr[0] = e1[1] * e2[2] - e1[2] * e2[1];
    cr[1] = e1[2] * e2[0] - e1[0] * e2[2];
    cr[2] = e1[0] * e2[1] - e1[1] * e2[0];
    v_normalize(cr, fn); /* wp.mesh_eval_face_normal */
    int base[3];
    float w[9];
    orc_stencil(fp, s->inv_dx, base, w, NULL);
    if (!in_splat_bounds(s, base)) continue;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int kk = 0; kk < 3; ++kk) {
          if (!in_box(&bx, base[0] + i, base[1] + j, base[2] + kk)) continue; /* (never taken with box_mode = 0) */
          float weight = w[i] * w[3 + j] * w[6 + kk];
          size_t g = gidx(s, base[0] + i, base[1] + j, base[2] + kk);
          float a3[3] = {weight * fv[0], weight * fv[1], weight * fv[2]};
          float b3[3] = {weight * fn[0], weight * fn[1], weight * fn[2]};
          add3(&mc->v_in[g * 3], a3);
          add3(&mc->normal[g * 3], b3);
          ORC_ATOMIC
          mc->weight[g] += weight;
        }
  }
  ORC_BOX_FOR(s, bx) { /* normalize_grid :882-890 */
    if (mc->weight[g] > 1e-15f) {
      float inv_w = 1.0f / mc->weight[g];
      for (int a = 0; a < 3; ++a) mc->v_out[g * 3 + a] = mc->v_in[g * 3 + a] * inv_w;
    }
  }
  ORC_BOX_END
  ORC_BOX_FOR(s, bx) { /* collide :892-917 */
    float *v = &s->grid_v_out[g * 3];
    if (mc->weight[g] > 1e-15f) {
      float vrel[3], nn[3], vproj[3], vfric[3];
      for (int a = 0; a < 3; ++a) vrel[a] = v[a] - mc->v_out[g * 3 + a];
      v_normalize(&mc->normal[g * 3], nn);
      float nc = vrel[0] * nn[0] + vrel[1] * nn[1] + vrel[2] * nn[2];
      float mn = fminf(nc, 0.0f);
      for (int a = 0; a < 3; ++a) vproj[a] = vrel[a] - mn * nn[a];
      float lp = v_len(vproj);
      if (nc < 0.0f && lp > 1e-20f) {
        float sc = fmaxf(0.0f, lp + nc * mc->friction), nv[3];
        v_normalize(vproj, nv);
        for (int a = 0; a < 3; ++a) vfric[a] = sc * nv[a];
      } else {
        for (int a = 0; a < 3; ++a) vfric[a] = vproj[a];
      }
      for (int a = 0; a < 3; ++a) v[a] = vfric[a] + mc->v_out[g * 3 + a];
    }
  }
  ORC_BOX_END
}

/* splat of a prescribed velocity at particle q, mpm_solver.py:677-788 */
static void mover_splat(orc_sim *s, orc_mover *mv, int q, const float *vel) {
  int base[3];
  float w[9];
  orc_stencil(&s->x[q * 3], s->inv_dx, base, w, NULL);
  if (!in_splat_bounds(s, base)) return;
  orc_box bx = box_of(s);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        if (!in_box(&bx, base[0] + i, base[1] + j, base[2] + k)) continue; /* (a particle's stencil lies inside the box by construction) */
        float weight = w[i] * w[3 + j] * w[6 + k];
        size_t g = gidx(s, base[0] + i, base[1] + j, base[2] + k);
        float a3[3] = {weight * vel[0], weight * vel[1], weight * vel[2]};
        add3(&mv->velocity[g * 3], a3);
        ORC_ATOMIC
        mv->weight[g] += weight;
      }
}

/* particle mover k, mpm_solver.py:428-481 + 669-799 */
void orc_particle_move(orc_sim *s, int k, const float *joint_t_v, int n_joint_t,
                       const float *joint_v_v, const float *joint_f_v) {
  orc_mover *mv = &s->movers[k];
  orc_box bx = box_of(s);
  int n_nv = s->n_particles - s->n_vertices;
  box_clear(s, &bx, mv->weight, 1);
  box_clear(s, &bx, mv->velocity, 3);
  if (joint_t_v) { /* :437-449, offset n_particles - n_vertices - joint_num */
    int off = n_nv - n_joint_t;
    ORC_PARALLEL_FOR
    for (int p = 0; p < n_joint_t; ++p) mover_splat(s, mv, p + off, &joint_t_v[p * 3]);
  }
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->num_joint_v; ++p) mover_splat(s, mv, p + n_nv, &joint_v_v[p * 3]);
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->num_joint_f; ++p) mover_splat(s, mv, p, &joint_f_v[p * 3]);
  ORC_BOX_FOR(s, bx) { /* normalize_grid :790-799: overwrite */
    if (mv->weight[g] > 1e-15f) {
      float inv_w = 1.0f / mv->weight[g];
      for (int a = 0; a < 3; ++a) s->grid_v_out[g * 3 + a] = mv->velocity[g * 3 + a] * inv_w;
    }
  }
  ORC_BOX_END
}

/* grid BC k, mpm_solver.py:600-655 / 950-981 / 993-1050 / 1341-1352 */
void orc_apply_bc(orc_sim *s, int k, float dt) {
  orc_bc *bc = &s->bc[k];
  int G = s->n_grid;
  float time = (float)s->time;
  orc_box bx = box_of(s);
  ORC_PARALLEL_FOR
  for (int gx = bx.lo[0]; gx <= bx.hi[0]; ++gx)
    for (int gy = bx.lo[1]; gy <= bx.hi[1]; ++gy)
      for (int gz = bx.lo[2]; gz <= bx.hi[2]; ++gz) {
        float *v = &s->grid_v_out[gidx(s, gx, gy, gz) * 3];
        if (bc->type == ORC_BC_SURFACE) {
          if (time >= bc->start_time && time < bc->end_time) {
            float off[3] = {(float)gx * s->dx - bc->point[0], (float)gy * s->dx - bc->point[1],
                            (float)gz * s->dx - bc->point[2]};
            float dotp = off[0] * bc->normal[0] + off[1] * bc->normal[1] + off[2] * bc->normal[2];
            if (dotp < 0.0f) {
              if (bc->surface_type == 0) { v[0] = v[1] = v[2] = 0.0f; }
              else if (bc->surface_type == 11) { /* :623-635 */
                if ((float)gz * s->dx < 0.4f || (float)gz * s->dx > 0.53f) { v[0] = v[1] = v[2] = 0.0f; }
                else { v[0] = v[0] * 0.3f; v[1] = 0.0f * 0.3f; v[2] = v[2] * 0.3f; }
              } else { v[0] = v[1] = v[2] = 0.0f; } /* quirk Q1: :653-655 writes zero anyway */
            }
          }
        } else if (bc->type == ORC_BC_CUBOID) {
          if (time >= bc->start_time && time < bc->end_time) {
            float off[3] = {(float)gx * s->dx - bc->point[0], (float)gy * s->dx - bc->point[1],
                            (float)gz * s->dx - bc->point[2]};
            if (fabsf(off[0]) < bc->size[0] && fabsf(off[1]) < bc->size[1] && fabsf(off[2]) < bc->size[2]) {
              v[0] = bc->velocity[0]; v[1] = bc->velocity[1]; v[2] = bc->velocity[2];
            }
          } else if (bc->reset == 1) {
            if (time < bc->end_time + 15.0f * dt) { v[0] = v[1] = v[2] = 0.0f; }
          }
        } else if (bc->type == ORC_BC_BBOX) {
          int padding = 3;
          if (time >= bc->start_time && time < bc->end_time) {
            if (gx < padding && v[0] < 0.f) v[0] = 0.0f;
            if (gx >= G - padding && v[0] > 0.f) v[0] = 0.0f;
            if (gy < padding && v[1] < 0.f) v[1] = 0.0f;
            if (gy >= G - padding && v[1] > 0.f) v[1] = 0.0f;
            if (gz < padding && v[2] < 0.f) v[2] = 0.0f;
            if (gz >= G - padding && v[2] > 0.f) v[2] = 0.0f;
          }
        } else if (bc->type == ORC_BC_GRIDMASK) {
          if (bc->mask[gidx(s, gx, gy, gz)] >= 1) { v[0] = v[1] = v[2] = 0.0f; }
        }
      }
  if (bc->type == ORC_BC_CUBOID) { /* host-side modify, :975-981 */
    if (time >= bc->start_time && time < bc->end_time)
      for (int a = 0; a < 3; ++a) bc->point[a] = bc->point[a] + dt * bc->velocity[a];
  }
}

/* shared gather of g2p_v / g2p_e, mpm_utils.py:726-763 / 798-836 */
static void g2p_gather(const orc_sim *s, const float *xp, float *new_v, float *new_C, float *new_F) {
  int base[3];
  float w[9], dw[9], fx[3];
  orc_stencil(xp, s->inv_dx, base, w, dw);
  stencil_fx(xp, s->inv_dx, fx);
  memset(new_v, 0, 3 * sizeof(float));
  memset(new_C, 0, 9 * sizeof(float));
  memset(new_F, 0, 9 * sizeof(float));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        float dpos[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
        float weight = w[i] * w[3 + j] * w[6 + k];
        const float *gv = &s->grid_v_out[gidx(s, base[0] + i, base[1] + j, base[2] + k) * 3];
        float dweight[3] = {dw[i] * w[3 + j] * w[6 + k] * s->inv_dx, w[i] * dw[3 + j] * w[6 + k] * s->inv_dx,
                            w[i] * w[3 + j] * dw[6 + k] * s->inv_dx};
        float cs = weight * s->inv_dx * 4.0f;
        for (int r = 0; r < 3; ++r) {
          new_v[r] = new_v[r] + gv[r] * weight;
          for (int c = 0; c < 3; ++c) {
            new_C[r * 3 + c] = new_C[r * 3 + c] + (gv[r] * dpos[c]) * cs;
            new_F[r * 3 + c] = new_F[r * 3 + c] + gv[r] * dweight[c];
          }
        }
      }
}

/* g2p_v, mpm_utils.py:716-786; launch dim n_particles-n_elements, offset n_elements */
void orc_g2p_v(orc_sim *s, float dt) {
  int n_nv = s->n_particles - s->n_vertices;
  ORC_PARALLEL_FOR
  for (int q = s->n_elements; q < s->n_particles; ++q) {
    if (s->selection[q] != 0) continue;
    float nv[3], nC[9], nF[9];
    g2p_gather(s, &s->x[q * 3], nv, nC, nF);
    memcpy(&s->v[q * 3], nv, sizeof nv);
    float dx = 1.0f / s->inv_dx, a_min = dx * 2.0f, a_max = s->grid_lim - dx * 2.0f;
    for (int a = 0; a < 3; ++a) s->x[q * 3 + a] = f_clamp(s->x[q * 3 + a] + dt * nv[a], a_min, a_max);
    memcpy(&s->C[q * 9], nC, sizeof nC);
    if (q < n_nv) { /* particle_traditional: F_trial = (I + dt*gradv) F, :783-786 */
      float M[9];
      for (int i = 0; i < 9; ++i) M[i] = nF[i] * dt;
      M[0] += 1.0f; M[4] += 1.0f; M[8] += 1.0f;
      m_mul(M, &s->F[q * 9], &s->F_trial[q * 9]);
    }
  }
}

/* g2p_e, mpm_utils.py:788-857; launch dim n_elements, offset n_nv */
void orc_g2p_e(orc_sim *s, float dt) {
  int n_nv = s->n_particles - s->n_vertices;
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->n_elements; ++p) {
    if (s->selection[p] != 0) continue;
    float nv[3], nC[9], nF[9];
    g2p_gather(s, &s->x[p * 3], nv, nC, nF);
    int v1 = (int)s->faces[p * 3] + n_nv, v2 = (int)s->faces[p * 3 + 1] + n_nv, v3 = (int)s->faces[p * 3 + 2] + n_nv;
    for (int a = 0; a < 3; ++a) {
      s->v[p * 3 + a] = (s->v[v1 * 3 + a] + s->v[v2 * 3 + a] + s->v[v3 * 3 + a]) / 3.0f;
      s->x[p * 3 + a] = (s->x[v1 * 3 + a] + s->x[v2 * 3 + a] + s->x[v3 * 3 + a]) / 3.0f;
    }
    memcpy(&s->C[p * 9], nC, sizeof nC);
    float d1[3], d2[3];
    for (int a = 0; a < 3; ++a) {
      d1[a] = s->x[v2 * 3 + a] - s->x[v1 * 3 + a];
      d2[a] = s->x[v3 * 3 + a] - s->x[v1 * 3 + a];
    }
    float *d = &s->d[p * 9];
    float d3[3] = {d[2], d[5], d[8]}, M[9], d3t[3];
    for (int i = 0; i < 9; ++i) M[i] = nF[i] * dt;
    M[0] += 1.0f; M[4] += 1.0f; M[8] += 1.0f;
    m_vec(M, d3, d3t);
    float nd[9] = {d1[0], d2[0], d3t[0], d1[1], d2[1], d3t[1], d1[2], d2[2], d3t[2]};
    memcpy(d, nd, sizeof nd);
  }
}

/* pre-p2g particle operations, mpm_solver.py:260-279 */
static void apply_pre(orc_sim *s, const orc_pre *op, float dt) {
  float time = (float)s->time;
  if (!(time >= op->start_time && time < op->end_time)) return;
  ORC_PARALLEL_FOR
  for (int p = 0; p < s->n_particles; ++p) {
    float *v = &s->v[p * 3];
    switch (op->type) {
      case ORC_PRE_IMPULSE:
        if (op->mask[p] == 1)
          for (int a = 0; a < 3; ++a) v[a] = v[a] + (op->force[a] / s->mass[p]) * dt;
        break;
      case ORC_PRE_IMPULSE_MASK:
        if (op->mask[p] >= 1)
          for (int a = 0; a < 3; ++a) v[a] = v[a] + op->force[a] * dt;
        break;
      case ORC_PRE_VEL_SET:
        if (op->mask[p] == 1)
          for (int a = 0; a < 3; ++a) v[a] = op->velocity[a];
        break;
      case ORC_PRE_VEL_ROTATE:
        if (op->mask[p] == 1) { /* :1225-1255 */
          float off[3], h[3];
          for (int a = 0; a < 3; ++a) off[a] = s->x[p * 3 + a] - op->point[a];
          float dn = off[0] * op->normal[0] + off[1] * op->normal[1] + off[2] * op->normal[2];
          for (int a = 0; a < 3; ++a) h[a] = off[a] - dn * op->normal[a];
          float hd = v_len(h);
          float cosine = (off[0] * op->axis1[0] + off[1] * op->axis1[1] + off[2] * op->axis1[2]) / hd;
          float theta = acosf(fminf(fmaxf(cosine, -1.0f), 1.0f)); /* wp.acos clamps its argument */
          if (!(off[0] * op->axis2[0] + off[1] * op->axis2[1] + off[2] * op->axis2[2] > 0.f)) theta = -theta;
          float a1 = -hd * sinf(theta) * op->rotation_scale;
          float a2 = hd * cosf(theta) * op->rotation_scale;
          for (int a = 0; a < 3; ++a)
            v[a] = a1 * op->axis1[a] + a2 * op->axis2[a] + op->translation_scale * op->normal[a];
        }
        break;
    }
  }
}

/* the two loops over pre-p2g particle operations, mpm_solver.py:260-279 */
void orc_pre_p2g(orc_sim *s, float dt) {
  for (int k = 0; k < s->n_pre; ++k)                                  /* :260 impulses */
    if (s->pre[k].type == ORC_PRE_IMPULSE || s->pre[k].type == ORC_PRE_IMPULSE_MASK) apply_pre(s, &s->pre[k], dt);
  for (int k = 0; k < s->n_pre; ++k)                                  /* :269 velocity modifiers */
    if (s->pre[k].type == ORC_PRE_VEL_SET || s->pre[k].type == ORC_PRE_VEL_ROTATE) apply_pre(s, &s->pre[k], dt);
}

/* MPMWARP.p2g2p, mpm_solver.py:229-536 */
void orc_p2g2p(orc_sim *s, double dt_host, const float *mesh_x, const float *mesh_v,
               const float *joint_t_v, int n_joint_t, const float *joint_v_v,
               const float *joint_f_v) {
  const float dt = (float)dt_host; /* kernels take fp32 dt; MPMWARP.time advances by the Python double (:536) */
#ifdef ORC_OMP
  if (s->n_threads > 0) omp_set_num_threads(s->n_threads);
#endif
  if (s->box_mode) box_from_particles(s);                             /* (mpm_oracle.h, ACTIVE BOX: not part of the algorithm) */
  orc_zero_grid(s);                                                   /* :244 */
  memset(s->vertex_force, 0, (size_t)s->n_vertices * 3 * sizeof(float)); /* :251 */
  orc_pre_p2g(s, dt);                                                 /* :260-279 */
  if (mesh_x) memcpy(s->mesh_points, mesh_x, (size_t)s->num_mesh_v * 3 * sizeof(float));     /* :285-299 */
  if (mesh_v) memcpy(s->mesh_velocities, mesh_v, (size_t)s->num_mesh_v * 3 * sizeof(float)); /* :301-315 */
  orc_compute_stress_from_F_trial(s, dt);                             /* :327 */
  orc_p2g(s, dt);                                                     /* :355 */
  orc_grid_normalization_and_gravity(s, dt);                          /* :366 */
  if (s->grid_v_damping_scale < 1.0f) orc_add_damping_via_grid(s, s->grid_v_damping_scale); /* :373 */
  for (int k = 0; k < s->n_mesh_colliders; ++k) orc_mesh_collide(s, k); /* :385-419 */
  if (joint_v_v && joint_f_v)                                         /* :421 */
    for (int k = 0; k < s->n_movers; ++k) orc_particle_move(s, k, joint_t_v, n_joint_t, joint_v_v, joint_f_v);
  for (int k = 0; k < s->n_bc; ++k) orc_apply_bc(s, k, dt);           /* :487-501 */
  orc_g2p_v(s, dt);                                                   /* :518 */
  orc_g2p_e(s, dt);                                                   /* :529 */
  s->time = s->time + dt_host;                                        /* :536 */
}

void orc_p2g2p_n(orc_sim *s, double dt_host, int n, const float *mesh_x, const float *mesh_v,
                 const float *joint_t_v, int n_joint_t, const float *joint_v_v,
                 const float *joint_f_v) {
  float *cur = NULL;
  size_t nm = (size_t)s->num_mesh_v * 3;
  if (mesh_x && mesh_v) cur = (float *)malloc(nm * sizeof(float));
  for (int k = 0; k < n; ++k) {
    const float *mx = mesh_x;
    if (cur) { /* train_material_params.py:623: mesh_x + substep_size*substep_local*mesh_v */
      float f = (float)dt_host * (float)k;
      for (size_t i = 0; i < nm; ++i) cur[i] = mesh_x[i] + f * mesh_v[i];
      mx = cur;
    }
    orc_p2g2p(s, dt_host, mx, mesh_v, joint_t_v, n_joint_t, joint_v_v, joint_f_v);
  }
  free(cur);
}

int orc_sizeof_sim(void) { return (int)sizeof(orc_sim); }
