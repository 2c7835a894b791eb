/*
 * oracle/mpm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Serial fp32 CPU restatement of the reference MPM substep
 * (KAISTChangmin/MPMAvatar, warp_mpm/{mpm_solver,mpm_utils,mpm_data_structure}.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product path (mpmavatar_amd/) never links, imports or calls it.
 *
 * PINNING: the reference path is NVIDIA-Warp DSL (warp-lang 0.10.1), which is not
 * installed in this image and cannot be (no network), and the reference ships no
 * tests, fixtures or golden vectors for this path.  Its kernels are plain Python
 * bodies, though: tests/golden/make_golden_ref.py imports the .py files of /root/reference/warp_mpm
 * UNCHANGED over a NumPy stand-in of the `warp` module (tests/golden/warp_standin: serial
 * `for tid`, one fp32 rounding per operation) and records single-substep traces and
 * multi-substep sequences; this restatement reproduces them kernel by kernel and over
 * whole sequences (tests/test_ref_golden.py, CPU suite) -- since round 2.  What that
 * does NOT pin: wp.svd3 / wp.qr3 are the stand-in's own (two conventions each, agreeing
 * to 2e-6 before a fixture is written): convention-pinned, not Warp-pinned.  Beside the
 * fixtures: analytic known-answer tests and an independent float64 NumPy twin
 * (oracle/twin.py), tests/test_oracle_*.py.  Every function cites the reference lines
 * it follows.
 *
 * Layout = the reference's Warp layout: AoS, vec3 = 3 floats, mat33 = 9 floats
 * row-major, grids C-order [x][y][z].  Particle index classes:
 *   [0,n_elements) elements | [n_elements,n_nv) traditional | [n_nv,n_particles) vertices
 * with n_nv = n_particles - n_vertices.
 */
#ifndef MPM_ORACLE_H
#define MPM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_BC 16
#define ORC_MAX_MESH_COLLIDERS 4
#define ORC_MAX_MOVERS 2
#define ORC_MAX_PRE 64

/* grid boundary conditions, reference mpm_solver.py:564-1053,1330-1355 */
enum { ORC_BC_SURFACE = 0, ORC_BC_CUBOID = 1, ORC_BC_BBOX = 2, ORC_BC_GRIDMASK = 3 };

typedef struct {
  int32_t type;
  int32_t surface_type; /* 0 sticky, 1 slip, 11 cut, 2 other (mpm_solver.py:587-594) */
  int32_t reset;
  int32_t pad_;
  float point[3];
  float normal[3];
  float size[3];
  float velocity[3];
  float friction;
  float start_time, end_time;
  const int32_t *mask; /* GRIDMASK: [G^3] */
} orc_bc;

/* pre-p2g particle operations, reference mpm_solver.py:1058-1417 */
enum { ORC_PRE_IMPULSE = 0,        /* v += force/mass*dt where mask==1   (:1093-1104) */
       ORC_PRE_IMPULSE_MASK = 1,   /* v += force*dt     where mask>=1   (:1399-1415) */
       ORC_PRE_VEL_SET = 2,        /* v  = velocity     where mask==1   (:1138-1149,:1315-1326) */
       ORC_PRE_VEL_ROTATE = 3 };   /* cylinder rotation where mask==1   (:1214-1255) */

typedef struct {
  int32_t type;
  int32_t pad_;
  float start_time, end_time;
  float force[3];
  float velocity[3];
  float point[3];
  float normal[3];
  float axis1[3], axis2[3];
  float rotation_scale, translation_scale;
  const int32_t *mask; /* [n_particles] */
} orc_pre;

typedef struct {
  float friction;
  float *weight;  /* [G^3]   */
  float *v_in;    /* [G^3*3] */
  float *v_out;   /* [G^3*3] */
  float *normal;  /* [G^3*3] */
} orc_mesh_collider;

typedef struct {
  float *weight;   /* [G^3]   */
  float *velocity; /* [G^3*3] */
} orc_mover;

typedef struct {
  /* sizes */
  int32_t n_particles, n_elements, n_vertices;
  int32_t n_grid;
  float grid_lim, dx, inv_dx;

  /* particle state (mpm_data_structure.py:13-49) */
  float *x, *v, *C;
  float *F, *F_trial, *stress;
  float *d, *R_inv, *faces, *vertex_force;
  float *vol, *mass, *density;
  int32_t *selection;

  /* grid */
  float *grid_m, *grid_v_in, *grid_v_out;

  /* model (mpm_data_structure.py:610-645) */
  float *E, *nu, *mu, *lam, *gamma, *kappa, *yield_stress;
  int32_t material;
  float friction_coeff, alpha;
  float g[3];
  float hardening, xi, plastic_viscosity, softening;
  float rpic_damping, grid_v_damping_scale;

  /* body mesh (mpm_solver.py:45-51) */
  int32_t num_mesh_v, num_mesh_f;
  float *mesh_points, *mesh_velocities;
  const int32_t *mesh_indices;

  int32_t n_mesh_colliders;
  orc_mesh_collider mesh_colliders[ORC_MAX_MESH_COLLIDERS];
  int32_t n_movers;
  orc_mover movers[ORC_MAX_MOVERS];
  int32_t num_joint_v, num_joint_f;

  int32_t n_bc;
  orc_bc bc[ORC_MAX_BC];
  int32_t n_pre;
  orc_pre pre[ORC_MAX_PRE];

  double time; /* MPMWARP.time, python float (mpm_solver.py:28,536) */
  int32_t n_threads; /* 1 = serial oracle; >1 = OpenMP baseline (atomics) */
  /* ACTIVE BOX (off by default; a speed option of the long ensemble runs of tests/test_gpu_fullsize.py, not part of the restated
     algorithm): with box_mode != 0 orc_p2g2p takes the bounding box of ALL particles' 3x3x3 stencils at the head of the substep and
     every grid-wide pass of that substep (the clears, normalisation, damping, collider, mover, BCs) visits the nodes of the box only;
     a collider face's splat skips nodes outside it.  Nothing outside the box is read by p2g / g2p of that substep, and every node
     inside it sees exactly the dense substep's operations, in the dense order per node: particle results are bit-identical to
     box_mode = 0 in the serial build (tests/test_oracle_box.py).  Grid arrays hold stale values OUTSIDE the box afterwards. */
  int32_t box_mode;
  int32_t box_lo[3], box_hi[3];
} orc_sim;

/* individual kernels (exposed so tests can pin them one at a time) */
void orc_zero_grid(orc_sim *s);
void orc_pre_p2g(orc_sim *s, float dt);
void orc_compute_stress_from_F_trial(orc_sim *s, float dt);
void orc_p2g(orc_sim *s, float dt);
void orc_grid_normalization_and_gravity(orc_sim *s, float dt);
void orc_add_damping_via_grid(orc_sim *s, float scale);
void orc_mesh_collide(orc_sim *s, int k);
void orc_particle_move(orc_sim *s, int k, const float *joint_t_v, int n_joint_t,
                       const float *joint_v_v, const float *joint_f_v);
void orc_apply_bc(orc_sim *s, int k, float dt);
void orc_g2p_v(orc_sim *s, float dt);
void orc_g2p_e(orc_sim *s, float dt);

/* one substep = MPMWARP.p2g2p (mpm_solver.py:229-536); NULL pointers = argument None */
void orc_p2g2p(orc_sim *s, double dt, const float *mesh_x, const float *mesh_v,
               const float *joint_t_v, int n_joint_t, const float *joint_v_v,
               const float *joint_f_v);

/* n substeps with the caller's mesh advection mesh_x + k*dt*mesh_v
 * (train_material_params.py:622-626); mesh_x may be NULL */
void orc_p2g2p_n(orc_sim *s, double dt, int n, const float *mesh_x, const float *mesh_v,
                 const float *joint_t_v, int n_joint_t, const float *joint_v_v,
                 const float *joint_f_v);

/* small pure functions exposed for known-answer tests */
void orc_svd3(const float *A, float *U, float *sig, float *V);
void orc_qr_signfixed(const float *d, float *Q, float *R);
void orc_anisotropy_return_mapping(const float *d, float gamma, float kappa,
                                   float friction_coeff, float *new_d);
void orc_kirchhoff_anisotropy(const float *R_inv, const float *d, float vol, float mu,
                              float lam, float gamma, float kappa, float *stress_out,
                              float *f1, float *f2, float *f3);
void orc_stencil(const float *x, float inv_dx, int *base, float *w, float *dw);
int orc_sizeof_sim(void);

#ifdef __cplusplus
}
#endif
#endif
