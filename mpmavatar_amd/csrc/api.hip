// api.hip -- the extern "C" boundary of libmpmhip.so (declared in include/mpmhip.h).
// Host-side orchestration only: argument validation, context lifetime, dispatch to the fast or
// baseline back end.  There is deliberately no CPU fallback: without a HIP device
// mpmhip_create() fails with MPMHIP_ERR_NO_DEVICE.
#include <cstring>

#include "bc.hpp"
#include "ctx.hpp"

using namespace mpm;

namespace {
std::string g_create_error;

// set_vec3_to_vec3(mesh.points, mesh_x) / (mesh.velocities, mesh_v), mpm_solver.py:285-315, as ONE device-side
// kernel that also applies the caller's advection mesh_x + (k*dt)*mesh_v (train_material_params.py:623; mul then
// add, not fused, like the torch expression)
__global__ void k_mesh_store(float *pts, float *vel, const float *x, const float *v, float f, size_t n) {
#pragma clang fp contract(off)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (x) {
    float p = x[i];
    if (v && f != 0.0f) { float a = f * v[i]; p = p + a; }
    pts[i] = p;
  }
  if (v) vel[i] = v[i];
}

bool fast_mode(const mpmhip_ctx *c) { return c->cfg.mode == MPMHIP_MODE_FAST; }
}  // namespace

extern "C" {

int mpmhip_version(void) { return MPMHIP_VERSION; }

int mpmhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *mpmhip_last_error(const mpmhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mpmhip_create(const mpmhip_config *cfg, mpmhip_ctx **out) {
  if (!cfg || !out) { g_create_error = "mpmhip_create: null argument"; return MPMHIP_ERR_INVALID; }
  *out = nullptr;
  if (cfg->n_particles < 0 || cfg->n_elements < 0 || cfg->n_vertices < 0 ||
      cfg->n_elements + cfg->n_vertices > cfg->n_particles || cfg->n_grid < 8 || !(cfg->grid_lim > 0.f) ||
      cfg->num_joint_v > cfg->n_vertices || cfg->num_joint_f > cfg->n_elements) {
    g_create_error = "mpmhip_create: inconsistent sizes";
    return MPMHIP_ERR_INVALID;
  }
  if (cfg->mode != MPMHIP_MODE_FAST && cfg->mode != MPMHIP_MODE_BASELINE) {
    g_create_error = "mpmhip_create: unknown mode";
    return MPMHIP_ERR_INVALID;
  }
  if (cfg->p2g_tile < MPMHIP_P2G_TILE_AUTO || cfg->p2g_tile > MPMHIP_P2G_TILE_F64 || cfg->reserved_ != 0) {
    g_create_error = "mpmhip_create: p2g_tile must be MPMHIP_P2G_TILE_AUTO / _FIXED / _F64 (and reserved_ 0)";
    return MPMHIP_ERR_INVALID;
  }
  int ndev = mpmhip_device_count();
  if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    g_create_error = "mpmhip_create: no HIP device " + std::to_string(cfg->device) + " (visible devices: " +
                     std::to_string(ndev) + "); libmpmhip has no CPU fallback";
    return MPMHIP_ERR_NO_DEVICE;
  }
  mpmhip_ctx *c = new mpmhip_ctx();
  c->cfg = *cfg;
  c->n_nv = cfg->n_particles - cfg->n_vertices;
  c->n_trad = c->n_nv - cfg->n_elements;
  // MPMModelStruct.init_other_params, mpm_data_structure.py:692-697 (python doubles -> fp32 fields)
  c->dx = (float)((double)cfg->grid_lim / (double)cfg->n_grid);
  c->inv_dx = (float)((double)cfg->n_grid / (double)cfg->grid_lim);
  c->sc.material = 0;
  c->sc.softening = 0.1f;
  c->sc.grid_v_damping_scale = 1.1f;
  auto bail = [&](int code, const std::string &m) {
    g_create_error = m;
    mpmhip_destroy(c);
    return code;
  };
  if (hipSetDevice(cfg->device) != hipSuccess) return bail(MPMHIP_ERR_HIP, "hipSetDevice failed");
  if (!cfg->own_stream) {
    c->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
      return bail(MPMHIP_ERR_HIP, "hipStreamCreate failed");
    c->own_stream = true;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->kev0) != hipSuccess ||
      hipEventCreate(&c->kev1) != hipSuccess)
    return bail(MPMHIP_ERR_HIP, "hipEventCreate failed");
  int rc = fast_mode(c) ? fast_init(c) : baseline_init(c);
  if (rc) return bail(rc, c->err);
  *out = c;
  return MPMHIP_OK;
}

void mpmhip_destroy(mpmhip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  if (c->fast) fast_destroy(c);
  for (float *p : {c->grid_m, c->grid_v_in, c->grid_v_out, c->mesh_points, c->mesh_vel})
    if (p) (void)hipFree(p);
  if (c->mesh_idx) (void)hipFree(c->mesh_idx);
  if (!fast_mode(c)) {
    for (auto &mc : c->colliders)
      for (float *p : {mc.weight, mc.v_in, mc.normal})
        if (p) (void)hipFree(p);
    for (auto &mv : c->movers)
      for (float *p : {mv.weight, mv.velocity})
        if (p) (void)hipFree(p);
  }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->kev0) (void)hipEventDestroy(c->kev0);
  if (c->kev1) (void)hipEventDestroy(c->kev1);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

#define CHECK_CTX(c)                       \
  if (!(c)) return MPMHIP_ERR_INVALID;     \
  (void)hipSetDevice((c)->cfg.device)

int mpmhip_bind_state(mpmhip_ctx *c, const mpmhip_state_ptrs *p) {
  CHECK_CTX(c);
  if (!p) return fail(c, MPMHIP_ERR_INVALID, "bind_state: null");
  const int n_p = c->cfg.n_particles, n_e = c->cfg.n_elements, n_v = c->cfg.n_vertices;
  bool ok = (n_p == 0) || (p->particle_x && p->particle_v && p->particle_C && p->particle_vol && p->particle_mass &&
                           p->particle_selection);
  if (c->n_nv > 0) ok = ok && p->particle_F && p->particle_F_trial && p->particle_stress;
  if (n_e > 0) ok = ok && p->particle_d && p->particle_R_inv && p->faces;
  if (n_v > 0) ok = ok && p->vertex_force;
  if (!ok) return fail(c, MPMHIP_ERR_INVALID, "bind_state: a required array pointer is null");
  // pending results of the internal state belong to the *old* arrays: flush them first
  if (fast_mode(c) && c->st_bound && c->internal_dirty) {
    int rc = fast_pull(c);
    if (rc) return rc;
  }
  c->st = *p;
  c->st_bound = true;
  c->caller_dirty = true;
  c->internal_dirty = false;
  return MPMHIP_OK;
}

int mpmhip_bind_model(mpmhip_ctx *c, const mpmhip_model_ptrs *p) {
  CHECK_CTX(c);
  if (!p || (c->cfg.n_particles > 0 && !(p->mu && p->lam && p->gamma && p->kappa && p->yield_stress)))
    return fail(c, MPMHIP_ERR_INVALID, "bind_model: a required array pointer is null");
  if (fast_mode(c) && c->st_bound && c->internal_dirty) {
    int rc = fast_pull(c);
    if (rc) return rc;
  }
  c->md = *p;
  c->md_bound = true;
  c->caller_dirty = true;
  return MPMHIP_OK;
}

int mpmhip_set_model_scalars(mpmhip_ctx *c, const mpmhip_model_scalars *s) {
  CHECK_CTX(c);
  if (!s) return fail(c, MPMHIP_ERR_INVALID, "set_model_scalars: null");
  if (s->material < 0 || s->material > 7) return fail(c, MPMHIP_ERR_INVALID, "Undefined material type");
  c->sc = *s;
  return MPMHIP_OK;
}

int mpmhip_push_state(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (fast_mode(c) && c->st_bound && c->internal_dirty) {
    // the caller is about to overwrite (part of) the arrays: make them current first so that
    // fields it does not touch keep the simulated values
    int rc = fast_pull(c);
    if (rc) return rc;
  }
  c->caller_dirty = true;
  return MPMHIP_OK;
}

int mpmhip_pull_state(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!c->st_bound) return fail(c, MPMHIP_ERR_STATE, "pull_state: state not bound");
  if (fast_mode(c) && c->internal_dirty) return fast_pull(c);
  return MPMHIP_OK;
}

int mpmhip_set_body_mesh(mpmhip_ctx *c, int32_t n_verts, int32_t n_faces, const float *verts, const int32_t *faces) {
  CHECK_CTX(c);
  if (n_verts <= 0 || n_faces <= 0 || !verts || !faces) return fail(c, MPMHIP_ERR_INVALID, "set_body_mesh: bad mesh");
  for (int i = 0; i < 3 * n_faces; ++i)
    if (faces[i] < 0 || faces[i] >= n_verts) return fail(c, MPMHIP_ERR_INVALID, "set_body_mesh: face index out of range");
  if (c->mesh_points) return fail(c, MPMHIP_ERR_INVALID, "set_body_mesh: mesh already set");
  size_t nb = (size_t)n_verts * 3 * sizeof(float);
  MPM_HIP_CHECK(c, hipMalloc(&c->mesh_points, nb));
  MPM_HIP_CHECK(c, hipMalloc(&c->mesh_vel, nb));
  MPM_HIP_CHECK(c, hipMalloc(&c->mesh_idx, (size_t)n_faces * 3 * sizeof(int32_t)));
  MPM_HIP_CHECK(c, hipMemcpy(c->mesh_points, verts, nb, hipMemcpyHostToDevice));
  MPM_HIP_CHECK(c, hipMemset(c->mesh_vel, 0, nb));
  MPM_HIP_CHECK(c, hipMemcpy(c->mesh_idx, faces, (size_t)n_faces * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
  c->num_mesh_v = n_verts;
  c->num_mesh_f = n_faces;
  return MPMHIP_OK;
}

int mpmhip_add_mesh_collider(mpmhip_ctx *c, float friction) {
  CHECK_CTX(c);
  if (!c->mesh_points) return fail(c, MPMHIP_ERR_STATE, "add_mesh_collider: no body mesh set");
  if (c->colliders.size() >= 4) return fail(c, MPMHIP_ERR_LIMIT, "add_mesh_collider: at most 4 mesh colliders");
  MeshCollider mc{};
  mc.friction = friction;
  int rc = fast_mode(c) ? fast_add_collider_storage(c, mc) : baseline_add_collider_storage(c, mc);
  if (rc) return rc;
  c->colliders.push_back(mc);
  return MPMHIP_OK;
}

int mpmhip_add_particle_mover(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (c->movers.size() >= 2) return fail(c, MPMHIP_ERR_LIMIT, "add_particle_mover: at most 2 movers");
  Mover mv{};
  int rc = fast_mode(c) ? fast_add_mover_storage(c, mv) : baseline_add_mover_storage(c, mv);
  if (rc) return rc;
  c->movers.push_back(mv);
  return MPMHIP_OK;
}

static int push_bc(mpmhip_ctx *c, const BC &bc) {
  if ((int)c->bcs.size() >= MAX_BC) return fail(c, MPMHIP_ERR_LIMIT, "too many grid boundary conditions");
  c->bcs.push_back(bc);
  return MPMHIP_OK;
}

int mpmhip_add_surface_collider(mpmhip_ctx *c, const float point[3], const float normal[3], int32_t surface_type,
                                float friction, float start_time, float end_time) {
  CHECK_CTX(c);
  if (surface_type == 0 && friction != 0.0f) return fail(c, MPMHIP_ERR_INVALID, "friction must be 0 on sticky surfaces.");
  BC bc{};
  bc.type = BC_SURFACE;
  bc.surface_type = surface_type;
  bc.friction = friction;
  bc.start_time = start_time;
  bc.end_time = end_time;
  memcpy(bc.point, point, sizeof bc.point);
  memcpy(bc.normal, normal, sizeof bc.normal);
  return push_bc(c, bc);
}

int mpmhip_add_velocity_cuboid(mpmhip_ctx *c, const float point[3], const float size[3], const float velocity[3],
                               float start_time, float end_time, int32_t reset) {
  CHECK_CTX(c);
  BC bc{};
  bc.type = BC_CUBOID;
  bc.reset = reset;
  bc.start_time = start_time;
  bc.end_time = end_time;
  memcpy(bc.point, point, sizeof bc.point);
  memcpy(bc.size, size, sizeof bc.size);
  memcpy(bc.velocity, velocity, sizeof bc.velocity);
  return push_bc(c, bc);
}

int mpmhip_add_bounding_box(mpmhip_ctx *c, float start_time, float end_time) {
  CHECK_CTX(c);
  BC bc{};
  bc.type = BC_BBOX;
  bc.start_time = start_time;
  bc.end_time = end_time;
  return push_bc(c, bc);
}

int mpmhip_add_grid_mask(mpmhip_ctx *c, const int32_t *mask) {
  CHECK_CTX(c);
  if (!mask) return fail(c, MPMHIP_ERR_INVALID, "add_grid_mask: null mask");
  BC bc{};
  bc.type = BC_GRIDMASK;
  bc.mask = mask;
  return push_bc(c, bc);
}

int mpmhip_select_box(mpmhip_ctx *c, const float point[3], const float size[3], int32_t *mask) {
  CHECK_CTX(c);
  if (!c->st_bound || !mask) return fail(c, MPMHIP_ERR_STATE, "select_box: state not bound / null mask");
  int rc = mpmhip_pull_state(c);
  if (rc) return rc;
  return launch_select_box(c, c->st.particle_x, point, size, mask);
}

int mpmhip_select_cylinder(mpmhip_ctx *c, const float point[3], const float normal[3], float half_height, float radius,
                           int32_t *mask) {
  CHECK_CTX(c);
  if (!c->st_bound || !mask) return fail(c, MPMHIP_ERR_STATE, "select_cylinder: state not bound / null mask");
  int rc = mpmhip_pull_state(c);
  if (rc) return rc;
  return launch_select_cylinder(c, c->st.particle_x, point, normal, half_height, radius, mask);
}

static int push_pre(mpmhip_ctx *c, const PreOp &op) {
  if (!op.mask) return fail(c, MPMHIP_ERR_INVALID, "particle operation: null mask");
  if (c->pre.size() >= 256) return fail(c, MPMHIP_ERR_LIMIT, "too many particle operations");
  c->pre.push_back(op);
  return MPMHIP_OK;
}

int mpmhip_add_impulse(mpmhip_ctx *c, const float force[3], const int32_t *mask, int32_t per_mass, float start_time,
                       float end_time) {
  CHECK_CTX(c);
  PreOp op{};
  op.type = per_mass ? PRE_IMPULSE : PRE_IMPULSE_MASK;
  op.start_time = start_time;
  op.end_time = end_time;
  op.mask = mask;
  memcpy(op.force, force, sizeof op.force);
  return push_pre(c, op);
}

int mpmhip_add_velocity_set(mpmhip_ctx *c, const float velocity[3], const int32_t *mask, float start_time,
                            float end_time) {
  CHECK_CTX(c);
  PreOp op{};
  op.type = PRE_VEL_SET;
  op.start_time = start_time;
  op.end_time = end_time;
  op.mask = mask;
  memcpy(op.velocity, velocity, sizeof op.velocity);
  return push_pre(c, op);
}

int mpmhip_add_velocity_rotation(mpmhip_ctx *c, const float point[3], const float normal[3], const float axis1[3],
                                 const float axis2[3], float rotation_scale, float translation_scale,
                                 const int32_t *mask, float start_time, float end_time) {
  CHECK_CTX(c);
  PreOp op{};
  op.type = PRE_VEL_ROTATE;
  op.start_time = start_time;
  op.end_time = end_time;
  op.mask = mask;
  op.rotation_scale = rotation_scale;
  op.translation_scale = translation_scale;
  memcpy(op.point, point, sizeof op.point);
  memcpy(op.normal, normal, sizeof op.normal);
  memcpy(op.axis1, axis1, sizeof op.axis1);
  memcpy(op.axis2, axis2, sizeof op.axis2);
  return push_pre(c, op);
}

static int step_checked(mpmhip_ctx *c, const StepArgs &a) {
  if (!c->st_bound || !c->md_bound) return fail(c, MPMHIP_ERR_STATE, "step: state/model not bound");
  if (a.n_joint_t < 0 || a.n_joint_t > c->n_trad) return fail(c, MPMHIP_ERR_INVALID, "step: n_joint_t out of range");
  if ((a.mesh_x || a.mesh_v) && !c->mesh_points) return fail(c, MPMHIP_ERR_STATE, "step: mesh_x/mesh_v given but no body mesh");
  // the backends read the body mesh through cur_pts + cur_f * cur_vel (device-to-device; the reference's
  // .cpu().numpy() round trip per substep, mpm_solver.py:282-302, is not reproduced)
  c->cur_pts = a.mesh_x ? a.mesh_x : c->mesh_points;
  c->cur_vel = a.mesh_v ? a.mesh_v : c->mesh_vel;
  c->cur_f = (a.mesh_x && a.mesh_v) ? a.mesh_f : 0.0f;
  int rc = fast_mode(c) ? fast_step(c, a) : baseline_step(c, a);
  if (rc) return rc;
  if (a.mesh_store && (a.mesh_x || a.mesh_v)) {
    ScopedPhase ph(c, "update_mesh_positions");
    size_t nm = (size_t)c->num_mesh_v * 3;
    hipLaunchKernelGGL(k_mesh_store, (unsigned)((nm + 255) / 256), 256, 0, c->stream, c->mesh_points, c->mesh_vel,
                       a.mesh_x, a.mesh_v, c->cur_f, nm);
  }
  c->time = c->time + c->time_inc(a.dt);  // mpm_solver.py:536
  c->substeps += 1;
  return MPMHIP_OK;
}

int mpmhip_step(mpmhip_ctx *c, float dt, const float *mesh_x, const float *mesh_v, const float *joint_traditional_v,
                int32_t n_joint_t, const float *joint_verts_v, const float *joint_faces_v) {
  CHECK_CTX(c);
  StepArgs a{dt, mesh_x, mesh_v, 0.0f, true, joint_traditional_v, joint_traditional_v ? n_joint_t : 0, joint_verts_v, joint_faces_v};
  return step_checked(c, a);
}

int mpmhip_steps(mpmhip_ctx *c, float dt, int32_t n, const float *mesh_x, const float *mesh_v,
                 const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                 const float *joint_faces_v) {
  CHECK_CTX(c);
  if (n < 0) return fail(c, MPMHIP_ERR_INVALID, "steps: n < 0");
  const bool fast = fast_mode(c) && c->fast && c->st_bound && c->md_bound && (!(mesh_x || mesh_v) || c->mesh_points);
  if (fast) {   // a body that does not move during this call is splatted once, not n times (fast_body_at_rest_begin, fast.hip)
    c->cur_vel = mesh_v ? mesh_v : c->mesh_vel;
    int rc = fast_body_at_rest_begin(c, n);
    if (rc) return rc;
  }
  struct AtRestEnd { mpmhip_ctx *c; bool on; ~AtRestEnd() { if (on) fast_body_at_rest_end(c); } } at_rest_end{c, fast};
  for (int k = 0; k < n; ++k) {
    // mesh_x + substep_size*substep_local*mesh_v, train_material_params.py:623, evaluated inside the kernels
    StepArgs a{dt, mesh_x, mesh_v, (float)((double)dt * (double)k), k == n - 1, joint_traditional_v,
               joint_traditional_v ? n_joint_t : 0, joint_verts_v, joint_faces_v, k < n - 1};
    int rc = step_checked(c, a);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

extern "C++" {
namespace mpm {
void mesh_store_launch(mpmhip_ctx *c, const StepArgs &a) {
  size_t nm = (size_t)c->num_mesh_v * 3;
  hipLaunchKernelGGL(k_mesh_store, (unsigned)((nm + 255) / 256), 256, 0, c->stream, c->mesh_points, c->mesh_vel, a.mesh_x, a.mesh_v, c->cur_f, nm);
}
}  // namespace mpm
}  // extern "C++"

int mpmhip_dist_enable(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_enable(c);
}
int mpmhip_dist_set_ghost_mode(mpmhip_ctx *c, int32_t ghosts_gather) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_set_ghost_mode(c, ghosts_gather);
}
int mpmhip_dist_set_mass_span(mpmhip_ctx *c, float min_mass, float max_mass) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  if (min_mass > 0.0f && !(max_mass >= min_mass)) return fail(c, MPMHIP_ERR_INVALID, "dist_set_mass_span: max_mass < min_mass");
  return fast_dist_set_mass_span(c, min_mass, max_mass);
}
int mpmhip_dist_ghost_pack(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_ghosts(c, 1);
}
int mpmhip_dist_ghost_unpack(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_ghosts(c, 0);
}
int mpmhip_dist_num_blocks(const mpmhip_ctx *c) { return (c && c->fast) ? fast_dist_num_blocks(c) : 0; }
int mpmhip_dist_drift_flag(mpmhip_ctx *c, int32_t *flag) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  if (!flag) return fail(c, MPMHIP_ERR_INVALID, "dist_drift_flag: null output");
  return fast_dist_drift_flag(c, flag);
}
int mpmhip_dist_rebin(mpmhip_ctx *c, uint8_t *active_map) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_rebin(c, active_map);
}
int mpmhip_dist_set_peers(mpmhip_ctx *c, int32_t n_peers, const mpmhip_dist_peer *peers) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_dist_set_peers(c, n_peers, peers);
}
int mpmhip_dist_step_begin(mpmhip_ctx *c, float dt, const float *mesh_x, const float *mesh_v, float mesh_advect,
                           const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                           const float *joint_faces_v) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  if (!c->st_bound || !c->md_bound) return fail(c, MPMHIP_ERR_STATE, "step: state/model not bound");
  if ((mesh_x || mesh_v) && !c->mesh_points) return fail(c, MPMHIP_ERR_STATE, "step: mesh_x/mesh_v given but no body mesh");
  StepArgs a{dt, mesh_x, mesh_v, mesh_advect, true, joint_traditional_v, joint_traditional_v ? n_joint_t : 0, joint_verts_v, joint_faces_v};
  c->cur_pts = a.mesh_x ? a.mesh_x : c->mesh_points;
  c->cur_vel = a.mesh_v ? a.mesh_v : c->mesh_vel;
  c->cur_f = (a.mesh_x && a.mesh_v) ? a.mesh_f : 0.0f;
  c->fast_dt = dt;
  int rc = fast_dist_phase(c, 0, a);
  if (rc) return rc;
  if (a.mesh_x || a.mesh_v) {  // keep the context's wp.Mesh copy current (used by the next collective re-sort)
    size_t nm = (size_t)c->num_mesh_v * 3;
    hipLaunchKernelGGL(k_mesh_store, (unsigned)((nm + 255) / 256), 256, 0, c->stream, c->mesh_points, c->mesh_vel,
                       a.mesh_x, a.mesh_v, c->cur_f, nm);
  }
  return MPMHIP_OK;
}
int mpmhip_dist_step_mid(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  StepArgs a{};
  return fast_dist_phase(c, 1, a);
}
int mpmhip_dist_step_end(mpmhip_ctx *c) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  StepArgs a{};
  int rc = fast_dist_phase(c, 2, a);
  if (rc) return rc;
  c->time = c->time + c->time_inc(c->fast_dt);
  c->substeps += 1;
  return MPMHIP_OK;
}

int mpmhip_rccl_unique_id(char id[128]) {
  if (!id) return MPMHIP_ERR_INVALID;
  return fast_rccl_unique_id(id, g_create_error);
}
int mpmhip_rccl_init(mpmhip_ctx *c, int32_t rank, int32_t world, const char id[128]) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !id) return fail(c, MPMHIP_ERR_INVALID, "rccl_init: fast mode context and an id are required");
  return fast_rccl_init(c, rank, world, id);
}
int mpmhip_rccl_set_ghosts(mpmhip_ctx *c, int32_t n_peers, const int32_t *peer_ranks, const int32_t *n_send_p,
                           const int32_t *const *send_p, const int32_t *n_recv_p, const int32_t *const *recv_p,
                           const int32_t *n_send_e, const int32_t *const *send_e, const int32_t *n_recv_e,
                           const int32_t *const *recv_e) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  return fast_rccl_set_ghosts(c, n_peers, peer_ranks, n_send_p, send_p, n_recv_p, recv_p, n_send_e, send_e, n_recv_e, recv_e);
}
int mpmhip_rccl_steps(mpmhip_ctx *c, float dt, int32_t n, int64_t step_index, int32_t rebin_interval, const float *mesh_x,
                      const float *mesh_v, const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                      const float *joint_faces_v) {
  CHECK_CTX(c);
  if (!fast_mode(c)) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  if (!c->st_bound || !c->md_bound) return fail(c, MPMHIP_ERR_STATE, "step: state/model not bound");
  if ((mesh_x || mesh_v) && !c->mesh_points) return fail(c, MPMHIP_ERR_STATE, "step: mesh_x/mesh_v given but no body mesh");
  c->fast_dt = dt;
  int rc = fast_rccl_steps(c, dt, n, step_index, rebin_interval, mesh_x, mesh_v, joint_traditional_v,
                           joint_traditional_v ? n_joint_t : 0, joint_verts_v, joint_faces_v);
  if (rc) return rc;
  if (n > 0 && (mesh_x || mesh_v)) {
    size_t nm = (size_t)c->num_mesh_v * 3;
    hipLaunchKernelGGL(k_mesh_store, (unsigned)((nm + 255) / 256), 256, 0, c->stream, c->mesh_points, c->mesh_vel, mesh_x,
                       mesh_v, c->cur_f, nm);
  }
  return MPMHIP_OK;
}

int mpmhip_synchronize(mpmhip_ctx *c) {
  CHECK_CTX(c);
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return MPMHIP_OK;
}

double mpmhip_get_time(const mpmhip_ctx *c) { return c ? c->time : 0.0; }
int mpmhip_set_host_dt(mpmhip_ctx *c, double dt) {
  CHECK_CTX(c);
  c->host_dt = dt;
  return MPMHIP_OK;
}
int mpmhip_set_time(mpmhip_ctx *c, double t) {
  if (!c) return MPMHIP_ERR_INVALID;
  c->time = t;
  return MPMHIP_OK;
}

int mpmhip_export_grid(mpmhip_ctx *c, float *grid_m, float *grid_v_in, float *grid_v_out) {
  CHECK_CTX(c);
  if (fast_mode(c)) return fast_export_grid(c, grid_m, grid_v_in, grid_v_out);
  size_t n = G3(c);
  if (grid_m) MPM_HIP_CHECK(c, hipMemcpyAsync(grid_m, c->grid_m, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  if (grid_v_in) MPM_HIP_CHECK(c, hipMemcpyAsync(grid_v_in, c->grid_v_in, 3 * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  if (grid_v_out) MPM_HIP_CHECK(c, hipMemcpyAsync(grid_v_out, c->grid_v_out, 3 * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return MPMHIP_OK;
}

int mpmhip_set_debug_flags(mpmhip_ctx *c, int32_t flags) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "set_debug_flags: fast mode only");
  return fast_set_debug_flags(c, flags);
}

int mpmhip_dist_halo_bytes(mpmhip_ctx *c, int64_t *out) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast || !out) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  *out = fast_dist_halo_bytes(c);
  return MPMHIP_OK;
}

int mpmhip_dist_halo_transport(mpmhip_ctx *c, int32_t *out) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast || !out) return fail(c, MPMHIP_ERR_INVALID, "dist: only the fast mode shards across GPUs");
  *out = fast_dist_halo_transport(c);
  return MPMHIP_OK;
}

int mpmhip_dist_fused_halo_steps(mpmhip_ctx *c, int64_t *out) {
  if (!c || !out) return MPMHIP_ERR_INVALID;
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "dist_fused_halo_steps: fast mode only");
  *out = fast_dist_fused_halo_steps(c);
  return MPMHIP_OK;
}

int mpmhip_debug_counter(mpmhip_ctx *c, int32_t index, int64_t *out) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "debug_counter: fast mode only");
  return fast_debug_counter(c, index, out);
}

int mpmhip_debug_wgtrace(mpmhip_ctx *c, int32_t kernel, uint64_t *out, int32_t max_wg) {
  if (!c) return MPMHIP_ERR_INVALID;
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "debug_wgtrace: fast mode only");
  return fast_debug_wgtrace(c, kernel, out, max_wg);
}

int mpmhip_debug_sort(mpmhip_ctx *c, const uint32_t *keys_in, int32_t n, int32_t bits, uint32_t *keys_out, int32_t *order_out) {
  CHECK_CTX(c);
  if (!fast_mode(c) || !c->fast) return fail(c, MPMHIP_ERR_INVALID, "debug_sort: fast mode only");
  if (n < 0 || bits < 1 || bits > 32 || (n > 0 && (!keys_in || !keys_out || !order_out)))
    return fail(c, MPMHIP_ERR_INVALID, "debug_sort: bad arguments");
  return fast_debug_sort(c, keys_in, n, bits, keys_out, order_out);
}

int mpmhip_get_stats(mpmhip_ctx *c, mpmhip_stats *out) {
  CHECK_CTX(c);
  if (!out) return fail(c, MPMHIP_ERR_INVALID, "get_stats: null");
  memset(out, 0, sizeof *out);
  out->substeps = c->substeps;
  if (fast_mode(c)) return fast_stats(c, out);
  size_t n = G3(c);
  int rc = count_nonzero(c, c->grid_m, n, 0.0f, &out->n_active_nodes);
  if (rc) return rc;
  if (!c->colliders.empty()) rc = count_nonzero(c, c->colliders[0].weight, n, 1e-15f, &out->n_collider_nodes);
  if (rc) return rc;
  if (!c->movers.empty()) rc = count_nonzero(c, c->movers[0].weight, n, 1e-15f, &out->n_mover_nodes);
  return rc;
}

int mpmhip_profile_enable(mpmhip_ctx *c, int32_t on) {
  if (!c) return MPMHIP_ERR_INVALID;
  c->profiling = on == 1;   // 1: the reference's phases, each its own launch + sync (time_profile keys)
  c->prof_fused = on == 2;  // 2: event pairs around the launches of the production loop (bench.py's roofline)
  return MPMHIP_OK;
}
int mpmhip_profile_count(const mpmhip_ctx *c) { return c ? (int)c->phases.size() : 0; }
int mpmhip_profile_get(const mpmhip_ctx *c, int32_t i, const char **name, double *total_ms, int64_t *samples) {
  if (!c || i < 0 || i >= (int)c->phases.size()) return MPMHIP_ERR_INVALID;
  if (name) *name = c->phases[i].name;
  if (total_ms) *total_ms = c->phases[i].total_ms;
  if (samples) *samples = c->phases[i].samples;
  return MPMHIP_OK;
}
int mpmhip_profile_get_kernel(const mpmhip_ctx *c, int32_t i, double *kernel_ms, int64_t *samples) {
  if (!c || i < 0 || i >= (int)c->phases.size()) return MPMHIP_ERR_INVALID;
  if (kernel_ms) *kernel_ms = c->phases[i].kernel_ms;
  if (samples) *samples = c->phases[i].kernel_samples;
  return MPMHIP_OK;
}
int mpmhip_profile_reset(mpmhip_ctx *c) {
  if (!c) return MPMHIP_ERR_INVALID;
  c->phases.clear();
  return MPMHIP_OK;
}

}  // extern "C"
