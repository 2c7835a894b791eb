// ctx.hpp -- solver context shared by the C-ABI layer and the two kernel back ends.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mpmhip.h"

namespace mpm {

enum BCType { BC_SURFACE = 0, BC_CUBOID = 1, BC_BBOX = 2, BC_GRIDMASK = 3 };
enum PreType { PRE_IMPULSE = 0, PRE_IMPULSE_MASK = 1, PRE_VEL_SET = 2, PRE_VEL_ROTATE = 3 };

struct BC {  // POD, passed to kernels by value
  int type, surface_type, reset, pad_;
  float point[3], normal[3], size[3], velocity[3];
  float friction, start_time, end_time;
  const int32_t *mask;
};
constexpr int MAX_BC = 8;
struct BCList {
  int n;
  BC bc[MAX_BC];
};

struct PreOp {
  int type;
  float start_time, end_time;
  float force[3], velocity[3], point[3], normal[3], axis1[3], axis2[3];
  float rotation_scale, translation_scale;
  const int32_t *mask;
};

struct MeshCollider {
  float friction;
  float *weight, *v_in, *normal;  // dense [G^3], [G^3*3] (baseline) or blocked (fast)
};
struct Mover {
  float *weight, *velocity;
};

struct StepArgs {
  float dt;
  const float *mesh_x, *mesh_v;  // body mesh for this substep: position = mesh_x + mesh_f * mesh_v
  float mesh_f;                  // fused advection factor k*dt of mpmhip_steps (0 for a plain step)
  bool mesh_store;               // write the advected mesh back into the context's wp.Mesh copy
  const float *joint_t_v;
  int n_joint_t;
  const float *joint_v_v, *joint_f_v;
  bool more = false;             // another substep of the same mpmhip_steps call follows at once (nothing reads particles in between)
};

struct Phase {
  const char *name;
  double total_ms = 0.0;
  int64_t samples = 0;
  double kernel_ms = 0.0;       // the launch's own start -> stop time (hot launches in prof_fused mode), without the bracket's cost
  int64_t kernel_samples = 0;
};

struct FastState;  // defined in fast.hip

}  // namespace mpm

struct mpmhip_ctx {
  mpmhip_config cfg{};
  int n_nv = 0, n_trad = 0;
  float dx = 0.f, inv_dx = 0.f;
  hipStream_t stream = nullptr;
  bool own_stream = false;

  mpmhip_state_ptrs st{};
  bool st_bound = false;
  mpmhip_model_ptrs md{};
  bool md_bound = false;
  mpmhip_model_scalars sc{};

  // dense reference-layout grid (baseline mode; fast mode allocates it lazily for export only)
  float *grid_m = nullptr, *grid_v_in = nullptr, *grid_v_out = nullptr;

  // body mesh (wp.Mesh): points/velocities updated every substep
  int num_mesh_v = 0, num_mesh_f = 0;
  float *mesh_points = nullptr, *mesh_vel = nullptr;
  int32_t *mesh_idx = nullptr;
  // body mesh as seen by the current substep (set by the API layer): points = cur_pts + cur_f * cur_vel
  const float *cur_pts = nullptr, *cur_vel = nullptr;
  float cur_f = 0.f;

  std::vector<mpm::MeshCollider> colliders;
  std::vector<mpm::Mover> movers;
  std::vector<mpm::BC> bcs;
  std::vector<mpm::PreOp> pre;

  double time = 0.0;
  double host_dt = 0.0;  // the caller's dt as the double it is in Python (mpmhip_set_host_dt); 0 = not given
  // MPMWARP.time advances by the Python float (mpm_solver.py:536: self.time + dt in double precision) while the kernels get
  // fp32 dt; with only the fp32 value known the increment is (double)(float)dt, which drifts from the reference's time by
  // 2.5e-8 relative and can move a time-windowed BC / particle operation by one substep
  double time_inc(float dt) const { return (host_dt != 0.0 && (float)host_dt == dt) ? host_dt : (double)dt; }
  float fast_dt = 0.f;  // dt of the substep in flight (dist phases)
  int64_t substeps = 0;
  std::string err;

  bool profiling = false;   // one launch per reference phase, one sync per phase (ScopedTimer semantics)
  bool prof_fused = false;  // event pairs around the launches of the production (fused) loop; same kernels as unprofiled
  std::vector<mpm::Phase> phases;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the launch's OWN start / stop timestamps (hipExtLaunchKernelGGL): what rocprofv3 --kernel-trace reports as the kernel's duration.
  // An event bracket around a launch adds ~3 us of packet processing; in prof_fused mode the hot launches carry these two events
  // and ScopedPhase reports both times (kernel, bracket).
  hipEvent_t kev0 = nullptr, kev1 = nullptr;
  bool kev_pending = false;
  bool phase_open = false;  // a ScopedPhase bracket is open: the context has ONE ev0 / ev1 pair, an inner bracket would re-record it

  mpm::FastState *fast = nullptr;
  bool caller_dirty = true;    // caller arrays newer than the internal state (fast mode)
  bool internal_dirty = false; // internal state newer than the caller arrays (fast mode)
};

namespace mpm {

inline int fail(mpmhip_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define MPM_HIP_CHECK(ctx, expr)                                                              \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return mpm::fail(ctx, MPMHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline size_t G3(const mpmhip_ctx *c) { return (size_t)c->cfg.n_grid * c->cfg.n_grid * c->cfg.n_grid; }

// profiling brackets: no-ops unless enabled (then one sync per phase, like ScopedTimer(synchronize=True))
struct ScopedPhase {
  mpmhip_ctx *c;
  int idx;
  ScopedPhase(mpmhip_ctx *ctx, const char *name) : c(ctx), idx(-1) {
    if (!c->profiling && !c->prof_fused) return;
    // nested bracket (e.g. the pending g2p flushed from inside the re-sort's bracket: rebin -> flush_elements -> flush_g2p): the
    // outer phase keeps the event pair and its time includes the inner work (the inner one used to re-record ev0 and the
    // outer "rebin" time was under-reported)
    if (c->phase_open) return;
    c->phase_open = true;
    for (size_t i = 0; i < c->phases.size(); ++i)
      if (c->phases[i].name == name || std::string(c->phases[i].name) == name) idx = (int)i;
    if (idx < 0) {
      c->phases.push_back(Phase{name});
      idx = (int)c->phases.size() - 1;
    }
    c->kev_pending = false;
    (void)hipEventRecord(c->ev0, c->stream);
  }
  ~ScopedPhase() {
    if (idx < 0) return;
    c->phase_open = false;
    (void)hipEventRecord(c->ev1, c->stream);
    (void)hipEventSynchronize(c->ev1);
    float ms = 0.f, kms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->phases[idx].total_ms += ms;
    c->phases[idx].samples += 1;
    if (c->kev_pending && hipEventElapsedTime(&kms, c->kev0, c->kev1) == hipSuccess) {
      c->phases[idx].kernel_ms += kms;
      c->phases[idx].kernel_samples += 1;
    }
    c->kev_pending = false;
  }
};

int baseline_init(mpmhip_ctx *ctx);
int baseline_step(mpmhip_ctx *ctx, const StepArgs &a);
int baseline_add_collider_storage(mpmhip_ctx *ctx, MeshCollider &mc);
int baseline_add_mover_storage(mpmhip_ctx *ctx, Mover &mv);

int fast_init(mpmhip_ctx *ctx);
void fast_destroy(mpmhip_ctx *ctx);
int fast_step(mpmhip_ctx *ctx, const StepArgs &a);
int fast_body_at_rest_begin(mpmhip_ctx *ctx, int n_substeps);   // head / tail of every mpmhip_steps call (fast.hip)
void fast_body_at_rest_end(mpmhip_ctx *ctx);
int fast_pull(mpmhip_ctx *ctx);
int fast_export_grid(mpmhip_ctx *ctx, float *m, float *v_in, float *v_out);
int fast_stats(mpmhip_ctx *ctx, mpmhip_stats *out);
int fast_set_debug_flags(mpmhip_ctx *ctx, int flags);
int fast_debug_counter(mpmhip_ctx *ctx, int index, int64_t *out);
int fast_debug_wgtrace(mpmhip_ctx *ctx, int kernel, uint64_t *out, int max_wg);
int fast_debug_sort(mpmhip_ctx *ctx, const uint32_t *keys_in, int n, int bits, uint32_t *keys_out, int32_t *order_out);
int fast_add_collider_storage(mpmhip_ctx *ctx, MeshCollider &mc);
int fast_add_mover_storage(mpmhip_ctx *ctx, Mover &mv);

int fast_dist_enable(mpmhip_ctx *ctx);
int fast_dist_set_ghost_mode(mpmhip_ctx *ctx, int ghosts_gather);
int fast_dist_set_mass_span(mpmhip_ctx *ctx, float min_mass, float max_mass);
int fast_dist_ghosts(mpmhip_ctx *ctx, int send);
int fast_dist_num_blocks(const mpmhip_ctx *ctx);
int64_t fast_dist_halo_bytes(const mpmhip_ctx *ctx);
int fast_dist_halo_transport(const mpmhip_ctx *ctx);
int64_t fast_dist_fused_halo_steps(const mpmhip_ctx *ctx);
int fast_dist_drift_flag(mpmhip_ctx *ctx, int32_t *out);
int fast_dist_rebin(mpmhip_ctx *ctx, unsigned char *active_map);
int fast_dist_set_peers(mpmhip_ctx *ctx, int n, const mpmhip_dist_peer *peers);
int fast_dist_phase(mpmhip_ctx *ctx, int phase, const StepArgs &a);

int fast_rccl_unique_id(char id[128], std::string &err);
int fast_rccl_init(mpmhip_ctx *ctx, int rank, int world, const char id[128]);
int fast_rccl_set_ghosts(mpmhip_ctx *ctx, int n, const int32_t *ranks, const int32_t *nsp, const int32_t *const *sp,
                         const int32_t *nrp, const int32_t *const *rp, const int32_t *nse, const int32_t *const *se,
                         const int32_t *nre, const int32_t *const *re);
int fast_rccl_steps(mpmhip_ctx *ctx, float dt, int n, int64_t step_index, int rebin_interval, const float *mesh_x,
                    const float *mesh_v, const float *jt, int n_jt, const float *jv, const float *jf);

// shared small kernels (common.hip)
int launch_pre_ops(mpmhip_ctx *ctx, float dt, float *v, const float *x, const float *mass, int n);
int launch_select_box(mpmhip_ctx *ctx, const float *x, const float point[3], const float size[3], int32_t *mask);
int launch_select_cylinder(mpmhip_ctx *ctx, const float *x, const float point[3], const float normal[3],
                           float half_height, float radius, int32_t *mask);
int count_nonzero(mpmhip_ctx *ctx, const float *a, size_t n, float thresh, int *out);

}  // namespace mpm
