/*
 * mpmhip.h -- C ABI of libmpmhip.so, the MI355X (gfx950) MPM substep solver.
 *
 * Drop-in boundary for the hot path of KAISTChangmin/MPMAvatar: one MPM substep,
 * `MPMWARP.p2g2p` (/root/reference/warp_mpm/mpm_solver.py:229-536) and the solver
 * state it advances.  The reference has no FFI for this path (it is NVIDIA-Warp DSL
 * called from Python); the entry points below are what a ctypes
 * binding of `warp_mpm/mpm_solver.py` + `mpm_data_structure.py` needs, one group per
 * reference interface.  The Python shim `mpmavatar_amd/warp_mpm/` is that binding.
 *
 * Conventions
 *   - every pointer marked [dev] is a device pointer on `config.device` (e.g.
 *     torch.Tensor.data_ptr() of a ROCm tensor); [host] pointers are read during the
 *     call and not retained.  No torch types cross this boundary.
 *   - particle arrays use the reference's layout: AoS fp32, vec3 = 3 floats,
 *     mat33 = 9 floats row-major; index classes [0,n_elements) elements,
 *     [n_elements,n_nv) traditional, [n_nv,n_particles) vertices,
 *     n_nv = n_particles - n_vertices (mpm_solver.py:19-26, SURVEY.md 8 layout).
 *   - calls are asynchronous on the context's stream unless stated; every function
 *     returns MPMHIP_OK (0) or a negative error code, mpmhip_last_error() has the text.
 *   - a context is bound to one GPU and is not thread-safe; distinct contexts are
 *     independent (one per rank in multi-GPU runs).
 */
#ifndef MPMHIP_H
#define MPMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPMHIP_VERSION 100

enum {
  MPMHIP_OK = 0,
  MPMHIP_ERR_INVALID = -1,   /* bad argument / inconsistent sizes  (reference: assert / RuntimeError) */
  MPMHIP_ERR_NO_DEVICE = -2, /* no HIP device visible: the solver never falls back to a CPU path */
  MPMHIP_ERR_HIP = -3,       /* a HIP runtime call failed */
  MPMHIP_ERR_STATE = -4,     /* called before the state/model were bound */
  MPMHIP_ERR_LIMIT = -5      /* too many colliders / boundary conditions */
};

enum { MPMHIP_MODE_FAST = 0,      /* cell-sorted SoA particles + block-sparse grid + LDS-tiled transfers */
       MPMHIP_MODE_BASELINE = 1   /* reference-structured kernels on the caller's AoS arrays, dense grid */ };

typedef struct mpmhip_ctx mpmhip_ctx;

/* MPMWARP.__init__/initialize arguments, mpm_solver.py:14-26 */
typedef struct {
  int32_t n_particles, n_elements, n_vertices;
  int32_t n_grid;
  float grid_lim;
  int32_t num_joint_t, num_joint_v, num_joint_f;
  int32_t device;         /* HIP device ordinal */
  int32_t mode;           /* MPMHIP_MODE_* */
  int32_t rebin_interval; /* fast mode: max substeps between particle re-sorts (a device-side drift flag triggers
                             earlier ones); 0 = default (256); < 0 = exactly every -n substeps, drift flag ignored */
  int32_t own_stream;     /* 1: create a private non-blocking stream and ignore `stream` */
  void *stream;           /* own_stream == 0: hipStream_t to launch on (NULL = the HIP null stream), e.g.
                             torch.cuda.current_stream().cuda_stream so that solver work is ordered with the
                             caller's tensor ops the way Warp's stream is with torch's */
  int32_t p2g_tile;       /* fast mode, the one setting that changes NUMERICS: accumulator of p2g's chunk tile.
                             MPMHIP_P2G_TILE_AUTO (0): packed fixed point, or fp64 when the particle masses of the scene span more
                             than 1e5 (decided at every import of the state); MPMHIP_P2G_TILE_FIXED (1); MPMHIP_P2G_TILE_F64 (2).
                             (The environment variable MPMHIP_P2G_TILE=fx|f64 overrides it for experiments.) */
  int32_t reserved_;      /* must be 0 */
} mpmhip_config;
#define MPMHIP_P2G_TILE_AUTO 0
#define MPMHIP_P2G_TILE_FIXED 1
#define MPMHIP_P2G_TILE_F64 2

/* MPMStateStruct fields the substep touches, mpm_data_structure.py:13-49.  All [dev]. */
typedef struct {
  float *particle_x;        /* [n_particles*3] */
  float *particle_v;        /* [n_particles*3] */
  float *particle_C;        /* [n_particles*9] */
  float *particle_F;        /* [n_nv*9] */
  float *particle_F_trial;  /* [n_nv*9] */
  float *particle_stress;   /* [n_nv*9] */
  float *particle_d;        /* [n_elements*9] */
  float *particle_R_inv;    /* [n_elements*3] */
  const float *faces;       /* [n_elements*3] float-encoded vertex ids (quirk Q8) */
  float *vertex_force;      /* [n_vertices*3] */
  const float *particle_vol;  /* [n_particles] */
  const float *particle_mass; /* [n_particles] */
  const int32_t *particle_selection; /* [n_particles], 0 = simulate, anything else = not simulated (mpm_utils.py:492); the
                                        value 2 marks a ghost copy, but only after mpmhip_dist_enable */
} mpmhip_state_ptrs;

/* MPMModelStruct arrays, mpm_data_structure.py:621-630.  All [dev], [n_particles]. */
typedef struct {
  float *mu, *lam, *gamma, *kappa, *yield_stress;
} mpmhip_model_ptrs;

/* MPMModelStruct scalars, mpm_data_structure.py:627-645 + init_other_params :686-715 */
typedef struct {
  int32_t material; /* 0 jelly 1 metal 2 sand 3 foam 4 snow 5 plasticine 6 neo-hookean 7 cloth */
  float friction_coeff, alpha;
  float g[3];
  float hardening, xi, plastic_viscosity, softening;
  float rpic_damping, grid_v_damping_scale;
} mpmhip_model_scalars;

typedef struct {
  int64_t substeps;        /* p2g2p calls since creation */
  int64_t rebins;          /* particle re-sorts (fast mode) */
  int32_t n_active_blocks; /* 4x4x4-node grid blocks currently swept (fast mode) */
  int32_t n_active_nodes;  /* nodes with mass > 0 after the last p2g (filled by mpmhip_measure) */
  int32_t n_collider_nodes;
  int32_t n_mover_nodes;
  int32_t n_fallback_particles; /* particles that left their tile margin since the last re-sort */
  int32_t n_dropped;            /* fast mode: scatter contributions that fell outside the active blocks (cumulative).
                                   Must stay 0: non-zero means particles outran the re-sorts (fixed-interval mode
                                   with an interval too long for their speed) and the results are not valid */
  int64_t g2p2g_launches;       /* fast mode, scenes of traditional particles only: substep boundaries that ran as ONE launch
                                   (g2p of substep n + stress and p2g of substep n + 1, csrc/g2p.hip k_g2p2g) */
  int32_t p2g_tile_in_use;      /* fast mode: the accumulator p2g's chunk tile runs on NOW -- MPMHIP_P2G_TILE_FIXED or MPMHIP_P2G_TILE_F64
                                   (what MPMHIP_P2G_TILE_AUTO resolved to at the last import; 0 in baseline mode) */
  int32_t kept_collider_substeps; /* fast mode: substeps of mpmhip_steps calls whose body was at rest (mesh_v == 0 for every vertex) that
                                     ran WITHOUT body-face splat workgroups: the collider field is splatted once per accumulator buffer
                                     and kept until the particle order changes or the call ends (csrc/fast.hip fast_body_at_rest_begin) */
} mpmhip_stats;

/* ---- lifetime ----------------------------------------------------------------- */
int mpmhip_version(void);
int mpmhip_device_count(void); /* 0 when no GPU is visible (never an error) */
/* MPMWARP(...) constructor, mpm_solver.py:14-51 */
int mpmhip_create(const mpmhip_config *cfg, mpmhip_ctx **out);
void mpmhip_destroy(mpmhip_ctx *ctx);
/* text of the last error on ctx (ctx may be NULL for mpmhip_create failures) */
const char *mpmhip_last_error(const mpmhip_ctx *ctx);

/* ---- state / model binding ------------------------------------------------------ */
/* MPMStateStruct.from_torch / reset_state / continue_from_torch rebind arrays
 * (mpm_data_structure.py:158-419): call again whenever any pointer changes.  The caller's
 * arrays are taken as the authoritative state at the next step. */
int mpmhip_bind_state(mpmhip_ctx *ctx, const mpmhip_state_ptrs *p);
/* MPMModelStruct.init + set_E_nu + prepare_mu_lam results, mpm_solver.py:128-227 */
int mpmhip_bind_model(mpmhip_ctx *ctx, const mpmhip_model_ptrs *p);
/* MPMWARP.set_parameters_dict scalars, mpm_solver.py:57-126 */
int mpmhip_set_model_scalars(mpmhip_ctx *ctx, const mpmhip_model_scalars *s);
/* The caller changed bound arrays in place (particle or model): re-import before the next step. */
int mpmhip_push_state(mpmhip_ctx *ctx);
/* Make the caller's particle_x/v/C/F/F_trial/stress/d/vertex_force current (the equivalent of the
 * zero-copy wp.to_torch(mpm_state.particle_x) read, run_demo.py:532).  No-op in baseline mode. */
int mpmhip_pull_state(mpmhip_ctx *ctx);

/* ---- body mesh, colliders, boundary conditions ----------------------------------- */
/* wp.Mesh(points, velocities=0, indices), mpm_solver.py:45-51.  verts/faces are [host]. */
int mpmhip_set_body_mesh(mpmhip_ctx *ctx, int32_t n_verts, int32_t n_faces, const float *verts,
                         const int32_t *faces);
/* MPMWARP.add_mesh_collider, mpm_solver.py:805-919 */
int mpmhip_add_mesh_collider(mpmhip_ctx *ctx, float friction);
/* MPMWARP.add_particle_mover, mpm_solver.py:661-802 */
int mpmhip_add_particle_mover(mpmhip_ctx *ctx);
/* MPMWARP.add_surface_collider, mpm_solver.py:564-658; normal already normalised,
 * surface_type 0 sticky / 1 slip / 11 cut / 2 other */
int mpmhip_add_surface_collider(mpmhip_ctx *ctx, const float point[3], const float normal[3],
                                int32_t surface_type, float friction, float start_time, float end_time);
/* MPMWARP.set_velocity_on_cuboid, mpm_solver.py:929-984 (the host-side `modify` is applied inside step) */
int mpmhip_add_velocity_cuboid(mpmhip_ctx *ctx, const float point[3], const float size[3],
                               const float velocity[3], float start_time, float end_time, int32_t reset);
/* MPMWARP.add_bounding_box, mpm_solver.py:986-1053 */
int mpmhip_add_bounding_box(mpmhip_ctx *ctx, float start_time, float end_time);
/* MPMWARP.enforce_grid_velocity_by_mask, mpm_solver.py:1330-1355; mask [dev] int32 [n_grid^3] */
int mpmhip_add_grid_mask(mpmhip_ctx *ctx, const int32_t *mask);

/* ---- pre-p2g particle operations (mpm_solver.py:1058-1417) ------------------------ */
/* selection kernels, mpm_utils.py:1198-1248: write 0/1 into mask [dev] int32 [n_particles] */
int mpmhip_select_box(mpmhip_ctx *ctx, const float point[3], const float size[3], int32_t *mask);
int mpmhip_select_cylinder(mpmhip_ctx *ctx, const float point[3], const float normal[3],
                           float half_height, float radius, int32_t *mask);
/* add_impulse_on_particles (per_mass=1: v += force/mass*dt where mask==1, :1093-1104) and
 * add_impulse_on_particles_with_mask (per_mass=0: v += force*dt where mask>=1, :1399-1415) */
int mpmhip_add_impulse(mpmhip_ctx *ctx, const float force[3], const int32_t *mask, int32_t per_mass,
                       float start_time, float end_time);
/* enforce_particle_velocity_translation / _by_mask, :1138-1149, :1315-1326 */
int mpmhip_add_velocity_set(mpmhip_ctx *ctx, const float velocity[3], const int32_t *mask,
                            float start_time, float end_time);
/* enforce_particle_velocity_rotation, :1156-1257 */
int mpmhip_add_velocity_rotation(mpmhip_ctx *ctx, const float point[3], const float normal[3],
                                 const float axis1[3], const float axis2[3], float rotation_scale,
                                 float translation_scale, const int32_t *mask, float start_time,
                                 float end_time);

/* ---- the substep --------------------------------------------------------------------- */
/* MPMWARP.p2g2p, mpm_solver.py:229-536.  mesh_x, mesh_v [dev][num_mesh_v*3]; joint_traditional_v
 * [dev][n_joint_t*3]; joint_verts_v [dev][num_joint_v*3]; joint_faces_v [dev][num_joint_f*3];
 * NULL = the Python argument None. */
int mpmhip_step(mpmhip_ctx *ctx, float dt, const float *mesh_x, const float *mesh_v,
                const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                const float *joint_faces_v);
/* n substeps with the caller's per-substep mesh advection mesh_x + k*dt*mesh_v fused on the device
 * (the loop at train_material_params.py:622-626 / run_demo.py:526-530) */
int mpmhip_steps(mpmhip_ctx *ctx, float dt, int32_t n, const float *mesh_x, const float *mesh_v,
                 const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                 const float *joint_faces_v);
int mpmhip_synchronize(mpmhip_ctx *ctx);
/* MPMWARP.time (never reset by reset_state, quirk Q3) */
double mpmhip_get_time(const mpmhip_ctx *ctx);
int mpmhip_set_time(mpmhip_ctx *ctx, double t);
/* mpm_solver.py:536: `self.time = self.time + dt` adds the caller's Python float (a double) while the kernels receive dt as
 * fp32.  Tell the library that double once (and whenever dt changes); steps whose fp32 dt equals (float)dt_host then advance
 * MPMWARP.time by dt_host exactly as the reference does.  Without it time advances by (double)(float)dt. */
int mpmhip_set_host_dt(mpmhip_ctx *ctx, double dt_host);

/* ---- multi-GPU (one process and one context per GPU; not in the reference) --------------------
 * Particles are sharded across ranks by the caller (mpmavatar_amd/dist.py: spatial x-slabs, re-cut at the particles' current positions when more than a
 * tenth of them have left their slab); every rank runs
 * its own context on its particles plus ghost copies (particle_selection == 2: stress yes, transfers no).  The
 * substep is split in three so that the caller can run its two neighbour exchanges (RCCL send/recv through
 * torch.distributed) between the phases:
 *   begin: [stress, p2g] + pack halo_send      -> exchange halo   (sum of the grid blocks both ranks touch)
 *   mid  : add halo_recv, [grid, g2p] + pack ghost_send -> exchange ghosts (x, v of vertices; d3 of elements)
 *   end  : unpack ghost_recv, [element finalise]
 * All ranks must re-sort at the same substep: mpmhip_dist_rebin() replaces the context's own re-sort policy.
 * With mpmhip_dist_set_ghost_mode(ctx, 1) the ghost copies gather for themselves (g2p yes, p2g no): their grid
 * neighbourhood is on both ranks' active lists and therefore complete after the halo sum, so the per-substep ghost
 * exchange disappears (mid / end no longer pack / unpack); owners and copies differ only by the rounding order of the
 * halo sums, and mpmhip_dist_ghost_pack / _unpack re-synchronise them around an exchange at each collective re-sort. */
typedef struct {
  int32_t n_blocks;          /* grid blocks on both ranks' active lists */
  const int32_t *blocks;     /* [dev] their ids in ascending order (identical on both ranks) */
  float *halo_send, *halo_recv; /* [dev] n_blocks * CH * 64 floats; CH = 8 when a particle mover exists, else 4 */
  int32_t n_send_p, n_recv_p, n_send_e, n_recv_e;
  const int32_t *send_p, *recv_p; /* [dev] caller-order indices of vertices/traditional particles sent / received */
  const int32_t *send_e, *recv_e; /* [dev] caller-order indices of elements whose d3 is sent / received */
  float *ghost_send, *ghost_recv; /* [dev] 6*n_p + 3*n_e floats */
} mpmhip_dist_peer;
int mpmhip_dist_enable(mpmhip_ctx *ctx);
/* 0 (default): ghost copies are overwritten by their owners' values every substep; 1: they gather for themselves.
 * Call before the first mpmhip_dist_rebin. */
int mpmhip_dist_set_ghost_mode(mpmhip_ctx *ctx, int32_t ghosts_gather);
int mpmhip_dist_ghost_pack(mpmhip_ctx *ctx);   /* fill every peer's ghost_send */
int mpmhip_dist_ghost_unpack(mpmhip_ctx *ctx); /* apply every peer's ghost_recv */
int mpmhip_dist_num_blocks(const mpmhip_ctx *ctx); /* size of the active-block byte map */
/* MPMHIP_P2G_TILE_AUTO in a sharded run: the smallest positive and the largest mass over the simulated particles of ALL ranks (the
 * caller all-reduces them).  The tile decision of every later import is taken from max(this rank's span, max_mass / min_mass), so
 * every rank switches to the fp64 tile together -- ranks that share halo blocks must not run different accumulator numerics -- and
 * from the masses actually bound (reset_density(update_mass) after the build included), never from a description of the scene.
 * AUTO is only ever widened to fp64 this way, never forced to the fixed-point tile.  min_mass <= 0: forget the global span. */
int mpmhip_dist_set_mass_span(mpmhip_ctx *ctx, float min_mass, float max_mass);
/* this rank's early-warning drift flag (1: some particle is about to leave the tile margin of the block it was sorted
 * into; cleared by the next re-sort).  Synchronous.  A sharded driver max-reduces it over the ranks to decide on a
 * collective re-sort -- the single-GPU adaptive policy (mpmhip_config.rebin_interval = 0) made collective. */
int mpmhip_dist_drift_flag(mpmhip_ctx *ctx, int32_t *flag);
/* import the bound state if needed, re-sort, and write this rank's active-block map (1 byte per block) [dev] */
int mpmhip_dist_rebin(mpmhip_ctx *ctx, uint8_t *active_map);
int mpmhip_dist_set_peers(mpmhip_ctx *ctx, int32_t n_peers, const mpmhip_dist_peer *peers);
int mpmhip_dist_step_begin(mpmhip_ctx *ctx, float dt, const float *mesh_x, const float *mesh_v, float mesh_advect,
                           const float *joint_traditional_v, int32_t n_joint_t, const float *joint_verts_v,
                           const float *joint_faces_v);
int mpmhip_dist_step_mid(mpmhip_ctx *ctx);
int mpmhip_dist_step_end(mpmhip_ctx *ctx);

/* RCCL transport inside the library (no Python in the substep loop): the communicator is created from a
 * ncclUniqueId that rank 0 obtains with mpmhip_rccl_unique_id() and the caller broadcasts (torch.distributed).
 * librccl.so.1 is dlopen'ed on first use (the copy torch has already loaded, if any). */
int mpmhip_rccl_unique_id(char id[128]);
int mpmhip_rccl_init(mpmhip_ctx *ctx, int32_t rank, int32_t world, const char id[128]);
/* static ghost lists per peer rank ([host] int arrays of caller-order particle indices, see mpmhip_dist_peer) */
int mpmhip_rccl_set_ghosts(mpmhip_ctx *ctx, int32_t n_peers, const int32_t *peer_ranks, const int32_t *n_send_p,
                           const int32_t *const *send_p, const int32_t *n_recv_p, const int32_t *const *recv_p,
                           const int32_t *n_send_e, const int32_t *const *send_e, const int32_t *n_recv_e,
                           const int32_t *const *recv_e);
/* n substeps (collective): re-sort + shared-block lists every rebin_interval substeps (counted from step_index) if
 * rebin_interval > 0; if <= 0, when the max-reduced drift flag of the ranks (ncclAllReduce of one int every 16
 * substeps, read 4 substeps later so that all ranks act at the same substep) asks for it, at the latest every 256
 * (or -rebin_interval) substeps.  Halo (and, in ghost mode 0, ghost) exchanges with ncclSend/ncclRecv groups on the context's stream; in ghost mode 1
 * the ghosts are re-synchronised before every re-sort instead.  Mesh advection factor of substep k is
 * (step_index + k) * dt.  joint_traditional_v: velocities of the LAST n_joint_t traditional particles this rank owns (the
 * staged release of run_demo.py:524; a rank's share of the held particles is a suffix of its owned ones), or NULL */
/* bytes this rank sends per substep in the halo exchange with the current shared-block lists (measurement) */
int mpmhip_dist_halo_bytes(mpmhip_ctx *ctx, int64_t *out);
/* how mpmhip_rccl_steps moves the halos: 1 = peer-mapped buffers (each pair of neighbouring ranks maps the other's
 * fine-grained receive arena with HIP IPC at the first collective re-sort; the pack kernel stores into it and raises a flag
 * there, the add kernel waits for its own flag -- no RCCL kernel in the substep), 0 = ncclSend/ncclRecv groups.  Peer
 * mapping is the default and is used only if a four-round handshake over every link of every rank succeeded (max-reduced);
 * MPMHIP_DIST_HALO=rccl keeps send/recv.  No counterpart in the reference (single GPU). */
int mpmhip_dist_halo_transport(mpmhip_ctx *ctx, int32_t *out);
/* substeps of mpmhip_rccl_steps so far that had NO halo kernels: with peer-mapped halos the pack rides in the p2g launch as
 * trailing workgroups (they wait until every scattering workgroup of that launch has counted itself done) and g2p adds the
 * neighbour's share to the shared blocks while it stages its tile.  Falls back to the pack / add kernels for an interval in
 * which a block is shared with more than one neighbour, a pair's arena is too small, or profiling brackets the launches;
 * MPMHIP_DIST_FUSED_HALO=0 switches it off. */
int mpmhip_dist_fused_halo_steps(mpmhip_ctx *ctx, int64_t *out);
int mpmhip_rccl_steps(mpmhip_ctx *ctx, float dt, int32_t n, int64_t step_index, int32_t rebin_interval,
                      const float *mesh_x, const float *mesh_v, const float *joint_traditional_v, int32_t n_joint_t,
                      const float *joint_verts_v, const float *joint_faces_v);

/* ---- after the solver: per-face frames and bound Gaussians --------------------------------
 * Stand-alone maps on [dev] arrays (no solver context; `stream` is a hipStream_t, NULL = default stream).
 * mpmhip_face_frames = MeshGaussianModel.set_mesh_by_verts (scene/mesh_gaussian_model.py:137-146) with
 * compute_face_orientation(return_scale=True) (utils/graphics_utils.py:88-106):
 *   face_center [n_f*3] = mean of the three vertices, face_orien_mat [n_f*9] row-major with columns a0 a1 a2,
 *   face_orien_quat [n_f*4] WXYZ = quat_xyzw_to_wxyz(rotmat_to_unitquat(mat)) (roma), face_scaling [n_f]. */
int mpmhip_face_frames(int32_t device, void *stream, const float *verts, const int32_t *faces, int32_t n_faces,
                       float *face_center, float *face_orien_mat, float *face_orien_quat, float *face_scaling);
/* GaussianModel.get_xyz / get_rotation / get_scaling with a face binding (scene/gaussian_model.py:112-151):
 *   xyz [n_g*3] = mat[b] xyz_local * scaling[b] + center[b];  rotation [n_g*4] WXYZ = normalize(quat[b]) (x)
 *   normalize(rotation_raw);  scaling [n_g*3] = exp(scaling_raw) * face_scaling[b].  Any output may be NULL. */
int mpmhip_bind_gaussians(int32_t device, void *stream, int32_t n_gaussians, const int32_t *binding, const float *xyz_local,
                          const float *rotation_raw, const float *scaling_raw, const float *face_center,
                          const float *face_orien_mat, const float *face_orien_quat, const float *face_scaling, float *xyz,
                          float *rotation, float *scaling);

/* The rasteriser's inputs, SURVEY.md 8(f) N4: what gaussian_renderer/__init__.py:52-103 hands to GaussianRasterizer for a mesh-bound
 * model plus the caller's `extra` primitives (run_demo.py:578-604: sand and chair), assembled in one launch into buffers of
 * n_gaussians + n_extra rows: means3D [*3] = get_xyz (scene/gaussian_model.py:141-151) | extra_xyz; means2D [*3] = 0 (:27);
 * opacities [*1] = sigmoid(opacity_raw) (:158-160) | extra_opacity; scales [*3] = get_scaling (:112-122) | extra_scales;
 * rotations [*4] WXYZ = get_rotation (:124-138) | extra_rotations -- the five torch.cat of :84-91.  Colours (shs or
 * colors_precomp) are the caller's tensors unchanged.  With frames taken from the solver's particle_x on the device this replaces
 * the per-frame OBJ write / re-read of train_material_params.py:819-845 for everything but the Blender AO bake. */
int mpmhip_render_inputs(int32_t device, void *stream, int32_t n_gaussians, int32_t n_extra, const int32_t *binding,
                         const float *xyz_local, const float *rotation_raw, const float *scaling_raw, const float *opacity_raw,
                         const float *face_center, const float *face_orien_mat, const float *face_orien_quat,
                         const float *face_scaling, const float *extra_xyz, const float *extra_opacity, const float *extra_scales,
                         const float *extra_rotations, float *means3D, float *means2D, float *opacities, float *scales,
                         float *rotations);

/* MPMWARP.export_particle_cov_to_torch (warp_mpm/mpm_solver.py:543-561) = kernel compute_cov_from_F
 * (warp_mpm/mpm_utils.py:1108-1132): new_cov[6p..] = upper triangle (xx xy xz yy yz zz) of F_trial[p] * sym(particle_cov[6p..])
 * * F_trial[p]^T for p < n (= n_particles - n_vertices).  Stand-alone map on [dev] arrays in the reference's AoS layout. */
int mpmhip_cov_from_F(int32_t device, void *stream, const float *particle_F_trial, const float *particle_cov, int32_t n,
                      float *new_cov);

/* ---- introspection ---------------------------------------------------------------------- */
/* dense reference-layout copies of grid_m [G^3], grid_v_in [G^3*3], grid_v_out [G^3*3] as they
 * stand after the last substep's grid stage ([dev] outputs, any may be NULL).  Synchronous. */
int mpmhip_export_grid(mpmhip_ctx *ctx, float *grid_m, float *grid_v_in, float *grid_v_out);
/* performance experiments only (kernel ablations, MPMHIP_DBG bit mask of csrc/fast_device.hpp; most bits make the results wrong).
 * The kernel switches exist only in -DMPMHIP_DEBUG=1 builds; the production build accepts bit 64 (host-side) alone. */
int mpmhip_set_debug_flags(mpmhip_ctx *ctx, int32_t flags);
int mpmhip_debug_counter(mpmhip_ctx *ctx, int32_t index, int64_t *out); /* device-side experiment counters, synchronous */
/* per-workgroup timeline of the newest p2g (kernel 0) / g2p (kernel 1) launch: out[wg * 8 + slot] in ticks of the 100 MHz
 * constant clock, slot 7 = (XCC_ID << 32) | HW_ID.  out == NULL starts recording.  Only libraries built with
 * -DMPMHIP_DEBUG=1 carry the stamps (and the kernel switches of mpmhip_set_debug_flags); the production build returns
 * MPMHIP_ERR_INVALID.  tools/gpu/wgtrace.py. */
int mpmhip_debug_wgtrace(mpmhip_ctx *ctx, int32_t kernel, uint64_t *out, int32_t max_wg);
/* the sort of the re-sort on its own (tests/test_gpu_sort.py): stable sort of n 32-bit keys by their low `bits` bits
 * ([dev] keys_in; bits above `bits` must be zero) -> [dev] keys_out (sorted), order_out (source index of each sorted key).
 * Uses the path the context's re-sorts use (csrc/resort.hip k_rs_*; rocPRIM with MPMHIP_SORT=rocprim or n > 2^21).
 * Synchronous; fast mode only. */
int mpmhip_debug_sort(mpmhip_ctx *ctx, const uint32_t *keys_in, int32_t n, int32_t bits, uint32_t *keys_out, int32_t *order_out);
/* counts for the algorithmic-bytes formula; synchronous, runs small count kernels */
int mpmhip_get_stats(mpmhip_ctx *ctx, mpmhip_stats *out);
/* MPMWARP.time_profile / print_time_profile, mpm_solver.py:16,538-541: when enabled every phase is
 * bracketed by hipEvents (forces a sync per substep, like ScopedTimer(synchronize=True)).
 * on = 1: every reference phase is its own launch (the reference's keys; un-fused kernels);
 * on = 2: event pairs around the launches of the production loop -- the same (fused) kernels an unprofiled run
 *         executes, keys compute_stress_from_F_trial / p2g / g2p_v / rebin (what bench.py's roofline uses);
 * on = 0: off (no events, no syncs). */
int mpmhip_profile_enable(mpmhip_ctx *ctx, int32_t on);
int mpmhip_profile_count(const mpmhip_ctx *ctx);
/* i-th phase: name, accumulated milliseconds, number of samples */
int mpmhip_profile_get(const mpmhip_ctx *ctx, int32_t i, const char **name, double *total_ms,
                       int64_t *samples);
/* on = 2 only: the i-th phase's launch timed by its OWN start / stop timestamps (the kernel duration a profiler's kernel trace
 * reports), without what the event bracket adds around it; samples = 0 for phases that are not a single hot launch.
 * (No reference counterpart: the reference's ScopedTimer only has the synchronised wall time, mpm_solver.py:16.) */
int mpmhip_profile_get_kernel(const mpmhip_ctx *ctx, int32_t i, double *kernel_ms, int64_t *samples);
int mpmhip_profile_reset(mpmhip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MPMHIP_H */
