// fast.hip -- MPMHIP_MODE_FAST: the MI355X-native substep.
//
// Data layout:
//   * particles live in a solver-owned SoA copy (component-major fp32 arrays), ordered class-major
//     (elements | traditional | vertices) and, inside each class, by (4x4x4-cell grid block, cell).  The order is
//     rebuilt (own radix sort of 30-bit keys, k_rs_*, + one gather pass) when a device-side drift flag asks for it, at the
//     latest every `rebin_interval` substeps; until then a particle may sit up to one cell outside its block, which
//     the transfer tiles absorb, and anything further out takes global-memory paths inside the same kernels.
//   * the grid is stored block-major: block b = (x>>2,y>>2,z>>2) owns 64 nodes, channel-major inside the
//     block ([block][channel][64 nodes]) so one wavefront reads one channel of one block as 256 B.
//     Only blocks on the active list (27-neighbourhoods of particle blocks) are ever touched.
// Launches per substep (single stream, no events):
//   1. stress      per-particle map; fuses the tail of the previous substep's g2p_e (element finalise) and carries
//                  extra workgroups that clear the grid accumulators the previous substep left loaded
//   2. p2g         one workgroup per 256-particle chunk of a block: LDS tile (8x8x8 nodes) in packed fixed point, DPP
//                  pre-reduction, two ds_add_u64 per node, coalesced flush; extra workgroups do the body-face and joint
//                  splats (fp64 tile, ds_add_f64)
//   3. g2p         same chunks; the tile is staged from the accumulators and every node goes through the grid stage
//                  (normalise, gravity, damping, collide, mover, BCs) on the way -- there is no grid kernel
// Profiling runs (one sync per reference phase) and export use the stand-alone k_grid instead.
// Reference semantics: /root/reference/warp_mpm/mpm_utils.py, mpm_solver.py:229-536 (cited per kernel).
#include "fast_state.hpp"

namespace mpm {

namespace {


// Stand-alone grid stage: writes v_out (and the node mass, for introspection).  ZERO = true is the classic form
// (profiling runs, where every phase of the reference gets its own launch); ZERO = false materialises v_out after a
// fused substep for export_grid / stats without disturbing the accumulators.
template <bool ZERO>
__global__ __launch_bounds__(TPB) void k_grid(const int *alist, int n_A, Dims d, GridPtrs g, GridParams gp, BCList bcl) {
  int w = xcd_slice(blockIdx.x, (n_A + 3) / 4);
  if (w < 0) return;
  int a = w * 4 + (threadIdx.x >> 6);
  if (a >= n_A) return;
  int blk = alist[a], l = threadIdx.x & 63;
  int ncol = 0, nmov = 0;
  float m;
  V3 v = node_update<ZERO>(blk, l, d, g, gp, bcl, m, ncol, nmov);
  float *po = g.vout + ((size_t)blk * GCH_VOUT) * 64 + l;
  po[0] = v.x; po[64] = v.y; po[128] = v.z; po[192] = m;
  if (ZERO && l == 0) { g.m_flag[blk] = 0; if (gp.has_col) g.col_flag[blk] = 0; }
  if (gp.count) {  // statistics for the algorithmic-bytes formula (N_coll, N_mov), one atomic per wavefront
    unsigned long long bc = __ballot(ncol), bm = __ballot(nmov);
    if (l == 0) {
      if (bc) atomicAdd(g.counters + 2, __popcll(bc));
      if (bm) atomicAdd(g.counters + 3, __popcll(bm));
    }
  }
}

__global__ __launch_bounds__(TPB) void k_zero_blocks(ZeroArgs z) { zero_blocks_wg(z, blockIdx.x); }


// second half of g2p_e (mpm_utils.py:838-857): x, v = mean of the three updated vertices; d1, d2 = edges
__global__ void k_elem_finalize(Bufs b, const int *face_slot, const SortKey *skeys, int blk_bits, int *counters, Dims d, int step_id) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.n_e) return;
  if (b.sel[e] == 1) return;
  bool ghost = b.sel[e] == 2;
  int v1 = d.n_nv + face_slot[e], v2 = d.n_nv + face_slot[d.n_e + e], v3i = d.n_nv + face_slot[2 * d.n_e + e];
  V3 x1 = ld3(b.all, A_X, v1), x2 = ld3(b.all, A_X, v2), x3 = ld3(b.all, A_X, v3i);
  V3 u1 = ld3(b.all, A_V, v1), u2 = ld3(b.all, A_V, v2), u3 = ld3(b.all, A_V, v3i);
  st3(b.all, A_V, e, v3((u1.x + u2.x + u3.x) / 3.0f, (u1.y + u2.y + u3.y) / 3.0f, (u1.z + u2.z + u3.z) / 3.0f));
  V3 xe = v3((x1.x + x2.x + x3.x) / 3.0f, (x1.y + x2.y + x3.y) / 3.0f, (x1.z + x2.z + x3.z) / 3.0f);
  st3(b.all, A_X, e, xe);
  {  // drift check against the block this element was sorted into
    int blk = key_block(skeys[e], blk_bits);
    int oz = 4 * (blk % d.NB) - 1, oy = 4 * ((blk / d.NB) % d.NB) - 1, ox = 4 * (blk / (d.NB * d.NB)) - 1;
    int nbx = (int)(xe.x * d.inv_dx - 0.5f) - ox, nby = (int)(xe.y * d.inv_dx - 0.5f) - oy, nbz = (int)(xe.z * d.inv_dx - 0.5f) - oz;
    if (!ghost && ((unsigned)nbx > 5u || (unsigned)nby > 5u || (unsigned)nbz > 5u)) raise_drift(counters, step_id);
  }
  V3 d1 = x2 - x1, d2 = x3 - x1;
  b.el.at(E_D + 0, e) = d1.x; b.el.at(E_D + 3, e) = d1.y; b.el.at(E_D + 6, e) = d1.z;
  b.el.at(E_D + 1, e) = d2.x; b.el.at(E_D + 4, e) = d2.y; b.el.at(E_D + 7, e) = d2.z;
}


// pre-p2g particle operations on the sorted state (masks are in the caller's particle order)
__global__ void k_pre_sorted(PreOp op, Bufs b, const int *perm, Dims d, float dt) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_p) return;
  int mk = op.mask[perm[s]];
  V3 pv = ld3(b.all, A_V, s);
  if (op.type == PRE_IMPULSE) {
    if (mk != 1) return;
    float m = b.all.at(A_MASS, s);
    pv = pv + dt * v3(op.force[0] / m, op.force[1] / m, op.force[2] / m);
  } else if (op.type == PRE_IMPULSE_MASK) {
    if (mk < 1) return;
    pv = pv + dt * v3(op.force[0], op.force[1], op.force[2]);
  } else if (op.type == PRE_VEL_SET) {
    if (mk != 1) return;
    pv = v3(op.velocity[0], op.velocity[1], op.velocity[2]);
  } else {
    if (mk != 1) return;
    V3 nrm = v3(op.normal[0], op.normal[1], op.normal[2]);
    V3 a1 = v3(op.axis1[0], op.axis1[1], op.axis1[2]), a2 = v3(op.axis2[0], op.axis2[1], op.axis2[2]);
    V3 off = ld3(b.all, A_X, s) - v3(op.point[0], op.point[1], op.point[2]);
    float hd = length(off - dot(off, nrm) * nrm);
    float theta = acosf(fminf(fmaxf(dot(off, a1) / hd, -1.f), 1.f));  // wp.acos clamps its argument
    if (!(dot(off, a2) > 0.0f)) theta = -theta;
    pv = (-hd * sinf(theta) * op.rotation_scale) * a1 + (hd * cosf(theta) * op.rotation_scale) * a2 + op.translation_scale * nrm;
  }
  st3(b.all, A_V, s, pv);
}


__global__ void k_export_grid(const int *alist, int n_A, Dims d, GridPtrs g, float *gm, float *gvo) {
  int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= n_A) return;
  int blk = alist[a], l = threadIdx.x & 63;
  int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
  int x = 4 * bx + (l >> 4), y = 4 * by + ((l >> 2) & 3), z = 4 * bz + (l & 3);
  if (!in_grid(x, y, z, d.G)) return;
  size_t dense = ((size_t)x * d.G + y) * d.G + z;
  const float *po = g.vout + ((size_t)blk * GCH_VOUT) * 64 + l;
  if (gm) gm[dense] = po[192];
  if (gvo) { gvo[3 * dense] = po[0]; gvo[3 * dense + 1] = po[64]; gvo[3 * dense + 2] = po[128]; }
}


__global__ void k_count_active(const int *alist, int n_A, GridPtrs g, int *out) {
  int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  int c = 0;
  if (a < n_A) c = g.vout[((size_t)alist[a] * GCH_VOUT) * 64 + 192 + (threadIdx.x & 63)] > 0.0f ? 1 : 0;
  unsigned long long bal = __ballot(c);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(out, __popcll(bal));
}

__global__ void k_null() {}

}  // namespace


// run the stand-alone element finalise if the last substep deferred it (and the plain g2p, if that was deferred: G2P2G)
int flush_elements(mpmhip_ctx *c) {
  FastState *f = c->fast;
  flush_g2p(c);
  if (f->elem_pending && f->d.n_e)
    hipLaunchKernelGGL(k_elem_finalize, nblk(f->d.n_e), TPB, 0, c->stream, f->buf[f->cur], f->face_slot, f->keys[1],
                       f->blk_bits, f->g.counters, f->d, f->g.step_id);
  f->elem_pending = false;
  return MPMHIP_OK;
}

// what has to be cleared after the last fused substep (the buffer g points at), marking it clean
ZeroArgs take_zero(FastState *f) {
  // (a buffer that holds the collider field of a body at rest keeps it: col_state 2)
  const int has_col = (f->dirty_col && f->col_state[f->par] != 2) ? 1 : 0;
  ZeroArgs z{f->alist, f->n_A, 0, has_col, f->dirty_mov, f->g.mv, f->g.col, f->g.mov, f->g.m_flag, f->g.col_flag};
  if (f->grid_dirty && f->n_A) z.n_wg = (f->n_A + PT / 64 - 1) / (PT / 64);
  if (f->grid_dirty && f->col_state[f->par] == 1) f->col_state[f->par] = 0;   // (cleared by the workgroups these arguments go to)
  f->grid_dirty = false;
  return z;
}
// collider fields kept for a body at rest: clear them now, over the active list as it stands (before it changes; before a body that
// moves splats into the buffers again)
void drop_kept_collider_fields(mpmhip_ctx *c) {
  FastState *f = c->fast;
  for (int k = 0; k < f->nbuf; ++k) {
    if (f->col_state[k] != 2) continue;
    if (f->n_A && f->col2[k]) {
      // col-only pass: an all-zero flag array stands in for the mass flags (m_flag of the buffer may be live)
      ZeroArgs z{f->alist, f->n_A, (f->n_A + PT / 64 - 1) / (PT / 64), 1, 0, f->mv2[k], f->col2[k], f->mov2[k], f->zero_flags, f->cflag2[k]};
      hipLaunchKernelGGL(k_zero_blocks, (unsigned)z.n_wg, PT, 0, c->stream, z);
    }
    f->col_state[k] = 0;
  }
}
__global__ void k_any_nonzero(const float *v, size_t n, int *flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool nz = i < n && v[i] != 0.0f;
  if (__any(nz) && (threadIdx.x & 63) == 0) *flag = 1;
}
// Head of an mpmhip_steps call: does the body move during the call?  One small kernel over the vertex velocities and one host wait per
// CALL (calls of >= 16 substeps only; a 400-substep frame pays ~20 us for it).  Whatever was kept from the call before is dropped:
// the pose may have changed between calls.
int fast_body_at_rest_begin(mpmhip_ctx *c, int n_substeps) {
  FastState *f = c->fast;
  f->col_at_rest = false;
  drop_kept_collider_fields(c);
  const bool eligible = f->col_keep && n_substeps >= 16 && !c->colliders.empty() && c->num_mesh_f && c->num_mesh_v && f->nbuf == 2 && f->fuse_grid &&
                        !c->profiling && !f->dist && c->cur_vel && !(MPMHIP_DEBUG && f->g.dbg);
  if (!eligible) return MPMHIP_OK;
  int *flag = f->g.counters + 13;
  MPM_HIP_CHECK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
  const size_t nm = (size_t)c->num_mesh_v * 3;
  hipLaunchKernelGGL(k_any_nonzero, (unsigned)((nm + 255) / 256), 256, 0, c->stream, c->cur_vel, nm, flag);
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 31, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  f->col_at_rest = f->h_pin[31] == 0;
  return MPMHIP_OK;
}
void fast_body_at_rest_end(mpmhip_ctx *c) { c->fast->col_at_rest = false; }   // (kept fields are dropped by whoever steps next)
void select_buffer(FastState *f, int par) {
  f->par = par;
  f->g.mv = f->mv2[par]; f->g.col = f->col2[par]; f->g.mov = f->mov2[par];
  f->g.m_flag = f->mflag2[par]; f->g.col_flag = f->cflag2[par];
}
// clear the accumulators now (the active list is about to change)
void flush_grid(mpmhip_ctx *c) {
  FastState *f = c->fast;
  ZeroArgs z = take_zero(f);
  if (z.n_wg) hipLaunchKernelGGL(k_zero_blocks, (unsigned)z.n_wg, PT, 0, c->stream, z);
}
// v_out of the last (fused) substep for export_grid / stats; the accumulators stay as they are
void materialize_grid(mpmhip_ctx *c, bool count) {
  FastState *f = c->fast;
  if (!f->grid_dirty || !f->n_A) return;
  GridParams gp = f->last_gp;
  gp.count = count ? 1 : 0;
  hipLaunchKernelGGL(k_grid<false>, xcd_grid((f->n_A + 3) / 4), TPB, 0, c->stream, f->alist, f->n_A, f->d, f->g, gp, f->last_bcl);
}


int fast_init(mpmhip_ctx *c) {
  const mpmhip_config &cfg = c->cfg;
  if (cfg.n_grid > 512) return fail(c, MPMHIP_ERR_INVALID, "fast mode supports n_grid <= 512");
  FastState *f = new FastState();
  c->fast = f;
  Dims &d = f->d;
  d.n_p = cfg.n_particles; d.n_e = cfg.n_elements; d.n_v = cfg.n_vertices; d.n_nv = c->n_nv; d.n_t = c->n_trad;
  d.G = cfg.n_grid; d.NB = (cfg.n_grid + 3) / 4;
  d.dx = c->dx; d.inv_dx = c->inv_dx; d.grid_lim = cfg.grid_lim;
  f->nblocks = (size_t)d.NB * d.NB * d.NB;
  f->blk_bits_plain = 1;
  while ((1ull << f->blk_bits_plain) < f->nblocks) ++f->blk_bits_plain;
  int cell_bits = f->blk_bits_plain + 8 + 2 + 2 <= 32 ? 8 : 6;  // 8: predictive sort (see make_key)
  if (const char *e = getenv("MPMHIP_PREDICTIVE_SORT")) if (atoi(e) == 0) cell_bits = 6;
  if (const char *e = getenv("MPMHIP_SORT")) f->sort_rocprim = std::string(e) == "rocprim";
  f->p2g_fixed = cfg.p2g_tile != MPMHIP_P2G_TILE_F64;
  f->p2g_fixed_forced = cfg.p2g_tile == MPMHIP_P2G_TILE_FIXED;
  if (const char *e = getenv("MPMHIP_P2G_TILE")) { f->p2g_fixed = std::string(e) != "f64"; f->p2g_fixed_forced = std::string(e) == "fx"; }
  f->p2g_fixed_now = f->p2g_fixed;
  if (const char *e = getenv("MPMHIP_G2P2G")) f->g2p2g = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_COL_KEEP")) f->col_keep = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_SPLIT_SPLAT")) f->split_splat = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_SPLIT_SPLAT_MAX")) f->split_splat_max_chunks = atoi(e);
  if (const char *e = getenv("MPMHIP_G2P2G_MAX")) f->g2p2g_max_chunks = atoi(e);
  f->key_bits = f->blk_bits_plain + cell_bits + 2 + 2;
  if (f->key_bits > 32) return fail(c, MPMHIP_ERR_INVALID, "grid too large for 32-bit sort keys");
  f->blk_bits = f->blk_bits_plain | (cell_bits << 8);
  // upper bound between re-sorts; the drift flag normally triggers one earlier (or never, for slow scenes)
  f->rebin_interval = cfg.rebin_interval > 0 ? cfg.rebin_interval : (cfg.rebin_interval < 0 ? -cfg.rebin_interval : 256);
  f->adaptive_rebin = cfg.rebin_interval >= 0;
  int rc;
  for (int i = 0; i < 2; ++i) {
    if ((rc = alloc_bufs(c, f->buf[i]))) return rc;
    if ((rc = dalloc(c, &f->perm[i], (size_t)d.n_p))) return rc;
    if ((rc = dalloc(c, &f->keys[i], (size_t)d.n_p))) return rc;
  }
  if ((rc = dalloc(c, &f->inv, (size_t)d.n_p))) return rc;
  if ((rc = dalloc(c, &f->face_slot, (size_t)3 * d.n_e))) return rc;
  if ((rc = dalloc(c, &f->eforce, (size_t)3 * d.n_e + 1))) return rc;
  if ((rc = dalloc(c, &f->adj_cnt, (size_t)d.n_v + 1))) return rc;
  if ((rc = dalloc(c, &f->order, (size_t)d.n_p))) return rc;
  if ((rc = dalloc(c, &f->iota, (size_t)d.n_p))) return rc;
  f->nbuf = (d.n_e == 0 && d.n_v == 0 && d.n_t > 0) ? 3 : 2;
  for (int i = 0; i < f->nbuf; ++i) {
    if ((rc = dalloc(c, &f->mv2[i], f->nblocks * GCH_MV * 64))) return rc;
    if ((rc = dalloc(c, &f->mflag2[i], f->nblocks))) return rc;
    if ((rc = dalloc(c, &f->cflag2[i], f->nblocks))) return rc;
  }
  select_buffer(f, 0);
  if ((rc = dalloc(c, &f->g.vout, f->nblocks * GCH_VOUT * 64))) return rc;
  if ((rc = dalloc(c, &f->g.counters, CNT_N))) return rc;
  if ((rc = dalloc(c, &f->pack_done, (size_t)DONE_SHARDS * DONE_STRIDE))) return rc;
  if ((rc = dalloc(c, &f->zero_flags, f->nblocks))) return rc;
  // one allocation, one memset per re-sort: [particle-block flags | active-block flags | device counts]
  static_assert(RC_N <= 64, "device counts of a re-sort");
  f->fc_tiles = (int)((f->nblocks + FC_TILE - 1) / FC_TILE);
  f->fc_groups = (f->fc_tiles + FC_GROUP - 1) / FC_GROUP;
  f->n_clear = (int)(2 * f->nblocks + 64 + 2 * f->fc_groups);
  if ((rc = dalloc(c, &f->pb_flag, (size_t)f->n_clear + 2 * f->fc_tiles))) return rc;
  f->ab_flag = f->pb_flag + f->nblocks;
  f->rcnt = f->pb_flag + 2 * f->nblocks;
  f->fc_gsum = f->rcnt + 64;
  f->fc_tcount = f->fc_gsum + 2 * f->fc_groups;
  if ((rc = dalloc(c, &f->pb_index, f->nblocks))) return rc;
  if ((rc = dalloc(c, &f->ab_index, f->nblocks))) return rc;
  f->g.ab_flag = f->ab_flag;
  if (const char *e = getenv("MPMHIP_DBG")) f->g.dbg = MPMHIP_DEBUG ? (int)strtoul(e, nullptr, 0) : ((int)strtoul(e, nullptr, 0) & 64);
  if (const char *e = getenv("MPMHIP_FUSE_GRID")) f->fuse_grid = atoi(e) != 0;
  f->g2p_two_pass = cfg.n_particles - cfg.n_elements - cfg.n_vertices == 0;
  if (const char *e = getenv("MPMHIP_G2P_TWO_PASS")) f->g2p_two_pass = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_FUSE_TRAD")) f->fuse_trad = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_DIST_FUSED_HALO")) f->fused_want = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_SPLAT_FIRST_MAX")) f->splat_first_max = atoi(e);
  // p2g's first round staggered by wave slot -- 2 x 1024 cycles per step, 5 groups, the first 1,280 workgroups -- where the launch is
  // more than two rounds of workgroups (decided per launch in step_phase_a): the headline scene +1.1 % at t = 0 and draped (round 4,
  // three alternating runs: 16.06-16.08 k -> 16.21-16.29 k; round 3 had measured the same and left it off because a launch of less
  // than one round loses up to 20 %).  MPMHIP_P2G_STAGGER="units[,groups[,first]]" forces a setting for every launch, "0" none.
  f->g.stagger = 0; f->g.stagger_groups = 5; f->g.stagger_first = 1280;
  f->stagger_auto = 2;
  if (const char *e = getenv("MPMHIP_P2G_STAGGER")) {
    int u = 0, gr = 2, first = 1280;
    sscanf(e, "%d,%d,%d", &u, &gr, &first);
    f->g.stagger = std::max(0, u); f->g.stagger_groups = std::max(1, gr); f->g.stagger_first = std::max(0, first);
    f->stagger_auto = -1;  // forced
  }
  if (const char *e = getenv("MPMHIP_G2P_MFLAG")) f->g2p_mflag = atoi(e) != 0;
  MPM_HIP_CHECK(c, hipHostMalloc((void **)&f->h_pin, 64 * sizeof(int), hipHostMallocDefault));
  MPM_HIP_CHECK(c, hipEventCreateWithFlags(&f->ev_flag, hipEventDisableTiming));
  {
    int *hs = nullptr, *ds = nullptr;
    MPM_HIP_CHECK(c, hipHostMalloc((void **)&hs, SIG_WORDS * sizeof(int), hipHostMallocMapped));
    memset(hs, 0, SIG_WORDS * sizeof(int));
    MPM_HIP_CHECK(c, hipHostGetDevicePointer((void **)&ds, hs, 0));
    f->h_sig = hs;
    f->g.host_sig = getenv("MPMHIP_FLAG_COPY") ? nullptr : ds;  // MPMHIP_FLAG_COPY=1: the former copy + event poll (A/B)
    if (const char *e = getenv("MPMHIP_HOST_LEAD")) f->host_lead = std::min(std::max(1, atoi(e)), 12);  // (ring of 16 entries)
    f->g.lookahead = DRIFT_LOOKAHEAD;
    if (const char *e = getenv("MPMHIP_DRIFT_LOOKAHEAD")) f->g.lookahead = (float)atof(e);
  }
  return MPMHIP_OK;
}

void fast_destroy(mpmhip_ctx *c) {
  FastState *f = c->fast;
  if (!f) return;
  for (auto &p : f->rpeers) {
    if (p.link_remote) (void)hipIpcCloseMemHandle(p.link_remote);
    if (p.link_local) (void)hipFree(p.link_local);
  }
  if (f->rccl.comm) (void)f->rccl.CommDestroy(f->rccl.comm);
  for (void *p : f->allocs) (void)hipFree(p);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  if (f->h_sig) (void)hipHostFree((void *)f->h_sig);
  f->h_ranges.release(); f->h_plist.release(); f->h_chunks.release(); f->h_chunks_g.release();
  if (f->ev_flag) (void)hipEventDestroy(f->ev_flag);
  delete f;
  c->fast = nullptr;
}

int fast_add_collider_storage(mpmhip_ctx *c, MeshCollider &mc) {
  FastState *f = c->fast;
  if (!c->colliders.empty()) {  // same body mesh, same splat: the extra collider only adds a collide step with its friction
    mc.weight = c->colliders[0].weight;
    return MPMHIP_OK;
  }
  int rc, nf = c->num_mesh_f;
  for (int i = 0; i < 2; ++i)
    if ((rc = dalloc(c, &f->fkeys[i], (size_t)nf))) return rc;
  if ((rc = dalloc(c, &f->forder, (size_t)nf))) return rc;
  if ((rc = dalloc(c, &f->fidx, (size_t)3 * nf))) return rc;
  if ((rc = dalloc(c, &f->fiota, (size_t)nf))) return rc;
  if ((rc = dalloc(c, &f->fb_start, f->nblocks))) return rc;
  if ((rc = dalloc(c, &f->fb_cnt, f->nblocks))) return rc;
  for (int i = 0; i < f->nbuf; ++i)
    if ((rc = dalloc(c, &f->col2[i], f->nblocks * GCH_COL * 64))) return rc;
  select_buffer(f, f->par);
  mc.weight = f->col2[0];
  return MPMHIP_OK;
}

int fast_add_mover_storage(mpmhip_ctx *c, Mover &mv) {
  FastState *f = c->fast;
  if (!c->movers.empty()) {  // every mover is handed the same joint velocities (mpm_solver.py:421-481) and OVERWRITES the
    mv.weight = c->movers[0].weight;  // touched nodes: a second one repeats the first one's result exactly
    return MPMHIP_OK;
  }
  int rc = MPMHIP_OK;
  for (int i = 0; i < f->nbuf && !rc; ++i) rc = dalloc(c, &f->mov2[i], f->nblocks * GCH_MOV * 64);
  select_buffer(f, f->par);
  mv.weight = f->mov2[0];
  return rc;
}

// One substep = three phases; the multi-GPU driver interleaves its exchanges between them:
//   A: [re-sort] pre-ops, body/joint splats (side stream), stress, p2g          -> halo exchange of shared blocks
//   B: grid stage, g2p (+ escaped queue)                                         -> ghost x/v/d3 exchange
//   C: element finalise, drift-flag bookkeeping
static void grid_stage_params(mpmhip_ctx *c, const StepArgs &a, GridParams &gp, BCList &bcl);

// G2P2G applies to scenes of traditional particles only, in the production loop of one GPU
static bool g2p2g_ok(const mpmhip_ctx *c) {
  const FastState *f = c->fast;
  const Dims &d = f->d;
  // ... and, as the kernel stands, where one round of workgroups holds the whole scene: hipcc gives the fused kernel 212-227 VGPRs
  // (two wavefronts per SIMD = 512 workgroup slots; either half alone needs 116-124, profiles/r04_experiments.md), which a scene
  // of more chunks pays for with more than it saves (block-512k -8 %, garment-120k-iso -11 %; cube-8k +23 %)
  return f->g2p2g && f->nbuf == 3 && !f->dist && !c->profiling && d.n_e == 0 && d.n_v == 0 && d.n_t > 0 && f->fuse_trad && f->fuse_grid &&
         !f->g.halo.slot && !f->g2p_mflag && !(MPMHIP_DEBUG && f->g.dbg) && f->n_chunks > 0 && f->n_chunks <= f->g2p2g_max_chunks;
}
// the deferred g2p of the last substep as a launch of its own (anything that reads or re-orders the particles comes here first),
// and the clearing of the buffer the last fused launch read
int flush_g2p(mpmhip_ctx *c) {
  FastState *f = c->fast;
  if (f->g2p_pending) {
    if (f->n_chunks_g) {
      ScopedPhase ph(c, "g2p_v");
      launch_g2p(c, true, f->g2p_two_pass, f->pend_dt, f->pend_gp, f->pend_bcl);
    }
    f->g2p_pending = false;
  }
  if (f->clear_later >= 0) {
    const int k = f->clear_later;
    ZeroArgs z{f->alist, f->n_A, 0, f->cl_col, f->cl_mov, f->mv2[k], f->col2[k], f->mov2[k], f->mflag2[k], f->cflag2[k]};
    if (f->n_A) {
      z.n_wg = (f->n_A + PT / 64 - 1) / (PT / 64);
      hipLaunchKernelGGL(k_zero_blocks, (unsigned)z.n_wg, PT, 0, c->stream, z);
    }
    f->clear_later = -1;
  }
  return MPMHIP_OK;
}

// Look at the drift flags the device has posted (ring in pinned host memory, or the copied flag of contexts without one) and turn a
// raised flag into "re-sort before the next substep" (steps_since_rebin = 1 << 30).  Called once per substep, at the head of
// step_phase_a.
static int poll_drift_flags(mpmhip_ctx *c) {
  FastState *f = c->fast;
  hipStream_t s = c->stream;
  {
    if (f->g.host_sig) {
      // the kernels report progress and their flags into pinned host memory (k_p2g): no stream operation here.  Before
      // substep s the host waits until substep s - host_lead has started and decides with THAT substep's entry: it keeps
      // host_lead substeps queued (enough to hide its launch latency) and no more, and a warning takes effect exactly
      // host_lead substeps after the launch that posted it (the copy + event scheme: 8-16) -- inside the look-ahead.
      const unsigned e = f->sig_seq + 1u - (unsigned)f->host_lead;  // the substep whose ring entry decides now
      bool arrived = true;
      for (long spins = 0; (int)((unsigned)f->h_sig[SIG_PROGRESS] - e) < 0; ++spins) {
        if ((spins & 0x3ff) == 0x3ff) {
          hipError_t q = hipStreamQuery(s);
          if (q == hipSuccess) { arrived = (int)((unsigned)f->h_sig[SIG_PROGRESS] - e) >= 0; break; }  // nothing in flight any more
          if (q != hipErrorNotReady) MPM_HIP_CHECK(c, q);
        }
        std::this_thread::yield();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (arrived && (int)(e - f->sig_at_rebin) > 0) {  // an entry written after the last re-sort
        unsigned v = (unsigned)f->h_sig[SIG_RING0 + (e & (unsigned)(SIG_RING_N - 1))];
        if ((v >> 2) == (e & 0x3fffffffu)) {
          if (v & 2u) f->face_flag_seen = true;
          if ((v & 1u) && f->adaptive_rebin) f->steps_since_rebin = 1 << 30;
        }
      }
    } else if (f->flag_pending && (f->steps_since_rebin & f->poll_mask) == 0) {
      MPM_HIP_CHECK(c, hipEventSynchronize(f->ev_flag));
      f->flag_pending = false;
      if (f->h_pin[24] && f->adaptive_rebin) f->steps_since_rebin = 1 << 30;
    }
  }
  return MPMHIP_OK;
}

int step_phase_a(mpmhip_ctx *c, const StepArgs &a) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  hipStream_t s = c->stream;
  int rc;
  (void)rc; (void)d; (void)s;
  if (c->caller_dirty) {
    if (f->dist) return fail(c, MPMHIP_ERR_STATE, "dist mode: call mpmhip_dist_rebin after (re)binding the state");
    if ((rc = do_import(c))) return rc;
  }
  const float dt = a.dt;
  if (!f->col_at_rest) drop_kept_collider_fields(c);   // (fields kept by an mpmhip_steps call whose body was at rest: this body may move)
  // pre-p2g particle operations, mpm_solver.py:260-279 (impulses first, then velocity modifiers)
  if (f->g2p_pending && (!g2p2g_ok(c) || !c->pre.empty() || dt != f->pend_dt)) flush_g2p(c);
  if (!c->pre.empty() && d.n_p) {
    flush_elements(c);
    float t = (float)c->time;
    for (int pass = 0; pass < 2; ++pass)
      for (auto &op : c->pre) {
        bool imp = op.type == PRE_IMPULSE || op.type == PRE_IMPULSE_MASK;
        if (imp != (pass == 0) || !(t >= op.start_time && t < op.end_time)) continue;
        hipLaunchKernelGGL(k_pre_sorted, nblk(d.n_p), TPB, 0, s, op, f->buf[f->cur], f->perm[f->cur], d, dt);
      }
  }
  if (!f->dist) {
    // Drift flag raised by g2p / element finalise / collider splat.  It is copied back every 8 substeps; before
    // the next copy is issued the host waits for the previous one, which also bounds how far the host may run
    // ahead of the GPU (<= 16 substeps) -- otherwise a fused mpmhip_steps(n) would have enqueued all n substeps
    // long before the first flag arrives.
    if ((rc = poll_drift_flags(c))) return rc;
    if (f->steps_since_rebin >= f->rebin_interval) {
      ScopedPhase ph(c, "rebin");
      // predictive sort: aim at the middle of the next interval, estimated from the one that just ended
      if (f->true_since_rebin > 0) {
        f->lead_steps = std::min(std::max(0.5f * (float)f->true_since_rebin, 4.0f), 48.0f);
        // fast material (short intervals): look at the flag more often, so that the host's lag stays well inside the
        // 20-substep early warning and nothing outruns the active blocks
        f->poll_mask = f->true_since_rebin <= 24 ? 1 : (f->true_since_rebin <= 48 ? 3 : 7);
      }
      f->last_dt = dt;
      if ((rc = rebin(c))) return rc;
      f->true_since_rebin = 0;
    }
  }
  f->last_dt = dt;
  f->true_since_rebin += 1;
  // The body-face and joint splats ride along in the p2g launch as extra workgroups (SplatArgs).  With profiling on
  // (one sync per phase, like the reference's ScopedTimer) they get a launch of their own under the reference's
  // phase names: the same kernel with no particle chunks.
  bool has_col = !c->colliders.empty() && c->num_mesh_f && f->n_fbins;
  bool mov_on = a.joint_v_v && a.joint_f_v && !c->movers.empty();
  // traditional particles: their stress update runs at the front of p2g (k_p2g<.., true, ..>) unless profiling wants
  // the reference's phases apart
  const bool trad_fused = d.n_t > 0 && f->fuse_trad && !c->profiling;
  const TradParams tp{c->sc.material, c->sc.alpha, c->sc.hardening, c->sc.xi, c->sc.plastic_viscosity, c->sc.softening};
  bool jt_tile = false;
  SplatArgs sa{};
  SplatArgs none{};
  none.z_first = 1 << 30;
  if (has_col) {
    sa.pts = c->cur_pts; sa.vel = c->cur_vel; sa.adv = c->cur_f; sa.fidx = f->fidx; sa.fbins = f->fbins;
    sa.n_fbins = f->n_fbins;
  }
  if (mov_on) {
    sa.js = JointSplatArgs{a.joint_t_v, a.joint_v_v, a.joint_f_v, (a.joint_t_v ? a.n_joint_t : 0), c->cfg.num_joint_v,
                           c->cfg.num_joint_f, d.n_nv - a.n_joint_t, d.n_nv, f->inv, f->perm[f->cur], 0};
    // many held traditional particles: splatted through the p2g tiles (needs the fused-stress kernel variant)
    jt_tile = trad_fused && sa.js.n_t >= 2048;
    sa.js.t_in_tile = jt_tile ? 1 : 0;
    int nj = (jt_tile ? 0 : sa.js.n_t) + sa.js.n_v + sa.js.n_f;
    sa.n_mov_wg = (int)(((size_t)nj * 32 + PT - 1) / PT);
    if (nj == 0) sa.n_mov_wg = 0;
  }
  // accumulators left loaded by the previous (fused) substep: this substep scatters into the other buffer and clears
  // the loaded one with extra workgroups of the p2g launch
  const bool do_g2p2g = f->g2p_pending && !jt_tile;  // (pending survives a re-sort decision above only when none happened)
  if (f->g2p_pending && !do_g2p2g) flush_g2p(c);
  GridRead rd{f->g.mv, f->g.col, f->g.mov, f->g.col_flag};  // (fused launch) the buffer the deferred g2p reads: the current one
  if (do_g2p2g) {
    // rotate: read R = current, scatter into W = the next, clear Z = what the fused launch before this one read
    const int R = f->par, Z = f->clear_later;
    sa.z = ZeroArgs{f->alist, f->n_A, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (Z >= 0 && f->n_A) {
      sa.z = ZeroArgs{f->alist, f->n_A, (f->n_A + PT / 64 - 1) / (PT / 64), f->cl_col, f->cl_mov, f->mv2[Z], f->col2[Z], f->mov2[Z],
                      f->mflag2[Z], f->cflag2[Z]};
    }
    f->clear_later = R; f->cl_col = f->dirty_col; f->cl_mov = f->dirty_mov;
    f->grid_dirty = false;
    select_buffer(f, (R + 1) % 3);
  } else
  sa.z = take_zero(f);
  if (sa.z.n_wg && !do_g2p2g) {
    if (c->profiling) {  // profiling runs keep one launch per reference phase: clear now
      hipLaunchKernelGGL(k_zero_blocks, (unsigned)sa.z.n_wg, PT, 0, s, sa.z);
      sa.z.n_wg = 0;
    } else {
      select_buffer(f, (f->par + 1) % f->nbuf);
    }
  }
  // a body at rest whose collider field this buffer already holds (col_state 2): no splat workgroups in this substep
  bool col_kept = false;
  if (has_col && f->col_at_rest && f->nbuf == 2 && !c->profiling && f->col_state[f->par] == 2) {
    sa.n_fbins = 0;
    col_kept = true;
    f->n_col_kept += 1;
  }
  // cloth scenes of the production loop: the splat's first pass rides in front of the stress launch (col_splat_wg)
  // ... where the p2g launch is at most one round of workgroups, i.e. as long as a workgroup's life is the launch's length
  // (garment-120k-aniso: stress 9.7 -> 12.0 us, p2g 20.0 -> 16.4 us, 24.7 k -> 25.6 k substeps/s; with several rounds of chunk
  // workgroups the splat hides among them and the split only lengthens the stress launch: sheet-500k -1 %, profiles/r04_experiments.md)
  const bool split_splat = f->split_splat && has_col && sa.n_fbins > 0 && d.n_e > 0 && f->elem_pending && !c->profiling && !f->dist &&
                           f->n_chunks <= f->split_splat_max_chunks && !(MPMHIP_DEBUG && f->g.dbg);
  sa.splat_passes = split_splat ? 2 : 3;
  sa.n_extra = (sa.n_fbins + sa.n_mov_wg + 7) & ~7;
  sa.z_first = sa.n_extra + (int)xcd_grid(f->n_chunks);
  sa.e0 = (sa.n_extra > f->splat_first_max && !c->profiling) ? (int)xcd_grid(f->n_chunks) : 0;
  const bool fused_halo_now = f->dist && f->fused_halo && !c->profiling && !c->prof_fused;
  if (fused_halo_now) {  // (multi-GPU) the halo pack rides in this launch, behind the clearing workgroups: see PackArgs
    HaloTab &tb = sa.pack.tb;
    tb.with_mov = c->movers.empty() ? 0 : 1;
    const int CH = tb.with_mov ? 8 : 4, par = (int)(f->halo_seq & 1u);
    for (auto &q : f->peers) {
      if (!q.n_blocks) continue;
      int k = tb.n++;
      tb.blocks[k] = q.blocks; tb.n_blocks[k] = q.n_blocks;
      tb.buf[k] = q.link_remote + LINK_DATA0 + (size_t)par * q.link_cap * 8 * 64;
      tb.sig[k] = (int *)q.link_remote + par * LINK_FLAG_STRIDE;
      tb.cnt[k] = q.link_cnt;
      tb.wg_off[k + 1] = tb.wg_off[k] + (int)(((size_t)q.n_blocks * CH * 64 + PT - 1) / PT);
    }
    tb.seq = (int)f->halo_seq;
    sa.pack.n_wg = tb.wg_off[tb.n];
    sa.pack.first = sa.z_first + sa.z.n_wg;
    sa.pack.count = 1;
    sa.pack.done = f->pack_done;
    f->pack_target += (unsigned)sa.z_first;  // every workgroup in front of the clearing ones counts itself done
    sa.pack.target = f->pack_target;
  }
  f->g.step_id = (int)++f->sig_seq;
  if (f->stagger_auto >= 0) f->g.stagger = (f->n_chunks >= 2 * 1280 && !c->profiling) ? f->stagger_auto : 0;
  const bool stress_here = d.n_e > 0;
  if (stress_here || (d.n_t && !trad_fused)) {  // (no empty event bracket when the stress update rides in p2g)
    ScopedPhase ph(c, "compute_stress_from_F_trial");
    if (stress_here) {
      launch_stress_elem(c, split_splat ? 2 : (f->elem_pending ? 1 : 0), sa);
      f->elem_pending = false;
    }
    if (d.n_t && !trad_fused) launch_stress_trad(c, dt);
  }
  if (c->profiling && sa.n_extra) {
    {
      ScopedPhase ph(c, "p2g");
      if (f->n_chunks)
        launch_p2g(c, false, false, xcd_grid(f->n_chunks), f->n_chunks, dt, none, tp);
    }
    if (sa.n_fbins) {
      ScopedPhase ph(c, "apply_Mesh_Collision_on_grid");
      SplatArgs only = sa;
      only.n_mov_wg = 0;
      only.n_extra = (only.n_fbins + 7) & ~7;
      only.z_first = 1 << 30;
      launch_p2g(c, false, false, (unsigned)only.n_extra, 0, dt, only, tp);
    }
    if (sa.n_mov_wg) {
      ScopedPhase ph(c, "apply_Particle_Moving_on_grid");
      SplatArgs only = sa;
      only.n_fbins = 0;
      only.n_extra = (only.n_mov_wg + 7) & ~7;
      only.z_first = 1 << 30;
      launch_p2g(c, false, false, (unsigned)only.n_extra, 0, dt, only, tp);
    }
  } else if (do_g2p2g) {
    ScopedPhase ph(c, "g2p2g");
    const unsigned grid = xcd_grid(f->n_chunks) + (unsigned)(sa.n_extra + sa.z.n_wg);
    launch_g2p2g(c, grid, dt, rd, sa, tp, f->pend_gp, f->pend_bcl);
    f->g2p_pending = false;
    f->n_g2p2g += 1;
  } else {
    ScopedPhase ph(c, "p2g");
    if (f->n_chunks || sa.n_extra || sa.z.n_wg || sa.pack.n_wg)
      launch_p2g(c, trad_fused, jt_tile, xcd_grid(f->n_chunks) + (unsigned)(sa.n_extra + sa.z.n_wg + sa.pack.n_wg), f->n_chunks, dt, sa, tp);
  }
  if (has_col && !col_kept && f->nbuf == 2) f->col_state[f->par] = (f->col_at_rest && !c->profiling) ? 2 : 1;
  return MPMHIP_OK;
}

// grid-stage parameters of this substep (the BC list as it stands BEFORE this substep's bc_host_modify)
static void grid_stage_params(mpmhip_ctx *c, const StepArgs &a, GridParams &gp, BCList &bcl) {
  const bool mov_on = a.joint_v_v && a.joint_f_v && !c->movers.empty();
  gp = GridParams{a.dt, c->sc.g[0], c->sc.g[1], c->sc.g[2], c->sc.grid_v_damping_scale, (float)c->time,
                  (c->colliders.empty() || !c->num_mesh_f) ? 0 : 1, c->movers.empty() ? 0 : 1, mov_on ? 1 : 0,
                  c->colliders.empty() ? 0.0f : c->colliders[0].friction, 0};
  gp.n_col_more = std::max(0, std::min(3, (int)c->colliders.size() - 1));
  for (int k = 0; k < gp.n_col_more; ++k) gp.col_friction_more[k] = c->colliders[k + 1].friction;
  bcl = BCList{};
  bcl.n = (int)c->bcs.size();
  for (int k = 0; k < bcl.n; ++k) bcl.bc[k] = c->bcs[k];
}

int step_phase_b(mpmhip_ctx *c, const StepArgs &a) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  hipStream_t s = c->stream;
  int rc;
  (void)rc; (void)d; (void)s;
  const float dt = a.dt;
  GridParams gp;
  BCList bcl;
  grid_stage_params(c, a, gp, bcl);
  const bool fused = f->fuse_grid && !c->profiling;
  if (!fused) {
    ScopedPhase ph(c, "grid_update");
    gp.count = 1;
    f->stat_steps += 1;
    if (f->n_A)
      hipLaunchKernelGGL(k_grid<true>, xcd_grid((f->n_A + 3) / 4), TPB, 0, s, f->alist, f->n_A, d, f->g, gp, bcl);
  }
  for (auto &bc : c->bcs) bc_host_modify(bc, (float)c->time, dt);
  if (fused && g2p2g_ok(c)) {  // G2P2G: the next substep's launch does this g2p in front of its p2g (or flush_g2p does)
    f->g2p_pending = true;
    f->pend_gp = gp; f->pend_bcl = bcl; f->pend_dt = dt;
  } else {
    ScopedPhase ph(c, "g2p_v");
    if (f->n_chunks_g) launch_g2p(c, fused, f->g2p_two_pass, dt, gp, bcl);
  }
  if (fused) {
    f->grid_dirty = true;
    f->dirty_col = gp.has_col;
    f->dirty_mov = gp.has_mov && gp.mov_on;
    f->last_gp = gp;
    f->last_bcl = bcl;
  }
  return MPMHIP_OK;
}

int step_phase_c(mpmhip_ctx *c, const StepArgs &a) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  hipStream_t s = c->stream;
  int rc;
  (void)rc; (void)d; (void)s;
  // unprofiled: the element finalize is deferred into the next substep's stress kernel (k_stress_elem<true>); multi-GPU
  // ranks have unpacked their ghost vertices by now, so the same holds there
  f->elem_pending = d.n_e > 0;
  if (c->profiling || (f->g.dbg & 64)) {
    ScopedPhase ph(c, "g2p_e");
    flush_elements(c);
  }
  f->steps_since_rebin += 1;
  if (!f->dist && !f->g.host_sig && !f->flag_pending && (f->steps_since_rebin & f->poll_mask) == 0) {
    MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 24, f->g.counters + 6, sizeof(int), hipMemcpyDeviceToHost, s));
    MPM_HIP_CHECK(c, hipEventRecord(f->ev_flag, s));
    f->flag_pending = true;
  }
  c->internal_dirty = true;
  MPM_HIP_CHECK(c, hipGetLastError());
  return MPMHIP_OK;
}


int fast_step(mpmhip_ctx *c, const StepArgs &a) {
  int rc;
  if (c->prof_fused && !c->profiling) {
    // What a bracket costs by itself.  Two empty event records measure the wrong thing (4.8 us: the command processor's time per
    // event packet; with a kernel between them most of that overlaps the kernel).  So: one bracket around ONE null kernel (B1) and
    // one around TWO (B2).  B2 - B1 is a null kernel, 2 B1 - B2 what the bracket adds around a launch; bench.py reports both.
    { ScopedPhase ph(c, "event_null1"); hipLaunchKernelGGL(k_null, 1, 64, 0, c->stream); }
    { ScopedPhase ph(c, "event_null2"); hipLaunchKernelGGL(k_null, 1, 64, 0, c->stream); hipLaunchKernelGGL(k_null, 1, 64, 0, c->stream); }
  }
  if ((rc = step_phase_a(c, a))) return rc;
  if ((rc = step_phase_b(c, a))) return rc;
  return step_phase_c(c, a);
}

int fast_export_grid(mpmhip_ctx *c, float *m, float *v_in, float *v_out) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  size_t n = G3(c);
  if (m) MPM_HIP_CHECK(c, hipMemsetAsync(m, 0, n * sizeof(float), c->stream));
  if (v_in) MPM_HIP_CHECK(c, hipMemsetAsync(v_in, 0, 3 * n * sizeof(float), c->stream));  // consumed by the grid stage
  if (v_out) MPM_HIP_CHECK(c, hipMemsetAsync(v_out, 0, 3 * n * sizeof(float), c->stream));
  materialize_grid(c, false);
  if (f->n_A) hipLaunchKernelGGL(k_export_grid, (unsigned)((f->n_A + 3) / 4), TPB, 0, c->stream, f->alist, f->n_A, d, f->g, m, v_out);
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return MPMHIP_OK;
}

int fast_set_debug_flags(mpmhip_ctx *c, int flags) {
  // bit 64 (stand-alone element finalize every substep; results stay right) is a host-side switch and exists in every build
  if (!MPMHIP_DEBUG && (flags & ~64))
    return fail(c, MPMHIP_ERR_INVALID, "set_debug_flags: this build carries no kernel ablation switches (build a variant with "
                                       "-DMPMHIP_DEBUG=1, tools/build_variants.py, and select it with MPMHIP_LIB)");
  c->fast->g.dbg = flags;
  return MPMHIP_OK;
}
// Per-workgroup timeline of the p2g (kernel 0) and g2p (kernel 1) launches, MPMHIP_DEBUG builds only.  out == nullptr: start
// recording (stamps of earlier launches are cleared); otherwise copy the stamps of the newest launch of `kernel`:
// out[wg * 8 + slot], slots as placed by WGT() in the kernels, in ticks of the 100 MHz constant clock.
int fast_debug_wgtrace(mpmhip_ctx *c, int kernel, uint64_t *out, int max_wg) {
  FastState *f = c->fast;
  if (!MPMHIP_DEBUG) return fail(c, MPMHIP_ERR_INVALID, "debug_wgtrace: needs a -DMPMHIP_DEBUG=1 build of the library");
  const size_t total = (size_t)WGT_KERNELS * WGT_MAX_WG * WGT_SLOTS;
  if (!out) {
    if (!f->g.trace) {
      int rc = dalloc(c, &f->g.trace, total);
      if (rc) return rc;
    }
    MPM_HIP_CHECK(c, hipMemsetAsync(f->g.trace, 0, total * sizeof(unsigned long long), c->stream));
    return MPMHIP_OK;
  }
  if (kernel < 0 || kernel >= WGT_KERNELS || max_wg <= 0 || !f->g.trace) return fail(c, MPMHIP_ERR_INVALID, "debug_wgtrace: bad arguments");
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  size_t n = (size_t)std::min(max_wg, WGT_MAX_WG) * WGT_SLOTS;
  MPM_HIP_CHECK(c, hipMemcpy(out, f->g.trace + (size_t)kernel * WGT_MAX_WG * WGT_SLOTS, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}
int fast_debug_counter(mpmhip_ctx *c, int index, int64_t *out) {
  if (index < 0 || index >= 16 || !out) return fail(c, MPMHIP_ERR_INVALID, "debug_counter: index out of range");
  int v = 0;
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  MPM_HIP_CHECK(c, hipMemcpy(&v, c->fast->g.counters + index, sizeof(int), hipMemcpyDeviceToHost));
  *out = v;
  return MPMHIP_OK;
}

int fast_stats(mpmhip_ctx *c, mpmhip_stats *out) {
  FastState *f = c->fast;
  flush_g2p(c);  // (a pending g2p counts its out-of-margin particles too)
  out->rebins = f->rebins;
  out->g2p2g_launches = f->n_g2p2g;
  out->p2g_tile_in_use = f->p2g_fixed_now ? MPMHIP_P2G_TILE_FIXED : MPMHIP_P2G_TILE_F64;
  out->kept_collider_substeps = (int32_t)std::min<int64_t>(f->n_col_kept, 0x7fffffff);
  out->n_active_blocks = f->n_A;
  int *dcnt = f->g.counters + 4;
  if (f->grid_dirty) {  // fused substeps do not count collider / mover nodes: count the last substep now
    MPM_HIP_CHECK(c, hipMemsetAsync(f->g.counters + 2, 0, 2 * sizeof(int), c->stream));
    materialize_grid(c, true);
    f->stat_steps = 1;
  }
  MPM_HIP_CHECK(c, hipMemsetAsync(dcnt, 0, sizeof(int), c->stream));
  if (f->n_A) hipLaunchKernelGGL(k_count_active, (unsigned)((f->n_A + 3) / 4), TPB, 0, c->stream, f->alist, f->n_A, f->g, dcnt);
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 8, f->g.counters, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  out->n_fallback_particles = f->h_pin[8];
  out->n_dropped = f->h_pin[9];  // contributions outside the active blocks (must stay 0)
  out->n_active_nodes = f->h_pin[12];
  if (f->stat_steps > 0) {  // per-substep averages since the previous call
    out->n_collider_nodes = (int)(f->h_pin[10] / f->stat_steps);
    out->n_mover_nodes = (int)(f->h_pin[11] / f->stat_steps);
  }
  MPM_HIP_CHECK(c, hipMemsetAsync(f->g.counters + 2, 0, 2 * sizeof(int), c->stream));
  f->stat_steps = 0;
  return MPMHIP_OK;
}


}  // namespace mpm
