// batch.hip -- one launch per phase for SEVERAL solver contexts (mpmhip_steps_multi; round 5).
// The caller's finite-difference training step (train_material_params.py:583-631) runs four independent simulations of the same garment.
// As four contexts on four streams they are 12 launches per joint substep, each of less than one round of workgroups (724 chunks on
// 1,280 slots), and the kernel trace shows the GPU without a hot kernel for a quarter of the time.  Here the contexts are stepped in
// lock step on ONE stream and every phase is ONE launch: workgroups [first[i], first[i + 1]) run context i's chunk list with context i's
// arguments -- the same device code (stress_elem_body, p2g_body, g2p_body), the same results bit for bit.
// Host side: the launchers of the three hot kernels (p2g.hip, g2p.hip) record instead of launching while FastState::batching is set;
// everything else a phase issues (re-sorts, flushes, clears) goes to the shared stream at once, which keeps every context's own order.
#include "fast_state.hpp"

namespace mpm {

namespace {

template <class T>
__device__ __forceinline__ int batch_entry(const Batch<T> &B, int bid) {  // (uniform)
  int c = 0;
  if (B.n > 1 && bid >= B.first[1]) c = 1;
  if (B.n > 2 && bid >= B.first[2]) c = 2;
  if (B.n > 3 && bid >= B.first[3]) c = 3;
  return c;
}

__global__ __launch_bounds__(TPB) void k_stress_elem_b(Batch<StressB> B) {
  __shared__ double tile[4 * TILE_PAD];
  const int c = batch_entry(B, (int)blockIdx.x);
  const StressB &a = B.a[c];
  const int bid = (int)blockIdx.x - B.first[c];
  if (bid >= a.grid) return;
  const GridPtrs g = expand(a.g);
  if (bid < a.n_splat) {
    const SplatArgs sa = expand(a.sa);
    col_splat_wg<1>(tile, sa, bid, a.d, g);
    return;
  }
  stress_elem_body<true>((bid - a.n_splat) * (int)blockDim.x + (int)threadIdx.x, a.b, a.ef, a.d, a.friction_coeff, a.face_slot, a.skeys, a.blk_bits,
                         g.counters, g.step_id);
}

__global__ __launch_bounds__(PT) void k_p2g_b(Batch<P2GB> B) {
  __shared__ double tile[P2G_TILE_DOUBLES];
  __shared__ int esc[CHUNK];
  __shared__ int esc_n;
  __shared__ float red[8];
  const int c = batch_entry(B, (int)blockIdx.x);
  const P2GB &a = B.a[c];
  const int bid = (int)blockIdx.x - B.first[c];
  if (bid >= a.grid) return;
  const GridPtrs g = expand(a.g);
  const SplatArgs sa = expand(a.sa);
  p2g_body<P2G_STEPS, false, false, true>(a.recs, a.n_chunks, a.b, a.va, a.d, a.rpic, a.dt, g, sa, TradParams{}, tile, esc, esc_n, red, bid);
}

__global__ __launch_bounds__(PT) void k_g2p_b(Batch<G2PB> B) {
  __shared__ float4 tile[TILE_PAD];
  const int c = batch_entry(B, (int)blockIdx.x);
  const G2PB &a = B.a[c];
  const int bid = (int)blockIdx.x - B.first[c];
  if (bid >= a.grid) return;
  const GridPtrs g = expand(a.g);
  g2p_body<true, true, false, false, false>(a.recs, a.n_chunks, a.b, a.d, a.dt, g, a.gp, *a.bcl, tile, bid);
}

inline int pad8(int n) { return (n + 7) & ~7; }

// one batched launch per kind over the contexts that have one recorded
void flush_kind(mpmhip_ctx **cs, int nc, int kind) {
  hipStream_t s = cs[0]->stream;
  for (int i0 = 0; i0 < nc;) {
    Batch<StressB> bs{};
    Batch<P2GB> bp{};
    Batch<G2PB> bg{};
    int n = 0, total = 0, i = i0;
    for (; i < nc && n < BATCH_MAX; ++i) {
      FastState *f = cs[i]->fast;
      if (kind == 0 && f->pend_stress) { bs.a[n] = f->ps; bs.first[n] = total; total += pad8(f->ps.grid); ++n; f->pend_stress = false; }
      if (kind == 1 && f->pend_p2g) { bp.a[n] = f->pp; bp.first[n] = total; total += pad8(f->pp.grid); ++n; f->pend_p2g = false; }
      if (kind == 2 && f->pend_g2p) { bg.a[n] = f->pg; bg.first[n] = total; total += pad8(f->pg.grid); ++n; f->pend_g2p = false; }
    }
    i0 = i;
    if (!n || !total) continue;
    if (kind == 0) { bs.n = n; bs.first[n] = total; hipLaunchKernelGGL(k_stress_elem_b, (unsigned)total, TPB, 0, s, bs); }
    if (kind == 1) { bp.n = n; bp.first[n] = total; hipLaunchKernelGGL(k_p2g_b, (unsigned)total, PT, 0, s, bp); }
    if (kind == 2) { bg.n = n; bg.first[n] = total; hipLaunchKernelGGL(k_g2p_b, (unsigned)total, PT, 0, s, bg); }
  }
}

}  // namespace

bool fast_batch_single(const mpmhip_ctx *c) { return c->fast && c->fast->batch_single; }

int batch_flush_ctx(mpmhip_ctx *c) {
  mpmhip_ctx *one[1] = {c};
  flush_kind(one, 1, 0);
  flush_kind(one, 1, 1);
  flush_kind(one, 1, 2);
  return MPMHIP_OK;
}

// n substeps of nc contexts in lock step (base[i]: context i's arguments of the fused call, as mpmhip_steps takes them)
int fast_steps_multi(mpmhip_ctx **cs, int nc, const StepArgs *base, int n) {
  int rc = MPMHIP_OK;
  for (int i = 0; i < nc; ++i) cs[i]->fast->batching = true;
  for (int k = 0; k < n && rc == MPMHIP_OK; ++k) {
    std::vector<StepArgs> a(base, base + nc);
    for (int i = 0; i < nc; ++i) {
      mpmhip_ctx *c = cs[i];
      a[i].mesh_f = (float)((double)a[i].dt * (double)k);   // mesh_x + substep_size * k * mesh_v, train_material_params.py:623
      a[i].mesh_store = k == n - 1;
      a[i].more = k < n - 1;
      c->cur_pts = a[i].mesh_x ? a[i].mesh_x : c->mesh_points;
      c->cur_vel = a[i].mesh_v ? a[i].mesh_v : c->mesh_vel;
      c->cur_f = (a[i].mesh_x && a[i].mesh_v) ? a[i].mesh_f : 0.0f;
      if ((rc = step_phase_a(c, a[i]))) break;
    }
    flush_kind(cs, nc, 0);
    flush_kind(cs, nc, 1);
    if (rc) break;
    for (int i = 0; i < nc; ++i)
      if ((rc = step_phase_b(cs[i], a[i]))) break;
    flush_kind(cs, nc, 2);
    if (rc) break;
    for (int i = 0; i < nc; ++i) {
      mpmhip_ctx *c = cs[i];
      if ((rc = step_phase_c(c, a[i]))) break;
      if (a[i].mesh_store && (a[i].mesh_x || a[i].mesh_v)) mesh_store_launch(c, a[i]);
      c->time = c->time + c->time_inc(a[i].dt);  // mpm_solver.py:536
      c->substeps += 1;
      c->fast->n_batched += 1;
    }
  }
  for (int i = 0; i < nc; ++i) { (void)batch_flush_ctx(cs[i]); cs[i]->fast->batching = false; }
  return rc;
}

}  // namespace mpm
