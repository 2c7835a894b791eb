// p2g.hip -- stress kernels and p2g (k_stress_elem, k_stress_trad, k_stress_elem_splat, k_p2g) and their launchers.
// (split out of fast.hip in round 4; shared device code: fast_device.hpp, shared host state: fast_state.hpp)
#include "fast_state.hpp"

namespace mpm {

namespace {

template <bool FINALIZE>
__global__ void k_stress_elem(Bufs b, F3 *ef, Dims d, float friction_coeff, const int *face_slot,
                              const SortKey *skeys, int blk_bits, int *counters, int step_id) {
  stress_elem_body<FINALIZE>(blockIdx.x * blockDim.x + threadIdx.x, b, ef, d, friction_coeff, face_slot, skeys, blk_bits, counters, step_id);
}


__global__ void k_stress_trad(Bufs b, Dims d, mpmhip_model_scalars sc, float dt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_t) return;
  int s = t + d.n_e;
  if (b.sel[s] != 0) return;
  M3 Ft = ld9(b.tr, T_FT, t), F, stress;
  float mu = b.nv.at(N_MU, s), lam = b.nv.at(N_LAM, s), ys = b.tr.at(T_YS, t);
  int m = sc.material;
  TradParams tp{m, sc.alpha, sc.hardening, sc.xi, sc.plastic_viscosity, sc.softening};
  traditional_update(Ft, tp, mu, lam, ys, dt, F, stress);
  if (m == 1 || m == 5) b.tr.at(T_YS, t) = ys;
  if (m == 5) { b.nv.at(N_MU, s) = mu; b.nv.at(N_LAM, s) = lam; }
  st9(b.tr, T_F, t, F);
  st9(b.nv, N_STRESS, s, stress);
}

// The cloth scenes' stress launch with the collider splat's first pass in front (see col_splat_wg): workgroups [0, n_splat) splat,
// the rest are k_stress_elem<true>.
__global__ __launch_bounds__(TPB) void k_stress_elem_splat(Bufs b, F3 *ef, Dims d, float friction_coeff, const int *face_slot,
                                                           const SortKey *skeys, int blk_bits, int n_splat, GridPtrs g, SplatArgs sa) {
  __shared__ double tile[4 * TILE_PAD];
  if ((int)blockIdx.x < n_splat) {
    col_splat_wg<1>(tile, sa, (int)blockIdx.x, d, g);
    return;
  }
  stress_elem_body<true>(((int)blockIdx.x - n_splat) * (int)blockDim.x + (int)threadIdx.x, b, ef, d, friction_coeff, face_slot, skeys,
                         blk_bits, g.counters, g.step_id);
}


// The chunk records come first in the argument list: the record load is the head of every workgroup's dependency chain.
template <int STEPS, bool TRAD, bool JT, bool FX>
__global__ __launch_bounds__(PT) void k_p2g(const ChunkRec *recs, int n_chunks, Bufs b, VAdj va, Dims d, float rpic, float dt,
                                             GridPtrs g, SplatArgs sa, TradParams tp) {
  __shared__ double tile[P2G_TILE_DOUBLES];  // (the one-pass small-bin splat needs 7 * 536 doubles, everything else 4 * TILE_PAD)
  __shared__ int esc[CHUNK];
  __shared__ int esc_n;
  __shared__ float red[8];
  p2g_body<STEPS, TRAD, JT, FX>(recs, n_chunks, b, va, d, rpic, dt, g, sa, tp, tile, esc, esc_n, red, (int)blockIdx.x);
}
}  // namespace

// ---- launchers of the substep's kernels (the only places that name their template instantiations) -------------------------
// k_p2g with / without the fused traditional stress update (trad) and the in-tile joint splat of held traditional particles (jt),
// on the chunk list's first n_chunks records (0: only the extra workgroups sa describes)
void launch_p2g(mpmhip_ctx *c, bool trad, bool jt, unsigned grid, int n_chunks, float dt, const SplatArgs &sa, const TradParams &tp) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
  const float rpic = c->sc.rpic_damping;
#define P2G_ARGS grid, PT, f->chunks, n_chunks, b, f->va(), d, rpic, dt, f->g, sa, tp
  if (!f->p2g_fixed_now) {  // mpmhip_config.p2g_tile = F64, or particle masses that span more than 1e5
    if (trad && jt) kstamp_launch(c, k_p2g<3, true, true, false>, P2G_ARGS);
    else if (trad) kstamp_launch(c, k_p2g<3, true, false, false>, P2G_ARGS);
    else kstamp_launch(c, k_p2g<3, false, false, false>, P2G_ARGS);
  } else if (trad && jt) kstamp_launch(c, k_p2g<P2G_STEPS, true, true, true>, P2G_ARGS);
  else if (trad) kstamp_launch(c, k_p2g<P2G_STEPS, true, false, true>, P2G_ARGS);
  else kstamp_launch(c, k_p2g<P2G_STEPS, false, false, true>, P2G_ARGS);
#undef P2G_ARGS
}
// compute_stress_from_F_trial of the elements: mode 0 = from the stored directors (first substep after an import), 1 = with the
// element finalize of the substep before fused in, 2 = the same with the collider splat's first pass in front (sa.n_fbins workgroups)
void launch_stress_elem(mpmhip_ctx *c, int mode, const SplatArgs &sa) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
  hipStream_t s = c->stream;
  if (mode == 2)
    kstamp_launch(c, k_stress_elem_splat, nblk(d.n_e) + (unsigned)sa.n_fbins, TPB, b, f->eforce, d, c->sc.friction_coeff, f->face_slot,
                  f->keys[1], f->blk_bits, sa.n_fbins, f->g, sa);
  else if (mode == 1)
    kstamp_launch(c, k_stress_elem<true>, nblk(d.n_e), TPB, b, f->eforce, d, c->sc.friction_coeff, f->face_slot, f->keys[1], f->blk_bits,
                  f->g.counters, f->g.step_id);
  else
    hipLaunchKernelGGL(k_stress_elem<false>, nblk(d.n_e), TPB, 0, s, b, f->eforce, d, c->sc.friction_coeff, f->face_slot, f->keys[1],
                       f->blk_bits, f->g.counters, f->g.step_id);
}
void launch_stress_trad(mpmhip_ctx *c, float dt) {
  FastState *f = c->fast;
  hipLaunchKernelGGL(k_stress_trad, nblk(f->d.n_t), TPB, 0, c->stream, f->buf[f->cur], f->d, c->sc, dt);
}

}  // namespace mpm
