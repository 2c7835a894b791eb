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

// p2g AND g2p of a cloth substep in one launch (PhaseGate, fast_device.hpp): workgroups [0, p_end) are k_p2g's splat and chunk
// workgroups, [p_end, p_end + g_grid) the gather workgroups of k_g2p<true, true, false> behind the gate, the rest clear the other
// accumulator buffer (they fill the gather's tail as they filled the scatter's).
constexpr unsigned PG_LDS_BYTES = P2G_TILE_DOUBLES * sizeof(double) + CHUNK * sizeof(int) + 8 * sizeof(float) + 16;
struct G2PTail {
  const ChunkRec *recs;   // the g2p chunk list (== the p2g list on one GPU)
  int n_chunks, p_end, g_grid;
};
// The kernel's arguments as ONE record: the scatter half is a function of its own (noinline: register allocation and the
// scheduler's occupancy target are per FUNCTION, and inlined beside the gather half -- 96 VGPRs by itself -- the pair compiles to
// 126-129, four wavefronts per SIMD) and reads what it needs through the kernel-argument segment pointer.
struct PGArgs {
  const ChunkRec *recs;
  int n_chunks;
  Bufs b;
  VAdj va;
  Dims d;
  float rpic, dt;
  GridPtrs g;
  SplatArgs sa;
  TradParams tp;
  GridParams gp;
  BCList bcl;
  G2PTail gt;
};
template <int STEPS, bool FX, bool B128>
__device__ __forceinline__ void pg_kernel_body() {
  // The arguments are read THROUGH THE KERNEL-ARGUMENT SEGMENT POINTER, where they are used, instead of as by-value parameters: those
  // are all loaded into SGPRs at the kernel's entry, for both halves at once, and ~110 of them end up spilled into VGPR lanes.  A
  // scalar that is read from constant memory can be loaded again instead of being kept (6 spilled SGPRs).
  const PGArgs &A = *reinterpret_cast<const PGArgs *>(__builtin_amdgcn_kernarg_segment_ptr());
  // DYNAMIC shared memory (PG_LDS_BYTES at the launch): with the 31 KB declared statically hipcc takes the launch for LDS-bound at low
  // occupancy and schedules the gather half for 120-126 VGPRs -- the same code under a 12 KB tile compiles to 96
  extern __shared__ __attribute__((aligned(16))) double pg_lds[];
  const int bid = (int)blockIdx.x;
  if (bid < A.gt.p_end) {
    double *tile = pg_lds;
    int *esc = reinterpret_cast<int *>(pg_lds + P2G_TILE_DOUBLES);
    float *red = reinterpret_cast<float *>(esc + CHUNK);
    int &esc_n = *reinterpret_cast<int *>(red + 8);
    p2g_body<STEPS, false, false, FX, false>(A.recs, A.n_chunks, A.b, A.va, A.d, A.rpic, A.dt, A.g, A.sa, A.tp, tile, esc, esc_n, red, bid);
  } else if (bid < A.gt.p_end + A.gt.g_grid) {
    g2p_body<true, true, false, false, B128, true>(A.gt.recs, A.gt.n_chunks, A.b, A.d, A.dt, A.g, A.gp, A.bcl, reinterpret_cast<float4 *>(pg_lds), bid - A.gt.p_end, A.sa.gate);
  } else {
    zero_blocks_wg(A.sa.z, bid - (A.gt.p_end + A.gt.g_grid));
  }
}
// Two halves that compile to 89 and 95 VGPRs by themselves come out at 125 together: the scheduler's occupancy target is per function,
// drops to four wavefronts per SIMD while it works on the scatter nest and is not raised again for the gather sweeps.  The attribute
// pins it at five: the two hot regions are instruction for instruction those of k_p2g / k_g2p (tools/isa_mix.py), ten dwords of the
// small-bin splat's face data go through scratch.
#ifndef PG_WAVES
#define PG_WAVES __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
template <int STEPS, bool FX>
__global__ __launch_bounds__(PT) PG_WAVES void k_p2g_g2p(PGArgs by_value) { pg_kernel_body<STEPS, FX, false>(); }
// chunk lists of at most one round of workgroups: the gather reads its tile with ds_read_b128 (see launch_g2p) and occupancy is not
// what bounds the launch -- no pin
template <int STEPS, bool FX>
__global__ __launch_bounds__(PT) void k_p2g_g2p_wide(PGArgs by_value) { pg_kernel_body<STEPS, FX, true>(); }
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
// the cloth form of p2g + the fused two-sweep g2p as ONE launch; sa.z_first must be out of range (the clearing workgroups come last here)
void launch_p2g_g2p(mpmhip_ctx *c, int n_chunks, float dt, const SplatArgs &sa, const TradParams &tp, const GridParams &gp, const BCList &bcl) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
  const float rpic = c->sc.rpic_damping;
  G2PTail gt{f->chunks_g, f->n_chunks_g, sa.n_extra + (int)xcd_grid(n_chunks), (int)xcd_grid(f->n_chunks_g)};
  const unsigned grid = (unsigned)(gt.p_end + gt.g_grid + sa.z.n_wg);
  const bool wide = f->n_chunks_g <= 1280;  // (ds_read_b128 in the sweeps where a launch is at most one round: see launch_g2p)
  const PGArgs A{f->chunks, n_chunks, b, f->va(), d, rpic, dt, f->g, sa, tp, gp, bcl, gt};
  if (!f->p2g_fixed_now) {
    if (wide) kstamp_launch_lds(c, k_p2g_g2p_wide<3, false>, grid, PT, PG_LDS_BYTES, A);
    else kstamp_launch_lds(c, k_p2g_g2p<3, false>, grid, PT, PG_LDS_BYTES, A);
  } else if (wide) kstamp_launch_lds(c, k_p2g_g2p_wide<P2G_STEPS, true>, grid, PT, PG_LDS_BYTES, A);
  else kstamp_launch_lds(c, k_p2g_g2p<P2G_STEPS, true>, grid, PT, PG_LDS_BYTES, A);
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
