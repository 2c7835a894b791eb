// batch_args.hpp -- argument records of the BATCHED launches (batch.hip): several solver contexts' workgroups in one launch per phase.
// The by-value argument sets of k_p2g / k_g2p are 1.0-1.2 KB each (GridPtrs carries the multi-GPU HaloIn, SplatArgs the PackArgs, g2p a
// 712-byte BC list); four of them do not fit the 4 KB kernel-argument segment.  The batched launches take lean forms without the
// multi-GPU members (a batched context is never sharded) and the BC list through a pointer to a device copy.
#pragma once
#include "g2p_device.hpp"

namespace mpm {

constexpr int BATCH_MAX = 4;

struct GridLean {
  float *mv, *vout, *col, *mov;
  const int *ab_flag;
  int *col_flag, *m_flag, *counters, *host_sig;
  int step_id;
  float lookahead;
  int stagger, stagger_groups, stagger_first;
  float *xprev;
};
inline GridLean lean(const GridPtrs &g) {
  return GridLean{g.mv, g.vout, g.col, g.mov, g.ab_flag, g.col_flag, g.m_flag, g.counters, g.host_sig, g.step_id, g.lookahead,
                  g.stagger, g.stagger_groups, g.stagger_first, g.xprev};
}
__device__ __forceinline__ GridPtrs expand(const GridLean &l) {
  GridPtrs g{};
  g.mv = l.mv; g.vout = l.vout; g.col = l.col; g.mov = l.mov; g.ab_flag = l.ab_flag; g.col_flag = l.col_flag; g.m_flag = l.m_flag;
  g.counters = l.counters; g.host_sig = l.host_sig; g.step_id = l.step_id; g.lookahead = l.lookahead; g.stagger = l.stagger;
  g.stagger_groups = l.stagger_groups; g.stagger_first = l.stagger_first; g.xprev = l.xprev;
  return g;
}
struct SplatLean {
  const float *pts, *vel;
  float adv;
  const int *fidx;
  const FaceBin *fbins;
  int n_fbins, splat_passes;
  JointSplatArgs js;
  int n_mov_wg, n_extra, e0;
  ZeroArgs z;
  int z_first;
};
inline SplatLean lean(const SplatArgs &s) {
  return SplatLean{s.pts, s.vel, s.adv, s.fidx, s.fbins, s.n_fbins, s.splat_passes, s.js, s.n_mov_wg, s.n_extra, s.e0, s.z, s.z_first};
}
__device__ __forceinline__ SplatArgs expand(const SplatLean &l) {
  SplatArgs s{};
  s.pts = l.pts; s.vel = l.vel; s.adv = l.adv; s.fidx = l.fidx; s.fbins = l.fbins; s.n_fbins = l.n_fbins; s.splat_passes = l.splat_passes;
  s.js = l.js; s.n_mov_wg = l.n_mov_wg; s.n_extra = l.n_extra; s.e0 = l.e0; s.z = l.z; s.z_first = l.z_first;
  return s;
}

struct StressB {   // k_stress_elem<true> / k_stress_elem_splat of one context
  Bufs b;
  F3 *ef;
  Dims d;
  float friction_coeff;
  const int *face_slot;
  const SortKey *skeys;
  int blk_bits, n_splat, grid;
  GridLean g;
  SplatLean sa;
};
struct P2GB {      // k_p2g<P2G_STEPS, false, false, true> of one context
  const ChunkRec *recs;
  int n_chunks, grid;
  Bufs b;
  VAdj va;
  Dims d;
  float rpic, dt;
  GridLean g;
  SplatLean sa;
};
struct G2PB {      // k_g2p<true, true, false, false> of one context
  const ChunkRec *recs;
  int n_chunks, grid;
  Bufs b;
  Dims d;
  float dt;
  GridLean g;
  GridParams gp;
  const BCList *bcl;   // device copy (FastState::bcl_dev)
};
template <class T>
struct Batch {
  int n;
  int first[BATCH_MAX + 1];   // workgroups [first[i], first[i + 1]) belong to entry i (each a multiple of 8: keeps the XCD mapping)
  T a[BATCH_MAX];
};
static_assert(sizeof(Batch<P2GB>) <= 3800 && sizeof(Batch<G2PB>) <= 3800 && sizeof(Batch<StressB>) <= 3800, "kernel-argument segment is 4 KB");

}  // namespace mpm
