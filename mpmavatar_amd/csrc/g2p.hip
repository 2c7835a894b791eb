// g2p.hip -- g2p with the fused grid stage (k_g2p, k_g2p_halo) and the fused g2p -> p2g launch (k_g2p2g), and their launchers.
// (split out of fast.hip in round 4; shared device code: fast_device.hpp, shared host state: fast_state.hpp)
#include "fast_state.hpp"
#include <cstring>

namespace mpm {

namespace {

// B128: the tile is read with ds_read_b128 instead of the ds_read_b96 hipcc narrows the load to (g2p_device.hpp tile_read)
template <bool FUSED, bool TWO_PASS, bool MFLAG, bool B128>
__global__ __launch_bounds__(PT) void k_g2p(const ChunkRec *recs, int n_chunks, Bufs b, Dims d, float dt, GridPtrs g, GridParams gp,
                                             BCList bcl) {
  __shared__ float4 tile[TILE_PAD];  // node velocity, 16 bytes per node
  g2p_body<FUSED, TWO_PASS, MFLAG, false, B128>(recs, n_chunks, b, d, dt, g, gp, bcl, tile, (int)blockIdx.x);
}

// multi-GPU: fused halo add (see HaloIn)
template <bool TWO_PASS>
__global__ __launch_bounds__(PT) void k_g2p_halo(const ChunkRec *recs, int n_chunks, Bufs b, Dims d, float dt, GridPtrs g, GridParams gp,
                                                  BCList bcl) {
  __shared__ float4 tile[TILE_PAD];
  g2p_body<true, TWO_PASS, false, true>(recs, n_chunks, b, d, dt, g, gp, bcl, tile, (int)blockIdx.x);
}

template <int STEPS, bool FX>
__global__ __launch_bounds__(PT) void k_g2p2g(const ChunkRec *recs, int n_chunks, Bufs b, VAdj va, Dims d, float rpic, float dt, GridPtrs g,
                                               GridRead rd, SplatArgs sa, TradParams tp, GridParams gp, BCList bcl) {
  __shared__ double tile[P2G_TILE_DOUBLES];  // (col_splat_wg<3> runs in this launch too: its one-pass tile is 7 * SPLAT7_S doubles)
  __shared__ int esc[CHUNK];
  __shared__ int esc_n;
  __shared__ float red[8];
  g2p2g_body<STEPS, FX>(recs, n_chunks, b, va, d, rpic, dt, g, rd, sa, tp, gp, bcl, tile, esc, esc_n, red);
}

}  // namespace

// k_g2p on the g2p chunk list: fused = with the grid stage (no grid kernel in the substep), two = the two-sweep gather of cloth scenes
void launch_g2p(mpmhip_ctx *c, bool fused, bool two, float dt, const GridParams &gp, const BCList &bcl) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
#define G2P_ARGS xcd_grid(f->n_chunks_g), PT, f->chunks_g, f->n_chunks_g, b, d, dt, f->g, gp, bcl
  // (two-sweep kernel: ds_read_b128 costs it its fifth wavefront per SIMD -- taken where a launch is at most one round of workgroups)
  const bool wide = f->n_chunks_g <= 1280;
  if (!fused) {
    if (two) kstamp_launch(c, k_g2p<false, true, true, false>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<false, false, true, true>, G2P_ARGS);
  } else if (f->g.halo.slot) {
    if (two) kstamp_launch(c, k_g2p_halo<true>, G2P_ARGS);
    else kstamp_launch(c, k_g2p_halo<false>, G2P_ARGS);
  } else if (f->g2p_mflag) {
    if (two) kstamp_launch(c, k_g2p<true, true, true, false>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<true, false, true, true>, G2P_ARGS);
  } else {
    if (two && wide) kstamp_launch(c, k_g2p<true, true, false, true>, G2P_ARGS);
    else if (two) kstamp_launch(c, k_g2p<true, true, false, false>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<true, false, false, true>, G2P_ARGS);
  }
#undef G2P_ARGS
}
// k_g2p2g: g2p of the substep before (gp, bcl, read side rd) + stress and p2g of this one, see the kernel
void launch_g2p2g(mpmhip_ctx *c, unsigned grid, float dt, const GridRead &rd, const SplatArgs &sa, const TradParams &tp, const GridParams &gp,
                  const BCList &bcl) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
  if (f->p2g_fixed_now)
    kstamp_launch(c, k_g2p2g<P2G_STEPS, true>, grid, PT, f->chunks, f->n_chunks, b, f->va(), d, c->sc.rpic_damping, dt, f->g, rd, sa, tp,
                  gp, bcl);
  else
    kstamp_launch(c, k_g2p2g<3, false>, grid, PT, f->chunks, f->n_chunks, b, f->va(), d, c->sc.rpic_damping, dt, f->g, rd, sa, tp, gp,
                  bcl);
}

}  // namespace mpm
