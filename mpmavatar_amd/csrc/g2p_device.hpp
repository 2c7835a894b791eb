// g2p_device.hpp -- device side of g2p (g2p_v / g2p_e, mpm_utils.py:716-857) with the fused grid stage, and of the fused g2p -> p2g
// workgroup of traditional-only scenes (g2p2g_body).  Kernels: g2p.hip.
#pragma once
#include "p2g_device.hpp"

namespace mpm {
inline namespace fk {

// ------------------------------------------------------------------------------------------------
// g2p (g2p_v / g2p_e, mpm_utils.py:716-857) with the v_out tile staged in LDS.
// Factored gather: for every (i,j) column first reduce over k
//   s0 = sum_k wz_k u_ijk,  s1 = sum_k dwz_k u_ijk,  s2 = sum_k k wz_k u_ijk          (u = grid_v_out)
// then v = sum wxy s0,  M1 = sum u (x) (i,j,k) w = [sum i wxy s0 | sum j wxy s0 | sum wxy s2],
// grad v = inv_dx [sum dwx wy s0 | sum wx dwy s0 | sum wxy s1],  C = 4 inv_dx (M1 - v (x) fx)
// (the reference accumulates outer(grid_v, dpos) * weight * inv_dx * 4 with dpos = (i,j,k) - fx, :753-763).
// ------------------------------------------------------------------------------------------------
struct G2PResult {
  V3 v;
  M3 C, F;  // C (APIC matrix) and grad v
};

__device__ __forceinline__ G2PResult g2p_finish(const Stencil &s, const Dims &d, V3 nv, V3 Mx, V3 My, V3 Mz, V3 Fx, V3 Fy,
                                                V3 Fz) {
  G2PResult r;
  r.v = nv;
  float c4 = 4.0f * d.inv_dx;
  r.C = m3_cols(c4 * (Mx - s.fx.x * nv), c4 * (My - s.fx.y * nv), c4 * (Mz - s.fx.z * nv));
  r.F = m3_cols(d.inv_dx * Fx, d.inv_dx * Fy, d.inv_dx * Fz);
  return r;
}

// One tile node = one ds_read_b128.  Left to itself hipcc narrows the load to ds_read_b96 because the gathers never use .w -- and a
// ds_read_b96 costs the LDS 8 cycles per wave instruction (eight lane groups) against 4 for a ds_read_b128 (MI355X_MICROARCH.md, LDS
// table): the sweeps are LDS-read bound, so the "narrower" load is the slower one.  .w is made a used value at the point of the
// load: x + 0 * w with w = 0 as staged (exact; without fast-math the compiler must keep it: w could be NaN for all it knows).  An
// empty asm on .w does the same but lets the scheduler park all 27 w registers until the end of the sweep (+29 VGPRs).
// B128 costs four VGPRs (one more per load in flight): the two-sweep cloth kernel goes from 96 to 100, i.e. from five to four
// wavefronts per SIMD -- measured (tools/gpu/r04n.sh): garment-120k-aniso k_g2p 11.0-11.4 -> 10.3-10.6 us (a launch of less than one round:
// a workgroup's latency is what counts), sheet-500k 18.7-19.2 -> 20.1-20.3 us (2.1 rounds: occupancy counts; forced back to five
// wavefronts it spills 21 registers: 21.4 us).  So B128 is a template parameter and the launcher picks it where occupancy is not what
// bounds the launch: chunk lists of at most one round, and the single-sweep kernels of traditional scenes (117 -> 121 VGPRs, four
// wavefronts either way).
#ifndef G2P_D3_SWEEP
#define G2P_D3_SWEEP 1  // experiment switch: second sweep of cloth-only wavefronts through g2p_gather_grad_d3
#endif
template <bool B128>
__device__ __forceinline__ float4 tile_read(const float4 *tile, int idx) {
  float4 t4 = tile[idx];
  if (B128) t4.x = fmaf(0.0f, t4.w, t4.x);
  return t4;
}

template <bool B128>
__device__ __forceinline__ G2PResult g2p_gather(const float4 *tile, int ox, int oy, int oz, V3 x, const Dims &d) {
  Stencil s = make_stencil(x, d.inv_dx);
  int base = tile_idx(s.bx - ox, s.by - oy, s.bz - oz);
  V3 nv = v3(0, 0, 0), Mx = v3(0, 0, 0), My = v3(0, 0, 0), Mz = v3(0, 0, 0);
  V3 Fx = v3(0, 0, 0), Fy = v3(0, 0, 0), Fz = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x), dwx = sel3(i, s.dw0.x, s.dw1.x, s.dw2.x);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float wy = sel3(j, s.w0.y, s.w1.y, s.w2.y), dwy = sel3(j, s.dw0.y, s.dw1.y, s.dw2.y);
      V3 s0 = v3(0, 0, 0), s1 = v3(0, 0, 0), s2 = v3(0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wzk = sel3(k, s.w0.z, s.w1.z, s.w2.z), dwzk = sel3(k, s.dw0.z, s.dw1.z, s.dw2.z);
        const float4 t4 = tile_read<B128>(tile, base + tile_idx(i, j, k));
        V3 u = v3(t4.x, t4.y, t4.z);
        s0 = s0 + wzk * u;
        s1 = s1 + dwzk * u;
        if (k > 0) s2 = s2 + ((float)k * wzk) * u;
      }
      float wxy = wx * wy;
      nv = nv + wxy * s0;
      if (i > 0) Mx = Mx + ((float)i * wxy) * s0;
      if (j > 0) My = My + ((float)j * wxy) * s0;
      Mz = Mz + wxy * s2;
      Fx = Fx + (dwx * wy) * s0;
      Fy = Fy + (wx * dwy) * s0;
      Fz = Fz + wxy * s1;
    }
  }
  return g2p_finish(s, d, nv, Mx, My, Mz, Fx, Fy, Fz);
}

// the same gather in two passes (velocity + APIC matrix, then the velocity gradient): 12 and 9 accumulators instead
// of 21 at a time
template <bool B128>
__device__ __forceinline__ void g2p_gather_vC(const float4 *tile, int ox, int oy, int oz, V3 x, const Dims &d, V3 &v, M3 &C) {
  Stencil s = make_stencil(x, d.inv_dx);
  int base = tile_idx(s.bx - ox, s.by - oy, s.bz - oz);
  V3 nv = v3(0, 0, 0), Mx = v3(0, 0, 0), My = v3(0, 0, 0), Mz = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float wx = bspline_w(i, s.fx.x);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float wy = bspline_w(j, s.fx.y);
      V3 s0 = v3(0, 0, 0), s2 = v3(0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wzk = sel3(k, s.w0.z, s.w1.z, s.w2.z);
        const float4 t4 = tile_read<B128>(tile, base + tile_idx(i, j, k));
        V3 u = v3(t4.x, t4.y, t4.z);
        s0 = s0 + wzk * u;
        if (k > 0) s2 = s2 + ((float)k * wzk) * u;
      }
      float wxy = wx * wy;
      nv = nv + wxy * s0;
      if (i > 0) Mx = Mx + ((float)i * wxy) * s0;
      if (j > 0) My = My + ((float)j * wxy) * s0;
      Mz = Mz + wxy * s2;
    }
  }
  float c4 = 4.0f * d.inv_dx;
  v = nv;
  C = m3_cols(c4 * (Mx - s.fx.x * nv), c4 * (My - s.fx.y * nv), c4 * (Mz - s.fx.z * nv));
}
template <bool B128>
__device__ __forceinline__ M3 g2p_gather_grad(const float4 *tile, int ox, int oy, int oz, V3 x, const Dims &d) {
  Stencil s = make_stencil(x, d.inv_dx);
  int base = tile_idx(s.bx - ox, s.by - oy, s.bz - oz);
  V3 Fx = v3(0, 0, 0), Fy = v3(0, 0, 0), Fz = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float wx = bspline_w(i, s.fx.x), dwx = bspline_dw(i, s.fx.x);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float wy = bspline_w(j, s.fx.y), dwy = bspline_dw(j, s.fx.y);
      V3 s0 = v3(0, 0, 0), s1 = v3(0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wzk = sel3(k, s.w0.z, s.w1.z, s.w2.z), dwzk = bspline_dw(k, s.fx.z);
        const float4 t4 = tile_read<B128>(tile, base + tile_idx(i, j, k));
        V3 u = v3(t4.x, t4.y, t4.z);
        s0 = s0 + wzk * u;
        s1 = s1 + dwzk * u;
      }
      Fx = Fx + (dwx * wy) * s0;
      Fy = Fy + (wx * dwy) * s0;
      Fz = Fz + (wx * wy) * s1;
    }
  }
  return m3_cols(d.inv_dx * Fx, d.inv_dx * Fy, d.inv_dx * Fz);
}

// Second sweep of a wavefront that holds cloth only (elements and vertices, no traditional particle): an element needs nothing of
// grad v but what it does to its third director, (grad v) d3 = inv_dx sum_n u_n (grad w_n . d3) (g2p_e, mpm_utils.py:843-857:
// new_d3 = (I + dt grad_v) d3) -- three accumulators instead of nine and 5 instead of 9 VALU instructions per node:
//   grad w_n . d3 = wz_k (dwx_i wy_j d3x + wx_i dwy_j d3y) + dwz_k (wx_i wy_j d3z)
// (round 4 had this only inside the one-sweep gather that died on its 171 VGPRs, profiles/r04_experiments.md 2).
template <bool B128>
__device__ __forceinline__ V3 g2p_gather_grad_d3(const float4 *tile, int ox, int oy, int oz, V3 x, V3 d3, const Dims &d) {
  Stencil s = make_stencil(x, d.inv_dx);
  int base = tile_idx(s.bx - ox, s.by - oy, s.bz - oz);
  V3 acc = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float wx = bspline_w(i, s.fx.x), dwx = bspline_dw(i, s.fx.x);
    float ax = dwx * d3.x, ay = wx * d3.y, az = wx * d3.z;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float wy = bspline_w(j, s.fx.y), dwy = bspline_dw(j, s.fx.y);
      float A = fmaf(ax, wy, ay * dwy), B = az * wy;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float wzk = sel3(k, s.w0.z, s.w1.z, s.w2.z), dwzk = bspline_dw(k, s.fx.z);
        const float4 t4 = tile_read<B128>(tile, base + tile_idx(i, j, k));
        float c = fmaf(wzk, A, dwzk * B);
        acc = acc + c * v3(t4.x, t4.y, t4.z);
      }
    }
  }
  return d.inv_dx * acc;
}

// same sums for a particle that drifted out of its tile margin: rolled loop over the global grid (zero outside
// active blocks); kept small so that it does not set the kernel's register budget
template <bool FUSED, bool HALO = false>
__device__ __forceinline__ G2PResult g2p_gather_global(V3 x, const Dims &d, const GridPtrs &g, const GridParams &gp,
                                                       const BCList &bcl) {
  Stencil s = make_stencil(x, d.inv_dx);
  V3 nv = v3(0, 0, 0), Mx = v3(0, 0, 0), My = v3(0, 0, 0), Mz = v3(0, 0, 0);
  V3 Fx = v3(0, 0, 0), Fy = v3(0, 0, 0), Fz = v3(0, 0, 0);
#pragma unroll 1
  for (int n = 0; n < 27; ++n) {
    int i = n / 9, j = (n / 3) % 3, k = n % 3;
    float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x), wy = sel3(j, s.w0.y, s.w1.y, s.w2.y), wz = sel3(k, s.w0.z, s.w1.z, s.w2.z);
    float dwx = sel3(i, s.dw0.x, s.dw1.x, s.dw2.x), dwy = sel3(j, s.dw0.y, s.dw1.y, s.dw2.y), dwz = sel3(k, s.dw0.z, s.dw1.z, s.dw2.z);
    int x_ = s.bx + i, y_ = s.by + j, z_ = s.bz + k;
    V3 u = v3(0, 0, 0);
    if (in_grid(x_, y_, z_, d.G)) {
      int blk = blk_of(x_, y_, z_, d.NB);
      if (g.ab_flag[blk]) {
        if (FUSED && HALO) {
          int nc = 0, nm = 0, l_ = loc_of(x_, y_, z_);
          const float *pm = g.mv + ((size_t)blk * GCH_MV) * 64 + l_;
          float m = pm[0], px = pm[64], py = pm[128], pz = pm[192];
          const float *rem_mov = nullptr;
          int hs = g.halo.slot[blk];
          if (hs >= 0) {
            link_wait_lane(halo_sig(g.halo, hs >> 24), g.halo.seq, g.counters + 10);
            rem_mov = halo_add_node(g.halo, hs, l_, m, px, py, pz);
          }
          u = node_finish<false>(blk, l_, m, px, py, pz, d, g, gp, bcl, nc, nm, true, 0xffffffffu, rem_mov);
        } else if (FUSED) {
          float m;
          int nc = 0, nm = 0;
          u = node_update<false>(blk, loc_of(x_, y_, z_), d, g, gp, bcl, m, nc, nm);
        } else {
          const float *p = g.vout + ((size_t)blk * GCH_VOUT) * 64 + loc_of(x_, y_, z_);
          u = v3(p[0], p[64], p[128]);
        }
      }
    }
    float w = wx * wy * wz;
    nv = nv + w * u;
    Mx = Mx + ((float)i * w) * u; My = My + ((float)j * w) * u; Mz = Mz + ((float)k * w) * u;
    Fx = Fx + (dwx * wy * wz) * u; Fy = Fy + (wx * dwy * wz) * u; Fz = Fz + (wx * wy * dwz) * u;
  }
  return g2p_finish(s, d, nv, Mx, My, Mz, Fx, Fy, Fz);
}

// particle update from the gathered values (g2p_v :765-786, first half of g2p_e :843-857)
// what the velocity gradient feeds: d3 of an element, F_trial of a traditional particle (g2p_e :843-857, g2p_v :780-786)
__device__ __forceinline__ void g2p_write_grad(const Bufs &b, int cls, int s, V3 d3, const M3 &F, const Dims &d, float dt) {
  if (cls == 0) {
    // elements: d3 <- (I + dt grad v) d3 now; x, v, d1, d2 in k_elem_finalize once all vertices are updated
    V3 d3n = (m3_identity() + dt * F) * d3;
    b.el.at(E_D + 2, s) = d3n.x; b.el.at(E_D + 5, s) = d3n.y; b.el.at(E_D + 8, s) = d3n.z;
  } else if (cls == 1) {
    st9(b.tr, T_FT, s - d.n_e, deform_update(F, dt, ld9(b.tr, T_F, s - d.n_e)));
  }
}
template <bool NO_GRAD = false>
__device__ __forceinline__ void g2p_write(const Bufs &b, int cls, int s, V3 x, V3 d3, const G2PResult &r, int ox, int oy,
                                          int oz, const Dims &d, float dt, const GridPtrs &g) {
  st9(b.all, A_C, s, r.C);
  if (cls == 0) {
    if (!NO_GRAD) g2p_write_grad(b, cls, s, d3, r.F, d, dt);
    return;
  }
  float a_min = (1.0f / d.inv_dx) * 2.0f, a_max = d.grid_lim - (1.0f / d.inv_dx) * 2.0f;
  st3(b.all, A_V, s, r.v);
  V3 nx = x + dt * r.v;
  nx = v3(fminf(fmaxf(nx.x, a_min), a_max), fminf(fmaxf(nx.y, a_min), a_max), fminf(fmaxf(nx.z, a_min), a_max));
  st3(b.all, A_X, s, nx);
  {  // already outside the tile margin of its block?  Ask the host for a re-sort.  (The early warning -- will it still
     // fit DRIFT_LOOKAHEAD substeps from now -- is raised by the next p2g, where x and v are in registers anyway; here
     // it cost hipcc 18-50 more VGPRs and an occupancy step.)
    int nbx = (int)(nx.x * d.inv_dx - 0.5f) - ox, nby = (int)(nx.y * d.inv_dx - 0.5f) - oy, nbz = (int)(nx.z * d.inv_dx - 0.5f) - oz;
    if ((unsigned)nbx > 5u || (unsigned)nby > 5u || (unsigned)nbz > 5u) raise_drift(g.counters, g.step_id);
  }
  if (cls == 1 && !NO_GRAD) g2p_write_grad(b, cls, s, d3, r.F, d, dt);
}

// FUSED = true: there is no grid kernel in the substep; the tile is staged from the accumulators and every node goes
// through node_update<false> on the way (normalise, gravity, damping, collide, mover, BCs).  Nodes shared by several
// tiles are evaluated once per tile (about 2x redundant arithmetic, ~50 VALU instructions per node) in exchange for
// one launch, one v_out round trip through HBM and one grid-wide dependency less per substep.
// TWO_PASS: gather velocity + APIC matrix first and the velocity gradient in a second sweep over the tile, the
// latter only by wavefronts that hold elements or traditional particles.  12 + 9 instead of 21 accumulators at a time:
// 95 instead of 114 VGPRs, a fifth wavefront per SIMD.  Pays when many lanes are vertices (cloth scenes: a third of the
// particles skip the second sweep); traditional-only scenes read every node twice and keep the single sweep.
// MFLAG = false: the accumulators of all 27 overlapped blocks are loaded as soon as the chunk record is there, without first
// asking m_flag which of them were scattered into (the ~70 % that were not read back zeros from L2).  One dependent memory
// level less at the head of every workgroup for more L2 traffic; node values are identical (an unflagged block holds zeros).
// HALO = true (multi-GPU, needs MFLAG = false): blocks shared with a neighbour rank get its contribution added on the way
// (HaloIn), after the workgroup has seen the neighbour's flag for this substep.
template <bool FUSED, bool TWO_PASS, bool MFLAG, bool HALO = false, bool B128 = false>
__device__ __forceinline__ void g2p_body(const ChunkRec *recs, int n_chunks, const Bufs &b, const Dims &d, float dt, const GridPtrs &g,
                                         const GridParams &gp, const BCList &bcl, float4 *tile, int wg) {
  WGT(g, 1, 0);
  // The front of a g2p workgroup is a chain of memory latencies (record -> positions + accumulators -> grid stage): few
  // instructions, long waits.  Its wavefronts get issue priority over wavefronts that are in the VALU / LDS-bound sweeps of another
  // workgroup on the same SIMD, so the next request of the chain goes out when its data is there: -0.5...-0.8 us on every scene
  // (profiles/r03_experiments.md).  (The same in p2g is neutral for cloth and costs the fused traditional stress update 7 us: its
  // SVD sits in that front.)
  __builtin_amdgcn_s_setprio(3);
  int w = xcd_slice(wg, n_chunks);
  if (w < 0) return;
  const ChunkRec cm = recs[w];
  // (the whole record with the first request: LLVM sinks the load of a field into the branch of map() that uses it, and e0 -- the
  // common case -- arrived one dependent memory level after the rest; the empty asm uses the three range starts here)
  asm volatile("" ::"s"(cm.e0), "s"(cm.t0), "s"(cm.v0));
  int blk = cm.blk, chunk = cm.chunk;
  int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
  int ox = 4 * bx - 1, oy = 4 * by - 1, oz = 4 * bz - 1;
  int cls = 0, s = 0;
  bool valid = cm.map(chunk * CHUNK + (int)threadIdx.x, cls, s);
  // Everything that depends on the chunk record alone is loaded NOW, back to back and without a branch in between: the
  // particle's position (and director), and -- MFLAG = false -- the accumulators of this thread's two tile nodes.  (With the
  // loads behind `if (valid)` / behind the flag ballots the compiler cannot issue them before the first wait, and every
  // dependent memory level costs a workgroup 1.3-1.5 us: profiles/r03_wg_timeline.md.)
  V3 x, d3;
  {
    int sx_ = valid ? s : 0, se = (valid && cls == 0) ? s : 0;
    x = ld3(b.all, A_X, sx_);
    d3 = v3(b.el.at(E_D + 2, se), b.el.at(E_D + 5, se), b.el.at(E_D + 8, se));
  }
  constexpr int NPT = TILE3 / PT;  // tile nodes per thread
  int nbk[NPT], nlk[NPT], hsl[NPT];
  float am[NPT], apx[NPT], apy[NPT], apz[NPT];
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    int t = (int)threadIdx.x + u * PT;
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    int gx = ox + ti, gy = oy + tj, gz = oz + tk;
    bool in = in_grid(gx, gy, gz, d.G);
    nbk[u] = in ? blk_of(gx, gy, gz, d.NB) : -1;
    nlk[u] = loc_of(gx, gy, gz);
    am[u] = apx[u] = apy[u] = apz[u] = 0.0f;
    if (FUSED && !MFLAG) {  // (all 27 overlapped blocks of a particle block are on the active list: cleared or loaded;
                            // a node outside the grid reads this block's instead -- no branch around the loads -- and drops it)
      const float *pm = g.mv + ((size_t)(in ? nbk[u] : blk) * GCH_MV) * 64 + nlk[u];
      float a0 = pm[0], a1 = pm[64], a2 = pm[128], a3 = pm[192];
      am[u] = in ? a0 : 0.0f; apx[u] = in ? a1 : 0.0f; apy[u] = in ? a2 : 0.0f; apz[u] = in ? a3 : 0.0f;
    }
    hsl[u] = -1;
    if (HALO) { int hs = g.halo.slot[in ? nbk[u] : blk]; hsl[u] = in ? hs : -1; }
  }
  // tile-level shortcuts for the fused node evaluation (both wave-uniform): which of the 27 overlapped blocks may
  // carry body-collider data this substep (flags set by the splat), and which BCs can reach this tile at all
  unsigned long long col_mask = 0, m_mask = 0;
  unsigned bc_mask = 0;
  if (FUSED) {
    int l = threadIdx.x & 63, fl = 0, fm = 0;
    if (l < 27) {
      int nx = bx + l / 9 - 1, ny = by + (l / 3) % 3 - 1, nz = bz + l % 3 - 1;
      if ((unsigned)nx < (unsigned)d.NB && (unsigned)ny < (unsigned)d.NB && (unsigned)nz < (unsigned)d.NB) {
        if (MFLAG) fm = g.m_flag[(nx * d.NB + ny) * d.NB + nz];
        if (gp.has_col) fl = g.col_flag[(nx * d.NB + ny) * d.NB + nz];
      }
    }
    if (gp.has_col) col_mask = __ballot(fl != 0);
    m_mask = MFLAG ? __ballot(fm != 0) : ~0ull;  // a block nobody scattered into: its nodes carry no mass, hence no weight in any gather
    for (int k = 0; k < bcl.n; ++k)
      if (bc_may_touch(bcl.bc[k], ox, oy, oz, ox + 7, oy + 7, oz + 7, d.G, d.dx, gp.time, gp.dt)) bc_mask |= 1u << k;
    if (HALO) {  // wait for the flag of every neighbour rank this tile shares a block with (the same set in every wavefront)
      int hs27 = -1;
      if (l < 27) {
        int nx = bx + l / 9 - 1, ny = by + (l / 3) % 3 - 1, nz = bz + l % 3 - 1;
        if ((unsigned)nx < (unsigned)d.NB && (unsigned)ny < (unsigned)d.NB && (unsigned)nz < (unsigned)d.NB)
          hs27 = g.halo.slot[(nx * d.NB + ny) * d.NB + nz];
      }
      for (int k = 0; k < g.halo.n_peers; ++k)
        if (__any(hs27 >= 0 && (hs27 >> 24) == k)) link_wait(halo_sig(g.halo, k), g.halo.seq, g.counters + 10);
    }
  }
  bool escaped = false;
  if (valid) {
    int lx = (int)(x.x * d.inv_dx - 0.5f) - ox, ly = (int)(x.y * d.inv_dx - 0.5f) - oy, lz = (int)(x.z * d.inv_dx - 0.5f) - oz;
    escaped = (unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u;  // drifted out of the tile margin
  }
  WGT(g, 1, 1);  // chunk record, particle positions, block flags (and, MFLAG = false, the accumulators) here
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    int t = (int)threadIdx.x + u * PT;
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    V3 v = v3(0, 0, 0);
    if (nbk[u] >= 0) {
      int nb = nbk[u], nl = nlk[u];
      if (FUSED) {
        int nc = 0, nm = 0;
        int nidx = ((((ox + ti) >> 2) - bx + 1) * 3 + (((oy + tj) >> 2) - by + 1)) * 3 + (((oz + tk) >> 2) - bz + 1);
        bool uc = (col_mask >> nidx) & 1ull;
        if (!MFLAG) {
          const float *rem_mov = nullptr;
          if (HALO && hsl[u] >= 0) rem_mov = halo_add_node(g.halo, hsl[u], nl, am[u], apx[u], apy[u], apz[u]);
          v = node_finish<false>(nb, nl, am[u], apx[u], apy[u], apz[u], d, g, gp, bcl, nc, nm, uc, bc_mask, rem_mov);
        } else if ((m_mask >> nidx) & 1ull) {
          float m;
          v = node_update<false>(nb, nl, d, g, gp, bcl, m, nc, nm, uc, bc_mask);
        }
      } else {
        const float *p = g.vout + ((size_t)nb * GCH_VOUT) * 64 + nl;
        v = v3(p[0], p[64], p[128]);
      }
    }
    tile[tile_idx(ti, tj, tk)] = make_float4(v.x, v.y, v.z, 0.0f);
  }
  WGT(g, 1, 2);  // wavefront 0 has staged its nodes (accumulator loads + grid stage)
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  WGT(g, 1, 3);  // tile complete
  {
    // lanes without a particle in the tile margin gather from the tile corner (in range, result unused)
    bool fit = valid && !escaped;
    V3 xg = fit ? x : v3((float)(ox + 2) * d.dx, (float)(oy + 2) * d.dx, (float)(oz + 2) * d.dx);
    // a wavefront without a single particle (the tail of a chunk: a flat sheet fills ~184 of the 256 lanes) has helped to
    // stage the tile and is done: the gather is ~600 VALU instructions per wavefront and the kernel is bound by VALU issue
    if (!__any(fit)) {
    } else if (!TWO_PASS) {
      G2PResult r = g2p_gather<B128>(tile, ox, oy, oz, xg, d);
      if (fit) g2p_write(b, cls, s, x, d3, r, ox, oy, oz, d, dt, g);
    } else {
      {
        G2PResult r;
        r.F = m3_zero();
        g2p_gather_vC<B128>(tile, ox, oy, oz, xg, d, r.v, r.C);
        if (fit) g2p_write<true>(b, cls, s, x, d3, r, ox, oy, oz, d, dt, g);
      }
      WGT(g, 1, 4);  // first sweep (v, C) of wavefront 0 stored
      if (__any(fit && cls != 2)) {  // elements and traditional particles also need grad v
        asm volatile("" : "+v"(xg.x), "+v"(xg.y), "+v"(xg.z));  // a fresh stencil: nothing of the first sweep stays live
        if (G2P_D3_SWEEP && !__any(fit && cls == 1)) {  // cloth only: (grad v) d3 with three accumulators (g2p_gather_grad_d3)
          V3 gd = g2p_gather_grad_d3<B128>(tile, ox, oy, oz, xg, d3, d);
          if (fit && cls == 0) {
            V3 d3n = d3 + dt * gd;
            b.el.at(E_D + 2, s) = d3n.x; b.el.at(E_D + 5, s) = d3n.y; b.el.at(E_D + 8, s) = d3n.z;
          }
        } else {
          M3 rF = g2p_gather_grad<B128>(tile, ox, oy, oz, xg, d);
          if (fit && cls != 2) g2p_write_grad(b, cls, s, d3, rF, d, dt);
        }
      }
    }
  }
  // A particle outside the tile margin (rare, and only until the re-sort its drift flag has already requested) is
  // finished here from the global grid with a rolled loop.  The empty asm makes its inputs opaque: otherwise the
  // optimizer shares stencil weights and store addresses with the tile path above and keeps them live across it
  // (202 instead of ~110 VGPRs).  A follow-up kernel for these particles cost 4.7 us per substep for nothing.
  if (__any(escaped)) {
    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(s), "+v"(cls), "+v"(d3.x), "+v"(d3.y), "+v"(d3.z));
    if (escaped) {
      G2PResult r = g2p_gather_global<FUSED, HALO>(x, d, g, gp, bcl);
      g2p_write(b, cls, s, x, d3, r, ox, oy, oz, d, dt, g);
      atomicAdd(g.counters + 0, 1);
    }
  }
  WGT(g, 1, 6);
}
// ------------------------------------------------------------------------------------------------
// G2P2G (round 4): scenes of traditional particles only run ONE launch per substep.  A workgroup finishes substep n for its
// chunk -- g2p_v (mpm_utils.py:716-786) from the accumulators p2g(n) filled, every node through the grid stage on the way -- and,
// with the particles still in registers, starts substep n + 1: compute_stress_from_F_trial (:1047-1103) and p2g (:484-557) into
// ANOTHER accumulator buffer.  Nothing grid-wide lies between g2p(n) and p2g(n + 1) of the same particle; what is grid-wide -- every
// p2g(n + 1) contribution must be in before any g2p(n + 1) reads -- is the kernel boundary to the next launch.  Cloth cannot do
// this: an element needs its three vertices' new positions and a vertex its elements' forces, both across chunks.
// Buffers rotate by three: this launch READS R (scattered by the launch before), scatters into W and clears Z (read by the launch
// before; nobody touches it now).  Saved per substep: a kernel boundary, the re-load of x / v / C / F_trial in p2g (they are
// registers), the store of F_trial (only the epilogue's plain g2p writes it: a pull always sees the end of a substep) and one
// record -> particle-loads chain per workgroup.  The host side (fast_step) keeps the g2p of the last substep PENDING and
// flushes it with a plain k_g2p whenever anything else looks at the particles (pull, re-sort, statistics, another dt, ...).
struct GridRead {  // the accumulator buffer g2p reads (GridPtrs g is the write side, as in k_p2g)
  float *mv, *col, *mov;
  int *col_flag;
};
template <int STEPS, bool FX>
__device__ __forceinline__ void g2p2g_body(const ChunkRec *recs, int n_chunks, const Bufs &b, const VAdj &va, const Dims &d, float rpic,
                                           float dt, const GridPtrs &g, const GridRead &rd, const SplatArgs &sa, const TradParams &tp,
                                           const GridParams &gp, const BCList &bcl, double *tile, int *esc, int &esc_n, float *red) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && g.host_sig) {  // progress + flags of the substep before (see p2g_body)
    int *prev = g.counters + CNT_PAR0 + 2 * ((g.step_id - 1) & 1);
    unsigned v = ((unsigned)g.step_id << 2) | (prev[0] != 0 ? 2u : 0u) | (prev[1] != 0 ? 1u : 0u);
    prev[0] = 0; prev[1] = 0;
    __hip_atomic_store(g.host_sig + SIG_RING0 + (g.step_id & (SIG_RING_N - 1)), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g.host_sig + SIG_PROGRESS, g.step_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if ((int)blockIdx.x >= sa.e0 && (int)blockIdx.x < sa.e0 + sa.n_extra) {  // splats of substep n + 1 (into W)
    int e = (int)blockIdx.x - sa.e0;
    if (e < sa.n_fbins) col_splat_wg<3>(tile, sa, e, d, g);
    else if (e < sa.n_fbins + sa.n_mov_wg) mover_splat_wg(b, sa.js, e - sa.n_fbins, d, g);
    return;
  }
  if ((int)blockIdx.x >= sa.z_first) {  // clearing of Z
    zero_blocks_wg(sa.z, (int)blockIdx.x - sa.z_first);
    return;
  }
  int w = xcd_slice((int)blockIdx.x - (sa.e0 == 0 ? sa.n_extra : 0), n_chunks);
  if (w < 0) return;
  __builtin_amdgcn_s_setprio(3);
  const ChunkRec cm = recs[w];
  asm volatile("" ::"s"(cm.e0), "s"(cm.t0), "s"(cm.v0));  // (the whole record with the first request, see g2p_body)
  int blk = cm.blk, chunk = cm.chunk;
  int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
  int ox = 4 * bx - 1, oy = 4 * by - 1, oz = 4 * bz - 1;
  int cls = 0, s = 0;
  bool valid = cm.map(chunk * CHUNK + (int)threadIdx.x, cls, s);
  valid = valid && cls == 1;  // (the host selects this kernel only for scenes without elements and vertices)
  const int sx = valid ? s : d.n_e, tx = sx - d.n_e;
  GridPtrs gr = g;  // the read side: same tables, the other accumulator buffer
  gr.mv = rd.mv; gr.col = rd.col; gr.mov = rd.mov; gr.col_flag = rd.col_flag;
  // ---- g2p of substep n: everything that depends on the record alone is loaded now (see g2p_body) ----
  V3 x = ld3(b.all, A_X, sx);
  // ... including what only the p2g half needs (mass, volume, Lame parameters, yield stress): loaded behind the g2p half's stores and the
  // barrier they were one more exposed memory level in the middle of a launch that is one workgroup's dependency chain long
  float pm_mass = b.all.at(A_MASS, sx), pm_vol = b.nv.at(N_VOL, sx), pm_mu = b.nv.at(N_MU, sx), pm_lam = b.nv.at(N_LAM, sx);
  float pm_ys = b.tr.at(T_YS, tx);
  constexpr int NPT = TILE3 / PT;
  int nbk[NPT], nlk[NPT];
  float am[NPT], apx[NPT], apy[NPT], apz[NPT];
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    int t = (int)threadIdx.x + u * PT;
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    int gx = ox + ti, gy = oy + tj, gz = oz + tk;
    bool in = in_grid(gx, gy, gz, d.G);
    nbk[u] = in ? blk_of(gx, gy, gz, d.NB) : -1;
    nlk[u] = loc_of(gx, gy, gz);
    const float *pm = gr.mv + ((size_t)(in ? nbk[u] : blk) * GCH_MV) * 64 + nlk[u];
    float a0 = pm[0], a1 = pm[64], a2 = pm[128], a3 = pm[192];
    am[u] = in ? a0 : 0.0f; apx[u] = in ? a1 : 0.0f; apy[u] = in ? a2 : 0.0f; apz[u] = in ? a3 : 0.0f;
  }
  unsigned long long col_mask = 0;
  unsigned bc_mask = 0;
  {
    int l = threadIdx.x & 63, fl = 0;
    if (l < 27 && gp.has_col) {
      int nx = bx + l / 9 - 1, ny = by + (l / 3) % 3 - 1, nz = bz + l % 3 - 1;
      if ((unsigned)nx < (unsigned)d.NB && (unsigned)ny < (unsigned)d.NB && (unsigned)nz < (unsigned)d.NB)
        fl = gr.col_flag[(nx * d.NB + ny) * d.NB + nz];
    }
    if (gp.has_col) col_mask = __ballot(fl != 0);
    for (int k = 0; k < bcl.n; ++k)
      if (bc_may_touch(bcl.bc[k], ox, oy, oz, ox + 7, oy + 7, oz + 7, d.G, d.dx, gp.time, gp.dt)) bc_mask |= 1u << k;
  }
  if (threadIdx.x == 0) esc_n = 0;
  bool escaped = false;
  if (valid) {
    int lx = (int)(x.x * d.inv_dx - 0.5f) - ox, ly = (int)(x.y * d.inv_dx - 0.5f) - oy, lz = (int)(x.z * d.inv_dx - 0.5f) - oz;
    escaped = (unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u;
  }
  float4 *vt = reinterpret_cast<float4 *>(tile);  // velocity tile of the g2p half; the same LDS is the p2g half's tile afterwards
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    int t = (int)threadIdx.x + u * PT;
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    V3 v = v3(0, 0, 0);
    if (nbk[u] >= 0) {
      int nc = 0, nm = 0;
      int nidx = ((((ox + ti) >> 2) - bx + 1) * 3 + (((oy + tj) >> 2) - by + 1)) * 3 + (((oz + tk) >> 2) - bz + 1);
      bool uc = (col_mask >> nidx) & 1ull;
      v = node_finish<false>(nbk[u], nlk[u], am[u], apx[u], apy[u], apz[u], d, gr, gp, bcl, nc, nm, uc, bc_mask);
    }
    vt[tile_idx(ti, tj, tk)] = make_float4(v.x, v.y, v.z, 0.0f);
  }
  asm volatile("" : "+v"(pm_mass), "+v"(pm_vol), "+v"(pm_mu), "+v"(pm_lam), "+v"(pm_ys));  // (here at the latest: they were issued before the accumulators)
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  const bool fit = valid && !escaped;
  V3 nx = x, nv = v3(0, 0, 0);
  M3 nC = m3_zero(), Ft = m3_identity();
  if (__any(fit)) {
    V3 xg = fit ? x : v3((float)(ox + 2) * d.dx, (float)(oy + 2) * d.dx, (float)(oz + 2) * d.dx);
    G2PResult r = g2p_gather<false>(vt, ox, oy, oz, xg, d);
    nv = r.v; nC = r.C;
    Ft = deform_update(r.F, dt, ld9(b.tr, T_F, tx));   // g2p_v :780-786
    float a_min = (1.0f / d.inv_dx) * 2.0f, a_max = d.grid_lim - (1.0f / d.inv_dx) * 2.0f;
    nx = x + dt * nv;
    nx = v3(fminf(fmaxf(nx.x, a_min), a_max), fminf(fmaxf(nx.y, a_min), a_max), fminf(fmaxf(nx.z, a_min), a_max));
    if (fit) {  // (F_trial stays in registers: see the header)
      st9(b.all, A_C, s, nC);
      st3(b.all, A_V, s, nv);
      st3(b.all, A_X, s, nx);
      int nbx = (int)(nx.x * d.inv_dx - 0.5f) - ox, nby = (int)(nx.y * d.inv_dx - 0.5f) - oy, nbz = (int)(nx.z * d.inv_dx - 0.5f) - oz;
      if ((unsigned)nbx > 5u || (unsigned)nby > 5u || (unsigned)nbz > 5u) raise_drift(g.counters, g.step_id);
    }
  }
  // A particle outside its tile margin (rare, and only until the re-sort its drift flag has requested) takes the global-memory
  // paths of both halves: the plain g2p update here, with everything stored, and p2g_escaped<true> -- which loads it back and runs
  // the stress update -- below (entry tagged with bit 16).
  if (__any(escaped)) {
    V3 xe = x;
    int se = s;
    asm volatile("" : "+v"(xe.x), "+v"(xe.y), "+v"(xe.z), "+v"(se));
    if (escaped) {
      G2PResult r = g2p_gather_global<true, false>(xe, d, gr, gp, bcl);
      g2p_write(b, 1, se, xe, v3(0, 0, 0), r, ox, oy, oz, d, dt, g);
      atomicAdd(g.counters + 0, 1);
      esc[atomicAdd(&esc_n, 1)] = (int)threadIdx.x | (1 << 16);
    }
  }
  __syncthreads();  // every wavefront is done with the velocity tile (and the escaped lanes' stores are visible in the workgroup)
  // ---- stress + p2g of substep n + 1, from registers ----
  for (int t = threadIdx.x; t < (FX ? 2 : 4) * TILE_PAD; t += PT) tile[t] = 0.0;
  if (fit) {  // early warning of the adaptive re-sort (see p2g_body)
    float la = g.lookahead * dt;
    int fx = (int)((nx.x + la * nv.x) * d.inv_dx - 0.5f) - ox, fy = (int)((nx.y + la * nv.y) * d.inv_dx - 0.5f) - oy,
        fz = (int)((nx.z + la * nv.z) * d.inv_dx - 0.5f) - oz;
    if ((unsigned)fx > 5u || (unsigned)fy > 5u || (unsigned)fz > 5u) raise_drift(g.counters, g.step_id);
  }
  // (what the two halves share goes through an empty asm: otherwise the optimizer hoists and keeps values of the second half
  // live across the gather of the first -- 227 VGPRs)
  asm volatile("" : "+v"(nx.x), "+v"(nx.y), "+v"(nx.z), "+v"(nv.x), "+v"(nv.y), "+v"(nv.z), "+v"(s));
  asm volatile("" : "+v"(nC.a00), "+v"(nC.a01), "+v"(nC.a02), "+v"(nC.a10), "+v"(nC.a11), "+v"(nC.a12), "+v"(nC.a20), "+v"(nC.a21), "+v"(nC.a22));
  asm volatile("" : "+v"(Ft.a00), "+v"(Ft.a01), "+v"(Ft.a02), "+v"(Ft.a10), "+v"(Ft.a11), "+v"(Ft.a12), "+v"(Ft.a20), "+v"(Ft.a21), "+v"(Ft.a22));
  P2GRaw raw;
  raw.x = nx; raw.v = nv; raw.C = nC; raw.S = Ft;
  raw.mass = pm_mass; raw.vol = pm_vol; raw.mu = pm_mu; raw.lam = pm_lam; raw.ys = pm_ys;  // (of slot s for every valid lane: only fit lanes use them)
#pragma unroll
  for (int u = 0; u < ADJ_BATCH; ++u) raw.ab.ent[u] = -1;
  P2GParticle q = p2g_finish<true>(raw, b, va, fit, 1, s, d, rpic, dt, false, tp);
  FxScale fs{1.0f, 1.0f, 1.0f, 1.0f};
  if (FX) {
    float bm, bp;
    fx_bounds(q, fit, bm, bp);
    fs = fx_scales(bm, bp, red);  // (barrier inside: also publishes the cleared tile)
    fx_apply(q, fs);
  } else {
    __syncthreads();
  }
  p2g_scatter<STEPS, FX>(tile, esc, &esc_n, q, fit, ox, oy, oz, d, g);
  __syncthreads();
  if (esc_n > 0) {
    for (int e = threadIdx.x; e < esc_n; e += PT) {
      int ec = 0, es = 0, ent = esc[e];
      if (!cm.map(chunk * CHUNK + (ent & 0xffff), ec, es)) continue;
      if (ent >> 16) p2g_escaped<true>(b, va, ec, es, d, rpic, dt, g, tp);  // left the margin before this launch: nothing of it ran yet
      else p2g_escaped<false>(b, va, ec, es, d, rpic, dt, g, tp);          // left it with this launch's move: its stress update ran above
    }
  }
  p2g_flush<false, false, FX>(tile, ox, oy, oz, d, g, fs);
}

}  // namespace fk
}  // namespace mpm
