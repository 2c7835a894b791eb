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
#include <hip/hip_ext.h>
#include <algorithm>
#include <array>
#include <cstring>
#include <string.h>
#include <cstdlib>
#include <dlfcn.h>

#include <rccl/rccl.h>

#include <atomic>
#include <thread>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "bc.hpp"
#include "ctx.hpp"
#include "mpm_math.hpp"

namespace mpm {

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 256;     // particles of one block handled by one workgroup of p2g / g2p (192, re-measured with
                               // the 92/95-VGPR kernels: sheet -1 us, garment and dense scenes 15 % slower; round 2, chunk size
                               // chosen per scene at run time: 128 / 64 are slower on every scene but demo-250 (-3 %), even on
                               // the 8k cube whose 256-particle chunks occupy a quarter of the CUs -- the cost is per workgroup:
                               // tile clear, two barriers, flush; profiles/r02_experiments.md)
constexpr int PT = CHUNK;      // threads of those workgroups (and of the extra workgroups riding in their launches)
constexpr int TILE = 8;        // tile edge in nodes: block (4) + 1 below + 3 above
constexpr int TILE3 = TILE * TILE * TILE;
inline unsigned nblk(size_t n) { return n ? (unsigned)((n + TPB - 1) / TPB) : 1u; }  // never an empty grid: kernels bound-check

// Kernel ablation switches and the per-workgroup timeline exist only in builds with -DMPMHIP_DEBUG=1
// (tools/build_variants.py dbg:-DMPMHIP_DEBUG=1, selected with MPMHIP_LIB): the production kernels carry neither the
// branches nor the stamps.
#ifndef MPMHIP_DEBUG
#define MPMHIP_DEBUG 0
#endif
#define DBG(g, bits) (MPMHIP_DEBUG && ((g).dbg & (bits)))
constexpr int WGT_MAX_WG = 16384, WGT_SLOTS = 8, WGT_KERNELS = 3;  // per-workgroup timeline: [kernel][workgroup][slot]
#if MPMHIP_DEBUG
// slot <- constant 100 MHz clock (the same on every CU and XCD), after everything issued before has completed; slot 7 of a
// workgroup holds where it ran (XCC_ID << 32 | HW_ID)
#define WGT(g, k, slot)                                                                                       \
  do {                                                                                                         \
    if ((g).trace && threadIdx.x == 0 && blockIdx.x < (unsigned)WGT_MAX_WG) {                                    \
      unsigned long long t_;                                                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
      (g).trace[((size_t)(k) * WGT_MAX_WG + blockIdx.x) * WGT_SLOTS + (slot)] = t_;                             \
      if ((slot) == 0)                                                                                         \
        (g).trace[((size_t)(k) * WGT_MAX_WG + blockIdx.x) * WGT_SLOTS + 7] =                                    \
            ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
    }                                                                                                          \
  } while (0)
#else
#define WGT(g, k, slot) do { } while (0)
#endif

// component-major array view: comp c of item i at p[c*n + i]
// Addressing: (uniform component base) + (zero-extended 32-bit byte offset of the item) -- the form the global_load / global_store
// "saddr" encoding takes (SGPR pair + one VGPR), instead of a 64-bit VGPR address per component (two more VGPRs and two more
// VALU instructions per access with the signed 64-bit index p[c * n + i]).  Arrays stay below 2^30 items per component.
struct Soa {
  float *p;
  int n;
  __device__ __forceinline__ float &at(int c, int i) const {
    return *reinterpret_cast<float *>(reinterpret_cast<char *>(p + (size_t)c * (size_t)n) + ((unsigned)i << 2));
  }
};
__device__ __forceinline__ V3 ld3(const Soa &a, int c0, int i) { return v3(a.at(c0, i), a.at(c0 + 1, i), a.at(c0 + 2, i)); }
__device__ __forceinline__ void st3(const Soa &a, int c0, int i, V3 v) {
  a.at(c0, i) = v.x; a.at(c0 + 1, i) = v.y; a.at(c0 + 2, i) = v.z;
}
__device__ __forceinline__ M3 ld9(const Soa &a, int c0, int i) {
  return M3{a.at(c0, i), a.at(c0 + 1, i), a.at(c0 + 2, i), a.at(c0 + 3, i), a.at(c0 + 4, i),
            a.at(c0 + 5, i), a.at(c0 + 6, i), a.at(c0 + 7, i), a.at(c0 + 8, i)};
}
__device__ __forceinline__ void st9(const Soa &a, int c0, int i, const M3 &m) {
  a.at(c0, i) = m.a00; a.at(c0 + 1, i) = m.a01; a.at(c0 + 2, i) = m.a02; a.at(c0 + 3, i) = m.a10;
  a.at(c0 + 4, i) = m.a11; a.at(c0 + 5, i) = m.a12; a.at(c0 + 6, i) = m.a20; a.at(c0 + 7, i) = m.a21;
  a.at(c0 + 8, i) = m.a22;
}

// component indices
enum { A_X = 0, A_V = 3, A_C = 6, A_MASS = 15, A_NC = 16 };                       // all particles
enum { N_STRESS = 0, N_VOL = 9, N_MU = 10, N_LAM = 11, N_NC = 12 };                // elements + traditional
enum { E_D = 0, E_RINV = 9, E_GAMMA = 12, E_KAPPA = 13, E_NC = 14 };               // elements
enum { T_F = 0, T_FT = 9, T_YS = 18, T_NC = 19 };                                  // traditional
enum { GCH_MV = 4, GCH_VOUT = 4, GCH_COL = 8, GCH_MOV = 4 };                       // grid channels per block

struct Bufs {
  Soa all, nv, el, tr;
  int *face_orig;  // [3][n_e] component-major original vertex-local ids
  int *sel;        // [n_p]
};

struct Dims {
  int n_p, n_e, n_nv, n_v, n_t;
  int G, NB;
  float dx, inv_dx, grid_lim;
};

__device__ __forceinline__ int blk_of(int x, int y, int z, int NB) { return ((x >> 2) * NB + (y >> 2)) * NB + (z >> 2); }
__device__ __forceinline__ int loc_of(int x, int y, int z) { return ((x & 3) << 4) | ((y & 3) << 2) | (z & 3); }
__device__ __forceinline__ bool in_grid(int x, int y, int z, int G) {
  return (unsigned)x < (unsigned)G && (unsigned)y < (unsigned)G && (unsigned)z < (unsigned)G;
}
// XCD-aware remap: consecutive workgroup ids land on different XCDs (observed: id % 8).  Work items are sorted by
// grid block, so runs of XCD_RUN consecutive items (neighbouring tiles) are given to the same XCD to share its L2,
// while successive runs rotate over the 8 XCDs so that a spatially concentrated load (e.g. the blocks around the
// body collider) is spread over the whole chip instead of landing on one or two XCDs.
constexpr int XCD_RUN = 16;
__device__ __forceinline__ int xcd_slice(int w, int n) {
  int xcd = w & 7, idx = w >> 3;
  int i = ((idx / XCD_RUN) * 8 + xcd) * XCD_RUN + (idx % XCD_RUN);
  return i < n ? i : -1;
}
inline unsigned xcd_grid(int n) { return (unsigned)(((n + 8 * XCD_RUN - 1) / (8 * XCD_RUN)) * (8 * XCD_RUN)); }

// ------------------------------------------------------------------------------------------------
// import / export between the caller's AoS arrays (reference layout) and the sorted SoA state
// ------------------------------------------------------------------------------------------------
__global__ void k_import(mpmhip_state_ptrs st, mpmhip_model_ptrs md, Bufs b, const int *perm, Dims d, int dist) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_p) return;
  int o = perm[s];
  for (int c = 0; c < 3; ++c) b.all.at(A_X + c, s) = st.particle_x[3 * (size_t)o + c];
  for (int c = 0; c < 3; ++c) b.all.at(A_V + c, s) = st.particle_v[3 * (size_t)o + c];
  for (int c = 0; c < 9; ++c) b.all.at(A_C + c, s) = st.particle_C[9 * (size_t)o + c];
  b.all.at(A_MASS, s) = st.particle_mass[o];
  // only selection == 0 is simulated (mpm_utils.py:492,725,797,1028).  The value 2 means "ghost copy of another rank's
  // particle" to the multi-GPU driver and only there; on a single context every nonzero value is "not simulated".
  int sel = st.particle_selection[o];
  b.sel[s] = dist ? sel : (sel != 0 ? 1 : 0);
  if (s < d.n_nv) {
    for (int c = 0; c < 9; ++c) b.nv.at(N_STRESS + c, s) = st.particle_stress[9 * (size_t)o + c];
    b.nv.at(N_VOL, s) = st.particle_vol[o];
    b.nv.at(N_MU, s) = md.mu[o];
    b.nv.at(N_LAM, s) = md.lam[o];
    if (s < d.n_e) {
      for (int c = 0; c < 9; ++c) b.el.at(E_D + c, s) = st.particle_d[9 * (size_t)o + c];
      for (int c = 0; c < 3; ++c) b.el.at(E_RINV + c, s) = st.particle_R_inv[3 * (size_t)o + c];
      b.el.at(E_GAMMA, s) = md.gamma[o];
      b.el.at(E_KAPPA, s) = md.kappa[o];
      for (int c = 0; c < 3; ++c) b.face_orig[c * d.n_e + s] = (int)st.faces[3 * (size_t)o + c];
    } else {
      int t = s - d.n_e;
      for (int c = 0; c < 9; ++c) b.tr.at(T_F + c, t) = st.particle_F[9 * (size_t)o + c];
      for (int c = 0; c < 9; ++c) b.tr.at(T_FT + c, t) = st.particle_F_trial[9 * (size_t)o + c];
      b.tr.at(T_YS, t) = md.yield_stress[o];
    }
  }
}

// Vertex forces without atomics: every element stores its corner forces f2, f3 (f1 = -(f2+f3), mpm_utils.py:
// 168-170) and every vertex sums over its incident (element, corner) pairs through an ELL adjacency table that is
// rebuilt in sorted index space at each re-sort.  Replaces the 9 scattered fp32 atomics per element of
// kirchoff_stress_Anisotropy (mpm_utils.py:173-175): scattered global atomics run at ~21 G/s on MI355X.
struct F3 { float x, y, z; };
struct VAdj {
  const int *adj;     // [K][n_v]: (element_slot << 2) | corner, -1 = empty
  const F3 *ef;       // [3][n_e] corner forces f1, f2, f3 per element + one zero entry at 3*n_e (12-byte loads)
  int K, n_v, n_e;
};
// One incidence = one 16-byte load: entry (e, c) reads ef[c*n_e + e]; empty entries read the zero slot, so there is no
// branch and all loads of a batch are in flight together.
constexpr int ADJ_BATCH = 8;
struct AdjBatch { int ent[ADJ_BATCH]; };
__device__ __forceinline__ AdjBatch adj_load(const VAdj &a, int vl, int k0) {
  AdjBatch r;
#pragma unroll
  for (int u = 0; u < ADJ_BATCH; ++u) r.ent[u] = (k0 + u < a.K) ? a.adj[(size_t)(k0 + u) * a.n_v + vl] : -1;
  return r;
}
__device__ __forceinline__ V3 adj_gather(const VAdj &a, const AdjBatch &r, V3 f) {
#pragma unroll
  for (int h = 0; h < ADJ_BATCH; h += 4) {  // four 16-byte loads in flight at a time (register budget of p2g)
    F3 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int ent = r.ent[h + u];
      g[u] = a.ef[ent < 0 ? 3 * a.n_e : (ent & 3) * a.n_e + (ent >> 2)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) f = f + v3(g[u].x, g[u].y, g[u].z);
  }
  return f;
}
__device__ __forceinline__ V3 vertex_force(const VAdj &a, int vl) {
  V3 f = v3(0, 0, 0);
  for (int k0 = 0; k0 < a.K; k0 += ADJ_BATCH) f = adj_gather(a, adj_load(a, vl, k0), f);
  return f;
}

__global__ void k_export(mpmhip_state_ptrs st, mpmhip_model_ptrs md, Bufs b, VAdj va, const int *perm,
                         Dims d, int export_model) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_p) return;
  int o = perm[s];
  for (int c = 0; c < 3; ++c) st.particle_x[3 * (size_t)o + c] = b.all.at(A_X + c, s);
  for (int c = 0; c < 3; ++c) st.particle_v[3 * (size_t)o + c] = b.all.at(A_V + c, s);
  for (int c = 0; c < 9; ++c) st.particle_C[9 * (size_t)o + c] = b.all.at(A_C + c, s);
  if (s < d.n_nv) {
    for (int c = 0; c < 9; ++c) st.particle_stress[9 * (size_t)o + c] = b.nv.at(N_STRESS + c, s);
    if (s < d.n_e) {
      for (int c = 0; c < 9; ++c) st.particle_d[9 * (size_t)o + c] = b.el.at(E_D + c, s);
    } else {
      int t = s - d.n_e;
      for (int c = 0; c < 9; ++c) st.particle_F[9 * (size_t)o + c] = b.tr.at(T_F + c, t);
      for (int c = 0; c < 9; ++c) st.particle_F_trial[9 * (size_t)o + c] = b.tr.at(T_FT + c, t);
      if (export_model) {
        md.yield_stress[o] = b.tr.at(T_YS, t);
        md.mu[o] = b.nv.at(N_MU, s);
        md.lam[o] = b.nv.at(N_LAM, s);
      }
    }
  } else {
    int v = s - d.n_nv;
    store_v3(st.vertex_force + 3 * (size_t)(o - d.n_nv), vertex_force(va, v));
  }
}

// ------------------------------------------------------------------------------------------------
// rebin: keys, permutation, block tables
// ------------------------------------------------------------------------------------------------
// key = class | state | block | cell ; `kf` packs the field widths (blk_bits | cell_bits << 8).
// state: 0 = simulated, 1 = ghost copy that gathers for itself (multi-GPU: g2p yes, p2g no), 2 = not transferred.
// PREDICTIVE SORT (cell_bits == 8): the block is the one the particle is expected to be in half a re-sort interval
// from now (x + lead * v, the shift clamped to one cell per axis), so that a coherently moving particle starts in the
// margin on one side of its block's tile and ends in the margin on the other: twice the travel before a re-sort is
// due.  The low bits then order by the CURRENT cell relative to that block's tile (6x6x6 positions), which is what
// the DPP pre-reduction of p2g wants to see in neighbouring lanes.  cell_bits == 6 (very large grids whose keys would
// not fit 32 bits otherwise): no prediction, cell = position inside the block.
typedef unsigned SortKey;  // class | state | block | tile cell: 2 + 2 + 18 + 8 = 30 bits at 256^3
__device__ __forceinline__ int kf_blk(int kf) { return kf & 255; }
__device__ __forceinline__ int kf_cell(int kf) { return kf >> 8; }
__device__ __forceinline__ int key_block(SortKey k, int kf) { return (int)((k >> kf_cell(kf)) & ((1u << kf_blk(kf)) - 1u)); }
__device__ __forceinline__ int key_state(SortKey k, int kf) { return (int)((k >> (kf_blk(kf) + kf_cell(kf))) & 3u); }
__device__ __forceinline__ bool key_inactive(SortKey k, int blk_bits) { return key_state(k, blk_bits) >= 2; }
__device__ __forceinline__ SortKey make_key(V3 x, V3 v, float lead, int cls, int state, const Dims &d, int kf) {
  int bb = kf_blk(kf), cb = kf_cell(kf);
  int cx = (int)(x.x * d.inv_dx - 0.5f), cy = (int)(x.y * d.inv_dx - 0.5f), cz = (int)(x.z * d.inv_dx - 0.5f);
  int bx = cx, by = cy, bz = cz;
  if (cb >= 8) {
    float lim = d.dx;  // at most one cell: the current cell must stay inside the predicted block's tile margin
    V3 xp = v3(x.x + fminf(fmaxf(lead * v.x, -lim), lim), x.y + fminf(fmaxf(lead * v.y, -lim), lim),
               x.z + fminf(fmaxf(lead * v.z, -lim), lim));
    int px = (int)(xp.x * d.inv_dx - 0.5f), py = (int)(xp.y * d.inv_dx - 0.5f), pz = (int)(xp.z * d.inv_dx - 0.5f);
    bx = min(max(px, cx - 1), cx + 1); by = min(max(py, cy - 1), cy + 1); bz = min(max(pz, cz - 1), cz + 1);
  }
  bx = min(max(bx, 0), d.G - 3); by = min(max(by, 0), d.G - 3); bz = min(max(bz, 0), d.G - 3);
  SortKey blk = (SortKey)blk_of(bx, by, bz, d.NB);
  SortKey cell;
  if (cb >= 8) {  // current cell in the predicted block's tile: 0..5 per axis when inside the margin (clamped otherwise)
    int lx = min(max(cx - (4 * (bx >> 2) - 1), 0), 5), ly = min(max(cy - (4 * (by >> 2) - 1), 0), 5), lz = min(max(cz - (4 * (bz >> 2) - 1), 0), 5);
    cell = (SortKey)((lx * 6 + ly) * 6 + lz);
  } else {
    cell = (SortKey)loc_of(bx, by, bz);
  }
  return ((SortKey)cls << (bb + cb + 2)) | ((SortKey)state << (bb + cb)) | (blk << cb) | cell;
}

// ---- the sort of the re-sort: LSD radix sort of (key, index) pairs, 8 bits a pass ----------------------------------------------
// rocPRIM sorts up to 2^20 pairs with a block sort + ~20 merge launches (115-135 us for the headline scene's 500k keys, half
// of a re-sort) and its Onesweep is slower still at this size: its decoupled look-back is a serial chain over the tiles
// (profiles/r02_experiments.md).  At this size every launch costs its 4-5 us of dispatch whatever it does, so a pass is TWO
// launches and nothing in them is a chain:
//   k_rs_hist     per-tile digit histogram [tile][digit], plus the same counts summed per GROUP of RS_GROUP tiles (integer atomics:
//                 the order of the adds does not matter);
//   k_rs_scatter  every workgroup works out by itself where its tile's pairs of each digit start -- pairs of smaller digits (a
//                 block scan over the digit totals) + pairs of this digit in earlier groups + in earlier tiles of its group:
//                 <= groups + RS_GROUP coalesced loads per thread instead of a scan launch -- and scatters.
// A tile is RS_TILE consecutive pairs, taken RS_TPB at a time in index order; the rank of a pair inside its tile = pairs of the
// same digit in earlier slices (run[]) + in earlier wavefronts of its slice (cnt[][]) + in lower lanes of its wavefront (ballot
// match).  Stable: the same permutation as rocPRIM's sort, bit for bit (tests/test_gpu_sort.py; MPMHIP_SORT=rocprim selects the
// library path).
#ifndef MPMHIP_RS_IPT
#define MPMHIP_RS_IPT 4
#endif
constexpr int RS_BITS = 8, RS_BINS = 1 << RS_BITS, RS_TPB = 256, RS_IPT = MPMHIP_RS_IPT, RS_TILE = RS_TPB * RS_IPT, RS_GROUP = 16;
static_assert(RS_TPB == RS_BINS, "one thread per digit value");

// lanes of this wavefront that hold the same 9-bit value (bit 8 = "no pair in this lane")
__device__ __forceinline__ unsigned long long rs_peers(int dg) {
  unsigned long long peers = ~0ull;
#pragma unroll
  for (int b = 0; b <= RS_BITS; ++b) {
    bool bit = (dg >> b) & 1;
    unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__global__ __launch_bounds__(RS_TPB) void k_rs_hist(const unsigned *keys, int n, int shift, int *hist, int *gsum) {
  __shared__ int h[RS_BINS];
  const int t = threadIdx.x, lane = t & 63;
  h[t] = 0;
  __syncthreads();
  const int base = (int)blockIdx.x * RS_TILE;
  unsigned kk[RS_IPT];
#pragma unroll
  for (int j = 0; j < RS_IPT; ++j) {
    int i = base + j * RS_TPB + t;
    kk[j] = i < n ? keys[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < RS_IPT; ++j) {
    bool in = base + j * RS_TPB + t < n;
    int dg = in ? (int)((kk[j] >> shift) & (RS_BINS - 1)) : RS_BINS;
    unsigned long long peers = rs_peers(dg);
    if (in && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[dg], __popcll(peers));
  }
  __syncthreads();
  int v = h[t];
  hist[(size_t)blockIdx.x * RS_BINS + t] = v;
  if (v) atomicAdd(gsum + (size_t)(blockIdx.x / RS_GROUP) * RS_BINS + t, v);
}

// IOTA: the values going in are 0, 1, 2, ... (first pass).  gsum_next: the group sums the NEXT pass accumulates, cleared here.
template <bool IOTA>
__global__ __launch_bounds__(RS_TPB) void k_rs_scatter(const unsigned *kin, const int *vin, unsigned *kout, int *vout, int n, int shift,
                                                       int n_tiles, const int *hist, const int *gsum, int *gsum_next, int clear_groups, int *mark, int mark_kf) {
  __shared__ int run[RS_BINS];          // where this tile's next pair of each digit goes
  __shared__ int cnt[2][4][RS_BINS];    // pairs of each digit in each wavefront of the current slice (double-buffered)
  __shared__ int ws[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int T = (int)blockIdx.x, base = T * RS_TILE, n_groups = (n_tiles + RS_GROUP - 1) / RS_GROUP, g0 = T / RS_GROUP;
  unsigned kk[RS_IPT];
  int vv[RS_IPT];
#pragma unroll
  for (int j = 0; j < RS_IPT; ++j) {
    int i = base + j * RS_TPB + t;
    kk[j] = i < n ? kin[i] : 0u;
    vv[j] = IOTA ? i : (i < n ? vin[i] : 0);
  }
  {  // thread t = digit t: start of this tile's pairs of that digit.  Fixed-size predicated batches: all loads of a batch are in
     // flight together (a loop with a run-time trip count issues them one latency after the other: +2 us per launch)
    int total = 0, pre = 0;
    for (int gb = 0; gb < n_groups; gb += 32) {
      int x[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) x[u] = gb + u < n_groups ? gsum[(size_t)(gb + u) * RS_BINS + t] : 0;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        total += x[u];
        pre += gb + u < g0 ? x[u] : 0;
      }
    }
    {
      int y[RS_GROUP - 1];
#pragma unroll
      for (int u = 0; u < RS_GROUP - 1; ++u) y[u] = g0 * RS_GROUP + u < T ? hist[(size_t)(g0 * RS_GROUP + u) * RS_BINS + t] : 0;
#pragma unroll
      for (int u = 0; u < RS_GROUP - 1; ++u) pre += y[u];
    }
    int inc = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int a = __shfl_up(inc, o);
      if (lane >= o) inc += a;
    }
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    int below = inc - total + (wv > 0 ? ws[0] : 0) + (wv > 1 ? ws[1] : 0) + (wv > 2 ? ws[2] : 0);
    run[t] = below + pre;
    for (int g = T; g < clear_groups; g += (int)gridDim.x) gsum_next[(size_t)g * RS_BINS + t] = 0;  // (every row: sorts of other sizes share the buffer)
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) (&cnt[0][0][0])[q * RS_TPB + t] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_IPT; ++j) {
    bool in = base + j * RS_TPB + t < n;
    int dg = in ? (int)((kk[j] >> shift) & (RS_BINS - 1)) : RS_BINS;
    unsigned long long peers = rs_peers(dg);
    int rank = __popcll(peers & ((1ull << lane) - 1));
    int(*c)[RS_BINS] = cnt[j & 1];
    if (in && rank == 0) c[wv][dg] = __popcll(peers);
    __syncthreads();
    if (in) {
      int off = run[dg] + rank;
      if (wv > 0) off += c[0][dg];
      if (wv > 1) off += c[1][dg];
      if (wv > 2) off += c[2][dg];
      kout[off] = kk[j];
      vout[off] = vv[j];
      // (last pass of the particle sort: flag the block of every transferred particle -- what k_mark_blocks would do next)
      if (mark && !key_inactive(kk[j], mark_kf)) mark[key_block(kk[j], mark_kf)] = 1;
    }
    __syncthreads();
    run[t] += c[0][t] + c[1][t] + c[2][t] + c[3][t];  // (read by the next slice after its first barrier)
    c[0][t] = 0; c[1][t] = 0; c[2][t] = 0; c[3][t] = 0;  // (written again two slices on: two barriers in between)
  }
}

__global__ void k_permute(Bufs src, Bufs dst, const int *order, const int *perm_src, int *perm_dst, int *inv, Dims d) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_p) return;
  int o = order[s];  // same class as s: the class is the key's most significant field
  for (int c = 0; c < A_NC; ++c) dst.all.at(c, s) = src.all.at(c, o);
  dst.sel[s] = src.sel[o];
  int po = perm_src[o];
  perm_dst[s] = po;
  inv[po] = s;
  if (s < d.n_nv) {
    for (int c = 0; c < N_NC; ++c) dst.nv.at(c, s) = src.nv.at(c, o);
    if (s < d.n_e) {
      for (int c = 0; c < E_NC; ++c) dst.el.at(c, s) = src.el.at(c, o);
      for (int c = 0; c < 3; ++c) dst.face_orig[c * d.n_e + s] = src.face_orig[c * d.n_e + o];
    } else {
      for (int c = 0; c < T_NC; ++c) dst.tr.at(c, s - d.n_e) = src.tr.at(c, o - d.n_e);
    }
  }
}

// cloth topology in sorted slots, one launch: thread i < n_e files the sorted (vertex-local) slots of element i's three
// vertices, thread i < n_v the sorted (element, corner) adjacency of vertex i
__global__ void k_topology_sorted(Bufs b, const int *inv, int *face_slot, const int *adj_o, int *adj_s, const int *perm, int K, Dims d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n_e)
    for (int c = 0; c < 3; ++c) face_slot[c * d.n_e + i] = inv[d.n_nv + b.face_orig[c * d.n_e + i]] - d.n_nv;
  if (i < d.n_v && adj_o) {
    int o = perm[d.n_nv + i] - d.n_nv;
    for (int k = 0; k < K; ++k) {
      int ent = adj_o[(size_t)k * d.n_v + o];
      adj_s[(size_t)k * d.n_v + i] = ent < 0 ? -1 : ((inv[ent >> 2] << 2) | (ent & 3));
    }
  }
}

// (the parameter named blk_bits below is the packed key format kf)

__global__ void k_mark_blocks(const SortKey *keys, int n, int blk_bits, int *pb_flag) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  SortKey k = keys[s];
  if (!key_inactive(k, blk_bits)) pb_flag[key_block(k, blk_bits)] = 1;
}

// rc: re-sort counts kept on the device so that the table kernels can be enqueued back to back without a host round trip
// (the host reads them once, at the end): [0] particle blocks, [1] active blocks, [2] chunks, [3] chunks incl. ghost copies,
// [4] any ghost copy, [5] capacity overflow bits (1 plist / ranges, 2 alist, 4 chunk records, 8 face bins), [6] face bins
enum { RC_NP = 0, RC_NA = 1, RC_NCH = 2, RC_NCHG = 3, RC_GHOST = 4, RC_OVER = 5, RC_NFB = 6, RC_N = 8 };
// compaction of the flagged blocks onto a list; thread 0 also files the total (rc[slot], overflow bit) and every thread
// clears its share of `clear` (the ranges table k_ranges fills next) -- both used to be launches of their own
__global__ void k_compact(const int *flag, const int *index, int n, int *list, int cap, int *rc, int slot, int over_bit, int *clear,
                          int n_clear) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = b; i < n_clear; i += (int)(gridDim.x * blockDim.x)) clear[i] = 0;
  if (b == 0 && rc) {
    int tot = index[n - 1] + flag[n - 1];
    rc[slot] = tot;
    if (tot > cap) atomicOr(rc + RC_OVER, over_bit);
  }
  if (b < n && flag[b] && index[b] < cap) list[index[b]] = b;
}

// The same compaction without a scan launch in front (the re-sort's two block lists; rocPRIM's scan is two launches): k_flag_count
// files the flagged blocks per tile of FC_TILE flags and per group of FC_GROUP tiles, k_compact_tiles works out every tile's start
// from those (uniform loads: <= groups + FC_GROUP scalars) and scans inside the tile.  index[] is filled as the exclusive scan
// would have filled it.
constexpr int FC_TILE = 1024, FC_GROUP = 16;
__global__ __launch_bounds__(256) void k_flag_count(const int *flag, int n, int *tcount, int *gsum) {
  __shared__ int ws[4];
  const int t = threadIdx.x, b0 = (int)blockIdx.x * FC_TILE + t * 4;
  int c = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) c += (b0 + u < n && flag[b0 + u]) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((t & 63) == 0) ws[t >> 6] = c;
  __syncthreads();
  if (t == 0) {
    int tot = ws[0] + ws[1] + ws[2] + ws[3];
    tcount[blockIdx.x] = tot;
    if (tot) atomicAdd(gsum + blockIdx.x / FC_GROUP, tot);
  }
}
__global__ __launch_bounds__(256) void k_compact_tiles(const int *flag, int n, const int *tcount, const int *gsum, int n_tiles, int *index,
                                                       int *list, int cap, int *rc, int slot, int over_bit, int *clear, int n_clear) {
  __shared__ int ws[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, T = (int)blockIdx.x, g0 = T / FC_GROUP;
  const int n_groups = (n_tiles + FC_GROUP - 1) / FC_GROUP;
  for (int i = T * 256 + t; i < n_clear; i += (int)gridDim.x * 256) clear[i] = 0;
  int total = 0, pre = 0;
  for (int g = 0; g < n_groups; ++g) {
    int x = gsum[g];
    total += x;
    pre += g < g0 ? x : 0;
  }
  for (int q = g0 * FC_GROUP; q < T; ++q) pre += tcount[q];
  if (T == 0 && t == 0) {
    rc[slot] = total;
    if (total > cap) atomicOr(rc + RC_OVER, over_bit);
  }
  const int b0 = T * FC_TILE + t * 4;
  int fl[4], c = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    fl[u] = (b0 + u < n && flag[b0 + u]) ? 1 : 0;
    c += fl[u];
  }
  int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int a = __shfl_up(inc, o);
    if (lane >= o) inc += a;
  }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  int idx = pre + inc - c + (wv > 0 ? ws[0] : 0) + (wv > 1 ? ws[1] : 0) + (wv > 2 ? ws[2] : 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (b0 + u >= n) break;
    index[b0 + u] = idx;
    if (fl[u]) {
      if (idx < cap) list[idx] = b0 + u;
      idx += 1;
    }
  }
}

// ranges[(cls*2+0)*n_P + slot] = first sorted index, [(cls*2+1)*n_P + slot] = one past the last
// n_P here is the STRIDE of the table (its capacity), not the number of particle blocks
__global__ void k_ranges(const SortKey *keys, Dims d, int blk_bits, const int *pb_index, int n_P, int *ranges) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_p) return;
  SortKey k = keys[s];
  if (key_inactive(k, blk_bits)) return;
  int cls = s < d.n_e ? 0 : (s < d.n_nv ? 1 : 2);
  int c0 = cls == 0 ? 0 : (cls == 1 ? d.n_e : d.n_nv), c1 = cls == 0 ? d.n_e : (cls == 1 ? d.n_nv : d.n_p);
  int slot = pb_index[key_block(k, blk_bits)];
  if (slot >= n_P) return;  // capacity overflow: flagged by k_compact_tiles, the host grows the tables and repeats
  int row = key_state(k, blk_bits) == 0 ? cls * 2 : (cls == 0 ? 6 : 8);  // ghosts: elements, vertices only
  int cb = kf_cell(blk_bits);
  if (s == c0 || (keys[s - 1] >> cb) != (k >> cb)) ranges[(row + 0) * n_P + slot] = s;
  if (s == c1 - 1 || (keys[s + 1] >> cb) != (k >> cb)) ranges[(row + 1) * n_P + slot] = s + 1;
}

__global__ void k_dilate(const int *plist, const int *rc, int cap_P, int NB, int *ab_flag) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int p = t / 27, nb = t % 27;
  if (p >= min(rc[RC_NP], cap_P)) return;
  int b = plist[p];
  int bz = b % NB, by = (b / NB) % NB, bx = b / (NB * NB);
  int x = bx + nb / 9 - 1, y = by + (nb / 3) % 3 - 1, z = bz + nb % 3 - 1;
  if ((unsigned)x < (unsigned)NB && (unsigned)y < (unsigned)NB && (unsigned)z < (unsigned)NB)
    ab_flag[(x * NB + y) * NB + z] = 1;
}

// ------------------------------------------------------------------------------------------------
// grid stage: grid_normalization_and_gravity (mpm_utils.py:561-572), damping (:1162-1174), mesh collide
// (mpm_solver.py:882-917), mover overwrite (:790-799), BCs in registration order (:487-501), and the re-zeroing of the
// accumulators (replaces zero_grid, :411-417).  In the fused substep there is no grid kernel: g2p evaluates the nodes
// of its tile on the fly (node_update<false>) and the accumulators are cleared by extra workgroups of the next
// substep's stress launch.
// ------------------------------------------------------------------------------------------------
// host-mapped signal words (FastState::h_sig / GridPtrs::host_sig)
enum { SIG_PROGRESS = 1, SIG_DFLAG = 2, SIG_DSEQ = 3, SIG_RING0 = 8, SIG_RING_N = 16, SIG_WORDS = 32 };
static_assert(SIG_RING0 + SIG_RING_N <= SIG_WORDS && (SIG_RING_N & (SIG_RING_N - 1)) == 0, "signal ring must fit its buffer");
// device counters (GridPtrs::counters): [0] particles outside their tile margin, [1] dropped contributions, [2] [3] collider /
// mover node counts, [4] active nodes, [5] a body face left its bin's tile (sticky), [6] drift flag (sticky; dist loops and the
// copy + event scheme read it), [7] all-reduced drift flag, [8] [9] experiment counters, [10]-[12] peer links,
// [CNT_PAR0 + 2 * parity + {0, 1}] the same two flags per substep parity: the kernels of substep s raise slot s & 1 and the
// p2g launch of substep s + 1 posts and clears it, so a ring entry holds exactly the flags of ONE finished substep (a plain
// snapshot of the sticky flags raced with the workgroups of the posting launch that raise them)
enum { CNT_FACE = 5, CNT_DRIFT = 6, CNT_PAR0 = 16, CNT_MMIN = 24, CNT_MMAX = 25, CNT_N = 32 };  // (MMIN / MMAX: smallest positive / largest
                                                                                             // particle mass as float bits, k_mass_span)

// Fused halo add (multi-GPU, peer-mapped halos): k_g2p<.., HALO = true> adds the neighbour rank's contribution to a shared
// block while it stages its tile -- own accumulator + the value the neighbour's pack stored into this rank's arena -- instead
// of a separate add kernel between p2g and g2p.  slot == nullptr: off.
constexpr int PEER_TAB = 8;
struct HaloIn {
  const int *slot;             // [blocks] -1, or (peer << 24) | index of the block in that peer's shared-block list
  const float *buf[PEER_TAB];  // this substep's receive buffer of each peer (arena of parity halo_seq & 1)
  const int *sig[PEER_TAB];    // its flag: reaches `seq` when the neighbour's pack of this substep has landed
  int n_peers, seq, ch;        // ch = 4 (m, momentum) or 8 (+ mover channels)
};

struct GridPtrs {
  float *mv;        // [block][4][64]: m, momentum xyz
  float *vout;      // [block][4][64]: v_out xyz, m (copy kept for introspection)
  float *col;       // [block][8][64]: weight, v_in xyz, normal xyz, pad
  float *mov;       // [block][4][64]: weight, velocity xyz
  const int *ab_flag;
  int *col_flag;    // [block] 1 = the body-face splat may have written this block's collider channels this substep
  int *m_flag;      // [block] 1 = p2g (or a halo sum) may have written this block's mass / momentum this substep
  int *counters;    // [0] particles outside their tile margin, [1] dropped contributions (inactive block)
  int *host_sig;    // host-mapped pinned memory: [SIG_PROGRESS] step_id of the newest k_p2g launch that started, [SIG_RING0 +
  int step_id;      // (step_id & 15)] the flags of substep step_id - 1 (see k_p2g); [SIG_DFLAG], [SIG_DSEQ] the sharded loop's
                    // reduced flag and its sequence number (k_post_flag)
  float lookahead;  // substeps the early warning of the adaptive re-sort looks ahead (k_p2g)
  HaloIn halo;      // multi-GPU: see HaloIn
  int stagger, stagger_groups, stagger_first;  // p2g: first-round workgroups wait (wave slot % groups) * stagger * 1024 cycles
  unsigned long long *trace;  // per-workgroup timeline (MPMHIP_DEBUG builds, mpmhip_debug_wgtrace); null otherwise
  int dbg;          // MPMHIP_DBG bitmask (MPMHIP_DEBUG builds only; perf experiments, results are wrong): 1 skip p2g flush, 2 skip the p2g
                    // scatter, 8 / 16 skip vertex-force / stress loads, 128 skip the LDS atomics only, 256 skip the splat workgroups, 2048 skip the clearing workgroups; 64 (results stay
                    // right) runs the stand-alone element finalize every substep instead of fusing it into the stress kernel
};

struct GridParams {
  float dt, gx, gy, gz, damping, time;
  int has_col, has_mov, mov_on;
  float col_friction;
  int count;
  // further mesh colliders (mpm_solver.py:385-419 loops over a list): they all splat the solver's one body mesh, so their
  // weight / velocity / normal fields are identical and only the friction of the collide step differs
  int n_col_more = 0;
  float col_friction_more[3] = {0.0f, 0.0f, 0.0f};
};

// smallest positive and largest particle mass of the simulated particles (float bits; positive floats order like ints)
__global__ void k_mass_span(const float *mass, const int *sel, int n, int *counters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float m = (i < n && sel[i] == 0) ? mass[i] : 0.0f;
  int lo = m > 0.0f ? __float_as_int(m) : 0x7f7fffff, hi = m > 0.0f ? __float_as_int(m) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(counters + CNT_MMIN, lo); atomicMax(counters + CNT_MMAX, hi); }
}

__device__ __forceinline__ void raise_drift(int *counters, int step_id) {
  counters[CNT_DRIFT] = 1;
  counters[CNT_PAR0 + 2 * (step_id & 1) + 1] = 1;
}
__device__ __forceinline__ void raise_face(int *counters, int step_id) {
  counters[CNT_FACE] = 1;
  counters[CNT_PAR0 + 2 * (step_id & 1)] = 1;
}

// One node of the grid stage.  ZERO = true consumes the accumulators (re-zeroes what it read); ZERO = false only reads
// them (g2p evaluates nodes on the fly while it stages its tile, k_zero_blocks / the zeroing workgroups of the next
// stress launch clear them afterwards).  Returns the node's v_out; m_out = accumulated mass.
// (the four accumulator values come in as arguments so that a caller can have issued their loads earlier: g2p does,
// together with its particle loads, to take one dependent memory level out of the head of every workgroup)
template <bool ZERO>
__device__ __forceinline__ V3 node_finish(int blk, int l, float m, float px, float py, float pz, const Dims &d, const GridPtrs &g,
                                          const GridParams &gp, const BCList &bcl, int &ncol, int &nmov, bool use_col,
                                          unsigned bc_mask, const float *rem_mov = nullptr) {
  V3 v = v3(0, 0, 0);
  if (m > 1e-15f) {
    float inv = 1.0f / m;
    v = v3(px * inv + gp.dt * gp.gx, py * inv + gp.dt * gp.gy, pz * inv + gp.dt * gp.gz);
  }
  if (gp.damping < 1.0f) v = v - (1.0f - gp.damping) * v;
  if (gp.has_col && use_col) {  // normalize_grid + collide, mpm_solver.py:882-917
    float *pc = g.col + ((size_t)blk * GCH_COL) * 64 + l;
    float wc = pc[0];
    if (wc != 0.0f) {
      V3 vin = v3(pc[64], pc[128], pc[192]), nrm = v3(pc[256], pc[320], pc[384]);
      if (wc > 1e-15f) {
        V3 vm = (1.0f / wc) * vin;
        v = collide_node(v, vm, nrm, gp.col_friction);
        for (int k = 0; k < gp.n_col_more; ++k) v = collide_node(v, vm, nrm, gp.col_friction_more[k]);
        ncol = 1;
      }
      if (ZERO) { pc[0] = 0.0f; pc[64] = 0.0f; pc[128] = 0.0f; pc[192] = 0.0f; pc[256] = 0.0f; pc[320] = 0.0f; pc[384] = 0.0f; }
    }
  }
  if (gp.has_mov && gp.mov_on) {
    float *pv = g.mov + ((size_t)blk * GCH_MOV) * 64 + l;
    float wv = pv[0], mx = 0.0f, my = 0.0f, mz = 0.0f;
    if (rem_mov) { wv += rem_mov[0]; mx = rem_mov[64]; my = rem_mov[128]; mz = rem_mov[192]; }  // the neighbour rank's share
    if (wv != 0.0f) {
      if (wv > 1e-15f) { v = (1.0f / wv) * v3(pv[64] + mx, pv[128] + my, pv[192] + mz); nmov = 1; }
      if (ZERO) { pv[0] = 0.0f; pv[64] = 0.0f; pv[128] = 0.0f; pv[192] = 0.0f; }
    }
  }
  if (bcl.n > 0 && bc_mask != 0u) {
    int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
    int gxn = 4 * bx + (l >> 4), gyn = 4 * by + ((l >> 2) & 3), gzn = 4 * bz + (l & 3);
    if (in_grid(gxn, gyn, gzn, d.G)) {
      size_t dense = ((size_t)gxn * d.G + gyn) * d.G + gzn;
      for (int k = 0; k < bcl.n; ++k)
        if ((bc_mask >> k) & 1u) apply_bc(bcl.bc[k], v, gxn, gyn, gzn, d.G, d.dx, gp.time, gp.dt, dense);
    }
  }
  return v;
}
template <bool ZERO>
__device__ __forceinline__ V3 node_update(int blk, int l, const Dims &d, const GridPtrs &g, const GridParams &gp,
                                          const BCList &bcl, float &m_out, int &ncol, int &nmov, bool use_col = true,
                                          unsigned bc_mask = 0xffffffffu) {
  float *pm = g.mv + ((size_t)blk * GCH_MV) * 64 + l;
  float m = pm[0], px = pm[64], py = pm[128], pz = pm[192];
  if (ZERO && (m != 0.0f || px != 0.0f || py != 0.0f || pz != 0.0f)) { pm[0] = 0.0f; pm[64] = 0.0f; pm[128] = 0.0f; pm[192] = 0.0f; }
  m_out = m;
  return node_finish<ZERO>(blk, l, m, px, py, pz, d, g, gp, bcl, ncol, nmov, use_col, bc_mask);
}

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

// Clear the accumulators a fused substep left loaded (what node_update<true> would have cleared).  The accumulators
// are double-buffered: substep n scatters into buffer n & 1, and the clearing of buffer (n - 1) & 1 rides in the p2g
// launch of substep n as extra workgroups -- it can run concurrently with the scatter because it touches the other
// buffer.  Stand-alone (k_zero_blocks) only before a re-sort: the active list is about to change.
struct ZeroArgs {
  const int *alist;
  int n_A, n_wg;  // n_wg workgroups clear 4 blocks each (0: nothing to clear)
  int has_col, has_mov;
  float *mv, *col, *mov;  // the buffer to clear
  int *m_flag, *col_flag;
};
__device__ __forceinline__ void zero_blocks_wg(const ZeroArgs &z, int wg) {
  int a = wg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6), l = threadIdx.x & 63;  // one block per wavefront
  if (a >= z.n_A) return;
  int blk = z.alist[a];
  if (z.m_flag[blk]) {  // wave-uniform
    float *pm = z.mv + ((size_t)blk * GCH_MV) * 64 + l;
    float m = pm[0], px = pm[64], py = pm[128], pz = pm[192];
    if (m != 0.0f || px != 0.0f || py != 0.0f || pz != 0.0f) { pm[0] = 0.0f; pm[64] = 0.0f; pm[128] = 0.0f; pm[192] = 0.0f; }
    if (l == 0) z.m_flag[blk] = 0;
  }
  if (z.has_col && z.col_flag[blk]) {  // wave-uniform
    float *pc = z.col + ((size_t)blk * GCH_COL) * 64 + l;
    if (pc[0] != 0.0f) { pc[0] = 0.0f; pc[64] = 0.0f; pc[128] = 0.0f; pc[192] = 0.0f; pc[256] = 0.0f; pc[320] = 0.0f; pc[384] = 0.0f; }
    if (l == 0) z.col_flag[blk] = 0;
  }
  if (z.has_mov) {
    float *pv = z.mov + ((size_t)blk * GCH_MOV) * 64 + l;
    if (pv[0] != 0.0f) { pv[0] = 0.0f; pv[64] = 0.0f; pv[128] = 0.0f; pv[192] = 0.0f; }
  }
}
__global__ __launch_bounds__(TPB) void k_zero_blocks(ZeroArgs z) { zero_blocks_wg(z, blockIdx.x); }

// Sort keys of all particles.  Tile-shaped like the sort's kernels (a workgroup = RS_TILE particles) because it also does the
// sort's first launch -- the digit histogram of the lowest RS_BITS bits (hist != nullptr) -- on the keys it has in registers,
// and clears the block flags and counts the table build starts from: two launches less per re-sort (a launch costs 4-5 us here
// whatever it does).
__global__ __launch_bounds__(RS_TPB) void k_keys(Bufs b, Dims d, int kf, float lead, int ghost_g2p, SortKey *keys, int *iota, int *clear,
                                                 int n_clear, int *hist, int *gsum, int n_tiles, ZeroArgs z) {
  __shared__ int h[RS_BINS];
  if ((int)blockIdx.x >= n_tiles) {  // behind the key workgroups: the grid accumulators of the OLD active list are cleared
    zero_blocks_wg(z, (int)blockIdx.x - n_tiles);  // (independent of everything else in a re-sort until the new list exists)
    return;
  }
  const int t = threadIdx.x, lane = t & 63, base = (int)blockIdx.x * RS_TILE;
  for (int i = (int)blockIdx.x * RS_TPB + t; i < n_clear; i += n_tiles * RS_TPB) clear[i] = 0;
  h[t] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_IPT; ++j) {
    int s = base + j * RS_TPB + t;
    bool in = s < d.n_p;
    SortKey k = 0;
    if (in) {
      int cls = s < d.n_e ? 0 : (s < d.n_nv ? 1 : 2);
      V3 x = ld3(b.all, A_X, s), v = ld3(b.all, A_V, s);
      int sel = b.sel[s];
      k = make_key(x, v, lead, cls, sel == 0 ? 0 : ((sel == 2 && ghost_g2p && cls != 1) ? 1 : 2), d, kf);
      keys[s] = k;
      iota[s] = s;
    }
    if (hist) {
      int dg = in ? (int)(k & (RS_BINS - 1)) : RS_BINS;
      unsigned long long peers = rs_peers(dg);
      if (in && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[dg], __popcll(peers));
    }
  }
  if (!hist) return;
  __syncthreads();
  int v = h[t];
  hist[(size_t)blockIdx.x * RS_BINS + t] = v;
  if (v) atomicAdd(gsum + (size_t)(blockIdx.x / RS_GROUP) * RS_BINS + t, v);
}


// ------------------------------------------------------------------------------------------------
// stress (compute_stress_from_F_trial, mpm_utils.py:1017-1105) on the sorted SoA state
// ------------------------------------------------------------------------------------------------
// FINALIZE = true fuses the tail of the previous substep's g2p_e (x, v = mean of the three updated vertices,
// d1, d2 = edges; mpm_utils.py:838-857) into this substep's stress kernel: one launch and one round trip of the
// director matrix less per substep.  The host runs the stand-alone k_elem_finalize instead whenever something needs
// finished elements earlier (re-sort, read-back, pre-p2g operations, joint-face splats, multi-GPU ghosts).
template <bool FINALIZE>
__device__ __forceinline__ void stress_elem_body(int e, const Bufs &b, F3 *ef, const Dims &d, float friction_coeff, const int *face_slot,
                                                 const SortKey *skeys, int blk_bits, int *counters, int step_id) {
  if (e >= d.n_e) return;
  if (b.sel[e] == 1) {  // not simulated (selection == 2 marks a ghost copy: stress yes, transfers no)
    for (int c = 0; c < 3; ++c) ef[c * d.n_e + e] = F3{0.0f, 0.0f, 0.0f};
    return;
  }
  M3 dm;
  if (FINALIZE) {
    int v1 = d.n_nv + face_slot[e], v2 = d.n_nv + face_slot[d.n_e + e], v3i = d.n_nv + face_slot[2 * d.n_e + e];
    V3 x1 = ld3(b.all, A_X, v1), x2 = ld3(b.all, A_X, v2), x3 = ld3(b.all, A_X, v3i);
    V3 u1 = ld3(b.all, A_V, v1), u2 = ld3(b.all, A_V, v2), u3 = ld3(b.all, A_V, v3i);
    st3(b.all, A_V, e, v3((u1.x + u2.x + u3.x) / 3.0f, (u1.y + u2.y + u3.y) / 3.0f, (u1.z + u2.z + u3.z) / 3.0f));
    V3 xe = v3((x1.x + x2.x + x3.x) / 3.0f, (x1.y + x2.y + x3.y) / 3.0f, (x1.z + x2.z + x3.z) / 3.0f);
    st3(b.all, A_X, e, xe);
    {  // drift check against the block this element was sorted into
      int blk = key_block(skeys[e], blk_bits);
      int oz = 4 * (blk % d.NB) - 1, oy = 4 * ((blk / d.NB) % d.NB) - 1, ox = 4 * (blk / (d.NB * d.NB)) - 1;
      // (no look-ahead here: an element follows its three vertices, whose g2p raises the flag early, see g2p_write)
      int nbx = (int)(xe.x * d.inv_dx - 0.5f) - ox, nby = (int)(xe.y * d.inv_dx - 0.5f) - oy, nbz = (int)(xe.z * d.inv_dx - 0.5f) - oz;
      if (b.sel[e] == 0 && ((unsigned)nbx > 5u || (unsigned)nby > 5u || (unsigned)nbz > 5u)) raise_drift(counters, step_id);
    }
    V3 d3o = v3(b.el.at(E_D + 2, e), b.el.at(E_D + 5, e), b.el.at(E_D + 8, e));
    V3 d1 = x2 - x1, d2 = x3 - x1;
    dm = m3_cols(d1, d2, d3o);
    // d1, d2 are not stored here: nothing reads them before the next finalize (every consumer of finished elements --
    // re-sort, read-back, ghosts -- runs k_elem_finalize first, which recomputes them from the vertices)
  } else {
    dm = ld9(b.el, E_D, e);
  }
  QR3 q = qr_cloth(dm);
  float gamma = b.el.at(E_GAMMA, e), kappa = b.el.at(E_KAPPA, e);
  float r02, r12, r22;
  V3 d3 = anisotropy_return_mapping(q, gamma, kappa, friction_coeff, r02, r12, r22);
  b.el.at(E_D + 2, e) = d3.x; b.el.at(E_D + 5, e) = d3.y; b.el.at(E_D + 8, e) = d3.z;
  M3 stress;
  V3 f1, f2, f3;
  kirchhoff_anisotropy(q, r02, r12, r22, d3, ld3(b.el, E_RINV, e), b.nv.at(N_VOL, e), b.nv.at(N_MU, e),
                       b.nv.at(N_LAM, e), gamma, kappa, stress, f1, f2, f3);
  st9(b.nv, N_STRESS, e, stress);
  ef[e] = F3{f1.x, f1.y, f1.z};
  ef[d.n_e + e] = F3{f2.x, f2.y, f2.z};
  ef[2 * d.n_e + e] = F3{f3.x, f3.y, f3.z};
}
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

// ------------------------------------------------------------------------------------------------
// p2g (p2g_apic_with_stress, mpm_utils.py:484-557): LDS tile accumulation per particle-block chunk
// ------------------------------------------------------------------------------------------------

// The LDS tile is stored with padded strides (i*99 + j*9 + k) so that the 27 nodes of a 3x3x3 stencil fall into
// different banks (99 = 3 mod 32, 9, 1).
constexpr int TS_I = 99, TS_J = 9;
constexpr int TILE_PAD = 768;  // entries per channel: 7*99 + 7*9 + 7 = 763 < 768
__device__ __forceinline__ int tile_idx(int i, int j, int k) { return i * TS_I + j * TS_J + k; }

// One chunk = up to 256 particles of ONE particle block, all three classes packed back to back (elements, then
// traditional, then vertices) so that lanes stay filled; lane t of chunk k takes combined index k*256 + t.
// The record is self-contained (48 bytes, one scalar load): a chunks -> plist -> ranges chain of three dependent
// loads in front of every particle load was a measurable part of p2g / g2p (both start with nothing else to do).
struct ChunkRec {
  int blk, chunk, e0, ne, t0, nt, v0, nv;
  int ge0, gne, gv0, gnv;  // ghost copies of the block (multi-GPU; only in the g2p list, empty in the p2g list)
  __device__ __forceinline__ bool map(int ci, int &cls, int &s) const {
    if (ci < ne) { cls = 0; s = e0 + ci; return true; }
    ci -= ne;
    if (ci < nt) { cls = 1; s = t0 + ci; return true; }
    ci -= nt;
    if (ci < nv) { cls = 2; s = v0 + ci; return true; }
    ci -= nv;
    if (ci < gne) { cls = 0; s = ge0 + ci; return true; }
    ci -= gne;
    if (ci < gnv) { cls = 2; s = gv0 + ci; return true; }
    return false;
  }
};

// Chunk records of all particle blocks, built on the device (one workgroup: a few thousand blocks at most): block p
// contributes ceil(particles / CHUNK) records to the p2g list and ceil((particles + ghost copies) / CHUNK) to the g2p list,
// in block order (neighbouring records = neighbouring tiles, what the XCD mapping wants).
// Thread t takes blocks t, t + 1024, ... (coalesced table reads), BC_R rounds at a time with all their loads in flight together:
// as one workgroup the kernel is a chain of memory latencies, and with one block after the other per thread it took 19-25 us.
constexpr int BC_R = 4;
__global__ __launch_bounds__(1024) void k_build_chunks(const int *plist, const int *ranges, int stride, int *rc, ChunkRec *recs,
                                                       ChunkRec *recs_g, int cap, int *counters) {
  __shared__ int sc[16], sg[16];
  __shared__ int any_ghost;
  const int t = threadIdx.x, n_P = min(rc[RC_NP], stride);
  const int lane = t & 63, wv = t >> 6;
  if (t == 0) any_ghost = 0;
  int base_c = 0, base_g = 0, gh = 0;  // records in front of the current round
  for (int p0 = 0; p0 < n_P; p0 += BC_R * 1024) {
    int R[BC_R][10], blk[BC_R];
#pragma unroll
    for (int r = 0; r < BC_R; ++r) {
      int p = p0 + r * 1024 + t;
      bool in = p < n_P;
      blk[r] = in ? plist[p] : 0;
#pragma unroll
      for (int k = 0; k < 10; ++k) R[r][k] = in ? ranges[(size_t)k * stride + p] : 0;
    }
#pragma unroll
    for (int r = 0; r < BC_R; ++r) {
      if (p0 + r * 1024 >= n_P) break;  // (uniform)
      ChunkRec q{blk[r], 0, R[r][0], R[r][1] - R[r][0], R[r][2], R[r][3] - R[r][2], R[r][4], R[r][5] - R[r][4], 0, 0, 0, 0};
      int tot = q.ne + q.nt + q.nv, g = (R[r][7] - R[r][6]) + (R[r][9] - R[r][8]);
      int c = (tot + CHUNK - 1) / CHUNK, cg = (tot + g + CHUNK - 1) / CHUNK;
      gh |= g > 0;
      // inclusive scan over the 1024 threads: inside each wavefront with shuffles, then over the 16 wavefront totals
      int ic = c, ig = cg;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        int a = __shfl_up(ic, o), b = __shfl_up(ig, o);
        if (lane >= o) { ic += a; ig += b; }
      }
      __syncthreads();
      if (lane == 63) { sc[wv] = ic; sg[wv] = ig; }
      __syncthreads();
      int tc = 0, tg = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        int a = sc[w], b = sg[w];
        if (w < wv) { ic += a; ig += b; }
        tc += a; tg += b;
      }
      int o = base_c + ic - c, og = base_g + ig - cg;
      for (int k = 0; k * CHUNK < tot; ++k, ++o) { q.chunk = k; if (o < cap) recs[o] = q; }
      q.ge0 = R[r][6]; q.gne = R[r][7] - R[r][6]; q.gv0 = R[r][8]; q.gnv = R[r][9] - R[r][8];
      tot += q.gne + q.gnv;
      for (int k = 0; k * CHUNK < tot; ++k, ++og) { q.chunk = k; if (og < cap) recs_g[og] = q; }
      base_c += tc; base_g += tg;
    }
  }
  if (gh) any_ghost = 1;
  __syncthreads();
  if (t == 0) {
    rc[RC_NCH] = base_c; rc[RC_NCHG] = base_g; rc[RC_GHOST] = any_ghost;
    if (base_c > cap || base_g > cap) atomicOr(rc + RC_OVER, 4);
    // the new order starts with no drift warning pending (two memsets after the host's wait before: 2 x 8 us of idle queue)
    counters[CNT_DRIFT] = 0;
    counters[CNT_PAR0] = 0; counters[CNT_PAR0 + 1] = 0; counters[CNT_PAR0 + 2] = 0; counters[CNT_PAR0 + 3] = 0;
  }
}

// ---- multi-GPU exchange helpers (mpmavatar_amd/dist.py drives them) ---------------------------------------
// halo: the (m, momentum) and mover channels of the grid blocks two ranks both have on their active lists
// One launch serves up to PEER_TAB neighbours: workgroups [wg_off[p], wg_off[p+1]) belong to peer p.
struct HaloTab {
  int n, with_mov;
  int wg_off[PEER_TAB + 1];
  const int *blocks[PEER_TAB];
  int n_blocks[PEER_TAB];
  float *buf[PEER_TAB];
  // peer-mapped halos (see "peer links" below): pack stores straight into the neighbour's receive buffer and the last
  // workgroup raises sig (a flag in the neighbour's memory) to seq; add waits for its own flag to reach seq.  null: none
  int *sig[PEER_TAB];
  int *cnt[PEER_TAB];
  int seq;
};
struct GhostTab {
  int n;
  int wg_off[PEER_TAB + 1];
  const int *ids_p[PEER_TAB], *ids_e[PEER_TAB];
  int n_p[PEER_TAB], n_e[PEER_TAB];
  float *buf[PEER_TAB];
};
template <class Tab>
__device__ __forceinline__ int tab_peer(const Tab &t, int wg) {
  int p = 0;
  while (p + 1 < t.n && wg >= t.wg_off[p + 1]) ++p;
  return p;
}
// ---- peer links: flags and data in fine-grained memory of the RECEIVING rank, mapped into the sender with HIP IPC ----------
// Producer: every thread fences its stores at system scope, the workgroup counts itself done, the last one to do so stores
// the flag with release semantics.  Consumer: one thread per workgroup polls the flag (acquire, system scope) with a
// wall-clock bound, so that a lost signal fails the run (counters[10]) instead of hanging the GPU.
constexpr long long LINK_TIMEOUT_TICKS = 20ll * 100000000ll;    // wall_clock64() ticks at 100 MHz: 20 s in a substep,
constexpr long long LINK_HANDSHAKE_TICKS = 3ll * 100000000ll;   // 3 s in the set-up handshake (failure = fall back to send/recv)
__device__ __forceinline__ void link_signal(int *cnt, int n_wg, int *flag, int seq) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == n_wg - 1) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__device__ __forceinline__ void link_wait(const int *flag, int seq, int *err, long long ticks = LINK_TIMEOUT_TICKS) {
  if (threadIdx.x == 0) {
    long long t0 = wall_clock64();
    while ((int)((unsigned)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)seq) < 0) {  // wraps
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > ticks) { *err = 1; break; }
    }
  }
  __syncthreads();
  (void)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);  // every thread orders its reads after the flag
}

// the pack of one workgroup.  L2 = true: the accumulators are read with agent-scope atomic loads (the caller runs in the
// SAME launch as the workgroups that scattered into them, see PackArgs: nothing may come from this CU's L1)
template <bool L2>
__device__ __forceinline__ void halo_pack_wg(const HaloTab &tb, const GridPtrs &g, int wg) {
  int p = tab_peer(tb, wg);
  int t = (wg - tb.wg_off[p]) * (int)blockDim.x + (int)threadIdx.x;
  int CH = tb.with_mov ? 8 : 4;
  if (t < tb.n_blocks[p] * CH * 64) {
    int l = t & 63, ch = (t >> 6) % CH, i = t / (CH * 64);
    int blk = tb.blocks[p][i];
    const float *src = ch < 4 ? g.mv + ((size_t)blk * GCH_MV + ch) * 64 + l : g.mov + ((size_t)blk * GCH_MOV + (ch - 4)) * 64 + l;
    tb.buf[p][t] = L2 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
  }
  if (tb.sig[p]) link_signal(tb.cnt[p], tb.wg_off[p + 1] - tb.wg_off[p], tb.sig[p], tb.seq);
}
// Halo pack INSIDE the p2g launch (multi-GPU, peer-mapped halos): the launch carries pack workgroups after its clearing
// workgroups.  Every workgroup in front of them (splats, chunks, XCD padding) counts itself done when its atomics are out; a pack
// workgroup waits for that count -- they are dispatched in order, so everything it waits for is resident or finished, no
// deadlock -- and then stores the shared blocks into the neighbour's arena and raises the neighbour's flag.  One launch
// (4-5 us at its floor) less per substep and rank than k_halo_pack.
struct PackArgs {
  HaloTab tb;
  unsigned *done;   // running count of finished workgroups (wraps)
  unsigned target;  // value it reaches when this launch's are all done
  int first, n_wg;  // pack workgroups: blockIdx in [first, first + n_wg); n_wg == 0: none
  int count;        // the scattering workgroups count themselves done (pack workgroups wait for them)
};
// What the waiting side reads are the accumulators, and those are only ever touched by device-scope atomics (performed at the
// memory side, coherent across the XCDs' L2s) -- so "done" needs no cache write-back: a workgroup waits until its own atomics
// are acknowledged (s_waitcnt vmcnt(0)) and then bumps a RELAXED counter.  (A release fence at agent scope instead costs an L2
// write-back per workgroup and those serialise: a launch of 5,100 workgroups took 427 us instead of 30, profiles/r03_experiments.md.)
// The count is spread over DONE_SHARDS addresses 64 bytes apart: arrivals on one address serialise too.
constexpr int DONE_SHARDS = 64, DONE_STRIDE = 16;
__device__ __forceinline__ void wg_done(const PackArgs &pk) {
  if (!pk.count) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(pk.done + (blockIdx.x & (DONE_SHARDS - 1)) * DONE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pack_wait(const PackArgs &pk, int *err) {
  if (threadIdx.x < 64) {  // wavefront 0: lane l reads shard l, the wavefront sums
    long long t0 = wall_clock64();
    for (;;) {
      unsigned v = __hip_atomic_load(pk.done + (threadIdx.x & (DONE_SHARDS - 1)) * DONE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if ((int)(v - pk.target) >= 0) break;
      __builtin_amdgcn_s_sleep(32);
      if (wall_clock64() - t0 > LINK_TIMEOUT_TICKS) { *err = 1; break; }
    }
  }
  __syncthreads();
}
// one lane waits for a peer's flag (g2p's out-of-margin path; the tile path waits per workgroup, link_wait)
__device__ __forceinline__ void link_wait_lane(const int *flag, int seq, int *err) {
  long long t0 = wall_clock64();
  while ((int)((unsigned)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)seq) < 0) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > LINK_TIMEOUT_TICKS) { *err = 1; break; }
  }
}
// the neighbour rank's share of node l of block blk (HaloIn): added to (m, px, py, pz); returns its mover channels or null
__device__ __forceinline__ const float *halo_add_node(const HaloIn &h, int hs, int l, float &m, float &px, float &py, float &pz) {
  int k = hs >> 24, idx = hs & 0xffffff;
  const float *base = h.buf[0];
#pragma unroll
  for (int q = 1; q < PEER_TAB; ++q) base = (k == q) ? h.buf[q] : base;
  const float *rp = base + ((size_t)idx * h.ch) * 64 + l;
  m += rp[0]; px += rp[64]; py += rp[128]; pz += rp[192];
  return h.ch == 8 ? rp + 256 : nullptr;
}
__device__ __forceinline__ const int *halo_sig(const HaloIn &h, int k) {
  const int *sg = h.sig[0];
#pragma unroll
  for (int q = 1; q < PEER_TAB; ++q) sg = (k == q) ? h.sig[q] : sg;
  return sg;
}

// ------------------------------------------------------------------------------------------------
// body-face splat (compute_mesh, mpm_solver.py:829-880) and joint splat (:677-788) into active blocks
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool splat_ok(int G, const Stencil &s) {
  return s.bx >= 0 && s.bx < G - 3 && s.by >= 0 && s.by < G - 3 && s.bz >= 0 && s.bz < G - 3;
}

// Body-mesh collider (compute_mesh, mpm_solver.py:829-880) with the same LDS-tile structure as p2g.  Faces are
// binned by grid block at each re-sort (rocPRIM sort of the centroid's block key).  Per substep one wavefront per
// ACTIVE block takes the faces binned there (lane = face: centroid, mean vertex velocity, unit normal with the
// caller's mesh advection applied), accumulates weight / weight*velocity / weight*normal into a 7-channel fp64
// LDS tile with ds_add_f64 and flushes the touched nodes to the block-major collider channels with coalesced
// atomics.  Faces in blocks outside the active list cannot reach a node that carries mass and are skipped; a
// face that drifted out of its tile margin since the last re-sort falls back to global atomics.
// (Tried and dropped: gathering the faces per node block inside the grid stage -- no atomics at all, but the few
// wavefronts next to the body serialise ~50 faces x 60 dependent instructions each and set the kernel's tail.)

__device__ __forceinline__ V3 face_centroid(const float *pts, const float *vel, float adv, const int32_t *idx, int f,
                                            V3 &p0, V3 &p1, V3 &p2) {
  int i0 = idx[3 * f], i1 = idx[3 * f + 1], i2 = idx[3 * f + 2];
  p0 = mesh_point(pts, vel, adv, i0); p1 = mesh_point(pts, vel, adv, i1); p2 = mesh_point(pts, vel, adv, i2);
  return v3((p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f);
}

__global__ void k_face_keys(const float *pts, const float *vel, float adv, const int32_t *idx, int n_f, Dims d,
                            unsigned *keys, int *iota) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_f) return;
  V3 p0, p1, p2;
  V3 fp = face_centroid(pts, vel, adv, idx, f, p0, p1, p2);
  // block in the high bits, cell of the block in the low six: faces of one cell end up in neighbouring lanes of the
  // splat workgroup, which pre-reduces runs of equal cells across lanes (col_splat_pass)
  int bx = min(max((int)(fp.x * d.inv_dx - 0.5f), 0), d.G - 1), by = min(max((int)(fp.y * d.inv_dx - 0.5f), 0), d.G - 1),
      bz = min(max((int)(fp.z * d.inv_dx - 0.5f), 0), d.G - 1);
  keys[f] = ((unsigned)blk_of(bx, by, bz, d.NB) << 6) | (unsigned)loc_of(bx, by, bz);
  iota[f] = f;
}

__global__ void k_face_bins(const unsigned *skeys, int n_f, int *fb_start, int *fb_cnt) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_f) return;
  unsigned k = skeys[j] >> 6;
  if (j == 0 || (skeys[j - 1] >> 6) != k) fb_start[k] = j;
  atomicAdd(fb_cnt + k, 1);
}

// non-empty face bins that lie on the active list (order irrelevant), as self-contained records
struct FaceBin { int blk, start, cnt, pad; };
__global__ void k_fbin_compact(const int *alist, const int *rc, int cap_A, const int *fb_start, const int *fb_cnt, FaceBin *list,
                               int cap_fbins) {
  const int n_A = min(rc[RC_NA], cap_A);
  int *counter = const_cast<int *>(rc) + RC_NFB;
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_A) return;
  int blk = alist[a];
  int cnt = fb_cnt[blk];
  if (cnt > 0) {
    int i = atomicAdd(counter, 1);
    if (i < cap_fbins) list[i] = FaceBin{blk, fb_start[blk], cnt, 0};
    else atomicOr(const_cast<int *>(rc) + RC_OVER, 8);
  }
}
// vertex ids of the faces in bin order (one indirection less per substep)
__global__ void k_face_sorted_idx(const int32_t *idx, const int *order, int n_f, int *fidx) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_f) return;
  int f = order[j];
  fidx[3 * j] = idx[3 * f]; fidx[3 * j + 1] = idx[3 * f + 1]; fidx[3 * j + 2] = idx[3 * f + 2];
}

// Joint splat (add_velocity_{traditional,verts,faces}, mpm_solver.py:677-788) as ONE launch: 32 lanes per joint
// particle, lane = stencil node (27 used), so every thread has a single short dependency chain instead of a 27-trip
// loop of dependent loads.  Group 0: the last n_t traditional particles, group 1: the first n_v vertices, group 2:
// the first n_f elements (caller-order indices; inv[] maps them to sorted slots).
struct JointSplatArgs {
  const float *vel_t, *vel_v, *vel_f;
  int n_t, n_v, n_f;
  int off_t, off_v;  // caller-order index of the first particle of group 0 / group 1 (group 2 starts at 0)
  const int *inv;    // caller order -> sorted slot
  const int *perm;   // sorted slot -> caller order
  int t_in_tile;     // 1: group 0 is splatted by the p2g chunks themselves (second tile pass), not by mover_splat_wg
};
__device__ __forceinline__ void mover_splat_wg(const Bufs &b, const JointSplatArgs &js, int wg, const Dims &d,
                                               const GridPtrs &g) {
  const int *inv = js.inv;
  int t = wg * PT + (int)threadIdx.x;
  int q = (t >> 5) + (js.t_in_tile ? js.n_t : 0), nn = t & 31;
  if (nn >= 27 || q >= js.n_t + js.n_v + js.n_f) return;
  const float *vel;
  int orig;
  if (q < js.n_t) { vel = js.vel_t + 3 * (size_t)q; orig = js.off_t + q; }
  else if (q < js.n_t + js.n_v) { vel = js.vel_v + 3 * (size_t)(q - js.n_t); orig = js.off_v + (q - js.n_t); }
  else { vel = js.vel_f + 3 * (size_t)(q - js.n_t - js.n_v); orig = q - js.n_t - js.n_v; }
  Stencil s = make_stencil(ld3(b.all, A_X, inv[orig]), d.inv_dx);
  if (!splat_ok(d.G, s)) return;  // mpm_solver.py:692,730,767
  int i = nn / 9, j = (nn / 3) % 3, k = nn % 3;
  float w = sel3(i, s.w0.x, s.w1.x, s.w2.x) * sel3(j, s.w0.y, s.w1.y, s.w2.y) * sel3(k, s.w0.z, s.w1.z, s.w2.z);
  int x = s.bx + i, y = s.by + j, z = s.bz + k;
  int blk = blk_of(x, y, z, d.NB);
  if (!g.ab_flag[blk]) { atomicAdd(g.counters + 1, 1); return; }
  V3 pv = load_v3(vel);
  float *p = g.mov + ((size_t)blk * GCH_MOV) * 64 + loc_of(x, y, z);
  atomicAdd(p, w);
  atomicAdd(p + 64, w * pv.x); atomicAdd(p + 128, w * pv.y); atomicAdd(p + 192, w * pv.z);
}

// The two splats are small, latency-bound and independent of the particle transfer, so they ride along in the p2g
// LAUNCH as extra workgroups (k_p2g: blockIdx < n_extra) instead of being kernels of their own: as separate launches
// they either sit on the critical path (17 us) or, on a side stream, cost two cross-queue barrier packets per
// substep (~6 us of idle GPU each, measured with rocprofv3 --kernel-trace).
struct SplatArgs {
  const float *pts, *vel;  // body mesh at this substep: pts + adv * vel
  float adv;
  const int *fidx;         // [n_f][3] vertex ids in bin order
  const FaceBin *fbins;
  int n_fbins;             // workgroups [0, n_fbins): one face bin each
  int splat_passes;        // 3: both passes of the body-face splat here; 2: only the normal pass (pass 0 rode in the stress launch)
  JointSplatArgs js;       // workgroups [n_fbins, n_fbins + n_mov_wg): joints
  int n_mov_wg;
  int n_extra;             // n_fbins + n_mov_wg rounded up to a multiple of 8 (keeps the XCD mapping of the chunks)
  int e0;                  // first workgroup of the splats: 0 (in front of the chunks) or xcd_grid(n_chunks) (behind them)
  ZeroArgs z;              // workgroups [z_first, z_first + z.n_wg), after the chunk workgroups: clear the other
  int z_first;             // accumulator buffer
  PackArgs pack;           // workgroups [pack.first, ...) after those: multi-GPU halo pack (see PackArgs)
};

// PASS 0: weight + weight*velocity (collider channels 0..3), PASS 1: weight*normal (channels 4..6); both passes use
// the 4-channel fp64 tile of p2g.
struct P2GParticle {
  Stencil s;
  float mass;
  float mass_s;  // mass * FxScale::sm (the mass channel of the fixed-point tile), else = mass
  V3 a0;    // v - dx * C' * fx
  M3 Cdx;   // dx * C'
  M3 Sdt;   // -dt * inv_dx * S   (elements: stress, traditional: vol*stress, vertices: 0)
  V3 vfdt;  // dt * vertex_force  (vertices only)
};

__device__ __forceinline__ P2GParticle p2g_zero(int ox, int oy, int oz, const Dims &d) {
  P2GParticle q;
  q.s = make_stencil(v3((float)(ox + 2) * d.dx, (float)(oy + 2) * d.dx, (float)(oz + 2) * d.dx), d.inv_dx);
  q.mass = 0.0f; q.mass_s = 0.0f; q.a0 = v3(0, 0, 0); q.Cdx = m3_zero(); q.Sdt = m3_zero(); q.vfdt = v3(0, 0, 0);
  return q;
}

// ---- the chunk tile in packed fixed point (template parameter FX; the default, MPMHIP_P2G_TILE=f64 selects the fp64 tile) ----
// The tile pass is bound by LDS atomic INSTRUCTIONS: 27 nodes x 4 channels per issuing lane, a ds_add_f64 costs the CU 6.7 +
// 0.17 x active lanes clocks and a ds_add_u64 7.5 + 0.04 x lanes (tools/ubench_lds_u64.hip; ds_add_f32 is 3 clocks PER LANE).  Two
// 32-bit fixed-point channels share one 64-bit integer add -- (mass | p_x) and (p_y | p_z): the sum of packed words is the packed
// word of the sums (two's complement: the signed low field borrows from the high one and the decode gives it back) -- so a node
// takes 2 atomics instead of 4, integer ones.  The scale of each field is a power of two chosen per workgroup so that the
// largest possible node sum of THIS chunk -- the sum over its lanes of a bound on |contribution| -- stays below 2^30: one unit
// is 2^-23..2^-22 of the sum of the chunk's largest contributions, i.e. an add is rounded like an fp32 add into a running sum
// of that size (what the reference's atomic_add does), and the sum itself is exact and order-independent.  Scaling by a power
// of two commutes with fp32 rounding: the DPP pre-reduction computes exactly what it computed before, times the scale.
struct FxScale { float sm, sp, inv_sm, inv_sp; };
__device__ __forceinline__ float fx_pow2(float bound, float &inv) {  // largest 2^k with bound * 2^k < 2^30, and 2^-k
  int eb = (__float_as_int(bound) >> 23) & 0xff;  // bound < 2^(eb - 126)
  eb = min(max(eb, 40), 240);
  inv = __int_as_float((eb - 29) << 23);
  return __int_as_float((283 - eb) << 23);
}
// bound on |what lane q adds to any one node|: mass channel, momentum channels (the largest of the three components)
__device__ __forceinline__ void fx_bounds(const P2GParticle &q, bool on, float &bm, float &bp) {
  const float W3 = 0.421875f, DW = 0.5625f;  // max w^3 (0.75^3), max |dw| w^2
  auto comp = [&](float a0, float cx, float cy, float cz, float s0, float s1, float s2, float vf) {
    return W3 * q.mass * (fabsf(a0) + 2.0f * (fabsf(cx) + fabsf(cy) + fabsf(cz))) + DW * (fabsf(s0) + fabsf(s1) + fabsf(s2)) +
           W3 * fabsf(vf);
  };
  const M3 &C = q.Cdx, &S = q.Sdt;
  float bx = comp(q.a0.x, C.a00, C.a01, C.a02, S.a00, S.a01, S.a02, q.vfdt.x);
  float by = comp(q.a0.y, C.a10, C.a11, C.a12, S.a10, S.a11, S.a12, q.vfdt.y);
  float bz = comp(q.a0.z, C.a20, C.a21, C.a22, S.a20, S.a21, S.a22, q.vfdt.z);
  bm = on ? W3 * q.mass : 0.0f;
  bp = on ? fmaxf(bx, fmaxf(by, bz)) : 0.0f;
}
// workgroup sums of the bounds -> the chunk's scales (contains the barrier that also publishes the cleared tile); red: 8 floats
__device__ __forceinline__ float dpp_shr_f(float v, int n);  // (defined with the DPP pre-reduction below)
// sum over the wavefront: inclusive DPP scan inside the four 16-lane rows, then the four row totals through v_readlane (a
// __shfl_xor butterfly is six dependent ds_bpermute round trips per value: it cost every chunk workgroup ~1 us of its ~10)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_shr_f(v, 1); v += dpp_shr_f(v, 2); v += dpp_shr_f(v, 4); v += dpp_shr_f(v, 8);
  int b = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(b, 15)) + __int_as_float(__builtin_amdgcn_readlane(b, 31))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 47)) + __int_as_float(__builtin_amdgcn_readlane(b, 63)));
}
__device__ __forceinline__ FxScale fx_scales(float bm, float bp, float *red) {
  bm = wave_sum(bm);
  bp = wave_sum(bp);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = bm; red[4 + (threadIdx.x >> 6)] = bp; }
  __syncthreads();
  float Bm = ((red[0] + red[1]) + (red[2] + red[3])) * 1.001f, Bp = ((red[4] + red[5]) + (red[6] + red[7])) * 1.001f;
  FxScale f;
  f.sm = fx_pow2(Bm, f.inv_sm);
  f.sp = fx_pow2(Bp, f.inv_sp);
  return f;
}
__device__ __forceinline__ void fx_apply(P2GParticle &q, const FxScale &f) {
  q.mass_s = q.mass * f.sm;
  q.a0 = f.sp * q.a0; q.Cdx = f.sp * q.Cdx; q.Sdt = f.sp * q.Sdt; q.vfdt = f.sp * q.vfdt;
}
__device__ __forceinline__ int fx_round(float x) {  // floor(x + 0.5)
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// All global loads of a particle are issued before anything waits on them: x, mass, C, v for every lane, stress / vol
// when the wavefront holds any element or traditional particle, the first ADJ_BATCH adjacency entries when it holds
// any vertex (wave-uniform branches; lanes of the other class read slot 0 and are masked afterwards).  The
// per-class `if` ladder this replaces serialised stress -> adjacency -> corner-force latencies.
// substeps the early warning looks ahead.  A warning raised in p2g(n) reaches the host with p2g(n + 1) and takes effect at
// most host_lead + 1 = 7 substeps later (fast_step), so 10 leave three in hand -- and every substep of look-ahead that is not
// needed is margin given away: with 20 the scenes re-sorted 2.4-2.9x as often in their fast-moving phases
// (profiles/r02_experiments.md).  The sharded loops see the all-reduced flag up to 16 + 4 substeps late and keep 20.
constexpr float DRIFT_LOOKAHEAD = 10.0f, DRIFT_LOOKAHEAD_DIST = 20.0f;
struct P2GRaw {
  V3 x, v;
  float mass, vol;
  M3 C, S;        // S: stress, or F_trial of a traditional particle when its stress update is fused into p2g (TRAD)
  float mu, lam, ys;  // TRAD only
  AdjBatch ab;
};
// TRAD = true fuses compute_stress_from_F_trial of the traditional particles (mpm_utils.py:1047-1103, k_stress_trad) into
// the front of p2g: the lane loads F_trial instead of the stress, runs the SVD / return mapping while the rest of its
// chunk's loads are still in flight, stores F / stress / hardening state exactly as the stand-alone kernel does and
// scatters with the fresh stress.  Same order of operations as the reference, one launch and one stress round trip less.
template <bool TRAD>
__device__ __forceinline__ P2GRaw p2g_issue(const Bufs &b, const VAdj &va, bool valid, int cls, int s, const Dims &d,
                                            bool w_nv, bool w_v) {
  P2GRaw r;
  int sa = valid ? s : 0;
  r.x = ld3(b.all, A_X, sa);
  r.mass = b.all.at(A_MASS, sa);
  r.C = ld9(b.all, A_C, sa);
  r.v = ld3(b.all, A_V, sa);
  r.S = m3_zero();
  r.vol = 1.0f;
  r.mu = r.lam = r.ys = 0.0f;
  if (w_nv) {
    int sn = (valid && cls != 2) ? s : 0;
    if (TRAD) {
      bool tr = valid && cls == 1;
      int t = tr ? s - d.n_e : 0;
      const float *base = tr ? b.tr.p + (size_t)T_FT * b.tr.n + t : b.nv.p + (size_t)N_STRESS * b.nv.n + sn;
      size_t st = tr ? (size_t)b.tr.n : (size_t)b.nv.n;
      r.S = M3{base[0], base[st], base[2 * st], base[3 * st], base[4 * st], base[5 * st], base[6 * st], base[7 * st], base[8 * st]};
      r.mu = b.nv.at(N_MU, sn);
      r.lam = b.nv.at(N_LAM, sn);
      if (d.n_t) r.ys = b.tr.at(T_YS, t);
    } else {
      r.S = ld9(b.nv, N_STRESS, sn);
    }
    r.vol = b.nv.at(N_VOL, sn);
  }
#pragma unroll
  for (int u = 0; u < ADJ_BATCH; ++u) r.ab.ent[u] = -1;
  if (w_v) r.ab = adj_load(va, (valid && cls == 2) ? s - d.n_nv : 0, 0);
  return r;
}
// Lanes without a particle (valid = false) loaded slot 0 and keep its (finite) stencil / velocity data with zero forces: they
// only have to stay finite -- their key is unique, so the segmented scan never merges them with a neighbour (a masked DPP
// step still multiplies the neighbour's value by 0.0) and they never issue an atomic.  No select between two particle
// records: hipcc lowers a select on the aggregate through scratch memory.
template <bool TRAD>
__device__ __forceinline__ P2GParticle p2g_finish(const P2GRaw &r, const Bufs &b, const VAdj &va, bool valid, int cls, int s,
                                                  const Dims &d, float rpic, float dt, bool w_v, const TradParams &tp) {
  V3 vf = v3(0, 0, 0);
  if (w_v) {
    vf = adj_gather(va, r.ab, vf);
    int vl = (valid && cls == 2) ? s - d.n_nv : 0;
    for (int k0 = ADJ_BATCH; k0 < va.K; k0 += ADJ_BATCH) vf = adj_gather(va, adj_load(va, vl, k0), vf);
  }
  P2GParticle q;
  q.s = make_stencil(r.x, d.inv_dx);
  q.mass = r.mass;
  q.mass_s = r.mass;
  M3 C = r.C;
  C = (1.0f - rpic) * C + (rpic / 2.0f) * (C - transpose(C));  // mpm_utils.py:530-532
  if (rpic < -0.001f) C = m3_zero();
  q.a0 = r.v - d.dx * (C * q.s.fx);
  q.Cdx = d.dx * C;
  q.Sdt = m3_zero();
  q.vfdt = v3(0, 0, 0);
  if (valid && cls == 0) {
    q.Sdt = (-dt * d.inv_dx) * r.S;
  } else if (valid && cls == 1) {
    M3 S = r.S;
    if (TRAD) {  // r.S holds F_trial: k_stress_trad's body
      int t = s - d.n_e;
      M3 F;
      float mu = r.mu, lam = r.lam, ys = r.ys;
      traditional_update(r.S, tp, mu, lam, ys, dt, F, S);
      if (tp.material == 1 || tp.material == 5) b.tr.at(T_YS, t) = ys;
      if (tp.material == 5) { b.nv.at(N_MU, s) = mu; b.nv.at(N_LAM, s) = lam; }
      st9(b.tr, T_F, t, F);
      st9(b.nv, N_STRESS, s, S);
    }
    q.Sdt = (-dt * d.inv_dx * r.vol) * S;
  } else if (valid) {
    q.vfdt = dt * vf;
  }
  return q;
}
// slow-path loader (escaped particles)
template <bool TRAD>
__device__ __forceinline__ P2GParticle p2g_load(const Bufs &b, const VAdj &va, int cls, int s, const Dims &d, float rpic,
                                                float dt, const TradParams &tp) {
  P2GRaw r = p2g_issue<TRAD>(b, va, true, cls, s, d, cls != 2, cls == 2);
  return p2g_finish<TRAD>(r, b, va, true, cls, s, d, rpic, dt, cls == 2, tp);
}

// ---- wave-level pre-reduction -------------------------------------------------------------------------
// After the cell sort neighbouring lanes mostly hold particles of the SAME cell, i.e. they add into the same
// 27 tile nodes.  Contributions are therefore summed across lanes first -- a segmented inclusive scan inside
// each 16-lane DPP row (row_shr 1,2,4,8; a segment = run of lanes with equal cell key) -- and only the last
// lane of every segment issues the LDS atomic.  Measured on MI355X the cost of a ds_add_f64 wave instruction
// is proportional to its active lanes (tools/ubench_lds_lanes.hip), and ds_add_f32 is ~10x slower than
// ds_add_f64 (tools/ubench_atomics.hip), hence fp64 accumulators in LDS for the splats -- and, cheaper still, two 32-bit
// fixed-point channels per ds_add_u64 for the particles ("the chunk tile in packed fixed point" above).
__device__ __forceinline__ int dpp_shr_i(int v, int old, int n) {  // lane l <- lane l-n of the same row, else old
  switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, false);
    case 2: return __builtin_amdgcn_update_dpp(old, v, 0x112, 0xf, 0xf, false);
    case 4: return __builtin_amdgcn_update_dpp(old, v, 0x114, 0xf, 0xf, false);
    default: return __builtin_amdgcn_update_dpp(old, v, 0x118, 0xf, 0xf, false);
  }
}
__device__ __forceinline__ float dpp_shr_f(float v, int n) { return __int_as_float(dpp_shr_i(__float_as_int(v), 0, n)); }

struct SegMask {
  float m1, m2, m4, m8;  // 1.0 where lane-d belongs to the same segment
  bool tail;             // last lane of its segment
};
__device__ __forceinline__ SegMask seg_masks(int key) {
  // m_d(l) = 1 iff lanes l-d .. l all carry the same key (one unbroken run).  Comparing key(l-d) with key(l) alone is
  // only equivalent while equal keys are contiguous, i.e. right after a re-sort: once particles have moved to other
  // cells of their block the lane order is no longer monotone (A B A ...), and the scan would jump over the B and add a
  // lane that also issues its own atomic.
  SegMask sm;
  int c1 = dpp_shr_i(key, ~key, 1) == key ? 1 : 0;
  int c2 = c1 & dpp_shr_i(c1, 0, 1);
  int c4 = c2 & dpp_shr_i(c2, 0, 2);
  int c8 = c4 & dpp_shr_i(c4, 0, 4);
  sm.m1 = c1 ? 1.0f : 0.0f;
  sm.m2 = c2 ? 1.0f : 0.0f;
  sm.m4 = c4 ? 1.0f : 0.0f;
  sm.m8 = c8 ? 1.0f : 0.0f;
  int next = __builtin_amdgcn_update_dpp(~key, key, 0x101, 0xf, 0xf, false);  // row_shl:1 -> lane l+1
  sm.tail = next != key;
  return sm;
}
__device__ __forceinline__ float seg_scan(float v, const SegMask &sm) {
  v = fmaf(dpp_shr_f(v, 1), sm.m1, v);
  v = fmaf(dpp_shr_f(v, 2), sm.m2, v);
  v = fmaf(dpp_shr_f(v, 4), sm.m4, v);
  v = fmaf(dpp_shr_f(v, 8), sm.m8, v);
  return v;
}
// The same scan for four values at once with the DPP source operand folded into the FMA
// (v_fmac_f32_dpp: dst += dpp(src0) * src1; lanes whose source falls outside the 16-lane row are left
// unchanged).  hipcc emits v_mov_b32_dpp + v_fmac_f32 for the C++ form above, i.e. twice the VALU issue slots,
// and this kernel is VALU-bound.  The four chains are interleaved so that a register written by one DPP op is
// read through DPP only three instructions later (gfx9 needs 2 wait states between a VALU write and a DPP
// read of the same VGPR); the leading s_nop covers values produced right before the block.
template <int STEPS>
__device__ __forceinline__ void seg_scan4(float &a, float &b, float &c, float &d, const SegMask &sm) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %1, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %2, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
      : "v"(sm.m1), "v"(sm.m2));
  if (STEPS >= 3)
    asm volatile(
        "s_nop 0\n"
        "v_fmac_f32_dpp %0, %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %1, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %2, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %3, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
        : "v"(sm.m4));
  if (STEPS >= 4)
    asm volatile(
        "s_nop 0\n"
        "v_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %1, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %2, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %3, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
        : "v"(sm.m8));
}

// contribution of q to stencil node (i,j,k) in the reference's form (mpm_utils.py:519-556); slow path only
__device__ __forceinline__ void p2g_node_ref(const P2GParticle &q, int i, int j, int k, float &wm, V3 &add) {
  const Stencil &s = q.s;
  float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x), wy = sel3(j, s.w0.y, s.w1.y, s.w2.y), wz = sel3(k, s.w0.z, s.w1.z, s.w2.z);
  float dwx = sel3(i, s.dw0.x, s.dw1.x, s.dw2.x), dwy = sel3(j, s.dw0.y, s.dw1.y, s.dw2.y), dwz = sel3(k, s.dw0.z, s.dw1.z, s.dw2.z);
  float weight = wx * wy * wz;
  V3 vel = q.a0 + (float)i * col0(q.Cdx) + (float)j * col1(q.Cdx) + (float)k * col2(q.Cdx);
  wm = weight * q.mass;
  add = wm * vel + q.Sdt * v3(dwx * wy * wz, wx * dwy * wz, wx * wy * dwz) + weight * q.vfdt;
}

// slow path for the (rare) particles that left their tile margin since the last re-sort: global atomics
template <bool TRAD>
__device__ __forceinline__ void p2g_escaped(const Bufs &b, const VAdj &va, int cls, int s, const Dims &d, float rpic,
                                         float dt, GridPtrs g, const TradParams &tp) {
  P2GParticle q = p2g_load<TRAD>(b, va, cls, s, d, rpic, dt, tp);
  atomicAdd(g.counters + 0, 1);
#pragma unroll 1
  for (int n = 0; n < 27; ++n) {
    int i = n / 9, j = (n / 3) % 3, k = n % 3;
    float wm;
    V3 add;
    p2g_node_ref(q, i, j, k, wm, add);
    int x = q.s.bx + i, y = q.s.by + j, z = q.s.bz + k;
    if (!in_grid(x, y, z, d.G)) continue;
    int blk = blk_of(x, y, z, d.NB);
    if (!g.ab_flag[blk]) { atomicAdd(g.counters + 1, 1); continue; }
    float *p = g.mv + ((size_t)blk * GCH_MV) * 64 + loc_of(x, y, z);
    g.m_flag[blk] = 1;
    atomicAdd(p, wm);
    atomicAdd(p + 64, add.x); atomicAdd(p + 128, add.y); atomicAdd(p + 192, add.z);
  }
}

// ---- pieces shared by the two p2g kernels -------------------------------------------------------------------
// tile pass of one chunk: margin check (out-of-margin lanes go to the esc list), DPP pre-reduction, LDS atomics
template <int STEPS, bool FX>
__device__ __forceinline__ void p2g_scatter(double *tile, int *esc, int *esc_n_p, P2GParticle &q, bool valid, int ox, int oy,
                                            int oz, const Dims &d, const GridPtrs &g) {
  int &esc_n = *esc_n_p;
  int key = -2 - (int)(threadIdx.x & 63), base = 0;
  if (valid) {
    int lx = q.s.bx - ox, ly = q.s.by - oy, lz = q.s.bz - oz;
    if ((unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u) {
      esc[atomicAdd(&esc_n, 1)] = (int)threadIdx.x;  // drifted out of the tile margin: handled after the tile pass
      valid = false;  // (keeps its finite values: unique key, no atomics -- see p2g_finish)
    } else {
      key = (lx * TILE + ly) * TILE + lz;
      base = tile_idx(lx, ly, lz);
    }
  }
  if (!DBG(g, 2) && __any(valid)) {  // wave-uniform: DPP needs converged lanes
    SegMask sm = seg_masks(key);
    // STEPS scan steps sum windows of 2^STEPS lanes: lanes at distances 0, W, 2W, ... from their segment's tail issue
    unsigned long long tails = __ballot(sm.tail);
    int dist = __ffsll((unsigned long long)(tails >> (threadIdx.x & 63))) - 1;
    bool do_add = valid && (dist & ((1 << STEPS) - 1)) == 0;
    if (DBG(g, 128)) do_add = false;
    if (DBG(g, 512)) {  // measurement: lanes that issue LDS atomics per lane that holds a particle
      unsigned long long ba = __ballot(do_add), bv = __ballot(valid);
      if ((threadIdx.x & 63) == 0) { atomicAdd(g.counters + 8, __popcll(ba)); atomicAdd(g.counters + 9, __popcll(bv)); }
    }
    const Stencil &st = q.s;
    // factored stencil: add_ijk = wz_k (wxym_ij (B_ij + k Cz) + P_ij) + dwz_k Q_ij,  wm = wxym_ij wz_k,  wxym = wx wy m
    V3 Cx = col0(q.Cdx), Cy = col1(q.Cdx), Cz = col2(q.Cdx);
    V3 S0 = col0(q.Sdt), S1 = col1(q.Sdt), S2 = col2(q.Sdt);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // x / y weights and all derivatives are recomputed from the fractional offsets where they are used: keeping the
      // stencil's 18 values live through the loop nest costs 12-16 VGPRs, i.e. a wavefront per SIMD
      float wx = bspline_w(i, st.fx.x), dwx = bspline_dw(i, st.fx.x);
      V3 Bi = q.a0 + (float)i * Cx;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float wy = bspline_w(j, st.fx.y), dwy = bspline_dw(j, st.fx.y);
        float wxy = wx * wy, wxym = wxy * q.mass, wxyms = wxy * q.mass_s;
        V3 Bij = Bi + (float)j * Cy;
        V3 T = wxym * Bij + ((dwx * wy) * S0 + (wx * dwy) * S1 + wxy * q.vfdt);
        V3 dT = wxym * Cz;
        V3 Q = wxy * S2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float wzk = sel3(k, st.w0.z, st.w1.z, st.w2.z), dwzk = bspline_dw(k, st.fx.z);
          float wm = wxyms * wzk;
          if (k > 0) T = T + dT;
          V3 add = wzk * T + dwzk * Q;
          float r0 = wm, r1 = add.x, r2 = add.y, r3 = add.z;
          seg_scan4<STEPS>(r0, r1, r2, r3, sm);
          if (do_add && FX) {
            unsigned long long *p = (unsigned long long *)tile + base + tile_idx(i, j, k);
            int i0 = fx_round(r0), i1 = fx_round(r1), i2 = fx_round(r2), i3 = fx_round(r3);  // (i0 >= 0: masses)
            atomicAdd(p, ((unsigned long long)(unsigned)i1 << 32) | (unsigned)i0);
            atomicAdd(p + TILE_PAD, ((unsigned long long)(unsigned)(i3 + (i2 >> 31)) << 32) | (unsigned)i2);
          } else if (do_add) {
            double *p = tile + base + tile_idx(i, j, k);
            atomicAdd(p, (double)r0);
            atomicAdd(p + TILE_PAD, (double)r1);
            atomicAdd(p + 2 * TILE_PAD, (double)r2);
            atomicAdd(p + 3 * TILE_PAD, (double)r3);
          }
        }
      }
    }
  }
}

// flush: skip untouched nodes; every touched node lies in an active block by construction.  REZERO leaves the tile
// cleared for the next chunk of a persistent workgroup.
template <bool REZERO, bool TO_MOV, bool FX>
__device__ __forceinline__ void p2g_flush(double *tile, int ox, int oy, int oz, const Dims &d, const GridPtrs &g, const FxScale &fs) {
  for (int t = threadIdx.x; t < TILE3; t += PT) {
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    float m, px, py, pz;
    if (FX) {
      unsigned long long *qd = (unsigned long long *)tile + tile_idx(ti, tj, tk);
      unsigned long long s0 = qd[0], s1 = qd[TILE_PAD];
      if ((s0 | s1) == 0ull) continue;
      if (REZERO) { qd[0] = 0ull; qd[TILE_PAD] = 0ull; }
      int im = (int)(unsigned)s0, ipx = (int)(s0 >> 32), ipy = (int)(unsigned)s1, ipz = (int)(s1 >> 32) + (ipy < 0 ? 1 : 0);
      m = (float)im * fs.inv_sm; px = (float)ipx * fs.inv_sp; py = (float)ipy * fs.inv_sp; pz = (float)ipz * fs.inv_sp;
    } else {
      double *qd = tile + tile_idx(ti, tj, tk);
      m = (float)qd[0]; px = (float)qd[TILE_PAD]; py = (float)qd[2 * TILE_PAD]; pz = (float)qd[3 * TILE_PAD];
      if (m == 0.0f && px == 0.0f && py == 0.0f && pz == 0.0f) continue;
      if (REZERO) { qd[0] = 0.0; qd[TILE_PAD] = 0.0; qd[2 * TILE_PAD] = 0.0; qd[3 * TILE_PAD] = 0.0; }
    }
    if (DBG(g, 1)) continue;
    int x = ox + ti, y = oy + tj, z = oz + tk;
    if (!in_grid(x, y, z, d.G)) continue;
    int nb = blk_of(x, y, z, d.NB);
    float *p = (TO_MOV ? g.mov : g.mv) + ((size_t)nb * 4) * 64 + loc_of(x, y, z);  // GCH_MV == GCH_MOV == 4
    if (!TO_MOV) g.m_flag[nb] = 1;
    atomicAdd(p, m);
    atomicAdd(p + 64, px); atomicAdd(p + 128, py); atomicAdd(p + 192, pz);
  }
}

// joint splat of one out-of-margin particle (second tile pass of k_p2g<.., JT = true>)
__device__ __forceinline__ void mover_escaped(V3 x, V3 pv, const Dims &d, const GridPtrs &g) {
  Stencil s = make_stencil(x, d.inv_dx);
#pragma unroll 1
  for (int n = 0; n < 27; ++n) {
    int i = n / 9, j = (n / 3) % 3, k = n % 3;
    float w = sel3(i, s.w0.x, s.w1.x, s.w2.x) * sel3(j, s.w0.y, s.w1.y, s.w2.y) * sel3(k, s.w0.z, s.w1.z, s.w2.z);
    int gx = s.bx + i, gy = s.by + j, gz = s.bz + k;
    int blk = blk_of(gx, gy, gz, d.NB);
    if (!g.ab_flag[blk]) { atomicAdd(g.counters + 1, 1); continue; }
    float *p = g.mov + ((size_t)blk * GCH_MOV) * 64 + loc_of(gx, gy, gz);
    atomicAdd(p, w);
    atomicAdd(p + 64, w * pv.x); atomicAdd(p + 128, w * pv.y); atomicAdd(p + 192, w * pv.z);
  }
}

// Faces are sorted by (block, cell of the centroid) at the re-sort, so neighbouring lanes mostly hold faces of the same
// cell and add into the same 27 tile nodes: the same segmented DPP pre-reduction as the particle scatter (p2g_scatter)
// leaves one lane per run issuing the LDS atomics.  DBG 4096 switches the pre-reduction off (every lane issues).
// Two passes through the four-channel tile per batch of faces -- (weight, weight * velocity), then weight * normal -- with
// the face, its stencil and the scan masks loaded / computed once for both (seven channels at once would need 43 KB of LDS:
// three instead of five workgroups per CU for the whole launch).
template <int PASS>
__device__ __forceinline__ void col_splat_scatter(double *tile, const Stencil &s, float on, V3 c, SegMask sm, bool do_add, int base) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float wx = sel3(i, s.w0.x, s.w1.x, s.w2.x) * on;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float wxy = wx * sel3(j, s.w0.y, s.w1.y, s.w2.y);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float w = wxy * sel3(k, s.w0.z, s.w1.z, s.w2.z);
        float r0 = w * c.x, r1 = w * c.y, r2 = w * c.z, r3 = w;
        seg_scan4<3>(r0, r1, r2, r3, sm);
        if (do_add) {
          double *p = tile + base + tile_idx(i, j, k);
          if (PASS == 0) {
            atomicAdd(p, (double)r3);
            atomicAdd(p + TILE_PAD, (double)r0); atomicAdd(p + 2 * TILE_PAD, (double)r1); atomicAdd(p + 3 * TILE_PAD, (double)r2);
          } else {
            atomicAdd(p, (double)r0); atomicAdd(p + TILE_PAD, (double)r1); atomicAdd(p + 2 * TILE_PAD, (double)r2);
          }
        }
      }
    }
  }
}
template <int PASS>
__device__ __forceinline__ void col_splat_flush(const double *tile, int ox, int oy, int oz, int bx, int by, int bz,
                                                unsigned long long act_mask, const Dims &d, const GridPtrs &g) {
  for (int t = threadIdx.x; t < TILE3; t += PT) {
    int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
    const double *q = tile + tile_idx(ti, tj, tk);
    float c0 = (float)q[0], c1 = (float)q[TILE_PAD], c2 = (float)q[2 * TILE_PAD];
    float c3 = PASS == 0 ? (float)q[3 * TILE_PAD] : 0.0f;
    if (PASS == 0 ? c0 == 0.0f : (c0 == 0.0f && c1 == 0.0f && c2 == 0.0f)) continue;
    int x = ox + ti, y = oy + tj, z = oz + tk;
    if (!in_grid(x, y, z, d.G)) continue;
    int nb = blk_of(x, y, z, d.NB);
    int nidx = (((x >> 2) - bx + 1) * 3 + ((y >> 2) - by + 1)) * 3 + ((z >> 2) - bz + 1);
    if (!((act_mask >> nidx) & 1ull)) continue;  // inactive block: never read by g2p, never re-zeroed
    float *p = g.col + ((size_t)nb * GCH_COL) * 64 + loc_of(x, y, z) + (PASS == 0 ? 0 : 256);
    atomicAdd(p, c0); atomicAdd(p + 64, c1); atomicAdd(p + 128, c2);
    if (PASS == 0) { atomicAdd(p + 192, c3); __hip_atomic_store(&g.col_flag[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}

constexpr int SPLAT_SMALL = 32;  // faces per bin up to which the splat workgroup maps lanes to (face, node) pairs
// PASSES: bit 0 = the weight / velocity pass (w, w v_face: collider channels 0-3, sets col_flag), bit 1 = the normal pass (w n:
// channels 4-6).  3 = both in one workgroup, as rounds 1-3 did.  Round 4: in cloth scenes the two passes ride in DIFFERENT
// launches -- pass 0 in front of the stress kernel, pass 1 in the p2g launch -- because a two-pass splat workgroup lives 10-17 us and
// set the length of the p2g launch in scenes that fit one round of workgroups (garment-120k: p2g 18 us for 10 us chunk
// workgroups), while the stress launch before it has room (9 us of streaming work, no LDS, one round).  Nothing reads the collider
// channels before g2p; the buffer they go into was cleared by the p2g launch of the substep before.
template <int PASSES>
__device__ __forceinline__ void col_splat_wg(double *tile, const SplatArgs &sa, int bin, const Dims &d, const GridPtrs &g) {
  const FaceBin fb = sa.fbins[bin];
  int blk = fb.blk;
  int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
  int ox = 4 * bx - 1, oy = 4 * by - 1, oz = 4 * bz - 1;
  // active flags of the 27 blocks the tile overlaps (lane n < 27 of every wavefront -> neighbour n)
  bool nb_act = false;
  const int l = threadIdx.x;
  if ((l & 63) < 27) {
    int n = l & 63;
    int x = bx + n / 9 - 1, y = by + (n / 3) % 3 - 1, z = bz + n % 3 - 1;
    if ((unsigned)x < (unsigned)d.NB && (unsigned)y < (unsigned)d.NB && (unsigned)z < (unsigned)d.NB)
      nb_act = g.ab_flag[(x * d.NB + y) * d.NB + z] != 0;
  }
  unsigned long long act_mask = __ballot(nb_act);
  const int end = fb.start + fb.cnt;
  if (fb.cnt <= SPLAT_SMALL) {
    // SMALL BIN (the common case once the cloth has draped: ~740 bins of ~27 faces): lane = (face, stencil node), 8 faces x 32
    // lanes (27 used) per step, <= 4 steps -- instead of lane = face with a 27-trip node loop of dependent DPP scans that 230
    // of the 256 lanes sit out.  Such a workgroup used to live 10-17 us (two 3 us scatter passes, profiles/r03_wg_timeline.md);
    // what is left is its chain of loads and the two flushes.  The per-step weights and normals stay in registers for the
    // second (normal) pass through the four-channel tile.
    const int fi = l >> 5, n = l & 31;
    const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
    float wk[SPLAT_SMALL / 8];
    V3 fnk[SPLAT_SMALL / 8];
    int basek[SPLAT_SMALL / 8];
    for (int t = l; t < 4 * TILE_PAD; t += PT) tile[t] = 0.0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SPLAT_SMALL / 8; ++it) {
      const int q = it * 8 + fi;
      const bool have = q < fb.cnt && n < 27;
      const int jq = q < fb.cnt ? fb.start + q : fb.start;
      int i0 = sa.fidx[3 * jq], i1 = sa.fidx[3 * jq + 1], i2 = sa.fidx[3 * jq + 2];
      V3 p0 = mesh_point(sa.pts, sa.vel, sa.adv, i0), p1 = mesh_point(sa.pts, sa.vel, sa.adv, i1), p2 = mesh_point(sa.pts, sa.vel, sa.adv, i2);
      V3 u0 = load_v3(sa.vel + 3 * i0), u1 = load_v3(sa.vel + 3 * i1), u2 = load_v3(sa.vel + 3 * i2);
      V3 fp = v3((p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f);
      V3 a = v3((u0.x + u1.x + u2.x) / 3.0f, (u0.y + u1.y + u2.y) / 3.0f, (u0.z + u1.z + u2.z) / 3.0f);
      V3 fn = normalize(cross(p1 - p0, p2 - p0));  // wp.mesh_eval_face_normal
      Stencil s = make_stencil(fp, d.inv_dx);
      const bool ok = have && splat_ok(d.G, s);  // mpm_solver.py:858
      const int lx = s.bx - ox, ly = s.by - oy, lz = s.bz - oz;
      const bool in_tile = !((unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u);
      const float w = sel3(ni, s.w0.x, s.w1.x, s.w2.x) * sel3(nj, s.w0.y, s.w1.y, s.w2.y) * sel3(nk, s.w0.z, s.w1.z, s.w2.z);
      wk[it] = 0.0f; fnk[it] = fn; basek[it] = 0;
      if (ok && in_tile) {
        wk[it] = w;
        basek[it] = tile_idx(lx + ni, ly + nj, lz + nk);
        if (PASSES & 1) {
          double *p = tile + basek[it];
          atomicAdd(p, (double)w);
          atomicAdd(p + TILE_PAD, (double)(w * a.x)); atomicAdd(p + 2 * TILE_PAD, (double)(w * a.y)); atomicAdd(p + 3 * TILE_PAD, (double)(w * a.z));
        }
      } else if (ok) {  // drifted out of the tile margin since the faces were binned: this lane's node through global atomics
        raise_drift(g.counters, g.step_id);
        raise_face(g.counters, g.step_id);
        int x = s.bx + ni, y = s.by + nj, z = s.bz + nk;
        int nb = blk_of(x, y, z, d.NB);
        if (g.ab_flag[nb]) {
          float *p = g.col + ((size_t)nb * GCH_COL) * 64 + loc_of(x, y, z);
          if (PASSES & 1) {
            __hip_atomic_store(&g.col_flag[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(p, w);
            atomicAdd(p + 64, w * a.x); atomicAdd(p + 128, w * a.y); atomicAdd(p + 192, w * a.z);
          }
          if (PASSES & 2) { atomicAdd(p + 256, w * fn.x); atomicAdd(p + 320, w * fn.y); atomicAdd(p + 384, w * fn.z); }
        }
      }
    }
    if (PASSES & 1) {
      __syncthreads();
      col_splat_flush<0>(tile, ox, oy, oz, bx, by, bz, act_mask, d, g);
    }
    if (!(PASSES & 2)) return;
    if (PASSES & 1) {
      __syncthreads();
      for (int t = l; t < 3 * TILE_PAD; t += PT) tile[t] = 0.0;
      __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < SPLAT_SMALL / 8; ++it)
      if (wk[it] != 0.0f) {
        double *p = tile + basek[it];
        atomicAdd(p, (double)(wk[it] * fnk[it].x)); atomicAdd(p + TILE_PAD, (double)(wk[it] * fnk[it].y));
        atomicAdd(p + 2 * TILE_PAD, (double)(wk[it] * fnk[it].z));
      }
    __syncthreads();
    col_splat_flush<1>(tile, ox, oy, oz, bx, by, bz, act_mask, d, g);
    return;
  }
  for (int j0 = fb.start; j0 < end; j0 += PT) {  // workgroup-uniform trip count: barriers and DPP need converged lanes
    for (int t = l; t < 4 * TILE_PAD; t += PT) tile[t] = 0.0;
    int jj = j0 + l;
    bool have = jj < end;
    int jq = have ? jj : fb.start;
    int i0 = sa.fidx[3 * jq], i1 = sa.fidx[3 * jq + 1], i2 = sa.fidx[3 * jq + 2];
    V3 p0 = mesh_point(sa.pts, sa.vel, sa.adv, i0), p1 = mesh_point(sa.pts, sa.vel, sa.adv, i1), p2 = mesh_point(sa.pts, sa.vel, sa.adv, i2);
    V3 u0 = load_v3(sa.vel + 3 * i0), u1 = load_v3(sa.vel + 3 * i1), u2 = load_v3(sa.vel + 3 * i2);
    V3 fp = v3((p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f);
    V3 a = v3((u0.x + u1.x + u2.x) / 3.0f, (u0.y + u1.y + u2.y) / 3.0f, (u0.z + u1.z + u2.z) / 3.0f);
    V3 fn = normalize(cross(p1 - p0, p2 - p0));  // wp.mesh_eval_face_normal
    Stencil s = make_stencil(fp, d.inv_dx);
    bool ok = have && splat_ok(d.G, s);  // mpm_solver.py:858
    int lx = s.bx - ox, ly = s.by - oy, lz = s.bz - oz;
    bool in_tile = !((unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u);
    bool tile_ok = ok && in_tile;
    // lanes without a face in the tile carry a unique key (never merged, never issue) and a zero contribution
    int key = tile_ok ? (lx * TILE + ly) * TILE + lz : -2 - (l & 63);
    int base = tile_ok ? tile_idx(lx, ly, lz) : 0;
    float on = tile_ok ? 1.0f : 0.0f;
    bool any = __any(tile_ok);
    SegMask sm = seg_masks(key);
    unsigned long long tails = __ballot(sm.tail);
    int dist = __ffsll((unsigned long long)(tails >> (l & 63))) - 1;
    bool do_add = tile_ok && (dist & 7) == 0;
    if (DBG(g, 4096)) { sm.m1 = sm.m2 = sm.m4 = sm.m8 = 0.0f; do_add = tile_ok; }
    __syncthreads();
    if (any && (PASSES & 1)) col_splat_scatter<0>(tile, s, on, a, sm, do_add, base);
    if (ok && !in_tile) {  // drifted out of the tile margin since the faces were binned
      raise_drift(g.counters, g.step_id);
      raise_face(g.counters, g.step_id);  // ... which is what makes the next re-sort bin the faces again (rebin)
#pragma unroll 1
      for (int n = 0; n < 27; ++n) {
        int i = n / 9, j = (n / 3) % 3, k = n % 3;
        float w = sel3(i, s.w0.x, s.w1.x, s.w2.x) * sel3(j, s.w0.y, s.w1.y, s.w2.y) * sel3(k, s.w0.z, s.w1.z, s.w2.z);
        int x = s.bx + i, y = s.by + j, z = s.bz + k;
        int nb = blk_of(x, y, z, d.NB);
        if (g.ab_flag[nb]) {
          float *p = g.col + ((size_t)nb * GCH_COL) * 64 + loc_of(x, y, z);
          if (PASSES & 1) {
            __hip_atomic_store(&g.col_flag[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(p, w);
            atomicAdd(p + 64, w * a.x); atomicAdd(p + 128, w * a.y); atomicAdd(p + 192, w * a.z);
          }
          if (PASSES & 2) { atomicAdd(p + 256, w * fn.x); atomicAdd(p + 320, w * fn.y); atomicAdd(p + 384, w * fn.z); }
        }
      }
    }
    if (PASSES & 1) {
      __syncthreads();
      col_splat_flush<0>(tile, ox, oy, oz, bx, by, bz, act_mask, d, g);
    }
    if (PASSES == 3) {
      __syncthreads();
      for (int t = l; t < 3 * TILE_PAD; t += PT) tile[t] = 0.0;
    }
    if (PASSES & 2) {
      __syncthreads();
      if (any) col_splat_scatter<1>(tile, s, on, fn, sm, do_add, base);
      __syncthreads();
      col_splat_flush<1>(tile, ox, oy, oz, bx, by, bz, act_mask, d, g);
    }
    __syncthreads();
  }
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

// JT = true: the mover holds MANY traditional particles (run_demo.py keeps 100k sand particles frozen for the first
// frames); their joint splat (weight, weight * joint velocity into the mover channels, mpm_solver.py:677-704) is a
// second pass through the same LDS tile by the chunk that owns them instead of 27 x 4 scattered global atomics each.
template <int STEPS, bool TRAD, bool JT, bool FX>
__device__ __forceinline__ void p2g_body(const ChunkRec *recs, int n_chunks, const Bufs &b, const VAdj &va, const Dims &d, float rpic,
                                         float dt, const GridPtrs &g, const SplatArgs &sa, const TradParams &tp, double *tile, int *esc,
                                         int &esc_n, float *red) {
  WGT(g, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0 && g.host_sig) {
    // progress + drift flag for the host (plain stores into pinned host memory instead of a copy + event every few
    // substeps: on the stream those cost a blit kernel and ~10-20 us of idle queue each).  Everything before this launch
    // has completed, so substep step_id - 1 is done and its parity slot of the flags holds every warning it raised (final: the
    // kernels of THIS substep raise the other slot); post it and clear it for substep step_id + 1.
    // One ring entry per launch -- (step_id, a body face left its bin's tile, drift flag) -- and the progress word after it.
    // The host decides at substep s with the entry of substep s - host_lead, whatever the GPU has done since: the re-sort
    // schedule is a function of the simulation, not of host / GPU timing, and a run stays bit-reproducible.
    int *prev = g.counters + CNT_PAR0 + 2 * ((g.step_id - 1) & 1);
    unsigned v = ((unsigned)g.step_id << 2) | (prev[0] != 0 ? 2u : 0u) | (prev[1] != 0 ? 1u : 0u);
    prev[0] = 0; prev[1] = 0;
    __hip_atomic_store(g.host_sig + SIG_RING0 + (g.step_id & (SIG_RING_N - 1)), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g.host_sig + SIG_PROGRESS, g.step_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // The splat workgroups go in front of the chunks (e0 = 0: the longest workgroups of the launch start first) or behind them
  // (e0 = xcd_grid(n_chunks), MPMHIP_SPLAT_FIRST_MAX): measured the same to 1 % early and in the draped state, where ~740 of them
  // take more than half of the first-round slots -- the dispatcher evens it out.
  if ((int)blockIdx.x >= sa.e0 && (int)blockIdx.x < sa.e0 + sa.n_extra) {
    int e = (int)blockIdx.x - sa.e0;
    if (DBG(g, 256)) return;
    if (e < sa.n_fbins) {   // (8192 / 16384: ablation switches)
      if (DBG(g, 8192)) {}
      else if (sa.splat_passes == 2) col_splat_wg<2>(tile, sa, e, d, g);   // (pass 0 rode in the stress launch)
      else col_splat_wg<3>(tile, sa, e, d, g);
    }
    else if (e < sa.n_fbins + sa.n_mov_wg) { if (!DBG(g, 16384)) mover_splat_wg(b, sa.js, e - sa.n_fbins, d, g); }
    WGT(g, 0, 6);
    wg_done(sa.pack);
    return;
  }
  if ((int)blockIdx.x >= sa.z_first) {  // ... and the clearing workgroups last: they fill the tail of the launch
    if (sa.pack.n_wg && (int)blockIdx.x >= sa.pack.first) {  // (multi-GPU) halo pack, once everything in front has scattered
      pack_wait(sa.pack, g.counters + 10);
      halo_pack_wg<true>(sa.pack.tb, g, (int)blockIdx.x - sa.pack.first);
      return;
    }
    if (!DBG(g, 2048)) zero_blocks_wg(sa.z, (int)blockIdx.x - sa.z_first);
    WGT(g, 0, 6);
    return;
  }
  int w = xcd_slice((int)blockIdx.x - (sa.e0 == 0 ? sa.n_extra : 0), n_chunks);
  if (w < 0) { wg_done(sa.pack); return; }
  if (g.stagger > 0 && (int)blockIdx.x < g.stagger_first) {
    // The workgroups of the first round all start within a microsecond, load together and then scatter together: memory
    // system and VALU / LDS pipelines take turns idling, and a first-round workgroup lives 12.4 us against 9.0 us for one
    // of the desynchronised second round (profiles/r03_wg_timeline.md).  Stagger them per CU by the wave slot they landed in.
    int slot = (int)(__builtin_amdgcn_s_getreg(63492) & 0xfu) % g.stagger_groups;
    for (int i = 0; i < slot * g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  const ChunkRec cm = recs[w];
  int blk = cm.blk, chunk = cm.chunk;
  int bz = blk % d.NB, by = (blk / d.NB) % d.NB, bx = blk / (d.NB * d.NB);
  int ox = 4 * bx - 1, oy = 4 * by - 1, oz = 4 * bz - 1;
  int cls = 0, s = 0;
  bool valid = cm.map(chunk * CHUNK + (int)threadIdx.x, cls, s);
  // issue the particle loads before the tile is cleared so that their latency overlaps
  bool w_nv = __any(valid && cls != 2), w_v = __any(valid && cls == 2);
  if (DBG(g, 8)) w_v = false;
  if (DBG(g, 16)) w_nv = false;
  WGT(g, 0, 1);  // chunk record here
  P2GRaw raw = p2g_issue<TRAD>(b, va, valid, cls, s, d, w_nv, w_v);
  for (int t = threadIdx.x; t < ((FX && !JT) ? 2 : 4) * TILE_PAD; t += PT) tile[t] = 0.0;  // (fixed point: two 64-bit words per node; the joint pass needs all four)
  if (threadIdx.x == 0) esc_n = 0;
  if (valid) {  // early warning for the adaptive re-sort: will this particle still fit the tile DRIFT_LOOKAHEAD substeps
                // from now (the host reads the flag with a lag of up to 16 substeps)?  The out-of-margin paths work
                // but cost ~100 scattered global atomics per particle and substep.
    float la = g.lookahead * dt;
    int fx = (int)((raw.x.x + la * raw.v.x) * d.inv_dx - 0.5f) - ox, fy = (int)((raw.x.y + la * raw.v.y) * d.inv_dx - 0.5f) - oy,
        fz = (int)((raw.x.z + la * raw.v.z) * d.inv_dx - 0.5f) - oz;
    if ((unsigned)fx > 5u || (unsigned)fy > 5u || (unsigned)fz > 5u) raise_drift(g.counters, g.step_id);
  }
  WGT(g, 0, 2);  // particle loads + first adjacency batch here, tile cleared
  P2GParticle q = p2g_finish<TRAD>(raw, b, va, valid, cls, s, d, rpic, dt, w_v, tp);
  FxScale fs{1.0f, 1.0f, 1.0f, 1.0f};
  if (FX) {
    float bm, bp;
    fx_bounds(q, valid, bm, bp);
    fs = fx_scales(bm, bp, red);  // (barrier inside)
    fx_apply(q, fs);
  } else {
    __syncthreads();
  }
  WGT(g, 0, 3);  // corner forces gathered (and the fused traditional stress update done) in every wavefront
  p2g_scatter<STEPS, FX>(tile, esc, &esc_n, q, valid, ox, oy, oz, d, g);
  WGT(g, 0, 4);  // wavefront 0 through its scatter
  __syncthreads();
  WGT(g, 0, 5);  // every wavefront through its scatter
  if (esc_n > 0) {
    for (int e = threadIdx.x; e < esc_n; e += PT) {
      int ec = 0, es = 0;
      // <false>: the fused stress update of this particle already ran (p2g_finish above) and stored its stress; running
      // it again would harden / soften the material twice
      if (cm.map(chunk * CHUNK + esc[e], ec, es)) p2g_escaped<false>(b, va, ec, es, d, rpic, dt, g, tp);
    }
  }
  p2g_flush<JT, false, FX>(tile, ox, oy, oz, d, g, fs);
  WGT(g, 0, 6);  // flush atomics of wavefront 0 acknowledged
  if (JT) {
    // held = one of the last js.n_t traditional particles in the caller's order, with the reference's range check
    int jq = -1;
    if (valid && cls == 1) {
      int o = sa.js.perm[s] - sa.js.off_t;
      if (o >= 0 && o < sa.js.n_t) jq = o;
    }
    V3 xq = raw.x;
    asm volatile("" : "+v"(xq.x), "+v"(xq.y), "+v"(xq.z), "+v"(jq));  // keep pass 2 from sharing live values with pass 1
    if (__syncthreads_or(jq >= 0)) {  // also orders the re-zeroing flush above before the atomics below
      if (threadIdx.x == 0) esc_n = 0;
      P2GParticle q2 = p2g_zero(ox, oy, oz, d);
      bool held = false;
      V3 pv = v3(0, 0, 0);
      if (jq >= 0) {
        Stencil st = make_stencil(xq, d.inv_dx);
        if (splat_ok(d.G, st)) {  // mpm_solver.py:692
          held = true;
          pv = load_v3(sa.js.vel_t + 3 * (size_t)jq);
          q2.s = st; q2.mass = 1.0f; q2.mass_s = 1.0f; q2.a0 = pv;  // contribution = (w, w * v): the scatter's mass / momentum channels
        }
      }
      // This pass stays on the fp64 tile: the grid stage pins EVERY node with a positive mover weight (mpm_utils.py: joint nodes
      // take the joint velocity), and a weight below half a fixed-point unit would round to "not held" -- released sand next to
      // the held pile then fell 3 % too fast (demo-250: x off by 2.3e-4 after 1000 substeps; with this 6e-6, tools/gpu/diag_demo.py).
      __syncthreads();
      p2g_scatter<STEPS, false>(tile, esc, &esc_n, q2, held, ox, oy, oz, d, g);
      __syncthreads();
      for (int e = threadIdx.x; e < esc_n; e += PT) {
        int ec = 0, es = 0;
        if (!cm.map(chunk * CHUNK + esc[e], ec, es)) continue;
        int o = sa.js.perm[es] - sa.js.off_t;
        mover_escaped(ld3(b.all, A_X, es), load_v3(sa.js.vel_t + 3 * (size_t)o), d, g);
      }
      p2g_flush<false, true, false>(tile, ox, oy, oz, d, g, FxScale{1.0f, 1.0f, 1.0f, 1.0f});
    }
  }
  wg_done(sa.pack);
}

// The chunk records come first in the argument list: the record load is the head of every workgroup's dependency chain.
template <int STEPS, bool TRAD, bool JT, bool FX>
__global__ __launch_bounds__(PT) void k_p2g(const ChunkRec *recs, int n_chunks, Bufs b, VAdj va, Dims d, float rpic, float dt,
                                             GridPtrs g, SplatArgs sa, TradParams tp) {
  __shared__ double tile[4 * TILE_PAD];
  __shared__ int esc[CHUNK];
  __shared__ int esc_n;
  __shared__ float red[8];
  p2g_body<STEPS, TRAD, JT, FX>(recs, n_chunks, b, va, d, rpic, dt, g, sa, tp, tile, esc, esc_n, red);
}
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
        const float4 t4 = tile[base + tile_idx(i, j, k)];   // one ds_read_b128 per node instead of three ds_read_b32
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
        const float4 t4 = tile[base + tile_idx(i, j, k)];
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
        const float4 t4 = tile[base + tile_idx(i, j, k)];
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
    st9(b.tr, T_FT, s - d.n_e, (m3_identity() + dt * F) * ld9(b.tr, T_F, s - d.n_e));
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
template <bool FUSED, bool TWO_PASS, bool MFLAG, bool HALO = false>
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
    int sx = valid ? s : 0, se = (valid && cls == 0) ? s : 0;
    x = ld3(b.all, A_X, sx);
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
      G2PResult r = g2p_gather(tile, ox, oy, oz, xg, d);
      if (fit) g2p_write(b, cls, s, x, d3, r, ox, oy, oz, d, dt, g);
    } else {
      {
        G2PResult r;
        r.F = m3_zero();
        g2p_gather_vC(tile, ox, oy, oz, xg, d, r.v, r.C);
        if (fit) g2p_write<true>(b, cls, s, x, d3, r, ox, oy, oz, d, dt, g);
      }
      WGT(g, 1, 4);  // first sweep (v, C) of wavefront 0 stored
      if (__any(fit && cls != 2)) {  // elements and traditional particles also need grad v
        asm volatile("" : "+v"(xg.x), "+v"(xg.y), "+v"(xg.z));  // a fresh stencil: nothing of the first sweep stays live
        M3 rF = g2p_gather_grad(tile, ox, oy, oz, xg, d);
        if (fit && cls != 2) g2p_write_grad(b, cls, s, d3, rF, d, dt);
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
template <bool FUSED, bool TWO_PASS, bool MFLAG>
__global__ __launch_bounds__(PT) void k_g2p(const ChunkRec *recs, int n_chunks, Bufs b, Dims d, float dt, GridPtrs g, GridParams gp,
                                             BCList bcl) {
  __shared__ float4 tile[TILE_PAD];  // node velocity, 16 bytes per node
  g2p_body<FUSED, TWO_PASS, MFLAG>(recs, n_chunks, b, d, dt, g, gp, bcl, tile, (int)blockIdx.x);
}
// multi-GPU: fused halo add (see HaloIn)
template <bool TWO_PASS>
__global__ __launch_bounds__(PT) void k_g2p_halo(const ChunkRec *recs, int n_chunks, Bufs b, Dims d, float dt, GridPtrs g, GridParams gp,
                                                  BCList bcl) {
  __shared__ float4 tile[TILE_PAD];
  g2p_body<true, TWO_PASS, false, true>(recs, n_chunks, b, d, dt, g, gp, bcl, tile, (int)blockIdx.x);
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
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  const bool fit = valid && !escaped;
  V3 nx = x, nv = v3(0, 0, 0);
  M3 nC = m3_zero(), Ft = m3_identity();
  if (__any(fit)) {
    V3 xg = fit ? x : v3((float)(ox + 2) * d.dx, (float)(oy + 2) * d.dx, (float)(oz + 2) * d.dx);
    G2PResult r = g2p_gather(vt, ox, oy, oz, xg, d);
    nv = r.v; nC = r.C;
    Ft = (m3_identity() + dt * r.F) * ld9(b.tr, T_F, tx);   // g2p_v :780-786
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
  const int sp = fit ? s : d.n_e, tp_ = sp - d.n_e;
  P2GRaw raw;
  raw.x = nx; raw.v = nv; raw.C = nC; raw.S = Ft;
  raw.mass = b.all.at(A_MASS, sp); raw.vol = b.nv.at(N_VOL, sp); raw.mu = b.nv.at(N_MU, sp); raw.lam = b.nv.at(N_LAM, sp);
  raw.ys = b.tr.at(T_YS, tp_);
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
template <int STEPS, bool FX>
__global__ __launch_bounds__(PT) void k_g2p2g(const ChunkRec *recs, int n_chunks, Bufs b, VAdj va, Dims d, float rpic, float dt, GridPtrs g,
                                               GridRead rd, SplatArgs sa, TradParams tp, GridParams gp, BCList bcl) {
  __shared__ double tile[4 * TILE_PAD];
  __shared__ int esc[CHUNK];
  __shared__ int esc_n;
  __shared__ float red[8];
  g2p2g_body<STEPS, FX>(recs, n_chunks, b, va, d, rpic, dt, g, rd, sa, tp, gp, bcl, tile, esc, esc_n, red);
}

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

// handshake at link set-up: `n` pattern words through the link's data area, checked on the other side (rccl_link_setup)
__device__ __forceinline__ unsigned link_pattern(int seq, int i) { return (unsigned)i * 2654435761u ^ ((unsigned)seq * 0x9E3779B9u); }
__global__ void k_link_ping(unsigned *data, int n, int *cnt, int *flag, int seq) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] = link_pattern(seq, i);
  link_signal(cnt, gridDim.x, flag, seq);
}
__global__ void k_link_check(const unsigned *data, int n, const int *flag, int seq, int *counters) {
  link_wait(flag, seq, counters + 10, LINK_HANDSHAKE_TICKS);
  int bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad += data[i] != link_pattern(seq, i);
  if (bad) atomicAdd(counters + 11, bad);
}

// the all-reduced drift flag of the sharded loop -> pinned host memory: value first, then its sequence number
__global__ void k_post_flag(const int *value, int *host_sig, int seq) {
  __hip_atomic_store(host_sig + SIG_DFLAG, *value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(host_sig + SIG_DSEQ, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_link_verdict(int *counters) { counters[12] = (counters[10] != 0 || counters[11] != 0) ? 1 : 0; }

__global__ void k_halo_pack(HaloTab tb, GridPtrs g) { halo_pack_wg<false>(tb, g, (int)blockIdx.x); }
// a block can be shared with more than one peer (slabs thinner than two blocks): atomic adds
__global__ void k_halo_add(HaloTab tb, GridPtrs g) {
  int p = tab_peer(tb, blockIdx.x);
  int t = ((int)blockIdx.x - tb.wg_off[p]) * blockDim.x + threadIdx.x;
  int CH = tb.with_mov ? 8 : 4;
  if (tb.sig[p]) link_wait(tb.sig[p], tb.seq, g.counters + 10);
  if (t >= tb.n_blocks[p] * CH * 64) return;
  int l = t & 63, ch = (t >> 6) % CH, i = t / (CH * 64);
  int blk = tb.blocks[p][i];
  float v = tb.buf[p][t];
  if (v == 0.0f) return;
  if (ch < 4) { atomicAdd(g.mv + ((size_t)blk * GCH_MV + ch) * 64 + l, v); g.m_flag[blk] = 1; }
  else atomicAdd(g.mov + ((size_t)blk * GCH_MOV + (ch - 4)) * 64 + l, v);
}
// ghosts: x, v of vertices / traditional particles (6 floats) and the director d3 of elements (3 floats);
// ids are the caller-order particle indices of this rank's solver, inv[] maps them to sorted slots
__global__ void k_ghost_pack(GhostTab tb, const int *inv, Bufs b) {
  int p = tab_peer(tb, blockIdx.x);
  int t = ((int)blockIdx.x - tb.wg_off[p]) * blockDim.x + threadIdx.x;
  int n_p_ids = tb.n_p[p], n_e_ids = tb.n_e[p];
  float *out = tb.buf[p];
  if (t < n_p_ids) {
    int s = inv[tb.ids_p[p][t]];
    V3 x = ld3(b.all, A_X, s), v = ld3(b.all, A_V, s);
    float *o = out + 6 * (size_t)t;
    o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = v.x; o[4] = v.y; o[5] = v.z;
  } else if (t < n_p_ids + n_e_ids) {
    int i = t - n_p_ids, s = inv[tb.ids_e[p][i]];
    float *o = out + 6 * (size_t)n_p_ids + 3 * (size_t)i;
    o[0] = b.el.at(E_D + 2, s); o[1] = b.el.at(E_D + 5, s); o[2] = b.el.at(E_D + 8, s);
  }
}
__global__ void k_ghost_unpack(GhostTab tb, const int *inv, Bufs b) {
  int p = tab_peer(tb, blockIdx.x);
  int t = ((int)blockIdx.x - tb.wg_off[p]) * blockDim.x + threadIdx.x;
  int n_p_ids = tb.n_p[p], n_e_ids = tb.n_e[p];
  const float *in = tb.buf[p];
  if (t < n_p_ids) {
    int s = inv[tb.ids_p[p][t]];
    const float *o = in + 6 * (size_t)t;
    st3(b.all, A_X, s, v3(o[0], o[1], o[2]));
    st3(b.all, A_V, s, v3(o[3], o[4], o[5]));
  } else if (t < n_p_ids + n_e_ids) {
    int i = t - n_p_ids, s = inv[tb.ids_e[p][i]];
    const float *o = in + 6 * (size_t)n_p_ids + 3 * (size_t)i;
    b.el.at(E_D + 2, s) = o[0]; b.el.at(E_D + 5, s) = o[1]; b.el.at(E_D + 8, s) = o[2];
  }
}
// halo_slot[b] = (peer << 24) | index of block b in that peer's shared-block list; *multi = 1 if a block is shared with
// more than one peer (slabs thinner than two blocks: those intervals keep the separate add kernel with its atomics)
__global__ void k_halo_slots(const int *flag, const int *index, int n, int peer, int *slot, int *multi) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n || !flag[b]) return;
  if (slot[b] != -1) *multi = 1;
  else slot[b] = (peer << 24) | index[b];
}
__global__ void k_shared_flags(const unsigned char *a, const unsigned char *b, int n, int *flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (a[i] && b[i]) ? 1 : 0;
}
__global__ void k_flags_to_bytes(const int *flag, int n, unsigned char *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = flag[i] ? 1 : 0;
}

// original-index ELL adjacency from the (float-encoded) faces: pass 0 counts valences, pass 1 fills
__global__ void k_adj_build(const float *faces, int n_e, int n_v, int *cnt, int *adj, int K, int fill) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_e) return;
  for (int c = 0; c < 3; ++c) {
    int v = (int)faces[3 * (size_t)e + c];
    if ((unsigned)v >= (unsigned)n_v) continue;
    int slot = atomicAdd(cnt + v, 1);
    if (fill && slot < K) adj[(size_t)slot * n_v + v] = (e << 2) | c;
  }
}
__global__ void k_max_int(const int *a, int n, int *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax(out, a[i]);
}
__global__ void k_null() {}
__global__ void k_iota(int *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct DistPeer {
  int n_blocks = 0;
  const int *blocks = nullptr;
  float *halo_send = nullptr, *halo_recv = nullptr;
  int n_send_p = 0, n_recv_p = 0, n_send_e = 0, n_recv_e = 0;
  const int *send_p = nullptr, *recv_p = nullptr, *send_e = nullptr, *recv_e = nullptr;
  float *ghost_send = nullptr, *ghost_recv = nullptr;
  // peer link (in-library loop only): this rank's receive arena and the neighbour's, mapped; see rccl_link_setup
  float *link_local = nullptr, *link_remote = nullptr;
  int link_cap = 0;
  int *link_cnt = nullptr;
};

struct Rccl {  // entry points resolved with dlsym: libmpmhip.so itself does not link librccl
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool load(std::string &err) {
    if (h) return true;
    // MPMHIP_RCCL_LIB: another library with the same ten entry points (tests/mock_rccl: shared-memory stand-in that lets
    // 2-3 ranks share the one GPU of a test box, which RCCL itself refuses)
    const char *over = getenv("MPMHIP_RCCL_LIB");
    if (over && *over) {
      h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!h) { err = std::string("dlopen MPMHIP_RCCL_LIB=") + over + ": " + dlerror(); return false; }
      // never silently: the collective library of a production run must be RCCL
      fprintf(stderr, "[mpmhip] WARNING: MPMHIP_RCCL_LIB is set -- the multi-GPU exchange uses %s INSTEAD OF librccl.so (test hook)\n", over);
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("dlopen librccl.so.1: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) err = std::string("dlsym ") + n; return p; };
    *(void **)&GetUniqueId = sym("ncclGetUniqueId"); *(void **)&CommInitRank = sym("ncclCommInitRank");
    *(void **)&CommDestroy = sym("ncclCommDestroy"); *(void **)&GroupStart = sym("ncclGroupStart");
    *(void **)&GroupEnd = sym("ncclGroupEnd"); *(void **)&Send = sym("ncclSend"); *(void **)&Recv = sym("ncclRecv");
    *(void **)&AllGather = sym("ncclAllGather"); *(void **)&GetErrorString = sym("ncclGetErrorString");
    *(void **)&AllReduce = sym("ncclAllReduce");
    return GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllGather && AllReduce &&
           GetErrorString;
  }
};

struct RcclPeer {  // one neighbour rank: static ghost lists + per-re-sort shared blocks, all buffers owned here
  int rank = -1;
  int *send_p = nullptr, *recv_p = nullptr, *send_e = nullptr, *recv_e = nullptr;
  int n_send_p = 0, n_recv_p = 0, n_send_e = 0, n_recv_e = 0;
  float *ghost_send = nullptr, *ghost_recv = nullptr;
  int *blocks = nullptr, *flag = nullptr, *index = nullptr;
  int n_blocks = 0, cap_blocks = 0;
  float *halo_send = nullptr, *halo_recv = nullptr;
  // peer link: arena = [flag parity 0 | flag parity 1 | data parity 0 | data parity 1], flags 64 B apart, data from word 32,
  // link_cap blocks x 8 channels x 64 nodes per parity.  link_local is fine-grained memory of this rank that the
  // neighbour writes; link_remote is the neighbour's arena for this rank (hipIpcOpenMemHandle)
  float *link_local = nullptr, *link_remote = nullptr;
  int link_cap = 0;
  int *link_cnt = nullptr;
  unsigned char *hbuf = nullptr;  // device staging of the two IPC handles (mine at 0, theirs at 64)
};
constexpr int LINK_DATA0 = 32, LINK_FLAG_STRIDE = 16;

struct FastState {
  Rccl rccl;
  std::vector<RcclPeer> rpeers;
  unsigned char *map_all = nullptr;  // [world][nblocks] active-block byte maps
  bool link_want = true, link_decided = false, link_on = false;  // peer-mapped halos: asked for / decided collectively / in use
  // fused halo (peer-mapped halos only): pack workgroups ride in the p2g launch, g2p adds the neighbour's share while it
  // stages its tile (PackArgs, HaloIn) -- no pack / add kernels in the substep.  Decided per collective re-sort (local decision:
  // what goes over the links is the same either way).  MPMHIP_DIST_FUSED_HALO=0: keep the two kernels.
  bool fused_want = true, fused_halo = false;
  int *halo_slot = nullptr, *halo_multi = nullptr;
  unsigned *pack_done = nullptr, pack_target = 0;
  int64_t fused_halo_steps = 0;  // substeps that ran without pack / add kernels (mpmhip_dist_fused_halo_steps)
  unsigned halo_seq = 0;             // substeps exchanged so far (+ handshake rounds): flag value and buffer parity (wraps: the
                                     // kernels compare (int)(flag - seq), long trainings run billions of substeps)
  Dims d{};
  bool dist = false;  // multi-GPU: re-sorts only on request (all ranks re-sort together)
  bool dist_keep_cur = false;  // re-sort inside mpmhip_rccl_steps: the caller's mesh pointers are valid
  bool g2p_two_pass = false;   // k_g2p<., true, .>: see there (default: scenes without traditional particles)
  int splat_first_max = 1 << 30;  // more splat workgroups than this go behind the chunk workgroups (MPMHIP_SPLAT_FIRST_MAX; measured
                                  // neutral early and late -- profiles/r03_experiments.md -- so they stay in front)
  bool p2g_fixed_now = true, p2g_fixed_forced = false, mass_span_pending = false;  // the tile in use (decided per import from the mass span)
  float mass_span = 1.0f;
  bool p2g_fixed = true;       // p2g's chunk tile in packed fixed point (k_p2g<.., FX = true>); MPMHIP_P2G_TILE=f64: the fp64 tile
  bool g2p_mflag = false;      // g2p asks m_flag before it loads a block's accumulators (one more dependent memory level at the head
                               // of every workgroup; the default loads them with the particle positions): MPMHIP_G2P_MFLAG=1
  // adaptive collective re-sorts (mpmhip_rccl_steps with rebin_interval <= 0): the ranks' drift flags are max-reduced
  // every DIST_POLL substeps and read DIST_LAG substeps later, so every rank takes the same decision at the same substep
  int dist_since = 0;
  bool dist_resort = false, dflag_pending = false, rccl_sorted = false;
  int64_t dflag_check_at = 0;
  unsigned dflag_seq = 0;  // sequence number of the last reduction posted to host memory (k_post_flag; wraps)
  std::vector<DistPeer> peers;
  StepArgs dist_args{};
  int blk_bits = 0, key_bits = 0;  // blk_bits: packed key format kf (field widths) as the kernels take it
  int blk_bits_plain = 0;          // bits of a block id (face-bin sort)
  float lead_steps = 12.0f;        // predictive sort: look this many substeps ahead (half the expected re-sort interval)
  float last_dt = 0.0f;
  int poll_mask = 7;               // the drift flag is read back every poll_mask + 1 substeps (host lag <= twice that)
  int true_since_rebin = 0;        // substeps since the last re-sort (steps_since_rebin is overwritten to force one)
  size_t nblocks = 0;
  Bufs buf[2]{};
  int cur = 0;
  int *perm[2] = {nullptr, nullptr}, *inv = nullptr, *face_slot = nullptr;
  // body-face bins (collider gather)
  unsigned *fkeys[2] = {nullptr, nullptr};
  int *forder = nullptr, *fiota = nullptr, *fb_start = nullptr, *fb_cnt = nullptr;
  bool faces_binned = false;
  int rebins_since_face_sort = 0;
  FaceBin *fbins = nullptr;
  int *fidx = nullptr;  // [n_f][3] face vertex ids in bin order
  int cap_fbins = 0, n_fbins = 0;
  F3 *eforce = nullptr;  // [3][n_e] + zero slot
  int *adj_cnt = nullptr, *adj_o = nullptr, *adj_s = nullptr;
  int adj_K = 0, adj_cap = 0;
  VAdj va() const { return VAdj{adj_s, eforce, adj_K, d.n_v, d.n_e}; }
  SortKey *keys[2] = {nullptr, nullptr};
  int *order = nullptr, *iota = nullptr;
  void *sort_tmp = nullptr, *scan_tmp = nullptr;
  size_t sort_tmp_bytes = 0, scan_tmp_bytes = 0;
  int *rs_hist = nullptr;      // [tiles][RS_BINS] digit counts of the radix sort (see k_rs_hist)
  int rs_tiles = 0;
  int *rs_gsum = nullptr;      // [2][groups][RS_BINS] the same counts per group of tiles: the pass in flight / the next one
  int rs_groups = 0;
  unsigned rs_seq = 0;
  bool sort_rocprim = false;   // MPMHIP_SORT=rocprim: the library's sort instead (same permutation)
  GridPtrs g{};
  int *pb_flag = nullptr, *pb_index = nullptr, *ab_flag = nullptr, *ab_index = nullptr;
  int *fc_gsum = nullptr, *fc_tcount = nullptr;  // [2][fc_groups], [2][fc_tiles]: flagged blocks per group / tile (k_flag_count)
  int fc_tiles = 0, fc_groups = 0, n_clear = 0;  // n_clear: ints from pb_flag on that a re-sort starts from zeroed
  int *plist = nullptr, *alist = nullptr, *ranges = nullptr;
  ChunkRec *chunks = nullptr, *chunks_g = nullptr;  // p2g list, g2p list (= p2g list unless there are ghost copies)
  int n_chunks_g = 0;
  bool ghost_g2p = false;  // multi-GPU: ghost copies (selection == 2) gather for themselves
  int cap_P = 0, cap_A = 0, cap_chunks = 0, cap_R = 0;  // cap_P: capacity the block tables are built with (grows on demand)
  int alloc_P = 0;     // allocated entries of plist
  int *rcnt = nullptr; // device: counts of the last re-sort (RC_*)
  int64_t stat_steps = 0;
  int n_P = 0, n_A = 0, n_chunks = 0;
  int *h_pin = nullptr;  // pinned host scratch
  // host staging of the re-sort in pinned memory: device->host copies really are asynchronous, and the chunk table can
  // be uploaded without waiting for the copy (the next re-sort synchronises long before it touches the buffer again)
  template <class T>
  struct Pinned {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    bool reserve(size_t want) {
      if (want <= cap) return true;
      size_t nc = want + want / 2 + 256;
      T *q = nullptr;
      if (hipHostMalloc((void **)&q, nc * sizeof(T), hipHostMallocDefault) != hipSuccess) return false;
      if (p) { memcpy(q, p, n * sizeof(T)); (void)hipHostFree(p); }
      p = q; cap = nc;
      return true;
    }
    bool resize(size_t want) { if (!reserve(want)) return false; n = want; return true; }
    bool push_back(const T &v) { if (n == cap && !reserve(n + 1)) return false; p[n++] = v; return true; }
    void clear() { n = 0; }
    size_t size() const { return n; }
    T *data() { return p; }
    T &operator[](size_t i) { return p[i]; }
    T *begin() { return p; }
    T *end() { return p + n; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
  };
  Pinned<int> h_ranges, h_plist;
  Pinned<ChunkRec> h_chunks, h_chunks_g;
  bool elem_pending = false;  // element finalise of the last substep still to be done (fused into the next stress)
  int steps_since_rebin = 0;
  hipEvent_t ev_flag = nullptr;
  bool flag_pending = false;
  volatile int *h_sig = nullptr;  // pinned, host-mapped, written by the kernels (GridPtrs::host_sig)
  unsigned sig_seq = 0;           // step_id of the last p2g launch issued (wraps)
  unsigned sig_at_rebin = 0;      // sig_seq when the last re-sort finished: ring entries up to it speak about the old order
  bool face_flag_seen = false;    // some ring entry since the last face sort had the face bit set
  int host_lead = 6;              // substeps the host may run ahead of the GPU (MPMHIP_HOST_LEAD)
  bool have_order = false;
  int64_t rebins = 0;
  int rebin_interval = 32;
  bool adaptive_rebin = true;
  // fused grid stage: after a substep the accumulators of the active blocks are still loaded (g2p only read them);
  // they are cleared by the next substep's stress launch (ZeroArgs) or, before a re-sort, by k_zero_blocks
  bool fuse_grid = true, grid_dirty = false, fuse_trad = true;
  int dirty_col = 0, dirty_mov = 0;
  // accumulator double buffer: g.{mv,col,mov,m_flag,col_flag} point at buffer `par`
  // (three for scenes that can run the fused g2p -> p2g launch, k_g2p2g: read / write / clear)
  float *mv2[3] = {nullptr, nullptr, nullptr}, *col2[3] = {nullptr, nullptr, nullptr}, *mov2[3] = {nullptr, nullptr, nullptr};
  int *mflag2[3] = {nullptr, nullptr, nullptr}, *cflag2[3] = {nullptr, nullptr, nullptr};
  int par = 0, nbuf = 2;
  // G2P2G: the g2p of the last substep has not been launched yet -- the next substep's launch does it in front of its own p2g
  // (k_g2p2g), or flush_g2p() does with a plain k_g2p when anything else needs the particles first
  bool g2p2g = true;           // MPMHIP_G2P2G=0: two launches per substep for traditional-only scenes as before
  int g2p2g_max_chunks = 512;  // MPMHIP_G2P2G_MAX
  int split_splat_max_chunks = 1024;  // MPMHIP_SPLIT_SPLAT_MAX
  bool split_splat = true;     // body-face splat: pass 0 in the stress launch, pass 1 in the p2g launch (MPMHIP_SPLIT_SPLAT=0: both in p2g)
  int64_t n_g2p2g = 0;         // fused launches so far (mpmhip_stats)
  bool g2p_pending = false;
  GridParams pend_gp{};
  BCList pend_bcl{};
  float pend_dt = 0.0f;
  int clear_later = -1, cl_col = 0, cl_mov = 0;  // buffer the last fused launch read: cleared by the next one (or by flush_g2p)
  GridParams last_gp{};
  BCList last_bcl{};  // false: ignore the drift flag (tests of the out-of-margin paths)
  std::vector<void *> allocs;
};

int flush_g2p(mpmhip_ctx *c);  // (defined with the step functions: launches the deferred g2p of a G2P2G sequence)

namespace {

template <class T>
int dalloc(mpmhip_ctx *c, T **p, size_t count, bool zero = true) {
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  MPM_HIP_CHECK(c, hipMalloc((void **)p, bytes));
  c->fast->allocs.push_back((void *)*p);
  if (zero) MPM_HIP_CHECK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
  return MPMHIP_OK;
}

int alloc_bufs(mpmhip_ctx *c, Bufs &b) {
  const Dims &d = c->fast->d;
  int rc;
  if ((rc = dalloc(c, &b.all.p, (size_t)A_NC * d.n_p))) return rc;
  if ((rc = dalloc(c, &b.nv.p, (size_t)N_NC * d.n_nv))) return rc;
  if ((rc = dalloc(c, &b.el.p, (size_t)E_NC * d.n_e))) return rc;
  if ((rc = dalloc(c, &b.tr.p, (size_t)T_NC * d.n_t))) return rc;
  if ((rc = dalloc(c, &b.face_orig, (size_t)3 * d.n_e))) return rc;
  if ((rc = dalloc(c, &b.sel, (size_t)d.n_p))) return rc;
  b.all.n = d.n_p; b.nv.n = d.n_nv; b.el.n = d.n_e; b.tr.n = d.n_t;
  return MPMHIP_OK;
}

int ensure_cap(mpmhip_ctx *c, int **p, int *cap, int need, int per) {
  if (need <= *cap) return MPMHIP_OK;
  int ncap = std::max(need + need / 2, 1024);
  int *np_ = nullptr;
  MPM_HIP_CHECK(c, hipMalloc((void **)&np_, (size_t)ncap * per * sizeof(int)));
  c->fast->allocs.push_back(np_);  // old buffer stays alive until destroy (in-flight kernels may use it)
  *p = np_;
  *cap = ncap;
  return MPMHIP_OK;
}

int scan_flags(mpmhip_ctx *c, const int *flag, int *index, int n, int *total) {
  FastState *f = c->fast;
  size_t need = 0;
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(nullptr, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  if (need > f->scan_tmp_bytes) {
    MPM_HIP_CHECK(c, hipMalloc(&f->scan_tmp, need));
    f->allocs.push_back(f->scan_tmp);
    f->scan_tmp_bytes = need;
  }
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(f->scan_tmp, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin, index + (n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 1, flag + (n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  *total = f->h_pin[0] + f->h_pin[1];
  return MPMHIP_OK;
}

// the same scan without the wait: the two addends of the total land in h_pin[slot], h_pin[slot + 1] once the stream gets there
int scan_flags_dev(mpmhip_ctx *c, const int *flag, int *index, int n) {  // exclusive scan, nothing read back
  FastState *f = c->fast;
  size_t need = 0;
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(nullptr, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  if (need > f->scan_tmp_bytes) {
    MPM_HIP_CHECK(c, hipMalloc(&f->scan_tmp, need));
    f->allocs.push_back(f->scan_tmp);
    f->scan_tmp_bytes = need;
  }
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(f->scan_tmp, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  return MPMHIP_OK;
}
int scan_flags_async(mpmhip_ctx *c, const int *flag, int *index, int n, int slot) {
  FastState *f = c->fast;
  size_t need = 0;
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(nullptr, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  if (need > f->scan_tmp_bytes) {
    MPM_HIP_CHECK(c, hipMalloc(&f->scan_tmp, need));
    f->allocs.push_back(f->scan_tmp);
    f->scan_tmp_bytes = need;
  }
  MPM_HIP_CHECK(c, rocprim::exclusive_scan(f->scan_tmp, need, flag, index, 0, (size_t)n, rocprim::plus<int>(), c->stream));
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + slot, index + (n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + slot + 1, flag + (n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  return MPMHIP_OK;
}

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

int do_import(mpmhip_ctx *c) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  if (!f->have_order) {
    if (d.n_p) hipLaunchKernelGGL(k_iota, nblk(d.n_p), TPB, 0, c->stream, f->perm[f->cur], d.n_p);
    f->have_order = true;
  }
  flush_g2p(c);  // (a pending g2p belongs to the state that is about to be replaced; its buffers must be left clean)
  if (d.n_p) {
    hipLaunchKernelGGL(k_import, nblk(d.n_p), TPB, 0, c->stream, c->st, c->md, f->buf[f->cur], f->perm[f->cur], d,
                       f->dist ? 1 : 0);
    // mass span of the scene: decides between the fixed-point and the fp64 chunk tile of p2g at the next re-sort (see rebin)
    MPM_HIP_CHECK(c, hipMemsetD32Async((hipDeviceptr_t)(f->g.counters + CNT_MMIN), 0x7f7fffff, 1, c->stream));
    MPM_HIP_CHECK(c, hipMemsetD32Async((hipDeviceptr_t)(f->g.counters + CNT_MMAX), 0, 1, c->stream));
    hipLaunchKernelGGL(k_mass_span, nblk(d.n_p), TPB, 0, c->stream, (const float *)c->st.particle_mass, (const int *)c->st.particle_selection,
                       d.n_p, f->g.counters);
    f->mass_span_pending = true;
  }
  if (d.n_e && d.n_v) {  // cloth topology -> ELL adjacency (original indices); K = max valence
    hipStream_t s = c->stream;
    MPM_HIP_CHECK(c, hipMemsetAsync(f->adj_cnt, 0, ((size_t)d.n_v + 1) * sizeof(int), s));
    hipLaunchKernelGGL(k_adj_build, nblk(d.n_e), TPB, 0, s, c->st.faces, d.n_e, d.n_v, f->adj_cnt, (int *)nullptr, 0, 0);
    hipLaunchKernelGGL(k_max_int, nblk(d.n_v), TPB, 0, s, f->adj_cnt, d.n_v, f->adj_cnt + d.n_v);
    MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 16, f->adj_cnt + d.n_v, sizeof(int), hipMemcpyDeviceToHost, s));
    MPM_HIP_CHECK(c, hipStreamSynchronize(s));
    int K = std::max(f->h_pin[16], 1);
    if (K > f->adj_cap) {
      int rc2;
      if ((rc2 = dalloc(c, &f->adj_o, (size_t)K * d.n_v, false))) return rc2;
      if ((rc2 = dalloc(c, &f->adj_s, (size_t)K * d.n_v, false))) return rc2;
      f->adj_cap = K;
    }
    f->adj_K = K;
    MPM_HIP_CHECK(c, hipMemsetAsync(f->adj_o, 0xff, (size_t)K * d.n_v * sizeof(int), s));
    MPM_HIP_CHECK(c, hipMemsetAsync(f->adj_cnt, 0, (size_t)d.n_v * sizeof(int), s));
    hipLaunchKernelGGL(k_adj_build, nblk(d.n_e), TPB, 0, s, c->st.faces, d.n_e, d.n_v, f->adj_cnt, f->adj_o, K, 1);
    MPM_HIP_CHECK(c, hipMemsetAsync(f->eforce, 0, ((size_t)3 * d.n_e + 1) * sizeof(F3), s));
  }
  f->elem_pending = false;
  c->caller_dirty = false;
  c->internal_dirty = false;
  f->steps_since_rebin = 1 << 30;  // force a rebin before the next transfer
  return MPMHIP_OK;
}

// what has to be cleared after the last fused substep (the buffer g points at), marking it clean
static ZeroArgs take_zero(FastState *f) {
  ZeroArgs z{f->alist, f->n_A, 0, f->dirty_col, f->dirty_mov, f->g.mv, f->g.col, f->g.mov, f->g.m_flag, f->g.col_flag};
  if (f->grid_dirty && f->n_A) z.n_wg = (f->n_A + PT / 64 - 1) / (PT / 64);
  f->grid_dirty = false;
  return z;
}
static void select_buffer(FastState *f, int par) {
  f->par = par;
  f->g.mv = f->mv2[par]; f->g.col = f->col2[par]; f->g.mov = f->mov2[par];
  f->g.m_flag = f->mflag2[par]; f->g.col_flag = f->cflag2[par];
}
// clear the accumulators now (the active list is about to change)
static void flush_grid(mpmhip_ctx *c) {
  FastState *f = c->fast;
  ZeroArgs z = take_zero(f);
  if (z.n_wg) hipLaunchKernelGGL(k_zero_blocks, (unsigned)z.n_wg, PT, 0, c->stream, z);
}
// v_out of the last (fused) substep for export_grid / stats; the accumulators stay as they are
static void materialize_grid(mpmhip_ctx *c, bool count) {
  FastState *f = c->fast;
  if (!f->grid_dirty || !f->n_A) return;
  GridParams gp = f->last_gp;
  gp.count = count ? 1 : 0;
  hipLaunchKernelGGL(k_grid<false>, xcd_grid((f->n_A + 3) / 4), TPB, 0, c->stream, f->alist, f->n_A, f->d, f->g, gp, f->last_bcl);
}

// Stable sort of n (key, index) pairs by the low `bits` bits of the keys: sorted keys in keys[1], the indices in order[] (= the
// source position of each sorted key); vtmp is scratch (n ints).  The caller writes the unsorted keys into
// keys[sort_input(...)]: the radix passes ping-pong between the two key buffers and must end in keys[1].
constexpr int RS_MAX_N = 1 << 21;  // above: the library (its Onesweep is made for large inputs)
static bool sort_custom(const FastState *f, int n) { return !f->sort_rocprim && n <= RS_MAX_N; }
static int sort_passes(int bits) { return (bits + RS_BITS - 1) / RS_BITS; }
static int sort_input(const FastState *f, int n, int bits) { return sort_custom(f, n) ? 1 - (sort_passes(bits) & 1) : 0; }
// room for the histograms of a sort of n pairs
static int sort_reserve(mpmhip_ctx *c, int n) {
  FastState *f = c->fast;
  const int tiles = (n + RS_TILE - 1) / RS_TILE, groups = (tiles + RS_GROUP - 1) / RS_GROUP;
  if (tiles > f->rs_tiles) {
    int rc;
    if ((rc = dalloc(c, &f->rs_hist, (size_t)RS_BINS * tiles, false))) return rc;
    if ((rc = dalloc(c, &f->rs_gsum, (size_t)2 * RS_BINS * groups))) return rc;  // zeroed here; from then on by k_rs_scatter
    f->rs_tiles = tiles;
    f->rs_groups = groups;
  }
  return MPMHIP_OK;
}
// the group sums the next pass adds into (the two sets alternate with every pass of every sort: a pass clears the set the one
// after it uses)
static int *sort_gsum(FastState *f, int next) { return f->rs_gsum + (size_t)((f->rs_seq + next) & 1) * RS_BINS * f->rs_groups; }
// hist0_done: the caller's key kernel has already filed the first pass's histogram (k_keys: into f->rs_hist / sort_gsum(f, 0))
static int sort_pairs(mpmhip_ctx *c, unsigned *const keys[2], int *vtmp, int *order, int n, int bits, bool hist0_done = false,
                      int *mark = nullptr, int mark_kf = 0) {
  FastState *f = c->fast;
  hipStream_t s = c->stream;
  if (n <= 0) return MPMHIP_OK;
  if (!sort_custom(f, n)) {  // vtmp holds 0, 1, 2, ... (written with the keys)
    size_t need = 0;
    MPM_HIP_CHECK(c, rocprim::radix_sort_pairs(nullptr, need, keys[0], keys[1], vtmp, order, (size_t)n, 0u, (unsigned)bits, s));
    if (need > f->sort_tmp_bytes) {
      MPM_HIP_CHECK(c, hipMalloc(&f->sort_tmp, need));
      f->allocs.push_back(f->sort_tmp);
      f->sort_tmp_bytes = need;
    }
    MPM_HIP_CHECK(c, rocprim::radix_sort_pairs(f->sort_tmp, need, keys[0], keys[1], vtmp, order, (size_t)n, 0u, (unsigned)bits, s));
    return MPMHIP_OK;
  }
  int rc;
  if ((rc = sort_reserve(c, n))) return rc;
  const int tiles = (n + RS_TILE - 1) / RS_TILE, P = sort_passes(bits), s0 = 1 - (P & 1);
  int *vb[2] = {vtmp, order};
  for (int p = 0; p < P; ++p) {
    const unsigned *kin = keys[(s0 + p) & 1];
    unsigned *kout = keys[(s0 + p + 1) & 1];
    const int *vin = vb[(s0 + p) & 1];
    int *vout = vb[(s0 + p + 1) & 1];
    int *gs = sort_gsum(f, 0), *gs_next = sort_gsum(f, 1);
    f->rs_seq += 1;
    if (p > 0 || !hist0_done) hipLaunchKernelGGL(k_rs_hist, (unsigned)tiles, RS_TPB, 0, s, kin, n, p * RS_BITS, f->rs_hist, gs);
    int *mk = p == P - 1 ? mark : nullptr;
    if (p == 0) hipLaunchKernelGGL(k_rs_scatter<true>, (unsigned)tiles, RS_TPB, 0, s, kin, vin, kout, vout, n, p * RS_BITS, tiles, f->rs_hist, gs, gs_next, f->rs_groups, mk, mark_kf);
    else hipLaunchKernelGGL(k_rs_scatter<false>, (unsigned)tiles, RS_TPB, 0, s, kin, vin, kout, vout, n, p * RS_BITS, tiles, f->rs_hist, gs, gs_next, f->rs_groups, mk, mark_kf);
  }
  return MPMHIP_OK;
}

int rebin(mpmhip_ctx *c) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  hipStream_t s = c->stream;
  int cur = f->cur, alt = 1 - cur;
  int rc;
  if (d.n_p == 0) { flush_grid(c); f->n_P = f->n_A = f->n_chunks = f->n_chunks_g = 0; f->steps_since_rebin = 0; return MPMHIP_OK; }
  flush_elements(c);
  // k_keys: the keys (written where the sort wants its input, so that the sorted keys end up in keys[1] and the order in f->order),
  // the sort's first histogram, the zeroing of the block flags / counts, and -- as extra workgroups -- the clearing of the grid
  // accumulators of the old active list (flush_grid)
  const bool fused_hist = sort_custom(f, d.n_p);
  if (fused_hist && (rc = sort_reserve(c, d.n_p))) return rc;
  const ZeroArgs z = take_zero(f);
  const int key_tiles = (d.n_p + RS_TILE - 1) / RS_TILE;
  hipLaunchKernelGGL(k_keys, (unsigned)(key_tiles + z.n_wg), RS_TPB, 0, s, f->buf[cur], d, f->blk_bits, f->lead_steps * f->last_dt,
                     f->ghost_g2p ? 1 : 0, f->keys[sort_input(f, d.n_p, f->key_bits)], f->iota, f->pb_flag, f->n_clear,
                     fused_hist ? f->rs_hist : nullptr, fused_hist ? sort_gsum(f, 0) : nullptr, key_tiles, z);
  // (custom sort: its last pass also flags the particle blocks, k_mark_blocks below)
  if ((rc = sort_pairs(c, f->keys, f->iota, f->order, d.n_p, f->key_bits, fused_hist, f->pb_flag, f->blk_bits))) return rc;
  hipLaunchKernelGGL(k_permute, nblk(d.n_p), TPB, 0, s, f->buf[cur], f->buf[alt], f->order, f->perm[cur], f->perm[alt],
                     f->inv, d);
  f->cur = cur = alt;
  if (d.n_e)
    hipLaunchKernelGGL(k_topology_sorted, nblk(std::max(d.n_e, d.n_v)), TPB, 0, s, f->buf[cur], f->inv, f->face_slot,
                       d.n_v ? f->adj_o : nullptr, f->adj_s, f->perm[cur], f->adj_K, d);
  const SortKey *skeys = f->keys[1];
  int nb = (int)f->nblocks;
  const bool with_faces = !c->colliders.empty() && c->num_mesh_f;
  const int nf = c->num_mesh_f;
  // The face bins survive a particle re-sort (they do not depend on the particle tables; only their compaction onto the
  // active list below does): the ~13 launches of the face sort run when a face has actually left its bin's tile since the
  // last one (counters[5], seen through host memory), at the latest every 16th re-sort, and always in the sharded loops.
  bool face_sort = with_faces;
  if (with_faces && f->faces_binned && f->g.host_sig && !f->dist && !f->face_flag_seen && f->rebins_since_face_sort < 16 &&
      !getenv("MPMHIP_FACE_SORT_ALWAYS"))
    face_sort = false;
  if (face_sort) {  // body faces: sort by block, per-block ranges (independent of the particle tables)
    hipLaunchKernelGGL(k_face_keys, nblk(nf), TPB, 0, s, c->cur_pts, c->cur_vel, c->cur_f, c->mesh_idx, nf, d,
                       f->fkeys[sort_input(f, nf, f->blk_bits_plain + 6)], f->fiota);
    if ((rc = sort_pairs(c, f->fkeys, f->fiota, f->forder, nf, f->blk_bits_plain + 6))) return rc;
    MPM_HIP_CHECK(c, hipMemsetAsync(f->fb_cnt, 0, f->nblocks * sizeof(int), s));
    hipLaunchKernelGGL(k_face_bins, nblk(nf), TPB, 0, s, f->fkeys[1], nf, f->fb_start, f->fb_cnt);
    hipLaunchKernelGGL(k_face_sorted_idx, nblk(nf), TPB, 0, s, c->mesh_idx, f->forder, nf, f->fidx);
    MPM_HIP_CHECK(c, hipMemsetAsync(f->g.counters + CNT_FACE, 0, sizeof(int), s));
    f->rebins_since_face_sort = 0;
    f->face_flag_seen = false;
  } else if (with_faces) {
    f->rebins_since_face_sort += 1;
  }
  // Block tables, chunk records and face bins: every kernel takes its counts from the device array f->rcnt and its array
  // sizes from CAPACITIES, so the whole sequence is enqueued without a host round trip; the host reads the counts once, at
  // the end.  A capacity that turns out too small (first re-sort of a scene, or a scene that spreads out quickly) is
  // grown and the tables are built again.
  if (f->cap_P == 0) f->cap_P = (int)std::min<long long>((long long)nb, std::max<long long>(1024, (long long)d.n_p / 16));
  for (int attempt = 0;; ++attempt) {
    if (attempt > 8) return fail(c, MPMHIP_ERR_HIP, "re-sort: table capacities do not converge");
    const int cap_P = f->cap_P;
    const int cap_A = (int)std::min<long long>((long long)nb, 27LL * cap_P);
    const int cap_ch = cap_P + d.n_p / CHUNK + 8;
    const int cap_fb = with_faces ? std::min(nf, cap_A) : 0;
    int dummy = 0;
    if ((rc = ensure_cap(c, &f->plist, &f->alloc_P, cap_P, 1))) return rc;
    if ((rc = ensure_cap(c, &f->ranges, &f->cap_R, cap_P, 10))) return rc;
    if ((rc = ensure_cap(c, &f->alist, &f->cap_A, cap_A, 1))) return rc;
    if (2 * cap_ch > f->cap_chunks) {
      if ((rc = dalloc(c, &f->chunks, (size_t)2 * cap_ch, false))) return rc;
      f->cap_chunks = 2 * cap_ch;
    }
    if (cap_fb > f->cap_fbins) {
      if ((rc = dalloc(c, &f->fbins, (size_t)cap_fb + 64, false))) return rc;
      f->cap_fbins = cap_fb + 64;
    }
    (void)dummy;
    if (attempt > 0)  // (the first time k_keys has cleared them)
      MPM_HIP_CHECK(c, hipMemsetAsync(f->pb_flag, 0, (size_t)f->n_clear * sizeof(int), s));  // pb_flag, ab_flag, rcnt, fc_gsum
    if (attempt > 0 || !fused_hist) hipLaunchKernelGGL(k_mark_blocks, nblk(d.n_p), TPB, 0, s, skeys, d.n_p, f->blk_bits, f->pb_flag);
    const int ft = f->fc_tiles, fg = f->fc_groups;
    hipLaunchKernelGGL(k_flag_count, (unsigned)ft, 256, 0, s, f->pb_flag, nb, f->fc_tcount, f->fc_gsum);
    hipLaunchKernelGGL(k_compact_tiles, (unsigned)ft, 256, 0, s, f->pb_flag, nb, f->fc_tcount, f->fc_gsum, ft, f->pb_index, f->plist, cap_P,
                       f->rcnt, (int)RC_NP, 1, f->ranges, cap_P * 10);
    hipLaunchKernelGGL(k_ranges, nblk(d.n_p), TPB, 0, s, skeys, d, f->blk_bits, f->pb_index, cap_P, f->ranges);
    hipLaunchKernelGGL(k_dilate, nblk((size_t)cap_P * 27), TPB, 0, s, f->plist, f->rcnt, cap_P, d.NB, f->ab_flag);
    hipLaunchKernelGGL(k_flag_count, (unsigned)ft, 256, 0, s, f->ab_flag, nb, f->fc_tcount + ft, f->fc_gsum + fg);
    hipLaunchKernelGGL(k_compact_tiles, (unsigned)ft, 256, 0, s, f->ab_flag, nb, f->fc_tcount + ft, f->fc_gsum + fg, ft, f->ab_index, f->alist,
                       cap_A, f->rcnt, (int)RC_NA, 2, (int *)nullptr, 0);
    hipLaunchKernelGGL(k_build_chunks, 1, 1024, 0, s, f->plist, f->ranges, cap_P, f->rcnt, f->chunks, f->chunks + cap_ch, cap_ch, f->g.counters);
    if (with_faces)
      hipLaunchKernelGGL(k_fbin_compact, nblk(cap_A), TPB, 0, s, f->alist, f->rcnt, cap_A, f->fb_start, f->fb_cnt, f->fbins, cap_fb);
    MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 32, f->rcnt, RC_N * sizeof(int), hipMemcpyDeviceToHost, s));
    if (f->mass_span_pending)
      MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 44, f->g.counters + CNT_MMIN, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MPM_HIP_CHECK(c, hipStreamSynchronize(s));  // the one wait of a re-sort
    if (f->mass_span_pending) {
      // The fixed-point chunk tile gives every chunk ONE scale, from the sum of its lanes' bounds: a particle whose mass is
      // below ~1e-5 of its chunk mates' loses its contributions to rounding (measured, tools/gpu/mass_ratio.py: cloth beside
      // sand 1e+6 times heavier v 1.1e-2 against 8.6e-5 with the fp64 tile; at 1e+4 both 2e-4).  Scenes whose particle masses
      // span more than 1e+5 therefore run the fp64 tile (MPMHIP_P2G_TILE=fx overrides).
      float lo, hi;
      memcpy(&lo, f->h_pin + 44, 4); memcpy(&hi, f->h_pin + 45, 4);
      f->mass_span = (hi > 0.0f && lo < 3.0e38f) ? hi / lo : 1.0f;
      f->p2g_fixed_now = f->p2g_fixed && (f->p2g_fixed_forced || f->mass_span <= 1.0e5f);
      f->mass_span_pending = false;
    }
    const int *h = f->h_pin + 32;
    if (h[RC_OVER]) {  // grow what was too small and build the tables again (the sorted particles stay as they are)
      f->cap_P = std::max(f->cap_P, std::min(nb, std::max(h[RC_NP], (h[RC_NCH] - d.n_p / CHUNK)) * 2 + 1024));
      if (h[RC_OVER] & ~1) f->cap_P = std::min(nb, f->cap_P * 2);
      continue;
    }
    f->n_P = h[RC_NP];
    f->n_A = h[RC_NA];
    f->n_chunks = h[RC_NCH];
    const bool any_ghost = h[RC_GHOST] != 0;
    f->n_chunks_g = any_ghost ? h[RC_NCHG] : f->n_chunks;
    f->chunks_g = any_ghost ? f->chunks + cap_ch : f->chunks;
    f->n_fbins = with_faces ? h[RC_NFB] : 0;
    if (with_faces) f->faces_binned = true;
    // next time: room for twice what this re-sort needed
    f->cap_P = std::min(nb, std::max(1024, 2 * f->n_P));
    break;
  }
  if (getenv("MPMHIP_VERBOSE")) {  // occupancy of the chunks (particles per chunk) after this re-sort
    std::vector<ChunkRec> hc((size_t)f->n_chunks);
    if (f->n_chunks) MPM_HIP_CHECK(c, hipMemcpy(hc.data(), f->chunks, hc.size() * sizeof(ChunkRec), hipMemcpyDeviceToHost));
    int hist[5] = {0, 0, 0, 0, 0};
    for (auto &r : hc) {
      int tot = r.ne + r.nt + r.nv, n = std::min(CHUNK, tot - r.chunk * CHUNK);
      hist[n <= 32 ? 0 : n <= 64 ? 1 : n <= 128 ? 2 : n < 256 ? 3 : 4]++;
    }
    fprintf(stderr, "[mpmhip] re-sort %ld: %d particle blocks, %d active blocks, %zu chunks (<=32: %d, <=64: %d, <=128: %d, <256: %d, full: %d), lead %.1f\n",
            (long)f->rebins, f->n_P, f->n_A, hc.size(), hist[0], hist[1], hist[2], hist[3], hist[4], f->lead_steps);
  }
  // (k_build_chunks has cleared the drift flag and the parity slots; ring entries up to sig_at_rebin are ignored anyway)
  f->h_pin[24] = 0;
  f->flag_pending = false;
  f->sig_at_rebin = f->sig_seq;  // ring entries of earlier substeps speak about the old order
  f->g.ab_flag = f->ab_flag;
  f->steps_since_rebin = 0;
  f->rebins += 1;
  return MPMHIP_OK;
}

}  // namespace

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
  f->g.stagger = 0; f->g.stagger_groups = 2; f->g.stagger_first = 0;
  if (const char *e = getenv("MPMHIP_P2G_STAGGER")) {  // "units[,groups[,first]]": units of 1024 cycles per group step
    int u = 0, gr = 2, first = 1280;
    sscanf(e, "%d,%d,%d", &u, &gr, &first);
    f->g.stagger = std::max(0, u); f->g.stagger_groups = std::max(1, gr); f->g.stagger_first = std::max(0, first);
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

int fast_pull(mpmhip_ctx *c) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  flush_elements(c);
  if (d.n_p && f->have_order) {
    int m = c->sc.material;
    hipLaunchKernelGGL(k_export, nblk(d.n_p), TPB, 0, c->stream, c->st, c->md, f->buf[f->cur], f->va(),
                       f->perm[f->cur], d, (m == 1 || m == 5) ? 1 : 0);
  }
  c->internal_dirty = false;
  return MPMHIP_OK;
}

// One substep = three phases; the multi-GPU driver interleaves its exchanges between them:
//   A: [re-sort] pre-ops, body/joint splats (side stream), stress, p2g          -> halo exchange of shared blocks
//   B: grid stage, g2p (+ escaped queue)                                         -> ghost x/v/d3 exchange
//   C: element finalise, drift-flag bookkeeping
// P2G_LAUNCH(trad, jt, grid, block, shmem, stream, args...): k_p2g with / without the fused traditional stress update
// and the in-tile joint splat of held traditional particles
#ifndef P2G_STEPS
#define P2G_STEPS 3  // DPP scan steps of the fixed-point instantiations (experiment switch)
#endif
// the hot launches: in prof_fused mode they carry the context's kernel-stamp events (ctx.hpp kev0 / kev1); otherwise a plain launch
template <class K, class... A>
inline void kstamp_launch(mpmhip_ctx *c, K kernel, unsigned grid, unsigned block, A &&...args) {
  if (c->prof_fused) {
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, c->stream, c->kev0, c->kev1, 0, args...);
    c->kev_pending = true;
  } else {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, c->stream, args...);
  }
}
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
// k_g2p on the g2p chunk list: fused = with the grid stage (no grid kernel in the substep), two = the two-sweep gather of cloth scenes
void launch_g2p(mpmhip_ctx *c, bool fused, bool two, float dt, const GridParams &gp, const BCList &bcl) {
  FastState *f = c->fast;
  const Dims &d = f->d;
  const Bufs &b = f->buf[f->cur];
#define G2P_ARGS xcd_grid(f->n_chunks_g), PT, f->chunks_g, f->n_chunks_g, b, d, dt, f->g, gp, bcl
  if (!fused) {
    if (two) kstamp_launch(c, k_g2p<false, true, true>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<false, false, true>, G2P_ARGS);
  } else if (f->g.halo.slot) {
    if (two) kstamp_launch(c, k_g2p_halo<true>, G2P_ARGS);
    else kstamp_launch(c, k_g2p_halo<false>, G2P_ARGS);
  } else if (f->g2p_mflag) {
    if (two) kstamp_launch(c, k_g2p<true, true, true>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<true, false, true>, G2P_ARGS);
  } else {
    if (two) kstamp_launch(c, k_g2p<true, true, false>, G2P_ARGS);
    else kstamp_launch(c, k_g2p<true, false, false>, G2P_ARGS);
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

static void grid_stage_params(mpmhip_ctx *c, const StepArgs &a, GridParams &gp, BCList &bcl);

// G2P2G applies to scenes of traditional particles only, in the production loop of one GPU
static bool g2p2g_ok(const mpmhip_ctx *c) {
  const FastState *f = c->fast;
  const Dims &d = f->d;
  // ... and, as the kernel stands, where one round of workgroups holds the whole scene: hipcc gives the fused kernel 212-227 VGPRs
  // (two wavefronts per SIMD = 512 workgroup slots; either half alone needs 116-124, profiles/r04_experiments.md), which a scene
  // of more chunks pays for with more than it saves (block-512k -8 %, garment-120k-iso -11 %; cube-8k +23 %)
  return f->g2p2g && f->nbuf == 3 && !f->dist && !c->profiling && d.n_e == 0 && d.n_v == 0 && d.n_t > 0 && f->fuse_trad && f->fuse_grid &&
         !f->g.halo.slot && !f->g2p_mflag && !(MPMHIP_DEBUG && f->g.dbg) && f->n_chunks <= f->g2p2g_max_chunks;
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

static int step_phase_a(mpmhip_ctx *c, const StepArgs &a) {
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
  if (d.n_e || (d.n_t && !trad_fused)) {  // (no empty event bracket when the stress update rides in p2g)
    ScopedPhase ph(c, "compute_stress_from_F_trial");
    if (d.n_e) {
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

static int step_phase_b(mpmhip_ctx *c, const StepArgs &a) {
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
    if (f->n_chunks_g) {
      launch_g2p(c, fused, f->g2p_two_pass, dt, gp, bcl);
    }
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

static int step_phase_c(mpmhip_ctx *c, const StepArgs &a) {
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

// ---- multi-GPU entry points --------------------------------------------------------------------------------
int fast_dist_enable(mpmhip_ctx *c) {
  c->fast->dist = true;
  if (!getenv("MPMHIP_DRIFT_LOOKAHEAD")) c->fast->g.lookahead = DRIFT_LOOKAHEAD_DIST;
  return MPMHIP_OK;
}
int fast_dist_set_ghost_mode(mpmhip_ctx *c, int ghosts_gather) {
  c->fast->ghost_g2p = ghosts_gather != 0;
  c->fast->steps_since_rebin = 1 << 30;
  return MPMHIP_OK;
}
int fast_dist_num_blocks(const mpmhip_ctx *c) { return (int)c->fast->nblocks; }
int64_t fast_dist_halo_bytes(const mpmhip_ctx *c) {  // bytes this rank sends per substep in the halo exchange (all peers)
  int CH = c->movers.empty() ? 4 : 8;
  int64_t n = 0;
  for (auto &p : c->fast->peers) n += (int64_t)p.n_blocks * CH * 64 * 4;
  return n;
}

int fast_dist_rebin(mpmhip_ctx *c, unsigned char *active_map) {
  FastState *f = c->fast;
  int rc;
  if (!c->st_bound || !c->md_bound) return fail(c, MPMHIP_ERR_STATE, "dist_rebin: state/model not bound");
  if (c->caller_dirty && (rc = do_import(c))) return rc;
  if (!f->dist_keep_cur) {
    // between substeps the caller's mesh tensors may be gone: bin the body faces from the context's own copy
    c->cur_pts = c->mesh_points;
    c->cur_vel = c->mesh_vel;
    c->cur_f = 0.0f;
  }
  if ((rc = rebin(c))) return rc;
  if (active_map)
    hipLaunchKernelGGL(k_flags_to_bytes, nblk(f->nblocks), TPB, 0, c->stream, f->ab_flag, (int)f->nblocks, active_map);
  return MPMHIP_OK;
}

int fast_dist_set_peers(mpmhip_ctx *c, int n, const mpmhip_dist_peer *peers) {
  FastState *f = c->fast;
  if (n < 0 || n > 64 || (n > 0 && !peers)) return fail(c, MPMHIP_ERR_INVALID, "dist_set_peers: bad peer list");
  f->peers.clear();
  for (int i = 0; i < n; ++i) {
    const mpmhip_dist_peer &p = peers[i];
    DistPeer q;
    q.n_blocks = p.n_blocks; q.blocks = p.blocks; q.halo_send = p.halo_send; q.halo_recv = p.halo_recv;
    q.n_send_p = p.n_send_p; q.n_recv_p = p.n_recv_p; q.n_send_e = p.n_send_e; q.n_recv_e = p.n_recv_e;
    q.send_p = p.send_p; q.recv_p = p.recv_p; q.send_e = p.send_e; q.recv_e = p.recv_e;
    q.ghost_send = p.ghost_send; q.ghost_recv = p.ghost_recv;
    f->peers.push_back(q);
  }
  return MPMHIP_OK;
}

static inline bool peer_linked(const FastState *f, const DistPeer &p) {
  return f->link_on && p.link_remote && p.n_blocks <= p.link_cap;
}
// halo (send = true: pack into halo_send, false: add halo_recv) for all peers, PEER_TAB per launch
static void launch_halo(mpmhip_ctx *c, bool send) {
  FastState *f = c->fast;
  int with_mov = c->movers.empty() ? 0 : 1, CH = with_mov ? 8 : 4;
  for (size_t i0 = 0; i0 < f->peers.size(); i0 += PEER_TAB) {
    HaloTab tb{};
    tb.with_mov = with_mov;
    for (size_t i = i0; i < std::min(f->peers.size(), i0 + PEER_TAB); ++i) {
      const DistPeer &p = f->peers[i];
      if (!p.n_blocks) continue;
      int k = tb.n++;
      tb.blocks[k] = p.blocks; tb.n_blocks[k] = p.n_blocks; tb.buf[k] = send ? p.halo_send : p.halo_recv;
      if (peer_linked(f, p)) {  // store into / read from the receive arena of this pair instead, flag in the same memory
        float *arena = send ? p.link_remote : p.link_local;
        int par = (int)(f->halo_seq & 1u);
        tb.buf[k] = arena + LINK_DATA0 + (size_t)par * p.link_cap * 8 * 64;
        tb.sig[k] = (int *)arena + par * LINK_FLAG_STRIDE;
        tb.cnt[k] = p.link_cnt;
      }
      tb.wg_off[k + 1] = tb.wg_off[k] + (int)nblk((size_t)p.n_blocks * CH * 64);
    }
    tb.seq = (int)f->halo_seq;
    if (!tb.n) continue;
    if (send) hipLaunchKernelGGL(k_halo_pack, (unsigned)tb.wg_off[tb.n], TPB, 0, c->stream, tb, f->g);
    else hipLaunchKernelGGL(k_halo_add, (unsigned)tb.wg_off[tb.n], TPB, 0, c->stream, tb, f->g);
  }
}
static void launch_ghosts(mpmhip_ctx *c, bool send) {
  FastState *f = c->fast;
  for (size_t i0 = 0; i0 < f->peers.size(); i0 += PEER_TAB) {
    GhostTab tb{};
    for (size_t i = i0; i < std::min(f->peers.size(), i0 + PEER_TAB); ++i) {
      const DistPeer &p = f->peers[i];
      int np = send ? p.n_send_p : p.n_recv_p, ne = send ? p.n_send_e : p.n_recv_e;
      if (np + ne == 0) continue;
      int k = tb.n++;
      tb.ids_p[k] = send ? p.send_p : p.recv_p; tb.ids_e[k] = send ? p.send_e : p.recv_e;
      tb.n_p[k] = np; tb.n_e[k] = ne; tb.buf[k] = send ? p.ghost_send : p.ghost_recv;
      tb.wg_off[k + 1] = tb.wg_off[k] + (int)nblk((size_t)(np + ne));
    }
    if (!tb.n) continue;
    if (send) hipLaunchKernelGGL(k_ghost_pack, (unsigned)tb.wg_off[tb.n], TPB, 0, c->stream, tb, f->inv, f->buf[f->cur]);
    else hipLaunchKernelGGL(k_ghost_unpack, (unsigned)tb.wg_off[tb.n], TPB, 0, c->stream, tb, f->inv, f->buf[f->cur]);
  }
}

int fast_dist_phase(mpmhip_ctx *c, int phase, const StepArgs &a) {
  FastState *f = c->fast;
  int rc;
  if (phase == 0) {
    f->dist_args = a;
    if ((rc = step_phase_a(c, a))) return rc;
    const bool fused_halo_now = f->fused_halo && !c->profiling && !c->prof_fused;   // (the pack rode in the p2g launch)
    if (!fused_halo_now) {
      ScopedPhase ph(c, "halo_pack");  // (profiling only) with peer links: the stores into the neighbour's memory + its flag
      launch_halo(c, true);
    }
  } else if (phase == 1) {
    const bool fused_halo_now = f->fused_halo && !c->profiling && !c->prof_fused;
    if (!fused_halo_now) {
      ScopedPhase ph(c, "halo_add");   // (profiling only) with peer links: includes the wait for the neighbour's flag
      launch_halo(c, false);
    } else {  // g2p adds the neighbours' shares itself: this substep's receive buffers and flags
      f->fused_halo_steps += 1;
      HaloIn &h = f->g.halo;
      h = HaloIn{};
      h.slot = f->halo_slot;
      h.n_peers = (int)f->peers.size();
      h.seq = (int)f->halo_seq;
      h.ch = c->movers.empty() ? 4 : 8;
      const int par = (int)(f->halo_seq & 1u);
      for (size_t i = 0; i < f->peers.size(); ++i) {
        const DistPeer &q = f->peers[i];
        if (!q.n_blocks || !q.link_local) continue;
        h.buf[i] = q.link_local + LINK_DATA0 + (size_t)par * q.link_cap * 8 * 64;
        h.sig[i] = (const int *)q.link_local + par * LINK_FLAG_STRIDE;
      }
    }
    rc = step_phase_b(c, f->dist_args);
    f->g.halo.slot = nullptr;
    if (rc) return rc;
    if (!f->ghost_g2p) launch_ghosts(c, true);
  } else {
    if (!f->ghost_g2p) launch_ghosts(c, false);
    if ((rc = step_phase_c(c, f->dist_args))) return rc;
  }
  MPM_HIP_CHECK(c, hipGetLastError());
  return MPMHIP_OK;
}
// re-synchronisation of the ghost copies around an exchange (ghost mode 1: at every collective re-sort)
int fast_dist_ghosts(mpmhip_ctx *c, int send) {
  if (!c->fast->have_order) return MPMHIP_OK;  // nothing sorted yet: the copies are still the caller's exact values
  if (!send) flush_elements(c);                // finished elements first: the unpack overwrites their d3
  launch_ghosts(c, send != 0);
  MPM_HIP_CHECK(c, hipGetLastError());
  return MPMHIP_OK;
}

// ---- RCCL transport inside the library ----------------------------------------------------------------------
#define MPM_NCCL_CHECK(c, r, expr)                                                                                \
  do {                                                                                                            \
    ncclResult_t e_ = (expr);                                                                                     \
    if (e_ != ncclSuccess) return fail(c, MPMHIP_ERR_HIP, std::string(#expr) + ": " + (r).GetErrorString(e_));   \
  } while (0)

int fast_rccl_unique_id(char id[128], std::string &err) {
  Rccl r;
  if (!r.load(err)) return MPMHIP_ERR_HIP;
  ncclUniqueId uid;
  ncclResult_t e = r.GetUniqueId(&uid);
  if (e != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r.GetErrorString(e); return MPMHIP_ERR_HIP; }
  static_assert(sizeof(uid) == 128, "ncclUniqueId size");
  memcpy(id, &uid, 128);
  return MPMHIP_OK;
}

int fast_rccl_init(mpmhip_ctx *c, int rank, int world, const char id[128]) {
  FastState *f = c->fast;
  if (world < 1 || rank < 0 || rank >= world) return fail(c, MPMHIP_ERR_INVALID, "rccl_init: bad rank/world");
  if (!f->rccl.load(c->err)) return MPMHIP_ERR_HIP;
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  MPM_NCCL_CHECK(c, f->rccl, f->rccl.CommInitRank(&f->rccl.comm, world, uid, rank));
  f->rccl.rank = rank;
  f->rccl.world = world;
  f->dist = true;
  if (!getenv("MPMHIP_DRIFT_LOOKAHEAD")) f->g.lookahead = DRIFT_LOOKAHEAD_DIST;
  const char *hm = getenv("MPMHIP_DIST_HALO");  // "rccl": keep the halos on ncclSend/ncclRecv; default: peer-mapped buffers
  f->link_want = !(hm && !strcmp(hm, "rccl"));
  int rc;
  if ((rc = dalloc(c, &f->map_all, (size_t)world * f->nblocks))) return rc;
  return MPMHIP_OK;
}

int fast_rccl_set_ghosts(mpmhip_ctx *c, int n, const int32_t *ranks, const int32_t *nsp, const int32_t *const *sp,
                         const int32_t *nrp, const int32_t *const *rp, const int32_t *nse, const int32_t *const *se,
                         const int32_t *nre, const int32_t *const *re) {
  FastState *f = c->fast;
  if (!f->rccl.comm) return fail(c, MPMHIP_ERR_STATE, "rccl_set_ghosts: call mpmhip_rccl_init first");
  // one peer slot per other rank (shared blocks may exist without ghosts, e.g. traditional particles only)
  f->rpeers.clear();
  for (int q = 0; q < f->rccl.world; ++q) {
    if (q == f->rccl.rank) continue;
    RcclPeer p;
    p.rank = q;
    int rc;
    if ((rc = dalloc(c, &p.flag, f->nblocks))) return rc;
    if ((rc = dalloc(c, &p.index, f->nblocks))) return rc;
    f->rpeers.push_back(p);
  }
  auto up = [&](int **dst, const int32_t *src, int cnt) -> int {
    int rc = dalloc(c, dst, (size_t)std::max(cnt, 1), false);
    if (rc) return rc;
    if (cnt) MPM_HIP_CHECK(c, hipMemcpy(*dst, src, (size_t)cnt * sizeof(int), hipMemcpyHostToDevice));
    return MPMHIP_OK;
  };
  for (int i = 0; i < n; ++i) {
    RcclPeer *p = nullptr;
    for (auto &q : f->rpeers) if (q.rank == ranks[i]) p = &q;
    if (!p) return fail(c, MPMHIP_ERR_INVALID, "rccl_set_ghosts: bad peer rank");
    int rc;
    p->n_send_p = nsp[i]; p->n_recv_p = nrp[i]; p->n_send_e = nse[i]; p->n_recv_e = nre[i];
    if ((rc = up(&p->send_p, sp[i], nsp[i])) || (rc = up(&p->recv_p, rp[i], nrp[i])) || (rc = up(&p->send_e, se[i], nse[i])) ||
        (rc = up(&p->recv_e, re[i], nre[i])))
      return rc;
    if ((rc = dalloc(c, &p->ghost_send, (size_t)6 * nsp[i] + 3 * nse[i] + 1))) return rc;
    if ((rc = dalloc(c, &p->ghost_recv, (size_t)6 * nrp[i] + 3 * nre[i] + 1))) return rc;
  }
  return MPMHIP_OK;
}

// Peer-mapped halo buffers.  At the first collective re-sort every pair of ranks that shares grid blocks allocates a
// fine-grained receive arena each, swaps the HIP IPC handles (64 bytes through ncclSend/ncclRecv), maps the other side's
// arena and pushes four rounds of a test pattern through both buffer parities with the same signal / wait primitives the
// substep uses.  The outcome is max-reduced over all ranks: only if every link of every rank works do the halos go
// through the links (k_halo_pack stores into the neighbour's memory and raises its flag, k_halo_add waits for the
// flag: no RCCL kernel in the substep); otherwise every rank stays on ncclSend/ncclRecv.  Pairs that start sharing
// blocks only later, or share more than link_cap of them, use send/recv for that interval (both sides see the same count).
static int rccl_link_setup(mpmhip_ctx *c) {
  FastState *f = c->fast;
  Rccl &r = f->rccl;
  hipStream_t s = c->stream;
  int rc, bad = 0;
  f->link_decided = true;
  std::vector<hipIpcMemHandle_t> mine(f->rpeers.size()), theirs(f->rpeers.size());
  for (size_t i = 0; i < f->rpeers.size(); ++i) {
    RcclPeer &p = f->rpeers[i];
    memset(&mine[i], 0, sizeof(hipIpcMemHandle_t));
    if (!p.n_blocks) continue;
    if ((rc = dalloc(c, &p.link_cnt, 1))) return rc;
    if ((rc = dalloc(c, &p.hbuf, 256))) return rc;  // [0, 96): my IPC handle + PCI bus id, [128, 224): the peer's
    p.link_cap = std::max(4 * p.n_blocks, 1024);
    size_t bytes = ((size_t)LINK_DATA0 + 2 * (size_t)p.link_cap * 8 * 64) * sizeof(float);
    if (hipExtMallocWithFlags((void **)&p.link_local, bytes, hipDeviceMallocFinegrained) != hipSuccess) { p.link_local = nullptr; bad = 1; continue; }
    if (hipMemsetAsync(p.link_local, 0, bytes, s) != hipSuccess || hipIpcGetMemHandle(&mine[i], p.link_local) != hipSuccess) {
      memset(&mine[i], 0, sizeof(hipIpcMemHandle_t));
      bad = 1;
    }
  }
  (void)hipGetLastError();
  // ... and with the handle this rank's PCI bus id: the receiver asks hipDeviceCanAccessPeer before it maps the arena
  char my_bus[32] = {0};
  int my_dev = 0;
  (void)hipGetDevice(&my_dev);
  if (hipDeviceGetPCIBusId(my_bus, (int)sizeof my_bus, my_dev) != hipSuccess) my_bus[0] = 0;
  (void)hipGetLastError();
  std::vector<std::array<char, 96>> msg_out(f->rpeers.size()), msg_in(f->rpeers.size());
  for (size_t i = 0; i < f->rpeers.size(); ++i) {
    memcpy(msg_out[i].data(), &mine[i], 64);
    memcpy(msg_out[i].data() + 64, my_bus, 32);
    if (f->rpeers[i].n_blocks) MPM_HIP_CHECK(c, hipMemcpyAsync(f->rpeers[i].hbuf, msg_out[i].data(), 96, hipMemcpyHostToDevice, s));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  MPM_NCCL_CHECK(c, r, r.GroupStart());
  for (auto &p : f->rpeers) {
    if (!p.n_blocks) continue;
    MPM_NCCL_CHECK(c, r, r.Send(p.hbuf, 96, ncclUint8, p.rank, r.comm, s));
    MPM_NCCL_CHECK(c, r, r.Recv(p.hbuf + 128, 96, ncclUint8, p.rank, r.comm, s));
  }
  MPM_NCCL_CHECK(c, r, r.GroupEnd());
  for (size_t i = 0; i < f->rpeers.size(); ++i)
    if (f->rpeers[i].n_blocks) MPM_HIP_CHECK(c, hipMemcpyAsync(msg_in[i].data(), f->rpeers[i].hbuf + 128, 96, hipMemcpyDeviceToHost, s));
  MPM_HIP_CHECK(c, hipStreamSynchronize(s));
  const bool verbose = getenv("MPMHIP_VERBOSE") != nullptr;
  for (size_t i = 0; i < f->rpeers.size(); ++i) {
    RcclPeer &p = f->rpeers[i];
    if (!p.n_blocks) continue;
    static const hipIpcMemHandle_t none{};
    memcpy(&theirs[i], msg_in[i].data(), 64);
    char peer_bus[33] = {0};
    memcpy(peer_bus, msg_in[i].data() + 64, 32);
    // Can this GPU reach the peer's memory at all?  Asked BEFORE the arena is mapped (round 4): a pair without peer access (another
    // PCIe root without xGMI, an IOMMU setting) is refused here with a reason, instead of failing inside hipIpcOpenMemHandle or --
    // worse -- passing it and faulting in the first substep.  The same GPU (ranks sharing a device in tests) needs no peer access; a
    // peer device this process cannot see (masked by HIP_VISIBLE_DEVICES) cannot be asked, and the mapping is attempted.
    int peer_dev = -1, can = 1;
    const char *why = "same device";
    if (peer_bus[0] && hipDeviceGetByPCIBusId(&peer_dev, peer_bus) == hipSuccess) {
      if (peer_dev != my_dev) {
        if (hipDeviceCanAccessPeer(&can, my_dev, peer_dev) != hipSuccess) can = 0;
        why = can ? "hipDeviceCanAccessPeer: yes" : "hipDeviceCanAccessPeer: NO";
      }
    } else {
      why = "peer device not visible to this process: not asked";
    }
    (void)hipGetLastError();
    bool mapped = false;
    if (can && memcmp(&theirs[i], &none, 64) &&
        hipIpcOpenMemHandle((void **)&p.link_remote, theirs[i], hipIpcMemLazyEnablePeerAccess) == hipSuccess)
      mapped = true;
    if (!mapped) {
      p.link_remote = nullptr;
      bad = 1;
    }
    if (verbose || !mapped)
      fprintf(stderr, "[mpmhip] rank %d (%s) <- rank %d (%s): %s; halo arena %s\n", r.rank, my_bus[0] ? my_bus : "?", p.rank,
              peer_bus[0] ? peer_bus : "?", why, mapped ? "mapped (HIP IPC)" : "NOT mapped: every rank falls back to ncclSend / ncclRecv");
  }
  (void)hipGetLastError();
  const char *fault = getenv("MPMHIP_LINK_FAULT");  // tests: this rank pretends its links failed
  if (fault && *fault && atoi(fault) == r.rank) bad = 1;
  // agree before the handshake: a rank without its links would leave its neighbours waiting for pings
  int *vote = f->g.counters + 12;
  MPM_HIP_CHECK(c, hipMemcpyAsync(vote, &bad, sizeof(int), hipMemcpyHostToDevice, s));
  MPM_NCCL_CHECK(c, r, r.AllReduce(vote, vote + 1, 1, ncclInt32, ncclMax, r.comm, s));
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 28, vote + 1, sizeof(int), hipMemcpyDeviceToHost, s));
  MPM_HIP_CHECK(c, hipStreamSynchronize(s));
  if (f->h_pin[28] == 0) {
    MPM_HIP_CHECK(c, hipMemsetAsync(f->g.counters + 10, 0, 2 * sizeof(int), s));
    for (int round = 0; round < 4; ++round) {
      int seq = (int)++f->halo_seq, par = seq & 1;
      for (auto &p : f->rpeers) {
        if (!p.link_remote) continue;
        int n = (int)std::min<size_t>((size_t)p.link_cap * 8 * 64, (size_t)1 << 16);
        hipLaunchKernelGGL(k_link_ping, 16, TPB, 0, s, (unsigned *)p.link_remote + LINK_DATA0 + (size_t)par * p.link_cap * 8 * 64, n,
                           p.link_cnt, (int *)p.link_remote + par * LINK_FLAG_STRIDE, seq);
      }
      for (auto &p : f->rpeers) {
        if (!p.link_remote) continue;
        int n = (int)std::min<size_t>((size_t)p.link_cap * 8 * 64, (size_t)1 << 16);
        hipLaunchKernelGGL(k_link_check, 16, TPB, 0, s, (const unsigned *)p.link_local + LINK_DATA0 + (size_t)par * p.link_cap * 8 * 64, n,
                           (const int *)p.link_local + par * LINK_FLAG_STRIDE, seq, f->g.counters);
      }
    }
    hipLaunchKernelGGL(k_link_verdict, 1, 1, 0, s, f->g.counters);
    MPM_NCCL_CHECK(c, r, r.AllReduce(vote, vote + 1, 1, ncclInt32, ncclMax, r.comm, s));
    MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 28, vote + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    MPM_HIP_CHECK(c, hipStreamSynchronize(s));
    MPM_HIP_CHECK(c, hipMemsetAsync(f->g.counters + 10, 0, 2 * sizeof(int), s));
  }
  f->link_on = f->h_pin[28] == 0;
  if (getenv("MPMHIP_VERBOSE"))
    fprintf(stderr, "[mpmhip] rank %d: halo transport %s\n", r.rank, f->link_on ? "peer-mapped buffers" : "ncclSend/ncclRecv");
  return MPMHIP_OK;
}

static int rccl_rebin(mpmhip_ctx *c) {
  FastState *f = c->fast;
  Rccl &r = f->rccl;
  hipStream_t s = c->stream;
  int rc, nb = (int)f->nblocks;
  unsigned char *mine = f->map_all + (size_t)r.rank * f->nblocks;
  if ((rc = fast_dist_rebin(c, mine))) return rc;
  MPM_NCCL_CHECK(c, r, r.AllGather(mine, f->map_all, f->nblocks, ncclUint8, r.comm, s));
  int CH = c->movers.empty() ? 4 : 8;
  f->peers.clear();
  for (auto &p : f->rpeers) {
    hipLaunchKernelGGL(k_shared_flags, nblk(nb), TPB, 0, s, mine, f->map_all + (size_t)p.rank * f->nblocks, nb, p.flag);
    if ((rc = scan_flags(c, p.flag, p.index, nb, &p.n_blocks))) return rc;
    if (p.n_blocks > p.cap_blocks) {
      int cap = std::max(p.n_blocks + p.n_blocks / 2, 256);
      if ((rc = dalloc(c, &p.blocks, (size_t)cap, false))) return rc;
      if ((rc = dalloc(c, &p.halo_send, (size_t)cap * 8 * 64, false))) return rc;
      if ((rc = dalloc(c, &p.halo_recv, (size_t)cap * 8 * 64, false))) return rc;
      p.cap_blocks = cap;
    }
    if (p.n_blocks) hipLaunchKernelGGL(k_compact, nblk(nb), TPB, 0, s, p.flag, p.index, nb, p.blocks, p.cap_blocks, (int *)nullptr, 0, 0, (int *)nullptr, 0);
    DistPeer q;
    q.n_blocks = p.n_blocks; q.blocks = p.blocks; q.halo_send = p.halo_send; q.halo_recv = p.halo_recv;
    q.n_send_p = p.n_send_p; q.n_recv_p = p.n_recv_p; q.n_send_e = p.n_send_e; q.n_recv_e = p.n_recv_e;
    q.send_p = p.send_p; q.recv_p = p.recv_p; q.send_e = p.send_e; q.recv_e = p.recv_e;
    q.ghost_send = p.ghost_send; q.ghost_recv = p.ghost_recv;
    f->peers.push_back(q);
  }
  (void)CH;
  if (f->link_want && !f->link_decided && (rc = rccl_link_setup(c))) return rc;
  for (size_t i = 0; i < f->peers.size(); ++i) {
    const RcclPeer &p = f->rpeers[i];
    DistPeer &q = f->peers[i];
    q.link_local = p.link_local; q.link_remote = p.link_remote; q.link_cap = p.link_cap; q.link_cnt = p.link_cnt;
  }
  // fused halo for this interval?
  f->fused_halo = false;
  if (f->fused_want && f->link_on && f->g2p_mflag == false && f->fuse_grid && f->peers.size() <= (size_t)PEER_TAB) {
    bool all = true, any = false;
    for (auto &q : f->peers)
      if (q.n_blocks) { any = true; all = all && peer_linked(f, q); }
    if (all && any) {
      if (!f->halo_slot) {
        if ((rc = dalloc(c, &f->halo_slot, f->nblocks + 1, false))) return rc;
        f->halo_multi = f->halo_slot + f->nblocks;
      }
      MPM_HIP_CHECK(c, hipMemsetAsync(f->halo_slot, 0xff, f->nblocks * sizeof(int), s));
      MPM_HIP_CHECK(c, hipMemsetAsync(f->halo_multi, 0, sizeof(int), s));
      for (size_t i = 0; i < f->peers.size(); ++i)
        if (f->peers[i].n_blocks)
          hipLaunchKernelGGL(k_halo_slots, nblk(nb), TPB, 0, s, f->rpeers[i].flag, f->rpeers[i].index, nb, (int)i, f->halo_slot, f->halo_multi);
      MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 30, f->halo_multi, sizeof(int), hipMemcpyDeviceToHost, s));
      MPM_HIP_CHECK(c, hipStreamSynchronize(s));
      f->fused_halo = f->h_pin[30] == 0;
    }
  }
  return MPMHIP_OK;
}

// f->peers[i] corresponds to f->rpeers[i]
static int rccl_exchange(mpmhip_ctx *c, bool halo) {
  FastState *f = c->fast;
  Rccl &r = f->rccl;
  int CH = c->movers.empty() ? 4 : 8;
  bool open = false;
  for (size_t i = 0; i < f->peers.size(); ++i) {
    const DistPeer &p = f->peers[i];
    int peer = f->rpeers[i].rank;
    if (halo && peer_linked(f, p)) continue;  // went through the pair's link (k_halo_pack / k_halo_add)
    size_t ns = halo ? (size_t)p.n_blocks * CH * 64 : (size_t)6 * p.n_send_p + 3 * p.n_send_e;
    size_t nr = halo ? (size_t)p.n_blocks * CH * 64 : (size_t)6 * p.n_recv_p + 3 * p.n_recv_e;
    if ((ns || nr) && !open) { MPM_NCCL_CHECK(c, r, r.GroupStart()); open = true; }
    if (ns) MPM_NCCL_CHECK(c, r, r.Send(halo ? p.halo_send : p.ghost_send, ns, ncclFloat, peer, r.comm, c->stream));
    if (nr) MPM_NCCL_CHECK(c, r, r.Recv(halo ? p.halo_recv : p.ghost_recv, nr, ncclFloat, peer, r.comm, c->stream));
  }
  if (open) MPM_NCCL_CHECK(c, r, r.GroupEnd());
  return MPMHIP_OK;
}

int fast_rccl_steps(mpmhip_ctx *c, float dt, int n, int64_t step_index, int rebin_interval, const float *mesh_x,
                    const float *mesh_v, const float *jt, int n_jt, const float *jv, const float *jf) {
  FastState *f = c->fast;
  if (!f->rccl.comm) return fail(c, MPMHIP_ERR_STATE, "rccl_steps: call mpmhip_rccl_init first");
  // rebin_interval > 0: every rank re-sorts at substeps that are multiples of it.  <= 0: when any rank's early-warning
  // drift flag is up (the single-GPU policy made collective), at the latest every 256 (or -rebin_interval) substeps.
  const bool adaptive = rebin_interval <= 0;
  const int cap = rebin_interval < 0 ? -rebin_interval : (rebin_interval == 0 ? 256 : rebin_interval);
  constexpr int DIST_POLL = 16, DIST_LAG = 4;
  int rc;
  for (int k = 0; k < n; ++k) {
    int64_t idx = step_index + k;
    StepArgs a{dt, mesh_x, mesh_v, (float)((double)dt * (double)idx), true, jt, jt ? n_jt : 0, jv, jf};
    c->cur_pts = a.mesh_x ? a.mesh_x : c->mesh_points;
    c->cur_vel = a.mesh_v ? a.mesh_v : c->mesh_vel;
    c->cur_f = (a.mesh_x && a.mesh_v) ? a.mesh_f : 0.0f;
    if (adaptive && f->dflag_pending && idx >= f->dflag_check_at) {
      if (f->g.host_sig) {  // posted by k_post_flag: wait for THIS reduction's sequence number, then read its value
        for (long spins = 0; (unsigned)f->h_sig[SIG_DSEQ] != f->dflag_seq; ++spins) {
          if ((spins & 0x3ff) == 0x3ff) {
            hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess && (unsigned)f->h_sig[SIG_DSEQ] != f->dflag_seq)
              return fail(c, MPMHIP_ERR_HIP, "rccl_steps: the reduced drift flag never reached host memory");
            if (e != hipSuccess && e != hipErrorNotReady) MPM_HIP_CHECK(c, e);
          }
          std::this_thread::yield();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (f->h_sig[SIG_DFLAG]) f->dist_resort = true;
      } else {
        MPM_HIP_CHECK(c, hipEventSynchronize(f->ev_flag));
        if (f->h_pin[26]) f->dist_resort = true;
      }
      f->dflag_pending = false;
    }
    bool due = adaptive ? (f->dist_resort || f->dist_since >= cap || !f->rccl_sorted) : (idx % cap == 0);
    if (due || c->caller_dirty) {
      if (f->ghost_g2p && f->have_order && !c->caller_dirty && !f->peers.empty()) {  // owners -> copies, then re-sort
        if ((rc = fast_dist_ghosts(c, 1))) return rc;
        if ((rc = rccl_exchange(c, false))) return rc;
        if ((rc = fast_dist_ghosts(c, 0))) return rc;
      }
      if (adaptive && f->true_since_rebin > 0)  // predictive sort: aim at the middle of the next interval
        f->lead_steps = std::min(std::max(0.5f * (float)f->true_since_rebin, 4.0f), 48.0f);
      f->dist_keep_cur = true;
      rc = rccl_rebin(c);
      f->dist_keep_cur = false;
      if (rc) return rc;
      f->true_since_rebin = 0;
      f->dist_since = 0;
      f->dist_resort = false;
      f->rccl_sorted = true;
      if (f->dflag_pending) {  // a reduction issued before this re-sort speaks about the old order: drop it (every rank does)
        if (!f->g.host_sig) MPM_HIP_CHECK(c, hipEventSynchronize(f->ev_flag));
        f->dflag_pending = false;  // (host memory: the next poll waits for a newer sequence number)
      }
    }
    f->halo_seq += 1;
    if ((rc = fast_dist_phase(c, 0, a))) return rc;
    {
      ScopedPhase ph(c, "halo_exchange");  // (profiling only: the ncclSend/ncclRecv group between the pack and the add kernel)
      if ((rc = rccl_exchange(c, true))) return rc;
    }
    if ((rc = fast_dist_phase(c, 1, a))) return rc;
    if (!f->ghost_g2p && (rc = rccl_exchange(c, false))) return rc;
    if ((rc = fast_dist_phase(c, 2, a))) return rc;
    c->time = c->time + c->time_inc(dt);
    c->substeps += 1;
    f->dist_since += 1;
    if (adaptive && !f->dflag_pending && f->dist_since % DIST_POLL == 0) {
      MPM_NCCL_CHECK(c, f->rccl, f->rccl.AllReduce(f->g.counters + 6, f->g.counters + 7, 1, ncclInt32, ncclMax, f->rccl.comm, c->stream));
      if (f->g.host_sig) {  // no copy + event on the stream (each costs an idle queue, see fast_step): one thread posts the result
        hipLaunchKernelGGL(k_post_flag, 1, 1, 0, c->stream, f->g.counters + 7, f->g.host_sig, (int)++f->dflag_seq);
      } else {
        MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 26, f->g.counters + 7, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MPM_HIP_CHECK(c, hipEventRecord(f->ev_flag, c->stream));
      }
      f->dflag_pending = true;
      f->dflag_check_at = idx + 1 + DIST_LAG;
    }
  }
  if (f->link_on) {  // a wait that ran into its wall-clock bound computed with an incomplete halo: fail the call
    MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 29, f->g.counters + 10, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (f->h_pin[29]) return fail(c, MPMHIP_ERR_HIP, "rccl_steps: a peer-mapped halo never arrived (flag wait timed out)");
  }
  return MPMHIP_OK;
}
int fast_dist_halo_transport(const mpmhip_ctx *c) { return c->fast->link_on ? 1 : 0; }
int64_t fast_dist_fused_halo_steps(const mpmhip_ctx *c) { return c->fast->fused_halo_steps; }

// the drift flag of this rank (set by the kernels when a particle is about to leave its tile margin); synchronous
int fast_dist_drift_flag(mpmhip_ctx *c, int32_t *out) {
  FastState *f = c->fast;
  MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 27, f->g.counters + 6, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MPM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  *out = f->h_pin[27];
  return MPMHIP_OK;
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
int fast_debug_sort(mpmhip_ctx *c, const uint32_t *keys_in, int n, int bits, uint32_t *keys_out, int32_t *order_out) {
  FastState *f = c->fast;
  if (n == 0) return MPMHIP_OK;
  unsigned *kb[2] = {nullptr, nullptr};
  int *vtmp = nullptr;
  auto done = [&](int rc) {
    for (void *p : {(void *)kb[0], (void *)kb[1], (void *)vtmp}) if (p) (void)hipFree(p);
    return rc;
  };
  MPM_HIP_CHECK(c, hipMalloc(&kb[0], (size_t)n * sizeof(unsigned)));
  if (hipMalloc(&kb[1], (size_t)n * sizeof(unsigned)) != hipSuccess || hipMalloc(&vtmp, (size_t)n * sizeof(int)) != hipSuccess)
    return done(fail(c, MPMHIP_ERR_HIP, "debug_sort: out of memory"));
  hipStream_t s = c->stream;
  (void)hipMemcpyAsync(kb[sort_input(f, n, bits)], keys_in, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(k_iota, nblk(n), TPB, 0, s, vtmp, n);
  int rc = sort_pairs(c, kb, vtmp, order_out, n, bits);
  if (rc == MPMHIP_OK) {
    (void)hipMemcpyAsync(keys_out, kb[1], (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, s);
    if (hipStreamSynchronize(s) != hipSuccess) rc = fail(c, MPMHIP_ERR_HIP, "debug_sort: stream error");
  }
  return done(rc);
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
