// fast_device.hpp -- device-side library of the fast back end (MPMHIP_MODE_FAST): layout types, the grid stage, the pieces of
// p2g / g2p / the re-sort / the multi-GPU exchange that are not kernels.  Kernels live in the .hip file that launches them
// (resort.hip, p2g.hip, g2p.hip, dist.hip, fast.hip); everything here is inline.
#pragma once
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
inline namespace fk {


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
#ifndef MPM_XCD_RUN
#define MPM_XCD_RUN 16  // (experiment switch; 8 and 32 measured in round 5)
#endif
constexpr int XCD_RUN = MPM_XCD_RUN;
__device__ __forceinline__ int xcd_slice(int w, int n) {
  int xcd = w & 7, idx = w >> 3;
  int i = ((idx / XCD_RUN) * 8 + xcd) * XCD_RUN + (idx % XCD_RUN);
  return i < n ? i : -1;
}
inline unsigned xcd_grid(int n) { return (unsigned)(((n + 8 * XCD_RUN - 1) / (8 * XCD_RUN)) * (8 * XCD_RUN)); }

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

// rc: re-sort counts kept on the device so that the table kernels can be enqueued back to back without a host round trip
// (the host reads them once, at the end): [0] particle blocks, [1] active blocks, [2] chunks, [3] chunks incl. ghost copies,
// [4] any ghost copy, [5] capacity overflow bits (1 plist / ranges, 2 alist, 4 chunk records, 8 face bins), [6] face bins
enum { RC_NP = 0, RC_NA = 1, RC_NCH = 2, RC_NCHG = 3, RC_GHOST = 4, RC_OVER = 5, RC_NFB = 6, RC_N = 8 };

// The same compaction without a scan launch in front (the re-sort's two block lists; rocPRIM's scan is two launches): k_flag_count
// files the flagged blocks per tile of FC_TILE flags and per group of FC_GROUP tiles, k_compact_tiles works out every tile's start
// from those (uniform loads: <= groups + FC_GROUP scalars) and scans inside the tile.  index[] is filled as the exclusive scan
// would have filled it.
constexpr int FC_TILE = 1024, FC_GROUP = 16;

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
enum { CNT_FACE = 5, CNT_DRIFT = 6, CNT_PAR0 = 16, CNT_MMIN = 24, CNT_MMAX = 25, CNT_NSEL = 26, CNT_N = 32 };  // (MMIN / MMAX: smallest positive / largest
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
  // Every load that depends on e alone goes out FIRST, in one burst, and every store comes after the last load: the pointers are not
  // restrict-qualified, so a store (or the drift flag) between two loads pins the later one behind it, and the straightforward order
  // -- selection -> vertex slots -> vertices -> store x, v -> sort key -> director -> gamma, kappa -> store d3 -> R^-1, vol, mu, lam
  // -> store stress -- was a chain of SEVEN dependent memory levels per wavefront (ISA: one s_waitcnt vmcnt(0) after each), which is
  // what a launch of one round of workgroups lasts.  Now: e-indexed data -> vertices -> stores.
  const int sel = b.sel[e];
  int s1 = 0, s2 = 0, s3 = 0;
  SortKey sk = 0;
  M3 dm;
  if (FINALIZE) {
    s1 = face_slot[e]; s2 = face_slot[d.n_e + e]; s3 = face_slot[2 * d.n_e + e];
    sk = skeys[e];
    dm = m3_zero();
    dm.a02 = b.el.at(E_D + 2, e); dm.a12 = b.el.at(E_D + 5, e); dm.a22 = b.el.at(E_D + 8, e);
  } else {
    dm = ld9(b.el, E_D, e);
  }
  float gamma = b.el.at(E_GAMMA, e), kappa = b.el.at(E_KAPPA, e);
  V3 rinv = ld3(b.el, E_RINV, e);
  float vol = b.nv.at(N_VOL, e), mu = b.nv.at(N_MU, e), lam = b.nv.at(N_LAM, e);
  // (an empty asm that "uses" the loaded values HERE: without it LLVM sinks the loads into the blocks behind the branches below, next
  // to their first use, and the chain is back)
  asm volatile("" : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(sk), "+v"(dm.a02), "+v"(dm.a12), "+v"(dm.a22));
  asm volatile("" : "+v"(gamma), "+v"(kappa), "+v"(rinv.x), "+v"(rinv.y), "+v"(rinv.z), "+v"(vol), "+v"(mu), "+v"(lam));
  if (sel == 1) {  // not simulated (selection == 2 marks a ghost copy: stress yes, transfers no)
    for (int c = 0; c < 3; ++c) ef[c * d.n_e + e] = F3{0.0f, 0.0f, 0.0f};
    return;
  }
  V3 xe = v3(0, 0, 0), ve = v3(0, 0, 0);
  if (FINALIZE) {
    int v1 = d.n_nv + s1, v2 = d.n_nv + s2, v3i = d.n_nv + s3;
    V3 x1 = ld3(b.all, A_X, v1), x2 = ld3(b.all, A_X, v2), x3 = ld3(b.all, A_X, v3i);
    V3 u1 = ld3(b.all, A_V, v1), u2 = ld3(b.all, A_V, v2), u3 = ld3(b.all, A_V, v3i);
    ve = v3((u1.x + u2.x + u3.x) / 3.0f, (u1.y + u2.y + u3.y) / 3.0f, (u1.z + u2.z + u3.z) / 3.0f);
    xe = v3((x1.x + x2.x + x3.x) / 3.0f, (x1.y + x2.y + x3.y) / 3.0f, (x1.z + x2.z + x3.z) / 3.0f);
    V3 d1 = x2 - x1, d2 = x3 - x1;
    dm = m3_cols(d1, d2, v3(dm.a02, dm.a12, dm.a22));
    // d1, d2 are not stored here: nothing reads them before the next finalize (every consumer of finished elements --
    // re-sort, read-back, ghosts -- runs k_elem_finalize first, which recomputes them from the vertices)
  }
  QR3 q = qr_cloth(dm);
  float r02, r12, r22;
  V3 d3 = anisotropy_return_mapping(q, gamma, kappa, friction_coeff, r02, r12, r22);
  M3 stress;
  V3 f1, f2, f3;
  kirchhoff_anisotropy(q, r02, r12, r22, d3, rinv, vol, mu, lam, gamma, kappa, stress, f1, f2, f3);
  if (FINALIZE) {
    st3(b.all, A_V, e, ve);
    st3(b.all, A_X, e, xe);
  }
  b.el.at(E_D + 2, e) = d3.x; b.el.at(E_D + 5, e) = d3.y; b.el.at(E_D + 8, e) = d3.z;
  st9(b.nv, N_STRESS, e, stress);
  ef[e] = F3{f1.x, f1.y, f1.z};
  ef[d.n_e + e] = F3{f2.x, f2.y, f2.z};
  ef[2 * d.n_e + e] = F3{f3.x, f3.y, f3.z};
  if (FINALIZE) {  // drift check against the block this element was sorted into
    int blk = key_block(sk, blk_bits);
    int oz = 4 * (blk % d.NB) - 1, oy = 4 * ((blk / d.NB) % d.NB) - 1, ox = 4 * (blk / (d.NB * d.NB)) - 1;
    // (no look-ahead here: an element follows its three vertices, whose g2p raises the flag early, see g2p_write)
    int nbx = (int)(xe.x * d.inv_dx - 0.5f) - ox, nby = (int)(xe.y * d.inv_dx - 0.5f) - oy, nbz = (int)(xe.z * d.inv_dx - 0.5f) - oz;
    if (sel == 0 && ((unsigned)nbx > 5u || (unsigned)nby > 5u || (unsigned)nbz > 5u)) raise_drift(counters, step_id);
  }
}

// ------------------------------------------------------------------------------------------------
// chunk tiles and chunk records: shared by p2g (p2g_device.hpp) and g2p (g2p_device.hpp)
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

// handshake at link set-up: `n` pattern words through the link's data area, checked on the other side (rccl_link_setup)
__device__ __forceinline__ unsigned link_pattern(int seq, int i) { return (unsigned)i * 2654435761u ^ ((unsigned)seq * 0x9E3779B9u); }


}  // namespace fk
}  // namespace mpm
