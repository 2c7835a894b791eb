// p2g_device.hpp -- device side of p2g (p2g_apic_with_stress, mpm_utils.py:484-557): the body-face and joint splats, the chunk tile in
// packed fixed point, the segmented DPP pre-reduction, scatter / flush and the chunk workgroup (p2g_body).  Kernels: p2g.hip.
#pragma once
#include "fast_device.hpp"

namespace mpm {
inline namespace fk {

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

// non-empty face bins that lie on the active list (order irrelevant), as self-contained records
struct FaceBin { int blk, start, cnt, pad; };

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

// One-pass form of the small-bin splat (PASSES == 3): all seven collider channels (weight, weight * velocity, weight * normal) in
// one tile of 8 x 8 x 8 nodes at strides (67, 8, 1) -- 2 * (67 i + 8 j + k) mod 64 puts 25 of a face's 27 nodes into different bank
// pairs -- so that a bin costs one clearing, one scatter and one flush instead of two of each with five barriers in between.  The
// workgroup tile is 7 * 536 doubles = 30 KB instead of 24.6 KB: still five workgroups per CU (VGPR-bound at five).
constexpr int SPLAT7_SI = 67, SPLAT7_SJ = 8, SPLAT7_S = 536;  // 7*67 + 7*8 + 7 = 532 < 536
#ifndef SPLAT_ONEPASS
#define SPLAT_ONEPASS 1  // experiment switch: 0 = small bins take the two-pass path through the four-channel tile (24.6 KB per workgroup)
#endif
constexpr int P2G_TILE_DOUBLES = (SPLAT_ONEPASS && 7 * SPLAT7_S > 4 * TILE_PAD) ? 7 * SPLAT7_S : 4 * TILE_PAD;
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
  // (the one-pass path takes the ballot -- i.e. the wait for the flags -- right before its flush: in front of the tile clearing it was
  // one more dependent memory level at the head of the workgroup)
  const bool one_pass = PASSES == 3 && SPLAT_ONEPASS && fb.cnt <= SPLAT_SMALL;
  unsigned long long act_mask = one_pass ? 0ull : __ballot(nb_act);
  const int end = fb.start + fb.cnt;
  if (fb.cnt <= SPLAT_SMALL) {
    // SMALL BIN (the common case once the cloth has draped: ~740 bins of ~27 faces): lane = (face, stencil node), 8 faces x 32
    // lanes (27 used) per step, <= 4 steps -- instead of lane = face with a 27-trip node loop of dependent DPP scans that 230
    // of the 256 lanes sit out.  Such a workgroup used to live 10-17 us (two 3 us scatter passes, profiles/r03_wg_timeline.md);
    // what is left is its chain of loads and the two flushes.  The per-step weights and normals stay in registers for the
    // second (normal) pass through the four-channel tile.
    const int fi = l >> 5, n = l & 31;
    const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
    if (PASSES == 3 && SPLAT_ONEPASS) {  // one pass through a seven-channel tile (see SPLAT7_S)
      // the first pair's face indices are requested together with the block flags, BEFORE the tile is cleared: behind the barrier they
      // were a memory level of their own (record -> flags -> [clear, barrier] -> indices -> vertices; now record -> flags + indices -> vertices)
      int pre_i[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = u * 8 + fi;
        const int jq = q < fb.cnt ? fb.start + q : fb.start;
        pre_i[u][0] = sa.fidx[3 * jq]; pre_i[u][1] = sa.fidx[3 * jq + 1]; pre_i[u][2] = sa.fidx[3 * jq + 2];
      }
      for (int t = l; t < 7 * SPLAT7_S; t += PT) tile[t] = 0.0;
      __syncthreads();
      WGT(g, 0, 2);  // (debug build: bin record, block flags, tile cleared)
      // Two steps' loads in flight at a time, then their LDS atomics; the global atomics of the out-of-margin lanes wait until all
      // steps are through.  With those inside the load loop (they may alias the vertex arrays) the compiler kept the four steps in
      // order and a bin paid index -> vertex latency four times: 5.3 us of the workgroup's 11 (profiles/r04_experiments.md 15); all
      // four steps' loads at once are 84 registers of raw vertex data and cost the whole kernel a wavefront per SIMD.
      auto face_eval = [&](int it, float &w, V3 &a, V3 &fn, Stencil &s, bool pre = false) -> bool {
        const int q = it * 8 + fi;
        const bool have = q < fb.cnt && n < 27;
        const int jq = q < fb.cnt ? fb.start + q : fb.start;
        int i0, i1, i2;
        if (pre) { i0 = pre_i[it & 1][0]; i1 = pre_i[it & 1][1]; i2 = pre_i[it & 1][2]; }  // (it < 2 only)
        else { i0 = sa.fidx[3 * jq]; i1 = sa.fidx[3 * jq + 1]; i2 = sa.fidx[3 * jq + 2]; }
        V3 p0 = mesh_point(sa.pts, sa.vel, sa.adv, i0), p1 = mesh_point(sa.pts, sa.vel, sa.adv, i1), p2 = mesh_point(sa.pts, sa.vel, sa.adv, i2);
        V3 u0 = load_v3(sa.vel + 3 * i0), u1 = load_v3(sa.vel + 3 * i1), u2 = load_v3(sa.vel + 3 * i2);
        V3 fp = v3((p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f);
        a = v3((u0.x + u1.x + u2.x) / 3.0f, (u0.y + u1.y + u2.y) / 3.0f, (u0.z + u1.z + u2.z) / 3.0f);
        fn = normalize(cross(p1 - p0, p2 - p0));  // wp.mesh_eval_face_normal
        s = make_stencil(fp, d.inv_dx);
        w = sel3(ni, s.w0.x, s.w1.x, s.w2.x) * sel3(nj, s.w0.y, s.w1.y, s.w2.y) * sel3(nk, s.w0.z, s.w1.z, s.w2.z);
        return have && splat_ok(d.G, s);  // mpm_solver.py:858
      };
      unsigned esc_mask = 0;  // steps whose face left the tile margin since the faces were binned
#pragma unroll
      for (int h = 0; h < SPLAT_SMALL / 8; h += 2) {
        if (h * 8 >= fb.cnt) break;  // (workgroup-uniform: a bin of at most 16 faces -- the average is 13 -- is done after the first pair)
        float w[2];
        V3 a[2], fn[2];
        int off[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          Stencil s;
          const bool ok = face_eval(h + u, w[u], a[u], fn[u], s, h == 0);
          const int lx = s.bx - ox, ly = s.by - oy, lz = s.bz - oz;
          const bool in_tile = !((unsigned)lx > 5u || (unsigned)ly > 5u || (unsigned)lz > 5u);
          off[u] = (ok && in_tile) ? (lx + ni) * SPLAT7_SI + (ly + nj) * SPLAT7_SJ + (lz + nk) : -1;
          if (ok && !in_tile) esc_mask |= 1u << (h + u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (off[u] >= 0) {
            double *p = tile + off[u];
            atomicAdd(p, (double)w[u]);
            atomicAdd(p + SPLAT7_S, (double)(w[u] * a[u].x)); atomicAdd(p + 2 * SPLAT7_S, (double)(w[u] * a[u].y));
            atomicAdd(p + 3 * SPLAT7_S, (double)(w[u] * a[u].z));
            atomicAdd(p + 4 * SPLAT7_S, (double)(w[u] * fn[u].x)); atomicAdd(p + 5 * SPLAT7_S, (double)(w[u] * fn[u].y));
            atomicAdd(p + 6 * SPLAT7_S, (double)(w[u] * fn[u].z));
          }
        asm volatile("" : "+v"(esc_mask)::"memory");  // (the next pair's loads stay behind this pair's)
      }
      if (esc_mask) {  // rare: this lane's node through global atomics; the flag makes the next re-sort bin the faces again
        raise_drift(g.counters, g.step_id);
        raise_face(g.counters, g.step_id);
#pragma unroll 1
        for (int it = 0; it < SPLAT_SMALL / 8; ++it) {
          if (!((esc_mask >> it) & 1u)) continue;
          float w;
          V3 a, fn;
          Stencil s;
          (void)face_eval(it, w, a, fn, s);
          int x = s.bx + ni, y = s.by + nj, z = s.bz + nk;
          int nb = blk_of(x, y, z, d.NB);
          if (g.ab_flag[nb]) {
            float *p = g.col + ((size_t)nb * GCH_COL) * 64 + loc_of(x, y, z);
            __hip_atomic_store(&g.col_flag[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(p, w);
            atomicAdd(p + 64, w * a.x); atomicAdd(p + 128, w * a.y); atomicAdd(p + 192, w * a.z);
            atomicAdd(p + 256, w * fn.x); atomicAdd(p + 320, w * fn.y); atomicAdd(p + 384, w * fn.z);
          }
        }
      }
      WGT(g, 0, 3);  // faces loaded, LDS atomics of wavefront 0 out
      act_mask = __ballot(nb_act);
      __syncthreads();
      WGT(g, 0, 4);
      for (int t = l; t < TILE3; t += PT) {  // (same rules as col_splat_flush<0> and <1>: a node without weight got nothing at all)
        int ti = t >> 6, tj = (t >> 3) & 7, tk = t & 7;
        const double *q = tile + (ti * SPLAT7_SI + tj * SPLAT7_SJ + tk);
        float c0 = (float)q[0];
        if (c0 == 0.0f) continue;
        int x = ox + ti, y = oy + tj, z = oz + tk;
        if (!in_grid(x, y, z, d.G)) continue;
        int nb = blk_of(x, y, z, d.NB);
        int nidx = (((x >> 2) - bx + 1) * 3 + ((y >> 2) - by + 1)) * 3 + ((z >> 2) - bz + 1);
        if (!((act_mask >> nidx) & 1ull)) continue;  // inactive block: never read by g2p, never re-zeroed
        float *p = g.col + ((size_t)nb * GCH_COL) * 64 + loc_of(x, y, z);
        atomicAdd(p, c0);
        atomicAdd(p + 64, (float)q[SPLAT7_S]); atomicAdd(p + 128, (float)q[2 * SPLAT7_S]); atomicAdd(p + 192, (float)q[3 * SPLAT7_S]);
        atomicAdd(p + 256, (float)q[4 * SPLAT7_S]); atomicAdd(p + 320, (float)q[5 * SPLAT7_S]); atomicAdd(p + 384, (float)q[6 * SPLAT7_S]);
        __hip_atomic_store(&g.col_flag[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    float wk[SPLAT_SMALL / 8];
    V3 fnk[SPLAT_SMALL / 8];
    int basek[SPLAT_SMALL / 8];
    for (int t = l; t < 4 * TILE_PAD; t += PT) tile[t] = 0.0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SPLAT_SMALL / 8; ++it) {
      wk[it] = 0.0f; fnk[it] = v3(0, 0, 0); basek[it] = 0;
      if (it * 8 >= fb.cnt) continue;  // (workgroup-uniform: no face left for this step)
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

// JT = true: the mover holds MANY traditional particles (run_demo.py keeps 100k sand particles frozen for the first
// frames); their joint splat (weight, weight * joint velocity into the mover channels, mpm_solver.py:677-704) is a
// second pass through the same LDS tile by the chunk that owns them instead of 27 x 4 scattered global atomics each.
template <int STEPS, bool TRAD, bool JT, bool FX>
__device__ __forceinline__ void p2g_body(const ChunkRec *recs, int n_chunks, const Bufs &b, const VAdj &va, const Dims &d, float rpic,
                                         float dt, const GridPtrs &g, const SplatArgs &sa, const TradParams &tp, double *tile, int *esc,
                                         int &esc_n, float *red, int bid) {
  WGT(g, 0, 0);
  if (bid == 0 && threadIdx.x == 0 && g.host_sig) {
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
  if (bid >= sa.e0 && bid < sa.e0 + sa.n_extra) {
    int e = bid - sa.e0;
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
  if (bid >= sa.z_first) {  // ... and the clearing workgroups last: they fill the tail of the launch
    if (sa.pack.n_wg && bid >= sa.pack.first) {  // (multi-GPU) halo pack, once everything in front has scattered
      pack_wait(sa.pack, g.counters + 10);
      halo_pack_wg<true>(sa.pack.tb, g, bid - sa.pack.first);
      return;
    }
    if (!DBG(g, 2048)) zero_blocks_wg(sa.z, bid - sa.z_first);
    WGT(g, 0, 6);
    return;
  }
  int w = xcd_slice(bid - (sa.e0 == 0 ? sa.n_extra : 0), n_chunks);
  if (w < 0) { wg_done(sa.pack); return; }
  if (g.stagger > 0 && bid < g.stagger_first) {
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
}  // namespace fk
}  // namespace mpm
