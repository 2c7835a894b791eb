// resort.hip -- the re-sort of the fast back end: keys, own radix sort, gather, block tables, chunk records, face bins; import / export of the caller's arrays.
// (split out of fast.hip in round 4; shared device code: fast_device.hpp, shared host state: fast_state.hpp)
#include "fast_state.hpp"

namespace mpm {

namespace {


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


// smallest positive and largest particle mass of the simulated particles (float bits; positive floats order like ints)
__global__ void k_mass_span(const float *mass, const int *sel, int n, int *counters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float m = (i < n && sel[i] == 0) ? mass[i] : 0.0f;
  int lo = m > 0.0f ? __float_as_int(m) : 0x7f7fffff, hi = m > 0.0f ? __float_as_int(m) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(counters + CNT_MMIN, lo); atomicMax(counters + CNT_MMAX, hi); }
  // ... and whether any particle is NOT simulated (selection != 0: frozen, or a ghost copy): the fused g2p + stress launch moves an
  // element's corners itself and only covers scenes in which every corner moves (FastState::all_simulated)
  unsigned long long ns = __ballot(i < n && sel[i] != 0);
  if ((threadIdx.x & 63) == 0 && ns) atomicAdd(counters + CNT_NSEL, (int)__popcll(ns));
}


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

__global__ void k_fbin_compact(const int *alist, const int *rc, int cap_A, const int *fb_start, const int *fb_cnt, FaceBin *list,
                               int cap_fbins) {
  const int n_A = min(rc[RC_NA], cap_A);
  int *counter = const_cast<int *>(rc) + RC_NFB;
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_A) return;
  int blk = alist[a];
  int cnt = fb_cnt[blk];
  if (cnt > 0) {
    // a bin of more than one workgroup's worth of faces (a coarse grid under a fine body mesh: hundreds of faces per block) becomes
    // several records of <= PT faces each, one splat workgroup per record -- they all flush into the same collider channels with
    // atomics anyway -- instead of one workgroup looping over the bin (round 4: the 64^3 training-size garment ran 2x slower than
    // the 128^3 one because a few such workgroups set the length of the launch)
    int parts = (cnt + PT - 1) / PT;
    int i = atomicAdd(counter, parts);
    if (i + parts <= cap_fbins) {
      int s0 = fb_start[blk];
      for (int k = 0; k < parts; ++k) list[i + k] = FaceBin{blk, s0 + k * PT, min(PT, cnt - k * PT), 0};
    } else {
      atomicOr(const_cast<int *>(rc) + RC_OVER, 8);
    }
  }
}

// vertex ids of the faces in bin order (one indirection less per substep)
__global__ void k_face_sorted_idx(const int32_t *idx, const int *order, int n_f, int *fidx) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_f) return;
  int f = order[j];
  fidx[3 * j] = idx[3 * f]; fidx[3 * j + 1] = idx[3 * f + 1]; fidx[3 * j + 2] = idx[3 * f + 2];
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

__global__ void k_iota(int *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

}  // namespace


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
    MPM_HIP_CHECK(c, hipMemsetD32Async((hipDeviceptr_t)(f->g.counters + CNT_MMAX), 0, 2, c->stream));   // (MMAX and NSEL)
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
  drop_kept_collider_fields(c);   // (a body at rest: its kept collider fields live on the OLD active list and in the old face bins)
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
    const int cap_fb = with_faces ? std::min(nf, cap_A + nf / PT + 1) : 0;   // (non-empty bins + the extra records of split bins)
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
      MPM_HIP_CHECK(c, hipMemcpyAsync(f->h_pin + 44, f->g.counters + CNT_MMIN, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    MPM_HIP_CHECK(c, hipStreamSynchronize(s));  // the one wait of a re-sort
    if (f->mass_span_pending) {
      // The fixed-point chunk tile gives every chunk ONE scale, from the sum of its lanes' bounds: a particle whose mass is
      // below ~1e-5 of its chunk mates' loses its contributions to rounding (measured, tools/gpu/mass_ratio.py: cloth beside
      // sand 1e+6 times heavier v 1.1e-2 against 8.6e-5 with the fp64 tile; at 1e+4 both 2e-4).  Scenes whose particle masses
      // span more than 1e+5 therefore run the fp64 tile (MPMHIP_P2G_TILE=fx overrides).
      float lo, hi;
      memcpy(&lo, f->h_pin + 44, 4); memcpy(&hi, f->h_pin + 45, 4);
      f->mass_span = (hi > 0.0f && lo < 3.0e38f) ? hi / lo : 1.0f;
      f->p2g_fixed_now = f->p2g_fixed && (f->p2g_fixed_forced || std::max(f->mass_span, f->global_mass_span) <= 1.0e5f);
      f->all_simulated = f->h_pin[46] == 0;
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

}  // namespace mpm
