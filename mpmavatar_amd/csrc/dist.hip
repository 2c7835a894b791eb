// dist.hip -- multi-GPU: halo / ghost kernels, peer links (HIP IPC), the RCCL binding and the in-library sharded loop.
// (split out of fast.hip in round 4; shared device code: fast_device.hpp, shared host state: fast_state.hpp)
#include "fast_state.hpp"

namespace mpm {

namespace {

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

}  // namespace


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
// the mass span of the whole sharded scene (all ranks): see mpmhip_dist_set_mass_span
int fast_dist_set_mass_span(mpmhip_ctx *c, float min_mass, float max_mass) {
  FastState *f = c->fast;
  f->global_mass_span = min_mass > 0.0f ? max_mass / min_mass : 0.0f;
  // (an import whose own span has already been read decides again; a pending one decides in rebin with the value just stored)
  if (!f->mass_span_pending)
    f->p2g_fixed_now = f->p2g_fixed && (f->p2g_fixed_forced || std::max(f->mass_span, f->global_mass_span) <= 1.0e5f);
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

}  // namespace mpm
