// fast_state.hpp -- host-side state of the fast back end (FastState and what hangs on it) and the functions its translation
// units share.
#pragma once
#include "g2p_device.hpp"

namespace mpm {


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
  float global_mass_span = 0.0f;  // sharded runs: the span over ALL ranks' simulated particles (mpmhip_dist_set_mass_span); 0: not set
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
  bool all_simulated = false;      // no particle with selection != 0 (counted at every import with the mass span)
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
  // A body AT REST inside one mpmhip_steps call (mesh_v == 0 for every vertex: checked once per call, fast_body_at_rest): its collider
  // field -- weight, weight * velocity, weight * normal per node -- is the same in every substep, so it is splatted ONCE into each of
  // the two accumulator buffers and then kept: no splat workgroups, no clearing of the collider channels, until the particle order
  // (the face bins, the active list) changes or the call ends.  col_state[buffer]: 0 clean, 1 holds this substep's splat (cleared by
  // the next launch's clearing workgroups, as ever), 2 holds the field of the body at rest (kept).
  bool col_at_rest = false;      // this mpmhip_steps call's body does not move
  int col_state[3] = {0, 0, 0};
  int64_t n_col_kept = 0;        // substeps that ran without splat workgroups because the field was kept (statistics)
  // accumulator double buffer: g.{mv,col,mov,m_flag,col_flag} point at buffer `par`
  // (three for scenes that can run the fused g2p -> p2g launch, k_g2p2g: read / write / clear)
  float *mv2[3] = {nullptr, nullptr, nullptr}, *col2[3] = {nullptr, nullptr, nullptr}, *mov2[3] = {nullptr, nullptr, nullptr};
  int *mflag2[3] = {nullptr, nullptr, nullptr}, *cflag2[3] = {nullptr, nullptr, nullptr};
  int par = 0, nbuf = 2;
  // G2P2G: the g2p of the last substep has not been launched yet -- the next substep's launch does it in front of its own p2g
  // (k_g2p2g), or flush_g2p() does with a plain k_g2p when anything else needs the particles first
  bool col_keep = true;        // MPMHIP_COL_KEEP=0: splat the body every substep whether it moves or not (A/B)
  int *zero_flags = nullptr;   // [blocks] zeros (drop_kept_collider_fields)
  bool g2p2g = true;           // MPMHIP_G2P2G=0: two launches per substep for traditional-only scenes as before
  int g2p2g_max_chunks = 512;  // MPMHIP_G2P2G_MAX
  int stagger_auto = 2;        // p2g first-round stagger units for chunk lists of at least two rounds; -1: forced by MPMHIP_P2G_STAGGER
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


template <class T>
int dalloc(mpmhip_ctx *c, T **p, size_t count, bool zero = true) {
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  MPM_HIP_CHECK(c, hipMalloc((void **)p, bytes));
  c->fast->allocs.push_back((void *)*p);
  if (zero) MPM_HIP_CHECK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
  return MPMHIP_OK;
}

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


// ---- functions shared by the translation units of the fast back end ------------------------------------------------------
// resort.hip
int alloc_bufs(mpmhip_ctx *c, Bufs &b);
int ensure_cap(mpmhip_ctx *c, int **p, int *cap, int need, int per);
int scan_flags(mpmhip_ctx *c, const int *flag, int *index, int n, int *total);
int scan_flags_dev(mpmhip_ctx *c, const int *flag, int *index, int n);
int scan_flags_async(mpmhip_ctx *c, const int *flag, int *index, int n, int slot);
int do_import(mpmhip_ctx *c);
int rebin(mpmhip_ctx *c);
// fast.hip
int flush_elements(mpmhip_ctx *c);
int flush_g2p(mpmhip_ctx *c);
ZeroArgs take_zero(FastState *f);
void select_buffer(FastState *f, int par);
void flush_grid(mpmhip_ctx *c);
void materialize_grid(mpmhip_ctx *c, bool count);
void drop_kept_collider_fields(mpmhip_ctx *c);
int step_phase_a(mpmhip_ctx *c, const StepArgs &a);
int step_phase_b(mpmhip_ctx *c, const StepArgs &a);
int step_phase_c(mpmhip_ctx *c, const StepArgs &a);
// p2g.hip / g2p.hip: the only places that name the template instantiations of the substep's kernels
void launch_p2g(mpmhip_ctx *c, bool trad, bool jt, unsigned grid, int n_chunks, float dt, const SplatArgs &sa, const TradParams &tp);
void launch_stress_elem(mpmhip_ctx *c, int mode, const SplatArgs &sa);
void launch_stress_trad(mpmhip_ctx *c, float dt);
void launch_g2p(mpmhip_ctx *c, bool fused, bool two, float dt, const GridParams &gp, const BCList &bcl);
// api.hip
void mesh_store_launch(mpmhip_ctx *c, const StepArgs &a);
void launch_g2p2g(mpmhip_ctx *c, unsigned grid, float dt, const GridRead &rd, const SplatArgs &sa, const TradParams &tp, const GridParams &gp,
                  const BCList &bcl);


}  // namespace mpm
