// TEST INFRASTRUCTURE -- not part of the product, never loaded unless MPMHIP_RCCL_LIB points at it.
//
// The ten RCCL entry points libmpmhip.so resolves with dlsym (csrc/fast.hip, struct Rccl), implemented over a POSIX
// shared-memory segment with host-staged copies, so that `mpmhip_rccl_steps` -- the in-library multi-GPU substep loop --
// can execute with 2-3 ranks on a box that has ONE GPU (real RCCL refuses two ranks on one device: "Duplicate GPU
// detected").  Semantics kept: send/recv are queued between ncclGroupStart/End and matched per ordered pair in issue
// order; element counts of a send and its recv must agree (a mismatch is an error here, where real RCCL would hang or
// corrupt); collectives are blocking; everything is ordered after the work already on the given stream and complete
// when the call returns.  Every wait has a timeout, so that a protocol bug fails a test instead of hanging the box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int MAX_WORLD = 8;
constexpr double TIMEOUT_S = 120.0;

struct Mailbox {  // one per ordered pair (src -> dst)
  std::atomic<uint64_t> written, read;
  uint64_t bytes;
  uint64_t pad[5];
};
struct Header {
  std::atomic<uint32_t> arrived, generation;  // sense-reversing barrier
  std::atomic<uint32_t> attached;
  uint32_t world;
  uint64_t box_bytes;
  Mailbox box[MAX_WORLD * MAX_WORLD];
};

struct Op { bool send; void *ptr; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
  Header *h = nullptr;
  unsigned char *data = nullptr;  // [world*world] mailboxes of box_bytes, then [world] collective slots of box_bytes
  size_t map_bytes = 0;
  int rank = 0, world = 1;
  std::string name;
  std::vector<unsigned char> stage;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;
thread_local char g_err[256] = "mock rccl: no error";

double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
ncclResult_t fail(const char *what) {
  snprintf(g_err, sizeof g_err, "mock rccl: %s", what);
  fprintf(stderr, "[mock_rccl] %s\n", what);
  return ncclSystemError;
}
template <class F> bool wait_for(F cond) {
  double t0 = now();
  for (int spin = 0; !cond(); ++spin) {
    if (spin > 64) sched_yield();
    if ((spin & 1023) == 1023) {
      if (now() - t0 > TIMEOUT_S) return false;
      usleep(50);
    }
  }
  return true;
}
size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
unsigned char *box_data(Comm *c, int src, int dst) { return c->data + ((size_t)src * c->world + dst) * c->h->box_bytes; }
unsigned char *slot_data(Comm *c, int r) { return c->data + ((size_t)c->world * c->world + r) * c->h->box_bytes; }

bool barrier(Comm *c) {
  uint32_t gen = c->h->generation.load(std::memory_order_acquire);
  if (c->h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    c->h->arrived.store(0, std::memory_order_relaxed);
    c->h->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  return wait_for([&] { return c->h->generation.load(std::memory_order_acquire) != gen; });
}

ncclResult_t run_group(Comm *c, std::vector<Op> &ops) {
  // everything queued on the streams first (the pack kernels), then all sends, then all receives
  for (auto &o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("hipStreamSynchronize failed");
  for (auto &o : ops) {
    if (!o.send) continue;
    if (o.bytes > c->h->box_bytes) return fail("message larger than MOCK_RCCL_BOX_BYTES");
    Mailbox &m = c->h->box[c->rank * MAX_WORLD + o.peer];
    if (!wait_for([&] { return m.read.load(std::memory_order_acquire) == m.written.load(std::memory_order_relaxed); }))
      return fail("timeout: the previous message to this peer was never received");
    if (hipMemcpy(box_data(c, c->rank, o.peer), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H copy failed");
    m.bytes = o.bytes;
    m.written.fetch_add(1, std::memory_order_release);
  }
  for (auto &o : ops) {
    if (o.send) continue;
    Mailbox &m = c->h->box[o.peer * MAX_WORLD + c->rank];
    if (!wait_for([&] { return m.written.load(std::memory_order_acquire) > m.read.load(std::memory_order_relaxed); }))
      return fail("timeout: a receive without a matching send");
    if (m.bytes != o.bytes) {
      char b[160];
      snprintf(b, sizeof b, "size mismatch: rank %d receives %zu bytes from rank %d, which sent %llu", c->rank, o.bytes, o.peer,
               (unsigned long long)m.bytes);
      return fail(b);
    }
    if (hipMemcpy(o.ptr, box_data(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D copy failed");
    m.read.fetch_add(1, std::memory_order_release);
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/mpmhip_mock_rccl_%d_%llx", (int)getpid(), (unsigned long long)(now() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return fail("bad rank / world");
  Comm *c = new Comm;
  c->rank = rank; c->world = world; c->name = id.internal;
  const char *e = getenv("MOCK_RCCL_BOX_BYTES");
  size_t box_bytes = e ? (size_t)atoll(e) : ((size_t)4 << 20);
  c->map_bytes = sizeof(Header) + ((size_t)world * world + world) * box_bytes;
  int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) return fail("shm_open failed");
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); return fail("ftruncate failed"); }  // new pages read as zero
  void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("mmap failed");
  c->h = (Header *)p;
  c->data = (unsigned char *)p + sizeof(Header);
  if (rank == 0) { c->h->world = (uint32_t)world; c->h->box_bytes = box_bytes; }
  c->h->attached.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_for([&] { return c->h->attached.load(std::memory_order_acquire) >= (uint32_t)world; })) return fail("timeout: not every rank attached");
  if (!wait_for([&] { return c->h->box_bytes == box_bytes; })) return fail("ranks disagree about MOCK_RCCL_BOX_BYTES");
  if (!barrier(c)) return fail("timeout in the attach barrier");
  if (rank == 0) shm_unlink(c->name.c_str());  // every rank has it mapped: the name can go
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  munmap((void *)c->h, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (ops.empty()) return ncclSuccess;
  return run_group(g_comm, ops);
}

static ncclResult_t p2p(bool send, void *ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  Comm *c = (Comm *)comm;
  if (peer < 0 || peer >= c->world || peer == c->rank) return fail("bad peer");
  g_comm = c;
  g_ops.push_back(Op{send, ptr, count * dtype_size(t), peer, s});
  if (g_depth == 0) {  // outside a group: one blocking operation
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_group(c, ops);
  }
  return ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(true, const_cast<void *>(buf), count, t, peer, comm, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(false, buf, count, t, peer, comm, s);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  Comm *c = (Comm *)comm;
  size_t bytes = count * dtype_size(t);
  if (bytes > c->h->box_bytes) return fail("all-gather chunk larger than MOCK_RCCL_BOX_BYTES");
  if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize failed");
  if (hipMemcpy(slot_data(c, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H copy failed");
  if (!barrier(c)) return fail("timeout in all-gather");
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy((unsigned char *)recv + (size_t)r * bytes, slot_data(c, r), bytes, hipMemcpyHostToDevice) != hipSuccess)
      return fail("H2D copy failed");
  if (!barrier(c)) return fail("timeout in all-gather");
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  Comm *c = (Comm *)comm;
  if (t != ncclInt32 || (op != ncclMax && op != ncclSum)) return fail("all-reduce: only int32 max / sum");
  size_t bytes = count * 4;
  if (bytes > c->h->box_bytes) return fail("all-reduce larger than MOCK_RCCL_BOX_BYTES");
  if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize failed");
  if (hipMemcpy(slot_data(c, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H copy failed");
  if (!barrier(c)) return fail("timeout in all-reduce");
  std::vector<int32_t> acc(count);
  memcpy(acc.data(), slot_data(c, 0), bytes);
  for (int r = 1; r < c->world; ++r) {
    const int32_t *v = (const int32_t *)slot_data(c, r);
    for (size_t i = 0; i < count; ++i) acc[i] = op == ncclMax ? (v[i] > acc[i] ? v[i] : acc[i]) : acc[i] + v[i];
  }
  if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D copy failed");
  if (!barrier(c)) return fail("timeout in all-reduce");
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t) { return g_err; }

}  // extern "C"
