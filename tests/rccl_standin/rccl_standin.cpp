// TEST-ONLY stand-in for librccl: the slice of the RCCL C API that csrc/comm_rccl.cpp binds (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart / End),
// implemented between PROCESSES OF ONE HOST over POSIX shared memory, on HOST pointers.
//
// Why: the pool's GPU box has one MI355X and RCCL refuses two ranks per device, so the library's C++ transport
// (per-peer offsets, forward / reverse role swap, widths up to 32, reduction-buffer capacity) could only ever run as a
// one-rank communicator there.  With this stand-in, two CPU processes on the test-only host backend drive the REAL
// comm_rccl.cpp -- selected through GENEO_RCCL_LIBRARY, the path override of its load_api() -- and the result is
// compared with the serial oracle (tests/test_rccl_two_peers.py).  Never built by, shipped with or loaded from the
// geneo4petsc_amd package; it says nothing about xGMI performance.
//
// Semantics kept from NCCL: point-to-point operations inside a group are posted together and complete at
// ncclGroupEnd (no ordering between them, so a send / receive pair between two ranks cannot deadlock); a receive must
// match the size of the send; ncclAllReduce(sum, double) in place.  The stream argument is ignored (host memory, the
// call returns when the data has moved).
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
constexpr int kSum = 0, kDouble = 8;
constexpr size_t BOX_BYTES = (size_t)1 << 20;       // payload of one mailbox: larger messages travel in chunks
constexpr size_t RED_DOUBLES = (size_t)1 << 16;
constexpr double TIMEOUT_S = 120.0;

struct Box {                                       // one directed pair (src -> dst)
  std::atomic<uint64_t> written, read;
  uint64_t bytes;                                  // payload of the chunk in flight
  uint64_t total;                                  // size of the whole message (the receiver checks it)
  char data[BOX_BYTES];
};
struct Header {
  std::atomic<int> arrived, generation, attached;
};
struct Comm {
  int rank = 0, size = 1;
  std::string name;
  void* base = nullptr;
  size_t bytes = 0;
  Header* hdr = nullptr;
  Box* boxes = nullptr;                            // size * size, box(src, dst) = boxes[src * size + dst]
  double* red = nullptr;                           // size * RED_DOUBLES
};
struct Op { bool send; char* buf; size_t bytes, done; int peer; Comm* comm; };
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_ops;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int barrier(Comm* c) {
  const int g = c->hdr->generation.load(std::memory_order_acquire);
  if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->size) {
    c->hdr->arrived.store(0, std::memory_order_relaxed);
    c->hdr->generation.fetch_add(1, std::memory_order_release);
    return ncclSuccess;
  }
  const double t0 = now();
  while (c->hdr->generation.load(std::memory_order_acquire) == g) {
    sched_yield();
    if (now() - t0 > TIMEOUT_S) return ncclSystemError;
  }
  return ncclSuccess;
}

// one step of one operation; returns true when something moved
bool progress(Op& o, int* err) {
  if (o.done >= o.bytes && !(o.bytes == 0 && o.done == 0)) return false;
  Comm* c = o.comm;
  if (o.send) {
    Box& b = c->boxes[(size_t)c->rank * c->size + o.peer];
    if (b.written.load(std::memory_order_acquire) != b.read.load(std::memory_order_acquire)) return false;   // not consumed yet
    const size_t len = std::min(BOX_BYTES, o.bytes - o.done);
    std::memcpy(b.data, o.buf + o.done, len);
    b.bytes = len;
    b.total = o.bytes;
    b.written.fetch_add(1, std::memory_order_release);
    o.done += len;
    if (o.bytes == 0) o.done = 1;                  // an empty message is one empty chunk
    return true;
  }
  Box& b = c->boxes[(size_t)o.peer * c->size + c->rank];
  if (b.written.load(std::memory_order_acquire) == b.read.load(std::memory_order_acquire)) return false;       // nothing there
  if (b.total != o.bytes) { *err = ncclInvalidUsage; return false; }   // a receive must match its send
  std::memcpy(o.buf + o.done, b.data, b.bytes);
  o.done += b.bytes;
  if (o.bytes == 0) o.done = 1;
  b.read.fetch_add(1, std::memory_order_release);
  return true;
}

int run_ops(std::vector<Op>& ops) {
  const double t0 = now();
  for (;;) {
    bool all = true, moved = false;
    int err = 0;
    for (Op& o : ops) {
      const bool finished = (o.bytes == 0) ? (o.done == 1) : (o.done >= o.bytes);
      if (finished) continue;
      all = false;
      moved = progress(o, &err) || moved;
      if (err) { ops.clear(); return err; }
    }
    if (all) break;
    if (!moved) {
      sched_yield();
      if (now() - t0 > TIMEOUT_S) { ops.clear(); return ncclSystemError; }
    }
  }
  ops.clear();
  return ncclSuccess;
}

int post(bool send, void* buf, size_t count, int dtype, int peer, Comm* c) {
  if (!c || dtype != kDouble || peer < 0 || peer >= c->size) return ncclInvalidArgument;
  t_ops.push_back({send, (char*)buf, count * sizeof(double), 0, peer, c});
  if (t_group_depth == 0) return run_ops(t_ops);
  return ncclSuccess;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> counter{0};
  std::memset(id->internal, 0, sizeof(id->internal));
  std::snprintf(id->internal, sizeof(id->internal), "/geneo_rccl_standin_%d_%d_%llx", (int)getpid(), counter.fetch_add(1),
                (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank;
  c->size = nranks;
  c->name = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
  c->bytes = sizeof(Header) + sizeof(Box) * (size_t)nranks * nranks + sizeof(double) * RED_DOUBLES * nranks + 64;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);     // every rank may be the first: same size, zero-filled
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
  c->hdr = (Header*)c->base;
  c->boxes = (Box*)((char*)c->base + ((sizeof(Header) + 63) / 64) * 64);
  c->red = (double*)((char*)c->boxes + sizeof(Box) * (size_t)nranks * nranks);
  c->hdr->attached.fetch_add(1);
  const double t0 = now();                       // everybody attached before anybody uses a mailbox
  while (c->hdr->attached.load() < nranks) {
    sched_yield();
    if (now() - t0 > TIMEOUT_S) { munmap(c->base, c->bytes); delete c; return ncclSystemError; }
  }
  if (int rc = barrier(c)) { munmap(c->base, c->bytes); delete c; return rc; }
  if (rank == 0) shm_unlink(c->name.c_str());    // the mappings keep the segment alive; nothing is left behind in /dev/shm
  *comm = c;
  return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  munmap(c->base, c->bytes);
  delete c;
  return ncclSuccess;
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "stand-in: shared-memory failure or peer timed out";
    case ncclInvalidArgument: return "stand-in: invalid argument (only ncclDouble / ncclSum, peer in range)";
    case ncclInvalidUsage: return "stand-in: receive size does not match the send";
    default: return "stand-in: internal error";
  }
}

int ncclGroupStart() { ++t_group_depth; return ncclSuccess; }
int ncclGroupEnd() {
  if (t_group_depth <= 0) return ncclInvalidUsage;
  if (--t_group_depth > 0) return ncclSuccess;
  return run_ops(t_ops);
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, (Comm*)comm);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  return post(false, buf, count, dtype, peer, (Comm*)comm);
}

int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, void*) {
  Comm* c = (Comm*)comm;
  if (!c || dtype != kDouble || op != kSum) return ncclInvalidArgument;
  const double* in = (const double*)sendbuf;
  double* out = (double*)recvbuf;
  for (size_t off = 0; off < count || (count == 0 && off == 0); off += RED_DOUBLES) {
    const size_t len = std::min(RED_DOUBLES, count - off);
    std::memcpy(c->red + (size_t)c->rank * RED_DOUBLES, in + off, len * sizeof(double));
    if (int rc = barrier(c)) return rc;
    for (size_t i = 0; i < len; ++i) {             // rank order: every rank computes the same bits
      double s = 0.0;
      for (int r = 0; r < c->size; ++r) s += c->red[(size_t)r * RED_DOUBLES + i];
      out[off + i] = s;
    }
    if (int rc = barrier(c)) return rc;
    if (count == 0) break;
  }
  return ncclSuccess;
}

}  // extern "C"
