// C++ transport of libgeneopc over RCCL / xGMI: one process per GPU, no Python in the data path.
//
// The GenEO hot path has exactly two exchange steps (the reference's VecScatter at geneo.cpp:1850 / :1881
// and its MPI reductions at :1474):
//   * halo forward / reverse  -- neighbour-only point-to-point: one ncclGroup of ncclSend / ncclRecv pairs per exchange,
//     which RCCL maps to the direct xGMI link of each neighbour (<= 7 per GPU on one node);
//   * small all-reduce (Z^T x of dimE doubles, Krylov dots, sizes) -- ncclAllReduce in place, latency-bound.
// Both are enqueued on the library's launch stream (bk::get_stream()), so they are stream-ordered with the kernels that
// pack the send buffer and consume the receive buffer: no host synchronisation anywhere.
//
// RCCL is resolved at run time (dlopen): a process that already carries an RCCL (PyTorch bundles one) keeps exactly that
// instance; libgeneopc itself has no link-time dependency on it and single-GPU users never load it.  The communicator is
// bootstrapped from a 128-byte unique id that the HOST distributes (rank 0 creates it with GeneoRcclUniqueId and
// broadcasts the bytes by whatever channel it has: MPI_Bcast, torch.distributed, a file).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/geneo_c.h"
#include "backend.h"

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x: rccl.h:40-43,:187,:220,:260,:339,:448-467,:611,:700)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;          // ncclSuccess = 0
enum { kNcclSum = 0, kNcclDouble = 8 };
struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, void*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, void*) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, void*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
Api g_api;
std::string g_err;

bool load_api() {
  if (g_api.handle) return true;
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  // GENEO_RCCL_LIBRARY: full path of the library to bind instead (a site build of RCCL; the two-process CPU test of this
  // transport binds a shared-memory stand-in this way, tests/test_rccl_two_peers.py)
  const char* over = getenv("GENEO_RCCL_LIBRARY");
  if (over && *over) {
    if (!(h = dlopen(over, RTLD_NOW | RTLD_LOCAL))) {
      g_err = std::string("GenEO: cannot load GENEO_RCCL_LIBRARY ") + over + ": " + (dlerror() ? dlerror() : "not found");
      return false;
    }
  }
  for (const char* n : names)            // the instance the process already has (same soname), if any
    if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) {
    g_err = std::string("GenEO: cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
    return false;
  }
  Api a;
  a.handle = h;
#define SYM(field, name)                                                        \
  *(void**)(&a.field) = dlsym(h, name);                                         \
  if (!a.field) { g_err = std::string("GenEO: RCCL lacks ") + name; return false; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(AllReduce, "ncclAllReduce")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
  g_api = a;
  return true;
}

bool ok(ncclResult_t r, const char* what) {
  if (r == 0) return true;
  g_err = std::string("GenEO: RCCL ") + what + " failed: " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
  return false;
}

}  // namespace

// one halo plan (counts, offsets, device buffers) per PC attached to the communicator
struct GeneoRcclPlan {
  struct _p_GeneoRccl* comm = nullptr;
  std::vector<int> send_counts, recv_counts, soff, roff;   // per peer, in vectors of ONE double per entry
  double *send = nullptr, *recv = nullptr, *red = nullptr;
  int red_cap = 0, width = 1;
};
struct _p_GeneoRccl {
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
  std::vector<GeneoRcclPlan*> plans;
};

// exchange callback of PCGenEOSetComm: flag = reverse | width << 1.  Forward: owners send the values of their DOFs to
// every rank that overlaps them (send_counts out, recv_counts in); reverse: halo contributions travel back (the two
// count arrays swap roles).  Entry-major buffers: the `width` values of one entry are contiguous.
static int rccl_exchange(void* user, int flag) {
  GeneoRcclPlan* p = (GeneoRcclPlan*)user;
  const int reverse = flag & 1;
  const size_t w = (size_t)((flag >> 1) > 0 ? (flag >> 1) : 1);
  if ((int)w > p->width) { g_err = "GenEO: halo exchange wider than the RCCL buffers"; return 1; }
  const std::vector<int>& outc = reverse ? p->recv_counts : p->send_counts;
  const std::vector<int>& outo = reverse ? p->roff : p->soff;
  const std::vector<int>& inc = reverse ? p->send_counts : p->recv_counts;
  const std::vector<int>& ino = reverse ? p->soff : p->roff;
  void* stream = bk::get_stream();
  if (!ok(g_api.GroupStart(), "ncclGroupStart")) return 1;
  bool good = true;
  for (int q = 0; q < p->comm->size && good; ++q) {
    if (outc[q] > 0)
      good = ok(g_api.Send(p->send + (size_t)outo[q] * w, (size_t)outc[q] * w, kNcclDouble, q, p->comm->comm, stream), "ncclSend");
    if (good && inc[q] > 0)
      good = ok(g_api.Recv(p->recv + (size_t)ino[q] * w, (size_t)inc[q] * w, kNcclDouble, q, p->comm->comm, stream), "ncclRecv");
  }
  const bool ended = ok(g_api.GroupEnd(), "ncclGroupEnd");
  return (good && ended) ? 0 : 1;
}

// in-place sum of red[0..n) over the ranks
static int rccl_allreduce(void* user, int n) {
  GeneoRcclPlan* p = (GeneoRcclPlan*)user;
  if (n > p->red_cap) { g_err = "GenEO: all-reduce longer than the RCCL reduction buffer"; return 1; }
  return ok(g_api.AllReduce(p->red, p->red, (size_t)n, kNcclDouble, kNcclSum, p->comm->comm, bk::get_stream()), "ncclAllReduce") ? 0 : 1;
}

extern "C" {

const char* GeneoRcclGetError(void) { return g_err.c_str(); }

PetscErrorCode GeneoRcclUniqueId(char* id128) {
  if (!id128 || !load_api()) return 1;
  ncclUniqueId id;
  if (!ok(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return 1;
  std::memcpy(id128, id.internal, 128);
  return 0;
}

PetscErrorCode GeneoRcclCreate(const char* id128, int rank, int size, GeneoRccl* out) {
  if (!id128 || !out || size < 1 || rank < 0 || rank >= size) { g_err = "GenEO: bad RCCL communicator arguments"; return 1; }
  if (!load_api()) return 1;
  ncclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  _p_GeneoRccl* c = new _p_GeneoRccl();
  c->rank = rank;
  c->size = size;
  if (!ok(g_api.CommInitRank(&c->comm, size, id, rank), "ncclCommInitRank")) {
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

PetscErrorCode PCGenEOSetCommRccl(GENEO_PC pc, GeneoRccl comm, int n_owned, const int* owned_gid, int n_halo,
                                  const int* halo_gid, const int* recv_counts, const int* send_counts,
                                  const int* send_idx, int max_width) {
  if (!pc || !comm || !recv_counts || !send_counts || max_width < 1) { g_err = "GenEO: bad RCCL halo plan"; return 1; }
  GeneoRcclPlan* p = new GeneoRcclPlan();
  p->comm = comm;
  p->width = max_width;
  p->send_counts.assign(send_counts, send_counts + comm->size);
  p->recv_counts.assign(recv_counts, recv_counts + comm->size);
  p->soff.assign(comm->size + 1, 0);
  p->roff.assign(comm->size + 1, 0);
  for (int q = 0; q < comm->size; ++q) {
    p->soff[q + 1] = p->soff[q] + send_counts[q];
    p->roff[q + 1] = p->roff[q] + recv_counts[q];
  }
  const size_t cap = (size_t)std::max(1, std::max(p->soff[comm->size], p->roff[comm->size])) * (size_t)max_width;
  p->red_cap = 1 << 16;
  try {
    p->send = (double*)bk::alloc(sizeof(double) * cap);
    p->recv = (double*)bk::alloc(sizeof(double) * cap);
    p->red = (double*)bk::alloc(sizeof(double) * (size_t)p->red_cap);
  } catch (std::exception& e) {
    g_err = e.what();
    bk::dfree(p->send); bk::dfree(p->recv); bk::dfree(p->red);
    delete p;
    return 1;
  }
  comm->plans.push_back(p);
  PetscErrorCode rc = PCGenEOSetComm(pc, comm->rank, comm->size, n_owned, owned_gid, n_halo, halo_gid, recv_counts,
                                     send_counts, send_idx, rccl_exchange, rccl_allreduce, p, p->send, p->recv, p->red,
                                     p->red_cap);
  if (!rc) rc = PCGenEOSetCommWidth(pc, max_width);
  if (rc) g_err = PCGenEOGetError(pc);
  return rc;
}

// Test / bring-up hook: runs the two callbacks of plan `which` directly (the library calls them from inside the solver).
// buffers: 0 send, 1 recv, 2 reduction -- device pointers of the plan, for the caller to fill and read back.
PetscErrorCode GeneoRcclPlanBuffers(GeneoRccl comm, int which, double** send_dev, double** recv_dev, double** red_dev) {
  if (!comm || which < 0 || which >= (int)comm->plans.size()) return 1;
  GeneoRcclPlan* p = comm->plans[which];
  if (send_dev) *send_dev = p->send;
  if (recv_dev) *recv_dev = p->recv;
  if (red_dev) *red_dev = p->red;
  return 0;
}
PetscErrorCode GeneoRcclPlanExchange(GeneoRccl comm, int which, int flag) {
  if (!comm || which < 0 || which >= (int)comm->plans.size()) return 1;
  return rccl_exchange(comm->plans[which], flag);
}
PetscErrorCode GeneoRcclPlanAllreduce(GeneoRccl comm, int which, int n) {
  if (!comm || which < 0 || which >= (int)comm->plans.size()) return 1;
  return rccl_allreduce(comm->plans[which], n);
}

PetscErrorCode GeneoRcclDestroy(GeneoRccl* comm) {
  if (!comm || !*comm) return 0;
  _p_GeneoRccl* c = *comm;
  for (GeneoRcclPlan* p : c->plans) {
    bk::dfree(p->send); bk::dfree(p->recv); bk::dfree(p->red);
    delete p;
  }
  if (c->comm && g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
  delete c;
  *comm = nullptr;
  return 0;
}

}  // extern "C"
