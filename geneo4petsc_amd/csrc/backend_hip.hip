// HIP / CDNA4 (gfx950, MI355X) implementation of the device primitives in backend.h.
// Wave = 64 lanes, 256-thread workgroups, FP64 data, 32-bit indices.
//
// Kernel inventory (roofline that bounds each, algorithmic bytes per unit -- see DESIGN.md):
//   k_spmv_lds        CSR SpMV, LDS-staged row blocks, XCD-aware block remap       HBM  (12 B/nnz + 20 B/row)
//   k_spmv_long       one workgroup per long row (> tile)                           HBM
//   k_spmm            CSR x row-major multi-vector (fused D pre/post scaling)       HBM  (12 B/nnz + 16 m B/row)
//   k_gram_mfma       tall-skinny S^T T, v_mfma_f64_16x16x4_f64, LDS slabs          MFMA/HBM ridge
//   k_blockmul_mfma   tall-skinny S C update, v_mfma_f64_16x16x4_f64                MFMA/HBM ridge
//   k_gram_fma / k_blockmul_fma   plain-FMA twins (GENEO_NO_MFMA=1; used to validate the MFMA maps)
//   gather / segsum / BLAS-1 / chunked CG / z(t)_apply                              HBM
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <string>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "backend.h"

#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e_) + " at " + \
                               __FILE__ + ":" + std::to_string(__LINE__));                 \
    }                                                                                      \
  } while (0)

namespace bk {

// The library stream.  One MAIN stream (set by the host: GeneoSetStream) carries the solver; a host thread may switch
// ITS launches, copies and allocations to a private side stream (side_stream_begin / side_stream_end): the level-1
// hierarchy is built that way while the main stream runs the eigensolve.  Every `g_stream` below is the calling
// thread's current stream.
static hipStream_t g_stream_main = nullptr;
static thread_local hipStream_t t_stream_side = nullptr;
static thread_local bool t_side = false;
static thread_local bool g_capturing = false;   // between graph_capture_begin and graph_capture_end (per thread)
static thread_local hipStream_t g_capture_stream = nullptr;   // launches of a capturing thread are recorded here
// The HIP "current device" is a property of the HOST THREAD and a new thread starts on device 0: on a node with several
// GPUs visible to the process (one rank per GPU under torchrun, LOCAL_RANK = 1 .. 7) a thread started by the library --
// the level-1 set-up on its side stream, the upload / download helpers -- would create its stream, allocate and launch on
// GPU 0.  The library therefore remembers the device of the thread that configured it (GeneoSetDevice, GeneoSetStream, or
// the first allocation) and every other thread is bound to that device before its first HIP call: bind_thread() sits in
// stream_cur(), which every launch and copy goes through, and in the calls that take no stream.
static std::atomic<int> g_device{-1};
static thread_local bool t_dev_bound = false;
static inline void bind_thread() {
  if (t_dev_bound) return;
  t_dev_bound = true;
  const int want = g_device.load(std::memory_order_acquire);
  if (want < 0) {                 // first HIP call of the library: this thread's device is the library's
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess) g_device.store(d, std::memory_order_release);
    else (void)hipGetLastError();
  } else {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d != want) {
      if (hipSetDevice(want) != hipSuccess) throw std::runtime_error("geneo: hipSetDevice failed on a library thread");
    }
  }
}
static inline hipStream_t stream_cur() {
  bind_thread();
  return g_capturing ? g_capture_stream : (t_side ? t_stream_side : g_stream_main);
}
#define g_stream (stream_cur())
static bool g_no_mfma = false;
static bool g_spgemm_fill_scan = getenv("GENEO_SPGEMM_SCAN_FILL") != nullptr;   // the owner-computes numeric pass of round 2
static bool g_gram_flat = getenv("GENEO_GRAM_NO_FLAT") == nullptr;               // 16-byte streaming form of LOBPCG's Grams
static bool g_init = false;
// set_variant("lp_fixed", 0) / GENEO_LP_FIXED=0: the 4-step loop for every slice of k_spmv_sell_lp instead of its two-latency
// forms (validation, A/B)
__device__ int g_lp_no_fixed = 0;

static void lazy_init() {
  if (g_init) return;
  g_init = true;
  const char* e = getenv("GENEO_NO_MFMA");
  g_no_mfma = (e && e[0] == '1');
  if (const char* f = getenv("GENEO_LP_FIXED")) {
    const int off = atoi(f) ? 0 : 1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lp_no_fixed), &off, sizeof(int));
  }
}

const char* name() { return "hip-gfx950"; }
void set_stream(void* s) {
  // a new main stream: everything queued on the old one is finished first (cached blocks and scratch buffers carry no
  // ordering between two main streams otherwise)
  if ((hipStream_t)s != g_stream_main) (void)hipStreamSynchronize(g_stream_main);
  g_stream_main = (hipStream_t)s;
  // the host hands over a stream of ITS current device: that is the device every library thread works on
  int d = 0;
  if (hipGetDevice(&d) == hipSuccess) g_device.store(d, std::memory_order_release);
  else (void)hipGetLastError();
  t_dev_bound = true;
}
void side_stream_begin(void* after, bool use_after) {
  if (t_side) return;
  bind_thread();
  if (!t_stream_side) HIPCHK(hipStreamCreateWithFlags(&t_stream_side, hipStreamNonBlocking));
  // ordered behind everything the parent stream (default: the main stream) has been given so far -- the matrices this
  // thread is going to read
  hipEvent_t ev;
  HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(ev, use_after ? (hipStream_t)after : g_stream_main));
  HIPCHK(hipStreamWaitEvent(t_stream_side, ev, 0));
  (void)hipEventDestroy(ev);
  t_side = true;
}
void side_stream_end() {
  if (!t_side) return;
  (void)hipStreamSynchronize(t_stream_side);     // whoever joins this thread may use its results on any stream
  t_side = false;
  (void)hipStreamDestroy(t_stream_side);
  t_stream_side = nullptr;
}
void* get_stream() { return (void*)g_stream; }
void sync() { HIPCHK(hipStreamSynchronize(g_stream)); }

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
int set_device(int ordinal) {
  const int n = device_count();
  if (n <= 0 || ordinal < 0) return -1;
  const int d = ordinal % n;
  if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); return -1; }
  g_device.store(d, std::memory_order_release);
  t_dev_bound = true;
  return d;
}
int current_device() { return g_device.load(std::memory_order_acquire); }
// test hook: the device a thread started by the library ends up on (after the binding every HIP path goes through)
int thread_device_check() {
  int got = -2;
  std::thread t([&]() {
    try {
      (void)stream_cur();
      int d = -1;
      if (hipGetDevice(&d) == hipSuccess) got = d;
    } catch (...) {
      got = -3;
    }
  });
  t.join();
  return got;
}
static double g_alloc_s = 0.0, g_free_s = 0.0;
static long long g_alloc_n = 0;
// Caching allocator.  hipMalloc of a multi-GB block costs ~18 ms per GB on this system (184^3 per GPU: 0.9 s of a 2.5 s
// set-up went into hipMalloc, most of it for blocks an earlier phase had just given back) and hipFree synchronises the
// device.  Freed blocks are kept (by size) and handed out again: to the next phase of the same set-up (the LOBPCG basis
// buffers become the blocks of the coarse-operator assembly) and to the next set-up of the process.  Every use of a
// block is ordered on the library stream, so a block may be reused without waiting for its last kernel.  The cache is
// released when a preconditioner is destroyed (alloc_cache_release), when hipMalloc fails, and never grows beyond
// GENEO_ALLOC_CACHE_GB (default 96); GENEO_ALLOC_CACHE=0 turns it off.
static std::mutex g_alloc_mu;
static std::unordered_map<void*, size_t> g_live;
// a parked block remembers the stream its last user ran on and an event recorded there when it was freed: the next
// owner on the SAME stream needs no synchronisation (stream order), one on another stream waits for the event
struct Parked { void* p; hipStream_t stream; hipEvent_t ev; };
static std::multimap<size_t, Parked> g_cache;
static std::vector<hipEvent_t> g_ev_pool;
static size_t g_cache_bytes = 0;
// device-memory accounting of the library's own blocks (bench.py's device_mem_peak_gb): bytes handed out, their
// high-water mark, and the high-water mark of handed out + parked (what the process holds of the card)
static size_t g_live_bytes = 0, g_live_peak = 0, g_foot_peak = 0;
static inline void mem_account(long long delta_live) {      // caller holds g_alloc_mu
  g_live_bytes = (size_t)((long long)g_live_bytes + delta_live);
  g_live_peak = std::max(g_live_peak, g_live_bytes);
  g_foot_peak = std::max(g_foot_peak, g_live_bytes + g_cache_bytes);
}
static bool alloc_cache_on() {
  static const bool on = !(getenv("GENEO_ALLOC_CACHE") && !strcmp(getenv("GENEO_ALLOC_CACHE"), "0"));
  return on;
}
static size_t alloc_cache_cap() {
  static const size_t cap = (size_t)((getenv("GENEO_ALLOC_CACHE_GB") ? atof(getenv("GENEO_ALLOC_CACHE_GB")) : 96.0) * 1073741824.0);
  return cap;
}
static inline size_t alloc_round(size_t bytes) {
  const size_t g = bytes < ((size_t)1 << 20) ? 512 : ((size_t)2 << 20);
  return (bytes + g - 1) / g * g;
}
void alloc_cache_release() {
  bind_thread();
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  if (g_cache.empty()) return;
  auto t0 = std::chrono::high_resolution_clock::now();
  (void)hipDeviceSynchronize();                  // parked blocks may have been used on any stream
  for (auto& kv : g_cache) {
    (void)hipFree(kv.second.p);
    if (kv.second.ev) g_ev_pool.push_back(kv.second.ev);
  }
  g_cache.clear();
  g_cache_bytes = 0;
  g_free_s += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
}
void* alloc(size_t bytes) {
  lazy_init();
  bind_thread();
  void* p = nullptr;
  if (bytes == 0) bytes = 8;
  const size_t sz = alloc_cache_on() ? alloc_round(bytes) : bytes;
  if (alloc_cache_on()) {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_cache.lower_bound(sz);
    if (it != g_cache.end() && it->first <= sz + sz / 8 + 4096) {
      p = it->second.p;
      if (it->second.ev) {
        if (it->second.stream != g_stream) (void)hipStreamWaitEvent(g_stream, it->second.ev, 0);
        g_ev_pool.push_back(it->second.ev);
      }
      g_live[p] = it->first;
      g_cache_bytes -= it->first;
      mem_account((long long)it->first);
      g_cache.erase(it);
    }
  }
  if (!p) {
    auto t0 = std::chrono::high_resolution_clock::now();
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess && alloc_cache_on()) {     // out of memory with blocks parked in the cache: give them back, once
      (void)hipGetLastError();
      alloc_cache_release();
      e = hipMalloc(&p, sz);
    }
    HIPCHK(e);
    const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    static const bool trace = getenv("GENEO_ALLOC_TRACE") != nullptr;
    if (trace && dt > 2e-4) fprintf(stderr, "[alloc] %.1f MB in %.3f ms\n", sz / 1e6, dt * 1e3);
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_alloc_s += dt;
    ++g_alloc_n;
    g_live[p] = sz;
    mem_account((long long)sz);
  }
  HIPCHK(hipMemsetAsync(p, 0, bytes, g_stream));
  return p;
}
void dfree(void* p) {
  if (!p) return;
  bind_thread();
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) {
      const size_t sz = it->second;
      g_live.erase(it);
      mem_account(-(long long)sz);
      if (alloc_cache_on() && g_cache_bytes + sz <= alloc_cache_cap()) {
        Parked pk{p, g_stream, nullptr};
        if (!g_capturing) {
          if (!g_ev_pool.empty()) { pk.ev = g_ev_pool.back(); g_ev_pool.pop_back(); }
          else if (hipEventCreateWithFlags(&pk.ev, hipEventDisableTiming) != hipSuccess) pk.ev = nullptr;
          if (pk.ev && hipEventRecord(pk.ev, g_stream) != hipSuccess) { g_ev_pool.push_back(pk.ev); pk.ev = nullptr; }
        }
        g_cache.emplace(sz, pk);
        g_cache_bytes += sz;
        return;
      }
    }
  }
  auto t0 = std::chrono::high_resolution_clock::now();
  (void)hipFree(p);
  g_free_s += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
}
void mem_info(double* live, double* live_peak, double* footprint_peak, double* cached, double* dev_free, double* dev_total,
              bool reset_peaks) {
  bind_thread();
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); f = t = 0; }
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  if (live) *live = (double)g_live_bytes;
  if (live_peak) *live_peak = (double)g_live_peak;
  if (footprint_peak) *footprint_peak = (double)g_foot_peak;
  if (cached) *cached = (double)g_cache_bytes;
  if (dev_free) *dev_free = (double)f;
  if (dev_total) *dev_total = (double)t;
  if (reset_peaks) {
    g_live_peak = g_live_bytes;
    g_foot_peak = g_live_bytes + g_cache_bytes;
  }
}
void alloc_stats(double* alloc_s, double* free_s, long long* n) {
  *alloc_s = g_alloc_s; *free_s = g_free_s; *n = g_alloc_n;
  g_alloc_s = g_free_s = 0.0; g_alloc_n = 0;
}
// Large transfers from / to pageable host memory (CSR arrays of the fine matrices, downloaded coarse operators)
// go through two pinned staging buffers: the host memcpy of one chunk overlaps the DMA of the other.  Small
// transfers take the plain path (their cost is the synchronisation, not the bandwidth).
constexpr size_t STAGE_BYTES = (size_t)16 << 20;
// Two staging sets (two pinned buffers + events each): a thread takes a free set for the duration of one large copy, so
// the main thread and a side-stream thread (section "side streams") stage their uploads side by side -- the host memcpy
// into pinned memory, ~10 GB/s per thread, is what bounds a pageable upload, not the link.
struct StageSet {
  char* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2];
  bool ready = false, busy = false, failed = false;
};
static StageSet g_stage_sets[2];
static std::mutex g_stage_mu;
static std::condition_variable g_stage_cv;
static StageSet* stage_acquire() {
  if (getenv("GENEO_NO_PINNED_STAGING")) return nullptr;
  bind_thread();
  std::unique_lock<std::mutex> lk(g_stage_mu);
  for (;;) {
    bool any_usable = false;
    for (StageSet& st : g_stage_sets) {
      if (st.failed) continue;
      any_usable = true;
      if (st.busy) continue;
      if (!st.ready) {
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) {
          if (hipHostMalloc((void**)&st.buf[i], STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (i == 1) (void)hipHostFree(st.buf[0]);
            st.buf[0] = st.buf[1] = nullptr;
            ok = false;
          } else if (hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming) != hipSuccess) {
            ok = false;
          }
        }
        if (!ok) { st.failed = true; continue; }
        st.ready = true;
      }
      st.busy = true;
      return &st;
    }
    if (!any_usable) return nullptr;
    g_stage_cv.wait(lk);
  }
}
static void stage_release(StageSet* st) {
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    st->busy = false;
  }
  g_stage_cv.notify_one();
}
struct StageGuard {
  StageSet* st;
  explicit StageGuard(StageSet* s) : st(s) {}
  ~StageGuard() { if (st) stage_release(st); }
};
// memcpy between pageable and pinned memory on four host threads: one thread moves ~10 GB/s, which is what bounded the
// 0.5 GB upload of a 6.5 M-row matrix (0.05-0.1 s), not the link
static void par_memcpy(void* dst, const void* src, size_t len) {
  constexpr int NT = 4;
  if (len < ((size_t)4 << 20)) { std::memcpy(dst, src, len); return; }
  std::thread th[NT - 1];
  const size_t part = (len / NT + 63) & ~(size_t)63;
  for (int t = 1; t < NT; ++t) {
    const size_t off = std::min(len, part * t), end = std::min(len, part * (t + 1));
    th[t - 1] = std::thread([=]() { std::memcpy((char*)dst + off, (const char*)src + off, end - off); });
  }
  std::memcpy(dst, src, std::min(len, part));
  for (auto& x : th) x.join();
}
void h2d(void* d, const void* h, size_t bytes) {
  if (!bytes) return;
  if (bytes >= 2 * STAGE_BYTES && !g_capturing) {
    StageGuard g(stage_acquire());
    if (g.st) {
      size_t off = 0;
      for (int i = 0; off < bytes; ++i, off += STAGE_BYTES) {
        const int b = i & 1;
        const size_t len = std::min(STAGE_BYTES, bytes - off);
        if (i >= 2) HIPCHK(hipEventSynchronize(g.st->ev[b]));      // the DMA that last used this buffer is done
        par_memcpy(g.st->buf[b], (const char*)h + off, len);
        HIPCHK(hipMemcpyAsync((char*)d + off, g.st->buf[b], len, hipMemcpyHostToDevice, g_stream));
        HIPCHK(hipEventRecord(g.st->ev[b], g_stream));
      }
      HIPCHK(hipStreamSynchronize(g_stream));
      return;
    }
  }
  HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));  // h may be pageable / reused by the caller
}
void d2h(void* h, const void* d, size_t bytes) {
  if (!bytes) return;
  if (bytes >= 2 * STAGE_BYTES && !g_capturing) {
    StageGuard g(stage_acquire());
    if (g.st) {
      const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
      auto issue = [&](size_t i) {
        const size_t off = i * STAGE_BYTES, len = std::min(STAGE_BYTES, bytes - off);
        HIPCHK(hipMemcpyAsync(g.st->buf[i & 1], (const char*)d + off, len, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipEventRecord(g.st->ev[i & 1], g_stream));
      };
      issue(0);
      for (size_t i = 0; i < nchunk; ++i) {
        if (i + 1 < nchunk) issue(i + 1);                             // next chunk's DMA runs during this memcpy
        HIPCHK(hipEventSynchronize(g.st->ev[i & 1]));
        const size_t off = i * STAGE_BYTES, len = std::min(STAGE_BYTES, bytes - off);
        par_memcpy((char*)h + off, g.st->buf[i & 1], len);
      }
      return;
    }
  }
  HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
}
void* pinned_alloc(size_t bytes) {
  bind_thread();
  void* p = nullptr;
  HIPCHK(hipHostMalloc(&p, std::max<size_t>(bytes, 8), hipHostMallocDefault));
  return p;
}
void pinned_free(void* p) {
  if (p) (void)hipHostFree(p);
}
void h2d_async(void* d, const void* h_pinned, size_t bytes) {
  if (!bytes) return;
  HIPCHK(hipMemcpyAsync(d, h_pinned, bytes, hipMemcpyHostToDevice, g_stream));
}
void d2d(void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
}
void zero(void* d, size_t bytes) {
  if (!bytes) return;
  HIPCHK(hipMemsetAsync(d, 0, bytes, g_stream));
}

// ---- HIP graphs: capture on a private stream, replay on the launch stream ----------------------------
// (the in-situ kernel timer counts what a replay launches: recorded per graph at capture, added at every replay)
static void graph_counts_reset();
static void graph_counts_store(void* exec);
static void graph_counts_add(void* exec);
static void graph_counts_drop(void* exec);
bool graph_capture_begin() {
  if (g_capturing || getenv("GENEO_NO_GRAPH")) return false;
  if (!g_capture_stream && hipStreamCreateWithFlags(&g_capture_stream, hipStreamNonBlocking) != hipSuccess) {
    g_capture_stream = nullptr;
    return false;
  }
  HIPCHK(hipStreamSynchronize(g_stream));
  if (hipStreamBeginCapture(g_capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  g_capturing = true;
  graph_counts_reset();
  return true;
}
void* graph_capture_end() {
  if (!g_capturing) return nullptr;
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(g_capture_stream, &graph);
  g_capturing = false;
  if (e != hipSuccess || !graph) return nullptr;
  hipGraphExec_t exec = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) return nullptr;
  graph_counts_store((void*)exec);
  return (void*)exec;
}
void graph_launch(void* exec) {
  HIPCHK(hipGraphLaunch((hipGraphExec_t)exec, g_stream));
  graph_counts_add(exec);
}
void graph_destroy(void* exec) {
  if (!exec) return;
  graph_counts_drop(exec);
  (void)hipGraphExecDestroy((hipGraphExec_t)exec);
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int grid1d(int64_t n, int per_block) {
  int g = cdiv(n, per_block);
  return g < 1 ? 1 : g;
}

static inline int gridv(int64_t n) { return std::min(grid1d(n, 1024), 2048); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum_256(double v, double* sm /*>=4*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];  // every thread gets the same, fixed-order value
}
// =============================================================================== CSR SpMV
constexpr int SPMV_TILE = 1792;  // nnz staged in LDS per workgroup (14 KB): 256 rows x 7 nnz
constexpr int SPMV_ROWS = 256;   // rows per row block (one row per thread in the reduce phase)
constexpr int SELL_LONG = 64;    // rows longer than this bypass the sliced layout
constexpr int SELL_VEC_MAX_ROWS = 1 << 18;   // ragged matrices below this size run the lanes-per-row kernel instead

// device-side construction of the sliced layout from the CSR arrays already in HBM
__global__ void k_slice_width(const int* __restrict__ rowptr, int n, int ns, int* __restrict__ w, int long_len) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  int m = 0;
  const int r1 = (64 * s + 64 < n) ? 64 * s + 64 : n;
  for (int i = 64 * s; i < r1; ++i) {
    const int len = rowptr[i + 1] - rowptr[i];
    if (len <= long_len && len > m) m = len;
  }
  w[s] = m;
}
__global__ __launch_bounds__(256) void k_sell_fill(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const double* __restrict__ val, int n, int ns,
                                                   const int64_t* __restrict__ sl_ptr, int* __restrict__ sl_col,
                                                   double* __restrict__ sl_val, int long_len) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int l = threadIdx.x & 63;
  const int i = 64 * s + l;
  const int64_t base = sl_ptr[s];
  const int w = (int)((sl_ptr[s + 1] - base) >> 6);
  int a = 0, len = 0;
  if (i < n) {
    a = rowptr[i];
    len = rowptr[i + 1] - a;
  }
  const bool use = (i < n) && len <= long_len;
  // padding entries (value 0) point at a column NEAR the slice's own -- the row's LAST one (its first one for a long row
  // that is not stored here), or for an empty row / the rows past the end of the last slice the entry just before: the
  // x gathers stay local and the slice's column span (16-bit offsets of the single-precision companion) is not widened
  // by the padding.  The last column rather than the first: the padding sits behind the row's entries, and a slice that
  // needs two column bases (k_lp_base) splits its entries by position -- low columns first, high columns last.
  int padc = 0;
  if (i < n && len > 0) padc = use ? col[a + len - 1] : col[a];
  else if (n > 0 && rowptr[n] > 0) {
    const int k = (i < n) ? a : rowptr[n];
    padc = col[k > 0 ? k - 1 : 0];
  }
  for (int k = 0; k < w; ++k) {
    const int64_t e = base + (int64_t)64 * k + l;
    if (use && k < len) {
      sl_col[e] = col[a + k];
      sl_val[e] = val[a + k];
    } else {
      sl_col[e] = padc;
      sl_val[e] = 0.0;
    }
  }
}

int spmv_kind();
static bool g_force_slices = false;   // tests: no lanes-per-row kernel for small ragged matrices (they exercise the wide-slice kernels)
static void finish_layout(Csr& a, const int* h_rowptr, const int* h_col = nullptr);

Csr csr_upload_raw(int n, const int* h_rowptr, const int* h_col, const double* h_val) {
  Csr a;
  a.n = n;
  a.nnz = h_rowptr[n];
  a.rowptr = (int*)alloc(sizeof(int) * (size_t)(n + 1));
  a.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)a.nnz));
  a.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  h2d(a.rowptr, h_rowptr, sizeof(int) * (size_t)(n + 1));
  h2d(a.col, h_col, sizeof(int) * (size_t)a.nnz);
  h2d(a.val, h_val, sizeof(double) * (size_t)a.nnz);
  return a;
}
__global__ void k_iota_ones(int* __restrict__ rp, double* __restrict__ val, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
    rp[i] = i;
    if (i < n) val[i] = 1.0;
  }
}
Csr csr_tentative_prolongator(int n, const int* agg_dev) {
  Csr a;
  a.n = n;
  a.nnz = n;
  a.rowptr = (int*)alloc(sizeof(int) * (size_t)(n + 1));
  a.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)n));
  a.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)n));
  hipLaunchKernelGGL(k_iota_ones, dim3(grid1d(n + 1, 256)), dim3(256), 0, stream_cur(), a.rowptr, a.val, n);
  if (n > 0) d2d(a.col, agg_dev, sizeof(int) * (size_t)n);
  return a;
}
Csr csr_upload(int n, const int* h_rowptr, const int* h_col, const double* h_val) {
  Csr a;
  a.n = n;
  a.nnz = h_rowptr[n];
  a.rowptr = (int*)alloc(sizeof(int) * (size_t)(n + 1));
  a.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)a.nnz));
  a.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  static const bool dbg = getenv("GENEO_DEBUG") != nullptr;
  auto t0 = std::chrono::high_resolution_clock::now();
  h2d(a.rowptr, h_rowptr, sizeof(int) * (size_t)(n + 1));
  h2d(a.col, h_col, sizeof(int) * (size_t)a.nnz);
  h2d(a.val, h_val, sizeof(double) * (size_t)a.nnz);
  auto t1 = std::chrono::high_resolution_clock::now();
  finish_layout(a, h_rowptr, h_col);
  if (dbg && a.nnz > 4000000) {
    HIPCHK(hipStreamSynchronize(g_stream));
    auto t2 = std::chrono::high_resolution_clock::now();
    fprintf(stderr, "[upload] n %d nnz %lld: copies %.4f s (%.1f GB/s), layouts %.4f s\n", n, (long long)a.nnz,
            std::chrono::duration<double>(t1 - t0).count(),
            (12.0 * a.nnz + 4.0 * n) / std::chrono::duration<double>(t1 - t0).count() / 1e9,
            std::chrono::duration<double>(t2 - t1).count());
  }
  return a;
}
// SpMV layouts (LDS row blocks, 64-row slices, long-row list, lanes-per-row choice) of a matrix whose CSR arrays are
// in HBM; h_rowptr = host copy of the row pointers
// Slice schedule of the sliced SpMM for matrices whose far neighbours lie further apart than the slices one XCD group
// holds in flight (a 3-D grid in natural order: the +-z neighbours are a whole plane away -- 186^2 rows = 8.9 MB of a
// 32-column block, against the 4 MiB L2 of an XCD).  The slices are re-ordered into compact blobs of the slice graph
// (breadth-first growth from the lowest unvisited slice, TILE slices per blob): the waves of a group then work side by
// side on one blob, so that a row of X fetched for one slice is still in the group's L2 when the neighbouring slices of
// the same blob ask for it; only the blob's surface is fetched twice.  Adjacency is sampled (4 rows per slice): the
// schedule is a traversal order, not part of the result.  Returns an empty vector when the natural order already keeps
// the neighbours within reach.
// Strip schedule for matrices with ONE far offset F per block (a 3-D stencil in natural order: F = a grid plane).  The
// blobs above bound the over-fetch of the +-F neighbours by their surface (1.56 x algorithmic at 6.5 M rows per block);
// a sweep does better: the plane is cut into strips of W consecutive rows (position of a row in its plane = row mod F),
// and a strip is walked through ALL planes before the next strip starts.  The +-F neighbours of the rows in flight are
// then the same strip one plane behind (already in the group's L2: it was the centre a moment ago) and one plane ahead
// (fetched now, the centre next): every row of X comes in once, plus the two lines either side of a strip (+-nx
// neighbours across the strip edge).  Needs nothing but F: slices are grouped into runs of equal furthest offset
// (blocks of a flat block-diagonal matrix may differ), the origin of a run stands for the block's first row (a shift
// of the origin only rotates the strips).  Empty when less than 80 % of the slices lie in runs of three planes or more
// (irregular matrices: the blobs stay).
static std::vector<int> strip_schedule(int nslice, const std::vector<int>& frow) {
  // rows of a strip (target); read per matrix, as the tile below: tests lower both to reach the schedules on small grids
  const int WT = std::max(64, getenv("GENEO_SPMM_STRIP") ? atoi(getenv("GENEO_SPMM_STRIP")) : 3072);
  std::vector<int> order;
  order.reserve(nslice);
  int64_t covered = 0;
  std::vector<std::vector<int>> bucket;
  for (int sa = 0; sa < nslice;) {
    int sb = sa + 1;
    while (sb < nslice && frow[sb] == frow[sa]) ++sb;
    const int F = frow[sa];
    const bool swept = F >= 4 * WT && (int64_t)(sb - sa) * 64 >= (int64_t)3 * F;
    if (!swept) {
      for (int s = sa; s < sb; ++s) order.push_back(s);
    } else {
      const int nst = std::max(1, (F + WT / 2) / WT);
      const int W = (F + nst - 1) / nst;
      bucket.assign(nst, {});
      for (int s = sa; s < sb; ++s) bucket[(int)((((int64_t)(s - sa) * 64) % F) / W)].push_back(s);
      for (int b = 0; b < nst; ++b) order.insert(order.end(), bucket[b].begin(), bucket[b].end());
      covered += sb - sa;
    }
    sa = sb;
  }
  if (covered * 10 < (int64_t)nslice * 8) return {};
  if (getenv("GENEO_DEBUG_SCHED"))
    fprintf(stderr, "[sched] strip: %d slices, %lld swept, far offset of the first run %d rows, strip target %d rows\n", nslice,
            (long long)covered, frow[0], WT);
  return order;
}

static std::vector<int> blob_schedule(int n, const int* h_rowptr, const int* h_col, int nslice) {
  const int TILE = std::max(4, getenv("GENEO_SPMM_TILE") ? atoi(getenv("GENEO_SPMM_TILE")) : 512);
  const char* mode = getenv("GENEO_SPMM_SCHED");   // natural | blob | (auto: strips, else blobs, else natural)
  if (mode && !strcmp(mode, "natural")) return {};
  if (nslice < 4 * TILE) return {};
  // Sampling: 4 rows per slice, two dependent cache misses each (row pointer, columns) -- latency-bound on one host thread
  // (25-45 ms for the 102 000 slices of a 6.5 M-row block), so the slices are sampled in ranges on up to 8 threads; the
  // per-slice lists are concatenated in slice order, i.e. the result does not depend on the number of threads.
  std::vector<int> adj_ptr((size_t)nslice + 1, 0), adj;
  constexpr int CAP = 60;
  const int nth = std::max(1, std::min(8, nslice / 4096));
  std::vector<std::vector<int>> part_adj(nth);
  std::vector<int> part_far(nth, 0);
  std::vector<int> cnts((size_t)nslice, 0);
  std::vector<int> frow((size_t)nslice, 0);   // furthest column of the probed rows, in rows (a grid plane for a 3-D stencil)
  auto sample = [&](int t) {
    const int s0 = (int)((int64_t)nslice * t / nth), s1 = (int)((int64_t)nslice * (t + 1) / nth);
    std::vector<int>& out = part_adj[t];
    out.reserve((size_t)(s1 - s0) * 8);
    int far_t = 0;
    for (int s = s0; s < s1; ++s) {
      int loc[64];
      int cnt = 0;
      const int r0 = 64 * s, r1 = std::min(n, r0 + 64);
      const int probe[4] = {r0, r0 + 21, r0 + 42, r1 - 1};
      for (int t4 = 0; t4 < 4; ++t4) {
        const int r = std::min(probe[t4], r1 - 1);
        for (int k = h_rowptr[r]; k < h_rowptr[r + 1] && cnt < CAP; ++k) {
          frow[s] = std::max(frow[s], std::abs(h_col[k] - r));
          const int o = h_col[k] >> 6;
          if (o == s || o >= nslice) continue;
          bool seen = false;
          for (int i = 0; i < cnt; ++i) seen = seen || loc[i] == o;
          if (!seen) loc[cnt++] = o;
        }
      }
      for (int i = 0; i < cnt; ++i) {
        out.push_back(loc[i]);
        far_t = std::max(far_t, std::abs(loc[i] - s));
      }
      cnts[s] = cnt;
    }
    part_far[t] = far_t;
  };
  if (nth > 1) {
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(sample, t);
    sample(0);
    for (auto& x : th) x.join();
  } else {
    sample(0);
  }
  int far = 0;
  for (int t = 0; t < nth; ++t) far = std::max(far, part_far[t]);
  for (int s = 0; s < nslice; ++s) adj_ptr[s + 1] = adj_ptr[s] + cnts[s];
  // natural order keeps the far neighbours in flight while they are within a fraction of the slices a group holds
  if (!(mode && !strcmp(mode, "blob")) && far < TILE / 4) return {};
  if (!(mode && !strcmp(mode, "blob"))) {
    std::vector<int> order = strip_schedule(nslice, frow);
    if (!order.empty()) return order;
  }
  adj.reserve((size_t)adj_ptr[nslice]);
  for (int t = 0; t < nth; ++t) adj.insert(adj.end(), part_adj[t].begin(), part_adj[t].end());
  std::vector<int> order;
  order.reserve(nslice);
  std::vector<char> seen((size_t)nslice, 0);
  std::vector<int> queue;
  int next_seed = 0;
  while ((int)order.size() < nslice) {
    while (next_seed < nslice && seen[next_seed]) ++next_seed;
    if (next_seed >= nslice) break;
    queue.clear();
    queue.push_back(next_seed);
    seen[next_seed] = 1;
    size_t head = 0;
    int taken = 0;
    while (head < queue.size() && taken < TILE) {
      const int s = queue[head++];
      order.push_back(s);
      ++taken;
      for (int k = adj_ptr[s]; k < adj_ptr[s + 1]; ++k) {
        const int o = adj[k];
        if (!seen[o]) { seen[o] = 1; queue.push_back(o); }
      }
    }
    for (; head < queue.size(); ++head) seen[queue[head]] = 0;   // discovered but not taken: back to the pool
  }
  return order;
}

static void finish_layout(Csr& a, const int* h_rowptr, const int* h_col) {
  const int n = a.n;
  static const bool dbg_l = getenv("GENEO_DEBUG") != nullptr;
  auto tl0 = std::chrono::high_resolution_clock::now();
  auto lapl = [&](const char* what) {
    if (!dbg_l || a.nnz < 4000000) return;
    HIPCHK(hipStreamSynchronize(g_stream));
    auto t = std::chrono::high_resolution_clock::now();
    fprintf(stderr, "[layout] %-22s %.4f s\n", what, std::chrono::duration<double>(t - tl0).count());
    tl0 = t;
  };
  int maxrow = 0;
  std::vector<int> longr;
  const int long_len = SELL_LONG;
  for (int i = 0; i < n; ++i) {
    const int len = h_rowptr[i + 1] - h_rowptr[i];
    maxrow = std::max(maxrow, len);
    if (len > long_len) longr.push_back(i);
  }
  a.max_row = maxrow;
  lapl("row lengths (host)");
  if (spmv_kind() == 0) {
    // row blocks of the LDS kernel: as many consecutive rows as fit SPMV_TILE nnz and SPMV_ROWS rows;
    // a row longer than the tile gets a block of its own (long-row path)
    std::vector<int> blk;
    blk.push_back(0);
    int r = 0;
    while (r < n) {
      int r1 = r;
      int nz = 0;
      while (r1 < n && (r1 - r) < SPMV_ROWS) {
        const int len = h_rowptr[r1 + 1] - h_rowptr[r1];
        if (nz + len > SPMV_TILE) break;
        nz += len;
        ++r1;
      }
      if (r1 == r) ++r1;  // single long row
      blk.push_back(r1);
      r = r1;
    }
    a.nblk = (int)blk.size() - 1;
    a.rowblk = (int*)alloc(sizeof(int) * blk.size());
    h2d(a.rowblk, blk.data(), sizeof(int) * blk.size());
  }
  // sliced layout, built on the device from the CSR arrays just uploaded
  {
    const int ns = (n + 63) / 64;
    int* dw = (int*)alloc(sizeof(int) * std::max(1, ns));
    if (ns > 0)
      hipLaunchKernelGGL(k_slice_width, dim3(grid1d(ns, 256)), dim3(256), 0, g_stream, a.rowptr, n, ns, dw, long_len);
    std::vector<int> w(std::max(1, ns));
    d2h(w.data(), dw, sizeof(int) * ns);
    dfree(dw);
    std::vector<int64_t> sp(ns + 1, 0);
    for (int s = 0; s < ns; ++s) sp[s + 1] = sp[s] + (int64_t)64 * w[s];
    a.nslice = ns;
    a.sl_nnz = sp[ns];
    a.sl_ptr = (int64_t*)alloc(sizeof(int64_t) * (ns + 1));
    a.sl_col = (int*)alloc(sizeof(int) * (size_t)std::max<int64_t>(1, sp[ns]));
    a.sl_val = (double*)alloc(sizeof(double) * (size_t)std::max<int64_t>(1, sp[ns]));
    h2d(a.sl_ptr, sp.data(), sizeof(int64_t) * (ns + 1));
    lapl("slice widths + scan");
    if (ns > 0)
      hipLaunchKernelGGL(k_sell_fill, dim3((ns + 3) / 4), dim3(256), 0, g_stream, a.rowptr, a.col, a.val, n, ns,
                         a.sl_ptr, a.sl_col, a.sl_val, long_len);
    lapl("slice fill");
    {   // traversal of the sliced SpMM: eight contiguous ranges of the slice schedule, one per XCD group
      int xp[9];
      for (int g = 0; g <= 8; ++g) xp[g] = (int)(((int64_t)ns * g) / 8);
      a.xcd_ptr = (int*)alloc(sizeof(int) * 9);
      h2d(a.xcd_ptr, xp, sizeof(int) * 9);
      if (h_col && longr.empty()) {
        const std::vector<int> order = blob_schedule(n, h_rowptr, h_col, ns);
        if ((int)order.size() == ns) {
          a.sched = (int*)alloc(sizeof(int) * (size_t)ns);
          h2d(a.sched, order.data(), sizeof(int) * (size_t)ns);
        }
      }
    }
    lapl("blob schedule");
    a.nlong = (int)longr.size();
    // Long or ragged rows (restriction operators, coarse Galerkin matrices): one lane group per row reads the
    // row coalesced and reduces with shuffles; the slices would pad every 64-row slice to its longest row.
    const double avg = n > 0 ? (double)a.nnz / n : 0.0;
    static const int vec_max_rows = getenv("GENEO_SELL_VEC_MAX_ROWS") ? atoi(getenv("GENEO_SELL_VEC_MAX_ROWS")) : SELL_VEC_MAX_ROWS;
    if (!g_force_slices && n < vec_max_rows && (a.nlong > 0 || avg >= 20.0)) a.vec_lpr = avg <= 24.0 ? 16 : (avg <= 48.0 ? 32 : 64);
    a.long_rows = (int*)alloc(sizeof(int) * std::max<size_t>(1, longr.size()));
    h2d(a.long_rows, longr.data(), sizeof(int) * longr.size());
  }
}
void csr_finish(Csr& a) {
  std::vector<int> rp((size_t)a.n + 1);
  d2h(rp.data(), a.rowptr, sizeof(int) * rp.size());
  finish_layout(a, rp.data());
}
void csr_download(const Csr& a, int* rowptr, int* col, double* val) {
  d2h(rowptr, a.rowptr, sizeof(int) * ((size_t)a.n + 1));
  d2h(col, a.col, sizeof(int) * (size_t)a.nnz);
  d2h(val, a.val, sizeof(double) * (size_t)a.nnz);
}
__global__ void k_map_int(int* __restrict__ out, const int* __restrict__ in, const int* __restrict__ map, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = map[in[i]];
}
Csr csr_remap_columns(const Csr& a, const int* map_dev) {
  Csr b = a;   // scalars (n, nnz, nslice, nlong, max_row, vec_lpr, ...) carry over; the arrays are re-made below
  auto dup = [](const void* src, size_t bytes) -> void* {
    void* p = alloc(std::max<size_t>(bytes, 8));
    if (src && bytes) d2d(p, src, bytes);
    return p;
  };
  const size_t nnz = (size_t)a.nnz;
  b.rowptr = (int*)dup(a.rowptr, sizeof(int) * ((size_t)a.n + 1));
  b.val = (double*)dup(a.val, sizeof(double) * nnz);
  b.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, nnz));
  if (nnz) hipLaunchKernelGGL(k_map_int, dim3(gridv((int64_t)nnz)), dim3(256), 0, g_stream, b.col, a.col, map_dev, (int64_t)nnz);
  b.rowblk = a.rowblk ? (int*)dup(a.rowblk, sizeof(int) * ((size_t)a.nblk + 1)) : nullptr;
  b.sl_ptr = (int64_t*)dup(a.sl_ptr, sizeof(int64_t) * ((size_t)a.nslice + 1));
  b.sl_val = (double*)dup(a.sl_val, sizeof(double) * (size_t)std::max<int64_t>(1, a.sl_nnz));
  b.sl_col = (int*)alloc(sizeof(int) * (size_t)std::max<int64_t>(1, a.sl_nnz));
  if (a.sl_nnz > 0)
    hipLaunchKernelGGL(k_map_int, dim3(gridv(a.sl_nnz)), dim3(256), 0, g_stream, b.sl_col, a.sl_col, map_dev, a.sl_nnz);
  b.long_rows = (int*)dup(a.long_rows, sizeof(int) * std::max<size_t>(1, (size_t)a.nlong));
  b.sched = a.sched ? (int*)dup(a.sched, sizeof(int) * std::max<size_t>(1, (size_t)a.nslice)) : nullptr;
  b.xcd_ptr = a.xcd_ptr ? (int*)dup(a.xcd_ptr, sizeof(int) * 9) : nullptr;
  return b;
}
// =============================================================================== sparse products (multigrid set-up)
// Row-wise SpGEMM, one wave per output row.  Distinct columns of the row are collected in an LDS hash set
// (atomicCAS on the keys only), compacted, sorted (bitonic) and written; the values are then accumulated
// owner-computes: lane t owns the sorted columns t, t+64, ... and scans ALL products of the row in the fixed
// (k, l) order, so every sum has a fixed order whatever the hash did.  Capacity: SPG_MAXD distinct columns per row.
constexpr int SPG_HS = 512;      // hash slots per wave (load factor <= 0.5 at the row capacity; typical rows hold 10-60 columns)
constexpr int SPG_MAXD = 256;    // distinct columns per output row
constexpr int SPG_KMAX = 256;    // entries of a row of A whose product offsets fit the wave's LDS prefix table
constexpr int SPG_CHUNK = 256;   // products staged per round of the numeric phase
__device__ int g_spgemm_no_small = 0;   // validation: 1 sends the <= 64-product rows through the hash table as well
// Round 3: the products of an output row are FLATTENED over the wave.  Round 2 walked the row of A entry by entry and let
// the lanes stride over the matching row of B -- with the 5-to-30-entry rows of P, A P and R that kept 5 to 30 of 64
// lanes busy in the symbolic pass, and the numeric pass fetched every product's (column, value) from global memory inside
// a serial loop (1 250 products per row of R (A P) at 126^3).  Now a prefix table of the B-row lengths (LDS, one entry per
// entry of the A row) maps product number p to (k, l); lane t takes the products t, t + 64, ...: the hash inserts of the
// symbolic pass run 64 wide, and the numeric pass stages SPG_CHUNK products at a time in LDS (all lanes loading) before
// the owner-computes scan reads them back as LDS broadcasts.  Product order (k-major, l-minor) and therefore every sum
// is unchanged: results are bit-identical to round 2's.  Rows of A longer than SPG_KMAX keep the serial walk.
__device__ __forceinline__ int spg_prefix(const int* __restrict__ arp, const int* __restrict__ acol, const int* __restrict__ brp,
                                          int row, int lane, int* ja, int* pre) {
  const int a0 = arp[row], na = arp[row + 1] - a0;
  if (na > SPG_KMAX) return -1;
  for (int k = lane; k < na; k += 64) {
    const int j = acol[a0 + k];
    ja[k] = j;
    pre[k + 1] = brp[j + 1] - brp[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    pre[0] = 0;
    for (int k = 0; k < na; ++k) pre[k + 1] += pre[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return pre[na];
}
// entry k of the A row that product p belongs to: pre[k] <= p < pre[k + 1]
__device__ __forceinline__ int spg_find(const int* pre, int na, int p) {
  int lo = 0, hi = na;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] <= p) lo = mid;
    else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void spg_insert(int* keys, int key, int* overflow) {
  unsigned h = ((unsigned)key * 2654435761u) & (SPG_HS - 1);
  for (int probe = 0; probe < SPG_HS; ++probe) {
    const int old = atomicCAS(&keys[h], -1, key);
    if (old == -1 || old == key) break;
    h = (h + 1) & (SPG_HS - 1);
    if (probe == SPG_HS - 1) *overflow = 1;
  }
}
// hash set of the distinct columns of output row `row` (total = spg_prefix's result: -1 = long row, serial walk)
__device__ __forceinline__ void spg_collect(const int* __restrict__ arp, const int* __restrict__ acol,
                                            const int* __restrict__ brp, const int* __restrict__ bcol, int row,
                                            int lane, int* keys, const int* ja, const int* pre, int total, int* overflow) {
  for (int s = lane; s < SPG_HS; s += 64) keys[s] = -1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (total >= 0) {
    const int na = arp[row + 1] - arp[row];
    for (int p = lane; p < total; p += 64) {
      const int k = spg_find(pre, na, p);
      spg_insert(keys, bcol[brp[ja[k]] + (p - pre[k])], overflow);
    }
  } else {
    for (int k = arp[row]; k < arp[row + 1]; ++k) {
      const int j = acol[k];
      for (int l = brp[j] + lane; l < brp[j + 1]; l += 64) spg_insert(keys, bcol[l], overflow);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Rows of at most 64 products (A P0 and A P of a stencil matrix: 7 and ~25) never touch the hash table: one product per
// lane, the (column, product number) pairs are sorted ACROSS THE LANES (bitonic network on shuffles, no LDS, no barrier),
// equal columns end up side by side in product order, and the head lane of each run folds its run from the left -- the
// same sum order as the scans, so the results are bit-identical, and the row leaves sorted.  Clearing, compacting and
// sorting a 512-slot table per row was most of what these rows cost (1 M rows of A P0: 2.1 ms, of which 7 products each).
__device__ __forceinline__ void spg_sort64(unsigned long long& ck, double& v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned long long ock = __shfl_xor(ck, j, 64);
      const double ov = __shfl_xor(v, j, 64);
      const bool take_min = (((lane & j) == 0) == ((lane & k) == 0));
      const bool swap = take_min ? (ock < ck) : (ock > ck);
      if (swap) { ck = ock; v = ov; }
    }
}
// one product per lane: composite sort key (column << 6 | product number) and value; lanes without a product sort last
__device__ __forceinline__ void spg_small_products(const int* __restrict__ brp, const int* __restrict__ bcol,
                                                   const double* __restrict__ aval, const double* __restrict__ bval,
                                                   int a0, int na, int total, int lane, const int* ja, const int* pre,
                                                   unsigned long long& ck, double& v) {
  ck = ~0ull;
  v = 0.0;
  if (lane < total) {
    const int k = spg_find(pre, na, lane);
    const int l = brp[ja[k]] + (lane - pre[k]);
    ck = ((unsigned long long)(unsigned)bcol[l] << 6) | (unsigned)lane;
    if (aval) v = aval[a0 + k] * bval[l];
  }
}
__global__ __launch_bounds__(256) void k_spgemm_count(int n, const int* __restrict__ arp, const int* __restrict__ acol,
                                                      const int* __restrict__ brp, const int* __restrict__ bcol,
                                                      int* __restrict__ cnt, int* __restrict__ overflow,
                                                      const unsigned char* __restrict__ defer) {
  __shared__ int keys[4][SPG_HS];
  __shared__ int ja[4][SPG_KMAX];
  __shared__ int pre[4][SPG_KMAX + 1];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= n) return;
  if (defer && !defer[row]) return;       // done by k_spgemm_small
  const int total = spg_prefix(arp, acol, brp, row, lane, ja[w], pre[w]);
  if (total >= 0 && total <= 64 && !g_spgemm_no_small) {
    unsigned long long ck;
    double v;
    spg_small_products(brp, bcol, nullptr, nullptr, arp[row], arp[row + 1] - arp[row], total, lane, ja[w], pre[w], ck, v);
    spg_sort64(ck, v, lane);
    const unsigned long long prev = __shfl_up(ck, 1, 64);
    const bool head = (ck != ~0ull) && (lane == 0 || (prev >> 6) != (ck >> 6));
    const unsigned long long heads = __ballot(head);
    if (lane == 0) cnt[row] = __builtin_popcountll(heads);
    return;
  }
  spg_collect(arp, acol, brp, bcol, row, lane, keys[w], ja[w], pre[w], total, overflow);
  int c = 0;
  for (int s = lane; s < SPG_HS; s += 64) c += (keys[w][s] != -1) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) {
    cnt[row] = c;
    if (c > SPG_MAXD) *overflow = 1;
  }
}
__device__ __forceinline__ void wave_bitonic_sort(int* key, double* val, int N, int lane) {   // N power of two
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < N; i += 64) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = ((i & k) == 0);
          const int a = key[i], b = key[ixj];
          if ((a > b) == up) {
            key[i] = b; key[ixj] = a;
            if (val) { const double t = val[i]; val[i] = val[ixj]; val[ixj] = t; }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(256) void k_spgemm_fill_scan(int n, const int* __restrict__ arp, const int* __restrict__ acol,
                                                     const double* __restrict__ aval, const int* __restrict__ brp,
                                                     const int* __restrict__ bcol, const double* __restrict__ bval,
                                                     const int* __restrict__ crp, int* __restrict__ ccol,
                                                     double* __restrict__ cval, int* __restrict__ overflow) {
  __shared__ __attribute__((aligned(16))) int keys[4][SPG_HS];
  __shared__ __attribute__((aligned(16))) int stage[4][SPG_CHUNK * 3];
  __shared__ int list[4][SPG_MAXD];
  __shared__ int ja[4][SPG_KMAX];
  __shared__ int pre[4][SPG_KMAX + 1];
  __shared__ int cntl[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= n) return;
  const int total = spg_prefix(arp, acol, brp, row, lane, ja[w], pre[w]);
  spg_collect(arp, acol, brp, bcol, row, lane, keys[w], ja[w], pre[w], total, overflow);
  const int base = crp[row], cnt = crp[row + 1] - base;
  if (cnt > SPG_MAXD) return;   // flagged by the count pass
  if (lane == 0) cntl[w] = 0;
  for (int t = lane; t < SPG_MAXD; t += 64) list[w][t] = 0x7fffffff;
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < SPG_HS; s += 64) {
    const int key = keys[w][s];
    if (key != -1) list[w][atomicAdd(&cntl[w], 1)] = key;
  }
  __builtin_amdgcn_wave_barrier();
  int N = 64;
  while (N < cnt) N <<= 1;
  wave_bitonic_sort(list[w], nullptr, N, lane);
  int mycol[SPG_MAXD / 64];
  double acc[SPG_MAXD / 64];
#pragma unroll
  for (int u = 0; u < SPG_MAXD / 64; ++u) {
    const int t = lane + 64 * u;
    mycol[u] = (t < cnt) ? list[w][t] : -2;
    acc[u] = 0.0;
  }
  const int nu = (cnt + 63) >> 6;
  if (total >= 0) {
    // the hash table has done its job (its keys are in `list`, sorted): its LDS now holds the staging tiles
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int* skey = stage[w] + 2 * SPG_CHUNK;
    double* sval = reinterpret_cast<double*>(stage[w]);
    const int a0 = arp[row], na = arp[row + 1] - a0;
    for (int p0 = 0; p0 < total; p0 += SPG_CHUNK) {
      const int nc = (total - p0 < SPG_CHUNK) ? total - p0 : SPG_CHUNK;
      for (int q = lane; q < nc; q += 64) {          // stage: every lane fetches its own products
        const int p = p0 + q;
        const int k = spg_find(pre[w], na, p);
        const int l = brp[ja[w][k]] + (p - pre[w][k]);
        skey[q] = bcol[l];
        sval[q] = aval[a0 + k] * bval[l];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // owner computes, products in their fixed order; eight (key, value) pairs are read per round (LDS broadcasts, all
      // in flight together) so that the LDS latency is paid once per eight products
      int q = 0;
      for (; q + 8 <= nc; q += 8) {
        int kk[8];
        double vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { kk[t] = skey[q + t]; vv[t] = sval[q + t]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
          for (int u = 0; u < SPG_MAXD / 64; ++u)
            if (u < nu && kk[t] == mycol[u]) acc[u] += vv[t];
        }
      }
      for (; q < nc; ++q) {
        const int key = skey[q];
        const double v = sval[q];
#pragma unroll
        for (int u = 0; u < SPG_MAXD / 64; ++u)
          if (u < nu && key == mycol[u]) acc[u] += v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    for (int k = arp[row]; k < arp[row + 1]; ++k) {
      const int j = acol[k];
      const double av = aval[k];
      for (int l = brp[j]; l < brp[j + 1]; ++l) {      // every lane walks the whole row of B: broadcast loads
        const int key = bcol[l];
        const double v = av * bval[l];
#pragma unroll
        for (int u = 0; u < SPG_MAXD / 64; ++u)
          if (u < nu && key == mycol[u]) acc[u] += v;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < SPG_MAXD / 64; ++u) {
    const int t = lane + 64 * u;
    if (t < cnt) {
      ccol[base + t] = mycol[u];
      cval[base + t] = acc[u];
    }
  }
}
// Numeric pass, round-3 form (default): the hash table of the symbolic collect keeps its keys and gets one FP64 accumulator
// per slot.  Products are taken 256 at a time (four per lane, all their loads in flight together), each finds its slot by
// the probe sequence that inserted its column, and the additions run SEGMENT BY SEGMENT: the products of one entry k of
// the A row hit pairwise different columns (a CSR row of B has none twice), so their read-modify-writes are independent,
// and the segments follow each other in k order -- every output entry is summed in the (k, l) order of the owner-computes
// scan above, bit for bit, but a product costs a handful of LDS operations instead of a compare in every lane
// (R (A P) at 126^3: 1 250 products for ~30 columns per row).  The (column, sum) pairs are then compacted and sorted.
__device__ __forceinline__ int spg_lookup(const int* keys, int key) {
  unsigned h = ((unsigned)key * 2654435761u) & (SPG_HS - 1);
  while (keys[h] != key) h = (h + 1) & (SPG_HS - 1);
  return (int)h;
}
__global__ __launch_bounds__(256) void k_spgemm_fill(int n, const int* __restrict__ arp, const int* __restrict__ acol,
                                                     const double* __restrict__ aval, const int* __restrict__ brp,
                                                     const int* __restrict__ bcol, const double* __restrict__ bval,
                                                     const int* __restrict__ crp, int* __restrict__ ccol,
                                                     double* __restrict__ cval, int* __restrict__ overflow,
                                                     const unsigned char* __restrict__ defer) {
  __shared__ int keys[4][SPG_HS];
  __shared__ double hval[4][SPG_HS];
  __shared__ int list[4][SPG_MAXD];
  __shared__ double lval[4][SPG_MAXD];
  __shared__ int ja[4][SPG_KMAX];
  __shared__ int pre[4][SPG_KMAX + 1];
  __shared__ int cntl[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= n) return;
  if (defer && !defer[row]) return;       // done by k_spgemm_small
  const int total = spg_prefix(arp, acol, brp, row, lane, ja[w], pre[w]);
  const int base = crp[row], cnt = crp[row + 1] - base;
  if (cnt > SPG_MAXD) return;   // flagged by the count pass
  if (total >= 0 && total <= 64 && !g_spgemm_no_small) {
    unsigned long long ck;
    double v;
    spg_small_products(brp, bcol, aval, bval, arp[row], arp[row + 1] - arp[row], total, lane, ja[w], pre[w], ck, v);
    spg_sort64(ck, v, lane);
    const unsigned long long prev = __shfl_up(ck, 1, 64);
    const bool head = (ck != ~0ull) && (lane == 0 || (prev >> 6) != (ck >> 6));
    double acc = 0.0 + v;                          // the scans start every sum at +0.0
    for (int d = 1; d < 64; ++d) {                 // left fold of each run into its head lane, in product order
      const unsigned long long nk = __shfl_down(ck, d, 64);
      const double nv = __shfl_down(v, d, 64);
      const bool more = head && lane + d < 64 && (nk >> 6) == (ck >> 6);
      if (more) acc += nv;
      if (!__ballot(more)) break;
    }
    const unsigned long long heads = __ballot(head);
    if (head) {
      const int pos = __builtin_popcountll(heads & ((1ull << lane) - 1ull));
      ccol[base + pos] = (int)(ck >> 6);
      cval[base + pos] = acc;
    }
    return;
  }
  spg_collect(arp, acol, brp, bcol, row, lane, keys[w], ja[w], pre[w], total, overflow);
  const int* kw = keys[w];
  double* hv = hval[w];
  for (int s = lane; s < SPG_HS; s += 64) hv[s] = 0.0;
  if (lane == 0) cntl[w] = 0;
  for (int t = lane; t < SPG_MAXD; t += 64) list[w][t] = 0x7fffffff;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int a0 = arp[row], na = arp[row + 1] - a0;
  if (total >= 0) {
    constexpr int NB = 4;                         // products per lane and round
    for (int p0 = 0; p0 < total; p0 += 64 * NB) {
      int kk[NB], slot[NB];
      double vv[NB];
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int p = p0 + 64 * t + lane;
        kk[t] = (p < total) ? spg_find(pre[w], na, p) : 0x7fffffff;
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        slot[t] = 0;
        vv[t] = 0.0;
        if (kk[t] != 0x7fffffff) {
          const int l = brp[ja[w][kk[t]]] + (p0 + 64 * t + lane - pre[w][kk[t]]);
          slot[t] = bcol[l];                      // the column for now
          vv[t] = aval[a0 + kk[t]] * bval[l];
        }
      }
#pragma unroll
      for (int t = 0; t < NB; ++t)
        if (kk[t] != 0x7fffffff) slot[t] = spg_lookup(kw, slot[t]);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        if (p0 + 64 * t >= total) break;          // wave-uniform
        int kc = __builtin_amdgcn_readfirstlane(kk[t]);   // lane 0 of a started batch holds a product
        while (true) {
          if (kk[t] == kc) hv[slot[t]] += vv[t];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          const unsigned long long later = __ballot(kk[t] != 0x7fffffff && kk[t] > kc);
          if (!later) break;
          kc = __shfl(kk[t], __builtin_ctzll(later), 64);
        }
      }
    }
  } else {
    for (int k = a0; k < a0 + na; ++k) {          // long row of A: entry by entry, lanes over the row of B
      const int j = acol[k];
      const double av = aval[k];
      for (int l = brp[j] + lane; l < brp[j + 1]; l += 64) hv[spg_lookup(kw, bcol[l])] += av * bval[l];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < SPG_HS; s += 64) {
    const int key = kw[s];
    if (key != -1) {
      const int t = atomicAdd(&cntl[w], 1);
      list[w][t] = key;
      lval[w][t] = hv[s];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int N = 64;
  while (N < cnt) N <<= 1;
  wave_bitonic_sort(list[w], lval[w], N, lane);
  for (int t = lane; t < cnt; t += 64) {
    ccol[base + t] = list[w][t];
    cval[base + t] = lval[w][t];
  }
}
// Short rows, G lanes per row (64 / G rows per wave): rows of A with at most G entries whose products number at most G
// -- A P0 of a stencil matrix (7 products: G = 8, eight rows per wave), A P (about 25: G = 32), the first coarse level's
// A P0 (31: G = 64).  One wave per row spent most of its time in per-row latency (prefix table and hash table in LDS,
// dependent loads) with 7 of 64 lanes busy: 2.29 M rows of A P0 took 1.8 + 4.4 ms (count + fill).  Here everything lives in
// registers: the B-row lengths are scanned with shuffles inside the group, lane p finds the entry its product belongs to,
// the (column, product number) keys are sorted by a bitonic network inside the group, run heads fold their run from the
// left (product order: the scans' sum order, bit-identical results) and write the row, already sorted.  Rows that do not fit
// are marked in `defer` and left to the wave-per-row kernels (FILL pass: recomputed, same decision).
template <int G, bool FILL>
__global__ __launch_bounds__(256) void k_spgemm_small(int n, const int* __restrict__ arp, const int* __restrict__ acol,
                                                      const double* __restrict__ aval, const int* __restrict__ brp,
                                                      const int* __restrict__ bcol, const double* __restrict__ bval,
                                                      int* __restrict__ cnt, const int* __restrict__ crp,
                                                      int* __restrict__ ccol, double* __restrict__ cval,
                                                      unsigned char* __restrict__ defer, int* __restrict__ flags) {
  constexpr int RPW = 64 / G;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gl = lane % G, gbase = lane - gl;
  const int row = (blockIdx.x * 4 + w) * RPW + lane / G;
  const bool live = row < n;
  const int a0 = live ? arp[row] : 0;
  const int na = live ? arp[row + 1] - a0 : 0;
  int b0 = 0, len = 0;
  double av = 0.0;
  if (gl < na && na <= G) {
    const int j = acol[a0 + gl];
    b0 = brp[j];
    len = brp[j + 1] - b0;
    if (FILL) av = aval[a0 + gl];
  }
  int incl = len;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) {
    const int t = __shfl_up(incl, o, G);
    if (gl >= o) incl += t;
  }
  const int total = __shfl(incl, G - 1, G);
  const bool fits = na <= G && total <= G;
  if (!FILL && live && gl == 0) {
    defer[row] = fits ? 0 : 1;
    if (!fits) atomicAdd(&flags[1], 1);
  }
  // product p = gl of the row: entry k of the A row with incl[k - 1] <= p < incl[k]
  int k = 0;
#pragma unroll
  for (int q = 0; q < G; ++q) k += (__shfl(incl, q, G) <= gl) ? 1 : 0;
  const bool has = fits && gl < total;
  const int ks = has ? k : 0;
  const int kb0 = __shfl(b0, ks, G);
  const int kexcl = __shfl(incl - len, ks, G);
  double kav = 0.0;
  if (FILL) kav = __shfl(av, ks, G);
  unsigned long long ck = ~0ull;
  double v = 0.0;
  if (has) {
    const int l = kb0 + (gl - kexcl);
    ck = ((unsigned long long)(unsigned)bcol[l] << 6) | (unsigned)gl;
    if (FILL) v = kav * bval[l];
  }
#pragma unroll
  for (int k2 = 2; k2 <= G; k2 <<= 1)
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const unsigned long long ock = __shfl_xor(ck, j, 64);
      double ov = 0.0;
      if (FILL) ov = __shfl_xor(v, j, 64);
      const bool take_min = (((gl & j) == 0) == ((gl & k2) == 0));
      const bool swap = take_min ? (ock < ck) : (ock > ck);
      if (swap) { ck = ock; if (FILL) v = ov; }
    }
  const unsigned long long prev = __shfl_up(ck, 1, G);
  const bool head = (ck != ~0ull) && (gl == 0 || (prev >> 6) != (ck >> 6));
  const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << gbase);
  const unsigned long long heads = __ballot(head) & gmask;
  if (!FILL) {
    if (live && fits && gl == 0) cnt[row] = __builtin_popcountll(heads);
    return;
  }
  double acc = 0.0 + v;                          // the scans start every sum at +0.0
  for (int d = 1; d < G; ++d) {                  // left fold of each run into its head lane, in product order
    const unsigned long long nk = __shfl_down(ck, d, G);
    const double nv = __shfl_down(v, d, G);
    const bool more = head && gl + d < G && (nk >> 6) == (ck >> 6);
    if (more) acc += nv;
    if (!__ballot(more)) break;
  }
  if (head) {
    const int pos = __builtin_popcountll(heads & ((1ull << lane) - 1ull));
    const int base = crp[row];
    ccol[base + pos] = (int)(ck >> 6);
    cval[base + pos] = acc;
  }
}
// rowptr[0..n] from the counts in rowptr[1..n]: inclusive scan on the device (round 2 took the counts to the host and back:
// two copies of n integers and a serial loop per product -- 10 ms per call at 6.5 M rows, four calls per level).
// Three-phase scan: 1024-element blocks (LDS), their totals scanned by the same kernel recursively, totals added back.
constexpr int SCAN_B = 1024;
__global__ __launch_bounds__(256) void k_scan_blocks(int* __restrict__ x, int64_t n, int64_t* __restrict__ totals) {
  __shared__ int64_t part[256];
  const int64_t b0 = (int64_t)blockIdx.x * SCAN_B;
  const int t = threadIdx.x;
  int64_t v[4], run = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b0 + 4 * t + u;
    run += (i < n) ? x[i] : 0;
    v[u] = run;
  }
  part[t] = run;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {           // Hillis-Steele over the 256 thread totals
    const int64_t add = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const int64_t before = (t > 0) ? part[t - 1] : 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b0 + 4 * t + u;
    if (i < n) x[i] = (int)(before + v[u]);
  }
  if (t == 255) totals[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void k_scan_blocks64(int64_t* __restrict__ x, int64_t n, int64_t* __restrict__ totals) {
  __shared__ int64_t part[256];
  const int64_t b0 = (int64_t)blockIdx.x * SCAN_B;
  const int t = threadIdx.x;
  int64_t v[4], run = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b0 + 4 * t + u;
    run += (i < n) ? x[i] : 0;
    v[u] = run;
  }
  part[t] = run;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int64_t add = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const int64_t before = (t > 0) ? part[t - 1] : 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b0 + 4 * t + u;
    if (i < n) x[i] = before + v[u];
  }
  if (t == 255) totals[blockIdx.x] = part[255];
}
__global__ void k_scan_add(int* __restrict__ x, int64_t n, const int64_t* __restrict__ totals_incl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + SCAN_B; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] += (int)totals_incl[i / SCAN_B - 1];
}
__global__ void k_scan_add64(int64_t* __restrict__ x, int64_t n, const int64_t* __restrict__ totals_incl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + SCAN_B; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] += totals_incl[i / SCAN_B - 1];
}
static void device_inclusive_scan64(int64_t* x, int64_t n) {       // block totals of a large scan (n <= a few 10^4)
  if (n <= 0) return;
  const int64_t nb = (n + SCAN_B - 1) / SCAN_B;
  int64_t* tot = (int64_t*)alloc(sizeof(int64_t) * (size_t)nb);
  hipLaunchKernelGGL(k_scan_blocks64, dim3((unsigned)nb), dim3(256), 0, g_stream, x, n, tot);
  if (nb > 1) {
    device_inclusive_scan64(tot, nb);
    hipLaunchKernelGGL(k_scan_add64, dim3((unsigned)std::min<int64_t>(nb, 4096)), dim3(256), 0, g_stream, x, n, tot);
  }
  dfree(tot);
}
static void host_exclusive_scan(int* dcnt_to_ptr, int n, int64_t* total) {   // rowptr[0..n] from counts in rowptr[1..n]
  if (n <= 0) { *total = 0; return; }
  static const bool on_host = getenv("GENEO_SCAN_HOST") != nullptr;
  if (on_host || (int64_t)n < 4 * SCAN_B) {
    std::vector<int> h((size_t)n + 1, 0);
    d2h(h.data() + 1, dcnt_to_ptr + 1, sizeof(int) * (size_t)n);
    int64_t run = 0;
    for (int i = 1; i <= n; ++i) {
      run += h[i];
      h[i] = (int)run;
    }
    h[0] = 0;
    h2d(dcnt_to_ptr, h.data(), sizeof(int) * ((size_t)n + 1));
    *total = run;
    return;
  }
  int* x = dcnt_to_ptr + 1;
  const int64_t nb = ((int64_t)n + SCAN_B - 1) / SCAN_B;
  int64_t* tot = (int64_t*)alloc(sizeof(int64_t) * (size_t)nb);
  hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(256), 0, g_stream, x, (int64_t)n, tot);
  device_inclusive_scan64(tot, nb);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)std::min<int64_t>(nb, 4096)), dim3(256), 0, g_stream, x, (int64_t)n, tot);
  HIPCHK(hipMemsetAsync(dcnt_to_ptr, 0, sizeof(int), g_stream));
  int64_t run = 0;
  d2h(&run, tot + (nb - 1), sizeof(int64_t));
  dfree(tot);
  if (run > 0x7fffffffLL) throw std::runtime_error("sparse product: more than 2^31 entries");
  *total = run;
}
static bool g_spgemm_small_host = true;     // host mirror of g_spgemm_no_small (validation switch "spgemm_small_rows")
Csr spgemm(const Csr& a, const Csr& b, int ncols_b, bool* ok) {
  (void)ncols_b;
  Csr c;
  *ok = true;
  c.n = a.n;
  if (a.n == 0) return c;
  int* dflag = (int*)alloc(2 * sizeof(int));          // [0] overflow, [1] rows the group pass left to the wave-per-row kernels
  c.rowptr = (int*)alloc(sizeof(int) * ((size_t)a.n + 1));
  // lanes per row of the short-row pass, from the average row: most rows of a mesh-like matrix look like the average one
  int G = 0;
  if (g_spgemm_small_host && !g_spgemm_fill_scan && a.n >= 256 && b.n > 0) {
    const double na = (double)a.nnz / a.n, est = na * ((double)b.nnz / b.n);
    if (na <= 7.5 && est <= 7.5) G = 8;
    else if (na <= 15.0 && est <= 14.0) G = 16;
    else if (na <= 30.0 && est <= 26.0) G = 32;
    else if (na <= 60.0 && est <= 52.0) G = 64;
  }
  unsigned char* defer = nullptr;
#define SPG_SMALL(GG, FILL)                                                                                                  \
  hipLaunchKernelGGL((k_spgemm_small<GG, FILL>), dim3((unsigned)((a.n + 256 / GG - 1) / (256 / GG))), dim3(256), 0, g_stream, \
                     a.n, a.rowptr, a.col, a.val, b.rowptr, b.col, b.val, c.rowptr + 1, c.rowptr, c.col, c.val, defer, dflag)
#define SPG_SMALL_G(FILL)                                      \
  do {                                                         \
    if (G == 8) SPG_SMALL(8, FILL);                            \
    else if (G == 16) SPG_SMALL(16, FILL);                     \
    else if (G == 32) SPG_SMALL(32, FILL);                     \
    else SPG_SMALL(64, FILL);                                  \
  } while (0)
  int hflag[2] = {0, 0};
  if (G) {
    defer = (unsigned char*)alloc((size_t)a.n);
    SPG_SMALL_G(false);
    d2h(hflag, dflag, 2 * sizeof(int));
  }
  const bool general = !G || hflag[1] > 0;
  if (general) {
    hipLaunchKernelGGL(k_spgemm_count, dim3((a.n + 3) / 4), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, b.rowptr, b.col,
                       c.rowptr + 1, dflag, defer);
    d2h(hflag, dflag, sizeof(int));
  }
  if (hflag[0]) {
    dfree(dflag);
    dfree(defer);
    dfree(c.rowptr);
    *ok = false;
    return Csr();
  }
  int64_t total = 0;
  host_exclusive_scan(c.rowptr, a.n, &total);
  c.nnz = total;
  c.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)total));
  c.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)total));
  if (G) SPG_SMALL_G(true);
  if (general) {
    if (g_spgemm_fill_scan)
      hipLaunchKernelGGL(k_spgemm_fill_scan, dim3((a.n + 3) / 4), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, b.rowptr,
                         b.col, b.val, c.rowptr, c.col, c.val, dflag);
    else
      hipLaunchKernelGGL(k_spgemm_fill, dim3((a.n + 3) / 4), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, b.rowptr,
                         b.col, b.val, c.rowptr, c.col, c.val, dflag, defer);
    d2h(hflag, dflag, sizeof(int));
  }
#undef SPG_SMALL_G
#undef SPG_SMALL
  dfree(dflag);
  dfree(defer);
  if (hflag[0]) {
    dfree(c.rowptr); dfree(c.col); dfree(c.val);
    *ok = false;
    return Csr();
  }
  return c;
}

// Transpose: column counts (integer atomics: exact), host scan, scatter with an atomic cursor, then every row of the
// result is sorted by its column index (= the source row), which restores a unique order.
constexpr int TR_MAXROW = 1024;
__global__ void k_col_count(const int* __restrict__ col, int64_t nnz, int* __restrict__ cnt) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&cnt[col[e]], 1);
}
__global__ void k_tr_scatter(int n, const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                             const int* __restrict__ trp, int* __restrict__ cursor, int* __restrict__ tcol,
                             double* __restrict__ tval) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int c = col[k];
      const int p = trp[c] + atomicAdd(&cursor[c], 1);
      tcol[p] = i;
      tval[p] = val[k];
    }
}
__global__ __launch_bounds__(256) void k_sort_rows(int n, const int* __restrict__ rp, int* __restrict__ col,
                                                   double* __restrict__ val, int* __restrict__ overflow) {
  __shared__ int key[4][TR_MAXROW];
  __shared__ double v[4][TR_MAXROW];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= n) return;
  const int base = rp[row], cnt = rp[row + 1] - base;
  if (cnt <= 1) return;
  if (cnt > TR_MAXROW) {
    if (lane == 0) *overflow = 1;
    return;
  }
  int N = 64;
  while (N < cnt) N <<= 1;
  for (int t = lane; t < N; t += 64) {
    key[w][t] = (t < cnt) ? col[base + t] : 0x7fffffff;
    v[w][t] = (t < cnt) ? val[base + t] : 0.0;
  }
  __builtin_amdgcn_wave_barrier();
  wave_bitonic_sort(key[w], v[w], N, lane);
  for (int t = lane; t < cnt; t += 64) {
    col[base + t] = key[w][t];
    val[base + t] = v[w][t];
  }
}
Csr transpose(const Csr& a, int ncols, bool* ok) {
  Csr t;
  *ok = true;
  t.n = ncols;
  t.nnz = a.nnz;
  t.rowptr = (int*)alloc(sizeof(int) * ((size_t)ncols + 1));
  t.col = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)a.nnz));
  t.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  if (a.nnz > 0)
    hipLaunchKernelGGL(k_col_count, dim3(gridv(a.nnz)), dim3(256), 0, g_stream, a.col, (int64_t)a.nnz, t.rowptr + 1);
  int64_t total = 0;
  host_exclusive_scan(t.rowptr, ncols, &total);
  int* cursor = (int*)alloc(sizeof(int) * std::max<size_t>(1, (size_t)ncols));
  int* dflag = (int*)alloc(sizeof(int));
  if (a.n > 0)
    hipLaunchKernelGGL(k_tr_scatter, dim3(gridv(a.n)), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, t.rowptr,
                       cursor, t.col, t.val);
  if (ncols > 0)
    hipLaunchKernelGGL(k_sort_rows, dim3((ncols + 3) / 4), dim3(256), 0, g_stream, ncols, t.rowptr, t.col, t.val, dflag);
  int hflag = 0;
  d2h(&hflag, dflag, sizeof(int));
  dfree(dflag);
  dfree(cursor);
  if (hflag) {
    dfree(t.rowptr); dfree(t.col); dfree(t.val);
    *ok = false;
    return Csr();
  }
  return t;
}
__global__ void k_smooth_p(int n, const int* __restrict__ rp, const int* __restrict__ col, double* __restrict__ val,
                           const int* __restrict__ agg, const double* __restrict__ dinv, double w) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double f = -w * dinv[i];
    const int mine = agg[i];
    for (int k = rp[i]; k < rp[i + 1]; ++k) val[k] = f * val[k] + (col[k] == mine ? 1.0 : 0.0);
  }
}
void smooth_prolongator(Csr& ap0, const int* agg_dev, const double* dinv_dev, double w) {
  if (ap0.n == 0) return;
  hipLaunchKernelGGL(k_smooth_p, dim3(gridv(ap0.n)), dim3(256), 0, g_stream, ap0.n, ap0.rowptr, ap0.col, ap0.val, agg_dev,
                     dinv_dev, w);
}

// M = P - w D^-1 (A P) over the pattern of A P (one thread per row; the rows of P are short: linear search)
__global__ void k_post_matrix(int n, const int* __restrict__ arp, const int* __restrict__ acol, double* __restrict__ aval,
                              const int* __restrict__ prp, const int* __restrict__ pcol, const double* __restrict__ pval,
                              const double* __restrict__ dinv, double w, int* __restrict__ miss) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int p0 = prp[i], p1 = prp[i + 1];
    const double f = w * dinv[i];
    int found = 0;
    for (int k = arp[i]; k < arp[i + 1]; ++k) {
      const int c = acol[k];
      double v = -f * aval[k];
      for (int q = p0; q < p1; ++q)
        if (pcol[q] == c) { v += pval[q]; ++found; break; }
      aval[k] = v;
    }
    if (found != p1 - p0) atomicExch(miss, 1);
  }
}
bool post_matrix(Csr& ap, const Csr& p, const double* dinv, double w) {
  if (ap.n == 0) return true;
  int* dmiss = (int*)alloc(sizeof(int));
  hipLaunchKernelGGL(k_post_matrix, dim3(gridv(ap.n)), dim3(256), 0, g_stream, ap.n, ap.rowptr, ap.col, ap.val, p.rowptr,
                     p.col, p.val, dinv, w, dmiss);
  int miss = 0;
  d2h(&miss, dmiss, sizeof(int));
  dfree(dmiss);
  return miss == 0;
}

void csr_free_lp(Csr& a) {
  dfree(a.lp_val);
  if (!a.alias) { dfree(a.lp_col); dfree(a.lp_base); }
  a.lp_val = nullptr; a.lp_col = nullptr; a.lp_base = nullptr;
}
void csr_free(Csr& a) {
  csr_free_lp(a);
  if (a.alias) {   // values-only copy: the index arrays belong to the matrix it was made from
    dfree(a.val); dfree(a.sl_val);
    a = Csr();
    return;
  }
  dfree(a.rowptr); dfree(a.col); dfree(a.val); dfree(a.rowblk);
  dfree(a.sl_ptr); dfree(a.sl_col); dfree(a.sl_val); dfree(a.long_rows);
  dfree(a.sched); dfree(a.xcd_ptr);
  a = Csr();
}

__global__ void k_scale_csr(int n, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                            const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double a = rs ? rs[r] : 1.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) out[k] = a * val[k] * (cs ? cs[col[k]] : 1.0);
}
__global__ __launch_bounds__(256) void k_scale_sell(int n, int ns, const int64_t* __restrict__ sl_ptr,
                                                    const int* __restrict__ sl_col, const double* __restrict__ sl_val,
                                                    const double* __restrict__ rs, const double* __restrict__ cs,
                                                    double* __restrict__ out) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int l = threadIdx.x & 63;
  const int r = 64 * s + l;
  const double a = (rs && r < n) ? rs[r] : 1.0;
  for (int64_t e = sl_ptr[s] + l; e < sl_ptr[s + 1]; e += 64) out[e] = a * sl_val[e] * (cs ? cs[sl_col[e]] : 1.0);
}
Csr csr_scaled_alias(const Csr& a, const double* row_scale, const double* col_scale, bool col_is_dinv) {
  Csr b = a;
  b.alias = true;
  b.lp_val = nullptr; b.lp_col = nullptr; b.lp_base = nullptr;
  b.col_scaled = col_is_dinv;
  b.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  b.sl_val = (double*)alloc(sizeof(double) * (size_t)std::max<int64_t>(1, a.sl_nnz));
  if (a.n > 0)
    hipLaunchKernelGGL(k_scale_csr, dim3(grid1d(a.n, 256)), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, row_scale,
                       col_scale, b.val);
  if (a.nslice > 0)
    hipLaunchKernelGGL(k_scale_sell, dim3((a.nslice + 3) / 4), dim3(256), 0, g_stream, a.n, a.nslice, a.sl_ptr, a.sl_col,
                       a.sl_val, row_scale, col_scale, b.sl_val);
  return b;
}

// XCD-aware remap: hardware deals workgroups round-robin over the 8 XCDs (b % 8 = XCD group).
// Give each XCD a contiguous eighth of the row blocks so that its private 4 MiB L2 sees a
// contiguous window of x instead of the whole vector.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int per = (nblk + 7) >> 3;
  const int t = (b & 7) * per + (b >> 3);
  return t;
}

__global__ __launch_bounds__(256) void k_spmv_lds(const int* __restrict__ rowblk, int nblk,
                                                  const int* __restrict__ rowptr,
                                                  const int* __restrict__ col,
                                                  const double* __restrict__ val,
                                                  const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[SPMV_TILE];
  const int t = xcd_remap(blockIdx.x, nblk);
  if (t >= nblk) return;
  const int r0 = rowblk[t], r1 = rowblk[t + 1];
  const int nz0 = rowptr[r0], nz1 = rowptr[r1];
  const int cnt = nz1 - nz0;
  const int tid = threadIdx.x;
  if (cnt > SPMV_TILE) {  // single long row: block-wide strided reduction
    double s = 0.0;
    for (int k = nz0 + tid; k < nz1; k += 256) s += val[k] * x[col[k]];
    prod[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) prod[tid] += prod[tid + w];
      __syncthreads();
    }
    if (tid == 0) y[r0] = prod[0];
    return;
  }
  // phase 1: coalesced stream of (col,val), gather x, products into LDS
  for (int k = tid; k < cnt; k += 256) prod[k] = val[nz0 + k] * x[col[nz0 + k]];
  __syncthreads();
  // phase 2: one row per thread, segmented sum out of LDS (fixed left-to-right order)
  const int r = r0 + tid;
  if (r < r1) {
    const int a = rowptr[r] - nz0, b = rowptr[r + 1] - nz0;
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    y[r] = s;
  }
}

// Column bases of the 16-bit column offsets (see k_lp_base): {b0, b1, ks, -} per slice, one 16-byte load.
__device__ __forceinline__ int4 lp_base_of(const int* __restrict__ base, int s) {
  return base ? reinterpret_cast<const int4*>(base)[s] : int4{0, 0, 0x7fffffff, 0};
}
// Sliced kernel: one wave per 64-row slice, lane i owns row i.  Per k the wave issues one coalesced
// 512-B val load, one 256-B col load and one x gather whose 64 addresses are the k-th neighbours of
// 64 consecutive rows (contiguous for stencil-like matrices) -- no LDS round trip, no barrier.
// UNR independent k-steps are in flight per lane; NT marks the once-read (col,val) stream
// non-temporal so that it does not displace x from the XCD's L2.
// COLT = unsigned short: the columns are read as 16-bit offsets from the slice's lowest column (cbase[s]; the index
// arrays of the single-precision companion, shared) -- 10 bytes per entry instead of 12 with the FP64 values untouched.
// Slices of KW <= 8 entries per row with every (col, val) load in front of the first gather and all gathers in flight
// together -- two dependent latencies instead of up to eight -- and the products summed in the order of the 4-step loop of
// k_spmv_sell<4, ..> below (KW / 4 rounds into four accumulators, the remainder into the first, then acc0 + acc1 + acc2 +
// acc3): bit-identical to it.  (k_spmv_sell_p8 further down is the round-3 form of the same idea with another summation
// order; the single-precision companion kernels have theirs in lp_row_sum_fixed.)
template <int KW, bool NT, typename COLT>
__device__ __forceinline__ double spmv_row_sum_fixed(const COLT* __restrict__ col, const double* __restrict__ val, int64_t e0,
                                                     const int4 cb, const double* __restrict__ x) {
  int c[KW];
  double v[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) {
    c[u] = (u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e0 + 64 * u) : col[e0 + 64 * u]);
    v[u] = NT ? __builtin_nontemporal_load(val + e0 + 64 * u) : val[e0 + 64 * u];
  }
  double xv[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) xv[u] = x[c[u]];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int NG = KW / 4;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += v[4 * g + u] * xv[4 * g + u];
#pragma unroll
  for (int k = 4 * NG; k < KW; ++k) acc[0] += v[k] * xv[k];
  return ((acc[0] + acc[1]) + acc[2]) + acc[3];
}
template <int UNR, bool NT, typename COLT = int>
__global__ __launch_bounds__(256) void k_spmv_sell(const int64_t* __restrict__ sl_ptr, int nslice, int n,
                                                   const COLT* __restrict__ col, const double* __restrict__ val,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   const int* __restrict__ cbase = nullptr) {
  const int nwb = (nslice + 3) >> 2;            // workgroups (4 slices each)
  const int t = xcd_remap(blockIdx.x, nwb);
  const int s = 4 * t + (threadIdx.x >> 6);
  if (t >= nwb || s >= nslice) return;
  const int l = threadIdx.x & 63;
  const int64_t a = sl_ptr[s], b = sl_ptr[s + 1];
  const int4 cb = lp_base_of(cbase, s);       // {b0, b1, ks}: entries k < ks count from b0, the others from b1
  if (UNR == 4 && !g_lp_no_fixed) {           // narrow slice: the two-latency form (same sums, same order)
    const int wd = (int)((b - a) >> 6);
    if (wd >= 1 && wd <= 8) {
      double fs;
      switch (wd) {
        case 1: fs = spmv_row_sum_fixed<1, NT, COLT>(col, val, a + l, cb, x); break;
        case 2: fs = spmv_row_sum_fixed<2, NT, COLT>(col, val, a + l, cb, x); break;
        case 3: fs = spmv_row_sum_fixed<3, NT, COLT>(col, val, a + l, cb, x); break;
        case 4: fs = spmv_row_sum_fixed<4, NT, COLT>(col, val, a + l, cb, x); break;
        case 5: fs = spmv_row_sum_fixed<5, NT, COLT>(col, val, a + l, cb, x); break;
        case 6: fs = spmv_row_sum_fixed<6, NT, COLT>(col, val, a + l, cb, x); break;
        case 7: fs = spmv_row_sum_fixed<7, NT, COLT>(col, val, a + l, cb, x); break;
        default: fs = spmv_row_sum_fixed<8, NT, COLT>(col, val, a + l, cb, x); break;
      }
      const int rr = 64 * s + l;
      if (rr < n) y[rr] = fs;
      return;
    }
  }
  double acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u] = 0.0;
  int64_t e = a + l;
  int k = 0;
  for (; e + 64 * (UNR - 1) < b; e += 64 * UNR, k += UNR) {
    int c[UNR];
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      c[u] = (k + u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e + 64 * u) : col[e + 64 * u]);
      v[u] = NT ? __builtin_nontemporal_load(val + e + 64 * u) : val[e + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc[u] += v[u] * x[c[u]];
  }
  for (; e < b; e += 64, ++k) {
    const int c0 = (k < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e) : col[e]);
    const double v0 = NT ? __builtin_nontemporal_load(val + e) : val[e];
    acc[0] += v0 * x[c0];
  }
  double sum = acc[0];
#pragma unroll
  for (int u = 1; u < UNR; ++u) sum += acc[u];
  const int r = 64 * s + l;
  if (r < n) y[r] = sum;
}

// The same traversal with EIGHT predicated k-steps per round: a slice of a 7-point matrix is 7 entries wide, and with
// the 4-step loop above its last three entries go through the one-at-a-time tail -- three more dependent
// (col -> x) latency pairs per wave.  Here every (col, val) pair of the slice is requested before the first x gather
// and all gathers are in flight together: two dependent latencies per slice.  Lanes past the slice's width load nothing
// (predicated), so no extra traffic.
template <bool NT, typename COLT = int>
__global__ __launch_bounds__(256) void k_spmv_sell_p8(const int64_t* __restrict__ sl_ptr, int nslice, int n,
                                                      const COLT* __restrict__ col, const double* __restrict__ val,
                                                      const double* __restrict__ x, double* __restrict__ y,
                                                      const int* __restrict__ cbase = nullptr) {
  const int nwb = (nslice + 3) >> 2;
  const int t = xcd_remap(blockIdx.x, nwb);
  const int s = 4 * t + (threadIdx.x >> 6);
  if (t >= nwb || s >= nslice) return;
  const int l = threadIdx.x & 63;
  const int64_t a = sl_ptr[s], b = sl_ptr[s + 1];
  const int4 cb = lp_base_of(cbase, s);
  double acc0 = 0.0, acc1 = 0.0;
  int k = 0;
  for (int64_t e = a + l; e < b; e += 64 * 8, k += 8) {
    int c[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = e + 64 * u < b;
      c[u] = cb.x;
      v[u] = 0.0;
      if (ok) {
        c[u] = (k + u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e + 64 * u) : col[e + 64 * u]);
        v[u] = NT ? __builtin_nontemporal_load(val + e + 64 * u) : val[e + 64 * u];
      }
    }
    double xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc0 += v[u] * xv[u];
      acc1 += v[u + 1] * xv[u + 1];
    }
  }
  const int r = 64 * s + l;
  if (r < n) y[r] = acc0 + acc1;
}
// Same traversal with the multigrid epilogues of backend.h fused in (EPI_RES / ADD / JAC / PRE): the vector
// passes that used to follow the SpMV (residual, correction, Jacobi update) ride on its output write.
template <bool NT, int EPI>
__global__ __launch_bounds__(256) void k_spmv_sell_epi(const int64_t* __restrict__ sl_ptr, int nslice, int n,
                                                       const int* __restrict__ col, const double* __restrict__ val,
                                                       const double* __restrict__ x, double* __restrict__ y,
                                                       const double* __restrict__ b, double* __restrict__ z,
                                                       const double* __restrict__ dinv, double w,
                                                       const double* __restrict__ cs /* EPI_PRE: column scaling, null when the values carry it */) {
  constexpr int UNR = 4;
  const int nwb = (nslice + 3) >> 2;
  const int t = xcd_remap(blockIdx.x, nwb);
  const int s = 4 * t + (threadIdx.x >> 6);
  if (t >= nwb || s >= nslice) return;
  const int l = threadIdx.x & 63;
  const int64_t a = sl_ptr[s], e1 = sl_ptr[s + 1];
  // the epilogue's own operands, loaded in front of the matrix stream (see k_spmv_sell_lp)
  const int r = 64 * s + l;
  double e_b = 0.0, e_z = 0.0, e_d = 0.0, e_x = 0.0;
  if (r < n) {
    if (EPI == EPI_RES || EPI == EPI_JAC || EPI == EPI_POST || EPI == EPI_PRE) e_b = b[r];
    if (EPI == EPI_ADD || EPI == EPI_POST) e_z = z[r];
    if (EPI == EPI_JAC || EPI == EPI_POST || (EPI == EPI_PRE && z)) e_d = dinv[r];
    if (EPI == EPI_JAC) e_x = x[r];
  }
  double acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u] = 0.0;
  int64_t e = a + l;
  for (; e + 64 * (UNR - 1) < e1; e += 64 * UNR) {
    int c[UNR];
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      c[u] = NT ? __builtin_nontemporal_load(col + e + 64 * u) : col[e + 64 * u];
      v[u] = NT ? __builtin_nontemporal_load(val + e + 64 * u) : val[e + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc[u] += v[u] * (EPI == EPI_PRE ? (cs ? b[c[u]] * cs[c[u]] : b[c[u]]) : x[c[u]]);
  }
  for (; e < e1; e += 64) {
    const int c0 = NT ? __builtin_nontemporal_load(col + e) : col[e];
    const double v0 = NT ? __builtin_nontemporal_load(val + e) : val[e];
    acc[0] += v0 * (EPI == EPI_PRE ? (cs ? b[c0] * cs[c0] : b[c0]) : x[c0]);
  }
  double sum = acc[0];
#pragma unroll
  for (int u = 1; u < UNR; ++u) sum += acc[u];
  if (r >= n) return;
  if (EPI == EPI_RES) {
    y[r] = e_b - sum;
  } else if (EPI == EPI_ADD) {
    y[r] = e_z + sum;
  } else if (EPI == EPI_JAC) {
    y[r] = e_x + w * e_d * (e_b - sum);
  } else if (EPI == EPI_POST) {
    y[r] = w * e_d * (e_z + e_b) + sum;
  } else {  // EPI_PRE
    if (z) z[r] = w * e_d * e_b;
    y[r] = e_b - w * sum;
  }
}
// Wide slices (coarse Galerkin operators, restrictions: 30-60 entries per row on a few thousand slices): with one wave
// per slice the wave's own chain of dependent (col,val) -> x loads is the whole run time and most of the chip idles.
// Here ONE WORKGROUP owns a slice: wave w takes the entries k = w, w + 4, ... (4 of them in flight per lane, 16 per row
// across the workgroup), the four partial sums meet in LDS and wave 0 runs the epilogue.  Summation order is fixed.
template <int EPI>
__global__ __launch_bounds__(256) void k_spmv_sell_wide(const int64_t* __restrict__ sl_ptr, int nslice, int n,
                                                        const int* __restrict__ col, const double* __restrict__ val,
                                                        const double* __restrict__ x, double* __restrict__ y,
                                                        const double* __restrict__ b, double* __restrict__ z,
                                                        const double* __restrict__ dinv, double w,
                                                        const double* __restrict__ cs) {
  constexpr int UNR = 4;
  __shared__ double part[3][64];
  const int s = xcd_remap(blockIdx.x, nslice);
  if (s >= nslice) return;
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int64_t a = sl_ptr[s], e1 = sl_ptr[s + 1];
  const double* __restrict__ xin = (EPI == EPI_PRE) ? b : x;
  double acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u] = 0.0;
  int64_t e = a + 64 * wv + l;
  for (; e + 256 * (UNR - 1) < e1; e += 256 * UNR) {
    int c[UNR];
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      c[u] = col[e + 256 * u];
      v[u] = val[e + 256 * u];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc[u] += v[u] * ((EPI == EPI_PRE && cs) ? xin[c[u]] * cs[c[u]] : xin[c[u]]);
  }
  for (; e < e1; e += 256) {
    const int c0 = col[e];
    acc[0] += val[e] * ((EPI == EPI_PRE && cs) ? xin[c0] * cs[c0] : xin[c0]);
  }
  double sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (wv > 0) part[wv - 1][l] = sum;
  __syncthreads();
  if (wv > 0) return;
  sum = (sum + part[0][l]) + (part[1][l] + part[2][l]);
  const int r = 64 * s + l;
  if (r >= n) return;
  if (EPI == EPI_NONE) {
    y[r] = sum;
  } else if (EPI == EPI_RES) {
    y[r] = b[r] - sum;
  } else if (EPI == EPI_ADD) {
    y[r] = z[r] + sum;
  } else if (EPI == EPI_JAC) {
    y[r] = x[r] + w * dinv[r] * (b[r] - sum);
  } else if (EPI == EPI_POST) {
    y[r] = w * dinv[r] * (z[r] + b[r]) + sum;
  } else {  // EPI_PRE
    const double bb = b[r];
    if (z) z[r] = w * dinv[r] * bb;
    y[r] = bb - w * sum;
  }
}
// Average slice width from which a workgroup (not a wave) owns a slice; 0 = never.  Single vectors: 16 (measured on the
// 126^3 hierarchy: the 12-wide post-smoothing matrix P - w D^-1 A P runs 20 % faster wave-per-slice).  Blocks: 9 -- as
// soon as a slice needs a second 8-entry chunk the wave-per-slice SpMM re-stages it for every row group (that matrix
// on 32 columns: 1.54 ms against 0.61 ms).
static int g_sell_wide = -1, g_sell_wide_mm = -1;
static inline bool sell_wide_at(const Csr& a, int width) {
  return width > 0 && a.nslice > 0 && a.nlong == 0 && a.sl_nnz >= (int64_t)64 * width * a.nslice;
}
static inline bool sell_wide(const Csr& a) {
  if (g_sell_wide < 0) g_sell_wide = getenv("GENEO_SELL_WIDE") ? atoi(getenv("GENEO_SELL_WIDE")) : 16;
  return sell_wide_at(a, g_sell_wide);
}
static inline bool sell_wide_mm(const Csr& a) {
  if (g_sell_wide_mm < 0) g_sell_wide_mm = getenv("GENEO_SELL_WIDE_MM") ? atoi(getenv("GENEO_SELL_WIDE_MM")) : 9;
  return sell_wide_at(a, g_sell_wide_mm);
}
template <int EPI>
static void spmv_wide_launch(const Csr& a, const double* x, double* y, const double* b, double* z, const double* dinv,
                             double w) {
  const int per = (a.nslice + 7) / 8;
  hipLaunchKernelGGL((k_spmv_sell_wide<EPI>), dim3(per * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice, a.n, a.sl_col,
                     a.sl_val, x, y, b, z, dinv, w, a.col_scaled ? (const double*)nullptr : dinv);
}
// In-situ kernel timing (bench.py): while profiling is on, every `every`-th launch of each kernel class is bracketed by
// two HIP events on the launch stream; nothing waits on the host until stop.  Classes (backend.h): fine-level CSR
// SpMV, fine-level SpMM, MFMA Gram, MFMA block update.
struct KProf {
  long long nlaunch = 0;     // direct launches seen (every `every`-th of them is timed)
  long long ngraph = 0;      // launches replayed from HIP graphs (counted, never timed)
  std::vector<hipEvent_t> e0, e1;
  std::vector<double> bytes, flops;
};
static bool g_prof_on = false;
static int g_prof_every = 1;
static double g_prof_min_bytes = 0.0;
static KProf g_kprof[PROF_NCLASS];
static long long g_capture_count[PROF_NCLASS] = {0, 0, 0, 0, 0};
struct GraphCounts { void* exec; long long n[PROF_NCLASS]; };
static std::vector<GraphCounts> g_graph_counts;
struct ProfScope {   // e0 at construction, e1 at destruction, when this launch is one of the sampled ones
  KProf* k = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(int cls, bool counted, double bytes, double flops) {
    if (g_capturing && counted) ++g_capture_count[cls];    // what one replay of the graph being recorded will launch
    if (!g_prof_on || g_capturing || !counted) return;
    KProf& kp = g_kprof[cls];
    if (kp.nlaunch++ % g_prof_every != 0 || kp.e0.size() >= 20000) return;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, g_stream));
    k = &kp;
    kp.bytes.push_back(bytes);
    kp.flops.push_back(flops);
  }
  ~ProfScope() {
    if (!k) return;
    (void)hipEventRecord(b, g_stream);
    k->e0.push_back(a);
    k->e1.push_back(b);
  }
};
static void kprof_clear(KProf& k) {
  for (hipEvent_t e : k.e0) (void)hipEventDestroy(e);
  for (hipEvent_t e : k.e1) (void)hipEventDestroy(e);
  k.e0.clear(); k.e1.clear(); k.bytes.clear(); k.flops.clear();
  k.nlaunch = 0;
  k.ngraph = 0;
}
static void graph_counts_reset() {
  for (long long& v : g_capture_count) v = 0;
}
static void graph_counts_store(void* exec) {
  GraphCounts gc;
  gc.exec = exec;
  for (int i = 0; i < PROF_NCLASS; ++i) gc.n[i] = g_capture_count[i];
  g_graph_counts.push_back(gc);
}
static void graph_counts_add(void* exec) {
  if (!g_prof_on) return;
  for (const GraphCounts& gc : g_graph_counts)
    if (gc.exec == exec) {
      for (int i = 0; i < PROF_NCLASS; ++i) g_kprof[i].ngraph += gc.n[i];
      return;
    }
}
static void graph_counts_drop(void* exec) {
  for (size_t i = 0; i < g_graph_counts.size(); ++i)
    if (g_graph_counts[i].exec == exec) {
      g_graph_counts.erase(g_graph_counts.begin() + i);
      return;
    }
}
void kernel_profile_start(int every, double spmv_min_bytes) {
  for (KProf& k : g_kprof) kprof_clear(k);
  g_prof_on = true;
  g_prof_every = every < 1 ? 1 : every;
  g_prof_min_bytes = spmv_min_bytes;
}
void kernel_profile_stop() {
  g_prof_on = false;
  HIPCHK(hipStreamSynchronize(g_stream));
}
void kernel_profile_get(int cls, double* ms_sum, double* bytes_sum, double* flops_sum, long long* nsampled,
                        long long* nlaunch) {
  double ms = 0.0, by = 0.0, fl = 0.0;
  if (cls < 0 || cls >= PROF_NCLASS) throw std::runtime_error("kernel_profile_get: unknown class");
  KProf& k = g_kprof[cls];
  for (size_t i = 0; i < k.e1.size(); ++i) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, k.e0[i], k.e1[i]));
    ms += t;
    by += k.bytes[i];
    fl += k.flops[i];
  }
  if (ms_sum) *ms_sum = ms;
  if (bytes_sum) *bytes_sum = by;
  if (flops_sum) *flops_sum = fl;
  if (nsampled) *nsampled = (long long)k.e1.size();
  if (nlaunch) *nlaunch = k.nlaunch + k.ngraph;
}
void spmv_profile_start(int every, double min_bytes) { kernel_profile_start(every, min_bytes); }
bool spmv_profiling() { return g_prof_on; }
void spmv_profile_stop(double* ms_sum, double* bytes_sum, long long* nsampled, long long* nlaunch) {
  kernel_profile_stop();
  kernel_profile_get(PROF_SPMV, ms_sum, bytes_sum, nullptr, nsampled, nlaunch);
}

// ------------------------------------------------------------------------------- single-precision companion
// (col, val) of the slices at 6 bytes per entry for the V-cycle of the local solves: the preconditioner of an FP64 PCG
// needs its operator to a few digits only, and its passes over the fine and first coarse matrices are pure HBM streams.
// Arithmetic stays FP64 (values are widened on load), vectors stay FP64.
// Column bases of a slice: 16-bit offsets need the columns of a slice within 65 536 of a base.  One base (the slice's
// lowest column) is enough while a 64-row slice of a grid block spans less than that; a 64-row slice of a 187^3 block
// (one subdomain per GPU at 368^3 / 8) spans 2 x 187^2 + 64 = 70 002 columns.  Such a slice gets TWO bases: the entries
// k < ks of every row (the slice's first ks stored columns, 64 entries each) count from b0, the others from b1 -- rows
// hold their columns in ascending order, so the low neighbours sit in the first entries and the high ones in the last.
// ks is the largest split the first base allows (greedy: the second group is then as small as it can be), wave-uniform
// in the kernels that read it.  base4[s] = {b0, b1, ks, 0}; ks = INT_MAX when one base does.  `limit` is 65535
// (GENEO_LP_SPAN_MAX lowers it so that small test cases take the two-base and the 32-bit paths).
__device__ __forceinline__ int lp_wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int lp_wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
__global__ __launch_bounds__(256) void k_lp_base(int ns, const int64_t* __restrict__ sl_ptr, const int* __restrict__ sl_col,
                                                 int4* __restrict__ base, int* __restrict__ fail, int limit,
                                                 int* __restrict__ nsplit) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int l = threadIdx.x & 63;
  const int64_t a = sl_ptr[s];
  const int wd = (int)((sl_ptr[s + 1] - a) >> 6);
  int lo = 0x7fffffff, hi = 0;
  for (int k = 0; k < wd; ++k) {
    const int c = sl_col[a + 64 * (int64_t)k + l];
    lo = c < lo ? c : lo;
    hi = c > hi ? c : hi;
  }
  lo = lp_wave_min(lo);
  hi = lp_wave_max(hi);
  if (lo == 0x7fffffff) lo = 0;          // empty slice
  int b1 = lo, ks = 0x7fffffff;
  if (hi - lo > limit) {
    ks = wd;
    for (int k = 0; k < wd; ++k) {       // first entry index whose columns leave the first base's window
      const int mx = lp_wave_max(sl_col[a + 64 * (int64_t)k + l]);
      if (mx - lo > limit) { ks = k; break; }
    }
    int lo2 = 0x7fffffff, hi2 = 0;
    for (int k = ks; k < wd; ++k) {
      const int c = sl_col[a + 64 * (int64_t)k + l];
      lo2 = c < lo2 ? c : lo2;
      hi2 = c > hi2 ? c : hi2;
    }
    lo2 = lp_wave_min(lo2);
    hi2 = lp_wave_max(hi2);
    b1 = lo2;
    if (l == 0) {
      if (hi2 - lo2 > limit) atomicExch(fail, 1);
      else atomicAdd(nsplit, 1);
    }
  }
  if (l == 0) base[s] = int4{lo, b1, ks, 0};
}
__global__ __launch_bounds__(256) void k_lp_fill(int ns, const int64_t* __restrict__ sl_ptr, const int* __restrict__ sl_col,
                                                 const double* __restrict__ sl_val, const int4* __restrict__ base,
                                                 unsigned short* __restrict__ c16, float* __restrict__ v32) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int l = threadIdx.x & 63;
  const int4 b = c16 ? base[s] : int4{0, 0, 0x7fffffff, 0};
  const int64_t a = sl_ptr[s];
  int k = 0;
  for (int64_t e = a + l; e < sl_ptr[s + 1]; e += 64, ++k) {
    if (c16) c16[e] = (unsigned short)(sl_col[e] - (k < b.z ? b.x : b.y));
    v32[e] = (float)sl_val[e];
  }
}
bool csr_make_lp(Csr& a, const Csr* index_owner) {
  if (spmv_kind() != 1 || a.nslice == 0 || !a.sl_ptr || a.vec_lpr > 0 || a.nlong > 0 || a.lp_val) return a.lp_val != nullptr;
  if (index_owner) {
    if (!index_owner->lp_val || index_owner->sl_col != a.sl_col) return false;
    a.lp_col = index_owner->lp_col;
    a.lp_base = index_owner->lp_base;
  } else {
    const char* lim_env = getenv("GENEO_LP_SPAN_MAX");       // read per call: tests lower it for one set-up
    int limit = lim_env ? atoi(lim_env) : 65535;
    if (limit < 1 || limit > 65535) limit = 65535;
    a.lp_base = (int*)alloc(sizeof(int4) * (size_t)a.nslice);
    int* dfail = (int*)alloc(2 * sizeof(int));
    HIPCHK(hipMemsetAsync(dfail, 0, 2 * sizeof(int), g_stream));
    hipLaunchKernelGGL(k_lp_base, dim3((a.nslice + 3) / 4), dim3(256), 0, g_stream, a.nslice, a.sl_ptr, a.sl_col,
                       (int4*)a.lp_base, dfail, limit, dfail + 1);
    int hfail[2] = {0, 0};
    d2h(hfail, dfail, 2 * sizeof(int));
    dfree(dfail);
    if (hfail[0]) {          // a slice needs more than two bases: float values over the 32-bit columns (8 B per entry)
      dfree(a.lp_base);
      a.lp_base = nullptr;
    } else {
      a.lp_col = (unsigned short*)alloc(sizeof(unsigned short) * (size_t)std::max<int64_t>(1, a.sl_nnz));
    }
    if (getenv("GENEO_DEBUG") && (hfail[0] || hfail[1]))
      fprintf(stderr, "[lp] %d-row matrix, %d slices: %s (span limit %d)\n", a.n, a.nslice,
              hfail[0] ? "32-bit columns kept (a slice needs more than two bases)"
                       : (std::to_string(hfail[1]) + " slices with two column bases").c_str(), limit);
  }
  a.lp_val = (float*)alloc(sizeof(float) * (size_t)std::max<int64_t>(1, a.sl_nnz));
  hipLaunchKernelGGL(k_lp_fill, dim3((a.nslice + 3) / 4), dim3(256), 0, g_stream, a.nslice, a.sl_ptr, a.sl_col, a.sl_val,
                     (const int4*)a.lp_base, (index_owner || !a.lp_col) ? (unsigned short*)nullptr : a.lp_col, a.lp_val);
  return true;
}
// A slice of KW <= 8 entries per row in TWO dependent memory latencies: every (col, val) pair of the slice is requested
// before the first x gather and all gathers are in flight together (what k_spmv_sell_p8 does for the FP64 SpMV).  The
// 4-step loop of k_spmv_sell_lp sends a 7-wide slice -- every fine-level operator of the benchmark -- through one round of
// four and three one-at-a-time tail steps: eight dependent latencies per wave, and a wave holds one 64-row slice, so its
// run time IS that chain (0.46-0.54 of the HBM peak while the FP64 SpMV on the same pattern reaches
// 0.72).  The products are summed exactly as that loop sums them (KW / 4 rounds into four accumulators, the remainder
// into the first): results are bit-identical.
template <int KW, int EPI, typename COLT, bool NT>
__device__ __forceinline__ double lp_row_sum_fixed(const COLT* __restrict__ col, const float* __restrict__ val, int64_t e0,
                                                   const int4 cb, const double* __restrict__ xin,
                                                   const double* __restrict__ cs) {
  int c[KW];
  double v[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) {
    c[u] = (u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e0 + 64 * u) : col[e0 + 64 * u]);
    v[u] = (double)(NT ? __builtin_nontemporal_load(val + e0 + 64 * u) : val[e0 + 64 * u]);
  }
  double xv[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) xv[u] = (EPI == EPI_PRE && cs) ? xin[c[u]] * cs[c[u]] : xin[c[u]];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int NG = KW / 4;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += v[4 * g + u] * xv[4 * g + u];
#pragma unroll
  for (int k = 4 * NG; k < KW; ++k) acc[0] += v[k] * xv[k];
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// The same for slices of 9 .. 16 entries per row (the post-smoothing matrices M = P - w D^-1 A P of the V-cycle: 12 wide on
// the 7-point problem, three rounds of the 4-step loop = six dependent latencies): sixteen predicated steps (wd is
// wave-uniform: scalar branches, lanes past the width load nothing), all loads, then all gathers, then the products in the
// loop's order -- entry u < 4 (wd / 4) goes to accumulator u mod 4, the remainder to the first, in increasing u.
// (The same sequence for the workgroup-per-slice kernel -- each of its four waves owning every fourth entry of a row -- was
//  measured on one box, three runs each at 6.5 M rows: companion passes 82 us with the wave-per-slice forms alone, 92 us with
//  both, 92 us with neither (profiles/r04_lp_two_latency_ab.log): it costs what the narrow forms gain, and is not built.)
template <int EPI, typename COLT, bool NT>
__device__ __forceinline__ double lp_row_sum_p16(const COLT* __restrict__ col, const float* __restrict__ val, int64_t e0,
                                                 int wd, const int4 cb, const double* __restrict__ xin,
                                                 const double* __restrict__ cs) {
  int c[16];
  double v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    c[u] = cb.x;
    v[u] = 0.0;
    if (u < wd) {
      c[u] = (u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e0 + 64 * u) : col[e0 + 64 * u]);
      v[u] = (double)(NT ? __builtin_nontemporal_load(val + e0 + 64 * u) : val[e0 + 64 * u]);
    }
  }
  double xv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    xv[u] = 0.0;
    if (u < wd) xv[u] = (EPI == EPI_PRE && cs) ? xin[c[u]] * cs[c[u]] : xin[c[u]];
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int ng4 = (wd >> 2) << 2;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (u < ng4) acc[u & 3] += v[u] * xv[u];
    else if (u < wd) acc[0] += v[u] * xv[u];
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// WPS = 1: one wave per slice (four slices per workgroup), WPS = 4: one workgroup per slice (wide slices), as the FP64
// kernels k_spmv_sell_epi / k_spmv_sell_wide above; summation order fixed.
// NT: the once-read (col, val) stream of a LARGE companion is marked non-temporal, as in k_spmv_sell (the vectors of the
// cycle, not the matrix, are what the caches should keep); a hint only -- same loads, same arithmetic.
template <int EPI, int WPS, typename COLT, bool NT>
__global__ __launch_bounds__(256) void k_spmv_sell_lp(const int64_t* __restrict__ sl_ptr, int nslice, int n,
                                                      const COLT* __restrict__ col, const float* __restrict__ val,
                                                      const int* __restrict__ base, const double* __restrict__ x,
                                                      double* __restrict__ y, const double* __restrict__ b,
                                                      double* __restrict__ z, const double* __restrict__ dinv, double w,
                                                      const double* __restrict__ cs) {
  constexpr int UNR = 4;
  __shared__ double part[3][64];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  int s;
  if (WPS == 1) {
    const int nwb = (nslice + 3) >> 2;
    const int t = xcd_remap(blockIdx.x, nwb);
    s = 4 * t + wv;
    if (t >= nwb || s >= nslice) return;
  } else {
    s = xcd_remap(blockIdx.x, nslice);
    if (s >= nslice) return;
  }
  const int64_t a = sl_ptr[s], e1 = sl_ptr[s + 1];
  const int4 cb = lp_base_of(base, s);   // 16-bit columns are offsets from the slice's bases (k_lp_base)
  const double* __restrict__ xin = (EPI == EPI_PRE) ? b : x;
  // The epilogue's own operands do not depend on the product: their loads are issued HERE, in front of the matrix stream,
  // instead of after the last gather has come back (one dependent memory round trip less per wave; a wave holds one
  // 64-row slice, so its run time is a chain of round trips, not bandwidth).  Same arithmetic, same results.
  const int r = 64 * s + l;
  const bool rok = r < n && (WPS == 1 || wv == 0);
  double e_b = 0.0, e_z = 0.0, e_d = 0.0, e_x = 0.0;
  if (rok) {
    if (EPI == EPI_RES || EPI == EPI_JAC || EPI == EPI_POST || EPI == EPI_PRE) e_b = b[r];
    if (EPI == EPI_ADD || EPI == EPI_POST) e_z = z[r];
    if (EPI == EPI_JAC || EPI == EPI_POST || (EPI == EPI_PRE && z)) e_d = dinv[r];
    if (EPI == EPI_JAC) e_x = x[r];
  }
  constexpr int STEP = 64 * WPS;
  double acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u] = 0.0;
  int64_t e = a + (WPS == 1 ? 0 : 64 * wv) + l;
  int k = (WPS == 1 ? 0 : wv);            // entry index of e within its row
  const int wd = (int)((e1 - a) >> 6);    // entries per row of this slice (wave-uniform)
  double fixed_sum = 0.0;
  const bool fixed = (WPS == 1 && wd >= 1 && wd <= 16 && !g_lp_no_fixed);
  if (fixed && wd > 8) {
    fixed_sum = lp_row_sum_p16<EPI, COLT, NT>(col, val, e, wd, cb, xin, cs);
    e = e1;
  } else if (fixed) {
    switch (wd) {
      case 1: fixed_sum = lp_row_sum_fixed<1, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 2: fixed_sum = lp_row_sum_fixed<2, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 3: fixed_sum = lp_row_sum_fixed<3, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 4: fixed_sum = lp_row_sum_fixed<4, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 5: fixed_sum = lp_row_sum_fixed<5, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 6: fixed_sum = lp_row_sum_fixed<6, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      case 7: fixed_sum = lp_row_sum_fixed<7, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
      default: fixed_sum = lp_row_sum_fixed<8, EPI, COLT, NT>(col, val, e, cb, xin, cs); break;
    }
    e = e1;                               // nothing left for the general loops below
  }
  for (; e + STEP * (UNR - 1) < e1; e += STEP * UNR, k += WPS * UNR) {
    int c[UNR];
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      c[u] = (k + WPS * u < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e + STEP * u) : col[e + STEP * u]);
      v[u] = (double)(NT ? __builtin_nontemporal_load(val + e + STEP * u) : val[e + STEP * u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc[u] += v[u] * ((EPI == EPI_PRE && cs) ? xin[c[u]] * cs[c[u]] : xin[c[u]]);
  }
  for (; e < e1; e += STEP, k += WPS) {
    const int c0 = (k < cb.z ? cb.x : cb.y) + (int)(NT ? __builtin_nontemporal_load(col + e) : col[e]);
    const double v0 = (double)(NT ? __builtin_nontemporal_load(val + e) : val[e]);
    acc[0] += v0 * ((EPI == EPI_PRE && cs) ? xin[c0] * cs[c0] : xin[c0]);
  }
  double sum = fixed ? fixed_sum : (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (WPS == 4) {
    if (wv > 0) part[wv - 1][l] = sum;
    __syncthreads();
    if (wv > 0) return;
    sum = (sum + part[0][l]) + (part[1][l] + part[2][l]);
  }
  if (r >= n) return;
  if (EPI == EPI_NONE) {
    y[r] = sum;
  } else if (EPI == EPI_RES) {
    y[r] = e_b - sum;
  } else if (EPI == EPI_ADD) {
    y[r] = e_z + sum;
  } else if (EPI == EPI_JAC) {
    y[r] = e_x + w * e_d * (e_b - sum);
  } else if (EPI == EPI_POST) {
    y[r] = w * e_d * (e_z + e_b) + sum;
  } else {  // EPI_PRE
    if (z) z[r] = w * e_d * e_b;
    y[r] = e_b - w * sum;
  }
}
template <int EPI>
static void spmv_lp_launch(const Csr& a, const double* x, double* y, const double* b, double* z, const double* dinv, double w) {
  if (!a.lp_val) throw std::runtime_error("spmv_lp: the matrix has no single-precision companion");
  if (a.n == 0) return;
  // in-situ timer class PROF_LP: the V-cycle's passes over the LARGE operators (fine level: A diag(dinv), M, R; >= 32 MB
  // of companion entries).  Algorithmic bytes: 6 per stored entry (float value + 16-bit column; 8 when the slice spans
  // more than 65535 columns) + 4 per slice + 8 per row and vector the epilogue streams (x or b in, y out, plus z / r).
  constexpr int nvec = EPI == EPI_NONE ? 2 : (EPI == EPI_POST ? 5 : (EPI == EPI_JAC ? 4 : 3));
  const double ebytes = a.lp_col ? 6.0 : 8.0;
  ProfScope prof(PROF_LP, (double)a.sl_nnz * ebytes >= 32e6, (double)a.sl_nnz * ebytes + 4.0 * a.nslice + 8.0 * nvec * (double)a.n,
                 2.0 * (double)a.nnz);
  const double* cs = (EPI == EPI_PRE && !a.col_scaled) ? dinv : nullptr;
  const int per = (a.nslice + 7) / 8;
  const int perw = ((a.nslice + 3) / 4 + 7) / 8;
  static const bool nt_off = getenv("GENEO_LP_NO_NT") != nullptr;
  const bool nt = !nt_off && (double)a.sl_nnz * ebytes >= 24e6;
#define LP_LAUNCH2(WPS, COLT, colp, grid, NT)                                                                          \
  hipLaunchKernelGGL((k_spmv_sell_lp<EPI, WPS, COLT, NT>), dim3(grid), dim3(256), 0, g_stream, a.sl_ptr, a.nslice, a.n, colp, \
                     a.lp_val, a.lp_base, x, y, b, z, dinv, w, cs)
#define LP_LAUNCH(WPS, COLT, colp, grid)          \
  do {                                            \
    if (nt) LP_LAUNCH2(WPS, COLT, colp, grid, true); \
    else LP_LAUNCH2(WPS, COLT, colp, grid, false);   \
  } while (0)
  if (sell_wide(a)) {
    if (a.lp_col) LP_LAUNCH(4, unsigned short, a.lp_col, per * 8);
    else LP_LAUNCH(4, int, a.sl_col, per * 8);
  } else {
    if (a.lp_col) LP_LAUNCH(1, unsigned short, a.lp_col, perw * 8);
    else LP_LAUNCH(1, int, a.sl_col, perw * 8);
  }
#undef LP_LAUNCH
#undef LP_LAUNCH2
}
void spmv_lp(const Csr& a, const double* x, double* y) { spmv_lp_launch<EPI_NONE>(a, x, y, nullptr, nullptr, nullptr, 0.0); }
void spmv_fused_lp(const Csr& a, int epi, const double* x, double* y, const double* b, double* z, const double* dinv, double w) {
  switch (epi) {
    case EPI_RES: spmv_lp_launch<EPI_RES>(a, x, y, b, z, dinv, w); break;
    case EPI_ADD: spmv_lp_launch<EPI_ADD>(a, x, y, b, z, dinv, w); break;
    case EPI_JAC: spmv_lp_launch<EPI_JAC>(a, x, y, b, z, dinv, w); break;
    case EPI_POST: spmv_lp_launch<EPI_POST>(a, x, y, b, z, dinv, w); break;
    case EPI_PRE: spmv_lp_launch<EPI_PRE>(a, x, y, b, z, dinv, w); break;
    default: throw std::runtime_error("spmv_fused_lp: unknown epilogue");
  }
}

// Lanes-per-row CSR kernel for long / ragged rows, with the same epilogues: LPR lanes stride over one row
// (coalesced col/val reads), partial sums are combined by a fixed butterfly, lane 0 writes.
template <int LPR, int EPI>
__global__ __launch_bounds__(256) void k_spmv_vec(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ b,
                                                  double* __restrict__ z, const double* __restrict__ dinv, double w) {
  constexpr int RPB = 256 / LPR;
  const int lane = threadIdx.x % LPR;
  const int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = r < n;
  double s = 0.0;
  if (live) {
    const int k1 = rowptr[r + 1];
    for (int k = rowptr[r] + lane; k < k1; k += LPR) {
      const int c = col[k];
      s += val[k] * (EPI == EPI_PRE ? b[c] * dinv[c] : x[c]);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
  if (!live || lane != 0) return;
  if (EPI == EPI_NONE) {
    y[r] = s;
  } else if (EPI == EPI_RES) {
    y[r] = b[r] - s;
  } else if (EPI == EPI_ADD) {
    y[r] = z[r] + s;
  } else if (EPI == EPI_JAC) {
    y[r] = x[r] + w * dinv[r] * (b[r] - s);
  } else if (EPI == EPI_POST) {
    y[r] = w * dinv[r] * (z[r] + b[r]) + s;
  } else {
    const double bb = b[r];
    if (z) z[r] = w * dinv[r] * bb;
    y[r] = bb - w * s;
  }
}
template <int EPI>
static void spmv_vec_launch(const Csr& a, const double* x, double* y, const double* b, double* z, const double* dinv,
                            double w) {
#define VEC_LAUNCH(L)                                                                                              \
  hipLaunchKernelGGL((k_spmv_vec<L, EPI>), dim3(grid1d(a.n, 256 / L)), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, \
                     a.val, x, y, b, z, dinv, w)
  if (a.vec_lpr <= 16) VEC_LAUNCH(16);
  else if (a.vec_lpr <= 32) VEC_LAUNCH(32);
  else VEC_LAUNCH(64);
#undef VEC_LAUNCH
}
// rows excluded from the slices (longer than SELL_LONG): one workgroup per row
__global__ __launch_bounds__(256) void k_spmv_long(const int* __restrict__ rows, const int* __restrict__ rowptr,
                                                   const int* __restrict__ col, const double* __restrict__ val,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double sm[4];
  const int r = rows[blockIdx.x];
  double s = 0.0;
  for (int k = rowptr[r] + threadIdx.x; k < rowptr[r + 1]; k += 256) s += val[k] * x[col[k]];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) y[r] = s;
}

static inline bool sell_nt(const Csr& a) { return (double)a.sl_nnz * 12.0 > 48e6; }
static int g_sell_variant = 0;  // tuning variants of the sliced kernel (kind = 1 + 10 * variant)
static int g_spmv_kind = -1;  // 0 = LDS row blocks, 1 = 64-row slices
int spmv_kind() {
  if (g_spmv_kind < 0) {
    const char* e = getenv("GENEO_SPMV");
    g_spmv_kind = (e && std::string(e) == "lds") ? 0 : 1;
    const char* v = getenv("GENEO_SELL_VARIANT");
    if (v) g_sell_variant = atoi(v);
  }
  return g_spmv_kind;
}
const char* spmv_kernel_name() { return spmv_kind() == 0 ? "k_spmv_lds" : "k_spmv_sell"; }
void set_spmv_kind(int kind) {     // kind = layout + 10 * variant + 100 * (1: ragged matrices keep the slices too)
  g_spmv_kind = (kind % 10) ? 1 : 0;
  g_sell_variant = (kind / 10) % 10;
  g_force_slices = (kind / 100) % 10 == 1;
}

void spmv(const Csr& a, const double* x, double* y) {
  if (a.n == 0) return;
  const int per = (a.nblk + 7) / 8;
  // algorithmic bytes: nnz*(8+4) + (n+1)*4 + n*8 (x once) + n*8 (y)
  const double abytes = (double)a.nnz * 12.0 + ((double)a.n + 1.0) * 4.0 + (double)a.n * 16.0;
  ProfScope prof(PROF_SPMV, a.fine || (g_prof_min_bytes > 0.0 && abytes >= g_prof_min_bytes), abytes, 2.0 * (double)a.nnz);
  if (spmv_kind() == 0) {
    if (!a.rowblk) throw std::runtime_error("spmv: matrix was uploaded without LDS row blocks (GENEO_SPMV=lds at upload)");
    hipLaunchKernelGGL(k_spmv_lds, dim3(per * 8), dim3(256), 0, g_stream, a.rowblk, a.nblk, a.rowptr,
                       a.col, a.val, x, y);
  } else if (a.vec_lpr > 0) {
    spmv_vec_launch<EPI_NONE>(a, x, y, nullptr, nullptr, nullptr, 0.0);
  } else if (sell_wide(a)) {
    spmv_wide_launch<EPI_NONE>(a, x, y, nullptr, nullptr, nullptr, 0.0);
  } else {
    const int nwb = (a.nslice + 3) / 4;
    const int perw = (nwb + 7) / 8;
#define SELL_LAUNCH(U, N)                                                                                  \
  hipLaunchKernelGGL((k_spmv_sell<U, N>), dim3(perw * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice, a.n, \
                     a.sl_col, a.sl_val, x, y)
    // default policy: 4 k-steps in flight; the once-read (col,val) stream of a large matrix is marked
    // non-temporal so that it does not displace x and the multigrid vectors from L2 / Infinity Cache.
    // Measured in situ (126^3 bench, 190 MB matrix, every fine-level launch sampled): 38.5 us with NT,
    // 42.2 us without; back-to-back re-reads of the same matrix (a micro-benchmark, not the solver's
    // access pattern) prefer the cached stream, which is why small (coarse-level) matrices keep it.
    int variant = g_sell_variant;
    if ((variant == 0 || variant >= 5) && a.lp_col && a.lp_base && sell_nt(a)) {   // 16-bit column offsets are available: 10 B per entry
      // Default: 4 steps in flight + one-at-a-time tail (the same accumulation order as the 32-bit-column kernel: the two
      // paths agree to the bit).  GENEO_SELL_VARIANT=5 / 7: the predicated eight-step form (round-3 experiment: 20 % faster
      // back to back on an evicted cache, no difference in situ -- 5037 vs 5017 GB/s -- and a different rounding).
      if (variant == 0 || variant == 6)
        hipLaunchKernelGGL((k_spmv_sell<4, true, unsigned short>), dim3(perw * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice,
                           a.n, a.lp_col, a.sl_val, x, y, a.lp_base);
      else if (variant == 7)     // predicated form without the non-temporal hint
        hipLaunchKernelGGL((k_spmv_sell_p8<false, unsigned short>), dim3(perw * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice,
                           a.n, a.lp_col, a.sl_val, x, y, a.lp_base);
      else
        hipLaunchKernelGGL((k_spmv_sell_p8<true, unsigned short>), dim3(perw * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice,
                           a.n, a.lp_col, a.sl_val, x, y, a.lp_base);
      if (a.nlong > 0)
        hipLaunchKernelGGL(k_spmv_long, dim3(a.nlong), dim3(256), 0, g_stream, a.long_rows, a.rowptr, a.col, a.val, x, y);
      return;
    }
    if (variant == 0) variant = sell_nt(a) ? 3 : 2;
    switch (variant) {
      case 1: SELL_LAUNCH(2, true); break;
      case 2: SELL_LAUNCH(4, false); break;
      case 3: SELL_LAUNCH(4, true); break;
      case 4: SELL_LAUNCH(7, true); break;
      default: SELL_LAUNCH(2, false); break;
    }
#undef SELL_LAUNCH
    if (a.nlong > 0)
      hipLaunchKernelGGL(k_spmv_long, dim3(a.nlong), dim3(256), 0, g_stream, a.long_rows, a.rowptr, a.col, a.val,
                         x, y);
  }
}

// =============================================================================== CSR SpMM
// LPR lanes cooperate on one row.  The row's (col,val) pairs are fetched LPR at a time with ONE
// coalesced load per lane group and broadcast with shuffles, so the X-row loads of a chunk carry no
// dependent global load in front of them and overlap; lane j owns output columns j, j+LPR, ...
template <int LPR, int EPI = 0>
__global__ __launch_bounds__(256) void k_spmm(int n, const int* __restrict__ rowptr,
                                              const int* __restrict__ col, const double* __restrict__ val,
                                              const double* __restrict__ X, int ldx, double* __restrict__ Y,
                                              int ldy, int m, const double* __restrict__ pre,
                                              const double* __restrict__ post, const double* __restrict__ B = nullptr,
                                              int ldb = 0, double* __restrict__ Z = nullptr, int ldz = 0,
                                              const double* __restrict__ dinv = nullptr, double w = 0.0) {
  constexpr int RPB = 256 / LPR;
  const int lane = threadIdx.x % LPR;
  const int rloc = threadIdx.x / LPR;
  const int nrounds = (n + gridDim.x * RPB - 1) / (gridDim.x * RPB);
  for (int rd = 0; rd < nrounds; ++rd) {
    const int64_t r = ((int64_t)rd * gridDim.x + blockIdx.x) * RPB + rloc;
    const bool live = r < n;   // keep every lane in the shuffles
    const int a = live ? rowptr[r] : 0, b = live ? rowptr[r + 1] : 0;
    for (int j0 = 0; j0 < m; j0 += LPR) {
      const int j = j0 + lane;
      double s0 = 0.0, s1 = 0.0;
      for (int base = a; base < b; base += LPR) {
        int myc = 0;
        double myv = 0.0;
        if (base + lane < b) {
          myc = col[base + lane];
          myv = val[base + lane];
          if (pre) myv *= pre[myc];
        }
        const int cnt = (b - base < LPR) ? b - base : LPR;
        int k = 0;
        // eight X rows in flight for the long rows of the restriction operators (100 entries per row, a few thousand
        // rows: the launch is one latency chain per row); the products are added in the order of the two-at-a-time
        // loop below, so the sums are bit-identical to it
        for (; k + 7 < cnt; k += 8) {
          double xv[8], vv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int cu = __shfl(myc, k + u, LPR);
            vv[u] = __shfl(myv, k + u, LPR);
            xv[u] = (j < m) ? X[(int64_t)cu * ldx + j] : 0.0;
          }
          if (j < m) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
              s0 += vv[u] * xv[u];
              s1 += vv[u + 1] * xv[u + 1];
            }
          }
        }
        for (; k + 1 < cnt; k += 2) {
          const int c0 = __shfl(myc, k, LPR), c1 = __shfl(myc, k + 1, LPR);
          const double v0 = __shfl(myv, k, LPR), v1 = __shfl(myv, k + 1, LPR);
          if (j < m) {
            s0 += v0 * X[(int64_t)c0 * ldx + j];
            s1 += v1 * X[(int64_t)c1 * ldx + j];
          }
        }
        if (k < cnt) {
          const int c0 = __shfl(myc, k, LPR);
          const double v0 = __shfl(myv, k, LPR);
          if (j < m) s0 += v0 * X[(int64_t)c0 * ldx + j];
        }
      }
      if (live && j < m) {
        double sacc = s0 + s1;
        if (post) sacc *= post[r];
        if (EPI == EPI_NONE) {
          Y[r * ldy + j] = sacc;
        } else if (EPI == EPI_RES) {
          Y[r * ldy + j] = B[r * ldb + j] - sacc;
        } else if (EPI == EPI_ADD) {
          Y[r * ldy + j] = Z[r * ldz + j] + sacc;
        } else if (EPI == EPI_JAC) {
          Y[r * ldy + j] = X[r * ldx + j] + w * dinv[r] * (B[r * ldb + j] - sacc);
        } else if (EPI == EPI_POST) {
          Y[r * ldy + j] = w * dinv[r] * (Z[r * ldz + j] + B[r * ldb + j]) + sacc;
        } else {  // EPI_PRE: X = B, pre = dinv
          const double bb = B[r * ldb + j];
          if (Z) Z[r * ldz + j] = w * dinv[r] * bb;
          Y[r * ldy + j] = bb - w * sacc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------- sliced SpMM
// Y = post .* (A (pre .* X)) on the 64-row slices, m = 2 LG columns.  LG lanes own one row (16 bytes = two columns per
// lane), so ONE wave-wide dwordx4 load gathers the k-th neighbour rows of 64 / LG consecutive rows (m = 32: four rows,
// 1 KiB per instruction) and the Y store of those rows is one contiguous 1 KiB segment.  The slice's (col, val) entries
// are read ONCE, coalesced and non-temporal (lane = row of the slice, as the SpMV does), staged in a wave-private LDS
// tile and handed to the row groups by broadcast reads: no dependent global load in front of the X gathers.
// The gathers of a step group (U steps x KW entries, e.g. 4 x 7 = 28 KiB per wave) are all issued before the first
// product: the body is instantiated per entry count KW so that no branch sits between them (the first version had a
// branch per entry and drained its 4 loads each time: 0.35 ms; see DESIGN.md section 7 for the history).
//
// Traversal: persistent grid; the hardware deals workgroups round-robin over the 8 XCDs, workgroup b belongs to XCD
// group b & 7 and that group walks ITS OWN contiguous range of the slice schedule (xptr[g] .. xptr[g + 1]) with all its
// waves side by side.  The rows of X a group touches at any time are the few hundred slices its waves hold plus their
// neighbours: that window, not the whole block, is what the XCD's private 4 MiB L2 has to keep.  (Which XCD a group
// lands on does not matter for correctness.)  sched == nullptr: slices in natural order.
typedef double spmm_d2 __attribute__((ext_vector_type(2)));
template <int LG, int U, int KW>
__device__ __forceinline__ void spmm_sell_accum(const int* mc, const double* mv, int rl0, const double* __restrict__ Xq,
                                                int ldx, spmm_d2* acc) {
  constexpr int RS = 64 / LG;
  spmm_d2 x[KW][U];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = mc[k * 64 + rl0 + u * RS];
      x[k][u] = *reinterpret_cast<const spmm_d2*>(Xq + (int64_t)c * ldx);
    }
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += mv[k * 64 + rl0 + u * RS] * x[k][u];
}

template <int LG, int EPI, int U>
__global__ __launch_bounds__(256) void k_spmm_sell(const int64_t* __restrict__ sl_ptr, const int* __restrict__ sl_col,
                                                   const double* __restrict__ sl_val, int n,
                                                   const int* __restrict__ sched, const int* __restrict__ xptr,
                                                   const double* __restrict__ X, int ldx, double* __restrict__ Y, int ldy,
                                                   const double* __restrict__ pre, const double* __restrict__ post,
                                                   const double* __restrict__ B, int ldb, double* __restrict__ Z, int ldz,
                                                   const double* __restrict__ dinv, double w) {
  typedef spmm_d2 d2;
  constexpr int KC = 8;              // entries per row staged per chunk
  constexpr int RS = 64 / LG;        // rows per wave-wide load
  constexpr int NSTEP = 64 / RS;     // steps per slice
  static_assert(NSTEP % U == 0, "steps in flight must divide the steps of a slice");
  __shared__ int lc[4][KC * 64];
  __shared__ double lv[4][KC * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / LG, q = lane % LG;
  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  int* mc = lc[wave];
  double* mv = lv[wave];
  const double* Xq = X + 2 * q;
  for (int it = xptr[xg] + slot * 4 + wave; it < xptr[xg + 1]; it += wpx * 4) {
    const int s = sched ? sched[it] : it;
    const int64_t base = sl_ptr[s];
    const int wd = (int)((sl_ptr[s + 1] - base) >> 6);
    const int nchunk = (wd + KC - 1) / KC;
    for (int g = 0; g < NSTEP / U; ++g) {
      d2 acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = d2{0.0, 0.0};
      for (int ch = 0; ch < nchunk; ++ch) {
        const int kc = (wd - ch * KC < KC) ? wd - ch * KC : KC;
        if (nchunk > 1 || g == 0) {          // narrow slices (the fine-level stencils) are staged once
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int k = 0; k < kc; ++k) {
            const int64_t e = base + (int64_t)64 * (ch * KC + k) + lane;
            const int c = __builtin_nontemporal_load(sl_col + e);
            double v = __builtin_nontemporal_load(sl_val + e);
            if (pre) v *= pre[c];
            mc[k * 64 + lane] = c;
            mv[k * 64 + lane] = v;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const int rl0 = g * U * RS + grp;
        switch (kc) {                         // wave-uniform
          case 8: spmm_sell_accum<LG, U, 8>(mc, mv, rl0, Xq, ldx, acc); break;
          case 7: spmm_sell_accum<LG, U, 7>(mc, mv, rl0, Xq, ldx, acc); break;
          case 6: spmm_sell_accum<LG, U, 6>(mc, mv, rl0, Xq, ldx, acc); break;
          case 5: spmm_sell_accum<LG, U, 5>(mc, mv, rl0, Xq, ldx, acc); break;
          case 4: spmm_sell_accum<LG, U, 4>(mc, mv, rl0, Xq, ldx, acc); break;
          case 3: spmm_sell_accum<LG, U, 3>(mc, mv, rl0, Xq, ldx, acc); break;
          case 2: spmm_sell_accum<LG, U, 2>(mc, mv, rl0, Xq, ldx, acc); break;
          case 1: spmm_sell_accum<LG, U, 1>(mc, mv, rl0, Xq, ldx, acc); break;
          default: break;
        }
      }
      // the epilogue's own rows (loading them in front of the gathers was measured: same time, 24 more registers)
      d2 e_b[U], e_z[U], e_x[U];
      double e_d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = (int64_t)64 * s + (g * U + u) * RS + grp;
        e_b[u] = e_z[u] = e_x[u] = d2{0.0, 0.0};
        e_d[u] = 0.0;
        if (EPI != EPI_NONE && r < n) {
          if (EPI == EPI_RES || EPI == EPI_JAC || EPI == EPI_POST || EPI == EPI_PRE) e_b[u] = *reinterpret_cast<const d2*>(B + r * ldb + 2 * q);
          if (EPI == EPI_ADD || EPI == EPI_POST) e_z[u] = *reinterpret_cast<const d2*>(Z + r * ldz + 2 * q);
          if (EPI == EPI_JAC) e_x[u] = *reinterpret_cast<const d2*>(X + r * ldx + 2 * q);
          if (EPI == EPI_JAC || EPI == EPI_POST || (EPI == EPI_PRE && Z)) e_d[u] = dinv[r];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = (int64_t)64 * s + (g * U + u) * RS + grp;
        if (r >= n) continue;
        d2 a2 = acc[u];
        if (post) a2 *= post[r];
        d2 out;
        if (EPI == EPI_NONE) {
          out = a2;
        } else if (EPI == EPI_RES) {
          out = e_b[u] - a2;
        } else if (EPI == EPI_ADD) {
          out = e_z[u] + a2;
        } else if (EPI == EPI_JAC) {
          out = e_x[u] + (w * e_d[u]) * (e_b[u] - a2);
        } else if (EPI == EPI_POST) {
          out = (w * e_d[u]) * (e_z[u] + e_b[u]) + a2;
        } else {  // EPI_PRE: X = B, pre = dinv
          if (Z) *reinterpret_cast<d2*>(Z + r * ldz + 2 * q) = (w * e_d[u]) * e_b[u];
          out = e_b[u] - w * a2;
        }
        __builtin_nontemporal_store(out, reinterpret_cast<d2*>(Y + r * ldy + 2 * q));
      }
    }
  }
}

// Wide slices (see k_spmv_sell_wide): one workgroup per slice.  The four waves stage a chunk of KC entries per row
// together (each entry of the slice is read ONCE -- the wave-per-slice kernel above re-stages a wide slice for every row
// group), then wave w accumulates the rows 16 w .. 16 w + 15 of the slice: 64 / LG rows per wave-wide gather, all its
// row groups and 8 entries in flight at a time.
template <int LG, int EPI>
__global__ __launch_bounds__(256) void k_spmm_sell_wide(const int64_t* __restrict__ sl_ptr, const int* __restrict__ sl_col,
                                                        const double* __restrict__ sl_val, int n, int nslice,
                                                        const double* __restrict__ X, int ldx, double* __restrict__ Y, int ldy,
                                                        const double* __restrict__ pre, const double* __restrict__ post,
                                                        const double* __restrict__ B, int ldb, double* __restrict__ Z, int ldz,
                                                        const double* __restrict__ dinv, double w, int nt /* large matrix: (col, val) and Y streamed non-temporally */) {
  typedef spmm_d2 d2;
  constexpr int KC = 16;             // entries per row staged per chunk
  constexpr int RS = 64 / LG;        // rows per wave-wide load
  constexpr int U = 16 / RS;         // row groups of one wave (its 16 rows)
  __shared__ int lc[KC * 64];
  __shared__ double lv[KC * 64];
  const int s = xcd_remap(blockIdx.x, nslice);
  if (s >= nslice) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / LG, q = lane % LG;
  const double* Xq = X + 2 * q;
  const int64_t base = sl_ptr[s];
  const int wd = (int)((sl_ptr[s + 1] - base) >> 6);
  d2 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = d2{0.0, 0.0};
  const int rl0 = 16 * wave + grp;
  // The epilogue's own rows do not depend on the product.  For the zero-guess sweep (PRE: one operand) they are loaded in
  // front of the gathers (first coarse level: 209 -> 191 us); for JAC and POST the three operands of four row groups cost
  // 110 registers and half the resident waves (POST: 578 -> 800 us): every other epilogue loads them behind the
  // accumulation as before.
  constexpr bool EARLY = (EPI == EPI_PRE);
  d2 e_b[U], e_z[U], e_x[U];
  double e_d[U];
  auto load_epi = [&]() {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = (int64_t)64 * s + 16 * wave + u * RS + grp;
      e_b[u] = e_z[u] = e_x[u] = d2{0.0, 0.0};
      e_d[u] = 0.0;
      if (EPI != EPI_NONE && r < n) {
        if (EPI == EPI_RES || EPI == EPI_JAC || EPI == EPI_POST || EPI == EPI_PRE) e_b[u] = *reinterpret_cast<const d2*>(B + r * ldb + 2 * q);
        if (EPI == EPI_ADD || EPI == EPI_POST) e_z[u] = *reinterpret_cast<const d2*>(Z + r * ldz + 2 * q);
        if (EPI == EPI_JAC) e_x[u] = *reinterpret_cast<const d2*>(X + r * ldx + 2 * q);
        if (EPI == EPI_JAC || EPI == EPI_POST || (EPI == EPI_PRE && Z)) e_d[u] = dinv[r];
      }
    }
  };
  if (EARLY) load_epi();
  for (int k0 = 0; k0 < wd; k0 += KC) {
    const int kc = (wd - k0 < KC) ? wd - k0 : KC;
    if (k0 > 0) __syncthreads();
    for (int k = wave; k < kc; k += 4) {
      const int64_t e = base + (int64_t)64 * (k0 + k) + lane;
      const int c = nt ? __builtin_nontemporal_load(sl_col + e) : sl_col[e];
      double v = nt ? __builtin_nontemporal_load(sl_val + e) : sl_val[e];
      if (pre) v *= pre[c];
      lc[k * 64 + lane] = c;
      lv[k * 64 + lane] = v;
    }
    __syncthreads();
    for (int kb = 0; kb < kc; kb += 8) {
      const int kw = (kc - kb < 8) ? kc - kb : 8;
      const int* mc = lc + kb * 64;
      const double* mv = lv + kb * 64;
      switch (kw) {                         // workgroup-uniform
        case 8: spmm_sell_accum<LG, U, 8>(mc, mv, rl0, Xq, ldx, acc); break;
        case 7: spmm_sell_accum<LG, U, 7>(mc, mv, rl0, Xq, ldx, acc); break;
        case 6: spmm_sell_accum<LG, U, 6>(mc, mv, rl0, Xq, ldx, acc); break;
        case 5: spmm_sell_accum<LG, U, 5>(mc, mv, rl0, Xq, ldx, acc); break;
        case 4: spmm_sell_accum<LG, U, 4>(mc, mv, rl0, Xq, ldx, acc); break;
        case 3: spmm_sell_accum<LG, U, 3>(mc, mv, rl0, Xq, ldx, acc); break;
        case 2: spmm_sell_accum<LG, U, 2>(mc, mv, rl0, Xq, ldx, acc); break;
        case 1: spmm_sell_accum<LG, U, 1>(mc, mv, rl0, Xq, ldx, acc); break;
        default: break;
      }
    }
  }
  if (!EARLY) load_epi();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t r = (int64_t)64 * s + 16 * wave + u * RS + grp;
    if (r >= n) continue;
    d2 a2 = acc[u];
    if (post) a2 *= post[r];
    d2 out;
    if (EPI == EPI_NONE) {
      out = a2;
    } else if (EPI == EPI_RES) {
      out = e_b[u] - a2;
    } else if (EPI == EPI_ADD) {
      out = e_z[u] + a2;
    } else if (EPI == EPI_JAC) {
      out = e_x[u] + (w * e_d[u]) * (e_b[u] - a2);
    } else if (EPI == EPI_POST) {
      out = (w * e_d[u]) * (e_z[u] + e_b[u]) + a2;
    } else {  // EPI_PRE: X = B, pre = dinv
      if (Z) *reinterpret_cast<d2*>(Z + r * ldz + 2 * q) = (w * e_d[u]) * e_b[u];
      out = e_b[u] - w * a2;
    }
    if (nt) __builtin_nontemporal_store(out, reinterpret_cast<d2*>(Y + r * ldy + 2 * q));
    else *reinterpret_cast<d2*>(Y + r * ldy + 2 * q) = out;
  }
}

static int g_spmm_wpx = -1;      // workgroups per XCD group of the sliced SpMM (GENEO_SPMM_WPX; 0 = old CSR kernel)
static int g_spmm_u = 2;         // steps in flight per wave (GENEO_SPMM_U: 1, 2 or 4)
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// true when the launch was taken by the sliced kernel
template <int EPI>
static bool spmm_sell_launch(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                             const double* post, const double* B, int ldb, double* Z, int ldz, const double* dinv,
                             double w) {
  if (g_spmm_wpx < 0) {
    const char* e = getenv("GENEO_SPMM_WPX");
    const char* u = getenv("GENEO_SPMM_U");
    if (u) g_spmm_u = atoi(u);
    // 126^3 fine level, time / PMC traffic over algorithmic per (steps in flight U, workgroups per XCD group):
    // (2, 64) 0.337 ms / 1.13   (1, 128) 0.332 / 1.31   (4, 64) 0.334 / 1.35   (2, 128) 0.356 / 1.58   (4, 128) 0.35 / 1.60:
    // the time is flat, the over-fetch grows with the gathers a group holds in flight (its L2 window)
    g_spmm_wpx = e ? atoi(e) : 64;
  }
  if (g_spmm_wpx == 0 || spmv_kind() != 1 || a.vec_lpr > 0 || a.nlong > 0 || !a.sl_ptr || !a.xcd_ptr) return false;
  if (m != 16 && m != 32 && m != 64) return false;
  if ((ldx | ldy | ldb | ldz) & 1) return false;
  if (!aligned16(X) || !aligned16(Y) || !aligned16(B) || !aligned16(Z)) return false;
  if (sell_wide_mm(a)) {
    const int per = (a.nslice + 7) / 8;
#define SELLW(L)                                                                                                           \
  hipLaunchKernelGGL((k_spmm_sell_wide<L, EPI>), dim3(per * 8), dim3(256), 0, g_stream, a.sl_ptr, a.sl_col, a.sl_val, a.n, \
                     a.nslice, X, ldx, Y, ldy, pre, post, B, ldb, Z, ldz, dinv, w, wide_nt)
    static const bool wide_nt_off = getenv("GENEO_SPMM_WIDE_NO_NT") != nullptr;
    const int wide_nt = (!wide_nt_off && sell_nt(a)) ? 1 : 0;
    if (m == 16) { SELLW(8); return true; }
    if (m == 32) { SELLW(16); return true; }
#undef SELLW
  }
  // small matrices: no more workgroups than there is work for (4 slices per workgroup and pass)
  int wpx = g_spmm_wpx;
  const int need = (a.nslice + 31) / 32;
  if (wpx > need) wpx = need < 1 ? 1 : need;
#define SELLMM2(L, UU)                                                                                                  \
  hipLaunchKernelGGL((k_spmm_sell<L, EPI, UU>), dim3(8 * wpx), dim3(256), 0, g_stream, a.sl_ptr, a.sl_col, a.sl_val, a.n, \
                     a.sched, a.xcd_ptr, X, ldx, Y, ldy, pre, post, B, ldb, Z, ldz, dinv, w)
#define SELLMM(L)                  \
  do {                             \
    if (g_spmm_u >= 4) SELLMM2(L, 4); \
    else if (g_spmm_u == 2) SELLMM2(L, 2); \
    else SELLMM2(L, 1);            \
  } while (0)
  if (m == 16) SELLMM(8);
  else if (m == 32) SELLMM(16);
  else SELLMM(32);
#undef SELLMM2
#undef SELLMM
  return true;
}

// ------------------------------------------------------------------------------- two operators, one pass over X
// Y1 = A1 X and Y2 = A2 X where A1 and A2 share ONE sliced pattern (LOBPCG: A_Neu laid out on A_Dir's pattern -- which
// contains it -- and D A_Dir D): the X rows, whose gathers are what the sliced SpMM spends its time on, are fetched once
// for both products.  Same traversal, staging and step groups as k_spmm_sell<LG, EPI_NONE, U>; per entry one more value
// in LDS and one more FMA pair.  Results are bit-identical to the two separate products: the extra pattern entries of
// A_Neu carry explicit zeros, which change no sum.
template <int LG, int U, int KW>
__device__ __forceinline__ void spmm_sell_accum2(const int* mc, const double* mv1, const double* mv2, int rl0,
                                                 const double* __restrict__ Xq, int ldx, spmm_d2* acc1, spmm_d2* acc2) {
  constexpr int RS = 64 / LG;
  spmm_d2 x[KW][U];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = mc[k * 64 + rl0 + u * RS];
      x[k][u] = *reinterpret_cast<const spmm_d2*>(Xq + (int64_t)c * ldx);
    }
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc1[u] += mv1[k * 64 + rl0 + u * RS] * x[k][u];
      acc2[u] += mv2[k * 64 + rl0 + u * RS] * x[k][u];
    }
}
// RES = 1: instead of the two products the kernel writes the LOBPCG residual block of X,
//     Y1[r, j] = mask[s, j] (A1 X - A2 X diag(lam_s))[r, j]      (s = subdomain of row r, found in suboff[0 .. nsub]),
// so that neither product travels through HBM (Y2 is not touched).
template <int LG, int U, int RES>
__global__ __launch_bounds__(256) void k_spmm_sell_dual(const int64_t* __restrict__ sl_ptr, const int* __restrict__ sl_col,
                                                        const double* __restrict__ v1, const double* __restrict__ v2, int n,
                                                        const int* __restrict__ sched, const int* __restrict__ xptr,
                                                        const double* __restrict__ X, int ldx, double* __restrict__ Y1,
                                                        double* __restrict__ Y2, int ldy, const int* __restrict__ suboff,
                                                        int nsub, const double* __restrict__ lam,
                                                        const double* __restrict__ mask) {
  typedef spmm_d2 d2;
  constexpr int KC = 8;
  constexpr int RS = 64 / LG;
  constexpr int NSTEP = 64 / RS;
  static_assert(NSTEP % U == 0, "steps in flight must divide the steps of a slice");
  __shared__ int lc[4][KC * 64];
  __shared__ double lv1[4][KC * 64];
  __shared__ double lv2[4][KC * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / LG, q = lane % LG;
  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  int* mc = lc[wave];
  double* m1 = lv1[wave];
  double* m2 = lv2[wave];
  const double* Xq = X + 2 * q;
  for (int it = xptr[xg] + slot * 4 + wave; it < xptr[xg + 1]; it += wpx * 4) {
    const int s = sched ? sched[it] : it;
    const int64_t base = sl_ptr[s];
    const int wd = (int)((sl_ptr[s + 1] - base) >> 6);
    const int nchunk = (wd + KC - 1) / KC;
    // RES: the subdomain of the slice's first and last row, once per slice (a per-row search would put three dependent
    // loads in front of every store); lam / mask of that subdomain ride in registers unless the slice straddles a boundary
    int sd0 = 0, sd1 = 0;
    d2 lm0 = d2{0.0, 0.0}, mk0 = d2{0.0, 0.0};
    if (RES) {
      const int rfirst = 64 * s, rlast = (64 * s + 63 < n) ? 64 * s + 63 : n - 1;
      int lo = 0, hi = nsub, lo1 = 0, hi1 = nsub;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (suboff[mid] <= rfirst) lo = mid; else hi = mid;
      }
      while (hi1 - lo1 > 1) {
        const int mid = (lo1 + hi1) >> 1;
        if (suboff[mid] <= rlast) lo1 = mid; else hi1 = mid;
      }
      sd0 = lo;
      sd1 = lo1;
      lm0 = *reinterpret_cast<const d2*>(lam + sd0 * (2 * LG) + 2 * q);
      mk0 = *reinterpret_cast<const d2*>(mask + sd0 * (2 * LG) + 2 * q);
    }
    for (int g = 0; g < NSTEP / U; ++g) {
      d2 acc1[U], acc2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { acc1[u] = d2{0.0, 0.0}; acc2[u] = d2{0.0, 0.0}; }
      for (int ch = 0; ch < nchunk; ++ch) {
        const int kc = (wd - ch * KC < KC) ? wd - ch * KC : KC;
        if (nchunk > 1 || g == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int k = 0; k < kc; ++k) {
            const int64_t e = base + (int64_t)64 * (ch * KC + k) + lane;
            mc[k * 64 + lane] = __builtin_nontemporal_load(sl_col + e);
            m1[k * 64 + lane] = __builtin_nontemporal_load(v1 + e);
            m2[k * 64 + lane] = __builtin_nontemporal_load(v2 + e);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const int rl0 = g * U * RS + grp;
        switch (kc) {
          case 8: spmm_sell_accum2<LG, U, 8>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 7: spmm_sell_accum2<LG, U, 7>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 6: spmm_sell_accum2<LG, U, 6>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 5: spmm_sell_accum2<LG, U, 5>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 4: spmm_sell_accum2<LG, U, 4>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 3: spmm_sell_accum2<LG, U, 3>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 2: spmm_sell_accum2<LG, U, 2>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          case 1: spmm_sell_accum2<LG, U, 1>(mc, m1, m2, rl0, Xq, ldx, acc1, acc2); break;
          default: break;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = (int64_t)64 * s + (g * U + u) * RS + grp;
        if (r >= n) continue;
        if (RES) {
          d2 lm = lm0, mk = mk0;
          if (sd0 != sd1) {                          // a slice across a subdomain boundary: this row's own subdomain
            int lo = sd0, hi = sd1 + 1;              // largest lo with suboff[lo] <= r
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if (suboff[mid] <= r) lo = mid; else hi = mid;
            }
            constexpr int mcols = 2 * LG;
            lm = *reinterpret_cast<const d2*>(lam + lo * mcols + 2 * q);
            mk = *reinterpret_cast<const d2*>(mask + lo * mcols + 2 * q);
          }
          d2 res;
          res.x = mk.x * (acc1[u].x - lm.x * acc2[u].x);
          res.y = mk.y * (acc1[u].y - lm.y * acc2[u].y);
          __builtin_nontemporal_store(res, reinterpret_cast<d2*>(Y1 + r * ldy + 2 * q));
        } else {
          __builtin_nontemporal_store(acc1[u], reinterpret_cast<d2*>(Y1 + r * ldy + 2 * q));
          __builtin_nontemporal_store(acc2[u], reinterpret_cast<d2*>(Y2 + r * ldy + 2 * q));
        }
      }
    }
  }
}
bool spmm_dual_available(const Csr& a, int m) {
  if (spmv_kind() != 1 || a.vec_lpr > 0 || a.nlong > 0 || !a.sl_ptr || !a.xcd_ptr || sell_wide_mm(a)) return false;
  return m == 16 || m == 32 || m == 64;
}
void spmm_dual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* Y1, double* Y2, int ldy,
               int m) {
  if (a.n == 0) return;
  if (!spmm_dual_available(a, m) || (ldx & 1) || (ldy & 1) || !aligned16(X) || !aligned16(Y1) || !aligned16(Y2))
    throw std::runtime_error("spmm_dual: operands not on the sliced path");
  ProfScope prof(PROF_SPMM, a.fine && m >= 16, (double)a.nnz * 20.0 + (double)a.n * 4.0 + 24.0 * m * (double)a.n,
                 4.0 * (double)a.nnz * m);
  int wpx = g_spmm_wpx < 0 ? 64 : (g_spmm_wpx == 0 ? 64 : g_spmm_wpx);
  const int need = (a.nslice + 31) / 32;
  if (wpx > need) wpx = need < 1 ? 1 : need;
#define DUAL(L)                                                                                                   \
  hipLaunchKernelGGL((k_spmm_sell_dual<L, 2, 0>), dim3(8 * wpx), dim3(256), 0, g_stream, a.sl_ptr, a.sl_col, v1, v2, a.n,  \
                     a.sched, a.xcd_ptr, X, ldx, Y1, Y2, ldy, (const int*)nullptr, 0, (const double*)nullptr,            \
                     (const double*)nullptr)
  if (m == 16) DUAL(8);
  else if (m == 32) DUAL(16);
  else DUAL(32);
#undef DUAL
}
// R = mask .* (A1 X - A2 X diag(lam)) per subdomain of c, both products in one pass over X and neither written (LOBPCG's
// residual block from X alone; lam / mask: nsub x m, subdomain-major).  Same accumulation order as spmm_dual.
void spmm_dual_residual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* R, int ldr,
                        int m, const Chunks& c, const double* lam, const double* mask) {
  if (a.n == 0) return;
  if (!spmm_dual_available(a, m) || (ldx & 1) || (ldr & 1) || !aligned16(X) || !aligned16(R) || !aligned16(lam) || !aligned16(mask))
    throw std::runtime_error("spmm_dual_residual: operands not on the sliced path");
  if (c.n != a.n) throw std::runtime_error("spmm_dual_residual: the subdomain list does not cover the matrix rows");
  ProfScope prof(PROF_SPMM, a.fine && m >= 16, (double)a.nnz * 20.0 + (double)a.n * 4.0 + 16.0 * m * (double)a.n,
                 4.0 * (double)a.nnz * m);
  int wpx = g_spmm_wpx < 0 ? 64 : (g_spmm_wpx == 0 ? 64 : g_spmm_wpx);
  const int need = (a.nslice + 31) / 32;
  if (wpx > need) wpx = need < 1 ? 1 : need;
#define DUALR(L)                                                                                                  \
  hipLaunchKernelGGL((k_spmm_sell_dual<L, 2, 1>), dim3(8 * wpx), dim3(256), 0, g_stream, a.sl_ptr, a.sl_col, v1, v2, a.n,  \
                     a.sched, a.xcd_ptr, X, ldx, R, (double*)nullptr, ldr, c.suboff, c.nsub, lam, mask)
  if (m == 16) DUALR(8);
  else if (m == 32) DUALR(16);
  else DUALR(32);
#undef DUALR
}
// values of b laid out on the sliced pattern of a: out[e] = b(r, a.sl_col[e]) for the stored entries of row r of a, 0 for a's
// entries b lacks and for the padding; *missing counts the entries of b that a's pattern does not hold
__global__ __launch_bounds__(256) void k_sell_values_on(int n, int ns, const int64_t* __restrict__ sl_ptr,
                                                        const int* __restrict__ sl_col, const int* __restrict__ arp,
                                                        const int* __restrict__ brp, const int* __restrict__ bcol,
                                                        const double* __restrict__ bval, double* __restrict__ out,
                                                        int* __restrict__ missing) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int l = threadIdx.x & 63;
  const int r = 64 * s + l;
  const int64_t base = sl_ptr[s];
  const int wd = (int)((sl_ptr[s + 1] - base) >> 6);
  const int la = r < n ? arp[r + 1] - arp[r] : 0;
  const int b0 = r < n ? brp[r] : 0, b1 = r < n ? brp[r + 1] : 0;
  int found = 0;
  for (int k = 0; k < wd; ++k) {
    const int64_t e = base + (int64_t)64 * k + l;
    double v = 0.0;
    if (k < la) {
      const int c = sl_col[e];
      for (int t = b0; t < b1; ++t)
        if (bcol[t] == c) { v = bval[t]; ++found; break; }
    }
    out[e] = v;
  }
  if (found != b1 - b0) atomicAdd(missing, 1);
}
double* sell_values_on(const Csr& a, const Csr& b) {
  if (spmv_kind() != 1 || a.nlong > 0 || a.vec_lpr > 0 || !a.sl_ptr || a.n != b.n || a.nslice == 0) return nullptr;
  double* out = (double*)alloc(sizeof(double) * (size_t)std::max<int64_t>(1, a.sl_nnz));
  int* dmiss = (int*)alloc(sizeof(int));
  hipLaunchKernelGGL(k_sell_values_on, dim3((a.nslice + 3) / 4), dim3(256), 0, g_stream, a.n, a.nslice, a.sl_ptr, a.sl_col,
                     a.rowptr, b.rowptr, b.col, b.val, out, dmiss);
  int h = 0;
  d2h(&h, dmiss, sizeof(int));
  dfree(dmiss);
  if (h) {            // b has entries outside a's pattern: no shared layout
    dfree(out);
    return nullptr;
  }
  return out;
}

static void spmm_ld(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                    const double* post) {
  if (a.n == 0 || m == 0) return;
  // algorithmic bytes: the matrix once, X once, Y once
  ProfScope prof(PROF_SPMM, a.fine && m >= 16, (double)a.nnz * 12.0 + (double)a.n * 4.0 + 16.0 * m * (double)a.n,
                 2.0 * (double)a.nnz * m);
  if (spmm_sell_launch<EPI_NONE>(a, X, ldx, Y, ldy, m, pre, post, nullptr, 0, nullptr, 0, nullptr, 0.0)) return;
  if (m <= 16) {
    int g = std::min(grid1d(a.n, 16), 8192);
    hipLaunchKernelGGL(k_spmm<16>, dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, X, ldx, Y,
                       ldy, m, pre, post);
  } else if (m <= 32) {
    int g = std::min(grid1d(a.n, 8), 8192);
    hipLaunchKernelGGL(k_spmm<32>, dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, X, ldx, Y,
                       ldy, m, pre, post);
  } else {
    int g = std::min(grid1d(a.n, 4), 8192);
    hipLaunchKernelGGL(k_spmm<64>, dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, X, ldx, Y,
                       ldy, m, pre, post);
  }
}
void spmm_strided(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                  const double* post) {
  spmm_ld(a, X, ldx, Y, ldy, m, pre, post);
}

bool csr_fusable(const Csr& a) { return spmv_kind() == 1 && (a.vec_lpr > 0 || a.nlong == 0); }

template <int EPI>
static void spmm_fused_t(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* B, int ldb,
                         double* Z, int ldz, const double* dinv, double w) {
  if (m == 1 && ldx <= 1 && ldy == 1 && ldb <= 1 && ldz <= 1 && csr_fusable(a)) {  // contiguous vectors: SpMV kernels
    if (a.vec_lpr > 0) {
      spmv_vec_launch<EPI>(a, X, Y, B, Z, dinv, w);
      return;
    }
    if (sell_wide(a)) {
      spmv_wide_launch<EPI>(a, X, Y, B, Z, dinv, w);
      return;
    }
    const int nwb = (a.nslice + 3) / 4;
    const int perw = (nwb + 7) / 8;
    // The fused multigrid kernels keep the cached (col,val) stream even for large matrices: measured in situ
    // (126^3) the non-temporal hint changes nothing for them (50.7 / 40.2 us vs 49.5 / 39.4 us), while the lines
    // they leave in the Infinity Cache are what the next plain SpMV of the same matrix hits (38.5 vs 48.0 us).
    hipLaunchKernelGGL((k_spmv_sell_epi<false, EPI>), dim3(perw * 8), dim3(256), 0, g_stream, a.sl_ptr, a.nslice,
                       a.n, a.sl_col, a.sl_val, X, Y, B, Z, dinv, w, a.col_scaled ? (const double*)nullptr : dinv);
    return;
  }
  const double* Xin = (EPI == EPI_PRE) ? B : X;
  const int ldin = (EPI == EPI_PRE) ? ldb : ldx;
  const double* pre = (EPI == EPI_PRE && !a.col_scaled) ? dinv : nullptr;
  // + the epilogue's block reads / writes: RES, ADD one more block in, JAC two, PRE one more out
  ProfScope prof(PROF_SPMM, a.fine && m >= 16,
                 (double)a.nnz * 12.0 + (double)a.n * 4.0 + (16.0 + ((EPI == EPI_JAC || EPI == EPI_POST) ? 16.0 : 8.0)) * m * (double)a.n,
                 2.0 * (double)a.nnz * m);
  if (spmm_sell_launch<EPI>(a, Xin, ldin, Y, ldy, m, pre, nullptr, B, ldb, Z, ldz, dinv, w)) return;
  if (m <= 16) {
    int g = std::min(grid1d(a.n, 16), 8192);
    hipLaunchKernelGGL((k_spmm<16, EPI>), dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, Xin, ldin, Y,
                       ldy, m, pre, nullptr, B, ldb, Z, ldz, dinv, w);
  } else if (m <= 32) {
    int g = std::min(grid1d(a.n, 8), 8192);
    hipLaunchKernelGGL((k_spmm<32, EPI>), dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, Xin, ldin, Y,
                       ldy, m, pre, nullptr, B, ldb, Z, ldz, dinv, w);
  } else {
    int g = std::min(grid1d(a.n, 4), 8192);
    hipLaunchKernelGGL((k_spmm<64, EPI>), dim3(g), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val, Xin, ldin, Y,
                       ldy, m, pre, nullptr, B, ldb, Z, ldz, dinv, w);
  }
}
void spmm_fused(const Csr& a, int epi, const double* X, int ldx, double* Y, int ldy, int m, const double* B, int ldb,
                double* Z, int ldz, const double* dinv, double w) {
  if (a.n == 0 || m == 0) return;
  switch (epi) {
    case EPI_RES: spmm_fused_t<EPI_RES>(a, X, ldx, Y, ldy, m, B, ldb, Z, ldz, dinv, w); break;
    case EPI_ADD: spmm_fused_t<EPI_ADD>(a, X, ldx, Y, ldy, m, B, ldb, Z, ldz, dinv, w); break;
    case EPI_JAC: spmm_fused_t<EPI_JAC>(a, X, ldx, Y, ldy, m, B, ldb, Z, ldz, dinv, w); break;
    case EPI_POST: spmm_fused_t<EPI_POST>(a, X, ldx, Y, ldy, m, B, ldb, Z, ldz, dinv, w); break;
    case EPI_PRE: spmm_fused_t<EPI_PRE>(a, X, ldx, Y, ldy, m, B, ldb, Z, ldz, dinv, w); break;
    default: throw std::runtime_error("spmm_fused: unknown epilogue");
  }
}

__global__ void k_csr_diag(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                           const double* __restrict__ val, double* __restrict__ d) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double v = 0.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
    if (col[k] == r) v += val[k];
  d[r] = v;
}
void csr_diag(const Csr& a, double* diag) {
  if (a.n == 0) return;
  hipLaunchKernelGGL(k_csr_diag, dim3(grid1d(a.n, 256)), dim3(256), 0, g_stream, a.n, a.rowptr, a.col, a.val,
                     diag);
}
__global__ void k_recip_positive(double* __restrict__ x, int n, int* __restrict__ bad) {
  int mine = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = x[i];
    if (v > 0.0) x[i] = 1.0 / v;
    else ++mine;
  }
  if (mine) atomicAdd(bad, mine);
}
int recip_positive(double* x, int n) {
  if (n <= 0) return 0;
  int* dbad = (int*)alloc(sizeof(int));
  hipLaunchKernelGGL(k_recip_positive, dim3(gridv(n)), dim3(256), 0, g_stream, x, n, dbad);
  int h = 0;
  d2h(&h, dbad, sizeof(int));
  dfree(dbad);
  return h;
}

// =============================================================================== index kernels
__global__ void k_gather(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ idx,
                         const double* __restrict__ d, int n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = in[idx[i]];
    if (d) v *= d[i];
    out[i] = v;
  }
}
void gather(double* out, const double* in, const int* idx, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather, dim3(std::min(grid1d(n, 256), 4096)), dim3(256), 0, g_stream, out, in, idx,
                     (const double*)nullptr, n);
}
void gather_mul(double* out, const double* in, const int* idx, const double* d, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather, dim3(std::min(grid1d(n, 256), 4096)), dim3(256), 0, g_stream, out, in, idx, d, n);
}
__global__ void k_segsum(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ ptr,
                         const int* __restrict__ idx, int nseg, int acc) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nseg; e += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = ptr[e]; k < ptr[e + 1]; ++k) s += in[idx[k]];
    out[e] = acc ? out[e] + s : s;
  }
}
__global__ void k_gather_rows(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ idx,
                              int64_t n, int w) {
  const int64_t tot = n * w;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / w;
    const int j = (int)(e - i * w);
    out[e] = in[(int64_t)idx[i] * w + j];
  }
}
void gather_rows(double* out, const double* in, const int* idx, int n, int w) {
  if (n <= 0 || w <= 0) return;
  hipLaunchKernelGGL(k_gather_rows, dim3(gridv((int64_t)n * w)), dim3(256), 0, g_stream, out, in, idx, (int64_t)n, w);
}
__global__ void k_segsum_rows(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ ptr,
                              const int* __restrict__ idx, int64_t nseg, int w, int acc) {
  const int64_t tot = nseg * w;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / w;
    const int j = (int)(e - r * w);
    double s = 0.0;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) s += in[(int64_t)idx[k] * w + j];   // fixed order
    out[e] = acc ? out[e] + s : s;
  }
}
void segsum_rows(double* out, const double* in, const int* ptr, const int* idx, int nseg, int w, bool accumulate) {
  if (nseg <= 0 || w <= 0) return;
  hipLaunchKernelGGL(k_segsum_rows, dim3(gridv((int64_t)nseg * w)), dim3(256), 0, g_stream, out, in, ptr, idx,
                     (int64_t)nseg, w, accumulate ? 1 : 0);
}
void segsum(double* out, const double* in, const int* ptr, const int* idx, int nseg, bool accumulate) {
  if (nseg <= 0) return;
  hipLaunchKernelGGL(k_segsum, dim3(std::min(grid1d(nseg, 256), 4096)), dim3(256), 0, g_stream, out, in, ptr,
                     idx, nseg, accumulate ? 1 : 0);
}

// =============================================================================== BLAS-1
__global__ void k_set(double* x, double v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_axpby(double* __restrict__ y, double a, const double* __restrict__ x, double b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}
__global__ void k_xmy(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * d[i];
}
__global__ void k_axpy_dev(double* __restrict__ y, const double* __restrict__ a, double sign,
                           const double* __restrict__ x, int64_t n) {
  const double s = sign * a[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] += s * x[i];
}
void set(double* x, double v, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_set, dim3(gridv(n)), dim3(256), 0, g_stream, x, v, (int64_t)n);
}
void copy(double* y, const double* x, int n) { d2d(y, x, sizeof(double) * (size_t)n); }
void axpy(double* y, double a, const double* x, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpby, dim3(gridv(n)), dim3(256), 0, g_stream, y, a, x, 1.0, (int64_t)n);
}
void axpby(double* y, double a, const double* x, double b, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpby, dim3(gridv(n)), dim3(256), 0, g_stream, y, a, x, b, (int64_t)n);
}
void xmy(double* y, const double* x, const double* d, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_xmy, dim3(gridv(n)), dim3(256), 0, g_stream, y, x, d, (int64_t)n);
}
void axpy_dev(double* y, const double* a_dev, double sign, const double* x, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpy_dev, dim3(gridv(n)), dim3(256), 0, g_stream, y, a_dev, sign, x, (int64_t)n);
}

// deterministic dot: fixed grid of DOT_BLOCKS partials, then one block sums them in order
constexpr int DOT_BLOCKS = 1024;
static double* g_dot_work = nullptr;
int dot_work_doubles() { return DOT_BLOCKS; }

__global__ __launch_bounds__(256) void k_dot1(const double* __restrict__ x, const double* __restrict__ y,
                                              int64_t n, double* __restrict__ part) {
  __shared__ double sm[4];
  // contiguous slice per block -> fixed association independent of scheduling
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t a = (int64_t)blockIdx.x * per, b = (a + per < n) ? a + per : n;
  double s = 0.0;
  for (int64_t i = a + threadIdx.x; i < b; i += 256) s += x[i] * y[i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dot2(const double* __restrict__ part, int np, double* __restrict__ out) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = s;
}
void dot(const double* x, const double* y, int n, double* out_dev) {
  if (!g_dot_work) g_dot_work = (double*)alloc(sizeof(double) * DOT_BLOCKS);
  int nb = std::min(DOT_BLOCKS, std::max(1, cdiv(n, 2048)));
  hipLaunchKernelGGL(k_dot1, dim3(nb), dim3(256), 0, g_stream, x, y, (int64_t)n, g_dot_work);
  hipLaunchKernelGGL(k_dot2, dim3(1), dim3(256), 0, g_stream, g_dot_work, nb, out_dev);
}

// =============================================================================== chunks
Chunks chunks_upload(int nsub, const int* h_suboff) {
  Chunks c;
  c.nsub = nsub;
  c.n = h_suboff[nsub] - h_suboff[0];
  std::vector<int> st, ln, sb, sp;
  sp.push_back(0);
  for (int s = 0; s < nsub; ++s) {
    for (int a = h_suboff[s]; a < h_suboff[s + 1]; a += CHUNK) {
      st.push_back(a);
      ln.push_back(std::min(CHUNK, h_suboff[s + 1] - a));
      sb.push_back(s);
    }
    c.maxsub_chunks = std::max(c.maxsub_chunks, (int)st.size() - sp.back());
    sp.push_back((int)st.size());
  }
  c.nchunk = (int)st.size();
  c.totals = (double*)alloc(sizeof(double) * 4 * (size_t)std::max(1, nsub));
  c.start = (int*)alloc(sizeof(int) * std::max<size_t>(1, st.size()));
  c.len = (int*)alloc(sizeof(int) * std::max<size_t>(1, st.size()));
  c.sub = (int*)alloc(sizeof(int) * std::max<size_t>(1, st.size()));
  c.subptr = (int*)alloc(sizeof(int) * sp.size());
  c.partial = (double*)alloc(sizeof(double) * 4 * std::max<size_t>(1, st.size()));
  c.suboff = (int*)alloc(sizeof(int) * (size_t)(nsub + 1));
  h2d(c.suboff, h_suboff, sizeof(int) * (size_t)(nsub + 1));
  h2d(c.start, st.data(), sizeof(int) * st.size());
  h2d(c.len, ln.data(), sizeof(int) * ln.size());
  h2d(c.sub, sb.data(), sizeof(int) * sb.size());
  h2d(c.subptr, sp.data(), sizeof(int) * sp.size());
  return c;
}
void gram_plan_drop(const Chunks& c);
void chunks_free(Chunks& c) {
  if (c.start) gram_plan_drop(c);
  dfree(c.start); dfree(c.len); dfree(c.sub); dfree(c.subptr); dfree(c.partial); dfree(c.suboff); dfree(c.totals);
  c = Chunks();
}

// sum of the chunk partials of subdomain s (slot k), same fixed order in every workgroup
__device__ __forceinline__ double sub_total(const double* __restrict__ part, int nchunk, int slot, int c0, int c1) {
  double s = 0.0;
  for (int c = c0; c < c1; ++c) s += part[(int64_t)slot * nchunk + c];
  return s;
}

// the same total computed cooperatively by the 256 threads of a workgroup (strided partial sums in a
// fixed pattern + the fixed-order block reduction => bitwise identical in every workgroup)
__device__ __forceinline__ double sub_total_block(const double* __restrict__ part, int nchunk, int slot, int c0,
                                                  int c1, double* sm) {
  double s = 0.0;
  for (int c = c0 + (int)threadIdx.x; c < c1; c += 256) s += part[(int64_t)slot * nchunk + c];
  return block_sum_256(s, sm);
}

// Large subdomains (the one-subdomain-per-GPU layout: 6 400 chunks): every per-subdomain reduction of chunk partials is
// done cooperatively -- one workgroup per subdomain (and slot) instead of one THREAD per subdomain, and once per launch
// instead of once per chunk workgroup.  Lists of at most g_par_reduce_min chunks keep the forms (and the summation
// order, hence the bits) they have had since round 1.  GENEO_PAR_REDUCE_MIN overrides the threshold (tests: 0).
static int g_par_reduce_min = getenv("GENEO_PAR_REDUCE_MIN") ? atoi(getenv("GENEO_PAR_REDUCE_MIN")) : 1024;
static inline bool big_subs(const Chunks& c) { return c.maxsub_chunks > g_par_reduce_min; }
void set_par_reduce_min(int chunks) { g_par_reduce_min = chunks; }
int get_par_reduce_min() { return g_par_reduce_min; }
// totals[s * 4 + slot] = sub_total_block(slot) for the slots of `mask` (bit k: slot k); grid (nsub, 4)
__global__ __launch_bounds__(256) void k_sub_totals(const int* __restrict__ subptr, const double* __restrict__ part,
                                                    int nchunk, int mask, double* __restrict__ totals) {
  __shared__ double sm[4];
  const int s = blockIdx.x, slot = blockIdx.y;
  if (!((mask >> slot) & 1)) return;
  const double t = sub_total_block(part, nchunk, slot, subptr[s], subptr[s + 1], sm);
  if (threadIdx.x == 0) totals[(int64_t)s * 4 + slot] = t;
}
static void sub_totals(const Chunks& c, int mask) {
  hipLaunchKernelGGL(k_sub_totals, dim3(c.nsub, 4), dim3(256), 0, g_stream, c.subptr, c.partial, c.nchunk, mask, c.totals);
}

__global__ __launch_bounds__(256) void k_seg_dot1(const int* __restrict__ start, const int* __restrict__ len,
                                                  const double* __restrict__ x, const double* __restrict__ y,
                                                  double* __restrict__ part, int nchunk, int slot) {
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int a = start[c], n = len[c];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += x[a + i] * y[a + i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[(int64_t)slot * nchunk + c] = s;
}
__global__ void k_seg_dot2(const int* __restrict__ subptr, const double* __restrict__ part, int nchunk, int pslot,
                           double* __restrict__ out, int stride, int slot, int nsub) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsub) return;
  out[(int64_t)s * stride + slot] = sub_total(part, nchunk, pslot, subptr[s], subptr[s + 1]);
}
__global__ __launch_bounds__(256) void k_seg_dot2_big(const int* __restrict__ subptr, const double* __restrict__ part,
                                                      int nchunk, int pslot, double* __restrict__ out, int stride, int slot) {
  __shared__ double sm[4];
  const int s = blockIdx.x;
  const double t = sub_total_block(part, nchunk, pslot, subptr[s], subptr[s + 1], sm);
  if (threadIdx.x == 0) out[(int64_t)s * stride + slot] = t;
}
void seg_dot(const Chunks& c, const double* x, const double* y, double* out, int stride, int slot) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_seg_dot1, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, x, y, c.partial,
                     c.nchunk, 3);
  if (big_subs(c)) {
    hipLaunchKernelGGL(k_seg_dot2_big, dim3(c.nsub), dim3(256), 0, g_stream, c.subptr, c.partial, c.nchunk, 3, out, stride,
                       slot);
    return;
  }
  hipLaunchKernelGGL(k_seg_dot2, dim3(grid1d(c.nsub, 64)), dim3(64), 0, g_stream, c.subptr, c.partial, c.nchunk,
                     3, out, stride, slot, c.nsub);
}

// --- batched CG (one independent CG per subdomain; scalars live in sc[s*8 + k]) -------------
// sc slots: 0 rz(parity 0) 1 rz(parity 1) 2 pAp 3 rr 4 alpha 5 beta 6 active 7 rr0
__global__ __launch_bounds__(256) void k_cg_start(const int* __restrict__ start, const int* __restrict__ len,
                                                  double* __restrict__ x, double* __restrict__ r,
                                                  double* __restrict__ z, double* __restrict__ p,
                                                  const double* __restrict__ b, const double* __restrict__ dinv,
                                                  double* __restrict__ part, int nchunk) {
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int a = start[c], n = len[c];
  double rz = 0.0, rr = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double rv = b[a + i];
    x[a + i] = 0.0;
    r[a + i] = rv;
    if (dinv) {
      const double zv = dinv[a + i] * rv;
      z[a + i] = zv;
      p[a + i] = zv;
      rz += rv * zv;
    }
    rr += rv * rv;
  }
  rz = block_sum_256(rz, sm);
  rr = block_sum_256(rr, sm);
  if (threadIdx.x == 0) {
    if (dinv) part[(int64_t)1 * nchunk + c] = rz;
    part[(int64_t)2 * nchunk + c] = rr;
  }
}
__global__ void k_cg_start2(const int* __restrict__ subptr, const double* __restrict__ part, int nchunk,
                            double* __restrict__ sc, int nsub, const double* __restrict__ totals) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsub) return;
  const double rz = totals ? totals[(int64_t)s * 4 + 1] : sub_total(part, nchunk, 1, subptr[s], subptr[s + 1]);
  const double rr = totals ? totals[(int64_t)s * 4 + 2] : sub_total(part, nchunk, 2, subptr[s], subptr[s + 1]);
  double* q = sc + (int64_t)s * 8;
  q[0] = rz; q[1] = rz; q[2] = 0.0; q[3] = rr; q[4] = 0.0; q[5] = 0.0;
  q[6] = (rr > 0.0) ? 1.0 : 0.0;
  q[7] = rr;
}
void cg_start(const Chunks& c, double* sc, double* x, double* r, double* z, double* p, const double* b,
              const double* dinv) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_cg_start, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, x, r, z, p, b, dinv,
                     c.partial, c.nchunk);
  const bool big = big_subs(c);
  if (big) sub_totals(c, 6);
  hipLaunchKernelGGL(k_cg_start2, dim3(grid1d(c.nsub, 64)), dim3(64), 0, g_stream, c.subptr, c.partial, c.nchunk,
                     sc, c.nsub, big ? c.totals : (const double*)nullptr);
}
void seg_partial(const Chunks& c, const double* x, const double* y, int slot) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_seg_dot1, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, x, y, c.partial, c.nchunk,
                     slot);
}
// second half of cg_start when the caller preconditions itself: p = z, rz slots from partial slot 1
__global__ void k_cg_start3(const int* __restrict__ subptr, const double* __restrict__ part, int nchunk,
                            double* __restrict__ sc, int nsub, const double* __restrict__ totals) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsub) return;
  const double rz = totals ? totals[(int64_t)s * 4 + 1] : sub_total(part, nchunk, 1, subptr[s], subptr[s + 1]);
  sc[(int64_t)s * 8 + 0] = rz;
  sc[(int64_t)s * 8 + 1] = rz;
}
void cg_set_rz(const Chunks& c, double* sc) {
  if (c.nchunk == 0) return;
  const bool big = big_subs(c);
  if (big) sub_totals(c, 2);
  hipLaunchKernelGGL(k_cg_start3, dim3(grid1d(c.nsub, 64)), dim3(64), 0, g_stream, c.subptr, c.partial, c.nchunk, sc,
                     c.nsub, big ? c.totals : (const double*)nullptr);
}
void seg_pap(const Chunks& c, const double* p, const double* q) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_seg_dot1, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, p, q, c.partial,
                     c.nchunk, 0);
}
__global__ __launch_bounds__(256) void k_cg_update(const int* __restrict__ start, const int* __restrict__ len,
                                                   const int* __restrict__ sub, const int* __restrict__ subptr,
                                                   double* __restrict__ sc, int parity, double* __restrict__ x,
                                                   double* __restrict__ r, double* __restrict__ z,
                                                   const double* __restrict__ p, const double* __restrict__ q,
                                                   const double* __restrict__ dinv, double* __restrict__ part,
                                                   int nchunk, const double* __restrict__ totals) {
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int s = sub[c];
  const int c0 = subptr[s], c1 = subptr[s + 1];
  // (totals: the same sum, computed once per launch by k_sub_totals -- bit-identical)
  const double pap = totals ? totals[(int64_t)s * 4 + 0] : sub_total_block(part, nchunk, 0, c0, c1, sm);
  const double rz = sc[(int64_t)s * 8 + parity];
  const double active = sc[(int64_t)s * 8 + 6];
  const double alpha = (active != 0.0 && pap != 0.0) ? rz / pap : 0.0;
  const int a = start[c], n = len[c];
  double nrz = 0.0, nrr = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double xv = x[a + i] + alpha * p[a + i];
    const double rv = r[a + i] - alpha * q[a + i];
    x[a + i] = xv;
    r[a + i] = rv;
    if (dinv) {
      const double zv = dinv[a + i] * rv;
      z[a + i] = zv;
      nrz += rv * zv;
    }
    nrr += rv * rv;
  }
  nrz = block_sum_256(nrz, sm);
  nrr = block_sum_256(nrr, sm);
  if (threadIdx.x == 0) {
    if (dinv) part[(int64_t)1 * nchunk + c] = nrz;
    part[(int64_t)2 * nchunk + c] = nrr;
    if (c == c0) {
      sc[(int64_t)s * 8 + 2] = pap;
      sc[(int64_t)s * 8 + 4] = alpha;
    }
  }
}
void cg_update(const Chunks& c, double* sc, int parity, double* x, double* r, double* z, const double* p,
                 const double* q, const double* dinv) {
  if (c.nchunk == 0) return;
  const bool big = big_subs(c);
  if (big) sub_totals(c, 1);
  hipLaunchKernelGGL(k_cg_update, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.subptr, sc,
                     parity, x, r, z, p, q, dinv, c.partial, c.nchunk, big ? c.totals : (const double*)nullptr);
}
__global__ __launch_bounds__(256) void k_cg_direction(const int* __restrict__ start, const int* __restrict__ len,
                                                      const int* __restrict__ sub, const int* __restrict__ subptr,
                                                      double* __restrict__ sc, int parity, double* __restrict__ p,
                                                      const double* __restrict__ z, double tol2,
                                                      const double* __restrict__ part, int nchunk,
                                                      const double* __restrict__ totals) {
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int s = sub[c];
  const int c0 = subptr[s], c1 = subptr[s + 1];
  const double nrz = totals ? totals[(int64_t)s * 4 + 1] : sub_total_block(part, nchunk, 1, c0, c1, sm);
  const double nrr = totals ? totals[(int64_t)s * 4 + 2] : sub_total_block(part, nchunk, 2, c0, c1, sm);
  const double rz = sc[(int64_t)s * 8 + parity];
  const double was_active = sc[(int64_t)s * 8 + 6];
  const double rr0 = sc[(int64_t)s * 8 + 7];
  const double beta = (was_active != 0.0 && rz != 0.0) ? nrz / rz : 0.0;
  const int a = start[c], n = len[c];
  if (was_active != 0.0) {
    for (int i = threadIdx.x; i < n; i += 256) p[a + i] = z[a + i] + beta * p[a + i];
  }
  if (threadIdx.x == 0 && c == c0) {
    sc[(int64_t)s * 8 + (parity ^ 1)] = nrz;
    sc[(int64_t)s * 8 + 3] = nrr;
    sc[(int64_t)s * 8 + 5] = beta;
    // NOTE: slot 6 is read by the other workgroups of this subdomain in THIS launch only through
    // was_active loaded above; writing it here is ordered by the kernel boundary for later launches.
  }
  (void)tol2; (void)rr0;
}
// second tiny launch flips the active flag (separate launch => no intra-kernel read/write race)
__global__ void k_cg_flag(double* __restrict__ sc, double tol2, int nsub) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsub) return;
  double* q = sc + (int64_t)s * 8;
  if (q[6] != 0.0 && q[3] <= tol2 * q[7]) q[6] = 0.0;
}
void cg_direction(const Chunks& c, double* sc, int parity, double* p, const double* z, double tol2) {
  if (c.nchunk == 0) return;
  const bool big = big_subs(c);
  if (big) sub_totals(c, 6);
  hipLaunchKernelGGL(k_cg_direction, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.subptr, sc,
                     parity, p, z, tol2, c.partial, c.nchunk, big ? c.totals : (const double*)nullptr);
  hipLaunchKernelGGL(k_cg_flag, dim3(grid1d(c.nsub, 64)), dim3(64), 0, g_stream, sc, tol2, c.nsub);
}

// =============================================================================== tall-skinny blocks
// Gram row ranges: GRAM_CH consecutive chunks of one subdomain per workgroup.
constexpr int GRAM_CH = 1;

using d4 = __attribute__((ext_vector_type(4))) double;

// G_partial[blk] (p x q) = S[rows]^T T[rows].  MFMA v_mfma_f64_16x16x4_f64:
//   A operand lane l: A[i = l&15][k = l>>4]   -> S[row0 + 4*step + (l>>4)][16*I + (l&15)]
//   B operand lane l: B[k = l>>4][j = l&15]   -> T[row0 + 4*step + (l>>4)][16*J + (l&15)]
//   C/D lane l, reg v: row i = (l>>4) + 4*v, col j = l&15         (cdna_hip_programming.md s3)
// The 4 waves form a 2 x 2 grid; wave (wi, wj) owns the TI x TJ block of 16x16 output tiles
// I = I0 + wi*TI + a, J = J0 + wj*TJ + b, so each 4-row step costs TI + TJ LDS fragment reads for
// TI*TJ MFMAs.  16-row slabs of S and T go HBM -> registers -> LDS; the next slab's loads are issued
// before the MFMA phase of the current one (register double buffering).
// NC = 64-column groups staged per row: ceil(max(p, q) / 64).  amdgpu_waves_per_eu(3): 164 VGPRs, no spills, 3 waves per
// SIMD instead of 2 (128 + 72 AGPRs): 1.30 -> 1.12 ms for the 96 x 96 Gram of 2.29 M rows.
template <int TI, int TJ, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_gram_mfma(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                   const int* __restrict__ gfirst, const int* __restrict__ gcount,
                                                   const double* __restrict__ S, int lds_, int p,
                                                   const double* __restrict__ T, int ldt_, int q,
                                                   double* __restrict__ Gpart, int I0, int J0,
                                                   const double* __restrict__ S2, int lds2_, int psplit) {
  // S2 != nullptr: the left operand is [S(:, 0:psplit) | S2(:, 0:p-psplit)] -- two blocks that live in different
  // buffers (A W and B W) against ONE pass over T
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ldS = (p % 32 == 0) ? p + 16 : p;  // rows r, r+1 land 32 banks apart
  const int ldT = (q % 32 == 0) ? q + 16 : q;
  double* sS = smem;             // 16 x ldS
  double* sT = smem + 16 * ldS;  // 16 x ldT
  const int g = blockIdx.x;
  const int c0 = gfirst[g], nc = gcount[g];
  const int row0 = cstart[c0];
  int nrows = 0;
  for (int c = 0; c < nc; ++c) nrows += clen[c0 + c];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int P16 = p >> 4, Q16 = q >> 4;
  const int wi = w >> 1, wj = w & 1;
  int aoff[TI], boff[TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    int I = I0 + wi * TI + a;
    if (I >= P16) I = P16 - 1;
    aoff[a] = 16 * I + (l & 15);
  }
#pragma unroll
  for (int b = 0; b < TJ; ++b) {
    int J = J0 + wj * TJ + b;
    if (J >= Q16) J = Q16 - 1;
    boff[b] = 16 * J + (l & 15);
  }
  d4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  double rs[4][NC], rt[4][NC];
  auto load_slab = [&](int r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 4 * w + u;
      const bool ok = rr < nrows;
      const double* srow = S + (int64_t)(row0 + (ok ? rr : 0)) * lds_;
      const double* srow2 = S2 ? S2 + (int64_t)(row0 + (ok ? rr : 0)) * lds2_ - psplit : srow;
      const double* trow = T + (int64_t)(row0 + (ok ? rr : 0)) * ldt_;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const int cc = l + 64 * ci;
        rs[u][ci] = (ok && cc < p) ? (cc < psplit ? srow[cc] : srow2[cc]) : 0.0;
        rt[u][ci] = (ok && cc < q) ? trow[cc] : 0.0;
      }
    }
  };
  load_slab(0);
  for (int r = 0; r < nrows; r += 16) {
    __syncthreads();  // every wave is done reading the previous slab
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const int cc = l + 64 * ci;
        if (cc < p) sS[(4 * w + u) * ldS + cc] = rs[u][ci];
        if (cc < q) sT[(4 * w + u) * ldT + cc] = rt[u][ci];
      }
    __syncthreads();
    if (r + 16 < nrows) load_slab(r + 16);  // in flight during the MFMA phase
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const int kr = 4 * step + (l >> 4);
      double av[TI], bv[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) av[a] = sS[kr * ldS + aoff[a]];
#pragma unroll
      for (int b = 0; b < TJ; ++b) bv[b] = sT[kr * ldT + boff[b]];
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  double* G = Gpart + (int64_t)g * p * q;
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    const int I = I0 + wi * TI + a;
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int J = J0 + wj * TJ + b;
      if (I < P16 && J < Q16) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + (l >> 4) + 4 * v, colj = 16 * J + (l & 15);
          G[(int64_t)row * q + colj] = acc[a][b][v];
        }
      }
    }
  }
}

// The two shapes LOBPCG launches every iteration / every 8th iteration, with 16-byte streaming loads (round 3):
//   TWO = true :  [A1 | A2]^T T, two left blocks of 16 TI columns each (A W and B W inside the 96-wide A S / B S rows)
//   TWO = false:  A1^T T, one left block of 32 TI columns
// against a right operand of 32 TJ columns.  A 16-row slab goes HBM -> registers as 16-byte units (the right operand's
// slab is one contiguous block when ldt == q: thread t copies units t, t + 256, ...), non-temporal (every byte is read
// once), -> LDS as ds_write_b128, and the MFMA phase is k_gram_mfma's: per output tile the same sequence of 4-row steps in
// ascending row order, so the partial matrices are bit-identical to k_gram_mfma<TI, TJ, *>'s.  W-row Gram of 2.29 M rows:
// 0.71 -> 0.55 ms (4.1 -> 5.3 TB/s, 39.6 -> 50.9 TFLOP/s; a read-only stream of the same bytes takes 0.43 ms);
// scripts/dev/gram_bench.hip holds the variants that were measured (32-row slabs, two LDS slabs: no better).
template <int TI, int TJ, bool TWO>
__global__ __launch_bounds__(256) void k_gram_flat(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                   const int* __restrict__ gfirst, const double* __restrict__ A1,
                                                   const double* __restrict__ A2, int lda, const double* __restrict__ T,
                                                   int ldt_, double* __restrict__ Gpart) {
  typedef spmm_d2 d2;
  constexpr int p = 32 * TI, q = 32 * TJ, ldS = p + 16, ldT = q + 16, SR = 16;
  constexpr int UT = q / 2;                        // 16-byte units per row of the right operand
  constexpr int UL = TWO ? p / 4 : p / 2;          // units per row of one left block
  constexpr int NUT = (SR * UT + 255) / 256, NUL = (SR * UL + 255) / 256;
  __shared__ __attribute__((aligned(16))) double sS[SR * ldS];
  __shared__ __attribute__((aligned(16))) double sT[SR * ldT];
  const int g = blockIdx.x;
  const int c0 = gfirst[g];
  const int row0 = cstart[c0], nrows = clen[c0];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int wi = w >> 1, wj = w & 1;
  int aoff[TI], boff[TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a) aoff[a] = 16 * (wi * TI + a) + (l & 15);
#pragma unroll
  for (int b = 0; b < TJ; ++b) boff[b] = 16 * (wj * TJ + b) + (l & 15);
  d4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  d2 rt[NUT], r1[NUL], r2[TWO ? NUL : 1];
  auto load_slab = [&](int r) {
    const int nr = (nrows - r < SR) ? nrows - r : SR;
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      const int u = tid + 256 * j, rr = u / UT, cu = u - rr * UT;
      const bool ok = (SR * UT % 256 == 0 || u < SR * UT) && rr < nr;
      rt[j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(T + (int64_t)(row0 + r + rr) * ldt_ + 2 * cu))
                 : d2{0.0, 0.0};
    }
#pragma unroll
    for (int j = 0; j < NUL; ++j) {
      const int u = tid + 256 * j, rr = u / UL, cu = u - rr * UL;
      const bool ok = (SR * UL % 256 == 0 || u < SR * UL) && rr < nr;
      const int64_t off = (int64_t)(row0 + r + (ok ? rr : 0)) * lda + 2 * cu;
      r1[j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(A1 + off)) : d2{0.0, 0.0};
      if (TWO) r2[j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(A2 + off)) : d2{0.0, 0.0};
    }
  };
  load_slab(0);
  for (int r = 0; r < nrows; r += SR) {
    __syncthreads();  // every wave is done reading the previous slab
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      const int u = tid + 256 * j, rr = u / UT, cu = u - rr * UT;
      if (SR * UT % 256 == 0 || u < SR * UT) *reinterpret_cast<d2*>(sT + rr * ldT + 2 * cu) = rt[j];
    }
#pragma unroll
    for (int j = 0; j < NUL; ++j) {
      const int u = tid + 256 * j, rr = u / UL, cu = u - rr * UL;
      if (SR * UL % 256 == 0 || u < SR * UL) {
        *reinterpret_cast<d2*>(sS + rr * ldS + 2 * cu) = r1[j];
        if (TWO) *reinterpret_cast<d2*>(sS + rr * ldS + p / 2 + 2 * cu) = r2[j];
      }
    }
    __syncthreads();
    if (r + SR < nrows) load_slab(r + SR);  // in flight during the MFMA phase
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const int kr = 4 * step + (l >> 4);
      double av[TI], bv[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) av[a] = sS[kr * ldS + aoff[a]];
#pragma unroll
      for (int b = 0; b < TJ; ++b) bv[b] = sT[kr * ldT + boff[b]];
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  double* G = Gpart + (int64_t)g * p * q;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * (wi * TI + a) + (l >> 4) + 4 * v, colj = 16 * (wj * TJ + b) + (l & 15);
        G[(int64_t)row * q + colj] = acc[a][b][v];
      }
}

// plain-FMA twin (any p, q): each thread owns output entries e = tid, tid+256, ...
__global__ __launch_bounds__(256) void k_gram_fma(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                  const int* __restrict__ gfirst, const int* __restrict__ gcount,
                                                  const double* __restrict__ S, int lds_, int p,
                                                  const double* __restrict__ T, int ldt_, int q,
                                                  double* __restrict__ Gpart, const double* __restrict__ S2, int lds2_,
                                                  int psplit) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sS = smem;           // 16 x p
  double* sT = smem + 16 * p;  // 16 x q
  const int g = blockIdx.x;
  const int c0 = gfirst[g], nc = gcount[g];
  const int row0 = cstart[c0];
  int nrows = 0;
  for (int c = 0; c < nc; ++c) nrows += clen[c0 + c];
  const int tid = threadIdx.x;
  const int npq = p * q;
  constexpr int MAXE = 40;  // 256*40 = 10240 >= 96*96
  double acc[MAXE];
#pragma unroll
  for (int i = 0; i < MAXE; ++i) acc[i] = 0.0;
  for (int r = 0; r < nrows; r += 16) {
    const int nr = (nrows - r < 16) ? nrows - r : 16;
    __syncthreads();
    for (int e = tid; e < 16 * p; e += 256) {
      const int rr = e / p, cc = e - rr * p;
      sS[e] = (rr < nr) ? (cc < psplit ? S[(int64_t)(row0 + r + rr) * lds_ + cc]
                                       : S2[(int64_t)(row0 + r + rr) * lds2_ + cc - psplit]) : 0.0;
    }
    for (int e = tid; e < 16 * q; e += 256) {
      const int rr = e / q, cc = e - rr * q;
      sT[e] = (rr < nr) ? T[(int64_t)(row0 + r + rr) * ldt_ + cc] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int e = tid + 256 * i;
      if (e < npq) {
        const int a = e / q, b = e - a * q;
        double s = acc[i];
        for (int rr = 0; rr < 16; ++rr) s += sS[rr * p + a] * sT[rr * q + b];
        acc[i] = s;
      }
    }
  }
  double* G = Gpart + (int64_t)g * npq;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = tid + 256 * i;
    if (e < npq) G[e] = acc[i];
  }
}

// sum the partial Grams of each subdomain in fixed order
// 64 output entries per workgroup, 4 thread sets: set k sums the partial matrices g = first + k, first + k + 4, ... in
// order, the four sums are combined in fixed order -- four independent load streams instead of one dependent chain
__global__ __launch_bounds__(256) void k_gram_reduce(const int* __restrict__ gsubptr, const double* __restrict__ Gpart,
                                                     int pq, double* __restrict__ G) {
  __shared__ double sm[4][64];
  const int s = blockIdx.y;
  const int el = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  double acc = 0.0;
  if (e < pq)
    for (int g = gsubptr[s] + k; g < gsubptr[s + 1]; g += 4) acc += Gpart[(int64_t)g * pq + e];
  sm[k][el] = acc;
  __syncthreads();
  if (k == 0 && e < pq) G[(int64_t)s * pq + e] = ((sm[0][el] + sm[1][el]) + sm[2][el]) + sm[3][el];
}

// host-side cache of the gram grouping for a Chunks object (keyed by its device pointer)
// Large subdomains (thousands of partial matrices each): two stages -- GZ workgroups per 64 entries, workgroup z sums the
// partial matrices first + 4 z + k, + 4 GZ, ... (k = thread set) and combines its four sets in order; the second stage adds
// the GZ results in order.  6 383 partial 64 x 96 Grams of ONE subdomain: 0.75 ms -> the time of reading them once.
__global__ __launch_bounds__(256) void k_gram_reduce_z(const int* __restrict__ gsubptr, const double* __restrict__ Gpart,
                                                       int pq, int GZ, double* __restrict__ out) {
  __shared__ double sm[4][64];
  const int s = blockIdx.y, z = blockIdx.z;
  const int el = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  double acc = 0.0;
  if (e < pq)
    for (int g = gsubptr[s] + 4 * z + k; g < gsubptr[s + 1]; g += 4 * GZ) acc += Gpart[(int64_t)g * pq + e];
  sm[k][el] = acc;
  __syncthreads();
  if (k == 0 && e < pq) out[((int64_t)s * GZ + z) * pq + e] = ((sm[0][el] + sm[1][el]) + sm[2][el]) + sm[3][el];
}
__global__ __launch_bounds__(256) void k_gram_reduce_fin(const double* __restrict__ part2, int pq, int GZ,
                                                         double* __restrict__ G) {
  const int s = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= pq) return;
  double acc = 0.0;
  for (int z = 0; z < GZ; ++z) acc += part2[((int64_t)s * GZ + z) * pq + e];
  G[(int64_t)s * pq + e] = acc;
}
struct GramPlan {
  const int* key = nullptr;
  int ngroup = 0, nsub = 0;
  int *gfirst = nullptr, *gcount = nullptr, *gsubptr = nullptr;
  double* part = nullptr;
  size_t part_doubles = 0;
  double* part2 = nullptr;      // stage-1 results of the two-stage reduction (large subdomains)
  size_t part2_doubles = 0;
};
static std::vector<GramPlan> g_plans;
static GramPlan& gram_plan(const Chunks& c) {
  for (auto& pl : g_plans)
    if (pl.key == c.start) return pl;
  GramPlan pl;
  pl.key = c.start;
  pl.nsub = c.nsub;
  std::vector<int> sp(c.nsub + 1);
  d2h(sp.data(), c.subptr, sizeof(int) * (c.nsub + 1));
  std::vector<int> gf, gc, gs;
  gs.push_back(0);
  for (int s = 0; s < c.nsub; ++s) {
    for (int a = sp[s]; a < sp[s + 1]; a += GRAM_CH) {
      gf.push_back(a);
      gc.push_back(std::min(GRAM_CH, sp[s + 1] - a));
    }
    gs.push_back((int)gf.size());
  }
  pl.ngroup = (int)gf.size();
  pl.gfirst = (int*)alloc(sizeof(int) * std::max<size_t>(1, gf.size()));
  pl.gcount = (int*)alloc(sizeof(int) * std::max<size_t>(1, gc.size()));
  pl.gsubptr = (int*)alloc(sizeof(int) * gs.size());
  h2d(pl.gfirst, gf.data(), sizeof(int) * gf.size());
  h2d(pl.gcount, gc.data(), sizeof(int) * gc.size());
  h2d(pl.gsubptr, gs.data(), sizeof(int) * gs.size());
  g_plans.push_back(pl);
  return g_plans.back();
}
void gram_plan_drop(const Chunks& c) {
  for (size_t i = 0; i < g_plans.size(); ++i)
    if (g_plans[i].key == c.start) {
      dfree(g_plans[i].gfirst); dfree(g_plans[i].gcount); dfree(g_plans[i].gsubptr); dfree(g_plans[i].part);
      dfree(g_plans[i].part2);
      g_plans.erase(g_plans.begin() + i);
      return;
    }
}

static void gram_impl(const Chunks& c, const double* S, int lds_, int p, const double* S2, int lds2_, int psplit,
                      const double* T, int ldt_, int q, double* G) {
  if (c.nchunk == 0 || p == 0 || q == 0) return;
  ProfScope prof(PROF_GRAM, p >= 32 && q >= 32, 8.0 * (double)c.n * (p + q), 2.0 * (double)c.n * p * q);
  GramPlan& pl = gram_plan(c);
  const size_t need = (size_t)pl.ngroup * p * q;
  if (pl.part_doubles < need) {
    dfree(pl.part);
    pl.part = (double*)alloc(sizeof(double) * need);
    pl.part_doubles = need;
  }
  const bool mfma_ok = !g_no_mfma && (p % 16 == 0) && (q % 16 == 0) && p <= 192 && q <= 192;
  // LOBPCG's two shapes on 16-byte streaming loads (k_gram_flat; bit-identical partial matrices)
  static_assert(GRAM_CH == 1, "k_gram_flat takes one chunk per workgroup");
  const bool flat_base = mfma_ok && g_gram_flat && q == 96 && (ldt_ % 2 == 0) && (lds_ % 2 == 0) && aligned16(S) && aligned16(T);
  if (flat_base && S2 && p == 64 && psplit == 32 && lds2_ == lds_ && aligned16(S2)) {
    hipLaunchKernelGGL((k_gram_flat<2, 3, true>), dim3(pl.ngroup), dim3(256), 0, g_stream, c.start, c.len, pl.gfirst, S, S2,
                       lds_, T, ldt_, pl.part);
  } else if (flat_base && !S2 && p == 96) {
    hipLaunchKernelGGL((k_gram_flat<3, 3, false>), dim3(pl.ngroup), dim3(256), 0, g_stream, c.start, c.len, pl.gfirst, S,
                       nullptr, lds_, T, ldt_, pl.part);
  } else if (mfma_ok) {
    const int ldS = (p % 32 == 0) ? p + 16 : p, ldT = (q % 32 == 0) ? q + 16 : q;
    const size_t sm = sizeof(double) * 16 * (size_t)(ldS + ldT);
    const int P16 = p / 16, Q16 = q / 16;
    const int ti = std::min(3, (P16 + 1) / 2), tj = std::min(3, (Q16 + 1) / 2);
    for (int I0 = 0; I0 < P16; I0 += 2 * ti)
      for (int J0 = 0; J0 < Q16; J0 += 2 * tj) {
#define GRAM_LAUNCH(A, B)                                                                                             \
  do {                                                                                                                \
    if (std::max(p, q) <= 128)                                                                                        \
      hipLaunchKernelGGL((k_gram_mfma<A, B, 2>), dim3(pl.ngroup), dim3(256), sm, g_stream, c.start, c.len, pl.gfirst, \
                         pl.gcount, S, lds_, p, T, ldt_, q, pl.part, I0, J0, S2, lds2_, psplit);                      \
    else                                                                                                              \
      hipLaunchKernelGGL((k_gram_mfma<A, B, 3>), dim3(pl.ngroup), dim3(256), sm, g_stream, c.start, c.len, pl.gfirst, \
                         pl.gcount, S, lds_, p, T, ldt_, q, pl.part, I0, J0, S2, lds2_, psplit);                      \
  } while (0)
        switch (ti * 10 + tj) {
          case 11: GRAM_LAUNCH(1, 1); break;
          case 12: GRAM_LAUNCH(1, 2); break;
          case 13: GRAM_LAUNCH(1, 3); break;
          case 21: GRAM_LAUNCH(2, 1); break;
          case 22: GRAM_LAUNCH(2, 2); break;
          case 23: GRAM_LAUNCH(2, 3); break;
          case 31: GRAM_LAUNCH(3, 1); break;
          case 32: GRAM_LAUNCH(3, 2); break;
          default: GRAM_LAUNCH(3, 3); break;
        }
#undef GRAM_LAUNCH
      }
  } else {
    if ((size_t)p * q > 256 * 40) throw std::runtime_error("gram: p*q too large for the FMA kernel");
    const size_t sm = sizeof(double) * 16 * (size_t)(p + q);
    hipLaunchKernelGGL(k_gram_fma, dim3(pl.ngroup), dim3(256), sm, g_stream, c.start, c.len, pl.gfirst, pl.gcount,
                       S, lds_, p, T, ldt_, q, pl.part, S2 ? S2 : S, lds2_, psplit);
  }
  if (big_subs(c)) {
    constexpr int GZ = 16;
    const size_t need2 = (size_t)c.nsub * GZ * p * q;
    if (pl.part2_doubles < need2) {
      dfree(pl.part2);
      pl.part2 = (double*)alloc(sizeof(double) * need2);
      pl.part2_doubles = need2;
    }
    hipLaunchKernelGGL(k_gram_reduce_z, dim3(grid1d(p * q, 64), c.nsub, GZ), dim3(256), 0, g_stream, pl.gsubptr, pl.part,
                       p * q, GZ, pl.part2);
    hipLaunchKernelGGL(k_gram_reduce_fin, dim3(grid1d(p * q, 256), c.nsub), dim3(256), 0, g_stream, pl.part2, p * q, GZ, G);
    return;
  }
  hipLaunchKernelGGL(k_gram_reduce, dim3(grid1d(p * q, 64), c.nsub), dim3(256), 0, g_stream, pl.gsubptr, pl.part,
                     p * q, G);
}
void gram(const Chunks& c, const double* S, int lds_, int p, const double* T, int ldt_, int q, double* G) {
  gram_impl(c, S, lds_, p, nullptr, 0, p, T, ldt_, q, G);
}
void gram2(const Chunks& c, const double* S1, int lds1, int p1, const double* S2, int lds2, int p2, const double* T,
           int ldt_, int q, double* G) {
  gram_impl(c, S1, lds1, p1 + p2, S2, lds2, p1, T, ldt_, q, G);
}

// Y[rows] (+)= S[rows] C_s : one workgroup per chunk, 64-row slabs of S through LDS, the wave's
// B fragments (its 16 output columns of C_s) held in registers for the whole chunk.
//   A operand lane l: A[i = l&15][k = l>>4] -> S[r + (l&15)][4*kk + (l>>4)]   (LDS, ld = 2 mod 32:
//                     the 32 lanes of a ds_read_b64 group hit 32 distinct 8-byte banks)
//   B operand lane l: B[k = l>>4][j = l&15] -> C[4*kk + (l>>4)][16*J + (l&15)] (registers)
//   D lane l, reg v : Y[r + (l>>4) + 4v][16*J + (l&15)]
// Wave w -> column tile J = Jbase + (w % nJ), row tiles rt = w / nJ, + 4/nJ, ...  (nJ = min(Q16,4))
template <int P4>
__global__ __launch_bounds__(256) void k_blockmul_mfma(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                       const int* __restrict__ csub, const double* __restrict__ S,
                                                       int lds_, const double* __restrict__ C, int q,
                                                       double* __restrict__ Y, int ldy, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int p = 4 * P4;
  // odd row stride: the 16 lanes of one ds_read_b64 group read 16 different rows at the same column, i.e. addresses
  // r * ldS; with ldS odd their 16 bank pairs are all different (ldS = 2 mod 32 left 2-way conflicts: PMC showed 41 %
  // of the LDS cycles in bank conflicts)
  constexpr int ldS = p + 1;
  constexpr int NC = (p + 63) / 64;
  constexpr int SR = 32;                           // slab rows (2 row tiles)
  double* sS = smem;                               // SR x ldS
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const double* Cs = C + (int64_t)s * p * q;
  const int Q16 = q >> 4;
  const int nJ = Q16 < 4 ? Q16 : 4;  // Q16 in {1,2,4,8,...}
  const int rtw = 4 / nJ;            // waves sharing one column tile
  const int jsel = w % nJ, rsel = w / nJ;
  double rg[SR / 4][NC];             // wave w stages slab rows 8w..8w+7
  auto load_slab = [&](int r) {
#pragma unroll
    for (int u = 0; u < SR / 4; ++u) {
      const int rr = r + (SR / 4) * w + u;
      const bool ok = rr < nrows;
      const double* srow = S + (int64_t)(row0 + (ok ? rr : 0)) * lds_;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const int cc = l + 64 * ci;
        rg[u][ci] = (ok && cc < p) ? srow[cc] : 0.0;
      }
    }
  };
  for (int Jbase = 0; Jbase < Q16; Jbase += nJ) {
    const int J = Jbase + jsel;
    double bfrag[P4];
#pragma unroll
    for (int kk = 0; kk < P4; ++kk) bfrag[kk] = Cs[(int64_t)(4 * kk + (l >> 4)) * q + 16 * J + (l & 15)];
    load_slab(0);
    for (int r = 0; r < nrows; r += SR) {
      const int nr = (nrows - r < SR) ? nrows - r : SR;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < SR / 4; ++u)
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          const int cc = l + 64 * ci;
          if (cc < p) sS[((SR / 4) * w + u) * ldS + cc] = rg[u][ci];
        }
      __syncthreads();
      if (r + SR < nrows) load_slab(r + SR);  // next slab in flight during the MFMA phase
      for (int rt = rsel; rt < SR / 16; rt += rtw) {
        if (16 * rt >= nr) break;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        if (accumulate) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 16 * rt + (l >> 4) + 4 * v;
            if (rr < nr) acc[v] = Y[(int64_t)(row0 + r + rr) * ldy + 16 * J + (l & 15)];
          }
        }
        const double* arow = sS + (16 * rt + (l & 15)) * ldS + (l >> 4);
#pragma unroll
        for (int kk = 0; kk < P4; ++kk)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * kk], bfrag[kk], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * rt + (l >> 4) + 4 * v;
          if (rr < nr) Y[(int64_t)(row0 + r + rr) * ldy + 16 * J + (l & 15)] = acc[v];
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_blockmul_fma(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                      const int* __restrict__ csub, const double* __restrict__ S,
                                                      int lds_, int p, const double* __restrict__ C, int q,
                                                      double* __restrict__ Y, int ldy, int accumulate) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const double* Cs = C + (int64_t)s * p * q;
  for (int e = threadIdx.x; e < nrows * q; e += 256) {
    const int rr = e / q, j = e - rr * q;
    const double* srow = S + (int64_t)(row0 + rr) * lds_;
    double acc = accumulate ? Y[(int64_t)(row0 + rr) * ldy + j] : 0.0;
    for (int k = 0; k < p; ++k) acc += srow[k] * Cs[k * q + j];
    Y[(int64_t)(row0 + rr) * ldy + j] = acc;
  }
}
template <int P4>
static void launch_blockmul(const Chunks& c, const double* S, int lds_, const double* C, int q, double* Y, int ldy,
                            bool accumulate) {
  constexpr int p = 4 * P4;
  constexpr int ldS = p + 1;
  const size_t sm = sizeof(double) * (size_t)32 * ldS;
  static bool attr_done = false;
  if (sm > 64 * 1024 && !attr_done) {
    HIPCHK(hipFuncSetAttribute((const void*)k_blockmul_mfma<P4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sm));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_blockmul_mfma<P4>, dim3(c.nchunk), dim3(256), sm, g_stream, c.start, c.len, c.sub, S, lds_,
                     C, q, Y, ldy, accumulate ? 1 : 0);
}
void block_mul(const Chunks& c, const double* S, int lds_, int p, const double* C, int q, double* Y, int ldy,
               bool accumulate) {
  if (c.nchunk == 0 || p == 0 || q == 0) return;
  ProfScope prof(PROF_BLOCKMUL, p >= 32 && q >= 32, 8.0 * (double)c.n * (p + q), 2.0 * (double)c.n * p * q);
  const int Q16 = q / 16;
  const bool qok = (q % 16 == 0) && (Q16 == 1 || Q16 == 2 || (Q16 % 4 == 0));
  if (!g_no_mfma && qok && p % 4 == 0) {
    switch (p / 4) {
      case 4:  launch_blockmul<4>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 8:  launch_blockmul<8>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 12: launch_blockmul<12>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 16: launch_blockmul<16>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 24: launch_blockmul<24>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 32: launch_blockmul<32>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      case 48: launch_blockmul<48>(c, S, lds_, C, q, Y, ldy, accumulate); return;
      default: break;
    }
  }
  hipLaunchKernelGGL(k_blockmul_fma, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, S, lds_, p, C,
                     q, Y, ldy, accumulate ? 1 : 0);
}

// ------------------------------------------------------------------------------- fused LOBPCG update (m = 32)
// One launch for the whole Rayleigh-Ritz update of an iteration:  [X' P'] = S C for S, A S and B S, and the residual
// block of the next iteration R = mask .* (A X' - B X' diag(lam)).  It uses the structure of C (core.cpp, rr_one):
//     X' = X Cxx + [P W] Cw ,   P' = keep .* ([P W] Cw)        (Cw = C[32:96, 0:32]; keep_j = 0 for locked pairs)
// so the [P W] Cw product is computed ONCE per operand (24 MFMAs per 16 x 16 tile pair instead of 48), and A X', B X'
// never travel back through HBM for the residual kernel (3 block passes).  Wave w of the workgroup owns row tile w >> 1
// and column tile w & 1 of every 32-row slab; the three operands' slabs go HBM -> registers -> LDS one after the other
// (next slab's loads in flight during the MFMA phase); the C fragments (24 doubles per lane) stay in registers.
// Lane maps as k_blockmul_mfma.  Bytes: reads 3 x 96 n, writes 3 x 64 n + 32 n doubles = 4.0 KB per row (separate
// kernels: 3 x (96 + 64) + (64 + 32) = 4.6 KB); flops 2 n (64 x 32 + 32 x 32) x 3.
// NOPS = 1 (round 3, the "lean" iteration of core.cpp): only S is carried -- A X', B X' are never formed, the residual
// comes from one two-operator product over X' (k_spmm_sell_dual<.., .., 1>) -- 1.28 KB per row instead of 4.0.
template <int NOPS>
__global__ __launch_bounds__(256) void k_lobpcg_update32(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                        const int* __restrict__ csub, const double* __restrict__ S,
                                                        const double* __restrict__ AS, const double* __restrict__ BS,
                                                        const double* __restrict__ C, const double* __restrict__ keep,
                                                        const double* __restrict__ lam, const double* __restrict__ mask,
                                                        double* __restrict__ T, double* __restrict__ AT,
                                                        double* __restrict__ BT, double* __restrict__ R) {
  // Round 3: a 32-row slab of a 96-wide operand is ONE contiguous 24 KiB block, copied as 16-byte units (thread t takes
  // units t, t + 256, ...: 6 per thread) with non-temporal loads, and the outputs leave through non-temporal stores --
  // every byte of this kernel is touched once.  Same MFMA sequence per output tile as before (bit-identical results);
  // 2.29 M rows: 1.69 -> 1.61 ms = 5.8 TB/s, where a plain copy of the same access pattern reaches 5.3 TB/s
  // (scripts/dev/update_bench.hip; neither two slabs in flight nor shorter workgroups change it).
  typedef spmm_d2 d2;
  constexpr int m = 32, p = 96, q = 64, ldS = p + 1, SR = 32, NU = SR * (p / 2) / 256;
  __shared__ double sS[SR * ldS];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], sd = csub[c];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int rt = w >> 1, ct = w & 1;
  const double* Cs = C + (int64_t)sd * p * q;
  const int col = 16 * ct + (l & 15);              // this lane's output column inside X' (and inside P')
  double bx[8], bw[16];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) bx[kk] = Cs[(int64_t)(4 * kk + (l >> 4)) * q + col];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) bw[kk] = Cs[(int64_t)(m + 4 * kk + (l >> 4)) * q + col];
  const double kp = keep[sd * m + col], lm = lam[sd * m + col], mk = mask ? mask[sd * m + col] : 1.0;
  const double* src[3] = {S, AS, BS};
  double* dst[3] = {T, AT, BT};
  int loff[NU];                                    // LDS offset of this thread's j-th unit (odd row stride: two b64 writes)
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int u = tid + 256 * j;
    loff[j] = (u / (p / 2)) * ldS + 2 * (u % (p / 2));
  }
  d2 rg[NU];
  auto load_slab = [&](int t) {        // t = 3 * slab + operand
    const int r = (t / NOPS) * SR;
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    const d2* base = reinterpret_cast<const d2*>(src[t % NOPS] + (int64_t)(row0 + r) * p);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = tid + 256 * j;
      rg[j] = (u < nr * (p / 2)) ? __builtin_nontemporal_load(base + u) : d2{0.0, 0.0};
    }
  };
  const int nt = NOPS * ((nrows + SR - 1) / SR);
  load_slab(0);
  d4 ax = (d4){0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < nt; ++t) {
    const int op = t % NOPS, r = (t / NOPS) * SR;
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      sS[loff[j]] = rg[j].x;
      sS[loff[j] + 1] = rg[j].y;
    }
    __syncthreads();
    if (t + 1 < nt) load_slab(t + 1);
    if (16 * rt < nr) {
      const double* arow = sS + (16 * rt + (l & 15)) * ldS + (l >> 4);
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[m + 4 * kk], bw[kk], acc, 0, 0, 0);
      double* out = dst[op];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * rt + (l >> 4) + 4 * v;
        if (rr < nr) __builtin_nontemporal_store(kp * acc[v], out + (int64_t)(row0 + r + rr) * p + m + col);
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * kk], bx[kk], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * rt + (l >> 4) + 4 * v;
        if (rr < nr) __builtin_nontemporal_store(acc[v], out + (int64_t)(row0 + r + rr) * p + col);
      }
      if (NOPS == 3 && op == 1) ax = acc;
      if (NOPS == 3 && op == 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * rt + (l >> 4) + 4 * v;
          if (rr < nr) __builtin_nontemporal_store(mk * (ax[v] - lm * acc[v]), R + (int64_t)(row0 + r + rr) * m + col);
        }
      }
    }
  }
}
void lobpcg_update32(const Chunks& c, const double* S, const double* AS, const double* BS, const double* C,
                     const double* keep, const double* lam, const double* mask, double* T, double* AT, double* BT,
                     double* R) {
  if (c.nchunk == 0) return;
  if (!aligned16(S) || !aligned16(AS) || !aligned16(BS)) throw std::runtime_error("lobpcg_update32: operands must be 16-byte aligned");
  // counted with the block updates it replaces (three of them): same class, bytes and flops of the fused form
  ProfScope prof(PROF_BLOCKMUL, true, 8.0 * (double)c.n * (3 * 96 + 3 * 64 + 32), 2.0 * (double)c.n * (64 * 32 + 32 * 32) * 3);
  hipLaunchKernelGGL(k_lobpcg_update32<3>, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, S, AS, BS, C, keep,
                     lam, mask, T, AT, BT, R);
}
void lobpcg_update32_basis(const Chunks& c, const double* S, const double* C, const double* keep, double* T) {
  if (c.nchunk == 0) return;
  if (!aligned16(S)) throw std::runtime_error("lobpcg_update32_basis: operand must be 16-byte aligned");
  ProfScope prof(PROF_BLOCKMUL, true, 8.0 * (double)c.n * (96 + 64), 2.0 * (double)c.n * (64 * 32 + 32 * 32));
  hipLaunchKernelGGL(k_lobpcg_update32<1>, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, S, S, S, C, keep,
                     keep, (const double*)nullptr, T, T, T, (double*)nullptr);
}
bool lobpcg_update32_available() { return !g_no_mfma; }

// per-chunk column sums -> per-subdomain column sums (fixed order)
// per-chunk column sums -> per-subdomain column sums: K thread sets per column (chunks c = first + k, + K, ... in
// order), combined in fixed order -- K independent load streams instead of one dependent chain per column
__global__ __launch_bounds__(1024) void k_colsum_reduce(const int* __restrict__ subptr, const double* __restrict__ part, int m,
                                                        int w, double* __restrict__ out) {
  extern __shared__ double smc[];            // K x w
  const int s = blockIdx.x;
  const int K = blockDim.x / w;
  const int j = threadIdx.x % w, k = threadIdx.x / w;
  double t = 0.0;
  if (j < m)
    for (int c = subptr[s] + k; c < subptr[s + 1]; c += K) t += part[(int64_t)c * m + j];
  smc[k * w + j] = t;
  __syncthreads();
  if (k == 0 && j < m) {
    double a = smc[j];
    for (int i = 1; i < K; ++i) a += smc[i * w + j];
    out[(int64_t)s * m + j] = a;
  }
}
static void launch_colsum_reduce(const Chunks& c, const double* part, int m, double* out) {
  const int w = ((m + 15) / 16) * 16;
  const int K = std::max(1, std::min(8, 1024 / w));
  if (w > 1024) throw std::runtime_error("colsum_reduce: more than 1024 columns");
  hipLaunchKernelGGL(k_colsum_reduce, dim3(c.nsub), dim3(K * w), sizeof(double) * K * w, g_stream, c.subptr, part, m, w, out);
}
static double* g_colpart = nullptr;
static size_t g_colpart_n = 0;
static double* colpart(size_t n) {
  if (g_colpart_n < n) {
    dfree(g_colpart);
    g_colpart = (double*)alloc(sizeof(double) * n);
    g_colpart_n = n;
  }
  return g_colpart;
}

// R = AX - BX diag(lam_s); column squared norms per subdomain
__global__ __launch_bounds__(256) void k_block_residual(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                        const int* __restrict__ csub, const double* __restrict__ AX,
                                                        int lda, const double* __restrict__ BX, int ldb,
                                                        const double* __restrict__ lam, int m,
                                                        double* __restrict__ R, int ldr, double* __restrict__ part) {
  __shared__ double sm[256];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int tid = threadIdx.x;
  const int rstep = 256 / m;  // m <= 256
  const int j = tid % m, rr0 = tid / m;
  double acc = 0.0;
  if (tid < rstep * m) {
    const double lj = lam[(int64_t)s * m + j];
    for (int rr = rr0; rr < nrows; rr += rstep) {
      const int64_t row = row0 + rr;
      const double v = AX[row * lda + j] - lj * BX[row * ldb + j];
      R[row * ldr + j] = v;
      acc += v * v;
    }
  }
  sm[tid] = (tid < rstep * m) ? acc : 0.0;
  __syncthreads();
  if (tid < m) {
    double t = 0.0;
    for (int k = 0; k < rstep; ++k) t += sm[k * m + tid];
    part[(int64_t)c * m + tid] = t;
  }
}
void block_residual(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam, int m,
                    double* R, int ldr, double* nrm) {
  if (c.nchunk == 0) return;
  if (m > 256) throw std::runtime_error("block_residual: m > 256");
  double* part = colpart((size_t)c.nchunk * m);
  hipLaunchKernelGGL(k_block_residual, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, AX, lda, BX,
                     ldb, lam, m, R, ldr, part);
  launch_colsum_reduce(c, part, m, nrm);
}
__global__ __launch_bounds__(256) void k_block_residual_norms(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                              const int* __restrict__ csub, const double* __restrict__ AX,
                                                              int lda, const double* __restrict__ BX, int ldb,
                                                              const double* __restrict__ lam, int m,
                                                              double* __restrict__ R, int ldr,
                                                              const double* __restrict__ mask, double* __restrict__ part) {
  __shared__ double sm[3][256];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int tid = threadIdx.x;
  const int rstep = 256 / m;  // m <= 85 keeps 3m <= 256 reducers
  const int j = tid % m, rr0 = tid / m;
  double ar = 0.0, aa = 0.0, ab = 0.0;
  if (tid < rstep * m) {
    const double lj = lam[(int64_t)s * m + j];
    const double mk = mask ? mask[(int64_t)s * m + j] : 1.0;
    for (int rr = rr0; rr < nrows; rr += rstep) {
      const int64_t row = row0 + rr;
      const double a = AX[row * lda + j], b = BX[row * ldb + j];
      const double v = a - lj * b;
      R[row * ldr + j] = mk * v;
      ar += v * v;
      aa += a * a;
      ab += b * b;
    }
  }
  const bool live = tid < rstep * m;
  sm[0][tid] = live ? ar : 0.0;
  sm[1][tid] = live ? aa : 0.0;
  sm[2][tid] = live ? ab : 0.0;
  __syncthreads();
  if (tid < 3 * m) {
    const int w = tid / m, jj = tid % m;
    double t = 0.0;
    for (int k = 0; k < rstep; ++k) t += sm[w][k * m + jj];
    part[(int64_t)c * 3 * m + tid] = t;
  }
}
void block_residual_norms(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam,
                          int m, double* R, int ldr, const double* colmask, double* nrm3) {
  if (c.nchunk == 0) return;
  if (3 * m > 256) throw std::runtime_error("block_residual_norms: m > 85");
  double* part = colpart((size_t)c.nchunk * 3 * m);
  hipLaunchKernelGGL(k_block_residual_norms, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, AX, lda,
                     BX, ldb, lam, m, R, ldr, colmask, part);
  if (nrm3) launch_colsum_reduce(c, part, 3 * m, nrm3);   // null: the caller only wants the residual block
}
__global__ __launch_bounds__(256) void k_block_colnorm(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                       const double* __restrict__ X, int ldx, int m,
                                                       double* __restrict__ part) {
  __shared__ double sm[256];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c];
  const int tid = threadIdx.x;
  const int rstep = 256 / m;
  const int j = tid % m, rr0 = tid / m;
  double acc = 0.0;
  if (tid < rstep * m)
    for (int rr = rr0; rr < nrows; rr += rstep) {
      const double v = X[(int64_t)(row0 + rr) * ldx + j];
      acc += v * v;
    }
  sm[tid] = (tid < rstep * m) ? acc : 0.0;
  __syncthreads();
  if (tid < m) {
    double t = 0.0;
    for (int k = 0; k < rstep; ++k) t += sm[k * m + tid];
    part[(int64_t)c * m + tid] = t;
  }
}
void block_colnorm(const Chunks& c, const double* X, int ldx, int m, double* nrm) {
  if (c.nchunk == 0) return;
  if (m > 256) throw std::runtime_error("block_colnorm: m > 256");
  double* part = colpart((size_t)c.nchunk * m);
  hipLaunchKernelGGL(k_block_colnorm, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, X, ldx, m, part);
  launch_colsum_reduce(c, part, m, nrm);
}

// strided elementwise: Y[i][j] = a*X[i][j] + b*Y[i][j]   (b == 0 => pure assignment, Y not read)
__global__ void k_block_axpby(double* __restrict__ Y, int ldy, double a, const double* __restrict__ X, int ldx,
                              double b, int64_t n, int m) {
  const int64_t tot = n * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / m;
    const int j = (int)(e - i * m);
    const double xv = a * X[i * ldx + j];
    Y[i * ldy + j] = (b == 0.0) ? xv : xv + b * Y[i * ldy + j];
  }
}
void block_axpby(double* Y, int ldy, double a, const double* X, int ldx, double b, int n, int m) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(k_block_axpby, dim3(gridv((int64_t)n * m)), dim3(256), 0, g_stream, Y, ldy, a, X, ldx, b,
                     (int64_t)n, m);
}
__global__ void k_jacobi_step(double* __restrict__ X, int ldx, const double* __restrict__ B, int ldb,
                              const double* __restrict__ AX, const double* __restrict__ dinv, double w, int64_t n,
                              int m, int zero_guess) {
  const int64_t tot = n * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / m;
    const int j = (int)(e - i * m);
    const double bv = B[i * ldb + j];
    if (zero_guess) X[i * ldx + j] = w * dinv[i] * bv;
    else X[i * ldx + j] += w * dinv[i] * (bv - AX[e]);
  }
}
void jacobi_step(double* X, int ldx, const double* B, int ldb, const double* AX, const double* dinv, double w, int n,
                 int m, bool zero_guess) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(k_jacobi_step, dim3(gridv((int64_t)n * m)), dim3(256), 0, g_stream, X, ldx, B, ldb, AX, dinv, w,
                     (int64_t)n, m, zero_guess ? 1 : 0);
}
// one Chebyshev step fused: r -= ad ; d = a * dinv .* r + b * d ; z += d
__global__ void k_cheb_update(double* __restrict__ r, const double* __restrict__ ad, double* __restrict__ d,
                              double* __restrict__ z, int ldz, const double* __restrict__ dinv, double a, double b,
                              int64_t n, int m) {
  const int64_t tot = n * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / m;
    const int j = (int)(e - i * m);
    const double rv = r[e] - ad[e];
    const double dv = a * dinv[i] * rv + b * d[e];
    r[e] = rv;
    d[e] = dv;
    z[i * ldz + j] += dv;
  }
}
void cheb_update(double* r, const double* ad, double* d, double* z, int ldz, const double* dinv, double a, double b,
                 int n, int m) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(k_cheb_update, dim3(gridv((int64_t)n * m)), dim3(256), 0, g_stream, r, ad, d, z, ldz, dinv, a, b,
                     (int64_t)n, m);
}
// Y[i][j] = a * d[i] * X[i][j] + b*Y[i][j]
__global__ void k_block_rowscale(double* __restrict__ Y, int ldy, const double* __restrict__ X, int ldx,
                                 const double* __restrict__ d, double a, double b, int64_t n, int m) {
  const int64_t tot = n * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / m;
    const int j = (int)(e - i * m);
    const double xv = a * d[i] * X[i * ldx + j];
    Y[i * ldy + j] = (b == 0.0) ? xv : xv + b * Y[i * ldy + j];
  }
}
void block_rowscale(double* Y, int ldy, const double* X, int ldx, const double* d, double a, double b, int n, int m) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(k_block_rowscale, dim3(gridv((int64_t)n * m)), dim3(256), 0, g_stream, Y, ldy, X, ldx, d, a, b,
                     (int64_t)n, m);
}
__global__ __launch_bounds__(256) void k_block_colscale(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                        const int* __restrict__ csub, double* __restrict__ X, int ldx,
                                                        int m, const double* __restrict__ cs) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  for (int e = threadIdx.x; e < nrows * m; e += 256) {
    const int rr = e / m, j = e - rr * m;
    X[(int64_t)(row0 + rr) * ldx + j] *= cs[(int64_t)s * m + j];
  }
}
void block_colscale(const Chunks& c, double* X, int ldx, int m, const double* colscale) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_block_colscale, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, X, ldx, m,
                     colscale);
}

// counter-based start block (splitmix64 of (seed, global subdomain id, local row, column))
__host__ __device__ inline double hash_unit(uint64_t seed, uint64_t gid, uint64_t row, uint64_t colj) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (gid + 1) + 0xBF58476D1CE4E5B9ull * (row + 1) +
               0x94D049BB133111EBull * (colj + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;  // [-0.5, 0.5)
}
double hash_unit_host(uint64_t seed, uint64_t gid, uint64_t row, uint64_t colj) {
  return hash_unit(seed, gid, row, colj);
}
__global__ __launch_bounds__(256) void k_block_init(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                    const int* __restrict__ csub, const int* __restrict__ suboff,
                                                    double* __restrict__ X, int ldx, int m,
                                                    const int* __restrict__ sub_gid, uint64_t seed) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int sub_row0 = suboff[s];
  for (int e = threadIdx.x; e < nrows * m; e += 256) {
    const int rr = e / m, j = e - rr * m;
    const int64_t lrow = (int64_t)(row0 + rr) - sub_row0;
    X[(int64_t)(row0 + rr) * ldx + j] =
        (j == 0) ? 1.0 : hash_unit(seed, (uint64_t)sub_gid[s], (uint64_t)lrow, (uint64_t)j);
  }
}
void block_init(const Chunks& c, double* X, int ldx, int m, const int* sub_gid, uint64_t seed) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_block_init, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, X, ldx,
                     m, sub_gid, seed);
}

__global__ __launch_bounds__(256) void k_block_extract(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                       const int* __restrict__ csub, const int* __restrict__ suboff,
                                                       const double* __restrict__ X, int ldx, int m,
                                                       const double* __restrict__ d, const int* __restrict__ sel,
                                                       const int* __restrict__ ksub, const int64_t* __restrict__ zbase,
                                                       double* __restrict__ Z) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int k = ksub[s];
  const int srow0 = suboff[s];
  const int ns = suboff[s + 1] - srow0;
  double* Zs = Z + zbase[s];
  for (int j = 0; j < k; ++j) {
    const int src = sel[(int64_t)s * m + j];
    for (int rr = threadIdx.x; rr < nrows; rr += 256) {
      const int64_t row = row0 + rr;
      const double v = (src < 0) ? 1.0 : X[row * ldx + src];  // src < 0: constant (Nicolaides) vector
      Zs[(int64_t)j * ns + (row - srow0)] = d[row] * v;
    }
  }
}
void block_extract(const Chunks& c, const double* X, int ldx, int m, const double* d, const int* sel,
                   const int* ksub, const int64_t* zbase, double* Z) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_block_extract, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, X,
                     ldx, m, d, sel, ksub, zbase, Z);
}

__global__ __launch_bounds__(256) void k_z_rowmajor(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                    const int* __restrict__ csub, const int* __restrict__ suboff,
                                                    const double* __restrict__ Z, const int64_t* __restrict__ zbase,
                                                    const int* __restrict__ ksub, double* __restrict__ ZR, int kp) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int k = ksub[s];
  const int srow0 = suboff[s];
  const int ns = suboff[s + 1] - srow0;
  const double* Zs = Z + zbase[s];
  for (int j = 0; j < kp; ++j)
    for (int rr = threadIdx.x; rr < nrows; rr += 256) {
      const int64_t row = row0 + rr;
      ZR[row * kp + j] = (j < k) ? Zs[(int64_t)j * ns + (row - srow0)] : 0.0;
    }
}
void z_rowmajor(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, double* ZR, int kp) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_z_rowmajor, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, Z, zbase,
                     ksub, ZR, kp);
}

// =============================================================================== coarse space
// Z_s column-major (k_s columns of length n_s).  One workgroup per chunk: x chunk in registers,
// loop over the k_s columns (coalesced), chunk partial per column, then ordered reduce.
constexpr int ZMAXK = 256;
__global__ __launch_bounds__(256) void k_zt_apply(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                  const int* __restrict__ csub, const int* __restrict__ suboff,
                                                  const double* __restrict__ Z, const int64_t* __restrict__ zbase,
                                                  const int* __restrict__ ksub, const double* __restrict__ xL,
                                                  double* __restrict__ part, int kmax) {
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int k = ksub[s];
  const int srow0 = suboff[s];
  const int ns = suboff[s + 1] - srow0;
  const double* Zs = Z + zbase[s];
  double xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rr = threadIdx.x + 256 * u;
    xv[u] = (rr < nrows) ? xL[row0 + rr] : 0.0;
  }
  for (int j = 0; j < k; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = threadIdx.x + 256 * u;
      if (rr < nrows) acc += Zs[(int64_t)j * ns + (row0 - srow0) + rr] * xv[u];
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[(int64_t)c * kmax + j] = acc;
  }
}
__global__ void k_zt_reduce(const int* __restrict__ subptr, const int* __restrict__ ksub, const int* __restrict__ zoff,
                            const double* __restrict__ part, int kmax, double* __restrict__ yE) {
  const int s = blockIdx.x;
  const int j = threadIdx.x;
  if (j >= ksub[s]) return;
  double t = 0.0;
  for (int c = subptr[s]; c < subptr[s + 1]; ++c) t += part[(int64_t)c * kmax + j];
  yE[zoff[s] + j] = t;
}
// large subdomains: one workgroup per (subdomain, column), strided partial sums + the fixed-order block reduction
__global__ __launch_bounds__(256) void k_zt_reduce_big(const int* __restrict__ subptr, const int* __restrict__ ksub,
                                                       const int* __restrict__ zoff, const double* __restrict__ part, int kmax,
                                                       double* __restrict__ yE) {
  __shared__ double sm[4];
  const int s = blockIdx.x, j = blockIdx.y;
  if (j >= ksub[s]) return;
  double t = 0.0;
  for (int c = subptr[s] + (int)threadIdx.x; c < subptr[s + 1]; c += 256) t += part[(int64_t)c * kmax + j];
  t = block_sum_256(t, sm);
  if (threadIdx.x == 0) yE[zoff[s] + j] = t;
}
void zt_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff, int kmax,
              const double* xL, double* yE, int dimE_total) {
  if (kmax > ZMAXK) throw std::runtime_error("zt_apply: more than 256 coarse vectors in one subdomain");
  zero(yE, sizeof(double) * (size_t)dimE_total);
  if (c.nchunk == 0 || kmax == 0) return;
  double* part = colpart((size_t)c.nchunk * kmax);
  hipLaunchKernelGGL(k_zt_apply, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, Z, zbase,
                     ksub, xL, part, kmax);
  if (big_subs(c))
    hipLaunchKernelGGL(k_zt_reduce_big, dim3(c.nsub, kmax), dim3(256), 0, g_stream, c.subptr, ksub, zoff, part, kmax, yE);
  else
    hipLaunchKernelGGL(k_zt_reduce, dim3(c.nsub), dim3(((kmax + 63) / 64) * 64), 0, g_stream, c.subptr, ksub, zoff,
                       part, kmax, yE);
}
__global__ __launch_bounds__(256) void k_z_apply(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                 const int* __restrict__ csub, const int* __restrict__ suboff,
                                                 const double* __restrict__ Z, const int64_t* __restrict__ zbase,
                                                 const int* __restrict__ ksub, const int* __restrict__ zoff,
                                                 const double* __restrict__ yE, double* __restrict__ wL, int acc) {
  __shared__ double sy[ZMAXK];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int k = ksub[s];
  const int srow0 = suboff[s];
  const int ns = suboff[s + 1] - srow0;
  const double* Zs = Z + zbase[s];
  for (int j = threadIdx.x; j < k; j += 256) sy[j] = yE[zoff[s] + j];
  __syncthreads();
  for (int rr = threadIdx.x; rr < nrows; rr += 256) {
    double a = 0.0;
    for (int j = 0; j < k; ++j) a += Zs[(int64_t)j * ns + (row0 - srow0) + rr] * sy[j];
    wL[row0 + rr] = acc ? wL[row0 + rr] + a : a;
  }
}
void z_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff,
             const double* yE, double* wL, bool accumulate) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_z_apply, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, Z, zbase,
                     ksub, zoff, yE, wL, accumulate ? 1 : 0);
}

// X_s = Inv_s B_s, Inv_s symmetric (read column-wise = coalesced).  One workgroup per chunk of rows.
__global__ __launch_bounds__(256) void k_dense_sym_apply(const int* __restrict__ cstart, const int* __restrict__ clen,
                                                         const int* __restrict__ csub, const int* __restrict__ suboff,
                                                         const double* __restrict__ inv, const int64_t* __restrict__ base,
                                                         const double* __restrict__ B, int ldb, double* __restrict__ X,
                                                         int ldx, int m) {
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], s = csub[c];
  const int s0 = suboff[s];
  const int ns = suboff[s + 1] - s0;
  const double* A = inv + base[s];
  for (int e = threadIdx.x; e < nrows * m; e += 256) {
    const int rr = e % nrows, j = e / nrows;   // consecutive threads -> consecutive rows (coalesced A reads)
    const int i = row0 - s0 + rr;
    // eight independent partial sums: the loads of eight columns are in flight together (the block is a few dozen
    // columns wide and one thread walks all of them: with a single chain the launch is 56 load latencies long)
    double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int k = 0;
    for (; k + 8 <= ns; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += A[(int64_t)(k + u) * ns + i] * B[(int64_t)(s0 + k + u) * ldb + j];
    }
    for (; k < ns; ++k) a8[0] += A[(int64_t)k * ns + i] * B[(int64_t)(s0 + k) * ldb + j];
    X[(int64_t)(row0 + rr) * ldx + j] = ((a8[0] + a8[4]) + (a8[1] + a8[5])) + ((a8[2] + a8[6]) + (a8[3] + a8[7]));
  }
}
void dense_sym_apply(const Chunks& c, const double* inv, const int64_t* base, const double* B, int ldb, double* X,
                     int ldx, int m) {
  if (c.nchunk == 0) return;
  hipLaunchKernelGGL(k_dense_sym_apply, dim3(c.nchunk), dim3(256), 0, g_stream, c.start, c.len, c.sub, c.suboff, inv,
                     base, B, ldb, X, ldx, m);
}

// =============================================================================== replicated coarse solve
// x = (L L^T)^-1 b for the replicated coarse operator E (dimE <= 1024), ONE workgroup, in place on the device: the two
// triangular sweeps of applyQ's coarse solve (geneo.cpp:1474-1513: KSPSolve(pcKSPL2)) stream-ordered behind the all-reduce
// of Z^T x instead of a download, two host sweeps and an upload.  Thread t owns unknown t in a
// register.  Column-oriented sweeps: step k publishes x_k through LDS (one barrier) and every later unknown subtracts
// L_tk x_k -- for unknown t the subtractions come in the order k = 0, 1, .. (forward) and k = n - 1, n - 2, .. (backward),
// exactly dense::cholesky_solve_lu's, so host and device produce the same bits.  The matrix entries of PB steps are
// loaded together (PB loads in flight per thread: column k of L is row k of L^T and vice versa, contiguous in t).
template <int PB>
__global__ __launch_bounds__(1024) void k_chol_solve(const double* __restrict__ L, const double* __restrict__ LT, int n,
                                                     double* __restrict__ y) {
  extern __shared__ double xs[];   // n published unknowns
  const int t = threadIdx.x;
  double yv = (t < n) ? y[t] : 0.0;
  const double dg = (t < n) ? L[(int64_t)t * n + t] : 1.0;
  for (int k0 = 0; k0 < n; k0 += PB) {                 // L z = b
    double c[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int k = k0 + p;
      c[p] = (k < n && t > k && t < n) ? LT[(int64_t)k * n + t] : 0.0;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int k = k0 + p;
      if (k < n) {                                     // (uniform)
        if (t == k) { yv = yv / dg; xs[k] = yv; }
        __syncthreads();
        if (t > k) yv -= c[p] * xs[k];
      }
    }
  }
  for (int k0 = 0; k0 < n; k0 += PB) {                 // L^T x = z, from the last unknown down
    double c[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int k = n - 1 - (k0 + p);
      c[p] = (k >= 0 && t < k) ? L[(int64_t)k * n + t] : 0.0;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int k = n - 1 - (k0 + p);
      if (k >= 0) {
        if (t == k) { yv = yv / dg; xs[k] = yv; }
        __syncthreads();
        if (t < k) yv -= c[p] * xs[k];
      }
    }
  }
  if (t < n) y[t] = yv;
}
bool chol_solve(const double* L, const double* LT, int n, double* y) {
  if (n <= 0) return true;
  if (n > 1024) return false;
  const int threads = ((n + 63) / 64) * 64;
  hipLaunchKernelGGL((k_chol_solve<16>), dim3(1), dim3(threads), sizeof(double) * n, g_stream, L, LT, n, y);
  return true;
}

// =============================================================================== self test / events
// A = I(16x16 in the first 4 k-steps pattern) check of the f64 MFMA operand/result maps.
__global__ void k_mfma_selftest(double* out) {
  const int l = threadIdx.x;
  // C = A(16x4) * B(4x16) with A[i][k] = (i == k), B[k][j] = 100*k + j  ->  C[i][j] = B[i][j] for i<4 else 0
  const int i = l & 15, k = l >> 4;
  const double a = (i == k) ? 1.0 : 0.0;
  const double b = 100.0 * k + (l & 15);
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) {
    const int row = (l >> 4) + 4 * v, colj = l & 15;
    out[row * 16 + colj] = acc[v];
  }
}
void set_mfma(bool enable) { lazy_init(); g_no_mfma = !enable; }
bool set_variant(const char* name, int value) {
  const std::string k(name ? name : "");
  if (k == "spgemm_fill_scan") { g_spgemm_fill_scan = value != 0; return true; }
  if (k == "gram_flat") { g_gram_flat = value != 0; return true; }
  if (k == "lp_fixed") {
    lazy_init();
    const int off = value ? 0 : 1;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lp_no_fixed), &off, sizeof(int)));
    return true;
  }
  if (k == "spgemm_small_rows") {
    lazy_init();
    const int off = value ? 0 : 1;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_spgemm_no_small), &off, sizeof(int)));
    g_spgemm_small_host = value != 0;
    return true;
  }
  return false;
}
int selftest_mfma_f64() {
  double* d = (double*)alloc(sizeof(double) * 256);
  hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, g_stream, d);
  double h[256];
  d2h(h, d, sizeof(h));
  dfree(d);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      const double want = (i < 4) ? 100.0 * i + j : 0.0;
      if (h[i * 16 + j] != want) return 1 + i * 16 + j;
    }
  return 0;
}

void* event_create() {
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  return (void*)e;
}
void event_record(void* ev) { HIPCHK(hipEventRecord((hipEvent_t)ev, g_stream)); }
static hipStream_t g_copy_stream = nullptr;
void d2h_after(void* host, const void* dev, size_t bytes, void* event) {
  if (!bytes) return;
  if (!g_copy_stream) HIPCHK(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamWaitEvent(g_copy_stream, (hipEvent_t)event, 0));
  HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, g_copy_stream));
  HIPCHK(hipStreamSynchronize(g_copy_stream));
}
float event_elapsed_ms(void* a, void* b) {
  HIPCHK(hipEventSynchronize((hipEvent_t)b));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
  return ms;
}
void event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

}  // namespace bk
