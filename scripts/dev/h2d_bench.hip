// Microbenchmark of the staged pageable -> device upload (scripts/dev: not part of the library).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_bench scripts/dev/h2d_bench.hip -lpthread && /tmp/h2d_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using clk = std::chrono::high_resolution_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
static void par_memcpy(char* dst, const char* src, size_t len, int nt) {
  std::vector<std::thread> th;
  const size_t part = (len / nt + 63) & ~(size_t)63;
  for (int t = 1; t < nt; ++t) {
    const size_t off = std::min(len, part * t), end = std::min(len, part * (t + 1));
    th.emplace_back([=]() { memcpy(dst + off, src + off, end - off); });
  }
  memcpy(dst, src, std::min(len, part));
  for (auto& x : th) x.join();
}
int main() {
  const size_t N = (size_t)512 << 20;
  char* h = (char*)malloc(N);
  memset(h, 1, N);
  char* d; CK(hipMalloc(&d, N));
  hipStream_t s; CK(hipStreamCreate(&s));
  // (1) pinned -> device
  char* p; CK(hipHostMalloc(&p, N, hipHostMallocDefault));
  memset(p, 2, N);
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = clk::now();
    CK(hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    printf("pinned 512 MB -> device: %.1f GB/s\n", N / secs(t0, clk::now()) / 1e9);
  }
  // (1b) pageable -> device, plain hipMemcpy
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = clk::now();
    CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice));
    printf("pageable 512 MB -> device (hipMemcpy): %.1f GB/s\n", N / secs(t0, clk::now()) / 1e9);
  }
  // (2) pageable -> pinned memcpy, nt threads
  for (int nt : {1, 2, 4, 8, 16}) {
    auto t0 = clk::now();
    par_memcpy(p, h, N, nt);
    printf("memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", nt, N / secs(t0, clk::now()) / 1e9);
  }
  // (3) staged pipeline: chunk size x threads
  for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20})
    for (int nt : {4, 8, 16}) {
      hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
      char* b[2] = {p, p + chunk};
      auto t0 = clk::now();
      size_t off = 0;
      for (int i = 0; off < N; ++i, off += chunk) {
        const size_t len = std::min(chunk, N - off);
        if (i >= 2) CK(hipEventSynchronize(ev[i & 1]));
        par_memcpy(b[i & 1], h + off, len, nt);
        CK(hipMemcpyAsync(d + off, b[i & 1], len, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[i & 1], s));
      }
      CK(hipStreamSynchronize(s));
      printf("staged, chunk %3zu MB, %2d threads: %.1f GB/s\n", chunk >> 20, nt, N / secs(t0, clk::now()) / 1e9);
    }
  // (4) hipHostRegister cost
  {
    auto t0 = clk::now();
    hipError_t e = hipHostRegister(h, N, hipHostRegisterDefault);
    double tr = secs(t0, clk::now());
    if (e == hipSuccess) {
      auto t1 = clk::now();
      CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
      double tc = secs(t1, clk::now());
      auto t2 = clk::now();
      CK(hipHostUnregister(h));
      printf("hipHostRegister 512 MB: %.1f ms, copy %.1f GB/s, unregister %.1f ms\n", tr * 1e3, N / tc / 1e9, secs(t2, clk::now()) * 1e3);
    } else printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
  }
  printf("host threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
