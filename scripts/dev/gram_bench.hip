// Micro-benchmark of LOBPCG's W-row Gram launch (gram2: [A W | B W]^T S, p = 64, q = 96, k_gram_mfma<2,3,2>) against
// candidate forms of the same arithmetic.  Every variant must reproduce the baseline's partial Gram matrices bit for bit
// (per output tile the same sequence of 4-row MFMA steps in ascending row order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/dev/gram_bench.hip -o scripts/dev/bin/gram_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using d4 = __attribute__((ext_vector_type(4))) double;
using d2 = __attribute__((ext_vector_type(2))) double;

// ---------------------------------------------------------------- baseline: the product kernel, TI = 2, TJ = 3, NC = 2
template <int TI, int TJ, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_gram_base(
    const int* __restrict__ cstart, const int* __restrict__ clen, const double* __restrict__ S, int lds_, int p,
    const double* __restrict__ T, int ldt_, int q, double* __restrict__ Gpart, const double* __restrict__ S2, int lds2_,
    int psplit) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ldS = (p % 32 == 0) ? p + 16 : p;
  const int ldT = (q % 32 == 0) ? q + 16 : q;
  double* sS = smem;
  double* sT = smem + 16 * ldS;
  const int g = blockIdx.x;
  const int row0 = cstart[g];
  const int nrows = clen[g];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int P16 = p >> 4, Q16 = q >> 4;
  const int wi = w >> 1, wj = w & 1;
  int aoff[TI], boff[TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    int I = wi * TI + a;
    if (I >= P16) I = P16 - 1;
    aoff[a] = 16 * I + (l & 15);
  }
#pragma unroll
  for (int b = 0; b < TJ; ++b) {
    int J = wj * TJ + b;
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
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const int cc = l + 64 * ci;
        if (cc < p) sS[(4 * w + u) * ldS + cc] = rs[u][ci];
        if (cc < q) sT[(4 * w + u) * ldT + cc] = rt[u][ci];
      }
    __syncthreads();
    if (r + 16 < nrows) load_slab(r + 16);
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
    const int I = wi * TI + a;
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int J = wj * TJ + b;
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

// ---------------------------------------------------------------- W-row Gram specialised: p = 32 + 32 (two buffers), q = 96
// SR-row slabs, 16-byte loads: the right operand's slab is one contiguous block when ldt == 96 (6 * SR / 32 units per
// thread), each left block is SR rows x 16 units.  NBUF = 2: two LDS slabs, one barrier per slab.
template <int SR, int NBUF, bool NTL, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void k_gram_w(
    const int* __restrict__ cstart, const int* __restrict__ clen, const double* __restrict__ A1, const double* __restrict__ A2,
    int lda, const double* __restrict__ T, double* __restrict__ Gpart) {
  constexpr int TI = 2, TJ = 3, p = 64, q = 96, ldS = 80, ldT = 112;
  constexpr int NUT = SR * 48 / 256;      // 16-byte units of the right slab per thread
  constexpr int NUS = SR * 16 / 256;      // units of each left block per thread
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int g = blockIdx.x;
  const int row0 = cstart[g], nrows = clen[g];
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
  d2 rt[NUT], r1[NUS], r2[NUS];
  auto ld = [&](const d2* ptr) { return NTL ? __builtin_nontemporal_load(ptr) : *ptr; };
  auto load_slab = [&](int r) {
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    const d2* tb = reinterpret_cast<const d2*>(T + (int64_t)(row0 + r) * q);
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      const int u = tid + 256 * j;
      rt[j] = (u < nr * 48) ? ld(tb + u) : d2{0.0, 0.0};
    }
#pragma unroll
    for (int j = 0; j < NUS; ++j) {
      const int u = tid + 256 * j;
      const int rr = u >> 4, cu = u & 15;
      const bool ok = rr < nr;
      const int64_t off = (int64_t)(row0 + r + (ok ? rr : 0)) * lda + 2 * cu;
      r1[j] = ok ? ld(reinterpret_cast<const d2*>(A1 + off)) : d2{0.0, 0.0};
      r2[j] = ok ? ld(reinterpret_cast<const d2*>(A2 + off)) : d2{0.0, 0.0};
    }
  };
  auto stage = [&](double* sS, double* sT) {
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      const int u = tid + 256 * j;
      *reinterpret_cast<d2*>(sT + (u / 48) * ldT + 2 * (u % 48)) = rt[j];
    }
#pragma unroll
    for (int j = 0; j < NUS; ++j) {
      const int u = tid + 256 * j;
      const int rr = u >> 4, cu = u & 15;
      *reinterpret_cast<d2*>(sS + rr * ldS + 2 * cu) = r1[j];
      *reinterpret_cast<d2*>(sS + rr * ldS + 32 + 2 * cu) = r2[j];
    }
  };
  auto compute = [&](const double* sS, const double* sT, int nr) {
    const int nstep = 4 * ((nr + 15) / 16);       // the baseline runs whole 16-row slabs (zero rows included)
#pragma unroll
    for (int step = 0; step < SR / 4; ++step) {
      if (step < nstep) {
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
  };
  constexpr int SLAB = SR * (ldS + ldT);
  load_slab(0);
  if (NBUF == 1) {
    for (int r = 0; r < nrows; r += SR) {
      __syncthreads();
      stage(smem, smem + SR * ldS);
      __syncthreads();
      if (r + SR < nrows) load_slab(r + SR);
      compute(smem, smem + SR * ldS, (nrows - r < SR) ? nrows - r : SR);
    }
  } else {
    int buf = 0;
    stage(smem, smem + SR * ldS);
    if (SR < nrows) load_slab(SR);
    __syncthreads();
    for (int r = 0; r < nrows; r += SR) {
      double* cur = smem + buf * SLAB;
      double* nxt = smem + (buf ^ 1) * SLAB;
      if (r + SR < nrows) {
        stage(nxt, nxt + SR * ldS);                  // the other buffer: its readers passed the barrier below
        if (r + 2 * SR < nrows) load_slab(r + 2 * SR);
      }
      compute(cur, cur + SR * ldS, (nrows - r < SR) ? nrows - r : SR);
      __syncthreads();
      buf ^= 1;
    }
  }
  double* G = Gpart + (int64_t)g * p * q;
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    const int I = wi * TI + a;
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int J = wj * TJ + b;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * I + (l >> 4) + 4 * v, colj = 16 * J + (l & 15);
        G[(int64_t)row * q + colj] = acc[a][b][v];
      }
    }
  }
}

// read-only stream of the same bytes (ceiling of the pattern)
__global__ __launch_bounds__(256) void k_read_pattern(int64_t n, const double* __restrict__ A1, const double* __restrict__ A2,
                                                      const double* __restrict__ T, double* __restrict__ out) {
  d2 s = d2{0.0, 0.0};
  const int64_t total = n * 48;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t row = u / 48;
    const int cu = (int)(u - row * 48);
    s += __builtin_nontemporal_load(reinterpret_cast<const d2*>(T) + u);
    if (cu >= 32) {
      s += __builtin_nontemporal_load(reinterpret_cast<const d2*>(A1) + u);
      s += __builtin_nontemporal_load(reinterpret_cast<const d2*>(A2) + u);
    }
  }
  if (s.x == 1e300) out[0] = s.y;
}

__global__ void k_diff(const double* a, const double* b, int64_t n, unsigned long long* cnt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long local = 0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (reinterpret_cast<const uint64_t*>(a)[i] != reinterpret_cast<const uint64_t*>(b)[i]) ++local;
  if (local) atomicAdd(cnt, local);
}
__global__ void k_fill(double* a, int64_t n, uint64_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    a[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

int main(int argc, char** argv) {
  const int nsub_rows = argc > 1 ? atoi(argv[1]) : 286336;
  const int nsub = argc > 2 ? atoi(argv[2]) : 8;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int CHUNK = 1024;
  std::vector<int> st, ln;
  for (int s = 0; s < nsub; ++s)
    for (int a = 0; a < nsub_rows; a += CHUNK) {
      st.push_back(s * nsub_rows + a);
      ln.push_back(std::min(CHUNK, nsub_rows - a));
    }
  const int nchunk = (int)st.size();
  const int64_t n = (int64_t)nsub * nsub_rows;
  int *dst_, *dln;
  CK(hipMalloc(&dst_, 4 * nchunk)); CK(hipMalloc(&dln, 4 * nchunk));
  CK(hipMemcpy(dst_, st.data(), 4 * nchunk, hipMemcpyHostToDevice));
  CK(hipMemcpy(dln, ln.data(), 4 * nchunk, hipMemcpyHostToDevice));
  double *S, *AS, *BS, *G[2];
  const size_t nb = sizeof(double) * (size_t)n * 96;
  CK(hipMalloc(&S, nb)); CK(hipMalloc(&AS, nb)); CK(hipMalloc(&BS, nb));
  const size_t gb = sizeof(double) * (size_t)nchunk * 64 * 96;
  CK(hipMalloc(&G[0], gb)); CK(hipMalloc(&G[1], gb));
  k_fill<<<4096, 256>>>(S, n * 96, 1); k_fill<<<4096, 256>>>(AS, n * 96, 2); k_fill<<<4096, 256>>>(BS, n * 96, 3);
  CK(hipDeviceSynchronize());
  unsigned long long* cnt;
  CK(hipMalloc(&cnt, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 8.0 * (double)n * (64 + 96), flops = 2.0 * (double)n * 64 * 96;
  printf("rows %lld  chunks %d  algorithmic bytes %.3f GB  flops %.2f G\n", (long long)n, nchunk, bytes * 1e-9, flops * 1e-9);
  auto timeit = [&](const char* name, auto launch, bool check) {
    CK(hipMemset(G[1], 0, gb));
    launch(1);
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    if (check) {
      CK(hipMemset(cnt, 0, 8));
      k_diff<<<1024, 256>>>(G[0], G[1], (int64_t)nchunk * 64 * 96, cnt);
      CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
    }
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch(1);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps; sum += ms; if (ms < best) best = ms;
    }
    printf("%-44s best %.4f ms  avg %.4f ms  %.3f TB/s (%.3f of 8)  %.1f TFLOP/s (%.3f of 78.6)  differing %llu\n", name, best,
           sum / 3, bytes / best * 1e-9, bytes / best * 1e-9 / 8.0, flops / best * 1e-9, flops / best * 1e-9 / 78.6, bad);
    fflush(stdout);
  };
  const size_t sm0 = sizeof(double) * 16 * (80 + 112);
  k_gram_base<2, 3, 2><<<nchunk, 256, sm0>>>(dst_, dln, AS + 64, 96, 64, S, 96, 96, G[0], BS + 64, 96, 32);
  CK(hipDeviceSynchronize());
  timeit("baseline k_gram_mfma<2,3,2>", [&](int k) { k_gram_base<2, 3, 2><<<nchunk, 256, sm0>>>(dst_, dln, AS + 64, 96, 64, S, 96, 96, G[k], BS + 64, 96, 32); }, true);
  timeit("read-only stream of the same bytes", [&](int k) { k_read_pattern<<<2048, 256>>>(n, AS, BS, S, G[k]); }, false);
#define RUN(SR, NB, NT, WPE)                                                                                              \
  do {                                                                                                                    \
    const size_t sm = sizeof(double) * SR * (80 + 112) * NB;                                                              \
    CK(hipFuncSetAttribute((const void*)k_gram_w<SR, NB, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
    timeit("slab " #SR " lds-buffers " #NB " nt " #NT " waves/eu " #WPE, [&](int k) {                                     \
      k_gram_w<SR, NB, NT, WPE><<<nchunk, 256, sm>>>(dst_, dln, AS + 64, BS + 64, 96, S, G[k]); }, true);                  \
  } while (0)
  RUN(16, 1, false, 3);
  RUN(16, 1, true, 3);
  RUN(32, 1, true, 3);
  RUN(16, 2, true, 3);
  RUN(32, 2, true, 2);
  RUN(16, 2, true, 2);
  RUN(16, 2, true, 4);
  RUN(32, 1, true, 2);
  RUN(64, 1, true, 2);
  return 0;
}
