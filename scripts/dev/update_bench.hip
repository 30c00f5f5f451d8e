// Micro-benchmark of the fused LOBPCG update (k_lobpcg_update32) against candidate forms of the same arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/dev/update_bench.hip -o /tmp/update_bench && /tmp/update_bench [rows per subdomain] [nsub]
// Every variant must reproduce the baseline's output BIT FOR BIT (same MFMA sequence per output tile); the program
// prints per variant the average launch time, the algorithmic TB/s (8 n (3*96 + 3*64 + 32) bytes) and the number of
// differing doubles.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using d4 = __attribute__((ext_vector_type(4))) double;
using d2 = __attribute__((ext_vector_type(2))) double;

// ---------------------------------------------------------------- baseline (copy of the product kernel's body)
__global__ __launch_bounds__(256) void k_base(const int* __restrict__ cstart, const int* __restrict__ clen,
                                              const int* __restrict__ csub, const double* __restrict__ S,
                                              const double* __restrict__ AS, const double* __restrict__ BS,
                                              const double* __restrict__ C, const double* __restrict__ keep,
                                              const double* __restrict__ lam, const double* __restrict__ mask,
                                              double* __restrict__ T, double* __restrict__ AT,
                                              double* __restrict__ BT, double* __restrict__ R) {
  constexpr int m = 32, p = 96, q = 64, ldS = p + 1, SR = 32;
  __shared__ double sS[SR * ldS];
  const int c = blockIdx.x;
  const int row0 = cstart[c], nrows = clen[c], sd = csub[c];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int rt = w >> 1, ct = w & 1;
  const double* Cs = C + (int64_t)sd * p * q;
  const int col = 16 * ct + (l & 15);
  double bx[8], bw[16];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) bx[kk] = Cs[(int64_t)(4 * kk + (l >> 4)) * q + col];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) bw[kk] = Cs[(int64_t)(m + 4 * kk + (l >> 4)) * q + col];
  const double kp = keep[sd * m + col], lm = lam[sd * m + col], mk = mask ? mask[sd * m + col] : 1.0;
  const double* src[3] = {S, AS, BS};
  double* dst[3] = {T, AT, BT};
  double rg[SR / 4][2];
  auto load_slab = [&](int t) {
    const double* base = src[t % 3];
    const int r = (t / 3) * SR;
#pragma unroll
    for (int u = 0; u < SR / 4; ++u) {
      const int rr = r + (SR / 4) * w + u;
      const bool ok = rr < nrows;
      const double* srow = base + (int64_t)(row0 + (ok ? rr : 0)) * p;
      rg[u][0] = ok ? srow[l] : 0.0;
      rg[u][1] = (ok && l < 32) ? srow[64 + l] : 0.0;
    }
  };
  const int nt = 3 * ((nrows + SR - 1) / SR);
  load_slab(0);
  d4 ax = (d4){0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < nt; ++t) {
    const int op = t % 3, r = (t / 3) * SR;
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SR / 4; ++u) {
      sS[((SR / 4) * w + u) * ldS + l] = rg[u][0];
      if (l < 32) sS[((SR / 4) * w + u) * ldS + 64 + l] = rg[u][1];
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
        if (rr < nr) out[(int64_t)(row0 + r + rr) * p + m + col] = kp * acc[v];
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * kk], bx[kk], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * rt + (l >> 4) + 4 * v;
        if (rr < nr) out[(int64_t)(row0 + r + rr) * p + col] = acc[v];
      }
      if (op == 1) ax = acc;
      if (op == 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * rt + (l >> 4) + 4 * v;
          if (rr < nr) R[(int64_t)(row0 + r + rr) * m + col] = mk * (ax[v] - lm * acc[v]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- flat 16-byte slab loads, optional split / depth / nt
// A slab of 32 rows is ONE contiguous 24 KiB block (ld = 96): thread t copies the 16-byte units t, t + 256, ... (6 per
// thread).  DEPTH = slabs in flight (1 or 2).  split = parts a chunk is cut into (more, shorter workgroups).
template <int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_flat(const int* __restrict__ cstart, const int* __restrict__ clen,
                                              const int* __restrict__ csub, const double* __restrict__ S,
                                              const double* __restrict__ AS, const double* __restrict__ BS,
                                              const double* __restrict__ C, const double* __restrict__ keep,
                                              const double* __restrict__ lam, const double* __restrict__ mask,
                                              double* __restrict__ T, double* __restrict__ AT,
                                              double* __restrict__ BT, double* __restrict__ R, int split) {
  constexpr int m = 32, p = 96, q = 64, ldS = p + 1, SR = 32, NU = 6;
  __shared__ double sS[SR * ldS];
  const int c = blockIdx.x / split, part = blockIdx.x - c * split;
  const int cn = clen[c], sd = csub[c];
  const int per = ((cn + split * SR - 1) / (split * SR)) * SR;
  const int rb = part * per;
  if (rb >= cn) return;
  const int nrows = (cn - rb < per) ? cn - rb : per;
  const int row0 = cstart[c] + rb;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int rt = w >> 1, ct = w & 1;
  const double* Cs = C + (int64_t)sd * p * q;
  const int col = 16 * ct + (l & 15);
  double bx[8], bw[16];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) bx[kk] = Cs[(int64_t)(4 * kk + (l >> 4)) * q + col];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) bw[kk] = Cs[(int64_t)(m + 4 * kk + (l >> 4)) * q + col];
  const double kp = keep[sd * m + col], lm = lam[sd * m + col], mk = mask ? mask[sd * m + col] : 1.0;
  const double* src[3] = {S, AS, BS};
  double* dst[3] = {T, AT, BT};
  int loff[NU];                                  // LDS offset of this thread's j-th unit
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int u = tid + 256 * j;
    loff[j] = (u / 48) * ldS + 2 * (u % 48);
  }
  d2 rg[DEPTH][NU];
  const int nt = 3 * ((nrows + SR - 1) / SR);
  auto load_slab = [&](int t, d2* dstreg) {
    const int r = (t / 3) * SR;
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    const d2* base = reinterpret_cast<const d2*>(src[t % 3] + (int64_t)(row0 + r) * p);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = tid + 256 * j;
      const bool ok = u < nr * 48;
      d2 v = d2{0.0, 0.0};
      if (ok) v = NTL ? __builtin_nontemporal_load(base + u) : base[u];
      dstreg[j] = v;
    }
  };
  auto stage = [&](const d2* reg) {
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      sS[loff[j]] = reg[j].x;
      sS[loff[j] + 1] = reg[j].y;
    }
  };
  d4 ax = (d4){0.0, 0.0, 0.0, 0.0};
  auto compute = [&](int t) {
    const int op = t % 3, r = (t / 3) * SR;
    const int nr = (nrows - r < SR) ? nrows - r : SR;
    if (16 * rt < nr) {
      const double* arow = sS + (16 * rt + (l & 15)) * ldS + (l >> 4);
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[m + 4 * kk], bw[kk], acc, 0, 0, 0);
      double* out = dst[op];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * rt + (l >> 4) + 4 * v;
        if (rr < nr) {
          double* o = out + (int64_t)(row0 + r + rr) * p + m + col;
          if (NTS) __builtin_nontemporal_store(kp * acc[v], o); else *o = kp * acc[v];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * kk], bx[kk], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * rt + (l >> 4) + 4 * v;
        if (rr < nr) {
          double* o = out + (int64_t)(row0 + r + rr) * p + col;
          if (NTS) __builtin_nontemporal_store(acc[v], o); else *o = acc[v];
        }
      }
      if (op == 1) ax = acc;
      if (op == 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * rt + (l >> 4) + 4 * v;
          if (rr < nr) {
            double* o = R + (int64_t)(row0 + r + rr) * m + col;
            const double val = mk * (ax[v] - lm * acc[v]);
            if (NTS) __builtin_nontemporal_store(val, o); else *o = val;
          }
        }
      }
    }
  };
  if (DEPTH == 1) {
    load_slab(0, rg[0]);
    for (int t = 0; t < nt; ++t) {
      __syncthreads();
      stage(rg[0]);
      __syncthreads();
      if (t + 1 < nt) load_slab(t + 1, rg[0]);
      compute(t);
    }
  } else {
    load_slab(0, rg[0]);
    if (1 < nt) load_slab(1, rg[DEPTH - 1]);
    for (int t = 0; t < nt; t += 2) {
      __syncthreads();
      stage(rg[0]);
      __syncthreads();
      if (t + 2 < nt) load_slab(t + 2, rg[0]);
      compute(t);
      if (t + 1 < nt) {
        __syncthreads();
        stage(rg[DEPTH - 1]);
        __syncthreads();
        if (t + 3 < nt) load_slab(t + 3, rg[DEPTH - 1]);
        compute(t + 1);
      }
    }
  }
}

// ---------------------------------------------------------------- the same bytes as a plain copy (ceiling of the pattern)
__global__ __launch_bounds__(256) void k_copy_pattern(int64_t n, const double* __restrict__ S, const double* __restrict__ AS,
                                                      const double* __restrict__ BS, double* __restrict__ T,
                                                      double* __restrict__ AT, double* __restrict__ BT,
                                                      double* __restrict__ R) {
  // one thread per 16-byte unit of a row block: reads 48 units per row and operand, writes the first 32 units (X' P') of
  // each operand and 16 units of R
  const d2* src[3] = {reinterpret_cast<const d2*>(S), reinterpret_cast<const d2*>(AS), reinterpret_cast<const d2*>(BS)};
  d2* dst[3] = {reinterpret_cast<d2*>(T), reinterpret_cast<d2*>(AT), reinterpret_cast<d2*>(BT)};
  d2* r2 = reinterpret_cast<d2*>(R);
  const int64_t total = n * 48;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t row = u / 48;
    const int cu = (int)(u - row * 48);
    d2 a = src[0][u], b = src[1][u], c = src[2][u];
    if (cu < 32) {
      dst[0][u] = a;
      dst[1][u] = b;
      dst[2][u] = c;
      if (cu < 16) r2[row * 16 + cu] = b - c;
    } else if (a.x == 1e300 && b.x == 1e300 && c.x == 1e300) {
      r2[0] = a;     // keeps the loads of the W columns alive
    }
  }
}

__global__ void k_diff(const double* a, const double* b, int64_t n, unsigned long long* cnt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long local = 0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t x = reinterpret_cast<const uint64_t*>(a)[i], y = reinterpret_cast<const uint64_t*>(b)[i];
    if (x != y) ++local;
  }
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
  std::vector<int> st, ln, sb;
  for (int s = 0; s < nsub; ++s)
    for (int a = 0; a < nsub_rows; a += CHUNK) {
      st.push_back(s * nsub_rows + a);
      ln.push_back(std::min(CHUNK, nsub_rows - a));
      sb.push_back(s);
    }
  const int nchunk = (int)st.size();
  const int64_t n = (int64_t)nsub * nsub_rows;
  int *dst_, *dln, *dsb;
  CK(hipMalloc(&dst_, 4 * nchunk)); CK(hipMalloc(&dln, 4 * nchunk)); CK(hipMalloc(&dsb, 4 * nchunk));
  CK(hipMemcpy(dst_, st.data(), 4 * nchunk, hipMemcpyHostToDevice));
  CK(hipMemcpy(dln, ln.data(), 4 * nchunk, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsb, sb.data(), 4 * nchunk, hipMemcpyHostToDevice));
  double *S, *AS, *BS, *C, *keep, *lam, *mask, *T[2], *AT[2], *BT[2], *R[2];
  const size_t nb = sizeof(double) * (size_t)n * 96;
  CK(hipMalloc(&S, nb)); CK(hipMalloc(&AS, nb)); CK(hipMalloc(&BS, nb));
  for (int k = 0; k < 2; ++k) {
    CK(hipMalloc(&T[k], nb)); CK(hipMalloc(&AT[k], nb)); CK(hipMalloc(&BT[k], nb));
    CK(hipMalloc(&R[k], sizeof(double) * (size_t)n * 32));
    CK(hipMemset(T[k], 0, nb)); CK(hipMemset(AT[k], 0, nb)); CK(hipMemset(BT[k], 0, nb));
    CK(hipMemset(R[k], 0, sizeof(double) * (size_t)n * 32));
  }
  CK(hipMalloc(&C, sizeof(double) * nsub * 96 * 64));
  CK(hipMalloc(&keep, 8 * nsub * 32)); CK(hipMalloc(&lam, 8 * nsub * 32)); CK(hipMalloc(&mask, 8 * nsub * 32));
  k_fill<<<4096, 256>>>(S, n * 96, 1); k_fill<<<4096, 256>>>(AS, n * 96, 2); k_fill<<<4096, 256>>>(BS, n * 96, 3);
  k_fill<<<64, 256>>>(C, (int64_t)nsub * 96 * 64, 4);
  k_fill<<<1, 256>>>(keep, nsub * 32, 5); k_fill<<<1, 256>>>(lam, nsub * 32, 6); k_fill<<<1, 256>>>(mask, nsub * 32, 7);
  CK(hipDeviceSynchronize());
  unsigned long long* cnt;
  CK(hipMalloc(&cnt, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 8.0 * (double)n * (3 * 96 + 3 * 64 + 32);
  printf("rows %lld  chunks %d  algorithmic bytes %.3f GB\n", (long long)n, nchunk, bytes * 1e-9);

  auto timeit = [&](const char* name, auto launch, bool check) {
    launch(1);
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    if (check) {
      CK(hipMemset(cnt, 0, 8));
      k_diff<<<4096, 256>>>(T[0], T[1], n * 96, cnt); k_diff<<<4096, 256>>>(AT[0], AT[1], n * 96, cnt);
      k_diff<<<4096, 256>>>(BT[0], BT[1], n * 96, cnt); k_diff<<<4096, 256>>>(R[0], R[1], n * 32, cnt);
      CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
      CK(hipMemset(T[1], 0, nb)); CK(hipMemset(AT[1], 0, nb)); CK(hipMemset(BT[1], 0, nb));
      CK(hipMemset(R[1], 0, sizeof(double) * (size_t)n * 32));
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
    printf("%-34s  best %.4f ms  avg %.4f ms  %.3f TB/s  of 8: %.3f   differing doubles %llu\n", name, best, sum / 3,
           bytes / best * 1e-9, bytes / best * 1e-9 / 8.0, bad);
    fflush(stdout);
  };
  // reference output into set 0
  k_base<<<nchunk, 256>>>(dst_, dln, dsb, S, AS, BS, C, keep, lam, mask, T[0], AT[0], BT[0], R[0]);
  CK(hipDeviceSynchronize());
  timeit("baseline", [&](int k) { k_base<<<nchunk, 256>>>(dst_, dln, dsb, S, AS, BS, C, keep, lam, mask, T[k], AT[k], BT[k], R[k]); }, true);
  timeit("copy of the same pattern", [&](int k) { k_copy_pattern<<<256 * 8, 256>>>(n, S, AS, BS, T[k], AT[k], BT[k], R[k]); }, false);
  CK(hipMemset(T[1], 0, nb)); CK(hipMemset(AT[1], 0, nb)); CK(hipMemset(BT[1], 0, nb));
  CK(hipMemset(R[1], 0, sizeof(double) * (size_t)n * 32));
#define RUN(D, NL, NS, SP)                                                                                         \
  timeit("flat depth " #D " ntl " #NL " nts " #NS " split " #SP, [&](int k) {                                      \
    k_flat<D, NL, NS><<<nchunk * SP, 256>>>(dst_, dln, dsb, S, AS, BS, C, keep, lam, mask, T[k], AT[k], BT[k], R[k], SP); }, true)
  // the same launch after an idle gap of the length of LOBPCG's host Rayleigh-Ritz phase (does the chip slow down?)
  for (int gap_us : {0, 100, 300, 700, 1500}) {
    float sum = 0, best = 1e30f;
    const int nrep = 20;
    for (int i = 0; i < nrep; ++i) {
      CK(hipDeviceSynchronize());
      if (gap_us) usleep(gap_us);
      CK(hipEventRecord(e0));
      k_flat<1, true, true><<<nchunk, 256>>>(dst_, dln, dsb, S, AS, BS, C, keep, lam, mask, T[1], AT[1], BT[1], R[1], 1);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      sum += ms; if (ms < best) best = ms;
    }
    printf("flat nt after an idle gap of %4d us: best %.4f ms  avg %.4f ms\n", gap_us, best, sum / nrep);
  }
  RUN(1, false, false, 1);
  RUN(1, false, false, 2);
  RUN(1, false, false, 4);
  RUN(1, true, false, 1);
  RUN(1, false, true, 1);
  RUN(1, true, true, 1);
  RUN(1, true, true, 4);
  RUN(2, false, false, 1);
  RUN(2, false, false, 4);
  RUN(2, true, true, 1);
  RUN(2, true, true, 4);
  return 0;
}
