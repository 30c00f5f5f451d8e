// Host-side Rayleigh-Ritz cost on the machine it runs on: gen_eig_rr on a 96 x 96 pencil, alone and on 8 threads the way
// lobpcg_solve() issues it (thread creation included).   hipcc -O3 -mavx2 -mfma -std=c++17 -x c++ -I geneo4petsc_amd/csrc
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>
#include "dense.h"
static double now() { return std::chrono::duration<double>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }
int main() {
  const int p = 96, n = 400, ns = 8;
  std::mt19937 g(1);
  std::normal_distribution<double> nd;
  std::vector<double> S((size_t)n * p), D(n);
  for (auto& v : S) v = nd(g);
  for (int i = 0; i < n; ++i) D[i] = 0.1 + i;
  std::vector<double> GA((size_t)p * p, 0), GB((size_t)p * p, 0);
  for (int a = 0; a < p; ++a)
    for (int b = 0; b < p; ++b) {
      double sa = 0, sb = 0;
      for (int i = 0; i < n; ++i) { sa += S[i * p + a] * D[i] * S[i * p + b]; sb += S[i * p + a] * S[i * p + b]; }
      GA[a * p + b] = sa; GB[a * p + b] = sb;
    }
  std::vector<double> th, C;
  double t0 = now();
  for (int k = 0; k < 50; ++k) dense::gen_eig_rr(GA, GB, p, 32, 1e-12, th, C);
  printf("gen_eig_rr alone            %.3f ms\n", (now() - t0) / 50 * 1e3);
  std::vector<double> M, w, V;
  t0 = now();
  for (int k = 0; k < 50; ++k) { M = GA; dense::sym_eig(M, p, w, V); }
  printf("sym_eig alone               %.3f ms\n", (now() - t0) / 50 * 1e3);
  t0 = now();
  for (int k = 0; k < 50; ++k) {
    std::vector<std::thread> tt;
    for (int t = 0; t < ns; ++t) tt.emplace_back([&]() { std::vector<double> a, c; dense::gen_eig_rr(GA, GB, p, 32, 1e-12, a, c); });
    for (auto& x : tt) x.join();
  }
  printf("8 threads x gen_eig_rr      %.3f ms per round (spawn + join included)\n", (now() - t0) / 50 * 1e3);
  t0 = now();
  for (int k = 0; k < 50; ++k) {
    std::vector<std::thread> tt;
    for (int t = 0; t < ns; ++t) tt.emplace_back([]() {});
    for (auto& x : tt) x.join();
  }
  printf("8 empty threads             %.3f ms per round\n", (now() - t0) / 50 * 1e3);
  printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
}
