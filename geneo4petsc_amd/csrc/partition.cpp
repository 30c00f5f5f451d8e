// Host k-way mesh partitioner of libgeneopc: the stand-in for the METIS_PartMeshDual / METIS_PartMeshNodal calls of the
// reference's driver (src/geneo4PETSc.cpp:381-445; Metis 5.1.0 is a dependency the image does not have).  Same inputs
// (element -> node lists, number of parts, dual: elements adjacent when they share >= 1 node / nodal: nodes adjacent when
// they share an element), same outputs (a part per element and per node), objective = edge cut, k-way by recursive
// bisection.  Pure host C++ (north_star: "subdomains are partitioned with Metis on the host"); no device code.
//
// One bisection = the multilevel scheme Metis itself uses (Karypis & Kumar, SIAM J. Sci. Comput. 20, 1998):
//   coarsening by heavy-edge matching (light pairs first, no pair heavier than 3x the average vertex);
//   initial cuts on the coarsest graph (<= 200 vertices): greedy graph growing from several seeds and sign cuts of the
//     lowest non-trivial eigenvectors of the weighted Laplacian (dense) and of their pairwise sums / differences;
//   uncoarsening with Fiduccia-Mattheyses boundary refinement (priority queues, hill climbing with roll-back);
// plus a MULTILEVEL SPECTRAL candidate (Barnard & Simon): the three lowest non-trivial eigenvectors of the coarsest
// Laplacian are prolonged through the same hierarchy and smoothed with damped-Jacobi sweeps, the direction in their span
// with the smallest cut is searched on an intermediate level (symmetric domains have a multiple Fiedler value: a cube's
// is threefold and the planar cut is a combination), the resulting scalar field is carried to the finest level, split at
// its quantile and refined.  The smaller of the two cuts wins.  Sizes are exact to one vertex per bisection.
//
// Deterministic: fixed visiting orders, fixed seeds, threads only in order-independent loops (smoothing sweeps write
// disjoint rows; cut counts are integers).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <queue>
#include <thread>
#include <vector>

#include "dense.h"

namespace {

struct Graph {
  int n = 0;
  std::vector<int> xadj, adj, ew;   // CSR, symmetric, no self loops; integer edge weights
  std::vector<int> vw;              // vertex weights
  int64_t total_vw() const { return std::accumulate(vw.begin(), vw.end(), (int64_t)0); }
};

struct Timers { double coarsen = 0, init = 0, fm = 0, spectral_smooth = 0, spectral_dir = 0, sub = 0, graph = 0; };
Timers g_t;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Tick { double& acc; double t0; explicit Tick(double& a) : acc(a), t0(now_s()) {} ~Tick() { acc += now_s() - t0; } };

int host_threads() {
  static const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  return nt;
}
// f(r0, r1) over row ranges; serial below `min_rows`
void parallel_rows(int n, int min_rows, const std::function<void(int, int)>& f) {
  const int nt = n < min_rows ? 1 : host_threads();
  if (nt == 1) { f(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { f((int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt)); });
  for (auto& x : th) x.join();
}

// ------------------------------------------------------------------------------ coarsening
// heavy-edge matching; returns the coarse graph and the fine -> coarse map
Graph coarsen(const Graph& g, std::vector<int>& cmap) {
  const int n = g.n;
  const int64_t tot = g.total_vw();
  const int64_t cap = std::max<int64_t>(2, 3 * tot / std::max(1, n));        // no pair heavier than 3x the average vertex
  std::vector<int> match(n, -1);
  // visiting order: a fixed pseudo-random permutation (stride walk), light vertices are met as often as heavy ones
  std::vector<int> order(n);
  {
    // blocks of 16 consecutive vertices in shuffled order, shuffled inside: as unbiased as a full shuffle for the
    // matching, but neighbouring visits share cache lines of the adjacency arrays
    const int B = 16, nb = (n + B - 1) / B;
    std::vector<int> blocks(nb);
    std::iota(blocks.begin(), blocks.end(), 0);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](uint64_t m) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (int)(s % m); };
    for (int i = nb - 1; i > 0; --i) std::swap(blocks[i], blocks[rnd((uint64_t)(i + 1))]);
    int p = 0;
    for (int b : blocks) {
      const int r0 = b * B, r1 = std::min(n, r0 + B), p0 = p;
      for (int v = r0; v < r1; ++v) order[p++] = v;
      for (int i = p - 1; i > p0; --i) std::swap(order[i], order[p0 + rnd((uint64_t)(i - p0 + 1))]);
    }
  }
  for (int v : order) {
    if (match[v] >= 0) continue;
    int best = -1;
    double bw = -1.0;
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
      const int u = g.adj[k];
      if (match[u] >= 0 || (int64_t)g.vw[v] + g.vw[u] > cap) continue;
      const double w = (double)g.ew[k] / ((double)g.vw[v] * (double)g.vw[u]);   // heavy edges between LIGHT vertices first
      if (w > bw) { bw = w; best = u; }
    }
    if (best >= 0) { match[v] = best; match[best] = v; }
    else match[v] = v;
  }
  cmap.assign(n, -1);
  int nc = 0;
  for (int v = 0; v < n; ++v)
    if (cmap[v] < 0) {
      cmap[v] = nc;
      cmap[match[v]] = nc;
      ++nc;
    }
  Graph c;
  c.n = nc;
  c.vw.assign(nc, 0);
  for (int v = 0; v < n; ++v) c.vw[cmap[v]] += g.vw[v];
  c.xadj.assign(nc + 1, 0);
  // members of each coarse vertex in fine order
  std::vector<int> first(nc, -1), second(nc, -1);
  for (int v = 0; v < n; ++v) {
    const int cv = cmap[v];
    if (first[cv] < 0) first[cv] = v;
    else second[cv] = v;
  }
  std::vector<int> pos(nc, -1);
  c.adj.reserve(g.adj.size() / 2 + 16);
  c.ew.reserve(g.adj.size() / 2 + 16);
  for (int cv = 0; cv < nc; ++cv) {
    const int start = (int)c.adj.size();
    for (int m = 0; m < 2; ++m) {
      const int v = m == 0 ? first[cv] : second[cv];
      if (v < 0) continue;
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int cu = cmap[g.adj[k]];
        if (cu == cv) continue;
        if (pos[cu] >= start) c.ew[pos[cu]] += g.ew[k];
        else {
          pos[cu] = (int)c.adj.size();
          c.adj.push_back(cu);
          c.ew.push_back(g.ew[k]);
        }
      }
    }
    c.xadj[cv + 1] = (int)c.adj.size();
  }
  return c;
}

// ------------------------------------------------------------------------------ refinement
int64_t cut_of(const Graph& g, const std::vector<char>& side) {
  int64_t c = 0;
  for (int v = 0; v < g.n; ++v)
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
      if (side[v] != side[g.adj[k]]) c += g.ew[k];
  return c / 2;
}

// Fiduccia-Mattheyses boundary refinement of a weighted bisection (side 0 = A).  target_a: weight A should hold;
// tol: allowed deviation (in weight).  Hill climbing with roll-back to the best prefix of every pass.
void fm_refine(const Graph& g, std::vector<char>& side, int64_t target_a, int64_t tol, int passes) {
  const int n = g.n;
  std::vector<int> gain(n), stamp(n, 0);
  std::vector<char> locked(n);
  int64_t wa = 0;
  for (int v = 0; v < n; ++v)
    if (!side[v]) wa += g.vw[v];
  typedef std::pair<int, std::pair<int, int>> Ent;     // (gain, (-stamp order, vertex)): max-heap, ties -> lower vertex id
  for (int pass = 0; pass < passes; ++pass) {
    std::priority_queue<Ent> q[2];
    std::fill(locked.begin(), locked.end(), 0);
    auto compute = [&](int v) {
      int ext = 0, in = 0;
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) (side[g.adj[k]] != side[v] ? ext : in) += g.ew[k];
      gain[v] = ext - in;
      return ext;
    };
    for (int v = 0; v < n; ++v) {
      const int ext = compute(v);
      if (ext > 0) q[(int)side[v]].push({gain[v], {-v, v}});
    }
    std::vector<int> moved;
    int64_t cur = 0, best = 0;
    int best_len = 0;
    int64_t best_dev = std::llabs(wa - target_a);
    const int limit = std::max(64, std::min(n / 50 + 1, 4000));
    int since = 0;
    while (since < limit) {
      // candidate from each side: valid top entries
      int cand[2] = {-1, -1};
      for (int s = 0; s < 2; ++s) {
        while (!q[s].empty()) {
          const Ent e = q[s].top();
          const int v = e.second.second;
          if (locked[v] || side[v] != s || gain[v] != e.first) { q[s].pop(); continue; }
          cand[s] = v;
          break;
        }
      }
      // a move from side s changes wa by +vw (s = 1 -> A) or -vw (s = 0 -> B); it must stay within tol or improve balance
      int pick = -1;
      for (int s = 0; s < 2; ++s) {
        const int v = cand[s];
        if (v < 0) continue;
        const int64_t nwa = wa + (s ? g.vw[v] : -g.vw[v]);
        const int64_t ndev = std::llabs(nwa - target_a), odev = std::llabs(wa - target_a);
        if (ndev > tol && ndev >= odev) continue;
        if (pick < 0 || gain[v] > gain[pick] || (gain[v] == gain[pick] && ndev < std::llabs(wa + (side[pick] ? g.vw[pick] : -g.vw[pick]) - target_a)))
          pick = v;
      }
      if (pick < 0) break;
      const int v = pick, s = side[v];
      q[s].pop();
      locked[v] = 1;
      cur += gain[v];
      wa += s ? g.vw[v] : -g.vw[v];
      side[v] = (char)(1 - s);
      moved.push_back(v);
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int u = g.adj[k];
        if (locked[u]) continue;
        gain[u] += (side[u] == s ? 2 : -2) * g.ew[k];      // v left side s: edges to s-vertices became external
        q[(int)side[u]].push({gain[u], {-u, u}});
      }
      const int64_t dev = std::llabs(wa - target_a);
      if (cur > best || (cur == best && dev < best_dev)) {
        best = cur;
        best_len = (int)moved.size();
        best_dev = dev;
        since = 0;
      } else {
        ++since;
      }
    }
    for (int i = (int)moved.size() - 1; i >= best_len; --i) {   // roll back behind the best prefix
      const int v = moved[i];
      side[v] = (char)(1 - side[v]);
      wa += side[v] ? -g.vw[v] : g.vw[v];
    }
    if (best_len == 0) break;
  }
}

// unit-ish weights: move the cheapest boundary vertices until A holds exactly target_a (weight); gains stay exact
void balance_exact(const Graph& g, std::vector<char>& side, int64_t target_a) {
  const int n = g.n;
  int64_t wa = 0;
  for (int v = 0; v < n; ++v)
    if (!side[v]) wa += g.vw[v];
  if (wa == target_a) return;
  const int from = wa > target_a ? 0 : 1;            // the side that gives vertices away
  std::vector<int> gain(n);
  typedef std::pair<int, int> Ent;
  std::priority_queue<Ent> q;
  auto compute = [&](int v) {
    int ext = 0, in = 0;
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) (side[g.adj[k]] != side[v] ? ext : in) += g.ew[k];
    gain[v] = ext - in;
    return ext;
  };
  bool any = false;
  for (int v = 0; v < n; ++v)
    if (side[v] == from && compute(v) > 0) { q.push({gain[v], -v}); any = true; }
  if (!any)
    for (int v = 0; v < n; ++v)
      if (side[v] == from) { compute(v); q.push({gain[v], -v}); }
  while (wa != target_a && !q.empty()) {
    const Ent e = q.top();
    q.pop();
    const int v = -e.second;
    if (side[v] != from || gain[v] != e.first) continue;
    const int64_t nwa = wa + (from ? g.vw[v] : -g.vw[v]);
    if (std::llabs(nwa - target_a) > std::llabs(wa - target_a)) continue;      // would overshoot (heavy vertex)
    side[v] = (char)(1 - from);
    wa = nwa;
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
      const int u = g.adj[k];
      if (side[u] != from) continue;
      compute(u);
      q.push({gain[u], -u});
    }
  }
}

// ------------------------------------------------------------------------------ initial bisection (coarsest graph)
std::vector<std::vector<double>> coarse_eigvecs(const Graph& g, int want) {
  const int n = g.n;
  std::vector<double> lap((size_t)n * n, 0.0), w, v;
  std::vector<double> sw(n);
  for (int i = 0; i < n; ++i) sw[i] = 1.0 / std::sqrt((double)std::max(1, g.vw[i]));
  for (int i = 0; i < n; ++i) {
    double deg = 0.0;
    for (int k = g.xadj[i]; k < g.xadj[i + 1]; ++k) {
      lap[(size_t)i * n + g.adj[k]] -= g.ew[k] * sw[i] * sw[g.adj[k]];
      deg += g.ew[k];
    }
    lap[(size_t)i * n + i] += deg * sw[i] * sw[i];
  }
  dense::sym_eig(lap, n, w, v);
  std::vector<std::vector<double>> out;
  for (int k = 1; k < n && (int)out.size() < want; ++k) {
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) x[i] = v[(size_t)i * n + k] * sw[i];
    out.push_back(std::move(x));
  }
  return out;
}

std::vector<char> split_at_quantile(const Graph& g, const std::vector<double>& f, int64_t target_a) {
  const int n = g.n;
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return f[a] < f[b]; });
  std::vector<char> side(n, 1);
  int64_t wa = 0;
  for (int i = 0; i < n; ++i) {
    if (wa >= target_a && i > 0) break;
    side[order[i]] = 0;
    wa += g.vw[order[i]];
  }
  return side;
}

std::vector<char> initial_bisection(const Graph& g, int64_t target_a) {
  const int n = g.n;
  const int64_t tot = g.total_vw();
  const int64_t tol = std::max<int64_t>(1, tot * 3 / 100);
  std::vector<char> best;
  int64_t best_cut = -1;
  auto consider = [&](std::vector<char> side) {
    fm_refine(g, side, target_a, tol, 6);
    const int64_t c = cut_of(g, side);
    int64_t wa = 0;
    for (int v = 0; v < n; ++v)
      if (!side[v]) wa += g.vw[v];
    if (std::llabs(wa - target_a) > 2 * tol + *std::max_element(g.vw.begin(), g.vw.end())) return;   // hopelessly unbalanced
    if (best_cut < 0 || c < best_cut) { best_cut = c; best = side; }
  };
  // greedy graph growing from several seeds: the region takes the frontier vertex that adds the least cut
  std::vector<int> seeds;
  for (int s = 0; s < 8; ++s) seeds.push_back((int)((int64_t)n * s / 8));
  {   // + a pseudo-peripheral vertex (two BFS sweeps from vertex 0)
    int root = 0;
    for (int rep = 0; rep < 2; ++rep) {
      std::vector<int> dist(n, -1), qu;
      qu.push_back(root);
      dist[root] = 0;
      for (size_t h = 0; h < qu.size(); ++h)
        for (int k = g.xadj[qu[h]]; k < g.xadj[qu[h] + 1]; ++k)
          if (dist[g.adj[k]] < 0) { dist[g.adj[k]] = dist[qu[h]] + 1; qu.push_back(g.adj[k]); }
      root = qu.back();
    }
    seeds.push_back(root);
  }
  for (int seed : seeds) {
    std::vector<char> in_a(n, 0);
    std::vector<int64_t> conn(n, 0), deg(n, 0);
    for (int v = 0; v < n; ++v)
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) deg[v] += g.ew[k];
    int64_t wa = 0;
    int v = seed;
    while (true) {
      in_a[v] = 1;
      wa += g.vw[v];
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) conn[g.adj[k]] += g.ew[k];
      if (wa >= target_a) break;
      int pick = -1;
      int64_t pg = 0;
      bool front = false;
      for (int u = 0; u < n; ++u) {
        if (in_a[u]) continue;
        const bool f = conn[u] > 0;
        const int64_t gn = 2 * conn[u] - deg[u];
        if (pick < 0 || (f && !front) || (f == front && gn > pg)) { pick = u; pg = gn; front = f; }
      }
      if (pick < 0) break;
      v = pick;
    }
    std::vector<char> side(n);
    for (int u = 0; u < n; ++u) side[u] = in_a[u] ? 0 : 1;
    consider(side);
  }
  if (n >= 8 && n <= 600) {   // spectral candidates
    std::vector<std::vector<double>> vs = coarse_eigvecs(g, 3);
    std::vector<std::vector<double>> cands = vs;
    for (size_t i = 0; i < vs.size(); ++i)
      for (size_t j = i + 1; j < vs.size(); ++j)
        for (double sgn : {1.0, -1.0}) {
          std::vector<double> f(n);
          for (int t = 0; t < n; ++t) f[t] = vs[i][t] + sgn * vs[j][t];
          cands.push_back(std::move(f));
        }
    for (auto& f : cands) consider(split_at_quantile(g, f, target_a));
  }
  if (best.empty()) {          // degenerate graphs: first vertices in order
    best.assign(n, 1);
    int64_t wa = 0;
    for (int v = 0; v < n && wa < target_a; ++v) { best[v] = 0; wa += g.vw[v]; }
  }
  return best;
}

// ------------------------------------------------------------------------------ one bisection
int64_t cut_parallel(const Graph& g, const std::vector<char>& side) {
  std::vector<int64_t> part(host_threads() + 1, 0);
  int slot = 0;
  std::vector<std::thread> th;
  const int nt = g.n < 200000 ? 1 : host_threads();
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((int64_t)g.n * t / nt), r1 = (int)((int64_t)g.n * (t + 1) / nt);
    const int my = slot++;
    auto f = [&, r0, r1, my]() {
      int64_t c = 0;
      for (int v = r0; v < r1; ++v)
        for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
          if (side[v] != side[g.adj[k]]) c += g.ew[k];
      part[my] = c;
    };
    if (nt == 1) f();
    else th.emplace_back(f);
  }
  for (auto& x : th) x.join();
  return std::accumulate(part.begin(), part.end(), (int64_t)0) / 2;
}

// side[] with exactly target_a of the weight on side 0 (to one vertex), smallest cut of the two multilevel candidates
std::vector<char> bisect(const Graph& g0, int64_t target_a) {
  std::vector<Graph> levels;             // the coarse graphs; gl[l] points at level l (gl[0] = the caller's graph)
  levels.reserve(64);
  std::vector<const Graph*> gl{&g0};
  std::vector<std::vector<int>> cmaps;
  while (gl.back()->n > 160) {
    std::vector<int> cmap;
    Tick tk(g_t.coarsen);
    Graph c = coarsen(*gl.back(), cmap);
    if (c.n > 0.92 * gl.back()->n) break;     // matching stalled (stars, cliques)
    levels.push_back(std::move(c));
    cmaps.push_back(std::move(cmap));
    gl.clear();
    gl.push_back(&g0);
    for (auto& l : levels) gl.push_back(&l);
  }
  const int L = (int)gl.size();
  const int64_t tot = g0.total_vw();
  auto tol_at = [&](int l) { return std::max<int64_t>(1, tot * (l == 0 ? 2 : 20) / 1000); };
  // ---- candidate 1: multilevel FM
  std::vector<char> side;
  {
    Tick tk(g_t.init);
    side = initial_bisection(*gl[L - 1], target_a);
  }
  Tick* tfm = new Tick(g_t.fm);
  // (the finest level is refined once, below, for the better of the two candidates only)
  for (int l = L - 2; l >= 0; --l) {
    std::vector<char> fine(gl[l]->n);
    const std::vector<int>& cm = cmaps[l];
    for (int v = 0; v < gl[l]->n; ++v) fine[v] = side[cm[v]];
    side.swap(fine);
    if (l > 0 || L == 1) fm_refine(*gl[l], side, target_a, tol_at(l), 6);
  }
  delete tfm;
  int64_t cut1 = cut_parallel(g0, side);
  // ---- candidate 2: multilevel spectral
  const Graph& gc = *gl[L - 1];
  if (gc.n >= 8 && gc.n <= 2500 && L >= 2) {
    std::vector<std::vector<double>> x = coarse_eigvecs(gc, 3);
    const int nv0 = (int)x.size();
    auto smooth = [&](const Graph& g, std::vector<std::vector<double>>& xs) {
      const int n = g.n;
      std::vector<double> deg(n, 0.0);
      for (int v = 0; v < n; ++v)
        for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) deg[v] += g.ew[k];
      double wsum = 0.0;
      for (int v = 0; v < n; ++v) wsum += g.vw[v];
      for (auto& xv : xs) {
        std::vector<double> y(n);
        for (int sweep = 0; sweep < 10; ++sweep) {
          parallel_rows(n, 100000, [&](int r0, int r1) {
            for (int v = r0; v < r1; ++v) {
              double s = 0.0;
              for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) s += g.ew[k] * xv[g.adj[k]];
              y[v] = deg[v] > 0 ? xv[v] - 0.7 * (xv[v] - s / deg[v]) : xv[v];
            }
          });
          double mean = 0.0;
          for (int v = 0; v < n; ++v) mean += g.vw[v] * y[v];
          mean /= wsum;
          for (int v = 0; v < n; ++v) xv[v] = y[v] - mean;
        }
      }
    };
    bool chosen = (nv0 <= 1);
    for (int l = L - 2; l >= 0; --l) {
      const Graph& g = *gl[l];
      const std::vector<int>& cm = cmaps[l];
      for (auto& xv : x) {
        std::vector<double> f(g.n);
        for (int v = 0; v < g.n; ++v) f[v] = xv[cm[v]];
        xv.swap(f);
      }
      {
        Tick tk(g_t.spectral_smooth);
        smooth(g, x);
      }
      Tick tkd(g_t.spectral_dir);
      // direction search on the last level that is still cheap to evaluate (or the finest)
      if (!chosen && (l == 0 || gl[l - 1]->n > 60000)) {
        const int n = g.n, k = (int)x.size();
        // Gram-Schmidt in the vertex-weight inner product
        for (int a = 0; a < k; ++a) {
          for (int b = 0; b < a; ++b) {
            double d = 0.0;
            for (int v = 0; v < n; ++v) d += g.vw[v] * x[a][v] * x[b][v];
            for (int v = 0; v < n; ++v) x[a][v] -= d * x[b][v];
          }
          double nr = 0.0;
          for (int v = 0; v < n; ++v) nr += g.vw[v] * x[a][v] * x[a][v];
          nr = std::sqrt(std::max(nr, 1e-300));
          for (int v = 0; v < n; ++v) x[a][v] /= nr;
        }
        auto field = [&](const std::vector<double>& d) {
          std::vector<double> f(n, 0.0);
          for (int a = 0; a < k; ++a)
            for (int v = 0; v < n; ++v) f[v] += d[a] * x[a][v];
          return f;
        };
        auto cut_dir = [&](const std::vector<double>& d) {
          const std::vector<double> f = field(d);
          // threshold at the quantile by COUNT (nth_element, O(n)): the coarse vertex weights are within 3x of each
          // other, good enough to rank directions; the chosen field is split at the exact weighted quantile later
          std::vector<double> tmp = f;
          const int na = (int)std::min<int64_t>(n - 1, std::max<int64_t>(1, (int64_t)n * target_a / std::max<int64_t>(1, tot)));
          std::nth_element(tmp.begin(), tmp.begin() + (na - 1), tmp.end());
          const double thr = tmp[na - 1];
          std::vector<char> sd(n);
          for (int v = 0; v < n; ++v) sd[v] = f[v] > thr;
          return cut_parallel(g, sd);
        };
        std::vector<std::vector<double>> dirs;
        if (k == 2) {
          for (int i = 0; i < 32; ++i) dirs.push_back({std::cos(M_PI * i / 32), std::sin(M_PI * i / 32)});
        } else {
          for (int i = 0; i < 64; ++i) {     // Fibonacci half sphere
            const double z = (i + 0.5) / 64, r = std::sqrt(std::max(0.0, 1.0 - z * z)), phi = i * M_PI * (3.0 - std::sqrt(5.0));
            dirs.push_back({r * std::cos(phi), r * std::sin(phi), z});
          }
        }
        std::vector<double> bd = dirs[0];
        int64_t bc = cut_dir(bd);
        for (size_t i = 1; i < dirs.size(); ++i) {
          const int64_t c = cut_dir(dirs[i]);
          if (c < bc) { bc = c; bd = dirs[i]; }
        }
        double step = 0.2;
        for (int it = 0; it < 5; ++it) {
          bool improved = false;
          for (int axis = 0; axis < k; ++axis)
            for (double sgn : {1.0, -1.0}) {
              std::vector<double> d = bd;
              d[axis] += sgn * step;
              double nr = 0.0;
              for (double t : d) nr += t * t;
              nr = std::sqrt(nr);
              for (double& t : d) t /= nr;
              const int64_t c = cut_dir(d);
              if (c < bc) { bc = c; bd = d; improved = true; }
            }
          if (!improved) step *= 0.5;
        }
        std::vector<double> f = field(bd);
        x.clear();
        x.push_back(std::move(f));          // from here on ONE field travels down
        chosen = true;
      }
    }
    if (chosen && x.size() == 1 && (int)x[0].size() == g0.n) {
      std::vector<char> s2 = split_at_quantile(g0, x[0], target_a);
      const int64_t cut2 = cut_parallel(g0, s2);
      if (cut2 < cut1) { side.swap(s2); cut1 = cut2; }
    }
  }
  {
    Tick tk(g_t.fm);
    fm_refine(g0, side, target_a, tol_at(0), 4);
    balance_exact(g0, side, target_a);
    fm_refine(g0, side, target_a, 0, 2);
  }
  return side;
}

// induced sub-graph of the vertices with mask == 1; ids[new] = old
Graph subgraph(const Graph& g, const std::vector<int>& verts, std::vector<int>& local /*scratch, size g.n, -1*/) {
  Graph s;
  s.n = (int)verts.size();
  for (int i = 0; i < s.n; ++i) local[verts[i]] = i;
  s.xadj.assign(s.n + 1, 0);
  s.vw.resize(s.n);
  for (int i = 0; i < s.n; ++i) {
    const int v = verts[i];
    s.vw[i] = g.vw[v];
    int c = 0;
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) c += local[g.adj[k]] >= 0;
    s.xadj[i + 1] = s.xadj[i] + c;
  }
  s.adj.resize(s.xadj[s.n]);
  s.ew.resize(s.xadj[s.n]);
  for (int i = 0; i < s.n; ++i) {
    const int v = verts[i];
    int p = s.xadj[i];
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
      const int u = local[g.adj[k]];
      if (u >= 0) { s.adj[p] = u; s.ew[p] = g.ew[k]; ++p; }
    }
  }
  for (int v : verts) local[v] = -1;
  return s;
}

void kway(const Graph& g, int nparts, int* part) {
  std::vector<int> local(g.n, -1);
  struct Job { std::vector<int> verts; int first, k; };
  std::vector<Job> stack;
  {
    Job j;
    j.verts.resize(g.n);
    std::iota(j.verts.begin(), j.verts.end(), 0);
    j.first = 0;
    j.k = nparts;
    stack.push_back(std::move(j));
  }
  while (!stack.empty()) {
    Job j = std::move(stack.back());
    stack.pop_back();
    if (j.k == 1) {
      for (int v : j.verts) part[v] = j.first;
      continue;
    }
    const int k1 = j.k / 2;
    const int nv = (int)j.verts.size();
    double t_sub = now_s();
    Graph s = (nv == g.n) ? g : subgraph(g, j.verts, local);
    g_t.sub += now_s() - t_sub;
    const int64_t tot = s.total_vw();
    int64_t ta = (tot * k1 + j.k / 2) / j.k;
    ta = std::min<int64_t>(std::max<int64_t>(ta, k1), tot - (j.k - k1));
    std::vector<char> side = bisect(s, ta);
    Job a, b;
    for (int i = 0; i < nv; ++i) (side[i] ? b.verts : a.verts).push_back(j.verts[i]);
    if (a.verts.empty() || b.verts.empty()) {          // cannot happen for k <= n; keep every part non-empty anyway
      a.verts.assign(j.verts.begin(), j.verts.begin() + std::max(1, nv * k1 / j.k));
      b.verts.assign(j.verts.begin() + a.verts.size(), j.verts.end());
    }
    a.first = j.first; a.k = k1;
    b.first = j.first + k1; b.k = j.k - k1;
    stack.push_back(std::move(b));
    stack.push_back(std::move(a));
  }
}

// stray fragments of a part (cut off from its main body by a later bisection) go to the neighbouring part they touch most
void absorb_fragments(const Graph& g, int nparts, int* part) {
  const int n = g.n;
  std::vector<int> comp(n, -1), csize, cpart;
  std::vector<int> qu;
  for (int v = 0; v < n; ++v) {
    if (comp[v] >= 0) continue;
    const int c = (int)csize.size();
    qu.clear();
    qu.push_back(v);
    comp[v] = c;
    for (size_t h = 0; h < qu.size(); ++h)
      for (int k = g.xadj[qu[h]]; k < g.xadj[qu[h] + 1]; ++k) {
        const int u = g.adj[k];
        if (comp[u] < 0 && part[u] == part[v]) { comp[u] = c; qu.push_back(u); }
      }
    csize.push_back((int)qu.size());
    cpart.push_back(part[v]);
  }
  std::vector<int> main_of(nparts, -1);
  for (int c = 0; c < (int)csize.size(); ++c)
    if (main_of[cpart[c]] < 0 || csize[c] > csize[main_of[cpart[c]]]) main_of[cpart[c]] = c;
  std::vector<int64_t> psize(nparts, 0);
  for (int v = 0; v < n; ++v) psize[part[v]]++;
  std::vector<int> touch(nparts);
  std::vector<std::vector<int>> members(csize.size());
  for (int v = 0; v < n; ++v)
    if (comp[v] != main_of[part[v]]) members[comp[v]].push_back(v);
  for (int c = 0; c < (int)csize.size(); ++c) {
    if (members[c].empty()) continue;
    if ((int64_t)csize[c] * 50 > psize[cpart[c]]) continue;      // a large second body is not a stray fragment
    std::fill(touch.begin(), touch.end(), 0);
    for (int v : members[c])
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
        if (part[g.adj[k]] != cpart[c]) touch[part[g.adj[k]]] += g.ew[k];
    const int dest = (int)(std::max_element(touch.begin(), touch.end()) - touch.begin());
    if (touch[dest] == 0) continue;                               // a genuinely separate component of the graph
    for (int v : members[c]) part[v] = dest;
    psize[dest] += csize[c];
    psize[cpart[c]] -= csize[c];
  }
}

Graph graph_from_csr(int n, const int* xadj, const int* adjncy) {
  Graph g;
  g.n = n;
  g.xadj.assign(xadj, xadj + n + 1);
  g.adj.assign(adjncy, adjncy + xadj[n]);
  g.ew.assign((size_t)xadj[n], 1);
  g.vw.assign(n, 1);
  return g;
}

// dual: vertices = elements, adjacent when they share a node; nodal: vertices = nodes, adjacent when they share an element
Graph graph_from_mesh(int ne, int nn, const int* eptr, const int* eind, bool dual) {
  // node -> elements incidence
  std::vector<int> nptr(nn + 1, 0);
  for (int k = 0; k < eptr[ne]; ++k) nptr[eind[k] + 1]++;
  for (int i = 0; i < nn; ++i) nptr[i + 1] += nptr[i];
  std::vector<int> nind(eptr[ne]);
  {
    std::vector<int> fill(nptr.begin(), nptr.end() - 1);
    for (int e = 0; e < ne; ++e)
      for (int k = eptr[e]; k < eptr[e + 1]; ++k) nind[fill[eind[k]]++] = e;
  }
  const int n = dual ? ne : nn;
  const int* aptr = dual ? eptr : nptr.data();       // vertex -> hubs
  const int* aind = dual ? eind : nind.data();
  const int* bptr = dual ? nptr.data() : eptr;       // hub -> vertices
  const int* bind = dual ? nind.data() : eind;
  Graph g;
  g.n = n;
  g.vw.assign(n, 1);
  g.xadj.assign(n + 1, 0);
  const int nt = n < 200000 ? 1 : host_threads();
  std::vector<std::vector<int>> chunks(nt);
  std::vector<std::vector<int>> counts(nt);
  auto work = [&](int t) {
    const int r0 = (int)((int64_t)n * t / nt), r1 = (int)((int64_t)n * (t + 1) / nt);
    std::vector<int> mark(n, -1);
    std::vector<int>& out = chunks[t];
    std::vector<int>& cnt = counts[t];
    cnt.assign(r1 - r0, 0);
    for (int v = r0; v < r1; ++v) {
      const size_t start = out.size();
      for (int k = aptr[v]; k < aptr[v + 1]; ++k) {
        const int h = aind[k];
        for (int l = bptr[h]; l < bptr[h + 1]; ++l) {
          const int u = bind[l];
          if (u != v && mark[u] != v) { mark[u] = v; out.push_back(u); }
        }
      }
      std::sort(out.begin() + start, out.end());
      cnt[v - r0] = (int)(out.size() - start);
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((int64_t)n * t / nt);
    for (size_t i = 0; i < counts[t].size(); ++i) g.xadj[r0 + (int)i + 1] = counts[t][i];
  }
  for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
  g.adj.resize(g.xadj[n]);
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((int64_t)n * t / nt);
    std::copy(chunks[t].begin(), chunks[t].end(), g.adj.begin() + g.xadj[r0]);
  }
  g.ew.assign(g.adj.size(), 1);
  return g;
}

int run_kway(const Graph& g, int nparts, int* objval, int* part) {
  if (nparts < 1 || nparts > std::max(1, g.n)) return 1;
  if (nparts == 1) {
    std::fill(part, part + g.n, 0);
    if (objval) *objval = 0;
    return 0;
  }
  g_t = Timers();
  const double t0 = now_s();
  kway(g, nparts, part);
  absorb_fragments(g, nparts, part);
  if (getenv("GENEO_DEBUG"))
    fprintf(stderr, "[partition] %d vertices -> %d parts in %.2f s: coarsening %.2f, initial cuts %.2f, FM refinement %.2f, spectral "
            "smoothing %.2f + direction search %.2f, sub-graphs %.2f s\n", g.n, nparts, now_s() - t0, g_t.coarsen, g_t.init, g_t.fm,
            g_t.spectral_smooth, g_t.spectral_dir, g_t.sub);
  if (objval) {
    int64_t c = 0;
    for (int v = 0; v < g.n; ++v)
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) c += part[v] != part[g.adj[k]];
    *objval = (int)(c / 2);
  }
  return 0;
}

}  // namespace

extern "C" {

// METIS_PartGraphKway counterpart on a CSR graph (symmetric, no self loops, unit weights): part[v] in [0, nparts)
int GeneoPartGraphKway(int n, const int* xadj, const int* adjncy, int nparts, int* objval, int* part) {
  if (n < 0 || !xadj || (!adjncy && xadj[n] > 0) || !part) return 1;
  try {
    return run_kway(graph_from_csr(n, xadj, adjncy), nparts, objval, part);
  } catch (...) {
    return 1;
  }
}

// METIS_PartMeshDual (driver:386-413; ncommon = 1) / METIS_PartMeshNodal: elements given as node lists (eptr / eind).
// The partitioned objects get the k-way partition; the other kind follows it (a node takes the part of its first
// element, an element the part of its first node), as Metis derives its second output.
static int part_mesh(int ne, int nn, const int* eptr, const int* eind, int nparts, bool dual, int* objval, int* epart, int* npart) {
  if (ne < 0 || nn < 0 || !eptr || !eind || !epart || !npart) return 1;
  try {
    Graph g = graph_from_mesh(ne, nn, eptr, eind, dual);
    std::vector<int> p(g.n);
    if (int rc = run_kway(g, nparts, objval, p.data())) return rc;
    if (dual) {
      std::copy(p.begin(), p.end(), epart);
      std::fill(npart, npart + nn, 0);
      std::vector<char> seen(nn, 0);
      for (int e = 0; e < ne; ++e)
        for (int k = eptr[e]; k < eptr[e + 1]; ++k)
          if (!seen[eind[k]]) { seen[eind[k]] = 1; npart[eind[k]] = p[e]; }
    } else {
      std::copy(p.begin(), p.end(), npart);
      for (int e = 0; e < ne; ++e) epart[e] = eptr[e + 1] > eptr[e] ? p[eind[eptr[e]]] : 0;
    }
    return 0;
  } catch (...) {
    return 1;
  }
}
int GeneoPartMeshDual(int ne, int nn, const int* eptr, const int* eind, int nparts, int* objval, int* epart, int* npart) {
  return part_mesh(ne, nn, eptr, eind, nparts, true, objval, epart, npart);
}
int GeneoPartMeshNodal(int ne, int nn, const int* eptr, const int* eind, int nparts, int* objval, int* epart, int* npart) {
  return part_mesh(ne, nn, eptr, eind, nparts, false, objval, epart, npart);
}

}  // extern "C"
