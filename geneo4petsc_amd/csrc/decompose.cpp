// Host-side domain decomposition of libgeneopc: the counterpart of the reference DRIVER's decompose + addOverlapLayers +
// buildDomain + fillALoc (src/geneo4PETSc.cpp:196-379, :447-494, :643-715) and of its structured input generators
// (tst/laplacian/laplacian.cpp:57-188, tst/heat/heat.cpp:64-261), in C++ behind the C ABI -- the numpy prototypes in
// geneo4petsc_amd/decomp.py (which the parity tests keep using as the readable restatement) took 4-14 s of host time per
// bench run.  Pure host code, no device.
//
//   per part p (driver:312-345): start from the elements of p (dual) or the elements with a node in p (nodal,
//   driver:196-215); every overlap layer adds the elements sharing a node with the current set (driver:244-269); the
//   domain's nodes are the nodes of its elements; node / element multiplicity = number of domains holding it.
//   A_Neu,p = sum of its elements' matrices weighted 1 / elemMult (driver:473-475, :683-715) in ascending-global local
//   numbering; A_Dir,p = R_p A R_p^T (every element touching the node set, restricted to it).
//
// All parts are grown together: one bit per part in a word array per node / per element (OR-propagation), so a
// decomposition costs (overlap + 1) passes over the element lists whatever the number of parts.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/geneo_c.h"

namespace {

struct Decomp {
  int nn = 0, ne = 0, w = 0, nparts = 0, words = 1;
  const int* nodes = nullptr;        // ne x w, -1 padded (borrowed: the caller keeps the mesh alive)
  const double* mats = nullptr;      // ne x w*w
  std::vector<uint64_t> nbits, ebits;
  std::vector<int> node_mult, elem_mult;
  bool has(const std::vector<uint64_t>& b, int64_t i, int p) const { return (b[(size_t)i * words + (p >> 6)] >> (p & 63)) & 1ull; }
};

int popcount_words(const uint64_t* p, int words) {
  int c = 0;
  for (int k = 0; k < words; ++k) c += __builtin_popcountll(p[k]);
  return c;
}

template <class T>
T* dup(const std::vector<T>& v) {
  T* p = (T*)malloc(sizeof(T) * std::max<size_t>(1, v.size()));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

// CSR of sum_e w_e * mat_e over the selected elements, rows / columns in local numbering (g2l >= 0), columns sorted and
// duplicates summed in element order
void assemble(const Decomp& d, const std::vector<int>& elems, const std::vector<double>* weight, const std::vector<int>& g2l,
              int nloc, std::vector<int>& rowptr, std::vector<int>& col, std::vector<double>& val) {
  const int w = d.w;
  std::vector<int64_t> cnt((size_t)nloc + 1, 0);
  for (int e : elems)
    for (int a = 0; a < w; ++a) {
      const int ga = d.nodes[(size_t)e * w + a];
      if (ga < 0 || g2l[ga] < 0) continue;
      for (int b = 0; b < w; ++b) {
        const int gb = d.nodes[(size_t)e * w + b];
        if (gb >= 0 && g2l[gb] >= 0) cnt[g2l[ga] + 1]++;
      }
    }
  for (int i = 0; i < nloc; ++i) cnt[i + 1] += cnt[i];
  std::vector<int> rc((size_t)cnt[nloc]);
  std::vector<double> rv((size_t)cnt[nloc]);
  {
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (size_t k = 0; k < elems.size(); ++k) {
      const int e = elems[k];
      const double we = weight ? (*weight)[k] : 1.0;
      for (int a = 0; a < w; ++a) {
        const int ga = d.nodes[(size_t)e * w + a];
        if (ga < 0 || g2l[ga] < 0) continue;
        for (int b = 0; b < w; ++b) {
          const int gb = d.nodes[(size_t)e * w + b];
          if (gb < 0 || g2l[gb] < 0) continue;
          const int64_t p = fill[g2l[ga]]++;
          rc[p] = g2l[gb];
          rv[p] = weight ? d.mats[(size_t)e * w * w + a * w + b] * we : d.mats[(size_t)e * w * w + a * w + b];
        }
      }
    }
  }
  // per row: stable sort by column, merge duplicates (row ranges on host threads)
  rowptr.assign((size_t)nloc + 1, 0);
  std::vector<int> uniq(nloc, 0);
  const int nt = nloc < 200000 ? 1 : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto sort_rows = [&](int r0, int r1) {
    std::vector<std::pair<int, double>> tmp;
    for (int i = r0; i < r1; ++i) {
      const int64_t a = cnt[i], b = cnt[i + 1];
      tmp.resize((size_t)(b - a));
      for (int64_t k = a; k < b; ++k) tmp[(size_t)(k - a)] = {rc[k], rv[k]};
      std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
      int64_t o = a;
      for (size_t k = 0; k < tmp.size(); ++k) {
        if (o > a && rc[o - 1] == tmp[k].first) rv[o - 1] += tmp[k].second;
        else { rc[o] = tmp[k].first; rv[o] = tmp[k].second; ++o; }
      }
      uniq[i] = (int)(o - a);
    }
  };
  if (nt == 1) sort_rows(0, nloc);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(sort_rows, (int)((int64_t)nloc * t / nt), (int)((int64_t)nloc * (t + 1) / nt));
    for (auto& x : th) x.join();
  }
  for (int i = 0; i < nloc; ++i) rowptr[i + 1] = rowptr[i] + uniq[i];
  col.resize((size_t)rowptr[nloc]);
  val.resize((size_t)rowptr[nloc]);
  for (int i = 0; i < nloc; ++i) {
    std::copy(rc.begin() + cnt[i], rc.begin() + cnt[i] + uniq[i], col.begin() + rowptr[i]);
    std::copy(rv.begin() + cnt[i], rv.begin() + cnt[i] + uniq[i], val.begin() + rowptr[i]);
  }
}

double kappa_of(int interp, double alpha, double beta, double x) {
  if (interp == 1) return alpha * x * x + beta;                  // quad
  if (interp == 2) return alpha * x + beta;                      // lin
  if (interp == 3) return x >= 2.0 * beta ? 1.0 : (x >= beta ? alpha : 1.0);   // minmax (laplacianServices.cpp:20-36)
  return 1.0;
}

}  // namespace

extern "C" {

struct _p_GeneoDecomp : Decomp {};

// nodes: nbElem x W node ids (-1 = unused slot), mats: nbElem x W*W element matrices; both must outlive the handle.
// elemPart (dual) or nodePart (nodal) is the k-way partition (driver:381-445).
PetscErrorCode GeneoDecompCreate(int nbNode, int nbElem, int W, const int* nodes, const double* mats, int nbPart,
                                 const int* elemPart, const int* nodePart, int dual, int addOverlap, GeneoDecomp* out) {
  if (!out || nbNode < 0 || nbElem < 0 || W < 1 || !nodes || !mats || nbPart < 1 || (dual ? !elemPart : !nodePart)) return 1;
  try {
    _p_GeneoDecomp* d = new _p_GeneoDecomp();
    d->nn = nbNode; d->ne = nbElem; d->w = W; d->nparts = nbPart; d->words = (nbPart + 63) / 64;
    d->nodes = nodes; d->mats = mats;
    const int words = d->words;
    d->nbits.assign((size_t)nbNode * words, 0);
    d->ebits.assign((size_t)nbElem * words, 0);
    auto bit = [&](uint64_t* p, int part) { p[part >> 6] |= 1ull << (part & 63); };
    // every node slot is either -1 (unused) or a node of the mesh, in both modes (the driver's "bad element" errors):
    // the loops below index per-node arrays with them
    for (size_t k = 0; k < (size_t)nbElem * W; ++k)
      if (nodes[k] < -1 || nodes[k] >= nbNode) { delete d; return 1; }
    for (int e = 0; e < nbElem; ++e) {
      uint64_t* eb = &d->ebits[(size_t)e * words];
      if (dual) {
        if (elemPart[e] < 0 || elemPart[e] >= nbPart) { delete d; return 1; }
        bit(eb, elemPart[e]);
      } else {
        for (int a = 0; a < W; ++a) {
          const int g = nodes[(size_t)e * W + a];
          if (g < 0) continue;
          if (g >= nbNode || nodePart[g] < 0 || nodePart[g] >= nbPart) { delete d; return 1; }
          bit(eb, nodePart[g]);
        }
      }
    }
    auto nodes_from_elems = [&]() {
      std::fill(d->nbits.begin(), d->nbits.end(), 0);
      for (int e = 0; e < nbElem; ++e)
        for (int a = 0; a < W; ++a) {
          const int g = nodes[(size_t)e * W + a];
          if (g < 0) continue;
          for (int k = 0; k < words; ++k) d->nbits[(size_t)g * words + k] |= d->ebits[(size_t)e * words + k];
        }
    };
    for (int layer = 0; layer < addOverlap; ++layer) {
      nodes_from_elems();
      for (int e = 0; e < nbElem; ++e)
        for (int a = 0; a < W; ++a) {
          const int g = nodes[(size_t)e * W + a];
          if (g < 0) continue;
          for (int k = 0; k < words; ++k) d->ebits[(size_t)e * words + k] |= d->nbits[(size_t)g * words + k];
        }
    }
    nodes_from_elems();
    d->node_mult.resize(nbNode);
    d->elem_mult.resize(nbElem);
    for (int i = 0; i < nbNode; ++i) d->node_mult[i] = popcount_words(&d->nbits[(size_t)i * words], words);
    for (int e = 0; e < nbElem; ++e) d->elem_mult[e] = popcount_words(&d->ebits[(size_t)e * words], words);
    *out = d;
    return 0;
  } catch (...) {
    return 1;
  }
}

PetscErrorCode GeneoDecompDomain(GeneoDecomp d, int p, int withDirichlet, GeneoDomain* out) {
  if (!d || !out || p < 0 || p >= d->nparts) return 1;
  try {
    memset(out, 0, sizeof(*out));
    const int nn = d->nn, ne = d->ne, w = d->w;
    std::vector<int> l2g, g2l((size_t)nn, -1);
    for (int i = 0; i < nn; ++i)
      if (d->has(d->nbits, i, p)) { g2l[i] = (int)l2g.size(); l2g.push_back(i); }
    const int nloc = (int)l2g.size();
    std::vector<int> mult(nloc);
    for (int i = 0; i < nloc; ++i) mult[i] = d->node_mult[l2g[i]];
    std::vector<int> own, touch;
    std::vector<double> wgt;
    for (int e = 0; e < ne; ++e) {
      if (d->has(d->ebits, e, p)) { own.push_back(e); wgt.push_back(1.0 / (double)d->elem_mult[e]); }
      if (withDirichlet) {
        bool t = false;
        for (int a = 0; a < w && !t; ++a) {
          const int g = d->nodes[(size_t)e * w + a];
          t = g >= 0 && g2l[g] >= 0;
        }
        if (t) touch.push_back(e);
      }
    }
    std::vector<int> rp, col;
    std::vector<double> val;
    assemble(*d, own, &wgt, g2l, nloc, rp, col, val);
    out->n = nloc;
    out->l2g = dup(l2g);
    out->mult = dup(mult);
    out->neu_rowptr = dup(rp); out->neu_col = dup(col); out->neu_val = dup(val);
    if (withDirichlet) {
      assemble(*d, touch, nullptr, g2l, nloc, rp, col, val);
      out->dir_rowptr = dup(rp); out->dir_col = dup(col); out->dir_val = dup(val);
    }
    // intersections with every other part: local indices of the shared nodes (hdr/geneo.hpp:34 intersectLoc)
    std::vector<int> iptr((size_t)d->nparts + 1, 0), iidx;
    for (int q = 0; q < d->nparts; ++q) {
      if (q != p)
        for (int i = 0; i < nloc; ++i)
          if (d->has(d->nbits, l2g[i], q)) iidx.push_back(i);
      iptr[q + 1] = (int)iidx.size();
    }
    out->inter_ptr = dup(iptr);
    out->inter_idx = dup(iidx);
    return 0;
  } catch (...) {
    return 1;
  }
}

void GeneoFreeDomain(GeneoDomain* dm) {
  if (!dm) return;
  free(dm->l2g); free(dm->mult); free(dm->neu_rowptr); free(dm->neu_col); free(dm->neu_val);
  free(dm->dir_rowptr); free(dm->dir_col); free(dm->dir_val); free(dm->inter_ptr); free(dm->inter_idx);
  memset(dm, 0, sizeof(*dm));
}

void GeneoDecompDestroy(GeneoDecomp* d) {
  if (!d || !*d) return;
  delete *d;
  *d = nullptr;
}

// tst/laplacian (laplacian.cpp:57-188) and tst/heat (heat.cpp:64-261) generators on an n^dim grid, optionally only the
// elements whose nodes all lie in the index box [wlo, whi) (node ids stay global).  1-D edge elements
// kappa [[1+eps, -1], [-1, 1+eps]] (+ mass / dt for heat) created from the lower endpoint, whose coordinates give
// kappa = kappa(x) kappa(y) kappa(z); one 1-node Dirichlet element kappa (1+eps) per node of the face {last coordinate
// = 0}.  Element order = the reference's: per node in ascending id, x-edge, y-edge, [Dirichlet in 2-D between them]
// ..., i.e. sorted by key = 6 node + (2 axis + 1 | 2 (dim - 1) for the Dirichlet element).
// interp: 0 none, 1 quad, 2 lin, 3 minmax.  Outputs are malloc'ed (free with GeneoFreeMesh): nodes nbElem x 2 (-1 in the
// second slot of a Dirichlet element), mats nbElem x 4.
PetscErrorCode GeneoGridMesh(int n, int dim, double inpEps, double kappaMax, int interp, int heat, double lbd, double dt,
                             const int* wlo, const int* whi, int* nbNode, int* nbElem, int** nodesOut, double** matsOut) {
  if (n < 1 || dim < 1 || dim > 3 || !nbNode || !nbElem || !nodesOut || !matsOut) return 1;
  try {
    const int d[3] = {n, dim >= 2 ? n : 1, dim >= 3 ? n : 1};
    int lo[3] = {0, 0, 0}, hi[3] = {d[0], d[1], d[2]};
    if (wlo && whi)
      for (int a = 0; a < 3; ++a) { lo[a] = wlo[a]; hi[a] = whi[a]; }
    const double xmax = (double)(n - 1);
    double alpha = 0.0, beta = 1.0;
    if (interp == 1) alpha = (kappaMax - beta) / (xmax * xmax);
    else if (interp == 2) alpha = (kappaMax - beta) / xmax;
    else if (interp == 3) { alpha = kappaMax; beta = xmax / 3.0; }
    std::vector<double> kx(hi[0] - lo[0]), ky(hi[1] - lo[1]), kz(hi[2] - lo[2]);
    for (int i = lo[0]; i < hi[0]; ++i) kx[i - lo[0]] = kappa_of(interp, alpha, beta, (double)i);
    for (int j = lo[1]; j < hi[1]; ++j) ky[j - lo[1]] = kappa_of(interp, alpha, beta, (double)j);
    for (int k = lo[2]; k < hi[2]; ++k) kz[k - lo[2]] = kappa_of(interp, alpha, beta, (double)k);
    const int64_t stride[3] = {1, d[0], (int64_t)d[0] * d[1]};
    std::vector<int> nodes;
    std::vector<double> mats;
    const size_t guess = (size_t)(hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) * (dim + 1);
    nodes.reserve(2 * guess);
    mats.reserve(4 * guess);
    auto push = [&](int64_t a, int64_t b, double kk, bool bc) {
      const double lap_d = (1.0 + inpEps) * kk;
      nodes.push_back((int)a);
      nodes.push_back(bc ? -1 : (int)b);
      if (heat) {
        const double dg = lbd * lap_d + (1.0 / 3.0) / dt, od = lbd * (-kk) + (1.0 / 6.0) / dt;
        mats.push_back(dg); mats.push_back(bc ? 0.0 : od); mats.push_back(bc ? 0.0 : od); mats.push_back(bc ? 0.0 : dg);
      } else {
        mats.push_back(lap_d); mats.push_back(bc ? 0.0 : -kk); mats.push_back(bc ? 0.0 : -kk); mats.push_back(bc ? 0.0 : lap_d);
      }
    };
    for (int k = lo[2]; k < hi[2]; ++k)
      for (int j = lo[1]; j < hi[1]; ++j)
        for (int i = lo[0]; i < hi[0]; ++i) {
          const int64_t c = i + stride[1] * j + stride[2] * k;
          const double kap = (kx[i - lo[0]] * ky[j - lo[1]]) * kz[k - lo[2]];
          const int co[3] = {i, j, k};
          for (int ax = 0; ax < 3; ++ax) {
            // key order: edge ax has key 2 ax + 1, the Dirichlet element 2 (dim - 1): it comes before the edge of the
            // last axis
            if (ax == dim - 1 && co[dim - 1] == 0) push(c, -1, kap, true);
            if (d[ax] > 1 && co[ax] < hi[ax] - 1) push(c, c + stride[ax], kap, false);
          }
        }
    *nbNode = d[0] * d[1] * d[2];
    *nbElem = (int)(nodes.size() / 2);
    *nodesOut = dup(nodes);
    *matsOut = dup(mats);
    return (*nodesOut && *matsOut) ? 0 : 1;
  } catch (...) {
    return 1;
  }
}

void GeneoFreeMesh(int* nodes, double* mats) {
  free(nodes);
  free(mats);
}

}  // extern "C"
