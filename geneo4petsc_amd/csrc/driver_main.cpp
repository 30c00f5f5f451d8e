// C++ counterpart of the reference CLI driver (src/geneo4PETSc.cpp main():1569) over the C ABI of include/geneo_c.h:
// read or generate the element list, partition, decompose, build the GenEO preconditioner, solve, print the reference's
// INFO: / TIME: lines (printIterativeGlobalSolveParameters / Results / Timing, driver:898-1231, in the shapes
// tst/plot.py:57-116 parses).  One process, --np N subdomains on its GPU (N stands for `mpirun -n N`).
//
//   geneo_driver --inpLibA <getInput plugin .so | laplacian | heat>#--size#48#--dim#3 --np 8 --metisNodal --addOverlap 2
//                -geneo_lvl SORAS,2 -ksp_type cg --timing
//
// Flags (driver:1396-1495): --inpFileA, --inpLibA, --inpFileB, --inpEps, --metisDual | --metisNodal, --addOverlap,
// --verbose, --timing, --shortRes, --cmdLine; --parts px,py,pz (structured blocks of a generated grid) and --partFile
// (one part id per line) where no Metis partition is wanted; every other option is forwarded to the PC (-geneo_*, -ksp_*,
// -els2_*, -dls1_*).  Exported as GeneoDriverMain (the tests call it in-process); tools/geneo_driver.cpp is the main().
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/geneo_c.h"

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

struct Mesh {                 // element list, padded to W nodes per element (-1: unused slot)
  int nbNode = 0, nbElem = 0, W = 0;
  std::vector<int> nodes;     // nbElem x W
  std::vector<double> mats;   // nbElem x W*W
  int grid_n = 0, grid_dim = 0;   // > 0: a structured grid of the built-in generators (for --parts)
};

struct Cli {
  std::string inpFileA, inpLibA, inpFileB, partFile;
  double inpEps = 1e-4;
  bool metisDual = true, timing = false, shortRes = false, cmdLine = false;
  int addOverlap = 0, verbose = 0, np = 1;
  int parts[3] = {0, 0, 0};
  std::vector<std::string> pc_args;
};

int fail(const std::string& m) {
  fprintf(stderr, "Error: %s\n", m.c_str());
  return 1;
}

void from_lists(Mesh& m, const std::vector<int>& ptr, const std::vector<int>& idx, const std::vector<std::vector<double>>& mats) {
  m.nbElem = (int)ptr.size() - 1;
  m.W = 0;
  m.nbNode = 0;
  for (int e = 0; e < m.nbElem; ++e) m.W = std::max(m.W, ptr[e + 1] - ptr[e]);
  for (int v : idx) m.nbNode = std::max(m.nbNode, v + 1);
  m.nodes.assign((size_t)m.nbElem * m.W, -1);
  m.mats.assign((size_t)m.nbElem * m.W * m.W, 0.0);
  for (int e = 0; e < m.nbElem; ++e) {
    const int k = ptr[e + 1] - ptr[e];
    for (int a = 0; a < k; ++a) {
      m.nodes[(size_t)e * m.W + a] = idx[ptr[e] + a];
      for (int b = 0; b < k; ++b) m.mats[((size_t)e * m.W + a) * m.W + b] = mats[e][(size_t)a * k + b];
    }
  }
}

// --inpFileA: 'dof dof ... [- a11 a12 ...]' per element, '#' / '%' comments (driver:98-194)
int read_input_text(const std::string& path, double inpEps, Mesh& m) {
  std::ifstream in(path);
  if (!in) return fail("can not open " + path);
  std::vector<int> ptr = {0}, idx;
  std::vector<std::vector<double>> mats;
  std::string line;
  while (std::getline(in, line)) {
    size_t p = line.find_first_not_of(" \t");
    if (p == std::string::npos || line[p] == '#' || line[p] == '%') continue;
    std::string head = line, tail;
    const size_t d = line.find(" - ");
    if (d != std::string::npos) { head = line.substr(0, d); tail = line.substr(d + 3); }
    std::istringstream hs(head), ts(tail);
    std::vector<int> dofs;
    std::string tok;
    while (hs >> tok) {
      char* end = nullptr;
      const long v = strtol(tok.c_str(), &end, 10);
      if (end && *end == 0 && v >= 0) dofs.push_back((int)v);
    }
    std::vector<double> vals;
    double v;
    while (ts >> v) vals.push_back(v);
    const int n = (int)dofs.size();
    if (n == 0) continue;
    if (vals.empty())                                   // default element matrix (driver:130-138)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) vals.push_back(i == j ? 1.0 + inpEps : -1.0 / (n - 1));
    if ((int)vals.size() != n * n) return fail("bad element: " + line);
    idx.insert(idx.end(), dofs.begin(), dofs.end());
    ptr.push_back((int)idx.size());
    mats.push_back(vals);
  }
  if (mats.empty()) return fail("no element in " + path);
  from_lists(m, ptr, idx, mats);
  std::vector<char> seen(m.nbNode, 0);
  for (int v : idx) seen[v] = 1;
  for (char c : seen)
    if (!c) return fail("bad node set in " + path);
  return 0;
}

int grid_size(int size, int weak, int dim) {               // laplacianServices.cpp: grid side from --size / --weakScaling
  if (dim == 1) return size * weak;
  if (dim == 2) return (int)std::sqrt((double)size * size * weak);
  const double r = (double)size * size * size * weak;
  int c = (int)std::lround(std::cbrt(r));
  while ((double)c * c * c > r) --c;
  while ((double)(c + 1) * (c + 1) * (c + 1) <= r) ++c;
  return c;
}

// --inpLibA: a getInput plugin built for the reference driver (loaded unchanged), or the bare name of a structured
// generator of the library (laplacian | heat) with the plugin's own arguments
int load_lib_input(const std::string& spec, Mesh& m) {
  const size_t h = spec.find('#');
  const std::string name = spec.substr(0, h), rest = h == std::string::npos ? "" : spec.substr(h + 1);
  if (std::ifstream(name).good()) {
    GeneoInput in;
    if (GeneoGetLibInput(name.c_str(), rest.c_str(), &in)) return fail(std::string("can not load input from ") + name);
    std::vector<int> ptr(in.nbElem + 1), idx(in.nIdx);
    for (unsigned e = 0; e <= in.nbElem; ++e) ptr[e] = (int)in.elemPtr[e];
    for (size_t k = 0; k < in.nIdx; ++k) idx[k] = (int)in.elemIdx[k];
    std::vector<std::vector<double>> mats(in.nbElem);
    size_t off = 0;
    for (unsigned e = 0; e < in.nbElem; ++e) {
      const int k = ptr[e + 1] - ptr[e];
      mats[e].assign(in.elemMat + off, in.elemMat + off + (size_t)k * k);
      off += (size_t)k * k;
    }
    from_lists(m, ptr, idx, mats);
    m.nbNode = (int)in.nbNode;
    GeneoFreeInput(&in);
    return 0;
  }
  std::string base = name.substr(name.find_last_of('/') == std::string::npos ? 0 : name.find_last_of('/') + 1);
  if (base.rfind("lib", 0) == 0) base = base.substr(3);
  if (base.size() > 3 && base.substr(base.size() - 3) == ".so") base = base.substr(0, base.size() - 3);
  if (base != "laplacian" && base != "heat") return fail("unknown input library " + name);
  std::string args = rest;
  for (char& c : args)
    if (c == '#') c = ' ';
  std::istringstream as(args);
  int size = 4, weak = 1, dim = 3, interp = 0;
  double eps = 1e-4, kmax = 1.0, lbd = 1.0, dt = 0.1;
  std::string t;
  while (as >> t) {
    if (t == "--size") as >> size;
    else if (t == "--weakScaling") as >> weak;
    else if (t == "--dim") as >> dim;
    else if (t == "--inpEps") as >> eps;
    else if (t == "--lbd") as >> lbd;
    else if (t == "--dt") as >> dt;
    else if (t == "--kappa") {
      std::string ip;
      as >> kmax >> ip;
      interp = ip == "quad" ? 1 : (ip == "lin" ? 2 : (ip == "minmax" ? 3 : 0));
    }
  }
  const int n = grid_size(size, weak, dim);
  int* nodes = nullptr;
  double* mats = nullptr;
  if (GeneoGridMesh(n, dim, eps, kmax, interp, base == "heat" ? 1 : 0, lbd, dt, nullptr, nullptr, &m.nbNode, &m.nbElem, &nodes, &mats))
    return fail("generator failed");
  m.W = 2;
  m.nodes.assign(nodes, nodes + (size_t)m.nbElem * 2);
  m.mats.assign(mats, mats + (size_t)m.nbElem * 4);
  GeneoFreeMesh(nodes, mats);
  m.grid_n = n;
  m.grid_dim = dim;
  return 0;
}

int parse_cli(int argc, const char* const* argv, Cli& o) {
  for (int i = 0; i < argc; ++i) {
    const std::string a = argv[i];
    const char* nxt = i + 1 < argc ? argv[i + 1] : nullptr;
    auto need = [&]() { return nxt != nullptr; };
    if (a == "--inpFileA" || a == "--inpLibA" || a == "--inpFileB" || a == "--partFile") {
      if (!need()) return fail("invalid option " + a);
      (a == "--inpFileA" ? o.inpFileA : a == "--inpLibA" ? o.inpLibA : a == "--inpFileB" ? o.inpFileB : o.partFile) = nxt;
      ++i;
    } else if (a == "--inpEps") { if (!need()) return fail("invalid option " + a); o.inpEps = atof(nxt); ++i; }
    else if (a == "--metisDual") o.metisDual = true;
    else if (a == "--metisNodal") o.metisDual = false;
    else if (a == "--addOverlap") { if (!need()) return fail("invalid option " + a); o.addOverlap = atoi(nxt); ++i; }
    else if (a == "--verbose") { if (!need()) return fail("invalid option " + a); o.verbose = atoi(nxt); ++i; }
    else if (a == "--np") { if (!need()) return fail("invalid option " + a); o.np = atoi(nxt); ++i; }
    else if (a == "--parts") {
      if (!need() || sscanf(nxt, "%d,%d,%d", &o.parts[0], &o.parts[1], &o.parts[2]) != 3) return fail("invalid option --parts");
      ++i;
    } else if (a == "--timing") o.timing = true;
    else if (a == "--shortRes") o.shortRes = true;
    else if (a == "--cmdLine") o.cmdLine = true;
    else if (a == "--debug") { if (nxt && nxt[0] != '-') ++i; }
    else o.pc_args.push_back(a);
  }
  if (o.np < 1) return fail("invalid --np");
  if (o.inpFileA.empty() && o.inpLibA.empty()) return fail("missing --inpFileA or --inpLibA");
  return 0;
}

std::string opt_of(const std::string& all, const std::string& key) {
  const size_t p = all.find(key + "=");
  if (p == std::string::npos) return "";
  const size_t e = all.find(';', p);
  return all.substr(p + key.size() + 1, e == std::string::npos ? std::string::npos : e - p - key.size() - 1);
}

}  // namespace

extern "C" int GeneoDriverMain(int argc, const char* const* argv) {
  Cli o;
  if (parse_cli(argc, argv, o)) return 1;
  auto t0 = clk::now();
  Mesh m;
  if (!o.inpFileA.empty() ? read_input_text(o.inpFileA, o.inpEps, m) : load_lib_input(o.inpLibA, m)) return 1;
  const double t_read = since(t0);
  t0 = clk::now();
  // ---- partition (driver:381-445) and decomposition (driver:196-379, :447-494, :643-715)
  std::vector<int> eptr(m.nbElem + 1, 0), eind, epart(m.nbElem, 0), npart(m.nbNode, 0);
  for (int e = 0; e < m.nbElem; ++e) {
    for (int a = 0; a < m.W; ++a)
      if (m.nodes[(size_t)e * m.W + a] >= 0) eind.push_back(m.nodes[(size_t)e * m.W + a]);
    eptr[e + 1] = (int)eind.size();
  }
  if (!o.partFile.empty()) {
    std::ifstream pf(o.partFile);
    if (!pf) return fail("can not open " + o.partFile);
    std::vector<int>& dst = o.metisDual ? epart : npart;
    for (size_t k = 0; k < dst.size(); ++k)
      if (!(pf >> dst[k])) return fail("bad partition file " + o.partFile);
  } else if (!o.metisDual && m.grid_n > 0 && o.parts[0] > 0) {
    const int n = m.grid_n, d[3] = {n, m.grid_dim >= 2 ? n : 1, m.grid_dim >= 3 ? n : 1};
    for (int k = 0; k < d[2]; ++k)
      for (int j = 0; j < d[1]; ++j)
        for (int i = 0; i < d[0]; ++i)
          npart[i + d[0] * (j + d[1] * k)] = (i * o.parts[0]) / d[0] + o.parts[0] * ((j * o.parts[1]) / d[1] + o.parts[1] * ((k * o.parts[2]) / d[2]));
  } else {
    int cut = 0;
    const int rc = o.metisDual ? GeneoPartMeshDual(m.nbElem, m.nbNode, eptr.data(), eind.data(), o.np, &cut, epart.data(), npart.data())
                               : GeneoPartMeshNodal(m.nbElem, m.nbNode, eptr.data(), eind.data(), o.np, &cut, epart.data(), npart.data());
    if (rc) return fail("partition failed");
  }
  GeneoDecomp dec = nullptr;
  if (GeneoDecompCreate(m.nbNode, m.nbElem, m.W, m.nodes.data(), m.mats.data(), o.np, epart.data(), npart.data(), o.metisDual ? 1 : 0,
                        o.addOverlap, &dec))
    return fail("decomposition failed (bad element or partition)");
  const double t_part = since(t0);
  t0 = clk::now();
  // ---- preconditioner (driver:1328-1367)
  GENEO_PC pc = nullptr;
  if (PCCreate_GenEO(&pc)) return fail("GenEO preconditioner is invalid");
  auto pcfail = [&](const char* what) {
    const std::string msg = std::string(what) + ": " + PCGenEOGetError(pc);
    PCDestroy_GenEO(&pc);
    GeneoDecompDestroy(&dec);
    return fail(msg);
  };
  std::vector<const char*> pargv;
  for (auto& s : o.pc_args) pargv.push_back(s.c_str());
  if (PCSetFromOptions_GenEO(pc, (int)pargv.size(), pargv.data())) return pcfail("options");
  if (PCGenEOSetSizes(pc, m.nbNode, o.np)) return pcfail("sizes");
  long long nnz = 0;
  std::vector<double> x(m.nbNode), b(m.nbNode, 0.0);
  for (int p = 0; p < o.np; ++p) {
    GeneoDomain d;
    if (GeneoDecompDomain(dec, p, 1, &d)) return pcfail("domain");
    nnz += d.neu_rowptr[d.n];
    if (o.inpFileB.empty())        // b = A (1, 2, ..., N) (driver:820-831): A = sum_p R_p^T A_Neu,p R_p, summed in domain order
      for (int i = 0; i < d.n; ++i) {
        double sum = 0.0;
        for (int k = d.neu_rowptr[i]; k < d.neu_rowptr[i + 1]; ++k) sum += d.neu_val[k] * (d.l2g[d.neu_col[k]] + 1.0);
        b[d.l2g[i]] += sum;
      }
    const GeneoCsr neu = {d.n, d.neu_rowptr, d.neu_col, d.neu_val}, dir = {d.n, d.dir_rowptr, d.dir_col, d.dir_val};
    const int rc = PCGenEOAddSubdomain(pc, p, d.n, d.l2g, d.mult, &neu, &dir);
    GeneoFreeDomain(&d);
    if (rc) return pcfail("subdomain");
  }
  const size_t bytes = sizeof(double) * (size_t)m.nbNode;
  double* d_x = (double*)GeneoDeviceAlloc(bytes);
  double* d_b = (double*)GeneoDeviceAlloc(bytes);
  double* d_r = (double*)GeneoDeviceAlloc(bytes);
  auto release = [&]() { GeneoDeviceFree(d_x); GeneoDeviceFree(d_b); GeneoDeviceFree(d_r); };
  if (!o.inpFileB.empty()) {      // --inpFileB: 'idx [value]' lines (driver:841-858)
    std::ifstream bf(o.inpFileB);
    if (!bf) { release(); return pcfail("can not open --inpFileB"); }
    std::string line;
    while (std::getline(bf, line)) {
      const size_t p = line.find_first_not_of(" \t");
      if (p == std::string::npos || line[p] == '#' || line[p] == '%') continue;
      std::istringstream ls(line);
      long idx = -1;
      double v = 1.0;
      ls >> idx;
      if (!(ls >> v)) v = 1.0;
      if (idx < 0 || idx >= m.nbNode) { release(); return pcfail("bad index in --inpFileB"); }
      b[idx] = v;
    }
  }
  GeneoH2D(d_b, b.data(), bytes);
  const double t_create = since(t0);
  if (PCGenEOSetRHS(pc, d_b) || PCSetUp_GenEO(pc)) { release(); return pcfail("GenEO - setup KO"); }
  int its = 0, reason = 0;
  double rnorm = 0.0;
  if (PCGenEOGetX0(pc, d_x)) { release(); return pcfail("GenEO - x0 KO"); }   // the initial guess the set-up wrote (geneo.cpp:1601-1607)
  if (KSPSolve_GenEO(pc, d_b, d_x, &its, &rnorm, &reason)) { release(); return pcfail("GenEO - solve KO"); }
  GeneoD2H(x.data(), d_x, bytes);
  MatMult_GenEO(pc, d_x, d_r);
  std::vector<double> ax(m.nbNode);
  GeneoD2H(ax.data(), d_r, bytes);
  double num = 0.0, den = 0.0;
  for (int i = 0; i < m.nbNode; ++i) { num += (ax[i] - b[i]) * (ax[i] - b[i]); den += b[i] * b[i]; }
  const double res_rel = den > 0.0 ? std::sqrt(num / den) : std::sqrt(num);
  GeneoInfo info;
  PCGenEOGetInfo(pc, &info);
  const std::string opts = PCGenEOGetOptionsString(pc), name = PCGenEOGetName(pc);
  const int lvl2 = atoi(opt_of(opts, "lvl2").c_str());
  // ---- output (driver:898-1231)
  if (o.cmdLine) {
    printf("CMD:");
    for (int i = 0; i < argc; ++i) printf(" %s", argv[i]);
    printf("\n");
  }
  if (o.verbose >= 1) {
    printf("The solution X is:\n");
    for (double v : x) printf("%g\n", v);
    printf("\n");
  }
  printf("INFO: nb DOFs %d, nb elements %d, nnz coefs %lld, nb partitions %d, overlap %d, metis %s\n", m.nbNode, m.nbElem, nnz, o.np,
         o.addOverlap, o.metisDual ? "dual" : "nodal");
  printf("INFO: %s ksp, eps rel %.1e, eps abs %.1e, max iterations %d\n", opt_of(opts, "ksp_type").c_str(), atof(opt_of(opts, "ksp_rtol").c_str()),
         atof(opt_of(opts, "ksp_atol").c_str()), atoi(opt_of(opts, "ksp_max_it").c_str()));
  std::string line = "INFO: " + name + " pc";
  char buf[256];
  if (name.find("ORAS") != std::string::npos) { snprintf(buf, sizeof(buf), ", optim %.2f", atof(opt_of(opts, "optim").c_str())); line += buf; }
  if (atoi(opt_of(opts, "effHybrid").c_str())) line += ", initial guess";
  line += ", L1 pcg-" + opt_of(opts, "dls1_pc") + (atoi(opt_of(opts, "hybrid").c_str()) ? " proj-fine-space" : " no-proj-fine-space");
  if (lvl2) {
    snprintf(buf, sizeof(buf), ", tau %.2f", atof(opt_of(opts, "tau").c_str()));
    line += buf;
    if (lvl2 >= 2) { snprintf(buf, sizeof(buf), ", gamma %.2f", atof(opt_of(opts, "gamma").c_str())); line += buf; }
    if (atoi(opt_of(opts, "offload").c_str())) line += ", offload";
    line += ", L2 lobpcg cholesky";
  }
  printf("%s\n", line.c_str());
  if (!o.shortRes) {
    if (lvl2) {
      std::vector<int> dims(o.np, 0);
      PCGenEOGetLocalDims(pc, dims.data(), o.np);
      int lo = dims[0], hi = dims[0];
      for (int v : dims) { lo = std::min(lo, v); hi = std::max(hi, v); }
      printf("INFO: setup - estim dimE %i (local: min %i, max %i), , real dimE %i (local: min %i, max %i), nicolaides %i\n", info.estimDimELoc,
             lo, hi, info.dimE, lo, hi, info.nicolaidesLoc);
    } else {
      printf("INFO: setup - none\n");
    }
  }
  static const char* reasons[] = {"KSP_CONVERGED_ITERATING", "", "KSP_CONVERGED_RTOL", "KSP_CONVERGED_ATOL"};
  const bool conv = reason > 0;
  std::string rs = reason >= 2 && reason <= 3 ? reasons[reason]
                   : reason == -3 ? "KSP_DIVERGED_ITS" : reason == -4 ? "KSP_DIVERGED_DTOL" : reason == -9 ? "KSP_DIVERGED_NANORINF"
                   : (conv ? "KSP_CONVERGED" : "KSP_DIVERGED");
  if (o.shortRes) printf("INFO: solve - %s\n", conv ? "converged" : "diverged");
  else
    printf("INFO: solve - %s (%s), %d iteration(s), residual norm %.10f, || AX - B || / || B || %.10f\n", conv ? "converged" : "diverged", rs.c_str(),
           its, rnorm, res_rel);
  if (o.timing) {
    printf("\nTIME: read input %.5f s, part / decomp %.5f s, create A %.5f s, solver set up %.5f s, solver iterations %.5f s, solve %.5f s\n", t_read,
           t_part, t_create, info.setupTime, info.solveTime, info.solveTime + info.setupTime);
    printf("      L1       setup: Minv %.5f s\n", info.lvl1SetupMinvTimeLoc);
    if (lvl2) printf("      L2       setup: eigen solve %.5f s, Z %.5f s, E %.5f s\n", info.lvl2SetupEigTimeLoc, info.lvl2SetupZTimeLoc, info.lvl2SetupETimeLoc);
    printf("      L1       solve: apply %.5f s - scatter %.5f s, Minv %.5f s, gather %.5f s\n", info.lvl1ApplyTimeLoc, info.lvl1ApplyScatterTimeLoc,
           info.lvl1ApplyMinvTimeLoc, info.lvl1ApplyGatherTimeLoc);
    if (lvl2) printf("      L2       solve: apply %.5f s - Zt %.5f s, Einv %.5f s, Z %.5f s\n", info.lvl2ApplyTimeLoc, info.lvl2ApplyZtTimeLoc, info.lvl2ApplyEinvTimeLoc,
                     info.lvl2ApplyZTimeLoc);
  }
  fflush(stdout);
  release();
  PCDestroy_GenEO(&pc);
  GeneoDecompDestroy(&dec);
  return 0;
}
