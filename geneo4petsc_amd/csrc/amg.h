// Smoothed-aggregation AMG hierarchy for the block-diagonal local matrices (one independent
// hierarchy per subdomain, stored concatenated so every level is ONE CSR / one launch).
//
// Role: preconditioner INSIDE the batched local PCG that replaces the reference's MUMPS solve
// (geneo.cpp:94-160,:1995) and inside LOBPCG (replacing ARPACK's shift-invert, geneo.cpp:626-744).
// The operator the outer Krylov method sees is still A_Dir^-1 to -dls1_ksp_rtol; AMG only changes how
// fast the inner iteration gets there.  Set-up (aggregation, Galerkin products) runs once on the host.
#pragma once
#include <vector>

#include "backend.h"
#include "core.h"

namespace geneo {

struct AmgParams {
  int max_levels = 10;
  int coarse_size = 600;      // stop coarsening when every subdomain block is at most this large
  int smooth_degree = 1;      // Chebyshev-Jacobi smoother degree (1 = damped Jacobi)
  double smooth_ratio = 4.0;  // smoothing interval [rho/ratio, 1.1 rho]
  double strength = 0.0;      // aggregation on levels >= 1 ties only strong connections |a_ij| >= strength 0.5^l sqrt(a_ii a_jj); 0 = all
  bool single = false;        // single-vector V-cycles read single-precision companions of the level matrices (FP64 arithmetic)
};

struct AmgLevelHost {
  HostCsr A, P, R;            // P: n x nc, R = P^T.  Level 0 borrows the caller's matrix: A stays empty there
  int n = 0;                  // rows of this level
  size_t nnz = 0;
  std::vector<double> dinv;
  std::vector<int> suboff;    // nsub+1 row offsets of the subdomain blocks on this level
  double rho = 2.0;           // Gershgorin bound of D^-1 A
};

// Host set-up: levels[0].A = the given matrix.  The last level carries no P/R; its dense inverse per
// subdomain is returned in coarse_inv (row-major, concatenated; coarse_base[s] = element offset).
void amg_setup_host(const HostCsr& A, const std::vector<int>& suboff, const AmgParams& prm,
                    std::vector<AmgLevelHost>& levels, std::vector<double>& coarse_inv,
                    std::vector<int64_t>& coarse_base);

// Host part of one level of a device-built hierarchy: Jacobi diagonal + Gershgorin bound, aggregates per subdomain
// block.  Level 0 needs nothing but the caller's host matrix, so the set-up computes it EARLY, on its own thread, while
// the matrices are still being uploaded (amg_level_host_part is what AmgDevice::build_on_device runs per level otherwise).
struct AmgLevelHostPart {
  std::vector<double> dinv;
  double rho = 2.0;
  std::vector<int> agg, csub;     // aggregate of every row (global numbering of the coarse level), block offsets
  int nc = 0;
  bool last = false;              // no further coarsening from this level
};
AmgLevelHostPart amg_level_host_part(const HostCsr& A, const std::vector<int>& suboff, const AmgParams& prm, int level);

// null pivots detected and fixed in the dense coarsest blocks (singular subdomain matrices: tuneSolver, geneo.cpp:76-92)
// by the hierarchies built since the last call
int amg_null_pivots_take();

// Device hierarchy + V-cycle on row-major blocks of m vectors (m = 1: SpMV kernels).
class AmgDevice {
 public:
  ~AmgDevice();
  // fine_dev: the level-0 matrix if it already lives in HBM (borrowed, not freed), else nullptr
  void upload(const std::vector<AmgLevelHost>& levels, const std::vector<double>& coarse_inv,
              const std::vector<int64_t>& coarse_base, const AmgParams& prm, int max_m, const bk::Csr* fine_dev);
  // The same hierarchy with the Galerkin products on the device: per level the host only aggregates (it needs the
  // level's matrix: the caller's at level 0, a download of the coarse one further down) and computes the Jacobi
  // diagonal; P = (I - w D^-1 A) P0, R = P^T, A P and R A P are device sparse products (backend.h) and never leave
  // HBM.  Returns false (nothing built) when a row exceeds the product kernels' capacity: the caller then takes
  // amg_setup_host + upload.
  bool build_on_device(const HostCsr& A, const std::vector<int>& suboff, const AmgParams& prm, int max_m,
                       const bk::Csr* fine_dev, const AmgLevelHostPart* level0 = nullptr);
  // X = V(B) with zero initial guess; B, X: n0 x m row-major with leading dimensions ldb / ldx
  void vcycle(const double* B, int ldb, double* X, int ldx, int m);
  int nlevels() const { return (int)lv.size(); }
  double operator_complexity() const { return opc; }
  void free_all();
  int lp_matrices() const { return nlp; }   // level matrices that carry a single-precision companion
  // The hierarchy from level l down as a preconditioner of ITS level matrix (the eigensolver's coarse start: LOBPCG on the
  // Galerkin pencil of level 1 before the fine one); B, X: n_l x m.  Level views for that caller: matrix, transfer
  // operators, Jacobi diagonal and the subdomain offsets of a level (borrowed; valid while the hierarchy lives).
  void vcycle_from(int l, const double* B, int ldb, double* X, int ldx, int m);
  const bk::Csr& level_A(int l) const { return lv[l].A; }
  const bk::Csr& level_P(int l) const { return lv[l].P; }     // n_l x n_{l+1}
  const bk::Csr& level_R(int l) const { return lv[l].R; }
  const double* level_dinv(int l) const { return lv[l].dinv; }
  int level_rows(int l) const { return lv[l].n; }
  const std::vector<int>& level_suboff(int l) const { return lv[l].suboff; }

 private:
  struct Lvl {
    bk::Csr A, P, R;
    bk::Csr Acs;              // A diag(dinv), values only: the zero-guess sweep (EPI_PRE) then gathers b alone
    bk::Csr M;                // P - w D^-1 A P: prolongation, correction and the post-smoothing sweep as ONE product with
                              // the coarse correction (EPI_POST); empty: the two-launch form (P, then the Jacobi sweep on A)
    double* dinv = nullptr;
    double *b = nullptr, *x = nullptr, *r = nullptr, *d = nullptr, *ad = nullptr;  // n x max_m work blocks
    int n = 0;
    double rho = 2.0;
    bool own_A = true;
    bool own_lp_A = false;    // the single-precision companion of a BORROWED level-0 matrix was made here (else its owner made it)
    bool fused = false;       // A and P have no long-row remainder: fused-epilogue cycle
    std::vector<int> suboff;  // row offsets of the subdomain blocks on this level (host)
  };
  std::vector<Lvl> lv;
  bk::Chunks cch;              // chunks of the coarsest level (per subdomain)
  double* d_inv = nullptr;
  int64_t* d_invbase = nullptr;
  AmgParams prm;
  int maxm = 1;
  double opc = 1.0;
  int nlp = 0;
  void make_single();
  void alloc_level_buffers(Lvl& L, bool coarse);
  void make_column_scaled(Lvl& L);
  void make_post_matrix(Lvl& L, bk::Csr* ap, int nc);
  double jacobi_weight(const Lvl& L) const;
  void applyA(const bk::Csr& a, const double* X, int ldx, double* Y, int ldy, int m);
  void smooth(Lvl& L, const double* B, int ldb, double* X, int ldx, int m, bool zero_guess);
  void cycle(int l, const double* B, int ldb, double* X, int ldx, int m);
};

}  // namespace geneo
