// Device-primitive interface of libgeneopc.
//
// The GenEO core (core.cpp) is written once against these primitives.  The PRODUCT library
// links exactly one implementation: backend_hip.hip (hand-written gfx950 kernels).  There is no
// CPU implementation inside the product.  tests/hostsim/backend_host.cpp implements the same
// interface with serial loops and is linked ONLY into a test-side library used by the CPU
// (`-m "not gpu"`) host-logic tests; it is never built by, shipped with, or loaded from the
// geneo4petsc_amd package.
//
// All pointers are device pointers unless named h_*.  All operations are ordered on the single
// stream set with bk::set_stream (default: the null stream).  FP64 values, 32-bit indices.
// Tall-skinny blocks are row-major with an explicit leading dimension (ld >= #columns).
#pragma once
#include <cstddef>
#include <cstdint>

namespace bk {

const char* name();                       // "hip-gfx950" | "hostsim"
void set_stream(void* hip_stream);        // stream used for every launch / async copy
void* get_stream();
void sync();                              // wait for the calling thread's stream
// A host thread may move ITS launches, copies and allocations to a private stream: begin orders that stream behind
// everything queued on the main stream so far, end waits for it (results are then safe on any stream) and returns the
// thread to the main stream.  Cached device blocks carry the stream of their last user, so blocks travel safely
// between the two.  The level-1 hierarchy is built this way while the main stream runs the eigensolve.
// (after / use_after: order the private stream behind THAT stream instead of the main one -- a thread that itself runs
//  on a side stream hands bk::get_stream() to the helper threads it starts)
void side_stream_begin(void* after = nullptr, bool use_after = false);
void side_stream_end();

int   device_count();                     // GPUs visible to this process
int   set_device(int ordinal);            // binds the calling thread's (and the library's) work to that GPU; returns it, -1 on failure
int   current_device();                  // the device the library's threads are bound to (-1: not configured yet)
int   thread_device_check();             // test hook: device of a freshly started library thread after the binding (see bind_thread)
void* alloc(size_t bytes);                // HBM allocation (zero-initialised)
void  dfree(void* p);
void  alloc_cache_release();   // hipFree every block the caching allocator holds (backend_hip.hip: alloc)
// bytes: handed out now / their high-water mark / high-water mark of handed out + parked in the cache / parked now /
// hipMemGetInfo free and total; reset_peaks restarts both high-water marks at the current state (any pointer may be null)
void  mem_info(double* live, double* live_peak, double* footprint_peak, double* cached, double* dev_free, double* dev_total,
               bool reset_peaks);
void  alloc_stats(double* alloc_s, double* free_s, long long* n);   // time spent in hipMalloc / hipFree since the last call
void  h2d(void* d, const void* h, size_t bytes);
void  d2h(void* h, const void* d, size_t bytes);   // synchronous w.r.t. the stream
void  d2d(void* dst, const void* src, size_t bytes);
void  zero(void* d, size_t bytes);

// ---- CSR -----------------------------------------------------------------------------------
struct Csr {
  int n = 0;              // rows
  int64_t nnz = 0;
  int* rowptr = nullptr;  // n+1
  int* col = nullptr;     // nnz
  double* val = nullptr;  // nnz
  int* rowblk = nullptr;  // nblk+1 : first row of each LDS row block
  int nblk = 0;
  int max_row = 0;
  // CSR tiles re-laid out for the wave: 64-row slices, entries k-major inside a slice
  // (element (row i, k-th nonzero) at sl_ptr[s] + 64*k + i, zero-padded to the slice's longest
  // short row); rows longer than SELL_LONG stay in CSR and go through the long-row kernel.
  int64_t* sl_ptr = nullptr;   // nslice+1
  int* sl_col = nullptr;
  double* sl_val = nullptr;
  int nslice = 0;
  int* long_rows = nullptr;    // rows handled by the long-row kernel
  int nlong = 0;
  int64_t sl_nnz = 0;          // stored entries incl. padding
  // sliced SpMM traversal: sched[i] = i-th slice to process (nullptr: natural order); the XCD group g of the persistent
  // grid walks the schedule entries xcd_ptr[g] .. xcd_ptr[g + 1]
  int* sched = nullptr;        // nslice
  int* xcd_ptr = nullptr;      // 9
  bool alias = false;          // values-only copy of another matrix (csr_scaled_alias): the index arrays are borrowed
  bool col_scaled = false;     // the values already carry the column scaling the EPI_PRE epilogue would apply (A diag(dinv))
  bool fine = false;           // a subdomain-level (fine) operator: its launches are the ones bench.py's in-situ timer samples
  // single-precision companion of the sliced layout (preconditioner use only: the V-cycle of the local solves): values as
  // float, columns as 16-bit offsets from the slice's column bases, same sl_ptr -- 6 bytes per entry instead of 12.
  // A slice has one base (its lowest column) or, when it spans more than 65535 columns (a 64-row slice of a 187^3 block:
  // 70 002), two: the first ks entries of every row count from the first, the others from the second.
  // lp_col / lp_base stay null when two bases are not enough either (the kernels then read sl_col: 8 bytes per
  // entry); all three are null when the matrix is not on the sliced path.
  float* lp_val = nullptr;
  unsigned short* lp_col = nullptr;
  int* lp_base = nullptr;      // 4 x nslice: {b0, b1, ks, -} per slice (ks = INT_MAX: one base)
  int vec_lpr = 0;             // > 0: long / ragged rows (restriction, coarse Galerkin operators): the SpMV runs the
                               // lanes-per-row CSR kernel with this many lanes per row instead of the slices
};
Csr  csr_upload(int n, const int* h_rowptr, const int* h_col, const double* h_val);
Csr  csr_upload_raw(int n, const int* h_rowptr, const int* h_col, const double* h_val);   // CSR arrays only (no SpMV layouts)
// tentative prolongator of an aggregation: one entry 1.0 per row in column agg[i] (agg_dev: n aggregate ids in HBM);
// CSR arrays only, built on the device (the host used to fill and upload 16 bytes per row of iota and ones)
Csr  csr_tentative_prolongator(int n, const int* agg_dev);
void csr_free(Csr& a);
// ---- sparse products on the device (multigrid set-up: Galerkin products without a host round trip) ----------------
// C = A B (B has ncols_b columns) and A^T (A has ncols columns).  Columns come out sorted inside every row and
// every sum runs in a fixed order: results are bitwise reproducible.  ok = false: a row exceeded the kernels'
// per-row capacity (the caller falls back to the host product); the returned matrix is then empty.
Csr  spgemm(const Csr& a, const Csr& b, int ncols_b, bool* ok);
Csr  transpose(const Csr& a, int ncols, bool* ok);
// P = P0 - w Dinv (A P0) given AP0 = A P0 (in place on its values): P0 = piecewise constant over agg[]
void smooth_prolongator(Csr& ap0, const int* agg_dev, const double* dinv_dev, double w);
// finish the SpMV layouts of a matrix whose CSR arrays were just written on the device (spgemm / transpose leave
// them out: intermediate products never see an SpMV)
void csr_finish(Csr& a);
void csr_download(const Csr& a, int* rowptr, int* col, double* val);   // host arrays sized n+1 / nnz / nnz
// same matrix with every column index c replaced by map_dev[c] (device-side copy: no host round trip)
Csr  csr_remap_columns(const Csr& a, const int* map_dev);
// diag(row_scale) A diag(col_scale) as a values-only copy: rowptr / col / layouts are BORROWED from `a` (which must
// outlive the copy); either scaling may be null.  col_is_dinv marks A diag(dinv) for the EPI_PRE epilogue, which then
// skips its own column scaling (one gather less per entry).
Csr  csr_scaled_alias(const Csr& a, const double* row_scale, const double* col_scale, bool col_is_dinv);
void spmv(const Csr& a, const double* x, double* y);                    // y = A x
// In-situ timing of the SpMV launches issued between start and stop: every `every`-th launch is
// bracketed by two HIP events on the backend stream (no host sync until stop).
void spmv_profile_start(int every, double min_bytes);   // only launches moving >= min_bytes algorithmic bytes
void spmv_profile_stop(double* ms_sum, double* bytes_sum, long long* nsampled, long long* nlaunch);
// all kernel classes at once: fine-level SpMV / SpMM (Csr::fine), MFMA Gram and block update (p, q >= 32)
// PROF_LP: the single-precision-companion passes of the local solves' V-cycle over the fine-level operators
enum { PROF_SPMV = 0, PROF_SPMM = 1, PROF_GRAM = 2, PROF_BLOCKMUL = 3, PROF_LP = 4, PROF_NCLASS = 5 };
void kernel_profile_start(int every, double spmv_min_bytes);
void kernel_profile_stop();
void kernel_profile_get(int cls, double* ms_sum, double* bytes_sum, double* flops_sum, long long* nsampled, long long* nlaunch);
bool spmv_profiling();   // between start and stop (callers replay one graph in eight as direct launches so that they are sampled)
// Y = post .* (A (pre .* X)); X (ldx), Y (ldy) row-major with m columns; pre/post may be null
void spmm_strided(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                  const double* post);
// Two operators on ONE sliced pattern, one pass over X (LOBPCG's A W and B W): sell_values_on lays the values of b out on
// a's sliced pattern (device array of a.sl_nnz doubles, zeros where b has no entry; nullptr when pattern(b) is not
// contained in pattern(a) row by row or a is not on the sliced path); spmm_dual computes Y1 = (a's pattern, v1) X and
// Y2 = (a's pattern, v2) X, m = 16 | 32 | 64 columns, bit-identical to two separate products.
double* sell_values_on(const Csr& a, const Csr& b);
bool spmm_dual_available(const Csr& a, int m);
void spmm_dual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* Y1, double* Y2, int ldy,
               int m);
struct Chunks;
// R = mask .* ((a's pattern, v1) X - (a's pattern, v2) X diag(lam)) per subdomain of c (lam, mask: nsub x m): LOBPCG's
// residual block straight from X, neither product written (R: n x m with leading dimension ldr)
void spmm_dual_residual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* R, int ldr,
                        int m, const Chunks& c, const double* lam, const double* mask);
// Fused epilogues of the multigrid cycle (one launch instead of SpMV + 1-2 vector kernels); blocks are
// row-major with m columns, m = 1 runs the sliced SpMV kernel.  A must be square for JAC / PRE.
//   EPI_RES : Y = B - A X
//   EPI_ADD : Y = Z + A X                         (prolongation + correction; Y must not alias Z or X)
//   EPI_JAC : Y = X + w dinv .* (B - A X)         (damped-Jacobi sweep, out of place)
//   EPI_PRE : Z = w dinv .* B ;  Y = B - A Z      (zero-guess sweep + residual; X unused; Z may be null: Z not stored)
//   EPI_POST: Y = w dinv .* (Z + B) + A X         (A = P - w D^-1 A P, Z = the cycle's right-hand side b, B = the residual
//                                                  r1 of the zero-guess sweep: prolongation + correction + post-smoothing
//                                                  sweep in one launch, see AmgDevice::cycle)
enum { EPI_NONE = 0, EPI_RES = 1, EPI_ADD = 2, EPI_JAC = 3, EPI_PRE = 4, EPI_POST = 5 };
// M = P - w diag(dinv) AP, written over the values of AP (pattern(P) must be contained in pattern(AP): true for
// AP = A P with a full diagonal in A).  Returns false (AP untouched in pattern, values undefined) when an entry of P has
// no slot in AP.
bool post_matrix(Csr& ap, const Csr& p, const double* dinv, double w);
// Builds the single-precision companion (false: the matrix is not on the sliced path).  index_owner: the matrix a
// values-only alias borrows its index arrays from (its companion must exist already).
bool csr_make_lp(Csr& a, const Csr* index_owner = nullptr);
void csr_free_lp(Csr& a);
inline bool csr_has_lp(const Csr& a) { return a.lp_val != nullptr; }
// y = A x / the fused epilogues of spmm_fused for ONE contiguous vector, reading the companion (FP64 arithmetic)
void spmv_lp(const Csr& a, const double* x, double* y);
void spmv_fused_lp(const Csr& a, int epi, const double* x, double* y, const double* b, double* z, const double* dinv, double w);
bool csr_fusable(const Csr& a);   // no long-row remainder and the sliced layout is in use
void spmm_fused(const Csr& a, int epi, const double* X, int ldx, double* Y, int ldy, int m, const double* B, int ldb,
                double* Z, int ldz, const double* dinv, double w);
void csr_diag(const Csr& a, double* diag);
// x[i] <- 1 / x[i] in place; returns the number of entries that are not > 0 (those are left untouched)
int  recip_positive(double* x, int n);

// ---- index kernels -------------------------------------------------------------------------
void gather(double* out, const double* in, const int* idx, int n);                       // out[i]=in[idx[i]]
void gather_mul(double* out, const double* in, const int* idx, const double* d, int n);  // ... * d[i]
// out[e] (+)= sum_{k in [ptr[e],ptr[e+1])} in[idx[k]]   (fixed order)
void segsum(double* out, const double* in, const int* ptr, const int* idx, int nseg, bool accumulate);

// row-major blocks of w columns (ld = w): out[i][:] = in[idx[i]][:] ; out[e][:] (+)= sum_k in[idx[k]][:]
void gather_rows(double* out, const double* in, const int* idx, int n, int w);
void segsum_rows(double* out, const double* in, const int* ptr, const int* idx, int nseg, int w, bool accumulate);

// ---- BLAS-1 --------------------------------------------------------------------------------
void set(double* x, double v, int n);
void copy(double* y, const double* x, int n);
void axpy(double* y, double a, const double* x, int n);             // y += a x
void axpby(double* y, double a, const double* x, double b, int n);  // y = a x + b y
void xmy(double* y, const double* x, const double* d, int n);       // y = x .* d
void axpy_dev(double* y, const double* a_dev, double sign, const double* x, int n);  // y += sign*a[0]*x
void dot(const double* x, const double* y, int n, double* out_dev);  // deterministic

// ---- chunked (per-subdomain) kernels on the concatenated local space ------------------------
// The local space L is the concatenation of the rank's subdomains, cut into chunks of at most
// CHUNK entries that never straddle two subdomains.  One workgroup handles one chunk.
constexpr int CHUNK = 1024;
struct Chunks {
  int nchunk = 0, nsub = 0, n = 0;
  int* start = nullptr;    // nchunk
  int* len = nullptr;      // nchunk
  int* sub = nullptr;      // nchunk
  int* subptr = nullptr;   // nsub+1 : first chunk of each subdomain
  int* suboff = nullptr;   // nsub+1 : first row of each subdomain
  double* partial = nullptr;  // 4*nchunk scratch
  double* totals = nullptr;   // 4*nsub: per-subdomain totals of the partial slots (large subdomains: reduced once per launch)
  int maxsub_chunks = 0;      // chunks of the largest subdomain
};
// per-subdomain reductions of chunk partials switch to their cooperative forms above this many chunks per subdomain
// (default 1024, GENEO_PAR_REDUCE_MIN; changing it while a HIP graph of the solver exists is not supported)
void set_par_reduce_min(int chunks);
int get_par_reduce_min();
Chunks chunks_upload(int nsub, const int* h_suboff /*nsub+1: absolute first rows, h_suboff[0] need not be 0*/);
void   chunks_free(Chunks& c);
// out[s*stride + slot] = sum over subdomain s of x.*y
void seg_dot(const Chunks& c, const double* x, const double* y, double* out, int stride, int slot);

// Batched Jacobi-PCG: one independent CG per subdomain, scalars in sc[s*8+k]:
//   0 rz(parity 0) 1 rz(parity 1) 2 pAp 3 rr 4 alpha 5 beta 6 active(0/1) 7 rr0
void cg_start(const Chunks& c, double* sc, double* x, double* r, double* z, double* p, const double* b,
              const double* dinv);                                       // x=0 r=b z=dinv.*r p=z
void seg_pap(const Chunks& c, const double* p, const double* q);         // chunk partials of p.q (slot 0)
void seg_partial(const Chunks& c, const double* x, const double* y, int slot);  // chunk partials of x.y
void cg_set_rz(const Chunks& c, double* sc);   // sc rz slots <- partial slot 1 (after cg_start with dinv == nullptr)
// (cg_start / cg_update accept dinv == nullptr: z and the r.z partial are then left to the caller,
//  who applies its own preconditioner and calls seg_partial(c, r, z, 1))
// X_s = Inv_s B_s for symmetric dense blocks Inv_s (n_s x n_s at inv + base[s]); blocks of m vectors
void dense_sym_apply(const Chunks& c, const double* inv, const int64_t* base, const double* B, int ldb, double* X,
                     int ldx, int m);
void cg_update(const Chunks& c, double* sc, int parity, double* x, double* r, double* z, const double* p,
               const double* q, const double* dinv);                     // alpha; x,r,z; partials
void cg_direction(const Chunks& c, double* sc, int parity, double* p, const double* z, double tol2);

// ---- tall-skinny block kernels, per subdomain -----------------------------------------------
// G[s] (p x q row-major at G + s*p*q) = S_s^T T_s
void gram(const Chunks& c, const double* S, int lds, int p, const double* T, int ldt, int q, double* G);
// G[s] ((p1 + p2) x q) = [S1 | S2]^T T: two left blocks that live in different buffers against ONE pass over T
void gram2(const Chunks& c, const double* S1, int lds1, int p1, const double* S2, int lds2, int p2, const double* T, int ldt,
           int q, double* G);
// Y_s (+)= S_s C_s : S (n x p), C[s] (p x q row-major at C + s*p*q), Y (n x q)
void block_mul(const Chunks& c, const double* S, int lds, int p, const double* C, int q, double* Y, int ldy,
               bool accumulate);
// R = AX - BX diag(lam_s) ; nrm[s*m+j] = ||R_j||_2^2 over subdomain s ; lam[s*m+j]
void block_residual(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam,
                    int m, double* R, int ldr, double* nrm);
void block_colnorm(const Chunks& c, const double* X, int ldx, int m, double* nrm);   // squared 2-norms
// One pass for the LOBPCG convergence test: R = mask .* (AX - BX diag(lam)) (colmask[s*m+j], may be null) and
// nrm3[s*3m + j] = ||AX_j - lam_j BX_j||^2 (unmasked), nrm3[s*3m + m + j] = ||AX_j||^2, nrm3[s*3m + 2m + j] = ||BX_j||^2
void block_residual_norms(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam,
                          int m, double* R, int ldr, const double* colmask, double* nrm3);
// Fused Rayleigh-Ritz update of one LOBPCG iteration for m = 32 (S, AS, BS: n x 96 row-major, [X | P | W]):
//   [X' P'] = S C (C: nsub x 96 x 64 row-major with the structure core.cpp gives it: the P columns equal the X columns on
//   the P / W rows for kept pairs, zero for the others -> keep[s*32+j] in {0, 1}), the same for AS and BS, written to
//   columns 0..63 of T, AT, BT, and R (n x 32 contiguous) = mask .* (A X' - B X' diag(lam)).
void lobpcg_update32(const Chunks& c, const double* S, const double* AS, const double* BS, const double* C,
                     const double* keep, const double* lam, const double* mask, double* T, double* AT, double* BT,
                     double* R);
// the basis part of it alone: columns 0..63 of T = [X' P'] from S (the "lean" iteration carries no A S / B S)
void lobpcg_update32_basis(const Chunks& c, const double* S, const double* C, const double* keep, double* T);
bool lobpcg_update32_available();
void block_axpby(double* Y, int ldy, double a, const double* X, int ldx, double b, int n, int m);
// Y = a * d .* X + b * Y  (row scaling by d[i])
void block_rowscale(double* Y, int ldy, const double* X, int ldx, const double* d, double a, double b, int n,
                    int m);
// damped-Jacobi step: zero_guess ? X = w*dinv.*B : X += w*dinv.*(B - AX)   (AX contiguous n x m)
void jacobi_step(double* X, int ldx, const double* B, int ldb, const double* AX, const double* dinv, double w, int n,
                 int m, bool zero_guess);
// fused Chebyshev step on contiguous n x m blocks (z strided): r -= ad ; d = a*dinv.*r + b*d ; z += d
void cheb_update(double* r, const double* ad, double* d, double* z, int ldz, const double* dinv, double a, double b,
                 int n, int m);
void block_colscale(const Chunks& c, double* X, int ldx, int m, const double* colscale);  // X[:,j]*=cs[s*m+j]
// deterministic counter-based start block; column 0 is the constant vector
void block_init(const Chunks& c, double* X, int ldx, int m, const int* sub_gid, uint64_t seed);
// Z_s (column-major, k_s columns of length n_s, at Z + zbase[s]):  Z_s[j][i] = d[i] * X[i][sel[s*m+j]]
// (sel < 0 : the constant vector)
void block_extract(const Chunks& c, const double* X, int ldx, int m, const double* d, const int* sel,
                   const int* ksub, const int64_t* zbase, double* Z);

// ---- coarse space ---------------------------------------------------------------------------
// yE[zoff[s]+j] = sum_i Z_s[j][i] * xL[i] ; entries of yE not owned by a local subdomain are zeroed
void zt_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff,
              int kmax, const double* xL, double* yE, int dimE_total);
// wL[i] (+)= sum_j Z_s[j][i] * yE[zoff[s]+j]
void z_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff,
             const double* yE, double* wL, bool accumulate);

// y <- (L L^T)^-1 y for a dense Cholesky factor held twice (L and L^T, n x n row-major, device): the replicated coarse
// solve, in place and stream-ordered.  Same arithmetic order as dense::cholesky_solve_lu.  false: n beyond the kernel's
// capacity (1024), nothing done -- the caller takes the host path.
bool chol_solve(const double* L, const double* LT, int n, double* y);

// ---- HIP graphs -------------------------------------------------------------------------------
// Launch-bound inner loops (one inner PCG chunk = ~70 small kernels) are captured once and replayed.
// Between begin and end every backend launch is recorded instead of executed: no allocation, copy to
// the host or synchronisation may happen in between.
bool  graph_capture_begin();          // false: capture unavailable (the caller runs the launches directly)
void* graph_capture_end();            // executable graph (nullptr on failure)
void  graph_launch(void* exec);
void  graph_destroy(void* exec);

// ZR (n_L x kp row-major, zero padded) from the per-subdomain column-major Z (blocked assembly of E)
void z_rowmajor(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, double* ZR, int kp);

// ---- misc -------------------------------------------------------------------------------------
void  set_spmv_kind(int kind);  // 0: LDS row-block kernel, 1: 64-row sliced kernel (default)
const char* spmv_kernel_name();
void  set_mfma(bool enable);   // false: run the plain-FMA twins of the MFMA kernels (validation)
// validation switches between two device forms of the same arithmetic ("spgemm_fill_scan", "gram_flat"); false: unknown name
bool  set_variant(const char* name, int value);
// Page-locked host memory for the small per-iteration transfers of LOBPCG (Gram rows down, Rayleigh-Ritz coefficients up):
// a copy to or from it is ONE DMA; from pageable memory the runtime stages it in pieces and synchronises in between.
void* pinned_alloc(size_t bytes);
void  pinned_free(void* p);
// stream-ordered upload from pinned memory, NO synchronisation: the source must stay untouched until the calling thread's
// stream has been synchronised (or has passed a later synchronous copy)
void  h2d_async(void* d, const void* h_pinned, size_t bytes);
int   selftest_mfma_f64();   // 0 = the f64 MFMA operand/result lane maps are as the kernels assume
void* event_create();
// device-to-host copy on a side stream as soon as `event` (recorded on the library stream) has completed: the library
// stream keeps running the launches queued behind the event
void  d2h_after(void* host, const void* dev, size_t bytes, void* event);
void  event_record(void* ev);
float event_elapsed_ms(void* a, void* b);   // syncs on b
void  event_destroy(void* ev);
double hash_unit_host(uint64_t seed, uint64_t gid, uint64_t row, uint64_t colj);

}  // namespace bk
