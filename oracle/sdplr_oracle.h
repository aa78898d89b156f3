/*
 * sdplr_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-threaded CPU restatement of the SDPLR+ numeric hot path
 * (luotuoqingshan/SDPLRPlus.jl v0.2.0: src/coreop.jl, src/lbfgs.jl, src/linesearch.jl,
 * src/preprocess.jl, src/structs.jl).  It is the parity checker for libsdplr_hip.so and the
 * "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (sdplrplus.jl_amd/) never does.
 *
 * Pinning: the reference is Julia and cannot be run here; it holds no stored
 * golden vectors.  This restatement is pinned by the reference's own self-checking test
 * identities, restated in tests/test_oracle_*.py against dense numpy recomputation
 * (test/coreop.jl:58-72,107-114,160-172,199-209; test/symlowrank.jl:6-15) and by its known-answer
 * tests (test/maxcut.jl:24,47,75; test/minimumbisection.jl:22).  Two third-party pieces of the
 * path are PARITY UNPINNED because no reference test reaches them: GenericArpack.symeigs on the
 * Lanczos tridiagonal (src/coreop.jl:509-511; replaced by an exact Sturm bisection) and
 * PolynomialRoots.roots in the line search (src/linesearch.jl:82,94; replaced by a bracketed
 * real-root finder — compare f(α*), not α*).
 *
 * The entry points mirror include/sdplr_hip.h one-for-one (prefix sdplr_oracle_ instead of
 * sdplr_hip_) so that a parity test is the same call sequence on both libraries.
 */
#ifndef SDPLR_ORACLE_H
#define SDPLR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdplr_oracle_solver sdplr_oracle_solver;

int32_t sdplr_oracle_device_count(int32_t* count);
int32_t sdplr_oracle_set_device(int32_t device);
const char* sdplr_oracle_last_error(const sdplr_oracle_solver* s);
const char* sdplr_oracle_version(void);
int32_t sdplr_oracle_device_synchronize(void);

int32_t sdplr_oracle_create(int64_t n, int64_t m, int64_t r, int64_t numlbfgsvecs,
                            sdplr_oracle_solver** out);
int32_t sdplr_oracle_set_sparse(sdplr_oracle_solver* s, int64_t index_base, int64_t n_sparse,
                                const int64_t* matptr, const int64_t* nzind,
                                const double* nzval_one, const double* nzval_two,
                                const int64_t* global_inds, int64_t nnzT,
                                const int64_t* triu_colptr, const int64_t* triu_rowval,
                                int64_t nnzS, const int64_t* full_colptr,
                                const int64_t* full_rowval, const int64_t* mappedto_triu);
int32_t sdplr_oracle_add_symlowrank(sdplr_oracle_solver* s, int64_t index_base,
                                    int64_t global_ind, int64_t s_cols, const double* B,
                                    const double* D);
int32_t sdplr_oracle_finalize(sdplr_oracle_solver* s);
int32_t sdplr_oracle_destroy(sdplr_oracle_solver* s);
int32_t sdplr_oracle_reset_rank(sdplr_oracle_solver* s, int64_t new_r);

int32_t sdplr_oracle_set_factor(sdplr_oracle_solver* s, int32_t slot, const double* host_rt);
int32_t sdplr_oracle_get_factor(sdplr_oracle_solver* s, int32_t slot, double* host_rt);
int32_t sdplr_oracle_set_vec(sdplr_oracle_solver* s, int32_t which, const double* host,
                             int64_t len);
int32_t sdplr_oracle_get_vec(sdplr_oracle_solver* s, int32_t which, double* host, int64_t len);
int32_t sdplr_oracle_set_scalar(sdplr_oracle_solver* s, int32_t which, double value);
int32_t sdplr_oracle_get_scalar(sdplr_oracle_solver* s, int32_t which, double* value);
int32_t sdplr_oracle_get_dims(const sdplr_oracle_solver* s, int64_t* n, int64_t* m, int64_t* r,
                              int64_t* h, int64_t* nnzT, int64_t* nnzS, int64_t* nnzAgg);

int32_t sdplr_oracle_A(sdplr_oracle_solver* s, int32_t u_slot, int32_t v_slot, int32_t out_vec);
int32_t sdplr_oracle_major_iteration(sdplr_oracle_solver* s, double normC, double normb, int32_t gtol_relative,
                                     int32_t ptol_relative, int32_t use_armijo, int32_t update_lambda, double sigma,
                                     double cur_gtol, double fprec_eps, int64_t max_local_iters, double time_budget_s,
                                     double* lagrangian, double* grad_norm, double* primal_vio_norm,
                                     double* last_alpha, int64_t* iters_done, int32_t* exit_reason);
int32_t sdplr_oracle_warmup(int32_t n_handles);
int32_t sdplr_oracle_trim_pools(void);
/* the lockstep batch calls of the shared ABI (include/sdplr_hip.h): here plain loops over the single-instance functions */
typedef struct sdplr_oracle_fg_item {
  sdplr_oracle_solver* s;
  double normC, normb;
  int32_t gtol_relative, ptol_relative;
  double lagrangian, grad_norm, primal_vio_norm, obj;
  int32_t status;
} sdplr_oracle_fg_item;
int32_t sdplr_oracle_batch_fg(int32_t count, sdplr_oracle_fg_item* items);
typedef struct sdplr_oracle_major_item {
  sdplr_oracle_solver* s;
  double normC, normb;
  int32_t gtol_relative, ptol_relative, use_armijo, update_lambda;
  double sigma, cur_gtol, fprec_eps;
  int64_t max_local_iters;
  double time_budget_s;
  double lagrangian, grad_norm, primal_vio_norm, last_alpha, obj;
  int64_t iters_done;
  int32_t exit_reason, status;
} sdplr_oracle_major_item;
int32_t sdplr_oracle_batch_major_iteration(int32_t count, sdplr_oracle_major_item* items);
typedef struct sdplr_oracle_dual_item {
  sdplr_oracle_solver* s;
  double trace_bound;
  int64_t iter;
  const double* v0;
  double* y_out;
  double dual_value, mineig;
  int32_t status;
} sdplr_oracle_dual_item;
int32_t sdplr_oracle_batch_dual_obj(int32_t count, sdplr_oracle_dual_item* items);
int32_t sdplr_oracle_set_sparse_coo(sdplr_oracle_solver* s, int64_t index_base, int64_t n_sparse,
                                    const int64_t* ent_ptr, const int64_t* I, const int64_t* J,
                                    const double* V, const int64_t* global_inds);
int32_t sdplr_oracle_get_layout(const sdplr_oracle_solver* s, int32_t which, int64_t* out_i, double* out_f,
                                int64_t cap, int64_t* len);
int32_t sdplr_oracle_At_preprocess(sdplr_oracle_solver* s);
int32_t sdplr_oracle_At_left(sdplr_oracle_solver* s, int32_t y_slot, int32_t x_slot);
int32_t sdplr_oracle_At_right(sdplr_oracle_solver* s, const double* x_host, double* y_host,
                              int64_t k);
int32_t sdplr_oracle_At_right_device(sdplr_oracle_solver* s, const double* x_dev, double* y_dev,
                                     int64_t k); /* the oracle's "device" is host memory */
int32_t sdplr_oracle_S_eigval(sdplr_oracle_solver* s, int64_t nev, int32_t which, int64_t ncv, double tol,
                              int64_t maxiter, const double* v0, double* evals, int64_t* n_matvec,
                              int64_t* n_converged); /* exact dense spectrum (n <= 1500); PARITY UNPINNED vs symeigs */
int32_t sdplr_oracle_factor_dot(sdplr_oracle_solver* s, int32_t slot_a, int32_t slot_b, double* out);
int32_t sdplr_oracle_get_stats(const sdplr_oracle_solver* s, int64_t* out, int32_t cap,
                               int32_t* n_written); /* all zero: no graphs here */
int32_t sdplr_oracle_f(sdplr_oracle_solver* s, double* lagrangian);
int32_t sdplr_oracle_g(sdplr_oracle_solver* s);
int32_t sdplr_oracle_fg(sdplr_oracle_solver* s, double normC, double normb,
                        int32_t gtol_relative, int32_t ptol_relative, double* lagrangian,
                        double* grad_norm, double* primal_vio_norm);
int32_t sdplr_oracle_lbfgs_clear(sdplr_oracle_solver* s);
int32_t sdplr_oracle_lbfgs_dir(sdplr_oracle_solver* s, int32_t negate, double* descent);
int32_t sdplr_oracle_descent_fallback(sdplr_oracle_solver* s);
int32_t sdplr_oracle_lbfgs_update(sdplr_oracle_solver* s, double stepsize);
int32_t sdplr_oracle_linesearch(sdplr_oracle_solver* s, double alpha_max, double* alpha,
                                double* lagrangian);
int32_t sdplr_oracle_linesearch_armijo(sdplr_oracle_solver* s, double alpha_max, double* alpha,
                                       double* lagrangian);
int32_t sdplr_oracle_axpy_R(sdplr_oracle_solver* s, double alpha);
int32_t sdplr_oracle_norms(sdplr_oracle_solver* s, double normC, double normb,
                           int32_t gtol_relative, int32_t ptol_relative, double* grad_norm,
                           double* primal_vio_norm);
int32_t sdplr_oracle_update_lambda(sdplr_oracle_solver* s);
int32_t sdplr_oracle_inner_loop(sdplr_oracle_solver* s, double normC, double normb,
                                int32_t gtol_relative, int32_t ptol_relative, int32_t use_armijo,
                                double cur_gtol, double fprec_eps, int64_t max_local_iters,
                                double time_budget_s, double* lagrangian, double* grad_norm,
                                double* primal_vio_norm, double* last_alpha, int64_t* iters_done,
                                int32_t* exit_reason);
int32_t sdplr_oracle_lanczos(sdplr_oracle_solver* s, int64_t q, const double* v0, double* alpha,
                             double* beta, int64_t* steps);
int32_t sdplr_oracle_tridiag_mineig(const double* alpha, const double* beta, int64_t k,
                                    double* mineig);
int32_t sdplr_oracle_approx_mineigval_lanczos(sdplr_oracle_solver* s, int64_t q,
                                              const double* v0, double* mineig);
int32_t sdplr_oracle_dual_obj(sdplr_oracle_solver* s, double trace_bound, int64_t iter,
                              const double* v0, double* dual_value, double* mineig);
int32_t sdplr_oracle_profile_enable(sdplr_oracle_solver* s, int32_t on);
int32_t sdplr_oracle_profile_filter(sdplr_oracle_solver* s, const char* name);
int32_t sdplr_oracle_profile_count(const sdplr_oracle_solver* s, int32_t* n_entries);
int32_t sdplr_oracle_profile_get(sdplr_oracle_solver* s, int32_t idx, char* name,
                                 int32_t name_cap, int64_t* launches, double* total_ms);

/* ---- oracle-only extras --------------------------------------------------------------------- */

/* preprocess_sparsecons (src/preprocess.jl:24-169) on a batch of nA sparse matrices given as
 * concatenated COO entry lists: matrix k owns entries ent_ptr[k] .. ent_ptr[k+1]-1 of (I, J, V)
 * (0-based ent_ptr; I/J in index_base), in the order `findnz` would yield them (column-major for
 * a CSC matrix, stored order for a COO matrix).  Outputs are malloc'ed, 0-based, and released with
 * sdplr_oracle_free.                                                                             */
typedef struct {
  int64_t n, nA, nnzT, nnzS, nnzAgg;
  int64_t* triu_colptr;   /* n+1   */
  int64_t* triu_rowval;   /* nnzT  */
  int64_t* full_colptr;   /* n+1   */
  int64_t* full_rowval;   /* nnzS  */
  int64_t* matptr;        /* nA+1  */
  int64_t* nzind;         /* nnzAgg */
  double* nzval_one;      /* nnzAgg */
  double* nzval_two;      /* nnzAgg */
  int64_t* mappedto_triu; /* nnzS  */
} sdplr_oracle_layout;
int32_t sdplr_oracle_preprocess(int64_t n, int64_t nA, int64_t index_base, const int64_t* ent_ptr,
                                const int64_t* I, const int64_t* J, const double* V,
                                sdplr_oracle_layout* out);
void sdplr_oracle_layout_free(sdplr_oracle_layout* l);

/* norm(A::SymLowRankMatrix, p), p ∈ {2 (Frobenius), Inf}; src/structs.jl:61-82; p_is_inf != 0 ⇒ Inf */
int32_t sdplr_oracle_symlowrank_norm(int64_t n, int64_t s_cols, const double* B, const double* D,
                                     int32_t p_is_inf, double* out);

/* exact-quartic scalar stage alone (src/linesearch.jl:44-112) on given coefficients:
 * biquadratic[5] ascending; returns α*, f(α*) — used to cross-check the device root finder.     */
int32_t sdplr_oracle_quartic_argmin(const double* biquadratic, double alpha_max, double* alpha,
                                    double* fval);

#ifdef __cplusplus
}
#endif
#endif
