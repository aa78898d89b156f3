/*
 * sdplr_hip.h — C ABI of libsdplr_hip.so: the MI355X (gfx950) device backend for the
 * numeric hot path of the SDPLR+ Burer–Monteiro augmented-Lagrangian SDP solver.
 *
 * This is the drop-in boundary described in SURVEY.md §8b.  Every entry point replaces one
 * operator (or one group of in-loop BLAS-1 calls) of the reference; the reference file:line it
 * replaces is cited next to each declaration (paths relative to the reference checkout,
 * luotuoqingshan/SDPLRPlus.jl v0.2.0).
 *
 * Conventions
 *   - plain C types only: pointers, int64_t sizes, doubles; no torch / C++ types.
 *   - every function returns an int32 status (0 = OK, < 0 = error, see SDPLR_ERR_*); the library
 *     never aborts or exits.  sdplr_hip_last_error() returns a human-readable message.
 *   - all arithmetic is IEEE FP64.
 *   - factor-shaped arrays ("Rt", "Gt", "dirt", L-BFGS s/y) are r×n column-major with leading
 *     dimension r — exactly Julia's `Rt` memory (src/structs.jl:195,236): row i of R is r
 *     contiguous doubles.  No transposition ever happens at the boundary.
 *   - index arrays handed to sdplr_hip_set_sparse may be 1-based (Julia) or 0-based; the caller
 *     says which with `index_base`.  They are converted to 0-based int32 once, on the host.
 *   - host pointers are borrowed only for the duration of the call.
 *   - calls on one handle must be serialised by the caller; each handle owns one HIP stream, and
 *     scalar-returning calls synchronise that stream only.  Different handles may be driven
 *     from different host threads / processes concurrently.
 *   - without a usable HIP device every compute entry point returns SDPLR_ERR_NO_DEVICE: there is
 *     no CPU fallback in this library.
 */
#ifndef SDPLR_HIP_H
#define SDPLR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdplr_hip_solver sdplr_hip_solver; /* opaque handle ("HIPAux" + device SolverVars) */

/* ---- status codes -------------------------------------------------------------------------- */
#define SDPLR_OK 0
#define SDPLR_ERR_INVALID_ARG (-1)
#define SDPLR_ERR_HIP (-2)         /* a HIP runtime call failed; message in last_error         */
#define SDPLR_ERR_NOT_DESCENT (-3) /* src/linesearch.jl:60-62: cubic[1] > eps()                 */
#define SDPLR_ERR_STATE (-4)       /* call-order violation (e.g. operator before finalize)     */
#define SDPLR_ERR_NO_DEVICE (-5)   /* no usable gfx950 device                                  */
#define SDPLR_ERR_ALLOC (-6)
#define SDPLR_ERR_UNSERVED (-7)    /* batch calls: the call returned before this item was served   */

/* ---- factor slots (r×n col-major, ld = r) ---------------------------------------------------- */
#define SDPLR_F_RT 0          /* var.Rt   src/structs.jl:195                                  */
#define SDPLR_F_GT 1          /* var.Gt   src/structs.jl:196                                  */
#define SDPLR_F_DIRT 2        /* dirt     src/sdplr.jl:174                                    */
#define SDPLR_F_LBFGS_S 100   /* + j, j = 0..h-1: lbfgshis.vecs[j+1].s   src/lbfgs.jl:9       */
#define SDPLR_F_LBFGS_Y 200   /* + j, j = 0..h-1: lbfgshis.vecs[j+1].y   src/lbfgs.jl:11      */
#define SDPLR_F_SCRATCH 300   /* + k, k = 0,1: two factor-shaped scratch arrays (allocated on first use) for
                                 operator calls on arrays that are not solver state: the Ut / Vt / x / y
                                 arguments of 𝒜!(out, aux, Ut[, Vt]) and 𝒜t!(y, x, aux, var) when a caller
                                 hands in its own matrices (src/coreop.jl:36,54,260)                   */

/* ---- vector slots (src/structs.jl:197-222) ---------------------------------------------------- */
#define SDPLR_V_LAMBDA 0      /* λ              m                                            */
#define SDPLR_V_LAMBDA_UB 1   /* λ_ub           m   (+Inf equality, 0 inequality)            */
#define SDPLR_V_B 2           /* b              m   (data.b, src/structs.jl:155)             */
#define SDPLR_V_Y 3           /* y              m+1                                          */
#define SDPLR_V_PV_RAW 4      /* primal_vio_raw m+1 (last slot = objective)                  */
#define SDPLR_V_PV_LB 5       /* primal_vio_lb  m                                            */
#define SDPLR_V_PV 6          /* primal_vio     m                                            */
#define SDPLR_V_A_RD 7        /* A_RD           m+1                                          */
#define SDPLR_V_A_DD 8        /* A_DD           m+1                                          */
#define SDPLR_V_LBFGS_RHO 9   /* ρ_j            h   (src/lbfgs.jl:13)                        */
#define SDPLR_V_LBFGS_A 10    /* a_j            h   (src/lbfgs.jl:15)                        */
#define SDPLR_V_UVT 11        /* aux.UVt        nnzT (src/structs.jl:288)                    */
#define SDPLR_V_TRIU_S_NZVAL 12 /* aux.triu_sparse_S.nzval  nnzT                             */
#define SDPLR_V_S_NZVAL 13    /* aux.sparse_S.nzval         nnzS                             */
#define SDPLR_V_SCRATCH 14    /* m+1: scratch output of 𝒜! for callers' own `out` vectors     */

/* ---- scalar slots ------------------------------------------------------------------------- */
#define SDPLR_S_SIGMA 0        /* var.σ[]   src/structs.jl:204                                 */
#define SDPLR_S_OBJ 1          /* var.obj[] src/structs.jl:205                                 */
#define SDPLR_S_LBFGS_LATEST 2 /* lbfgshis.latest[] (1-based, starts at h) src/lbfgs.jl:27,45  */

/* ---- device management -------------------------------------------------------------------- */
int32_t sdplr_hip_device_count(int32_t* count);
int32_t sdplr_hip_set_device(int32_t device); /* process-wide and sticky: recorded, and every later library call — from
                                                  any host thread — binds its thread to it; a handle remembers the device
                                                  it was created on and its entry points rebind to THAT (one rank = one GPU) */
const char* sdplr_hip_last_error(const sdplr_hip_solver* s); /* s may be NULL: the calling thread's last error of a call without a handle (create, warmup, batch argument checks) */
const char* sdplr_hip_version(void);
int32_t sdplr_hip_device_synchronize(void); /* hipDeviceSynchronize on the current device */
/* Optional: primes the library's pools (HIP streams, events, pinned staging) for n_handles handles alive at once, so
 * that the first batch of solves does not pay for them (a HIP stream costs milliseconds to create).                */
int32_t sdplr_hip_warmup(int32_t n_handles);
/* The pools keep device blocks (≤ SDPLR_HIP_POOL_MAX_MB, default 4096), streams, events and pinned staging blocks of
 * destroyed handles for the life of the process; this returns everything no live handle uses to the HIP runtime (a process
 * that shares the GPU with torch / RCCL calls it between batches; the library calls it itself when hipMalloc runs dry). */
int32_t sdplr_hip_trim_pools(void);

/* ---- construction: replaces SolverVars + SolverAuxiliary construction ------------------------
 * src/sdplr.jl:114-123, src/structs.jl:225-263 (SolverVars), :296-361 (SolverAuxiliary),
 * src/lbfgs.jl:35-47 (lbfgs_init: h zeroed (s,y,ρ,a) slots, latest = h).  Any numlbfgsvecs ≤ 4096: up to 16 the two-loop
 * recursion runs in Gram form (one pass over the history per direction; ≤ 4: fused with the step), beyond that as the
 * reference writes it (one dot + one axpy kernel per history vector).
 * Order: create → [set_sparse] → [add_symlowrank]* → finalize → set_factor/set_vec … → operators. */
int32_t sdplr_hip_create(int64_t n, int64_t m, int64_t r, int64_t numlbfgsvecs,
                         sdplr_hip_solver** out);

/* Aggregated sparse layout produced by preprocess_sparsecons (src/preprocess.jl:24-169) and held
 * in SolverAuxiliary (src/structs.jl:278-288).  Field ↔ argument:
 *   n_sparse_matrices            n_sparse
 *   triu_agg_sparse_A_matptr     matptr        [n_sparse+1]
 *   triu_agg_sparse_A_nzind      nzind         [nnzAgg]   position in the triu pattern
 *   triu_agg_sparse_A_nzval_one  nzval_one     [nnzAgg]
 *   triu_agg_sparse_A_nzval_two  nzval_two     [nnzAgg]
 *   sparse_As_global_inds        global_inds   [n_sparse] index into the (m+1)-vectors
 *   triu_sparse_S.colptr/rowval  triu_colptr [n+1], triu_rowval [nnzT]
 *   sparse_S.colptr/rowval       full_colptr [n+1], full_rowval [nnzS]
 *   agg_sparse_A_mappedto_triu   mappedto_triu [nnzS]
 * index_base is 1 for Julia arrays, 0 for C/numpy arrays.                                     */
int32_t sdplr_hip_set_sparse(sdplr_hip_solver* s, int64_t index_base, int64_t n_sparse,
                             const int64_t* matptr, const int64_t* nzind,
                             const double* nzval_one, const double* nzval_two,
                             const int64_t* global_inds, int64_t nnzT,
                             const int64_t* triu_colptr, const int64_t* triu_rowval,
                             int64_t nnzS, const int64_t* full_colptr,
                             const int64_t* full_rowval, const int64_t* mappedto_triu);

/* The same layout built INSIDE the library: preprocess_sparsecons (src/preprocess.jl:24-169) on the sparse matrices
 * themselves, given as concatenated COO triplets in findnz order (src/preprocess.jl:66-67): matrix k owns entries
 * ent_ptr[k] .. ent_ptr[k+1]-1 of (I, J, V), both triangles present as in the reference's input; global_inds as above.
 * Replaces the reference's host-side preprocessing call (src/structs.jl:331-339).  Use instead of set_sparse.         */
int32_t sdplr_hip_set_sparse_coo(sdplr_hip_solver* s, int64_t index_base, int64_t n_sparse,
                                 const int64_t* ent_ptr, const int64_t* I, const int64_t* J,
                                 const double* V, const int64_t* global_inds);
/* The aggregated layout held between set_sparse[_coo] and finalize, 0-based (SolverAuxiliary's fields,
 * src/structs.jl:278-288).  which: 0 matptr, 1 nzind, 2 global_inds, 3 triu_colptr, 4 triu_rowval, 5 full_colptr,
 * 6 full_rowval, 7 mappedto_triu (→ out_i); 8 nzval_one, 9 nzval_two (→ out_f).  Copies min(*len, cap) entries.       */
int32_t sdplr_hip_get_layout(const sdplr_hip_solver* s, int32_t which, int64_t* out_i, double* out_f,
                             int64_t cap, int64_t* len);

/* One SymLowRankMatrix B·D·Bᵀ (src/structs.jl:11-24): B is n×s column-major, D the s diagonal
 * entries; global_ind indexes the (m+1)-vectors (m+1 ⇒ the cost matrix C), in index_base.      */
int32_t sdplr_hip_add_symlowrank(sdplr_hip_solver* s, int64_t index_base, int64_t global_ind,
                                 int64_t s_cols, const double* B, const double* D);

int32_t sdplr_hip_finalize(sdplr_hip_solver* s); /* uploads layout, allocates all device state */
int32_t sdplr_hip_destroy(sdplr_hip_solver* s);

/* rank_update! (src/coreop.jl:518-526) is a cold restart at a new rank: every factor-shaped array
 * and y, primal_vio_raw, primal_vio, A_RD, A_DD, λ are zeroed, L-BFGS history re-initialised;
 * λ_ub, primal_vio_lb, b and the layout are kept.  The caller then uploads Rt0, λ0 and σ0.     */
int32_t sdplr_hip_reset_rank(sdplr_hip_solver* s, int64_t new_r);

/* ---- state transfer ------------------------------------------------------------------------ */
int32_t sdplr_hip_set_factor(sdplr_hip_solver* s, int32_t slot, const double* host_rt);
int32_t sdplr_hip_get_factor(sdplr_hip_solver* s, int32_t slot, double* host_rt);
int32_t sdplr_hip_set_vec(sdplr_hip_solver* s, int32_t which, const double* host, int64_t len);
int32_t sdplr_hip_get_vec(sdplr_hip_solver* s, int32_t which, double* host, int64_t len);
int32_t sdplr_hip_set_scalar(sdplr_hip_solver* s, int32_t which, double value);
int32_t sdplr_hip_get_scalar(sdplr_hip_solver* s, int32_t which, double* value);
int32_t sdplr_hip_get_dims(const sdplr_hip_solver* s, int64_t* n, int64_t* m, int64_t* r,
                           int64_t* h, int64_t* nnzT, int64_t* nnzS, int64_t* nnzAgg);

/* ---- constraint operator ------------------------------------------------------------------ */
/* 𝒜!(out, aux, Ut)      src/coreop.jl:36-49   (v_slot < 0)  : out = 𝒜(UUᵀ)
 * 𝒜!(out, aux, Ut, Vt)  src/coreop.jl:54-70   (v_slot >= 0) : out = 𝒜((UVᵀ+VUᵀ)/2)
 * covering 𝒜_sparse! :72-113, 𝒜_sparse_formUUt!/UVt! :174-203, 𝒜_symlowrank! :132-151.
 * out_vec ∈ {SDPLR_V_PV_RAW, SDPLR_V_A_RD, SDPLR_V_A_DD}.                                     */
int32_t sdplr_hip_A(sdplr_hip_solver* s, int32_t u_slot, int32_t v_slot, int32_t out_vec);

/* 𝒜t_preprocess!(var, aux)  src/coreop.jl:248-258 (+ :205-227): S.nzval from the device y.     */
int32_t sdplr_hip_At_preprocess(sdplr_hip_solver* s);
/* 𝒜t!(y, x, aux, var)  src/coreop.jl:260-279: y = x·S + Σ coeff·x·B·D·Bᵀ on factor slots.        */
int32_t sdplr_hip_At_left(sdplr_hip_solver* s, int32_t y_slot, int32_t x_slot);
/* 𝒜t!(y, aux, x, var)  src/coreop.jl:281-300: y = S·x + Σ coeff·B·D·Bᵀ·x; x, y are n×k
 * column-major HOST arrays (the Lanczos path keeps its vectors on the device, see below).      */
int32_t sdplr_hip_At_right(sdplr_hip_solver* s, const double* x_host, double* y_host, int64_t k);
/* The same product on DEVICE vectors (x_dev, y_dev: n×k column-major in this GPU's memory, e.g. a ROCArray /
 * torch tensor; must not alias): nothing crosses PCIe.  Runs on the handle's stream and returns when y is
 * complete.  What an eigensolver living on the device calls (SDP_S_eigval's operator, src/coreop.jl:363-366). */
int32_t sdplr_hip_At_right_device(sdplr_hip_solver* s, const double* x_dev, double* y_dev, int64_t k);

/* ---- Lagrangian value / gradient ----------------------------------------------------------- */
/* f!(data, var, aux)  src/coreop.jl:11-31 */
int32_t sdplr_hip_f(sdplr_hip_solver* s, double* lagrangian);
/* g!(var, aux)        src/coreop.jl:305-317 (copy2y_λ_sub_pvio! :229, 𝒜t_preprocess!, 𝒜t!, ×2) */
int32_t sdplr_hip_g(sdplr_hip_solver* s);
/* fg!(…)              src/coreop.jl:323-349; *_relative = (config.*tol_mode == :relative)       */
int32_t sdplr_hip_fg(sdplr_hip_solver* s, double normC, double normb, int32_t gtol_relative,
                     int32_t ptol_relative, double* lagrangian, double* grad_norm,
                     double* primal_vio_norm);

/* ---- L-BFGS ------------------------------------------------------------------------------- */
/* lbfgs_clear!  src/lbfgs.jl:52-59 */
int32_t sdplr_hip_lbfgs_clear(sdplr_hip_solver* s);
/* lbfgs_dir!(dirt, his, Gt; negate) src/lbfgs.jl:77-124, followed by descent = dot(dirt, Gt)
 * (src/sdplr.jl:201).  descent may be NULL.  A NaN descent is returned as NaN.                 */
int32_t sdplr_hip_lbfgs_dir(sdplr_hip_solver* s, int32_t negate, double* descent);
/* steepest-descent fallback, src/sdplr.jl:203-204: Gt ← −Gt; dirt ← Gt                           */
int32_t sdplr_hip_descent_fallback(sdplr_hip_solver* s);
/* lbfgs_update!(dirt, his, Gt, stepsize) src/lbfgs.jl:129-149 (dirt is scaled in place)          */
int32_t sdplr_hip_lbfgs_update(sdplr_hip_solver* s, double stepsize);

/* ---- line search ---------------------------------------------------------------------------- */
/* linesearch!(var, aux, dirt; α_max)  src/linesearch.jl:4-127 — exact quartic.
 * Returns SDPLR_ERR_NOT_DESCENT where the reference throws (:60-62).                           */
int32_t sdplr_hip_linesearch(sdplr_hip_solver* s, double alpha_max, double* alpha,
                             double* lagrangian);
/* linesearch_armijo!(var, aux, dirt; α_max)  src/linesearch.jl:139-191                           */
int32_t sdplr_hip_linesearch_armijo(sdplr_hip_solver* s, double alpha_max, double* alpha,
                                    double* lagrangian);

/* ---- in-loop BLAS-1 of _sdplr ---------------------------------------------------------------- */
/* axpy!(α, dirt, var.Rt)  src/sdplr.jl:219 */
int32_t sdplr_hip_axpy_R(sdplr_hip_solver* s, double alpha);
/* grad_norm, primal_vio_norm  src/sdplr.jl:224-234 */
int32_t sdplr_hip_norms(sdplr_hip_solver* s, double normC, double normb, int32_t gtol_relative,
                        int32_t ptol_relative, double* grad_norm, double* primal_vio_norm);
/* λ update at the end of a major iteration, src/sdplr.jl:358-362 */
int32_t sdplr_hip_update_lambda(sdplr_hip_solver* s);

/* One whole pass of the inner `while` body, src/sdplr.jl:190-278, repeated until one of the
 * reference's own exits fires.  This is THE hot path; the host only reads back O(1) scalars.
 *   in:  lagrangian / grad_norm as returned by the preceding fg!; cur_gtol; fprec_eps =
 *        config.fprec·eps(); max_local_iters (≥ 1; the reference's maxiter − iter budget, or a
 *        fixed count for benchmarking); time_budget_s (≤ 0 ⇒ none); use_armijo.
 *   out: updated lagrangian, grad_norm, primal_vio_norm, last step size, iterations done and
 *        exit_reason: 0 grad_norm ≤ cur_gtol (:190), 1 rel_delta < fprec·eps (:238-241),
 *        2 iteration budget (:272-277), 3 time budget (:272-277).                              */
int32_t sdplr_hip_inner_loop(sdplr_hip_solver* s, double normC, double normb,
                             int32_t gtol_relative, int32_t ptol_relative, int32_t use_armijo,
                             double cur_gtol, double fprec_eps, int64_t max_local_iters,
                             double time_budget_s, double* lagrangian, double* grad_norm,
                             double* primal_vio_norm, double* last_alpha, int64_t* iters_done,
                             int32_t* exit_reason);

/* The device work of ONE major iteration of _sdplr as one call (src/sdplr.jl:358-362 or :366-369, then :384, :389, then
 * the next pass of :190-278):  [update_lambda ≠ 0: λᵢ ← min(λ_ubᵢ, λᵢ − σ·primal_vio_rawᵢ) with the CURRENT σ]  →
 * var.σ[] = sigma  →  lbfgs_clear!  →  fg!  →  the inner while loop on fg!'s (ℒ, grad_norm) — exactly
 * sdplr_hip_update_lambda / set_scalar / lbfgs_clear / fg / inner_loop in that order, whose arguments these are.  On
 * small instances (the resident route) all five are ONE kernel launch; out parameters as sdplr_hip_inner_loop
 * (iterations 0, exit_reason 0 when fg!'s gradient norm is already ≤ cur_gtol).
 * update_lambda = SDPLR_MAJOR_RESUME: none of the prologue — the `while` CONTINUES on the state an earlier call left when
 * it ran out of its iteration budget (exit_reason 2): lagrangian / grad_norm / primal_vio_norm are in/out (on entry what
 * that call returned), sigma must be the σ of that call.  A driver that caps the iterations per call and resumes gets the
 * iterates of the uncapped call bit for bit (the lockstep batches do that so that a round does not last as long as its
 * slowest member).                                                                                                  */
#define SDPLR_MAJOR_RESUME 2
int32_t sdplr_hip_major_iteration(sdplr_hip_solver* s, double normC, double normb, int32_t gtol_relative,
                                  int32_t ptol_relative, int32_t use_armijo, int32_t update_lambda,
                                  double sigma, double cur_gtol, double fprec_eps, int64_t max_local_iters,
                                  double time_budget_s, double* lagrangian, double* grad_norm,
                                  double* primal_vio_norm, double* last_alpha, int64_t* iters_done,
                                  int32_t* exit_reason);

/* ---- dual bound / minimum eigenvalue -------------------------------------------------------- */
/* The Lanczos recurrence of approx_mineigval_lanczos, src/coreop.jl:461-500, on the S left by the
 * last 𝒜t_preprocess!: q is clipped to n−1 (:465); v0 (length n, HOST) replaces `randn` (:473) and
 * is normalised (:474).  alpha[q], beta[q] receive the raw coefficients, *steps the number of
 * steps taken (early exit :494-496).                                                           */
int32_t sdplr_hip_lanczos(sdplr_hip_solver* s, int64_t q, const double* v0, double* alpha,
                          double* beta, int64_t* steps);
/* Smallest eigenvalue of SymTridiagonal(alpha .+ 1, beta[1:k-1]) minus 1, src/coreop.jl:502-513.
 * Host-side; exact (Sturm bisection) where the reference calls GenericArpack.symeigs(tol=1e-4). */
int32_t sdplr_hip_tridiag_mineig(const double* alpha, const double* beta, int64_t k,
                                 double* mineig);
/* approx_mineigval_lanczos(var, aux, q)  src/coreop.jl:461-514 */
int32_t sdplr_hip_approx_mineigval_lanczos(sdplr_hip_solver* s, int64_t q, const double* v0,
                                           double* mineig);
/* dual_obj(data, var, aux, trace_bound, iter)  src/coreop.jl:376-415 (Lanczos branch;
 * q = 2⌈√max(iter,100)·ln n⌉ :402)                                                             */
int32_t sdplr_hip_dual_obj(sdplr_hip_solver* s, double trace_bound, int64_t iter,
                           const double* v0, double* dual_value, double* mineig);

/* ---- many small instances in lockstep: one launch for a whole batch ------------------------------------------------
 * The reference runs exps/batch_test.txt as independent sdplr() calls (exps/exp.jl:18-72).  A driver that advances B such
 * solves side by side hands the SAME step of all of them to the library as one call: the instances on the resident route
 * (one workgroup owns a small instance) that share a kernel shape (even ranks / odd ranks) go out as ONE launch with one
 * workgroup per instance
 * — B CUs busy from one stream, one argument table up, one result table back — and every other instance of the batch is
 * served by the single-instance entry point named in each struct (on a few host threads of the library's own, each handle
 * on its stream), so the call is total.  In sdplr_hip_batch_major_iteration the instances of the multi-launch EDGE path
 * (disjoint single-entry constraints: Lovász-θ) that share every launch dimension run their while loops (src/sdplr.jl:190-278)
 * behind one launch per kernel for the group, block row ↔ instance, after their own λ update / fg! (k_group.h).  Each item is exactly the
 * argument list of that entry point; `status` is what it would have returned for that instance (the function itself
 * returns the first non-zero status, or an argument error).  Results are bit-identical to the single-instance calls.
 * Handles of one call must be distinct and must not be used by other threads during the call.                        */
typedef struct sdplr_hip_fg_item {            /* sdplr_hip_fg */
  sdplr_hip_solver* s;
  double normC, normb;
  int32_t gtol_relative, ptol_relative;
  double lagrangian, grad_norm, primal_vio_norm, obj;   /* out (obj: var.obj[] as f! leaves it) */
  int32_t status;
} sdplr_hip_fg_item;
int32_t sdplr_hip_batch_fg(int32_t count, sdplr_hip_fg_item* items);

typedef struct sdplr_hip_major_item {         /* sdplr_hip_major_iteration */
  sdplr_hip_solver* s;
  double normC, normb;
  int32_t gtol_relative, ptol_relative, use_armijo, update_lambda;
  double sigma, cur_gtol, fprec_eps;
  int64_t max_local_iters;
  double time_budget_s;
  double lagrangian, grad_norm, primal_vio_norm, last_alpha, obj;   /* out */
  int64_t iters_done;
  int32_t exit_reason, status;
} sdplr_hip_major_item;
int32_t sdplr_hip_batch_major_iteration(int32_t count, sdplr_hip_major_item* items);

typedef struct sdplr_hip_dual_item {          /* sdplr_hip_dual_obj */
  sdplr_hip_solver* s;
  double trace_bound;
  int64_t iter;
  const double* v0;                           /* [n of that instance] */
  double* y_out;                              /* NULL, or [m+1]: var.y as dual_obj leaves it (−λ of the bound, src/sdplr.jl:325) */
  double dual_value, mineig;                  /* out */
  int32_t status;
} sdplr_hip_dual_item;
int32_t sdplr_hip_batch_dual_obj(int32_t count, sdplr_hip_dual_item* items);

/* ---- high-precision eigen path / DIMACS ----------------------------------------------------------------------
 * SDP_S_eigval(var, aux, nevs, preprocessed; which, ncv, tol, maxiter)  src/coreop.jl:351-374: the nev smallest
 * (which = 0, :SA) or largest (which = 1, :LA) eigenvalues of the S left by the last 𝒜t_preprocess!.  The reference
 * hands x ↦ S·x + x to GenericArpack.symeigs (implicitly restarted Lanczos) and subtracts the shift; here thick-restart
 * Lanczos with full re-orthogonalisation runs ON THE DEVICE: the ≤ ncv basis vectors stay in HBM, the operator is the
 * device SpMV of 𝒜t!(y, aux, x, var), and only the ncv×ncv projected matrix crosses PCIe, once per restart cycle.
 * Convergence test: ARPACK's, |β·y_last| ≤ tol·max(eps^⅔, |θ + 1|) on the shifted operator; tol ≤ 0 ⇒ machine
 * precision; ncv ≤ 0 ⇒ min(100, n) (:439); maxiter counts restart cycles.  v0 (HOST, length n) or NULL for a fixed
 * internal start vector.  evals[nev] ascending for :SA, descending for :LA; *n_converged < nev ⇒ the cycle budget ran
 * out (the values returned are the current Ritz values).                                                        */
int32_t sdplr_hip_S_eigval(sdplr_hip_solver* s, int64_t nev, int32_t which, int64_t ncv, double tol,
                           int64_t maxiter, const double* v0, double* evals, int64_t* n_matvec,
                           int64_t* n_converged);
/* dot(A, B) of two factor slots on the device: err6 of DIMACS_errors is dot(Rt, Rt·S) (src/coreop.jl:449) =
 * At_left(SDPLR_F_SCRATCH, SDPLR_F_RT) then factor_dot(SDPLR_F_RT, SDPLR_F_SCRATCH).                            */
int32_t sdplr_hip_factor_dot(sdplr_hip_solver* s, int32_t slot_a, int32_t slot_b, double* out);

/* ---- library counters ------------------------------------------------------------------------------
 * out[0] hipGraph captures that succeeded, out[1] captures that FAILED (this handle then launches eagerly for
 * good — visible here, never silent), out[2] captures skipped because the process-wide capture lock was busy
 * (that call ran eagerly; the next one tries again), out[3] inner-loop batches replayed from a graph,
 * out[4] inner-loop batches launched eagerly, out[5] Lanczos graph replays, out[6] Lanczos rounds launched
 * eagerly, out[7] inner iterations run, out[8] inner loops run as ONE resident launch (small instances: one
 * workgroup owns the instance for the whole loop), out[9] Lanczos runs as one resident launch, out[10] fg! calls as
 * one resident launch, out[11] of those launches (loops, Lanczos runs, fg!) the ones this instance shared with others
 * (sdplr_hip_batch_*), out[12] the inner loops (calls) that ran the step kernel WITHOUT P = A_g·R (the gradient carried
 * forward from G_old: cost matrix = the general sparse matrix, no low-rank term on the multi-launch route), out[13] the
 * inner loops this instance ran behind launches SHARED with other instances of a batch call on the multi-launch edge
 * path (one launch per kernel of the while body for the whole group, sdplr_hip_batch_major_iteration), out[14] the
 * inner loops (calls) that ran on the RING form of the L-BFGS history (lbfgshis.vecs[j].s / .y of src/lbfgs.jl:4-12
 * kept as (α_j, dir_j) and (G_j, G_{j+1}) inside the loop: the same values, two stores per iteration less), out[15]
 * the times that form was turned back into stored s_j, y_j because something outside the loop looked.  Writes
 * min(cap, 16) entries, *n_written says how many.                                                               */
int32_t sdplr_hip_get_stats(const sdplr_hip_solver* s, int64_t* out, int32_t cap, int32_t* n_written);

/* ---- per-kernel device timing (hipEvent pairs on the handle's stream) ------------------------ */
int32_t sdplr_hip_profile_enable(sdplr_hip_solver* s, int32_t on); /* also resets the counters */
/* time only launches of the kernel called `name` (NULL or "" ⇒ all); keeps the timed region undisturbed */
int32_t sdplr_hip_profile_filter(sdplr_hip_solver* s, const char* name);
int32_t sdplr_hip_profile_count(const sdplr_hip_solver* s, int32_t* n_entries);
int32_t sdplr_hip_profile_get(sdplr_hip_solver* s, int32_t idx, char* name, int32_t name_cap,
                              int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* SDPLR_HIP_H */
