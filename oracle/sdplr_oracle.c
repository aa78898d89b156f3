/*
 * sdplr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.  See sdplr_oracle.h for scope and pinning.
 *
 * Plain C99, single-threaded (see the note on the inert OpenMP pragmas at ddot), FP64, no BLAS: a CPU restatement of the reference's numeric hot path,
 * function by function, each citing the reference file:line it follows.  Deliberately written as
 * the obvious loops (the reference's own structure), not as an optimised implementation.
 */
#define _POSIX_C_SOURCE 200809L
#include "sdplr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OK 0
#define ERR_INVALID (-1)
#define ERR_NOT_DESCENT (-3)
#define ERR_STATE (-4)
#define ERR_ALLOC (-6)

#define F_RT 0
#define F_GT 1
#define F_DIRT 2
#define F_LBFGS_S 100
#define F_LBFGS_Y 200
#define F_SCRATCH 300

#define V_LAMBDA 0
#define V_LAMBDA_UB 1
#define V_B 2
#define V_Y 3
#define V_PV_RAW 4
#define V_PV_LB 5
#define V_PV 6
#define V_A_RD 7
#define V_A_DD 8
#define V_LBFGS_RHO 9
#define V_LBFGS_A 10
#define V_UVT 11
#define V_TRIU_S_NZVAL 12
#define V_S_NZVAL 13
#define V_SCRATCH 14

#define S_SIGMA 0
#define S_OBJ 1
#define S_LBFGS_LATEST 2

typedef struct {
  int64_t s;     /* columns of B */
  int64_t gid;   /* 0-based index into the (m+1)-vectors */
  double* B;     /* n×s column-major   (src/structs.jl:13) */
  double* D;     /* s                  (src/structs.jl:12) */
} lowrank_t;

struct sdplr_oracle_solver {
  int64_t n, m, r, h;
  int finalized;
  /* --- SolverAuxiliary, src/structs.jl:274-294 (all 0-based here) --- */
  int64_t n_sparse, nnzAgg, nnzT, nnzS;
  int64_t *matptr, *nzind, *gids;
  double *nzval_one, *nzval_two;
  int64_t *triu_colptr, *triu_rowval;
  double* triu_nzval;
  int64_t *colptr, *rowval, *mapped;
  double* nzval;
  double* UVt;
  int64_t n_lr;
  lowrank_t* lr;
  /* --- SolverVars, src/structs.jl:194-223 --- */
  double *Rt, *Gt, *dirt;
  double *lambda, *lambda_ub, *b, *y, *pv_raw, *pv_lb, *pv, *A_RD, *A_DD;
  double sigma, obj;
  /* --- LBFGSHistory, src/lbfgs.jl:21-28 --- */
  double **hs, **hy;
  double *rho, *a;
  int64_t latest; /* 1-based, as in the reference */
  char err[256];
  double* scratchF[2]; /* F_SCRATCH + k, allocated on first use */
  double* scratchV;    /* V_SCRATCH, m+1 */
};
typedef struct sdplr_oracle_solver S;

static char g_err[256] = "";

static int fail(S* s, int code, const char* msg) {
  if (s) {
    snprintf(s->err, sizeof s->err, "%s", msg);
  } else {
    snprintf(g_err, sizeof g_err, "%s", msg);
  }
  return code;
}

const char* sdplr_oracle_last_error(const S* s) { return s ? s->err : g_err; }
const char* sdplr_oracle_version(void) { return "sdplr_oracle 0.1 (CPU restatement, test infrastructure)"; }
int32_t sdplr_oracle_device_count(int32_t* c) {
  if (c) *c = 0;
  return OK;
}
int32_t sdplr_oracle_set_device(int32_t d) {
  (void)d;
  return OK;
}

static double* dalloc(int64_t k) { return (double*)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }
static int64_t* ialloc(int64_t k) { return (int64_t*)calloc((size_t)(k > 0 ? k : 1), sizeof(int64_t)); }

int32_t sdplr_oracle_create(int64_t n, int64_t m, int64_t r, int64_t h, S** out) {
  if (!out || n < 1 || m < 0 || r < 1 || h < 0) return fail(NULL, ERR_INVALID, "create: bad sizes");
  S* s = (S*)calloc(1, sizeof(S));
  if (!s) return fail(NULL, ERR_ALLOC, "create: alloc");
  s->n = n;
  s->m = m;
  s->r = r;
  s->h = h;
  s->sigma = 2.0; /* config.σ_0 default, src/options.jl:5 */
  *out = s;
  return OK;
}

int32_t sdplr_oracle_set_sparse(S* s, int64_t base, int64_t n_sparse, const int64_t* matptr,
                                const int64_t* nzind, const double* one, const double* two,
                                const int64_t* gids, int64_t nnzT, const int64_t* tcp,
                                const int64_t* trv, int64_t nnzS, const int64_t* fcp,
                                const int64_t* frv, const int64_t* mapped) {
  if (!s || s->finalized) return fail(s, ERR_STATE, "set_sparse: bad state");
  if (base != 0 && base != 1) return fail(s, ERR_INVALID, "set_sparse: index_base must be 0 or 1");
  if (n_sparse < 0 || nnzT < 0 || nnzS < 0) return fail(s, ERR_INVALID, "set_sparse: negative size");
  int64_t n = s->n;
  s->n_sparse = n_sparse;
  s->nnzT = nnzT;
  s->nnzS = nnzS;
  s->matptr = ialloc(n_sparse + 1);
  for (int64_t k = 0; k <= n_sparse; k++) s->matptr[k] = matptr[k] - base;
  s->nnzAgg = s->matptr[n_sparse];
  if (s->matptr[0] != 0 || s->nnzAgg < 0) return fail(s, ERR_INVALID, "set_sparse: bad matptr");
  s->nzind = ialloc(s->nnzAgg);
  s->nzval_one = dalloc(s->nnzAgg);
  s->nzval_two = dalloc(s->nnzAgg);
  for (int64_t e = 0; e < s->nnzAgg; e++) {
    s->nzind[e] = nzind[e] - base;
    if (s->nzind[e] < 0 || s->nzind[e] >= nnzT) return fail(s, ERR_INVALID, "set_sparse: nzind out of range");
    s->nzval_one[e] = one[e];
    s->nzval_two[e] = two[e];
  }
  s->gids = ialloc(n_sparse);
  for (int64_t k = 0; k < n_sparse; k++) {
    s->gids[k] = gids[k] - base;
    if (s->gids[k] < 0 || s->gids[k] > s->m) return fail(s, ERR_INVALID, "set_sparse: global index out of range");
  }
  s->triu_colptr = ialloc(n + 1);
  s->colptr = ialloc(n + 1);
  for (int64_t j = 0; j <= n; j++) {
    s->triu_colptr[j] = tcp[j] - base;
    s->colptr[j] = fcp[j] - base;
  }
  if (s->triu_colptr[n] != nnzT || s->colptr[n] != nnzS) return fail(s, ERR_INVALID, "set_sparse: colptr/nnz mismatch");
  s->triu_rowval = ialloc(nnzT);
  for (int64_t p = 0; p < nnzT; p++) {
    s->triu_rowval[p] = trv[p] - base;
    if (s->triu_rowval[p] < 0 || s->triu_rowval[p] >= n) return fail(s, ERR_INVALID, "set_sparse: triu rowval out of range");
  }
  s->rowval = ialloc(nnzS);
  s->mapped = ialloc(nnzS);
  for (int64_t p = 0; p < nnzS; p++) {
    s->rowval[p] = frv[p] - base;
    s->mapped[p] = mapped[p] - base;
    if (s->rowval[p] < 0 || s->rowval[p] >= n || s->mapped[p] < 0 || s->mapped[p] >= nnzT)
      return fail(s, ERR_INVALID, "set_sparse: full pattern out of range");
  }
  return OK;
}

int32_t sdplr_oracle_add_symlowrank(S* s, int64_t base, int64_t gid, int64_t sc, const double* B,
                                    const double* D) {
  if (!s || s->finalized) return fail(s, ERR_STATE, "add_symlowrank: bad state");
  if (sc < 1 || gid - base < 0 || gid - base > s->m) return fail(s, ERR_INVALID, "add_symlowrank: bad args");
  s->lr = (lowrank_t*)realloc(s->lr, (size_t)(s->n_lr + 1) * sizeof(lowrank_t));
  lowrank_t* L = &s->lr[s->n_lr++];
  L->s = sc;
  L->gid = gid - base;
  L->B = dalloc(s->n * sc);
  L->D = dalloc(sc);
  memcpy(L->B, B, (size_t)(s->n * sc) * sizeof(double));
  memcpy(L->D, D, (size_t)sc * sizeof(double));
  return OK;
}

static void free_factors(S* s) {
  free(s->Rt);
  free(s->Gt);
  free(s->dirt);
  for (int64_t j = 0; j < s->h; j++) {
    if (s->hs) free(s->hs[j]);
    if (s->hy) free(s->hy[j]);
  }
  free(s->hs);
  free(s->hy);
  free(s->scratchF[0]);
  free(s->scratchF[1]);
  s->scratchF[0] = s->scratchF[1] = NULL;
  s->Rt = s->Gt = s->dirt = NULL;
  s->hs = s->hy = NULL;
}

/* SolverVars(Rt0, λ0, λ_ub, r, σ_0) src/structs.jl:242-263 + lbfgs_init src/lbfgs.jl:35-47 */
static int alloc_factors(S* s) {
  int64_t N = s->n * s->r;
  s->Rt = dalloc(N);
  s->Gt = dalloc(N);
  s->dirt = dalloc(N);
  s->hs = (double**)calloc((size_t)(s->h > 0 ? s->h : 1), sizeof(double*));
  s->hy = (double**)calloc((size_t)(s->h > 0 ? s->h : 1), sizeof(double*));
  for (int64_t j = 0; j < s->h; j++) {
    s->hs[j] = dalloc(N);
    s->hy[j] = dalloc(N);
  }
  s->latest = s->h; /* src/lbfgs.jl:45 */
  return OK;
}

int32_t sdplr_oracle_finalize(S* s) {
  if (!s || s->finalized) return fail(s, ERR_STATE, "finalize: bad state");
  int64_t m = s->m;
  if (!s->matptr) { /* no sparse part: empty patterns */
    s->matptr = ialloc(1);
    s->triu_colptr = ialloc(s->n + 1);
    s->colptr = ialloc(s->n + 1);
  }
  s->triu_nzval = dalloc(s->nnzT);
  s->nzval = dalloc(s->nnzS);
  s->UVt = dalloc(s->nnzT);
  s->lambda = dalloc(m);
  s->lambda_ub = dalloc(m);
  s->b = dalloc(m);
  s->pv_lb = dalloc(m);
  s->pv = dalloc(m);
  for (int64_t i = 0; i < m; i++) { /* all-equality default, src/structs.jl:266-268, :247 */
    s->lambda_ub[i] = INFINITY;
    s->pv_lb[i] = -INFINITY;
  }
  s->y = dalloc(m + 1);
  s->pv_raw = dalloc(m + 1);
  s->A_RD = dalloc(m + 1);
  s->A_DD = dalloc(m + 1);
  s->rho = dalloc(s->h);
  s->a = dalloc(s->h);
  alloc_factors(s);
  s->finalized = 1;
  return OK;
}

int32_t sdplr_oracle_destroy(S* s) {
  if (!s) return OK;
  free_factors(s);
  free(s->matptr); free(s->nzind); free(s->gids); free(s->nzval_one); free(s->nzval_two);
  free(s->triu_colptr); free(s->triu_rowval); free(s->triu_nzval);
  free(s->colptr); free(s->rowval); free(s->mapped); free(s->nzval); free(s->UVt);
  for (int64_t t = 0; t < s->n_lr; t++) { free(s->lr[t].B); free(s->lr[t].D); }
  free(s->lr);
  free(s->lambda); free(s->lambda_ub); free(s->b); free(s->y); free(s->pv_raw); free(s->pv_lb);
  free(s->pv); free(s->A_RD); free(s->A_DD); free(s->rho); free(s->a); free(s->scratchV);
  free(s);
  return OK;
}

/* rank_update! src/coreop.jl:518-526 → SolverVars(data, newr, config) src/structs.jl:225-263 */
int32_t sdplr_oracle_reset_rank(S* s, int64_t new_r) {
  if (!s || !s->finalized) return fail(s, ERR_STATE, "reset_rank: not finalized");
  if (new_r < 1) return fail(s, ERR_INVALID, "reset_rank: bad rank");
  free_factors(s);
  s->r = new_r;
  alloc_factors(s);
  memset(s->lambda, 0, (size_t)s->m * sizeof(double));
  memset(s->pv, 0, (size_t)s->m * sizeof(double));
  memset(s->y, 0, (size_t)(s->m + 1) * sizeof(double));
  memset(s->pv_raw, 0, (size_t)(s->m + 1) * sizeof(double));
  memset(s->A_RD, 0, (size_t)(s->m + 1) * sizeof(double));
  memset(s->A_DD, 0, (size_t)(s->m + 1) * sizeof(double));
  memset(s->rho, 0, (size_t)s->h * sizeof(double));
  memset(s->a, 0, (size_t)s->h * sizeof(double));
  s->obj = 0.0;
  return OK;
}

/* ---- state access --------------------------------------------------------------------------- */
static double* factor_ptr(S* s, int32_t slot) {
  if (slot == F_RT) return s->Rt;
  if (slot == F_GT) return s->Gt;
  if (slot == F_DIRT) return s->dirt;
  if (slot >= F_LBFGS_S && slot < F_LBFGS_S + s->h) return s->hs[slot - F_LBFGS_S];
  if (slot >= F_LBFGS_Y && slot < F_LBFGS_Y + s->h) return s->hy[slot - F_LBFGS_Y];
  if (slot == F_SCRATCH || slot == F_SCRATCH + 1) {
    if (!s->scratchF[slot - F_SCRATCH]) s->scratchF[slot - F_SCRATCH] = dalloc(s->n * s->r);
    return s->scratchF[slot - F_SCRATCH];
  }
  return NULL;
}
static double* vec_ptr(S* s, int32_t which, int64_t* len) {
  int64_t m = s->m;
  switch (which) {
    case V_LAMBDA: *len = m; return s->lambda;
    case V_LAMBDA_UB: *len = m; return s->lambda_ub;
    case V_B: *len = m; return s->b;
    case V_Y: *len = m + 1; return s->y;
    case V_PV_RAW: *len = m + 1; return s->pv_raw;
    case V_PV_LB: *len = m; return s->pv_lb;
    case V_PV: *len = m; return s->pv;
    case V_A_RD: *len = m + 1; return s->A_RD;
    case V_A_DD: *len = m + 1; return s->A_DD;
    case V_LBFGS_RHO: *len = s->h; return s->rho;
    case V_LBFGS_A: *len = s->h; return s->a;
    case V_UVT: *len = s->nnzT; return s->UVt;
    case V_TRIU_S_NZVAL: *len = s->nnzT; return s->triu_nzval;
    case V_S_NZVAL: *len = s->nnzS; return s->nzval;
    case V_SCRATCH:
      *len = m + 1;
      if (!s->scratchV) s->scratchV = dalloc(m + 1);
      return s->scratchV;
    default: *len = 0; return NULL;
  }
}
#define NEED_FINAL(s) do { if (!(s) || !(s)->finalized) return fail((s), ERR_STATE, "not finalized"); } while (0)

int32_t sdplr_oracle_set_factor(S* s, int32_t slot, const double* h) {
  NEED_FINAL(s);
  double* p = factor_ptr(s, slot);
  if (!p || !h) return fail(s, ERR_INVALID, "set_factor: bad slot");
  memcpy(p, h, (size_t)(s->n * s->r) * sizeof(double));
  return OK;
}
int32_t sdplr_oracle_get_factor(S* s, int32_t slot, double* h) {
  NEED_FINAL(s);
  double* p = factor_ptr(s, slot);
  if (!p || !h) return fail(s, ERR_INVALID, "get_factor: bad slot");
  memcpy(h, p, (size_t)(s->n * s->r) * sizeof(double));
  return OK;
}
int32_t sdplr_oracle_set_vec(S* s, int32_t which, const double* h, int64_t len) {
  NEED_FINAL(s);
  int64_t L;
  double* p = vec_ptr(s, which, &L);
  if (!p || len != L || (!h && L > 0)) return fail(s, ERR_INVALID, "set_vec: bad slot/length");
  memcpy(p, h, (size_t)L * sizeof(double));
  return OK;
}
int32_t sdplr_oracle_get_vec(S* s, int32_t which, double* h, int64_t len) {
  NEED_FINAL(s);
  int64_t L;
  double* p = vec_ptr(s, which, &L);
  if (!p || len != L || (!h && L > 0)) return fail(s, ERR_INVALID, "get_vec: bad slot/length");
  memcpy(h, p, (size_t)L * sizeof(double));
  return OK;
}
int32_t sdplr_oracle_set_scalar(S* s, int32_t which, double v) {
  NEED_FINAL(s);
  if (which == S_SIGMA) s->sigma = v;
  else if (which == S_OBJ) s->obj = v;
  else if (which == S_LBFGS_LATEST) s->latest = (int64_t)v;
  else return fail(s, ERR_INVALID, "set_scalar: bad slot");
  return OK;
}
int32_t sdplr_oracle_get_scalar(S* s, int32_t which, double* v) {
  NEED_FINAL(s);
  if (!v) return fail(s, ERR_INVALID, "get_scalar: null");
  if (which == S_SIGMA) *v = s->sigma;
  else if (which == S_OBJ) *v = s->obj;
  else if (which == S_LBFGS_LATEST) *v = (double)s->latest;
  else return fail(s, ERR_INVALID, "get_scalar: bad slot");
  return OK;
}
int32_t sdplr_oracle_get_dims(const S* s, int64_t* n, int64_t* m, int64_t* r, int64_t* h,
                              int64_t* nnzT, int64_t* nnzS, int64_t* nnzAgg) {
  if (!s) return ERR_INVALID;
  if (n) *n = s->n;
  if (m) *m = s->m;
  if (r) *r = s->r;
  if (h) *h = s->h;
  if (nnzT) *nnzT = s->nnzT;
  if (nnzS) *nnzS = s->nnzS;
  if (nnzAgg) *nnzAgg = s->nnzAgg;
  return OK;
}

/* ---- BLAS-1 as plain loops ------------------------------------------------------------------ */
/* The `#pragma omp` lines below are inert in the checker build (libsdplr_oracle.so is compiled WITHOUT
 * -fopenmp: one thread, the reference's own protocol, exps/test.jl:46).  They only take effect in the
 * separate all-cores timing build (make omp → libsdplr_oracle_omp.so) that bench.py's cpu_baseline
 * reports beside the one-thread number; tests/test_oracle_omp.py checks that build
 * against this one. */
static double ddot(int64_t N, const double* x, const double* y) {
#ifdef _OPENMP
  /* all-cores timing build only: every thread sums one contiguous range and the ranges' sums are added in thread order,
   * so that the result does not depend on which thread finishes first (an OpenMP `reduction` combines in arrival order:
   * two runs of the same solve took different numbers of major iterations on MinBisection n = 1e5) */
  if (N > 65536) {
    enum { MAXT = 512 };
    double part[MAXT];
    int used = 1;
#pragma omp parallel
    {
      const int t = omp_get_thread_num();
      int T = omp_get_num_threads();
      if (T > MAXT) T = MAXT;
      if (t < T) {
        const int64_t lo = N * t / T, hi = N * (t + 1) / T;
        double a = 0.0;
        for (int64_t i = lo; i < hi; i++) a += x[i] * y[i];
        part[t] = a;
      }
      if (t == 0) used = T;
    }
    double acc = 0.0;
    for (int t = 0; t < used; t++) acc += part[t];
    return acc;
  }
#endif
  double acc = 0.0;
  for (int64_t i = 0; i < N; i++) acc += x[i] * y[i];
  return acc;
}
static void daxpy(int64_t N, double a, const double* x, double* y) {
#pragma omp parallel for schedule(static) if (N > 65536)
  for (int64_t i = 0; i < N; i++) y[i] += a * x[i];
}
static void dscal(int64_t N, double a, double* x) {
#pragma omp parallel for schedule(static) if (N > 65536)
  for (int64_t i = 0; i < N; i++) x[i] *= a;
}
/* copyto! of a factor-shaped array (a memcpy in the one-thread build) */
static void dcopy(int64_t N, const double* x, double* y) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (N > 65536)
  for (int64_t i = 0; i < N; i++) y[i] = x[i];
#else
  memcpy(y, x, (size_t)N * sizeof(double));
#endif
}
static double dnrm2(int64_t N, const double* x) { return sqrt(ddot(N, x, x)); }

/* ---- 𝒜 ------------------------------------------------------------------------------------- */

/* mydot(Rt, row, col)  src/coreop.jl:153-160 */
static double mydot1(const double* Ut, int64_t r, int64_t row, int64_t col) {
  double rval = 0.0;
  const double *a = Ut + row * r, *b = Ut + col * r;
  for (int64_t i = 0; i < r; i++) rval += a[i] * b[i];
  return rval;
}
/* mydot(Ut, Vt, row, col)  src/coreop.jl:162-172 */
static double mydot2(const double* Ut, const double* Vt, int64_t r, int64_t row, int64_t col) {
  double rval = 0.0;
  for (int64_t i = 0; i < r; i++) rval += Ut[row * r + i] * Vt[col * r + i];
  for (int64_t i = 0; i < r; i++) rval += Vt[row * r + i] * Ut[col * r + i];
  return rval / 2;
}

/* 𝒜_sparse_formUUt!  src/coreop.jl:174-186 */
static void A_sparse_formUUt(S* s, const double* Ut) {
  memset(s->UVt, 0, (size_t)s->nnzT * sizeof(double));
#pragma omp parallel for schedule(dynamic, 512)
  for (int64_t col = 0; col < s->n; col++)
    for (int64_t nzi = s->triu_colptr[col]; nzi < s->triu_colptr[col + 1]; nzi++) {
      int64_t row = s->triu_rowval[nzi];
      s->UVt[nzi] = mydot1(Ut, s->r, col, row);
    }
}
/* 𝒜_sparse_formUVt!  src/coreop.jl:188-203 */
static void A_sparse_formUVt(S* s, const double* Ut, const double* Vt) {
  memset(s->UVt, 0, (size_t)s->nnzT * sizeof(double));
#pragma omp parallel for schedule(dynamic, 512)
  for (int64_t col = 0; col < s->n; col++)
    for (int64_t nzi = s->triu_colptr[col]; nzi < s->triu_colptr[col + 1]; nzi++) {
      int64_t row = s->triu_rowval[nzi];
      s->UVt[nzi] = mydot2(Ut, Vt, s->r, col, row);
    }
}
/* the `UUt' * SparseMatrixCSC(nnzT, n_sparse, matptr, nzind, nzval_two)` product and scatter,
 * src/coreop.jl:80-90 / :102-112 */
static void A_sparse_reduce(S* s, double* out) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t k = 0; k < s->n_sparse; k++) {
    double v = 0.0;
    for (int64_t e = s->matptr[k]; e < s->matptr[k + 1]; e++) v += s->UVt[s->nzind[e]] * s->nzval_two[e];
    out[s->gids[k]] = v;
  }
}
/* tr_UtAU src/coreop.jl:115-120 and tr_UtAV :122-130.  Ut*A.B is r×s. */
static double tr_UtAV(S* s, const lowrank_t* L, const double* Ut, const double* Vt) {
  int64_t n = s->n, r = s->r;
  double total = 0.0;
  double* UtB = dalloc(r);
  double* VtB = dalloc(r);
  for (int64_t t = 0; t < L->s; t++) {
    const double* Bt = L->B + t * n;
    for (int64_t c = 0; c < r; c++) UtB[c] = VtB[c] = 0.0;
    for (int64_t i = 0; i < n; i++)
      for (int64_t c = 0; c < r; c++) {
        UtB[c] += Ut[i * r + c] * Bt[i];
        VtB[c] += Vt[i * r + c] * Bt[i];
      }
    for (int64_t c = 0; c < r; c++) total += UtB[c] * VtB[c] * L->D[t];
  }
  free(UtB);
  free(VtB);
  return total;
}
/* 𝒜! one- and two-argument, src/coreop.jl:36-70 */
static void A_op(S* s, double* out, const double* Ut, const double* Vt) {
  memset(out, 0, (size_t)(s->m + 1) * sizeof(double));
  if (s->n_sparse > 0) {
    if (Vt) A_sparse_formUVt(s, Ut, Vt);
    else A_sparse_formUUt(s, Ut);
    A_sparse_reduce(s, out);
  }
  /* 𝒜_symlowrank! src/coreop.jl:132-151 */
  for (int64_t t = 0; t < s->n_lr; t++) out[s->lr[t].gid] = tr_UtAV(s, &s->lr[t], Ut, Vt ? Vt : Ut);
}

int32_t sdplr_oracle_A(S* s, int32_t u_slot, int32_t v_slot, int32_t out_vec) {
  NEED_FINAL(s);
  double* U = factor_ptr(s, u_slot);
  double* V = v_slot >= 0 ? factor_ptr(s, v_slot) : NULL;
  if (!U || (v_slot >= 0 && !V)) return fail(s, ERR_INVALID, "A: bad factor slot");
  int64_t olen;
  double* out = out_vec == V_PV_RAW ? s->pv_raw : out_vec == V_A_RD ? s->A_RD : out_vec == V_A_DD ? s->A_DD :
                out_vec == V_SCRATCH ? vec_ptr(s, V_SCRATCH, &olen) : NULL;
  if (!out) return fail(s, ERR_INVALID, "A: bad output vector");
  A_op(s, out, U, V);
  return OK;
}

/* ---- 𝒜ᵀ ------------------------------------------------------------------------------------- */

/* 𝒜t_preprocess! src/coreop.jl:248-258 and 𝒜t_preprocess_sparse! :205-227 */
static void At_preprocess(S* s) {
  if (s->n_sparse <= 0) return;
  memset(s->triu_nzval, 0, (size_t)s->nnzT * sizeof(double));
  /* mul!(triu_nzval, CSC(nnzT × n_sparse; matptr, nzind, nzval_one), v),  v = y[global ids] */
  for (int64_t k = 0; k < s->n_sparse; k++) {
    double v = s->y[s->gids[k]];
    for (int64_t e = s->matptr[k]; e < s->matptr[k + 1]; e++) s->triu_nzval[s->nzind[e]] += s->nzval_one[e] * v;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < s->nnzS; i++) s->nzval[i] = s->triu_nzval[s->mapped[i]];
}
int32_t sdplr_oracle_At_preprocess(S* s) {
  NEED_FINAL(s);
  At_preprocess(s);
  return OK;
}

/* 𝒜t!(y, x, aux, var) src/coreop.jl:260-279 with mul!(Y, X, A, α, β) src/structs.jl:135-145 */
static void At_left(S* s, double* y, const double* x) {
  int64_t n = s->n, r = s->r;
  memset(y, 0, (size_t)(n * r) * sizeof(double));
  if (s->n_sparse > 0)
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t j = 0; j < n; j++)
      for (int64_t p = s->colptr[j]; p < s->colptr[j + 1]; p++) {
        double v = s->nzval[p];
        const double* xi = x + s->rowval[p] * r;
        for (int64_t c = 0; c < r; c++) y[j * r + c] += xi[c] * v;
      }
  for (int64_t t = 0; t < s->n_lr; t++) {
    const lowrank_t* L = &s->lr[t];
    double coeff = s->y[L->gid];
    double* XB = dalloc(r);
    for (int64_t q = 0; q < L->s; q++) {
      const double* Bq = L->B + q * n;
      for (int64_t c = 0; c < r; c++) XB[c] = 0.0;
      for (int64_t i = 0; i < n; i++)
        for (int64_t c = 0; c < r; c++) XB[c] += x[i * r + c] * Bq[i];
      for (int64_t c = 0; c < r; c++) XB[c] *= L->D[q];
      for (int64_t j = 0; j < n; j++)
        for (int64_t c = 0; c < r; c++) y[j * r + c] += coeff * XB[c] * Bq[j];
    }
    free(XB);
  }
}
int32_t sdplr_oracle_At_left(S* s, int32_t ys, int32_t xs) {
  NEED_FINAL(s);
  double *y = factor_ptr(s, ys), *x = factor_ptr(s, xs);
  if (!y || !x || y == x) return fail(s, ERR_INVALID, "At_left: bad slots");
  At_left(s, y, x);
  return OK;
}

/* 𝒜t!(y, aux, x, var) src/coreop.jl:281-300 with mul!(Y, A, X, α, β) src/structs.jl:117-127;
 * x, y are n×k column-major */
static void At_right(S* s, double* y, const double* x, int64_t k) {
  int64_t n = s->n;
  memset(y, 0, (size_t)(n * k) * sizeof(double));
  if (s->n_sparse > 0)
    for (int64_t c = 0; c < k; c++)
      for (int64_t j = 0; j < n; j++) {
        double xj = x[c * n + j];
        for (int64_t p = s->colptr[j]; p < s->colptr[j + 1]; p++) y[c * n + s->rowval[p]] += s->nzval[p] * xj;
      }
  for (int64_t t = 0; t < s->n_lr; t++) {
    const lowrank_t* L = &s->lr[t];
    double coeff = s->y[L->gid];
    for (int64_t c = 0; c < k; c++)
      for (int64_t q = 0; q < L->s; q++) {
        const double* Bq = L->B + q * n;
        double btx = 0.0;
        for (int64_t i = 0; i < n; i++) btx += Bq[i] * x[c * n + i];
        btx *= L->D[q];
        for (int64_t i = 0; i < n; i++) y[c * n + i] += coeff * Bq[i] * btx;
      }
  }
}
int32_t sdplr_oracle_At_right(S* s, const double* x, double* y, int64_t k) {
  NEED_FINAL(s);
  if (!x || !y || k < 1) return fail(s, ERR_INVALID, "At_right: bad args");
  At_right(s, y, x, k);
  return OK;
}
int32_t sdplr_oracle_At_right_device(S* s, const double* x, double* y, int64_t k) {
  return sdplr_oracle_At_right(s, x, y, k); /* the oracle's "device" memory is host memory */
}
/* SDP_S_eigval src/coreop.jl:351-374.  The reference's eigensolver (GenericArpack.symeigs) is third-party code
 * outside the tree: PARITY UNPINNED there.  The checker computes what that solver converges to: the exact spectrum
 * of the dense S (𝒜t! right applied to the identity, then cyclic Jacobi) — small n only. */
int32_t sdplr_oracle_S_eigval(S* s, int64_t nev, int32_t which, int64_t ncv, double tol, int64_t maxiter,
                              const double* v0, double* evals, int64_t* n_matvec, int64_t* n_converged) {
  NEED_FINAL(s);
  (void)ncv; (void)tol; (void)maxiter; (void)v0;
  int64_t n = s->n;
  if (!evals || nev < 1 || nev > n) return fail(s, ERR_INVALID, "S_eigval: bad args");
  if (n > 1500) return fail(s, ERR_INVALID, "S_eigval: the dense checker handles n <= 1500");
  double* A = dalloc(n * n);
  double* e = dalloc(n);
  for (int64_t j = 0; j < n; j++) {
    memset(e, 0, (size_t)n * sizeof(double));
    e[j] = 1.0;
    At_right(s, A + j * n, e, 1);
  }
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = i + 1; j < n; j++) A[i * n + j] = A[j * n + i] = 0.5 * (A[i * n + j] + A[j * n + i]);
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0, tot = 0.0;
    for (int64_t i = 0; i < n; i++)
      for (int64_t j = 0; j < n; j++) { tot += A[i * n + j] * A[i * n + j]; if (i != j) off += A[i * n + j] * A[i * n + j]; }
    if (off <= 1e-30 * tot) break;
    for (int64_t p = 0; p < n - 1; p++)
      for (int64_t q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int64_t k = 0; k < n; k++) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - sn * akq;
          A[k * n + q] = sn * akp + c * akq;
        }
        for (int64_t k = 0; k < n; k++) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - sn * aqk;
          A[q * n + k] = sn * apk + c * aqk;
        }
      }
  }
  for (int64_t i = 0; i < n; i++) e[i] = A[i * n + i];
  for (int64_t i = 1; i < n; i++) { /* insertion sort, ascending */
    double v = e[i];
    int64_t j = i - 1;
    while (j >= 0 && e[j] > v) { e[j + 1] = e[j]; j--; }
    e[j + 1] = v;
  }
  for (int64_t i = 0; i < nev; i++) evals[i] = which == 0 ? e[i] : e[n - 1 - i];
  if (n_matvec) *n_matvec = n;
  if (n_converged) *n_converged = nev;
  free(A);
  free(e);
  return OK;
}
/* dot(A, B) of two factor slots (err6 = dot(Rt, Rt·S), src/coreop.jl:449) */
int32_t sdplr_oracle_factor_dot(S* s, int32_t slot_a, int32_t slot_b, double* out) {
  NEED_FINAL(s);
  double *a = factor_ptr(s, slot_a), *b = factor_ptr(s, slot_b);
  if (!a || !b || !out) return fail(s, ERR_INVALID, "factor_dot: bad args");
  *out = ddot(s->n * s->r, a, b);
  return OK;
}
int32_t sdplr_oracle_get_stats(const S* s, int64_t* out, int32_t cap, int32_t* n_written) {
  if (!s || !out || cap < 0) return ERR_INVALID;
  int32_t k = cap < 8 ? cap : 8;
  for (int32_t i = 0; i < k; i++) out[i] = 0;
  if (n_written) *n_written = k;
  return OK;
}

/* ---- f!, g!, fg! ---------------------------------------------------------------------------- */

/* f! src/coreop.jl:11-31 */
static double f_op(S* s) {
  int64_t m = s->m;
  A_op(s, s->pv_raw, s->Rt, NULL);
  s->obj = s->pv_raw[m];
  for (int64_t i = 0; i < m; i++) s->pv_raw[i] -= s->b[i];
  for (int64_t i = 0; i < m; i++) s->pv[i] = fmax(s->pv_raw[i], s->pv_lb[i]);
  double sigma = s->sigma, L = s->obj;
  for (int64_t i = 0; i < m; i++) {
    double yi = fmin(s->lambda_ub[i], s->lambda[i] - sigma * s->pv_raw[i]);
    L += (yi * yi - s->lambda[i] * s->lambda[i]) / (2 * sigma);
  }
  return L;
}
/* copy2y_λ_sub_pvio! src/coreop.jl:229-236 */
static void copy2y_lambda_sub_pvio(S* s) {
  int64_t m = s->m;
  for (int64_t i = 0; i < m; i++) s->y[i] = -fmin(s->lambda_ub[i], s->lambda[i] - s->sigma * s->pv_raw[i]);
  s->y[m] = 1.0;
}
/* g! src/coreop.jl:305-317 */
static void g_op(S* s) {
  copy2y_lambda_sub_pvio(s);
  At_preprocess(s);
  At_left(s, s->Gt, s->Rt);
  dscal(s->n * s->r, 2.0, s->Gt);
}
/* norms as in fg! src/coreop.jl:334-347 and _sdplr src/sdplr.jl:224-234 */
static void norms_op(S* s, double normC, double normb, int grel, int prel, double* gn, double* pn) {
  double g = dnrm2(s->n * s->r, s->Gt);
  double p = dnrm2(s->m, s->pv);
  *gn = grel ? g / normC : g;
  *pn = prel ? p / normb : p;
}
int32_t sdplr_oracle_f(S* s, double* L) {
  NEED_FINAL(s);
  double v = f_op(s);
  if (L) *L = v;
  return OK;
}
int32_t sdplr_oracle_g(S* s) {
  NEED_FINAL(s);
  g_op(s);
  return OK;
}
int32_t sdplr_oracle_fg(S* s, double normC, double normb, int32_t grel, int32_t prel, double* L,
                        double* gn, double* pn) {
  NEED_FINAL(s);
  double v = f_op(s);
  g_op(s);
  for (int64_t i = 0; i < s->m; i++) s->pv[i] = fmax(s->pv_raw[i], s->pv_lb[i]); /* :340-342 */
  double g, p;
  norms_op(s, normC, normb, grel, prel, &g, &p);
  if (L) *L = v;
  if (gn) *gn = g;
  if (pn) *pn = p;
  return OK;
}
int32_t sdplr_oracle_norms(S* s, double normC, double normb, int32_t grel, int32_t prel,
                           double* gn, double* pn) {
  NEED_FINAL(s);
  double g, p;
  norms_op(s, normC, normb, grel, prel, &g, &p);
  if (gn) *gn = g;
  if (pn) *pn = p;
  return OK;
}
/* src/sdplr.jl:219 */
int32_t sdplr_oracle_axpy_R(S* s, double alpha) {
  NEED_FINAL(s);
  daxpy(s->n * s->r, alpha, s->dirt, s->Rt);
  return OK;
}
/* src/sdplr.jl:358-362 */
int32_t sdplr_oracle_update_lambda(S* s) {
  NEED_FINAL(s);
  for (int64_t i = 0; i < s->m; i++) s->lambda[i] = fmin(s->lambda_ub[i], s->lambda[i] - s->sigma * s->pv_raw[i]);
  return OK;
}

/* ---- L-BFGS --------------------------------------------------------------------------------- */

/* lbfgs_clear! src/lbfgs.jl:52-59 */
int32_t sdplr_oracle_lbfgs_clear(S* s) {
  NEED_FINAL(s);
  int64_t N = s->n * s->r;
  for (int64_t i = 0; i < s->h; i++) {
    memset(s->hs[i], 0, (size_t)N * sizeof(double));
    memset(s->hy[i], 0, (size_t)N * sizeof(double));
    s->rho[i] = 0.0;
    s->a[i] = 0.0;
  }
  return OK;
}
/* lbfgs_dir! src/lbfgs.jl:77-124 (indices j are 1-based as in the reference) */
static void lbfgs_dir(S* s, int negate) {
  int64_t N = s->n * s->r, m = s->h, lst = s->latest;
  double *dir = s->dirt, *grad = s->Gt;
  dcopy(N, grad, dir);
  if (m == 0) return;
  int64_t j = lst;
  for (int64_t it = 0; it < m; it++) {
    double alpha = s->rho[j - 1] * ddot(N, s->hs[j - 1], dir);
    daxpy(N, -alpha, s->hy[j - 1], dir);
    s->a[j - 1] = alpha;
    j -= 1;
    if (j == 0) j = m;
  }
  j = lst % m + 1;
  for (int64_t it = 0; it < m; it++) {
    double beta = s->rho[j - 1] * ddot(N, s->hy[j - 1], dir);
    double gamma = s->a[j - 1] - beta;
    daxpy(N, gamma, s->hs[j - 1], dir);
    j += 1;
    if (j == m + 1) j = 1;
  }
  if (negate) dscal(N, -1.0, dir);
  j = s->latest % m + 1;
  dcopy(N, grad, s->hy[j - 1]);
  dscal(N, -1.0, s->hy[j - 1]);
}
int32_t sdplr_oracle_lbfgs_dir(S* s, int32_t negate, double* descent) {
  NEED_FINAL(s);
  lbfgs_dir(s, negate);
  if (descent) *descent = ddot(s->n * s->r, s->dirt, s->Gt); /* src/sdplr.jl:201 */
  return OK;
}
/* src/sdplr.jl:203-204 */
int32_t sdplr_oracle_descent_fallback(S* s) {
  NEED_FINAL(s);
  int64_t N = s->n * s->r;
  dscal(N, -1.0, s->Gt);
  dcopy(N, s->Gt, s->dirt);
  return OK;
}
/* lbfgs_update! src/lbfgs.jl:129-149 */
static void lbfgs_update(S* s, double stepsize) {
  if (s->h == 0) return;
  int64_t N = s->n * s->r;
  int64_t j = s->latest % s->h + 1;
  dscal(N, stepsize, s->dirt);
  dcopy(N, s->dirt, s->hs[j - 1]);
  daxpy(N, 1.0, s->Gt, s->hy[j - 1]);
  s->rho[j - 1] = 1 / ddot(N, s->hy[j - 1], s->hs[j - 1]);
  s->latest = j;
}
int32_t sdplr_oracle_lbfgs_update(S* s, double stepsize) {
  NEED_FINAL(s);
  lbfgs_update(s, stepsize);
  return OK;
}

/* ---- line search ---------------------------------------------------------------------------- */

static double horner(const double* c, int deg, double x) {
  double v = c[deg];
  for (int i = deg - 1; i >= 0; i--) v = v * x + c[i];
  return v;
}
/* real roots in [lo, hi] of the polynomial c[0] + c[1]x + … + c[deg]x^deg, deg ≤ 3, found by
 * splitting [lo, hi] at the stationary points of the polynomial and bisecting every sign change.
 * Stands in for PolynomialRoots.roots (src/linesearch.jl:82,94) — PARITY UNPINNED, see header.  */
static int bracketed_roots(const double* c, int deg, double lo, double hi, double* out) {
  while (deg > 0 && c[deg] == 0.0) deg--;
  if (deg == 0) return 0;
  double brk[4];
  int nb = 0;
  brk[nb++] = lo;
  if (deg == 3) { /* stationary points: c1 + 2c2 x + 3c3 x² = 0 */
    double A = 3 * c[3], B = 2 * c[2], C = c[1];
    double disc = B * B - 4 * A * C;
    if (disc > 0) {
      double sq = sqrt(disc);
      double qq = -0.5 * (B + (B >= 0 ? sq : -sq));
      double t1 = qq / A, t2 = (qq != 0.0) ? C / qq : t1;
      if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
      if (t1 > lo && t1 < hi) brk[nb++] = t1;
      if (t2 > lo && t2 < hi && t2 != t1) brk[nb++] = t2;
    }
  } else if (deg == 2) {
    double t = -c[1] / (2 * c[2]);
    if (t > lo && t < hi) brk[nb++] = t;
  }
  brk[nb++] = hi;
  int nr = 0;
  for (int i = 0; i + 1 < nb; i++) {
    double a = brk[i], b = brk[i + 1];
    double fa = horner(c, deg, a), fb = horner(c, deg, b);
    if (fa == 0.0) {
      if (nr == 0 || out[nr - 1] != a) out[nr++] = a;
      continue;
    }
    if (fb == 0.0) {
      if (i + 2 == nb) out[nr++] = b; /* interior zero endpoints are picked up as the next `a` */
      continue;
    }
    if ((fa < 0) == (fb < 0)) continue;
    for (int it = 0; it < 400; it++) {
      double mid = 0.5 * (a + b);
      if (mid <= a || mid >= b) break;
      double fm = horner(c, deg, mid);
      if (fm == 0.0) { a = b = mid; break; }
      if ((fm < 0) == (fa < 0)) { a = mid; fa = fm; } else { b = mid; fb = fm; }
    }
    out[nr++] = (a == b) ? a : (fabs(fa) <= fabs(fb) ? a : b);
  }
  return nr;
}

/* the scalar stage of linesearch!, src/linesearch.jl:58-112, given the quartic's coefficients */
static int quartic_argmin(const double* bq, double alpha_max, double* alpha, double* fval) {
  double cubic[4];
  cubic[0] = 1.0 * bq[1];
  if (cubic[0] > DBL_EPSILON) return ERR_NOT_DESCENT; /* :60-62 */
  cubic[1] = 2.0 * bq[2];
  cubic[2] = 3.0 * bq[3];
  cubic[3] = 4.0 * bq[4];
  double roots[4];
  int nr;
  if (fabs(cubic[3]) < DBL_EPSILON) nr = bracketed_roots(cubic, 2, 0.0, alpha_max, roots); /* :70-83 */
  else nr = bracketed_roots(cubic, 3, 0.0, alpha_max, roots);                              /* :86-95 */
  roots[nr++] = alpha_max;
  double a_star = 0.0, f_star = bq[0]; /* :78-80 / :90-92 */
  for (int i = 0; i < nr; i++) {       /* :98-112 */
    double root = roots[i];
    if (root < 0 || root > alpha_max) continue;
    double fa = horner(bq, 4, root);
    if (fa < f_star) { f_star = fa; a_star = root; }
  }
  *alpha = a_star;
  *fval = f_star;
  return OK;
}
int32_t sdplr_oracle_quartic_argmin(const double* bq, double alpha_max, double* alpha, double* fval) {
  if (!bq || !alpha || !fval) return ERR_INVALID;
  return quartic_argmin(bq, alpha_max, alpha, fval);
}

/* the common head of both line searches, src/linesearch.jl:8-18 / :145-153 */
static void linesearch_head(S* s) {
  A_op(s, s->A_RD, s->Rt, s->dirt);
  for (int64_t i = 0; i <= s->m; i++) s->A_RD[i] *= 2.0;
  A_op(s, s->A_DD, s->dirt, s->dirt);
}
/* the common tail, src/linesearch.jl:118-124 / :184-188 */
static void linesearch_commit(S* s, double a) {
  for (int64_t i = 0; i <= s->m; i++) s->pv_raw[i] += a * (a * s->A_DD[i] + s->A_RD[i]);
  s->obj = s->pv_raw[s->m];
  for (int64_t i = 0; i < s->m; i++) s->pv[i] = fmax(s->pv_raw[i], s->pv_lb[i]);
}
/* linesearch! src/linesearch.jl:4-127 */
static int linesearch(S* s, double alpha_max, double* alpha, double* Lval) {
  int64_t m = s->m;
  linesearch_head(s);
  double p0 = s->obj, p1 = s->A_RD[m], p2 = s->A_DD[m], sigma = s->sigma;
  const double *nq0 = s->pv_raw, *q1 = s->A_RD, *q2 = s->A_DD, *lam = s->lambda;
  double bq[5];
  bq[0] = p0 - ddot(m, lam, nq0) + sigma * ddot(m, nq0, nq0) / 2;
  bq[1] = p1 - ddot(m, lam, q1) + sigma * ddot(m, nq0, q1);
  double d = 0.0; /* dot(λ − σ·neg_q0, q2) :52 */
  for (int64_t i = 0; i < m; i++) d += (lam[i] - sigma * nq0[i]) * q2[i];
  bq[2] = p2 - d + sigma * ddot(m, q1, q1) / 2;
  bq[3] = sigma * ddot(m, q1, q2);
  bq[4] = sigma * ddot(m, q2, q2) / 2;
  double a, f;
  int rc = quartic_argmin(bq, alpha_max, &a, &f);
  if (rc != OK) return rc;
  linesearch_commit(s, a);
  *alpha = a;
  *Lval = f;
  return OK;
}
int32_t sdplr_oracle_linesearch(S* s, double alpha_max, double* alpha, double* L) {
  NEED_FINAL(s);
  double a = 0, f = 0;
  int rc = linesearch(s, alpha_max, &a, &f);
  if (rc == ERR_NOT_DESCENT) return fail(s, rc, "Error: cubic[1] should be less than 0.");
  if (alpha) *alpha = a;
  if (L) *L = f;
  return rc;
}
/* eval_AL, src/linesearch.jl:157-165 */
static double armijo_eval(S* s, double a) {
  int64_t m = s->m;
  double L = s->obj + a * s->A_RD[m] + a * a * s->A_DD[m];
  for (int64_t i = 0; i < m; i++) {
    double gi = s->pv_raw[i] + a * s->A_RD[i] + a * a * s->A_DD[i];
    double lt = fmin(s->lambda_ub[i], s->lambda[i] - s->sigma * gi);
    L += (lt * lt - s->lambda[i] * s->lambda[i]) / (2 * s->sigma);
  }
  return L;
}
/* linesearch_armijo! src/linesearch.jl:139-191 */
static int linesearch_armijo(S* s, double alpha_max, double* alpha, double* Lval) {
  int64_t m = s->m;
  linesearch_head(s);
  double L0 = armijo_eval(s, 0.0);
  double slope = s->A_RD[m] + ddot(m, s->y, s->A_RD); /* :171 */
  double c = 1e-4, a = alpha_max;
  double La = armijo_eval(s, a);
  for (int it = 0; it < 50; it++) {
    if (La <= L0 + c * a * slope) break;
    a /= 2;
    La = armijo_eval(s, a);
  }
  linesearch_commit(s, a);
  *alpha = a;
  *Lval = La;
  return OK;
}
int32_t sdplr_oracle_linesearch_armijo(S* s, double alpha_max, double* alpha, double* L) {
  NEED_FINAL(s);
  double a = 0, f = 0;
  int rc = linesearch_armijo(s, alpha_max, &a, &f);
  if (alpha) *alpha = a;
  if (L) *L = f;
  return rc;
}

/* ---- the inner while loop, src/sdplr.jl:190-278 ---------------------------------------------- */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
int32_t sdplr_oracle_inner_loop(S* s, double normC, double normb, int32_t grel, int32_t prel,
                                int32_t use_armijo, double cur_gtol, double fprec_eps,
                                int64_t max_local_iters, double time_budget_s, double* Lio,
                                double* gnio, double* pnio, double* last_alpha, int64_t* iters,
                                int32_t* exit_reason) {
  NEED_FINAL(s);
  if (!Lio || !gnio || !pnio || max_local_iters < 1) return fail(s, ERR_INVALID, "inner_loop: bad args");
  double L = *Lio, gn = *gnio, pn = *pnio, alpha = 0.0;
  int64_t N = s->n * s->r, localiter = 0;
  int32_t why = 0;
  double t0 = now_s();
  while (gn > cur_gtol) { /* :190 */
    localiter++;
    lbfgs_dir(s, 1);                                  /* :197 */
    double descent = ddot(N, s->dirt, s->Gt);         /* :201 */
    if (isnan(descent) || descent >= 0) {             /* :202-205 */
      dscal(N, -1.0, s->Gt);
      dcopy(N, s->Gt, s->dirt);
    }
    double lastval = L;                               /* :207 */
    int rc = use_armijo ? linesearch_armijo(s, 1.0, &alpha, &L) : linesearch(s, 1.0, &alpha, &L);
    if (rc != OK) return fail(s, rc, "Error: cubic[1] should be less than 0.");
    daxpy(N, alpha, s->dirt, s->Rt);                  /* :219 */
    g_op(s);                                          /* :221 */
    norms_op(s, normC, normb, grel, prel, &gn, &pn);  /* :224-234 */
    double rel_delta = (lastval - L) / fmax(1.0, fmax(fabs(L), fabs(lastval))); /* :238 */
    if (rel_delta < fprec_eps) { why = 1; break; }    /* :239-241 */
    if (s->h > 0) lbfgs_update(s, alpha);             /* :244-246 */
    if (localiter >= max_local_iters) { why = 2; break; }                        /* :272-277 */
    if (time_budget_s > 0 && now_s() - t0 > time_budget_s) { why = 3; break; }   /* :272-277 */
  }
  *Lio = L;
  *gnio = gn;
  *pnio = pn;
  if (last_alpha) *last_alpha = alpha;
  if (iters) *iters = localiter;
  if (exit_reason) *exit_reason = why;
  return OK;
}

/* ---- Lanczos / dual bound ------------------------------------------------------------------- */

/* major_iteration of the shared ABI: the sequence the name stands for (src/sdplr.jl:358-369, :384, :389, :190-278) */
int32_t sdplr_oracle_major_iteration(S* s, double normC, double normb, int32_t grel, int32_t prel, int32_t use_armijo,
                                     int32_t update_lambda, double sigma, double cur_gtol, double fprec_eps,
                                     int64_t max_local_iters, double time_budget_s, double* L, double* gn, double* pn,
                                     double* last_alpha, int64_t* iters, int32_t* exit_reason) {
  NEED_FINAL(s);
  if (!L || !gn || !pn || max_local_iters < 1) return fail(s, ERR_INVALID, "major_iteration: bad args");
  if (update_lambda == 2) /* SDPLR_MAJOR_RESUME: the while loop continues where a capped call left it (L, gn, pn in/out) */
    return sdplr_oracle_inner_loop(s, normC, normb, grel, prel, use_armijo, cur_gtol, fprec_eps, max_local_iters,
                                   time_budget_s, L, gn, pn, last_alpha, iters, exit_reason);
  int32_t rc;
  if (update_lambda && (rc = sdplr_oracle_update_lambda(s))) return rc;
  if ((rc = sdplr_oracle_set_scalar(s, 0, sigma))) return rc;
  if ((rc = sdplr_oracle_lbfgs_clear(s))) return rc;
  if ((rc = sdplr_oracle_fg(s, normC, normb, grel, prel, L, gn, pn))) return rc;
  if (last_alpha) *last_alpha = 0.0;
  if (iters) *iters = 0;
  if (exit_reason) *exit_reason = 0;
  if (!(*gn > cur_gtol)) return OK; /* :190 */
  return sdplr_oracle_inner_loop(s, normC, normb, grel, prel, use_armijo, cur_gtol, fprec_eps, max_local_iters,
                                 time_budget_s, L, gn, pn, last_alpha, iters, exit_reason);
}

/* the recurrence of approx_mineigval_lanczos, src/coreop.jl:461-500 */
int32_t sdplr_oracle_lanczos(S* s, int64_t q, const double* v0, double* alpha, double* beta,
                             int64_t* steps) {
  NEED_FINAL(s);
  int64_t n = s->n;
  if (!v0 || !alpha || !beta || !steps || q < 1) return fail(s, ERR_INVALID, "lanczos: bad args");
  if (q > n - 1) q = n - 1; /* :465 */
  double *v = dalloc(n), *Av = dalloc(n), *vpre = dalloc(n);
  memcpy(v, v0, (size_t)n * sizeof(double));
  double nv = dnrm2(n, v);
  for (int64_t i = 0; i < n; i++) v[i] /= nv; /* :474 */
  int64_t iter = 0;
  for (int64_t i = 0; i < q; i++) {
    iter++;
    At_right(s, Av, v, 1);              /* :483 */
    alpha[i] = ddot(n, v, Av);          /* :484 */
    if (i == 0) for (int64_t t = 0; t < n; t++) Av[t] -= alpha[i] * v[t];                            /* :487 */
    else for (int64_t t = 0; t < n; t++) Av[t] -= alpha[i] * v[t] + beta[i - 1] * vpre[t];            /* :489 */
    beta[i] = dnrm2(n, Av);             /* :492 */
    if (fabs(beta[i]) < sqrt((double)n) * DBL_EPSILON) break; /* :494-496 */
    for (int64_t t = 0; t < n; t++) Av[t] /= beta[i];
    memcpy(vpre, v, (size_t)n * sizeof(double));
    memcpy(v, Av, (size_t)n * sizeof(double));
  }
  *steps = iter;
  free(v);
  free(Av);
  free(vpre);
  return OK;
}

/* number of eigenvalues of SymTridiagonal(d, e) strictly below x (Sturm sequence) */
static int64_t sturm_count(const double* d, const double* e, int64_t k, double x) {
  int64_t cnt = 0;
  double q = d[0] - x;
  if (q < 0) cnt++;
  for (int64_t i = 1; i < k; i++) {
    double den = (q == 0.0) ? DBL_MIN : q;
    q = d[i] - x - e[i - 1] * e[i - 1] / den;
    if (q < 0) cnt++;
  }
  return cnt;
}
/* src/coreop.jl:502-513: min eigenvalue of SymTridiagonal(alpha .+ 1, beta[1:k-1]) − 1.
 * The reference calls GenericArpack.symeigs(…; tol=1e-4) — PARITY UNPINNED; this is exact. */
int32_t sdplr_oracle_tridiag_mineig(const double* alpha, const double* beta, int64_t k, double* out) {
  if (!alpha || !out || k < 1 || (k > 1 && !beta)) return ERR_INVALID;
  if (k == 1) { /* :505-507 */
    *out = (alpha[0] + 1) - 1;
    return OK;
  }
  double* d = dalloc(k);
  for (int64_t i = 0; i < k; i++) d[i] = alpha[i] + 1;
  double lo = INFINITY, hi = -INFINITY; /* Gershgorin */
  for (int64_t i = 0; i < k; i++) {
    double rad = (i > 0 ? fabs(beta[i - 1]) : 0.0) + (i + 1 < k ? fabs(beta[i]) : 0.0);
    lo = fmin(lo, d[i] - rad);
    hi = fmax(hi, d[i] + rad);
  }
  for (int it = 0; it < 200; it++) {
    double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_count(d, beta, k, mid) >= 1) hi = mid; else lo = mid;
  }
  *out = 0.5 * (lo + hi) - 1;
  free(d);
  return OK;
}
int32_t sdplr_oracle_approx_mineigval_lanczos(S* s, int64_t q, const double* v0, double* mineig) {
  NEED_FINAL(s);
  if (q < 1) return fail(s, ERR_INVALID, "lanczos: q < 1");
  double *al = dalloc(q), *be = dalloc(q);
  int64_t steps = 0;
  int rc = sdplr_oracle_lanczos(s, q, v0, al, be, &steps);
  if (rc == OK) rc = sdplr_oracle_tridiag_mineig(al, be, steps, mineig);
  free(al);
  free(be);
  return rc;
}
/* dual_obj, src/coreop.jl:376-415 (Lanczos branch) */
int32_t sdplr_oracle_dual_obj(S* s, double trace_bound, int64_t iter, const double* v0,
                              double* dual_value, double* mineig) {
  NEED_FINAL(s);
  copy2y_lambda_sub_pvio(s);
  At_preprocess(s);
  double it = (double)(iter > 100 ? iter : 100);
  int64_t eig_iter = (int64_t)(2 * ceil(pow(it, 0.5) * log((double)s->n))); /* :402 */
  double ev = 0.0;
  int rc = sdplr_oracle_approx_mineigval_lanczos(s, eig_iter, v0, &ev);
  if (rc != OK) return rc;
  double dv = -ddot(s->m, s->y, s->b) + trace_bound * fmin(ev, 0.0); /* :412 */
  if (dual_value) *dual_value = dv;
  if (mineig) *mineig = ev;
  return OK;
}

int32_t sdplr_oracle_profile_enable(S* s, int32_t on) { (void)s; (void)on; return OK; }
int32_t sdplr_oracle_profile_filter(S* s, const char* name) { (void)s; (void)name; return OK; }
int32_t sdplr_oracle_device_synchronize(void) { return OK; }
int32_t sdplr_oracle_warmup(int32_t n_handles) { (void)n_handles; return OK; }
int32_t sdplr_oracle_trim_pools(void) { return OK; }   /* (no pools on the CPU side: same ABI, nothing to do) */
int32_t sdplr_oracle_profile_count(const S* s, int32_t* n) { (void)s; if (n) *n = 0; return OK; }
int32_t sdplr_oracle_profile_get(S* s, int32_t idx, char* name, int32_t cap, int64_t* launches, double* ms) {
  (void)s; (void)idx; (void)name; (void)cap; (void)launches; (void)ms;
  return ERR_INVALID;
}

/* ---- preprocess_sparsecons, src/preprocess.jl:24-169 ----------------------------------------- */
typedef struct { int64_t i, j; } pair_t;
static int pair_cmp(const void* a, const void* b) {
  const pair_t *x = (const pair_t*)a, *y = (const pair_t*)b;
  if (x->j != y->j) return x->j < y->j ? -1 : 1;
  if (x->i != y->i) return x->i < y->i ? -1 : 1;
  return 0;
}
/* `sparse(I, J, ones, n, n)` pattern: CSC, rows sorted within a column, duplicates merged */
static int64_t build_csc(pair_t* p, int64_t cnt, int64_t n, int64_t** colptr, int64_t** rowval) {
  qsort(p, (size_t)cnt, sizeof(pair_t), pair_cmp);
  int64_t u = 0;
  for (int64_t k = 0; k < cnt; k++)
    if (k == 0 || p[k].i != p[u - 1].i || p[k].j != p[u - 1].j) p[u++] = p[k];
  *colptr = ialloc(n + 1);
  *rowval = ialloc(u);
  for (int64_t k = 0; k < u; k++) {
    (*colptr)[p[k].j + 1]++;
    (*rowval)[k] = p[k].i;
  }
  for (int64_t j = 0; j < n; j++) (*colptr)[j + 1] += (*colptr)[j];
  return u;
}
/* binary search for `row` in column `col` of a CSC pattern, src/preprocess.jl:112-122, :147-157 */
static int64_t csc_find(const int64_t* colptr, const int64_t* rowval, int64_t col, int64_t row) {
  int64_t low = colptr[col], high = colptr[col + 1] - 1;
  while (low <= high) {
    int64_t mid = (low + high) / 2;
    if (rowval[mid] == row) return mid;
    if (rowval[mid] < row) low = mid + 1; else high = mid - 1;
  }
  return -1;
}
int32_t sdplr_oracle_preprocess(int64_t n, int64_t nA, int64_t base, const int64_t* ent_ptr,
                                const int64_t* I, const int64_t* J, const double* V,
                                sdplr_oracle_layout* out) {
  if (!out || n < 1 || nA < 0 || !ent_ptr) return ERR_INVALID;
  memset(out, 0, sizeof *out);
  int64_t total = ent_ptr[nA], total_triu = 0;
  for (int64_t e = 0; e < total; e++) {
    int64_t i = I[e] - base, j = J[e] - base;
    if (i < 0 || i >= n || j < 0 || j >= n) return ERR_INVALID;
    if (i <= j) total_triu++; /* triu: i <= j, src/preprocess.jl:9 */
  }
  pair_t* all = (pair_t*)malloc((size_t)(total > 0 ? total : 1) * sizeof(pair_t));
  pair_t* tri = (pair_t*)malloc((size_t)(total_triu > 0 ? total_triu : 1) * sizeof(pair_t));
  int64_t ct = 0;
  for (int64_t e = 0; e < total; e++) { /* :63-79 */
    all[e].i = I[e] - base;
    all[e].j = J[e] - base;
    if (all[e].i <= all[e].j) tri[ct++] = all[e];
  }
  out->n = n;
  out->nA = nA;
  out->nnzT = build_csc(tri, total_triu, n, &out->triu_colptr, &out->triu_rowval); /* :87 */
  out->nnzS = build_csc(all, total, n, &out->full_colptr, &out->full_rowval);      /* :90 */
  free(all);
  free(tri);
  out->nnzAgg = total_triu;
  out->matptr = ialloc(nA + 1);
  out->nzind = ialloc(total_triu);
  out->nzval_one = dalloc(total_triu);
  out->nzval_two = dalloc(total_triu);
  int64_t cumul = 0;
  for (int64_t k = 0; k < nA; k++) { /* :97-131 */
    out->matptr[k] = cumul;
    for (int64_t e = ent_ptr[k]; e < ent_ptr[k + 1]; e++) {
      int64_t row = I[e] - base, col = J[e] - base;
      if (row > col) continue;
      int64_t pos = csc_find(out->triu_colptr, out->triu_rowval, col, row);
      out->nzind[cumul] = pos;
      out->nzval_one[cumul] = V[e];
      out->nzval_two[cumul] = (row == col) ? V[e] : 2.0 * V[e]; /* :121-128 */
      cumul++;
    }
  }
  out->matptr[nA] = total_triu; /* :132 */
  out->mappedto_triu = ialloc(out->nnzS);
  for (int64_t col = 0; col < n; col++) /* :135-156 */
    for (int64_t nzi = out->full_colptr[col]; nzi < out->full_colptr[col + 1]; nzi++) {
      int64_t row = out->full_rowval[nzi];
      int64_t rr = row < col ? row : col, cc = row < col ? col : row;
      out->mappedto_triu[nzi] = csc_find(out->triu_colptr, out->triu_rowval, cc, rr);
    }
  return OK;
}
void sdplr_oracle_layout_free(sdplr_oracle_layout* l) {
  if (!l) return;
  free(l->triu_colptr); free(l->triu_rowval); free(l->full_colptr); free(l->full_rowval);
  free(l->matptr); free(l->nzind); free(l->nzval_one); free(l->nzval_two); free(l->mappedto_triu);
  memset(l, 0, sizeof *l);
}

/* set_sparse_coo of the shared ABI: preprocess_sparsecons (the literal restatement above) + set_sparse */
int32_t sdplr_oracle_set_sparse_coo(S* s, int64_t base, int64_t n_sparse, const int64_t* ent_ptr, const int64_t* I,
                                    const int64_t* J, const double* V, const int64_t* gids) {
  if (!s || s->finalized) return fail(s, ERR_STATE, "set_sparse_coo: bad state");
  if (base != 0 && base != 1) return fail(s, ERR_INVALID, "set_sparse_coo: index_base must be 0 or 1");
  int64_t* ptr0 = ialloc(n_sparse + 1);
  for (int64_t k = 0; k <= n_sparse; k++) ptr0[k] = ent_ptr[k] - base;
  sdplr_oracle_layout L;
  int32_t rc = sdplr_oracle_preprocess(s->n, n_sparse, base, ptr0, I, J, V, &L);
  free(ptr0);
  if (rc != OK) return fail(s, rc, "set_sparse_coo: constraint entry outside the n×n matrix");
  for (int64_t p = 0; p < L.nnzS; p++)
    if (L.mappedto_triu[p] < 0) {
      sdplr_oracle_layout_free(&L);
      return fail(s, ERR_INVALID, "set_sparse_coo: a constraint matrix is not symmetric: a lower-triangular entry has no "
                                  "upper-triangular mirror in the aggregated pattern");
    }
  int64_t* g0 = ialloc(n_sparse);
  for (int64_t k = 0; k < n_sparse; k++) g0[k] = gids[k] - base;
  rc = sdplr_oracle_set_sparse(s, 0, n_sparse, L.matptr, L.nzind, L.nzval_one, L.nzval_two, g0, L.nnzT, L.triu_colptr,
                               L.triu_rowval, L.nnzS, L.full_colptr, L.full_rowval, L.mappedto_triu);
  free(g0);
  sdplr_oracle_layout_free(&L);
  return rc;
}
int32_t sdplr_oracle_get_layout(const S* s, int32_t which, int64_t* out_i, double* out_f, int64_t cap, int64_t* len) {
  if (!s) return ERR_INVALID;
  if (s->finalized || !s->matptr) return ERR_STATE;
  const int64_t* vi = NULL;
  const double* vf = NULL;
  int64_t L = 0;
  switch (which) {
    case 0: vi = s->matptr; L = s->n_sparse + 1; break;
    case 1: vi = s->nzind; L = s->nnzAgg; break;
    case 2: vi = s->gids; L = s->n_sparse; break;
    case 3: vi = s->triu_colptr; L = s->n + 1; break;
    case 4: vi = s->triu_rowval; L = s->nnzT; break;
    case 5: vi = s->colptr; L = s->n + 1; break;
    case 6: vi = s->rowval; L = s->nnzS; break;
    case 7: vi = s->mapped; L = s->nnzS; break;
    case 8: vf = s->nzval_one; L = s->nnzAgg; break;
    case 9: vf = s->nzval_two; L = s->nnzAgg; break;
    default: return ERR_INVALID;
  }
  if (len) *len = L;
  for (int64_t k = 0; k < (L < cap ? L : cap); k++) {
    if (vi && out_i) out_i[k] = vi[k];
    if (vf && out_f) out_f[k] = vf[k];
  }
  return OK;
}

/* norm(A::SymLowRankMatrix, p) src/structs.jl:61-82 */
int32_t sdplr_oracle_symlowrank_norm(int64_t n, int64_t sc, const double* B, const double* D,
                                     int32_t p_is_inf, double* out) {
  if (!B || !D || !out || n < 1 || sc < 1) return ERR_INVALID;
  double res = 0.0;
  double* tmpv = dalloc(n);
  for (int64_t i = 0; i < n; i++) {
    /* tmpv = (B·D) · Bt[:, i] */
    for (int64_t k = 0; k < n; k++) tmpv[k] = 0.0;
    for (int64_t t = 0; t < sc; t++) {
      double w = D[t] * B[t * n + i];
      for (int64_t k = 0; k < n; k++) tmpv[k] += B[t * n + k] * w;
    }
    if (p_is_inf) {
      for (int64_t k = 0; k < n; k++) res = fmax(res, fabs(tmpv[k]));
    } else {
      double nn = dnrm2(n, tmpv);
      res += nn * nn;
    }
  }
  free(tmpv);
  *out = p_is_inf ? res : sqrt(res);
  return OK;
}

/* ---- the lockstep batch calls of the shared ABI: loops over the single-instance functions ------------------------- */
int32_t sdplr_oracle_batch_fg(int32_t count, sdplr_oracle_fg_item* it) {
  if (count < 0 || (count > 0 && !it)) return ERR_INVALID;
  int32_t first = OK;
  for (int32_t i = 0; i < count; i++) {
    sdplr_oracle_fg_item* q = &it[i];
    q->status = sdplr_oracle_fg(q->s, q->normC, q->normb, q->gtol_relative, q->ptol_relative, &q->lagrangian, &q->grad_norm,
                                &q->primal_vio_norm);
    if (!q->status) q->status = sdplr_oracle_get_scalar(q->s, S_OBJ, &q->obj);
    if (q->status && !first) first = q->status;
  }
  return first;
}
int32_t sdplr_oracle_batch_major_iteration(int32_t count, sdplr_oracle_major_item* it) {
  if (count < 0 || (count > 0 && !it)) return ERR_INVALID;
  int32_t first = OK;
  for (int32_t i = 0; i < count; i++) {
    sdplr_oracle_major_item* q = &it[i];
    q->status = sdplr_oracle_major_iteration(q->s, q->normC, q->normb, q->gtol_relative, q->ptol_relative, q->use_armijo,
                                             q->update_lambda, q->sigma, q->cur_gtol, q->fprec_eps, q->max_local_iters,
                                             q->time_budget_s, &q->lagrangian, &q->grad_norm, &q->primal_vio_norm,
                                             &q->last_alpha, &q->iters_done, &q->exit_reason);
    if (!q->status) q->status = sdplr_oracle_get_scalar(q->s, S_OBJ, &q->obj);
    if (q->status && !first) first = q->status;
  }
  return first;
}
int32_t sdplr_oracle_batch_dual_obj(int32_t count, sdplr_oracle_dual_item* it) {
  if (count < 0 || (count > 0 && !it)) return ERR_INVALID;
  int32_t first = OK;
  for (int32_t i = 0; i < count; i++) {
    sdplr_oracle_dual_item* q = &it[i];
    q->status = sdplr_oracle_dual_obj(q->s, q->trace_bound, q->iter, q->v0, &q->dual_value, &q->mineig);
    if (!q->status && q->y_out) q->status = sdplr_oracle_get_vec(q->s, V_Y, q->y_out, q->s->m + 1);
    if (q->status && !first) first = q->status;
  }
  return first;
}
