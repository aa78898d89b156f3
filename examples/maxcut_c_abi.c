/* A MaxCut SDP solved through the C ABI alone — no Python, no torch: what a host written in any language does with
 * include/sdplr_hip.h.  The graph is the circulant C_n(1,2,3) (every vertex joined to its three neighbours on either
 * side), the problem the reference's test/problem.jl:16-30 builds (C = −¼·(Diag(d) − A), A_i = e_i e_iᵀ, b_i = 1);
 * the aggregated layout is the one preprocess_sparsecons (src/preprocess.jl:24-169) produces, written out by hand
 * here because every A_i has one entry and C carries the whole pattern.  The driver below is the skeleton of
 * _sdplr (src/sdplr.jl:140-449): fg!, then major iterations of [inner loop → λ update → dual bound], with the
 * reference's adaptive schedule for σ and the tolerances left out for brevity (σ doubles every round, eight rounds).
 *
 *   gcc -O2 -Iinclude examples/maxcut_c_abi.c -Lsdplrplus.jl_amd/lib -lsdplr_hip -lm -o maxcut_c_abi
 *   LD_LIBRARY_PATH=sdplrplus.jl_amd/lib ./maxcut_c_abi [n] [r]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sdplr_hip.h"

#define CK(call)                                                                                     \
  do {                                                                                               \
    int32_t rc__ = (call);                                                                           \
    if (rc__ != SDPLR_OK) {                                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc__, sdplr_hip_last_error(s));                  \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)

static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096, r = argc > 2 ? atoll(argv[2]) : 8, m = n, h = 4;
  sdplr_hip_solver* s = NULL;
  if (n < 8) return 2;

  /* ---- the full pattern of C (CSC, 0-based): column j holds j and j ± 1, 2, 3 (mod n), rows ascending ---- */
  const int64_t nnzS = 7 * n;
  int64_t* fcp = malloc((n + 1) * sizeof *fcp);
  int64_t* frv = malloc(nnzS * sizeof *frv);
  for (int64_t j = 0; j < n; j++) {
    fcp[j] = 7 * j;
    for (int d = -3; d <= 3; d++) frv[7 * j + d + 3] = ((j + d) % n + n) % n;
    qsort(frv + 7 * j, 7, sizeof *frv, cmp_i64);
  }
  fcp[n] = nnzS;
  /* ---- its upper triangle, and for every full entry the position of (min, max) in it ---- */
  int64_t* tcp = malloc((n + 1) * sizeof *tcp);
  int64_t* trv = malloc(nnzS * sizeof *trv);
  int64_t nnzT = 0;
  for (int64_t j = 0; j < n; j++) {
    tcp[j] = nnzT;
    for (int64_t p = fcp[j]; p < fcp[j + 1]; p++)
      if (frv[p] <= j) trv[nnzT++] = frv[p];
  }
  tcp[n] = nnzT;
  int64_t* mapped = malloc(nnzS * sizeof *mapped);
  for (int64_t j = 0; j < n; j++)
    for (int64_t p = fcp[j]; p < fcp[j + 1]; p++) {
      const int64_t i = frv[p], lo = i < j ? i : j, hi = i < j ? j : i;
      int64_t q = tcp[hi];
      while (trv[q] != lo) q++;
      mapped[p] = q;
    }
  /* ---- per-matrix segments: A_0 … A_{n−1} (one diagonal entry each), then C (global index m) ---- */
  const int64_t n_sparse = n + 1, nnzAgg = n + nnzT;
  int64_t* matptr = malloc((n_sparse + 1) * sizeof *matptr);
  int64_t* nzind = malloc(nnzAgg * sizeof *nzind);
  int64_t* gids = malloc(n_sparse * sizeof *gids);
  double* one = malloc(nnzAgg * sizeof *one);
  double* two = malloc(nnzAgg * sizeof *two);
  for (int64_t k = 0; k < n; k++) {
    matptr[k] = k;
    gids[k] = k;
    nzind[k] = tcp[k + 1] - 1; /* the diagonal is the last entry of column k's upper part */
    one[k] = two[k] = 1.0;
  }
  matptr[n] = n;
  gids[n] = m;
  for (int64_t j = 0, e = n; j < n; j++)
    for (int64_t q = tcp[j]; q < tcp[j + 1]; q++, e++) {
      const double v = (trv[q] == j) ? -0.25 * 6.0 : 0.25; /* −¼(d_j) on the diagonal, +¼ per edge */
      nzind[e] = q;
      one[e] = v;
      two[e] = (trv[q] == j) ? v : 2.0 * v; /* off-diagonal entries count twice, src/preprocess.jl:121-128 */
    }
  matptr[n + 1] = nnzAgg;

  int32_t ndev = 0;
  (void)sdplr_hip_device_count(&ndev);
  printf("%s, %d device(s); MaxCut on C_%lld(1,2,3), r = %lld\n", sdplr_hip_version(), (int)ndev, (long long)n, (long long)r);
  CK(sdplr_hip_create(n, m, r, h, &s));
  CK(sdplr_hip_set_sparse(s, 0, n_sparse, matptr, nzind, one, two, gids, nnzT, tcp, trv, nnzS, fcp, frv, mapped));
  CK(sdplr_hip_finalize(s));

  /* ---- SolverVars: R0 ~ U(−1, 1) (src/structs.jl:236), λ0 = 0, σ0 = 2, b = 1 ---- */
  double* R = malloc((size_t)(n * r) * sizeof *R);
  double* b = malloc((size_t)m * sizeof *b);
  double* v0 = malloc((size_t)n * sizeof *v0);
  uint64_t x = 88172645463325252ull;
  for (int64_t i = 0; i < n * r; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    R[i] = 2.0 * (double)(x >> 11) / 9007199254740992.0 - 1.0;
  }
  for (int64_t i = 0; i < m; i++) b[i] = 1.0;
  for (int64_t i = 0; i < n; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    v0[i] = (double)(x >> 11) / 9007199254740992.0 - 0.5;
  }
  CK(sdplr_hip_set_factor(s, SDPLR_F_RT, R));
  CK(sdplr_hip_set_vec(s, SDPLR_V_B, b, m));
  CK(sdplr_hip_set_scalar(s, SDPLR_S_SIGMA, 2.0));
  double normC = 0.0;
  for (int64_t e = n; e < nnzAgg; e++) normC += (one[e] == two[e] ? 1.0 : 2.0) * one[e] * one[e];
  normC = sqrt(normC);
  const double normb = sqrt((double)m);

  double L, gnorm, pvnorm, alpha, obj, dual = 0.0, mineig = 0.0;
  int64_t iters, total = 0;
  int32_t reason;
  CK(sdplr_hip_fg(s, normC, normb, 1, 1, &L, &gnorm, &pvnorm));
  printf("start:  L = %.6e  |grad| = %.3e  |pv| = %.3e\n", L, gnorm, pvnorm);
  double sigma = 2.0;
  for (int major = 1; major <= 8; major++) {
    CK(sdplr_hip_inner_loop(s, normC, normb, 1, 1, 0, 1e-3 / major, 1e8 * 2.220446049250313e-16, 2000, 0.0, &L, &gnorm,
                            &pvnorm, &alpha, &iters, &reason));
    total += iters;
    CK(sdplr_hip_update_lambda(s));          /* λ ← λ − σ·primal_vio, src/sdplr.jl:358-362 */
    CK(sdplr_hip_lbfgs_clear(s));            /* src/sdplr.jl:384 */
    sigma *= 2.0;                            /* (the reference raises σ when the violation stalls, :364-376) */
    CK(sdplr_hip_set_scalar(s, SDPLR_S_SIGMA, sigma));
    CK(sdplr_hip_fg(s, normC, normb, 1, 1, &L, &gnorm, &pvnorm));
    CK(sdplr_hip_dual_obj(s, (double)n, total, v0, &dual, &mineig));
    CK(sdplr_hip_get_scalar(s, SDPLR_S_OBJ, &obj));
    printf("major %d: %5lld inner iterations (exit %d)  obj = %.6f  dual bound = %.6f  |pv| = %.2e  λ_min(S) = %.2e\n",
           major, (long long)iters, (int)reason, obj, dual, pvnorm, mineig);
  }
  CK(sdplr_hip_get_factor(s, SDPLR_F_RT, R));
  double worst = 0.0; /* diag(RRᵀ) = 1 */
  for (int64_t i = 0; i < n; i++) {
    double t = 0.0;
    for (int64_t k = 0; k < r; k++) t += R[i * r + k] * R[i * r + k];
    if (fabs(t - 1.0) > worst) worst = fabs(t - 1.0);
  }
  const int ok = worst < 1e-2 && dual <= obj + 1e-6 * fabs(obj) && (obj - dual) <= 2e-2 * fabs(obj);
  printf("max |‖R_i‖² − 1| = %.2e, gap = %.3e  ->  %s\n", worst, (obj - dual) / fabs(obj), ok ? "OK" : "CHECK FAILED");
  CK(sdplr_hip_destroy(s));
  return ok ? 0 : 3;
}
