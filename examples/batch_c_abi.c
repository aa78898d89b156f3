/* A batch of MaxCut SDPs solved SIDE BY SIDE through the C ABI alone — the lockstep calls of include/sdplr_hip.h
 * (sdplr_hip_batch_major_iteration / _batch_dual_obj / _batch_fg): every step of all B solves is ONE library call, and on
 * small instances one kernel launch with a workgroup per instance.  The reference runs such a batch as one process per
 * graph (exps/batch_test.txt under GNU parallel, exps/README.md:17-21).
 *
 * Instance k is the circulant graph C_{n_k}(1,…,d_k) (each vertex joined to its d_k neighbours on either side), n_k and d_k
 * differing from instance to instance; the problem is the one of the reference's test/problem.jl:16-30 (C = −¼·L,
 * A_i = e_i e_iᵀ, b_i = 1) and goes in as COO triplets (sdplr_hip_set_sparse_coo runs preprocess_sparsecons,
 * src/preprocess.jl:24-169, inside the library).  The driver is the skeleton of _sdplr (src/sdplr.jl:140-449) with the
 * adaptive schedule left out for brevity: eight rounds of [λ update → σ doubles → lbfgs_clear! → fg! → inner loop] — one
 * sdplr_hip_major_iteration per instance, sent as one batch — each followed by one batch of dual bounds.
 *
 *   gcc -O2 -Iinclude examples/batch_c_abi.c -Lsdplrplus.jl_amd/lib -lsdplr_hip -lm -o batch_c_abi
 *   LD_LIBRARY_PATH=sdplrplus.jl_amd/lib ./batch_c_abi [B] [r]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sdplr_hip.h"

#define CK(call, handle)                                                                             \
  do {                                                                                               \
    int32_t rc__ = (call);                                                                           \
    if (rc__ != SDPLR_OK) {                                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc__, sdplr_hip_last_error(handle));             \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double uniform01(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 12;
  const int64_t r = argc > 2 ? atoll(argv[2]) : 6, h = 4;
  if (B < 1 || B > 4096 || r < 1) return 2;
  sdplr_hip_solver** s = calloc((size_t)B, sizeof *s);
  int64_t* n = malloc((size_t)B * sizeof *n);
  double* normC = malloc((size_t)B * sizeof *normC);
  double** v0 = malloc((size_t)B * sizeof *v0);
  int32_t ndev = 0;
  (void)sdplr_hip_device_count(&ndev);
  printf("%s, %d device(s); %d MaxCut instances side by side, r = %lld\n", sdplr_hip_version(), (int)ndev, B, (long long)r);

  /* ---- set-up, instance by instance: the matrices as COO triplets (both triangles, as findnz returns them) ---- */
  for (int k = 0; k < B; k++) {
    const int64_t nk = 96 + 24 * (k % 7) + 8 * (k / 7), d = 2 + k % 3, m = nk;
    n[k] = nk;
    const int64_t n_sparse = nk + 1, E = nk + nk * (2 * d + 1);
    int64_t* ent_ptr = malloc((size_t)(n_sparse + 1) * sizeof *ent_ptr);
    int64_t* I = malloc((size_t)E * sizeof *I);
    int64_t* J = malloc((size_t)E * sizeof *J);
    double* V = malloc((size_t)E * sizeof *V);
    int64_t* gids = malloc((size_t)n_sparse * sizeof *gids);
    int64_t e = 0;
    for (int64_t i = 0; i < nk; i++) { /* A_i = e_i e_iᵀ */
      ent_ptr[i] = e;
      gids[i] = i;
      I[e] = J[e] = i;
      V[e++] = 1.0;
    }
    ent_ptr[nk] = e; /* C = −¼(Diag(deg) − A), global index m */
    gids[nk] = m;
    double c2 = 0.0;
    for (int64_t j = 0; j < nk; j++)
      for (int64_t t = -d; t <= d; t++) {
        I[e] = ((j + t) % nk + nk) % nk;
        J[e] = j;
        V[e] = t == 0 ? -0.25 * (double)(2 * d) : 0.25;
        c2 += V[e] * V[e];
        e++;
      }
    ent_ptr[n_sparse] = e;
    normC[k] = sqrt(c2);
    CK(sdplr_hip_create(nk, m, r, h, &s[k]), NULL);
    CK(sdplr_hip_set_sparse_coo(s[k], 0, n_sparse, ent_ptr, I, J, V, gids), s[k]);
    CK(sdplr_hip_finalize(s[k]), s[k]);
    free(ent_ptr); free(I); free(J); free(V); free(gids);
    /* SolverVars: R0 ~ U(−1, 1) (src/structs.jl:236), λ0 = 0, σ0 = 2, b = 1 */
    double* R = malloc((size_t)(nk * r) * sizeof *R);
    double* b = malloc((size_t)m * sizeof *b);
    v0[k] = malloc((size_t)nk * sizeof **v0);
    for (int64_t i = 0; i < nk * r; i++) R[i] = 2.0 * uniform01() - 1.0;
    for (int64_t i = 0; i < m; i++) b[i] = 1.0;
    for (int64_t i = 0; i < nk; i++) v0[k][i] = uniform01() - 0.5;
    CK(sdplr_hip_set_factor(s[k], SDPLR_F_RT, R), s[k]);
    CK(sdplr_hip_set_vec(s[k], SDPLR_V_B, b, m), s[k]);
    CK(sdplr_hip_set_scalar(s[k], SDPLR_S_SIGMA, 2.0), s[k]);
    free(R); free(b);
  }

  /* ---- the solves, in lockstep ---- */
  sdplr_hip_major_item* mi = calloc((size_t)B, sizeof *mi);
  sdplr_hip_dual_item* di = calloc((size_t)B, sizeof *di);
  sdplr_hip_fg_item* fi = calloc((size_t)B, sizeof *fi);
  int64_t* total = calloc((size_t)B, sizeof *total);
  double sigma = 2.0;
  for (int major = 1; major <= 8; major++) {
    for (int k = 0; k < B; k++) {
      mi[k].s = s[k];
      mi[k].normC = normC[k];
      mi[k].normb = sqrt((double)n[k]);
      mi[k].gtol_relative = mi[k].ptol_relative = 1;
      mi[k].use_armijo = 0;
      mi[k].update_lambda = major > 1; /* λ ← λ − σ·primal_vio with the σ of the round before, src/sdplr.jl:358-362 */
      mi[k].sigma = sigma;             /* (the reference raises σ when the violation stalls, :364-376) */
      mi[k].cur_gtol = 1e-3 / major;
      mi[k].fprec_eps = 1e8 * 2.220446049250313e-16;
      mi[k].max_local_iters = 2000;
      mi[k].time_budget_s = 0.0;
    }
    CK(sdplr_hip_batch_major_iteration(B, mi), NULL);
    for (int k = 0; k < B; k++) {
      total[k] += mi[k].iters_done;
      di[k].s = s[k];
      di[k].trace_bound = (double)n[k];
      di[k].iter = total[k];
      di[k].v0 = v0[k];
    }
    CK(sdplr_hip_batch_dual_obj(B, di), NULL);
    int64_t it_min = mi[0].iters_done, it_max = mi[0].iters_done;
    double worst_gap = 0.0;
    for (int k = 0; k < B; k++) {
      if (mi[k].iters_done < it_min) it_min = mi[k].iters_done;
      if (mi[k].iters_done > it_max) it_max = mi[k].iters_done;
      const double gap = (mi[k].obj - di[k].dual_value) / fabs(mi[k].obj);
      if (gap > worst_gap) worst_gap = gap;
    }
    printf("major %d: σ = %5.0f  inner iterations %lld…%lld per instance  worst relative gap %.3e\n", major, sigma,
           (long long)it_min, (long long)it_max, worst_gap);
    sigma *= 2.0;
  }
  for (int k = 0; k < B; k++) {
    fi[k].s = s[k];
    fi[k].normC = normC[k];
    fi[k].normb = sqrt((double)n[k]);
    fi[k].gtol_relative = fi[k].ptol_relative = 1;
  }
  CK(sdplr_hip_batch_fg(B, fi), NULL); /* src/sdplr.jl:396 */

  /* ---- checks: diag(RRᵀ) = 1, weak duality, a closed gap ---- */
  int ok = 1;
  for (int k = 0; k < B; k++) {
    double* R = malloc((size_t)(n[k] * r) * sizeof *R);
    CK(sdplr_hip_get_factor(s[k], SDPLR_F_RT, R), s[k]);
    double worst = 0.0;
    for (int64_t i = 0; i < n[k]; i++) {
      double t = 0.0;
      for (int64_t q = 0; q < r; q++) t += R[i * r + q] * R[i * r + q];
      if (fabs(t - 1.0) > worst) worst = fabs(t - 1.0);
    }
    free(R);
    const double obj = fi[k].obj, dual = di[k].dual_value;
    /* (the primal point is feasible to ≈ 1e-6 only, so its objective may undercut the bound by that much) */
    const int good = worst < 1e-2 && dual <= obj + 1e-4 * fabs(obj) && (obj - dual) <= 2e-2 * fabs(obj);
    if (!good || k < 3)
      printf("instance %2d: n = %4lld  %5lld iterations  obj = %.6f  dual bound = %.6f  max |‖R_i‖² − 1| = %.1e%s\n", k,
             (long long)n[k], (long long)total[k], obj, dual, worst, good ? "" : "  <-- CHECK FAILED");
    ok = ok && good;
    CK(sdplr_hip_destroy(s[k]), NULL);
    free(v0[k]);
  }
  free(s); free(n); free(normC); free(v0); free(mi); free(di); free(fi); free(total);
  printf("%s\n", ok ? "OK" : "CHECK FAILED");
  return ok ? 0 : 3;
}
