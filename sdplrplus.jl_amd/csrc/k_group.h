// Shared launches for a group of instances on the multi-launch EDGE path (Lovász-θ and alike).
//
// A small instance of this family (Gset G1: n = 800, 19 176 edge constraints, rank 10) cannot take the resident route — its
// (m+1)-vectors alone are 5 × 150 KB — and on the multi-launch route its seven kernels per inner iteration each fill a few
// CUs for 5–15 µs: one stream per instance is a chain of small dependent launches, and a batch overlaps only as far as the
// host threads of the batch call get their launches in.  Here the seven kernels of the while body
// (src/sdplr.jl:190-278) are launched ONCE for the whole group, blockIdx.y ↔ instance: every kernel is the single-instance
// body (k_dense.h / k_sparse.h / k_scalar.h, `*_body`) behind a wrapper that fetches its arguments from row blockIdx.y of
// a table in device memory instead of the launch's argument block.  The instances keep their own control blocks, so each
// leaves the loop on its own tests (its kernels fall through from then on) and the results are those of the
// single-instance route bit for bit: the bodies, their grids (the group shares every launch dimension) and hence every
// order of summation are the same.
#pragma once
#include "k_dense.h"
#include "k_scalar.h"
#include "k_sparse.h"

struct GrpSeamArgs {
  DevCtrl* c;
  int h, jfixed, fin_mode, do_loop, do_coeff, nb_partials;
  const double* partials;
  int desc_mode, m;
  double* pv_raw;
};
struct GrpDirArgs {
  const DevCtrl* c;
  FactorArena A;
  long long N;
  int h, negate, check_done, inline_fallback;
  double* partials;
};
struct GrpSddmmArgs {
  DevSparse sp;
  DevLowRank lr;
  int m, r;
  const double *U, *V;
  double *A_RD, *A_DD, *lr_part, *partials;
  const DevCtrl* c;
};
struct GrpSumsArgs {
  int nout, nb, nb_c, m, n_extra;
  const double* lr_part;
  double *W, *red2;
  const int* extra;
  const double *lam, *pv_raw, *A_RD, *A_DD;
  double* partials;
  const DevCtrl* c;
};
struct GrpLsArgs {
  DevCtrl* c;
  int m, gid_g, n_extra, nb, lr_ST, r, check_done, lr_tail, lr_n;
  const int* extra;
  double *A_RD, *A_DD;
  const double *lam, *lam_ub;
  double* pv_raw;
  const double* lb;
  double *pv, *y;
  const int* lr_col_gid;
  const double* lr_D;
  double *lrW, *lrWS;
  const double* partials;
  const int *lr_mat_ptr, *lr_mat_gid;
  const double* red2;
  ExtraHead eh;
};
struct GrpStepArgs {
  DevSparse sp;
  DevCtrl* c;
  double* R;
  const double* D;
  long long N;
  int nb_ax, nb_c, m, n_extra, nb_gnorm;
  const int* extra;
  double* pv_raw;
  const double *A_RD, *A_DD, *lb;
  double *pv, *y;
  const double *lam, *lam_ub;
  double* partials;
};
struct GrpSpmmArgs {
  DevSparse sp;
  DevLowRank lr;
  FactorArena A;
  const double* X;
  double* Y;
  int r, slot, nb_long, h;
  double scale;
  const double* WS;
  double* partials;
  DevCtrl* c;
  const double* D;
};

__global__ void __launch_bounds__(1024) k_grp_boundary(const GrpSeamArgs* __restrict__ tab) {
  const GrpSeamArgs a = tab[blockIdx.y];
  lbfgs_boundary_body(a.c, a.h, a.jfixed, a.fin_mode, a.do_loop, a.do_coeff, a.nb_partials, a.partials, a.desc_mode, a.pv_raw, a.m);
}
template <int HM>
__global__ void __launch_bounds__(SDPLR_NT) k_grp_dir(const GrpDirArgs* __restrict__ tab) {
  const GrpDirArgs a = tab[blockIdx.y];
  lbfgs_dir_body<HM, false>(a.c, a.A, a.N, a.h, a.negate, a.check_done, a.partials, a.inline_fallback);
}
template <int LPR, int VEC, int LRN>
__global__ void __launch_bounds__(SDPLR_NT) k_grp_sddmm_edge(const GrpSddmmArgs* __restrict__ tab) {
  const GrpSddmmArgs a = tab[blockIdx.y];
  sddmm_edge_body<LPR, VEC, LRN>(a.sp, a.m, a.U, a.V, a.r, a.A_RD, a.A_DD, a.lr, a.lr_part, a.partials, a.c);
}
__global__ void __launch_bounds__(SDPLR_NT) k_grp_edge_sums(const GrpSumsArgs* __restrict__ tab) {
  const GrpSumsArgs a = tab[blockIdx.y];
  edge_sums_body(a.nout, a.nb, a.lr_part, a.W, a.red2, a.nb_c, a.m, a.n_extra, a.extra, a.lam, a.pv_raw, a.A_RD, a.A_DD, a.partials, a.c);
}
__global__ void __launch_bounds__(SDPLR_LSF_NT) k_grp_ls_solve_fast(const GrpLsArgs* __restrict__ tab) {
  const GrpLsArgs a = tab[blockIdx.y];
  ls_solve_fast_body(a.c, a.m, a.gid_g, a.n_extra, a.extra, a.nb, a.A_RD, a.A_DD, a.lam, a.lam_ub, a.pv_raw, a.lb, a.pv, a.y, a.lr_ST, a.r,
                     a.lr_col_gid, a.lr_D, a.lrW, a.lrWS, a.partials, a.check_done, a.lr_tail, a.lr_n, a.lr_mat_ptr, a.lr_mat_gid, a.red2,
                     a.eh);
}
__global__ void __launch_bounds__(SDPLR_NT) k_grp_edge_step(const GrpStepArgs* __restrict__ tab) {
  const GrpStepArgs a = tab[blockIdx.y];
  edge_step_body(a.sp, a.c, a.R, a.D, a.N, a.nb_ax, a.nb_c, a.m, a.n_extra, a.extra, a.pv_raw, a.A_RD, a.A_DD, a.lb, a.pv, a.y, a.lam,
                 a.lam_ub, a.partials, a.nb_gnorm);
}
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT) k_grp_spmm_both_upd(const GrpSpmmArgs* __restrict__ tab) {
  const GrpSpmmArgs a = tab[blockIdx.y];
  spmm_both_upd_body<LPR, VEC, 4, false>(a.sp, a.X, a.Y, a.r, a.scale, a.lr, a.WS, a.slot, a.partials, a.c, a.nb_long, a.A, a.h, a.D, nullptr);
}
// done flags of the group's control blocks, gathered for ONE copy back per batch of iterations
__global__ void __launch_bounds__(64) k_grp_poll(const DevCtrl* const* __restrict__ ctrls, int count, int* __restrict__ out) {
  for (int b = threadIdx.x; b < count; b += 64) out[b] = ctrls[b]->done;
}
