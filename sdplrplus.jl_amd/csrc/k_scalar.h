// k_scalar.h — O(m) vector kernels and the single-block "scalar stage" kernels that keep every
// solver scalar on the device: Lagrangian value, line-search coefficient sums and root selection,
// primal-violation bookkeeping, the inner loop's exit tests, Armijo backtracking, Lanczos updates.
#pragma once
#include "common.h"

#define SDPLR_EPS 2.220446049250313e-16

// ---- f! tail: src/coreop.jl:16-29 on pv_raw = 𝒜(RRᵀ) ------------------------------------------------
__global__ void __launch_bounds__(SDPLR_NT)
k_f_tail(DevCtrl* __restrict__ c, int m, double* __restrict__ pv_raw, const double* __restrict__ b,
         const double* __restrict__ lb, double* __restrict__ pv, const double* __restrict__ lam,
         const double* __restrict__ lam_ub, double* __restrict__ partials) {
  __shared__ double sh[8];
  const double sigma = c->sigma;
  double t = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < m; i += stride) {
    const double v = pv_raw[i] - b[i];                      // v .-= b            (:20)
    pv_raw[i] = v;
    pv[i] = fmax(v, lb[i]);                                 // capped violation   (:22)
    const double l = lam[i];
    const double yi = fmin(lam_ub[i], l - sigma * v);       // (:27)
    t += (yi * yi - l * l) / (2 * sigma);                   // (:28)
  }
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_F)[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SDPLR_NT)
k_f_finalize(DevCtrl* __restrict__ c, int m, const double* __restrict__ pv_raw, int nb,
             const double* __restrict__ partials) {
  __shared__ double sh[8];
  const double s = reduce_partials(slot_partials(partials, SLOT_F), nb, sh);
  if (threadIdx.x == 0) {
    c->obj = pv_raw[m];      // (:16)
    c->L = c->obj + s;       // (:25-30)
  }
}

// copy2y_λ_sub_pvio!  src/coreop.jl:229-236
__global__ void __launch_bounds__(SDPLR_NT)
k_copy2y(const DevCtrl* __restrict__ c, int m, double* __restrict__ y, const double* __restrict__ lam,
         const double* __restrict__ lam_ub, const double* __restrict__ pv_raw, int check_done) {
  if (check_done && c->done) return;
  const double sigma = c->sigma;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i <= m; i += stride)
    y[i] = (i == m) ? 1.0 : -fmin(lam_ub[i], lam[i] - sigma * pv_raw[i]);
}

// primal_vio = max(primal_vio_raw, lb) and ‖primal_vio‖² partials  (src/coreop.jl:340-347)
__global__ void __launch_bounds__(SDPLR_NT)
k_pv_norm(int m, const double* __restrict__ pv_raw, const double* __restrict__ lb, double* __restrict__ pv,
          int recompute, double* __restrict__ partials) {
  __shared__ double sh[8];
  double t = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < m; i += stride) {
    double v;
    if (recompute) {
      v = fmax(pv_raw[i], lb[i]);
      pv[i] = v;
    } else {
      v = pv[i];
    }
    t += v * v;
  }
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_PVNORM2)[blockIdx.x] = t;
}

// grad_norm / primal_vio_norm from the ‖·‖² partials (src/sdplr.jl:224-234, src/coreop.jl:334-347) and,
// with `loop_tail`, the relative-decrease exit of src/sdplr.jl:238-241.  One block.
__global__ void __launch_bounds__(SDPLR_NT)
k_norms(DevCtrl* __restrict__ c, int nb_g, int nb_p, int loop_tail, int check_done,
        const double* __restrict__ partials) {
  __shared__ double sh[8];
  if (check_done && c->done) return;
  const double g2 = reduce_partials(slot_partials(partials, SLOT_GNORM2), nb_g, sh);
  __syncthreads();
  const double p2 = reduce_partials(slot_partials(partials, SLOT_PVNORM2), nb_p, sh);
  if (threadIdx.x != 0) return;
  const double g = sqrt(g2), p = sqrt(p2);
  c->gnorm = c->grel ? g / c->normC : g;
  c->pvnorm = c->prel ? p / c->normb : p;
  if (loop_tail) {
    const double L = c->L, last = c->lastval;
    const double rel_delta = (last - L) / fmax(1.0, fmax(fabs(L), fabs(last)));
    if (rel_delta < c->fprec_eps) {
      c->done = 1;
      c->exit_reason = EXIT_RELDELTA;
    }
  }
}

// λᵢ ← min(λ_ubᵢ, λᵢ − σ·pv_rawᵢ)   (src/sdplr.jl:358-362)
__global__ void __launch_bounds__(SDPLR_NT)
k_update_lambda(const DevCtrl* __restrict__ c, int m, double* __restrict__ lam,
                const double* __restrict__ lam_ub, const double* __restrict__ pv_raw) {
  const double sigma = c->sigma;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < m; i += stride)
    lam[i] = fmin(lam_ub[i], lam[i] - sigma * pv_raw[i]);
}

// Σ partials of one slot → *dst.  One block.
__global__ void __launch_bounds__(SDPLR_NT)
k_reduce_slot(double* __restrict__ dst, int slot, int nb, const double* __restrict__ partials) {
  __shared__ double sh[8];
  const double s = reduce_partials(slot_partials(partials, slot), nb, sh);
  if (threadIdx.x == 0) *dst = s;
}
// ⟨x, y⟩ partials of two short vectors
__global__ void __launch_bounds__(SDPLR_NT)
k_dot(int len, const double* __restrict__ x, const double* __restrict__ y, int slot, double* __restrict__ partials) {
  __shared__ double sh[8];
  double t = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < len; i += stride) t += x[i] * y[i];
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = t;
}

// ---- exact line search: coefficient sums (src/linesearch.jl:36-56) -----------------------------------
//   s0 = λ·(−q0)  s1 = ‖q0‖²  s2 = λ·q1  s3 = (−q0)·q1  s4 = (λ − σ(−q0))·q2  s5 = ‖q1‖²  s6 = q1·q2  s7 = ‖q2‖²
__global__ void __launch_bounds__(SDPLR_NT)
k_ls_partials(const DevCtrl* __restrict__ c, int m, const double* __restrict__ lam,
              const double* __restrict__ pv_raw, const double* __restrict__ A_RD,
              const double* __restrict__ A_DD, double* __restrict__ partials, int check_done) {
  __shared__ double sh[8 * (SDPLR_NT / 64)];
  if (check_done && c->done) return;
  const double sigma = c->sigma;
  double s[8];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < m; i += stride) {
    const double l = lam[i], nq0 = pv_raw[i], q1 = A_RD[i], q2 = A_DD[i];
    s[0] += l * nq0;
    s[1] += nq0 * nq0;
    s[2] += l * q1;
    s[3] += nq0 * q1;
    s[4] += (l - sigma * nq0) * q2;
    s[5] += q1 * q1;
    s[6] += q1 * q2;
    s[7] += q2 * q2;
  }
  block_sum<8>(s, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) slot_partials(partials, SLOT_LS + k)[blockIdx.x] = s[k];
  }
}

__device__ inline double horner4(const double* b, double x) {
  return (((b[4] * x + b[3]) * x + b[2]) * x + b[1]) * x + b[0];
}
// real roots of c0 + c1 x + c2 x² + c3 x³ (c3 ≠ 0): trigonometric / Cardano closed form, then Newton
// polish on the unscaled cubic.  Stands in for PolynomialRoots.roots (src/linesearch.jl:94) — the
// reference keeps a root only if |imag| < eps (:100); real roots are produced here directly.
__device__ inline int cubic_real_roots(double c0, double c1, double c2, double c3, double* roots) {
  const double a = c2 / c3, b = c1 / c3, cc = c0 / c3;
  const double p = b - a * a / 3.0;
  const double q = 2.0 * a * a * a / 27.0 - a * b / 3.0 + cc;
  const double disc = q * q / 4.0 + p * p * p / 27.0;
  int nr = 0;
  if (disc > 0.0) {
    const double sq = sqrt(disc);
    const double A = (q > 0.0) ? -cbrt(q / 2.0 + sq) : cbrt(-q / 2.0 + sq);
    const double B = (A != 0.0) ? -p / (3.0 * A) : 0.0;
    roots[nr++] = A + B - a / 3.0;
  } else if (p == 0.0) {
    roots[nr++] = -a / 3.0;  // triple root
  } else {
    const double mm = 2.0 * sqrt(-p / 3.0);
    double arg = 3.0 * q / (p * mm);
    arg = fmin(1.0, fmax(-1.0, arg));
    const double th = acos(arg) / 3.0;
    const double twopi3 = 2.0943951023931953;
    for (int k = 0; k < 3; k++) roots[nr++] = mm * cos(th - twopi3 * k) - a / 3.0;
  }
  for (int i = 0; i < nr; i++) {
    double x = roots[i];
    double fx = ((c3 * x + c2) * x + c1) * x + c0;
    for (int it = 0; it < 4; it++) {
      const double d = (3.0 * c3 * x + 2.0 * c2) * x + c1;
      if (d == 0.0) break;
      const double xn = x - fx / d;
      const double fn = ((c3 * xn + c2) * xn + c1) * xn + c0;
      if (!(fabs(fn) < fabs(fx))) break;
      x = xn;
      fx = fn;
    }
    roots[i] = x;
  }
  return nr;
}
// scalar stage of linesearch!  src/linesearch.jl:58-112.  Returns 0 or −3 (not a descent direction).
__device__ inline int quartic_argmin(const double* bq, double alpha_max, double* alpha, double* fval) {
  const double k0 = 1.0 * bq[1];
  if (k0 > SDPLR_EPS) return -3;  // :60-62
  const double k1 = 2.0 * bq[2], k2 = 3.0 * bq[3], k3 = 4.0 * bq[4];
  double roots[4];
  int nr = 0;
  if (fabs(k3) < SDPLR_EPS) {     // :70-83 quadratic (or lower) derivative
    if (k2 != 0.0) {
      const double disc = k1 * k1 - 4.0 * k2 * k0;
      if (disc >= 0.0) {
        const double sq = sqrt(disc);
        const double qq = -0.5 * (k1 + (k1 >= 0.0 ? sq : -sq));
        roots[nr++] = qq / k2;
        if (qq != 0.0) roots[nr++] = k0 / qq;
      }
    } else if (k1 != 0.0) {
      roots[nr++] = -k0 / k1;
    }
  } else {                        // :86-95
    nr = cubic_real_roots(k0, k1, k2, k3, roots);
  }
  roots[nr++] = alpha_max;
  double a_star = 0.0, f_star = bq[0];
  for (int i = 0; i < nr; i++) {  // :98-112
    const double root = roots[i];
    if (!(root >= 0.0) || root > alpha_max) continue;
    const double fa = horner4(bq, root);
    if (fa < f_star) {
      f_star = fa;
      a_star = root;
    }
  }
  *alpha = a_star;
  *fval = f_star;
  return 0;
}

// One block: quartic coefficients (src/linesearch.jl:44-56), root selection, α* and ℒ(α*).
__global__ void __launch_bounds__(SDPLR_NT)
k_ls_solve(DevCtrl* __restrict__ c, int m, int nb, const double* __restrict__ A_RD,
           const double* __restrict__ A_DD, const double* __restrict__ partials, int check_done,
           int loop_mode) {
  __shared__ double sh[8 * (SDPLR_NT / 64)];
  if (check_done && c->done) return;
  double s[8];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = 0.0;
  for (int i = threadIdx.x; i < nb; i += SDPLR_NT) {
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] += slot_partials(partials, SLOT_LS + k)[i];
  }
  block_sum<8>(s, sh);
  if (threadIdx.x != 0) return;
  const double sigma = c->sigma, p0 = c->obj, p1 = A_RD[m], p2 = A_DD[m];
  double bq[5];
  bq[0] = p0 - s[0] + sigma * s[1] / 2;
  bq[1] = p1 - s[2] + sigma * s[3];
  bq[2] = p2 - s[4] + sigma * s[5] / 2;
  bq[3] = sigma * s[6];
  bq[4] = sigma * s[7] / 2;
  for (int k = 0; k < 5; k++) c->biquad[k] = bq[k];
  double a = 0.0, f = bq[0];
  const int rc = quartic_argmin(bq, c->alpha_max, &a, &f);
  if (rc != 0) {
    c->err = rc;
    c->done = 1;
    return;
  }
  c->alpha = a;
  c->L = f;
  if (loop_mode) {  // src/sdplr.jl:238-241, decided as soon as ℒ(α*) is known; acted on after g! and the norms
    const double last = c->lastval;
    const double rel_delta = (last - f) / fmax(1.0, fmax(fabs(f), fabs(last)));
    c->reldelta_exit = (rel_delta < c->fprec_eps) ? 1 : 0;
  }
}

// commit of either line search (src/linesearch.jl:118-124 / :184-188): pv_raw += α(α·A_DD + A_RD),
// obj, capped violation and its ‖·‖² partials; with `fuse_y`, y of the following g! (copy2y_λ_sub_pvio!)
__global__ void __launch_bounds__(SDPLR_NT)
k_ls_commit(DevCtrl* __restrict__ c, int m, double* __restrict__ pv_raw, const double* __restrict__ A_RD,
            const double* __restrict__ A_DD, const double* __restrict__ lb, double* __restrict__ pv,
            int fuse_y, double* __restrict__ y, const double* __restrict__ lam,
            const double* __restrict__ lam_ub, double* __restrict__ partials, int check_done,
            int loop_mode, int nb_gnorm) {
  __shared__ double sh[8];
  if (check_done && c->done) return;
  const double a = c->alpha, sigma = c->sigma;
  double t = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i <= m; i += stride) {
    const double v = pv_raw[i] + a * (a * A_DD[i] + A_RD[i]);
    pv_raw[i] = v;
    if (i == m) {
      c->obj = v;
      if (fuse_y) y[m] = 1.0;
    } else {
      const double pc = fmax(v, lb[i]);
      pv[i] = pc;
      t += pc * pc;
      if (fuse_y) y[i] = -fmin(lam_ub[i], lam[i] - sigma * v);
    }
  }
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_PVNORM2)[blockIdx.x] = t;
    if (loop_mode && blockIdx.x == 0) {  // the g! that follows produces the ‖G‖² partials; the seam kernel folds both
      c->norms_pending = 1;
      c->nb_pvnorm = gridDim.x;
      c->nb_gnorm = nb_gnorm;
    }
  }
}

// ---- Armijo backtracking (src/linesearch.jl:139-191), all 51 trial steps in one sweep ---------------
// sum[k] = Σ_i (λ̃ᵢ(α_k)² − λᵢ²)/(2σ), α_k = α_max/2^k, k = 0..50; sum[51] the same at α = 0;
// sum[52] = Σ_i yᵢ·A_RDᵢ (slope, :171).
__global__ void __launch_bounds__(SDPLR_NT)
k_armijo_partials(const DevCtrl* __restrict__ c, int m, const double* __restrict__ lam,
                  const double* __restrict__ lam_ub, const double* __restrict__ pv_raw,
                  const double* __restrict__ A_RD, const double* __restrict__ A_DD,
                  const double* __restrict__ y, double* __restrict__ partials, int check_done) {
  __shared__ double sh[53 * (SDPLR_NT / 64)];
  if (check_done && c->done) return;
  const double sigma = c->sigma, amax = c->alpha_max;
  double s[53];
#pragma unroll
  for (int k = 0; k < 53; k++) s[k] = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < m; i += stride) {
    const double l = lam[i], ub = lam_ub[i], g0 = pv_raw[i], q1 = A_RD[i], q2 = A_DD[i];
    double a = amax;
#pragma unroll
    for (int k = 0; k < 51; k++) {
      const double gi = g0 + a * q1 + a * a * q2;            // :160
      const double lt = fmin(ub, l - sigma * gi);            // :161
      s[k] += (lt * lt - l * l) / (2 * sigma);               // :162
      a /= 2;
    }
    const double lt0 = fmin(ub, l - sigma * g0);
    s[51] += (lt0 * lt0 - l * l) / (2 * sigma);
    s[52] += y[i] * q1;
  }
  block_sum<53>(s, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 53; k++) slot_partials(partials, SLOT_ARMIJO + k)[blockIdx.x] = s[k];
  }
}
__global__ void __launch_bounds__(SDPLR_NT)
k_armijo_pick(DevCtrl* __restrict__ c, int m, int nb, const double* __restrict__ A_RD,
              const double* __restrict__ A_DD, const double* __restrict__ partials, int check_done,
              int loop_mode) {
  __shared__ double tot[53];
  __shared__ double sh[8];
  if (check_done && c->done) return;
  for (int k = 0; k < 53; k++) {
    const double v = reduce_partials(slot_partials(partials, SLOT_ARMIJO + k), nb, sh);
    if (threadIdx.x == 0) tot[k] = v;
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double obj = c->obj, p1 = A_RD[m], p2 = A_DD[m];
  const double L0 = obj + tot[51];                      // eval_AL(0)      :167
  const double slope = p1 + tot[52];                    // :171
  const double cc = 1e-4;                               // :173
  double a = c->alpha_max;
  int k = 0;
  double La = obj + a * p1 + a * a * p2 + tot[0];       // :175
  for (int it = 0; it < 50; it++) {                     // :177-181
    if (La <= L0 + cc * a * slope) break;
    a /= 2;
    k++;
    La = obj + a * p1 + a * a * p2 + tot[k];
  }
  c->alpha = a;
  c->L = La;
  if (loop_mode) {
    const double last = c->lastval;
    const double rel_delta = (last - La) / fmax(1.0, fmax(fabs(La), fabs(last)));
    c->reldelta_exit = (rel_delta < c->fprec_eps) ? 1 : 0;
  }
}

// ---- Lanczos vector updates (src/coreop.jl:473-499) ---------------------------------------------------
// v = v0/‖v0‖  (:474)
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_init(int n, const double* __restrict__ v0, double* __restrict__ v, int nb, const double* __restrict__ partials) {
  __shared__ double sh[8];
  const double nv = sqrt(reduce_partials(slot_partials(partials, SLOT_V0), nb, sh));
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < n; i += stride) v[i] = v0[i] / nv;
}
// alpha[i] = v·Av (:484); Av −= alpha[i]·v (+ beta[i−1]·v_pre) (:486-490); ‖Av‖² partials (:492)
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_update1(DevCtrl* __restrict__ c, int n, int step, const double* __restrict__ v, double* __restrict__ Av,
             const double* __restrict__ vpre, double* __restrict__ alpha_out, int nb_in,
             double* __restrict__ partials) {
  __shared__ double sh[8];
  if (c->lz_done) return;
  const double al = reduce_partials(slot_partials(partials, SLOT_LZ_A), nb_in, sh);
  const double be = c->lz_beta_prev;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    alpha_out[step] = al;
    c->lz_steps = step + 1;  // iter += 1 (:482)
  }
  double t = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < n; i += stride) {
    double a = Av[i];
    if (step == 0) a -= al * v[i];
    else a -= al * v[i] + be * vpre[i];
    Av[i] = a;
    t += a * a;
  }
  __syncthreads();
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_LZ_B)[blockIdx.x] = t;
}
// beta[i] = ‖Av‖ (:492); stop if |beta| < √n·eps (:494-496); else Av ./= beta (:497)
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_update2(DevCtrl* __restrict__ c, int n, int step, double* __restrict__ Av, double* __restrict__ beta_out,
             int nb_in, const double* __restrict__ partials) {
  __shared__ double sh[8];
  if (c->lz_done) return;
  const double be = sqrt(reduce_partials(slot_partials(partials, SLOT_LZ_B), nb_in, sh));
  const bool stop = fabs(be) < sqrt((double)n) * SDPLR_EPS;
  // every block has read lz_done before any block can have set it below only if the flag is written
  // last: a late-starting block could otherwise see lz_done = 1 and skip its share of the scaling —
  // harmless, because after a stop the vectors are never used again.
  if (!stop) {
    const int stride = gridDim.x * SDPLR_NT;
    for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i < n; i += stride) Av[i] /= be;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    beta_out[step] = be;
    c->lz_beta_prev = be;
    if (stop) c->lz_done = 1;
  }
}

// ---- scalar stage of the singleton fast path (k_sparse.h) ----------------------------------------------
// One block: fold the row-attached line-search sums, add the extra slots (A_g and the low-rank matrices),
// solve the quartic (src/linesearch.jl:44-112), decide the relative-decrease exit (src/sdplr.jl:238-241),
// commit the extra slots (src/linesearch.jl:118-124, src/coreop.jl:229-236) and prepare the low-rank
// coefficients WS of the moved point.
// The sums come from ≤ 1024 partials each, fetched four per thread in one round trip (unconditional loads: the slot
// arrays are SDPLR_MAXNB wide); `red2` ≠ null: sums 8 and 9 (the big matrix) were already folded by k_edge_sums.
#define SDPLR_LSF_NT SDPLR_NT
#define SDPLR_LSF_EXMAX 16
#define SDPLR_LSF_LRW 512     /* doubles of low-rank projections staged in LDS (2·ST·r) */
#define SDPLR_LSF_LRN 16      /* low-rank matrices whose line-search values are kept in LDS */
struct ExtraHead { int k[4]; };   // the first extra slots by value: their data is requested without waiting for the index list
__device__ __forceinline__ void ls_solve_fast_body(DevCtrl* __restrict__ c, int m, int gid_g, int n_extra, const int* __restrict__ extra, int nb, double* __restrict__ A_RD, double* __restrict__ A_DD, const double* __restrict__ lam, const double* __restrict__ lam_ub, double* __restrict__ pv_raw, const double* __restrict__ lb, double* __restrict__ pv, double* __restrict__ y, int lr_ST, int r, const int* __restrict__ lr_col_gid, const double* __restrict__ lr_D, double* __restrict__ lrW, double* __restrict__ lrWS, const double* __restrict__ partials, int check_done, int lr_tail, int lr_n, const int* __restrict__ lr_mat_ptr, const int* __restrict__ lr_mat_gid, const double* __restrict__ red2, ExtraHead eh) {
  __shared__ double sh[10 * (SDPLR_LSF_NT / 64)];
#ifdef SDPLR_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
  const long long stamp_it = c->iters;
#endif
  const int dn = check_done ? c->done : 0;
  // every scalar the serial part needs, requested up front: their latency overlaps the partial sums below
  // instead of forming a chain of dependent global round trips in thread 0
  const double sigma = c->sigma, obj0 = c->obj, amax = c->alpha_max, last = c->lastval, feps = c->fprec_eps;
  // A_RD[m], A_DD[m] as stored (the low-rank tail below overrides them when the cost matrix is one of its matrices)
  double rd_m = A_RD[m], dd_m = A_DD[m];
  // low-rank tail: the projections W = [RᵀB; DᵀB] and the matrices' column ranges are requested now, with everything
  // else, into LDS (≤ SDPLR_LSF_LRW doubles of W, ≤ SDPLR_LSF_LRN matrices; larger sets take the global-memory route)
  __shared__ double lw[SDPLR_LSF_LRW];
  __shared__ int lr_p[SDPLR_LSF_LRN + 1], lr_g[SDPLR_LSF_LRN];
  __shared__ double lr_rd[SDPLR_LSF_LRN], lr_dd[SDPLR_LSF_LRN], lr_dv[SDPLR_LSF_NT];
  const bool lr_lds = lr_tail && 2 * lr_ST * r <= SDPLR_LSF_LRW && lr_n <= SDPLR_LSF_LRN && lr_ST <= SDPLR_LSF_NT;
  // Everything is requested into REGISTERS first and parked in LDS only after the partials below have been requested
  // as well: an LDS store of a loaded value makes the wave wait for that load at the store's place in the program — with
  // the stores up here the kernel paid three memory round trips (projections, extra slots, partials) instead of one
  // (in-kernel stamps: 8.1 k cycles until the loads were in, against 5.5 k without the low-rank data).
  constexpr int LWT = SDPLR_LSF_LRW / SDPLR_LSF_NT;
  double lw_r[LWT];
  int lrp_r = 0, lrg_r = 0;
  double lrd_r = 0.0;
  int lrc_r = 0;
  if (lr_lds) {
    lrd_r = lr_D[min((int)threadIdx.x, lr_ST - 1)];
    lrc_r = lr_col_gid[min((int)threadIdx.x, lr_ST - 1)];
#pragma unroll
    for (int q = 0; q < LWT; q++) {
      const int t = (int)threadIdx.x + q * SDPLR_LSF_NT;
      lw_r[q] = lrW[min(t, 2 * lr_ST * r - 1)];
    }
    lrp_r = lr_mat_ptr[min((int)threadIdx.x, lr_n)];
    lrg_r = lr_mat_gid[min((int)threadIdx.x, lr_n - 1)];
  }
  // the extra slots' own data (≤ SDPLR_LSF_EXMAX of them staged; more fall back to global reads), fetched by one thread
  // each alongside the partials: the serial part below then reads LDS instead of chaining global round trips
  __shared__ int ex_k[SDPLR_LSF_EXMAX];
  __shared__ double ex_v[SDPLR_LSF_EXMAX][4];   // λ, λ_ub, primal_vio_raw, lb
  __shared__ double ex_y[SDPLR_LSF_EXMAX];      // y of the extra slots as committed below
  __shared__ int lr_cg[SDPLR_LSF_NT];           // owner (constraint index) of each low-rank column
  const bool ex_mine = (int)threadIdx.x < n_extra && threadIdx.x < SDPLR_LSF_EXMAX;
  int ex_kr = eh.k[0];   // (every thread requests: unconditional loads, clamped — threads past n_extra repeat slot 0's)
  if (threadIdx.x == 1 && n_extra > 1) ex_kr = eh.k[1];
  if (threadIdx.x == 2 && n_extra > 2) ex_kr = eh.k[2];
  if (threadIdx.x == 3 && n_extra > 3) ex_kr = eh.k[3];
  if (n_extra > 4 && threadIdx.x >= 4) ex_kr = extra[min((int)threadIdx.x, n_extra - 1)];
  double ex_r[4];
  {
    const int kc = min(ex_kr, m - (m > 0));
    ex_r[0] = lam[kc];
    ex_r[1] = lam_ub[kc];
    ex_r[2] = pv_raw[ex_kr];
    ex_r[3] = lb[kc];
  }
  double s[10];
  {
    // 16 bytes per lane (two neighbouring partials), and only the columns that hold partials — this one CU issues a
    // wave-load every ≈ 16 cycles whatever its width; absent entries are masked, so the sums do not depend on it.
    constexpr int PT = 2;                       // double2 columns of NT lanes: 2·PT·NT = 1024 partials
    double2 v[10][PT];
    const int ncols = (nb + 2 * SDPLR_LSF_NT - 1) / (2 * SDPLR_LSF_NT);
#pragma unroll
    for (int q = 0; q < PT; q++) {
      const int i = (int)threadIdx.x + SDPLR_LSF_NT * q;
      if (q < ncols) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k][q] = reinterpret_cast<const double2*>(slot_partials(partials, SLOT_LS + k))[i];
        v[8][q] = reinterpret_cast<const double2*>(slot_partials(partials, SLOT_PD))[i];
        v[9][q] = reinterpret_cast<const double2*>(slot_partials(partials, SLOT_DW))[i];
      } else {
#pragma unroll
        for (int k = 0; k < 10; k++) v[k][q].x = v[k][q].y = 0.0;
      }
    }
#pragma unroll
    for (int k = 0; k < 10; k++) {
      s[k] = 0.0;
#pragma unroll
      for (int q = 0; q < PT; q++) {
        const int i = 2 * ((int)threadIdx.x + SDPLR_LSF_NT * q);
        s[k] += (i < nb) ? v[k][q].x : 0.0;
        s[k] += (i + 1 < nb) ? v[k][q].y : 0.0;
      }
    }
    for (int i = threadIdx.x + 2 * PT * SDPLR_LSF_NT; i < nb; i += SDPLR_LSF_NT) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] += slot_partials(partials, SLOT_LS + k)[i];
      s[8] += slot_partials(partials, SLOT_PD)[i];
      s[9] += slot_partials(partials, SLOT_DW)[i];
    }
    if (red2) {
      s[8] = (threadIdx.x == 0) ? red2[0] : 0.0;
      s[9] = (threadIdx.x == 0) ? red2[1] : 0.0;
    }
  }
  if (lr_lds) {
#pragma unroll
    for (int q = 0; q < LWT; q++) {
      const int t = (int)threadIdx.x + q * SDPLR_LSF_NT;
      if (t < 2 * lr_ST * r) lw[t] = lw_r[q];
    }
    if ((int)threadIdx.x <= lr_n) lr_p[threadIdx.x] = lrp_r;
    if ((int)threadIdx.x < lr_n) lr_g[threadIdx.x] = lrg_r;
    if ((int)threadIdx.x < lr_ST) {
      lr_dv[threadIdx.x] = lrd_r;
      lr_cg[threadIdx.x] = lrc_r;
    }
  }
  if (ex_mine) {
    ex_k[threadIdx.x] = ex_kr;
#pragma unroll
    for (int q = 0; q < 4; q++) ex_v[threadIdx.x][q] = ex_r[q];
  }
#ifdef SDPLR_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long st_ld = __builtin_amdgcn_s_memtime();
#endif
  if (dn) return;
  if (lr_tail) {
    // the tail of k_lr_finalize, mode 2 (src/linesearch.jl:10-16 for the low-rank matrices):
    // A_RD[gid] = 2·Σ_c D_c⟨W0_c, W1_c⟩, A_DD[gid] = Σ_c D_c‖W1_c‖², from the projections W = [RᵀB; DᵀB]
    const int per = lr_ST * r;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lr_lds) __syncthreads();
    for (int t = wave; t < lr_n; t += SDPLR_LSF_NT / 64) {   // one wave per matrix, lanes over the rank
      double s0 = 0.0, s1 = 0.0;
      const int c0 = lr_lds ? lr_p[t] : lr_mat_ptr[t], c1 = lr_lds ? lr_p[t + 1] : lr_mat_ptr[t + 1];
      for (int cc = c0; cc < c1; cc++) {
        double d0 = 0.0, d1 = 0.0;
        for (int k = lane; k < r; k += 64) {
          const double w0 = lr_lds ? lw[cc * r + k] : lrW[cc * r + k];
          const double w1 = lr_lds ? lw[per + cc * r + k] : lrW[per + cc * r + k];
          d0 += w0 * w1;
          d1 += w1 * w1;
        }
        const double dc = lr_lds ? lr_dv[cc] : lr_D[cc];
        s0 += wave_sum(d0) * dc;
        s1 += wave_sum(d1) * dc;
      }
      if (lane == 0) {
        const int gid = lr_lds ? lr_g[t] : lr_mat_gid[t];
        A_RD[gid] = 2.0 * s0;
        A_DD[gid] = s1;
        if (lr_lds) {
          lr_rd[t] = 2.0 * s0;
          lr_dd[t] = s1;
        }
      }
    }
    __syncthreads();
    if (lr_lds) {       // the values just formed are read back from LDS, not through global memory
      for (int t = 0; t < lr_n; t++)
        if (lr_g[t] == m) {
          rd_m = lr_rd[t];
          dd_m = lr_dd[t];
        }
    } else {
      rd_m = A_RD[m];
      dd_m = A_DD[m];
    }
  }
  {  // block sums over the 16 waves, fixed order
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 10; k++) s[k] = wave_sum(s[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 10; k++) sh[k * (SDPLR_LSF_NT / 64) + wave] = s[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 10; k++) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < SDPLR_LSF_NT / 64; w++) t += sh[k * (SDPLR_LSF_NT / 64) + w];
      s[k] = t;
    }
  }
  __shared__ double sh_alpha;
  __shared__ int sh_err;
#ifdef SDPLR_STAMPS
  const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
  if (threadIdx.x == 0) {
    const double g_rd = s[8] + s[8], g_dd = s[9];
    // 𝒜 values of an extra slot: A_g's from the sums above, a low-rank matrix's from the tail's LDS copy
    auto extra_q = [&](int k, double& q1, double& q2) {
      if (k == gid_g) {
        q1 = g_rd;
        q2 = g_dd;
        return;
      }
      if (lr_lds)
        for (int t = 0; t < lr_n; t++)
          if (lr_g[t] == k) {
            q1 = lr_rd[t];
            q2 = lr_dd[t];
            return;
          }
      q1 = A_RD[k];
      q2 = A_DD[k];
    };
    if (gid_g >= 0) {     // (−1: the edge path without a multi-entry matrix)
      A_RD[gid_g] = g_rd;   // ⟨A_g, RDᵀ+DRᵀ⟩ = 2⟨P, D⟩
      A_DD[gid_g] = g_dd;   // ⟨A_g, DDᵀ⟩ = ⟨D, W⟩
    }
    for (int t = 0; t < n_extra; t++) {
      const bool st = t < SDPLR_LSF_EXMAX;
      const int k = st ? ex_k[t] : extra[t];
      if (k >= m) continue;
      const double l = st ? ex_v[t][0] : lam[k], nq0 = st ? ex_v[t][2] : pv_raw[k];
      double q1, q2;
      extra_q(k, q1, q2);
      s[0] += l * nq0;
      s[1] += nq0 * nq0;
      s[2] += l * q1;
      s[3] += nq0 * q1;
      s[4] += (l - sigma * nq0) * q2;
      s[5] += q1 * q1;
      s[6] += q1 * q2;
      s[7] += q2 * q2;
    }
    const double p0 = obj0, p1 = (gid_g == m) ? g_rd : rd_m, p2 = (gid_g == m) ? g_dd : dd_m;
    double bq[5];
    bq[0] = p0 - s[0] + sigma * s[1] / 2;
    bq[1] = p1 - s[2] + sigma * s[3];
    bq[2] = p2 - s[4] + sigma * s[5] / 2;
    bq[3] = sigma * s[6];
    bq[4] = sigma * s[7] / 2;
    for (int k = 0; k < 5; k++) c->biquad[k] = bq[k];
    double a = 0.0, f = bq[0];
    const int rc = quartic_argmin(bq, amax, &a, &f);
    sh_err = rc;
    sh_alpha = a;
    if (rc != 0) {
      c->err = rc;
      c->done = 1;
    } else {
      c->alpha = a;
      c->L = f;
      const double rel_delta = (last - f) / fmax(1.0, fmax(fabs(f), fabs(last)));
      c->reldelta_exit = (rel_delta < feps) ? 1 : 0;
      // commit of the extra slots
      double pv2 = 0.0;
      for (int t = 0; t < n_extra; t++) {
        const bool st = t < SDPLR_LSF_EXMAX;
        const int k = st ? ex_k[t] : extra[t];
        double q1, q2;
        extra_q(k, q1, q2);
        const double v = (st ? ex_v[t][2] : pv_raw[k]) + a * (a * q2 + q1);
        pv_raw[k] = v;
        double yk = 1.0;
        if (k == m) {
          c->obj = v;
        } else {
          const double pc = fmax(v, st ? ex_v[t][3] : lb[k]);
          pv[k] = pc;
          pv2 += pc * pc;
          yk = -fmin(st ? ex_v[t][1] : lam_ub[k], (st ? ex_v[t][0] : lam[k]) - sigma * v);
        }
        y[k] = yk;
        if (st) ex_y[t] = yk;   // read back by the low-rank tail below without a trip through global memory
      }
      c->pv2_extra = pv2;
    }
  }
  __syncthreads();
#ifdef SDPLR_STAMPS
  if (threadIdx.x == 0 && (stamp_it % 64) == 33) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
    printf("[ls_solve_fast] loads in %llu  loads+sums %llu  serial+stores %llu (s_memtime ticks)\n", st_ld - st0, st1 - st0, st2 - st1);
  }
#endif
  if (sh_err != 0) return;
  const double a = sh_alpha;
  const int per = lr_ST * r;
  const int n_st = min(n_extra, SDPLR_LSF_EXMAX);
  for (int t = threadIdx.x; t < per; t += SDPLR_LSF_NT) {  // W0 ← W0 + α·W1 = R_newᵀB ; WS = y·D·W0
    const int cc = t / r;
    double w, yc, dc;
    if (lr_lds) {   // everything is at hand in LDS: the projections, D_c, the owner and — committed above — its y
      w = lw[t] + a * lw[per + t];
      dc = lr_dv[cc];
      const int gid = lr_cg[cc];
      bool found = false;
      yc = 0.0;
      for (int e = 0; e < n_st; e++)
        if (ex_k[e] == gid) {
          yc = ex_y[e];
          found = true;
        }
      if (!found) yc = y[gid];
    } else {
      w = lrW[t] + a * lrW[per + t];
      dc = lr_D[cc];
      yc = y[lr_col_gid[cc]];
    }
    lrW[t] = w;
    lrWS[t] = yc * dc * w;
  }
}
__global__ void __launch_bounds__(SDPLR_LSF_NT)
k_ls_solve_fast(DevCtrl* __restrict__ c, int m, int gid_g, int n_extra, const int* __restrict__ extra,
                int nb, double* __restrict__ A_RD, double* __restrict__ A_DD,
                const double* __restrict__ lam, const double* __restrict__ lam_ub,
                double* __restrict__ pv_raw, const double* __restrict__ lb, double* __restrict__ pv,
                double* __restrict__ y, int lr_ST, int r, const int* __restrict__ lr_col_gid,
                const double* __restrict__ lr_D, double* __restrict__ lrW, double* __restrict__ lrWS,
                const double* __restrict__ partials, int check_done,
                int lr_tail, int lr_n, const int* __restrict__ lr_mat_ptr, const int* __restrict__ lr_mat_gid,
                const double* __restrict__ red2, ExtraHead eh) {
  ls_solve_fast_body(c, m, gid_g, n_extra, extra, nb, A_RD, A_DD, lam, lam_ub, pv_raw, lb, pv, y, lr_ST, r, lr_col_gid, lr_D, lrW, lrWS, partials, check_done, lr_tail, lr_n, lr_mat_ptr, lr_mat_gid, red2, eh);
}
