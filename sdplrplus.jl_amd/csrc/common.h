// common.h — shared definitions of the gfx950 device backend (libsdplr_hip.so).
//
// Design (DESIGN.md has the long form):
//  * every factor-shaped array is n rows × r contiguous doubles (the reference's r×n column-major
//    `Rt`, src/structs.jl:195,236); slots live in one arena with a 256-B aligned stride.
//  * all O(1) solver scalars live in ONE device-resident control block (DevCtrl).  Kernels read and
//    write it directly; the host copies it back once per batch of inner iterations, so the
//    reference's per-iteration scalar traffic (α, ℒ, ‖G‖, ‖pv‖, dot) never crosses PCIe mid-loop.
//  * grid-wide reductions are deterministic: every block writes one partial per quantity, and the
//    (single-block or multi-block) consumer adds the partials in a fixed order.  No float atomics.
//  * wavefront = 64 lanes everywhere; a factor row is covered by a sub-wave group of LPR lanes,
//    VEC doubles per lane (r = 32 ⇒ 16 lanes × 16-B loads, 4 rows per wave instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SDPLR_HMAX 16          // largest numlbfgsvecs supported by the fused L-BFGS kernels
#define SDPLR_NT 256           // threads per block for every kernel (4 waves)
#define SDPLR_MAXNB 4096       // largest grid of a partial-producing kernel (width of a partial slot)
#define SDPLR_NSLOT 160        // reduction slots in the partials buffer
#define SDPLR_LRMAX 8          // low-rank columns handled per register pass

// exit reasons of the inner loop (include/sdplr_hip.h, sdplr_hip_inner_loop)
#define EXIT_GTOL 0
#define EXIT_RELDELTA 1
#define EXIT_ITERS 2
#define EXIT_TIME 3

// Device-resident control block: loop control + every O(1) scalar of SolverVars / LBFGSHistory.
struct DevCtrl {
  // ---- inner-loop control (src/sdplr.jl:190-278) ----
  int done;            // 1 ⇒ every later kernel of the batch returns immediately
  int exit_reason;
  int err;             // 0 or SDPLR_ERR_NOT_DESCENT (src/linesearch.jl:60-62)
  int use_armijo;
  long long iters;     // localiter
  long long max_iters;
  double cur_gtol, fprec_eps, normC, normb;
  int grel, prel;
  int reldelta_exit;   // set by the line-search scalar stage when rel_delta < fprec·eps (src/sdplr.jl:238-241)
  int norms_pending;   // ‖G‖² / ‖pv‖² partials of the current iteration wait to be folded (by the seam kernel)
  int nb_gnorm, nb_pvnorm;  // number of partials behind them
  // ---- SolverVars scalars (src/structs.jl:203-205) ----
  double sigma, obj;
  // ---- per-iteration scalars ----
  double L, lastval, gnorm, pvnorm, alpha, descent, alpha_max;
  double pv2_extra;    // ‖pv‖² share of slots committed by a scalar kernel (singleton fast path)
  // the cost slot's new value decided by the step kernel's line-search head (k_fast_step2<…, LSH>): every block of that
  // kernel READS obj, so block 0 leaves the new value here and the seam kernel that follows stores it (obj and pv_raw[m])
  double obj_next;
  int obj_pending, pad2_;
  double biquad[5];
  // ---- Lanczos (src/coreop.jl:461-500) ----
  int lz_done;
  long long lz_steps;
  double lz_beta_prev;
  double lz_gamma_cur, lz_gamma_prev;  // norms of the current / previous unnormalised Lanczos vectors
  long long lz_qmax;
  double lz_mineig;    // resident dual bound: min eigenvalue of the Lanczos tridiagonal, minus the shift (src/coreop.jl:502-513)
  // ---- L-BFGS (src/lbfgs.jl:4-28) ----
  int latest;          // 1-based, as in the reference
  int gram_pending;    // set by k_lbfgs_update, consumed by k_lbfgs_boundary
  int fallback;        // set by the seam kernel: the coming direction is not a descent direction (src/sdplr.jl:202)
  int pad_;
  double rho[SDPLR_HMAX], a[SDPLR_HMAX];
  double c_alpha[SDPLR_HMAX], c_gamma[SDPLR_HMAX];   // two-loop coefficients of the current direction
  double SY[SDPLR_HMAX * SDPLR_HMAX];                // SY[a][b] = ⟨s_a, y_b⟩
  double YY[SDPLR_HMAX * SDPLR_HMAX];                // YY[a][b] = ⟨y_a, y_b⟩
  double Sg[SDPLR_HMAX], Yg[SDPLR_HMAX];             // ⟨s_a, G⟩, ⟨y_a, G⟩ for the current G
  // ---- ring form of the history (k_dense.h, "ring form") ----
  int ring_on;         // the loop keeps G_k and D_k of the last h + 1 iterations instead of s_j, y_j
  int ring_k;          // ring position (0..h) of the current G; the coming direction goes to the same position of the D ring
  int ring_n;          // pairs in ring form (≤ h): the i-th newest is (D, G) at position ring_k − 1 − i and G at ring_k − i;
                       // the older ones are still the stored s_j, y_j the loop found in their slots
  int ring_unc;        // the last step was not followed by lbfgs_update! (relative-decrease exit): its G_new sits at ring_k + 1
  int ring_j0;         // latest mod h when the ring was entered: position p ≥ 1 lives in history slot (ring_j0 + p − 1) mod h
  int ring_pad_;
  double ring_alpha[SDPLR_HMAX];                     // by pair slot: s_j = ring_alpha[j]·D
};

// ---- deterministic reductions --------------------------------------------------------------------
// A 16-lane DPP row summed by rotations of 8, 4, 2, 1 (v_mov_b32_dpp row_ror: VALU only, no trip over the LDS
// crossbar).  Lane i adds the same partners in the same order as an xor butterfly of the same widths (after the step
// of width o the partial sums have period o), so every lane of the row ends with the same value.
#define SDPLR_ROR(n)                                                                              \
  v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + n, 0xf, 0xf, true),   \
                        __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + n, 0xf, 0xf, true));
__device__ __forceinline__ double row_sum16(double v) {
  SDPLR_ROR(8) SDPLR_ROR(4) SDPLR_ROR(2) SDPLR_ROR(1)
  return v;
}
__device__ __forceinline__ double lane_value(double v, int l) {   // l: compile-time constant (v_readlane_b32)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Sum over the 64 lanes of a wave, all lanes active: the four row totals, then ((r0 + r1) + r2) + r3 — the same value
// in every lane, a fixed order.  (Six ds_bpermute round trips per sum before: ten sums at the end of a kernel kept the
// LDS pipe busy for ≈ 3 k cycles — k_ls_solve_fast, in-kernel stamps.)
__device__ __forceinline__ double wave_sum(double v) {
  v = row_sum16(v);
  const double r0 = lane_value(v, 0), r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
  return ((r0 + r1) + r2) + r3;
}

// sum over a sub-wave group of G lanes (G power of two ≤ 64); every lane of the group gets it
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (G == 16) {
    return row_sum16(v);
  } else {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
}

// block-wide sums of K values; result valid in every thread.  sh must hold K*(NT/64) doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* sh) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  constexpr int NW = SDPLR_NT / 64;
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) sh[k * NW + w] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NW; i++) t += sh[k * NW + i];
    v[k] = t;
  }
}
__device__ __forceinline__ double block_sum1(double v, double* sh) {
  double a[1] = {v};
  block_sum<1>(a, sh);
  return a[0];
}

// sum of the nb per-block partials of one slot, same value and same order in every calling block
__device__ __forceinline__ double reduce_partials(const double* __restrict__ p, int nb, double* sh) {
  double t = 0.0;
  for (int i = threadIdx.x; i < nb; i += SDPLR_NT) t += p[i];
  return block_sum1(t, sh);
}

__device__ __forceinline__ const double* slot_partials(const double* base, int slot) {
  return base + (size_t)slot * SDPLR_MAXNB;
}
__device__ __forceinline__ double* slot_partials(double* base, int slot) {
  return base + (size_t)slot * SDPLR_MAXNB;
}

// reduction slots in the partials buffer
enum {
  SLOT_DESCENT = 0,   // ⟨dirt, G⟩
  SLOT_GNORM2 = 1,    // ‖G‖²
  SLOT_PVNORM2 = 2,   // ‖primal_vio‖²
  SLOT_F = 3,         // Σ (ỹ² − λ²)/(2σ)
  SLOT_LZ_A = 9,      // Lanczos v·Av
  SLOT_LZ_B = 10,     // Lanczos ‖Av‖²
  SLOT_V0 = 11,       // ‖v0‖²
  SLOT_DUALYB = 12,   // ⟨y[1:m], b⟩
  SLOT_LZ_N = 15,     // Lanczos ‖u‖² of the vector about to be multiplied
  SLOT_PD = 13,       // ⟨P, D⟩ (structured fast path)
  SLOT_DW = 14,       // ⟨D, W⟩
  SLOT_GRAM = 16,     // 16 .. 16+5*HMAX-1: partials of lbfgs_update (see k_dense.h)
  SLOT_ARMIJO = 96,   // 96 .. 96+53-1: ℒ(α_max/2^k), k = 0..50, ℒ(0), slope (see k_scalar.h)
  SLOT_LS = 150,      // 150..157: the eight sums behind the quartic's coefficients (see k_scalar.h)
};
static_assert(SLOT_GRAM + 5 * SDPLR_HMAX <= SLOT_ARMIJO, "slot overlap");
static_assert(SLOT_ARMIJO + 53 <= SLOT_LS && SLOT_LS + 8 <= SDPLR_NSLOT, "slot overflow");
