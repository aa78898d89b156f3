// k_resident.h — the RESIDENT route for small instances (BASELINE config 5: n = 800, rank 10 MaxCut, the
// reference's own batch protocol, exps/batch_test.txt:1-9).
//
// On an instance whose factor is a few tens of KB every kernel of the multi-launch routes is a launch seam: ten
// launches of ≈ 4.5 µs per inner iteration with 255 of 256 CUs idle.  Here ONE workgroup owns the instance for a whole
// call of the inner `while` (src/sdplr.jl:190-278): one launch runs up to max_iters iterations, every exit test is
// taken on the device, and the host reads the control block once at the end.  A batch of instances is ONE grid with a
// workgroup per instance (k_rs_*_batch: block b runs row b of an argument table through the same body, rs_*_run, that the
// single-instance kernel runs; the scalars a single call moves through the control block ride the table rows) — 64
// instances occupy 64 CUs at the same time from one stream (sdplr_hip_batch_*), or a set of independent single-CU
// launches on the instances' own streams (threaded drivers).
//
//   * the direction D lives in LDS (n·r doubles — the SpMM W = A_g·D gathers ≈ 48 rows of it per row at LDS speed);
//     R, G, P, W and the 2h history arrays stay in global memory, which for ≤ 1 MB of state means the XCD's L2;
//   * the SpMM runs one ROW PER LANE over a sliced-ELL copy of A_g (RsEll: 64 rows per slice, entry k of the 64 rows in
//     one 256-byte line, 4 bytes per entry): per entry a lane issues one coalesced load, the LDS reads of the gathered
//     row and the multiply-adds — no index hand-offs, no row pointers.  (The first version gave a row to 8 lanes, as
//     the multi-launch kernels do: every entry then cost a wave ≈ 15 instructions for 8 entries and a dependent global
//     round trip per 32 — 137 µs of a 171 µs iteration on Gset G1.)  The lane that owns a row also forms that row's
//     line-search dots while W_j is in its registers; W goes out row-major and STEP streams it element by element;
//   * per-row quantities (⟨R_j,D_j⟩, ‖D_j‖², the diagonal coefficient d_j of S) are LDS vectors; the data of the
//     one singleton constraint attached to a row (λ, λ_ub, lb, primal_vio_raw, value) sit in the registers of the
//     thread that owns the row for the whole call;
//   * the control block is staged in LDS; thread 0 runs the same serial code as the seam kernel (k_dense.h:
//     seam_serial[_small]) and the scalar stage of the exact line search (k_scalar.h: quartic_argmin);
//   * all sums are formed in a fixed order (lane → wave → block): two runs are bit-identical.
//
// Applies to the singleton form of the structured fast path (k_sparse.h) without low-rank matrices and with at most one
// singleton constraint per row (MaxCut, CutNorm): S(y) = y_g·A_g + Diag(d(y)).  Phases of one iteration, separated by
// workgroup barriers (reference lines as in DESIGN.md §1):
//   SEAM   fold Gram sums / norms, loop tests, two-loop coefficients, descent test        lbfgs.jl:93-113, sdplr.jl:190,201-202,224-241,272-277
//   DIR    D = ∓(G − Σ α_l y_l + Σ γ_l s_l)  → LDS                                         lbfgs.jl:77-120, sdplr.jl:203-204
//   SPMM   W = A_g·D, ⟨R_j,D_j⟩, ‖D_j‖², partials of ⟨R,W⟩, ⟨D,W⟩                           coreop.jl:153-203, linesearch.jl:10-16
//   LSSUM  the eight line-search sums over the row-attached constraints                   linesearch.jl:36-56
//   SOLVE  quartic, α*, ℒ(α*), relative-decrease test, commit of A_g's slot               linesearch.jl:58-124, sdplr.jl:238
//   COMMIT primal_vio_raw, primal_vio, y of the row-attached constraints, d_j              linesearch.jl:118-124, coreop.jl:229-236
//   STEP   R += αD, P += αW, G = 2(y_g·P + d∘R), ‖G‖², lbfgs_update! with its Gram dots      sdplr.jl:219-234, lbfgs.jl:129-149
//
// k_rs_lanczos is the same idea for approx_mineigval_lanczos (src/coreop.jl:461-500): the three Lanczos vectors live
// in LDS, one launch runs all q steps on the assembled S.
#pragma once
#include <type_traits>

#include "common.h"
#include "k_dense.h"
#include "k_scalar.h"
#include "k_sparse.h"

#ifndef SDPLR_RS_NT
#define SDPLR_RS_NT 512        /* threads of the resident workgroup (8 waves: 256 VGPRs per lane) */
#endif
#define SDPLR_RS_NW (SDPLR_RS_NT / 64)
#define SDPLR_RS_RPT 2         /* rows whose constraint data a thread keeps in registers (n ≤ RPT·NT) */
#ifndef SDPLR_RS_EARLY_MAX
#define SDPLR_RS_EARLY_MAX 8   /* the SpMM asks for R_j before its gather loop when the row is at most this many doubles wide */
#endif
#ifndef SDPLR_RS_NCHMAX
#define SDPLR_RS_NCHMAX 8      /* most VEC-wide chunks of a row a lane carries through one pass of the SpMM */
#endif
#ifndef SDPLR_RS_ELL_PF
#define SDPLR_RS_ELL_PF 8      /* entries of a row in flight per lane in the SpMM (a register ring) */
#endif


// doubles of the LDS copy of a factor-shaped array: the n rows and one more, row n, that stays zero — the SpMM sends the
// lanes of rows shorter than their slice's longest there (rounded to an even count: 16-byte alignment of what follows)
__host__ __device__ inline long long rs_npad(long long n, long long r) { return (n * r + r + 1) & ~1LL; }

// sums of K values over the workgroup, delivered to thread 0 only (fixed order: lanes by DPP, then waves 0..NW−1)
template <int K>
__device__ __forceinline__ void rs_sum_to0(double (&v)[K], double* sh) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) sh[k * SDPLR_RS_NW + w] = v[k];
  }
  __syncthreads();
  // (lane k of wave 0 adds the NW wave partials of sum k — waves 0..NW−1 in that order, as thread 0 used to for all K sums one
  // LDS read after the other: ≈ 180 dependent reads, 11 k cycles for the 22 sums of STEP — and hands its total to lane 0)
  if (w == 0) {
    double t = 0.0;
    if (l < K) {
#pragma unroll
      for (int i = 0; i < SDPLR_RS_NW; i++) t += sh[l * SDPLR_RS_NW + i];
    }
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = __shfl(t, k, 64);
  }
}

// A_g for the resident route: sliced ELL, one row per lane.  Rows are taken in order of decreasing off-diagonal count
// (perm), 64 rows per slice; entry k of a slice's 64 rows is one contiguous 256-byte line: the k-th off-diagonal entry
// of each row in ascending column order: its column, and its value in `val` unless all off-diagonal entries of A_g share
// ONE value (unit-weight graphs: 4 bytes per entry instead of 12).  The diagonal of A_g is a vector of its own.
// (A palette of ≤ 255 values looked up in LDS per entry was tried for the weighted case: the look-ups in front of the
// gathers made the compiler keep every entry of the register ring's trip in flight — 256 VGPRs and scratch.)
struct RsEll {
  int n_slices;
  const int* perm;        // [64·n_slices] row of (slice, lane), −1 past the end
  const int* len;         // [64·n_slices] its number of off-diagonal entries
  const int* sptr;        // [n_slices + 1] first 64-entry line of each slice
  const unsigned* ent;    // [64·sptr[n_slices]]
  const double* val;      // same shape, or null: every off-diagonal entry has the value `one`
  double one;
  const double* gdiag;    // [n]
  int sl0 = 0, sl_step = 1;   // the slices this workgroup takes: sl0, sl0 + sl_step, … (a team member: its rank, the team's size)
};

// NCH chunks (VEC doubles each, from chunk c0) of Y_j = Σ_k a_jk·X_k for the 64 rows of one slice, X in LDS (row-major):
// the terms of a row are summed by increasing column, the diagonal last.  MODE 0: one off-diagonal value (the rows of X
// are summed, the sum scaled once), 2: value array.  Lanes whose row is shorter than the slice's longest gather the zero
// row n of X (MODE 0) or multiply row 0 by zero.
template <int VEC, int MODE, int NCH>
__device__ __forceinline__ void rs_ell_pass(const unsigned* __restrict__ ep, const double* __restrict__ vp, int width, int len,
                                            double one, const double* Xl, int r, int c0, int j, double gd, int zrow,
                                            vecd<VEC> (&w)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int q = 0; q < VEC; q++) w[c].v[q] = 0.0;
  // A lane's entries come in through a register ring of PF lines: the loads of the next PF entries are issued before
  // the current PF are consumed (the compiler does not pipeline the loop by itself: with two entries per trip the
  // SpMM of Gset G1 waited 215 k cycles per iteration on ≈ 40 dependent round trips per wave).
  constexpr int PF = SDPLR_RS_ELL_PF;
  unsigned er[PF];
  double vr[MODE == 2 ? PF : 1];
#pragma unroll
  for (int q = 0; q < PF; q++) {
    const int kc = min(q, width - 1);
    er[q] = ep[(size_t)kc * 64];
    if (MODE == 2) vr[q] = vp[(size_t)kc * 64];
  }
#pragma nounroll
  for (int k0 = 0; k0 < width; k0 += PF) {
    unsigned ec[PF];
    double vc[MODE == 2 ? PF : 1];
#pragma unroll
    for (int q = 0; q < PF; q++) {
      ec[q] = er[q];
      if (MODE == 2) vc[q] = vr[q];
    }
#pragma unroll
    for (int q = 0; q < PF; q++) {   // (clamped: the last trip re-reads the slice's last line)
      const int kc = min(k0 + PF + q, width - 1);
      er[q] = ep[(size_t)kc * 64];
      if (MODE == 2) vr[q] = vp[(size_t)kc * 64];
    }
#pragma unroll
    for (int q = 0; q < PF; q++) {
      const int k = k0 + q;
      const bool on = k < len;     // (len ≤ width: also false past the slice's end)
      const unsigned e = ec[q];
      // MODE 0 (one value for every off-diagonal entry): the gathered rows are SUMMED and the sum is scaled once, after
      // the loop — half the FP64 instructions of the multiply-add form; a lane past its row's end gathers the zero row
      const int col = on ? (int)(e & 0xFFFFu) : (MODE == 0 ? zrow : 0);
      double v = 0.0;
      if (MODE != 0) v = on ? vc[q] : 0.0;
      const double* xr = Xl + (long long)col * r + c0 * VEC;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const vecd<VEC> x = ldrow<VEC>(xr + c * VEC);
#pragma unroll
        for (int qq = 0; qq < VEC; qq++) {
          if (MODE == 0) w[c].v[qq] += x.v[qq];
          else w[c].v[qq] += x.v[qq] * v;
        }
      }
      // (entries are consumed two at a time: left to itself the scheduler hoists the LDS reads of all PF entries —
      // PF·NCH·VEC doubles of registers — and the kernel spills)
      if (q & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const double* xj = Xl + (long long)j * r + c0 * VEC;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const vecd<VEC> x = ldrow<VEC>(xj + c * VEC);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      if (MODE == 0) w[c].v[q] *= one;
      w[c].v[q] += x.v[q] * gd;
    }
  }
}
// Y = X·A_g for all rows, chunks [c0, c0 + NCH): one slice per wave and trip; Y row-major like every factor-shaped array
// (a lane stores its row's chunks: 16-byte pieces of 64 different rows per store instruction — 64 KB per SpMM, and the
// consumers stream Y element by element).
// DOTS (the loop's W = A_g·D): the lane that owns row j also forms that row's four line-search dots while W_j is in its
// registers — ⟨R_j,D_j⟩ and ‖D_j‖² to the LDS vectors rdl / ddl, ⟨R_j,W_j⟩ and ⟨D_j,W_j⟩ into the lane's running sums rw / dw
// (R_j is requested before the gather loop and arrives under it; D_j is in LDS) — so that no pass re-reads W and R for them.
// one rank-one matrix D·b·bᵀ among the constraints (MinBisection's 1ᵀX1 = 0, test/problem.jl:78-94): the only low-rank
// structure the resident route takes.  Its products are r-vectors: w0 = Rᵀb, w1 = Dᵀb (src/coreop.jl:115-151), its share of
// the gradient 2·b_j·(y_c·D·w0) (src/structs.jl:117-145).
struct RsLr {
  const double* B;   // [n] the column b; null: no low-rank matrix
  double D;
  int gid;           // its slot in the (m+1)-vectors (a constraint: gid < m)
};
// out[c] = Σ_j b_j·X[j][c] for the r columns of a factor-shaped LDS array: one column per wave and trip, the lanes stride the
// rows (fixed order: lane partials, then the DPP wave sum) — a barrier must separate it from the writes of X and the reads of out
__device__ __forceinline__ void rs_colsum(const double* Xl, const double* __restrict__ b, int n, int r, double* out) {
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  for (int c = wave; c < r; c += SDPLR_RS_NW) {
    double t = 0.0;
    for (int j = wl; j < n; j += 64) t += b[j] * Xl[(long long)j * r + c];
    t = wave_sum(t);
    if (wl == 0) out[c] = t;
  }
}
struct RsDots {
  const double* R;
  double *rdl, *ddl;
};
template <int VEC, int MODE, int NCH, bool DOTS>
__device__ __forceinline__ void rs_ell_spmm_chunks(const RsEll& E, const double* Xl, int n, int r, int c0, double* out,
                                                   const RsDots& dots, double& rw, double& dw) {
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  const double one = E.one;
#pragma nounroll
  for (int sl = E.sl0 + wave * E.sl_step; sl < E.n_slices; sl += SDPLR_RS_NW * E.sl_step) {
    const int idx = sl * 64 + wl;
    const int jp = E.perm[idx];
    const int len = E.len[idx];
    const int l0 = E.sptr[sl], width = E.sptr[sl + 1] - l0;
    const unsigned* ep = E.ent + (size_t)l0 * 64 + wl;
    const double* vp = (MODE == 2) ? E.val + (size_t)l0 * 64 + wl : nullptr;
    const int j = max(jp, 0);
    const double gd = (jp >= 0) ? E.gdiag[j] : 0.0;
    // (R_j is asked for before the gather loop where the registers allow it — up to five chunks: with eight, the 32
    // registers it would hold through the loop spill — and after it otherwise)
    constexpr bool EARLY = DOTS && NCH * VEC <= SDPLR_RS_EARLY_MAX;
    vecd<VEC> xr[DOTS ? NCH : 1];
    if (EARLY) {
#pragma unroll
      for (int c = 0; c < NCH; c++) xr[c] = ldrow_nt<VEC>(dots.R + (long long)j * r + (c0 + c) * VEC);   // (past L1: in a team the row may be a team-mate's)
    }
    vecd<VEC> w[NCH];
    rs_ell_pass<VEC, MODE, NCH>(ep, vp, width, len, one, Xl, r, c0, j, gd, n, w);
    if (DOTS && !EARLY) {
#pragma unroll
      for (int c = 0; c < NCH; c++) xr[c] = ldrow_nt<VEC>(dots.R + (long long)j * r + (c0 + c) * VEC);
    }
    if (jp >= 0) {
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        strow<VEC>(out + (long long)j * r + (c0 + c) * VEC, w[c]);
      }
      if (DOTS) {
        double rd = 0.0, dd = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const vecd<VEC> xd = ldrow<VEC>(Xl + (long long)j * r + (c0 + c) * VEC);
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            rd += xr[c].v[q] * xd.v[q];
            dd += xd.v[q] * xd.v[q];
            rw += xr[c].v[q] * w[c].v[q];
            dw += xd.v[q] * w[c].v[q];
          }
        }
        // (a row is owned by the same lane in every pass over the chunks: the passes after the first add to its sums)
        dots.rdl[j] = (c0 == 0 ? 0.0 : dots.rdl[j]) + rd;
        dots.ddl[j] = (c0 == 0 ? 0.0 : dots.ddl[j]) + dd;
      }
    }
  }
}
// (the dispatch on the number of chunks sits OUTSIDE the slice loop: with the switch inside it the compiler merged the
// heads of all its cases and the kernel needed 256 VGPRs and scratch; each case by itself takes 56–101)
template <int VEC, int MODE, bool DOTS>
__device__ __forceinline__ void rs_ell_spmm(const RsEll& E, const double* Xl, int n, int r, double* out, const RsDots& dots,
                                            double& rw, double& dw) {
  const int NC = r / VEC;   // (VEC = 2 only for even r)
  constexpr int CMAX = SDPLR_RS_NCHMAX;
#pragma nounroll
  for (int c0 = 0; c0 < NC; c0 += CMAX) {
    switch (min(NC - c0, CMAX)) {
      case 1: rs_ell_spmm_chunks<VEC, MODE, 1, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
      case 2: rs_ell_spmm_chunks<VEC, MODE, 2, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
      case 3: rs_ell_spmm_chunks<VEC, MODE, 3, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
      case 4: rs_ell_spmm_chunks<VEC, MODE, 4, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
#if SDPLR_RS_NCHMAX > 4
      case 5: rs_ell_spmm_chunks<VEC, MODE, 5, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
#endif
#if SDPLR_RS_NCHMAX > 5
      case 6: rs_ell_spmm_chunks<VEC, MODE, 6, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
#endif
#if SDPLR_RS_NCHMAX > 6
      case 7: rs_ell_spmm_chunks<VEC, MODE, 7, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
#endif
#if SDPLR_RS_NCHMAX > 7
      case 8: rs_ell_spmm_chunks<VEC, MODE, 8, DOTS>(E, Xl, n, r, c0, out, dots, rw, dw); break;
#endif
      default: break;
    }
  }
}
template <int VEC>
__device__ __forceinline__ void rs_ell_spmm_any(const RsEll& E, const double* Xl, int n, int r, double* out) {
  const RsDots none{nullptr, nullptr, nullptr};
  double u0 = 0.0, u1 = 0.0;
  if (E.val == nullptr) rs_ell_spmm<VEC, 0, false>(E, Xl, n, r, out, none, u0, u1);
  else rs_ell_spmm<VEC, 2, false>(E, Xl, n, r, out, none, u0, u1);
}
// W = A_g·D with the row dots of the line search (RsDots)
template <int VEC>
__device__ __forceinline__ void rs_ell_spmm_dots(const RsEll& E, const double* Xl, int n, int r, double* out, const RsDots& dots,
                                                 double& rw, double& dw) {
  if (E.val == nullptr) rs_ell_spmm<VEC, 0, true>(E, Xl, n, r, out, dots, rw, dw);
  else rs_ell_spmm<VEC, 2, true>(E, Xl, n, r, out, dots, rw, dw);
}

// ---- fg! (src/coreop.jl:323-349) as one launch on the instances of the resident loop ----------------------------------
// f! (:11-31): primal_vio_raw = 𝒜(RRᵀ) − b with 𝒜 row-local — v_j·‖R_j‖² for the constraint attached to row j, ⟨R, P⟩ for
// A_g's slot, P = A_g·R by the ELL SpMM (left in place: the loop that follows starts from a fresh P) — then obj, the capped
// violation, ℒ; g! (:305-317): y, G = 2(y_g·P + d(y)∘R), ‖G‖; the two norms of :334-347.
struct RsFgArgs {
  int n, m, r;
  int gid_g;
  const int* row_k;
  const double* row_v;
  RsEll E;
  const double* R;
  double *G, *P;
  double *y, *pv_raw, *pv;
  const double *lam, *lam_ub, *lb, *b;
  DevCtrl* c;
  RsLr lr;
  double* rowvec;                       // null, or 3n doubles of global memory for the per-row vectors (see RsLoopArgs)
  // batched launches (sdplr_hip_batch_fg): the norm parameters ride the argument row instead of a pushed control block,
  // and the scalars the host reads back go to a row of a result table instead of a pulled one
  int in_set, in_grel, in_prel;
  double in_normC, in_normb;
  double* out;                          // [4]: ℒ, ‖G‖, ‖primal_vio‖, obj
};
// the body of fg!: every thread of the resident workgroup calls it; Rl[Npad] | rrl[n] | djl[n] are LDS work arrays, sred holds
// 3·NW doubles, c is the control block (global memory in k_rs_fg, the LDS copy in the loop's prologue)
struct RsFgShared {
  double yg, obj_row;
  int has_obj_row;
};
// lrl: LDS, 2r doubles — w0 = Rᵀb and ws = y_c·D·w0 of the rank-one matrix (left there for the loop that follows)
template <int VEC>
__device__ __forceinline__ void rs_fg_body(const RsFgArgs& a, DevCtrl& c, double* Rl, double* rrl, double* djl, double* sred,
                                           RsFgShared& sh, double* lrl) {
  constexpr int NT = SDPLR_RS_NT;
  const int tid = threadIdx.x;
  const int n = a.n, m = a.m, r = a.r;
  const long long N = (long long)n * r;
  const double sigma = c.sigma, normC = c.normC, normb = c.normb;
  const int grel = c.grel, prel = c.prel;
  if (tid == 0) sh.has_obj_row = 0;
  for (long long e = tid; e < N; e += NT) Rl[e] = a.R[e];
  __syncthreads();
  // P = A_g·R; the lane that owns row j forms ‖R_j‖² (→ rrl) and its part of ⟨R, P⟩ while P_j is in its registers
  double acc[3] = {0.0, 0.0, 0.0};   // ⟨R, P⟩
  {
    const RsDots dots{Rl, rrl, djl};   // (djl: scratch for the second, identical, row sum — rewritten below)
    double same = 0.0;
    rs_ell_spmm_dots<VEC>(a.E, Rl, n, r, a.P, dots, acc[0], same);
  }
  rs_sum_to0<1>(reinterpret_cast<double(&)[1]>(acc[0]), sred);
  const bool has_lr = a.lr.B != nullptr;
  if (has_lr) rs_colsum(Rl, a.lr.B, n, r, lrl);   // w0 = Rᵀb
  __syncthreads();
  double fs = 0.0, pn = 0.0;
  if (tid == 0 && has_lr) {   // the rank-one matrix's slot: 𝒜(RRᵀ)_c = D·‖Rᵀb‖²  (src/coreop.jl:115-120,132-139)
    const int kc = a.lr.gid;
    double q = 0.0;
    for (int ch = 0; ch < r; ch++) q += lrl[ch] * lrl[ch];
    double v = a.lr.D * q - a.b[kc];
    const double pc = fmax(v, a.lb[kc]);
    a.pv[kc] = pc;
    pn += pc * pc;
    const double l = a.lam[kc], yt = fmin(a.lam_ub[kc], l - sigma * v);
    fs += (yt * yt - l * l) / (2 * sigma);
    a.pv_raw[kc] = v;
    a.y[kc] = -yt;
    for (int ch = 0; ch < r; ch++) lrl[r + ch] = -yt * a.lr.D * lrl[ch];   // ws = y_c·D·w0
  }
  if (tid == 0) {   // A_g's slot
    const int kg = a.gid_g;
    double v = acc[0];
    double yk = 1.0;
    if (kg < m) {
      v -= a.b[kg];                                          // (:20)
      const double pc = fmax(v, a.lb[kg]);                   // (:22)
      a.pv[kg] = pc;
      pn += pc * pc;
      const double l = a.lam[kg], yt = fmin(a.lam_ub[kg], l - sigma * v);   // (:27)
      fs += (yt * yt - l * l) / (2 * sigma);                 // (:28)
      yk = -yt;                                              // src/coreop.jl:233
    }
    a.pv_raw[kg] = v;
    a.y[kg] = yk;
    sh.yg = yk;
  }
  for (int j = tid; j < n; j += NT) {
    const int k = a.row_k[j];
    double dj = 0.0;
    if (k >= 0) {
      const double rv = a.row_v[j];
      double v = rv * rrl[j];
      double yk = 1.0;
      if (k < m) {
        v -= a.b[k];
        const double pc = fmax(v, a.lb[k]);
        a.pv[k] = pc;
        pn += pc * pc;
        const double l = a.lam[k], yt = fmin(a.lam_ub[k], l - sigma * v);
        fs += (yt * yt - l * l) / (2 * sigma);
        yk = -yt;
      } else {            // the cost matrix as a row-attached entry
        sh.obj_row = v;
        sh.has_obj_row = 1;
      }
      a.pv_raw[k] = v;
      a.y[k] = yk;
      dj = rv * yk;
    }
    djl[j] = dj;
  }
  double two[2] = {fs, pn};
  rs_sum_to0<2>(two, sred);
  __syncthreads();
  const double yg = sh.yg;
  double gn = 0.0;
  {   // element by element in units of VEC doubles; the row of a unit — for d_j — advances with it
    const long long U = N / VEC;
    const int adv = NT * VEC, adv_q = adv / r, adv_r = adv % r;
    int j = (tid * VEC) / r, ch = (tid * VEC) % r;
#pragma nounroll
    for (long long u = tid; u < U; u += NT) {
      const long long e = u * VEC;
      const vecd<VEC> x = ldrow<VEC>(Rl + e), p = ldrow<VEC>(a.P + e);
      const double dj = djl[j];
      const double bj = has_lr ? a.lr.B[j] : 0.0;
      const int ch0 = ch;
      ch += adv_r;
      j += adv_q;
      if (ch >= r) { ch -= r; j++; }
      vecd<VEC> g;
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        g.v[k] = p.v[k] * yg + x.v[k] * dj;
        if (has_lr) g.v[k] += lrl[r + ch0 + k] * bj;          // + Σ_c WS[c]·B[c]  (src/coreop.jl:271-278)
        g.v[k] *= 2.0;                                        // src/coreop.jl:315
        gn += g.v[k] * g.v[k];
      }
      strow<VEC>(a.G + e, g);
    }
  }
  double one[1] = {gn};
  rs_sum_to0<1>(one, sred);
  if (tid == 0) {
    const double obj = (a.gid_g == m) ? acc[0] : (sh.has_obj_row ? sh.obj_row : a.pv_raw[m]);
    c.obj = obj;                                              // (:16)
    c.L = obj + two[0];                                       // (:25-30)
    const double g2 = sqrt(one[0]), p2 = sqrt(two[1]);
    c.gnorm = grel ? g2 / normC : g2;                         // (:334-338)
    c.pvnorm = prel ? p2 / normb : p2;                        // (:340-347)
  }
  __syncthreads();
}
template <int VEC>
__device__ __forceinline__ void rs_fg_run(const RsFgArgs& a) {
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];
  __shared__ double sred[3 * SDPLR_RS_NW];
  __shared__ RsFgShared sh;
  const long long N = (long long)a.n * a.r;
  const long long Npad = rs_npad(a.n, a.r);
  for (long long e = N + threadIdx.x; e < Npad; e += SDPLR_RS_NT) rs_lds[e] = 0.0;   // (the SpMM's zero row)
  if (a.in_set) {
    if (threadIdx.x == 0) {   // (set_norm_params of the single-instance entry point)
      a.c->normC = a.in_normC; a.c->normb = a.in_normb; a.c->grel = a.in_grel; a.c->prel = a.in_prel;
      a.c->done = 0; a.c->err = 0;
    }
    __syncthreads();
  }
  if (a.rowvec != nullptr) rs_fg_body<VEC>(a, *a.c, rs_lds, a.rowvec, a.rowvec + a.n, sred, sh, rs_lds + Npad);
  else rs_fg_body<VEC>(a, *a.c, rs_lds, rs_lds + Npad, rs_lds + Npad + a.n, sred, sh, rs_lds + Npad + 2 * a.n);
  if (a.out != nullptr && threadIdx.x == 0) {
    a.out[0] = a.c->L; a.out[1] = a.c->gnorm; a.out[2] = a.c->pvnorm; a.out[3] = a.c->obj;
  }
}
template <int VEC>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_fg(RsFgArgs a) { rs_fg_run<VEC>(a); }
// one workgroup per instance: block b takes row b of the argument table
template <int VEC>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_fg_batch(const RsFgArgs* __restrict__ items) {
  const RsFgArgs a = items[blockIdx.x];
  rs_fg_run<VEC>(a);
}

struct RsLoopArgs {
  int n, m, r, h;
  int gid_g;                            // slot of A_g in the (m+1)-vectors
  const int* row_k;                     // [n] the one singleton constraint attached to row j (−1: none) …
  const double* row_v;                  // [n] … and its value
  RsEll E;                              // A_g
  FactorArena A;                        // R, G, D, s_*, y_*
  double *P, *W;
  double *y, *pv_raw, *pv, *A_RD, *A_DD;
  const double *lam, *lam_ub, *lb;
  DevCtrl* c;
  RsLr lr;
  // The per-row vectors (⟨R_j,D_j⟩, ‖D_j‖², d_j: 3n doubles) live in LDS next to the direction — or, when the direction
  // alone fills the CU's LDS (CutNorm on a Gset graph: n = 1600, rank 10: 128 KB), in this global scratch: each is
  // written once and read once per phase by other threads of the same workgroup, behind its barriers (through the L2,
  // like W and the 𝒜 values).
  double* rowvec;
  int refresh_P;                        // P = A_g·R from scratch before the first iteration
  // the head of a major iteration in the same launch (sdplr_hip_major_iteration): λ update (src/sdplr.jl:358-362),
  // lbfgs_clear! (:384, src/lbfgs.jl:52-59), fg! (:389) — then the while loop (:190-278) on what fg! returned
  int pre_lambda, pre_clear, pre_fg;
  const double* b;
  double* lam_rw;                       // λ, writable (pre_lambda)
  long long budget_ticks;               // wall_clock64() ticks this call may run (≤ 0: no limit), src/sdplr.jl:272-277
  // batched launches (sdplr_hip_batch_major_iteration): what the host otherwise writes into the control block before the
  // launch rides the argument row, and what it reads back afterwards goes to a row of a result table
  int in_set, in_grel, in_prel;
  double in_sigma, in_gtol, in_fprec, in_normC, in_normb;
  long long in_max_iters;
  double* out;                          // [8]: ℒ, ‖G‖, ‖primal_vio‖, α, obj, iters, exit_reason, err
  // TEAM (rs_loop_run<…, TEAM>): team_w workgroups of ONE XCD share the instance; this one is team_rank.  xch: the team's
  // exchange block in global memory — [team_w][32] Gram / norm partials | [team_w][16] line-search partials | as unsigned:
  // arrival counter, failure flag, the members' XCC ids (zeroed by the host before the launch).
  int team_w, team_rank;
  double* xch;
  int team_test_fail;                   // tests: the placement check answers "not one XCD" (SDPLR_HIP_TEAM_TEST_FAIL)
};
#define SDPLR_RS_TEAM_MAX 4
#define SDPLR_RS_XCH_DOUBLES (48 * SDPLR_RS_TEAM_MAX + 8)
#define SDPLR_ERR_TEAM_PLACEMENT (-101)   /* the members did not land on one XCD, or not all of them arrived: nothing was touched, the host relaunches without a team */
#define SDPLR_ERR_TEAM_TIMEOUT (-102)     /* a member never arrived at a team barrier (the grid was not co-resident) */

// Team barrier (k_rs_loop<…, TEAM>): every wave's stores are complete (s_waitcnt + the workgroup barrier) before ONE lane adds to
// the team's counter in L2; the poll is an agent-scope atomic load.  What a member then reads of its team-mates' data it reads
// past its L1 (non-temporal loads) — the members sit on one XCD (checked at entry: HW_REG_XCC_ID) and share its L2, so no
// L2 write-back / invalidate (≈ 3.5 µs per member and barrier) is needed.  Every spin is bounded.
struct RsTeamSync {
  unsigned* ctr;
  unsigned target;
  int W;
};
__device__ __forceinline__ bool rs_team_barrier(RsTeamSync& ts, int* sh_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ts.target += (unsigned)ts.W;
  if (threadIdx.x == 0) {
    // (the add returns the count: the last member to arrive does not poll at all; bit 31 of the counter is the failure flag)
    unsigned v = __hip_atomic_fetch_add(ts.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    long long spins = 0;
    while (!(v & 0x80000000u) && (v & 0x7FFFFFFFu) < ts.target) {
      __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(ts.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > (1LL << 20)) {   // (≈ 1 s)
        (void)__hip_atomic_fetch_or(ts.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v |= 0x80000000u;
      }
    }
    *sh_ok = (v & 0x80000000u) ? 0 : 1;
  }
  __syncthreads();
  return *sh_ok != 0;
}
__device__ __forceinline__ double rs_ld_nt(const double* p) { return __builtin_nontemporal_load(p); }

// constraint data of the rows a thread owns (row j = tid + q·NT): registers for q < RPT, global memory beyond
// (λ_ub and the lower bound of the violation — ±∞ on equality constraints — are read where COMMIT needs them, once per
// iteration from L2: eight registers per lane less across every phase of the loop, which is what kept the even-rank
// shape at 256 VGPRs + scratch)
struct RsRow {
  int k;
  double v, lam, pvr;
};
__device__ __forceinline__ RsRow rs_load_row(const RsLoopArgs& a, int j) {
  RsRow o;
  o.k = a.row_k[j];
  o.v = a.row_v[j];
  const int kc = (o.k >= 0 && o.k < a.m) ? o.k : 0;
  const bool has = o.k >= 0 && o.k < a.m && a.m > 0;
  o.lam = has ? a.lam[kc] : 0.0;
  o.pvr = (o.k >= 0) ? a.pv_raw[o.k] : 0.0;
  return o;
}

// PDROP (A_g is the cost matrix: y_g ≡ 1; the caller vouches that G is the gradient at the device's state, or has fg! run
// in the prologue): STEP carries G forward — G_new = G_old + 2(αW + d_new∘R_new − d_old∘R_old), as k_fast_step2<…, PDROP>
// does — and P is neither read nor written: two of STEP's seventeen streams gone, and no P = A_g·R at the loop's entry.
// TEAM (with PDROP): team_w workgroups run this loop on ONE instance.  Each member forms the whole
// direction in its own LDS (DIR is a ninth of the iteration and needs every row anyway) and owns the rows [row_lo, row_hi) in
// everything else: its slices of the sliced ELL hold exactly those rows (build_rs_ell cuts the matrix for the team), so the
// SpMM's rows of W and their dots, the line-search sums, the commit and STEP never cross members.  Two exchanges per
// iteration through the team's block in global memory, each behind a team barrier: the Gram / norm partials of STEP (→ SEAM,
// made by every member for itself: same sums, same order, the same control block in every member — and the barrier is also
// what makes every member's rows of G and of the history visible to DIR), and the ten line-search sums (→ SOLVE, again by
// every member).  Rank 0 alone runs the prologue, stores the scalars of the extra slot and the control block, and leaves dirt.
template <int VEC, int HM, bool PDROP, bool TEAM = false>
__device__ __forceinline__ void rs_loop_run(const RsLoopArgs& a) {
  static_assert(!TEAM || PDROP, "teams run the P-less loop");
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];   // (16-byte LDS reads: an 8-byte-aligned base behind the static LDS made every ds_read_b128 a misaligned access — 5× slower)
  __shared__ SeamLds gd;
  __shared__ double sred[(5 * HM + 2 > 10 ? 5 * HM + 2 : 10) * SDPLR_RS_NW];
  __shared__ double sh_alpha, sh_yg, sh_p1, sh_p2;
  __shared__ int sh_err, sh_upd;
  static_assert(sizeof(SeamLds) + sizeof(double) * ((5 * HM + 2 > 10 ? 5 * HM + 2 : 10) * SDPLR_RS_NW + 8) <= 10 * 1024,
                "static LDS of the resident loop: the host budgets 150 KB of dynamic LDS next to it");
  constexpr int NT = SDPLR_RS_NT;
  const int tid0 = threadIdx.x;
  int tid = tid0;
  const int n = a.n, m = a.m, r = a.r, h = a.h;
  const long long N = (long long)n * r;
  const long long Npad = rs_npad(n, r);
  // this workgroup's share of the rows (everything, without a team)
  const int TW = TEAM ? a.team_w : 1, trank = TEAM ? a.team_rank : 0;
  const int rpm = (n + TW - 1) / TW;
  const int row_lo = TEAM ? min(n, trank * rpm) : 0, row_hi = TEAM ? min(n, row_lo + rpm) : n;
  __shared__ int sh_team_ok;
  __shared__ double sh_pvg, sh_lrpv, sh_tflag;
  double* const xg = a.xch;                                   // [TW][32]
  double* const xl = TEAM ? a.xch + 32 * TW : nullptr;        // [TW][16]
  unsigned* const xi = TEAM ? reinterpret_cast<unsigned*>(a.xch + 48 * TW) : nullptr;
  RsTeamSync ts{xi, 0u, TW};
  double* Dl = rs_lds;                 // [Npad] the direction (or R while P is being refreshed), then the zero row
  for (long long e = N + threadIdx.x; e < Npad; e += SDPLR_RS_NT) Dl[e] = 0.0;
  const bool rows_global = a.rowvec != nullptr;
  double* rdl = rows_global ? a.rowvec : Dl + Npad;   // [n] ⟨R_j, D_j⟩
  double* ddl = rdl + n;               // [n] ‖D_j‖²
  double* djl = ddl + n;               // [n] d_j = v_j·y[k_j]
  double* const w0l = rows_global ? Dl + Npad : Dl + Npad + 3 * (long long)n;   // [r] Rᵀb of the rank-one matrix, [r] y_c·D·w0 (same order as rs_fg_body leaves them) …
  double* const wsl = w0l + r;
  double* const w1l = wsl + r;         // … [r] Dᵀb, [r] the previous y_c·D·w0 (PDROP)
  double* const wsol = w1l + r;
  const bool has_lr = a.lr.B != nullptr;
  double* const R = aslot(a.A, AS_R);
  double* const Gm = aslot(a.A, AS_G);
  // the control block → LDS (one coalesced load)
  for (int t = tid; t < (int)(sizeof(DevCtrl) / 8); t += NT)
    reinterpret_cast<unsigned long long*>(&gd.c)[t] = reinterpret_cast<const unsigned long long*>(a.c)[t];
  const long long t_start = (long long)wall_clock64();
  __syncthreads();
  // σ as the block holds it on entry: the λ update of the prologue runs BEFORE var.σ[] is written (src/sdplr.jl:358-362
  // come before :366-369), so it must not see the σ the argument row brings
  const double sigma_on_entry = gd.c.sigma;
  if constexpr (TEAM) {   // all members on one XCD?  (placement is the dispatcher's: observed round-robin, never promised)
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      __hip_atomic_store(xi + 2 + trank, (xcc & 0xFu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (a member that never arrives HERE — the grid is not co-resident: the GPU is oversubscribed — is a placement failure
    // like any other: nothing has been touched yet.  Past this barrier every member is resident for good.)
    bool ok = rs_team_barrier(ts, &sh_team_ok);
    int code = ok ? 0 : SDPLR_ERR_TEAM_PLACEMENT;
    if (ok) {
      __syncthreads();
      if (tid == 0) {
        const unsigned x0 = __hip_atomic_load(xi + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        for (int k = 1; k < TW; k++) same &= (__hip_atomic_load(xi + 2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == x0) ? 1 : 0;
        sh_team_ok = a.team_test_fail ? 0 : same;
      }
      __syncthreads();
      if (!sh_team_ok) code = SDPLR_ERR_TEAM_PLACEMENT;
    }
    if (code != 0) {   // nothing has been touched: the host relaunches the instance without a team
      if (trank == 0 && tid == 0) {
        a.c->err = code;
        if (a.out != nullptr) a.out[7] = (double)code;
      }
      return;
    }
  }
  if (!TEAM || trank == 0) {   // (rank 0 runs the head of the major iteration alone)
  if (a.in_set) {   // (sdplr_hip_major_iteration: var.σ[], the loop's parameters and counters)
    __syncthreads();
    if (tid == 0) {
      DevCtrl& c = gd.c;
      c.sigma = a.in_sigma;
      c.done = 0; c.exit_reason = 0; c.err = 0; c.use_armijo = 0;
      c.iters = 0; c.max_iters = a.in_max_iters; c.reldelta_exit = 0; c.norms_pending = 0; c.pv2_extra = 0.0;
      c.cur_gtol = a.in_gtol; c.fprec_eps = a.in_fprec; c.normC = a.in_normC; c.normb = a.in_normb;
      c.grel = a.in_grel; c.prel = a.in_prel;
      c.alpha = 0.0; c.alpha_max = 1.0;
    }
  }
  if (a.pre_lambda || a.pre_clear || a.pre_fg) {
    __shared__ RsFgShared fsh;
    __syncthreads();
    if (a.pre_lambda) {      // λᵢ ← min(λ_ubᵢ, λᵢ − σ·primal_vio_rawᵢ)  (src/sdplr.jl:358-362)
      const double sigma = sigma_on_entry;
      for (int k = tid; k < m; k += NT) a.lam_rw[k] = fmin(a.lam_ub[k], a.lam_rw[k] - sigma * a.pv_raw[k]);
    }
    if (a.pre_clear) {       // lbfgs_clear!: s, y ← 0 (neighbours in the arena), ρ = a = 0, and with them the Gram data
      double* const hist = aslot(a.A, AS_S0);
      const long long len = 2LL * h * a.A.stride;
      for (long long e = tid; e < len; e += NT) hist[e] = 0.0;
      for (int k = tid; k < SDPLR_HMAX * SDPLR_HMAX; k += NT) {
        gd.c.SY[k] = 0.0;
        gd.c.YY[k] = 0.0;
        if (k < SDPLR_HMAX) gd.c.rho[k] = gd.c.a[k] = gd.c.c_alpha[k] = gd.c.c_gamma[k] = gd.c.Sg[k] = gd.c.Yg[k] = 0.0;
      }
    }
    __syncthreads();
    if (a.pre_fg) {
      RsFgArgs f{};
      f.n = n; f.m = m; f.r = r; f.gid_g = a.gid_g; f.row_k = a.row_k; f.row_v = a.row_v; f.E = a.E;
      f.R = R; f.G = Gm; f.P = a.P; f.y = a.y; f.pv_raw = a.pv_raw; f.pv = a.pv;
      f.lam = a.lam; f.lam_ub = a.lam_ub; f.lb = a.lb; f.b = a.b; f.c = a.c;
      f.lr = a.lr;
      f.rowvec = a.rowvec;
      rs_fg_body<VEC>(f, gd.c, Dl, rdl, djl, sred, fsh, w0l);
    }
  }
  }
  if constexpr (TEAM) {   // the control block as the head left it → the other members (through global memory)
    __syncthreads();
    if (trank == 0)
      for (int t = tid; t < (int)(sizeof(DevCtrl) / 8); t += NT)
        reinterpret_cast<unsigned long long*>(a.c)[t] = reinterpret_cast<const unsigned long long*>(&gd.c)[t];
    if (!rs_team_barrier(ts, &sh_team_ok)) {
      if (trank == 0 && tid == 0) {
        a.c->err = SDPLR_ERR_TEAM_TIMEOUT;
        if (a.out != nullptr) a.out[7] = (double)SDPLR_ERR_TEAM_TIMEOUT;
      }
      return;
    }
    if (trank != 0)
      for (int t = tid; t < (int)(sizeof(DevCtrl) / 8); t += NT)
        reinterpret_cast<unsigned long long*>(&gd.c)[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(a.c) + t);
    if (tid == 0) {
      sh_pvg = rs_ld_nt(a.pv_raw + a.gid_g);
      sh_lrpv = has_lr ? rs_ld_nt(a.pv_raw + a.lr.gid) : 0.0;
      sh_tflag = 0.0;
      for (int k = 0; k < 5 * SDPLR_HMAX; k++) gd.red[k] = 0.0;
      gd.nrm[0] = gd.nrm[1] = 0.0;
    }
    __syncthreads();
  }
  // (in a team every member forms them for itself: fg! in the prologue left them in rank 0's LDS only)
  if (has_lr && (TEAM || !a.pre_fg)) {   // w0 = Rᵀb and ws = y_c·D·w0 at the point and multipliers the previous g! left
    for (long long e = tid; e < N; e += NT) Dl[e] = R[e];
    __syncthreads();
    rs_colsum(Dl, a.lr.B, n, r, w0l);
    __syncthreads();
    if (tid < r) wsl[tid] = (TEAM ? rs_ld_nt(a.y + a.lr.gid) : a.y[a.lr.gid]) * a.lr.D * w0l[tid];
    __syncthreads();
  }
  // constraint data of this thread's rows: constant over the call except primal_vio_raw, which the thread owns
  // (a team member owns at most ⌈n/W⌉ rows: one per thread up to n = 1024; the second register set is what tipped the team
  // kernels into scratch)
  constexpr int RPT = TEAM ? 1 : SDPLR_RS_RPT;
  RsRow rw_[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) {
    const int j = row_lo + tid + q * NT;
    if (j < row_hi) rw_[q] = rs_load_row(a, j);
    else { rw_[q].k = -1; rw_[q].v = rw_[q].lam = rw_[q].pvr = 0.0; }
  }
  // (in a team every member forms its rows' d_j itself: fg! in the prologue left them in rank 0's LDS)
  if (PDROP && (TEAM || !a.pre_fg)) {   // d_j at the multipliers the previous g! left (fg! in the prologue has just formed them)
#pragma unroll
    for (int q = 0; q < RPT; q++) {
      const int j = row_lo + tid + q * NT;
      if (j < row_hi) djl[j] = (rw_[q].k >= 0) ? rw_[q].v * (TEAM ? rs_ld_nt(a.y + rw_[q].k) : a.y[rw_[q].k]) : 0.0;
    }
    for (int j = row_lo + tid + RPT * NT; j < row_hi; j += NT) {
      const int k = a.row_k[j];
      djl[j] = (k >= 0) ? a.row_v[j] * (TEAM ? rs_ld_nt(a.y + k) : a.y[k]) : 0.0;
    }
  }
  if (!PDROP && a.refresh_P && !a.pre_fg) {   // P = A_g·R (entry of the loop: R was written outside, or the incremental P is due for a refresh)
    for (long long e = tid; e < N; e += NT) Dl[e] = R[e];
    __syncthreads();
    rs_ell_spmm_any<VEC>(a.E, Dl, n, r, a.P);
  }
  __syncthreads();
  bool have_upd = false, have_norms = false;
  bool dir_ran = false;
#ifdef SDPLR_RS_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#define RS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif
  for (;;) {
    // (opaque to the optimiser: per-thread addresses are re-derived in every iteration instead of being hoisted out
    // of the persistent loop, where two dozen 64-bit address pairs would stay live across all phases — 256 VGPRs and
    // 116 bytes of scratch per lane without this, 217 and none with it)
    tid = tid0;
    asm volatile("" : "+v"(tid));
    if constexpr (TEAM) {   // every member's partials of the 5h Gram sums and the two norms → every member
      // (lane k of wave 0 carries entry k: one store, one round of TW loads in flight together, the sum in rank order)
      if (tid < 32) {
        const int q = tid / HM, l = tid % HM;
        double mine = 0.0;
        if (tid < 5 * HM) mine = gd.red[q * SDPLR_HMAX + l];
        else if (tid < 5 * HM + 2) mine = gd.nrm[tid - 5 * HM];
        else if (tid == 31 && trank == 0) mine = (a.budget_ticks > 0 && (long long)wall_clock64() - t_start > a.budget_ticks) ? 1.0 : 0.0;
        xg[32 * trank + tid] = mine;
      }
      if (!rs_team_barrier(ts, &sh_team_ok)) {
        if (tid == 0) { gd.c.err = SDPLR_ERR_TEAM_TIMEOUT; gd.c.done = 1; }
        __syncthreads();
        break;
      }
      if (tid < 32) {
        double v[SDPLR_RS_TEAM_MAX];
#pragma unroll
        for (int k = 0; k < SDPLR_RS_TEAM_MAX; k++) v[k] = rs_ld_nt(xg + 32 * min(k, TW - 1) + tid);
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < SDPLR_RS_TEAM_MAX; k++) t += (k < TW) ? v[k] : 0.0;
        const int q = tid / HM, l = tid % HM;
        if (tid < 5 * HM) gd.red[q * SDPLR_HMAX + l] = t;
        else if (tid < 5 * HM + 2) gd.nrm[tid - 5 * HM] = t;
        else if (tid == 31) sh_tflag = v[0];
      }
      __syncthreads();
    }
    // ================= SEAM =================
    // (the Gram sums and the norms of the iteration that just ended were folded into gd.red / gd.nrm by STEP)
    if (tid == 0) {
      switch (h) {
        case 1: seam_serial_small<1>(gd, 0, 1, 1, 1, have_upd, have_norms, 1); break;
        case 2: seam_serial_small<2>(gd, 0, 1, 1, 1, have_upd, have_norms, 1); break;
        case 3: seam_serial_small<3>(gd, 0, 1, 1, 1, have_upd, have_norms, 1); break;
        case 4: seam_serial_small<4>(gd, 0, 1, 1, 1, have_upd, have_norms, 1); break;
        default: seam_serial(gd, h, 0, 1, 1, 1, have_upd, have_norms, 1);
      }
      // time budget (src/sdplr.jl:272-277): tested where the iteration budget is; the device's own exits win
      // (a team takes rank 0's reading of the clock, exchanged with the partials above: every member must decide alike)
      if (!gd.c.done && (TEAM ? sh_tflag != 0.0 : (a.budget_ticks > 0 && (long long)wall_clock64() - t_start > a.budget_ticks))) {
        gd.c.iters -= 1;
        gd.c.done = 1;
        gd.c.exit_reason = EXIT_TIME;
      }
    }
    __syncthreads();
    RS_STAMP(0);
    if (gd.c.done) break;
    // ================= DIR =================
    const int latest = gd.c.latest;
    const int fb = gd.c.fallback;
    const int jslot = latest % h;
    {
      double ca[HM], cg[HM];
      const double* yp[HM];
      const double* sp_[HM];
      int j = latest - 1;
#pragma unroll
      for (int i = 0; i < HM; i++) {
        const int slot = (i < h) ? j : 0;
        ca[i] = (i < h) ? gd.c.c_alpha[slot] : 0.0;
        cg[i] = (i < h) ? gd.c.c_gamma[slot] : 0.0;
        yp[i] = aslot(a.A, as_y0(a.A) + slot);
        sp_[i] = aslot(a.A, AS_S0 + slot);
        j = (j <= 0) ? h - 1 : j - 1;
      }
      dir_ran = true;
      // Two doubles per request whatever the rank (nothing here knows about rows; the arrays are 16-byte aligned): with
      // 8-byte requests this phase ran at the ≈ 67 GB/s a CU gets out of L2 that way (20.6 k cycles for 576 KB), with
      // 16-byte ones it takes half of that.  The same sums per element, in the same order.
      auto dir_unit = [&](auto tag, long long e) {
        constexpr int W = decltype(tag)::value;
        // (a team member reads G and the history past its L1: most of their rows are its team-mates')
        const vecd<W> g = TEAM ? ldrow_nt<W>(Gm + e) : ldrow<W>(Gm + e);
        vecd<W> yv[HM], sv[HM];
#pragma unroll
        for (int k = 0; k < HM; k++) {
          yv[k] = TEAM ? ldrow_nt<W>(yp[k] + e) : ldrow<W>(yp[k] + e);
          sv[k] = TEAM ? ldrow_nt<W>(sp_[k] + e) : ldrow<W>(sp_[k] + e);
        }
        vecd<W> d;
#pragma unroll
        for (int q = 0; q < W; q++) {
          double rr = g.v[q];
#pragma unroll
          for (int k = 0; k < HM; k++) rr -= ca[k] * yv[k].v[q];          // newest → oldest (lbfgs.jl:94-102)
#pragma unroll
          for (int k = HM - 1; k >= 0; k--) rr += cg[k] * sv[k].v[q];     // oldest → newest (:104-113)
          d.v[q] = fb ? -g.v[q] : -rr;                                    // (:116-118); fallback: src/sdplr.jl:202-205
        }
        // (G ← −G of the fallback, src/sdplr.jl:202-205: in a team G keeps its sign — another member may still be reading
        // the element — and STEP takes G_old with the sign it has)
        if (fb && !TEAM) strow<W>(Gm + e, d);
        strow<W>(Dl + e, d);
      };
      const long long U2 = N / 2;
#pragma unroll 2
      for (long long u = tid; u < U2; u += NT) dir_unit(std::integral_constant<int, 2>{}, 2 * u);
      if ((N & 1) && tid == 0) dir_unit(std::integral_constant<int, 1>{}, N - 1);
    }
    __syncthreads();
    if (has_lr) rs_colsum(Dl, a.lr.B, n, r, w1l);   // w1 = Dᵀb (read by SOLVE, two barriers on)
    RS_STAMP(1);
    // ================= SPMM =================
    // W = A_g·D, one row per lane — and, by the lane that owns the row, ⟨R_j,D_j⟩, ‖D_j‖² (→ rdl, ddl) and this lane's part
    // of ⟨R,W⟩ (= ⟨P,D⟩: A_g symmetric) and ⟨D,W⟩
    double rw_sum = 0.0, dw_sum = 0.0;
    {
      const RsDots dots{R, rdl, ddl};
      if constexpr (TEAM) {
        RsEll Et = a.E;   // (this member's slices hold exactly its rows: W and the dots stay with their owner)
        const int spp = (rpm + 63) / 64;   // slices per member: build_rs_ell cuts the matrix into runs of ⌈rpm/64⌉ slices
        Et.sl0 = trank * spp;
        Et.n_slices = min(a.E.n_slices, (trank + 1) * spp);
        rs_ell_spmm_dots<VEC>(Et, Dl, n, r, a.W, dots, rw_sum, dw_sum);
      } else {
        rs_ell_spmm_dots<VEC>(a.E, Dl, n, r, a.W, dots, rw_sum, dw_sum);
      }
    }
    if (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as in front of a team barrier: without it the team kernels' register allocation tips into 600 bytes of scratch)
    __syncthreads();
    RS_STAMP(7);
    RS_STAMP(2);
    double acc[10];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0.0;
    acc[8] = rw_sum;
    acc[9] = dw_sum;
    // ================= LSSUM =================
    {
      const double sigma = gd.c.sigma;
      auto ls_row = [&](int j, const RsRow& rw) {
        if (rw.k < 0) return;
        int kk = rw.k;
        asm volatile("" : "+v"(kk));   // (opaque: see commit_row)
        const double rd = rdl[j], dd = ddl[j];
        const double q1 = rw.v * (rd + rd), q2 = rw.v * dd;
        a.A_RD[kk] = q1;
        a.A_DD[kk] = q2;
        if (kk < m) {
          const double l = rw.lam, nq0 = rw.pvr;
          acc[0] += l * nq0;
          acc[1] += nq0 * nq0;
          acc[2] += l * q1;
          acc[3] += nq0 * q1;
          acc[4] += (l - sigma * nq0) * q2;
          acc[5] += q1 * q1;
          acc[6] += q1 * q2;
          acc[7] += q2 * q2;
        } else {          // the cost slot attached to a row
          sh_p1 = q1;
          sh_p2 = q2;
        }
      };
#pragma unroll
      for (int q = 0; q < RPT; q++)
        if (row_lo + tid + q * NT < row_hi) ls_row(row_lo + tid + q * NT, rw_[q]);
      for (int j = row_lo + tid + RPT * NT; j < row_hi; j += NT) ls_row(j, rs_load_row(a, j));
    }
    rs_sum_to0<10>(acc, sred);
    if constexpr (TEAM) {   // the ten sums of every member → every member (added in rank order)
      __shared__ double sh_ls[10];
      if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) sh_ls[k] = acc[k];
      }
      __syncthreads();
      if (tid < 10) xl[16 * trank + tid] = sh_ls[tid];
      if (!rs_team_barrier(ts, &sh_team_ok)) {
        if (tid == 0) { gd.c.err = SDPLR_ERR_TEAM_TIMEOUT; gd.c.done = 1; }
        __syncthreads();
        break;
      }
      if (tid < 10) {
        double v[SDPLR_RS_TEAM_MAX];
#pragma unroll
        for (int k = 0; k < SDPLR_RS_TEAM_MAX; k++) v[k] = rs_ld_nt(xl + 16 * min(k, TW - 1) + tid);
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < SDPLR_RS_TEAM_MAX; k++) t += (k < TW) ? v[k] : 0.0;
        sh_ls[tid] = t;
      }
      __syncthreads();
      if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) acc[k] = sh_ls[k];
      }
    }
    RS_STAMP(3);
    // ================= SOLVE =================
    if (tid == 0) {
      DevCtrl& c = gd.c;
      const double sigma = c.sigma;
      const int kg = a.gid_g;
      const double g_rd = acc[8] + acc[8], g_dd = acc[9];
      const bool st0 = !TEAM || trank == 0;   // (every member solves; rank 0 stores)
      if (st0) {
        a.A_RD[kg] = g_rd;   // ⟨A_g, RDᵀ+DRᵀ⟩ = 2⟨P, D⟩
        a.A_DD[kg] = g_dd;   // ⟨A_g, DDᵀ⟩ = ⟨D, W⟩
      }
      double pvg = TEAM ? sh_pvg : a.pv_raw[kg], lg = 0.0, lubg = 0.0, lbg = 0.0;
      if (kg < m) {
        lg = a.lam[kg];
        lubg = a.lam_ub[kg];
        lbg = a.lb[kg];
        acc[0] += lg * pvg;
        acc[1] += pvg * pvg;
        acc[2] += lg * g_rd;
        acc[3] += pvg * g_rd;
        acc[4] += (lg - sigma * pvg) * g_dd;
        acc[5] += g_rd * g_rd;
        acc[6] += g_rd * g_dd;
        acc[7] += g_dd * g_dd;
      }
      // the rank-one matrix's slot: ⟨A_c, RDᵀ+DRᵀ⟩ = 2·D·⟨w0, w1⟩, ⟨A_c, DDᵀ⟩ = D·‖w1‖²  (src/coreop.jl:122-130,141-151)
      double lr_q1 = 0.0, lr_q2 = 0.0, lr_pv = 0.0, lr_l = 0.0;
      if (has_lr) {
        const int kc = a.lr.gid;
        double s01 = 0.0, s11 = 0.0;
        for (int ch = 0; ch < r; ch++) {
          s01 += w0l[ch] * w1l[ch];
          s11 += w1l[ch] * w1l[ch];
        }
        lr_q1 = 2.0 * (a.lr.D * s01);
        lr_q2 = a.lr.D * s11;
        if (st0) {
          a.A_RD[kc] = lr_q1;
          a.A_DD[kc] = lr_q2;
        }
        lr_pv = TEAM ? sh_lrpv : a.pv_raw[kc];
        lr_l = a.lam[kc];
        acc[0] += lr_l * lr_pv;
        acc[1] += lr_pv * lr_pv;
        acc[2] += lr_l * lr_q1;
        acc[3] += lr_pv * lr_q1;
        acc[4] += (lr_l - sigma * lr_pv) * lr_q2;
        acc[5] += lr_q1 * lr_q1;
        acc[6] += lr_q1 * lr_q2;
        acc[7] += lr_q2 * lr_q2;
      }
      const double p0 = c.obj, p1 = (kg == m) ? g_rd : sh_p1, p2 = (kg == m) ? g_dd : sh_p2;
      double bq[5];
      bq[0] = p0 - acc[0] + sigma * acc[1] / 2;
      bq[1] = p1 - acc[2] + sigma * acc[3];
      bq[2] = p2 - acc[4] + sigma * acc[5] / 2;
      bq[3] = sigma * acc[6];
      bq[4] = sigma * acc[7] / 2;
      for (int k = 0; k < 5; k++) c.biquad[k] = bq[k];
      double al = 0.0, f = bq[0];
      const int rc = quartic_argmin(bq, c.alpha_max, &al, &f);
      sh_err = rc;
      sh_alpha = al;
      if (rc != 0) {
        c.err = rc;
        c.done = 1;
      } else {
        c.alpha = al;
        c.L = f;
        const double last = c.lastval;
        const double rel_delta = (last - f) / fmax(1.0, fmax(fabs(f), fabs(last)));
        c.reldelta_exit = (rel_delta < c.fprec_eps) ? 1 : 0;
        sh_upd = c.reldelta_exit ? 0 : 1;
        // commit of A_g's slot (src/linesearch.jl:118-124, src/coreop.jl:229-236)
        const double v = pvg + al * (al * g_dd + g_rd);
        if (st0) a.pv_raw[kg] = v;
        if (TEAM) sh_pvg = v;
        double yk = 1.0, pv2 = 0.0;
        if (kg == m) {
          c.obj = v;
        } else {
          const double pc = fmax(v, lbg);
          if (st0) a.pv[kg] = pc;
          pv2 = pc * pc;
          yk = -fmin(lubg, lg - sigma * v);
        }
        if (st0) a.y[kg] = yk;
        sh_yg = yk;
        if (has_lr) {   // commit of the rank-one matrix's slot; W0 ← W0 + α·W1 = R_newᵀb; WS = y_c·D·W0  (k_ls_solve_fast's tail)
          const int kc = a.lr.gid;
          const double vc = lr_pv + al * (al * lr_q2 + lr_q1);
          if (st0) a.pv_raw[kc] = vc;
          if (TEAM) sh_lrpv = vc;
          const double pcc = fmax(vc, a.lb[kc]);
          if (st0) a.pv[kc] = pcc;
          pv2 += pcc * pcc;
          const double yc = -fmin(a.lam_ub[kc], lr_l - sigma * vc);
          if (st0) a.y[kc] = yc;
          for (int ch = 0; ch < r; ch++) {
            const double w = w0l[ch] + al * w1l[ch];
            w0l[ch] = w;
            wsol[ch] = wsl[ch];
            wsl[ch] = yc * a.lr.D * w;
          }
        }
        c.pv2_extra = pv2;
      }
    }
    __syncthreads();
    RS_STAMP(4);
    if (sh_err != 0) break;
    const double al = sh_alpha, yg = sh_yg;
    const bool upd = sh_upd != 0;
    // ================= COMMIT =================
    double red[5 * HM + 2];   // Gram sums of lbfgs_update!, then ‖G‖², ‖pv‖²
#pragma unroll
    for (int k = 0; k < 5 * HM + 2; k++) red[k] = 0.0;
    double* const gram = red;
    double* const nrm = red + 5 * HM;
    {
      const double sigma = gd.c.sigma;
      auto commit_row = [&](int j, RsRow& rw) {
        double dj = 0.0;
        if (rw.k >= 0) {
          // (opaque: the addresses derived from a row's constraint index are loop-invariant per thread, and every address
          // pair the compiler hoists out of the persistent loop stays live across all its phases — a spill)
          int kk = rw.k;
          asm volatile("" : "+v"(kk));
          const int kb = kk < m ? kk : 0;
          const double lub = (kk < m) ? a.lam_ub[kb] : 0.0, lbv = (kk < m) ? a.lb[kb] : 0.0;
          const double rd = rdl[j], dd = ddl[j];
          const double q1 = rw.v * (rd + rd), q2 = rw.v * dd;
          const double v = rw.pvr + al * (al * q2 + q1);     // src/linesearch.jl:118
          rw.pvr = v;
          a.pv_raw[kk] = v;
          double yk;
          if (kk < m) {
            yk = -fmin(lub, rw.lam - sigma * v);              // src/coreop.jl:233
            const double pc = fmax(v, lbv);                   // src/linesearch.jl:122-124
            a.pv[kk] = pc;
            nrm[1] += pc * pc;
          } else {
            yk = 1.0;                                         // the cost slot, src/coreop.jl:235
            gd.c.obj = v;
          }
          a.y[kk] = yk;
          dj = rw.v * yk;
        }
        if (PDROP) rdl[j] = djl[j];    // d_j at the old multipliers, for STEP (⟨R_j,D_j⟩ has been consumed above)
        djl[j] = dj;
      };
#pragma unroll
      for (int q = 0; q < RPT; q++)
        if (row_lo + tid + q * NT < row_hi) commit_row(row_lo + tid + q * NT, rw_[q]);
      for (int j = row_lo + tid + RPT * NT; j < row_hi; j += NT) {
        RsRow rw = rs_load_row(a, j);
        commit_row(j, rw);
      }
    }
    __syncthreads();
    RS_STAMP(5);
    // ================= STEP =================
    {
      // y_j = G_new − G_old: G_old is still in the G array (sign-flipped if the fallback negated it)
      const double gs = (fb && !TEAM) ? 1.0 : -1.0;
      double* const Sj = aslot(a.A, AS_S0 + jslot);
      double* const Yj = aslot(a.A, as_y0(a.A) + jslot);
      const double* slp[HM];
      const double* ylp[HM];
#pragma unroll
      for (int l = 0; l < HM; l++) {
        slp[l] = aslot(a.A, AS_S0 + ((l < h) ? l : 0));
        ylp[l] = aslot(a.A, as_y0(a.A) + ((l < h) ? l : 0));
      }
      // element by element in units of VEC doubles, like DIR (every stream fully coalesced, all lanes busy whatever the
      // rank); the row of a unit — for d_j — advances with it
      {
        const long long U0 = (long long)row_lo * r / VEC, U = (long long)row_hi * r / VEC;   // (VEC = 2 only for even r)
        const int adv = NT * VEC, adv_q = adv / r, adv_r = adv % r;
        int j = (int)(((U0 + tid) * VEC) / r), ch = (int)(((U0 + tid) * VEC) % r);
#pragma nounroll
        for (long long u = U0 + tid; u < U; u += NT) {
          const long long e = u * VEC;
          const vecd<VEC> x0 = ldrow<VEC>(R + e), gold = ldrow<VEC>(Gm + e);
          vecd<VEC> p0;
          if (!PDROP) p0 = ldrow<VEC>(a.P + e);
          const vecd<VEC> w = ldrow<VEC>(a.W + e);
          const vecd<VEC> d = ldrow<VEC>(Dl + e);
          vecd<VEC> sv[HM], yv[HM];
          if (upd) {
#pragma unroll
            for (int l = 0; l < HM; l++) {
              sv[l] = ldrow<VEC>(slp[l] + e);
              yv[l] = ldrow<VEC>(ylp[l] + e);
            }
          }
          const double dj = djl[j];
          const double djo = PDROP ? rdl[j] : 0.0;
          const double bj = has_lr ? a.lr.B[j] : 0.0;
          const int ch0 = ch;
          ch += adv_r;
          j += adv_q;
          if (ch >= r) { ch -= r; j++; }
          vecd<VEC> x, pp, g;
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            x.v[q] = x0.v[q] + al * d.v[q];                 // src/sdplr.jl:219
            if (PDROP) {
              double t = al * w.v[q] + (x.v[q] * dj - x0.v[q] * djo);
              if (has_lr) t += bj * (wsl[ch0 + q] - wsol[ch0 + q]);   // + b_j·(WS_new − WS_old)
              g.v[q] = -gs * gold.v[q] + 2.0 * t;           // G_old + 2(αW + d_new∘R_new − d_old∘R_old [+ low-rank])
            } else {
              pp.v[q] = p0.v[q] + al * w.v[q];
              g.v[q] = pp.v[q] * yg + x.v[q] * dj;
              if (has_lr) g.v[q] += wsl[ch0 + q] * bj;      // + Σ_c WS[c]·B[c]  (src/coreop.jl:271-278)
              g.v[q] *= 2.0;                                // src/coreop.jl:315
            }
            nrm[0] += g.v[q] * g.v[q];
          }
          strow<VEC>(R + e, x);
          if (!PDROP) strow<VEC>(a.P + e, pp);
          strow<VEC>(Gm + e, g);
          if (!upd) {   // relative-decrease exit: no update, y_next = −G_old as lbfgs_dir! leaves it (lbfgs.jl:121-123)
            vecd<VEC> go;
#pragma unroll
            for (int q = 0; q < VEC; q++) go.v[q] = gs * gold.v[q];
            strow<VEC>(Yj + e, go);
          } else {
            vecd<VEC> sn, yn;
#pragma unroll
            for (int q = 0; q < VEC; q++) {
              sn.v[q] = al * d.v[q];                        // BLAS.scal!(stepsize, dir)  (lbfgs.jl:142)
              yn.v[q] = gs * gold.v[q] + g.v[q];            // y_j = −G_old + G_new  (:122,145)
            }
            strow<VEC>(Sj + e, sn);                         // copy!(s_j, dir)  (:143)
            strow<VEC>(Yj + e, yn);
#pragma unroll
            for (int l = 0; l < HM; l++)
              if (l < h) {
                const vecd<VEC> sl = (l == jslot) ? sn : sv[l];
                const vecd<VEC> yl = (l == jslot) ? yn : yv[l];
#pragma unroll
                for (int q = 0; q < VEC; q++) {
                  gram[0 * HM + l] += sn.v[q] * yl.v[q];
                  gram[1 * HM + l] += sl.v[q] * yn.v[q];
                  gram[2 * HM + l] += yn.v[q] * yl.v[q];
                  gram[3 * HM + l] += sl.v[q] * g.v[q];
                  gram[4 * HM + l] += yl.v[q] * g.v[q];
                }
              }
          }
        }
      }
    }
    rs_sum_to0<5 * HM + 2>(red, sred);
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < 5; q++)
#pragma unroll
        for (int l = 0; l < HM; l++) gd.red[q * SDPLR_HMAX + l] = red[q * HM + l];
      gd.nrm[0] = red[5 * HM];
      gd.nrm[1] = red[5 * HM + 1];
    }
    have_upd = upd;
    have_norms = true;
    __syncthreads();
    RS_STAMP(6);
  }
#ifdef SDPLR_RS_STAMPS
  if (tid0 == 0 && gd.c.iters > 20)
    printf("[rs_loop] iters %lld  cycles/iter: seam %llu dir %llu spmm %llu rowdots %llu lssum %llu solve %llu commit %llu step %llu\n", gd.c.iters,
           stamp_acc[0] / gd.c.iters, stamp_acc[1] / gd.c.iters, stamp_acc[7] / gd.c.iters, stamp_acc[2] / gd.c.iters, stamp_acc[3] / gd.c.iters,
           stamp_acc[4] / gd.c.iters, stamp_acc[5] / gd.c.iters, stamp_acc[6] / gd.c.iters);
#endif
  // dirt as the reference leaves it when the loop is left
  __syncthreads();
  if (TEAM && trank != 0) return;   // (every member's stores were complete at the last team barrier; rank 0 closes the call)
  if (dir_ran) {
    // … = the unscaled direction after a relative-decrease exit (no lbfgs_update!), s_latest = α·dirt otherwise
    // (`dirt *= α`, src/lbfgs.jl:142, is not stored inside the loop)
    double* const Dg = aslot(a.A, AS_D);
    const bool scaled = gd.c.iters > 0 && gd.c.err == 0 && gd.c.exit_reason != EXIT_RELDELTA;
    const double* const Sl = aslot(a.A, AS_S0 + (gd.c.latest - 1));
    for (long long e = tid; e < N; e += NT) Dg[e] = scaled ? (TEAM ? rs_ld_nt(Sl + e) : Sl[e]) : Dl[e];
  }
  // the control block goes back whole (done / exit_reason / iters / L / norms / Gram data / latest …)
  if (tid == 0) {
    gd.c.gram_pending = 0;
    gd.c.norms_pending = 0;
  }
  __syncthreads();
  for (int t = tid; t < (int)(sizeof(DevCtrl) / 8); t += NT)
    reinterpret_cast<unsigned long long*>(a.c)[t] = reinterpret_cast<const unsigned long long*>(&gd.c)[t];
  if (a.out != nullptr && tid == 0) {
    const DevCtrl& c = gd.c;
    a.out[0] = c.L; a.out[1] = c.gnorm; a.out[2] = c.pvnorm; a.out[3] = c.alpha; a.out[4] = c.obj;
    a.out[5] = (double)c.iters; a.out[6] = (double)c.exit_reason; a.out[7] = (double)c.err;
  }
}
template <int VEC, int HM, bool PDROP>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_loop(RsLoopArgs a) { rs_loop_run<VEC, HM, PDROP>(a); }
// one workgroup per instance: block b takes row b of the argument table (64 small instances: one launch on 64 CUs instead
// of 64 launches that share the device only as far as the hardware queues allow)
template <int VEC, int HM, bool PDROP>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_loop_batch(const RsLoopArgs* __restrict__ items) {
  const RsLoopArgs a = items[blockIdx.x];
  rs_loop_run<VEC, HM, PDROP>(a);
}
// Teams: the members of a team are blocks b, b + 8, b + 16, … — dealt to the same XCD by the dispatcher's round-robin
// (checked by the members themselves).  Grid 8·W·⌈count/8⌉: block b is rank (b/8) mod W of instance ((b/8)/W)·8 + b mod 8.
template <int VEC, int HM>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_team_batch(const RsLoopArgs* __restrict__ items, int count, int W) {
  const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
  const int inst = (q / W) * 8 + xcd;
  if (inst >= count) return;
  RsLoopArgs a = items[inst];
  a.team_w = W;
  a.team_rank = q % W;
  rs_loop_run<VEC, HM, true, true>(a);
}
template <int VEC, int HM>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_team(RsLoopArgs a) {   // one instance: grid 8·(W − 1) + 1, the blocks b ≡ 0 (mod 8) are the team
  if (blockIdx.x % 8 != 0) return;
  a.team_rank = blockIdx.x / 8;
  rs_loop_run<VEC, HM, true, true>(a);
}

// ---- approx_mineigval_lanczos's recurrence (src/coreop.jl:473-500) in one launch -----------------------------------
// v, v_pre, Av in LDS; S = the assembled full pattern (column j of the CSC = row j: S is symmetric) + the low-rank
// terms y[gid]·D_c·B_c B_cᵀ.  The raw coefficients go to alpha_out / beta_out, the number of steps to c->lz_steps.
struct RsLzArgs {
  int n, q;
  const int *colptr, *rowval;
  const double* nzval;
  DevLowRank lr;
  const double* yvec;
  const double* v0;
  double *alpha_out, *beta_out;
  DevCtrl* c;
};
#define SDPLR_RS_LZ_LPR 8
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_lanczos(RsLzArgs a) {
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];   // (16-byte LDS reads: an 8-byte-aligned base behind the static LDS made every ds_read_b128 a misaligned access — 5× slower)
  __shared__ double sred[2 * SDPLR_RS_NW];
  __shared__ double sh_s[2];
  constexpr int NT = SDPLR_RS_NT, LPR = SDPLR_RS_LZ_LPR, G = NT / LPR;
  const int tid = threadIdx.x, grp = tid / LPR, lane = tid % LPR;
  const int n = a.n;
  double* v = rs_lds;
  double* vpre = v + n;
  double* av = vpre + n;
  auto bsum = [&](double x) -> double {   // block sum, every thread gets it (fixed order)
    x = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) sred[tid >> 6] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < SDPLR_RS_NW; i++) t += sred[i];
    return t;
  };
  {   // v = v0/‖v0‖  (:473-474)
    double s = 0.0;
    for (int i = tid; i < n; i += NT) {
      const double x = a.v0[i];
      v[i] = x;
      vpre[i] = 0.0;
      s += x * x;
    }
    const double nv = sqrt(bsum(s));
    for (int i = tid; i < n; i += NT) v[i] = v[i] / nv;
  }
  __syncthreads();
  double beta_prev = 0.0;
  int steps = 0;
  const double tiny = sqrt((double)n) * 2.220446049250313e-16;
  for (int it = 0; it < a.q; it++) {
    // low-rank coefficients y[gid]·D_c·⟨B_c, v⟩ (src/structs.jl:117-127)
    double coef[SDPLR_LRMAX];
#pragma unroll
    for (int cc = 0; cc < SDPLR_LRMAX; cc++) {
      coef[cc] = 0.0;
      if (cc < a.lr.ST) {
        double s = 0.0;
        for (int i = tid; i < n; i += NT) s += a.lr.Bcat[(long long)cc * n + i] * v[i];
        coef[cc] = a.yvec[a.lr.col_gid[cc]] * a.lr.Dcat[cc] * bsum(s);
      }
    }
    // Av = S·v  (:483)
    double dot = 0.0;
    for (int j = grp; j < n; j += G) {
      const int beg = a.colptr[j], end = a.colptr[j + 1];
      double tj = 0.0;
      for (int p = beg + lane; p < end; p += LPR) tj += a.nzval[p] * v[a.rowval[p]];
      tj = group_sum<LPR>(tj);
      if (lane == 0) {
#pragma unroll
        for (int cc = 0; cc < SDPLR_LRMAX; cc++)
          if (cc < a.lr.ST) tj += coef[cc] * a.lr.Bcat[(long long)cc * n + j];
        av[j] = tj;
        dot += v[j] * tj;
      }
    }
    const double al = bsum(dot);                                  // alpha[i] = v'·Av  (:484)
    double nn = 0.0;
    for (int i = tid; i < n; i += NT) {
      double x = av[i];
      if (it == 0) x -= al * v[i];                                // (:486-490)
      else x -= al * v[i] + beta_prev * vpre[i];
      av[i] = x;
      nn += x * x;
    }
    const double be = sqrt(bsum(nn));                             // beta[i] = ‖Av‖  (:492)
    steps = it + 1;
    if (tid == 0) {
      a.alpha_out[it] = al;
      a.beta_out[it] = be;
    }
    if (fabs(be) < tiny) break;                                   // (:494-496)
    for (int i = tid; i < n; i += NT) av[i] = av[i] / be;         // (:497)
    __syncthreads();
    double* t = vpre;                                             // (:498-499) as a rotation
    vpre = v;
    v = av;
    av = t;
    beta_prev = be;
  }
  if (tid == 0) {
    a.c->lz_steps = steps;
    a.c->lz_beta_prev = beta_prev;
    sh_s[0] = 0.0;
  }
}

// ---- the same recurrence on the structured form of S (instances of the resident loop) ------------------------------
// S(y) = y_g·A_g + Diag(d(y)): the off-diagonal part is the sliced ELL of A_g (one row per lane, RsEll), the diagonal
// sdiag_j = y_g·A_g[j,j] + v_j·y[k_j] a vector formed once per run — no assembled S is read.  With unit weights and room
// in LDS (ELL_LDS) the columns are packed two to a dword into LDS by the prologue and the q steps never leave the CU:
// per entry one half of a 4-byte LDS read and one 8-byte LDS gather of v.
struct RsLzEllArgs {
  int n, q;
  RsEll E;
  int gid_g;
  const int* row_k;
  const double* row_v;
  const double* yvec;
  const double* v0;
  double *alpha_out, *beta_out;
  DevCtrl* c;
  RsLr lr;
  // dual_obj (src/coreop.jl:376-415) around the recurrence, in the same launch: copy2y_λ_sub_pvio! (:384) before it,
  // ⟨y[1:m], b⟩ (:412) and the smallest eigenvalue of the tridiagonal (:502-513) after it
  int dual, m;
  double* y_rw;
  const double *lam, *lam_ub, *pv_raw, *b;
  double* out;                          // batched launches: [3] smallest eigenvalue, ⟨y, b⟩, steps
  double* y_copy;                       // batched launches: y as copy2y leaves it, once more ([m+1], or null) — the result table's
};
// number of eigenvalues of SymTridiagonal(d, e) below x (Sturm count), d = alpha + 1
__device__ __forceinline__ int rs_sturm_below(const double* al, const double* be, int k, double x) {
  int cnt = 0;
  double q = (al[0] + 1.0) - x;
  if (q < 0) cnt++;
  for (int i = 1; i < k; i++) {
    const double den = (q == 0.0) ? 2.2250738585072014e-308 : q;
    q = (al[i] + 1.0) - x - be[i - 1] * be[i - 1] / den;
    if (q < 0) cnt++;
  }
  return cnt;
}
template <bool ELL_LDS>
__device__ __forceinline__ void rs_lanczos_ell_run(const RsLzEllArgs& a) {
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];
  __shared__ double sred[SDPLR_RS_NW];
  __shared__ int lp[1026];     // ELL_LDS: first pair-line of each slice in the packed copy
  constexpr int NT = SDPLR_RS_NT, PF = SDPLR_RS_ELL_PF;
  const int tid = threadIdx.x, wave = tid >> 6, wl = tid & 63;
  const int n = a.n;
  double* v = rs_lds;
  double* vpre = v + n;
  double* av = vpre + n;
  double* sdiag = av + n;
  unsigned* ell = reinterpret_cast<unsigned*>(sdiag + n);
  auto bsum = [&](double x) -> double {   // block sum, every thread gets it (fixed order)
    x = wave_sum(x);
    __syncthreads();
    if (wl == 0) sred[wave] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < SDPLR_RS_NW; i++) t += sred[i];
    return t;
  };
  if (a.dual) {   // copy2y_λ_sub_pvio!  src/coreop.jl:229-236
    const double sigma = a.c->sigma;
    for (int i = tid; i <= a.m; i += NT) {
      const double yi = (i == a.m) ? 1.0 : -fmin(a.lam_ub[i], a.lam[i] - sigma * a.pv_raw[i]);
      a.y_rw[i] = yi;
      if (a.y_copy != nullptr) a.y_copy[i] = yi;
    }
    __syncthreads();
  }
  const double yg = a.yvec[a.gid_g];
  const bool uniform = a.E.val == nullptr;
  const double s_off = yg * a.E.one;         // the value every off-diagonal entry of S has (unit weights)
  // the rank-one matrix's share of S·v: y_c·D·b·⟨b, v⟩  (src/coreop.jl:291-298, src/structs.jl:135-145)
  const bool has_lr = a.lr.B != nullptr;
  const double lr_coef = has_lr ? a.yvec[a.lr.gid] * a.lr.D : 0.0;
  {   // v = v0/‖v0‖ (:473-474); the diagonal of S
    double s = 0.0;
    for (int i = tid; i < n; i += NT) {
      const double x = a.v0[i];
      v[i] = x;
      vpre[i] = 0.0;
      s += x * x;
      const int k = a.row_k[i];
      sdiag[i] = (k >= 0 ? a.row_v[i] * a.yvec[k] : 0.0) + a.E.gdiag[i] * yg;
    }
    if (ELL_LDS) {
      if (tid == 0) {
        int t = 0;
        for (int sl = 0; sl < a.E.n_slices; sl++) {
          lp[sl] = t;
          t += (a.E.sptr[sl + 1] - a.E.sptr[sl] + 1) >> 1;
        }
        lp[a.E.n_slices] = t;
      }
    }
    const double nv = sqrt(bsum(s));
    for (int i = tid; i < n; i += NT) v[i] = v[i] / nv;
    if (ELL_LDS) {
      for (int sl = wave; sl < a.E.n_slices; sl += SDPLR_RS_NW) {
        const int l0 = a.E.sptr[sl], width = a.E.sptr[sl + 1] - l0;
        const unsigned* ep = a.E.ent + (size_t)l0 * 64 + wl;
        unsigned* dst = ell + (size_t)lp[sl] * 64 + wl;
        for (int kp = 0; 2 * kp < width; kp++) {
          const unsigned c0 = ep[(size_t)(2 * kp) * 64];
          const unsigned c1 = (2 * kp + 1 < width) ? ep[(size_t)(2 * kp + 1) * 64] : 0u;
          dst[(size_t)kp * 64] = (c0 & 0xFFFFu) | (c1 << 16);
        }
      }
    }
  }
  __syncthreads();
  double beta_prev = 0.0;
  int steps = 0;
  const double tiny = sqrt((double)n) * 2.220446049250313e-16;
  constexpr int SLC = 2;
  int c_jp[SLC], c_len[SLC], c_l0[SLC], c_w[SLC];
#pragma unroll
  for (int q = 0; q < SLC; q++) {
    const int sl = wave + q * SDPLR_RS_NW;
    const bool in = sl < a.E.n_slices;
    const int idx = (in ? sl : 0) * 64 + wl;
    c_jp[q] = in ? a.E.perm[idx] : -1;
    c_len[q] = in ? a.E.len[idx] : 0;
    c_l0[q] = in ? a.E.sptr[sl] : 0;
    c_w[q] = in ? a.E.sptr[sl + 1] - c_l0[q] : 0;
  }
#ifdef SDPLR_RS_STAMPS
  unsigned long long lz_acc[5] = {0, 0, 0, 0, 0}, lz_last = __builtin_amdgcn_s_memtime();
#define LZ_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); lz_acc[i] += t_ - lz_last; lz_last = t_; } while (0)
#else
#define LZ_STAMP(i) do { } while (0)
#endif
  for (int it = 0; it < a.q; it++) {
    // Av = S·v  (:483), one row per lane
    double dot = 0.0;
    double lr_bv = 0.0;
    if (has_lr) {
      double t = 0.0;
      for (int i = tid; i < n; i += NT) t += a.lr.B[i] * v[i];
      lr_bv = lr_coef * bsum(t);
    }
    auto do_slice = [&](int sl, int jp, int len, int l0, int width) {
      double acc = 0.0;
      if (ELL_LDS) {
        const unsigned* el = ell + (size_t)lp[sl] * 64 + wl;
#pragma unroll 4
        for (int kp = 0; 2 * kp < width; kp++) {
          const unsigned u = el[(size_t)kp * 64];
          const double x0 = v[(2 * kp < len) ? (u & 0xFFFFu) : 0u], x1 = v[(2 * kp + 1 < len) ? (u >> 16) : 0u];
          acc += ((2 * kp < len) ? s_off : 0.0) * x0;
          acc += ((2 * kp + 1 < len) ? s_off : 0.0) * x1;
        }
      } else {
        const unsigned* ep = a.E.ent + (size_t)l0 * 64 + wl;
        const double* vp = uniform ? nullptr : a.E.val + (size_t)l0 * 64 + wl;
        unsigned er[PF];
        double vr[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int kc = min(q, width - 1);
          er[q] = ep[(size_t)kc * 64];
          vr[q] = uniform ? 0.0 : vp[(size_t)kc * 64];
        }
#pragma nounroll
        for (int k0 = 0; k0 < width; k0 += PF) {
          unsigned ec[PF];
          double vc[PF];
#pragma unroll
          for (int q = 0; q < PF; q++) {
            ec[q] = er[q];
            vc[q] = vr[q];
          }
#pragma unroll
          for (int q = 0; q < PF; q++) {
            const int kc = min(k0 + PF + q, width - 1);
            er[q] = ep[(size_t)kc * 64];
            vr[q] = uniform ? 0.0 : vp[(size_t)kc * 64];
          }
#pragma unroll
          for (int q = 0; q < PF; q++) {
            const bool on = k0 + q < len;
            const double sv = uniform ? s_off : vc[q] * yg;      // S[j,k] = y_g·A_g[j,k]  (src/coreop.jl:221)
            acc += (on ? sv : 0.0) * v[on ? (ec[q] & 0xFFFFu) : 0u];
          }
        }
      }
      if (jp >= 0) {
        const double vj = v[jp];
        double t = acc + sdiag[jp] * vj;
        if (has_lr) t += a.lr.B[jp] * lr_bv;
        av[jp] = t;
        dot += vj * t;
      }
    };
    // (a wave's first SLC slices — all of them up to n = 1024 — keep their row, length and extent in registers: fetched from
    // global memory in every step they were a dependent round trip in front of each slice, ≈ 1.6 of the step's 5.8 µs)
#pragma unroll
    for (int q = 0; q < SLC; q++)
      if (wave + q * SDPLR_RS_NW < a.E.n_slices) do_slice(wave + q * SDPLR_RS_NW, c_jp[q], c_len[q], c_l0[q], c_w[q]);
#pragma nounroll
    for (int sl = wave + SLC * SDPLR_RS_NW; sl < a.E.n_slices; sl += SDPLR_RS_NW) {
      const int idx = sl * 64 + wl;
      do_slice(sl, a.E.perm[idx], a.E.len[idx], a.E.sptr[sl], a.E.sptr[sl + 1] - a.E.sptr[sl]);
    }
    LZ_STAMP(0);
    const double al = bsum(dot);                                  // alpha[i] = v'·Av  (:484)
    LZ_STAMP(1);
    double nn = 0.0;
    for (int i = tid; i < n; i += NT) {
      double x = av[i];
      if (it == 0) x -= al * v[i];                                // (:486-490)
      else x -= al * v[i] + beta_prev * vpre[i];
      av[i] = x;
      nn += x * x;
    }
    LZ_STAMP(2);
    const double be = sqrt(bsum(nn));                             // beta[i] = ‖Av‖  (:492)
    LZ_STAMP(3);
    steps = it + 1;
    if (tid == 0) {
      a.alpha_out[it] = al;
      a.beta_out[it] = be;
    }
    if (fabs(be) < tiny) break;                                   // (:494-496)
    for (int i = tid; i < n; i += NT) av[i] = av[i] / be;         // (:497)
    __syncthreads();
    double* t = vpre;                                             // (:498-499) as a rotation
    vpre = v;
    v = av;
    av = t;
    beta_prev = be;
    LZ_STAMP(4);
  }
#ifdef SDPLR_RS_STAMPS
  if (tid == 0 && steps > 20)
    printf("[rs_lanczos] steps %d  cycles/step: spmv %llu dot-sum %llu update %llu norm-sum %llu scale %llu\n", steps, lz_acc[0] / steps,
           lz_acc[1] / steps, lz_acc[2] / steps, lz_acc[3] / steps, lz_acc[4] / steps);
#endif
  if (tid == 0) {
    a.c->lz_steps = steps;
    a.c->lz_beta_prev = beta_prev;
  }
  if (!a.dual) return;
  double yb;
  {   // ⟨y[1:m], b⟩  (:412)
    double t = 0.0;
    for (int i = tid; i < a.m; i += NT) t += a.yvec[i] * a.b[i];
    yb = bsum(t);
    if (tid == 0) a.c->descent = yb;
  }
  // The smallest eigenvalue of SymTridiagonal(alpha .+ 1, beta) minus 1 (:502-513) — the host routine's bisection
  // (sdplr_hip_tridiag_mineig), with 64 trial points per round by wave 0 until the bracket is a few ulps wide and the
  // scalar loop finishes it: the bracket it ends on is the one the sequential bisection ends on.
  // (α and β go to LDS first — the Lanczos vectors are done with: read from global memory inside the Sturm recurrence, two
  // loads in front of every one of its k dependent divisions, the bisection took ≈ 300 of the run's 1050 µs)
  __syncthreads();
  const bool tri_lds = steps <= n;
  if (tri_lds) {
    __threadfence_block();
    for (int i = tid; i < steps; i += NT) {
      av[i] = a.alpha_out[i];
      vpre[i] = a.beta_out[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
    __threadfence_block();
    const double* al = tri_lds ? av : a.alpha_out;   // (written by thread 0 above: same wave, global memory, fenced)
    const double* be = tri_lds ? vpre : a.beta_out;
    const int k = steps;
    double ev;
    if (k == 1) {
      ev = (al[0] + 1.0) - 1.0;                                   // (:505-507)
    } else {
      double lo = 1.0 / 0.0, hi = -lo;
      for (int i = 0; i < k; i++) {
        const double rad = (i > 0 ? fabs(be[i - 1]) : 0.0) + (i + 1 < k ? fabs(be[i]) : 0.0);
        lo = fmin(lo, (al[i] + 1.0) - rad);
        hi = fmax(hi, (al[i] + 1.0) + rad);
      }
      for (int round = 0; round < 40; round++) {
        const double w = hi - lo;
        const double x = lo + w * ((double)(wl + 1) / 65.0);
        const bool inside = x > lo && x < hi;
        const bool below = inside && rs_sturm_below(al, be, k, x) >= 1;
        const unsigned long long mk = __ballot(below), mi = __ballot(inside);
        if (mi != ~0ull) break;                                    // the trial points no longer separate: finish below
        const int f = mk ? __ffsll((long long)mk) - 1 : 64;        // first point with an eigenvalue below it
        const double xhi = __shfl(x, min(f, 63), 64), xlo = __shfl(x, max(f - 1, 0), 64);
        if (f < 64) hi = xhi;
        if (f > 0) lo = xlo;
      }
      for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (mid <= lo || mid >= hi) break;
        if (rs_sturm_below(al, be, k, mid) >= 1) hi = mid; else lo = mid;
      }
      ev = 0.5 * (lo + hi) - 1.0;                                  // cancel the shift (:513)
    }
    if (wl == 0) {
      a.c->lz_mineig = ev;
      if (a.out != nullptr) { a.out[0] = ev; a.out[1] = yb; a.out[2] = (double)steps; }
    }
  }
}
template <bool ELL_LDS>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_lanczos_ell(RsLzEllArgs a) { rs_lanczos_ell_run<ELL_LDS>(a); }
template <bool ELL_LDS>
__global__ void __launch_bounds__(SDPLR_RS_NT)
k_rs_lanczos_ell_batch(const RsLzEllArgs* __restrict__ items) {
  const RsLzEllArgs a = items[blockIdx.x];
  rs_lanczos_ell_run<ELL_LDS>(a);
}
