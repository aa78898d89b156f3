// k_sparse.h — the gather-bound operators on the aggregated sparse layout (src/structs.jl:274-294):
//   SDDMM  UVt[q] = ⟨U_col, V_row⟩ over the upper-triangular pattern   (src/coreop.jl:153-203)
//   segmented reduction  out[k] = Σ_{e∈seg k} nzval_two[e]·UVt[nzind[e]]     (src/coreop.jl:80-90)
//   S assembly  triu_nzval = Agg_one·y ; nzval = triu_nzval[mappedto_triu]     (src/coreop.jl:205-227)
//   SpMM   Y = scale·(X·S + low-rank terms)                                    (src/coreop.jl:260-279)
//   SpMV   y = S·x + low-rank terms                                            (src/coreop.jl:281-300)
//   low-rank projections  W = Xᵀ·B                                             (src/coreop.jl:115-130)
//
// A factor row (r doubles, 256 B at r = 32) is covered by a sub-wave group of LPR lanes holding
// VEC doubles each, so one wave64 instruction moves 64/LPR whole rows, each a contiguous
// (LPR·VEC·8)-byte segment.  Rows are gathered at random from a factor that fits the 256 MiB
// Infinity Cache but not the 4 MiB per-XCD L2, so these kernels are bound by gathered-row bandwidth,
// not by HBM streaming; there is no matrix-core work here (FP64 dot products of length r).
#pragma once
#include "common.h"
#include "k_dense.h"
#include "k_scalar.h"   // quartic_argmin (the line-search head of k_fast_step2)

struct DevSparse {
  int n, nnzT, nnzS, nnzAgg, n_sparse;
  const int *triu_colptr, *triu_rowval, *triu_colidx;  // upper-triangular aggregated pattern (CSC + explicit col)
  const int *colptr, *rowval, *mapped;                 // full aggregated pattern, map → triu position
  double *nzval, *triu_nzval;                          // S values (full / triu)
  const int *matptr, *nzind, *gids;                    // per-matrix segments over the triu pattern
  const double *nzval_one, *nzval_two;
  const int *tptr, *tmat;                              // transpose of the segments: triu position → (y index, value)
  const double* tval;
  // segmented-reduction plan
  int n_short, n_chunks, n_long, n_short_blocks;
  const int *short_ids, *chunk_beg, *chunk_end, *long_ids, *long_chunk_ptr;
  double *chunk_partial0, *chunk_partial1;
  double *UVt0, *UVt1;
  // rows of the full pattern with more than long_thresh nonzeros (hub vertices): one block each (spmm_long_rows)
  int n_long_rows, long_thresh;
  const int* long_rows;
  // "edge path" (disjoint supports: every position of the pattern belongs to at most one matrix — Lovász-θ's one
  // constraint per edge): the owner of each position, so that S need not be assembled and 𝒜 needs no segment pass
  const int* q_gid;        // [nnzT] slot of the owning matrix in the (m+1)-vectors, −1 ⇒ none
  const double* q_two;     // [nnzT] its nzval_two (value ×2 off the diagonal, src/preprocess.jl:124-131)
  const double* q_rec;     // [nnzT][4] the same as one 32-byte record per position: {row | col << 32, gid, two, 0}
                           //           (bit patterns): four lanes of a group fetch it with ONE load instruction
  const int* s_gid;        // [nnzS] the same per entry of the full pattern …
  const double* s_one;     // [nnzS] … with nzval_one: S.nzval[p] = s_one[p]·y[s_gid[p]]   (src/coreop.jl:205-227)
  int big_gid;             // the one matrix with several entries (Lovász-θ: the identity), −1 ⇒ none
  // S assembly by owner (k_edge_step): the ≤ 2 full-pattern entries of each single-entry matrix, and the big one's
  const int* own_pos;      // [2(m+1)] positions p0, p1 in the full pattern of slot k's matrix, −1 ⇒ none
  const double* own_one;   // [m+1]    its nzval_one
  int n_big_pos;
  const int* big_pos;      // [n_big_pos] full-pattern positions of the big matrix's entries
  const double* big_one;   // [n_big_pos]
};

struct DevLowRank {
  int n_lr, ST;            // matrices, total columns
  const double* Bcat;      // [ST][n]   column c contiguous
  const double* Dcat;      // [ST]
  const int* col_gid;      // [ST]  index into the (m+1)-vectors of the owning matrix
  const int* mat_ptr;      // [n_lr+1] column range of matrix t
  const int* mat_gid;      // [n_lr]
};

template <int VEC> struct vecd;
template <> struct vecd<1> { double v[1]; };
template <> struct vecd<2> { double v[2]; };

template <int VEC>
__device__ __forceinline__ vecd<VEC> ldrow(const double* __restrict__ p) {
  vecd<VEC> o;
  if constexpr (VEC == 2) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    o.v[0] = t.x;
    o.v[VEC - 1] = t.y;
  } else {
    o.v[0] = *p;
  }
  return o;
}
template <int VEC>
__device__ __forceinline__ void strow(double* __restrict__ p, const vecd<VEC>& o) {
  if constexpr (VEC == 2) {
    double2 t;
    t.x = o.v[0];
    t.y = o.v[VEC - 1];
    *reinterpret_cast<double2*>(p) = t;
  } else {
    *p = o.v[0];
  }
}

// non-temporal row store (experiments: -DSDPLR_RING_NT_G / -DSDPLR_RING_NT_R)
template <int VEC>
__device__ __forceinline__ void strow_nt(double* __restrict__ p, const vecd<VEC>& o) {
  if constexpr (VEC == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t;
    t.x = o.v[0];
    t.y = o.v[VEC - 1];
    __builtin_nontemporal_store(t, reinterpret_cast<d2*>(p));
  } else {
    __builtin_nontemporal_store(o.v[0], p);
  }
}

#ifndef SDPLR_STEP_TR
#define SDPLR_STEP_TR 4   /* rows per tile of k_fast_step2 (2: 4541, 4: 4565, 8: 4458, 16: 4389 it/s on the north-star instance) */
#endif
// base + 32-bit byte offset: with a wave-uniform base the load/store takes the base from SGPRs and the offset
// from ONE VGPR (global_load … v_off, s[base:base+1]) instead of a 64-bit address pair per array
__device__ __forceinline__ const double* rowat(const double* base, unsigned off) {
  return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ double* rowat(double* base, unsigned off) {
  return reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off);
}
// non-temporal row load: history that streams through once per kernel should not displace reused lines
template <int VEC>
__device__ __forceinline__ vecd<VEC> ldrow_nt(const double* __restrict__ p) {
  vecd<VEC> o;
  if constexpr (VEC == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 t = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    o.v[0] = t.x;
    o.v[VEC - 1] = t.y;
  } else {
    o.v[0] = __builtin_nontemporal_load(p);
  }
  return o;
}

// ---- SDDMM over the upper-triangular pattern --------------------------------------------------------
// MODE 0: UVt0[q] = ⟨U_c, U_r⟩                                   𝒜_sparse_formUUt!  src/coreop.jl:174-186
// MODE 1: UVt0[q] = (⟨U_c, V_r⟩ + ⟨V_c, U_r⟩)/2                   𝒜_sparse_formUVt!  src/coreop.jl:188-203
// MODE 2: UVt0[q] = ⟨U_c, V_r⟩ + ⟨V_c, U_r⟩ ; UVt1[q] = ⟨V_c, V_r⟩  both line-search passes in one sweep
//         (src/linesearch.jl:10-16 with U = Rt, V = dirt; the ×2 of :13 is already applied)
template <int LPR, int VEC, int MODE>
__global__ void __launch_bounds__(SDPLR_NT)
k_sddmm(DevSparse sp, const double* __restrict__ U, const double* __restrict__ V, int r,
        const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  for (long long q = (long long)blockIdx.x * G + threadIdx.x / LPR; q < sp.nnzT; q += total) {
    const long long row = sp.triu_rowval[q], col = sp.triu_colidx[q];
    double a0 = 0.0, a1 = 0.0;
    for (int ch = lane * VEC; ch < r; ch += LPR * VEC) {
      const vecd<VEC> uc = ldrow<VEC>(U + col * r + ch), ur = ldrow<VEC>(U + row * r + ch);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < VEC; k++) a0 += uc.v[k] * ur.v[k];
      } else {
        const vecd<VEC> vc = ldrow<VEC>(V + col * r + ch), vr = ldrow<VEC>(V + row * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          a0 += uc.v[k] * vr.v[k];
          a0 += vc.v[k] * ur.v[k];
          if (MODE == 2) a1 += vc.v[k] * vr.v[k];
        }
      }
    }
    a0 = group_sum<LPR>(a0);
    if (MODE == 2) a1 = group_sum<LPR>(a1);
    if (lane == 0) {
      sp.UVt0[q] = (MODE == 1) ? a0 / 2 : a0;
      if (MODE == 2) sp.UVt1[q] = a1;
    }
  }
}

// ---- segmented reduction over the per-matrix segments ----------------------------------------------
// short segments: one thread each; long segments: cut into chunks, one block per chunk writes a
// partial, k_seg_finalize adds a segment's partials in chunk order (deterministic, no atomics).
template <bool DUAL>
__global__ void __launch_bounds__(SDPLR_NT)
k_segreduce(DevSparse sp, double* __restrict__ out0, double* __restrict__ out1,
            const DevCtrl* __restrict__ c, int check_done) {
  __shared__ double sh[2 * (SDPLR_NT / 64)];
  if (check_done && c->done) return;
  if ((int)blockIdx.x < sp.n_short_blocks) {
    const int t = blockIdx.x * SDPLR_NT + threadIdx.x;
    if (t >= sp.n_short) return;
    const int k = sp.short_ids[t];
    double v0 = 0.0, v1 = 0.0;
    for (int e = sp.matptr[k]; e < sp.matptr[k + 1]; e++) {
      const int q = sp.nzind[e];
      const double w = sp.nzval_two[e];
      v0 += sp.UVt0[q] * w;
      if (DUAL) v1 += sp.UVt1[q] * w;
    }
    out0[sp.gids[k]] = v0;
    if (DUAL) out1[sp.gids[k]] = v1;
  } else {
    const int ci = blockIdx.x - sp.n_short_blocks;
    double v[2] = {0.0, 0.0};
    for (int e = sp.chunk_beg[ci] + threadIdx.x; e < sp.chunk_end[ci]; e += SDPLR_NT) {
      const int q = sp.nzind[e];
      const double w = sp.nzval_two[e];
      v[0] += sp.UVt0[q] * w;
      if (DUAL) v[1] += sp.UVt1[q] * w;
    }
    block_sum<2>(v, sh);
    if (threadIdx.x == 0) {
      sp.chunk_partial0[ci] = v[0];
      if (DUAL) sp.chunk_partial1[ci] = v[1];
    }
  }
}
template <bool DUAL>
__global__ void __launch_bounds__(SDPLR_NT)
k_seg_finalize(DevSparse sp, double* __restrict__ out0, double* __restrict__ out1,
               const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  const int w = (blockIdx.x * SDPLR_NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= sp.n_long) return;
  const int k = sp.long_ids[w];
  double v0 = 0.0, v1 = 0.0;
  for (int ci = sp.long_chunk_ptr[w] + lane; ci < sp.long_chunk_ptr[w + 1]; ci += 64) {
    v0 += sp.chunk_partial0[ci];
    if (DUAL) v1 += sp.chunk_partial1[ci];
  }
  v0 = wave_sum(v0);
  if (DUAL) v1 = wave_sum(v1);
  if (lane == 0) {
    out0[sp.gids[k]] = v0;
    if (DUAL) out1[sp.gids[k]] = v1;
  }
}

// ---- S assembly ------------------------------------------------------------------------------------
// triu_nzval[q] = Σ_e nzval_one[e]·y[global id of e's matrix], summed in ascending matrix order — the
// order in which the reference's CSC×vector product (src/coreop.jl:214-221) accumulates into it.
__global__ void __launch_bounds__(SDPLR_NT)
k_assemble_triu(DevSparse sp, const double* __restrict__ y, const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long q = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; q < sp.nnzT; q += stride) {
    double v = 0.0;
    for (int e = sp.tptr[q]; e < sp.tptr[q + 1]; e++) v += sp.tval[e] * y[sp.tmat[e]];
    sp.triu_nzval[q] = v;
  }
}
// sparse_S.nzval[i] = triu_nzval[mappedto_triu[i]]   (src/coreop.jl:223-226)
__global__ void __launch_bounds__(SDPLR_NT)
k_assemble_full(DevSparse sp, const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long p = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; p < sp.nnzS; p += stride)
    sp.nzval[p] = sp.triu_nzval[sp.mapped[p]];
}

// The three independent jobs between the line-search commit and the SpMM of g! on the generic in-loop path, in
// ONE grid (each of them alone sits on the ≈4.5 µs floor of a dependent launch):
//   blocks [0, nb_ax):            R += α·dirt                                   (src/sdplr.jl:219; k_axpy_R)
//   blocks [nb_ax, nb_ax+nb_s):   S assembly, both copies from the per-position lists (k_assemble_triu + _full:
//                                 each full-pattern entry recomputes its triangular value — same terms, same
//                                 order — so the two copies of an off-diagonal entry write identical bits)
//   the last block (lr.ST > 0):   low-rank coefficients at the moved point     (k_fast_lr_ws)
__global__ void __launch_bounds__(SDPLR_NT)
k_step_jobs(DevSparse sp, const double* __restrict__ y, const DevCtrl* __restrict__ c, double* __restrict__ R,
            const double* __restrict__ D, long long N, int nb_ax, int nb_s, DevLowRank lr, int r,
            double* __restrict__ lrW, double* __restrict__ lrWS) {
  if (c->done) return;
  const int b = blockIdx.x;
  if (b < nb_ax) {
    const double alpha = c->alpha;
    const long long N2 = N >> 1;
    const long long stride = (long long)nb_ax * SDPLR_NT;
    for (long long i = (long long)b * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
      double2 x = reinterpret_cast<double2*>(R)[i];
      const double2 d = reinterpret_cast<const double2*>(D)[i];
      x.x += alpha * d.x;
      x.y += alpha * d.y;
      reinterpret_cast<double2*>(R)[i] = x;
    }
    if ((N & 1) && b == 0 && threadIdx.x == 0) R[N - 1] += alpha * D[N - 1];
  } else if (b < nb_ax + nb_s) {
    const long long stride = (long long)nb_s * SDPLR_NT;
    for (long long p = (long long)(b - nb_ax) * SDPLR_NT + threadIdx.x; p < sp.nnzS; p += stride) {
      const int q = sp.mapped[p];
      double v = 0.0;
      for (int e = sp.tptr[q]; e < sp.tptr[q + 1]; e++) v += sp.tval[e] * y[sp.tmat[e]];
      sp.nzval[p] = v;
      sp.triu_nzval[q] = v;
    }
  } else {
    const double a = c->alpha;
    const int per = lr.ST * r;
    for (int t = threadIdx.x; t < per; t += SDPLR_NT) {
      const double w = lrW[t] + a * lrW[per + t];
      lrW[t] = w;
      lrWS[t] = y[lr.col_gid[t / r]] * lr.Dcat[t / r] * w;
    }
  }
}

// ---- lbfgs_update! (src/lbfgs.jl:129-149) riding a kernel that produces G_new row by row ----------------
// The lane that holds a chunk of G_new takes the same chunk of dirt, of G_old (still in the G array: read before
// the new value is stored; sign gs — it is −G_old's role that lbfgs_dir! would have parked in slot j, flipped
// once more if the steepest-descent fallback negated G) and of the HMU history pairs, stores s_j = α·dirt and
// y_j = G_new − G_old, and adds the five families of Gram dots into its own column of the LDS array `accl`
// ([5·HMU][NT], ds_add_f64 without return).  `dirt *= α` itself is left to the copy at loop exit.
struct UpdCtx {
  FactorArena A;
  const double* D;     // dirt
  double* accl;
  double alpha, gs;
  int h, jslot, upd;
};
template <int HMU>
__device__ __forceinline__ UpdCtx upd_ctx(const DevCtrl* c, FactorArena A, int h, const double* D, double* accl) {
  UpdCtx u;
  u.A = A;
  u.D = D;
  u.accl = accl;
  u.alpha = c->alpha;
  u.gs = c->fallback ? 1.0 : -1.0;
  u.h = h;
  u.jslot = c->latest % (h > 0 ? h : 1);
  u.upd = c->reldelta_exit == 0;
#pragma unroll
  for (int k = 0; k < 5 * HMU; k++) accl[k * SDPLR_NT + threadIdx.x] = 0.0;
  return u;
}
// g: the chunk of G_new; gold: the same chunk of the G array as it was; e: element offset of the chunk
template <int VEC, int HMU>
__device__ __forceinline__ void upd_row(const UpdCtx& u, const vecd<VEC>& g, const vecd<VEC>& gold, long long e) {
  double* Yj = aslot(u.A, as_y0(u.A) + u.jslot);
  if (!u.upd) {   // relative-decrease exit: no update, y_next = −G_old parked as lbfgs_dir! would have
    vecd<VEC> go;
#pragma unroll
    for (int q = 0; q < VEC; q++) go.v[q] = u.gs * gold.v[q];
    strow<VEC>(Yj + e, go);
    return;
  }
  const vecd<VEC> d = ldrow<VEC>(u.D + e);
  vecd<VEC> sv[HMU], yv[HMU];
#pragma unroll
  for (int l = 0; l < HMU; l++) {
    const int slot = (l < u.h) ? l : 0;
    sv[l] = ldrow<VEC>(aslot(u.A, AS_S0 + slot) + e);
    yv[l] = ldrow<VEC>(aslot(u.A, as_y0(u.A) + slot) + e);
  }
  vecd<VEC> sn, yn;
#pragma unroll
  for (int q = 0; q < VEC; q++) {
    sn.v[q] = u.alpha * d.v[q];              // BLAS.scal!(stepsize, dir)  (:142)
    yn.v[q] = u.gs * gold.v[q] + g.v[q];     // y_j = −G_old + G_new  (:122,145)
  }
  strow<VEC>(aslot(u.A, AS_S0 + u.jslot) + e, sn);   // copy!(s_j, dir)  (:143)
  strow<VEC>(Yj + e, yn);
#pragma unroll
  for (int l = 0; l < HMU; l++)
    if (l < u.h) {
      const vecd<VEC> sl = (l == u.jslot) ? sn : sv[l];
      const vecd<VEC> yl = (l == u.jslot) ? yn : yv[l];
      double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0;
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        q0 += sn.v[q] * yl.v[q];
        q1 += sl.v[q] * yn.v[q];
        q2 += yn.v[q] * yl.v[q];
        q3 += sl.v[q] * g.v[q];
        q4 += yl.v[q] * g.v[q];
      }
      double* ac = u.accl + threadIdx.x;
      (void)__hip_atomic_fetch_add(ac + (0 * HMU + l) * SDPLR_NT, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(ac + (1 * HMU + l) * SDPLR_NT, q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(ac + (2 * HMU + l) * SDPLR_NT, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(ac + (3 * HMU + l) * SDPLR_NT, q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(ac + (4 * HMU + l) * SDPLR_NT, q4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// block sums of the running Gram sums straight out of LDS, one wave per sum → partial entry `pidx`
template <int HMU>
__device__ __forceinline__ void upd_flush(const UpdCtx& u, DevCtrl* c, double* partials, int pidx, bool first_block) {
  __syncthreads();
  if (!u.upd) return;
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  for (int k = wave; k < 5 * HMU; k += SDPLR_NT / 64) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < SDPLR_NT / 64; i++) t += u.accl[k * SDPLR_NT + wl + 64 * i];
    t = wave_sum(t);
    const int q = k / HMU, l = k % HMU;
    if (wl == 0 && l < u.h) slot_partials(partials, SLOT_GRAM + q * SDPLR_HMAX + l)[pidx] = t;
  }
  if (first_block && threadIdx.x == 0) c->gram_pending = 1;  // consumed by k_lbfgs_boundary
}

// ---- SpMM: Y[j,:] = scale·( Σ_{p∈col j} S[p]·X[rowval[p],:] + Σ_c WS[c,:]·B[c][j] ) ----------------------
// 𝒜t!(y, x, aux, var) src/coreop.jl:260-279 (S is symmetric: column j of the CSC pattern lists row j's
// neighbours); scale = 2 fuses BLAS.scal!(2, Gt) of g! (:315); with slot ≥ 0 the ‖Y‖² partials of
// norm(Gt) (src/sdplr.jl:225) are produced on the way out.  WS[c,:] = y[gid]·D_c·(XᵀB)[c,:] is the
// low-rank product of src/structs.jl:135-145 prepared by k_lr_finalize.
// (bid, nblk: this block's index and the number of blocks working on the short rows — the body is shared by
// k_spmm and by k_spmm_both, where the hub-row blocks come first in the same grid)
// S value of full-pattern entry p: the assembled array, or (EDGE) the owner's y on the fly
template <bool EDGE>
__device__ __forceinline__ double s_value(const DevSparse& sp, const double* __restrict__ yv, int p) {
  if constexpr (EDGE) {
    const int g = sp.s_gid[p];
    return g >= 0 ? sp.s_one[p] * yv[g] : 0.0;
  } else {
    return sp.nzval[p];
  }
}

template <int LPR, int VEC, int HMU = 0, bool EDGE = false>
__device__ __forceinline__ void
spmm_rows(DevSparse sp, const double* __restrict__ X, double* Y, int r, double scale,
          DevLowRank lr, const double* __restrict__ WS, int slot, double* __restrict__ partials,
          const DevCtrl* __restrict__ c, int check_done, const double* __restrict__ Xdot, int bid, int nblk,
          const UpdCtx* u = nullptr, const double* __restrict__ yv = nullptr) {
  __shared__ double sh[8];
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)nblk * G;
  double nrm = 0.0;
  for (long long j = (long long)bid * G + threadIdx.x / LPR; j < sp.n; j += total) {
    const int beg = sp.colptr[j], end = sp.colptr[j + 1];
    if (sp.n_long_rows > 0 && end - beg > sp.long_thresh) continue;  // hub row: spmm_long_rows
    for (int chb = 0; chb < r; chb += LPR * VEC) {
      // all lanes of the group run the loop: the (index, value) pairs of the row are fetched LPR at a time and
      // handed round with shuffles, eight row gathers in flight per lane (see k_spmm_fast)
      const int ch = chb + lane * VEC;
      const bool act = ch < r;
      vecd<VEC> acc;
#pragma unroll
      for (int k = 0; k < VEC; k++) acc.v[k] = 0.0;
      for (int base = beg; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_i = (lane < cnt) ? sp.rowval[base + lane] : 0;
        const double my_v = (lane < cnt) ? s_value<EDGE>(sp, yv, base + lane) : 0.0;
        for (int k0 = 0; k0 < cnt; k0 += 8) {
          vecd<VEC> x[8];
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int src = min(k0 + q, LPR - 1);
            const long long i = __shfl(my_i, src, LPR);
            v[q] = __shfl(my_v, src, LPR);
            if (act && k0 + q < cnt) {
              x[q] = ldrow<VEC>(X + i * r + ch);
            } else {
#pragma unroll
              for (int k = 0; k < VEC; k++) x[q].v[k] = 0.0;
              v[q] = 0.0;
            }
          }
#pragma unroll
          for (int q = 0; q < 8; q++)
#pragma unroll
            for (int k = 0; k < VEC; k++) acc.v[k] += x[q].v[k] * v[q];
        }
      }
      if (!act) continue;
      for (int cc = 0; cc < lr.ST; cc++) {
        const double b = lr.Bcat[(long long)cc * sp.n + j];
        const vecd<VEC> w = ldrow<VEC>(WS + (long long)cc * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) acc.v[k] += w.v[k] * b;
      }
#pragma unroll
      for (int k = 0; k < VEC; k++) acc.v[k] *= scale;
      if (Xdot) {  // partial of ⟨Xdot, Y⟩ instead of ‖Y‖²
        const vecd<VEC> xd = ldrow<VEC>(Xdot + j * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) nrm += acc.v[k] * xd.v[k];
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) nrm += acc.v[k] * acc.v[k];
      }
      if constexpr (HMU > 0) {
        const vecd<VEC> gold = ldrow<VEC>(Y + j * r + ch);
        upd_row<VEC, HMU>(*u, acc, gold, j * r + ch);
      }
      strow<VEC>(Y + j * r + ch, acc);
    }
  }
  if (slot >= 0) {
    nrm = block_sum1(nrm, sh);
    if (threadIdx.x == 0) slot_partials(partials, slot)[bid] = nrm;
  }
}
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT)
k_spmm(DevSparse sp, const double* __restrict__ X, double* __restrict__ Y, int r, double scale,
       DevLowRank lr, const double* __restrict__ WS, int slot, double* __restrict__ partials,
       const DevCtrl* __restrict__ c, int check_done, const double* __restrict__ Xdot = nullptr) {
  spmm_rows<LPR, VEC>(sp, X, Y, r, scale, lr, WS, slot, partials, c, check_done, Xdot, blockIdx.x, gridDim.x);
}

// ---- SpMV on an n-vector: y = S·x + Σ_c coef[c]·B[c][:], with the partial of ⟨x, y⟩ ----------------
// 𝒜t!(y, aux, x, var) src/coreop.jl:281-300 for one column, 8 lanes per row; coef[c] =
// y[gid]·D_c·⟨B[c], x⟩ (src/structs.jl:117-127) prepared by k_lr_btx_finalize.  The dot is
// alpha[i] = v'·Av of the Lanczos recurrence (src/coreop.jl:484).
__global__ void __launch_bounds__(SDPLR_NT)
k_spmv(DevSparse sp, const double* __restrict__ x, double* __restrict__ y, DevLowRank lr,
       const double* __restrict__ coef, int slot, double* __restrict__ partials,
       const int* __restrict__ stop_flag) {
  __shared__ double sh[8];
  if (stop_flag && *stop_flag) return;
  constexpr int LPR = 8, G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  double dot = 0.0;
  for (long long j = (long long)blockIdx.x * G + threadIdx.x / LPR; j < sp.n; j += total) {
    if (sp.n_long_rows > 0 && sp.colptr[j + 1] - sp.colptr[j] > sp.long_thresh) continue;  // k_spmv_long
    double t = 0.0;
    for (int p = sp.colptr[j] + lane; p < sp.colptr[j + 1]; p += LPR) t += sp.nzval[p] * x[sp.rowval[p]];
    t = group_sum<LPR>(t);
    if (lane == 0) {
      for (int cc = 0; cc < lr.ST; cc++) t += coef[cc] * lr.Bcat[(long long)cc * sp.n + j];
      y[j] = t;
      dot += x[j] * t;
    }
  }
  if (slot >= 0) {
    dot = block_sum1(dot, sh);
    if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = dot;
  }
}

// ---- low-rank projections W[f][c][k] = Σ_i B[c][i]·X_f[i][k] -------------------------------------------
// `Ut * A.B` of tr_UtAU / tr_UtAV (src/coreop.jl:116,125-126) and `X * A.B` of mul! (src/structs.jl:142),
// for up to two factors in one pass.  Per-block partials lr_part[(((f*ST + c)*r + k)*gridDim.x + blk]
// (block index fastest, so that the finalize kernel reads each output's partials contiguously).
template <int LPR, int VEC, int F>
__global__ void __launch_bounds__(SDPLR_NT)
k_lr_project(DevLowRank lr, const double* __restrict__ X0, const double* __restrict__ X1, int n, int r,
             double* __restrict__ lr_part, const DevCtrl* __restrict__ c, int check_done) {
  __shared__ double sh[SDPLR_NT * VEC * SDPLR_LRMAX];
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR, g = threadIdx.x / LPR;
  const long long rows_per_block = ((long long)n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * rows_per_block;
  const long long hi = (lo + rows_per_block < n) ? lo + rows_per_block : n;
  // one pass over the rows for all F factors (their loads in flight together), SDPLR_LRMAX columns at a time
  for (int c0 = 0; c0 < lr.ST; c0 += SDPLR_LRMAX) {
    const int nc = (lr.ST - c0 < SDPLR_LRMAX) ? lr.ST - c0 : SDPLR_LRMAX;
    for (int chb = 0; chb < r; chb += LPR * VEC) {
      const int ch = chb + lane * VEC;
      double acc[F][SDPLR_LRMAX][VEC];
#pragma unroll
      for (int f = 0; f < F; f++)
#pragma unroll
        for (int cc = 0; cc < SDPLR_LRMAX; cc++)
#pragma unroll
          for (int k = 0; k < VEC; k++) acc[f][cc][k] = 0.0;
      if (ch < r)
        for (long long i = lo + g; i < hi; i += 4 * G) {   // four rows per trip: 4·F row loads in flight per lane
          vecd<VEC> x[4][F];
          long long ii[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            ii[u] = (i + u * G < hi) ? i + u * G : i;        // (a repeated row gets weight 0 below)
            x[u][0] = ldrow<VEC>(X0 + ii[u] * r + ch);
            if (F > 1) x[u][F - 1] = ldrow<VEC>(X1 + ii[u] * r + ch);
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int cc = 0; cc < SDPLR_LRMAX; cc++)
              if (cc < nc) {
                const double b = (i + u * G < hi) ? lr.Bcat[(long long)(c0 + cc) * n + ii[u]] : 0.0;
#pragma unroll
                for (int f = 0; f < F; f++)
#pragma unroll
                  for (int k = 0; k < VEC; k++) acc[f][cc][k] += x[u][f].v[k] * b;
              }
        }
#pragma unroll
      for (int f = 0; f < F; f++) {
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < SDPLR_LRMAX; cc++)
#pragma unroll
          for (int k = 0; k < VEC; k++) sh[(g * SDPLR_LRMAX + cc) * (LPR * VEC) + lane * VEC + k] = acc[f][cc][k];
        __syncthreads();
        for (int t = threadIdx.x; t < nc * LPR * VEC; t += SDPLR_NT) {
          const int cc = t / (LPR * VEC), k = t % (LPR * VEC);
          if (chb + k < r) {
            double sum = 0.0;
            for (int gg = 0; gg < G; gg++) sum += sh[(gg * SDPLR_LRMAX + cc) * (LPR * VEC) + k];
            lr_part[((((long long)f * lr.ST + c0 + cc) * r + chb + k)) * gridDim.x + blockIdx.x] = sum;
          }
        }
      }
    }
  }
}

// One block: W = Σ_blocks partials, then the mode-specific tail.
//  mode 0: out0[gid_t] = Σ_c D_c Σ_k W0²            tr_UtAU, src/coreop.jl:115-120   (F = 1)
//  mode 1: out0[gid_t] = Σ_c D_c Σ_k W0·W1          tr_UtAV, src/coreop.jl:122-130   (F = 2)
//  mode 2: out0[gid_t] = 2·Σ D W0·W1 ; out1[gid_t] = Σ D W1²   line search (src/linesearch.jl:10-16)
//  mode 3: WS[c][k] = yvec[gid_c]·D_c·W0[c][k]      mul!(Y, X, A, α, β), src/structs.jl:142-144
__global__ void __launch_bounds__(1024)
k_lr_finalize(DevLowRank lr, int r, int F, int nb, const double* __restrict__ lr_part, double* __restrict__ W,
              int mode, double* __restrict__ out0, double* __restrict__ out1, const double* __restrict__ yvec,
              double* __restrict__ WS, const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  const int per = lr.ST * r;
  const int nout = F * per;
  if (nb > 0) {  // one wave per output: lanes stride the nb contiguous per-block partials, fixed-order butterfly
                 // (nb == 0: W has already been summed by k_lr_reduce)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int t = wave; t < nout; t += nw) {
      const double* p = lr_part + (long long)t * nb;
      double s = 0.0;
      for (int b = lane; b < nb; b += 64) s += p[b];
      s = wave_sum(s);
      if (lane == 0) W[t] = s;
    }
  }
  __syncthreads();
  if (mode == 3) {
    for (int t = threadIdx.x; t < per; t += blockDim.x) {
      const int cc = t / r;
      WS[t] = yvec[lr.col_gid[cc]] * lr.Dcat[cc] * W[t];
    }
    return;
  }
  for (int t = threadIdx.x; t < lr.n_lr; t += blockDim.x) {
    double s0 = 0.0, s1 = 0.0;
    for (int cc = lr.mat_ptr[t]; cc < lr.mat_ptr[t + 1]; cc++) {
      double d0 = 0.0, d1 = 0.0;
      for (int k = 0; k < r; k++) {
        const double w0 = W[cc * r + k];
        if (mode == 0) d0 += w0 * w0;
        else {
          const double w1 = W[per + cc * r + k];
          d0 += w0 * w1;
          if (mode == 2) d1 += w1 * w1;
        }
      }
      s0 += d0 * lr.Dcat[cc];
      s1 += d1 * lr.Dcat[cc];
    }
    out0[lr.mat_gid[t]] = (mode == 2) ? 2.0 * s0 : s0;
    if (mode == 2) out1[lr.mat_gid[t]] = s1;
  }
}

// first stage of k_lr_finalize alone: used when the partials come from many blocks (the fused projections of
// k_spmm_tile / k_sddmm_edge) and one block would spend its time reading them.  One BLOCK per output; its ≤ 4096
// partials are fetched 16 per thread in one round trip (clamped, unconditional) and summed in a fixed order.
__global__ void __launch_bounds__(SDPLR_NT)
k_lr_reduce(int nout, int nb, const double* __restrict__ lr_part, double* __restrict__ W,
            const DevCtrl* __restrict__ c, int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;
  const int t = blockIdx.x;
  const double* p = lr_part + (long long)t * nb;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; q++) v[q] = p[min((int)threadIdx.x + SDPLR_NT * q, nb - 1)];
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q++) s += ((int)threadIdx.x + SDPLR_NT * q < nb) ? v[q] : 0.0;
  for (int b = threadIdx.x + 16 * SDPLR_NT; b < nb; b += SDPLR_NT) s += p[b];
  s = block_sum1(s, sh);
  if (dn) return;
  if (threadIdx.x == 0) W[t] = s;
}

// ⟨B[c], x⟩ partials for the SpMV low-rank term; grid.y = column
__global__ void __launch_bounds__(SDPLR_NT)
k_lr_btx(DevLowRank lr, const double* __restrict__ x, int n, double* __restrict__ part /* [ST][gridDim.x] */,
         const int* __restrict__ stop_flag) {
  __shared__ double sh[8];
  if (stop_flag && *stop_flag) return;
  const int cc = blockIdx.y;
  double t = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < n; i += stride)
    t += lr.Bcat[(long long)cc * n + i] * x[i];
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) part[(long long)cc * gridDim.x + blockIdx.x] = t;
}
__global__ void __launch_bounds__(SDPLR_NT)
k_lr_btx_finalize(DevLowRank lr, int nb, const double* __restrict__ part, const double* __restrict__ yvec,
                  double* __restrict__ coef, const int* __restrict__ stop_flag) {
  __shared__ double sh[8];
  if (stop_flag && *stop_flag) return;
  for (int cc = 0; cc < lr.ST; cc++) {
    const double s = reduce_partials(part + (long long)cc * nb, nb, sh);
    if (threadIdx.x == 0) coef[cc] = yvec[lr.col_gid[cc]] * lr.Dcat[cc] * s;
    __syncthreads();
  }
}

// ================================================================================================
// Structured fast path: exactly one sparse matrix A_g has off-diagonal entries
// (for MaxCut / MinBisection / CutNorm / μ-conductance: the cost matrix), every other sparse matrix is
// diagonal-only.  Then  S(y) = y_g·A_g + Diag(d(y)) + low-rank,  and with  P = A_g·R  kept resident
// (P += α·W after each step, W = A_g·D) the whole iteration needs ONE gather pass (W = A_g·D):
//   ⟨A_g, RDᵀ+DRᵀ⟩ = 2⟨P, D⟩,  ⟨A_g, DDᵀ⟩ = ⟨D, W⟩,  diagonal-only rows need only ⟨R_i,D_i⟩, ‖D_i‖²,
//   G = 2·(y_g·P + d(y)∘R + low-rank).
// ================================================================================================
struct DevFast {
  int gid_g;                 // slot of A_g in the (m+1)-vectors
  const int* diagpos;        // [n] position of (i,i) in the triu pattern, −1 if absent
  const int *drow_ptr, *drow_gid;   // per row: entries of the diagonal-only matrices (y index, value)
  const double* drow_val;
#ifdef SDPLR_PROBE_TILE_HIST
  const double* probe_hist[7];      // experiment: the streams a polynomial-in-α Gram epilogue would read (G, s_l, y_l)
#endif
};

// per-row dots for the diagonal-only matrices and the partial of ⟨P, D⟩:
// UVt0[diag(i)] = 2⟨R_i, D_i⟩, UVt1[diag(i)] = ‖D_i‖²  (what k_sddmm<…,2> would put there)
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT)
k_rowdots(DevSparse sp, DevFast ff, const double* __restrict__ R, const double* __restrict__ D,
          const double* __restrict__ P, int r, int slot, double* __restrict__ partials,
          const DevCtrl* __restrict__ c, int check_done) {
  __shared__ double sh[8];
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  double pd = 0.0;
  // two rows per group and trip: six independent 16-B loads in flight per lane
  for (long long j0 = (long long)blockIdx.x * G + threadIdx.x / LPR; j0 < sp.n; j0 += 2 * total) {
    const long long j1 = j0 + total;
    const bool has1 = j1 < sp.n;
    double rd0 = 0.0, dd0 = 0.0, rd1 = 0.0, dd1 = 0.0;
    for (int ch = lane * VEC; ch < r; ch += LPR * VEC) {
      const vecd<VEC> x0 = ldrow<VEC>(R + j0 * r + ch), d0 = ldrow<VEC>(D + j0 * r + ch), p0 = ldrow<VEC>(P + j0 * r + ch);
      vecd<VEC> x1, d1, p1;
      if (has1) {
        x1 = ldrow<VEC>(R + j1 * r + ch);
        d1 = ldrow<VEC>(D + j1 * r + ch);
        p1 = ldrow<VEC>(P + j1 * r + ch);
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) x1.v[k] = d1.v[k] = p1.v[k] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        rd0 += x0.v[k] * d0.v[k];
        dd0 += d0.v[k] * d0.v[k];
        pd += p0.v[k] * d0.v[k];
        rd1 += x1.v[k] * d1.v[k];
        dd1 += d1.v[k] * d1.v[k];
        pd += p1.v[k] * d1.v[k];
      }
    }
    rd0 = group_sum<LPR>(rd0);
    dd0 = group_sum<LPR>(dd0);
    rd1 = group_sum<LPR>(rd1);
    dd1 = group_sum<LPR>(dd1);
    if (lane == 0) {
      const int q0 = ff.diagpos[j0];
      if (q0 >= 0) {
        sp.UVt0[q0] = rd0 + rd0;
        sp.UVt1[q0] = dd0;
      }
      if (has1) {
        const int q1 = ff.diagpos[j1];
        if (q1 >= 0) {
          sp.UVt0[q1] = rd1 + rd1;
          sp.UVt1[q1] = dd1;
        }
      }
    }
  }
  pd = block_sum1(pd, sh);
  if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = pd;
}

// A_RD[g] = 2·⟨P, D⟩, A_DD[g] = ⟨D, W⟩ from the partials of k_rowdots / k_spmm.  One block.
__global__ void __launch_bounds__(SDPLR_NT)
k_fast_fill(DevFast ff, double* __restrict__ A_RD, double* __restrict__ A_DD, int slot_pd, int nb_pd,
            int slot_dw, int nb_dw, const double* __restrict__ partials, const DevCtrl* __restrict__ c,
            int check_done) {
  __shared__ double sh[8];
  if (check_done && c->done) return;
  const double pd = reduce_partials(slot_partials(partials, slot_pd), nb_pd, sh);
  __syncthreads();
  const double dw = reduce_partials(slot_partials(partials, slot_dw), nb_dw, sh);
  if (threadIdx.x == 0) {
    A_RD[ff.gid_g] = pd + pd;
    A_DD[ff.gid_g] = dw;
  }
}

// low-rank coefficients at the moved point: W0 ← W0 + α·W1 (= R_newᵀB), WS[c] = y[gid_c]·D_c·W0[c]
__global__ void __launch_bounds__(SDPLR_NT)
k_fast_lr_ws(DevLowRank lr, int r, double* __restrict__ W, const double* __restrict__ yvec,
             double* __restrict__ WS, const DevCtrl* __restrict__ c, int check_done) {
  if (check_done && c->done) return;
  const double a = c->alpha;
  const int per = lr.ST * r;
  for (int t = threadIdx.x; t < per; t += SDPLR_NT) {
    const double w = W[t] + a * W[per + t];
    W[t] = w;
    WS[t] = yvec[lr.col_gid[t / r]] * lr.Dcat[t / r] * w;
  }
}

// ================================================================================================
// Fast path, singleton form: every diagonal-only matrix has exactly one entry (MaxCut / MinBisection /
// CutNorm rows e_i e_iᵀ).  Its 𝒜 values are then row-local — A_RD[k] = 2v⟨R_i,D_i⟩, A_DD[k] = v‖D_i‖² —
// so the row-dot pass also produces the line-search sums of src/linesearch.jl:36-56 for those rows, and the
// step kernel also commits them (src/linesearch.jl:118-124) and forms y (src/coreop.jl:229-236): the
// segmented reduction, ls_partials and ls_commit launches disappear.  Slots not attached to a row — A_g
// itself and the low-rank matrices ("extra" slots) — are handled by the single-block scalar kernel.
// ================================================================================================
// The step of the structured fast path: R += α·D (src/sdplr.jl:219), P += α·W, then g! (src/coreop.jl:305-317)
// as G = 2·(y_g·P + d(y)∘R + Σ_c WS[c]·B[c]) with the ‖G‖² partials of norm(Gt) (src/sdplr.jl:225).
// COMMIT (singleton form): also the commit of the row-attached constraints and their y; the extra slots were
// committed by k_ls_solve_fast, whose y values (y_g, low-rank owners) are read here.  !COMMIT (general
// diagonal-only matrices): k_ls_commit has done both, d_j = Σ v·y[gid] is read off.
// HMU > 0: lbfgs_update! (src/lbfgs.jl:129-149) rides the same pass — k_lbfgs_update<HMU, true>'s work on the
// row chunk this lane already holds: dir *= α, s_j = dir, y_j += G_new, and the five families of Gram dots of
// the new pair against the HMU history slots.  One launch, one read of D and no re-read of G saved per
// iteration.  The 5·HMU running sums of a lane live in LDS (ds_add_f64 without return, one address per lane and
// sum: applied in issue order, so each is an ordinary sequential sum) because in registers they pushed the fused
// kernel to 166 VGPRs and scratch.  Skipped, as lbfgs_update! is, when the relative-decrease exit has been
// decided (src/sdplr.jl:239-241).
// PDROP (needs HMU > 0 with G_old read from the G array, COMMIT, A_g = the cost matrix — y_g ≡ 1 — and no low-rank
// matrices): P is neither read nor written.  With y_g fixed, G_new − G_old = 2·(α·W + d(y_new)∘R_new − d(y_old)∘R_old),
// and G_old streams through this kernel anyway (for y_j): the gradient is carried forward incrementally, exactly as P was
// (P += α·W), at 2N bytes less per iteration.  d(y_old) is read off y before the commit overwrites it.  G is rebuilt from
// scratch by every fg! / g! (each major iteration) and after SDPLR_HIP_P_REFRESH_ITERS incremental steps.
// LSH (with PDROP; A_g = the cost matrix is the only slot not attached to a row): the scalar stage of the exact line search
// (k_ls_solve_fast: fold of the ten line-search sums, quartic, α*, ℒ(α*), relative-decrease test, commit of the cost slot) runs
// as a prologue of EVERY block — the same partials summed in the same order, the same arithmetic: the same α in every block, bit
// for bit — instead of as a single-block kernel between the gather kernel and this one: one launch and one boundary less per
// iteration (≈ 8 µs) for ≈ 4 µs of prologue.  Block 0 stores the scalars; obj, which every block reads, is handed to the seam.
template <int LPR, int VEC, int HMU, bool COMMIT, bool PDROP = false, bool LSH = false>
__global__ void __launch_bounds__(SDPLR_NT)
k_fast_step2(int n, int m, DevFast ff, double* __restrict__ R, double* __restrict__ D,
             double* __restrict__ P, const double* __restrict__ W, double* Gout, int r,
             double* __restrict__ yvec, const double* __restrict__ lam, const double* __restrict__ lam_ub,
             double* __restrict__ pv_raw, const double* __restrict__ lb, double* __restrict__ pv,
             const double* __restrict__ A_RD, const double* __restrict__ A_DD, DevLowRank lr,
             const double* __restrict__ WS, double* __restrict__ partials, DevCtrl* __restrict__ c,
             int check_done, FactorArena A, int h, int gold_in_G, int nb_ls = 0) {
  constexpr int HA = HMU > 0 ? HMU : 1;
  __shared__ double sh[2 * (SDPLR_NT / 64)];
  extern __shared__ double accl[];  // HMU > 0: [5·HMU][NT] running Gram sums, one column per lane
  // the flag is fetched together with the scalars and the first row, and tested before anything is stored:
  // one memory round trip at the head of the kernel instead of three
  const int dn = check_done ? c->done : 0;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  const double sigma = c->sigma;
  double a, yg;
  bool upd;
  if constexpr (LSH) {
    __shared__ double shl[10 * (SDPLR_NT / 64)];
    const double obj0 = c->obj, amax = c->alpha_max, last = c->lastval, feps = c->fprec_eps;
    double s10[10];
    {
      // the partials, 16 bytes per lane and column (k_ls_solve_fast's fetch: nb_ls ≤ 1024 producers, absent entries masked)
      constexpr int PT = 2;
      double2 v[10][PT];
      const int ncols = (nb_ls + 2 * SDPLR_NT - 1) / (2 * SDPLR_NT);
#pragma unroll
      for (int q = 0; q < PT; q++) {
        const int i = (int)threadIdx.x + SDPLR_NT * q;
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
        s10[k] = 0.0;
#pragma unroll
        for (int q = 0; q < PT; q++) {
          const int i = 2 * ((int)threadIdx.x + SDPLR_NT * q);
          s10[k] += (i < nb_ls) ? v[k][q].x : 0.0;
          s10[k] += (i + 1 < nb_ls) ? v[k][q].y : 0.0;
        }
      }
    }
    block_sum<10>(s10, shl);   // (k_ls_solve_fast's order: lane → wave → the four waves)
    const double g_rd = s10[8] + s10[8], g_dd = s10[9];
    double bq[5];
    bq[0] = obj0 - s10[0] + sigma * s10[1] / 2;      // src/linesearch.jl:44-56 (the cost slot is p0, p1, p2)
    bq[1] = g_rd - s10[2] + sigma * s10[3];
    bq[2] = g_dd - s10[4] + sigma * s10[5] / 2;
    bq[3] = sigma * s10[6];
    bq[4] = sigma * s10[7] / 2;
    double al = 0.0, f = bq[0];
    const int rc = quartic_argmin(bq, amax, &al, &f);
    if (rc != 0) {
      if (blockIdx.x == 0 && threadIdx.x == 0 && !dn) {
        c->err = rc;
        c->done = 1;
      }
      return;
    }
    const double rel_delta = (last - f) / fmax(1.0, fmax(fabs(f), fabs(last)));   // src/sdplr.jl:238
    upd = HMU > 0 && !(rel_delta < feps);
    a = al;
    yg = 1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !dn) {
      for (int k = 0; k < 5; k++) c->biquad[k] = bq[k];
      c->alpha = al;
      c->L = f;
      c->reldelta_exit = upd ? 0 : 1;
      const_cast<double*>(A_RD)[m] = g_rd;    // ⟨A_g, RDᵀ+DRᵀ⟩ = 2⟨P, D⟩ (= 2⟨R, W⟩)
      const_cast<double*>(A_DD)[m] = g_dd;    // ⟨A_g, DDᵀ⟩ = ⟨D, W⟩
      c->obj_next = obj0 + al * (al * g_dd + g_rd);   // commit of the cost slot (src/linesearch.jl:118-121): stored by the seam
      c->obj_pending = 1;
      c->pv2_extra = 0.0;
    }
  } else {
    a = c->alpha;
    yg = yvec[ff.gid_g];
    upd = HMU > 0 && c->reldelta_exit == 0;
  }
  // slot j's stream carries −G_old (parked by lbfgs_dir!) or, with gold_in_G, the G array itself: ±G_old
  const double gs = (HMU > 0 && gold_in_G) ? (c->fallback ? 1.0 : -1.0) : 1.0;
  const int jslot = HMU > 0 ? (c->latest % h) : 0;
  double red[2] = {0.0, 0.0};  // ‖G‖², ‖pv‖² (row-attached slots)
  const int ch0 = lane * VEC;
  double* const Sj = HMU > 0 ? aslot(A, AS_S0 + jslot) : nullptr;
  double* const Yj = HMU > 0 ? aslot(A, as_y0(A) + jslot) : nullptr;
  const double* slp[HA];
  const double* ylp[HA];
#pragma unroll
  for (int l = 0; l < HA; l++) {   // slots l ≥ h alias slot 0 (loaded, never used): every load unconditional
    slp[l] = HMU > 0 ? aslot(A, AS_S0 + ((l < h) ? l : 0)) : nullptr;
    ylp[l] = HMU > 0 ? aslot(A, as_y0(A) + ((l < h) ? l : 0)) : nullptr;
    // y_j = G_new − G_old (src/lbfgs.jl:121-123,145): G_old is still in the G array when this lane reaches the
    // element — it is read there, in place of the −G_old that lbfgs_dir! would have parked in slot j (the
    // in-loop direction kernel skips that store: N bytes per iteration less)
    // (gold_in_G; if the direction kernel took the steepest-descent fallback it has flipped G: sign gs)
    if (HMU > 0 && gold_in_G && l == jslot) ylp[l] = Gout;
    // the old s_j is overwritten, never used: its (unconditional) load is pointed at the D row this lane has just
    // requested — a cache hit instead of N bytes from memory
    if (HMU > 0 && l == jslot) slp[l] = D;
  }
  if (HMU > 0) {
#pragma unroll
    for (int k = 0; k < 5 * HMU; k++) accl[k * SDPLR_NT + threadIdx.x] = 0.0;
  }
  // A group takes TR ≤ LPR consecutive rows at a time.  Phase A: lane k walks the constraints attached to row
  // j0 + k (commit of src/linesearch.jl:118-124, y of src/coreop.jl:229-236, the row's diagonal coefficient
  // d_j) — one dependent pointer → slot → values chain for TR rows instead of one per row.  Phase B: the rows
  // stream through, two at a time, with d_j handed over from the owning lane.
  constexpr int TR = LPR < SDPLR_STEP_TR ? LPR : SDPLR_STEP_TR;   // rows per tile
  const long long ntiles = ((long long)n + TR - 1) / TR;
  for (long long tile = (long long)blockIdx.x * G + threadIdx.x / LPR; tile < ntiles; tile += total) {
    const long long j0 = tile * TR;
    const int nrows = (int)min((long long)TR, (long long)n - j0);
    if (dn) return;
    double djl = 0.0, djo = 0.0;     // d_j at the new / (PDROP) the old multipliers
    if (lane < nrows) {
      const long long j = j0 + lane;
      const int e0 = ff.drow_ptr[j], e1 = ff.drow_ptr[j + 1];
      for (int e = e0; e < e1; e++) {
        const int k = ff.drow_gid[e];
        if (!COMMIT) {               // non-singleton form: k_ls_commit has committed and formed y already
          djl += ff.drow_val[e] * yvec[k];
          continue;
        }
        if (PDROP) djo += ff.drow_val[e] * yvec[k];                 // y as the previous g! left it
        const double v = pv_raw[k] + a * (a * A_DD[k] + A_RD[k]);   // src/linesearch.jl:118
        double yk;
        if (k < m) {
          yk = -fmin(lam_ub[k], lam[k] - sigma * v);                // src/coreop.jl:233
          const double pc = fmax(v, lb[k]);                         // src/linesearch.jl:122-124
          pv[k] = pc;
          red[1] += pc * pc;
        } else {
          yk = 1.0;                                                  // the cost slot, src/coreop.jl:235
          c->obj = v;
        }
        djl += ff.drow_val[e] * yk;
        yvec[k] = yk;
        pv_raw[k] = v;
      }
    }
    for (int i0 = 0; i0 < nrows; i0 += 2) {
      vecd<VEC> xx[2], pq[2], dd[2], ww[2], sv[2][HA], yv[2][HA];
      double dj[2], dq[2];
      unsigned off[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = min(i0 + u, nrows - 1);                        // (odd tail: row repeated, not stored twice)
        const long long j = j0 + i;
        dj[u] = __shfl(djl, i, LPR);
        dq[u] = PDROP ? __shfl(djo, i, LPR) : 0.0;
        off[u] = (unsigned)(((unsigned long long)j * (unsigned)r + (unsigned)ch0) * 8ull);
#pragma unroll
        for (int q = 0; q < VEC; q++) xx[u].v[q] = pq[u].v[q] = dd[u].v[q] = ww[u].v[q] = 0.0;
#pragma unroll
        for (int l = 0; l < HA; l++)
#pragma unroll
          for (int q = 0; q < VEC; q++) sv[u][l].v[q] = yv[u][l].v[q] = 0.0;
        if (ch0 < r) {
          if (HMU > 0) {
            // (fused form: one 32-bit byte offset serves all eighteen row accesses — the bases are uniform,
            // the load adds them — instead of a 64-bit address pair per array; launched only when 8·n·r < 2³²)
            xx[u] = ldrow<VEC>(rowat(R, off[u]));
            if (!PDROP) pq[u] = ldrow<VEC>(rowat(P, off[u]));
            dd[u] = ldrow<VEC>(rowat(D, off[u]));
            ww[u] = ldrow<VEC>(rowat(W, off[u]));
#pragma unroll
            for (int l = 0; l < HMU; l++) {
              sv[u][l] = ldrow_nt<VEC>(rowat(slp[l], off[u]));
              yv[u][l] = ldrow_nt<VEC>(rowat(ylp[l], off[u]));
            }
          } else {
            xx[u] = ldrow<VEC>(R + j * r + ch0);
            pq[u] = ldrow<VEC>(P + j * r + ch0);
            dd[u] = ldrow<VEC>(D + j * r + ch0);
            ww[u] = ldrow<VEC>(W + j * r + ch0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (i0 + u >= nrows) break;
        const long long j = j0 + i0 + u;
        for (int ch = ch0; ch < r; ch += LPR * VEC) {
          vecd<VEC> x = xx[u], pp = pq[u], d = dd[u], w = ww[u];
          const unsigned offc = off[u] + (unsigned)(ch - ch0) * 8u;   // this chunk (ranks beyond LPR·VEC: several per row)
          if (ch != ch0) {
            x = ldrow<VEC>(R + j * r + ch);
            if (!PDROP) pp = ldrow<VEC>(P + j * r + ch);
            d = ldrow<VEC>(D + j * r + ch);
            w = ldrow<VEC>(W + j * r + ch);
            if (HMU > 0) {
#pragma unroll
              for (int l = 0; l < HMU; l++) {
                sv[u][l] = ldrow_nt<VEC>(rowat(slp[l], offc));
                yv[u][l] = ldrow_nt<VEC>(rowat(ylp[l], offc));
              }
            }
          }
          vecd<VEC> g;
          if constexpr (PDROP) {
            // G_new = G_old + 2·(α·W + d_new∘R_new − d_old∘R_old)   (y_g ≡ 1: A_g is the cost matrix)
            vecd<VEC> go;
#pragma unroll
            for (int q = 0; q < VEC; q++) go.v[q] = 0.0;
#pragma unroll
            for (int l = 0; l < HMU; l++)      // slot j's stream is the G array: ±G_old (sign flipped by a fallback)
              if (l == jslot) go = yv[u][l];
#pragma unroll
            for (int q = 0; q < VEC; q++) {
              const double xo = x.v[q];
              x.v[q] += a * d.v[q];
              const double t = a * w.v[q] + (x.v[q] * dj[u] - xo * dq[u]);
              g.v[q] = -gs * go.v[q] + 2.0 * t;
              red[0] += g.v[q] * g.v[q];
            }
          } else {
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            x.v[q] += a * d.v[q];
            pp.v[q] += a * w.v[q];
            g.v[q] = pp.v[q] * yg + x.v[q] * dj[u];
          }
          for (int cc = 0; cc < lr.ST; cc++) {
            const double bb = lr.Bcat[(long long)cc * n + j];
            const vecd<VEC> ws = ldrow<VEC>(WS + (long long)cc * r + ch);
#pragma unroll
            for (int q = 0; q < VEC; q++) g.v[q] += ws.v[q] * bb;
          }
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            g.v[q] *= 2.0;
            red[0] += g.v[q] * g.v[q];
          }
          }
          if (HMU > 0) {
            strow<VEC>(rowat(R, offc), x);
            if (!PDROP) strow<VEC>(rowat(P, offc), pp);
            strow<VEC>(rowat(Gout, offc), g);
          } else {
            strow<VEC>(R + j * r + ch, x);
            strow<VEC>(P + j * r + ch, pp);
            strow<VEC>(Gout + j * r + ch, g);
          }
          if (HMU > 0 && !upd) {   // leaving through the relative-decrease exit: y_next = −G_old as lbfgs_dir! leaves it
            vecd<VEC> go;
#pragma unroll
            for (int q = 0; q < VEC; q++) go.v[q] = 0.0;
#pragma unroll
            for (int l = 0; l < HMU; l++)
              if (l == jslot) go = yv[u][l];
#pragma unroll
            for (int q = 0; q < VEC; q++) go.v[q] = gs * go.v[q];
            if (gold_in_G) strow<VEC>(rowat(Yj, offc), go);
          }
          if (HMU > 0 && upd) {
            vecd<VEC> sn, yn;
#pragma unroll
            for (int q = 0; q < VEC; q++) yn.v[q] = 0.0;
#pragma unroll
            for (int l = 0; l < HMU; l++)      // slot j's stream is the G array: G_old
              if (l == jslot) yn = yv[u][l];
#pragma unroll
            for (int q = 0; q < VEC; q++) {
              sn.v[q] = a * d.v[q];            // BLAS.scal!(stepsize, dir)  (src/lbfgs.jl:142)
              yn.v[q] = gs * yn.v[q] + g.v[q]; // y_j = −G_old + G_new  (:122,145)
            }
            // dir *= α (:142) is not stored here: nothing reads the scaled direction before lbfgs_dir!
            // overwrites it, and the host copies s_j into dirt when the loop is left (sdplr_hip_inner_loop)
            strow<VEC>(rowat(Sj, offc), sn);   // copy!(s_j, dir)  (:143)
            strow<VEC>(rowat(Yj, offc), yn);
#pragma unroll
            for (int l = 0; l < HMU; l++)
              if (l < h) {
                const vecd<VEC> sl = (l == jslot) ? sn : sv[u][l];
                const vecd<VEC> yl = (l == jslot) ? yn : yv[u][l];
                double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0;
#pragma unroll
                for (int q = 0; q < VEC; q++) {
                  q0 += sn.v[q] * yl.v[q];
                  q1 += sl.v[q] * yn.v[q];
                  q2 += yn.v[q] * yl.v[q];
                  q3 += sl.v[q] * g.v[q];
                  q4 += yl.v[q] * g.v[q];
                }
                double* ac = accl + threadIdx.x;
                (void)__hip_atomic_fetch_add(ac + (0 * HMU + l) * SDPLR_NT, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(ac + (1 * HMU + l) * SDPLR_NT, q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(ac + (2 * HMU + l) * SDPLR_NT, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(ac + (3 * HMU + l) * SDPLR_NT, q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(ac + (4 * HMU + l) * SDPLR_NT, q4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
          }
          if (HMU > 0 && LPR < 64) break;   // a row is one chunk unless the sub-wave group is a whole wave (r > 64·VEC)
        }
      }
    }
  }
  if (dn) return;
  if constexpr (HMU > 0) {
    // block sums of the running Gram sums straight out of LDS, one wave per sum (lanes stride the NT columns,
    // fixed-order butterfly) — a register-resident block_sum<20> here cost the whole kernel its occupancy
    __syncthreads();
    if (upd) {
      const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
      for (int k = wave; k < 5 * HMU; k += SDPLR_NT / 64) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < SDPLR_NT / 64; i++) t += accl[k * SDPLR_NT + wl + 64 * i];
        t = wave_sum(t);
        const int q = k / HMU, l = k % HMU;
        if (wl == 0 && l < h) slot_partials(partials, SLOT_GRAM + q * SDPLR_HMAX + l)[blockIdx.x] = t;
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) c->gram_pending = 1;  // consumed by k_lbfgs_boundary
    }
  }
  block_sum<2>(red, sh);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_GNORM2)[blockIdx.x] = red[0];
    if (COMMIT) {     // (otherwise k_ls_commit has produced the ‖pv‖² partials and armed the seam kernel)
      slot_partials(partials, SLOT_PVNORM2)[blockIdx.x] = red[1];
      if (blockIdx.x == 0) {
        c->norms_pending = 1;
        c->nb_gnorm = gridDim.x;
        c->nb_pvnorm = gridDim.x;
      }
    }
  }
}

// The singleton step kernel (k_fast_step2<LPR, VEC, HMU, true, …>) on the ring form of the history (k_dense.h): G_old and D
// are read at position ring_k of their rings, G_new goes to position ring_k + 1, and neither s_j nor y_j is stored — the Gram
// dots of the new pair with the other pairs are formed from y_i = G(ring_k − i) − G(ring_k − i − 1) and s_i = α_i·D(ring_k − 1 − i),
// the values the stored form would have read back: 2N bytes less written per iteration.  A row is one chunk (r ≤ LPR·VEC).
// PB = false: k_fast_step2<…, true, true, true> — P-less (G carried forward), the line-search scalar stage as its head.
// PB = true:  k_fast_step2<…, true> — P += α·W, G = 2·(y_g·P + d(y)∘R + Σ_c WS[c]·B[c]); the scalars come from k_ls_solve_fast.
template <int LPR, int VEC, int HMU, bool PB = false>
__global__ void __launch_bounds__(SDPLR_NT)
k_fast_step_ring(int n, int m, DevFast ff, double* __restrict__ R, const double* __restrict__ W, int r,
                 double* __restrict__ yvec, const double* __restrict__ lam, const double* __restrict__ lam_ub,
                 double* __restrict__ pv_raw, const double* __restrict__ lb, double* __restrict__ pv,
                 const double* __restrict__ A_RD, const double* __restrict__ A_DD, double* __restrict__ partials,
                 DevCtrl* __restrict__ c, int check_done, FactorArena A, int h, int nb_ls,
                 double* __restrict__ P = nullptr, DevLowRank lr = DevLowRank{}, const double* __restrict__ WS = nullptr) {
  static_assert(HMU >= 1, "the ring form rides the fused update");
  __shared__ double sh[2 * (SDPLR_NT / 64)];
  __shared__ double shl[10 * (SDPLR_NT / 64)];
  extern __shared__ double accl[];  // [5·HMU][NT] running Gram sums, one column per lane
  const int dn = check_done ? c->done : 0;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  const double sigma = c->sigma;
  const int latest = c->latest, rk = c->ring_k, rn = c->ring_n, rj0 = c->ring_j0;
  double a, yg = 1.0;
  bool upd;
  if constexpr (PB) {
    a = c->alpha;
    yg = yvec[ff.gid_g];
    upd = c->reldelta_exit == 0;
  } else {
    const double obj0 = c->obj, amax = c->alpha_max, last = c->lastval, feps = c->fprec_eps;
    double s10[10];
    {
      constexpr int PT = 2;
      double2 v[10][PT];
      const int ncols = (nb_ls + 2 * SDPLR_NT - 1) / (2 * SDPLR_NT);
#pragma unroll
      for (int q = 0; q < PT; q++) {
        const int i = (int)threadIdx.x + SDPLR_NT * q;
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
        s10[k] = 0.0;
#pragma unroll
        for (int q = 0; q < PT; q++) {
          const int i = 2 * ((int)threadIdx.x + SDPLR_NT * q);
          s10[k] += (i < nb_ls) ? v[k][q].x : 0.0;
          s10[k] += (i + 1 < nb_ls) ? v[k][q].y : 0.0;
        }
      }
    }
    block_sum<10>(s10, shl);
    const double g_rd = s10[8] + s10[8], g_dd = s10[9];
    double bq[5];
    bq[0] = obj0 - s10[0] + sigma * s10[1] / 2;      // src/linesearch.jl:44-56
    bq[1] = g_rd - s10[2] + sigma * s10[3];
    bq[2] = g_dd - s10[4] + sigma * s10[5] / 2;
    bq[3] = sigma * s10[6];
    bq[4] = sigma * s10[7] / 2;
    double al = 0.0, f = bq[0];
    const int rc = quartic_argmin(bq, amax, &al, &f);
    if (rc != 0) {
      if (blockIdx.x == 0 && threadIdx.x == 0 && !dn) {
        c->err = rc;
        c->done = 1;
      }
      return;
    }
    const double rel_delta = (last - f) / fmax(1.0, fmax(fabs(f), fabs(last)));   // src/sdplr.jl:238
    upd = !(rel_delta < feps);
    a = al;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !dn) {
      for (int k = 0; k < 5; k++) c->biquad[k] = bq[k];
      c->alpha = al;
      c->L = f;
      c->reldelta_exit = upd ? 0 : 1;
      const_cast<double*>(A_RD)[m] = g_rd;
      const_cast<double*>(A_DD)[m] = g_dd;
      c->obj_next = obj0 + al * (al * g_dd + g_rd);   // src/linesearch.jl:118-121: stored by the seam
      c->obj_pending = 1;
      c->pv2_extra = 0.0;
    }
  }
  const int jslot = latest % h;
  // the pairs that stay: i = 0 (newest) … nv − 1 (the h-th is the one the new pair replaces); of those the first rn are in
  // ring form, the others still in their slots as stored
  const int nv = h - 1;
  constexpr int HO = HMU > 1 ? HMU - 1 : 1;
  const double* gp[HMU];     // gp[0] = G_old; ring pair i: G(ring_k − i − 1) at gp[i + 1]; stored pair i: its y at gp[i + 1]
  const double* dp[HO];      // D(ring_k − 1 − i), or the stored s
  double ra[HO];
  int lsl[HO];               // pair slot of the i-th newest pair
  gp[0] = ring_G(A, rk, rj0);
#pragma unroll
  for (int i = 0; i < HO; i++) {
    int l = (latest - 1 - i) % h;
    if (l < 0) l += h;
    lsl[i] = l;
    const double r0 = c->ring_alpha[l];
    ra[i] = (i < rn) ? r0 : 1.0;
    if (i >= nv) {
      dp[i] = gp[0];
      if (i + 1 < HMU) gp[i + 1] = gp[0];
    } else if (i < rn) {
      dp[i] = ring_D(A, ring_back(rk, 1 + i, h), rj0);
      if (i + 1 < HMU) gp[i + 1] = ring_G(A, ring_back(rk, 1 + i, h), rj0);
    } else {
      dp[i] = aslot(A, AS_S0 + l);
      if (i + 1 < HMU) gp[i + 1] = aslot(A, as_y0(A) + l);
    }
  }
  double* const D = ring_D(A, rk, rj0);
  double* const Gnew = ring_G(A, ring_back(rk, -1, h), rj0);
  double red[2] = {0.0, 0.0};  // ‖G‖², ‖pv‖² (row-attached slots)
  const int ch0 = lane * VEC;
#pragma unroll
  for (int k = 0; k < 5 * HMU; k++) accl[k * SDPLR_NT + threadIdx.x] = 0.0;
  constexpr int TR = LPR < SDPLR_STEP_TR ? LPR : SDPLR_STEP_TR;   // rows per tile
  const long long ntiles = ((long long)n + TR - 1) / TR;
  for (long long tile = (long long)blockIdx.x * G + threadIdx.x / LPR; tile < ntiles; tile += total) {
    const long long j0 = tile * TR;
    const int nrows = (int)min((long long)TR, (long long)n - j0);
    if (dn) return;
    double djl = 0.0, djo = 0.0;     // d_j at the new / the old multipliers
    if (lane < nrows) {
      const long long j = j0 + lane;
      const int e0 = ff.drow_ptr[j], e1 = ff.drow_ptr[j + 1];
      for (int e = e0; e < e1; e++) {
        const int k = ff.drow_gid[e];
        if (!PB) djo += ff.drow_val[e] * yvec[k];                   // y as the previous g! left it
        const double v = pv_raw[k] + a * (a * A_DD[k] + A_RD[k]);   // src/linesearch.jl:118
        double yk;
        if (k < m) {
          yk = -fmin(lam_ub[k], lam[k] - sigma * v);                // src/coreop.jl:233
          const double pc = fmax(v, lb[k]);                         // src/linesearch.jl:122-124
          pv[k] = pc;
          red[1] += pc * pc;
        } else {
          yk = 1.0;                                                  // the cost slot, src/coreop.jl:235
          c->obj = v;
        }
        djl += ff.drow_val[e] * yk;
        yvec[k] = yk;
        pv_raw[k] = v;
      }
    }
    for (int i0 = 0; i0 < nrows; i0 += 2) {
      vecd<VEC> xx[2], pq[2], dd[2], ww[2], gv[2][HMU], dv[2][HO];
      double dj[2], dq[2];
      unsigned off[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = min(i0 + u, nrows - 1);                        // (odd tail: row repeated, not stored twice)
        const long long j = j0 + i;
        dj[u] = __shfl(djl, i, LPR);
        dq[u] = __shfl(djo, i, LPR);
        off[u] = (unsigned)(((unsigned long long)j * (unsigned)r + (unsigned)ch0) * 8ull);
#pragma unroll
        for (int q = 0; q < VEC; q++) xx[u].v[q] = pq[u].v[q] = dd[u].v[q] = ww[u].v[q] = 0.0;
#pragma unroll
        for (int l = 0; l < HMU; l++)
#pragma unroll
          for (int q = 0; q < VEC; q++) gv[u][l].v[q] = 0.0;
#pragma unroll
        for (int l = 0; l < HO; l++)
#pragma unroll
          for (int q = 0; q < VEC; q++) dv[u][l].v[q] = 0.0;
        if (ch0 < r) {
          xx[u] = ldrow<VEC>(rowat(R, off[u]));
          if (PB) pq[u] = ldrow<VEC>(rowat(P, off[u]));
          dd[u] = ldrow<VEC>(rowat(D, off[u]));
          ww[u] = ldrow<VEC>(rowat(W, off[u]));
#pragma unroll
          for (int l = 0; l < HMU; l++) gv[u][l] = ldrow_nt<VEC>(rowat(gp[l], off[u]));
#pragma unroll
          for (int l = 0; l < HO; l++) dv[u][l] = ldrow_nt<VEC>(rowat(dp[l], off[u]));
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (i0 + u >= nrows) break;
        if (ch0 >= r) continue;
        vecd<VEC> x = xx[u];
        const vecd<VEC> d = dd[u], w = ww[u], go = gv[u][0];
        vecd<VEC> g;
        if constexpr (PB) {
          vecd<VEC> pp = pq[u];
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            x.v[q] += a * d.v[q];
            pp.v[q] += a * w.v[q];
            g.v[q] = pp.v[q] * yg + x.v[q] * dj[u];
          }
          const long long jrow = j0 + i0 + u;
          for (int cc = 0; cc < lr.ST; cc++) {
            const double bb = lr.Bcat[(long long)cc * n + jrow];
            const vecd<VEC> ws = ldrow<VEC>(WS + (long long)cc * r + ch0);
#pragma unroll
            for (int q = 0; q < VEC; q++) g.v[q] += ws.v[q] * bb;
          }
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            g.v[q] *= 2.0;
            red[0] += g.v[q] * g.v[q];
          }
          strow<VEC>(rowat(P, off[u]), pp);
        } else {
          // G_new = G_old + 2·(α·W + d_new∘R_new − d_old∘R_old)   (y_g ≡ 1: A_g is the cost matrix)
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            const double xo = x.v[q];
            x.v[q] += a * d.v[q];
            const double t = a * w.v[q] + (x.v[q] * dj[u] - xo * dq[u]);
            g.v[q] = go.v[q] + 2.0 * t;
            red[0] += g.v[q] * g.v[q];
          }
        }
#ifdef SDPLR_RING_NT_R
        strow_nt<VEC>(rowat(R, off[u]), x);
#else
        strow<VEC>(rowat(R, off[u]), x);
#endif
#ifdef SDPLR_RING_NT_G
        strow_nt<VEC>(rowat(Gnew, off[u]), g);
#else
        strow<VEC>(rowat(Gnew, off[u]), g);
#endif
        if (upd) {
          vecd<VEC> sn, yn;
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            sn.v[q] = a * d.v[q];            // BLAS.scal!(stepsize, dir)  (src/lbfgs.jl:142): kept as (α, D)
            yn.v[q] = g.v[q] - go.v[q];      // y_j = −G_old + G_new  (:122,145): kept as (G_old, G_new)
          }
          double* ac = accl + threadIdx.x;
#pragma unroll
          for (int i = 0; i < HO; i++)
            if (i < nv) {
              double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0;
#pragma unroll
              for (int q = 0; q < VEC; q++) {
                const double sl = ra[i] * dv[u][i].v[q];
                const double yn1 = gv[u][i + 1 < HMU ? i + 1 : i].v[q];
                const double yl = (i < rn) ? gv[u][i].v[q] - yn1 : yn1;
                q0 += sn.v[q] * yl;
                q1 += sl * yn.v[q];
                q2 += yn.v[q] * yl;
                q3 += sl * g.v[q];
                q4 += yl * g.v[q];
              }
              const int l = lsl[i];
              (void)__hip_atomic_fetch_add(ac + (0 * HMU + l) * SDPLR_NT, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              (void)__hip_atomic_fetch_add(ac + (1 * HMU + l) * SDPLR_NT, q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              (void)__hip_atomic_fetch_add(ac + (2 * HMU + l) * SDPLR_NT, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              (void)__hip_atomic_fetch_add(ac + (3 * HMU + l) * SDPLR_NT, q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              (void)__hip_atomic_fetch_add(ac + (4 * HMU + l) * SDPLR_NT, q4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          {   // the new pair with itself
            double q0 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0;
#pragma unroll
            for (int q = 0; q < VEC; q++) {
              q0 += sn.v[q] * yn.v[q];
              q2 += yn.v[q] * yn.v[q];
              q3 += sn.v[q] * g.v[q];
              q4 += yn.v[q] * g.v[q];
            }
            (void)__hip_atomic_fetch_add(ac + (0 * HMU + jslot) * SDPLR_NT, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_add(ac + (1 * HMU + jslot) * SDPLR_NT, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_add(ac + (2 * HMU + jslot) * SDPLR_NT, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_add(ac + (3 * HMU + jslot) * SDPLR_NT, q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_add(ac + (4 * HMU + jslot) * SDPLR_NT, q4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  if (dn) return;
  __syncthreads();
  if (upd) {
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
    for (int k = wave; k < 5 * HMU; k += SDPLR_NT / 64) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < SDPLR_NT / 64; i++) t += accl[k * SDPLR_NT + wl + 64 * i];
      t = wave_sum(t);
      const int q = k / HMU, l = k % HMU;
      if (wl == 0 && l < h) slot_partials(partials, SLOT_GRAM + q * SDPLR_HMAX + l)[blockIdx.x] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) c->gram_pending = 1;  // consumed by k_lbfgs_boundary
  }
  block_sum<2>(red, sh);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_GNORM2)[blockIdx.x] = red[0];
    slot_partials(partials, SLOT_PVNORM2)[blockIdx.x] = red[1];
    if (blockIdx.x == 0) {
      c->norms_pending = 1;
      c->nb_gnorm = gridDim.x;
      c->nb_pvnorm = gridDim.x;
    }
  }
}

// W = A_g·D with everything row-local of the line-search head riding along (singleton fast path):
// while row j's neighbours are being gathered, R_j, D_j, P_j stream in, giving ⟨R_j,D_j⟩, ‖D_j‖², the
// partials of ⟨P,D⟩ and ⟨D,W⟩, the 𝒜 values of the row's singleton constraints and their share of the
// line-search sums (row dots, line-search sums and the SpMM in one sweep; the gather latency hides the extra streams).
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT)
k_spmm_fast(DevSparse sg, int m, DevFast ff, const double* __restrict__ R, const double* __restrict__ D,
            const double* __restrict__ P, double* __restrict__ W, int r, const double* __restrict__ lam,
            const double* __restrict__ pv_raw, double* __restrict__ A_RD, double* __restrict__ A_DD,
            double* __restrict__ partials, const DevCtrl* __restrict__ c, int check_done) {
  __shared__ double sh[10 * (SDPLR_NT / 64)];
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * G;
  const double sigma = c->sigma;
  double acc[10];  // 0..7 line-search sums, 8 ⟨P,D⟩, 9 ⟨D,W⟩
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  for (long long j = (long long)blockIdx.x * G + threadIdx.x / LPR; j < sg.n; j += total) {
    const int beg = sg.colptr[j], end = sg.colptr[j + 1];
    double rd = 0.0, dd = 0.0;
    for (int chb = 0; chb < r; chb += LPR * VEC) {
      // every lane of the group runs the loop (the index hand-off below is a shuffle); lanes past the row's
      // end (r not a multiple of LPR·VEC) just do not touch memory
      const int ch = chb + lane * VEC;
      const bool act = ch < r;
      vecd<VEC> xr, xd, w;
#pragma unroll
      for (int k = 0; k < VEC; k++) xr.v[k] = xd.v[k] = w.v[k] = 0.0;
      if (act) {
        xr = ldrow<VEC>(R + j * r + ch);
        xd = ldrow<VEC>(D + j * r + ch);
      }
      // The row's (index, value) pairs are fetched LPR at a time by the group's lanes and handed round with
      // shuffles, so the gathers — eight in flight per lane — never wait on an index load: per row the chain is
      // one index fetch + ⌈len/8⌉ gather rounds instead of ⌈len/4⌉ × (index fetch → gather).
      for (int base = beg; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_i = (lane < cnt) ? sg.rowval[base + lane] : 0;
        const double my_v = (lane < cnt) ? sg.nzval[base + lane] : 0.0;
        for (int k0 = 0; k0 < cnt; k0 += 8) {
          vecd<VEC> x[8];
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int src = min(k0 + q, LPR - 1);
            const long long i = __shfl(my_i, src, LPR);
            v[q] = __shfl(my_v, src, LPR);
            if (act && k0 + q < cnt) {
              x[q] = ldrow<VEC>(D + i * r + ch);
            } else {
#pragma unroll
              for (int k = 0; k < VEC; k++) x[q].v[k] = 0.0;
              v[q] = 0.0;
            }
          }
#pragma unroll
          for (int q = 0; q < 8; q++)
#pragma unroll
            for (int k = 0; k < VEC; k++) w.v[k] += x[q].v[k] * v[q];
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        rd += xr.v[k] * xd.v[k];
        dd += xd.v[k] * xd.v[k];
        acc[8] += xr.v[k] * w.v[k];   // ⟨R, W⟩ = ⟨R, A_g·D⟩ = ⟨A_g·R, D⟩ = ⟨P, D⟩ (A_g symmetric): no read of P here
        acc[9] += xd.v[k] * w.v[k];
      }
      if (act) strow<VEC>(W + j * r + ch, w);
    }
    rd = group_sum<LPR>(rd);
    dd = group_sum<LPR>(dd);
    if (lane == 0)
      for (int e = ff.drow_ptr[j]; e < ff.drow_ptr[j + 1]; e++) {
        const int k = ff.drow_gid[e];
        const double v = ff.drow_val[e];
        const double q1 = v * (rd + rd), q2 = v * dd;
        A_RD[k] = q1;
        A_DD[k] = q2;
        if (k < m) {
          const double l = lam[k], nq0 = pv_raw[k];
          acc[0] += l * nq0;
          acc[1] += nq0 * nq0;
          acc[2] += l * q1;
          acc[3] += nq0 * q1;
          acc[4] += (l - sigma * nq0) * q2;
          acc[5] += q1 * q1;
          acc[6] += q1 * q2;
          acc[7] += q2 * q2;
        }
      }
  }
  block_sum<10>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) slot_partials(partials, SLOT_LS + k)[blockIdx.x] = acc[k];
    slot_partials(partials, SLOT_PD)[blockIdx.x] = acc[8];
    slot_partials(partials, SLOT_DW)[blockIdx.x] = acc[9];
  }
}

// ---- fg! on the singleton fast path (src/coreop.jl:323-349) ---------------------------------------------------------
// The gather kernel run on (R, R) instead of (R, D) leaves P = A_g·R, v_k·‖R_j‖² = 𝒜(RRᵀ)_k for the row-attached constraints
// (in its A_DD output) and the block partials of ⟨R, P⟩ = ⟨A_g, RRᵀ⟩: f! and g! are then two m- / n-sized passes.
// k_fg_fast_tail: f!'s tail (:16-29) + copy2y_λ_sub_pvio! (:229-236) for the row-attached slots — one constraint per
// thread — and, by block 0, A_g's own slot from the ⟨R, P⟩ partials.  Partials: SLOT_F (Σ(ỹ² − λ²)/2σ), SLOT_PVNORM2.
__global__ void __launch_bounds__(SDPLR_NT)
k_fg_fast_tail(DevCtrl* __restrict__ c, int m, int gid_g, const double* __restrict__ AA, double* __restrict__ pv_raw,
               const double* __restrict__ b, const double* __restrict__ lb, double* __restrict__ pv,
               const double* __restrict__ lam, const double* __restrict__ lam_ub, double* __restrict__ y,
               int nb_pd, double* __restrict__ partials) {
  __shared__ double sh[8];
  const double sigma = c->sigma;
  double fs = 0.0, pn = 0.0;
  if (blockIdx.x == 0) {   // A_g's slot: 𝒜(RRᵀ)_g = ⟨R, P⟩
    const double v0 = reduce_partials(slot_partials(partials, SLOT_PD), nb_pd, sh);
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = v0, yk = 1.0;                               // (the cost slot: y = 1, src/coreop.jl:235)
      if (gid_g < m) {
        v -= b[gid_g];
        const double pc = fmax(v, lb[gid_g]);
        pv[gid_g] = pc;
        pn += pc * pc;
        const double l = lam[gid_g], yt = fmin(lam_ub[gid_g], l - sigma * v);
        fs += (yt * yt - l * l) / (2 * sigma);
        yk = -yt;
      }
      pv_raw[gid_g] = v;
      y[gid_g] = yk;
    }
  }
  const int stride = gridDim.x * SDPLR_NT;
  for (int i = blockIdx.x * SDPLR_NT + threadIdx.x; i <= m; i += stride) {
    if (i == gid_g) continue;
    if (i == m) {   // the cost matrix as a row-attached entry
      pv_raw[i] = AA[i];
      y[i] = 1.0;
      continue;
    }
    const double v = AA[i] - b[i];                           // (:20)
    pv_raw[i] = v;
    const double pc = fmax(v, lb[i]);                        // (:22)
    pv[i] = pc;
    pn += pc * pc;
    const double l = lam[i], yt = fmin(lam_ub[i], l - sigma * v);   // (:27)
    fs += (yt * yt - l * l) / (2 * sigma);                   // (:28)
    y[i] = -yt;                                              // src/coreop.jl:233
  }
  double two[2] = {fs, pn};
  __syncthreads();
  __shared__ double sh2[2 * (SDPLR_NT / 64)];
  block_sum<2>(two, sh2);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_F)[blockIdx.x] = two[0];
    slot_partials(partials, SLOT_PVNORM2)[blockIdx.x] = two[1];
  }
}
// G = 2·(y_g·P + d(y)∘R) (src/coreop.jl:305-317 on the structured form) with the ‖G‖² partials
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT)
k_fg_fast_G(int n, DevFast ff, const double* __restrict__ R, const double* __restrict__ P, double* __restrict__ G, int r,
            const double* __restrict__ yvec, double* __restrict__ partials) {
  __shared__ double sh[8];
  constexpr int GR = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR;
  const long long total = (long long)gridDim.x * GR;
  const double yg = yvec[ff.gid_g];
  double nrm = 0.0;
  for (long long j = (long long)blockIdx.x * GR + threadIdx.x / LPR; j < n; j += total) {
    double dj = 0.0;
    for (int e = ff.drow_ptr[j]; e < ff.drow_ptr[j + 1]; e++) dj += ff.drow_val[e] * yvec[ff.drow_gid[e]];
    for (int ch = lane * VEC; ch < r; ch += LPR * VEC) {
      const vecd<VEC> x = ldrow<VEC>(R + j * r + ch), pp = ldrow<VEC>(P + j * r + ch);
      vecd<VEC> g;
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        g.v[q] = pp.v[q] * yg + x.v[q] * dj;
        g.v[q] *= 2.0;
        nrm += g.v[q] * g.v[q];
      }
      strow<VEC>(G + j * r + ch, g);
    }
  }
  nrm = block_sum1(nrm, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_GNORM2)[blockIdx.x] = nrm;
}
// obj, ℒ, grad_norm, primal_vio_norm (src/coreop.jl:16,25-30,334-347).  One block.
__global__ void __launch_bounds__(SDPLR_NT)
k_fg_fast_fin(DevCtrl* __restrict__ c, int m, const double* __restrict__ pv_raw, int nb_m, int nb_g,
              const double* __restrict__ partials) {
  __shared__ double sh[8];
  const double fs = reduce_partials(slot_partials(partials, SLOT_F), nb_m, sh);
  __syncthreads();
  const double p2 = reduce_partials(slot_partials(partials, SLOT_PVNORM2), nb_m, sh);
  __syncthreads();
  const double g2 = reduce_partials(slot_partials(partials, SLOT_GNORM2), nb_g, sh);
  if (threadIdx.x != 0) return;
  const double obj = pv_raw[m];
  c->obj = obj;
  c->L = obj + fs;
  const double g = sqrt(g2), pq = sqrt(p2);
  c->gnorm = c->grel ? g / c->normC : g;
  c->pvnorm = c->prel ? pq / c->normb : pq;
}

// ---- column-sweep form of k_spmm_fast --------------------------------------------------------------------
// A sub-wave group owns a tile of K consecutive rows.  The tile's nonzeros are stored sorted by COLUMN
// (entry = local row << SDPLR_TILE_COLBITS | column, value), so every group walks its list from column 0 to column n−1, and
// since the lists are equally long on average all the groups resident on the chip read the same narrow band
// of D at the same time: the band lives in the XCD's 4 MiB L2 instead of each row gather going out to the
// Infinity Cache (a uniformly random gather keeps only 4 MiB / |D| of its rows in L2).  The K partial rows
// live in LDS (private to the group, read-modify-write in list order, so the sums are formed in the same
// order as in k_spmm_fast — per row by increasing column — and W is bit-identical).  One list per group
// also removes the per-row pointer → index → gather dependency chain: the next index chunk is fetched
// while the current gathers are in flight.
#define SDPLR_TILE_COLBITS 24
#ifndef SDPLR_TILE_WIN
#define SDPLR_TILE_WIN 4   /* gathers in flight per group: 2 and 4 equal, 8 and 16 slower (L1/TA-bound, not latency-bound) */
#endif
struct DevTile {
  int K, n_tiles;        // K = most rows in any tile
  const int* row0;       // [n_tiles + 1] first row of each tile
  const int* ptr;        // [n_tiles + 1]
  const int* ent;        // local row << SDPLR_TILE_COLBITS (24) | column
  const double* val;
  // UNIFORM form (every off-diagonal entry of A_g has the same value — unit-weight graphs): the lists hold the
  // off-diagonal entries only and no values (4 bytes per nonzero instead of 12; the loop loses its value load, its two
  // broadcasts and its multiplies: the gathered rows are SUMMED, the sum is scaled once); the diagonal is a vector.
  double one;
  const double* gdiag;   // [n], or null: the general form above
};

// value held by lane `src` (0 ≤ src < LPR, constant after unrolling) of the caller's LPR-lane group
template <int LPR>
__device__ __forceinline__ int group_bcast(int v, int src) {
  if constexpr (LPR == 16) {  // the group is one DPP row: v_mov_b32_dpp row_newbcast, no LDS crossbar
    switch (src) {
#define SDPLR_BC(i) case i: return __builtin_amdgcn_mov_dpp(v, 0x150 + i, 0xf, 0xf, true);
      SDPLR_BC(0) SDPLR_BC(1) SDPLR_BC(2) SDPLR_BC(3) SDPLR_BC(4) SDPLR_BC(5) SDPLR_BC(6) SDPLR_BC(7)
      SDPLR_BC(8) SDPLR_BC(9) SDPLR_BC(10) SDPLR_BC(11) SDPLR_BC(12) SDPLR_BC(13) SDPLR_BC(14)
#undef SDPLR_BC
      default: return __builtin_amdgcn_mov_dpp(v, 0x150 + 15, 0xf, 0xf, true);
    }
  } else {
    return __shfl(v, src, LPR);
  }
}
template <int LPR>
__device__ __forceinline__ double group_bcast(double v, int src) {
  return __hiloint2double(group_bcast<LPR>(__double2hiint(v), src), group_bcast<LPR>(__double2loint(v), src));
}

// (launch bounds: the tile's LDS rows leave room for two blocks per CU — two waves per SIMD — whatever the second
// argument says; for rows wider than a DPP row (LPR ≥ 32: the hand-offs are ds_bpermute results that stay live across the
// unrolled chunk) the 128-VGPR cap of "4" cost 20–268 bytes of scratch per lane inside the gather loop)
template <int LPR, int VEC, int LRN, bool UNI = false>
#ifdef SDPLR_PROBE_TILE_HIST
__global__ void __launch_bounds__(SDPLR_NT, 2)
#else
__global__ void __launch_bounds__(SDPLR_NT, LPR >= 32 ? 2 : 4)
#endif
k_spmm_tile(DevTile tl, int n, int m, DevFast ff, const double* __restrict__ R, const double* __restrict__ Darg,
            const double* __restrict__ P, double* __restrict__ W, int r, const double* __restrict__ lam,
            const double* __restrict__ pv_raw, double* __restrict__ A_RD, double* __restrict__ A_DD,
            double* __restrict__ partials, const DevCtrl* __restrict__ c, int check_done,
            DevLowRank lr, double* __restrict__ lr_part, int rowout, int ring = 0, FactorArena RA = FactorArena{}) {
  // [G][K+1][LPR·VEC] partial rows, [G][K][2] row dots, [G][8] line-search sums
  extern __shared__ double tile_lds[];
  __shared__ double sh[10 * (SDPLR_NT / 64)];
  const int dn = check_done ? c->done : 0;  // tested below, once the first tile's pointers are on their way
  // ring form of the history (k_dense.h): the direction sits at position ring_k of the D ring
  const double* __restrict__ D = ring ? ring_D(RA, c->ring_k, c->ring_j0) : Darg;
  constexpr int G = SDPLR_NT / LPR;
  constexpr int RW = LPR * VEC;
  const int lane = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int K = tl.K;
  // row k of this group: rows + k·RW; row K is a dump for the ring slots past the end of the list
  double* rows = tile_lds + (size_t)grp * (K + 1) * RW + lane * VEC;
  double* dots = tile_lds + (size_t)G * (K + 1) * RW + (size_t)grp * K * 2;
  double* ls = tile_lds + (size_t)G * (K + 1) * RW + (size_t)G * K * 2 + (size_t)grp * 8;  // per group
  // per WAVE: running low-rank projections [2·LRN][RW] (kept out of the registers the ring needs)
  double* lrw = tile_lds + (size_t)G * (K + 1) * RW + (size_t)G * (K * 2 + 8) +
                (size_t)(threadIdx.x >> 6) * (2 * LRN * RW) + lane * VEC;
  const double sigma = c->sigma;
  double pd = 0.0, dw = 0.0;  // ⟨P,D⟩, ⟨D,W⟩
#ifdef SDPLR_PROBE_TILE_HIST
  double probe_acc = 0.0;
#endif
  // LRN > 0 (single chunk only): the projections RᵀB_c and DᵀB_c of k_lr_project<…,2> for the first LRN = ST
  // low-rank columns ride on the rows the epilogue loads anyway (src/coreop.jl:125-126) — one pass over R and D
  // saved per iteration
  constexpr int LRA = LRN > 0 ? LRN : 1;
  if ((threadIdx.x & 63) < LPR) {
#pragma unroll
    for (int cc = 0; cc < 2 * LRN; cc++)
#pragma unroll
      for (int q = 0; q < VEC; q++) lrw[cc * RW + q] = 0.0;
  }
  for (int k = lane; k < 8; k += LPR) ls[k] = 0.0;
  for (long long tile = (long long)blockIdx.x * G + grp; tile < tl.n_tiles; tile += (long long)gridDim.x * G) {
    const int beg = tl.ptr[tile], end = tl.ptr[tile + 1];
    const long long j0 = tl.row0[tile];
    const int nrows = tl.row0[tile + 1] - (int)j0;
    if (dn) return;
    for (int k = lane; k < K; k += LPR) dots[2 * k] = dots[2 * k + 1] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int chb = 0; chb < r; chb += RW) {
      const int ch = chb + lane * VEC;
      const bool act = ch < r;
      for (int k = 0; k < K; k++)
#pragma unroll
        for (int q = 0; q < VEC; q++) rows[k * RW + q] = 0.0;
      // (ce, cv): the list chunk being consumed, one entry per lane; (ne, nv): the chunk after it.  A ring of
      // WIN row gathers stays in flight: as soon as entry t has been folded into its LDS row, the gather of
      // entry t + WIN takes its slot, so the memory pipe never drains between batches.
      // The loop body is branch-free and as short as it can be made — at four entries per wave instruction
      // its VALU issue, not memory, was the limit: the host pads every list to a multiple of LPR with dump
      // entries (so no validity tests), the column sits in the low 24 bits of the entry (v_mad_u32_u24 reads
      // just those: byte offset = column·row bytes + channel bytes, 32 bits, added to the uniform base by
      // the load itself), and for LPR = 16 the hand-offs are DPP row broadcasts, not LDS-crossbar shuffles.
      constexpr int WIN = LPR < SDPLR_TILE_WIN ? LPR : SDPLR_TILE_WIN;
      const unsigned rowb = (unsigned)r * 8u, chb8 = (unsigned)(act ? ch : 0) * 8u;
      const char* Db = reinterpret_cast<const char*>(D);
      const int dump = (K << SDPLR_TILE_COLBITS) | (int)j0;  // row K of the group's LDS rows is a dump
      int ce = tl.ent[beg + lane], ne = tl.ent[beg + LPR + lane];
      double cv = 0.0, nv = 0.0;
      if (!UNI) {
        cv = tl.val[beg + lane];
        nv = tl.val[beg + LPR + lane];
      }
      vecd<VEC> x[WIN];
      double* fold[WIN];  // LDS row the slot's gather will be added to
#pragma unroll
      for (int q = 0; q < WIN; q++) {
        int e = group_bcast<LPR>(ce, q);
        e = (beg < end) ? e : dump;
        fold[q] = rows + ((unsigned)e >> SDPLR_TILE_COLBITS) * (unsigned)RW;
        x[q] = ldrow<VEC>(reinterpret_cast<const double*>(Db + (__umul24((unsigned)e, rowb) + chb8)));
      }
      // enter the loop with nothing but the ring outstanding: the compiler's wait-count merge at the loop
      // header is then exact instead of falling back to vmcnt(0) on every trip
      __builtin_amdgcn_sched_barrier(0);
      for (int base = beg; base < end; base += LPR) {
        const bool more = base + LPR < end;
        // the chunk after next, fetched first so that by the time it is rotated in (a register copy, which
        // must wait for the load) a whole chunk of gathers has been issued behind it and nothing drains
        // (the arrays carry 2·64 entries of tail padding, so this never reads past their end)
        const int fe = tl.ent[base + 2 * LPR + lane];
        double fv = 0.0;
        if (!UNI) fv = tl.val[base + 2 * LPR + lane];
#pragma unroll
        for (int k0 = 0; k0 < LPR; k0 += WIN) {
#pragma unroll
          for (int q = 0; q < WIN; q++) {
            // fold one …  (an LDS atomic add without return, ds_add_f64: the LDS unit applies a wave's adds in
            // issue order and each lane owns its addresses, so the sum is formed in list order exactly as a
            // read-add-write would form it, but the wave never waits for it)
            double v = 1.0;
            if (!UNI) v = group_bcast<LPR>(cv, k0 + q);
#pragma unroll
            for (int k = 0; k < VEC; k++)
              (void)__hip_atomic_fetch_add(fold[q] + k, UNI ? x[q].v[k] : x[q].v[k] * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // … refill one
            int e2;
            if (k0 + WIN < LPR) {
              e2 = group_bcast<LPR>(ce, (k0 + WIN + q) % LPR);
            } else {
              e2 = group_bcast<LPR>(ne, q);
              e2 = more ? e2 : dump;   // past the end of the list: nothing to gather
            }
            fold[q] = rows + ((unsigned)e2 >> SDPLR_TILE_COLBITS) * (unsigned)RW;
            x[q] = ldrow<VEC>(reinterpret_cast<const double*>(Db + (__umul24((unsigned)e2, rowb) + chb8)));
            __builtin_amdgcn_sched_barrier(0);  // no clustering of the refills behind the folds
          }
        }
        ce = ne;
        cv = nv;
        ne = fe;
        nv = fv;
      }
      constexpr int EB = LRN > 0 ? 2 : 4;   // rows per step (the projections need registers of their own)
      // the tile's rows, EB at a time (one memory round trip for twelve row loads, eight interleaved group
      // sums): W out, row dots, ⟨P,D⟩ and ⟨D,W⟩ partials.  Loads are clamped, not predicated.
      double lr0[LRA][VEC], lr1[LRA][VEC];
#pragma unroll
      for (int cc = 0; cc < LRA; cc++)
#pragma unroll
        for (int q = 0; q < VEC; q++) lr0[cc][q] = lr1[cc][q] = 0.0;
      for (int k = 0; k < nrows; k += EB) {
        vecd<VEC> xr[EB], xd[EB], w[EB];
        const long long chs = act ? ch : 0;
#pragma unroll
        for (int i = 0; i < EB; i++) {
          const long long j = j0 + min(k + i, nrows - 1);
          xr[i] = ldrow<VEC>(R + j * r + chs);
          xd[i] = ldrow<VEC>(D + j * r + chs);
#pragma unroll
          for (int q = 0; q < VEC; q++) w[i].v[q] = rows[min(k + i, nrows - 1) * RW + q];
          if (UNI) {   // scale the sum of the gathered rows once; the diagonal entry, left out of the lists, last
            const double gd = tl.gdiag[j];
#pragma unroll
            for (int q = 0; q < VEC; q++) w[i].v[q] = w[i].v[q] * tl.one + xd[i].v[q] * gd;
          }
        }
#ifdef SDPLR_PROBE_TILE_HIST
#pragma unroll
        for (int i0 = 0; i0 < EB; i0 += 2) {
          vecd<VEC> hx[2][7];
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const long long j = j0 + min(k + i0 + i, nrows - 1);
#pragma unroll
            for (int l = 0; l < 7; l++) hx[i][l] = ldrow_nt<VEC>(ff.probe_hist[l] + j * r + chs);
          }
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int l = 0; l < 7; l++)
#pragma unroll
              for (int q = 0; q < VEC; q++) {
                double t1 = hx[i][l].v[q] * w[i0 + i].v[q], t2 = hx[i][l].v[q] * xd[i0 + i].v[q], t3 = hx[i][l].v[q] * xr[i0 + i].v[q];
                probe_acc += (t1 + t2) + t3;
              }
        }
#endif
        double rd[EB], dd[EB];
#pragma unroll
        for (int i = 0; i < EB; i++) {
          const bool ok = act && k + i < nrows;
          if (ok) strow<VEC>(W + (j0 + k + i) * r + ch, w[i]);
          rd[i] = dd[i] = 0.0;
          double tp = 0.0, tw = 0.0;
#pragma unroll
          for (int q = 0; q < VEC; q++) {
            rd[i] += xr[i].v[q] * xd[i].v[q];
            dd[i] += xd[i].v[q] * xd[i].v[q];
            tp += xr[i].v[q] * w[i].v[q];    // ⟨R, W⟩ = ⟨P, D⟩ (A_g symmetric): the P stream stays out of this kernel
            tw += xd[i].v[q] * w[i].v[q];
          }
          rd[i] = ok ? rd[i] : 0.0;
          dd[i] = ok ? dd[i] : 0.0;
          pd += ok ? tp : 0.0;
          dw += ok ? tw : 0.0;
#pragma unroll
          for (int cc = 0; cc < LRN; cc++) {
            const double b = ok ? lr.Bcat[(long long)cc * n + j0 + k + i] : 0.0;
#pragma unroll
            for (int q = 0; q < VEC; q++) {
              lr0[cc][q] += xr[i].v[q] * b;
              lr1[cc][q] += xd[i].v[q] * b;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < EB; i++) {
          rd[i] = group_sum<LPR>(rd[i]);
          dd[i] = group_sum<LPR>(dd[i]);
        }
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < EB; i++)
            if (k + i < nrows) {
              dots[2 * (k + i)] += rd[i];
              dots[2 * (k + i) + 1] += dd[i];
            }
        }
      }
#pragma unroll
      for (int cc = 0; cc < LRN; cc++)
#pragma unroll
        for (int q = 0; q < VEC; q++) {   // the wave's groups first (fixed order), then into the wave's LDS slot
          double a0 = lr0[cc][q], a1 = lr1[cc][q];
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) {
            a0 += __shfl_xor(a0, o, 64);
            a1 += __shfl_xor(a1, o, 64);
          }
          if ((threadIdx.x & 63) < LPR) {
            lrw[cc * RW + q] += a0;
            lrw[(LRN + cc) * RW + q] += a1;
          }
        }
      __builtin_amdgcn_wave_barrier();
    }
    // the constraints attached to the tile's rows, one lane per row; their line-search sums are reduced over
    // the group and kept in LDS (8 doubles per group), not in registers the gather ring needs
    double t8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t8[k] = 0.0;
    for (int k = lane; k < nrows; k += LPR) {
      const long long j = j0 + k;
      const double rd = dots[2 * k], dd = dots[2 * k + 1];
      if (rowout) {   // general diagonal-only matrices (k_rowdots's output): the dots go where k_sddmm<…,2> would
        const int q = ff.diagpos[j];      // put them, A_RD / A_DD here being the two UVt arrays; k_segreduce follows
        if (q >= 0) {
          A_RD[q] = rd + rd;
          A_DD[q] = dd;
        }
        continue;
      }
      const int e0 = ff.drow_ptr[j], e1 = ff.drow_ptr[j + 1];
      for (int e = e0; e < e1; e++) {
        const int gid = ff.drow_gid[e];
        const double v = ff.drow_val[e];
        const double q1 = v * (rd + rd), q2 = v * dd;
        A_RD[gid] = q1;
        A_DD[gid] = q2;
        if (gid < m) {
          const double l = lam[gid], nq0 = pv_raw[gid];
          t8[0] += l * nq0;
          t8[1] += nq0 * nq0;
          t8[2] += l * q1;
          t8[3] += nq0 * q1;
          t8[4] += (l - sigma * nq0) * q2;
          t8[5] += q1 * q1;
          t8[6] += q1 * q2;
          t8[7] += q2 * q2;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) t8[k] = group_sum<LPR>(t8[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) ls[k] += t8[k];
    }
    __builtin_amdgcn_wave_barrier();
  }
  double lsk[8];
#pragma unroll
  for (int k = 0; k < 8; k++) lsk[k] = (lane == 0) ? ls[k] : 0.0;
  if constexpr (LRN > 0) {  // block partials of the projections, in k_lr_project's layout (block index fastest)
    __syncthreads();
    const double* all = tile_lds + (size_t)G * (K + 1) * RW + (size_t)G * (K * 2 + 8);
    for (int t = threadIdx.x; t < 2 * LRN * RW; t += SDPLR_NT) {
      const int ch = t % RW;
      if (ch < r) {
        double sum = 0.0;
        for (int w = 0; w < SDPLR_NT / 64; w++) sum += all[(size_t)w * (2 * LRN * RW) + t];
        // t = (f·LRN + cc)·RW + ch
        lr_part[((long long)(t / RW / LRN) * lr.ST + (t / RW) % LRN) * r * gridDim.x + (long long)ch * gridDim.x + blockIdx.x] = sum;
      }
    }
  }
  double acc[10];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = lsk[k];
  acc[8] = pd;
  acc[9] = dw;
  block_sum<10>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) slot_partials(partials, SLOT_LS + k)[blockIdx.x] = acc[k];
    slot_partials(partials, SLOT_PD)[blockIdx.x] = acc[8];
    slot_partials(partials, SLOT_DW)[blockIdx.x] = acc[9];
  }
#ifdef SDPLR_PROBE_TILE_HIST
  if (probe_acc == 1.2345e301) slot_partials(partials, SLOT_V0)[blockIdx.x] = probe_acc;   // keeps the probe's loads alive
#endif
}

// ---- hub rows: one block per row of the full pattern with more than long_thresh nonzeros ------------------
// Same result as k_spmm for those rows (which k_spmm skips): the row's nonzeros are dealt round-robin to the
// block's sub-wave groups, the group partials are added in group order (deterministic).  Partials of the
// norm / dot go to slot entries [pbase, pbase + n_long_rows).
template <int LPR, int VEC, int HMU = 0, bool EDGE = false>
__device__ __forceinline__ void
spmm_long_rows(DevSparse sp, const double* __restrict__ X, double* Y, int r, double scale,
               DevLowRank lr, const double* __restrict__ WS, int slot, int pbase, double* __restrict__ partials,
               const DevCtrl* __restrict__ c, int check_done, const double* __restrict__ Xdot, int bid, int nblk,
               const UpdCtx* u = nullptr, const double* __restrict__ yv = nullptr) {
  __shared__ double shg[SDPLR_NT * VEC];
  __shared__ double sh[8];
  if (check_done && c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  const int lane = threadIdx.x % LPR, g = threadIdx.x / LPR;
  double nrm = 0.0;
  for (int lrow = bid; lrow < sp.n_long_rows; lrow += nblk) {
  const long long j = sp.long_rows[lrow];
  const int beg = sp.colptr[j], end = sp.colptr[j + 1];
  for (int chb = 0; chb < r; chb += LPR * VEC) {
    const int ch = chb + lane * VEC;
    vecd<VEC> acc;
#pragma unroll
    for (int k = 0; k < VEC; k++) acc.v[k] = 0.0;
    if (ch < r) {
      int p = beg + g;
      for (; p + 7 * G < end; p += 8 * G) {  // eight independent row gathers in flight per group: the longest
        long long ii[8];                      // hub row sets this kernel's duration
        double vv[8];
        vecd<VEC> xx[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          ii[q] = sp.rowval[p + q * G];
          vv[q] = s_value<EDGE>(sp, yv, p + q * G);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) xx[q] = ldrow<VEC>(X + ii[q] * r + ch);
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
          for (int k = 0; k < VEC; k++) acc.v[k] += xx[q].v[k] * vv[q];
      }
      for (; p + 3 * G < end; p += 4 * G) {  // four
        const long long i0 = sp.rowval[p], i1 = sp.rowval[p + G], i2 = sp.rowval[p + 2 * G], i3 = sp.rowval[p + 3 * G];
        const double v0 = s_value<EDGE>(sp, yv, p), v1 = s_value<EDGE>(sp, yv, p + G), v2 = s_value<EDGE>(sp, yv, p + 2 * G), v3 = s_value<EDGE>(sp, yv, p + 3 * G);
        const vecd<VEC> x0 = ldrow<VEC>(X + i0 * r + ch), x1 = ldrow<VEC>(X + i1 * r + ch);
        const vecd<VEC> x2 = ldrow<VEC>(X + i2 * r + ch), x3 = ldrow<VEC>(X + i3 * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          acc.v[k] += x0.v[k] * v0;
          acc.v[k] += x1.v[k] * v1;
          acc.v[k] += x2.v[k] * v2;
          acc.v[k] += x3.v[k] * v3;
        }
      }
      for (; p < end; p += G) {
        const long long i = sp.rowval[p];
        const double v = s_value<EDGE>(sp, yv, p);
        const vecd<VEC> x = ldrow<VEC>(X + i * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) acc.v[k] += x.v[k] * v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VEC; k++) shg[(g * LPR + lane) * VEC + k] = acc.v[k];
    __syncthreads();
    if (g == 0 && ch < r) {
      vecd<VEC> tot;
#pragma unroll
      for (int k = 0; k < VEC; k++) tot.v[k] = 0.0;
      for (int gg = 0; gg < G; gg++)
#pragma unroll
        for (int k = 0; k < VEC; k++) tot.v[k] += shg[(gg * LPR + lane) * VEC + k];
      for (int cc = 0; cc < lr.ST; cc++) {
        const double b = lr.Bcat[(long long)cc * sp.n + j];
        const vecd<VEC> w = ldrow<VEC>(WS + (long long)cc * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) tot.v[k] += w.v[k] * b;
      }
#pragma unroll
      for (int k = 0; k < VEC; k++) tot.v[k] *= scale;
      if (Xdot) {
        const vecd<VEC> xd = ldrow<VEC>(Xdot + j * r + ch);
#pragma unroll
        for (int k = 0; k < VEC; k++) nrm += tot.v[k] * xd.v[k];
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) nrm += tot.v[k] * tot.v[k];
      }
      if constexpr (HMU > 0) {
        const vecd<VEC> gold = ldrow<VEC>(Y + j * r + ch);
        upd_row<VEC, HMU>(*u, tot, gold, j * r + ch);
      }
      strow<VEC>(Y + j * r + ch, tot);
    }
  }
  }
  if (slot >= 0) {
    nrm = block_sum1(nrm, sh);
    if (threadIdx.x == 0) slot_partials(partials, slot)[pbase + bid] = nrm;
  }
}
// Both in one grid: the first nb_long blocks take the hub rows (the longest work items start first), the rest
// the short rows — the two row sets are disjoint, and as two launches the hub rows (a few hundred rows, tens of
// µs of dependent gather rounds for the longest) ran after the short ones instead of beside them.
template <int LPR, int VEC>
__global__ void __launch_bounds__(SDPLR_NT)
k_spmm_both(DevSparse sp, const double* __restrict__ X, double* __restrict__ Y, int r, double scale,
            DevLowRank lr, const double* __restrict__ WS, int slot, double* __restrict__ partials,
            const DevCtrl* __restrict__ c, int check_done, const double* __restrict__ Xdot, int nb_long) {
  const int nb_short = gridDim.x - nb_long;
  if ((int)blockIdx.x < nb_long)
    spmm_long_rows<LPR, VEC>(sp, X, Y, r, scale, lr, WS, slot, nb_short, partials, c, check_done, Xdot, blockIdx.x, nb_long);
  else
    spmm_rows<LPR, VEC>(sp, X, Y, r, scale, lr, WS, slot, partials, c, check_done, Xdot, blockIdx.x - nb_long, nb_short);
}

// G = 2·R·S of the in-loop g! with lbfgs_update! riding along (see upd_row): hub blocks first, then short rows;
// every block contributes one entry to each Gram partial slot (hub block b → b, short block b → nb_long + b)
// EDGE: S is not read from its assembled array — entry p is s_one[p]·y[s_gid[p]] (see DevSparse, "edge path")
template <int LPR, int VEC, int HMU, bool EDGE>
__device__ __forceinline__ void spmm_both_upd_body(DevSparse sp, const double* __restrict__ X, double* Y, int r, double scale, DevLowRank lr, const double* __restrict__ WS, int slot, double* __restrict__ partials, DevCtrl* __restrict__ c, int nb_long, FactorArena A, int h, const double* __restrict__ D, const double* __restrict__ yv) {
  extern __shared__ double upd_accl[];
  if (c->done) return;
  const UpdCtx u = upd_ctx<HMU>(c, A, h, D, upd_accl);
  const int nb_short = gridDim.x - nb_long;
  if ((int)blockIdx.x < nb_long)
    spmm_long_rows<LPR, VEC, HMU, EDGE>(sp, X, Y, r, scale, lr, WS, slot, nb_short, partials, c, 0, nullptr, blockIdx.x, nb_long, &u, yv);
  else
    spmm_rows<LPR, VEC, HMU, EDGE>(sp, X, Y, r, scale, lr, WS, slot, partials, c, 0, nullptr, blockIdx.x - nb_long, nb_short, &u, yv);
  upd_flush<HMU>(u, c, partials, blockIdx.x, blockIdx.x == 0);
}
template <int LPR, int VEC, int HMU, bool EDGE = false>
__global__ void __launch_bounds__(SDPLR_NT)
k_spmm_both_upd(DevSparse sp, const double* __restrict__ X, double* Y, int r, double scale, DevLowRank lr,
                const double* __restrict__ WS, int slot, double* __restrict__ partials, DevCtrl* __restrict__ c,
                int nb_long, FactorArena A, int h, const double* __restrict__ D, const double* __restrict__ yv = nullptr) {
  spmm_both_upd_body<LPR, VEC, HMU, EDGE>(sp, X, Y, r, scale, lr, WS, slot, partials, c, nb_long, A, h, D, yv);
}

// hub rows of the SpMV: one block per row, 256 lanes stride the row
__global__ void __launch_bounds__(SDPLR_NT)
k_spmv_long(DevSparse sp, const double* __restrict__ x, double* __restrict__ y, DevLowRank lr,
            const double* __restrict__ coef, int slot, int pbase, double* __restrict__ partials,
            const int* __restrict__ stop_flag) {
  __shared__ double sh[8];
  if (stop_flag && *stop_flag) return;
  double dot = 0.0;
  for (int lrow = blockIdx.x; lrow < sp.n_long_rows; lrow += gridDim.x) {
    const long long j = sp.long_rows[lrow];
    double t = 0.0;
    for (int p = sp.colptr[j] + threadIdx.x; p < sp.colptr[j + 1]; p += SDPLR_NT) t += sp.nzval[p] * x[sp.rowval[p]];
    __syncthreads();
    t = block_sum1(t, sh);
    if (threadIdx.x == 0) {
      for (int cc = 0; cc < lr.ST; cc++) t += coef[cc] * lr.Bcat[(long long)cc * sp.n + j];
      y[j] = t;
      dot += x[j] * t;
    }
  }
  if (slot >= 0 && threadIdx.x == 0) slot_partials(partials, slot)[pbase + blockIdx.x] = dot;
}

// ================================================================================================
// Lanczos recurrence (src/coreop.jl:481-500) on UNNORMALISED vectors, two kernels per step.
//   u_1 = v0, γ_1 = ‖v0‖;  u_{i+1} = r_i = S·v_i − α_i v_i − β_{i−1} v_{i−1},  γ_{i+1} = β_i = ‖r_i‖,  v_i = u_i/γ_i.
//   K1 k_lz_spmv: t = S·u_i (+ low-rank), partials of u_i·t;  its block 0 first closes the previous step:
//      γ_i from the ‖u_i‖² partials, beta[i−1] = γ_i, the break test |β| < √n·eps (:494-496), the q-step cap.
//   K2 k_lz_step: α_i = (u_i·t)/γ_i², r_i = t/γ_i − (α_i·u_i/γ_i + β_{i−1}·u_{i−1}/γ_{i−1}) written over t,
//      partials of ‖r_i‖² and of ⟨B_c, r_i⟩ for the next K1's low-rank term.
// Normalising lazily removes one grid-wide dependency per step (three kernels → two) and the kernels carry
// no step index, so three steps (one rotation of the three vector buffers) replay as a hipGraph.
// ================================================================================================
template <int LPR>
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_spmv(DevSparse sp, DevCtrl* __restrict__ c, const double* __restrict__ u, double* __restrict__ t,
          DevLowRank lr, const double* __restrict__ yvec, const double* __restrict__ btx_part, int nb_prev,
          double* __restrict__ coef_out, double* __restrict__ beta_out, double* __restrict__ partials) {
  __shared__ double sh[8];
  // Everything the head needs is requested at once — the stop flag, the first trip's row pointers, the partials
  // of the low-rank coefficients and (block 0) of ‖u‖² — and every wave sums the partials for itself (lanes
  // stride them, fixed-order butterfly): no block barrier and one memory round trip before the SpMV proper
  // instead of flag → partials → barrier → partials → pointers.
  const int dn = c->lz_done;
  constexpr int G = SDPLR_NT / LPR, RIF = 4;
  const int lane = threadIdx.x % LPR, wl = threadIdx.x & 63;
  const long long total = (long long)gridDim.x * G;
  const long long jfirst = (long long)blockIdx.x * G + threadIdx.x / LPR;
  int beg0[RIF], end0[RIF];
#pragma unroll
  for (int k = 0; k < RIF; k++) {
    const long long j = min(jfirst + k * total, (long long)sp.n - 1);
    beg0[k] = sp.colptr[j];
    end0[k] = sp.colptr[j + 1];
  }
  double coef[SDPLR_LRMAX];
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++) {  // low-rank coefficients y[gid]·D_c·⟨B_c, u⟩ (src/structs.jl:117-127)
    coef[cc] = 0.0;
    if (cc < lr.ST) {
      double sidx = 0.0;
      for (int i = wl; i < nb_prev; i += 64) sidx += btx_part[(long long)cc * nb_prev + i];
      sidx = wave_sum(sidx);
      coef[cc] = yvec[lr.col_gid[cc]] * lr.Dcat[cc] * sidx;
      if (blockIdx.x == 0 && threadIdx.x == 0) coef_out[cc] = coef[cc];
    }
  }
  double nn = 0.0;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const double* pn = slot_partials(partials, SLOT_LZ_N);
    for (int i = wl; i < nb_prev; i += 64) nn += pn[i];
    nn = wave_sum(nn);
  }
  if (dn) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // close the previous step (see header)
    const double g = sqrt(nn);
    const long long st = c->lz_steps;
    if (st > 0) {
      beta_out[st - 1] = g;                                                  // beta[i] = ‖Av‖  (:492)
      if (fabs(g) < sqrt((double)sp.n) * 2.220446049250313e-16) c->lz_done = 1;   // (:494-496)
    }
    if (st >= c->lz_qmax) c->lz_done = 1;
    c->lz_gamma_prev = c->lz_gamma_cur;
    c->lz_gamma_cur = g;
    c->lz_beta_prev = (st > 0) ? g : 0.0;
  }
  // Four rows per sub-wave group in flight: the row pointers, then the first (index, value) pair of every
  // row, then the four gathers are issued back to back — the kernel is bound by the chain
  // colptr → rowval → x[rowval] of each row, not by bandwidth.
  double dot = 0.0;
  for (long long j0 = jfirst; j0 < sp.n; j0 += RIF * total) {
    int beg[RIF], end[RIF];
#pragma unroll
    for (int k = 0; k < RIF; k++) {
      const long long j = j0 + k * total;
      if (j < sp.n) {
        beg[k] = (j0 == jfirst) ? beg0[k] : sp.colptr[j];
        end[k] = (j0 == jfirst) ? end0[k] : sp.colptr[j + 1];
        if (sp.n_long_rows > 0 && end[k] - beg[k] > sp.long_thresh) end[k] = beg[k];  // hub row: k_spmv_long
      } else {
        beg[k] = end[k] = 0;
      }
    }
    double tj[RIF];
#pragma unroll
    for (int k = 0; k < RIF; k++) tj[k] = 0.0;
    int maxlen = 0;
#pragma unroll
    for (int k = 0; k < RIF; k++) maxlen = max(maxlen, end[k] - beg[k]);
    maxlen = max(maxlen, __shfl_xor(maxlen, 32, 64));  // wave-uniform trip count is not required; group-uniform is
    for (int off = lane; off < ((maxlen + LPR - 1) / LPR) * LPR; off += LPR) {
      int rv[RIF];
      double nv[RIF];
#pragma unroll
      for (int k = 0; k < RIF; k++) {
        const int p = beg[k] + off;
        const bool ok = p < end[k];
        rv[k] = ok ? sp.rowval[p] : 0;
        nv[k] = ok ? sp.nzval[p] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < RIF; k++) tj[k] += nv[k] * u[rv[k]];
    }
#pragma unroll
    for (int k = 0; k < RIF; k++) {
      const long long j = j0 + k * total;
      const double v = group_sum<LPR>(tj[k]);
      if (lane == 0 && j < sp.n && !(sp.n_long_rows > 0 && sp.colptr[j + 1] - sp.colptr[j] > sp.long_thresh)) {
        double tv = v;
#pragma unroll
        for (int cc = 0; cc < SDPLR_LRMAX; cc++)
          if (cc < lr.ST) tv += coef[cc] * lr.Bcat[(long long)cc * sp.n + j];
        t[j] = tv;
        dot += u[j] * tv;
      }
    }
  }
  dot = block_sum1(dot, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_LZ_A)[blockIdx.x] = dot;
}

__global__ void __launch_bounds__(SDPLR_NT)
k_lz_step(int n, DevCtrl* __restrict__ c, const double* __restrict__ uprev, const double* __restrict__ u,
          double* __restrict__ t, DevLowRank lr, double* __restrict__ btx_part, int nb_a,
          double* __restrict__ alpha_out, double* __restrict__ partials) {
  __shared__ double sh[8];
  if (c->lz_done) return;
  const double gi = c->lz_gamma_cur, gp = c->lz_gamma_prev, be = c->lz_beta_prev;
  const double al = reduce_partials(slot_partials(partials, SLOT_LZ_A), nb_a, sh) / (gi * gi);  // v'·Av (:484)
  double nrm = 0.0;
  double bt[SDPLR_LRMAX];
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++) bt[cc] = 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int k = blockIdx.x * SDPLR_NT + threadIdx.x; k < n; k += stride) {
    const double vi = u[k] / gi, avk = t[k] / gi, vp = uprev[k] / gp;
    const double r = avk - (al * vi + be * vp);                 // (:486-490)
    t[k] = r;
    nrm += r * r;
#pragma unroll
    for (int cc = 0; cc < SDPLR_LRMAX; cc++)
      if (cc < lr.ST) bt[cc] += lr.Bcat[(long long)cc * n + k] * r;
  }
  __syncthreads();
  nrm = block_sum1(nrm, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_LZ_N)[blockIdx.x] = nrm;
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++)
    if (cc < lr.ST) {
      __syncthreads();
      const double v = block_sum1(bt[cc], sh);
      if (threadIdx.x == 0) btx_part[(long long)cc * gridDim.x + blockIdx.x] = v;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long st = c->lz_steps;
    alpha_out[st] = al;
    c->lz_steps = st + 1;                                       // iter += 1 (:482)
  }
}

// ================================================================================================
// Edge path (Lovász-θ and alike): the sparse matrices have DISJOINT supports and all but at most one of them a
// single entry of the upper-triangular pattern (one constraint per edge; the exception is the identity).  Then
//   𝒜(·)_k = two_q·UVt[q] for the one position q of matrix k — written by the SDDMM lane that holds the dot, no
//            segmented reduction (src/coreop.jl:80-90) and no UVt round trip; the big matrix is a plain sum;
//   S_p   = s_one[p]·y[s_gid[p]]  — gathered by the SpMM itself, no assembly pass (src/coreop.jl:205-227).
// One inner iteration is 7 launches instead of 12: seam, direction, this kernel, projection sums, scalar stage
// (k_ls_solve_fast: the big matrix and the low-rank matrices are its "extra slots"), step (R += αD ‖ commit),
// SpMM + lbfgs_update!.
// ================================================================================================
// Both line-search passes (src/linesearch.jl:10-16) in one sweep over the pattern, as k_sddmm<…,2>, plus:
// A_RD/A_DD of the single-entry matrices, the big matrix's sums (block partials → SLOT_PD: half of it, SLOT_DW) and
// — LRN = 1, every row has a diagonal position — the projections RᵀB, DᵀB of the one low-rank column on the
// diagonal positions (src/coreop.jl:125-126).
// The kernel is bound by the NUMBER of vector-memory instructions a CU can issue (one wave instruction per ≈ 16
// cycles whatever its width — the Lanczos band kernel taught that), not by bytes: with row, column, owner, weight,
// λ, primal_vio_raw and B fetched by seven separate (group-uniform) loads per position it took 42 µs against 21 µs
// for the bare k_sddmm<…,2> with its two.  So the per-position scalars come as ONE load — lanes 0..3 of the group
// fetch the four words of a 32-byte record and hand them round by DPP broadcast — and the line-search sums of the
// single-entry matrices moved to k_edge_sums, which reads λ and primal_vio_raw coalesced.
#ifndef SDPLR_EDGE_POS
#define SDPLR_EDGE_POS 2   /* pattern positions per sub-wave group and trip (1 and 2 equal, 4: 196 VGPRs, slower) */
#endif
template <int LPR, int VEC, int LRN>
__device__ __forceinline__ void sddmm_edge_body(DevSparse sp, int m, const double* __restrict__ U, const double* __restrict__ V, int r, double* __restrict__ A_RD, double* __restrict__ A_DD, DevLowRank lr, double* __restrict__ lr_part, double* __restrict__ partials, const DevCtrl* __restrict__ c) {
  __shared__ double sh[2 * (SDPLR_NT / 64)];
  __shared__ double lrs[LRN > 0 ? 2 * SDPLR_NT * VEC : 1];
  if (c->done) return;
  constexpr int G = SDPLR_NT / LPR;
  constexpr int POS = SDPLR_EDGE_POS;
  const int lane = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const long long total = (long long)gridDim.x * G;
  double acc[2] = {0.0, 0.0};   // ½·Σ big·UVt0, Σ big·UVt1
  double l0[VEC], l1[VEC];
#pragma unroll
  for (int k = 0; k < VEC; k++) l0[k] = l1[k] = 0.0;
  for (long long q0 = (long long)blockIdx.x * G + grp; q0 < sp.nnzT; q0 += POS * total) {
    long long row[POS], col[POS];
    int gid[POS];
    double w[POS], bb[POS];
#pragma unroll
    for (int t = 0; t < POS; t++) {
      const long long qq = min(q0 + t * total, (long long)sp.nnzT - 1);
      const bool in = q0 + t * total < sp.nnzT;
      if constexpr (LPR >= 4) {
        const double word = sp.q_rec[qq * 4 + (lane & 3)];
        const double w0 = group_bcast<LPR>(word, 0), w1 = group_bcast<LPR>(word, 1);
        row[t] = (unsigned)__double2loint(w0);
        col[t] = (unsigned)__double2hiint(w0);
        gid[t] = in ? __double2loint(w1) : -1;
        w[t] = group_bcast<LPR>(word, 2);
      } else {
        row[t] = sp.triu_rowval[qq];
        col[t] = sp.triu_colidx[qq];
        gid[t] = in ? sp.q_gid[qq] : -1;
        w[t] = sp.q_two[qq];
      }
    }
#pragma unroll
    for (int t = 0; t < POS; t++) bb[t] = LRN > 0 ? lr.Bcat[row[t]] : 0.0;   // (with the row chunks: inside the chunk loop
                                                                             //  it would be one more dependent round trip)
    double a0[POS], a1[POS];
#pragma unroll
    for (int t = 0; t < POS; t++) a0[t] = a1[t] = 0.0;
    for (int ch = lane * VEC; ch < r; ch += LPR * VEC) {
      vecd<VEC> uc[POS], ur[POS], vc[POS], vr[POS];
#pragma unroll
      for (int t = 0; t < POS; t++) {
        uc[t] = ldrow<VEC>(U + col[t] * r + ch);
        ur[t] = ldrow<VEC>(U + row[t] * r + ch);
        vc[t] = ldrow<VEC>(V + col[t] * r + ch);
        vr[t] = ldrow<VEC>(V + row[t] * r + ch);
      }
#pragma unroll
      for (int t = 0; t < POS; t++) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          a0[t] += uc[t].v[k] * vr[t].v[k];
          a0[t] += vc[t].v[k] * ur[t].v[k];
          a1[t] += vc[t].v[k] * vr[t].v[k];
        }
        if (LRN > 0 && row[t] == col[t] && gid[t] != -2 && q0 + t * total < sp.nnzT) {   // (single chunk when LRN > 0: r ≤ LPR·VEC)
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            l0[k] += uc[t].v[k] * bb[t];
            l1[k] += vc[t].v[k] * bb[t];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < POS; t++) {
      a0[t] = group_sum<LPR>(a0[t]);
      a1[t] = group_sum<LPR>(a1[t]);
    }
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < POS; t++) {
        if (gid[t] < 0) continue;
        const double q1 = w[t] * a0[t], q2 = w[t] * a1[t];
        if (gid[t] == sp.big_gid) {
          acc[0] += 0.5 * q1;
          acc[1] += q2;
        } else {
          A_RD[gid[t]] = q1;
          A_DD[gid[t]] = q2;
        }
      }
    }
  }
  if constexpr (LRN > 0) {   // block partials of the projections, in k_lr_project's layout (block index fastest)
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      lrs[(0 * SDPLR_NT + threadIdx.x) * VEC + k] = l0[k];
      lrs[(1 * SDPLR_NT + threadIdx.x) * VEC + k] = l1[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * LPR * VEC; t += SDPLR_NT) {
      const int f = t / (LPR * VEC), ch = t % (LPR * VEC);   // channel ch = lane·VEC + k
      if (ch < r) {
        double sum = 0.0;
        for (int g = 0; g < G; g++) sum += lrs[(f * SDPLR_NT + g * LPR + ch / VEC) * VEC + ch % VEC];
        lr_part[((long long)f * lr.ST * r + ch) * gridDim.x + blockIdx.x] = sum;
      }
    }
    __syncthreads();
  }
  block_sum<2>(acc, sh);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_PD)[blockIdx.x] = acc[0];
    slot_partials(partials, SLOT_DW)[blockIdx.x] = acc[1];
  }
}
template <int LPR, int VEC, int LRN>
__global__ void __launch_bounds__(SDPLR_NT)
k_sddmm_edge(DevSparse sp, int m, const double* __restrict__ U, const double* __restrict__ V, int r,
             double* __restrict__ A_RD, double* __restrict__ A_DD, DevLowRank lr, double* __restrict__ lr_part,
             double* __restrict__ partials, const DevCtrl* __restrict__ c) {
  sddmm_edge_body<LPR, VEC, LRN>(sp, m, U, V, r, A_RD, A_DD, lr, lr_part, partials, c);
}

// Everything between k_sddmm_edge and the scalar stage that is a sum, in one grid:
//   blocks [0, nout):            projection sums W[t] = Σ_blocks lr_part[t][·]                 (as k_lr_reduce)
//   blocks [nout, nout + 2):     the big matrix's two sums → red2[0..1]
//   the rest (nb_c blocks):      the eight line-search sums (src/linesearch.jl:36-56) over the constraints the
//                                scalar stage does not take itself — λ, primal_vio_raw, A_RD, A_DD read coalesced —
//                                as block partials in SLOT_LS + 0..7
__device__ __forceinline__ void edge_sums_body(int nout, int nb, const double* __restrict__ lr_part, double* __restrict__ W, double* __restrict__ red2, int nb_c, int m, int n_extra, const int* __restrict__ extra, const double* __restrict__ lam, const double* __restrict__ pv_raw, const double* __restrict__ A_RD, const double* __restrict__ A_DD, double* __restrict__ partials, const DevCtrl* __restrict__ c) {
  __shared__ double sh[8 * (SDPLR_NT / 64)];
  const int dn = c->done;
  const int t = blockIdx.x;
  if (t < nout + 2) {
    const double* p = t < nout ? lr_part + (long long)t * nb : slot_partials(partials, t == nout ? SLOT_PD : SLOT_DW);
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = p[min((int)threadIdx.x + SDPLR_NT * q, nb - 1)];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) s += ((int)threadIdx.x + SDPLR_NT * q < nb) ? v[q] : 0.0;
    for (int b = threadIdx.x + 16 * SDPLR_NT; b < nb; b += SDPLR_NT) s += p[b];
    s = block_sum1(s, sh);
    if (dn) return;
    if (threadIdx.x == 0) (t < nout ? W[t] : red2[t - nout]) = s;
    return;
  }
  if (dn) return;
  const double sigma = c->sigma;
  const int bc = t - nout - 2;
  double s[8];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = 0.0;
  for (int i = bc * SDPLR_NT + threadIdx.x; i < m; i += nb_c * SDPLR_NT) {
    bool skip = false;
    for (int e = 0; e < n_extra; e++) skip |= (extra[e] == i);
    const double l = lam[i], nq0 = pv_raw[i], q1 = A_RD[i], q2 = A_DD[i];
    if (skip) continue;
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
    for (int k = 0; k < 8; k++) slot_partials(partials, SLOT_LS + k)[bc] = s[k];
  }
}
__global__ void __launch_bounds__(SDPLR_NT)
k_edge_sums(int nout, int nb, const double* __restrict__ lr_part, double* __restrict__ W, double* __restrict__ red2,
            int nb_c, int m, int n_extra, const int* __restrict__ extra, const double* __restrict__ lam,
            const double* __restrict__ pv_raw, const double* __restrict__ A_RD, const double* __restrict__ A_DD,
            double* __restrict__ partials, const DevCtrl* __restrict__ c) {
  edge_sums_body(nout, nb, lr_part, W, red2, nb_c, m, n_extra, extra, lam, pv_raw, A_RD, A_DD, partials, c);
}

// The two independent jobs between the scalar stage and the SpMM of g!, in one grid:
//   blocks [0, nb_ax):   R += α·dirt                                                          (src/sdplr.jl:219)
//   blocks [nb_ax, nb_ax + nb_c): commit of every constraint slot the scalar stage has not committed (the single-entry
//                        matrices and uncovered slots): primal_vio_raw, primal_vio, y (src/linesearch.jl:118-124,
//                        src/coreop.jl:229-236), the ‖pv‖² partials (src/sdplr.jl:229-234) — and S assembly BY
//                        OWNER: the thread that forms y_k stores y_k·nzval_one into the ≤ 2 entries of S that
//                        matrix k owns (src/coreop.jl:205-227 without the scatter-add or the map)
//   the rest:            the entries of S the multi-entry matrix owns (its y comes from the scalar stage)
__device__ __forceinline__ void edge_step_body(DevSparse sp, DevCtrl* __restrict__ c, double* __restrict__ R, const double* __restrict__ D, long long N, int nb_ax, int nb_c, int m, int n_extra, const int* __restrict__ extra, double* __restrict__ pv_raw, const double* __restrict__ A_RD, const double* __restrict__ A_DD, const double* __restrict__ lb, double* __restrict__ pv, double* __restrict__ y, const double* __restrict__ lam, const double* __restrict__ lam_ub, double* __restrict__ partials, int nb_gnorm) {
  __shared__ double sh[8];
  if (c->done) return;
  const double a = c->alpha, sigma = c->sigma;
  const int b = blockIdx.x;
  if (b < nb_ax) {
    const long long N2 = N >> 1;
    const long long stride = (long long)nb_ax * SDPLR_NT;
    for (long long i = (long long)b * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
      double2 x = reinterpret_cast<double2*>(R)[i];
      const double2 d = reinterpret_cast<const double2*>(D)[i];
      x.x += a * d.x;
      x.y += a * d.y;
      reinterpret_cast<double2*>(R)[i] = x;
    }
    if ((N & 1) && b == 0 && threadIdx.x == 0) R[N - 1] += a * D[N - 1];
    return;
  }
  if (b >= nb_ax + nb_c) {
    const int nb_b = gridDim.x - nb_ax - nb_c;
    const double yb = sp.big_gid >= 0 ? y[sp.big_gid] : 0.0;
    for (int e = (b - nb_ax - nb_c) * SDPLR_NT + threadIdx.x; e < sp.n_big_pos; e += nb_b * SDPLR_NT)
      sp.nzval[sp.big_pos[e]] = sp.big_one[e] * yb;
    return;
  }
  const int bc = b - nb_ax;
  double t = 0.0;
  for (int i = bc * SDPLR_NT + threadIdx.x; i <= m; i += nb_c * SDPLR_NT) {
    bool skip = false;
    for (int e = 0; e < n_extra; e++) skip |= (extra[e] == i);
    if (skip) continue;
    const int p0 = sp.own_pos[2 * i], p1 = sp.own_pos[2 * i + 1];
    const double one = sp.own_one[i];
    const double v = pv_raw[i] + a * (a * A_DD[i] + A_RD[i]);
    pv_raw[i] = v;
    double yi = 1.0;
    if (i == m) {
      c->obj = v;
    } else {
      const double pc = fmax(v, lb[i]);
      pv[i] = pc;
      t += pc * pc;
      yi = -fmin(lam_ub[i], lam[i] - sigma * v);
    }
    y[i] = yi;
    if (p0 >= 0) sp.nzval[p0] = one * yi;
    if (p1 >= 0) sp.nzval[p1] = one * yi;
  }
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) {
    slot_partials(partials, SLOT_PVNORM2)[bc] = t;
    if (bc == 0) {   // the SpMM that follows produces the ‖G‖² partials; the seam kernel folds both
      c->norms_pending = 1;
      c->nb_pvnorm = nb_c;
      c->nb_gnorm = nb_gnorm;
    }
  }
}
__global__ void __launch_bounds__(SDPLR_NT)
k_edge_step(DevSparse sp, DevCtrl* __restrict__ c, double* __restrict__ R, const double* __restrict__ D, long long N, int nb_ax, int nb_c,
            int m, int n_extra, const int* __restrict__ extra, double* __restrict__ pv_raw,
            const double* __restrict__ A_RD, const double* __restrict__ A_DD, const double* __restrict__ lb,
            double* __restrict__ pv, double* __restrict__ y, const double* __restrict__ lam,
            const double* __restrict__ lam_ub, double* __restrict__ partials, int nb_gnorm) {
  edge_step_body(sp, c, R, D, N, nb_ax, nb_c, m, n_extra, extra, pv_raw, A_RD, A_DD, lb, pv, y, lam, lam_ub, partials, nb_gnorm);
}

// ================================================================================================
// Lanczos SpMV through LDS-resident bands of x  (k_lz_band + k_lz_step_band; replaces k_lz_spmv + k_lz_step
// when the band plan exists: 2¹⁴ ≤ n, ≤ 16 bands).
//
// k_lz_spmv gathers x[rowval[p]] — 2.1 M random 8-byte reads per step at the north-star size.  x (0.8 MB) sits in
// every XCD's L2, but each 8-byte read costs a whole L2 request and line fill (profiles/r02_lanczos_pmc.csv:
// 2.5 M TCC read requests per launch, 88 % hits, the waves 54 % of their cycles in s_waitcnt) — the kernel runs at
// the L2 REQUEST rate, 18 µs for 31 MB of compulsory bytes.  Here the columns are cut into NB bands of BW ≤ 16 384
// entries and the rows into NC chunks; block (b, c) stages band b of x into LDS with coalesced loads (NB·NC·BW·8 B
// ≈ 29 MB of L2 reads per step instead of 2.1 M requests of 128 B), then sweeps the nonzeros of its rows that fall
// into the band — the random reads hit LDS — and writes a PARTIAL t for its chunk; the recurrence kernel adds the NB
// partials of a row in band order while it streams u and u_prev anyway.
//   Layout ("sliced ELL", built once on the host): within a block the rows are sorted by their nonzero count in the
// band and cut into slots of 64 (one wave, one row per lane); a slot stores its entries lane-fastest and padded to
// the slot's longest row, so a wave reads 64 consecutive (column u16, value f64) pairs per step of a uniform loop.
// Values are re-gathered from sparse_S.nzval once per Lanczos run (k_lz_band_fill): S is fixed during the q steps.
// Sums are formed in a fixed order (row: ascending column inside a band; then bands ascending): deterministic.
// Hub rows (more than long_thresh nonzeros) are left out of the plan: k_spmv_long writes them to `textra`.
// ================================================================================================
struct DevBand {
  int NB, NC, BW, CH, n_slots, n_groups;   // n_groups counts the real groups; group n_groups is an all-padding dummy
  const int* blk_slot;            // [NB·NC + 1] slots of block b·NC + c
  const int* slot_g;              // [n_slots + 1] first group of each slot
  // one GROUP = 4 entries of each of a slot's 64 rows (lane-fastest); padding entries are (column 0, value 0)
  const uint4* cw;                // [groups·64] {c0 | c1 << 16, c2 | c3 << 16, row local to the chunk (0xFFFF ⇒ none), 0}
  double2* vA;                    // [groups·64] values of entries 0, 1   (refreshed by k_lz_band_fill)
  double2* vB;                    // [groups·64] values of entries 2, 3
  const int4* pos;                // [groups·64] positions in sparse_S.nzval; −1 ⇒ padding
  double* tpart;                  // [NB][n] band partials of t = S·u
  double* textra;                 // [n] hub rows of t (k_spmv_long), zero elsewhere
  // PALETTE form (structured problems: every off-diagonal entry of S is y_g·A_g[i,j], and A_g has ≤ 255 distinct
  // off-diagonal values — unit or small-integer edge weights): an entry is a 2-byte column and a 1-byte value code,
  // packed into the SAME 16-byte word per lane and four entries ({c0|c1, c2|c3, row|code0|code1, code2|code3}); no value
  // arrays, no refill per run (the palette is static, y_g a scalar).  The diagonal — different in every row — is left
  // out of the groups and rides band 0's partial (`sdiag`, refreshed once per Lanczos run).
  int pal_mode, gid_g;
  const double* pal;              // [256] pal[0] = 0 (padding), pal[c] the c-th distinct off-diagonal value of A_g
  double* sdiag;                  // [n] S[k,k] (0 for hub rows: k_spmv_long has them whole)
  const int* diagpos;             // [n] position of (k,k) in sparse_S.nzval, −1 ⇒ none / hub row
};

// sdiag[k] = S[k,k] of the S left by the last 𝒜t_preprocess! (palette form, once per Lanczos run)
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_diag_fill(DevBand bd, int n, const double* __restrict__ nzval) {
  const int stride = gridDim.x * SDPLR_NT;
  for (int k = blockIdx.x * SDPLR_NT + threadIdx.x; k < n; k += stride) {
    const int p = bd.diagpos[k];
    bd.sdiag[k] = p >= 0 ? nzval[p] : 0.0;
  }
}

__global__ void __launch_bounds__(SDPLR_NT)
k_lz_band_fill(DevBand bd, const double* __restrict__ nzval) {
  const long long total = (long long)(bd.n_groups + 1) * 64;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long e = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; e < total; e += stride) {
    const int4 p = bd.pos[e];
    double2 a, b;
    a.x = p.x >= 0 ? nzval[p.x] : 0.0;
    a.y = p.y >= 0 ? nzval[p.y] : 0.0;
    b.x = p.z >= 0 ? nzval[p.z] : 0.0;
    b.y = p.w >= 0 ? nzval[p.w] : 0.0;
    bd.vA[e] = a;
    bd.vB[e] = b;
  }
}

#define SDPLR_LZB_NT 1024
#define SDPLR_LZB_MAXS 4      /* slots per wave: CH ≤ 4096 rows = 64 slots over 16 waves */
#define SDPLR_LZB_G0 3        /* groups of a wave's FIRST slot requested up front (the block's 16 longest slots) */
// K1: partial t for (band, chunk) + the block's share of u·t; the grid's last block closes the previous step exactly
// as k_lz_spmv's block 0 does and publishes the low-rank coefficients (coef_out[c] = y[gid]·D_c·⟨B_c,u⟩,
// coef_out[ST + c] = ⟨B_c,u⟩).
// Everything a sweeping block reads from global memory is requested in ONE round trip before anything is waited for,
// in as few wave instructions as the layout allows (a CU's texture path issues one wave-load per ≈ 8-16 cycles
// whatever its width: the first version — staging loop, then slot after slot, one 2-byte and one 8-byte load per
// entry — sat 8.5 of its 12 µs in the request phase): the band of x (≤ 8 × 16 B per thread), per slot three 16-byte
// loads per lane for four entries (packed columns + row, two value pairs), the chunk's rows of u for the dot.  Slots
// past the block's end and groups past a slot's end are skipped by a WAVE-UNIFORM branch (a slot belongs to one wave)
// and stand in the registers as all-padding groups, so nothing in the sweep is conditional.
template <bool PAL>
__global__ void __launch_bounds__(SDPLR_LZB_NT)
k_lz_band(DevBand bd, int n, DevCtrl* __restrict__ c, const double* __restrict__ u, DevLowRank lr,
          const double* __restrict__ yvec, const double* __restrict__ btx_part, int nb_prev,
          double* __restrict__ coef_out, double* __restrict__ beta_out, double* __restrict__ partials,
          DevSparse sp, int nb_hub) {
  extern __shared__ double lzb_lds[];   // xs[BW] | ys[CH]
  __shared__ double shw[SDPLR_LZB_NT / 64];
  __shared__ double pals[PAL ? 256 : 1];   // PAL: y_g·pal[code] = the S values themselves (fl(y_g·a), as assembled)
  constexpr int MAXS = SDPLR_LZB_MAXS, G0 = SDPLR_LZB_G0, NW = SDPLR_LZB_NT / 64;
  constexpr int NG = G0 + MAXS - 1;     // groups requested up front per lane
  double* xs = lzb_lds;
  double* ys = lzb_lds + bd.BW;
  const int dn = c->lz_done;
  if (blockIdx.x == gridDim.x - 1) {
    // The grid's LAST block only closes the previous step (one wave; its partials — nb_prev ≤ 1024 each — are fetched
    // 16 per lane in one round trip, unconditionally: the arrays are SDPLR_MAXNB wide).  A block of its own, so that
    // neither its registers nor its round trip sit on a sweeping block's path.
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double nn = 0.0;
    {
      const double* pn = slot_partials(partials, SLOT_LZ_N);
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = pn[lane + 64 * q];
#pragma unroll
      for (int q = 0; q < 16; q++) nn += (lane + 64 * q < nb_prev) ? v[q] : 0.0;
      for (int i = lane + 1024; i < nb_prev; i += 64) nn += pn[i];
      nn = wave_sum(nn);
    }
    double bsum[SDPLR_LRMAX];
#pragma unroll
    for (int cc = 0; cc < SDPLR_LRMAX; cc++) {
      bsum[cc] = 0.0;
      if (cc < lr.ST) {
        const double* pb = btx_part + (long long)cc * nb_prev;
        double v[16], sidx = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = pb[min(lane + 64 * q, nb_prev - 1)];
#pragma unroll
        for (int q = 0; q < 16; q++) sidx += (lane + 64 * q < nb_prev) ? v[q] : 0.0;
        for (int i = lane + 1024; i < nb_prev; i += 64) sidx += pb[i];
        bsum[cc] = wave_sum(sidx);
      }
    }
    if (!dn && lane == 0) {   // close the previous step (see "Lanczos recurrence" above)
#pragma unroll
      for (int cc = 0; cc < SDPLR_LRMAX; cc++)
        if (cc < lr.ST) {
          coef_out[cc] = yvec[lr.col_gid[cc]] * lr.Dcat[cc] * bsum[cc];
          coef_out[lr.ST + cc] = bsum[cc];
        }
      const double g = sqrt(nn);
      const long long st = c->lz_steps;
      if (st > 0) {
        beta_out[st - 1] = g;                                                      // beta[i] = ‖Av‖  (:492)
        if (fabs(g) < sqrt((double)n) * 2.220446049250313e-16) c->lz_done = 1;     // (:494-496)
      }
      if (st >= c->lz_qmax) c->lz_done = 1;
      c->lz_gamma_prev = c->lz_gamma_cur;
      c->lz_gamma_cur = g;
      c->lz_beta_prev = (st > 0) ? g : 0.0;
    }
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= bd.NB * bd.NC) {
    // Hub rows (left out of the band plan): the nb_hub blocks between the sweeping blocks and the closing one take them
    // whole from the assembled S, as k_spmv_long does — in this grid, so that a power-law instance does not pay a
    // third launch per step.  t goes to `textra`, the block's share of u·t to the partial slot behind the sweepers'.
    if (dn) return;
    const int hb = (int)blockIdx.x - bd.NB * bd.NC, quarter = tid >> 8, qt = tid & 255;   // four rows per block and trip
    double dot = 0.0;
    for (int l0 = hb * 4; l0 < sp.n_long_rows; l0 += nb_hub * 4) {
      const int lrow = l0 + quarter;
      const bool have = lrow < sp.n_long_rows;
      const long long j = sp.long_rows[have ? lrow : l0];
      double t = 0.0;
      if (have)
        for (int p = sp.colptr[j] + qt; p < sp.colptr[j + 1]; p += 256) t += sp.nzval[p] * u[sp.rowval[p]];
      t = wave_sum(t);
      __syncthreads();
      if (lane == 0) shw[wave] = t;
      __syncthreads();
      if (qt == 0 && have) {
        const double tt = ((shw[4 * quarter] + shw[4 * quarter + 1]) + shw[4 * quarter + 2]) + shw[4 * quarter + 3];
        bd.textra[j] = tt;
        dot += u[j] * tt;
      }
    }
    // the block's share of u·t: its four quarter sums in a fixed order
    __syncthreads();
    if (qt == 0) shw[quarter] = dot;
    __syncthreads();
    if (tid == 0) slot_partials(partials, SLOT_LZ_A)[blockIdx.x] = ((shw[0] + shw[1]) + shw[2]) + shw[3];
    return;
  }
  const int b = blockIdx.x / bd.NC, ch = blockIdx.x % bd.NC;
  const int col0 = b * bd.BW, ncol = max(0, min(bd.BW, n - col0));
  const int row0 = ch * bd.CH, nrow = max(0, min(bd.CH, n - row0));
  const int s0 = bd.blk_slot[blockIdx.x], s1 = bd.blk_slot[blockIdx.x + 1];
  // ---- requests -------------------------------------------------------------------------------------------------
  // group index of up-front request j: j < G0 → groups 0..G0−1 of the wave's first slot, then group 0 of slots 1..
  int gfirst[MAXS], gend[MAXS];
#pragma unroll
  for (int q = 0; q < MAXS; q++) {
    const int sl = s0 + wave + NW * q;
    const int slc = min(sl, max(bd.n_slots - 1, 0));
    const int g0 = bd.slot_g[slc], g1 = bd.slot_g[slc + 1];
    const bool ok = sl < s1;
    gfirst[q] = ok ? g0 : bd.n_groups;      // (bd.n_groups = the dummy group)
    gend[q] = ok ? g1 : bd.n_groups;
  }
  uint4 cw[NG];
  double2 va[PAL ? 1 : NG], vb[PAL ? 1 : NG];
  double palv = 0.0, ygv = 0.0;
  if (PAL) {
    palv = bd.pal[tid & 255];
    ygv = yvec[bd.gid_g];
  }
#pragma unroll
  for (int j = 0; j < NG; j++) {
    const int q = j < G0 ? 0 : j - G0 + 1;
    const int g = gfirst[q] + (j < G0 ? j : 0);
    if (g < gend[q]) {     // wave-uniform (a scalar branch): slots and groups past the end are not requested at all
      const long long e = (long long)g * 64 + lane;
      cw[j] = bd.cw[e];
      if (!PAL) {
        va[j] = bd.vA[e];
        vb[j] = bd.vB[e];
      }
    } else {
      cw[j] = make_uint4(0u, 0u, 0xFFFFu, 0u);
      if (!PAL) va[j].x = va[j].y = vb[j].x = vb[j].y = 0.0;
    }
  }
  constexpr int XT = 16384 / 2 / SDPLR_LZB_NT;    // double2 per thread for the widest band
  const int npair = ncol >> 1;
  double2 xr[XT];
  const double2* u2 = reinterpret_cast<const double2*>(u + col0);   // col0 is a multiple of 64
#pragma unroll
  for (int k = 0; k < XT; k++) xr[k] = u2[min(tid + SDPLR_LZB_NT * k, max(npair - 1, 0))];   // (vectors are padded by 64)
  double ur[MAXS];   // the chunk's rows of u (CH ≤ 4096 = MAXS·1024)
#pragma unroll
  for (int k = 0; k < MAXS; k++) ur[k] = u[min(row0 + tid + SDPLR_LZB_NT * k, n - 1)];
  double sdr[PAL ? MAXS : 1];   // palette form: the rows' diagonal entries ride band 0's partial
  if (PAL) {
#pragma unroll
    for (int k = 0; k < MAXS; k++) sdr[k] = (b == 0) ? bd.sdiag[min(row0 + tid + SDPLR_LZB_NT * k, n - 1)] : 0.0;
  }
  if (dn) return;
  // ---- the band of x into LDS --------------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < XT; k++) {
    const int i = tid + SDPLR_LZB_NT * k;
    if (i < npair) reinterpret_cast<double2*>(xs)[i] = xr[k];
  }
  if ((ncol & 1) && tid == 0) xs[ncol - 1] = u[col0 + ncol - 1];
  for (int i = tid; i < nrow; i += SDPLR_LZB_NT) ys[i] = 0.0;
  if (PAL && tid < 256) pals[tid] = ygv * palv;
  __syncthreads();
  // ---- the sweep: one row per lane and slot; a row's products are added in ascending column order ------------------
  auto fold = [&](double acc, const uint4& w, const double2& a, const double2& bb) {
    acc += a.x * xs[w.x & 0xFFFFu];
    acc += a.y * xs[w.x >> 16];
    acc += bb.x * xs[w.y & 0xFFFFu];
    acc += bb.y * xs[w.y >> 16];
    return acc;
  };
  auto fold_pal = [&](double acc, const uint4& w) {
    acc += pals[(w.z >> 16) & 0xFFu] * xs[w.x & 0xFFFFu];
    acc += pals[w.z >> 24] * xs[w.x >> 16];
    acc += pals[w.w & 0xFFu] * xs[w.y & 0xFFFFu];
    acc += pals[(w.w >> 8) & 0xFFu] * xs[w.y >> 16];
    return acc;
  };
#pragma unroll
  for (int q = 0; q < MAXS; q++) {
    double acc = 0.0;
    const int j0 = q == 0 ? 0 : G0 + q - 1, nj = q == 0 ? G0 : 1;
#pragma unroll
    for (int j = 0; j < nj; j++) acc = PAL ? fold_pal(acc, cw[j0 + j]) : fold(acc, cw[j0 + j], va[PAL ? 0 : j0 + j], vb[PAL ? 0 : j0 + j]);
    for (int g = gfirst[q] + nj; g < gend[q]; g++) {      // rows longer than the up-front groups (rare)
      const long long e = (long long)g * 64 + lane;
      if (PAL) acc = fold_pal(acc, bd.cw[e]);
      else acc = fold(acc, bd.cw[e], bd.vA[e], bd.vB[e]);
    }
    const unsigned lrow = cw[j0].z & 0xFFFFu;
    if (lrow != 0xFFFFu) ys[lrow] = acc;
  }
  __syncthreads();
  double dot = 0.0;
  double* tp = bd.tpart + (long long)b * n + row0;
#pragma unroll
  for (int k = 0; k < MAXS; k++) {
    const int i = tid + SDPLR_LZB_NT * k;
    if (i < nrow) {
      double t = ys[i];
      if (PAL) t += sdr[k] * ur[k];
      tp[i] = t;
      dot += ur[k] * t;
    }
  }
  dot = wave_sum(dot);
  if (lane == 0) shw[wave] = dot;
  __syncthreads();
  if (tid == 0) {
    double tsum = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) tsum += shw[w];
    slot_partials(partials, SLOT_LZ_A)[blockIdx.x] = tsum;
  }
}

// K2: t_k = Σ_b tpart[b][k] (+ hub rows + low-rank), α_i, r_i and the partials for the next step (see k_lz_step).
// The grid covers n in one trip (n ≤ 2¹⁸ whenever the band plan exists), and a thread's row — its ≤ 16 band
// partials, u, u_prev — is requested together with the flags and the partials of u·t, before the reduction's
// barriers: one memory round trip, not two.
#define SDPLR_LZB_NBMAX 16
__global__ void __launch_bounds__(SDPLR_NT)
k_lz_step_band(int n, DevCtrl* __restrict__ c, DevBand bd, int has_long, const double* __restrict__ uprev,
               const double* __restrict__ u, double* __restrict__ t, DevLowRank lr,
               const double* __restrict__ coef, double* __restrict__ btx_part, int nb_a,
               double* __restrict__ alpha_out, double* __restrict__ partials) {
  __shared__ double sh[8];
  const int dn = c->lz_done;
  const double gi = c->lz_gamma_cur, gp = c->lz_gamma_prev, be = c->lz_beta_prev;
  const int stride = gridDim.x * SDPLR_NT;
  const int k0 = blockIdx.x * SDPLR_NT + threadIdx.x;
  const int kc = min(k0, n - 1);           // clamped: every load of the first trip is unconditional
  double tpv[SDPLR_LZB_NBMAX];
#pragma unroll
  for (int b = 0; b < SDPLR_LZB_NBMAX; b++) tpv[b] = bd.tpart[(long long)min(b, bd.NB - 1) * n + kc];
  const double te0v = bd.textra[kc];
  const double te0 = has_long ? te0v : 0.0;
  const double u0 = u[kc], up0 = uprev[kc];
  double cf[SDPLR_LRMAX], lrdot = 0.0;
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++) {
    cf[cc] = 0.0;
    if (cc < lr.ST) {
      cf[cc] = coef[cc];
      lrdot += cf[cc] * coef[lr.ST + cc];     // coef_c·⟨B_c,u⟩: the low-rank part of u·(S·u)
    }
  }
  double pa = 0.0;
  for (int i = threadIdx.x; i < nb_a; i += SDPLR_NT) pa += slot_partials(partials, SLOT_LZ_A)[i];
  if (dn) return;
  const double al = (block_sum1(pa, sh) + lrdot) / (gi * gi);  // v'·Av (:484)
  double nrm = 0.0;
  double bt[SDPLR_LRMAX];
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++) bt[cc] = 0.0;
  for (int k = k0; k < n; k += stride) {
    double tv = 0.0, uk, upk;
    if (k == k0) {
#pragma unroll
      for (int b = 0; b < SDPLR_LZB_NBMAX; b++) tv += (b < bd.NB) ? tpv[b] : 0.0;
      tv += te0;
      uk = u0;
      upk = up0;
    } else {
      for (int b = 0; b < bd.NB; b++) tv += bd.tpart[(long long)b * n + k];
      if (has_long) tv += bd.textra[k];
      uk = u[k];
      upk = uprev[k];
    }
#pragma unroll
    for (int cc = 0; cc < SDPLR_LRMAX; cc++)
      if (cc < lr.ST) tv += cf[cc] * lr.Bcat[(long long)cc * n + k];
    const double vi = uk / gi, avk = tv / gi, vp = upk / gp;
    const double r = avk - (al * vi + be * vp);                 // (:486-490)
    t[k] = r;
    nrm += r * r;
#pragma unroll
    for (int cc = 0; cc < SDPLR_LRMAX; cc++)
      if (cc < lr.ST) bt[cc] += lr.Bcat[(long long)cc * n + k] * r;
  }
  __syncthreads();
  nrm = block_sum1(nrm, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_LZ_N)[blockIdx.x] = nrm;
#pragma unroll
  for (int cc = 0; cc < SDPLR_LRMAX; cc++)
    if (cc < lr.ST) {
      __syncthreads();
      const double v = block_sum1(bt[cc], sh);
      if (threadIdx.x == 0) btx_part[(long long)cc * gridDim.x + blockIdx.x] = v;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long st = c->lz_steps;
    alpha_out[st] = al;
    c->lz_steps = st + 1;                                       // iter += 1 (:482)
  }
}
