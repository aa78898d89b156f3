// k_dense.h — dense n×r sweeps: the L-BFGS direction and update, the step R += αD, and the small
// in-loop BLAS-1 of _sdplr.  All are HBM-bound streaming kernels over flat arrays of N = n·r doubles:
// 16-B (double2) loads/stores per lane, grid-stride, ≤ SDPLR_MAXNB blocks of 256 threads.
//
// L-BFGS is evaluated in "Gram" form.  The two-loop recursion of src/lbfgs.jl:77-124 only ever
// takes inner products between the history vectors and the running direction, and the running
// direction is always a linear combination of {G, y_l, s_l}; so every α_l, β_l follows from the
// small Gram data  SY[a][b] = ⟨s_a,y_b⟩, YY[a][b] = ⟨y_a,y_b⟩, Sg[a] = ⟨s_a,G⟩, Yg[a] = ⟨y_a,G⟩
// by O(h²) scalar arithmetic (k_lbfgs_coeff), and the direction is ONE pass
//     dir = ∓(G − Σ α_l y_l + Σ γ_l s_l)                                  (k_lbfgs_dir, (2h+1)N read)
// instead of the reference's 2h dependent dot/axpy sweeps (48 N of traffic at h = 4).  The Gram
// rows of the newest pair and the dots with the new gradient are accumulated by the pass that
// writes the pair (k_lbfgs_update, src/lbfgs.jl:129-149) — direct dots on the stored vectors, no
// differences of dots.  The history vectors themselves are kept exactly as the reference keeps them — outside the loop;
// inside it the singleton loops keep them as the arrays they are functions of ("ring form" below): the same values, formed
// on the fly.
#pragma once
#include "common.h"

struct FactorArena {
  double* base;      // slot k at base + k*stride
  long long stride;  // doubles, multiple of 32 (256 B)
  int h;             // numlbfgsvecs: slots are R, G, D, s_0..s_{h-1}, y_0..y_{h-1}
};
// arena slot numbers
#define AS_R 0
#define AS_G 1
#define AS_D 2
#define AS_S0 3                     // s_j at AS_S0 + j

// streaming (read-once / write-once) accesses: non-temporal, so that the 300 MB of history that streams
// through every iteration does not evict the 25.6 MB direction the gather kernel is about to re-read
#ifdef SDPLR_NO_NT
__device__ __forceinline__ double2 ldnt2(const double* p, long long i) { return reinterpret_cast<const double2*>(p)[i]; }
__device__ __forceinline__ void stnt2(double* p, long long i, double2 v) { reinterpret_cast<double2*>(p)[i] = v; }
#else
typedef double sdplr_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldnt2(const double* p, long long i) {
  const sdplr_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sdplr_d2*>(p) + i);
  double2 o;
  o.x = t.x;
  o.y = t.y;
  return o;
}
__device__ __forceinline__ void stnt2(double* p, long long i, double2 v) {
  sdplr_d2 t;
  t.x = v.x;
  t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<sdplr_d2*>(p) + i);
}
#endif
__host__ __device__ __forceinline__ double* aslot(const FactorArena& A, int k) { return A.base + (long long)k * A.stride; }
__host__ __device__ __forceinline__ int as_y0(const FactorArena& A) { return AS_S0 + A.h; }  // y_j at as_y0(A) + j

// ---- ring form of the history ---------------------------------------------------------------------------------------
// s_j = α_j·D_j and y_j = G_{j+1} − G_j are functions of arrays the loop writes anyway: the direction D_k (k_lbfgs_dir) and
// the gradient G_{k+1} (the step kernel).  In ring form the loop keeps the last h + 1 of each — the h + 1 "G-like" arrays
// of the arena (G, y_0..y_{h-1}) and its h + 1 "D-like" ones (dirt, s_0..s_{h-1}) used as two rings — and every reader forms
// s and y on the fly: fl(α·d) and fl(g' − g) are the very values lbfgs_update! would have stored (src/lbfgs.jl:142-145), so
// the iterates are those of the stored form bit for bit, and the step kernel writes 2N bytes less per iteration (no s_j, no
// y_j).  Ring position 0 is the array's own place (G / dirt); position p ≥ 1 is history slot (j0 + p − 1) mod h, j0 = latest
// mod h when the ring was entered — the slot of the p-th pair the loop replaces, so the ring grows into the slots of the
// stored pairs exactly as they expire: a loop may enter ring form on ANY stored history (an empty one is h stored pairs of
// zeros), and for its first h iterations the pairs older than the ring are read from their slots as stored.  The control
// block holds the position of the current G (ring_k), the number of pairs in ring form (ring_n), j0 and each ring pair's α
// (ring_alpha, by pair slot); the seam kernel advances them when it folds an update.  Everything outside the loop sees the
// stored form: the host turns the rings back into s_j, y_j, G (k_ring_materialize) before anything else touches the arena.
__host__ __device__ __forceinline__ int ring_slot(int p, int j0, int h) { return (j0 + p - 1) % h; }   // p ≥ 1
__host__ __device__ __forceinline__ double* ring_G(const FactorArena& A, int p, int j0) {
  return p == 0 ? aslot(A, AS_G) : aslot(A, as_y0(A) + ring_slot(p, j0, A.h));
}
__host__ __device__ __forceinline__ double* ring_D(const FactorArena& A, int p, int j0) {
  return p == 0 ? aslot(A, AS_D) : aslot(A, AS_S0 + ring_slot(p, j0, A.h));
}
__host__ __device__ __forceinline__ int ring_back(int k, int i, int h) {   // position i steps behind k
  const int p = (k - i) % (h + 1);
  return p < 0 ? p + h + 1 : p;
}

// ---- two-loop coefficients from the Gram data (one thread, control block staged in LDS) -------------
// Mirrors src/lbfgs.jl:93-113 step by step; j runs newest → oldest, then oldest → newest.
struct SeamLds {
  DevCtrl c;                              // the whole control block: one coalesced load, one coalesced store
  double red[5 * SDPLR_HMAX];
  double nrm[2];                          // Σ‖G‖² partials, Σ‖pv‖² partials
  double al[SDPLR_HMAX], ga[SDPLR_HMAX];  // two-loop work arrays: in LDS so that dynamic indexing
  int order[SDPLR_HMAX];                  // does not fall back to scratch memory
};
static_assert(sizeof(DevCtrl) % 8 == 0, "DevCtrl is copied as 8-byte words");
__device__ inline void lbfgs_coefficients(SeamLds& gd, int h, int latest) {
  if (h == 0) return;
  DevCtrl& c = gd.c;
  int* order = gd.order;
  double *al = gd.al, *ga = gd.ga;
  int j = latest - 1;  // 0-based newest
  for (int i = 0; i < h; i++) {
    order[i] = j;
    j = (j <= 0) ? h - 1 : j - 1;
  }
  for (int i = 0; i < h; i++) {  // α_j = ρ_j ⟨s_j, q⟩,  q = G − Σ_{newer l} α_l y_l
    const int jj = order[i];
    double sq = c.Sg[jj];
    for (int k = 0; k < i; k++) sq -= al[order[k]] * c.SY[jj * SDPLR_HMAX + order[k]];
    al[jj] = c.rho[jj] * sq;
  }
  for (int i = h - 1; i >= 0; i--) {  // β_j = ρ_j ⟨y_j, r⟩,  r = q + Σ_{older l} γ_l s_l
    const int jj = order[i];
    double yr = c.Yg[jj];
    for (int k = 0; k < h; k++) yr -= al[order[k]] * c.YY[jj * SDPLR_HMAX + order[k]];
    for (int k = h - 1; k > i; k--) yr += ga[order[k]] * c.SY[order[k] * SDPLR_HMAX + jj];
    const double beta = c.rho[jj] * yr;
    ga[jj] = al[jj] - beta;      // γ = a − β  (:107)
  }
  for (int l = 0; l < h; l++) {
    c.a[l] = al[l];              // lbfgshis.vecs[j].a[] = α  (:97)
    c.c_alpha[l] = al[l];
    c.c_gamma[l] = ga[l];
  }
}

// the serial part of the seam, on the LDS copy of the control block (thread 0)
__device__ inline void seam_serial(SeamLds& gd, int h, int jfixed, int fin_mode, int do_loop, int do_coeff,
                                   bool fin, bool norms, int desc_mode) {
  DevCtrl& c = gd.c;
  int latest = c.latest;
  if (fin) {
    const int j = (fin_mode == 1) ? (latest % h) : jfixed;
    for (int l = 0; l < h; l++) {
      c.SY[j * SDPLR_HMAX + l] = gd.red[0 * SDPLR_HMAX + l];
      if (l != j) c.SY[l * SDPLR_HMAX + j] = gd.red[1 * SDPLR_HMAX + l];
      c.YY[j * SDPLR_HMAX + l] = gd.red[2 * SDPLR_HMAX + l];
      c.YY[l * SDPLR_HMAX + j] = gd.red[2 * SDPLR_HMAX + l];
      c.Sg[l] = gd.red[3 * SDPLR_HMAX + l];
      c.Yg[l] = gd.red[4 * SDPLR_HMAX + l];
    }
    if (fin_mode == 1) {
      c.rho[j] = 1.0 / c.SY[j * SDPLR_HMAX + j];
      latest = j + 1;
      c.latest = latest;
      c.gram_pending = 0;
    }
  }
  if (norms) {
    const double g = sqrt(gd.nrm[0]), pn = sqrt(gd.nrm[1] + c.pv2_extra);
    c.gnorm = c.grel ? g / c.normC : g;
    c.pvnorm = c.prel ? pn / c.normb : pn;
    c.norms_pending = 0;
  }
  if (do_loop) {
    if (c.done) return;
    if (c.reldelta_exit) {                            // :238-241 (after g! and the norms)
      c.reldelta_exit = 0;
      c.done = 1;
      c.exit_reason = EXIT_RELDELTA;
      return;
    }
    if (c.iters > 0 && c.iters >= c.max_iters) {      // :272-277 (checked after the update)
      c.done = 1;
      c.exit_reason = EXIT_ITERS;
      return;
    }
    if (!(c.gnorm > c.cur_gtol)) {                    // :190
      c.done = 1;
      c.exit_reason = EXIT_GTOL;
      return;
    }
    c.iters += 1;
    c.lastval = c.L;
  }
  if (do_coeff) lbfgs_coefficients(gd, h, latest);
  if (desc_mode) {
    // descent = ⟨dir, G⟩ (src/sdplr.jl:201) of the direction the coefficients above define,
    // dir = ∓(G − Σ α_l y_l + Σ γ_l s_l), from the same Gram data — before the direction exists, so that
    // k_lbfgs_dir can take the steepest-descent fallback of :202-205 in its own pass and no reduction kernel
    // has to sit between the direction and its first consumer.  desc_mode: 1 = negated direction, 2 = not.
    const double g = c.grel ? c.gnorm * c.normC : c.gnorm;
    double d = norms ? gd.nrm[0] : g * g;
    for (int l = 0; l < h; l++) d -= c.c_alpha[l] * c.Yg[l];
    for (int l = 0; l < h; l++) d += c.c_gamma[l] * c.Sg[l];
    if (desc_mode == 1) d = -d;
    c.descent = d;
    c.fallback = (isnan(d) || d >= 0.0) ? 1 : 0;
  }
}

// seam_serial for h = H ≤ 4, the same arithmetic in the same order, laid out for latency: thread 0 fetches what it
// needs from the LDS copy in a few batches of independent reads (the Gram data permuted into time order, newest first,
// so that every index below is a compile-time constant and the recursion runs in registers) instead of one dependent LDS
// round trip per operand — measured with s_memtime on the north-star instance: 11.7 k cycles → see DESIGN.md §10.
template <int H>
__device__ inline void seam_serial_small(SeamLds& gd, int jfixed, int fin_mode, int do_loop, int do_coeff,
                                         bool fin, bool norms, int desc_mode) {
  DevCtrl& c = gd.c;
  // batch 1: every scalar of the loop tests and the norms
  int latest = c.latest;
  const int done0 = c.done, rde = c.reldelta_exit, grel = c.grel, prel = c.prel;
  const long long iters = c.iters, max_iters = c.max_iters;
  const double cur_gtol = c.cur_gtol, Lcur = c.L, normC = c.normC, normb = c.normb, pv2x = c.pv2_extra;
  const double nrm0 = gd.nrm[0], nrm1 = gd.nrm[1];
  double gnorm = c.gnorm;
  if (fin) {
    const int j = (fin_mode == 1) ? (latest % H) : jfixed;
    double rd[5][H];
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
      for (int l = 0; l < H; l++) rd[q][l] = gd.red[q * SDPLR_HMAX + l];
#pragma unroll
    for (int l = 0; l < H; l++) {
      c.SY[j * SDPLR_HMAX + l] = rd[0][l];
      if (l != j) c.SY[l * SDPLR_HMAX + j] = rd[1][l];
      c.YY[j * SDPLR_HMAX + l] = rd[2][l];
      c.YY[l * SDPLR_HMAX + j] = rd[2][l];
      c.Sg[l] = rd[3][l];
      c.Yg[l] = rd[4][l];
    }
    if (fin_mode == 1) {
      double syjj = 0.0;   // = c.SY[j][j] as just stored
#pragma unroll
      for (int l = 0; l < H; l++)
        if (l == j) syjj = rd[0][l];
      c.rho[j] = 1.0 / syjj;
      latest = j + 1;
      c.latest = latest;
      c.gram_pending = 0;
    }
  }
  if (norms) {
    const double g = sqrt(nrm0), pn = sqrt(nrm1 + pv2x);
    gnorm = grel ? g / normC : g;
    c.gnorm = gnorm;
    c.pvnorm = prel ? pn / normb : pn;
    c.norms_pending = 0;
  }
  if (do_loop) {
    if (done0) return;
    if (rde) {                                        // :238-241 (after g! and the norms)
      c.reldelta_exit = 0;
      c.done = 1;
      c.exit_reason = EXIT_RELDELTA;
      return;
    }
    if (iters > 0 && iters >= max_iters) {            // :272-277 (checked after the update)
      c.done = 1;
      c.exit_reason = EXIT_ITERS;
      return;
    }
    if (!(gnorm > cur_gtol)) {                        // :190
      c.done = 1;
      c.exit_reason = EXIT_GTOL;
      return;
    }
    c.iters = iters + 1;
    c.lastval = Lcur;
  }
  // batch 2: the Gram data in time order (position 0 = newest slot), src/lbfgs.jl:93-113
  int ord[H];
  {
    int j = latest - 1;
#pragma unroll
    for (int i = 0; i < H; i++) {
      ord[i] = j;
      j = (j <= 0) ? H - 1 : j - 1;
    }
  }
  double Sg_s[H], Yg_s[H];   // by slot (the descent sum below runs over slots)
#pragma unroll
  for (int l = 0; l < H; l++) {
    Sg_s[l] = c.Sg[l];
    Yg_s[l] = c.Yg[l];
  }
  double al_s[H], ga_s[H];   // coefficients by slot
#pragma unroll
  for (int l = 0; l < H; l++) {
    al_s[l] = c.c_alpha[l];
    ga_s[l] = c.c_gamma[l];
  }
  if (do_coeff) {
    double SYp[H][H], YYp[H][H], Sgp[H], Ygp[H], rhop[H];
#pragma unroll
    for (int a = 0; a < H; a++) {
      Sgp[a] = c.Sg[ord[a]];
      Ygp[a] = c.Yg[ord[a]];
      rhop[a] = c.rho[ord[a]];
#pragma unroll
      for (int b = 0; b < H; b++) {
        SYp[a][b] = c.SY[ord[a] * SDPLR_HMAX + ord[b]];
        YYp[a][b] = c.YY[ord[a] * SDPLR_HMAX + ord[b]];
      }
    }
    double al[H], ga[H];
#pragma unroll
    for (int i = 0; i < H; i++) {        // α_j = ρ_j ⟨s_j, q⟩,  q = G − Σ_{newer l} α_l y_l
      double sq = Sgp[i];
#pragma unroll
      for (int k = 0; k < i; k++) sq -= al[k] * SYp[i][k];
      al[i] = rhop[i] * sq;
    }
#pragma unroll
    for (int i = H - 1; i >= 0; i--) {   // β_j = ρ_j ⟨y_j, r⟩,  r = q + Σ_{older l} γ_l s_l
      double yr = Ygp[i];
#pragma unroll
      for (int k = 0; k < H; k++) yr -= al[k] * YYp[i][k];
#pragma unroll
      for (int k = H - 1; k > i; k--) yr += ga[k] * SYp[k][i];
      const double beta = rhop[i] * yr;
      ga[i] = al[i] - beta;              // γ = a − β  (:107)
    }
#pragma unroll
    for (int i = 0; i < H; i++) {
      c.a[ord[i]] = al[i];               // lbfgshis.vecs[j].a[] = α  (:97)
      c.c_alpha[ord[i]] = al[i];
      c.c_gamma[ord[i]] = ga[i];
#pragma unroll
      for (int l = 0; l < H; l++)
        if (ord[i] == l) {
          al_s[l] = al[i];
          ga_s[l] = ga[i];
        }
    }
  }
  if (desc_mode) {
    const double g = grel ? gnorm * normC : gnorm;
    double d = norms ? nrm0 : g * g;
#pragma unroll
    for (int l = 0; l < H; l++) d -= al_s[l] * Yg_s[l];
#pragma unroll
    for (int l = 0; l < H; l++) d += ga_s[l] * Sg_s[l];
    if (desc_mode == 1) d = -d;
    c.descent = d;
    c.fallback = (isnan(d) || d >= 0.0) ? 1 : 0;
  }
}

// The seam between two inner iterations, one block of 1024 threads:
//  1. fold the partials of the preceding k_lbfgs_update into the Gram data —
//     fin_mode 1: only if that kernel ran (c->gram_pending), then ρ_j = 1/⟨y_j,s_j⟩ (src/lbfgs.jl:146)
//                 and latest = j (:148);  fin_mode 2: row `jfixed` recomputed from stored vectors;
//  2. do_loop: the loop tests of src/sdplr.jl:272-277 and :190, localiter += 1, lastval = ℒ (:207);
//  3. do_coeff: the two-loop coefficients of the next direction.
// Each of the 5h sums is reduced by one wave (lanes stride the per-block partials, fixed order).
// The control block is staged through LDS whole (one coalesced load while the partials are being summed,
// one coalesced store at the end): the serial part then runs on LDS latencies, not on a dozen dependent
// global round trips.
__device__ __forceinline__ void lbfgs_boundary_body(DevCtrl* __restrict__ c, int h, int jfixed, int fin_mode, int do_loop, int do_coeff, int nb_partials, const double* __restrict__ partials, int desc_mode, double* __restrict__ pv_raw, int m) {
  __shared__ SeamLds gd;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef SDPLR_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
  // Everything this kernel reads from global memory is requested in one go — the flags, the control block
  // and the partials (summed whether or not the flags will want them) — so the kernel pays one memory round
  // trip, not flag → count → partials.
  const int gp = c->gram_pending, np = c->norms_pending;
  const int op = c->obj_pending;   // the cost slot's new value left by the step kernel's line-search head (k_fast_step2<…, LSH>)
  // (the control block's words go to registers here and into LDS only after the partials have been requested: an LDS
  // store of a loaded value makes the wave wait for that load where the store stands in the program)
  static_assert(sizeof(DevCtrl) / 8 <= 1024, "one word of the control block per thread");
  const unsigned long long cw = reinterpret_cast<const unsigned long long*>(c)[min(tid, (int)(sizeof(DevCtrl) / 8) - 1)];
  // The partials are fetched 16 bytes per lane (two neighbouring partials): this one CU issues a wave-load every ≈ 16
  // cycles whatever its width, and with 8-byte loads the ≈ 200 of them WERE the load phase (in-kernel stamps: partials
  // in after 3.3 k cycles).  Entries past the producers' count are masked (the slot arrays are SDPLR_MAXNB wide).
  if (wave >= 14) {  // ‖G‖², ‖pv‖² of the iteration that just ended (src/sdplr.jl:224-234): waves 14 and 15
    const int nbp = (wave == 14) ? c->nb_gnorm : c->nb_pvnorm;   // ≤ 1024 on every path
    const double* p = slot_partials(partials, wave == 14 ? SLOT_GNORM2 : SLOT_PVNORM2);
    const double2* p2 = reinterpret_cast<const double2*>(p);
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p2[lane + 64 * u];
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = 2 * (lane + 64 * u);
      t += (i < nbp) ? v[u].x : 0.0;
      t += (i + 1 < nbp) ? v[u].y : 0.0;
    }
    for (int i = lane + 1024; i < nbp; i += 64) t += p[i];
    t = wave_sum(t);
    if (lane == 0) gd.nrm[wave - 14] = t;
  } else if (fin_mode != 0) {  // the 5h Gram sums, two per wave and trip
    const int np2 = (nb_partials + 1) >> 1;
    for (int s0 = wave; s0 < 5 * h; s0 += 28) {
      const int s1 = s0 + 14;
      const bool two = s1 < 5 * h;
      const double2* pa = reinterpret_cast<const double2*>(slot_partials(partials, SLOT_GRAM + (s0 / h) * SDPLR_HMAX + s0 % h));
      const double2* pb = two ? reinterpret_cast<const double2*>(slot_partials(partials, SLOT_GRAM + (s1 / h) * SDPLR_HMAX + s1 % h)) : pa;
      double ta = 0.0, tb = 0.0;
#pragma unroll 4
      for (int i = lane; i < np2; i += 64) {
        const double2 a = pa[i], b = pb[i];
        const bool hi = 2 * i + 1 < nb_partials;
        ta += a.x;
        ta += hi ? a.y : 0.0;
        tb += b.x;
        tb += hi ? b.y : 0.0;
      }
      ta = wave_sum(ta);
      tb = wave_sum(tb);
      if (lane == 0) {
        gd.red[(s0 / h) * SDPLR_HMAX + s0 % h] = ta;
        if (two) gd.red[(s1 / h) * SDPLR_HMAX + s1 % h] = tb;
      }
    }
  }
  if (tid < (int)(sizeof(DevCtrl) / 8)) reinterpret_cast<unsigned long long*>(&gd.c)[tid] = cw;
  const bool fin = (fin_mode == 2) || (fin_mode == 1 && gp);
  const bool norms = np != 0;
  if (!fin && !do_coeff && !do_loop && !norms && !op) return;
  __syncthreads();
  if (tid == 0 && op) {   // obj[] = primal_vio_raw[m+1] (src/linesearch.jl:118-121), on behalf of that head
    gd.c.obj = gd.c.obj_next;
    gd.c.obj_pending = 0;
    if (pv_raw != nullptr) pv_raw[m] = gd.c.obj_next;
  }
  if (tid == 0 && fin_mode == 1 && gd.c.ring_on) {   // ring form: the pair the step kernel has just completed
    if (gp) {
      gd.c.ring_alpha[gd.c.latest % h] = gd.c.alpha;
      gd.c.ring_k = (gd.c.ring_k + 1) % (h + 1);
      gd.c.ring_n = min(gd.c.ring_n + 1, h);
    } else if (gd.c.reldelta_exit) {
      gd.c.ring_unc = 1;
    }
  }
#ifdef SDPLR_STAMPS
  const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
  if (tid == 0) {
    switch (h) {
      case 1: seam_serial_small<1>(gd, jfixed, fin_mode, do_loop, do_coeff, fin, norms, desc_mode); break;
      case 2: seam_serial_small<2>(gd, jfixed, fin_mode, do_loop, do_coeff, fin, norms, desc_mode); break;
      case 3: seam_serial_small<3>(gd, jfixed, fin_mode, do_loop, do_coeff, fin, norms, desc_mode); break;
      case 4: seam_serial_small<4>(gd, jfixed, fin_mode, do_loop, do_coeff, fin, norms, desc_mode); break;
      default: seam_serial(gd, h, jfixed, fin_mode, do_loop, do_coeff, fin, norms, desc_mode);
    }
  }
  __syncthreads();
#ifdef SDPLR_STAMPS
  const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
  {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&gd.c);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(c);
    for (int t = tid; t < (int)(sizeof(DevCtrl) / 8); t += 1024) dst[t] = src[t];
  }
#ifdef SDPLR_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long st3 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && do_loop && (gd.c.iters % 64) == 33) {
    printf("[seam] loads+sums %llu  serial %llu  store %llu (s_memtime ticks)\n", st1 - st0, st2 - st1, st3 - st2);
  }
#endif
}
__global__ void __launch_bounds__(1024)
k_lbfgs_boundary(DevCtrl* __restrict__ c, int h, int jfixed, int fin_mode, int do_loop, int do_coeff,
                 int nb_partials, const double* __restrict__ partials, int desc_mode, double* __restrict__ pv_raw = nullptr,
                 int m = 0) {
  lbfgs_boundary_body(c, h, jfixed, fin_mode, do_loop, do_coeff, nb_partials, partials, desc_mode, pv_raw, m);
}

// ---- direction: dir = ∓(G − Σ α_l y_l + Σ γ_l s_l); y_next = −G; partial ⟨dir, G⟩ -----------------
// src/lbfgs.jl:84 (copy), :94-113 (as one combination), :116-118 (negate), :121-123 (y_next = −grad),
// src/sdplr.jl:201 (descent).
// NTH: the history is read with non-temporal loads — when the iteration's working set (the factor arena) is larger than the
// 256 MiB Infinity Cache, so that the once-per-kernel history streams do not evict the direction the gather kernel is about
// to re-read (north-star instance, 333 MB: +4 %); a working set that FITS the cache (Lovász-θ stand-in, 140 MB) is served
// from it on every pass, and there the hint only loses that (−2.3 %): plain loads.
template <int HM, bool NTH>
__device__ __forceinline__ void lbfgs_dir_body(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int negate, int check_done, double* __restrict__ partials, int inline_fallback) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;  // fetched with the coefficients, tested before the first pass
  double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  const int latest = c->latest;
  // inline_fallback: the seam kernel has already evaluated ⟨dir, G⟩ from the Gram data and decided
  // (c->fallback) whether this direction is replaced by steepest descent, src/sdplr.jl:202-205: G ← −G, dir ← G
  const int fb = inline_fallback ? c->fallback : 0;
  int order[HM];
  double ca[HM], cg[HM];
  {
    int j = latest - 1;
#pragma unroll
    for (int i = 0; i < HM; i++) {
      order[i] = j;
      if (i < h) {
        ca[i] = c->c_alpha[j];
        cg[i] = c->c_gamma[j];
      } else {
        ca[i] = 0.0;
        cg[i] = 0.0;
      }
      j = (j <= 0) ? h - 1 : j - 1;
    }
  }
  // inline_fallback == 2: the fused step + update kernel reads G_old from the G array itself, nothing to park
  double* ynext = (h > 0 && inline_fallback != 2) ? aslot(A, as_y0(A) + (latest % h)) : nullptr;
  const double sgn = negate ? -1.0 : 1.0;
  if (dn) return;
  double desc = 0.0;
  const long long N2 = N >> 1;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  // Slot pointers for the HM unrolled positions: positions k ≥ h reuse a valid slot with a zero coefficient,
  // so that every load of a trip is unconditional and the compiler issues all 2·HM + 1 of them back to back
  // (with `if (k < h)` around the loads hipcc waited for each pair before issuing the next: a chain of memory
  // round trips per trip instead of one).
  const double* yp[HM];
  const double* sp_[HM];
#pragma unroll
  for (int k = 0; k < HM; k++) {
    const int slot = (k < h) ? order[k] : 0;
    yp[k] = (h > 0) ? aslot(A, as_y0(A) + slot) : G;
    sp_[k] = (h > 0) ? aslot(A, AS_S0 + slot) : G;
  }
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
    const double2 g = ldnt2(G, i);
    double2 yv[HM], sv[HM];
#pragma unroll
    for (int k = 0; k < HM; k++) {
      if constexpr (NTH) {
        yv[k] = ldnt2(yp[k], i);
        sv[k] = ldnt2(sp_[k], i);
      } else {
        yv[k] = reinterpret_cast<const double2*>(yp[k])[i];
        sv[k] = reinterpret_cast<const double2*>(sp_[k])[i];
      }
    }
    double2 r = g;
#pragma unroll
    for (int k = 0; k < HM; k++) {  // newest → oldest: q −= α y   (:94-102); ca = 0 beyond h
      r.x -= ca[k] * yv[k].x;
      r.y -= ca[k] * yv[k].y;
    }
#pragma unroll
    for (int k = HM - 1; k >= 0; k--) {  // oldest → newest: r += γ s  (:104-113); cg = 0 beyond h
      r.x += cg[k] * sv[k].x;
      r.y += cg[k] * sv[k].y;
    }
    double2 d;
    d.x = sgn * r.x;
    d.y = sgn * r.y;
    if (fb) {
      d.x = -g.x;
      d.y = -g.y;
      reinterpret_cast<double2*>(G)[i] = d;
    }
    reinterpret_cast<double2*>(dir)[i] = d;
    if (ynext) {
      double2 ng;
      ng.x = -g.x;
      ng.y = -g.y;
      stnt2(ynext, i, ng);
    }
    desc += d.x * g.x + d.y * g.y;
  }
  if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd tail element
    const long long e = N - 1;
    const double g = G[e];
    double r = g;
#pragma unroll
    for (int k = 0; k < HM; k++)
      if (k < h) r -= ca[k] * aslot(A, as_y0(A) + order[k])[e];
#pragma unroll
    for (int k = HM - 1; k >= 0; k--)
      if (k < h) r += cg[k] * aslot(A, AS_S0 + order[k])[e];
    double d = sgn * r;
    if (fb) {
      d = -g;
      G[e] = d;
    }
    dir[e] = d;
    if (ynext) ynext[e] = -g;
    desc += d * g;
  }
  if (inline_fallback) return;  // no consumer for the partials: ⟨dir, G⟩ came from the seam kernel
  desc = block_sum1(desc, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_DESCENT)[blockIdx.x] = desc;
}
template <int HM, bool NTH = true>
__global__ void __launch_bounds__(SDPLR_NT)
k_lbfgs_dir(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int negate,
            int check_done, double* __restrict__ partials, int inline_fallback) {
  lbfgs_dir_body<HM, NTH>(c, A, N, h, negate, check_done, partials, inline_fallback);
}

// The in-loop direction on the ring form (see "ring form" above): the same combination from the same values — y_i =
// G(ring_k − i) − G(ring_k − i − 1), s_i = α_i·D(ring_k − 1 − i), i = 0 the newest pair — written to position ring_k of the D
// ring.  2h + 1 loads per element, as the stored form.  The steepest-descent fallback (decided by the seam kernel from the
// Gram data) writes −G and leaves G alone: the flip G ← −G of src/sdplr.jl:203 is undone by the step kernel's y_j = G_new −
// G_old in the stored form, and here nothing else reads G in between.  N even.
template <int HM, bool NTH>
__global__ void __launch_bounds__(SDPLR_NT)
k_lbfgs_dir_ring(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int check_done) {
  const int dn = check_done ? c->done : 0;
  const int latest = c->latest, rk = c->ring_k, rn = c->ring_n, j0 = c->ring_j0, fb = c->fallback;
  double ca[HM], cg[HM], ra[HM];
  int slot[HM];
  {
    int j = latest - 1;
#pragma unroll
    for (int i = 0; i < HM; i++) {
      const int jj = (i < h && j >= 0) ? j : 0;
      slot[i] = jj;
      const double a0 = c->c_alpha[jj], g0 = c->c_gamma[jj], r0 = c->ring_alpha[jj];
      ca[i] = (i < h) ? a0 : 0.0;
      cg[i] = (i < h) ? g0 : 0.0;
      ra[i] = (i < rn) ? r0 : 1.0;     // a pair still in its slot as stored: s = 1·s
      j = (j <= 0) ? h - 1 : j - 1;
    }
  }
  if (dn) return;
  // gp[0..rn]: the G chain of the ring pairs; gp[i + 1], i ≥ rn: the stored y of the i-th newest pair.  Positions past
  // the history alias the current G (loaded, never used): every load of a trip is unconditional
  const double* gp[HM + 1];
  const double* dp[HM];
  gp[0] = ring_G(A, rk, j0);
#pragma unroll
  for (int i = 0; i < HM; i++) {
    if (i >= h) {
      gp[i + 1] = gp[0];
      dp[i] = gp[0];
    } else if (i < rn) {
      gp[i + 1] = ring_G(A, ring_back(rk, i + 1, h), j0);
      dp[i] = ring_D(A, ring_back(rk, i + 1, h), j0);
    } else {
      gp[i + 1] = aslot(A, as_y0(A) + slot[i]);
      dp[i] = aslot(A, AS_S0 + slot[i]);
    }
  }
  double* dir = ring_D(A, rk, j0);
  const long long N2 = N >> 1;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
    double2 gv[HM + 1], dv[HM];
    gv[0] = ldnt2(gp[0], i);
#pragma unroll
    for (int k = 0; k < HM; k++) {
      if constexpr (NTH) {
        gv[k + 1] = ldnt2(gp[k + 1], i);
        dv[k] = ldnt2(dp[k], i);
      } else {
        gv[k + 1] = reinterpret_cast<const double2*>(gp[k + 1])[i];
        dv[k] = reinterpret_cast<const double2*>(dp[k])[i];
      }
    }
    double2 r = gv[0];
#pragma unroll
    for (int k = 0; k < HM; k++) {  // newest → oldest: q −= α y   (:94-102); ca = 0 beyond h
      const bool ring = k < rn;
      const double yx = ring ? gv[k].x - gv[k + 1].x : gv[k + 1].x, yy = ring ? gv[k].y - gv[k + 1].y : gv[k + 1].y;
      r.x -= ca[k] * yx;
      r.y -= ca[k] * yy;
    }
#pragma unroll
    for (int k = HM - 1; k >= 0; k--) {  // oldest → newest: r += γ s  (:104-113); cg = 0 beyond h
      const double sx = ra[k] * dv[k].x, sy = ra[k] * dv[k].y;
      r.x += cg[k] * sx;
      r.y += cg[k] * sy;
    }
    double2 d;
    d.x = -1.0 * r.x;
    d.y = -1.0 * r.y;
    if (fb) {
      d.x = -gv[0].x;
      d.y = -gv[0].y;
    }
    reinterpret_cast<double2*>(dir)[i] = d;
  }
}

// Ring form → stored form, in place: every element's 2h + 2 ring values are loaded before any of them is overwritten.
// G ← the current gradient; s_j ← α_j·D, y_j ← G' − G for the pairs in ring form (the older ones are in their slots as
// stored: untouched); after a step without lbfgs_update! (unc: the relative-decrease exit) the slot lbfgs_dir! had claimed
// holds y = −G_old (src/lbfgs.jl:121-123) and dirt the unscaled direction.  Otherwise dirt is left to the lazy copy dirt ←
// s_latest.  Arrays by physical index: 0 = G / dirt, 1 + j = history slot j.
template <int HM>
__global__ void __launch_bounds__(SDPLR_NT)
k_ring_materialize(FactorArena A, long long N, int h, int rk, int rn, int unc, int latest, int j0, const DevCtrl* __restrict__ c) {
  auto phys = [&](int p) { return p == 0 ? 0 : 1 + ring_slot(p, j0, h); };
  double ra[HM];
  int pd[HM], pg0[HM], pg1[HM];      // physical index of pair slot j's D, G_old, G_new; −1: the pair is in its slot as stored
#pragma unroll
  for (int j = 0; j < HM; j++) {
    int age = (latest - 1 - j) % (h > 0 ? h : 1);   // 0 = newest
    if (age < 0) age += h;
    const bool ring = j < h && age < rn;
    const int pos = ring_back(rk, 1 + age, h);
    pd[j] = ring ? phys(pos) : -1;
    pg0[j] = ring ? phys(pos) : -1;
    pg1[j] = ring ? phys(ring_back(pos, -1, h)) : -1;
    ra[j] = ring ? c->ring_alpha[j < h ? j : 0] : 0.0;
  }
  const int jnext = latest % (h > 0 ? h : 1);
  double* gp[HM + 1];
  double* dp[HM + 1];
#pragma unroll
  for (int q = 0; q <= HM; q++) {
    gp[q] = (q == 0 || q > h) ? aslot(A, AS_G) : aslot(A, as_y0(A) + q - 1);
    dp[q] = (q == 0 || q > h) ? aslot(A, AS_D) : aslot(A, AS_S0 + q - 1);
  }
  const int qcur = phys(unc ? ring_back(rk, -1, h) : rk), qold = phys(rk);
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) {
    double gv[HM + 1], dv[HM + 1];
#pragma unroll
    for (int q = 0; q <= HM; q++) {
      gv[q] = gp[q][i];
      dv[q] = dp[q][i];
    }
    auto pick = [&](const double* v, int q0) {
      double o = 0.0;
#pragma unroll
      for (int q = 0; q <= HM; q++) o = (q == q0) ? v[q] : o;
      return o;
    };
    const double gcur = pick(gv, qcur);
    const double gold = pick(gv, qold), dcur = pick(dv, qold);
    double sj[HM], yj[HM];
#pragma unroll
    for (int j = 0; j < HM; j++) {
      sj[j] = dv[j + 1];          // as stored
      yj[j] = gv[j + 1];
      if (pd[j] >= 0) {
        sj[j] = ra[j] * pick(dv, pd[j]);
        yj[j] = pick(gv, pg1[j]) - pick(gv, pg0[j]);
      }
      if (unc && j == jnext) yj[j] = -gold;
    }
    gp[0][i] = gcur;
    if (unc) dp[0][i] = dcur;
#pragma unroll
    for (int j = 0; j < HM; j++)
      if (j < h) {
        dp[j + 1][i] = sj[j];
        gp[j + 1][i] = yj[j];
      }
  }
}
// dirt ← scale·(position p of the D ring): all that lbfgs_clear! needs of a ring it is about to drop
__global__ void __launch_bounds__(SDPLR_NT)
k_ring_dirt(double* dirt, const double* src, long long N, double scale) {
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) dirt[i] = scale * src[i];
}

// descent = Σ partials (src/sdplr.jl:201); with `apply`, the steepest-descent fallback of
// src/sdplr.jl:202-205 (G ← −G; dirt ← G) is taken on the device when descent is NaN or ≥ 0.
__global__ void __launch_bounds__(SDPLR_NT)
k_descent(DevCtrl* __restrict__ c, FactorArena A, long long N, int nb_partials, int apply,
          int check_done, const double* __restrict__ partials) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;  // in flight together with the partials
  const double desc = reduce_partials(slot_partials(partials, SLOT_DESCENT), nb_partials, sh);
  if (dn) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) c->descent = desc;
  if (!apply) return;
  if (!(isnan(desc) || desc >= 0.0)) return;
  double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) {
    const double g = -G[i];
    G[i] = g;
    dir[i] = g;
  }
}

// unconditional fallback for the stand-alone operator sdplr_hip_descent_fallback
__global__ void __launch_bounds__(SDPLR_NT) k_neg_copy(double* __restrict__ G, double* __restrict__ dir, long long N) {
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) {
    const double g = -G[i];
    G[i] = g;
    dir[i] = g;
  }
}

// ---- history update + Gram rows of slot j ----------------------------------------------------------
// UPDATE: src/lbfgs.jl:140-146 — dirt *= α; s_j = dirt; y_j += G, j = latest mod h (0-based).
// Both variants accumulate, for the slots l of ONE window of four (l0 ≤ l < l0 + 4; with the new s_j, y_j for l = j):
//   q=0: ⟨s_j, y_l⟩   q=1: ⟨s_l, y_j⟩   q=2: ⟨y_j, y_l⟩   q=3: ⟨s_l, G⟩   q=4: ⟨y_l, G⟩
// into partial slot SLOT_GRAM + q*SDPLR_HMAX + l.  A history longer than four pairs takes one launch per window — the
// first one makes the update (UPDATE), the others read the pair it wrote (jmode 1: j = latest mod h from the control
// block) — instead of one kernel with 5·h running sums per lane: at h = 8 / 16 that kernel spilled 292 / 580 bytes per
// lane to scratch.  !UPDATE with jmode 0 recomputes row `jfixed` from the stored vectors (used when the host has written
// history slots or G behind the library's back).
template <bool UPDATE>
__global__ void __launch_bounds__(SDPLR_NT, 2)
k_lbfgs_update(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int jfixed, int jmode, int l0,
               int check_done, double* __restrict__ partials) {
  constexpr int HW = 4;
  __shared__ double sh[5 * HW * (SDPLR_NT / 64)];
  const int dn = check_done ? (c->done | c->reldelta_exit) : 0;  // :239-241 breaks before lbfgs_update!
  const int j = (UPDATE || jmode) ? (c->latest % h) : jfixed;
  const double alpha = c->alpha;
  if (dn) return;  // (all four scalars were requested together)
  const double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  double* Sj = aslot(A, AS_S0 + j);
  double* Yj = aslot(A, as_y0(A) + j);
  double acc[5 * HW];
#pragma unroll
  for (int k = 0; k < 5 * HW; k++) acc[k] = 0.0;
  const long long N2 = N >> 1;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  // all 2·HW + 4 loads of a trip are unconditional (slots l ≥ h alias slot 0 and are masked out of the sums):
  // see k_lbfgs_dir
  const double* slp[HW];
  const double* ylp[HW];
#pragma unroll
  for (int l = 0; l < HW; l++) {
    const int slot = (l0 + l < h) ? l0 + l : 0;
    slp[l] = aslot(A, AS_S0 + slot);
    ylp[l] = aslot(A, as_y0(A) + slot);
  }
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
    const double2 g = ldnt2(G, i);
    const double2 d = ldnt2(dir, i);
    double2 sn = ldnt2(Sj, i), yn = ldnt2(Yj, i);   // the stored s_j, y_j (y_j holds −G_old, written by lbfgs_dir!)
    double2 sv[HW], yv[HW];
#pragma unroll
    for (int l = 0; l < HW; l++) {
      sv[l] = ldnt2(slp[l], i);
      yv[l] = ldnt2(ylp[l], i);
    }
    if (UPDATE) {
      sn.x = alpha * d.x;  // BLAS.scal!(stepsize, dir)  (:142)
      sn.y = alpha * d.y;
      yn.x += g.x;         // axpy!(1, grad, y_j)  (:145)
      yn.y += g.y;
      reinterpret_cast<double2*>(dir)[i] = sn;
      reinterpret_cast<double2*>(Sj)[i] = sn;  // copy!(s_j, dir)  (:143)
      reinterpret_cast<double2*>(Yj)[i] = yn;
    }
#pragma unroll
    for (int l = 0; l < HW; l++)
      if (l0 + l < h) {
        const double2 sl = (l0 + l == j) ? sn : sv[l];
        const double2 yl = (l0 + l == j) ? yn : yv[l];
        acc[0 * HW + l] += sn.x * yl.x + sn.y * yl.y;
        acc[1 * HW + l] += sl.x * yn.x + sl.y * yn.y;
        acc[2 * HW + l] += yn.x * yl.x + yn.y * yl.y;
        acc[3 * HW + l] += sl.x * g.x + sl.y * g.y;
        acc[4 * HW + l] += yl.x * g.x + yl.y * g.y;
      }
  }
  if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long e = N - 1;
    const double g = G[e];
    double sn, yn;
    if (UPDATE) {
      sn = alpha * dir[e];
      dir[e] = sn;
      Sj[e] = sn;
      yn = Yj[e] + g;
      Yj[e] = yn;
    } else {
      sn = Sj[e];
      yn = Yj[e];
    }
#pragma unroll
    for (int l = 0; l < HW; l++)
      if (l0 + l < h) {
        const double sl = (l0 + l == j) ? sn : aslot(A, AS_S0 + l0 + l)[e];
        const double yl = (l0 + l == j) ? yn : aslot(A, as_y0(A) + l0 + l)[e];
        acc[0 * HW + l] += sn * yl;
        acc[1 * HW + l] += sl * yn;
        acc[2 * HW + l] += yn * yl;
        acc[3 * HW + l] += sl * g;
        acc[4 * HW + l] += yl * g;
      }
  }
  block_sum<5 * HW>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
      for (int l = 0; l < HW; l++)
        if (l0 + l < h) slot_partials(partials, SLOT_GRAM + q * SDPLR_HMAX + l0 + l)[blockIdx.x] = acc[q * HW + l];
    if (UPDATE && blockIdx.x == 0) const_cast<DevCtrl*>(c)->gram_pending = 1;  // consumed by k_lbfgs_boundary
  }
}

// ---- numlbfgsvecs > SDPLR_HMAX: the two-loop recursion as the reference writes it --------------------------------
// The Gram form keeps two h×h blocks in the control block that the seam kernel stages in LDS: fine for the h ≤ 16 anyone
// runs (the reference's default is 4), not for an arbitrary m (`lbfgs_init` allocates any, src/lbfgs.jl:35-47).  Longer
// histories take the literal route: dir ← G (:84); newest → oldest: α_j = ρ_j⟨s_j, dir⟩, dir −= α_j·y_j (:94-102); oldest →
// newest: β = ρ_j⟨y_j, dir⟩, dir += (α_j − β)·s_j (:104-113); dir ← −dir (:116-118), y_next ← −G (:121-123) — one dot kernel
// and one axpy kernel per history vector (4h + 2 launches, 12h·N of traffic: the reference's own cost), every slot found on
// the device from `latest`, so that the sequence is the same enqueue in every iteration (hipGraph replay).  ρ and a live in
// two device arrays of their own.
__device__ __forceinline__ int lit_slot(int latest, int h, int i) {   // slot (0-based) of the i-th newest pair
  int j = (latest - 1 - i) % h;
  return j < 0 ? j + h : j;
}
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_copy(const DevCtrl* __restrict__ c, FactorArena A, long long N, int check_done) {
  if (check_done && c->done) return;
  const double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) dir[i] = G[i];
}
// partials of ⟨v, dir⟩, v = s (sy = 0) or y (sy = 1) of the i-th newest pair
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_dot(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int i, int sy, int slot,
          double* __restrict__ partials, int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;
  const int j = lit_slot(c->latest, h, i);
  if (dn) return;
  const double* v = aslot(A, (sy ? as_y0(A) : AS_S0) + j);
  const double* dir = aslot(A, AS_D);
  double t = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long e = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; e < N; e += stride) t += v[e] * dir[e];
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = t;
}
// mode 0 (first loop): α = ρ_j·Σ, a[j] = α, dir −= α·y_j;  mode 1 (second loop): β = ρ_j·Σ, dir += (a[j] − β)·s_j
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_axpy(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int i, int mode, const double* __restrict__ rho,
           double* __restrict__ a, int slot, int nb, const double* __restrict__ partials, int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;
  const int j = lit_slot(c->latest, h, i);
  const double dot = reduce_partials(slot_partials(partials, slot), nb, sh);   // every block: the same sum in the same order
  if (dn) return;
  double coef;
  if (mode == 0) {
    const double al = rho[j] * dot;
    coef = -al;
    if (blockIdx.x == 0 && threadIdx.x == 0) a[j] = al;
  } else {
    coef = a[j] - rho[j] * dot;
  }
  const double* v = aslot(A, (mode == 0 ? as_y0(A) : AS_S0) + j);
  double* dir = aslot(A, AS_D);
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long e = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; e < N; e += stride) dir[e] += coef * v[e];
}
// dir ← −dir (negate), y_next ← −G, partials of ⟨dir, G⟩ (src/sdplr.jl:201) for k_descent
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_finish(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, int negate, double* __restrict__ partials,
             int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? c->done : 0;
  const int latest = c->latest;
  if (dn) return;
  const double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  double* ynext = aslot(A, as_y0(A) + (latest % h));
  double desc = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long e = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; e < N; e += stride) {
    const double g = G[e];
    const double d = negate ? -dir[e] : dir[e];
    dir[e] = d;
    ynext[e] = -g;
    desc += d * g;
  }
  desc = block_sum1(desc, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_DESCENT)[blockIdx.x] = desc;
}
// lbfgs_update! (src/lbfgs.jl:129-149): dir *= α, s_j = dir, y_j += G, partials of ⟨y_j, s_j⟩; j = latest mod h
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_update(const DevCtrl* __restrict__ c, FactorArena A, long long N, int h, double* __restrict__ partials, int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? (c->done | c->reldelta_exit) : 0;   // :239-241 breaks before lbfgs_update!
  const int j = c->latest % h;
  const double alpha = c->alpha;
  if (dn) return;
  const double* G = aslot(A, AS_G);
  double* dir = aslot(A, AS_D);
  double* Sj = aslot(A, AS_S0 + j);
  double* Yj = aslot(A, as_y0(A) + j);
  double t = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long e = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; e < N; e += stride) {
    const double sn = alpha * dir[e];
    const double yn = Yj[e] + G[e];
    dir[e] = sn;
    Sj[e] = sn;
    Yj[e] = yn;
    t += yn * sn;
  }
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, SLOT_GRAM)[blockIdx.x] = t;
}
// ρ_j = 1/⟨y_j, s_j⟩ (:146), latest = j (:148).  One block.
__global__ void __launch_bounds__(SDPLR_NT)
k_lit_rho(DevCtrl* __restrict__ c, int h, double* __restrict__ rho, int nb, const double* __restrict__ partials, int check_done) {
  __shared__ double sh[8];
  const int dn = check_done ? (c->done | c->reldelta_exit) : 0;
  const int j = c->latest % h;
  const double ys = reduce_partials(slot_partials(partials, SLOT_GRAM), nb, sh);
  if (dn) return;
  if (threadIdx.x == 0) {
    rho[j] = 1.0 / ys;
    c->latest = j + 1;
  }
}

// ---- R += α·dirt  (src/sdplr.jl:219) ------------------------------------------------------------------
__global__ void __launch_bounds__(SDPLR_NT)
k_axpy_R(const DevCtrl* __restrict__ c, double* __restrict__ R, const double* __restrict__ D,
         long long N, int check_done) {
  if (check_done && c->done) return;
  const double alpha = c->alpha;
  const long long N2 = N >> 1;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N2; i += stride) {
    double2 r = reinterpret_cast<double2*>(R)[i];
    const double2 d = reinterpret_cast<const double2*>(D)[i];
    r.x += alpha * d.x;
    r.y += alpha * d.y;
    reinterpret_cast<double2*>(R)[i] = r;
  }
  if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) R[N - 1] += alpha * D[N - 1];
}

// ‖x‖² partials of a flat array (stand-alone norm of G when no fused producer ran)
__global__ void __launch_bounds__(SDPLR_NT)
k_sumsq(const double* __restrict__ x, long long N, int slot, double* __restrict__ partials) {
  __shared__ double sh[8];
  double t = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) t += x[i] * x[i];
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = t;
}
