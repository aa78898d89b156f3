// k_eig.h — device kernels of the high-precision eigen path: SDP_S_eigval (src/coreop.jl:351-374) and the ⟨Rt, Rt·S⟩ of
// DIMACS_errors (src/coreop.jl:449).  The reference hands x ↦ S·x + x to GenericArpack.symeigs (implicitly restarted
// Lanczos, ncv ≤ 100 basis vectors); here the same family of method — thick-restart Lanczos with full
// re-orthogonalisation — keeps its basis in HBM ([ncv + 1][n], one vector per row) and only the ncv×ncv projected
// matrix crosses PCIe, once per restart cycle.  All kernels are streaming passes over n-vectors (HBM-bound,
// 8-byte coalesced accesses); reductions are per-block partials + a fixed-order second stage (deterministic).
#pragma once
#include "common.h"

#define SDPLR_EIG_TILE 8   /* basis vectors projected per pass of k_eig_proj (8 running sums per thread) */

// part[i][blk] = Σ_{p ∈ block's stride} V[i][p]·w[p], i < m
__global__ void __launch_bounds__(SDPLR_NT)
k_eig_proj(const double* __restrict__ V, long long ldv, int m, const double* __restrict__ w, int n,
           double* __restrict__ part) {
  __shared__ double sh[SDPLR_EIG_TILE * (SDPLR_NT / 64)];
  const int stride = gridDim.x * SDPLR_NT;
  for (int i0 = 0; i0 < m; i0 += SDPLR_EIG_TILE) {
    double acc[SDPLR_EIG_TILE];
#pragma unroll
    for (int q = 0; q < SDPLR_EIG_TILE; q++) acc[q] = 0.0;
    for (int p = blockIdx.x * SDPLR_NT + threadIdx.x; p < n; p += stride) {
      const double wp = w[p];
#pragma unroll
      for (int q = 0; q < SDPLR_EIG_TILE; q++) {
        const int i = min(i0 + q, m - 1);            // clamped: every load unconditional; surplus sums are dropped below
        acc[q] += V[(long long)i * ldv + p] * wp;
      }
    }
    __syncthreads();
    block_sum<SDPLR_EIG_TILE>(acc, sh);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < SDPLR_EIG_TILE; q++)
        if (i0 + q < m) part[(long long)(i0 + q) * gridDim.x + blockIdx.x] = acc[q];
    }
  }
}

// h[i] = Σ_blk part[i][blk] (one wave per i); tcol[i] (+)= h[i]
__global__ void __launch_bounds__(SDPLR_NT)
k_eig_reduce(int m, int nb, const double* __restrict__ part, double* __restrict__ h, double* __restrict__ tcol,
             int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (SDPLR_NT / 64) + wave;
  if (i >= m) return;
  double s = 0.0;
  for (int b = lane; b < nb; b += 64) s += part[(long long)i * nb + b];
  s = wave_sum(s);
  if (lane == 0) {
    h[i] = s;
    tcol[i] = accumulate ? tcol[i] + s : s;
  }
}

// w −= Σ_{i<m} h[i]·V[i]
__global__ void __launch_bounds__(SDPLR_NT)
k_eig_axpy(const double* __restrict__ V, long long ldv, int m, const double* __restrict__ h, double* __restrict__ w, int n) {
  extern __shared__ double hs[];
  for (int i = threadIdx.x; i < m; i += SDPLR_NT) hs[i] = h[i];
  __syncthreads();
  const int stride = gridDim.x * SDPLR_NT;
  for (int p = blockIdx.x * SDPLR_NT + threadIdx.x; p < n; p += stride) {
    double acc = w[p];
    for (int i = 0; i < m; i++) acc -= hs[i] * V[(long long)i * ldv + p];
    w[p] = acc;
  }
}

// β = ‖w‖ from the partials of k_sumsq; vout = w/β (zero when β is at round-off level: invariant subspace)
__global__ void __launch_bounds__(SDPLR_NT)
k_eig_scale(const double* __restrict__ w, int n, int slot, int nb_partials, const double* __restrict__ partials,
            double* __restrict__ beta_out, double tiny, double* __restrict__ vout) {
  __shared__ double sh[8];
  const double beta = sqrt(reduce_partials(slot_partials(partials, slot), nb_partials, sh));
  if (blockIdx.x == 0 && threadIdx.x == 0) *beta_out = beta;
  const double inv = beta > tiny ? 1.0 / beta : 0.0;
  const int stride = gridDim.x * SDPLR_NT;
  for (int p = blockIdx.x * SDPLR_NT + threadIdx.x; p < n; p += stride) vout[p] = w[p] * inv;
}

// thick restart: Vout[i] = Σ_{l<m} Y[i][l]·V[l], i < k   (Y: k×m row-major, on the device)
__global__ void __launch_bounds__(SDPLR_NT)
k_eig_rotate(const double* __restrict__ V, long long ldv, int m, const double* __restrict__ Y, int k,
             double* __restrict__ Vout, int n) {
  const int stride = gridDim.x * SDPLR_NT;
  for (int p = blockIdx.x * SDPLR_NT + threadIdx.x; p < n; p += stride)
    for (int i0 = 0; i0 < k; i0 += SDPLR_EIG_TILE) {
      double acc[SDPLR_EIG_TILE];
#pragma unroll
      for (int q = 0; q < SDPLR_EIG_TILE; q++) acc[q] = 0.0;
      for (int l = 0; l < m; l++) {
        const double v = V[(long long)l * ldv + p];
#pragma unroll
        for (int q = 0; q < SDPLR_EIG_TILE; q++) acc[q] += Y[(long long)min(i0 + q, k - 1) * m + l] * v;   // (uniform, cached)
      }
#pragma unroll
      for (int q = 0; q < SDPLR_EIG_TILE; q++)
        if (i0 + q < k) Vout[(long long)(i0 + q) * ldv + p] = acc[q];
    }
}

// partials of ⟨a, b⟩ over flat arrays (dot(Rt, Rt·S), src/coreop.jl:449)
__global__ void __launch_bounds__(SDPLR_NT)
k_dot_flat(const double* __restrict__ a, const double* __restrict__ b, long long N, int slot, double* __restrict__ partials) {
  __shared__ double sh[8];
  double t = 0.0;
  const long long stride = (long long)gridDim.x * SDPLR_NT;
  for (long long i = (long long)blockIdx.x * SDPLR_NT + threadIdx.x; i < N; i += stride) t += a[i] * b[i];
  t = block_sum1(t, sh);
  if (threadIdx.x == 0) slot_partials(partials, slot)[blockIdx.x] = t;
}
