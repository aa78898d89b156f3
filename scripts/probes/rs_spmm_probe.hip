// dev tool: what bounds the resident route's SpMM (k_resident.h, rs_ell_spmm) on ONE CU?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I sdplrplus.jl_amd/csrc scripts/probes/rs_spmm_probe.hip -o /tmp/rs_spmm_probe
// Variants: real columns / all columns 0 (no bank conflicts: broadcast) / columns = own row (no conflicts, distinct rows)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "k_resident.h"

template <int VEC>
__global__ void __launch_bounds__(SDPLR_RS_NT) k_probe(RsEll E, int n, int r, const double* X, double* out, int reps, long long* cycles) {
  extern __shared__ double lds[];
  for (int e = threadIdx.x; e < n * r; e += SDPLR_RS_NT) lds[e] = X[e];
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < reps; i++) {
    rs_ell_spmm_any<VEC, false>(E, lds, n, r, out);
    __syncthreads();
  }
  if (threadIdx.x == 0) *cycles = (clock64() - t0) / reps;
}

int main(int argc, char** argv) {
  const int n = 800, deg = 48;
  const int r = argc > 1 ? atoi(argv[1]) : 10;
  const int S = (n + 63) / 64;
  for (int variant = 0; variant < 3; variant++) {
    std::vector<int> perm(S * 64, -1), len(S * 64, 0), sptr(S + 1, 0);
    for (int j = 0; j < n; j++) { perm[j] = j; len[j] = deg; }
    for (int s = 0; s < S; s++) sptr[s + 1] = sptr[s] + deg;
    std::vector<unsigned> ent((size_t)sptr[S] * 64, 0u);
    srand(1);
    for (int s = 0; s < S; s++)
      for (int k = 0; k < deg; k++)
        for (int l = 0; l < 64; l++) {
          const int j = s * 64 + l;
          unsigned col = variant == 0 ? (unsigned)(rand() % n) : (variant == 1 ? 0u : (unsigned)std::min(j, n - 1));
          ent[((size_t)sptr[s] + k) * 64 + l] = col;
        }
    std::vector<double> gd(n, 1.0), X((size_t)n * r, 1.0);
    RsEll E{};
    E.n_slices = S; E.one = 0.25;
    int *dperm, *dlen, *dsptr; unsigned* dent; double *dgd, *dX, *dout; long long* dcy;
    hipMalloc(&dperm, perm.size() * 4); hipMalloc(&dlen, len.size() * 4); hipMalloc(&dsptr, sptr.size() * 4);
    hipMalloc(&dent, ent.size() * 4); hipMalloc(&dgd, n * 8); hipMalloc(&dX, X.size() * 8); hipMalloc(&dout, X.size() * 8); hipMalloc(&dcy, 8);
    hipMemcpy(dperm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dlen, len.data(), len.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsptr, sptr.data(), sptr.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dgd, gd.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    E.perm = dperm; E.len = dlen; E.sptr = dsptr; E.ent = dent; E.val = nullptr; E.gdiag = dgd;
    hipFuncSetAttribute((const void*)&k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const size_t lds = (size_t)n * r * 8;
    for (int it = 0; it < 2; it++) k_probe<2><<<1, SDPLR_RS_NT, lds>>>(E, n, r, dX, dout, 50, dcy);
    hipDeviceSynchronize();
    long long cy = 0;
    hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
    printf("r=%d variant %d (%s): %lld clock64 ticks per SpMM (%d entries)\n", r, variant,
           variant == 0 ? "random columns" : variant == 1 ? "all columns 0" : "column = own row", cy, n * deg);
  }
  return 0;
}
