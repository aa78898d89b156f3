// sdplr_hip.hip — host side of libsdplr_hip.so: the C ABI of include/sdplr_hip.h over the gfx950
// kernels in k_dense.h / k_sparse.h / k_scalar.h.  One solver handle = one SDP instance resident in
// HBM + one HIP stream.  There is no CPU fallback: without a device every entry point fails.
#include "../../include/sdplr_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"
#include "k_dense.h"
#include "k_scalar.h"
#include "k_sparse.h"
#include "k_eig.h"
#include "k_resident.h"
#include "k_group.h"

namespace {

thread_local std::string g_err;   // errors of calls without a handle (create, warmup, the batch calls' argument checks): per calling thread
// ROCm 7.2: while one host thread records a hipGraph, HIP calls made by OTHER threads fail ("operation
// failed due to a previous error during capture", any capture mode).  Every entry point therefore holds
// this lock shared; a capture takes it exclusively — but only if it gets it within a bounded wait
// (SDPLR_HIP_CAPTURE_WAIT_MS, default 50): other handles' inner loops hold the lock shared for up to their whole
// time budget, and glibc's rwlock prefers readers, so an unbounded wait could starve.  A capture that cannot
// get the lock is skipped: that call launches eagerly (same kernels, same results) and the next call tries again.
using ApiMutex = std::shared_timed_mutex;
using ApiLock = std::shared_lock<ApiMutex>;
ApiMutex g_api_rw;
// The device is sticky: sdplr_hip_set_device records it process-wide, a handle remembers the device it was created on,
// and every entry point binds the CALLING thread to that device before it touches HIP (hipSetDevice is per thread and
// a new thread starts on device 0 — a thread pool setting handles up would otherwise put their streams, pools and arenas
// on GPU 0 whatever the rank).
std::atomic<int> g_device{-1};
inline void bind_device(int dev) {
  if (dev < 0) return;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != dev) (void)hipSetDevice(dev);
}
struct ApiShared {
  ApiLock l{g_api_rw};
  ApiShared() { bind_device(g_device.load(std::memory_order_relaxed)); }
  explicit ApiShared(int dev) { bind_device(dev >= 0 ? dev : g_device.load(std::memory_order_relaxed)); }
};
int capture_wait_ms() {
  static const int ms = getenv("SDPLR_HIP_CAPTURE_WAIT_MS") ? std::max(0, atoi(getenv("SDPLR_HIP_CAPTURE_WAIT_MS"))) : 50;
  return ms;
}

struct LowRankHost {
  int64_t gid, s;
  std::vector<double> B, D;
};

struct ProfEntry {
  int64_t launches = 0;
  double ms = 0.0;
};

}  // namespace

struct BandHost {   // the band plan between its host half and its upload
  bool ok = false, pal = false;
  int NB = 0, NC = 0, BW = 0, CH = 0;
  size_t n_groups = 0;
  std::vector<int> fcp, frv;            // the full pattern (moved out of the solver's staging)
  std::vector<int> blk_slot, slot_g, diagpos;
  std::vector<uint4> cw;
  std::vector<int4> pos;
  std::vector<double> palv;
  // the test knobs, read by the thread that calls finalize (the plan itself may be built on another one, later)
  int env_min_n = 0, env_bw = 0, env_ch = 0, env_nc = 0;
  bool env_no_band = false, env_no_pal = false;
};

struct sdplr_hip_solver {
  int64_t n = 0, m = 0, r = 0, h = 0;
  bool finalized = false;
  hipStream_t stream = nullptr;
  std::string err;

  // ---- host staging of the layout (released at finalize) ----
  bool have_sparse = false;
  int64_t n_sparse = 0, nnzT = 0, nnzS = 0, nnzAgg = 0;
  std::vector<int> h_matptr, h_nzind, h_gids, h_tcp, h_trv, h_fcp, h_frv, h_mapped;
  std::vector<double> h_one, h_two;
  std::vector<LowRankHost> h_lr;

  // ---- device state ----
  // finalize's small uploads and zero-fills are sub-allocated from chunks that go to the device in ONE copy each (a small
  // instance made ≈ 70 synchronous hipMemcpy / hipMemset calls: 1.3 ms alone, 16 ms with 16 handles being built at once)
  struct UploadArena {
    bool on = false;
    char* dev = nullptr;      // uploads: a chunk mirrored in a pinned staging block, one asynchronous copy at commit
    char* host = nullptr;
    size_t cap = 0, used = 0;
    char* zdev = nullptr;     // zero-filled arrays: a chunk of their own, one device fill at commit (nothing crosses PCIe)
    size_t zcap = 0, zused = 0;
  } up;
  int device = -1;           // the HIP device this handle lives on (bound at create; every entry point rebinds the calling thread)
  std::vector<void*> allocs;
  DevSparse sp{};
  DevLowRank lr{};
  bool all_covered = false;  // every slot of an (m+1)-vector is written by some matrix
  // structured fast path (k_sparse.h, DevFast): one sparse matrix with off-diagonal entries
  bool fast = false;
  DevFast ff{};
  DevSparse sp_fast{};       // sp with the segmented-reduction plan restricted to the diagonal-only matrices
  DevSparse spg{};           // the general matrix A_g alone, as a symmetric CSR with fixed values
  DevTile tile{};            // the same matrix as column-sorted K-row tiles (k_spmm_tile)
  bool edge = false;         // edge path (k_sparse.h): disjoint supports, at most one multi-entry sparse matrix
  const int* edge_extra = nullptr;   // its extra slots: the multi-entry matrix and the low-rank matrices
  ExtraHead edge_extra_head{}, extra_head{};   // the first four of each list, passed to k_ls_solve_fast by value
  int n_edge_extra = 0, nb_edge = 1;
  bool edge_lr_fused = false;
  DevBand band{};            // full pattern as LDS-band slices for the Lanczos SpMV (k_lz_band)
  bool use_band = false;
  std::unique_ptr<BandHost> band_host;   // the plan being built on a host thread (finalize → first Lanczos run)
  std::thread band_thread;
  int nb_tile = 0, nb_step = 1, tile_blocks = 512;
  bool use_tile = false;
  bool tiles_deferred = false;   // an instance of the resident route: the tiles (multi-launch route) are built when first needed
  bool tile_attr_done = false;
  bool tile_panels = false;  // SDPLR_HIP_TILE_PANELS: 128-byte half-row gathers, two passes over the lists (experiment)
  // numlbfgsvecs > SDPLR_HMAX: the two-loop recursion as written (k_dense.h, k_lit_*), ρ and a in device arrays of their own
  bool lit = false;
  double *lit_rho = nullptr, *lit_a = nullptr;
  bool no_lshead = false;    // SDPLR_HIP_NO_LSHEAD: the line-search scalar stage stays a kernel of its own
  bool no_team = false;      // a team launch found its members on different XCDs: this handle runs without teams from then on
  double* rs_xch = nullptr;  // the team's exchange block (k_resident.h, RsLoopArgs::xch)
  int rs_ell_parts = 1;      // the team size the sliced ELL was cut for, and each member's first slice
  int rs_ell_part_sl[SDPLR_RS_TEAM_MAX + 1] = {0, 0, 0, 0, 0};
  bool no_pdrop = false;     // SDPLR_HIP_NO_PDROP: the step kernel keeps P = A_g·R (P += α·W) instead of carrying G forward
  bool pdrop_now = false;    // this inner loop runs the P-less step kernel (decided at loop entry)
  // Ring form of the history (k_dense.h): the singleton loops on the tile path keep (α_j, D_j) and (G_j, G_{j+1}) instead of
  // s_j, y_j — 2N bytes less written per iteration.  ring_on: the arena IS in ring form (from a loop that entered it until
  // something outside the loop needs the stored form: ensure_canonical, or lbfgs_clear! drops it).  The ring's scalars as the
  // last loop left them are kept here for those two.
  bool no_ring = false;      // SDPLR_HIP_NO_RING
  bool ring_on = false, ring_now = false;
  int ring_k = 0, ring_n = 0, ring_unc = 0, ring_latest = 0, ring_j0 = 0;
  double ring_alpha[SDPLR_HMAX] = {};
  // G is the gradient at the device's (R, λ, σ) with y as its g! left it: true after fg! / g! / an inner loop, cleared by
  // every other entry point that enqueues work (NEED_FINAL_RW).  The P-less step kernel carries G forward incrementally
  // and needs that; when in doubt the loop takes the P-based kernel, which rebuilds G from P and y.
  bool G_consistent = false;
  int64_t G_age = 0;         // incremental steps since G was last formed from scratch
  // `dirt *= α` of the last lbfgs_update! (src/lbfgs.jl:142) is not stored inside the device-driven loop: dirt = s_latest.
  // The copy is made when something outside the loop looks at dirt (ensure_dirt) — a loop followed by another loop, the
  // common case, overwrites it unread.  ≥ 0: the history slot whose s IS dirt; −1: the D array is current.
  int dirt_from = -1;
  bool no_updfuse = false;   // SDPLR_HIP_NO_UPDFUSE: lbfgs_update! as a kernel of its own on the singleton fast path
  int gram_nb = 1;           // number of Gram partials the latest enqueued producer writes (k_lbfgs_update / fused step)
  bool force_graph = false;  // SDPLR_HIP_FORCE_GRAPH: hipGraph batches on small instances too (the tests' default)
  bool no_lrfuse = false;    // SDPLR_HIP_NO_LRFUSE: low-rank projections by k_lr_project even on the tile path
  bool dot_descent = false;  // SDPLR_HIP_DOT_DESCENT: in-loop ⟨dir, G⟩ by reduction (k_descent) instead of the Gram form
  int tile_lpr = 0;          // the lists are padded to multiples of this sub-wave width
  std::vector<int> h_gptr, h_gcol;   // host CSR of A_g (kept: the tiles are rebuilt when the rank changes)
  std::vector<double> h_gval;
  std::vector<void*> tile_allocs;
  bool fast_singleton = false;   // every diagonal-only matrix has exactly one entry (k_sparse.h, singleton form)
  // resident route (k_resident.h): singleton form, no low-rank matrices, at most one singleton constraint per row
  bool rs_ok = false;
  double* rs_rowvec = nullptr;   // 3n doubles: the resident kernels' per-row vectors when they do not fit the LDS (rs_rows_global)
  const int* rs_row_k = nullptr;     // [n] the constraint attached to row j (−1: none)
  const double* rs_row_v = nullptr;  // [n] its value
  RsEll rs_ell{};                    // A_g as a sliced ELL, one row per lane
  size_t rs_ell_pair_lines = 0;      // Σ over slices of ⌈width/2⌉: 256-byte lines of the two-columns-per-dword copy (Lanczos, LDS)
  const int* extra_slots = nullptr;  // slots not attached to a row: A_g and the low-rank matrices
  int n_extra = 0;
  FactorArena arena{};
  long long N = 0;
  double *lambda = nullptr, *lambda_ub = nullptr, *b = nullptr, *y = nullptr, *pv_raw = nullptr,
         *pv_lb = nullptr, *pv = nullptr, *A_RD = nullptr, *A_DD = nullptr;
  DevCtrl* ctrl = nullptr;   // device
  DevCtrl* hc = nullptr;     // pinned host shadow
  bool hc_valid = false;     // the shadow equals the device block: nothing was enqueued since the last pull / push
  DevCtrl* snap[2] = {nullptr, nullptr};   // pinned snapshots of the control block, one per queued batch
  hipEvent_t snap_ev[2] = {nullptr, nullptr};
  double* partials = nullptr;
  double *lz_buf[3] = {nullptr, nullptr, nullptr}, *lz_v0 = nullptr, *lz_alpha = nullptr, *lz_beta = nullptr;
  int64_t lz_cap = 0;
  double *lr_part = nullptr, *lr_W = nullptr, *lr_WS = nullptr, *lr_coef = nullptr, *lr_btx_part = nullptr;
  double* red10 = nullptr;   // the ten line-search sums, when summed ahead of the scalar stage (k_lr_reduce)
  int nb_lr = 1;

  // kernel shapes
  int LPR = 1, VEC = 1, HM = 4;
  int nb_lzv = 1;  // grid of k_lz_spmv
  int lz_graph_reps = 0;
  int nb_upd = 1;  // grid of k_lbfgs_update (its per-block partials are folded by one block)
  int nb_dense = 1, nb_m = 1, nb_sddmm = 1, nb_spmm = 1, nb_spmv = 1, nb_nnzT = 1, nb_nnzS = 1, nb_n = 1;

  // captured batch of inner iterations (hipGraph), per line-search kind; rebuilt after reset_rank
  hipGraphExec_t graph_exec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // exact line search, Armijo, exact with the P-less step kernel, the last and the first on the ring form
  int graph_iters = 8;
  bool graph_disabled = false;
  hipGraphExec_t lz_graph = nullptr;   // three Lanczos steps (one rotation of the vector buffers)

  // Gram bookkeeping (see k_dense.h)
  bool gram_dirty = false, sg_stale = false, ynext_pending = false;
  // structured paths: P = A_g·R is resident and updated incrementally; it is recomputed from scratch when R was
  // written outside the loop or after SDPLR_P_REFRESH_ITERS incremental updates (drift control).  The loop leaves y
  // current but not the assembled S: whoever reads S next assembles it first (ensure_S).
  bool P_valid = false, S_stale = false;
  // S (assembled, or still to be assembled: S_stale) is S(y) of the device's y.  Cleared when the host writes y or the
  // S values themselves: the Lanczos forms that take S from y and the fixed structure (palette bands, resident ELL)
  // are then skipped in favour of the ones that read the assembled values.
  bool S_from_y = true;
  bool lz_band_now = false;          // this Lanczos run uses the band plan
  int64_t P_age = 0;

  // scratch arrays for operator calls on the caller's own matrices / vectors (SDPLR_F_SCRATCH, SDPLR_V_SCRATCH)
  double* scratchF[2] = {nullptr, nullptr};
  double* scratchV = nullptr;
  // counters (sdplr_hip_get_stats)
  int64_t st_captures = 0, st_capture_failed = 0, st_capture_skipped = 0, st_graph_batches = 0,
          st_eager_batches = 0, st_lz_graph = 0, st_lz_eager = 0, st_iters = 0, st_rs_loops = 0, st_rs_lz = 0, st_rs_fg = 0, st_rs_shared = 0, st_pdrop = 0, st_grp_loops = 0, st_ring_loops = 0, st_ring_materialized = 0;

  // profiling
  bool prof_on = false;
  std::string prof_filter;
  std::vector<std::string> prof_names;
  std::map<std::string, int> prof_index;
  std::vector<ProfEntry> prof;
  struct Pending { int id; hipEvent_t a, b; };
  std::vector<Pending> prof_pending;
  std::vector<hipEvent_t> ev_pool;
};

namespace {
using S = sdplr_hip_solver;
inline int dev_of(const S* s) { return s ? s->device : -1; }
void ensure_S(S* s);

int fail(S* s, int code, const std::string& msg) {
  if (s) s->err = msg; else g_err = msg;
  return code;
}
#define HIPCK(s, call)                                                                      \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail((s), SDPLR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));  \
  } while (0)
#define NEED_FINAL(s)                                                            \
  do {                                                                           \
    if (!(s)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");        \
    if (!(s)->finalized) return fail((s), SDPLR_ERR_STATE, "not finalized");     \
  } while (0)
// entry points that enqueue kernels: the host shadow of the control block is stale until the next pull
int ensure_dirt(S* s);
int ensure_canonical(S* s);
int blocks_for(long long work, int per_block, int cap);
#define NEED_FINAL_RW(s)                                   \
  do {                                                     \
    NEED_FINAL(s);                                         \
    (s)->hc_valid = false;                                 \
    (s)->G_consistent = false;                             \
    if ((s)->ring_on) {                                    \
      const int rc_r__ = ensure_canonical(s);              \
      if (rc_r__) return rc_r__;                           \
    }                                                      \
    if ((s)->dirt_from >= 0) {                             \
      const int rc_d__ = ensure_dirt(s);                   \
      if (rc_d__) return rc_d__;                           \
    }                                                      \
  } while (0)

// … for the entry points that read or write nothing of the factor arena but R (the dual bound and the eigensolvers work on
// S(y) and their own vectors, the λ update on the constraint vectors): a history in ring form (k_dense.h) stays a ring, and the
// lazy dirt ← s_latest stays pending
#define NEED_FINAL_RW_KEEP_HISTORY(s)                      \
  do {                                                     \
    NEED_FINAL(s);                                         \
    (s)->hc_valid = false;                                 \
    (s)->G_consistent = false;                             \
  } while (0)

int have_device() {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

// Host-side set-up loops over independent items (tiles, band blocks, columns) on a few host threads: the set-up of a
// north-star-size instance was ≈ 150 ms of single-threaded sorting and packing (SDPLR_HIP_HOST_THREADS, default ≤ 8).
int host_threads() {
  static const int n = [] {
    int k = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("SDPLR_HIP_HOST_THREADS")) k = atoi(e);
    return std::max(1, std::min(k, 8));
  }();
  return n;
}
template <typename F>
void parallel_for(int64_t n_items, int64_t min_per_thread, F&& body) {   // body(begin, end)
  const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(host_threads(), n_items / std::max<int64_t>(min_per_thread, 1)));
  if (nt <= 1) { body((int64_t)0, n_items); return; }
  std::vector<std::thread> th;
  for (int64_t t = 0; t < nt; t++) {
    const int64_t b = n_items * t / nt, e = n_items * (t + 1) / nt;
    th.emplace_back([&body, b, e] { body(b, e); });
  }
  for (auto& x : th) x.join();
}

// ---- device-memory pool ---------------------------------------------------------------------------------------
// hipMalloc / hipFree take process-wide locks and hipFree waits for the whole device: with eight small instances in
// flight (BASELINE config 5: n = 800, ≈ 60 arrays per handle, a handle created and destroyed per instance) the handles
// stalled each other in the allocator.  Blocks of ≤ 16 MiB go back to a per-device free list keyed by their exact size
// (instances of one batch have the same sizes) instead of to the runtime, ≤ 4 GiB in all (a small instance holds ≈ 14 MB, and a lockstep batch has all of its instances alive at once); larger blocks (the factor
// arena of a big instance) bypass the pool.  A block is only ever returned after the owning handle's stream has been
// drained.  SDPLR_HIP_NO_POOL=1 switches the pool off.
struct DevPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;
  std::unordered_map<void*, std::pair<int, size_t>> live;
  std::vector<void*> pinned_free;   // pinned host blocks of sizeof(DevCtrl)
  std::vector<void*> pinned_chunks_free;   // pinned staging blocks of finalize's upload arena
  // streams and events of destroyed handles: hipStreamCreate / hipStreamDestroy cost 1–2 ms each, more than the rest of
  // finalize on a small instance (a batch creates and destroys a handle per instance)
  std::map<int, std::vector<hipStream_t>> streams_free;
  std::map<int, std::vector<hipEvent_t>> events_free;   // (an event belongs to the device it was created on)
  size_t cached = 0;
  // Slabs: blocks of ≤ SLAB_ITEM bytes are carved out of 64-MiB allocations instead of being hipMalloc'ed one by one — the first
  // batch of 64 small instances made ≈ 750 hipMalloc calls (45 ms of host time over its eight set-up threads: the first
  // set-up 26 ms against 18 ms later; with slabs 20 ms); carved blocks live in the free lists like any other and go back to the runtime with their slab, when
  // pool_trim finds every block of it free.  SDPLR_HIP_NO_SLAB=1: every block its own allocation.
  struct Slab {
    int dev;
    char* base;
    size_t size, used, n_carved, n_cached;
  };
  std::vector<Slab> slabs;
  std::mutex slab_mu;
  const bool no_slab = getenv("SDPLR_HIP_NO_SLAB") != nullptr;
  static constexpr size_t SLAB_BYTES = (size_t)64 << 20, SLAB_ITEM = (size_t)8 << 20;
  int slab_of(const void* p) const {
    for (size_t i = 0; i < slabs.size(); i++)
      if (slabs[i].base != nullptr && (const char*)p >= slabs[i].base && (const char*)p < slabs[i].base + slabs[i].size) return (int)i;
    return -1;
  }
  const bool off = getenv("SDPLR_HIP_NO_POOL") != nullptr;
  static constexpr size_t MAX_BLOCK = (size_t)16 << 20;
  // most bytes of device memory kept cached (SDPLR_HIP_POOL_MAX_MB; default 4 GiB: a lockstep batch has all of its
  // instances alive at once); a process that shares the GPU with other allocators can lower it or call sdplr_hip_trim_pools
  const size_t MAX_CACHED = getenv("SDPLR_HIP_POOL_MAX_MB") ? (size_t)std::max(0LL, atoll(getenv("SDPLR_HIP_POOL_MAX_MB"))) << 20 : (size_t)4 << 30;
};
DevPool& pool() {
  static DevPool* p = new DevPool();   // never destroyed: the HIP runtime may be gone by the time statics are
  return *p;
}
void pool_trim();
// sizes are rounded up to classes an eighth of a power of two apart (≤ 12.5 % more): the arrays of two instances of one family
// differ by a few hundred bytes (their nonzero counts), and a free list keyed by the exact size served only the very same
// instance again
size_t pool_size_class(size_t bytes) {
  bytes = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
  if (bytes <= 4096 || bytes > ((size_t)16 << 20)) return bytes;   // (blocks the pool does not keep are not padded)
  size_t p2 = 1;
  while (p2 * 2 <= bytes) p2 *= 2;
  const size_t step = std::max<size_t>(256, p2 / 8);
  return (bytes + step - 1) / step * step;
}
hipError_t pool_malloc(void** out, size_t bytes) {
  DevPool& P = pool();
  bytes = P.off ? (std::max<size_t>(bytes, 1) + 255) / 256 * 256 : pool_size_class(bytes);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!P.off && bytes <= DevPool::MAX_BLOCK) {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.free_blocks.find({dev, bytes});
    if (it != P.free_blocks.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      P.cached -= bytes;
      P.live[*out] = {dev, bytes};
      const int sl = P.slab_of(*out);
      if (sl >= 0) P.slabs[sl].n_cached--;
      return hipSuccess;
    }
  }
  if (!P.off && !P.no_slab && bytes <= DevPool::SLAB_ITEM) {   // carve it out of a slab
    auto try_carve = [&]() -> bool {
      std::lock_guard<std::mutex> g(P.mu);
      for (auto& sb : P.slabs)
        if (sb.base != nullptr && sb.dev == dev && sb.used + bytes <= sb.size) {
          *out = sb.base + sb.used;
          sb.used += bytes;
          sb.n_carved++;
          P.live[*out] = {dev, bytes};
          return true;
        }
      return false;
    };
    if (try_carve()) return hipSuccess;
    {
      std::lock_guard<std::mutex> one(P.slab_mu);   // (one new slab at a time: eight set-up threads miss together at the start of a batch)
      if (try_carve()) return hipSuccess;
      void* base = nullptr;
      if (hipMalloc(&base, DevPool::SLAB_BYTES) == hipSuccess) {
        {
          std::lock_guard<std::mutex> g(P.mu);
          P.slabs.push_back(DevPool::Slab{dev, (char*)base, DevPool::SLAB_BYTES, 0, 0, 0});
        }
        if (try_carve()) return hipSuccess;
      } else {
        (void)hipGetLastError();   // (no room for a slab: the block gets an allocation of its own below)
      }
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && !P.off) {   // give the cached blocks back and try once more
    (void)hipGetLastError();
    pool_trim();
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess && !P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    P.live[*out] = {dev, bytes};
  }
  return e;
}
void pool_free(void* p) {
  if (!p) return;
  DevPool& P = pool();
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      const auto key = it->second;
      P.live.erase(it);
      const int sl = P.slab_of(p);
      if (sl >= 0) P.slabs[sl].n_cached++;
      if (sl >= 0 || (key.second <= DevPool::MAX_BLOCK && P.cached + key.second <= P.MAX_CACHED)) {   // (a slab's blocks cannot go back one by one)
        P.free_blocks[key].push_back(p);
        P.cached += key.second;
        return;
      }
    }
  }
  (void)hipFree(p);
}
hipError_t pool_host_ctrl(void** out) {
  DevPool& P = pool();
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    if (!P.pinned_free.empty()) {
      *out = P.pinned_free.back();
      P.pinned_free.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc(out, sizeof(DevCtrl), hipHostMallocDefault);
}
void pool_host_ctrl_free(void* p) {
  if (!p) return;
  DevPool& P = pool();
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    if (P.pinned_free.size() < 1024) {
      P.pinned_free.push_back(p);
      return;
    }
  }
  (void)hipHostFree(p);
}

hipError_t pool_stream(hipStream_t* out) {
  DevPool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.streams_free[dev];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void pool_stream_free(hipStream_t st) {   // (drained by the caller)
  if (!st) return;
  DevPool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.streams_free[dev];
    if (v.size() < 256) {   // (a lockstep batch has every instance's handle — and stream — alive at once)
      v.push_back(st);
      return;
    }
  }
  (void)hipStreamDestroy(st);
}
hipError_t pool_event(hipEvent_t* out) {
  DevPool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.events_free[dev];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  // (blocking: a host thread waiting for a resident loop — milliseconds — sleeps instead of spinning; with 16 instances in
  // flight the spinning waiters starved the threads that were setting their own instances up)
  return hipEventCreateWithFlags(out, hipEventDisableTiming | hipEventBlockingSync);
}
void pool_event_free(hipEvent_t e) {
  if (!e) return;
  DevPool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.events_free[dev];
    if (v.size() < 1024) {
      v.push_back(e);
      return;
    }
  }
  (void)hipEventDestroy(e);
}
// everything the pools hold and no handle uses goes back to the runtime (sdplr_hip_trim_pools; also tried once when an
// allocation fails for lack of memory)
void pool_trim() {
  DevPool& P = pool();
  std::map<std::pair<int, size_t>, std::vector<void*>> blocks;
  std::vector<void*> pinned, chunks;
  std::map<int, std::vector<hipStream_t>> streams;
  std::map<int, std::vector<hipEvent_t>> events;
  std::vector<std::pair<int, void*>> slab_bases;
  {
    std::lock_guard<std::mutex> g(P.mu);
    blocks.swap(P.free_blocks);
    // the blocks of a slab that still has a live block stay cached; a slab all of whose blocks are free goes back whole
    for (auto& kv : blocks) {
      std::vector<void*> rest;
      for (void* q : kv.second) {
        const int sl = P.slab_of(q);
        if (sl < 0) { rest.push_back(q); continue; }
        DevPool::Slab& sb = P.slabs[sl];
        if (sb.n_cached == sb.n_carved) continue;                       // (freed with its slab below)
        P.free_blocks[kv.first].push_back(q);
      }
      kv.second.swap(rest);
    }
    size_t still = 0;
    for (auto& kv : P.free_blocks) still += kv.first.second * kv.second.size();
    for (auto& sb : P.slabs)
      if (sb.base != nullptr && sb.n_cached == sb.n_carved) {
        slab_bases.push_back({sb.dev, (void*)sb.base});
        sb.base = nullptr;
        sb.size = sb.used = sb.n_carved = sb.n_cached = 0;
      }
    pinned.swap(P.pinned_free);
    chunks.swap(P.pinned_chunks_free);
    streams.swap(P.streams_free);
    events.swap(P.events_free);
    P.cached = still;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& kv : blocks) { (void)hipSetDevice(kv.first.first); for (void* q : kv.second) (void)hipFree(q); }
  for (auto& sb : slab_bases) { (void)hipSetDevice(sb.first); (void)hipFree(sb.second); }
  for (auto& kv : streams) { (void)hipSetDevice(kv.first); for (hipStream_t st : kv.second) (void)hipStreamDestroy(st); }
  for (auto& kv : events) { (void)hipSetDevice(kv.first); for (hipEvent_t e : kv.second) (void)hipEventDestroy(e); }
  (void)hipSetDevice(cur);
  for (void* q : pinned) (void)hipHostFree(q);
  for (void* q : chunks) (void)hipHostFree(q);
}

template <typename T>
int dalloc(S* s, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = pool_malloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
  s->allocs.push_back(q);
  *p = (T*)q;
  return SDPLR_OK;
}
constexpr size_t ARENA_ITEM_MAX = (size_t)256 << 10, ARENA_CHUNK = (size_t)4 << 20;
hipError_t pool_host_chunk(void** out) {   // pinned staging blocks of ARENA_CHUNK bytes
  DevPool& P = pool();
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    if (!P.pinned_chunks_free.empty()) {
      *out = P.pinned_chunks_free.back();
      P.pinned_chunks_free.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc(out, ARENA_CHUNK, hipHostMallocDefault);
}
void pool_host_chunk_free(void* p) {
  if (!p) return;
  DevPool& P = pool();
  if (!P.off) {
    std::lock_guard<std::mutex> g(P.mu);
    if (P.pinned_chunks_free.size() < 32) {
      P.pinned_chunks_free.push_back(p);
      return;
    }
  }
  (void)hipHostFree(p);
}
int arena_commit(S* s) {   // the pending chunks → device
  S::UploadArena& u = s->up;
  bool wait = false;
  if (u.dev && u.used > 0) {
    HIPCK(s, hipMemcpyAsync(u.dev, u.host, u.used, hipMemcpyHostToDevice, s->stream));
    wait = true;
  }
  if (u.zdev && u.zused > 0) HIPCK(s, hipMemsetAsync(u.zdev, 0, u.zused, s->stream));
  if (wait) HIPCK(s, hipStreamSynchronize(s->stream));   // (the staging block goes back to the pool)
  if (u.host) pool_host_chunk_free(u.host);
  u.dev = u.host = u.zdev = nullptr;
  u.cap = u.used = u.zcap = u.zused = 0;
  return SDPLR_OK;
}
// room for `bytes` in the current chunk (src == nullptr: zeros); *out = nullptr ⇒ not an arena item
int arena_take(S* s, size_t bytes, const void* src, void** out) {
  *out = nullptr;
  S::UploadArena& u = s->up;
  const size_t b = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
  if (!u.on || b > ARENA_ITEM_MAX) return SDPLR_OK;
  if (src == nullptr) {
    if (u.zused + b > u.zcap) {
      if (u.zdev && u.zused > 0) HIPCK(s, hipMemsetAsync(u.zdev, 0, u.zused, s->stream));
      void* q = nullptr;
      hipError_t e = pool_malloc(&q, ARENA_CHUNK);
      if (e != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
      s->allocs.push_back(q);
      u.zdev = (char*)q;
      u.zcap = ARENA_CHUNK;
      u.zused = 0;
    }
    *out = u.zdev + u.zused;
    u.zused += b;
    return SDPLR_OK;
  }
  if (u.used + b > u.cap) {
    if (u.dev && u.used > 0) {
      HIPCK(s, hipMemcpyAsync(u.dev, u.host, u.used, hipMemcpyHostToDevice, s->stream));
      HIPCK(s, hipStreamSynchronize(s->stream));
    }
    void* q = nullptr;
    hipError_t e = pool_malloc(&q, ARENA_CHUNK);
    if (e != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    s->allocs.push_back(q);
    if (!u.host) {
      void* hp = nullptr;
      e = pool_host_chunk(&hp);
      if (e != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, std::string("hipHostMalloc: ") + hipGetErrorString(e));
      u.host = (char*)hp;
    }
    u.dev = (char*)q;
    u.cap = ARENA_CHUNK;
    u.used = 0;
  }
  if (bytes) memcpy(u.host + u.used, src, bytes);
  *out = u.dev + u.used;
  u.used += b;
  return SDPLR_OK;
}
// Large host → device copies of pageable memory (the layout arrays of finalize, a factor from set_factor) go through two
// pooled pinned 4 MB blocks, the host copy of one overlapping the DMA of the other: the runtime's own staging of
// pageable memory moved ≈ 1 GB/s (25.6 MB of R₀ in 32 ms; 60 MB of layout arrays in 26 ms).  Small copies stay plain.
int h2d_copy(S* s, void* dst, const void* src, size_t bytes) {
  if (bytes < ((size_t)1 << 20) || getenv("SDPLR_HIP_NO_PINNED_PIPE") != nullptr) {
    HIPCK(s, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
    HIPCK(s, hipStreamSynchronize(s->stream));
    return SDPLR_OK;
  }
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; k++) {
    e = pool_host_chunk(&buf[k]);
    if (e == hipSuccess) e = pool_event(&ev[k]);
  }
  bool used[2] = {false, false};
  for (size_t off = 0, k = 0; off < bytes && e == hipSuccess; off += ARENA_CHUNK, k ^= 1) {
    const size_t len = std::min(ARENA_CHUNK, bytes - off);
    if (used[k]) e = hipEventSynchronize(ev[k]);      // the block's previous DMA has left it
    if (e != hipSuccess) break;
    memcpy(buf[k], (const char*)src + off, len);
    e = hipMemcpyAsync((char*)dst + off, buf[k], len, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[k], s->stream);
    used[k] = true;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  for (int k = 0; k < 2; k++) {
    if (buf[k]) pool_host_chunk_free(buf[k]);
    if (ev[k]) pool_event_free(ev[k]);
  }
  if (e != hipSuccess) return fail(s, SDPLR_ERR_HIP, std::string("h2d_copy: ") + hipGetErrorString(e));
  return SDPLR_OK;
}
template <typename T>
int upload(S* s, const T** dst, const std::vector<T>& v) {
  void* a = nullptr;
  int rc = arena_take(s, v.size() * sizeof(T), v.data(), &a);
  if (rc) return rc;
  if (a) { *dst = (const T*)a; return SDPLR_OK; }
  T* p = nullptr;
  rc = dalloc(s, &p, v.size());
  if (rc) return rc;
  if (!v.empty() && (rc = h2d_copy(s, p, v.data(), v.size() * sizeof(T)))) return rc;
  *dst = p;
  return SDPLR_OK;
}
int dzero(S* s, double** p, size_t count) {
  void* a = nullptr;
  int rc = arena_take(s, count * sizeof(double), nullptr, &a);
  if (rc) return rc;
  if (a) { *p = (double*)a; return SDPLR_OK; }
  rc = dalloc(s, p, count);
  if (rc) return rc;
  // zero-fill ON THE SOLVER'S STREAM: a legacy-stream hipMemset is not ordered with this non-blocking
  // stream, so a late zero-fill could wipe an upload issued after it (seen with 8 concurrent handles)
  HIPCK(s, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(double), s->stream));
  return SDPLR_OK;
}

// ---- profiling (hipEvent pairs on the solver's stream) -----------------------------------------------
struct ProfScope {
  S* s;
  bool on;
  S::Pending p{};
  ProfScope(S* s_, const char* name) : s(s_), on(s_->prof_on) {
    if (on && !s->prof_filter.empty() && s->prof_filter != name) on = false;
    if (!on) return;
    auto it = s->prof_index.find(name);
    int id;
    if (it == s->prof_index.end()) {
      id = (int)s->prof_names.size();
      s->prof_names.push_back(name);
      s->prof_index[name] = id;
      s->prof.emplace_back();
    } else {
      id = it->second;
    }
    auto get = [&]() {
      hipEvent_t e;
      if (!s->ev_pool.empty()) { e = s->ev_pool.back(); s->ev_pool.pop_back(); }
      else (void)hipEventCreate(&e);
      return e;
    };
    p.id = id;
    p.a = get();
    p.b = get();
    (void)hipEventRecord(p.a, s->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(p.b, s->stream);
    s->prof_pending.push_back(p);
  }
};
void prof_drain(S* s) {
  if (s->prof_pending.empty()) return;
  (void)hipStreamSynchronize(s->stream);
  for (auto& p : s->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      s->prof[p.id].launches++;
      s->prof[p.id].ms += ms;
    }
    s->ev_pool.push_back(p.a);
    s->ev_pool.push_back(p.b);
  }
  s->prof_pending.clear();
}

// ---- control block transfer -------------------------------------------------------------------------
int pull(S* s) {
  HIPCK(s, hipMemcpyAsync(s->hc, s->ctrl, sizeof(DevCtrl), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  s->hc_valid = true;
  return SDPLR_OK;
}
// the same, for waits that are long (a resident loop, a resident Lanczos run): the host thread sleeps on a blocking event
int pull_blocking(S* s) {
  HIPCK(s, hipMemcpyAsync(s->hc, s->ctrl, sizeof(DevCtrl), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipEventRecord(s->snap_ev[0], s->stream));
  HIPCK(s, hipEventSynchronize(s->snap_ev[0]));
  s->hc_valid = true;
  return SDPLR_OK;
}
// the shadow, fetched only if something may have changed the device block since it was last seen (a small solve makes
// ≈ 40 scalar reads / writes: each was a device round trip)
int pull_if_stale(S* s) { return s->hc_valid ? SDPLR_OK : pull(s); }
int push(S* s) {
  HIPCK(s, hipMemcpyAsync(s->ctrl, s->hc, sizeof(DevCtrl), hipMemcpyHostToDevice, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SDPLR_OK;   // (hc_valid is left as it is: a caller that enqueues kernels after this clears it at its entry)
}
int ensure_dirt(S* s) {   // dirt ← s_latest (see S::dirt_from); enqueued on the handle's stream, ahead of whatever reads dirt next
  { const int rc_r = ensure_canonical(s); if (rc_r) return rc_r; }   // (s_latest as a stored vector)
  if (s->dirt_from < 0) return SDPLR_OK;
  const int j = s->dirt_from;
  s->dirt_from = -1;
  HIPCK(s, hipMemcpyAsync(aslot(s->arena, AS_D), aslot(s->arena, AS_S0 + j), s->N * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  return SDPLR_OK;
}
// ring form → stored form (k_dense.h, k_ring_materialize), enqueued on the handle's stream ahead of whatever asked for it
int ensure_canonical(S* s) {
  if (!s->ring_on) return SDPLR_OK;
  s->ring_on = false;
  const int nb = blocks_for(s->N, SDPLR_NT, 2048);
  k_ring_materialize<4><<<nb, SDPLR_NT, 0, s->stream>>>(s->arena, s->N, (int)s->h, s->ring_k, s->ring_n, s->ring_unc, s->ring_latest, s->ring_j0, s->ctrl);
  HIPCK(s, hipGetLastError());
  // the device's control block leaves ring form too (the stored-form kernels never look, the seam does)
  HIPCK(s, hipMemsetAsync(&s->ctrl->ring_on, 0, sizeof(int), s->stream));
  if (s->hc) s->hc->ring_on = 0;
  s->st_ring_materialized++;
  return SDPLR_OK;
}
int sync_check(S* s) {
  HIPCK(s, hipGetLastError());
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SDPLR_OK;
}

// dynamic LDS the resident loop needs (k_resident.h) and its budget; tiles_can_wait: "small enough for that route"
constexpr size_t RS_LDS_MAX = 150 * 1024;   // dynamic LDS of a resident kernel (≈ 9.6 KB more are static: control block, reduction scratch)
// D (with the SpMM's zero row) | ⟨R_j,D_j⟩ | ‖D_j‖² | d_j | four r-vectors of the rank-one matrix.  When that is more than
// a CU has but the direction alone fits, the three per-row vectors move to global memory (k_resident.h, RsLoopArgs::rowvec).
// Workgroups per instance on the resident loop (k_resident.h, TEAM): a function of the instance alone — never of the batch it
// travels in — so that batch calls and single calls run the same kernel on it.  The P-less loop without a rank-one matrix.
int rs_team_want(const S* s) {
  if (s->no_team) return 1;
  const char* e = getenv("SDPLR_HIP_TEAM");
  // (three: a 64-instance batch — 192 workgroups — is still co-resident on the 256 CUs, lockstep calls 26.9 → 25.3 ms against two;
  // four — 29 µs per iteration alone — needs every CU of the GPU for such a batch and loses: 33.6 ms)
  const int want = e ? atoi(e) : 3;
  if (want < 2 || s->n < 128 || s->h < 1 || s->h > 4 || s->no_pdrop || s->ff.gid_g != (int)s->m) return 1;
  return std::min(want, SDPLR_RS_TEAM_MAX);
}
// … and the sliced ELL has to have been cut for that team (build_rs_ell: the members' row ranges are runs of whole slices)
int rs_team_w(const S* s) {
  const int want = rs_team_want(s);
  return (want > 1 && want == s->rs_ell_parts) ? want : 1;
}
bool rs_rows_global(const S* s) {
  const size_t full = ((size_t)rs_npad(s->n, s->r) + 3 * (size_t)s->n + 4 * (size_t)s->r) * sizeof(double);
  return full > RS_LDS_MAX && getenv("SDPLR_HIP_NO_RESIDENT_ROWVEC") == nullptr;
}
size_t rs_loop_lds(const S* s) {
  const size_t rows = rs_rows_global(s) ? 0 : 3 * (size_t)s->n;
  return ((size_t)rs_npad(s->n, s->r) + rows + 4 * (size_t)s->r) * sizeof(double);
}
RsLr rs_lr(const S* s) {   // the one rank-one matrix of an instance of the resident route (finalize: lr_one), or none
  RsLr l{};
  l.B = nullptr; l.D = 0.0; l.gid = -1;
  if (s->h_lr.size() == 1 && s->h_lr[0].s == 1) { l.B = s->lr.Bcat; l.D = s->h_lr[0].D[0]; l.gid = (int)s->h_lr[0].gid; }
  return l;
}
bool tiles_can_wait(const S* s) {
  return !s->force_graph && getenv("SDPLR_HIP_NO_RESIDENT") == nullptr && rs_loop_lds(s) <= RS_LDS_MAX;
}

int blocks_for(long long work, int per_block, int cap) {
  long long b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// release one tracked allocation
void dfree(S* s, void* p) {
  if (!p) return;
  pool_free(p);
  s->allocs.erase(std::remove(s->allocs.begin(), s->allocs.end(), p), s->allocs.end());
}

double* factor_ptr(S* s, int32_t slot) {
  if (slot == SDPLR_F_SCRATCH || slot == SDPLR_F_SCRATCH + 1) {
    double*& p = s->scratchF[slot - SDPLR_F_SCRATCH];
    if (!p && dzero(s, &p, (size_t)s->N) != SDPLR_OK) return nullptr;
    return p;
  }
  if (slot == SDPLR_F_RT) return aslot(s->arena, AS_R);
  if (slot == SDPLR_F_GT) return aslot(s->arena, AS_G);
  if (slot == SDPLR_F_DIRT) return aslot(s->arena, AS_D);
  if (slot >= SDPLR_F_LBFGS_S && slot < SDPLR_F_LBFGS_S + s->h) return aslot(s->arena, AS_S0 + (slot - SDPLR_F_LBFGS_S));
  if (slot >= SDPLR_F_LBFGS_Y && slot < SDPLR_F_LBFGS_Y + s->h) return aslot(s->arena, as_y0(s->arena) + (slot - SDPLR_F_LBFGS_Y));
  return nullptr;
}
void note_factor_written(S* s, int32_t slot) {
  if (slot == SDPLR_F_RT) s->P_valid = false;
  if (slot == SDPLR_F_GT) s->sg_stale = true;
  if (slot >= SDPLR_F_LBFGS_S && slot < SDPLR_F_SCRATCH) s->gram_dirty = true;
}

// sub-wave shape for a given rank: VEC doubles per lane, LPR lanes per row (power of two ≤ 64)
void choose_shape(S* s) {
  s->VEC = (s->r % 2 == 0) ? 2 : 1;
  long long need = (s->r + s->VEC - 1) / s->VEC;
  int lpr = 1;
  while (lpr < need && lpr < 64) lpr <<= 1;
  s->LPR = lpr;
  s->HM = s->h <= 4 ? 4 : (s->h <= 8 ? 8 : 16);
}

// Column-sorted row tiles of A_g for k_spmm_tile (k_sparse.h), from the host copy of its CSR.  Tiles are
// ⌊n/T⌋ or ⌈n/T⌉ consecutive rows; T fills 1024 blocks (4 per CU) exactly when that keeps a tile within 8
// rows, so every CU carries the same number of equally long lists and all sweeps stay in phase.  Every list is
// padded to a multiple of the sub-wave width with (dump row K, column = the tile's first row, value 0) entries,
// and the arrays end with 2·64 more entries so that the kernel's look-ahead stays in bounds.
// Sub-wave width of the column-sweep tile kernel.  Rows wider than one DPP row (r > 32 at two doubles per lane) are
// swept in several passes of the 16-lane shape — the kernel's loop over row chunks — instead of with 32- or 64-lane
// groups: those hand entries round by ds_bpermute, run at 2 waves per SIMD, and put half / a quarter as many rows in a
// block (n = 1e5: 172 µs per launch at r = 64 and 355 µs at r = 128 against 54 µs at r = 32, for 2× / 4× the bytes).
int tile_shape_lpr(const S* s) {
  static const bool off = getenv("SDPLR_HIP_TILE_WIDE_GROUPS") != nullptr;
  return (!off && s->VEC == 2 && s->LPR > 16) ? 16 : s->LPR;
}
int build_tiles(S* s) {
  const int64_t n = s->n;
  struct ArenaOff {   // (the tile arrays are released one by one when the rank changes: never arena items)
    S* s; bool was;
    explicit ArenaOff(S* s_) : s(s_), was(s_->up.on) { s->up.on = false; }
    ~ArenaOff() { s->up.on = was; }
  } arena_off(s);
  for (void* p : s->tile_allocs) {
    pool_free(p);
    s->allocs.erase(std::remove(s->allocs.begin(), s->allocs.end(), p), s->allocs.end());
  }
  s->tile_allocs.clear();
  s->use_tile = false;
  s->tile_attr_done = false;
  s->tile = DevTile{};
  if (s->h_gptr.empty() || n >= (1LL << SDPLR_TILE_COLBITS) || getenv("SDPLR_HIP_NO_TILE") != nullptr) return SDPLR_OK;
  const std::vector<int>&g_ptr = s->h_gptr, &g_col = s->h_gcol;
  const std::vector<double>& g_val = s->h_gval;
  choose_shape(s);
  const int L = tile_shape_lpr(s), G = SDPLR_NT / L;
  // Tile height and grid: the tallest tiles of which TWO blocks fit a CU's 160 KB of LDS (16 rows at r = 32), cut for
  // 512 blocks — one resident round of 2 blocks per CU.  Measured on the north-star instance (µs per launch):
  // 1024 blocks × ≤ 8 rows 77, 768 × ≤ 10 64, 512 × ≤ 16 58, 256 × ≤ 32 69; a grid that is not a whole number of
  // resident rounds (896, 640 blocks; 512 blocks with ≤ 12 rows) costs 77–84.  Longer column-sorted lists keep the
  // groups closer together in the sweep (the spread of a list's position shrinks as 1/√length), so more of the band
  // of D they are reading is still in the XCD's L2; fewer resident groups narrow the band further.
  const size_t lds_row = (size_t)SDPLR_NT * s->VEC * sizeof(double), lds_fix = (size_t)G * 8 * sizeof(double) + 2048;
  int kmax = (int)std::max<size_t>(1, std::min<size_t>(16, (80 * 1024 - lds_fix) / (lds_row + (size_t)G * 16) - 1));
  int tblocks = 512;
  if (const char* e = getenv("SDPLR_HIP_TILE_KMAX")) kmax = std::max(1, std::min(atoi(e), 40));
  if (const char* e = getenv("SDPLR_HIP_TILE_BLOCKS")) tblocks = std::max(64, std::min(atoi(e), 1024));
  const int64_t cap = (int64_t)tblocks * G, nnz = g_ptr[n];
  std::vector<int> t_row;
  if (const char* e = getenv("SDPLR_HIP_TILE_K")) {   // fixed tile height (tests)
    const int kk = std::max(1, std::min(atoi(e), 16));
    for (int64_t r0 = 0; r0 < n; r0 += kk) t_row.push_back((int)r0);
  } else {
    // Tiles of (nearly) equal NONZERO count, at most 8 rows each: a group's position in the column sweep is
    // the fraction of its list it has consumed, so equally long lists keep all groups on the same narrow band
    // of D (with equal row counts the lists differ by ±15 % and the band outgrows the L2).  The tile count
    // is steered to at most `cap` so that the grid is one resident wave of blocks.
    int64_t want = std::min<int64_t>(n, cap);
    for (int attempt = 0; attempt < 6; attempt++) {
      t_row.clear();
      int64_t r0 = 0, t = 0;
      while (r0 < n) {
        t_row.push_back((int)r0);
        const double goal = (double)(t + 1) * (double)(nnz + n) / (double)want;   // rows count as one entry too
        int64_t r1 = r0 + 1;
        while (r1 < n && r1 - r0 < kmax && (double)(g_ptr[r1 + 1] + r1 + 1) <= goal) r1++;
        r0 = r1;
        t++;
      }
      if ((int64_t)t_row.size() <= cap || (n + kmax - 1) / kmax > cap) break;
      want -= ((int64_t)t_row.size() - cap) + 8;
      if (want < 1) { want = 1; }
    }
  }
  const int64_t nt = (int64_t)t_row.size();
  t_row.push_back((int)n);
  std::vector<int> t_ptr(nt + 1, 0), t_ent;
  int K = 1;
  for (int64_t t = 0; t < nt; t++) K = std::max(K, t_row[t + 1] - t_row[t]);
  std::vector<double> t_val;
  // uniform form (k_sparse.h, DevTile): all off-diagonal values equal — the lists then hold the off-diagonal entries only
  bool uni = getenv("SDPLR_HIP_NO_TILE_UNIFORM") == nullptr;
  double uni_one = 0.0;
  std::vector<double> gdiag;
  std::vector<int> offd_cnt;   // per row: off-diagonal entries
  {
    bool have = false;
    for (int64_t j = 0; j < n && uni; j++)
      for (int q = g_ptr[j]; q < g_ptr[j + 1]; q++) {
        if (g_col[q] == (int)j) continue;
        if (!have) { uni_one = g_val[q]; have = true; }
        else if (!(g_val[q] == uni_one)) { uni = false; break; }
      }
    if (!have) uni = false;
    if (uni) {
      gdiag.assign(n, 0.0);
      offd_cnt.assign(n, 0);
      for (int64_t j = 0; j < n; j++)
        for (int q = g_ptr[j]; q < g_ptr[j + 1]; q++) {
          if (g_col[q] == (int)j) gdiag[j] += g_val[q];
          else offd_cnt[j]++;
        }
    }
  }
  for (int64_t t = 0; t < nt; t++) {   // list lengths, padded to the sub-wave width
    int len = g_ptr[t_row[t + 1]] - g_ptr[t_row[t]];
    if (uni) {
      len = 0;
      for (int j = t_row[t]; j < t_row[t + 1]; j++) len += offd_cnt[j];
    }
    t_ptr[t + 1] = t_ptr[t] + (len + L - 1) / L * L;
  }
  t_ent.assign((size_t)t_ptr[nt] + 128, 0);
  t_val.assign(uni ? 1 : (size_t)t_ptr[nt] + 128, 0.0);
  parallel_for(nt, 64, [&](int64_t t0, int64_t t1) {   // the tiles are independent: sorted and packed on a few host threads
    std::vector<unsigned long long> tmp;  // column << 32 | position in the tile's slice of the CSR
    std::vector<int> rowof;
    for (int64_t t = t0; t < t1; t++) {
      const int64_t r0 = t_row[t], r1 = t_row[t + 1];
      const int b = g_ptr[r0], e = g_ptr[r1];
      tmp.resize(e - b);
      for (int q = b; q < e; q++) tmp[q - b] = ((unsigned long long)(unsigned)g_col[q] << 32) | (unsigned)(q - b);
      std::sort(tmp.begin(), tmp.end());  // by column, then by CSR position (= by row)
      int64_t row = r0;
      rowof.assign(e - b, 0);
      for (int q = b; q < e; q++) { while (q >= g_ptr[row + 1]) row++; rowof[q - b] = (int)(row - r0); }
      size_t at0 = (size_t)t_ptr[t];
      for (int q = 0; q < e - b; q++) {
        const int col = (int)(tmp[q] >> 32), at = (int)(tmp[q] & 0xFFFFFFFFull);
        if (uni && col == (int)r0 + rowof[at]) continue;   // the diagonal entry rides the epilogue (gdiag)
        t_ent[at0] = (rowof[at] << SDPLR_TILE_COLBITS) | col;
        if (!uni) t_val[at0] = g_val[b + at];
        at0++;
      }
      for (; at0 < (size_t)t_ptr[t + 1]; at0++) { t_ent[at0] = (K << SDPLR_TILE_COLBITS) | (int)r0; if (!uni) t_val[at0] = 0.0; }
    }
  });
  int rc;
  const size_t first = s->allocs.size();
  if ((rc = upload(s, &s->tile.row0, t_row))) return rc;
  if ((rc = upload(s, &s->tile.ptr, t_ptr))) return rc;
  if ((rc = upload(s, &s->tile.ent, t_ent))) return rc;
  if ((rc = upload(s, &s->tile.val, t_val))) return rc;
  s->tile.one = uni_one;
  s->tile.gdiag = nullptr;
  if (uni && (rc = upload(s, &s->tile.gdiag, gdiag))) return rc;
  s->tile_allocs.assign(s->allocs.begin() + first, s->allocs.end());
  s->tile.K = K;
  s->tile.n_tiles = (int)nt;
  s->tile_blocks = tblocks;
  s->tile_lpr = L;
  s->use_tile = true;
  s->nb_tile = blocks_for(nt, G, tblocks);   // (also set by alloc_factors; tiles built later — tiles_deferred — come through here)
  return SDPLR_OK;
}

// The band plan of the Lanczos SpMV (k_sparse.h, DevBand) from the host copy of the full pattern (CSC of the
// symmetric S: column j lists row j's neighbours).  NB bands of BW ≤ 16 384 columns (x band in LDS: ≤ 128 KB),
// NC row chunks of CH ≤ 4096 rows (partial t of the chunk in LDS: ≤ 32 KB), one 1024-thread block per (band,
// chunk), about one block per CU.  Hub rows are left to k_spmv_long.
// host half (no HIP calls): may run on a thread of its own while the first inner loops occupy the device
void band_host(S* s, BandHost& H) {
  H.ok = false;
  std::vector<int>& h_fcp = H.fcp;
  std::vector<int>& h_frv = H.frv;
  const int64_t n = s->n;
  // (SDPLR_HIP_LZBAND_MIN_N / _BW / _CH: test knobs — small instances with several narrow bands and chunks)
  int64_t min_n = 1 << 14;
  if (H.env_min_n > 0) min_n = H.env_min_n;
  if (!s->have_sparse || s->nnzS == 0 || n < min_n || H.env_no_band) return;
  int BWMAX = 16384, CHMAX = 4096;
  if (H.env_bw > 0) BWMAX = std::max(64, std::min(H.env_bw, 16384)) / 64 * 64;
  if (H.env_ch > 0) CHMAX = std::max(1, std::min(H.env_ch, 4096));
  const int NB = (int)((n + BWMAX - 1) / BWMAX);
  if (NB > 16) return;
  const int BW = (int)(((n + NB - 1) / NB + 63) / 64 * 64);
  // one resident round of blocks (256 CUs, one 1024-thread block each): the sweeping blocks, the hub-row blocks, the closer
  const int hub_blocks = s->sp.n_long_rows > 0 ? std::min((s->sp.n_long_rows + 3) / 4, 64) : 0;
  int NC = std::max<int>((int)((n + CHMAX - 1) / CHMAX), std::max(1, (256 - hub_blocks - (hub_blocks > 0 ? 1 : 0)) / NB));
  if (H.env_nc > 0) NC = std::max(NC, H.env_nc);
  // the kernel's static LDS (wave sums, the palette) comes out of the same 160 KB as the x band and the chunk's partial t:
  // more chunks (a shorter t) where a full band and a full chunk would not fit together (n = 2¹⁷, 2¹⁸ and just below)
  constexpr int64_t LZB_LDS_BUDGET = 160 * 1024 - (16 + 256) * 8 - 256;
  while (((int64_t)BW + (n + NC - 1) / NC) * 8 > LZB_LDS_BUDGET && (int64_t)NB * (NC + 1) <= 1024) NC++;
  if ((int64_t)NB * NC > 1024 || ((int64_t)BW + (n + NC - 1) / NC) * 8 > LZB_LDS_BUDGET) return;
  const int CH = (int)((n + NC - 1) / NC);
  const int thresh = s->sp.long_thresh;
  auto is_hub = [&](int64_t j) { return s->sp.n_long_rows > 0 && h_fcp[j + 1] - h_fcp[j] > thresh; };
  // PALETTE form (k_sparse.h, DevBand): on the structured path every off-diagonal entry of S is y_g·A_g[i,j]; with
  // ≤ 255 distinct off-diagonal values in A_g an entry shrinks from 12 bytes (2-byte column, 8-byte value + padding
  // share) to 3, and the per-run refill of the values disappears.  The plan is then built on the pattern WITHOUT its
  // diagonal (ocp/orv, codes oc); the diagonal rides band 0's partial from `sdiag`.
  bool pal = s->fast && !s->h_gptr.empty() && !H.env_no_pal;
  std::vector<double>& palv = H.palv;
  palv.assign(1, 0.0);
  std::vector<int> ocp, orv;
  std::vector<int>& diagpos = H.diagpos;
  std::vector<unsigned char> oc;
  if (pal) {   // distinct off-diagonal values of A_g, ascending; more than 255 ⇒ the value-array form
    std::vector<double> all;
    for (int64_t j = 0; j < n && pal; j++)
      for (int q = s->h_gptr[j]; q < s->h_gptr[j + 1]; q++) {
        if (s->h_gcol[q] == (int)j) continue;
        const double v = s->h_gval[q];
        auto it = std::lower_bound(all.begin(), all.end(), v);
        if (it != all.end() && *it == v) continue;
        if (!(v == v) || all.size() >= 255) { pal = false; break; }
        all.insert(it, v);
      }
    if (pal) palv.insert(palv.end(), all.begin(), all.end());
  }
  if (pal) {
    ocp.assign(n + 1, 0);
    diagpos.assign(n, -1);
    orv.reserve(h_frv.size());
    oc.reserve(h_frv.size());
    for (int64_t j = 0; j < n && pal; j++) {
      int q = s->h_gptr[j];   // A_g's row j and the full pattern's row j are both sorted by column
      for (int p = h_fcp[j]; p < h_fcp[j + 1]; p++) {
        const int col = h_frv[p];
        if (col == (int)j) {
          if (!is_hub(j)) diagpos[j] = p;
          continue;
        }
        while (q < s->h_gptr[j + 1] && s->h_gcol[q] < col) q++;
        if (q >= s->h_gptr[j + 1] || s->h_gcol[q] != col) { pal = false; break; }   // an off-diagonal entry A_g does not hold
        const int code = (int)(std::lower_bound(palv.begin() + 1, palv.end(), s->h_gval[q]) - palv.begin());
        orv.push_back(col);
        oc.push_back((unsigned char)code);
      }
      ocp[j + 1] = (int)orv.size();
    }
  }
  const std::vector<int>&cp = pal ? ocp : h_fcp, &rv = pal ? orv : h_frv;
  // entries of row j that fall into band b, in ascending column order: counted, then filled
  std::vector<int> cnt((size_t)NB * n, 0);
  for (int64_t j = 0; j < n; j++) {
    if (is_hub(j)) continue;
    for (int p = cp[j]; p < cp[j + 1]; p++) cnt[(size_t)(rv[p] / BW) * n + j]++;
  }
  std::vector<int>&blk_slot = H.blk_slot, &slot_g = H.slot_g;
  blk_slot.assign(1, 0);
  slot_g.assign(1, 0);
  std::vector<uint4>& cw = H.cw;
  std::vector<int4>& pos = H.pos;
  std::vector<int> rows;
  auto push_group = [&](const int* first, const int* rowsp, int nl, int k0, int64_t r0, int b) {
    for (int l = 0; l < 64; l++) {
      unsigned cc[4] = {0, 0, 0, 0}, code[4] = {0, 0, 0, 0};
      int pp[4] = {-1, -1, -1, -1};
      unsigned lrow = 0xFFFFu;
      if (l < nl) {
        const int j = rowsp[l];
        lrow = (unsigned)(j - r0);
        for (int k = 0; k < 4; k++)
          if (k0 + k < cnt[(size_t)b * n + j]) {
            const int p = first[l] + k0 + k;
            cc[k] = (unsigned)(rv[p] - b * BW);
            pp[k] = p;
            if (pal) code[k] = oc[p];
          }
      }
      if (pal) {
        cw.push_back(make_uint4(cc[0] | (cc[1] << 16), cc[2] | (cc[3] << 16), lrow | (code[0] << 16) | (code[1] << 24), code[2] | (code[3] << 8)));
      } else {
        cw.push_back(make_uint4(cc[0] | (cc[1] << 16), cc[2] | (cc[3] << 16), lrow, 0u));
        pos.push_back(make_int4(pp[0], pp[1], pp[2], pp[3]));
      }
    }
  };
  // a row's entries are sorted by column, so its entries inside band b start where those of bands < b end
  std::vector<int> band_first(n);
  for (int64_t j = 0; j < n; j++) band_first[j] = cp[j];
  {
    size_t total = 0;
    for (size_t t = 0; t < cnt.size(); t++) total += (size_t)(cnt[t] + 3) / 4;
    cw.reserve((total + 1) * 64 + 64 * (size_t)NB * NC);
    if (!pal) pos.reserve((total + 1) * 64 + 64 * (size_t)NB * NC);
  }
  for (int b = 0; b < NB; b++) {
    if (b > 0)
      for (int64_t j = 0; j < n; j++) band_first[j] += cnt[(size_t)(b - 1) * n + j];
    for (int c = 0; c < NC; c++) {
      const int64_t r0 = (int64_t)c * CH, r1 = std::min<int64_t>(r0 + CH, n);
      rows.clear();
      for (int64_t j = r0; j < r1; j++)
        if (cnt[(size_t)b * n + j] > 0) rows.push_back((int)j);
      std::stable_sort(rows.begin(), rows.end(), [&](int a, int bb) { return cnt[(size_t)b * n + a] > cnt[(size_t)b * n + bb]; });
      for (size_t q0 = 0; q0 < rows.size(); q0 += 64) {
        const int nl = (int)std::min<size_t>(64, rows.size() - q0);
        const int len = cnt[(size_t)b * n + rows[q0]];
        int first[64];   // first entry of each lane's row inside band b
        for (int l = 0; l < nl; l++) first[l] = band_first[rows[q0 + l]];
        for (int k0 = 0; k0 < len; k0 += 4) push_group(first, rows.data() + q0, nl, k0, r0, b);
        slot_g.push_back((int)(cw.size() / 64));
      }
      blk_slot.push_back((int)slot_g.size() - 1);
    }
  }
  const size_t n_groups = cw.size() / 64;
  if ((n_groups + 1) * 64 >= (size_t)1 << 31) return;
  push_group(nullptr, nullptr, 0, 0, 0, 0);   // the all-padding dummy group
  H.NB = NB; H.NC = NC; H.BW = BW; H.CH = CH; H.n_groups = n_groups; H.pal = pal;
  H.ok = true;
}

// device half: uploads + the LDS attribute; joins the host half first.  Called by whoever is about to use the plan.
int ensure_band(S* s) {
  if (!s->band_host) return SDPLR_OK;
  if (s->band_thread.joinable()) s->band_thread.join();
  std::unique_ptr<BandHost> Hp = std::move(s->band_host);
  BandHost& H = *Hp;
  s->use_band = false;
  if (!H.ok) return SDPLR_OK;
  const int64_t n = s->n;
  const int NB = H.NB, NC = H.NC, BW = H.BW, CH = H.CH;
  const bool pal = H.pal;
  std::vector<int>&blk_slot = H.blk_slot, &slot_g = H.slot_g, &diagpos = H.diagpos;
  std::vector<uint4>& cw = H.cw;
  std::vector<int4>& pos = H.pos;
  std::vector<double>& palv = H.palv;
  const size_t n_groups = H.n_groups;
  constexpr int64_t LZB_LDS_BUDGET = 160 * 1024 - (16 + 256) * 8 - 256;
  DevBand& bd = s->band;
  bd = DevBand{};
  bd.NB = NB; bd.NC = NC; bd.BW = BW; bd.CH = CH; bd.n_slots = (int)slot_g.size() - 1; bd.n_groups = (int)n_groups;
  int rc;
  if ((rc = upload(s, &bd.blk_slot, blk_slot))) return rc;
  if ((rc = upload(s, &bd.slot_g, slot_g))) return rc;
  if ((rc = upload(s, &bd.cw, cw))) return rc;
  if (pal) {
    palv.resize(256, 0.0);
    bd.pal_mode = 1;
    bd.gid_g = s->ff.gid_g;
    if ((rc = upload(s, &bd.pal, palv))) return rc;
    if ((rc = upload(s, &bd.diagpos, diagpos))) return rc;
    if ((rc = dzero(s, &bd.sdiag, (size_t)n))) return rc;
  } else {
    if ((rc = upload(s, &bd.pos, pos))) return rc;
    if ((rc = dalloc(s, &bd.vA, cw.size()))) return rc;
    if ((rc = dalloc(s, &bd.vB, cw.size()))) return rc;
    HIPCK(s, hipMemsetAsync(bd.vA, 0, cw.size() * sizeof(double2), s->stream));
    HIPCK(s, hipMemsetAsync(bd.vB, 0, cw.size() * sizeof(double2), s->stream));
  }
  if ((rc = dzero(s, &bd.tpart, (size_t)NB * n))) return rc;
  if ((rc = dzero(s, &bd.textra, (size_t)n))) return rc;
  // more than 64 KB of dynamic LDS has to be asked for: a per-function, PROCESS-wide attribute — set once, to the fixed
  // upper bound (a per-handle value would lower the cap under another handle's feet)
  {
    static std::mutex mu;
    static bool done = false;
    std::lock_guard<std::mutex> g(mu);
    if (!done) {
      HIPCK(s, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz_band<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZB_LDS_BUDGET));
      HIPCK(s, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz_band<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZB_LDS_BUDGET));
      done = true;
    }
  }
  s->use_band = true;
  return SDPLR_OK;
}

// A_g as the resident route's sliced ELL (k_resident.h, RsEll), from its symmetric CSR (rows sorted by column).  Only for
// instances the resident kernels can take at some rank: n < 2¹⁶ (16-bit columns) and a factor of rank 1 within the LDS
// budget.  Sets rs_ok.
int build_rs_ell(S* s, const std::vector<int>& g_ptr, const std::vector<int>& g_col, const std::vector<double>& g_val) {
  const int64_t n = s->n;
  if (n >= (1 << 16) || (size_t)4 * n * sizeof(double) > (size_t)150 * 1024 || getenv("SDPLR_HIP_NO_RESIDENT") != nullptr) return SDPLR_OK;
  std::vector<int> len(n, 0);
  std::vector<double> gdiag(n, 0.0);
  bool uniform = true, have_one = false;   // every off-diagonal entry the same value?
  double one = 0.0;
  for (int64_t j = 0; j < n; j++)
    for (int q = g_ptr[j]; q < g_ptr[j + 1]; q++) {
      if (g_col[q] == (int)j) { gdiag[j] += g_val[q]; continue; }
      len[j]++;
      if (!have_one) { one = g_val[q]; have_one = true; }
      else if (!(g_val[q] == one)) uniform = false;
    }
  // Rows in order of decreasing length, 64 to a slice — within each TEAM MEMBER's range of rows (k_resident.h, TEAM: member w of
  // W owns rows [w·⌈n/W⌉, (w+1)·⌈n/W⌉) in every phase, so its slices hold exactly those rows and the SpMM's output never
  // crosses members); one range, the whole matrix, without a team.  A slice's lanes past its range's end hold no row (−1).
  const int parts = rs_team_want(s);
  const int64_t rpm = (n + parts - 1) / parts;
  std::vector<int> perm, plen, sptr(1, 0);
  s->rs_ell_parts = parts;
  for (int w = 0; w < parts; w++) {
    const int64_t lo = std::min<int64_t>(n, w * rpm), hi = std::min<int64_t>(n, lo + rpm);
    s->rs_ell_part_sl[w] = (int)(perm.size() / 64);
    std::vector<int> order((size_t)(hi - lo));
    for (int64_t j = lo; j < hi; j++) order[j - lo] = (int)j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
    for (int v : order) { perm.push_back(v); plen.push_back(len[v]); }
    while (perm.size() % 64) { perm.push_back(-1); plen.push_back(0); }
  }
  for (int w = parts; w <= SDPLR_RS_TEAM_MAX; w++) s->rs_ell_part_sl[w] = (int)(perm.size() / 64);
  const int S_ = (int)(perm.size() / 64);
  for (int sl = 0; sl < S_; sl++) sptr.push_back(sptr[sl] + plen[(size_t)sl * 64]);   // the slice's first row is its longest
  s->rs_ell_pair_lines = 0;
  for (int sl = 0; sl < S_; sl++) s->rs_ell_pair_lines += (size_t)(sptr[sl + 1] - sptr[sl] + 1) / 2;
  std::vector<unsigned> ent((size_t)std::max(sptr[S_], 1) * 64, 0u);
  std::vector<double> val;
  if (!uniform) val.assign(ent.size(), 0.0);
  for (int sl = 0; sl < S_; sl++)
    for (int l = 0; l < 64; l++) {
      const int j = perm[(size_t)sl * 64 + l];
      if (j < 0) continue;
      int k = 0;
      for (int q = g_ptr[j]; q < g_ptr[j + 1]; q++) {
        if (g_col[q] == j) continue;
        const size_t at = ((size_t)sptr[sl] + k) * 64 + l;
        if (!uniform) val[at] = g_val[q];
        ent[at] = (unsigned)g_col[q];
        k++;
      }
    }
  RsEll& E = s->rs_ell;
  E = RsEll{};
  E.n_slices = S_;
  E.one = one;
  int rc;
  if ((rc = upload(s, &E.perm, perm))) return rc;
  if ((rc = upload(s, &E.len, plen))) return rc;
  if ((rc = upload(s, &E.sptr, sptr))) return rc;
  if ((rc = upload(s, &E.ent, ent))) return rc;
  if (!uniform && (rc = upload(s, &E.val, val))) return rc;
  if ((rc = upload(s, &E.gdiag, gdiag))) return rc;
  if (!s->rs_rowvec && (rc = dzero(s, &s->rs_rowvec, (size_t)3 * n))) return rc;   // (12.8 KB at n = 1600; used only when rs_rows_global)
  if (!s->rs_xch && (rc = dzero(s, &s->rs_xch, SDPLR_RS_XCH_DOUBLES))) return rc;
  s->rs_ok = true;
  return SDPLR_OK;
}

// lbfgs_update! rides k_fast_step2 (structured fast paths) for h ≤ 4 and 32-bit row offsets
bool step_fuses_update(const S* s) {
  return s->h >= 1 && s->h <= 4 && s->n * s->r * 8 < (1LL << 32) && !s->no_updfuse;
}
// k_fast_step2<…, PDROP>: the singleton fast path with A_g = the cost matrix (y_g ≡ 1), no low-rank matrices, the update
// fused and G_old read from the G array
bool step_can_drop_P(const S* s) {
  return s->fast && s->fast_singleton && step_fuses_update(s) && !s->dot_descent && s->ff.gid_g == (int)s->m && s->lr.ST == 0 &&
         !s->no_pdrop;
}

// the ring form of the history (k_dense.h) rides the P-less step kernel with the line-search head on the tile path: rows
// of one chunk, 16-byte pieces, an even number of elements, h ≤ 4
int tile_shape_lpr(const S* s);
bool ring_shape_common(const S* s) {
  return s->fast && s->fast_singleton && step_fuses_update(s) && !s->dot_descent && !s->no_ring && s->VEC == 2 && s->LPR >= 4 &&
         s->r <= (int64_t)s->LPR * s->VEC && (s->N & 1) == 0 && s->use_tile && s->tile_lpr == tile_shape_lpr(s) &&
         !s->tile_panels && s->HM == 4 && !s->lit;
}
// … the P-less step kernel with the line-search head (MaxCut-shaped instances)
bool ring_shape_ok(const S* s) {
  return ring_shape_common(s) && step_can_drop_P(s) && !s->no_lshead && s->n_extra == 1 && s->nb_tile <= 4 * SDPLR_NT;
}
// … the P-based one (a rank-one constraint — MinBisection — or a P-less loop that may not trust its G): the projections
// come out of the tile kernel, or there are none (k_lr_project reads dirt where it always is)
bool ring_shape_pb_ok(const S* s) {
  const bool lr_fused = s->lr.ST == 1 && s->r <= (int64_t)s->tile_lpr * s->VEC && !s->no_lrfuse;
  return ring_shape_common(s) && (s->lr.ST == 0 || lr_fused);
}

int alloc_factors(S* s) {
  s->N = s->n * s->r;
  long long stride = (s->N + 31) / 32 * 32;
  s->arena.stride = stride;
  s->arena.h = (int)s->h;
  const size_t used = (size_t)(3 + 2 * s->h + (s->fast ? 2 : 0));  // R, G, dirt, s_*, y_* [, P, W]
  double* base = nullptr;
  hipError_t e = pool_malloc((void**)&base, std::max<size_t>(used * stride, 1) * sizeof(double));
  if (e != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, std::string("hipMalloc(factors): ") + hipGetErrorString(e));
  HIPCK(s, hipMemsetAsync(base, 0, used * stride * sizeof(double), s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  s->arena.base = base;
  choose_shape(s);
  const int G = SDPLR_NT / s->LPR;
  s->nb_dense = blocks_for((s->N + 1) / 2, SDPLR_NT, 1024);
  if (const char* e = getenv("SDPLR_HIP_NB_DENSE")) s->nb_dense = std::max(1, std::min(atoi(e), SDPLR_MAXNB));
  s->nb_upd = std::min(s->nb_dense, 256);  // one block per CU streams at full rate; fewer Gram partials for the seam kernel
  if (const char* e = getenv("SDPLR_HIP_NB_UPD")) s->nb_upd = std::max(1, std::min(atoi(e), SDPLR_MAXNB));
  s->nb_sddmm = blocks_for(s->nnzT, G, 8192);
  s->nb_edge = blocks_for(s->nnzT, SDPLR_EDGE_POS * G, SDPLR_MAXNB);   // k_sddmm_edge: its per-block partials are folded by one block
  s->nb_spmm = blocks_for(s->n, G, 768);  // + up to 256 hub-row blocks share the partial slot
  if (const char* e = getenv("SDPLR_HIP_NB_SPMM")) s->nb_spmm = std::max(1, std::min(atoi(e), 768));
  s->nb_tile = blocks_for(s->tile.n_tiles, SDPLR_NT / std::max(1, s->tile_lpr > 0 ? s->tile_lpr : s->LPR), s->tile_blocks);   // one resident round; taller instances stride
  s->nb_lr = (s->N >= (1LL << 22)) ? 1024 : 256;   // low-rank projection grid (lr_part is sized for 1024)
  {
    // (SDPLR_HIP_NB_STEP: the fused kernel's grid — its 5h Gram sums are folded by ONE block, so more blocks lengthen the seam)
    static const int nb_step_cap = getenv("SDPLR_HIP_NB_STEP") ? std::max(64, std::min(atoi(getenv("SDPLR_HIP_NB_STEP")), 2048)) : 512;
    const int tr = std::min<int>(s->LPR, SDPLR_STEP_TR);
    s->nb_step = blocks_for((s->n + tr - 1) / tr, G, step_fuses_update(s) ? nb_step_cap : 1024);
  }  // one group per tile of LPR rows; its ‖G‖², ‖pv‖² (and Gram)
                                                                    // partials are folded by the one-block seam kernel: keep them few
  if (const char* e = getenv("SDPLR_HIP_NB_STEP")) s->nb_step = std::max(1, std::min(atoi(e), SDPLR_MAXNB));
  return SDPLR_OK;
}

}  // namespace

// ================================================================================================
// device management / construction
// ================================================================================================
extern "C" {

const char* sdplr_hip_version(void) { return "sdplr_hip 0.1 (gfx950, FP64)"; }
const char* sdplr_hip_last_error(const sdplr_hip_solver* s) { return s ? s->err.c_str() : g_err.c_str(); }

int32_t sdplr_hip_device_synchronize(void) {
  ApiShared api_guard;
  if (have_device() <= 0) return fail(nullptr, SDPLR_ERR_NO_DEVICE, "no HIP device");
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(nullptr, SDPLR_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
  return SDPLR_OK;
}
int32_t sdplr_hip_device_count(int32_t* count) {
  ApiShared api_guard;
  if (!count) return SDPLR_ERR_INVALID_ARG;
  *count = have_device();
  return SDPLR_OK;
}
int32_t sdplr_hip_set_device(int32_t device) {
  ApiShared api_guard;
  if (have_device() <= 0) return fail(nullptr, SDPLR_ERR_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= have_device()) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "set_device: no such device");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // (the refusal must not surface as the next call's "last error")
    return fail(nullptr, SDPLR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  g_device.store(device, std::memory_order_relaxed);   // sticky: every later call without a handle, from any thread, binds to it
  return SDPLR_OK;
}

// Primes the library's per-device pools for `n_handles` solver handles alive at the same time: HIP streams (creating
// one costs milliseconds — tens of them when 16 threads ask at once), events, the pinned blocks of the control-block
// shadows and of finalize's staging.  Optional; part of bringing the device context up, like the first HIP call.
int32_t sdplr_hip_warmup(int32_t n_handles) {
  ApiShared api_guard;
  if (have_device() <= 0) return fail(nullptr, SDPLR_ERR_NO_DEVICE, "no HIP device");
  if (n_handles < 1 || n_handles > 256) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "warmup: 1 ≤ n_handles ≤ 256");
  std::vector<hipStream_t> st(n_handles, nullptr);
  std::vector<hipEvent_t> ev(2 * (size_t)n_handles, nullptr);
  std::vector<void*> pc(3 * (size_t)n_handles, nullptr), ch(std::min(n_handles, 32), nullptr);
  hipError_t e = hipSuccess;
  for (auto& x : st) if (e == hipSuccess) e = pool_stream(&x);
  for (auto& x : ev) if (e == hipSuccess) e = pool_event(&x);
  for (auto& x : pc) if (e == hipSuccess) e = pool_host_ctrl(&x);
  for (auto& x : ch) if (e == hipSuccess) e = pool_host_chunk(&x);
  for (auto x : st) pool_stream_free(x);
  for (auto x : ev) pool_event_free(x);
  for (auto x : pc) pool_host_ctrl_free(x);
  for (auto x : ch) pool_host_chunk_free(x);
  if (e != hipSuccess) return fail(nullptr, SDPLR_ERR_HIP, std::string("warmup: ") + hipGetErrorString(e));
  return SDPLR_OK;
}

// Returns what the library's pools cache and no handle uses to the runtime: device blocks, streams, events, pinned blocks.
int32_t sdplr_hip_trim_pools(void) {
  ApiShared api_guard;
  if (have_device() <= 0) return fail(nullptr, SDPLR_ERR_NO_DEVICE, "no HIP device");
  pool_trim();
  return SDPLR_OK;
}

int32_t sdplr_hip_create(int64_t n, int64_t m, int64_t r, int64_t h, sdplr_hip_solver** out) {
  ApiShared api_guard;
  if (!out) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "create: null out");
  *out = nullptr;
  if (n < 1 || m < 0 || r < 1 || h < 0) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "create: bad sizes");
  if (h > 4096) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "create: numlbfgsvecs > 4096");   // (> 16: the literal two-loop route, k_dense.h)
  if (n >= (1LL << 31) - 1 || m >= (1LL << 31) - 2 || n * r >= (1LL << 40))
    return fail(nullptr, SDPLR_ERR_INVALID_ARG, "create: sizes exceed the int32 index range of the device layout");
  if (have_device() <= 0) return fail(nullptr, SDPLR_ERR_NO_DEVICE, "no usable HIP device (libsdplr_hip has no CPU fallback)");
  S* s = new S();
  s->n = n; s->m = m; s->r = r; s->h = h;
  if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;   // (the guard above has bound this thread to the sticky device)
  *out = s;
  return SDPLR_OK;
}

int32_t sdplr_hip_set_sparse(S* s, int64_t base, int64_t n_sparse, const int64_t* matptr,
                             const int64_t* nzind, const double* one, const double* two,
                             const int64_t* gids, int64_t nnzT, const int64_t* tcp, const int64_t* trv,
                             int64_t nnzS, const int64_t* fcp, const int64_t* frv, const int64_t* mapped) {
  ApiShared api_guard(dev_of(s));
  if (!s) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");
  if (s->finalized || s->have_sparse) return fail(s, SDPLR_ERR_STATE, "set_sparse: already set / finalized");
  if (base != 0 && base != 1) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: index_base must be 0 or 1");
  if (n_sparse < 0 || nnzT < 0 || nnzS < 0 || !matptr || !tcp || !fcp)
    return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: bad sizes / null arrays");
  const int64_t n = s->n, lim = (1LL << 31) - 2;
  if (nnzT > lim || nnzS > lim || n_sparse > lim) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: nnz exceeds int32");
  const int64_t nnzAgg = matptr[n_sparse] - base;
  if (matptr[0] - base != 0 || nnzAgg < 0 || nnzAgg > lim) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: bad matptr");
  s->h_matptr.resize(n_sparse + 1);
  for (int64_t k = 0; k <= n_sparse; k++) {
    s->h_matptr[k] = (int)(matptr[k] - base);
    if (k > 0 && s->h_matptr[k] < s->h_matptr[k - 1]) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: matptr not monotone");
  }
  s->h_nzind.resize(nnzAgg); s->h_one.resize(nnzAgg); s->h_two.resize(nnzAgg);
  for (int64_t e = 0; e < nnzAgg; e++) {
    const int64_t q = nzind[e] - base;
    if (q < 0 || q >= nnzT) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: nzind out of range");
    s->h_nzind[e] = (int)q; s->h_one[e] = one[e]; s->h_two[e] = two[e];
  }
  s->h_gids.resize(n_sparse);
  for (int64_t k = 0; k < n_sparse; k++) {
    const int64_t g = gids[k] - base;
    if (g < 0 || g > s->m) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: global index out of range");
    s->h_gids[k] = (int)g;
  }
  s->h_tcp.resize(n + 1); s->h_fcp.resize(n + 1);
  for (int64_t j = 0; j <= n; j++) {
    s->h_tcp[j] = (int)(tcp[j] - base); s->h_fcp[j] = (int)(fcp[j] - base);
    if (j > 0 && (s->h_tcp[j] < s->h_tcp[j - 1] || s->h_fcp[j] < s->h_fcp[j - 1]))
      return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: colptr not monotone");
  }
  if (s->h_tcp[0] != 0 || s->h_fcp[0] != 0 || s->h_tcp[n] != nnzT || s->h_fcp[n] != nnzS)
    return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: colptr/nnz mismatch");
  s->h_trv.resize(nnzT);
  for (int64_t p = 0; p < nnzT; p++) {
    const int64_t i = trv[p] - base;
    if (i < 0 || i >= n) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: triu rowval out of range");
    s->h_trv[p] = (int)i;
  }
  s->h_frv.resize(nnzS); s->h_mapped.resize(nnzS);
  for (int64_t p = 0; p < nnzS; p++) {
    const int64_t i = frv[p] - base, q = mapped[p] - base;
    if (i < 0 || i >= n || q < 0 || q >= nnzT) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse: full pattern out of range");
    s->h_frv[p] = (int)i; s->h_mapped[p] = (int)q;
  }
  s->n_sparse = n_sparse; s->nnzT = nnzT; s->nnzS = nnzS; s->nnzAgg = nnzAgg;
  s->have_sparse = true;
  return SDPLR_OK;
}

// preprocess_sparsecons (src/preprocess.jl:24-169) inside the library: the same aggregated layout set_sparse takes,
// built from the matrices themselves — counting passes and per-column sorts of short lists instead of the reference's
// per-matrix walk, a binary search only where the reference has one (:110-119, :143-156).
int32_t sdplr_hip_set_sparse_coo(S* s, int64_t base, int64_t n_sparse, const int64_t* ent_ptr, const int64_t* I,
                                 const int64_t* J, const double* V, const int64_t* gids) {
  ApiShared api_guard(dev_of(s));
  if (!s) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");
  if (s->finalized || s->have_sparse) return fail(s, SDPLR_ERR_STATE, "set_sparse_coo: already set / finalized");
  if (base != 0 && base != 1) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: index_base must be 0 or 1");
  if (n_sparse < 0 || !ent_ptr || (n_sparse > 0 && !gids)) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: bad sizes / null arrays");
  const int64_t n = s->n, lim = (1LL << 31) - 2;
  const int64_t E = ent_ptr[n_sparse] - base;
  if (ent_ptr[0] - base != 0 || E < 0 || E > lim || n_sparse > lim) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: bad ent_ptr");
  if (E > 0 && (!I || !J || !V)) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: null entry arrays");
  for (int64_t k = 0; k < n_sparse; k++) {
    if (ent_ptr[k + 1] < ent_ptr[k]) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: ent_ptr not monotone");
    const int64_t g = gids[k] - base;
    if (g < 0 || g > s->m) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: global index out of range");
  }
  // ---- the aggregated full pattern sparse(all_I, all_J, ones) (:87): entries bucketed by column, each column's rows
  // sorted and made unique; its upper triangle (:90) is a prefix-free filter of the same lists ----
  std::vector<int> cnt(n + 1, 0);
  int64_t n_upper = 0;
  for (int64_t e = 0; e < E; e++) {
    const int64_t i = I[e] - base, j = J[e] - base;
    if (i < 0 || i >= n || j < 0 || j >= n) return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: constraint entry outside the n×n matrix");
    cnt[j + 1]++;
    n_upper += (i <= j);
  }
  for (int64_t j = 0; j < n; j++) cnt[j + 1] += cnt[j];
  std::vector<int> rows(E);
  {
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int64_t e = 0; e < E; e++) rows[fill[J[e] - base]++] = (int)(I[e] - base);
  }
  std::vector<int>&fcp = s->h_fcp, &frv = s->h_frv, &tcp = s->h_tcp, &trv = s->h_trv;
  fcp.assign(n + 1, 0);
  tcp.assign(n + 1, 0);
  // (columns are independent: sorted and made unique on a few host threads, compacted after a prefix sum)
  parallel_for(n, 4096, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; j++) {
      int* b = rows.data() + cnt[j];
      int* e = rows.data() + cnt[j + 1];
      if (!std::is_sorted(b, e)) std::sort(b, e);
      int* u = std::unique(b, e);
      fcp[j + 1] = (int)(u - b);
      tcp[j + 1] = (int)(std::upper_bound(b, u, (int)j) - b);   // rows ≤ j: the column's part of the upper triangle
    }
  });
  for (int64_t j = 0; j < n; j++) { fcp[j + 1] += fcp[j]; tcp[j + 1] += tcp[j]; }
  frv.resize(fcp[n]);
  trv.resize(tcp[n]);
  parallel_for(n, 4096, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; j++) {
      const int* b = rows.data() + cnt[j];
      std::copy(b, b + (fcp[j + 1] - fcp[j]), frv.begin() + fcp[j]);
      std::copy(b, b + (tcp[j + 1] - tcp[j]), trv.begin() + tcp[j]);
    }
  });
  const int64_t nnzS = (int64_t)frv.size(), nnzT = (int64_t)trv.size();
  auto find_triu = [&](int col, int row) -> int {   // position of (row, col) in the triu pattern, −1 if absent
    const int* b = trv.data() + tcp[col];
    const int* e = trv.data() + tcp[col + 1];
    const int* p = std::lower_bound(b, e, row);
    return (p < e && *p == row) ? (int)(p - trv.data()) : -1;
  };
  // ---- per-matrix segments over the upper-triangular entries (:97-132) ----
  s->h_matptr.assign(n_sparse + 1, 0);
  s->h_nzind.resize(n_upper); s->h_one.resize(n_upper); s->h_two.resize(n_upper);
  s->h_gids.resize(n_sparse);
  {
    // (the per-matrix segments are disjoint ranges of the output: counted, then filled on a few host threads — cut at
    // equal numbers of ENTRIES, so that one multi-entry matrix among 1e5 singletons does not land on one thread)
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(host_threads(), E / 65536));
    std::vector<int64_t> kcut(nt + 1, n_sparse);
    kcut[0] = 0;
    for (int64_t t = 1; t < nt; t++) {
      const int64_t want = E * t / nt + base;
      kcut[t] = std::lower_bound(ent_ptr, ent_ptr + n_sparse + 1, want) - ent_ptr;
      kcut[t] = std::min<int64_t>(std::max<int64_t>(kcut[t], kcut[t - 1]), n_sparse);
    }
    std::vector<int64_t> ucnt(nt, 0);
    auto run = [&](int64_t t, bool fill, int64_t start) {
      int64_t cum = start;
      for (int64_t k = kcut[t]; k < kcut[t + 1]; k++) {
        if (fill) { s->h_matptr[k] = (int)cum; s->h_gids[k] = (int)(gids[k] - base); }
        // (a matrix's entries usually come column by column with ascending rows — findnz order — and so do the pattern's:
        // the position is then found by walking a cursor along the column instead of a binary search per entry)
        int pcol = -1, prow = -1, cur = 0;
        const bool walk = ent_ptr[k + 1] - ent_ptr[k] > 4;       // (a singleton's diagonal entry is the LAST of its column: searched)
        for (int64_t e = ent_ptr[k] - base; e < ent_ptr[k + 1] - base; e++) {
          const int i = (int)(I[e] - base), j = (int)(J[e] - base);
          if (i > j) continue;                                   // triu keeps i ≤ j (:9)
          if (fill) {
            int pos;
            if (j != pcol) { pcol = j; prow = -1; cur = tcp[j]; }
            if (!walk || i < prow) {
              pos = find_triu(j, i);
            } else {
              const int end = tcp[j + 1];
              while (cur < end && trv[cur] < i) cur++;
              pos = (cur < end && trv[cur] == i) ? cur : -1;
              prow = i;
            }
            s->h_nzind[cum] = pos;
            s->h_one[cum] = V[e];
            s->h_two[cum] = (i == j) ? V[e] : 2.0 * V[e];         // off-diagonal entries count twice (:121-128)
          }
          cum++;
        }
      }
      return cum - start;
    };
    std::vector<int64_t> ustart(nt + 1, 0);
    if (nt == 1) {
      ustart[1] = run(0, true, 0);
    } else {
      {
        std::vector<std::thread> th;
        for (int64_t t = 0; t < nt; t++) th.emplace_back([&, t] { ucnt[t] = run(t, false, 0); });
        for (auto& x : th) x.join();
      }
      for (int64_t t = 0; t < nt; t++) ustart[t + 1] = ustart[t] + ucnt[t];
      std::vector<std::thread> th;
      for (int64_t t = 0; t < nt; t++) th.emplace_back([&, t] { (void)run(t, true, ustart[t]); });
      for (auto& x : th) x.join();
    }
    s->h_matptr[n_sparse] = (int)ustart[nt];
  }
  // ---- full pattern → position of (min, max) in the upper-triangular pattern (:135-156): the upper entries of column j
  // ARE column j of the triu pattern, in order; a lower entry (i > j) is searched in column i ----
  s->h_mapped.resize(nnzS);
  std::atomic<int> missing{0};
  if (n <= 4096) {
    // (one chunk: the lower entries of successive columns j ask column i of the triu pattern for ascending rows j — a cursor
    // per column that only moves forward, instead of a binary search per entry)
    std::vector<int> cursor(tcp.begin(), tcp.end() - 1);
    for (int64_t j = 0; j < n; j++) {
      int up = tcp[j];
      for (int p = fcp[j]; p < fcp[j + 1]; p++) {
        const int i = frv[p];
        if (i <= (int)j) { s->h_mapped[p] = up++; continue; }
        int& c = cursor[i];
        const int end = tcp[i + 1];
        while (c < end && trv[c] < (int)j) c++;
        const int q = (c < end && trv[c] == (int)j) ? c : -1;
        if (q < 0) missing.store(1, std::memory_order_relaxed);
        s->h_mapped[p] = q;
      }
    }
  } else
  parallel_for(n, 4096, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; j++) {
      int up = tcp[j];
      for (int p = fcp[j]; p < fcp[j + 1]; p++) {
        const int i = frv[p];
        if (i <= (int)j) { s->h_mapped[p] = up++; continue; }
        const int q = find_triu(i, (int)j);
        if (q < 0) missing.store(1, std::memory_order_relaxed);
        s->h_mapped[p] = q;
      }
    }
  });
  if (missing.load()) {
    s->h_matptr.clear(); s->h_nzind.clear(); s->h_one.clear(); s->h_two.clear(); s->h_gids.clear();
    fcp.clear(); frv.clear(); tcp.clear(); trv.clear(); s->h_mapped.clear();
    return fail(s, SDPLR_ERR_INVALID_ARG, "set_sparse_coo: a constraint matrix is not symmetric: a lower-triangular entry has no "
                                          "upper-triangular mirror in the aggregated pattern");
  }
  s->n_sparse = n_sparse; s->nnzT = nnzT; s->nnzS = nnzS; s->nnzAgg = n_upper;
  s->have_sparse = true;
  return SDPLR_OK;
}

// The aggregated layout as the library holds it between set_sparse[_coo] and finalize (0-based), for callers that want
// to keep it (SolverAuxiliary's fields, src/structs.jl:278-288) and for the parity tests of the native preprocessing.
// which: 0 matptr [n_sparse+1], 1 nzind [nnzAgg], 2 global_inds [n_sparse], 3 triu_colptr [n+1], 4 triu_rowval [nnzT],
// 5 full_colptr [n+1], 6 full_rowval [nnzS], 7 mappedto_triu [nnzS] → out_i; 8 nzval_one, 9 nzval_two [nnzAgg] → out_f.
int32_t sdplr_hip_get_layout(const S* s, int32_t which, int64_t* out_i, double* out_f, int64_t cap, int64_t* len) {
  if (!s) return SDPLR_ERR_INVALID_ARG;
  if (!s->have_sparse || s->finalized) return SDPLR_ERR_STATE;
  const std::vector<int>* vi = nullptr;
  const std::vector<double>* vf = nullptr;
  switch (which) {
    case 0: vi = &s->h_matptr; break;
    case 1: vi = &s->h_nzind; break;
    case 2: vi = &s->h_gids; break;
    case 3: vi = &s->h_tcp; break;
    case 4: vi = &s->h_trv; break;
    case 5: vi = &s->h_fcp; break;
    case 6: vi = &s->h_frv; break;
    case 7: vi = &s->h_mapped; break;
    case 8: vf = &s->h_one; break;
    case 9: vf = &s->h_two; break;
    default: return SDPLR_ERR_INVALID_ARG;
  }
  const int64_t L = vi ? (int64_t)vi->size() : (int64_t)vf->size();
  if (len) *len = L;
  if (vi && out_i) for (int64_t k = 0; k < std::min(L, cap); k++) out_i[k] = (*vi)[k];
  if (vf && out_f) for (int64_t k = 0; k < std::min(L, cap); k++) out_f[k] = (*vf)[k];
  return SDPLR_OK;
}

int32_t sdplr_hip_add_symlowrank(S* s, int64_t base, int64_t gid, int64_t sc, const double* B, const double* D) {
  ApiShared api_guard(dev_of(s));
  if (!s) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");
  if (s->finalized) return fail(s, SDPLR_ERR_STATE, "add_symlowrank: already finalized");
  if (sc < 1 || !B || !D || gid - base < 0 || gid - base > s->m) return fail(s, SDPLR_ERR_INVALID_ARG, "add_symlowrank: bad args");
  LowRankHost L;
  L.gid = gid - base; L.s = sc;
  L.B.assign(B, B + s->n * sc);
  L.D.assign(D, D + sc);
  s->h_lr.push_back(std::move(L));
  return SDPLR_OK;
}

int32_t sdplr_hip_finalize(S* s) {
  ApiShared api_guard(dev_of(s));
  if (!s) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");
  if (s->finalized) return fail(s, SDPLR_ERR_STATE, "finalize: already finalized");
  const int64_t n = s->n, m = s->m;
  int rc;
  s->up.on = getenv("SDPLR_HIP_NO_UPLOAD_ARENA") == nullptr;
  // SDPLR_HIP_TIMING=1: where the set-up time goes (stderr)
  const bool timing = getenv("SDPLR_HIP_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  HIPCK(s, pool_stream(&s->stream));
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[sdplr_hip finalize] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  lap("stream");
  // ---- sparse layout ----
  DevSparse& sp = s->sp;
  sp.n = (int)n; sp.nnzT = (int)s->nnzT; sp.nnzS = (int)s->nnzS; sp.nnzAgg = (int)s->nnzAgg; sp.n_sparse = (int)s->n_sparse;
  if (!s->have_sparse) { s->h_tcp.assign(n + 1, 0); s->h_fcp.assign(n + 1, 0); s->h_matptr.assign(1, 0); }
  std::vector<int> colidx(s->nnzT);
  for (int64_t j = 0; j < n; j++)
    for (int p = s->h_tcp[j]; p < s->h_tcp[j + 1]; p++) colidx[p] = (int)j;
  // transpose of the per-matrix segments: triu position → list of (y index, nzval_one), ascending matrix
  std::vector<int> tptr(s->nnzT + 1, 0), tmat(s->nnzAgg);
  std::vector<double> tval(s->nnzAgg);
  for (int64_t e = 0; e < s->nnzAgg; e++) tptr[s->h_nzind[e] + 1]++;
  for (int64_t q = 0; q < s->nnzT; q++) tptr[q + 1] += tptr[q];
  {
    std::vector<int> fill(tptr.begin(), tptr.end() - 1);
    for (int64_t k = 0; k < s->n_sparse; k++)
      for (int e = s->h_matptr[k]; e < s->h_matptr[k + 1]; e++) {
        const int pos = fill[s->h_nzind[e]]++;
        tmat[pos] = s->h_gids[k];
        tval[pos] = s->h_one[e];
      }
  }
  // segmented-reduction plan
  const int SHORT_MAX = 16, CHUNK = 2048;
  std::vector<int> short_ids, chunk_beg, chunk_end, long_ids, long_chunk_ptr(1, 0);
  for (int64_t k = 0; k < s->n_sparse; k++) {
    const int len = s->h_matptr[k + 1] - s->h_matptr[k];
    if (len <= SHORT_MAX) { short_ids.push_back((int)k); continue; }
    long_ids.push_back((int)k);
    for (int bgn = s->h_matptr[k]; bgn < s->h_matptr[k + 1]; bgn += CHUNK) {
      chunk_beg.push_back(bgn);
      chunk_end.push_back(std::min(bgn + CHUNK, s->h_matptr[k + 1]));
    }
    long_chunk_ptr.push_back((int)chunk_beg.size());
  }
  sp.n_short = (int)short_ids.size(); sp.n_chunks = (int)chunk_beg.size(); sp.n_long = (int)long_ids.size();
  sp.n_short_blocks = (sp.n_short + SDPLR_NT - 1) / SDPLR_NT;
  {
    std::vector<char> cov(m + 1, 0);
    for (int g : s->h_gids) cov[g] = 1;
    for (auto& L : s->h_lr) cov[L.gid] = 1;
    s->all_covered = std::all_of(cov.begin(), cov.end(), [](char c) { return c != 0; });
  }
  lap("transpose + reduction plan");
#define UP(field, vec) if ((rc = upload(s, &sp.field, vec))) return rc
  UP(triu_colptr, s->h_tcp); UP(triu_rowval, s->h_trv); UP(triu_colidx, colidx);
  UP(colptr, s->h_fcp); UP(rowval, s->h_frv); UP(mapped, s->h_mapped);
  UP(matptr, s->h_matptr); UP(nzind, s->h_nzind); UP(gids, s->h_gids);
  UP(nzval_one, s->h_one); UP(nzval_two, s->h_two);
  UP(tptr, tptr); UP(tmat, tmat); UP(tval, tval);
  UP(short_ids, short_ids); UP(chunk_beg, chunk_beg); UP(chunk_end, chunk_end);
  UP(long_ids, long_ids); UP(long_chunk_ptr, long_chunk_ptr);
#undef UP
  // hub rows (power-law graphs): rows of the full pattern far longer than the rest get a block each
  auto plan_long = [&](const std::vector<int>& ptr, DevSparse& dst) -> int {
    std::vector<int> deg(n);
    for (int64_t j = 0; j < n; j++) deg[j] = ptr[j + 1] - ptr[j];
    // a hub is a row far longer than the rest: more than 64 nonzeros AND more than 3× the mean (the G(n,p)
    // configurations have none; nor do the dense-ish Gset graphs of the batch configuration — mean degree 48, longest
    // row 67 — which a fixed threshold of 64 pushed off the singleton fast path and onto a third Lanczos kernel)
    const long long mean3 = n > 0 ? 3LL * ptr[n] / n : 0;
    int thresh = (int)std::min<long long>(std::max<long long>(64, mean3), 1 << 30);
    if (const char* e = getenv("SDPLR_HIP_HUB_THRESH")) thresh = std::max(8, atoi(e));
    std::vector<int> rows;
    for (int64_t j = 0; j < n; j++)
      if (deg[j] > thresh) rows.push_back((int)j);
    dst.n_long_rows = (int)rows.size();
    dst.long_thresh = thresh;
    return upload(s, &dst.long_rows, rows);
  };
  lap("uploads (layout)");
  if ((rc = plan_long(s->h_fcp, sp))) return rc;
  if ((rc = dzero(s, &sp.nzval, s->nnzS))) return rc;
  if ((rc = dzero(s, &sp.triu_nzval, s->nnzT))) return rc;
  if ((rc = dzero(s, &sp.UVt0, s->nnzT))) return rc;
  if ((rc = dzero(s, &sp.UVt1, s->nnzT))) return rc;
  if ((rc = dzero(s, &sp.chunk_partial0, sp.n_chunks))) return rc;
  if ((rc = dzero(s, &sp.chunk_partial1, sp.n_chunks))) return rc;
  // ---- low-rank matrices ----
  DevLowRank& lr = s->lr;
  lr.n_lr = (int)s->h_lr.size();
  {
    std::vector<double> Bcat, Dcat;
    std::vector<int> col_gid, mat_ptr(1, 0), mat_gid;
    for (auto& L : s->h_lr) {
      Bcat.insert(Bcat.end(), L.B.begin(), L.B.end());
      Dcat.insert(Dcat.end(), L.D.begin(), L.D.end());
      for (int64_t t = 0; t < L.s; t++) col_gid.push_back((int)L.gid);
      mat_ptr.push_back((int)col_gid.size());
      mat_gid.push_back((int)L.gid);
    }
    lr.ST = (int)col_gid.size();
    if ((rc = upload(s, &lr.Bcat, Bcat))) return rc;
    if ((rc = upload(s, &lr.Dcat, Dcat))) return rc;
    if ((rc = upload(s, &lr.col_gid, col_gid))) return rc;
    if ((rc = upload(s, &lr.mat_ptr, mat_ptr))) return rc;
    if ((rc = upload(s, &lr.mat_gid, mat_gid))) return rc;
  }
  // ---- vectors ----
  if ((rc = dzero(s, &s->lambda, m))) return rc;
  if ((rc = dzero(s, &s->b, m))) return rc;
  if ((rc = dzero(s, &s->pv, m))) return rc;
  if ((rc = dzero(s, &s->y, m + 1))) return rc;
  if ((rc = dzero(s, &s->pv_raw, m + 1))) return rc;
  if ((rc = dzero(s, &s->A_RD, m + 1))) return rc;
  if ((rc = dzero(s, &s->A_DD, m + 1))) return rc;
  {  // all-equality defaults (src/structs.jl:266-268, :247)
    std::vector<double> inf(m, std::numeric_limits<double>::infinity()), ninf(m, -std::numeric_limits<double>::infinity());
    const double *ub = nullptr, *lb = nullptr;
    if ((rc = upload(s, &ub, inf))) return rc;
    if ((rc = upload(s, &lb, ninf))) return rc;
    s->lambda_ub = const_cast<double*>(ub);
    s->pv_lb = const_cast<double*>(lb);
  }
  if ((rc = dzero(s, &s->partials, (size_t)SDPLR_NSLOT * SDPLR_MAXNB))) return rc;
  for (int k = 0; k < 3; k++)
    if ((rc = dzero(s, &s->lz_buf[k], n + 64))) return rc;   // (k_lz_band reads a clamped 16-byte pair at a ragged band end)
  if ((rc = dzero(s, &s->lz_v0, n))) return rc;
  s->nb_lr = 256;   // (alloc_factors raises it to 1024 for large factors: n·r ≥ 4M)
  if ((rc = dzero(s, &s->lr_btx_part, (size_t)std::max(lr.ST, 1) * SDPLR_MAXNB))) return rc;
  if ((rc = dzero(s, &s->lr_coef, 2 * (size_t)std::max(lr.ST, 1)))) return rc;
  if ((rc = dzero(s, &s->red10, 16))) return rc;
  {
    DevCtrl* d = nullptr;
    if ((rc = dalloc(s, &d, 1))) return rc;
    s->ctrl = d;
    HIPCK(s, pool_host_ctrl((void**)&s->hc));
    memset(s->hc, 0, sizeof(DevCtrl));
    s->hc->sigma = 2.0;             // config.σ_0 default, src/options.jl:5
    s->hc->alpha_max = 1.0;
    s->hc->latest = (int)s->h;      // src/lbfgs.jl:45
    HIPCK(s, hipMemcpyAsync(s->ctrl, s->hc, sizeof(DevCtrl), hipMemcpyHostToDevice, s->stream));
    HIPCK(s, hipStreamSynchronize(s->stream));
    for (int k = 0; k < 2; k++) {
      HIPCK(s, pool_host_ctrl((void**)&s->snap[k]));
      HIPCK(s, pool_event(&s->snap_ev[k]));
    }
  }
  lap("vectors + control block");
  // ---- structured fast path: classify the sparse matrices ----
  s->dot_descent = getenv("SDPLR_HIP_DOT_DESCENT") != nullptr;
  s->lit = s->h > SDPLR_HMAX;
  if (s->lit) {
    s->dot_descent = true;      // no Gram data: ⟨dir, G⟩ by reduction (k_descent)
    if ((rc = dzero(s, &s->lit_rho, (size_t)s->h))) return rc;
    if ((rc = dzero(s, &s->lit_a, (size_t)s->h))) return rc;
  }
  if (const char* e = getenv("SDPLR_HIP_GRAPH_ITERS")) s->graph_iters = std::max(1, std::min(atoi(e), 64));
  s->no_lrfuse = getenv("SDPLR_HIP_NO_LRFUSE") != nullptr;
  s->force_graph = getenv("SDPLR_HIP_FORCE_GRAPH") != nullptr;
  s->no_updfuse = getenv("SDPLR_HIP_NO_UPDFUSE") != nullptr;
  s->no_pdrop = getenv("SDPLR_HIP_NO_PDROP") != nullptr;
  s->no_lshead = getenv("SDPLR_HIP_NO_LSHEAD") != nullptr;
  s->no_ring = getenv("SDPLR_HIP_NO_RING") != nullptr;
  s->tile_panels = getenv("SDPLR_HIP_TILE_PANELS") != nullptr;
  if (s->have_sparse && getenv("SDPLR_HIP_NO_FAST") == nullptr) {
    std::vector<int> general;
    for (int64_t k = 0; k < s->n_sparse; k++) {
      bool diag_only = true;
      for (int e = s->h_matptr[k]; e < s->h_matptr[k + 1] && diag_only; e++)
        diag_only = (s->h_trv[s->h_nzind[e]] == colidx[s->h_nzind[e]]);
      if (!diag_only) general.push_back((int)k);
    }
    if (general.size() == 1) {
      const int kg = general[0];
      // A_g as a symmetric CSR (mirror the upper-triangular entries, merge duplicates)
      std::vector<int> g_ptr(n + 1, 0), g_col;
      std::vector<double> g_val;
      const int eb = s->h_matptr[kg], ee = s->h_matptr[kg + 1];
      bool ascending = true;   // positions of the (column-major) triu pattern, strictly ascending: the usual case
      for (int e = eb + 1; e < ee && ascending; e++) ascending = s->h_nzind[e] > s->h_nzind[e - 1];
      if (ascending) {
        // two counting passes: row j first receives its mirrored entries (i < j, ascending i as column j is
        // walked), then row i its own upper entries (j ≥ i, ascending j) — every row comes out sorted by column
        for (int e = eb; e < ee; e++) {
          const int q = s->h_nzind[e], i = s->h_trv[q], j = colidx[q];
          g_ptr[i + 1]++;
          if (i != j) g_ptr[j + 1]++;
        }
        for (int64_t i = 0; i < n; i++) g_ptr[i + 1] += g_ptr[i];
        g_col.resize(g_ptr[n]);
        g_val.resize(g_ptr[n]);
        std::vector<int> fillg(g_ptr.begin(), g_ptr.end() - 1);
        for (int e = eb; e < ee; e++) {
          const int q = s->h_nzind[e], i = s->h_trv[q], j = colidx[q];
          if (i != j) { const int pos = fillg[j]++; g_col[pos] = i; g_val[pos] = s->h_one[e]; }
        }
        for (int e = eb; e < ee; e++) {
          const int q = s->h_nzind[e], i = s->h_trv[q], j = colidx[q];
          const int pos = fillg[i]++; g_col[pos] = j; g_val[pos] = s->h_one[e];
        }
      } else {
        struct Ent { int i, j; double v; };
        std::vector<Ent> ents;
        for (int e = eb; e < ee; e++) {
          const int q = s->h_nzind[e], i = s->h_trv[q], j = colidx[q];
          ents.push_back({i, j, s->h_one[e]});
          if (i != j) ents.push_back({j, i, s->h_one[e]});
        }
        std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
        for (size_t t = 0; t < ents.size(); t++) {
          if (t > 0 && ents[t].i == ents[t - 1].i && ents[t].j == ents[t - 1].j) { g_val.back() += ents[t].v; continue; }
          g_col.push_back(ents[t].j);
          g_val.push_back(ents[t].v);
          g_ptr[ents[t].i + 1]++;
        }
        for (int64_t i = 0; i < n; i++) g_ptr[i + 1] += g_ptr[i];
      }
      lap("A_g CSR");
      // diagonal-only matrices, by row, ascending matrix order
      std::vector<int> d_ptr(n + 1, 0);
      for (int64_t k = 0; k < s->n_sparse; k++)
        if ((int)k != kg)
          for (int e = s->h_matptr[k]; e < s->h_matptr[k + 1]; e++) d_ptr[s->h_trv[s->h_nzind[e]] + 1]++;
      for (int64_t i = 0; i < n; i++) d_ptr[i + 1] += d_ptr[i];
      std::vector<int> d_gid(d_ptr[n]), fillp(d_ptr.begin(), d_ptr.end() - 1);
      std::vector<double> d_val(d_ptr[n]);
      for (int64_t k = 0; k < s->n_sparse; k++)
        if ((int)k != kg)
          for (int e = s->h_matptr[k]; e < s->h_matptr[k + 1]; e++) {
            const int pos = fillp[s->h_trv[s->h_nzind[e]]]++;
            d_gid[pos] = s->h_gids[k];
            d_val[pos] = s->h_one[e];
          }
      std::vector<int> diagpos(n, -1);
      for (int64_t j = 0; j < n; j++)
        for (int q = s->h_tcp[j]; q < s->h_tcp[j + 1]; q++)
          if (s->h_trv[q] == (int)j) diagpos[j] = q;
      // segmented-reduction plan without A_g
      std::vector<int> f_short, f_cb, f_ce, f_long, f_lcp(1, 0);
      for (int64_t k = 0; k < s->n_sparse; k++) {
        if ((int)k == kg) continue;
        const int len = s->h_matptr[k + 1] - s->h_matptr[k];
        if (len <= SHORT_MAX) { f_short.push_back((int)k); continue; }
        f_long.push_back((int)k);
        for (int bgn = s->h_matptr[k]; bgn < s->h_matptr[k + 1]; bgn += CHUNK) {
          f_cb.push_back(bgn);
          f_ce.push_back(std::min(bgn + CHUNK, s->h_matptr[k + 1]));
        }
        f_lcp.push_back((int)f_cb.size());
      }
      s->sp_fast = sp;
      DevSparse& sf = s->sp_fast;
      sf.n_short = (int)f_short.size(); sf.n_chunks = (int)f_cb.size(); sf.n_long = (int)f_long.size();
      sf.n_short_blocks = (sf.n_short + SDPLR_NT - 1) / SDPLR_NT;
      if ((rc = upload(s, &sf.short_ids, f_short))) return rc;
      if ((rc = upload(s, &sf.chunk_beg, f_cb))) return rc;
      if ((rc = upload(s, &sf.chunk_end, f_ce))) return rc;
      if ((rc = upload(s, &sf.long_ids, f_long))) return rc;
      if ((rc = upload(s, &sf.long_chunk_ptr, f_lcp))) return rc;
      DevSparse& sg = s->spg;
      sg = DevSparse{};
      sg.n = (int)n; sg.nnzS = (int)g_col.size();
      if ((rc = plan_long(g_ptr, sg))) return rc;
      if ((rc = upload(s, &sg.colptr, g_ptr))) return rc;
      if ((rc = upload(s, &sg.rowval, g_col))) return rc;
      { const double* gv = nullptr; if ((rc = upload(s, &gv, g_val))) return rc; sg.nzval = const_cast<double*>(gv); }
      // host copy of A_g's CSR: the column-sweep tiles depend on the sub-wave width, i.e. on the rank, and are
      // rebuilt by reset_rank
      s->h_gptr = g_ptr; s->h_gcol = g_col; s->h_gval = g_val;
      lap("diagonal lists + uploads");
      // an instance small enough for the resident route never walks the tiles unless it leaves that route (rank growth,
      // Armijo, profiling): their construction — more than half of a small instance's finalize — waits for that day
      s->tiles_deferred = tiles_can_wait(s);
      if (!s->tiles_deferred && (rc = build_tiles(s))) return rc;
      lap("tiles");
      s->ff.gid_g = s->h_gids[kg];
      if ((rc = upload(s, &s->ff.diagpos, diagpos))) return rc;
      if ((rc = upload(s, &s->ff.drow_ptr, d_ptr))) return rc;
      if ((rc = upload(s, &s->ff.drow_gid, d_gid))) return rc;
      if ((rc = upload(s, &s->ff.drow_val, d_val))) return rc;
      s->fast = true;
      bool single = s->all_covered;
      for (int64_t k = 0; k < s->n_sparse && single; k++)
        if ((int)k != kg && s->h_matptr[k + 1] - s->h_matptr[k] != 1) single = false;
      std::vector<int> extra(1, s->h_gids[kg]);
      for (auto& L : s->h_lr) extra.push_back((int)L.gid);
      if ((rc = upload(s, &s->extra_slots, extra))) return rc;
      for (size_t t = 0; t < 4; t++) s->extra_head.k[t] = t < extra.size() ? extra[t] : 0;
      s->n_extra = (int)extra.size();
      s->fast_singleton = single && s->spg.n_long_rows == 0 && getenv("SDPLR_HIP_NO_FAST2") == nullptr;
      // resident route for small instances (k_resident.h): row-local constraint data, one constraint per row at most
      // … and at most one rank-one matrix among the constraints (MinBisection), A_g being the cost matrix
      const bool lr_none = s->h_lr.empty();
      const bool lr_one = s->h_lr.size() == 1 && s->h_lr[0].s == 1 && s->h_lr[0].gid < s->m && s->h_gids[kg] == (int)s->m &&
                          getenv("SDPLR_HIP_NO_RESIDENT_LR") == nullptr;
      bool one_per_row = s->fast_singleton && (lr_none || lr_one);
      for (int64_t i = 0; i < n && one_per_row; i++) one_per_row = d_ptr[i + 1] - d_ptr[i] <= 1;
      if (one_per_row) {
        std::vector<int> row_k(n, -1);
        std::vector<double> row_v(n, 0.0);
        for (int64_t i = 0; i < n; i++)
          if (d_ptr[i + 1] > d_ptr[i]) { row_k[i] = d_gid[d_ptr[i]]; row_v[i] = d_val[d_ptr[i]]; }
        if ((rc = upload(s, &s->rs_row_k, row_k))) return rc;
        if ((rc = upload(s, &s->rs_row_v, row_v))) return rc;
        if ((rc = build_rs_ell(s, g_ptr, g_col, g_val))) return rc;
      }
    }
  }
  // ---- edge path: disjoint supports (every pattern position owned by at most one sparse matrix) and at most one
  // sparse matrix with more than one upper-triangular entry (Lovász-θ: one constraint per edge + the identity) ----
  sp.big_gid = -1;
  if (s->have_sparse && !s->fast && s->n_sparse > 0 && getenv("SDPLR_HIP_NO_EDGE") == nullptr) {
    bool disjoint = true;
    for (int64_t q = 0; q < s->nnzT && disjoint; q++) disjoint = tptr[q + 1] - tptr[q] <= 1;
    int n_big = 0, big = -1;
    for (int64_t k = 0; k < s->n_sparse; k++)
      if (s->h_matptr[k + 1] - s->h_matptr[k] > 1) { n_big++; big = s->h_gids[k]; }
    if (disjoint && n_big <= 1) {
      std::vector<int> q_gid(s->nnzT, -1), s_gid(s->nnzS, -1);
      std::vector<double> q_two(s->nnzT, 0.0), s_one(s->nnzS, 0.0);
      for (int64_t k = 0; k < s->n_sparse; k++)
        for (int e = s->h_matptr[k]; e < s->h_matptr[k + 1]; e++) {
          q_gid[s->h_nzind[e]] = s->h_gids[k];
          q_two[s->h_nzind[e]] = s->h_two[e];
        }
      for (int64_t p = 0; p < s->nnzS; p++) {
        const int q = s->h_mapped[p];
        if (tptr[q + 1] > tptr[q]) { s_gid[p] = tmat[tptr[q]]; s_one[p] = tval[tptr[q]]; }
      }
      {
        std::vector<double> rec((size_t)s->nnzT * 4, 0.0);
        for (int64_t q = 0; q < s->nnzT; q++) {
          const unsigned long long w0 = (unsigned long long)(unsigned)s->h_trv[q] | ((unsigned long long)(unsigned)colidx[q] << 32);
          const unsigned long long w1 = (unsigned long long)(unsigned)q_gid[q];
          memcpy(&rec[q * 4 + 0], &w0, 8);
          memcpy(&rec[q * 4 + 1], &w1, 8);
          rec[q * 4 + 2] = q_two[q];
        }
        if ((rc = upload(s, &sp.q_rec, rec))) return rc;
      }
      if ((rc = upload(s, &sp.q_gid, q_gid))) return rc;
      if ((rc = upload(s, &sp.q_two, q_two))) return rc;
      if ((rc = upload(s, &sp.s_gid, s_gid))) return rc;
      if ((rc = upload(s, &sp.s_one, s_one))) return rc;
      sp.big_gid = big;
      {   // S assembly by owner: the full-pattern entries of each single-entry matrix; the big matrix's list
        std::vector<int> own_pos(2 * (m + 1), -1), big_pos;
        std::vector<double> own_one(m + 1, 0.0), big_one;
        for (int64_t p = 0; p < s->nnzS; p++) {
          const int g = s_gid[p];
          if (g < 0) continue;
          if (g == big) { big_pos.push_back((int)p); big_one.push_back(s_one[p]); continue; }
          if (own_pos[2 * g] < 0) own_pos[2 * g] = (int)p; else own_pos[2 * g + 1] = (int)p;
          own_one[g] = s_one[p];
        }
        if ((rc = upload(s, &sp.own_pos, own_pos))) return rc;
        if ((rc = upload(s, &sp.own_one, own_one))) return rc;
        if ((rc = upload(s, &sp.big_pos, big_pos))) return rc;
        if ((rc = upload(s, &sp.big_one, big_one))) return rc;
        sp.n_big_pos = (int)big_pos.size();
      }
      std::vector<int> extra;
      if (big >= 0) extra.push_back(big);
      for (auto& L : s->h_lr) extra.push_back((int)L.gid);
      if ((rc = upload(s, &s->edge_extra, extra))) return rc;
      for (size_t t = 0; t < 4; t++) s->edge_extra_head.k[t] = t < extra.size() ? extra[t] : 0;
      s->n_edge_extra = (int)extra.size();
      // the low-rank projections ride the SDDMM when there is one column and every row has a diagonal position
      bool all_diag = true;
      {
        std::vector<char> has(n, 0);
        for (int64_t j = 0; j < n; j++)
          for (int q = s->h_tcp[j]; q < s->h_tcp[j + 1]; q++)
            if (s->h_trv[q] == (int)j) has[j] = 1;
        all_diag = std::all_of(has.begin(), has.end(), [](char c) { return c != 0; });
      }
      s->edge_lr_fused = lr.ST == 1 && all_diag && !s->no_lrfuse;
      s->edge = true;
    }
  }
  lap("classification / edge plan");
  // The band plan of the Lanczos SpMV (after the classification: the palette form needs to know A_g) is ≈ 60 ms of host
  // work at the north-star size and is not needed before the first dual bound: built on a thread of its own while the
  // first inner loops run, uploaded by the first Lanczos run (ensure_band).  SDPLR_HIP_BAND_SYNC=1: built here.
  s->use_band = false;
  s->band_host.reset(new BandHost());
  s->band_host->fcp = s->h_fcp;
  s->band_host->frv = std::move(s->h_frv);
  {
    BandHost& H = *s->band_host;
    auto envi = [](const char* k) { const char* e = getenv(k); return e ? std::max(1, atoi(e)) : 0; };
    H.env_min_n = envi("SDPLR_HIP_LZBAND_MIN_N"); H.env_bw = envi("SDPLR_HIP_LZBAND_BW");
    H.env_ch = envi("SDPLR_HIP_LZBAND_CH"); H.env_nc = envi("SDPLR_HIP_LZBAND_NC");
    H.env_no_band = getenv("SDPLR_HIP_NO_LZBAND") != nullptr; H.env_no_pal = getenv("SDPLR_HIP_NO_LZPAL") != nullptr;
  }
  // (no thread for an instance the plan does not apply to — n < 2¹⁴: band_host returns at once — a batch of small instances
  // would start and join one per instance for nothing)
  const bool band_possible = s->have_sparse && s->nnzS > 0 && !s->band_host->env_no_band &&
                             n >= (s->band_host->env_min_n > 0 ? s->band_host->env_min_n : (1 << 14));
  if (getenv("SDPLR_HIP_BAND_SYNC") != nullptr || !band_possible) {
    band_host(s, *s->band_host);
  } else {
    S* sp_ = s;
    BandHost* hp_ = s->band_host.get();
    s->band_thread = std::thread([sp_, hp_] { band_host(sp_, *hp_); });
  }
  lap("band plan (started)");
  if ((rc = alloc_factors(s))) return rc;
  lap("factors");
  s->nb_m = blocks_for(m + 1, SDPLR_NT, 256);
  s->nb_spmv = blocks_for(n, SDPLR_NT / 8, 768);
  s->nb_nnzT = blocks_for(s->nnzT, SDPLR_NT, 4096);
  s->nb_nnzS = blocks_for(s->nnzS, SDPLR_NT, 4096);
  s->nb_n = blocks_for(n, SDPLR_NT, 1024);
  lap("before commit");
  if ((rc = arena_commit(s))) return rc;
  s->up.on = false;
  lap("arena commit");
  // low-rank scratch depends on r: allocated for the largest rank seen (reset_rank re-allocates)
  if ((rc = dzero(s, &s->lr_part, (size_t)std::max(s->nb_lr, SDPLR_MAXNB) * 2 * std::max(lr.ST, 1) * s->r))) return rc;
  if ((rc = dzero(s, &s->lr_W, (size_t)2 * std::max(lr.ST, 1) * s->r))) return rc;
  if ((rc = dzero(s, &s->lr_WS, (size_t)std::max(lr.ST, 1) * s->r))) return rc;
  if (s->tiles_deferred && !s->rs_ok) {   // not an instance of the resident route after all
    s->tiles_deferred = false;
    if ((rc = build_tiles(s))) return rc;
  }
  // release host staging
  std::vector<int>().swap(s->h_nzind); std::vector<int>().swap(s->h_trv); std::vector<int>().swap(s->h_frv);
  std::vector<int>().swap(s->h_mapped); std::vector<double>().swap(s->h_one); std::vector<double>().swap(s->h_two);
  for (auto& L : s->h_lr) { std::vector<double>().swap(L.B); }
  lap("rank-sized scratch");
  HIPCK(s, hipStreamSynchronize(s->stream));
  lap("final sync");
  s->finalized = true;
  return SDPLR_OK;
}

int32_t sdplr_hip_destroy(S* s) {
  ApiShared api_guard(dev_of(s));
  if (!s) return SDPLR_OK;
  if (s->band_thread.joinable()) s->band_thread.join();
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (auto& p : s->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : s->ev_pool) (void)hipEventDestroy(e);
  for (void* p : s->allocs) pool_free(p);
  if (s->arena.base) pool_free(s->arena.base);
  if (s->lz_alpha) pool_free(s->lz_alpha);
  if (s->lz_beta) pool_free(s->lz_beta);
  if (s->hc) pool_host_ctrl_free(s->hc);
  if (s->up.host) pool_host_chunk_free(s->up.host);   // (a finalize that failed half-way)
  if (s->lz_graph) (void)hipGraphExecDestroy(s->lz_graph);
  for (int k = 0; k < 2; k++) {
    if (s->snap[k]) pool_host_ctrl_free(s->snap[k]);
    if (s->snap_ev[k]) pool_event_free(s->snap_ev[k]);
    if (s->graph_exec[k]) (void)hipGraphExecDestroy(s->graph_exec[k]);
  }
  for (int k = 2; k < 5; k++)
    if (s->graph_exec[k]) (void)hipGraphExecDestroy(s->graph_exec[k]);
  if (s->stream) { (void)hipStreamSynchronize(s->stream); pool_stream_free(s->stream); }
  delete s;
  return SDPLR_OK;
}

int32_t sdplr_hip_reset_rank(S* s, int64_t new_r) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  if (new_r < 1 || s->n * new_r >= (1LL << 40)) return fail(s, SDPLR_ERR_INVALID_ARG, "reset_rank: bad rank");
  HIPCK(s, hipStreamSynchronize(s->stream));
  if (s->arena.base) pool_free(s->arena.base);
  s->arena.base = nullptr;
  for (int k = 0; k < 5; k++)
    if (s->graph_exec[k]) { (void)hipGraphExecDestroy(s->graph_exec[k]); s->graph_exec[k] = nullptr; }
  // everything sized by the rank is released before it is re-allocated (the graphs above hold their pointers)
  for (double** p : {&s->lr_part, &s->lr_W, &s->lr_WS, &s->scratchF[0], &s->scratchF[1]}) { dfree(s, *p); *p = nullptr; }
  s->r = new_r;
  int rc = SDPLR_OK;
  if (s->fast) {
    s->tiles_deferred = s->rs_ok && tiles_can_wait(s);
    if (!s->tiles_deferred) rc = build_tiles(s);
  }
  if (rc) return rc;
  if ((rc = alloc_factors(s))) return rc;
  const int64_t m = s->m;
  HIPCK(s, hipMemsetAsync(s->lambda, 0, std::max<int64_t>(m, 1) * sizeof(double), s->stream));
  HIPCK(s, hipMemsetAsync(s->pv, 0, std::max<int64_t>(m, 1) * sizeof(double), s->stream));
  for (double* p : {s->y, s->pv_raw, s->A_RD, s->A_DD}) HIPCK(s, hipMemsetAsync(p, 0, (m + 1) * sizeof(double), s->stream));
  if ((rc = dzero(s, &s->lr_part, (size_t)std::max(s->nb_lr, SDPLR_MAXNB) * 2 * std::max(s->lr.ST, 1) * s->r))) return rc;
  if ((rc = dzero(s, &s->lr_W, (size_t)2 * std::max(s->lr.ST, 1) * s->r))) return rc;
  if ((rc = dzero(s, &s->lr_WS, (size_t)std::max(s->lr.ST, 1) * s->r))) return rc;
  if ((rc = pull(s))) return rc;
  DevCtrl keep = *s->hc;
  memset(s->hc, 0, sizeof(DevCtrl));
  s->hc->sigma = keep.sigma;
  s->hc->alpha_max = 1.0;
  s->hc->latest = (int)s->h;
  s->gram_dirty = s->sg_stale = s->ynext_pending = false;
  s->P_valid = false;
  s->S_stale = false;
  return push(s);
}

// ================================================================================================
// state transfer
// ================================================================================================
int32_t sdplr_hip_set_factor(S* s, int32_t slot, const double* h) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  s->G_consistent = false;   // (the host writes state behind G's back)
  { const int rc_r = ensure_canonical(s); if (rc_r) return rc_r; }
  { const int rc_d = ensure_dirt(s); if (rc_d) return rc_d; }
  double* p = factor_ptr(s, slot);
  if (!p || !h) return fail(s, SDPLR_ERR_INVALID_ARG, "set_factor: bad slot");
  {
    const int rc = h2d_copy(s, p, h, (size_t)s->N * sizeof(double));
    if (rc) return rc;
  }
  note_factor_written(s, slot);
  return SDPLR_OK;
}
int32_t sdplr_hip_get_factor(S* s, int32_t slot, double* h) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  if (slot != SDPLR_F_RT) {   // (R is never in ring form, and does not wait for the lazy dirt ← s_latest)
    const int rc_d = ensure_dirt(s);
    if (rc_d) return rc_d;
  }
  double* p = factor_ptr(s, slot);
  if (!p || !h) return fail(s, SDPLR_ERR_INVALID_ARG, "get_factor: bad slot");
  HIPCK(s, hipMemcpyAsync(h, p, s->N * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SDPLR_OK;
}
}  // extern "C"

namespace {
double* vec_ptr(S* s, int32_t which, int64_t* len, bool* in_ctrl) {
  const int64_t m = s->m;
  *in_ctrl = false;
  switch (which) {
    case SDPLR_V_LAMBDA: *len = m; return s->lambda;
    case SDPLR_V_LAMBDA_UB: *len = m; return s->lambda_ub;
    case SDPLR_V_B: *len = m; return s->b;
    case SDPLR_V_Y: *len = m + 1; return s->y;
    case SDPLR_V_PV_RAW: *len = m + 1; return s->pv_raw;
    case SDPLR_V_PV_LB: *len = m; return s->pv_lb;
    case SDPLR_V_PV: *len = m; return s->pv;
    case SDPLR_V_A_RD: *len = m + 1; return s->A_RD;
    case SDPLR_V_A_DD: *len = m + 1; return s->A_DD;
    case SDPLR_V_LBFGS_RHO: *len = s->h; *in_ctrl = !s->lit; return s->lit ? s->lit_rho : s->hc->rho;
    case SDPLR_V_LBFGS_A: *len = s->h; *in_ctrl = !s->lit; return s->lit ? s->lit_a : s->hc->a;
    case SDPLR_V_UVT: *len = s->nnzT; return s->sp.UVt0;
    case SDPLR_V_TRIU_S_NZVAL: *len = s->nnzT; return s->sp.triu_nzval;
    case SDPLR_V_S_NZVAL: *len = s->nnzS; return s->sp.nzval;
    case SDPLR_V_SCRATCH:
      *len = m + 1;
      if (!s->scratchV && dzero(s, &s->scratchV, (size_t)m + 1) != SDPLR_OK) *len = -1;
      return s->scratchV;
    default: *len = -1; return nullptr;
  }
}
}  // namespace

extern "C" {
int32_t sdplr_hip_set_vec(S* s, int32_t which, const double* h, int64_t len) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  s->G_consistent = false;   // (the host writes state behind G's back)
  int64_t L; bool in_ctrl;
  double* p = vec_ptr(s, which, &L, &in_ctrl);
  if (L < 0 || len != L || (!h && L > 0)) return fail(s, SDPLR_ERR_INVALID_ARG, "set_vec: bad slot/length");
  if (L == 0) return SDPLR_OK;
  if (in_ctrl) {
    int rc = pull_if_stale(s);
    if (rc) return rc;
    memcpy(p, h, L * sizeof(double));
    return push(s);
  }
  HIPCK(s, hipMemcpyAsync(p, h, L * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  if (which == SDPLR_V_Y) s->S_from_y = false;   // until the next 𝒜t_preprocess!
  if (which == SDPLR_V_S_NZVAL || which == SDPLR_V_TRIU_S_NZVAL) {
    s->S_from_y = false;
    s->S_stale = false;                          // the caller's values stand: nothing may re-assemble over them
  }
  return SDPLR_OK;
}
int32_t sdplr_hip_get_vec(S* s, int32_t which, double* h, int64_t len) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  int64_t L; bool in_ctrl;
  double* p = vec_ptr(s, which, &L, &in_ctrl);
  if (L < 0 || len != L || (!h && L > 0)) return fail(s, SDPLR_ERR_INVALID_ARG, "get_vec: bad slot/length");
  if (which == SDPLR_V_S_NZVAL || which == SDPLR_V_TRIU_S_NZVAL) ensure_S(s);
  if (L == 0) return SDPLR_OK;
  if (in_ctrl) {
    int rc = pull_if_stale(s);
    if (rc) return rc;
    memcpy(h, p, L * sizeof(double));
    return SDPLR_OK;
  }
  HIPCK(s, hipMemcpyAsync(h, p, L * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SDPLR_OK;
}
int32_t sdplr_hip_set_scalar(S* s, int32_t which, double v) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  s->G_consistent = false;   // (the host writes state behind G's back)
  int rc = pull_if_stale(s);
  if (rc) return rc;
  if (which == SDPLR_S_SIGMA) s->hc->sigma = v;
  else if (which == SDPLR_S_OBJ) s->hc->obj = v;
  else if (which == SDPLR_S_LBFGS_LATEST) {
    if (s->h > 0 && (v < 1 || v > (double)s->h)) return fail(s, SDPLR_ERR_INVALID_ARG, "set_scalar: latest out of range");
    s->hc->latest = (int)v;
    s->gram_dirty = true;
  } else return fail(s, SDPLR_ERR_INVALID_ARG, "set_scalar: bad slot");
  return push(s);
}
int32_t sdplr_hip_get_scalar(S* s, int32_t which, double* v) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  if (!v) return fail(s, SDPLR_ERR_INVALID_ARG, "get_scalar: null");
  int rc = pull_if_stale(s);
  if (rc) return rc;
  if (which == SDPLR_S_SIGMA) *v = s->hc->sigma;
  else if (which == SDPLR_S_OBJ) *v = s->hc->obj;
  else if (which == SDPLR_S_LBFGS_LATEST) *v = (double)s->hc->latest;
  else return fail(s, SDPLR_ERR_INVALID_ARG, "get_scalar: bad slot");
  return SDPLR_OK;
}
int32_t sdplr_hip_get_dims(const S* s, int64_t* n, int64_t* m, int64_t* r, int64_t* h, int64_t* nnzT,
                           int64_t* nnzS, int64_t* nnzAgg) {
  if (!s) return SDPLR_ERR_INVALID_ARG;
  if (n) *n = s->n;
  if (m) *m = s->m;
  if (r) *r = s->r;
  if (h) *h = s->h;
  if (nnzT) *nnzT = s->nnzT;
  if (nnzS) *nnzS = s->nnzS;
  if (nnzAgg) *nnzAgg = s->nnzAgg;
  return SDPLR_OK;
}
}  // extern "C"

// ================================================================================================
// kernel sequences (enqueue only; no synchronisation)
// ================================================================================================
namespace {

#define LV_CASE(L, V, CALL) if (s->LPR == L && s->VEC == V) { constexpr int LPR = L, VEC = V; CALL; } else
#define LV_DISPATCH(CALL)                                                                              \
  LV_CASE(1, 1, CALL) LV_CASE(2, 1, CALL) LV_CASE(4, 1, CALL) LV_CASE(8, 1, CALL) LV_CASE(16, 1, CALL) \
  LV_CASE(32, 1, CALL) LV_CASE(64, 1, CALL) LV_CASE(1, 2, CALL) LV_CASE(2, 2, CALL) LV_CASE(4, 2, CALL) \
  LV_CASE(8, 2, CALL) LV_CASE(16, 2, CALL) LV_CASE(32, 2, CALL) LV_CASE(64, 2, CALL) {}
// (the tile kernel's own sub-wave shape: tile_shape_lpr)
#define TLV_CASE(L, V, CALL) if (s->tile_lpr == L && s->VEC == V) { constexpr int LPR = L, VEC = V; CALL; } else
#define TLV_DISPATCH(CALL)                                                                                \
  TLV_CASE(1, 1, CALL) TLV_CASE(2, 1, CALL) TLV_CASE(4, 1, CALL) TLV_CASE(8, 1, CALL) TLV_CASE(16, 1, CALL) \
  TLV_CASE(32, 1, CALL) TLV_CASE(64, 1, CALL) TLV_CASE(1, 2, CALL) TLV_CASE(2, 2, CALL) TLV_CASE(4, 2, CALL) \
  TLV_CASE(8, 2, CALL) TLV_CASE(16, 2, CALL) TLV_CASE(32, 2, CALL) TLV_CASE(64, 2, CALL) {}
// (the shapes the ring form of the history is built for: ring_shape_ok)
#define RING_DISPATCH(CALL)                                                                            \
  LV_CASE(4, 2, CALL) LV_CASE(8, 2, CALL) LV_CASE(16, 2, CALL) LV_CASE(32, 2, CALL) LV_CASE(64, 2, CALL) {}
#define HM_DISPATCH(CALL)                                       \
  if (s->HM == 4) { constexpr int HM = 4; CALL; }               \
  else if (s->HM == 8) { constexpr int HM = 8; CALL; }          \
  else { constexpr int HM = 16; CALL; }

// tiles taller than 8 rows need more than 64 KB of dynamic LDS per block: asked for once per (shape, handle), outside
// any stream capture (sdplr_hip_inner_loop calls this before it enqueues anything)
int tile_lds_attr(S* s) {
  if (s->tiles_deferred) {   // first use of the multi-launch route by an instance that was set up for the resident one
    s->tiles_deferred = false;
    int rc = build_tiles(s);
    if (rc) return rc;
  }
  if (!s->use_tile || s->tile_attr_done) return SDPLR_OK;
  const int bytes = (int)(((size_t)SDPLR_NT * s->VEC * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->tile_lpr) * (2 * s->tile.K + 8) + (size_t)(SDPLR_NT / 64) * 2 * s->tile_lpr * s->VEC) * sizeof(double));
  // The attribute is per function and PROCESS-wide: each instantiation keeps the largest value any handle has asked for
  // (a smaller request from another handle, or after a rank change, must not lower the cap under the first one's feet).
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  hipError_t e = hipSuccess;
#define TILE_ATTR(LRN, UNI)                                                                                                      \
  TLV_DISPATCH(({                                                                                                                  \
    static int cur_max = 0;                                                                                                       \
    if (bytes > cur_max) {                                                                                                        \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_tile<LPR, VEC, LRN, UNI>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
      if (e == hipSuccess) cur_max = bytes;                                                                                       \
    }                                                                                                                             \
  }))
  TILE_ATTR(0, false)
  if (e == hipSuccess) { TILE_ATTR(1, false) }
  if (e == hipSuccess) { TILE_ATTR(0, true) }
  if (e == hipSuccess) { TILE_ATTR(1, true) }
#undef TILE_ATTR
  if (e != hipSuccess) return fail(s, SDPLR_ERR_HIP, std::string("hipFuncSetAttribute(k_spmm_tile): ") + hipGetErrorString(e));
  s->tile_attr_done = true;
  return SDPLR_OK;
}

// W = X0ᵀB (and X1ᵀB) followed by the mode-specific tail of k_lr_finalize
void enq_lowrank(S* s, const double* X0, const double* X1, int F, int mode, double* out0, double* out1, int chk) {
  if (s->lr.ST == 0) return;
  {
    ProfScope ps(s, "lr_project");
    if (F == 1) { LV_DISPATCH((k_lr_project<LPR, VEC, 1><<<s->nb_lr, SDPLR_NT, 0, s->stream>>>(s->lr, X0, X1, (int)s->n, (int)s->r, s->lr_part, s->ctrl, chk))) }
    else { LV_DISPATCH((k_lr_project<LPR, VEC, 2><<<s->nb_lr, SDPLR_NT, 0, s->stream>>>(s->lr, X0, X1, (int)s->n, (int)s->r, s->lr_part, s->ctrl, chk))) }
  }
  int nbf = s->nb_lr;
  if (s->nb_lr > 256) {   // many blocks: their partials are summed by many waves, not by the one finalize block
    ProfScope ps(s, "lr_reduce");
    const int nout = F * s->lr.ST * (int)s->r;
    k_lr_reduce<<<nout, SDPLR_NT, 0, s->stream>>>(nout, s->nb_lr, s->lr_part, s->lr_W, s->ctrl, chk);
    nbf = 0;
  }
  ProfScope ps(s, "lr_finalize");
  k_lr_finalize<<<1, 1024, 0, s->stream>>>(s->lr, (int)s->r, F, nbf, s->lr_part, s->lr_W, mode, out0, out1, s->y, s->lr_WS, s->ctrl, chk);
}

// mode 0: out0 = 𝒜(UUᵀ); mode 1: out0 = 𝒜((UVᵀ+VUᵀ)/2); mode 2: out0 = 2·𝒜((UVᵀ+VUᵀ)/2), out1 = 𝒜(VVᵀ)
void enq_A(S* s, const double* U, const double* V, int mode, double* out0, double* out1, int chk) {
  if (!s->all_covered) {
    (void)hipMemsetAsync(out0, 0, (s->m + 1) * sizeof(double), s->stream);  // fill!(out, 0), src/coreop.jl:39,60
    if (mode == 2) (void)hipMemsetAsync(out1, 0, (s->m + 1) * sizeof(double), s->stream);
  }
  if (s->n_sparse > 0) {
    {
      ProfScope ps(s, mode == 2 ? "sddmm_linesearch" : (mode == 1 ? "sddmm_uv" : "sddmm_uu"));
      if (mode == 0) { LV_DISPATCH((k_sddmm<LPR, VEC, 0><<<s->nb_sddmm, SDPLR_NT, 0, s->stream>>>(s->sp, U, V, (int)s->r, s->ctrl, chk))) }
      else if (mode == 1) { LV_DISPATCH((k_sddmm<LPR, VEC, 1><<<s->nb_sddmm, SDPLR_NT, 0, s->stream>>>(s->sp, U, V, (int)s->r, s->ctrl, chk))) }
      else { LV_DISPATCH((k_sddmm<LPR, VEC, 2><<<s->nb_sddmm, SDPLR_NT, 0, s->stream>>>(s->sp, U, V, (int)s->r, s->ctrl, chk))) }
    }
    const int nb = s->sp.n_short_blocks + s->sp.n_chunks;
    if (nb > 0) {
      ProfScope ps(s, "segreduce");
      if (mode == 2) k_segreduce<true><<<nb, SDPLR_NT, 0, s->stream>>>(s->sp, out0, out1, s->ctrl, chk);
      else k_segreduce<false><<<nb, SDPLR_NT, 0, s->stream>>>(s->sp, out0, out1, s->ctrl, chk);
    }
    if (s->sp.n_long > 0) {
      ProfScope ps(s, "seg_finalize");
      const int nbf = (s->sp.n_long * 64 + SDPLR_NT - 1) / SDPLR_NT;
      if (mode == 2) k_seg_finalize<true><<<nbf, SDPLR_NT, 0, s->stream>>>(s->sp, out0, out1, s->ctrl, chk);
      else k_seg_finalize<false><<<nbf, SDPLR_NT, 0, s->stream>>>(s->sp, out0, out1, s->ctrl, chk);
    }
  }
  if (mode == 0) enq_lowrank(s, U, U, 1, 0, out0, out1, chk);
  else enq_lowrank(s, U, V, 2, mode, out0, out1, chk);
}

void enq_At_preprocess(S* s, int chk) {
  s->S_from_y = true;
  if (s->n_sparse <= 0) return;
  {
    ProfScope ps(s, "assemble_triu");
    k_assemble_triu<<<s->nb_nnzT, SDPLR_NT, 0, s->stream>>>(s->sp, s->y, s->ctrl, chk);
  }
  ProfScope ps(s, "assemble_full");
  k_assemble_full<<<s->nb_nnzS, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, chk);
}

// the inner loop of the structured / edge paths leaves y current but S unassembled: assemble on first use
void ensure_S(S* s) {
  if (!s->S_stale) return;
  enq_At_preprocess(s, 0);
  s->S_stale = false;
}

// Y = scale·(X·S + low-rank); slot ≥ 0 also yields the ‖Y‖² partials
// lr_step: X is the point the line search has just moved to and lr_W still holds [X_oldᵀB; DᵀB] from its
// 𝒜 pass, so X_newᵀB = X_oldᵀB + α·DᵀB (exact by linearity) and no pass over X is needed for the low-rank term
void enq_spmm_S(S* s, double* Y, const double* X, double scale, int slot, int chk);
void enq_At_left(S* s, double* Y, const double* X, double scale, int slot, int chk, bool lr_step = false) {
  if (lr_step && s->lr.ST > 0) {
    ProfScope ps(s, "fast_lr_ws");
    k_fast_lr_ws<<<1, SDPLR_NT, 0, s->stream>>>(s->lr, (int)s->r, s->lr_W, s->y, s->lr_WS, s->ctrl, chk);
  } else {
    enq_lowrank(s, X, X, 1, 3, nullptr, nullptr, chk);
  }
  enq_spmm_S(s, Y, X, scale, slot, chk);
}
// the SpMM with S alone (low-rank coefficients lr_WS already in place)
void enq_spmm_S(S* s, double* Y, const double* X, double scale, int slot, int chk) {
  ProfScope ps(s, "spmm");
  if (s->sp.n_long_rows > 0) {
    const int nbl = std::min(s->sp.n_long_rows, 256);
    LV_DISPATCH((k_spmm_both<LPR, VEC><<<s->nb_spmm + nbl, SDPLR_NT, 0, s->stream>>>(s->sp, X, Y, (int)s->r, scale, s->lr, s->lr_WS, slot, s->partials, s->ctrl, chk, nullptr, nbl)))
  } else {
    LV_DISPATCH((k_spmm<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>(s->sp, X, Y, (int)s->r, scale, s->lr, s->lr_WS, slot, s->partials, s->ctrl, chk)))
  }
}

void enq_copy2y(S* s, int chk) {
  ProfScope ps(s, "copy2y");
  k_copy2y<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->y, s->lambda, s->lambda_ub, s->pv_raw, chk);
}

// g!: src/coreop.jl:305-317; y_done = 1 when the line-search commit already produced y
void enq_g(S* s, int chk, bool y_done, bool lr_step = false) {
  if (!y_done) enq_copy2y(s, chk);
  enq_At_preprocess(s, chk);
  enq_At_left(s, aslot(s->arena, AS_G), aslot(s->arena, AS_R), 2.0, SLOT_GNORM2, chk, lr_step);
}

// f!: src/coreop.jl:11-31
void enq_f(S* s) {
  enq_A(s, aslot(s->arena, AS_R), nullptr, 0, s->pv_raw, nullptr, 0);
  {
    ProfScope ps(s, "f_tail");
    k_f_tail<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->pv_raw, s->b, s->pv_lb, s->pv, s->lambda, s->lambda_ub, s->partials);
  }
  ProfScope ps(s, "f_finalize");
  k_f_finalize<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->pv_raw, s->nb_m, s->partials);
}

// 1-block seam kernel: fold update partials / loop tests / two-loop coefficients (k_dense.h)
void enq_boundary(S* s, int jfixed, int fin_mode, int do_loop, int do_coeff, int desc_mode = 0) {
  ProfScope ps(s, "lbfgs_boundary");
  if (s->lit) {   // histories beyond SDPLR_HMAX keep no Gram data: the seam folds the norms and makes the loop tests only
    k_lbfgs_boundary<<<1, 1024, 0, s->stream>>>(s->ctrl, 0, 0, 0, do_loop, 0, 1, s->partials, 0, s->pv_raw, (int)s->m);
    return;
  }
  k_lbfgs_boundary<<<1, 1024, 0, s->stream>>>(s->ctrl, (int)s->h, jfixed, fin_mode, do_loop, do_coeff, s->gram_nb, s->partials, desc_mode, s->pv_raw, (int)s->m);
}
void enq_gram_row(S* s, int j) {
  s->gram_nb = s->nb_upd;
  for (int l0 = 0; l0 < (int)s->h; l0 += 4)   // one launch per window of four history slots
    k_lbfgs_update<false><<<s->nb_upd, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, j, 0, l0, 0, s->partials);
  enq_boundary(s, j, 2, 0, 0);
}
// make the Gram data consistent with the stored history and the current G (see k_dense.h)
void ensure_gram(S* s) {
  if (s->h == 0 || s->lit) { s->gram_dirty = s->sg_stale = s->ynext_pending = false; return; }
  if (s->gram_dirty || s->ynext_pending) {
    ProfScope ps(s, "gram_recompute");
    for (int j = 0; j < (int)s->h; j++) enq_gram_row(s, j);
  } else if (s->sg_stale) {
    ProfScope ps(s, "gram_recompute");
    enq_gram_row(s, 0);
  }
  s->gram_dirty = s->sg_stale = s->ynext_pending = false;
}

// lbfgs_dir! (+ descent): [seam: pending Gram rows, loop tests, coefficients] → direction → descent
// (with the optional on-device steepest-descent fallback)
void enq_lbfgs_dir(S* s, int negate, int in_loop, int apply_fallback, bool skip_ynext = false) {
  // m == 0: the reference returns right after copyto!(dir, grad), before the negation
  // (src/lbfgs.jl:88-91); the caller's descent test then falls back to −G (src/sdplr.jl:202-205).
  if (s->h == 0) negate = 0;
  // Inside the device-driven loop ⟨dir, G⟩ is evaluated by the seam kernel from the Gram data (the same data
  // the direction's coefficients come from) and the fallback is taken by k_lbfgs_dir itself: one launch fewer
  // per iteration.  The stand-alone operator (and SDPLR_HIP_DOT_DESCENT=1) reduces the dot product instead.
  const int analytic = (in_loop && apply_fallback && !s->dot_descent && !s->lit) ? 1 : 0;
  enq_boundary(s, 0, in_loop ? 1 : 0, in_loop, 1, analytic ? (negate ? 1 : 2) : 0);
  if (s->lit) {   // numlbfgsvecs > SDPLR_HMAX: the recursion as written (k_dense.h, k_lit_*)
    ProfScope ps(s, "lbfgs_dir");
    const int h = (int)s->h, nb = s->nb_upd;
    k_lit_copy<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, in_loop);
    for (int i = 0; i < h; i++) {          // newest → oldest (src/lbfgs.jl:94-102)
      k_lit_dot<<<nb, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, h, i, 0, SLOT_GRAM + 1, s->partials, in_loop);
      k_lit_axpy<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, h, i, 0, s->lit_rho, s->lit_a, SLOT_GRAM + 1, nb, s->partials, in_loop);
    }
    for (int i = h - 1; i >= 0; i--) {     // oldest → newest (:104-113)
      k_lit_dot<<<nb, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, h, i, 1, SLOT_GRAM + 1, s->partials, in_loop);
      k_lit_axpy<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, h, i, 1, s->lit_rho, s->lit_a, SLOT_GRAM + 1, nb, s->partials, in_loop);
    }
    k_lit_finish<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, h, negate, s->partials, in_loop);
  } else {
    ProfScope ps(s, "lbfgs_dir");
    // (history loads non-temporal only when the arena does not fit the 256 MiB Infinity Cache: k_dense.h, NTH)
    static const long long nt_above = getenv("SDPLR_HIP_DIR_NT_ABOVE_MB") ? atoll(getenv("SDPLR_HIP_DIR_NT_ABOVE_MB")) << 20 : 200LL << 20;
    const long long arena_bytes = (long long)(3 + 2 * s->h + (s->fast ? 2 : 0)) * s->arena.stride * (long long)sizeof(double);
    if (arena_bytes > nt_above) {
      HM_DISPATCH((k_lbfgs_dir<HM, true><<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, negate, in_loop, s->partials, (analytic && skip_ynext) ? 2 : analytic)))
    } else {
      HM_DISPATCH((k_lbfgs_dir<HM, false><<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, negate, in_loop, s->partials, (analytic && skip_ynext) ? 2 : analytic)))
    }
  }
  if (analytic) return;
  ProfScope ps(s, "descent");
  const int nb = apply_fallback ? s->nb_dense : 1;
  k_descent<<<nb, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, s->nb_dense, apply_fallback, in_loop, s->partials);
}

// the in-loop seam and direction on the ring form of the history (k_dense.h)
void enq_dir_ring(S* s) {
  enq_boundary(s, 0, 1, 1, 1, 1);
  ProfScope ps(s, "lbfgs_dir");
  // (history loads non-temporal only when the arena does not fit the 256 MiB Infinity Cache: k_dense.h, NTH)
  static const long long nt_above = getenv("SDPLR_HIP_DIR_NT_ABOVE_MB") ? atoll(getenv("SDPLR_HIP_DIR_NT_ABOVE_MB")) << 20 : 200LL << 20;
  const long long arena_bytes = (long long)(3 + 2 * s->h + (s->fast ? 2 : 0)) * s->arena.stride * (long long)sizeof(double);
  if (arena_bytes > nt_above) k_lbfgs_dir_ring<4, true><<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, 1);
  else k_lbfgs_dir_ring<4, false><<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, 1);
}

// lbfgs_update!: the pass over the history; its partials are folded by the next seam kernel
void enq_lbfgs_update(S* s, int chk) {
  if (s->h == 0) return;
  s->gram_nb = s->nb_upd;
  ProfScope ps(s, "lbfgs_update");
  if (s->lit) {
    k_lit_update<<<s->nb_upd, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, s->partials, chk);
    k_lit_rho<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->h, s->lit_rho, s->nb_upd, s->partials, chk);
    return;
  }
  k_lbfgs_update<true><<<s->nb_upd, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, 0, 1, 0, chk, s->partials);
  for (int l0 = 4; l0 < (int)s->h; l0 += 4)   // longer histories: the dots of the new pair with the other windows of four slots
    k_lbfgs_update<false><<<s->nb_upd, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->arena, s->N, (int)s->h, 0, 1, l0, chk, s->partials);
}

// both line searches up to and including the commit; fuse_y also writes y of the following g!
void enq_linesearch(S* s, int armijo, int chk, int fuse_y) {
  enq_A(s, aslot(s->arena, AS_R), aslot(s->arena, AS_D), 2, s->A_RD, s->A_DD, chk);
  if (armijo) {
    {
      ProfScope ps(s, "armijo_partials");
      k_armijo_partials<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->lambda, s->lambda_ub, s->pv_raw, s->A_RD, s->A_DD, s->y, s->partials, chk);
    }
    ProfScope ps(s, "armijo_pick");
    k_armijo_pick<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->nb_m, s->A_RD, s->A_DD, s->partials, chk, fuse_y);
  } else {
    {
      ProfScope ps(s, "ls_partials");
      k_ls_partials<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, chk);
    }
    ProfScope ps(s, "ls_solve");
    k_ls_solve<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->nb_m, s->A_RD, s->A_DD, s->partials, chk, fuse_y);
  }
  ProfScope ps(s, "ls_commit");
  k_ls_commit<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->pv_raw, s->A_RD, s->A_DD, s->pv_lb, s->pv, fuse_y, s->y, s->lambda, s->lambda_ub, s->partials, chk, fuse_y, s->nb_spmm + std::min(s->sp.n_long_rows, 256));
}

void enq_axpy_R(S* s, int chk) {
  ProfScope ps(s, "axpy_R");
  k_axpy_R<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(s->ctrl, aslot(s->arena, AS_R), aslot(s->arena, AS_D), s->N, chk);
}

// one pass of the while body, src/sdplr.jl:190-278, entirely device-driven
// lbfgs_update! rides the SpMM of g! on the generic in-loop path (k_spmm_both_upd) for h ≤ 4
bool spmm_fuses_update(const S* s) {
  return s->h >= 1 && s->h <= 4 && s->n_sparse > 0 && !s->no_lrfuse && !s->no_updfuse && !s->dot_descent;
}
int spmm_upd_blocks(const S* s) { return s->nb_spmm + std::min(s->sp.n_long_rows, 256); }

void enq_iteration(S* s, int armijo) {
  const bool upd_fused = spmm_fuses_update(s);
  s->gram_nb = upd_fused ? spmm_upd_blocks(s) : s->nb_upd;   // the Gram partials' producer grid (read by the seam kernel)
  enq_lbfgs_dir(s, 1, 1, 1, upd_fused);      // :197-205
  enq_linesearch(s, armijo, 1, 1);           // :210-214
  if (s->n_sparse > 0 && !s->no_lrfuse) {     // :219 and the head of g! (:221) as one launch, then the SpMM
    {
      ProfScope ps(s, "step_jobs");
      k_step_jobs<<<s->nb_dense + s->nb_nnzS + (s->lr.ST > 0 ? 1 : 0), SDPLR_NT, 0, s->stream>>>(s->sp, s->y, s->ctrl, aslot(s->arena, AS_R), aslot(s->arena, AS_D), s->N, s->nb_dense, s->nb_nnzS, s->lr, (int)s->r, s->lr_W, s->lr_WS);
    }
    if (upd_fused) {   // … and lbfgs_update! (:244-246) on the rows of G as they are produced
      ProfScope ps(s, "spmm");
      const int nbl = std::min(s->sp.n_long_rows, 256);
      LV_DISPATCH((k_spmm_both_upd<LPR, VEC, 4><<<s->nb_spmm + nbl, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>(s->sp, aslot(s->arena, AS_R), aslot(s->arena, AS_G), (int)s->r, 2.0, s->lr, s->lr_WS, SLOT_GNORM2, s->partials, s->ctrl, nbl, s->arena, (int)s->h, aslot(s->arena, AS_D))))
      return;
    }
    enq_spmm_S(s, aslot(s->arena, AS_G), aslot(s->arena, AS_R), 2.0, SLOT_GNORM2, 1);
  } else {
    enq_axpy_R(s, 1);                          // :219
    enq_g(s, 1, true, !s->no_lrfuse);          // :221
  }
  // norms and the exit tests (:224-241, :272-277, :190) are folded by the next seam kernel
  enq_lbfgs_update(s, 1);                    // :244-246
}

// edge path (k_sparse.h): exact line search, lbfgs_update! fused into the SpMM (h ≤ 4): 7 launches
bool edge_applies(const S* s, int armijo) {
  return s->edge && !armijo && spmm_fuses_update(s);
}
void enq_iteration_edge(S* s) {
  double *R = aslot(s->arena, AS_R), *G = aslot(s->arena, AS_G), *D = aslot(s->arena, AS_D);
  s->gram_nb = spmm_upd_blocks(s);
  enq_lbfgs_dir(s, 1, 1, 1, true);                                                       // :197-205
  if (!s->all_covered) {
    (void)hipMemsetAsync(s->A_RD, 0, (s->m + 1) * sizeof(double), s->stream);
    (void)hipMemsetAsync(s->A_DD, 0, (s->m + 1) * sizeof(double), s->stream);
  }
  const bool lrf = s->edge_lr_fused && s->r <= (int64_t)s->LPR * s->VEC;
  const int nb_c = blocks_for(s->m + 1, SDPLR_NT, 1024);   // one constraint per thread: the m-passes are latency-bound
  {
    ProfScope ps(s, "sddmm_linesearch");   // 𝒜(RDᵀ+DRᵀ), 𝒜(DDᵀ) of the sparse matrices + line-search sums (+ projections)
    if (lrf) { LV_DISPATCH((k_sddmm_edge<LPR, VEC, 1><<<s->nb_edge, SDPLR_NT, 0, s->stream>>>(s->sp, (int)s->m, R, D, (int)s->r, s->A_RD, s->A_DD, s->lr, s->lr_part, s->partials, s->ctrl))) }
    else { LV_DISPATCH((k_sddmm_edge<LPR, VEC, 0><<<s->nb_edge, SDPLR_NT, 0, s->stream>>>(s->sp, (int)s->m, R, D, (int)s->r, s->A_RD, s->A_DD, s->lr, s->lr_part, s->partials, s->ctrl))) }
  }
  if (!lrf) enq_lowrank(s, R, D, 2, 2, s->A_RD, s->A_DD, 1);
  {
    ProfScope ps(s, "edge_sums");   // projection sums ‖ the big matrix's sums ‖ line-search partials of the other constraints
    const int nout = lrf ? 2 * s->lr.ST * (int)s->r : 0;
    k_edge_sums<<<nout + 2 + nb_c, SDPLR_NT, 0, s->stream>>>(nout, s->nb_edge, s->lr_part, s->lr_W, s->red10, nb_c, (int)s->m, s->n_edge_extra, s->edge_extra, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl);
  }
  {
    ProfScope ps(s, "ls_solve_fast");
    k_ls_solve_fast<<<1, SDPLR_LSF_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->sp.big_gid, s->n_edge_extra, s->edge_extra, nb_c, s->A_RD, s->A_DD, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->y, s->lr.ST, (int)s->r, s->lr.col_gid, s->lr.Dcat, s->lr_W, s->lr_WS, s->partials, 1, lrf ? 1 : 0, s->lr.n_lr, s->lr.mat_ptr, s->lr.mat_gid, s->red10, s->edge_extra_head);
  }
  const int nbl = std::min(s->sp.n_long_rows, 256);
  {
    ProfScope ps(s, "step_jobs");                                                         // :219 ‖ commit + y
    const int nb_big = s->sp.n_big_pos > 0 ? blocks_for(s->sp.n_big_pos, SDPLR_NT, 256) : 0;
    k_edge_step<<<s->nb_dense + nb_c + nb_big, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, R, D, s->N, s->nb_dense, nb_c, (int)s->m, s->n_edge_extra, s->edge_extra, s->pv_raw, s->A_RD, s->A_DD, s->pv_lb, s->pv, s->y, s->lambda, s->lambda_ub, s->partials, s->nb_spmm + nbl);
  }
  ProfScope ps(s, "spmm");                                                                // g! (:221) + lbfgs_update! (:244-246)
  if (getenv("SDPLR_HIP_EDGE_GATHER_S")) {   // S gathered by the SpMM itself (an extra dependent round trip per batch: slower; kept for the comparison)
    LV_DISPATCH((k_spmm_both_upd<LPR, VEC, 4, true><<<s->nb_spmm + nbl, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>(s->sp, R, G, (int)s->r, 2.0, s->lr, s->lr_WS, SLOT_GNORM2, s->partials, s->ctrl, nbl, s->arena, (int)s->h, D, s->y)))
  } else {
    LV_DISPATCH((k_spmm_both_upd<LPR, VEC, 4, false><<<s->nb_spmm + nbl, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>(s->sp, R, G, (int)s->r, 2.0, s->lr, s->lr_WS, SLOT_GNORM2, s->partials, s->ctrl, nbl, s->arena, (int)s->h, D)))
  }
}

// ---- structured fast path (k_sparse.h, DevFast) -----------------------------------------------------
inline double* fast_P(S* s) { return aslot(s->arena, 3 + 2 * (int)s->h); }
inline double* fast_W(S* s) { return aslot(s->arena, 3 + 2 * (int)s->h + 1); }

// P = A_g·R from scratch (entry of the inner loop: removes any drift accumulated by P += α·W)
void enq_fast_refresh_P(S* s) {
  ProfScope ps(s, "spmm_P");
  DevLowRank none{};
  if (s->spg.n_long_rows > 0) {
    const int nbl = std::min(s->spg.n_long_rows, 256);
    LV_DISPATCH((k_spmm_both<LPR, VEC><<<s->nb_spmm + nbl, SDPLR_NT, 0, s->stream>>>(s->spg, aslot(s->arena, AS_R), fast_P(s), (int)s->r, 1.0, none, nullptr, -1, s->partials, s->ctrl, 0, nullptr, nbl)))
  } else {
    LV_DISPATCH((k_spmm<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>(s->spg, aslot(s->arena, AS_R), fast_P(s), (int)s->r, 1.0, none, nullptr, -1, s->partials, s->ctrl, 0)))
  }
}

// one pass of the while body, src/sdplr.jl:190-278, with ONE gather pass (W = A_g·D)
void enq_iteration_fast(S* s, int armijo) {
  double *R = aslot(s->arena, AS_R), *G = aslot(s->arena, AS_G), *D = aslot(s->arena, AS_D);
  double *P = fast_P(s), *W = fast_W(s);
  const bool upd_fused = step_fuses_update(s);
  s->gram_nb = upd_fused ? s->nb_step : s->nb_upd;
  enq_lbfgs_dir(s, 1, 1, 1, upd_fused && !s->dot_descent);                            // :197-205
  // ---- line search head: 𝒜(RDᵀ+DRᵀ), 𝒜(DDᵀ)  (src/linesearch.jl:8-18) ----
  if (!s->all_covered) {
    (void)hipMemsetAsync(s->A_RD, 0, (s->m + 1) * sizeof(double), s->stream);
    (void)hipMemsetAsync(s->A_DD, 0, (s->m + 1) * sizeof(double), s->stream);
  }
  // W = A_g·D and the row dots: the column-sweep tile kernel when it applies (no hub rows), else two kernels
  const bool tiled = s->use_tile && s->tile_lpr == tile_shape_lpr(s) && s->n * s->r * 8 < (1LL << 32) && s->spg.n_long_rows == 0;
  if (tiled) {
    ProfScope ps(s, "spmm_W");
    const size_t lds = ((size_t)SDPLR_NT * s->VEC * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->tile_lpr) * (2 * s->tile.K + 8)) * sizeof(double);
    if (s->tile.gdiag) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0, true><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->sp_fast.UVt0, s->sp_fast.UVt1, s->partials, s->ctrl, 1, s->lr, s->lr_part, 1))) }
    else { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->sp_fast.UVt0, s->sp_fast.UVt1, s->partials, s->ctrl, 1, s->lr, s->lr_part, 1))) }
  } else {
  {
    ProfScope ps(s, "rowdots");
    LV_DISPATCH((k_rowdots<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>(s->sp_fast, s->ff, R, D, P, (int)s->r, SLOT_PD, s->partials, s->ctrl, 1)))
  }
  {
    ProfScope ps(s, "spmm_W");
    DevLowRank none{};
    if (s->spg.n_long_rows > 0) {
      const int nbl = std::min(s->spg.n_long_rows, 256);
      LV_DISPATCH((k_spmm_both<LPR, VEC><<<s->nb_spmm + nbl, SDPLR_NT, 0, s->stream>>>(s->spg, D, W, (int)s->r, 1.0, none, nullptr, SLOT_DW, s->partials, s->ctrl, 1, D, nbl)))
    } else {
      LV_DISPATCH((k_spmm<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>(s->spg, D, W, (int)s->r, 1.0, none, nullptr, SLOT_DW, s->partials, s->ctrl, 1, D)))
    }
  }
  }
  {
    const int nb = s->sp_fast.n_short_blocks + s->sp_fast.n_chunks;
    if (nb > 0) {
      ProfScope ps(s, "segreduce");
      k_segreduce<true><<<nb, SDPLR_NT, 0, s->stream>>>(s->sp_fast, s->A_RD, s->A_DD, s->ctrl, 1);
    }
    if (s->sp_fast.n_long > 0) {
      ProfScope ps(s, "seg_finalize");
      const int nbf = (s->sp_fast.n_long * 64 + SDPLR_NT - 1) / SDPLR_NT;
      k_seg_finalize<true><<<nbf, SDPLR_NT, 0, s->stream>>>(s->sp_fast, s->A_RD, s->A_DD, s->ctrl, 1);
    }
  }
  {
    ProfScope ps(s, "fast_fill");
    if (tiled) k_fast_fill<<<1, SDPLR_NT, 0, s->stream>>>(s->ff, s->A_RD, s->A_DD, SLOT_PD, s->nb_tile, SLOT_DW, s->nb_tile, s->partials, s->ctrl, 1);
    else k_fast_fill<<<1, SDPLR_NT, 0, s->stream>>>(s->ff, s->A_RD, s->A_DD, SLOT_PD, s->nb_spmm, SLOT_DW, s->nb_spmm + std::min(s->spg.n_long_rows, 256), s->partials, s->ctrl, 1);
  }
  enq_lowrank(s, R, D, 2, 2, s->A_RD, s->A_DD, 1);
  // ---- scalar stage + commit (with y of the following g!) ----
  if (armijo) {
    {
      ProfScope ps(s, "armijo_partials");
      k_armijo_partials<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->lambda, s->lambda_ub, s->pv_raw, s->A_RD, s->A_DD, s->y, s->partials, 1);
    }
    ProfScope ps(s, "armijo_pick");
    k_armijo_pick<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->nb_m, s->A_RD, s->A_DD, s->partials, 1, 1);
  } else {
    {
      ProfScope ps(s, "ls_partials");
      k_ls_partials<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, 1);
    }
    ProfScope ps(s, "ls_solve");
    k_ls_solve<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->nb_m, s->A_RD, s->A_DD, s->partials, 1, 1);
  }
  {
    ProfScope ps(s, "ls_commit");
    k_ls_commit<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->pv_raw, s->A_RD, s->A_DD, s->pv_lb, s->pv, 1, s->y, s->lambda, s->lambda_ub, s->partials, 1, 1, s->nb_step);
  }
  if (s->lr.ST > 0) {
    ProfScope ps(s, "fast_lr_ws");
    k_fast_lr_ws<<<1, SDPLR_NT, 0, s->stream>>>(s->lr, (int)s->r, s->lr_W, s->y, s->lr_WS, s->ctrl, 1);
  }
  {
    ProfScope ps(s, "fast_step");                                                     // :219-221 (and :244-246 when fused)
    if (upd_fused) {
      LV_DISPATCH((k_fast_step2<LPR, VEC, 4, false><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, s->dot_descent ? 0 : 1)))
    } else {
      LV_DISPATCH((k_fast_step2<LPR, VEC, 0, false><<<s->nb_step, SDPLR_NT, 0, s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, 0)))
    }
  }
  // norms and the exit tests (:224-241, :272-277, :190) are folded by the next seam kernel
  if (!upd_fused) enq_lbfgs_update(s, 1);                                             // :244-246
}

// singleton form of the fast path (exact line search only): 5 launches per iteration (seam, direction, gather,
// line-search solve, step + update), 6 with a rank-1 dense constraint, 6 when lbfgs_update! is not fused (h > 4)
void enq_iteration_fast2(S* s) {
  double *R = aslot(s->arena, AS_R), *G = aslot(s->arena, AS_G), *D = aslot(s->arena, AS_D);
  double *P = fast_P(s), *W = fast_W(s);
  bool lr_fused = false;
  // lbfgs_update! fused into the step kernel for h ≤ 4 (its Gram partials then come from nb_step blocks)
  const bool upd_fused = step_fuses_update(s);
  s->gram_nb = upd_fused ? s->nb_step : s->nb_upd;
  if (s->ring_now && s->pdrop_now) {   // ring form of the history (k_dense.h): seam, direction, gather, step — the same four launches
    enq_dir_ring(s);
    {
      ProfScope ps(s, "spmm_W");
      const size_t lds = ((size_t)SDPLR_NT * s->VEC * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->tile_lpr) * (2 * s->tile.K + 8)) * sizeof(double);
      if (s->tile.gdiag) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0, true><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, 1, s->arena))) }
      else { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, 1, s->arena))) }
    }
    ProfScope ps(s, "fast_step");
    RING_DISPATCH((k_fast_step_ring<LPR, VEC, 4><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, W, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->arena, (int)s->h, s->nb_tile)))
    return;
  }
  const bool ring_pb = s->ring_now;   // … the P-based step kernel on the ring form: the launches below, D found on the ring
  if (ring_pb) enq_dir_ring(s);
  else
  enq_lbfgs_dir(s, 1, 1, 1, upd_fused && !s->dot_descent);                            // :197-205
#ifdef SDPLR_PROBE_TILE_HIST
  s->ff.probe_hist[0] = G;
  for (int l = 0; l < 3; l++) {
    s->ff.probe_hist[1 + l] = aslot(s->arena, AS_S0 + std::min<int>(l, (int)s->h - 1));
    s->ff.probe_hist[4 + l] = aslot(s->arena, as_y0(s->arena) + std::min<int>(l, (int)s->h - 1));
  }
#endif
  {
    ProfScope ps(s, "spmm_W");   // W = A_g·D + row dots + line-search sums of the row-attached constraints
    if (s->use_tile && s->tile_lpr == tile_shape_lpr(s) && s->n * s->r * 8 < (1LL << 32)) {
      lr_fused = s->lr.ST == 1 && s->r <= (int64_t)s->tile_lpr * s->VEC && !s->no_lrfuse;
      const size_t lds = ((size_t)SDPLR_NT * s->VEC * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->tile_lpr) * (2 * s->tile.K + 8) + (lr_fused ? (size_t)(SDPLR_NT / 64) * 2 * s->tile_lpr * s->VEC : 0)) * sizeof(double);
      if (lr_fused && s->tile.gdiag) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 1, true><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, ring_pb ? 1 : 0, s->arena))) }
      else if (lr_fused) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 1><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, ring_pb ? 1 : 0, s->arena))) }
      else if (s->tile_panels && s->VEC == 2 && s->tile_lpr == 16 && s->LPR == 16 && !s->tile.gdiag) {
        // rank panels: the same kernel with 8 bytes per lane walks the lists once per 16-column half of the rank, so a
        // gathered row is 128 B and the XCD's L2 holds twice as many of them (tile kernel)
        const size_t lds1 = ((size_t)SDPLR_NT * 1 * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->LPR) * (2 * s->tile.K + 8)) * sizeof(double);
        k_spmm_tile<16, 1, 0><<<s->nb_tile, SDPLR_NT, lds1, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0);
      }
      else if (s->tile.gdiag) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0, true><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, ring_pb ? 1 : 0, s->arena))) }
      else { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->lr, s->lr_part, 0, ring_pb ? 1 : 0, s->arena))) }
    } else {
      LV_DISPATCH((k_spmm_fast<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>(s->spg, (int)s->m, s->ff, R, D, P, W, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 1)))
    }
  }
  if (lr_fused) {  // the projections came out of the tile kernel: sum the block partials (many blocks, one wave
                   // per output); the mode-2 tail of k_lr_finalize is taken by k_ls_solve_fast
    ProfScope ps(s, "lr_reduce");
    const int nout = 2 * s->lr.ST * (int)s->r;
    k_lr_reduce<<<nout, SDPLR_NT, 0, s->stream>>>(nout, s->nb_tile, s->lr_part, s->lr_W, s->ctrl, 1);
  } else {
    enq_lowrank(s, R, D, 2, 2, s->A_RD, s->A_DD, 1);
  }
  // the scalar stage as the step kernel's own prologue (k_sparse.h, LSH): with the P-less kernel, the cost matrix the only
  // slot not attached to a row, and ≤ 1024 producers of line-search partials
  const int nb_ls = (s->use_tile && s->tile_lpr == tile_shape_lpr(s) && s->n * s->r * 8 < (1LL << 32)) ? s->nb_tile : s->nb_spmm;
  const bool lsh = upd_fused && s->pdrop_now && s->n_extra == 1 && s->lr.ST == 0 && nb_ls <= 4 * SDPLR_NT && !s->no_lshead;
  if (!lsh) {
    ProfScope ps(s, "ls_solve_fast");
    k_ls_solve_fast<<<1, SDPLR_LSF_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->ff.gid_g, s->n_extra, s->extra_slots, (s->use_tile && s->tile_lpr == tile_shape_lpr(s) && s->n * s->r * 8 < (1LL << 32)) ? s->nb_tile : s->nb_spmm, s->A_RD, s->A_DD, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->y, s->lr.ST, (int)s->r, s->lr.col_gid, s->lr.Dcat, s->lr_W, s->lr_WS, s->partials, 1, lr_fused ? 1 : 0, s->lr.n_lr, s->lr.mat_ptr, s->lr.mat_gid, nullptr, s->extra_head);
  }
  // (a variant fusing this step kernel with lbfgs_update! was measured at 111 µs against 38 + 59 µs for the
  // two kernels — 166 VGPRs and scratch — and dropped)
  {
    ProfScope ps(s, "fast_step");                                                     // :219-234
    if (lsh) {   // … P-less, with the line-search scalar stage as its prologue
      LV_DISPATCH((k_fast_step2<LPR, VEC, 4, true, true, true><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, 1, nb_ls)))
    } else if (upd_fused && s->pdrop_now) {   // … P-less: G carried forward from G_old (k_sparse.h, PDROP)
      LV_DISPATCH((k_fast_step2<LPR, VEC, 4, true, true><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, 1)))
    } else if (ring_pb) {   // … the same on the ring form of the history (no s_j, y_j stored)
      RING_DISPATCH((k_fast_step_ring<LPR, VEC, 4, true><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, W, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->partials, s->ctrl, 1, s->arena, (int)s->h, 0, P, s->lr, s->lr_WS)))
    } else if (upd_fused) {   // … and lbfgs_update! (:244-246) in the same pass
      LV_DISPATCH((k_fast_step2<LPR, VEC, 4, true><<<s->nb_step, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, s->dot_descent ? 0 : 1)))
    } else {
      LV_DISPATCH((k_fast_step2<LPR, VEC, 0, true><<<s->nb_step, SDPLR_NT, 0, s->stream>>>((int)s->n, (int)s->m, s->ff, R, D, P, W, G, (int)s->r, s->y, s->lambda, s->lambda_ub, s->pv_raw, s->pv_lb, s->pv, s->A_RD, s->A_DD, s->lr, s->lr_WS, s->partials, s->ctrl, 1, s->arena, (int)s->h, 0)))
    }
  }
  if (!upd_fused) enq_lbfgs_update(s, 1);                                             // :244-246
}

// ---- resident route for small instances (k_resident.h) ----------------------------------------------------------
// One workgroup owns the instance for a whole inner loop / Lanczos run: one launch per call.
double wall_clock_hz() {   // rate of wall_clock64() on this device
  static const double hz = [] {
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    return 1e3 * (double)khz;
  }();
  return hz;
}
// SDPLR_HIP_FORCE_GRAPH ("treat this instance as a large one": the tests' default) and SDPLR_HIP_NO_RESIDENT keep an
// instance on the multi-launch routes.
bool rs_loop_applies(const S* s, int armijo) {
  if (!s->rs_ok || armijo || s->h < 1 || s->h > 4 || s->prof_on) return false;
  if (s->force_graph || getenv("SDPLR_HIP_NO_RESIDENT") != nullptr) return false;
  return rs_loop_lds(s) <= RS_LDS_MAX;
}
// the structured form: the instances of the resident loop (any rank), S = S(y) of the device's y
bool rs_lanczos_ell_applies(const S* s) {
  if (!s->rs_ok || !s->S_from_y || s->prof_on) return false;
  if (s->force_graph || getenv("SDPLR_HIP_NO_RESIDENT") != nullptr || getenv("SDPLR_HIP_CLASSIC_LANCZOS") != nullptr) return false;
  return (size_t)4 * s->n * sizeof(double) <= RS_LDS_MAX;
}
bool rs_fg_applies(const S* s) {
  if (!s->rs_ok || s->prof_on) return false;
  if (s->force_graph || getenv("SDPLR_HIP_NO_RESIDENT") != nullptr) return false;
  return rs_loop_lds(s) <= RS_LDS_MAX;
}
bool rs_lanczos_applies(const S* s) {
  if (s->force_graph || getenv("SDPLR_HIP_NO_RESIDENT") != nullptr || getenv("SDPLR_HIP_CLASSIC_LANCZOS") != nullptr) return false;
  if (s->prof_on || !s->have_sparse || s->lr.ST > SDPLR_LRMAX) return false;
  int64_t max_nnz = 1 << 14;   // (12 bytes per entry from L2 at ≈ 67 GB/s per CU: beyond this two launches per step are faster)
  if (const char* e = getenv("SDPLR_HIP_RESIDENT_LZ_NNZ")) max_nnz = atoll(e);
  return s->nnzS <= max_nnz && (size_t)3 * s->n * sizeof(double) <= RS_LDS_MAX;
}
// the resident kernels come in two shapes: 16-byte pieces of a row (even ranks) or 8-byte ones
#define RS_VEC_DISPATCH(s_, CALL)                                  \
  if ((s_)->r % 2 == 0) { constexpr int VEC = 2; CALL; }           \
  else { constexpr int VEC = 1; CALL; }
inline int rs_vec(const S* s) { return s->r % 2 == 0 ? 2 : 1; }
// more than 64 KB of dynamic LDS has to be asked for: a per-function, process-wide attribute — set once per
// instantiation, to the fixed upper bound
#define RS_SET_ATTR(kernel)                                                                                          \
  do {                                                                                                               \
    static std::atomic<int> attr_done{0};                                                                            \
    if (!attr_done.load(std::memory_order_acquire)) {                                                                \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                              (int)RS_LDS_MAX) == hipSuccess)                                                       \
        attr_done.store(1, std::memory_order_release);                                                               \
    }                                                                                                                \
  } while (0)

RsLoopArgs rs_loop_args(S* s, double time_budget_s, bool refresh_P, bool pre_lambda, bool pre_clear_fg) {
  RsLoopArgs a{};
  a.lr = rs_lr(s);
  a.rowvec = rs_rows_global(s) ? s->rs_rowvec : nullptr;
  a.pre_lambda = pre_lambda ? 1 : 0;
  a.pre_clear = a.pre_fg = pre_clear_fg ? 1 : 0;
  a.b = s->b;
  a.lam_rw = s->lambda;
  a.n = (int)s->n; a.m = (int)s->m; a.r = (int)s->r; a.h = (int)s->h;
  a.gid_g = s->ff.gid_g;
  a.row_k = s->rs_row_k; a.row_v = s->rs_row_v;
  a.E = s->rs_ell;
  a.A = s->arena;
  a.P = aslot(s->arena, 3 + 2 * (int)s->h); a.W = aslot(s->arena, 3 + 2 * (int)s->h + 1);
  a.y = s->y; a.pv_raw = s->pv_raw; a.pv = s->pv; a.A_RD = s->A_RD; a.A_DD = s->A_DD;
  a.lam = s->lambda; a.lam_ub = s->lambda_ub; a.lb = s->pv_lb;
  a.c = s->ctrl;
  a.refresh_P = refresh_P ? 1 : 0;
  a.budget_ticks = time_budget_s > 0 ? std::max<long long>(1, (long long)(time_budget_s * wall_clock_hz())) : 0;
  return a;
}
struct RsLoopIn {   // what a major iteration writes into the control block in front of its loop
  double sigma, gtol, fprec, normC, normb;
  int grel, prel;
  long long max_iters;
};
void rs_loop_set_in(RsLoopArgs& a, const RsLoopIn& in) {
  a.in_set = 1;
  a.in_sigma = in.sigma; a.in_gtol = in.gtol; a.in_fprec = in.fprec; a.in_normC = in.normC; a.in_normb = in.normb;
  a.in_grel = in.grel; a.in_prel = in.prel; a.in_max_iters = in.max_iters;
}
// the resident loop without P (k_resident.h, PDROP): A_g is the cost matrix
bool rs_can_drop_P(const S* s) { return s->ff.gid_g == (int)s->m && !s->no_pdrop; }
int enq_resident_loop(S* s, double time_budget_s, bool refresh_P, bool pre_lambda = false, bool pre_clear_fg = false,
                      const RsLoopIn* in = nullptr, bool pdrop = false) {
  RsLoopArgs a = rs_loop_args(s, time_budget_s, refresh_P, pre_lambda, pre_clear_fg);
  if (in) rs_loop_set_in(a, *in);
  const size_t lds = rs_loop_lds(s);
  const int W = pdrop ? rs_team_w(s) : 1;
  if (W > 1) {   // a team: W workgroups of one XCD (blocks 0, 8, …: k_resident.h)
    a.team_w = W;
    a.xch = s->rs_xch;
    a.team_test_fail = getenv("SDPLR_HIP_TEAM_TEST_FAIL") != nullptr;
    HIPCK(s, hipMemsetAsync(s->rs_xch + 48 * W, 0, 8 * sizeof(double), s->stream));   // arrival counter, failure flag, XCC ids
    RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_team<VEC, 4>)); k_rs_team<VEC, 4><<<8 * (W - 1) + 1, SDPLR_RS_NT, lds, s->stream>>>(a); }))
  } else if (pdrop) { RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_loop<VEC, 4, true>)); k_rs_loop<VEC, 4, true><<<1, SDPLR_RS_NT, lds, s->stream>>>(a); })) }
  else { RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_loop<VEC, 4, false>)); k_rs_loop<VEC, 4, false><<<1, SDPLR_RS_NT, lds, s->stream>>>(a); })) }
  HIPCK(s, hipGetLastError());
  s->st_rs_loops++;
  return SDPLR_OK;
}

// launch + wait + the host-side bookkeeping of one resident loop (the control block has been pushed by the caller);
// pre_lambda / pre_clear_fg: the head of a major iteration rides the same launch (sdplr_hip_major_iteration)
int run_resident_loop(S* s, double time_budget_s, bool pre_lambda, bool pre_clear_fg, double* Lio, double* gnio, double* pnio,
                      double* last_alpha, int64_t* iters, int32_t* exit_reason, const RsLoopIn* in = nullptr,
                      bool G_was_consistent = false, bool resume = false) {
  static const int64_t refresh_iters = getenv("SDPLR_HIP_P_REFRESH_ITERS") ? atoll(getenv("SDPLR_HIP_P_REFRESH_ITERS")) : 256;
  // without P: G is carried forward from the fg! of the prologue, or from a G known to be the gradient at the device's
  // state and not older than the refresh interval (otherwise this call runs the P-based loop, which rebuilds G at every step).
  // resume (SDPLR_MAJOR_RESUME): the continuation of a capped loop takes the kernel the capped call took and refreshes
  // nothing by age, so that cap + resume ≡ one uncapped call, bit for bit.
  const bool pdrop = rs_can_drop_P(s) && (pre_clear_fg || (G_was_consistent && (resume ? s->pdrop_now : s->G_age < refresh_iters)));
  const bool refresh = pre_clear_fg || !s->P_valid || (!resume && s->P_age >= refresh_iters);   // (fg! rebuilds P = A_g·R itself)
  int rc = enq_resident_loop(s, time_budget_s, refresh, pre_lambda, pre_clear_fg, in, pdrop);
  if (rc) return rc;
  s->pdrop_now = pdrop;
  s->P_valid = !pdrop;
  if (refresh) s->P_age = 0;
  if (pre_clear_fg || !pdrop) s->G_age = 0;
  s->S_stale = true;   // y is current, S is assembled by whoever reads it next (ensure_S)
  s->S_from_y = true;
  if (pre_clear_fg) { s->st_rs_fg++; s->gram_dirty = false; }
  if ((rc = pull_blocking(s))) return rc;
  DevCtrl* c = s->hc;
  if (c->err == SDPLR_ERR_TEAM_PLACEMENT) {   // the members were not dealt to one XCD: nothing was touched — once more, without a team
    c->err = 0;
    s->no_team = true;
    if ((rc = push(s))) return rc;
    if ((rc = enq_resident_loop(s, time_budget_s, refresh, pre_lambda, pre_clear_fg, in, pdrop))) return rc;
    s->st_rs_loops--;
    if ((rc = pull_blocking(s))) return rc;
  }
  if (c->err == SDPLR_ERR_TEAM_TIMEOUT) {
    c->err = 0; c->done = 0;
    (void)push(s);
    return fail(s, SDPLR_ERR_HIP, "resident loop: a team member never arrived at a barrier (the grid was not co-resident)");
  }
  const int why_rs = c->exit_reason;
  // (`dirt *= α`, src/lbfgs.jl:142: the kernel itself left dirt = s_latest)
  // no iteration ran after the fg! of the prologue: the dots with G are those of a cleared history (zero, exact)
  s->sg_stale = (why_rs == EXIT_RELDELTA);
  s->ynext_pending = (why_rs == EXIT_RELDELTA) && s->h > 0;
  if (c->err == SDPLR_ERR_NOT_DESCENT) {
    c->err = 0; c->done = 0;
    (void)push(s);
    return fail(s, SDPLR_ERR_NOT_DESCENT, "Error: cubic[1] should be less than 0.");
  }
  s->st_iters += c->iters;
  s->P_age += c->iters;
  if (pdrop) { s->G_age += c->iters; s->st_pdrop++; }
  s->G_consistent = true;
  *Lio = c->L; *gnio = c->gnorm; *pnio = c->pvnorm;
  if (last_alpha) *last_alpha = c->alpha;
  if (iters) *iters = c->iters;
  if (exit_reason) *exit_reason = why_rs;
  return SDPLR_OK;   // (pull_blocking has drained the stream and checked the launch)
}

// host restatement of the Sturm bisection for the Lanczos tridiagonal
int64_t sturm_below(const std::vector<double>& d, const double* e, int64_t k, double x) {
  int64_t cnt = 0;
  double q = d[0] - x;
  if (q < 0) cnt++;
  for (int64_t i = 1; i < k; i++) {
    const double den = (q == 0.0) ? std::numeric_limits<double>::min() : q;
    q = d[i] - x - e[i - 1] * e[i - 1] / den;
    if (q < 0) cnt++;
  }
  return cnt;
}

int ensure_lz_capacity(S* s, int64_t q) {
  if (q <= s->lz_cap) return SDPLR_OK;
  HIPCK(s, hipStreamSynchronize(s->stream));
  if (s->lz_alpha) pool_free(s->lz_alpha);
  if (s->lz_beta) pool_free(s->lz_beta);
  if (s->lz_graph) { (void)hipGraphExecDestroy(s->lz_graph); s->lz_graph = nullptr; }
  q = std::max<int64_t>(q, 4096);
  HIPCK(s, pool_malloc((void**)&s->lz_alpha, q * sizeof(double)));
  HIPCK(s, pool_malloc((void**)&s->lz_beta, q * sizeof(double)));
  s->lz_cap = q;
  return SDPLR_OK;
}

// y = S·x (+ low-rank) for device n-vectors; slot ≥ 0 adds the ⟨x, y⟩ partials
void enq_spmv(S* s, const double* x, double* yv, int slot, const int* stop_flag) {
  if (s->lr.ST > 0) {
    ProfScope ps(s, "lr_btx");
    k_lr_btx<<<dim3(64, s->lr.ST), SDPLR_NT, 0, s->stream>>>(s->lr, x, (int)s->n, s->lr_btx_part, stop_flag);
    k_lr_btx_finalize<<<1, SDPLR_NT, 0, s->stream>>>(s->lr, 64, s->lr_btx_part, s->y, s->lr_coef, stop_flag);
  }
  ProfScope ps(s, "spmv");
  k_spmv<<<s->nb_spmv, SDPLR_NT, 0, s->stream>>>(s->sp, x, yv, s->lr, s->lr_coef, slot, s->partials, stop_flag);
  if (s->sp.n_long_rows > 0)
    k_spmv_long<<<std::min(s->sp.n_long_rows, 256), SDPLR_NT, 0, s->stream>>>(s->sp, x, yv, s->lr, s->lr_coef, slot, s->nb_spmv, s->partials, stop_flag);
}

int run_lanczos_classic(S* s, int64_t q, const double* v0, double* alpha, double* beta, int64_t* steps) {
  const int64_t n = s->n;
  if (q > n - 1) q = n - 1;  // src/coreop.jl:465
  if (q < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "lanczos: q < 1 (needs n ≥ 2)");
  int rc = ensure_lz_capacity(s, q);
  if (rc) return rc;
  if ((rc = pull(s))) return rc;
  s->hc->lz_done = 0; s->hc->lz_steps = 0; s->hc->lz_beta_prev = 0.0;
  if ((rc = push(s))) return rc;
  HIPCK(s, hipMemcpyAsync(s->lz_v0, v0, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIPCK(s, hipMemsetAsync(s->lz_alpha, 0, q * sizeof(double), s->stream));
  HIPCK(s, hipMemsetAsync(s->lz_beta, 0, q * sizeof(double), s->stream));
  double *v = s->lz_buf[0], *Av = s->lz_buf[1], *vpre = s->lz_buf[2];
  HIPCK(s, hipMemsetAsync(vpre, 0, n * sizeof(double), s->stream));
  const int* stop = &s->ctrl->lz_done;
  {
    ProfScope ps(s, "lanczos_init");
    k_sumsq<<<s->nb_n, SDPLR_NT, 0, s->stream>>>(s->lz_v0, n, SLOT_V0, s->partials);
    k_lz_init<<<s->nb_n, SDPLR_NT, 0, s->stream>>>((int)n, s->lz_v0, v, s->nb_n, s->partials);
  }
  for (int64_t i = 0; i < q; i++) {
    enq_spmv(s, v, Av, SLOT_LZ_A, stop);                                  // :483
    {
      ProfScope ps(s, "lanczos_update");
      k_lz_update1<<<s->nb_n, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)n, (int)i, v, Av, vpre, s->lz_alpha, s->nb_spmv + std::min(s->sp.n_long_rows, 256), s->partials);
      k_lz_update2<<<s->nb_n, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)n, (int)i, Av, s->lz_beta, s->nb_n, s->partials);
    }
    double* t = vpre;  // copyto!(v_pre, v); copyto!(v, Av)  (:498-499) as a pointer rotation
    vpre = v; v = Av; Av = t;
  }
  HIPCK(s, hipGetLastError());
  HIPCK(s, hipMemcpyAsync(alpha, s->lz_alpha, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipMemcpyAsync(beta, s->lz_beta, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if ((rc = pull(s))) return rc;
  *steps = s->hc->lz_steps;
  return SDPLR_OK;
}

// A hipGraph capture failed (not: was skipped): the handle launches eagerly from now on.  Never silent — counted
// (sdplr_hip_get_stats out[1]), kept in last_error, and printed once per process.
void note_capture_failure(S* s, const char* what) {
  const hipError_t e = hipGetLastError();
  s->graph_disabled = true;
  s->st_capture_failed++;
  s->err = std::string("hipGraph capture failed (") + what + "): " + hipGetErrorString(e) + " — eager launches from now on";
  static bool said = false;
  if (!said) {
    said = true;
    fprintf(stderr, "[sdplr_hip] warning: %s\n", s->err.c_str());
  }
}

// hub-row blocks inside k_lz_band's grid: four rows per 1024-thread block, at most 64 blocks (the band plan leaves
// them room in the one resident round of blocks: build_band)
int lz_hub_blocks(const S* s) { return s->sp.n_long_rows > 0 ? std::min((s->sp.n_long_rows + 3) / 4, 64) : 0; }

// one Lanczos step = k_lz_spmv (+ hub rows) + k_lz_step on the buffer triple (uprev, u, t)
void enq_lz_step(S* s, double* uprev, double* u, double* t) {
  const int* stop = &s->ctrl->lz_done;
  if (s->lz_band_now) {   // LDS-band form (k_sparse.h, DevBand): partial t per (band, chunk), summed by the recurrence kernel
    const DevBand& bd = s->band;
    const int nbk = bd.NB * bd.NC, nbl = lz_hub_blocks(s);
    {
      ProfScope ps(s, "lz_spmv");
      // (+ 1: the grid's last block closes the previous step and does nothing else)
      // grid: nbk sweeping blocks, nbl hub-row blocks, and the last block, which only closes the previous step
      if (bd.pal_mode) k_lz_band<true><<<nbk + nbl + 1, SDPLR_LZB_NT, (size_t)(bd.BW + bd.CH) * sizeof(double), s->stream>>>(bd, (int)s->n, s->ctrl, u, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials, s->sp, nbl);
      else k_lz_band<false><<<nbk + nbl + 1, SDPLR_LZB_NT, (size_t)(bd.BW + bd.CH) * sizeof(double), s->stream>>>(bd, (int)s->n, s->ctrl, u, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials, s->sp, nbl);
    }
    ProfScope ps(s, "lz_step");
    k_lz_step_band<<<s->nb_n, SDPLR_NT, 0, s->stream>>>((int)s->n, s->ctrl, bd, nbl > 0 ? 1 : 0, uprev, u, t, s->lr, s->lr_coef, s->lr_btx_part, nbk + nbl, s->lz_alpha, s->partials);
    return;
  }
  {
    ProfScope ps(s, "lz_spmv");
    static const int lz_lpr = getenv("SDPLR_HIP_LZ_LPR") ? atoi(getenv("SDPLR_HIP_LZ_LPR")) : 8;
    static const int lz_nb = getenv("SDPLR_HIP_LZ_NB") ? atoi(getenv("SDPLR_HIP_LZ_NB")) : 768;
    const int nbv = blocks_for(s->n, SDPLR_NT / lz_lpr, lz_nb);  // more blocks only lengthen the consumer-side reductions
    s->nb_lzv = nbv;
    if (lz_lpr == 4) k_lz_spmv<4><<<nbv, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, u, t, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials);
    else if (lz_lpr == 16) k_lz_spmv<16><<<nbv, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, u, t, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials);
    else if (lz_lpr == 32) k_lz_spmv<32><<<nbv, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, u, t, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials);
    else k_lz_spmv<8><<<nbv, SDPLR_NT, 0, s->stream>>>(s->sp, s->ctrl, u, t, s->lr, s->y, s->lr_btx_part, s->nb_n, s->lr_coef, s->lz_beta, s->partials);
    if (s->sp.n_long_rows > 0)
      k_spmv_long<<<std::min(s->sp.n_long_rows, 256), SDPLR_NT, 0, s->stream>>>(s->sp, u, t, s->lr, s->lr_coef, SLOT_LZ_A, s->nb_lzv, s->partials, stop);
  }
  ProfScope ps(s, "lz_step");
  k_lz_step<<<s->nb_n, SDPLR_NT, 0, s->stream>>>((int)s->n, s->ctrl, uprev, u, t, s->lr, s->lr_btx_part, s->nb_lzv + std::min(s->sp.n_long_rows, 256), s->lz_alpha, s->partials);
}

// the structured resident Lanczos (k_rs_lanczos_ell); dual: with copy2y in front and ⟨y, b⟩ + the tridiagonal's smallest
// eigenvalue behind it — the whole of dual_obj (src/coreop.jl:376-415) in one launch
RsLzEllArgs rs_lz_ell_args(S* s, int64_t q, bool dual) {
  const int64_t n = s->n;
  RsLzEllArgs a{};
  a.lr = rs_lr(s);
  a.n = (int)n; a.q = (int)q;
  a.E = s->rs_ell; a.gid_g = s->ff.gid_g; a.row_k = s->rs_row_k; a.row_v = s->rs_row_v;
  a.yvec = s->y; a.v0 = s->lz_v0;
  a.alpha_out = s->lz_alpha; a.beta_out = s->lz_beta; a.c = s->ctrl;
  a.dual = dual ? 1 : 0; a.m = (int)s->m; a.y_rw = s->y;
  a.lam = s->lambda; a.lam_ub = s->lambda_ub; a.pv_raw = s->pv_raw; a.b = s->b;
  return a;
}
// dynamic LDS of the run: v, v_pre, Av, the diagonal — and, with unit weights and room for it, the packed columns
size_t rs_lz_ell_lds(const S* s, bool* ell_in_lds) {
  const size_t base = (size_t)4 * s->n * sizeof(double), packed = s->rs_ell_pair_lines * 64 * sizeof(unsigned);
  *ell_in_lds = s->rs_ell.val == nullptr && base + packed <= RS_LDS_MAX && getenv("SDPLR_HIP_RESIDENT_LZ_STREAM") == nullptr;
  return *ell_in_lds ? base + packed : base;
}
int enq_lanczos_ell(S* s, int64_t q, bool dual) {
  const RsLzEllArgs a = rs_lz_ell_args(s, q, dual);
  bool in_lds = false;
  const size_t lds = rs_lz_ell_lds(s, &in_lds);
  if (in_lds) {
    RS_SET_ATTR(k_rs_lanczos_ell<true>);
    k_rs_lanczos_ell<true><<<1, SDPLR_RS_NT, lds, s->stream>>>(a);
  } else {
    RS_SET_ATTR(k_rs_lanczos_ell<false>);
    k_rs_lanczos_ell<false><<<1, SDPLR_RS_NT, lds, s->stream>>>(a);
  }
  HIPCK(s, hipGetLastError());
  s->st_rs_lz++;
  return SDPLR_OK;
}

// approx_mineigval_lanczos's recurrence, src/coreop.jl:461-500 (see k_sparse.h "Lanczos recurrence")
int run_lanczos(S* s, int64_t q, const double* v0, double* alpha, double* beta, int64_t* steps,
                ApiLock* api_lock) {
  const int64_t n = s->n;
  if (s->lr.ST > SDPLR_LRMAX || getenv("SDPLR_HIP_CLASSIC_LANCZOS") != nullptr)
    return run_lanczos_classic(s, q, v0, alpha, beta, steps);
  if (q > n - 1) q = n - 1;  // src/coreop.jl:465
  if (q < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "lanczos: q < 1 (needs n ≥ 2)");
  int rc = ensure_lz_capacity(s, q);
  if (rc) return rc;
  if (rs_lanczos_ell_applies(s) || rs_lanczos_applies(s)) {   // resident route (k_resident.h): all q steps in one launch, the vectors in LDS
    HIPCK(s, hipMemcpyAsync(s->lz_v0, v0, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIPCK(s, hipMemsetAsync(s->lz_alpha, 0, q * sizeof(double), s->stream));
    HIPCK(s, hipMemsetAsync(s->lz_beta, 0, q * sizeof(double), s->stream));
    if (rs_lanczos_ell_applies(s)) {   // S(y) = y_g·A_g + Diag(d(y)) straight from the ELL of A_g and y: no assembled S is read
      if ((rc = enq_lanczos_ell(s, q, false))) return rc;
      HIPCK(s, hipMemcpyAsync(alpha, s->lz_alpha, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      HIPCK(s, hipMemcpyAsync(beta, s->lz_beta, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      if ((rc = pull_blocking(s))) return rc;
      *steps = s->hc->lz_steps;
      return SDPLR_OK;
    }
    RsLzArgs a{};
    a.n = (int)n; a.q = (int)q;
    a.colptr = s->sp.colptr; a.rowval = s->sp.rowval; a.nzval = s->sp.nzval;
    a.lr = s->lr; a.yvec = s->y; a.v0 = s->lz_v0;
    a.alpha_out = s->lz_alpha; a.beta_out = s->lz_beta; a.c = s->ctrl;
    RS_SET_ATTR(k_rs_lanczos);
    k_rs_lanczos<<<1, SDPLR_RS_NT, (size_t)3 * n * sizeof(double), s->stream>>>(a);
    HIPCK(s, hipGetLastError());
    s->st_rs_lz++;
    HIPCK(s, hipMemcpyAsync(alpha, s->lz_alpha, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCK(s, hipMemcpyAsync(beta, s->lz_beta, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if ((rc = pull(s))) return rc;
    *steps = s->hc->lz_steps;
    return SDPLR_OK;
  }
  if ((rc = pull(s))) return rc;
  DevCtrl* c = s->hc;
  c->lz_done = 0; c->lz_steps = 0; c->lz_beta_prev = 0.0; c->lz_gamma_cur = 1.0; c->lz_gamma_prev = 1.0; c->lz_qmax = q;
  if ((rc = push(s))) return rc;
  double *b0 = s->lz_buf[0], *b1 = s->lz_buf[1], *b2 = s->lz_buf[2];
  if ((rc = ensure_band(s))) return rc;
  // the palette form of the band plan takes the off-diagonal values from the live y: only when S is S(y)
  s->lz_band_now = s->use_band && (!s->band.pal_mode || s->S_from_y);
  if (s->lz_band_now) {   // S is fixed for the q steps: the band layout's values are gathered once per run
    ProfScope ps(s, "lanczos_init");
    if (s->band.pal_mode) k_lz_diag_fill<<<s->nb_n, SDPLR_NT, 0, s->stream>>>(s->band, (int)n, s->sp.nzval);
    else k_lz_band_fill<<<blocks_for((long long)(s->band.n_groups + 1) * 64, SDPLR_NT, 2048), SDPLR_NT, 0, s->stream>>>(s->band, s->sp.nzval);
  }
  HIPCK(s, hipMemcpyAsync(b1, v0, n * sizeof(double), hipMemcpyHostToDevice, s->stream));   // u_1 = v0
  HIPCK(s, hipMemsetAsync(b0, 0, n * sizeof(double), s->stream));                            // u_0 = 0
  HIPCK(s, hipMemsetAsync(s->lz_alpha, 0, q * sizeof(double), s->stream));
  HIPCK(s, hipMemsetAsync(s->lz_beta, 0, q * sizeof(double), s->stream));
  {
    ProfScope ps(s, "lanczos_init");
    k_sumsq<<<s->nb_n, SDPLR_NT, 0, s->stream>>>(b1, n, SLOT_LZ_N, s->partials);                // ‖v0‖² (:474)
    if (s->lr.ST > 0)
      k_lr_btx<<<dim3(s->nb_n, s->lr.ST), SDPLR_NT, 0, s->stream>>>(s->lr, b1, (int)n, s->lr_btx_part, nullptr);
  }
  // three steps = one rotation of (uprev, u, t): (b0,b1,b2) → (b1,b2,b0) → (b2,b0,b1)
  auto three = [&]() {
    enq_lz_step(s, b0, b1, b2);
    enq_lz_step(s, b1, b2, b0);
    enq_lz_step(s, b2, b0, b1);
  };
  // one graph = `reps` rotations (24 steps by default): a graph launch costs several µs of idle device time,
  // a step that falls through after the last one ≈ 9 µs
  int reps = 8;
  if (const char* e = getenv("SDPLR_HIP_LZ_REPS")) reps = std::max(1, std::min(atoi(e), 64));
  if (s->lz_graph && s->lz_graph_reps != reps) { (void)hipGraphExecDestroy(s->lz_graph); s->lz_graph = nullptr; }
  const int64_t rounds = (q + 1 + 2) / 3;   // q steps + the closing k_lz_spmv of step q+1
  bool use_graph = !s->prof_on && !s->graph_disabled && getenv("SDPLR_HIP_NO_GRAPH") == nullptr && (n >= (1 << 14) || s->force_graph) &&
                   s->lz_band_now == s->use_band;   // (the captured steps are the band plan's when there is one)
  if (use_graph && !s->lz_graph && api_lock) {
    api_lock->unlock();
    {
      std::unique_lock<ApiMutex> excl(g_api_rw, std::chrono::milliseconds(capture_wait_ms()));
      if (!excl.owns_lock()) {
        s->st_capture_skipped++;   // busy: this call launches eagerly, the next one tries again
      } else {
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
          for (int t = 0; t < reps; t++) three();
          ok = hipStreamEndCapture(s->stream, &graph) == hipSuccess && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&s->lz_graph, graph, nullptr, nullptr, 0) == hipSuccess;
        s->lz_graph_reps = reps;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) {
          s->st_captures++;
        } else {
          note_capture_failure(s, "Lanczos");
          s->lz_graph = nullptr;
        }
      }
    }
    api_lock->lock();
  }
  if (!s->lz_graph) use_graph = false;
  if (use_graph) {
    for (int64_t k = 0; k < rounds; k += reps) { HIPCK(s, hipGraphLaunch(s->lz_graph, s->stream)); s->st_lz_graph++; }
  } else {
    for (int64_t k = 0; k < rounds; k++) three();
    s->st_lz_eager += rounds;
  }
  HIPCK(s, hipGetLastError());
  HIPCK(s, hipMemcpyAsync(alpha, s->lz_alpha, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipMemcpyAsync(beta, s->lz_beta, q * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if ((rc = pull(s))) return rc;
  *steps = s->hc->lz_steps;
  return SDPLR_OK;
}

}  // namespace

// ================================================================================================
// operators
// ================================================================================================
extern "C" {

int32_t sdplr_hip_A(S* s, int32_t u_slot, int32_t v_slot, int32_t out_vec) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  double* U = factor_ptr(s, u_slot);
  double* V = v_slot >= 0 ? factor_ptr(s, v_slot) : nullptr;
  if (!U || (v_slot >= 0 && !V)) return fail(s, SDPLR_ERR_INVALID_ARG, "A: bad factor slot");
  double* out = out_vec == SDPLR_V_PV_RAW ? s->pv_raw : out_vec == SDPLR_V_A_RD ? s->A_RD : out_vec == SDPLR_V_A_DD ? s->A_DD : nullptr;
  if (out_vec == SDPLR_V_SCRATCH) {
    int64_t L; bool in_ctrl;
    out = vec_ptr(s, SDPLR_V_SCRATCH, &L, &in_ctrl);
  }
  if (!out) return fail(s, SDPLR_ERR_INVALID_ARG, "A: bad output vector");
  enq_A(s, U, V, V ? 1 : 0, out, nullptr, 0);
  return sync_check(s);
}
int32_t sdplr_hip_At_preprocess(S* s) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  enq_At_preprocess(s, 0);
  s->S_stale = false;
  return sync_check(s);
}
int32_t sdplr_hip_At_left(S* s, int32_t ys, int32_t xs) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  double *Y = factor_ptr(s, ys), *X = factor_ptr(s, xs);
  if (!Y || !X || Y == X) return fail(s, SDPLR_ERR_INVALID_ARG, "At_left: bad slots");
  ensure_S(s);
  enq_At_left(s, Y, X, 1.0, -1, 0);
  note_factor_written(s, ys);
  return sync_check(s);
}
int32_t sdplr_hip_At_right(S* s, const double* x, double* yh, int64_t k) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  if (!x || !yh || k < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "At_right: bad args");
  ensure_S(s);
  const int64_t n = s->n;
  for (int64_t c = 0; c < k; c++) {
    HIPCK(s, hipMemcpyAsync(s->lz_buf[0], x + c * n, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    enq_spmv(s, s->lz_buf[0], s->lz_buf[1], -1, nullptr);
    HIPCK(s, hipMemcpyAsync(yh + c * n, s->lz_buf[1], n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCK(s, hipStreamSynchronize(s->stream));
  }
  return sync_check(s);
}

int32_t sdplr_hip_At_right_device(S* s, const double* x, double* yd, int64_t k) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  if (!x || !yd || k < 1 || x == yd) return fail(s, SDPLR_ERR_INVALID_ARG, "At_right_device: bad args");
  hipPointerAttribute_t ax{}, ay{};
  if (hipPointerGetAttributes(&ax, x) != hipSuccess || hipPointerGetAttributes(&ay, yd) != hipSuccess ||
      ax.type != hipMemoryTypeDevice || ay.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    return fail(s, SDPLR_ERR_INVALID_ARG, "At_right_device: x and y must be device pointers");
  }
  ensure_S(s);
  const int64_t n = s->n;
  for (int64_t c = 0; c < k; c++) enq_spmv(s, x + c * n, yd + c * n, -1, nullptr);
  return sync_check(s);
}

int32_t sdplr_hip_get_stats(const S* s, int64_t* out, int32_t cap, int32_t* n_written) {
  if (!s || !out || cap < 0) return SDPLR_ERR_INVALID_ARG;
  const int64_t v[16] = {s->st_captures, s->st_capture_failed, s->st_capture_skipped, s->st_graph_batches,
                         s->st_eager_batches, s->st_lz_graph, s->st_lz_eager, s->st_iters, s->st_rs_loops, s->st_rs_lz, s->st_rs_fg,
                         s->st_rs_shared, s->st_pdrop, s->st_grp_loops, s->st_ring_loops, s->st_ring_materialized};
  const int32_t k = std::min<int32_t>(cap, 16);
  for (int32_t i = 0; i < k; i++) out[i] = v[i];
  if (n_written) *n_written = k;
  return SDPLR_OK;
}

int32_t sdplr_hip_f(S* s, double* L) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  enq_f(s);
  int rc = pull(s);
  if (rc) return rc;
  if (L) *L = s->hc->L;
  return sync_check(s);
}
int32_t sdplr_hip_g(S* s) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  enq_g(s, 0, false);
  s->S_stale = false;
  s->sg_stale = true;
  s->G_consistent = true;
  s->G_age = 0;
  return sync_check(s);
}
}  // extern "C"

namespace {
RsFgArgs rs_fg_args(S* s) {
  RsFgArgs a{};
  a.lr = rs_lr(s);
  a.rowvec = rs_rows_global(s) ? s->rs_rowvec : nullptr;
  a.n = (int)s->n; a.m = (int)s->m; a.r = (int)s->r;
  a.gid_g = s->ff.gid_g; a.row_k = s->rs_row_k; a.row_v = s->rs_row_v; a.E = s->rs_ell;
  a.R = aslot(s->arena, AS_R); a.G = aslot(s->arena, AS_G); a.P = aslot(s->arena, 3 + 2 * (int)s->h);
  a.y = s->y; a.pv_raw = s->pv_raw; a.pv = s->pv;
  a.lam = s->lambda; a.lam_ub = s->lambda_ub; a.lb = s->pv_lb; a.b = s->b;
  a.c = s->ctrl;
  return a;
}
int set_norm_params(S* s, double normC, double normb, int grel, int prel) {
  int rc = pull(s);
  if (rc) return rc;
  s->hc->normC = normC; s->hc->normb = normb; s->hc->grel = grel; s->hc->prel = prel;
  s->hc->done = 0; s->hc->err = 0;
  return push(s);
}
}  // namespace

extern "C" {
int32_t sdplr_hip_fg(S* s, double normC, double normb, int32_t grel, int32_t prel, double* L, double* gn, double* pn) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  int rc = set_norm_params(s, normC, normb, grel, prel);
  if (rc) return rc;
  if (rs_fg_applies(s)) {   // resident route (k_resident.h): one launch; P = A_g·R stays for the loop that follows
    const RsFgArgs a = rs_fg_args(s);
    const size_t lds = rs_loop_lds(s);
    RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_fg<VEC>)); k_rs_fg<VEC><<<1, SDPLR_RS_NT, lds, s->stream>>>(a); }))
    HIPCK(s, hipGetLastError());
    s->P_valid = true;
    s->P_age = 0;
    s->S_stale = true;   // y is current; S is assembled by whoever reads it (ensure_S)
    s->S_from_y = true;
    s->sg_stale = true;
    s->G_consistent = true;
    s->G_age = 0;
    s->st_rs_fg++;
    if ((rc = pull(s))) return rc;
    if (L) *L = s->hc->L;
    if (gn) *gn = s->hc->gnorm;
    if (pn) *pn = s->hc->pvnorm;
    return sync_check(s);
  }
  // the singleton fast path without low-rank matrices (MaxCut, CutNorm): P = A_g·R by the gather kernel on (R, R), whose
  // epilogue also leaves 𝒜(RRᵀ) of the row-attached constraints and the partials of ⟨R, P⟩ — then two m- / n-sized passes
  // instead of the generic twelve kernels (SDDMM, segmented reduction, S assembly, SpMM: ≈ 190 µs at the north-star size)
  if (s->fast && s->fast_singleton && s->lr.ST == 0 && s->use_tile && !s->tiles_deferred && s->tile_lpr == tile_shape_lpr(s) &&
      s->n * s->r * 8 < (1LL << 32) && getenv("SDPLR_HIP_NO_FAST_FG") == nullptr) {
    if ((rc = tile_lds_attr(s))) return rc;
    double *R = aslot(s->arena, AS_R), *G = aslot(s->arena, AS_G), *P = fast_P(s);
    if (!s->all_covered) HIPCK(s, hipMemsetAsync(s->A_DD, 0, (s->m + 1) * sizeof(double), s->stream));   // fill!(out, 0), src/coreop.jl:39
    {
      ProfScope ps(s, "spmm_P");
      const size_t lds = ((size_t)SDPLR_NT * s->VEC * (s->tile.K + 1) + (size_t)(SDPLR_NT / s->tile_lpr) * (2 * s->tile.K + 8)) * sizeof(double);
      if (s->tile.gdiag) { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0, true><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, R, P, P, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 0, s->lr, s->lr_part, 0))) }
      else { TLV_DISPATCH((k_spmm_tile<LPR, VEC, 0><<<s->nb_tile, SDPLR_NT, lds, s->stream>>>(s->tile, (int)s->n, (int)s->m, s->ff, R, R, P, P, (int)s->r, s->lambda, s->pv_raw, s->A_RD, s->A_DD, s->partials, s->ctrl, 0, s->lr, s->lr_part, 0))) }
    }
    {
      ProfScope ps(s, "fg_tail");
      k_fg_fast_tail<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->ff.gid_g, s->A_DD, s->pv_raw, s->b, s->pv_lb, s->pv, s->lambda, s->lambda_ub, s->y, s->nb_tile, s->partials);
    }
    {
      ProfScope ps(s, "fg_G");
      LV_DISPATCH((k_fg_fast_G<LPR, VEC><<<s->nb_spmm, SDPLR_NT, 0, s->stream>>>((int)s->n, s->ff, R, P, G, (int)s->r, s->y, s->partials)))
    }
    k_fg_fast_fin<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->pv_raw, s->nb_m, s->nb_spmm, s->partials);
    s->P_valid = true; s->P_age = 0;
    s->S_stale = true; s->S_from_y = true;   // y is current, S is assembled by whoever reads it next (ensure_S)
    s->sg_stale = true;
    s->G_consistent = true;
    s->G_age = 0;
    if ((rc = pull(s))) return rc;
    if (L) *L = s->hc->L;
    if (gn) *gn = s->hc->gnorm;
    if (pn) *pn = s->hc->pvnorm;
    return sync_check(s);
  }
  enq_f(s);
  enq_g(s, 0, false);
  s->S_stale = false;
  {
    ProfScope ps(s, "pv_norm");  // src/coreop.jl:340-347
    k_pv_norm<<<s->nb_m, SDPLR_NT, 0, s->stream>>>((int)s->m, s->pv_raw, s->pv_lb, s->pv, 1, s->partials);
    k_norms<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->nb_spmm + std::min(s->sp.n_long_rows, 256), s->nb_m, 0, 0, s->partials);
  }
  s->sg_stale = true;
  s->G_consistent = true;
  s->G_age = 0;
  if ((rc = pull(s))) return rc;
  if (L) *L = s->hc->L;
  if (gn) *gn = s->hc->gnorm;
  if (pn) *pn = s->hc->pvnorm;
  return sync_check(s);
}
int32_t sdplr_hip_norms(S* s, double normC, double normb, int32_t grel, int32_t prel, double* gn, double* pn) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  int rc = set_norm_params(s, normC, normb, grel, prel);
  if (rc) return rc;
  {
    ProfScope ps(s, "norms");
    k_sumsq<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(aslot(s->arena, AS_G), s->N, SLOT_GNORM2, s->partials);
    k_pv_norm<<<s->nb_m, SDPLR_NT, 0, s->stream>>>((int)s->m, s->pv_raw, s->pv_lb, s->pv, 0, s->partials);
    k_norms<<<1, SDPLR_NT, 0, s->stream>>>(s->ctrl, s->nb_dense, s->nb_m, 0, 0, s->partials);
  }
  if ((rc = pull(s))) return rc;
  if (gn) *gn = s->hc->gnorm;
  if (pn) *pn = s->hc->pvnorm;
  return sync_check(s);
}
int32_t sdplr_hip_axpy_R(S* s, double alpha) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  int rc = pull(s);
  if (rc) return rc;
  s->hc->alpha = alpha;
  if ((rc = push(s))) return rc;
  s->P_valid = false;
  enq_axpy_R(s, 0);
  return sync_check(s);
}
int32_t sdplr_hip_update_lambda(S* s) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW_KEEP_HISTORY(s);
  k_update_lambda<<<s->nb_m, SDPLR_NT, 0, s->stream>>>(s->ctrl, (int)s->m, s->lambda, s->lambda_ub, s->pv_raw);
  return sync_check(s);
}

// ---- L-BFGS ------------------------------------------------------------------------------------------
// lbfgs_clear! on a history in ring form: of the ring only G (the current gradient) and dirt (= s_latest; the unscaled
// direction after a step without lbfgs_update!) survive the clear — two N-sized passes instead of the whole materialisation
static int ring_drop_for_clear(S* s) {
  if (!s || !s->finalized || !s->ring_on) return SDPLR_OK;
  const int h = (int)s->h, rk = s->ring_k, j0 = s->ring_j0;
  const int pG = s->ring_unc ? ring_back(rk, -1, h) : rk;
  if (pG != 0) HIPCK(s, hipMemcpyAsync(aslot(s->arena, AS_G), ring_G(s->arena, pG, j0), s->N * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  const int nb = blocks_for(s->N, SDPLR_NT, 2048);
  if (s->dirt_from >= 0 && s->ring_n >= 1) {        // s_latest = α·D of the newest ring pair
    k_ring_dirt<<<nb, SDPLR_NT, 0, s->stream>>>(aslot(s->arena, AS_D), ring_D(s->arena, ring_back(rk, 1, h), j0), s->N, s->ring_alpha[s->dirt_from]);
    s->dirt_from = -1;
  } else if (s->dirt_from >= 0) {                   // (not one pair in ring form: s_latest is in its slot as stored)
    HIPCK(s, hipMemcpyAsync(aslot(s->arena, AS_D), aslot(s->arena, AS_S0 + s->dirt_from), s->N * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    s->dirt_from = -1;
  } else if (s->ring_unc && rk != 0) {
    HIPCK(s, hipMemcpyAsync(aslot(s->arena, AS_D), ring_D(s->arena, rk, j0), s->N * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  }
  HIPCK(s, hipGetLastError());
  HIPCK(s, hipMemsetAsync(&s->ctrl->ring_on, 0, sizeof(int), s->stream));
  if (s->hc) s->hc->ring_on = 0;
  s->ring_on = false;
  return SDPLR_OK;
}
int32_t sdplr_hip_lbfgs_clear(S* s) {
  ApiShared api_guard(dev_of(s));
  { const int rc_r = ring_drop_for_clear(s); if (rc_r) return rc_r; }
  NEED_FINAL_RW(s);
  // the 2h history slots are neighbours in the arena: one fill (a small solve clears the history once per major iteration)
  if (s->h > 0) HIPCK(s, hipMemsetAsync(aslot(s->arena, AS_S0), 0, (size_t)2 * s->h * s->arena.stride * sizeof(double), s->stream));
  if (s->lit) {
    HIPCK(s, hipMemsetAsync(s->lit_rho, 0, (size_t)s->h * sizeof(double), s->stream));
    HIPCK(s, hipMemsetAsync(s->lit_a, 0, (size_t)s->h * sizeof(double), s->stream));
  }
  int rc = pull(s);
  if (rc) return rc;
  memset(s->hc->rho, 0, sizeof s->hc->rho); memset(s->hc->a, 0, sizeof s->hc->a);
  memset(s->hc->SY, 0, sizeof s->hc->SY); memset(s->hc->YY, 0, sizeof s->hc->YY);
  memset(s->hc->Sg, 0, sizeof s->hc->Sg); memset(s->hc->Yg, 0, sizeof s->hc->Yg);
  memset(s->hc->c_alpha, 0, sizeof s->hc->c_alpha); memset(s->hc->c_gamma, 0, sizeof s->hc->c_gamma);
  s->gram_dirty = s->sg_stale = s->ynext_pending = false;  // zero vectors ⇒ zero Gram data, exact for any G
  s->hc->ring_on = 0;
  return push(s);
}
int32_t sdplr_hip_lbfgs_dir(S* s, int32_t negate, double* descent) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  ensure_gram(s);
  enq_lbfgs_dir(s, negate ? 1 : 0, 0, 0);
  if (s->h > 0) s->ynext_pending = true;
  int rc = pull(s);
  if (rc) return rc;
  if (descent) *descent = s->hc->descent;
  return sync_check(s);
}
int32_t sdplr_hip_descent_fallback(S* s) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  k_neg_copy<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(aslot(s->arena, AS_G), aslot(s->arena, AS_D), s->N);
  s->sg_stale = true;
  return sync_check(s);
}
int32_t sdplr_hip_lbfgs_update(S* s, double stepsize) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  if (s->h == 0) return SDPLR_OK;
  if (s->gram_dirty) {  // rows other than the updated one must be valid first
    s->ynext_pending = false;
    ensure_gram(s);
  }
  int rc = pull(s);
  if (rc) return rc;
  s->hc->alpha = stepsize;
  if ((rc = push(s))) return rc;
  enq_lbfgs_update(s, 0);
  enq_boundary(s, 0, 1, 0, 0);
  s->ynext_pending = false;
  s->sg_stale = false;
  s->hc_valid = false;   // (the seam kernel has just rewritten ρ, latest and the Gram data on the device)
  return sync_check(s);
}

// ---- line searches -----------------------------------------------------------------------------------
static int32_t linesearch_common(S* s, int armijo, double alpha_max, double* alpha, double* L) {
  int rc = pull(s);
  if (rc) return rc;
  s->hc->alpha_max = alpha_max; s->hc->done = 0; s->hc->err = 0; s->hc->reldelta_exit = 0;
  if ((rc = push(s))) return rc;
  enq_linesearch(s, armijo, 1, 0);   // chk = 1: a refused direction must stop the commit
  if ((rc = pull(s))) return rc;
  if (s->hc->err == SDPLR_ERR_NOT_DESCENT) {
    s->hc->err = 0; s->hc->done = 0;
    (void)push(s);
    char buf[128];
    snprintf(buf, sizeof buf, "Error: cubic[1] = %.17g should be less than 0.", s->hc->biquad[1]);
    return fail(s, SDPLR_ERR_NOT_DESCENT, buf);
  }
  if (alpha) *alpha = s->hc->alpha;
  if (L) *L = s->hc->L;
  return sync_check(s);
}
int32_t sdplr_hip_linesearch(S* s, double alpha_max, double* alpha, double* L) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  return linesearch_common(s, 0, alpha_max, alpha, L);
}
int32_t sdplr_hip_linesearch_armijo(S* s, double alpha_max, double* alpha, double* L) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  return linesearch_common(s, 1, alpha_max, alpha, L);
}

// ---- the inner loop ----------------------------------------------------------------------------------
static int32_t inner_loop_impl(S* s, double normC, double normb, int32_t grel, int32_t prel, int32_t use_armijo,
                               double cur_gtol, double fprec_eps, int64_t max_local_iters, double time_budget_s,
                               double* Lio, double* gnio, double* pnio, double* last_alpha, int64_t* iters,
                               int32_t* exit_reason, bool resume);
int32_t sdplr_hip_inner_loop(S* s, double normC, double normb, int32_t grel, int32_t prel, int32_t use_armijo,
                             double cur_gtol, double fprec_eps, int64_t max_local_iters, double time_budget_s,
                             double* Lio, double* gnio, double* pnio, double* last_alpha, int64_t* iters,
                             int32_t* exit_reason) {
  return inner_loop_impl(s, normC, normb, grel, prel, use_armijo, cur_gtol, fprec_eps, max_local_iters, time_budget_s, Lio, gnio,
                         pnio, last_alpha, iters, exit_reason, false);
}
static int32_t inner_loop_impl(S* s, double normC, double normb, int32_t grel, int32_t prel, int32_t use_armijo,
                               double cur_gtol, double fprec_eps, int64_t max_local_iters, double time_budget_s,
                               double* Lio, double* gnio, double* pnio, double* last_alpha, int64_t* iters,
                               int32_t* exit_reason, bool resume) {
  ApiLock api_lock(g_api_rw);
  bind_device(dev_of(s));
  const bool hc_was_valid = s && s->finalized && s->hc_valid;
  const bool G_was_consistent = s && s->finalized && s->G_consistent;
  // (the loop's first kernel after the seam overwrites dirt: a pending dirt ← s_latest is dropped, not made — unless the
  // loop turns out not to run a single iteration, see below)
  const int dirt_was_from = (s && s->finalized) ? s->dirt_from : -1;
  // Ring form of the history (k_dense.h): kept across calls while nothing else has touched the arena.  Decided before
  // NEED_FINAL_RW, which would turn a ring back into the stored form.
  bool ring_keep = false;
  if (s && s->finalized && s->ring_on) {
    const bool fast2_ = s->fast && s->fast_singleton && !use_armijo;
    static const int64_t refresh_iters_ = getenv("SDPLR_HIP_P_REFRESH_ITERS") ? atoll(getenv("SDPLR_HIP_P_REFRESH_ITERS")) : 256;
    // (a P-less loop due for its refresh of G writes G at its own place: not on a ring)
    const bool would_pdrop = fast2_ && step_can_drop_P(s) && s->G_consistent && (!resume || s->pdrop_now);
    const bool refresh_due = would_pdrop && !resume && s->G_age >= refresh_iters_;
    ring_keep = fast2_ && ring_shape_common(s) && !refresh_due && !s->ring_unc && Lio && gnio && pnio &&
                max_local_iters >= 1 && !rs_loop_applies(s, use_armijo) &&
                !(s->gram_dirty || s->ynext_pending || s->sg_stale);
    if (ring_keep) s->ring_on = false;   // (NEED_FINAL_RW leaves it alone; set again below)
  }
  if (s && s->finalized && max_local_iters >= 1 && Lio && gnio && pnio) s->dirt_from = -1;
  NEED_FINAL_RW(s);
  if (ring_keep) s->ring_on = true;
  if (!Lio || !gnio || !pnio || max_local_iters < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "inner_loop: bad args");
  const bool gram_work = s->h > 0 && (s->gram_dirty || s->ynext_pending || s->sg_stale);
  ensure_gram(s);
  int rc = rs_loop_applies(s, use_armijo) ? SDPLR_OK : tile_lds_attr(s);
  if (rc) return rc;
  s->hc_valid = hc_was_valid && !gram_work;   // (nothing has been enqueued since the shadow was last in step)
  if ((rc = pull_if_stale(s))) return rc;
  s->hc_valid = false;
  DevCtrl* c = s->hc;
  c->done = 0; c->exit_reason = 0; c->err = 0; c->use_armijo = use_armijo;
  c->iters = 0; c->max_iters = max_local_iters; c->reldelta_exit = 0; c->norms_pending = 0; c->pv2_extra = 0.0;
  c->cur_gtol = cur_gtol; c->fprec_eps = fprec_eps; c->normC = normC; c->normb = normb;
  c->grel = grel; c->prel = prel;
  c->L = *Lio; c->gnorm = *gnio; c->pvnorm = *pnio; c->alpha = 0.0; c->alpha_max = 1.0;
  // (no wait: the kernels are ordered behind the copy on the stream, and the host does not touch its shadow again before
  // the closing snapshot has been waited for)
  HIPCK(s, hipMemcpyAsync(s->ctrl, s->hc, sizeof(DevCtrl), hipMemcpyHostToDevice, s->stream));
  // Iterations are enqueued in batches — a captured hipGraph of `graph_iters` passes of the while
  // body, or eager launches when per-kernel event timing is on.  The device decides every exit;
  // once `done` is set the remaining kernels of a batch fall through.  The control block is
  // snapshotted after each batch, and batch k+1 is already queued while the host inspects the
  // snapshot of batch k, so the GPU never waits for the host.
  if (rs_loop_applies(s, use_armijo)) {
    // resident route (k_resident.h): the whole while loop is ONE launch of one workgroup; every exit — the time
    // budget too — is taken on the device, the host reads the control block once
    s->dirt_from = dirt_was_from;   // (the resident kernel leaves dirt itself: a pending copy from a multi-launch loop is made first)
    if ((rc = ensure_dirt(s))) return rc;
    return run_resident_loop(s, time_budget_s, false, false, Lio, gnio, pnio, last_alpha, iters, exit_reason, nullptr, G_was_consistent, resume);
  }
  const bool fastp = s->fast;
  const bool fast2 = fastp && s->fast_singleton && !use_armijo;
  static const int64_t refresh_iters = getenv("SDPLR_HIP_P_REFRESH_ITERS") ? atoll(getenv("SDPLR_HIP_P_REFRESH_ITERS")) : 256;
  // the P-less step kernel carries G forward: only from a G known to be the gradient at the device's state, and only for
  // so many steps before G is formed from scratch again (every fg! does that anyway)
  // (resume — SDPLR_MAJOR_RESUME — continues a capped loop: the kernel the capped call took, and nothing refreshed by age, so
  // that cap + resume ≡ one uncapped call bit for bit)
  s->pdrop_now = fast2 && step_can_drop_P(s) && G_was_consistent && (!resume || s->pdrop_now);
  if (s->pdrop_now && !resume && s->G_age >= refresh_iters) {
    enq_g(s, 0, false);          // g! from scratch (src/coreop.jl:305-317): same y, S assembled, G = 2·R·S
    s->S_stale = false;
    s->G_age = 0;
  }
  // (a ring is entered on whatever history is stored: k_dense.h)
  s->ring_now = fast2 && (s->pdrop_now ? ring_shape_ok(s) : ring_shape_pb_ok(s));
  if (getenv("SDPLR_HIP_DEBUG"))
    fprintf(stderr, "[sdplr_hip] inner_loop: ring %d (kept %d, pdrop %d, shape %d: can_drop %d n_extra %d VEC %d LPR %d tile %d/%d nb_tile %d)\n",
            (int)s->ring_now, (int)ring_keep, (int)s->pdrop_now, (int)ring_shape_ok(s), (int)step_can_drop_P(s), s->n_extra,
            s->VEC, s->LPR, (int)s->use_tile, s->tile_lpr, s->nb_tile);
  if (s->ring_on && !s->ring_now) {   // (a ring kept for a loop that turns out not to take it)
    if ((rc = ensure_canonical(s))) return rc;
  }
  {
    const bool fresh = s->ring_now && !s->ring_on;
    const int was = c->ring_on;
    if (fresh) {   // G and dirt at their own places are position 0; every pair is still in its slot as stored
      c->ring_k = 0; c->ring_n = 0; c->ring_unc = 0;
      c->ring_j0 = c->latest % (int)s->h;
      memset(c->ring_alpha, 0, sizeof c->ring_alpha);
    }
    c->ring_on = s->ring_now ? 1 : 0;
    s->ring_on = s->ring_now;
    // (the control block went to the device above, before the route was known: the ring's scalars follow it)
    if (fresh || was != c->ring_on)
      HIPCK(s, hipMemcpyAsync(&s->ctrl->ring_on, &c->ring_on, sizeof(DevCtrl) - offsetof(DevCtrl, ring_on), hipMemcpyHostToDevice, s->stream));
  }
  const int ar = use_armijo ? 1 : (s->ring_now ? (s->pdrop_now ? 3 : 4) : (s->pdrop_now ? 2 : 0));
  const bool edgep = !fastp && edge_applies(s, use_armijo);
  auto enq_iter = [&]() {
    if (fast2) enq_iteration_fast2(s);
    else if (fastp) enq_iteration_fast(s, use_armijo);
    else if (edgep) enq_iteration_edge(s);
    else enq_iteration(s, use_armijo);
  };
  // the structured loops leave y current and S unassembled: said before the first enqueue, so that an error return on the
  // way cannot leave a stale S marked as assembled
  if (fastp || edgep) { s->S_stale = true; s->S_from_y = true; }
  if (s->pdrop_now) {
    s->P_valid = false;          // P is neither read nor kept up to date by this loop
  } else if (fastp) {
    if (!s->P_valid || (!resume && s->P_age >= refresh_iters)) {
      enq_fast_refresh_P(s);
      s->P_valid = true;
      s->P_age = 0;
    }
  }
  // hipGraph batches pay for their capture (milliseconds, and exclusive of every other handle's HIP calls) only
  // on instances whose solve is long: small factors (config 5's n = 800 batch: 64 solves 1.3 → 0.95 s) stay eager
  bool use_graph = !s->prof_on && !s->graph_disabled && max_local_iters >= 4 && getenv("SDPLR_HIP_NO_GRAPH") == nullptr &&
                   (s->N >= (1LL << 17) || s->force_graph);
  if (use_graph && !s->graph_exec[ar]) {
    // Capture is fragile on ROCm 7.2 when OTHER host threads issue HIP calls meanwhile (observed: 8 handles
    // driven by 8 threads → "operation failed due to a previous error during capture", any capture mode).
    // So: one capture at a time, under the process-wide lock taken exclusively — with a bounded wait (other
    // handles may sit in their own inner loops for seconds): no lock in time ⇒ this call runs eagerly and the
    // next call tries again.  A capture that FAILS is not an error either, but it is recorded
    // (note_capture_failure) and this handle launches eagerly for good.
    api_lock.unlock();
    {
      std::unique_lock<ApiMutex> excl(g_api_rw, std::chrono::milliseconds(capture_wait_ms()));
      if (!excl.owns_lock()) {
        s->st_capture_skipped++;
      } else {
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
          for (int i = 0; i < s->graph_iters; i++) enq_iter();
          ok = hipStreamEndCapture(s->stream, &graph) == hipSuccess && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&s->graph_exec[ar], graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) {
          s->st_captures++;
        } else {
          note_capture_failure(s, "inner loop");
          s->graph_exec[ar] = nullptr;
        }
      }
    }
    api_lock.lock();
  }
  if (!s->graph_exec[ar]) use_graph = false;
  // the time budget starts here: after any wait for the capture lock
  const auto t0 = std::chrono::steady_clock::now();
  const int64_t eager_batch = 8;
  // The plan never runs past the iteration budget: whole graph batches while they fit, the remainder as eager
  // launches, then ONE loop-test seam (it sets done / EXIT_ITERS, src/sdplr.jl:272-277) — no batch of kernels that
  // could only fall through, and nothing speculative behind the last planned batch.
  int64_t planned = 0;
  bool plan_closed = false;
  const bool dbg = getenv("SDPLR_HIP_DEBUG") != nullptr;
  double t_enq = 0.0, t_wait = 0.0;
  int n_batches = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto launch_batch = [&](int slot) -> int {
    const double ta = now();
    n_batches++;
    const int64_t left = max_local_iters - planned;
    if (use_graph && left >= s->graph_iters) {
      HIPCK(s, hipGraphLaunch(s->graph_exec[ar], s->stream));
      planned += s->graph_iters;
      s->st_graph_batches++;
    } else if (left > 0) {
      const int64_t nb_it = std::min<int64_t>(left, eager_batch);
      for (int64_t i = 0; i < nb_it; i++) enq_iter();
      HIPCK(s, hipGetLastError());
      planned += nb_it;
      s->st_eager_batches++;
    }
    if (planned >= max_local_iters && !plan_closed) {
      const bool lf = fastp ? step_fuses_update(s) : spmm_fuses_update(s);
      s->gram_nb = !lf ? s->nb_upd : (fastp ? s->nb_step : spmm_upd_blocks(s));
      enq_boundary(s, 0, 1, 1, 0);       // fold the last update, then the loop tests: the budget exit fires here
      plan_closed = true;
    }
    HIPCK(s, hipMemcpyAsync(s->snap[slot], s->ctrl, sizeof(DevCtrl), hipMemcpyDeviceToHost, s->stream));
    HIPCK(s, hipEventRecord(s->snap_ev[slot], s->stream));
    t_enq += now() - ta;
    return SDPLR_OK;
  };
  int why = -1;
  int final_snap = -1;   // the snapshot that holds the closed plan's final control block, if the device got that far
  if (s->prof_on) {
    // per-kernel event timing: no speculation, and the trailing loop-test pass (whose kernels fall
    // through) is left untimed, so that the averages are over real launches only
    int64_t passes = 0;
    for (;;) {
      const int64_t nb = std::min<int64_t>(8, max_local_iters - passes);
      for (int64_t i = 0; i < nb; i++) enq_iter();
      passes += std::max<int64_t>(nb, 0);
      if (nb <= 0) {
        s->prof_on = false;
        enq_iter();
        s->prof_on = true;
      }
      HIPCK(s, hipGetLastError());
      if ((rc = pull(s))) return rc;
      if (c->done) break;
      if (time_budget_s > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_budget_s) {
        why = EXIT_TIME;
        break;
      }
    }
  } else {
    int cur = 0;
    if ((rc = launch_batch(cur))) return rc;
    for (;;) {
      const bool more = !plan_closed;
      if (more && (rc = launch_batch(cur ^ 1))) return rc;  // speculative: queued behind batch `cur`
      const double tw = now();
      HIPCK(s, hipEventSynchronize(s->snap_ev[cur]));
      t_wait += now() - tw;
      if (s->snap[cur]->done) {
        // (usable as the final state only if nothing was enqueued behind it — no speculative batch — and it was taken
        // behind the closing seam)
        if (!more && plan_closed) final_snap = cur;
        break;
      }
      if (time_budget_s > 0) {
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > time_budget_s) { why = EXIT_TIME; break; }
      }
      if (!more) break;   // (the closing seam always sets done; not reached)
      cur ^= 1;
    }
  }
  // fold the partials of a last lbfgs_update (time-budget exit); a replayed graph does not pass through the
  // enqueue functions, so the producer's partial count is set here, not inherited
  const bool loop_fused = fastp ? step_fuses_update(s) : spmm_fuses_update(s);   // lbfgs_update! rode another kernel
  s->gram_nb = !loop_fused ? s->nb_upd : (fastp ? s->nb_step : spmm_upd_blocks(s));
  if (fastp || edgep) { s->S_stale = true; s->S_from_y = true; }   // y is current, S is assembled by whoever reads it next (ensure_S)
  if (final_snap >= 0) {
    // the device closed the plan itself: the closing seam (Gram fold + loop tests) ran in front of this snapshot, nothing
    // has been enqueued behind it — it IS the control block (one launch, one copy and one wait less per call)
    memcpy(s->hc, s->snap[final_snap], sizeof(DevCtrl));
    s->hc_valid = true;
  } else {
    enq_boundary(s, 0, 1, 0, 0);
    if ((rc = pull(s))) return rc;
  }
  if (c->done) why = c->exit_reason;   // the device's verdict wins over the host's time check
  if (loop_fused && c->iters > 0 && c->err == 0 && why != EXIT_RELDELTA) {
    // the kernels lbfgs_update! rides on leave `dirt *= α` (src/lbfgs.jl:142) to a copy dirt ← s_latest, made when
    // something looks at dirt (ensure_dirt)
    s->dirt_from = c->latest - 1;
  } else if (c->iters == 0) {
    s->dirt_from = dirt_was_from;   // not one iteration ran: dirt is what it was
  }
  if (dbg)
    fprintf(stderr, "[sdplr_hip] inner_loop: %d batches (%s), host enqueue %.3f ms, host wait %.3f ms, iters %lld\n",
            n_batches, use_graph ? "graph" : "eager", 1e3 * t_enq, 1e3 * t_wait, (long long)c->iters);
  // G was rewritten by g! after the last update when the loop left through the relative-decrease
  // exit (no lbfgs_update!, src/sdplr.jl:239-241): the dots with G are stale then.
  s->sg_stale = (why == EXIT_RELDELTA);
  s->ynext_pending = (why == EXIT_RELDELTA) && s->h > 0;
  if (c->err == SDPLR_ERR_NOT_DESCENT) {
    c->err = 0; c->done = 0;
    (void)push(s);
    return fail(s, SDPLR_ERR_NOT_DESCENT, "Error: cubic[1] should be less than 0.");
  }
  s->st_iters += c->iters;
  s->P_age += c->iters;
  s->G_age = s->pdrop_now ? s->G_age + c->iters : 0;   // (the P-based kernels form G from P and y at every step)
  if (s->pdrop_now) s->st_pdrop++;
  if (s->ring_now) {   // the ring as the loop left it (ensure_canonical / lbfgs_clear work from these)
    s->st_ring_loops++;
    s->ring_k = c->ring_k; s->ring_n = c->ring_n; s->ring_unc = c->ring_unc; s->ring_latest = c->latest; s->ring_j0 = c->ring_j0;
    memcpy(s->ring_alpha, c->ring_alpha, sizeof s->ring_alpha);
  }
  s->G_consistent = true;                              // g! has run at the loop's last point
  *Lio = c->L; *gnio = c->gnorm; *pnio = c->pvnorm;
  if (last_alpha) *last_alpha = c->alpha;
  if (iters) *iters = c->iters;
  if (exit_reason) *exit_reason = why;
  return sync_check(s);
}

// One major iteration's device work as ONE call: [λ update (src/sdplr.jl:358-362)] → σ → lbfgs_clear! (:384) → fg! (:389)
// → the inner while loop (:190-278) on what fg! returned.  On the resident route the five are one launch; otherwise this
// is the sequence of the entry points above.
int32_t sdplr_hip_major_iteration(S* s, double normC, double normb, int32_t grel, int32_t prel, int32_t use_armijo,
                                  int32_t update_lambda, double sigma, double cur_gtol, double fprec_eps,
                                  int64_t max_local_iters, double time_budget_s, double* L, double* gn, double* pn,
                                  double* last_alpha, int64_t* iters, int32_t* exit_reason) {
  if (!s) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "null handle");
  if (!L || !gn || !pn || max_local_iters < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "major_iteration: bad args");
  if (update_lambda == SDPLR_MAJOR_RESUME)   // the while loop continues where a capped call left it: ℒ and the norms are in/out
    return inner_loop_impl(s, normC, normb, grel, prel, use_armijo, cur_gtol, fprec_eps, max_local_iters, time_budget_s, L, gn, pn,
                           last_alpha, iters, exit_reason, true);
  bool resident = false;
  {
    ApiShared api_guard(dev_of(s));
    { const int rc_r = ring_drop_for_clear(s); if (rc_r) return rc_r; }   // (lbfgs_clear! follows on every route)
    NEED_FINAL_RW(s);
    resident = rs_loop_applies(s, use_armijo) && rs_fg_applies(s);
    if (resident) {
      // var.σ[], the loop's parameters and its counters ride the launch's arguments (the kernel writes them into its copy
      // of the control block after the λ update has read the σ it found there): no pull / push of the block in front
      RsLoopIn in{};
      in.sigma = sigma; in.gtol = cur_gtol; in.fprec = fprec_eps; in.normC = normC; in.normb = normb;
      in.grel = grel; in.prel = prel; in.max_iters = max_local_iters;
      s->sg_stale = s->ynext_pending = false;   // (cleared history: see sdplr_hip_lbfgs_clear)
      return run_resident_loop(s, time_budget_s, update_lambda != 0, true, L, gn, pn, last_alpha, iters, exit_reason, &in);
    }
  }
  int32_t rc;
  if (update_lambda && (rc = sdplr_hip_update_lambda(s))) return rc;
  if ((rc = sdplr_hip_set_scalar(s, SDPLR_S_SIGMA, sigma))) return rc;
  if ((rc = sdplr_hip_lbfgs_clear(s))) return rc;
  if ((rc = sdplr_hip_fg(s, normC, normb, grel, prel, L, gn, pn))) return rc;
  if (last_alpha) *last_alpha = 0.0;
  if (iters) *iters = 0;
  if (exit_reason) *exit_reason = EXIT_GTOL;
  if (!(*gn > cur_gtol)) return SDPLR_OK;           // src/sdplr.jl:190
  return sdplr_hip_inner_loop(s, normC, normb, grel, prel, use_armijo, cur_gtol, fprec_eps, max_local_iters, time_budget_s,
                              L, gn, pn, last_alpha, iters, exit_reason);
}

// ---- Lanczos / dual bound ------------------------------------------------------------------------------
int32_t sdplr_hip_lanczos(S* s, int64_t q, const double* v0, double* alpha, double* beta, int64_t* steps) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW_KEEP_HISTORY(s);
  if (!v0 || !alpha || !beta || !steps || q < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "lanczos: bad args");
  ensure_S(s);
  return run_lanczos(s, q, v0, alpha, beta, steps, &api_guard.l);
}
int32_t sdplr_hip_tridiag_mineig(const double* alpha, const double* beta, int64_t k, double* out) {
  if (!alpha || !out || k < 1 || (k > 1 && !beta)) return SDPLR_ERR_INVALID_ARG;
  if (k == 1) { *out = (alpha[0] + 1) - 1; return SDPLR_OK; }  // src/coreop.jl:505-507
  std::vector<double> d(k);
  for (int64_t i = 0; i < k; i++) d[i] = alpha[i] + 1;          // shift by I  (:502)
  double lo = std::numeric_limits<double>::infinity(), hi = -lo;
  for (int64_t i = 0; i < k; i++) {
    const double rad = (i > 0 ? std::fabs(beta[i - 1]) : 0.0) + (i + 1 < k ? std::fabs(beta[i]) : 0.0);
    lo = std::min(lo, d[i] - rad);
    hi = std::max(hi, d[i] + rad);
  }
  for (int it = 0; it < 200; it++) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_below(d, beta, k, mid) >= 1) hi = mid; else lo = mid;
  }
  *out = 0.5 * (lo + hi) - 1;                                   // cancel the shift (:513)
  return SDPLR_OK;
}
static int32_t approx_mineig_impl(S* s, int64_t q, const double* v0, double* mineig, ApiLock* api_lock) {
  std::vector<double> al(q), be(q);
  int64_t steps = 0;
  int rc = run_lanczos(s, q, v0, al.data(), be.data(), &steps, api_lock);
  if (rc) return rc;
  return sdplr_hip_tridiag_mineig(al.data(), be.data(), steps, mineig);
}
int32_t sdplr_hip_approx_mineigval_lanczos(S* s, int64_t q, const double* v0, double* mineig) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW_KEEP_HISTORY(s);
  if (!v0 || !mineig || q < 1) return fail(s, SDPLR_ERR_INVALID_ARG, "approx_mineigval_lanczos: bad args");
  ensure_S(s);
  return approx_mineig_impl(s, q, v0, mineig, &api_guard.l);
}
int32_t sdplr_hip_dual_obj(S* s, double trace_bound, int64_t iter, const double* v0, double* dual_value, double* mineig) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW_KEEP_HISTORY(s);
  if (!v0) return fail(s, SDPLR_ERR_INVALID_ARG, "dual_obj: null v0");
  if (rs_lanczos_ell_applies(s) && s->n >= 2 && getenv("SDPLR_HIP_NO_FUSED_DUAL") == nullptr) {
    // resident route: copy2y (:384), the q Lanczos steps on S(y) (:461-500), the tridiagonal's smallest eigenvalue
    // (:502-513) and ⟨y, b⟩ (:412) are ONE launch; S itself is assembled (:385) only if somebody reads it later
    const double itd = (double)std::max<int64_t>(iter, 100);
    int64_t q = (int64_t)(2 * std::ceil(std::pow(itd, 0.5) * std::log((double)s->n)));   // :402
    q = std::min<int64_t>(q, s->n - 1);                                                    // :465
    int rc = ensure_lz_capacity(s, q);
    if (rc) return rc;
    HIPCK(s, hipMemcpyAsync(s->lz_v0, v0, s->n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if ((rc = enq_lanczos_ell(s, q, true))) return rc;
    s->S_stale = true;
    s->S_from_y = true;
    if ((rc = pull_blocking(s))) return rc;
    const double ev = s->hc->lz_mineig;
    if (dual_value) *dual_value = -s->hc->descent + trace_bound * std::min(ev, 0.0);        // :412
    if (mineig) *mineig = ev;
    return SDPLR_OK;
  }
  enq_copy2y(s, 0);              // src/coreop.jl:384
  s->S_from_y = true;
  if (rs_lanczos_ell_applies(s)) {
    s->S_stale = true;           // the structured Lanczos takes S from y: S is assembled (:385) only if somebody reads it
  } else {
    enq_At_preprocess(s, 0);     // :385
    s->S_stale = false;
  }
  const double it = (double)std::max<int64_t>(iter, 100);
  const int64_t eig_iter = (int64_t)(2 * std::ceil(std::pow(it, 0.5) * std::log((double)s->n)));  // :402
  double ev = 0.0;
  int rc = approx_mineig_impl(s, eig_iter, v0, &ev, &api_guard.l);
  if (rc) return rc;
  k_dot<<<s->nb_m, SDPLR_NT, 0, s->stream>>>((int)s->m, s->y, s->b, SLOT_DUALYB, s->partials);
  k_reduce_slot<<<1, SDPLR_NT, 0, s->stream>>>(&s->ctrl->descent, SLOT_DUALYB, s->nb_m, s->partials);
  if ((rc = pull(s))) return rc;
  const double dv = -s->hc->descent + trace_bound * std::min(ev, 0.0);  // :412
  if (dual_value) *dual_value = dv;
  if (mineig) *mineig = ev;
  return sync_check(s);
}

// ---- high-precision eigen path ---------------------------------------------------------------------------
}  // extern "C"
namespace {
// eigen-decomposition of a small dense symmetric matrix (cyclic Jacobi): A (m×m, row-major) is destroyed,
// w gets the eigenvalues ascending, Z (m×m) the eigenvectors as ROWS (Z[i] ↔ w[i])
void jacobi_eigh(std::vector<double>& A, int m, std::vector<double>& w, std::vector<double>& Z) {
  std::vector<double> Vv((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++) Vv[(size_t)i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < m; i++)
      for (int j = 0; j < m; j++) (i == j ? diag : off) += A[(size_t)i * m + j] * A[(size_t)i * m + j];
    if (off <= 1e-30 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < m - 1; p++)
      for (int q = p + 1; q < m; q++) {
        const double apq = A[(size_t)p * m + q];
        if (apq == 0.0) continue;
        const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < m; k++) {   // A ← A·J
          const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
          A[(size_t)k * m + p] = c * akp - sn * akq;
          A[(size_t)k * m + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < m; k++) {   // A ← Jᵀ·A
          const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
          A[(size_t)p * m + k] = c * apk - sn * aqk;
          A[(size_t)q * m + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < m; k++) {   // eigenvectors as columns of Vv
          const double vkp = Vv[(size_t)k * m + p], vkq = Vv[(size_t)k * m + q];
          Vv[(size_t)k * m + p] = c * vkp - sn * vkq;
          Vv[(size_t)k * m + q] = sn * vkp + c * vkq;
        }
      }
  }
  std::vector<int> order(m);
  for (int i = 0; i < m; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * m + a] < A[(size_t)b * m + b]; });
  w.resize(m);
  Z.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++) {
    w[i] = A[(size_t)order[i] * m + order[i]];
    for (int k = 0; k < m; k++) Z[(size_t)i * m + k] = Vv[(size_t)k * m + order[i]];
  }
}
}  // namespace
extern "C" {

// SDP_S_eigval, src/coreop.jl:351-374 — see include/sdplr_hip.h
int32_t sdplr_hip_S_eigval(S* s, int64_t nev, int32_t which, int64_t ncv_in, double tol, int64_t maxiter,
                           const double* v0, double* evals, int64_t* n_matvec, int64_t* n_converged) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  const int64_t n = s->n;
  if (!evals || nev < 1 || nev > n || (which != 0 && which != 1) || maxiter < 1)
    return fail(s, SDPLR_ERR_INVALID_ARG, "S_eigval: bad args");
  ensure_S(s);
  int m = (int)std::min<int64_t>(n, std::max<int64_t>(ncv_in > 0 ? ncv_in : 100, std::min<int64_t>(n, 2 * nev + 1)));
  if (m > 256) m = 256;
  if (nev > m) return fail(s, SDPLR_ERR_INVALID_ARG, "S_eigval: nev exceeds the basis size");
  const double eps = std::numeric_limits<double>::epsilon();
  if (!(tol > 0)) tol = eps;                       // ARPACK: tol = 0 ⇒ machine precision
  const long long ldv = (n + 31) / 32 * 32;
  const int nb = blocks_for(n, SDPLR_NT, 256);
  // scratch of this call: two bases [m + 1][ldv], the column of T being formed, β's, the rotation matrix
  double *V = nullptr, *V2 = nullptr, *part = nullptr, *h = nullptr, *Tdev = nullptr, *bdev = nullptr, *Ydev = nullptr;
  auto cleanup = [&]() { for (double* p : {V, V2, part, h, Tdev, bdev, Ydev}) dfree(s, p); };
  int rc;
#define EIG_ALLOC(ptr, cnt) if ((rc = dzero(s, &ptr, (size_t)(cnt)))) { cleanup(); return rc; }
  EIG_ALLOC(V, (size_t)(m + 1) * ldv) EIG_ALLOC(V2, (size_t)(m + 1) * ldv) EIG_ALLOC(part, (size_t)(m + 1) * nb)
  EIG_ALLOC(h, m + 1) EIG_ALLOC(Tdev, (size_t)m * m) EIG_ALLOC(bdev, m) EIG_ALLOC(Ydev, (size_t)m * m)
#undef EIG_ALLOC
  {  // start vector: the caller's v0 (the reference lets ARPACK draw one), or a fixed pseudo-random one
    std::vector<double> st(n);
    if (v0) std::copy(v0, v0 + n, st.begin());
    else { unsigned long long z = 0x9E3779B97F4A7C15ull; for (int64_t i = 0; i < n; i++) { z = z * 6364136223846793005ull + 1442695040888963407ull; st[i] = (double)(z >> 11) / 9007199254740992.0 - 0.5; } }
    hipError_t e = hipMemcpyAsync(V2, st.data(), n * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { cleanup(); return fail(s, SDPLR_ERR_HIP, std::string("S_eigval: ") + hipGetErrorString(e)); }
    k_sumsq<<<nb, SDPLR_NT, 0, s->stream>>>(V2, n, SLOT_V0, s->partials);
    k_eig_scale<<<nb, SDPLR_NT, 0, s->stream>>>(V2, (int)n, SLOT_V0, nb, s->partials, bdev, 0.0, V);
  }
  std::vector<double> T((size_t)m * m, 0.0), theta, Y, beta(m), Tcols((size_t)m * m);
  int k = 0;                 // vectors kept from the previous cycle (V[0..k) Ritz vectors, V[k] the new direction)
  int64_t matvecs = 0, nconv = 0;
  double* w = V + (size_t)m * ldv;   // row m of the basis: the vector being orthogonalised
  for (int64_t cycle = 0; cycle < maxiter; cycle++) {
    for (int j = k; j < m; j++) {
      double* vj = V + (size_t)j * ldv;
      w = V + (size_t)(j + 1) * ldv;
      enq_spmv(s, vj, w, -1, nullptr);                                     // w = S·v_j  (𝒜t!(y, aux, x, var), :364)
      double* tcol = Tdev + (size_t)j * m;
      for (int pass = 0; pass < 2; pass++) {                               // classical Gram–Schmidt, twice
        k_eig_proj<<<nb, SDPLR_NT, 0, s->stream>>>(V, ldv, j + 1, w, (int)n, part);
        k_eig_reduce<<<(j + 1 + 3) / 4, SDPLR_NT, 0, s->stream>>>(j + 1, nb, part, h, tcol, pass);
        k_eig_axpy<<<nb, SDPLR_NT, (size_t)(j + 1) * sizeof(double), s->stream>>>(V, ldv, j + 1, h, w, (int)n);
      }
      k_sumsq<<<nb, SDPLR_NT, 0, s->stream>>>(w, n, SLOT_V0, s->partials);
      k_eig_scale<<<nb, SDPLR_NT, 0, s->stream>>>(w, (int)n, SLOT_V0, nb, s->partials, bdev + j, 1e-300, w);
      matvecs++;
    }
    hipError_t e = hipMemcpyAsync(Tcols.data(), Tdev, (size_t)m * m * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(beta.data(), bdev, m * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { cleanup(); return fail(s, SDPLR_ERR_HIP, std::string("S_eigval: ") + hipGetErrorString(e)); }
    // T = Vᵀ·S·V: columns k..m−1 were projected in this cycle; the leading k×k block is diag(θ) of the last one
    double tnorm = 0.0;
    for (int j = k; j < m; j++)
      for (int i = 0; i <= j; i++) T[(size_t)i * m + j] = T[(size_t)j * m + i] = Tcols[(size_t)j * m + i];
    for (int i = 0; i < m; i++) tnorm = std::max(tnorm, std::fabs(T[(size_t)i * m + i]));
    // an invariant subspace (β at round-off level) ends the Krylov space there: exact eigenvalues of the block
    int meff = m;
    for (int j = k; j < m - 1; j++)
      if (beta[j] <= 1e-13 * std::max(tnorm, 1e-300)) { meff = j + 1; break; }
    std::vector<double> Tw((size_t)meff * meff);
    for (int i = 0; i < meff; i++)
      for (int j = 0; j < meff; j++) Tw[(size_t)i * meff + j] = T[(size_t)i * m + j];
    jacobi_eigh(Tw, meff, theta, Y);
    const double bres = meff == m ? beta[m - 1] : 0.0;
    auto idx = [&](int i) { return which == 0 ? i : meff - 1 - i; };       // i-th wanted Ritz pair
    nconv = 0;
    for (int i = 0; i < (int)std::min<int64_t>(nev, meff); i++) {
      const double res = std::fabs(bres * Y[(size_t)idx(i) * meff + (meff - 1)]);
      // ARPACK's test on the shifted operator S + I the reference hands over (:365-366)
      if (res <= tol * std::max(std::pow(eps, 2.0 / 3.0), std::fabs(theta[idx(i)] + 1.0))) nconv++;
      else break;
    }
    if (nconv >= nev || meff < m || meff == n || cycle + 1 == maxiter) {
      for (int i = 0; i < (int)std::min<int64_t>(nev, meff); i++) evals[i] = theta[idx(i)];
      for (int64_t i = meff; i < nev; i++) evals[i] = std::numeric_limits<double>::quiet_NaN();
      if (meff < m || meff == n) nconv = std::min<int64_t>(nev, meff);      // exact: the Krylov space closed
      break;
    }
    // thick restart: keep the knew wanted-end Ritz vectors, then continue from the residual direction V[m]
    const int knew = std::max((int)nev + 1, std::min(m - 1, (int)nev + (m - (int)nev) / 2));
    std::vector<double> Yk((size_t)knew * m);
    for (int i = 0; i < knew; i++)
      for (int l = 0; l < m; l++) Yk[(size_t)i * m + l] = Y[(size_t)idx(i) * m + l];
    e = hipMemcpyAsync(Ydev, Yk.data(), Yk.size() * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e != hipSuccess) { cleanup(); return fail(s, SDPLR_ERR_HIP, std::string("S_eigval: ") + hipGetErrorString(e)); }
    k_eig_rotate<<<nb, SDPLR_NT, 0, s->stream>>>(V, ldv, m, Ydev, knew, V2, (int)n);
    e = hipMemcpyAsync(V2 + (size_t)knew * ldv, V + (size_t)m * ldv, n * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);   // (Yk lives on this frame)
    if (e != hipSuccess) { cleanup(); return fail(s, SDPLR_ERR_HIP, std::string("S_eigval: ") + hipGetErrorString(e)); }
    std::swap(V, V2);
    std::fill(T.begin(), T.end(), 0.0);
    for (int i = 0; i < knew; i++) T[(size_t)i * m + i] = theta[idx(i)];
    k = knew;
  }
  if (n_matvec) *n_matvec = matvecs;
  if (n_converged) *n_converged = nconv;
  rc = sync_check(s);
  cleanup();
  return rc;
}

// dot of two factor-shaped arrays on the device (err6 = dot(Rt, Rt·S), src/coreop.jl:449)
int32_t sdplr_hip_factor_dot(S* s, int32_t slot_a, int32_t slot_b, double* out) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL_RW(s);
  double *a = factor_ptr(s, slot_a), *b = factor_ptr(s, slot_b);
  if (!a || !b || !out) return fail(s, SDPLR_ERR_INVALID_ARG, "factor_dot: bad args");
  k_dot_flat<<<s->nb_dense, SDPLR_NT, 0, s->stream>>>(a, b, s->N, SLOT_DUALYB, s->partials);
  k_reduce_slot<<<1, SDPLR_NT, 0, s->stream>>>(&s->ctrl->descent, SLOT_DUALYB, s->nb_dense, s->partials);
  int rc = pull(s);
  if (rc) return rc;
  *out = s->hc->descent;
  return sync_check(s);
}

// ---- profiling ---------------------------------------------------------------------------------------
int32_t sdplr_hip_profile_enable(S* s, int32_t on) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  prof_drain(s);
  for (auto& p : s->prof) p = ProfEntry();
  s->prof_on = on != 0;
  return SDPLR_OK;
}
int32_t sdplr_hip_profile_filter(S* s, const char* name) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  s->prof_filter = name ? name : "";
  return SDPLR_OK;
}
int32_t sdplr_hip_profile_count(const S* s, int32_t* n_entries) {
  if (!s || !n_entries) return SDPLR_ERR_INVALID_ARG;
  *n_entries = (int32_t)s->prof_names.size();
  return SDPLR_OK;
}
int32_t sdplr_hip_profile_get(S* s, int32_t idx, char* name, int32_t cap, int64_t* launches, double* ms) {
  ApiShared api_guard(dev_of(s));
  NEED_FINAL(s);
  if (idx < 0 || idx >= (int32_t)s->prof_names.size()) return fail(s, SDPLR_ERR_INVALID_ARG, "profile_get: bad index");
  prof_drain(s);
  if (name && cap > 0) snprintf(name, cap, "%s", s->prof_names[idx].c_str());
  if (launches) *launches = s->prof[idx].launches;
  if (ms) *ms = s->prof[idx].ms;
  return SDPLR_OK;
}

}  // extern "C"

// ---- batches of small instances in lockstep (include/sdplr_hip.h) -----------------------------------------------------
// The resident instances of a batch that share a kernel shape go out as ONE launch, block b ↔ instance b: an argument table
// up (one copy from a pinned block), one grid, a result table back (one copy), one wait.  What the single-instance entry
// points write into / read out of each control block around their launch rides the table rows instead, so no per-instance
// HIP call is made at all.  Everything else in the batch is served by the single-instance entry point.
namespace {
struct BatchBuf {   // one pinned block + one device block of ARENA_CHUNK bytes + a blocking event, from the pools
  char* host = nullptr;
  char* dev = nullptr;
  hipEvent_t ev = nullptr;
  hipError_t init() {
    hipError_t e = pool_host_chunk((void**)&host);
    if (e == hipSuccess) e = pool_malloc((void**)&dev, ARENA_CHUNK);
    if (e == hipSuccess) e = pool_event(&ev);
    return e;
  }
  ~BatchBuf() {
    if (host) pool_host_chunk_free(host);
    if (dev) pool_free(dev);
    if (ev) pool_event_free(ev);
  }
};
constexpr size_t BATCH_ALIGN = 256;
size_t batch_up(size_t x) { return (x + BATCH_ALIGN - 1) & ~(BATCH_ALIGN - 1); }
using ShapeKey = std::pair<int, int>;   // (0, bytes per piece of a row / 8): the two shapes of the resident kernels

// one grid dereferences every handle's arrays and runs on the first handle's stream: all handles on ONE device
bool batch_one_device(const std::vector<const S*>& hs, int* dev) {
  *dev = -1;
  for (const S* s : hs) {
    if (!s) continue;
    if (*dev < 0) *dev = s->device;
    else if (s->device != *dev) return false;
  }
  return true;
}
bool batch_handles_distinct(const std::vector<const S*>& hs) {
  std::vector<const S*> v;
  for (const S* s : hs) if (s) v.push_back(s);
  std::sort(v.begin(), v.end());
  return std::adjacent_find(v.begin(), v.end()) == v.end();
}
// table up → launch → results back → wait; `launch` enqueues the grid on `st`
// INVARIANT the shared grid relies on: it runs on the FIRST handle's stream and touches the arenas and control blocks of
// all the others with no event dependency on their streams.  That is correct because every entry point of this library
// drains its handle's stream before it returns (pull / pull_blocking / the D2H copies of the results), so no handle of
// the batch has work in flight when a batch call starts, and this function itself waits for the grid before returning.
// An entry point that returned with work still enqueued would have to record an event here for each participant.
template <typename Launch>
int batch_round_trip(S* s0, BatchBuf& bb, size_t table_bytes, size_t res_off, size_t res_bytes, Launch&& launch) {
  hipStream_t st = s0->stream;
  HIPCK(s0, hipMemcpyAsync(bb.dev, bb.host, table_bytes, hipMemcpyHostToDevice, st));
  launch(st);
  HIPCK(s0, hipGetLastError());
  HIPCK(s0, hipMemcpyAsync(bb.host + res_off, bb.dev + res_off, res_bytes, hipMemcpyDeviceToHost, st));
  HIPCK(s0, hipEventRecord(bb.ev, st));
  HIPCK(s0, hipEventSynchronize(bb.ev));
  return SDPLR_OK;
}
// the items a batch call serves through the single-instance entry point: on a few host threads (each handle has its own
// stream), so that a batch of instances of the multi-launch routes overlaps as the threaded driver's would
template <typename F>
void run_singles(const std::vector<int>& idx, F&& one) {
  const size_t nt = std::min<size_t>(idx.size(), 8);
  if (nt <= 1) {
    for (int i : idx) one(i);
    return;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++)
    th.emplace_back([&] {
      (void)hipSetDevice(dev);   // (a new thread starts on device 0)
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= idx.size()) break;
        one(idx[k]);
      }
    });
  for (auto& x : th) x.join();
}

// ---- a group of edge-path instances behind shared launches (k_group.h) ------------------------------------------------
// What the group shares: every launch dimension and template argument of the seven kernels of enq_iteration_edge.
bool edge_group_applies(const S* s, const sdplr_hip_major_item& q) {
  const bool off = getenv("SDPLR_HIP_NO_GROUP_LAUNCH") != nullptr;
  if (off || !s || !s->finalized || s->fast || s->lit || s->prof_on || s->h < 1 || s->HM != 4) return false;
  if (q.update_lambda == SDPLR_MAJOR_RESUME || q.max_local_iters < 1) return false;
  if (!edge_applies(s, q.use_armijo) || rs_loop_applies(s, q.use_armijo) || !s->all_covered) return false;
  return s->edge_lr_fused && s->r <= (int64_t)s->LPR * s->VEC;
}
std::vector<long long> edge_group_key(const S* s) {
  const int nb_c = blocks_for(s->m + 1, SDPLR_NT, 1024);
  const int nb_big = s->sp.n_big_pos > 0 ? blocks_for(s->sp.n_big_pos, SDPLR_NT, 256) : 0;
  return {s->LPR, s->VEC, s->h, s->r, s->nb_dense, s->nb_edge, nb_c, nb_big, s->nb_spmm, std::min(s->sp.n_long_rows, 256), s->lr.ST};
}
struct GroupMember {
  sdplr_hip_major_item* q;
  int dirt_was_from;
};
// The inner while loops (src/sdplr.jl:190-278) of `mem` — ≥ 2 instances of one key, each with ℒ, ‖∇ℒ‖, ‖pv‖ of its fg! in
// its item — as shared launches on the first member's stream.  Entry and exit per member are those of inner_loop_impl on
// the edge path (control block in, pending Gram rows, closing seam, bookkeeping); the caller holds the shared API lock.
int run_edge_group(std::vector<GroupMember>& mem) {
  const int B = (int)mem.size();
  S* s = mem[0].q->s;               // the leader: its stream carries the shared launches, its shape is everybody's
  hipStream_t st = s->stream;
  for (GroupMember& g : mem) {      // ---- entry, per member
    sdplr_hip_major_item& q = *g.q;
    S* sk = q.s;
    const bool hc_was_valid = sk->hc_valid;
    g.dirt_was_from = sk->dirt_from;
    sk->dirt_from = -1;             // (the direction kernel overwrites dirt: a pending dirt ← s_latest is dropped)
    sk->hc_valid = false;
    sk->G_consistent = false;
    const bool gram_work = sk->gram_dirty || sk->ynext_pending || sk->sg_stale;
    ensure_gram(sk);
    sk->hc_valid = hc_was_valid && !gram_work;
    int rc = pull_if_stale(sk);
    if (rc) return rc;
    sk->hc_valid = false;
    DevCtrl* c = sk->hc;
    c->done = 0; c->exit_reason = 0; c->err = 0; c->use_armijo = 0;
    c->iters = 0; c->max_iters = q.max_local_iters; c->reldelta_exit = 0; c->norms_pending = 0; c->pv2_extra = 0.0;
    c->cur_gtol = q.cur_gtol; c->fprec_eps = q.fprec_eps; c->normC = q.normC; c->normb = q.normb;
    c->grel = q.gtol_relative; c->prel = q.ptol_relative;
    c->L = q.lagrangian; c->gnorm = q.grad_norm; c->pvnorm = q.primal_vio_norm; c->alpha = 0.0; c->alpha_max = 1.0;
    if ((rc = push(sk))) return rc;   // (waits: the shared launches run on another stream)
    sk->S_stale = true; sk->S_from_y = true;
  }
  // ---- the argument tables: one row per member and kernel, one copy up
  BatchBuf bb;
  hipEvent_t ev2 = nullptr;
  if (bb.init() != hipSuccess || pool_event(&ev2) != hipSuccess) return fail(s, SDPLR_ERR_ALLOC, "batch_major_iteration: staging blocks");
  struct EvFree { hipEvent_t e; ~EvFree() { if (e) pool_event_free(e); } } ev2_free{ev2};
  size_t off = 0;
  auto region = [&](size_t bytes) { const size_t o = off; off += batch_up(bytes); return o; };
  const size_t o_seam = region(B * sizeof(GrpSeamArgs)), o_dir = region(B * sizeof(GrpDirArgs)), o_sd = region(B * sizeof(GrpSddmmArgs)),
               o_sum = region(B * sizeof(GrpSumsArgs)), o_ls = region(B * sizeof(GrpLsArgs)), o_step = region(B * sizeof(GrpStepArgs)),
               o_sp = region(B * sizeof(GrpSpmmArgs)), o_ctrl = region(B * sizeof(DevCtrl*));
  const size_t table_bytes = off;
  const size_t o_poll = region(2 * B * sizeof(int));
  if (off > ARENA_CHUNK) return fail(s, SDPLR_ERR_INVALID_ARG, "batch_major_iteration: group too large for one staging block");
  const int nb_c = blocks_for(s->m + 1, SDPLR_NT, 1024);
  const int nb_big = s->sp.n_big_pos > 0 ? blocks_for(s->sp.n_big_pos, SDPLR_NT, 256) : 0;
  const int nbl = std::min(s->sp.n_long_rows, 256);
  const int nout = 2 * s->lr.ST * (int)s->r;
  for (int b = 0; b < B; b++) {
    S* k = mem[b].q->s;
    double *R = aslot(k->arena, AS_R), *G = aslot(k->arena, AS_G), *D = aslot(k->arena, AS_D);
    GrpSeamArgs& a0 = reinterpret_cast<GrpSeamArgs*>(bb.host + o_seam)[b];
    a0 = GrpSeamArgs{k->ctrl, (int)k->h, 0, 1, 1, 1, spmm_upd_blocks(k), k->partials, 1, (int)k->m, k->pv_raw};
    GrpDirArgs& a1 = reinterpret_cast<GrpDirArgs*>(bb.host + o_dir)[b];
    a1 = GrpDirArgs{k->ctrl, k->arena, k->N, (int)k->h, 1, 1, 2, k->partials};
    GrpSddmmArgs& a2 = reinterpret_cast<GrpSddmmArgs*>(bb.host + o_sd)[b];
    a2 = GrpSddmmArgs{k->sp, k->lr, (int)k->m, (int)k->r, R, D, k->A_RD, k->A_DD, k->lr_part, k->partials, k->ctrl};
    GrpSumsArgs& a3 = reinterpret_cast<GrpSumsArgs*>(bb.host + o_sum)[b];
    a3 = GrpSumsArgs{nout, k->nb_edge, nb_c, (int)k->m, k->n_edge_extra, k->lr_part, k->lr_W, k->red10, k->edge_extra,
                     k->lambda, k->pv_raw, k->A_RD, k->A_DD, k->partials, k->ctrl};
    GrpLsArgs& a4 = reinterpret_cast<GrpLsArgs*>(bb.host + o_ls)[b];
    a4 = GrpLsArgs{k->ctrl, (int)k->m, k->sp.big_gid, k->n_edge_extra, nb_c, k->lr.ST, (int)k->r, 1, 1, k->lr.n_lr, k->edge_extra,
                   k->A_RD, k->A_DD, k->lambda, k->lambda_ub, k->pv_raw, k->pv_lb, k->pv, k->y, k->lr.col_gid, k->lr.Dcat, k->lr_W,
                   k->lr_WS, k->partials, k->lr.mat_ptr, k->lr.mat_gid, k->red10, k->edge_extra_head};
    GrpStepArgs& a5 = reinterpret_cast<GrpStepArgs*>(bb.host + o_step)[b];
    a5 = GrpStepArgs{k->sp, k->ctrl, R, D, k->N, k->nb_dense, nb_c, (int)k->m, k->n_edge_extra, k->nb_spmm + nbl, k->edge_extra,
                     k->pv_raw, k->A_RD, k->A_DD, k->pv_lb, k->pv, k->y, k->lambda, k->lambda_ub, k->partials};
    GrpSpmmArgs& a6 = reinterpret_cast<GrpSpmmArgs*>(bb.host + o_sp)[b];
    a6 = GrpSpmmArgs{k->sp, k->lr, k->arena, R, G, (int)k->r, SLOT_GNORM2, nbl, (int)k->h, 2.0, k->lr_WS, k->partials, k->ctrl, D};
    reinterpret_cast<DevCtrl**>(bb.host + o_ctrl)[b] = k->ctrl;
    k->gram_nb = spmm_upd_blocks(k);
  }
  HIPCK(s, hipMemcpyAsync(bb.dev, bb.host, table_bytes, hipMemcpyHostToDevice, st));
  const dim3 g1(1, B), g_dir(s->nb_dense, B), g_sd(s->nb_edge, B), g_sum(nout + 2 + nb_c, B), g_step(s->nb_dense + nb_c + nb_big, B),
      g_sp(s->nb_spmm + nbl, B);
  auto enq_iter = [&]() {
    k_grp_boundary<<<g1, 1024, 0, st>>>(reinterpret_cast<const GrpSeamArgs*>(bb.dev + o_seam));
    k_grp_dir<4><<<g_dir, SDPLR_NT, 0, st>>>(reinterpret_cast<const GrpDirArgs*>(bb.dev + o_dir));
    LV_DISPATCH((k_grp_sddmm_edge<LPR, VEC, 1><<<g_sd, SDPLR_NT, 0, st>>>(reinterpret_cast<const GrpSddmmArgs*>(bb.dev + o_sd))))
    k_grp_edge_sums<<<g_sum, SDPLR_NT, 0, st>>>(reinterpret_cast<const GrpSumsArgs*>(bb.dev + o_sum));
    k_grp_ls_solve_fast<<<g1, SDPLR_LSF_NT, 0, st>>>(reinterpret_cast<const GrpLsArgs*>(bb.dev + o_ls));
    k_grp_edge_step<<<g_step, SDPLR_NT, 0, st>>>(reinterpret_cast<const GrpStepArgs*>(bb.dev + o_step));
    LV_DISPATCH((k_grp_spmm_both_upd<LPR, VEC><<<g_sp, SDPLR_NT, 5 * 4 * SDPLR_NT * sizeof(double), st>>>(reinterpret_cast<const GrpSpmmArgs*>(bb.dev + o_sp))))
  };
  hipEvent_t ev[2] = {bb.ev, ev2};
  const bool dbg = getenv("SDPLR_HIP_DEBUG") != nullptr;
  double t_enq = 0.0, t_wait = 0.0;
  int n_batches = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto launch_batch = [&](int slot) -> int {
    const double ta = now();
    n_batches++;
    for (int i = 0; i < 8; i++) enq_iter();
    int* dpoll = reinterpret_cast<int*>(bb.dev + o_poll) + slot * B;
    k_grp_poll<<<1, 64, 0, st>>>(reinterpret_cast<const DevCtrl* const*>(bb.dev + o_ctrl), B, dpoll);
    HIPCK(s, hipGetLastError());
    HIPCK(s, hipMemcpyAsync(bb.host + o_poll + slot * B * sizeof(int), dpoll, B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCK(s, hipEventRecord(ev[slot], st));
    t_enq += now() - ta;
    return SDPLR_OK;
  };
  // (the device decides every exit — the iteration budget too, at the seam — except the time budget: the smallest one given
  // stops the group, and whoever is still running leaves with EXIT_TIME as on the single-instance route)
  double budget = 0.0;
  for (GroupMember& g : mem) if (g.q->time_budget_s > 0) budget = budget > 0 ? std::min(budget, g.q->time_budget_s) : g.q->time_budget_s;
  const auto t0 = std::chrono::steady_clock::now();
  bool timed_out = false;
  int rc, cur = 0;
  if ((rc = launch_batch(cur))) return rc;
  for (;;) {
    if ((rc = launch_batch(cur ^ 1))) return rc;   // speculative: queued behind batch `cur`
    const double tw = now();
    HIPCK(s, hipEventSynchronize(ev[cur]));
    t_wait += now() - tw;
    const int* dn = reinterpret_cast<const int*>(bb.host + o_poll) + cur * B;
    bool all = true;
    for (int b = 0; b < B; b++) all = all && dn[b] != 0;
    if (all) break;
    if (budget > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget) { timed_out = true; break; }
    cur ^= 1;
  }
  HIPCK(s, hipStreamSynchronize(st));
  if (dbg)
    fprintf(stderr, "[sdplr_hip] edge group of %d: %d batches of 8 iterations, host enqueue %.3f ms, host wait %.3f ms, wall %.3f ms\n", B,
            n_batches, 1e3 * t_enq, 1e3 * t_wait, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  int first_rc = SDPLR_OK;
  for (GroupMember& g : mem) {      // ---- exit, per member
    sdplr_hip_major_item& q = *g.q;
    S* sk = q.s;
    sk->gram_nb = spmm_upd_blocks(sk);
    sk->S_stale = true; sk->S_from_y = true;
    enq_boundary(sk, 0, 1, 0, 0);
    if ((rc = pull(sk))) { q.status = rc; if (!first_rc) first_rc = rc; continue; }
    DevCtrl* c = sk->hc;
    int why = c->done ? c->exit_reason : (timed_out ? EXIT_TIME : -1);
    if (c->iters > 0 && c->err == 0 && why != EXIT_RELDELTA) sk->dirt_from = c->latest - 1;
    else if (c->iters == 0) sk->dirt_from = g.dirt_was_from;
    sk->sg_stale = (why == EXIT_RELDELTA);
    sk->ynext_pending = (why == EXIT_RELDELTA) && sk->h > 0;
    if (c->err == SDPLR_ERR_NOT_DESCENT) {
      c->err = 0; c->done = 0;
      (void)push(sk);
      q.status = fail(sk, SDPLR_ERR_NOT_DESCENT, "Error: cubic[1] should be less than 0.");
      if (!first_rc) first_rc = q.status;
      continue;
    }
    sk->st_iters += c->iters;
    sk->P_age += c->iters;
    sk->G_age = 0;
    sk->G_consistent = true;
    sk->st_grp_loops++;
    q.lagrangian = c->L; q.grad_norm = c->gnorm; q.primal_vio_norm = c->pvnorm; q.last_alpha = c->alpha; q.obj = c->obj;
    q.iters_done = c->iters; q.exit_reason = why;
    q.status = SDPLR_OK;
  }
  return first_rc;
}
}  // namespace

extern "C" {

int32_t sdplr_hip_batch_fg(int32_t count, sdplr_hip_fg_item* it) {
  if (count < 0 || (count > 0 && !it)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_fg: bad args");
  std::vector<int> single;
  // an item keeps SDPLR_ERR_UNSERVED until it has actually been served: a call that returns early (argument checks, staging
  // blocks) must not leave OK beside zeroed outputs
  for (int i = 0; i < count; i++) it[i].status = SDPLR_ERR_UNSERVED;
  {
    std::vector<const S*> hs;
    for (int i = 0; i < count; i++) hs.push_back(it[i].s);
    if (!batch_handles_distinct(hs)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_fg: a handle appears twice");   // (before any handle is dereferenced)
    int dev = -1;
    if (!batch_one_device(hs, &dev)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_fg: the handles live on different devices");
    ApiShared api_guard(dev);
    std::map<ShapeKey, std::vector<int>> groups;
    for (int i = 0; i < count; i++) {
      S* s = it[i].s;
      if (s) s->G_consistent = false;
      if (s && s->finalized && rs_fg_applies(s)) groups[{0, rs_vec(s)}].push_back(i);
      else single.push_back(i);
    }
    const size_t max_rows = ARENA_CHUNK / (batch_up(sizeof(RsFgArgs)) + 64);
    for (auto& g : groups) {
      std::vector<int>& idx = g.second;
      if (idx.size() < 2) { single.insert(single.end(), idx.begin(), idx.end()); continue; }
      for (size_t lo = 0; lo < idx.size(); lo += max_rows) {
        const size_t nb = std::min(max_rows, idx.size() - lo);
        S* s = it[idx[lo]].s;   // (the shape all rows share: RS_VEC_DISPATCH reads the parity of s->r)
        BatchBuf bb;
        if (bb.init() != hipSuccess) {
          for (size_t k = 0; k < nb; k++) it[idx[lo + k]].status = SDPLR_ERR_ALLOC;
          return fail(s, SDPLR_ERR_ALLOC, "batch_fg: staging blocks");
        }
        RsFgArgs* tab = reinterpret_cast<RsFgArgs*>(bb.host);
        const size_t res_off = batch_up(nb * sizeof(RsFgArgs));
        size_t lds = 0;
        for (size_t k = 0; k < nb; k++) {
          const sdplr_hip_fg_item& q = it[idx[lo + k]];
          S* sk = q.s;
          sk->hc_valid = false;
          RsFgArgs a = rs_fg_args(sk);
          a.in_set = 1; a.in_normC = q.normC; a.in_normb = q.normb; a.in_grel = q.gtol_relative; a.in_prel = q.ptol_relative;
          a.out = reinterpret_cast<double*>(bb.dev + res_off) + 4 * k;
          tab[k] = a;
          lds = std::max(lds, rs_loop_lds(sk));
        }
        const RsFgArgs* dtab = reinterpret_cast<const RsFgArgs*>(bb.dev);
        int rc = batch_round_trip(s, bb, nb * sizeof(RsFgArgs), res_off, nb * 4 * sizeof(double), [&](hipStream_t st) {
          RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_fg_batch<VEC>)); k_rs_fg_batch<VEC><<<(int)nb, SDPLR_RS_NT, lds, st>>>(dtab); }))
        });
        const double* res = reinterpret_cast<const double*>(bb.host + res_off);
        for (size_t k = 0; k < nb; k++) {
          sdplr_hip_fg_item& q = it[idx[lo + k]];
          S* sk = q.s;
          q.status = rc;
          if (rc) { if (sk != s) sk->err = s->err; continue; }
          sk->P_valid = true; sk->P_age = 0; sk->S_stale = true; sk->S_from_y = true; sk->sg_stale = true;
          sk->G_consistent = true; sk->G_age = 0;
          sk->st_rs_fg++; sk->st_rs_shared++;
          q.lagrangian = res[4 * k]; q.grad_norm = res[4 * k + 1]; q.primal_vio_norm = res[4 * k + 2]; q.obj = res[4 * k + 3];
        }
      }
    }
  }
  run_singles(single, [&](int i) {
    sdplr_hip_fg_item& q = it[i];
    if (!q.s) { q.status = SDPLR_ERR_INVALID_ARG; return; }
    q.status = sdplr_hip_fg(q.s, q.normC, q.normb, q.gtol_relative, q.ptol_relative, &q.lagrangian, &q.grad_norm, &q.primal_vio_norm);
    if (!q.status) q.status = sdplr_hip_get_scalar(q.s, SDPLR_S_OBJ, &q.obj);
  });
  for (int i = 0; i < count; i++) if (it[i].status) return it[i].status;
  return SDPLR_OK;
}

int32_t sdplr_hip_batch_major_iteration(int32_t count, sdplr_hip_major_item* it) {
  if (count < 0 || (count > 0 && !it)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_major_iteration: bad args");
  std::vector<int> single;
  std::map<std::vector<long long>, std::vector<int>> egroups;   // edge-path instances that can share their launches (k_group.h)
  int dev = -1;
  // an item keeps SDPLR_ERR_UNSERVED until it has actually been served: a call that returns early (argument checks, staging
  // blocks) must not leave OK beside zeroed outputs
  for (int i = 0; i < count; i++) it[i].status = SDPLR_ERR_UNSERVED;
  {
    std::vector<const S*> hs;
    for (int i = 0; i < count; i++) hs.push_back(it[i].s);
    if (!batch_handles_distinct(hs)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_major_iteration: a handle appears twice");   // (before any handle is dereferenced)
    if (!batch_one_device(hs, &dev)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_major_iteration: the handles live on different devices");
    ApiShared api_guard(dev);
    std::map<ShapeKey, std::vector<int>> groups;
    std::vector<char> was_cons(count, 0);
    for (int i = 0; i < count; i++) {
      S* s = it[i].s;
      const bool was_consistent = s && s->G_consistent;
      was_cons[i] = was_consistent ? 1 : 0;
      if (s) s->G_consistent = false;
      const bool resume = it[i].update_lambda == SDPLR_MAJOR_RESUME;
      // (fg! runs in the prologue: G is fresh, the P-less loop applies; a resumed loop takes the kernel it was capped in)
      const bool pd = s && rs_can_drop_P(s) && (!resume || (was_consistent && s->pdrop_now));
      if (s && s->finalized && it[i].max_local_iters >= 1 && rs_loop_applies(s, it[i].use_armijo) && rs_fg_applies(s)) {
        groups[{(pd ? 1 : 0) + 10 * (pd ? rs_team_w(s) : 1), rs_vec(s)}].push_back(i);   // (kernel, team size | shape)
      } else {
        if (s) s->G_consistent = was_consistent;   // (the single-instance entry point looks at it itself)
        if (edge_group_applies(s, it[i])) egroups[edge_group_key(s)].push_back(i);
        else single.push_back(i);
      }
    }
    for (auto& g : egroups)
      if (g.second.size() < 2) { single.insert(single.end(), g.second.begin(), g.second.end()); g.second.clear(); }
    for (auto& g : groups) {
      std::vector<int>& idx = g.second;
      const bool g_pd = g.first.first % 10 != 0;
      const int g_w = g.first.first / 10;          // workgroups per instance (a team: k_resident.h, TEAM)
      // (a team's members must all be resident at once: one 512-thread workgroup per CU, 256 CUs)
      const size_t max_rows = std::min<size_t>(ARENA_CHUNK / (batch_up(sizeof(RsLoopArgs)) + 128 + (g_w > 1 ? SDPLR_RS_XCH_DOUBLES * sizeof(double) : 0)),
                                               g_w > 1 ? (size_t)(192 / g_w) : (size_t)1 << 30);
      if (idx.size() < 2) {   // a lone shape: the single-instance entry point (which looks at G_consistent itself)
        for (int i : idx) it[i].s->G_consistent = was_cons[i] != 0;
        single.insert(single.end(), idx.begin(), idx.end());
        continue;
      }
      for (size_t lo = 0; lo < idx.size(); lo += max_rows) {
        const size_t nb = std::min(max_rows, idx.size() - lo);
        S* s = it[idx[lo]].s;
        BatchBuf bb;
        if (bb.init() != hipSuccess) {
          for (size_t k = 0; k < nb; k++) it[idx[lo + k]].status = SDPLR_ERR_ALLOC;
          return fail(s, SDPLR_ERR_ALLOC, "batch_major_iteration: staging blocks");
        }
        RsLoopArgs* tab = reinterpret_cast<RsLoopArgs*>(bb.host);
        // (teams: each instance's exchange block rides the staging block — zeroed here, copied up with the table)
        const size_t xch_off = batch_up(nb * sizeof(RsLoopArgs));
        const size_t xch_bytes = g_w > 1 ? nb * SDPLR_RS_XCH_DOUBLES * sizeof(double) : 0;
        const size_t res_off = batch_up(xch_off + xch_bytes);
        if (xch_bytes) memset(bb.host + xch_off, 0, xch_bytes);
        size_t lds = 0;
        for (size_t k = 0; k < nb; k++) {
          const sdplr_hip_major_item& q = it[idx[lo + k]];
          S* sk = q.s;
          sk->hc_valid = false;
          if (sk->dirt_from >= 0) {   // (an instance that came over from the multi-launch route with dirt ← s_latest pending)
            if (ensure_dirt(sk) == SDPLR_OK) (void)hipStreamSynchronize(sk->stream);
          }
          const bool resume = q.update_lambda == SDPLR_MAJOR_RESUME;
          if (!resume) sk->sg_stale = sk->ynext_pending = false;   // (cleared history: see sdplr_hip_lbfgs_clear)
          RsLoopArgs a = resume ? rs_loop_args(sk, q.time_budget_s, !sk->P_valid, false, false)
                                : rs_loop_args(sk, q.time_budget_s, true, q.update_lambda != 0, true);
          RsLoopIn in{};
          in.sigma = q.sigma; in.gtol = q.cur_gtol; in.fprec = q.fprec_eps; in.normC = q.normC; in.normb = q.normb;
          in.grel = q.gtol_relative; in.prel = q.ptol_relative; in.max_iters = q.max_local_iters;
          rs_loop_set_in(a, in);
          a.out = reinterpret_cast<double*>(bb.dev + res_off) + 8 * k;
          if (g_w > 1) {
            a.team_w = g_w;
            a.xch = reinterpret_cast<double*>(bb.dev + xch_off) + k * SDPLR_RS_XCH_DOUBLES;
            a.team_test_fail = getenv("SDPLR_HIP_TEAM_TEST_FAIL") != nullptr;
          }
          tab[k] = a;
          lds = std::max(lds, rs_loop_lds(sk));
        }
        const RsLoopArgs* dtab = reinterpret_cast<const RsLoopArgs*>(bb.dev);
        int rc = batch_round_trip(s, bb, g_w > 1 ? xch_off + xch_bytes : nb * sizeof(RsLoopArgs), res_off, nb * 8 * sizeof(double), [&](hipStream_t st) {
          if (g_w > 1) { RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_team_batch<VEC, 4>)); k_rs_team_batch<VEC, 4><<<8 * g_w * (((int)nb + 7) / 8), SDPLR_RS_NT, lds, st>>>(dtab, (int)nb, g_w); })) }
          else if (g_pd) { RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_loop_batch<VEC, 4, true>)); k_rs_loop_batch<VEC, 4, true><<<(int)nb, SDPLR_RS_NT, lds, st>>>(dtab); })) }
          else { RS_VEC_DISPATCH(s, ({ RS_SET_ATTR((k_rs_loop_batch<VEC, 4, false>)); k_rs_loop_batch<VEC, 4, false><<<(int)nb, SDPLR_RS_NT, lds, st>>>(dtab); })) }
        });
        const double* res = reinterpret_cast<const double*>(bb.host + res_off);
        for (size_t k = 0; k < nb; k++) {
          sdplr_hip_major_item& q = it[idx[lo + k]];
          S* sk = q.s;
          q.status = rc;
          if (rc) { if (sk != s) sk->err = s->err; continue; }
          const double* o = res + 8 * k;
          const int why = (int)o[6], err = (int)o[7];
          const int64_t iters = (int64_t)o[5];
          if (err == SDPLR_ERR_TEAM_PLACEMENT) {   // not one XCD for this team: nothing was touched — the single-instance route, without teams
            sk->no_team = true;
            sk->G_consistent = was_cons[idx[lo + k]] != 0;
            q.status = SDPLR_ERR_UNSERVED;
            single.push_back(idx[lo + k]);
            continue;
          }
          if (err == SDPLR_ERR_TEAM_TIMEOUT) {
            q.status = fail(sk, SDPLR_ERR_HIP, "resident loop: a team member never arrived at a barrier (the grid was not co-resident)");
            continue;
          }
          // (the bookkeeping of run_resident_loop)
          const bool pd = g_pd, resumed = q.update_lambda == SDPLR_MAJOR_RESUME;
          sk->pdrop_now = pd;
          if (pd) sk->st_pdrop++;
          sk->P_valid = !pd; sk->P_age = (resumed ? sk->P_age : 0) + iters; sk->S_stale = true; sk->S_from_y = true;
          sk->G_consistent = true; sk->G_age = pd ? (resumed ? sk->G_age : 0) + iters : 0;
          if (!resumed) { sk->st_rs_fg++; sk->gram_dirty = false; }
          sk->st_rs_loops++; sk->st_rs_shared++;
          sk->sg_stale = (why == EXIT_RELDELTA);
          sk->ynext_pending = (why == EXIT_RELDELTA) && sk->h > 0;
          if (err == SDPLR_ERR_NOT_DESCENT) {
            int r2 = pull(sk);
            if (!r2) { sk->hc->err = 0; sk->hc->done = 0; r2 = push(sk); }
            q.status = r2 ? r2 : fail(sk, SDPLR_ERR_NOT_DESCENT, "Error: cubic[1] should be less than 0.");
            continue;
          }
          sk->st_iters += iters;
          q.lagrangian = o[0]; q.grad_norm = o[1]; q.primal_vio_norm = o[2]; q.last_alpha = o[3]; q.obj = o[4];
          q.iters_done = iters; q.exit_reason = why;
        }
      }
    }
  }
  // edge-path groups: [λ update] → σ → lbfgs_clear! → fg! per instance through the single-instance entry points (a dozen launches
  // each, once per major iteration), then the while loops of those that enter it (src/sdplr.jl:190) as shared launches
  for (auto& g : egroups) {
    const std::vector<int>& idx = g.second;
    if (idx.empty()) continue;
    run_singles(idx, [&](int i) {
      sdplr_hip_major_item& q = it[i];
      int32_t rc = SDPLR_OK;
      if (q.update_lambda) rc = sdplr_hip_update_lambda(q.s);
      if (!rc) rc = sdplr_hip_set_scalar(q.s, SDPLR_S_SIGMA, q.sigma);
      if (!rc) rc = sdplr_hip_lbfgs_clear(q.s);
      if (!rc) rc = sdplr_hip_fg(q.s, q.normC, q.normb, q.gtol_relative, q.ptol_relative, &q.lagrangian, &q.grad_norm, &q.primal_vio_norm);
      q.last_alpha = 0.0; q.iters_done = 0; q.exit_reason = EXIT_GTOL;
      if (!rc) rc = sdplr_hip_get_scalar(q.s, SDPLR_S_OBJ, &q.obj);
      q.status = rc ? rc : SDPLR_ERR_UNSERVED;
    });
    std::vector<GroupMember> live;
    for (int i : idx) {
      sdplr_hip_major_item& q = it[i];
      if (q.status != SDPLR_ERR_UNSERVED) continue;               // (its prologue failed)
      if (!(q.grad_norm > q.cur_gtol)) { q.status = SDPLR_OK; continue; }   // src/sdplr.jl:190: the loop is not entered
      live.push_back(GroupMember{&q, -1});
    }
    if (live.size() >= 2) {
      ApiShared api_guard(dev);
      (void)run_edge_group(live);   // (the members' status carries the outcome)
    } else if (live.size() == 1) {
      sdplr_hip_major_item& q = *live[0].q;
      q.status = sdplr_hip_inner_loop(q.s, q.normC, q.normb, q.gtol_relative, q.ptol_relative, q.use_armijo, q.cur_gtol, q.fprec_eps,
                                      q.max_local_iters, q.time_budget_s, &q.lagrangian, &q.grad_norm, &q.primal_vio_norm, &q.last_alpha,
                                      &q.iters_done, &q.exit_reason);
      if (!q.status) q.status = sdplr_hip_get_scalar(q.s, SDPLR_S_OBJ, &q.obj);
    }
  }
  run_singles(single, [&](int i) {
    sdplr_hip_major_item& q = it[i];
    if (!q.s) { q.status = SDPLR_ERR_INVALID_ARG; return; }
    q.status = sdplr_hip_major_iteration(q.s, q.normC, q.normb, q.gtol_relative, q.ptol_relative, q.use_armijo, q.update_lambda,
                                         q.sigma, q.cur_gtol, q.fprec_eps, q.max_local_iters, q.time_budget_s, &q.lagrangian,
                                         &q.grad_norm, &q.primal_vio_norm, &q.last_alpha, &q.iters_done, &q.exit_reason);
    if (!q.status) q.status = sdplr_hip_get_scalar(q.s, SDPLR_S_OBJ, &q.obj);
  });
  for (int i = 0; i < count; i++) if (it[i].status) return it[i].status;
  return SDPLR_OK;
}

int32_t sdplr_hip_batch_dual_obj(int32_t count, sdplr_hip_dual_item* it) {
  if (count < 0 || (count > 0 && !it)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_dual_obj: bad args");
  std::vector<int> single;
  // an item keeps SDPLR_ERR_UNSERVED until it has actually been served: a call that returns early (argument checks, staging
  // blocks) must not leave OK beside zeroed outputs
  for (int i = 0; i < count; i++) it[i].status = SDPLR_ERR_UNSERVED;
  {
    std::vector<const S*> hs;
    for (int i = 0; i < count; i++) hs.push_back(it[i].s);
    if (!batch_handles_distinct(hs)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_dual_obj: a handle appears twice");   // (before any handle is dereferenced)
    int dev = -1;
    if (!batch_one_device(hs, &dev)) return fail(nullptr, SDPLR_ERR_INVALID_ARG, "batch_dual_obj: the handles live on different devices");
    ApiShared api_guard(dev);
    const bool fused_ok = getenv("SDPLR_HIP_NO_FUSED_DUAL") == nullptr;
    std::map<int, std::vector<int>> groups;   // keyed by "the packed columns fit in LDS" (the kernel's template flag)
    for (int i = 0; i < count; i++) {
      S* s = it[i].s;
      if (s) s->G_consistent = false;   // (conservative, as the single-instance entry point)
      if (s && s->finalized && it[i].v0 && fused_ok && s->n >= 2 && rs_lanczos_ell_applies(s)) {
        bool in_lds = false;
        (void)rs_lz_ell_lds(s, &in_lds);
        groups[in_lds ? 1 : 0].push_back(i);
      } else {
        single.push_back(i);
      }
    }
    for (auto& g : groups) {
      std::vector<int>& idx = g.second;
      if (idx.size() < 2) { single.insert(single.end(), idx.begin(), idx.end()); continue; }
      size_t lo = 0;
      while (lo < idx.size()) {
        // rows [lo, hi): table + result rows + the start vectors fit one staging block
        size_t hi = lo, bytes = 0;
        while (hi < idx.size()) {
          const size_t add = batch_up(sizeof(RsLzEllArgs)) + 64 + batch_up((size_t)it[idx[hi]].s->n * sizeof(double)) +
                             (it[idx[hi]].y_out ? batch_up((size_t)(it[idx[hi]].s->m + 1) * sizeof(double)) : 0);
          if (hi > lo && bytes + add + 2 * BATCH_ALIGN > ARENA_CHUNK) break;
          bytes += add;
          hi++;
        }
        const size_t nb = hi - lo;
        S* s = it[idx[lo]].s;
        if (bytes + 2 * BATCH_ALIGN > ARENA_CHUNK) { single.push_back(idx[lo]); lo = hi; continue; }   // (one instance too large for the block)
        BatchBuf bb;
        if (bb.init() != hipSuccess) {
          for (size_t k = 0; k < nb; k++) it[idx[lo + k]].status = SDPLR_ERR_ALLOC;
          return fail(s, SDPLR_ERR_ALLOC, "batch_dual_obj: staging blocks");
        }
        RsLzEllArgs* tab = reinterpret_cast<RsLzEllArgs*>(bb.host);
        size_t v0_off = batch_up(nb * sizeof(RsLzEllArgs));
        size_t lds = 0;
        int rc = SDPLR_OK;
        std::vector<size_t> v0_at(nb);
        for (size_t k = 0; k < nb && !rc; k++) {   // the start vectors, behind the table: they travel in the same copy
          S* sk = it[idx[lo + k]].s;
          v0_at[k] = v0_off;
          memcpy(bb.host + v0_off, it[idx[lo + k]].v0, (size_t)sk->n * sizeof(double));
          v0_off += batch_up((size_t)sk->n * sizeof(double));
        }
        const size_t res_off = v0_off;
        // behind the result rows: the copies of y for the items that ask for them (they come back in the same transfer)
        size_t y_off = res_off + batch_up(nb * 4 * sizeof(double));
        std::vector<size_t> y_at(nb, 0);
        for (size_t k = 0; k < nb; k++)
          if (it[idx[lo + k]].y_out) {
            y_at[k] = y_off;
            y_off += batch_up((size_t)(it[idx[lo + k]].s->m + 1) * sizeof(double));
          }
        const size_t back_bytes = y_off - res_off;
        for (size_t k = 0; k < nb && !rc; k++) {
          const sdplr_hip_dual_item& q = it[idx[lo + k]];
          S* sk = q.s;
          sk->hc_valid = false;
          const double itd = (double)std::max<int64_t>(q.iter, 100);
          int64_t steps = (int64_t)(2 * std::ceil(std::pow(itd, 0.5) * std::log((double)sk->n)));   // src/coreop.jl:402
          steps = std::min<int64_t>(steps, sk->n - 1);                                              // :465
          if ((rc = ensure_lz_capacity(sk, steps))) { if (sk != s) s->err = sk->err; break; }
          RsLzEllArgs a = rs_lz_ell_args(sk, steps, true);
          a.v0 = reinterpret_cast<const double*>(bb.dev + v0_at[k]);
          a.out = reinterpret_cast<double*>(bb.dev + res_off) + 4 * k;
          a.y_copy = q.y_out ? reinterpret_cast<double*>(bb.dev + y_at[k]) : nullptr;
          tab[k] = a;
          bool in_lds = false;
          lds = std::max(lds, rs_lz_ell_lds(sk, &in_lds));
        }
        if (!rc) {
          const RsLzEllArgs* dtab = reinterpret_cast<const RsLzEllArgs*>(bb.dev);
          const bool in_lds = g.first == 1;
          rc = batch_round_trip(s, bb, res_off, res_off, back_bytes, [&](hipStream_t st) {
            if (in_lds) {
              RS_SET_ATTR(k_rs_lanczos_ell_batch<true>);
              k_rs_lanczos_ell_batch<true><<<(int)nb, SDPLR_RS_NT, lds, st>>>(dtab);
            } else {
              RS_SET_ATTR(k_rs_lanczos_ell_batch<false>);
              k_rs_lanczos_ell_batch<false><<<(int)nb, SDPLR_RS_NT, lds, st>>>(dtab);
            }
          });
        }
        const double* res = reinterpret_cast<const double*>(bb.host + res_off);
        for (size_t k = 0; k < nb; k++) {
          sdplr_hip_dual_item& q = it[idx[lo + k]];
          S* sk = q.s;
          q.status = rc;
          if (rc) { if (sk != s) sk->err = s->err; continue; }
          sk->S_stale = true; sk->S_from_y = true;
          sk->st_rs_lz++; sk->st_rs_shared++;
          const double ev = res[4 * k], yb = res[4 * k + 1];
          q.mineig = ev;
          q.dual_value = -yb + q.trace_bound * std::min(ev, 0.0);                                   // :412
          if (q.y_out) memcpy(q.y_out, bb.host + y_at[k], (size_t)(sk->m + 1) * sizeof(double));
        }
        lo = hi;
      }
    }
  }
  run_singles(single, [&](int i) {
    sdplr_hip_dual_item& q = it[i];
    if (!q.s) { q.status = SDPLR_ERR_INVALID_ARG; return; }
    q.status = sdplr_hip_dual_obj(q.s, q.trace_bound, q.iter, q.v0, &q.dual_value, &q.mineig);
    if (!q.status && q.y_out) q.status = sdplr_hip_get_vec(q.s, SDPLR_V_Y, q.y_out, q.s->m + 1);
  });
  for (int i = 0; i < count; i++) if (it[i].status) return it[i].status;
  return SDPLR_OK;
}

}  // extern "C"
