// altro_batch.hip -- C-ABI (include/altro_batch.h) of the MI355X batched ALTRO solver.
//
// Host logic only: owns device memory (lane layout, DESIGN.md "Data layout in HBM"), converts
// the caller's instance-major host arrays to/from it with small pack kernels, and launches the
// solve kernel of solve_dpp16.h.  There is no CPU compute path in this library: every entry
// point that computes launches HIP kernels, and creation fails if no HIP device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/altro_batch.h"
#include "launch_ring.h"
#include "pn_polish.h"
#include "solve_dpp16.h"
// The one-wave-per-instance kernels are compiled in translation units of their own (wide_inst.hip, _lib.build) and only
// declared here; -DALTRO_WIDE_SINGLE_TU (and the development builds) instantiate them in this unit as before.
#if !defined(ALTRO_WIDE_SINGLE_TU) && !defined(ALTRO_DEV_HEADLINE_ONLY)
#define ALTRO_WIDE_EXTERN
#endif
#include "wide_backend.h"

using altro::IPW;
using altro::LW;

static thread_local std::string g_create_err;

// Diagnostic switches (altro_debug_set, include/altro_batch.h).  The library reads NOTHING from the environment: a test or
// a measuring tool that wants a scheduling feature off says so through the entry point -- with a null handle for the
// handles this thread creates afterwards, with a handle for that handle.
struct DebugSwitches {
  int lone = 1, shadow = 1, resync = 1, reuse = 1, useqz = 1, mate = 1;
  int group = 1;             // 0 off, 1 sorted, 2-4: other slot orders (k_group_rank)
  int group_max_steps = 32;
  int trace_wave = -1;       // -DALTRO_PHASE_STAMPS builds
  int force_wide = 0;        // every (n, m) on the one-wave-per-instance backend
  int keep_gains = 0;        // -DALTRO_DEBUG builds only
  int wide_compact = -1, wide_coop = -1, wide_static_mask = -1;   // -1: the backend's default
};
static thread_local DebugSwitches g_dbg;

struct altro_handle {
  altro_wide::WideBackend* wide = nullptr;  // set: this handle runs on the one-wave-per-instance kernel
  altro_dims d{};
  altro_opts o{};
  int device = 0;
  int Bp = 0;  // batch padded to a multiple of IPW
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  altro::LaunchRing ring;        // start/end event pairs of the most recent solve launches
  double* Zsave = nullptr;       // [Bp][N][16]: Z0 of altro_batch_benchmark_solve
  hipEvent_t bev0 = nullptr, bev1 = nullptr;
  long long *n_backward = nullptr, *n_rollout = nullptr, *wave_cycles = nullptr;
  long long *n_solves = nullptr, *n_iters = nullptr, *n_ok = nullptr, *n_trials = nullptr, *n_gconf = nullptr;
  int* dzero = nullptr;  // [Bp] the last iteration of the last solve was costate-confirmed: its d is zero
  // problem data (device)
  double *Gcol = nullptr, *Grow = nullptr, *fvec = nullptr;
  double *wd = nullptr, *wf = nullptr, *zmin = nullptr, *zmax = nullptr;
  double *x0 = nullptr, *Zref = nullptr, *Z = nullptr, *Lb = nullptr, *mu = nullptr,
         *KD = nullptr, *Qz = nullptr, *Dff = nullptr, *kmu = nullptr;
  altro::ASet* ahash = nullptr;  // [Bp][16] active set of the backward pass behind the gains in KD (gain reuse, solve_dpp16.h)
  long long* n_fo = nullptr;
  // projected-Newton polish (pn_polish.h): per-instance results and the workspace, allocated by the first solve that asks for it
  int *pn_ran = nullptr, *pn_failed = nullptr, *pn_dfail = nullptr;
  double *pn_res = nullptr, *pn_dres0 = nullptr, *pn_dres = nullptr;
  double *pnE = nullptr, *pndv = nullptr, *pnLd = nullptr, *pnLo = nullptr, *pnvec = nullptr, *pntz = nullptr;
  int *pnnb = nullptr, *pnnst = nullptr, *pnrinfo = nullptr;
  int pn_bm = 0;
  double *noise = nullptr, *noise_w = nullptr;
  int* noise_grp = nullptr;
  int noise_mode = 0;
  int mpc_shift = 1;
  // scheduling switches (altro_debug_set; the defaults are the product's behaviour)
  int reuse = 1;  // gain reuse (solve_dpp16.h fosweep); "no_reuse" switches it off (tests)
  int lone = 1;  // backward_lone (solve_dpp16.h); "no_lone" switches it off (tests: lone == four-row pass bit for bit)
  int group_max_steps = 32;  // fused launches of more steps are not grouped ("group_max_steps": diagnostic)
  int shadow = 1;  // "no_shadow": rows that sit a phase out keep their own instance (solve_dpp16.h shadow_enter)
  int useqz = 1;   // "no_qz_pass": backward passes always recompute their cost / box expansion (solve_dpp16.h backward QV)
  int* cur = nullptr;
  int *perm = nullptr, *gscore = nullptr;  // [Bp] wave slot -> instance of a grouped MPC launch, and its sort key
  bool debug_keep_gains = false;           // "keep_gains" (-DALTRO_DEBUG builds only): the setters do NOT drop the stored gains
                                           // (exists to show that the tests notice stale gains)
  int group = 1;                           // "no_group": identity; "group_mode": other slot orders
  int dbg_wave = -1;                       // "trace_wave" (diagnostic builds only)
  int resync = 1;                          // "no_resync": rows never wait for their wave-mates
  int mate = 1;                            // "no_mate_rank": issue priority from the row's own state only, not from its SIMD mates'
  unsigned* simd_tab = nullptr;            // [65536][16], zero between launches
  int *iters = nullptr, *iters_outer = nullptr, *status = nullptr;
  double *cost = nullptr, *cmax = nullptr, *Jtrace = nullptr, *ctrace = nullptr, *atrace = nullptr;
  double* stage = nullptr;  // device staging buffer for host<->device layout conversion
  size_t stage_bytes = 0;
  int Nt = 0;    // knots held by Zref
  int kref = 0;  // current reference window start
  int noise_steps = 0;
  int box_k0 = 0, box_k1 = -1, box_id = -1;
  // generic affine constraints packed into 4 quads of 4 constraint rows (see solve_dpp16.h)
  double *Acon = nullptr, *bcon = nullptr, *Lc = nullptr;
  int* cmeta = nullptr;
  int* ckn = nullptr;   // [16] canonical knot of each constraint lane (time-invariant tables, see pack_constraints)
  int con_inv = 0;
  std::vector<double> Acon_h, bcon_h;  // per-knot tables [ninst][N][16][16], [ninst][N][16]; ninst = 1 (shared) or Bp
  bool con_per_instance = false;
  size_t acon_elems = 0;               // elements the device table Acon currently holds
  std::vector<int> cmeta_h;            // [N][16][4]
  int ncrows = 0;       // 16 once any generic constraint exists (kernel flag)
  bool con_dirty = false, con_locked = false;  // packing is redone until the first solve
  struct ConBlock {
    int id, kind, sense, k0, k1, p;
    int per_knot, per_instance;
    std::vector<double> A, b;  // A row-major p x nz blocks: [instance if per_instance][knot of the range if per_knot]
    int lanes[LW];
  };
  std::vector<ConBlock> cons;
  int* lanebuf = nullptr;  // device scratch [16] for dual transfers
  int* bslot = nullptr;  // device [16]
  int bslot_h[LW];       // host copy: slot of z element j among the bounded ones, -1 if none
  int nbp = 1;           // slots per side of the compact dual rows
  int ncon = 0;
  bool have_dyn = false, have_cost = false, have_ref = false, have_x0 = false;
  bool dyn_per_instance = false;
  double dt = 0.0;
  std::string err;
};

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return ALTRO_ERR_HIP;                                                               \
    }                                                                                     \
  } while (0)

#define FAIL(h, code, msg) \
  do {                     \
    (h)->err = (msg);      \
    return (code);         \
  } while (0)

// forward an entry point to the wide backend when the handle runs on it
#define WIDE_FWD(h, call)                           \
  do {                                              \
    if ((h) && (h)->wide) {                         \
      const int rc_ = (h)->wide->call;              \
      if (rc_) (h)->err = (h)->wide->err;           \
      return rc_;                                   \
    }                                               \
  } while (0)

static int ensure_stage(altro_handle* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return ALTRO_OK;
  if (h->stage) HIPCHK(h, hipFree(h->stage));
  h->stage = nullptr;
  h->stage_bytes = 0;
  HIPCHK(h, hipMalloc(&h->stage, bytes));
  h->stage_bytes = bytes;
  return ALTRO_OK;
}

// ------------------------------------------------------------------ layout kernels
// All take one thread per (instance slot, lane); padded slots mirror instance B-1.

__global__ void k_pack_traj(const double* __restrict__ X, const double* __restrict__ U, double* __restrict__ Zp,
                            const int* __restrict__ cur, size_t plane, int B, int Bp, int N, int n, int m,
                            int use_cur, int have_x) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  const int b = inst < B ? inst : B - 1;
  double* dst = Zp + (size_t)inst * (2 * (size_t)N + 1) * LW + (use_cur ? (size_t)cur[inst] * plane : 0);
  for (int k = 0; k < N; ++k) {
    double v = 0.0;
    bool wr = false;
    if (j < n) {
      if (have_x) { v = X[((size_t)b * N + k) * n + j]; wr = true; }
    } else if (j < n + m) {
      if (k < N - 1) v = U[((size_t)b * (N - 1) + k) * m + (j - n)];
      wr = true;
    } else {
      wr = true;
    }
    if (wr) dst[(size_t)k * LW + j] = v;
  }
}

__global__ void k_unpack_traj(double* __restrict__ X, double* __restrict__ U, const double* __restrict__ Zp,
                              const int* __restrict__ cur, size_t plane, int B, int Bp, int N, int n, int m) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * LW) return;
  const int inst = t / LW, j = t % LW;
  const double* src = Zp + (size_t)inst * (2 * (size_t)N + 1) * LW + (size_t)cur[inst] * plane;
  for (int k = 0; k < N; ++k) {
    const double v = src[(size_t)k * LW + j];
    if (j < n) {
      if (X) X[((size_t)inst * N + k) * n + j] = v;
    } else if (j < n + m && k < N - 1) {
      if (U) U[((size_t)inst * (N - 1) + k) * m + (j - n)] = v;
    }
  }
}

// reference: Xref [B][Nt][n], Uref [B][Nt-1][m] -> Zref [Bp][Nt][16]
__global__ void k_pack_ref(const double* __restrict__ X, const double* __restrict__ U, double* __restrict__ Zr,
                           int B, int Bp, int Nt, int n, int m) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  const int b = inst < B ? inst : B - 1;
  for (int k = 0; k < Nt; ++k) {
    double v = 0.0;
    if (j < n) v = X[((size_t)b * Nt + k) * n + j];
    else if (j < n + m && k < Nt - 1) v = U[((size_t)b * (Nt - 1) + k) * m + (j - n)];
    Zr[((size_t)inst * Nt + k) * LW + j] = v;
  }
}

// dynamics: A [nb][n*n] col-major, Bm [nb][n*m] col-major, f [nb][n]; nb = B or 1
__global__ void k_pack_dyn(const double* __restrict__ A, const double* __restrict__ Bm, const double* __restrict__ f,
                           double* __restrict__ Gcol, double* __restrict__ Grow, double* __restrict__ fvec,
                           int B, int Bp, int n, int m, int per_instance) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  const int b = per_instance ? (inst < B ? inst : B - 1) : 0;
  const double* Ab = A + (size_t)b * n * n;
  const double* Bb = Bm + (size_t)b * n * m;
  // Gcol[inst][k][j] = G[k][j], G = [A B]
  for (int k = 0; k < n; ++k) {
    double v = 0.0;
    if (j < n) v = Ab[k + n * j];
    else if (j < n + m) v = Bb[k + n * (j - n)];
    Gcol[((size_t)inst * n + k) * LW + j] = v;
  }
  // Grow[inst][c][i=j] = G[j][c]
  for (int c = 0; c < LW; ++c) {
    double v = 0.0;
    if (j < n) {
      if (c < n) v = Ab[j + n * c];
      else if (c < n + m) v = Bb[j + n * (c - n)];
    }
    Grow[((size_t)inst * LW + c) * LW + j] = v;
  }
  fvec[(size_t)inst * LW + j] = (f && j < n) ? f[(size_t)b * n + j] : 0.0;
}

__global__ void k_pack_x0(const double* __restrict__ x0, double* __restrict__ dst, int B, int Bp, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  const int b = inst < B ? inst : B - 1;
  dst[t] = j < n ? x0[(size_t)b * n + j] : 0.0;
}

__global__ void k_unpack_x0(double* __restrict__ x0, const double* __restrict__ src, int B, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * LW) return;
  const int inst = t / LW, j = t % LW;
  if (j < n) x0[(size_t)inst * n + j] = src[t];
}

// box duals: host [B][nk][2][nz] (dense, zero for unbounded elements)  <->  Lb [Bp][N+1][2][nbp]
__global__ void k_duals(double* __restrict__ host, double* __restrict__ Lb, const int* __restrict__ bslot, int nbp,
                        int B, int Bp, int N, int nz, int k0, int k1, int to_host) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  if (j >= nz) return;
  const int b = inst < B ? inst : B - 1;
  const int nk = k1 - k0 + 1;
  const int sl = bslot[j];
  for (int k = k0; k <= k1; ++k) {
    const size_t hi = (((size_t)b * nk + (k - k0)) * 2 + 0) * nz + j;
    const size_t lo = (((size_t)b * nk + (k - k0)) * 2 + 1) * nz + j;
    const size_t dh = (((size_t)inst * (N + 1) + k) * 2 + 0) * nbp + (sl >= 0 ? sl : 0);
    const size_t dl = (((size_t)inst * (N + 1) + k) * 2 + 1) * nbp + (sl >= 0 ? sl : 0);
    if (to_host) {
      if (inst < B) { host[hi] = sl >= 0 ? Lb[dh] : 0.0; host[lo] = sl >= 0 ? Lb[dl] : 0.0; }
    } else if (sl >= 0) {
      Lb[dh] = host[hi];
      Lb[dl] = host[lo];
    }
  }
}

// RD.shift_fill!(Z) on the current plane and Altro.shift_fill!(conSet) on the box duals
// (random_linear_problem.jl:136,139): entry k <- entry k+1, last entry kept.
__global__ void k_shift(double* __restrict__ Zp, const int* __restrict__ cur, size_t plane, double* __restrict__ Lb,
                        int nbp, int Bp, int N, int n, int m, int k0, int k1, int primal, int dual,
                        double* __restrict__ Lc, const int* __restrict__ cmeta, int ncrows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  const size_t ks = LW;  // rows of an instance are consecutive (instance-major arrays)
  if (primal) {
    double* z = Zp + (size_t)inst * (2 * (size_t)N + 1) * LW + (size_t)cur[inst] * plane + j;
    const int kend = (j < n) ? N - 1 : N - 2;  // x: knots 0..N-2 take k+1; u: knots 0..N-3
    if (j < n + m)
      for (int k = 0; k < kend; ++k) z[(size_t)k * ks] = z[(size_t)(k + 1) * ks];
  }
  if (dual && k1 >= k0) {  // the 16 threads of the instance share the 2*nbp elements of its compact rows
    const size_t ls = (size_t)2 * nbp;
    for (int e = j; e < 2 * nbp; e += LW) {
      double* l = Lb + (size_t)inst * (N + 1) * 2 * nbp + e;
      for (int k = k0; k < k1; ++k) l[(size_t)k * ls] = l[(size_t)(k + 1) * ls];
    }
  }
  if (dual && j < ncrows) {  // generic constraint rows on lane j: each constraint shifts inside its own range
    double* lc = Lc + (size_t)inst * (N + 1) * LW + j;
    for (int k = 0; k < N - 1; ++k) {
      const int* cm = cmeta + ((size_t)k * LW + j) * 4;
      if (cm[0] != 0 && k < cm[2]) lc[(size_t)k * ks] = lc[(size_t)(k + 1) * ks];
    }
  }
}

// duals of one generic constraint block: host [B][nk][p]  <->  Lc [Bp][N+1][16], rows on lanes lane0..lane0+p-1
__global__ void k_cduals(double* __restrict__ host, double* __restrict__ Lc, int B, int N, const int* __restrict__ lanes,
                         int p, int k0, int k1, int to_host) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * p) return;
  const int inst = t / p, r = t % p;
  const int nk = k1 - k0 + 1;
  for (int k = k0; k <= k1; ++k) {
    const size_t hi = ((size_t)inst * nk + (k - k0)) * p + r;
    const size_t di = ((size_t)inst * (N + 1) + k) * LW + lanes[r];
    if (to_host) host[hi] = Lc[di];
    else Lc[di] = host[hi];
  }
}

// benchmark_solve!: Z0 = copy(get_trajectory(solver)) and initial_trajectory!(solver, Z0) on the current plane
__global__ void k_plane_copy(double* __restrict__ Zp, double* __restrict__ Zs, const int* __restrict__ cur, size_t plane,
                             int Bp, int N, int save) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  double* z = Zp + (size_t)inst * (2 * (size_t)N + 1) * LW + (size_t)cur[inst] * plane + j;
  double* zs = Zs + (size_t)inst * N * LW + j;
  for (int k = 0; k < N; ++k) {
    if (save) zs[(size_t)k * LW] = z[(size_t)k * LW];
    else z[(size_t)k * LW] = zs[(size_t)k * LW];
  }
}

// ---- grouping of the instances of a fused MPC launch (solve_dpp16.h: a wave executes the union of the phases its
// four rows need, so rows with the same needs belong in the same wave).  Whether an instance will need backward
// passes is decided by whether its window holds an active box row: then the active set moves with the window from one
// step to the next and the gains cannot be taken from memory (47 % of the headline's instances need no pass in 20
// steps, 40 % one at nearly every step; tools/gpu_class_persist.py).  The tracking problem follows its reference, so
// the REFERENCE tells which windows those are: score = number of the launch's steps whose window holds a reference
// knot at (or within 2 % of) a bound.  It is a scheduling heuristic only: results do not depend on which rows share a
// wave (tests: instance results do not depend on the batch; lone-row == four-row pass bit for bit).
// one 16-lane row per instance (lane j = element j of z: coalesced 128-byte reads), 16 instances per block
__global__ void k_group_score(const double* __restrict__ Zref, const double* __restrict__ zmin, const double* __restrict__ zmax,
                              int* __restrict__ score, int Bp, int Nt, int first, int nsteps, int k0, int k1, int nz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = t / LW, j = t % LW;
  const bool live = b < Bp;
  const int bb = live ? b : Bp - 1;
  const int W = k1 - k0 + 1;                 // knots of a window that carry the box rows
  const int a0 = first + 1 + k0;             // first absolute knot touched by the launch's windows
  const int na = nsteps + W - 1;             // absolute knots touched
  if (na > 256 || W < 1) { if (live && j == 0) score[b] = 0; return; }
  const double lo = zmin[j], hi = zmax[j];
  const bool fl = (j < nz) && lo > -1e300, fh = (j < nz) && hi < 1e300;
  const double m = 0.02 * ((fl && fh) ? 0.5 * (hi - lo) : fmax(1.0, fabs(fh ? hi : lo)));
  unsigned long long bits[4] = {0ull, 0ull, 0ull, 0ull};
  const int sh = (threadIdx.x & 63) & ~15;   // this row's 16 bits of the wave's ballot
  for (int a = 0; a < na; ++a) {
    const double z = Zref[((size_t)bb * Nt + (a0 + a)) * LW + j];
    const bool act = (fh && z >= hi - m) || (fl && z <= lo + m);
    const bool any = ((__ballot(act) >> sh) & 0xFFFFull) != 0ull;
    if (any) bits[a >> 6] |= 1ull << (a & 63);
  }
  if (!live || j != 0) return;
  int cnt = 0, sc = 0;
  for (int a = 0; a < W; ++a) cnt += (int)((bits[a >> 6] >> (a & 63)) & 1ull);
  for (int st = 0; st < nsteps; ++st) {
    sc += cnt > 0 ? 1 : 0;
    const int out = st, in = st + W;
    cnt -= (int)((bits[out >> 6] >> (out & 63)) & 1ull);
    if (in < na) cnt += (int)((bits[in >> 6] >> (in & 63)) & 1ull);
  }
  score[b] = sc;
}

// perm = the instances in ascending order of (score, index): a stable counting sort in ONE block of 256 threads (thread t
// owns a contiguous chunk of instances; scores are at most GROUP_BINS - 1 = the steps of a grouped launch)
constexpr int GROUP_BINS = 33;
__global__ void __launch_bounds__(256) k_group_rank(const int* __restrict__ score, int* __restrict__ perm, int Bp, int mode) {
  __shared__ int cnt[GROUP_BINS][257];
  __shared__ int base[GROUP_BINS + 1];
  const int t = threadIdx.x;
  const int chunk = (Bp + 255) / 256;
  const int i0 = t * chunk, i1 = (i0 + chunk < Bp) ? i0 + chunk : Bp;
  for (int b = 0; b < GROUP_BINS; ++b) cnt[b][t] = 0;
  for (int i = i0; i < i1; ++i) {
    const int sc = score[i] < GROUP_BINS - 1 ? score[i] : GROUP_BINS - 1;
    cnt[sc][t] += 1;
  }
  __syncthreads();
  if (t < GROUP_BINS) {  // exclusive prefix over the threads, per bin
    int acc = 0;
    for (int q = 0; q < 256; ++q) {
      const int c = cnt[t][q];
      cnt[t][q] = acc;
      acc += c;
    }
    cnt[t][256] = acc;
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int b = 0; b < GROUP_BINS; ++b) { base[b] = acc; acc += cnt[b][256]; }
  }
  __syncthreads();
  const int W = Bp / 4;
  int off[GROUP_BINS];
  for (int b = 0; b < GROUP_BINS; ++b) off[b] = base[b] + cnt[b][t];
  for (int i = i0; i < i1; ++i) {
    const int sc = score[i] < GROUP_BINS - 1 ? score[i] : GROUP_BINS - 1;
    const int rank = off[sc]++;
    int slot = rank;                                  // mode 1: sorted, the instances with the fewest expected passes first
    if (mode == 2) {                                  // sorted waves, light and heavy ones alternating in the block order
      const int wr = rank / 4, q = rank % 4;          // (a permutation for odd W too: the lighter (W + 1) / 2 take the even slots)
      slot = ((wr < (W + 1) / 2) ? 2 * wr : 2 * (W - 1 - wr) + 1) * 4 + q;
    } else if (mode == 3) {                           // every wave gets one instance of each quartile
      slot = (rank % W) * 4 + rank / W;
    } else if (mode == 4) {                           // sorted waves, the upper half in DESCENDING order: with W = 2 waves per
      const int wr = rank / 4, q = rank % 4;          // SIMD the dispatcher puts blocks i and i + W / 2 on one SIMD, so the
      const int hw = (W + 1) / 2;                     // wave with the most expected passes shares its SIMD with the one
      slot = ((wr < hw) ? wr : hw + (W - 1 - wr)) * 4 + q;   // with the fewest (min-max pairing), not with a middle one
    }
    perm[slot] = i;
  }
}

// initial_trajectory!(prob, Z): plane 0 of Z <- the first N knots of the track (mpc.jl:19-20,45)
__global__ void k_window_copy(double* __restrict__ Zp, const double* __restrict__ Zr, int Bp, int N, int Nt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Bp * LW) return;
  const int inst = t / LW, j = t % LW;
  double* z = Zp + (size_t)inst * (2 * (size_t)N + 1) * LW + j;
  const double* r = Zr + (size_t)inst * Nt * LW + j;
  for (int k = 0; k < N; ++k) z[(size_t)k * LW] = r[(size_t)k * LW];
}

__global__ void k_fill(double* p, double v, size_t nelem) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nelem) p[t] = v;
}

// ------------------------------------------------------------------ helpers
static inline dim3 grid_for(size_t threads, int block = 256) { return dim3((unsigned)((threads + block - 1) / block)); }

static bool supported_dims(int n, int m) {
  return (n == 12 && m == 4) || (n == 6 && m == 3) || (n == 6 && m == 6) || (n == 8 && m == 4) || (n == 12 && m == 3);
}

static int launch_solve(altro_handle* h, int first_step, int nsteps, int prepare_only = 0) {
  altro::SolveParams p{};
  p.prepare_only = prepare_only;
  p.B = h->d.batch; p.Bp = h->Bp; p.N = h->d.N; p.Nt = h->Nt;
  p.kref = h->kref;
  p.first_step = first_step; p.nsteps = nsteps;
  p.noise = h->noise; p.noise_w = h->noise_w; p.noise_grp = h->noise_grp; p.noise_mode = h->noise_mode; p.mpc_shift = h->mpc_shift;
  p.box_k0 = h->box_k0; p.box_k1 = h->box_k1;
  p.Gcol = h->Gcol; p.Grow = h->Grow; p.fvec = h->fvec;
  p.wd = h->wd; p.wf = h->wf; p.zmin = h->zmin; p.zmax = h->zmax;
  p.x0 = h->x0; p.Zref = h->Zref; p.Z = h->Z; p.cur = h->cur;
  p.Lb = h->Lb; p.bslot = h->bslot; p.nbp = h->nbp; p.mu = h->mu; p.lone = h->lone; p.useqz = h->useqz; p.shadow = h->shadow; p.reuse = h->reuse; p.resync = h->resync; p.dbg_wave = h->dbg_wave;
  p.Dff = h->Dff; p.ahash = h->ahash; p.kmu = h->kmu; p.n_fo = h->n_fo;
  p.Qz = h->Qz;
  p.Acon = h->Acon; p.bcon = h->bcon; p.cmeta = h->cmeta; p.ckn = h->ckn; p.con_inv = h->con_inv;
  p.con_istride = h->con_per_instance ? (unsigned)(h->d.N * LW * LW) : 0u; p.Lc = h->Lc; p.ncrows = h->ncrows; p.KD = h->KD;
  p.iters = h->iters; p.iters_outer = h->iters_outer; p.status = h->status;
  p.cost = h->cost; p.cmax = h->cmax; p.Jtrace = h->Jtrace; p.ctrace = h->ctrace; p.atrace = h->atrace;
  p.n_backward = h->n_backward; p.n_rollout = h->n_rollout; p.wave_cycles = h->wave_cycles;
  p.simd_tab = h->mate ? h->simd_tab : nullptr;
  p.n_solves = h->n_solves; p.n_iters = h->n_iters; p.n_ok = h->n_ok; p.n_trials = h->n_trials;
  p.n_gconf = h->n_gconf; p.dzero = h->dzero;
  p.o = h->o;
  if (h->o.projected_newton) {  // solve!(::ALTROSolver): the AL stage only has to reach the polish's tolerance
    if (h->o.projected_newton_tolerance >= 0) p.o.constraint_tolerance = h->o.projected_newton_tolerance;
    else { p.o.constraint_tolerance = 0.0; p.o.kickout_max_penalty = 1; }
  }
  p.perm = nullptr;
  // fused MPC launches of box-constrained problems: group the instances by how many of the launch's steps will need
  // backward passes (see k_group_score); everything else runs in instance order
  // (short launches only: over 100 steps nearly every window meets a bound at some point, the score stops separating the
  //  instances and clustering the pass-heavy rows -- they are also the ones with the hard solves -- lengthens the tail:
  //  measured 20 steps +2 %, 100 steps -3 %, tools/gpu_ab.py)
  if (h->group && h->reuse && !h->o.strict && nsteps >= 4 && nsteps <= h->group_max_steps && !prepare_only && h->ncrows == 0 && h->box_k1 >= h->box_k0 && h->Bp <= 32768) {
    hipLaunchKernelGGL(k_group_score, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->Zref, h->zmin, h->zmax, h->gscore, h->Bp, h->Nt,
                       first_step, nsteps, h->box_k0, h->box_k1, h->d.n + h->d.m);
    hipLaunchKernelGGL(k_group_rank, dim3(1), dim3(256), 0, h->stream, h->gscore, h->perm, h->Bp, h->group);
    p.perm = h->perm;
  }
  const dim3 grid(h->Bp / IPW), block(64);
  const int n = h->d.n, m = h->d.m;
  const bool cones = h->ncrows > 0;
#define ALTRO_LAUNCH(NX_, NU_)                                                                            \
  do {                                                                                                    \
    if (cones) hipLaunchKernelGGL((altro::solve_kernel<NX_, NU_, true>), grid, block, 0, h->stream, p);  \
    else hipLaunchKernelGGL((altro::solve_kernel<NX_, NU_, false>), grid, block, 0, h->stream, p);       \
  } while (0)
#ifdef ALTRO_DEV_HEADLINE_ONLY  // development builds: only the headline instantiation (compiles in a fraction of the time)
  if (n == 12 && m == 4 && !cones) hipLaunchKernelGGL((altro::solve_kernel<12, 4, false>), grid, block, 0, h->stream, p);
  else FAIL(h, ALTRO_ERR_UNSUPPORTED, "development build: only (12, 4) without cones");
#else
  if (n == 12 && m == 4) ALTRO_LAUNCH(12, 4);
  else if (n == 6 && m == 3) ALTRO_LAUNCH(6, 3);
  else if (n == 6 && m == 6) ALTRO_LAUNCH(6, 6);
  else if (n == 8 && m == 4) ALTRO_LAUNCH(8, 4);
  else if (n == 12 && m == 3) ALTRO_LAUNCH(12, 3);
  else FAIL(h, ALTRO_ERR_UNSUPPORTED, "no kernel built for this (n, m)");
#endif
#undef ALTRO_LAUNCH
  HIPCHK(h, hipGetLastError());
  return ALTRO_OK;
}

// The gains kept in KD for reuse (solve_dpp16.h fosweep) depend on the dynamics, the cost weights, the set of bounded
// elements and the options: every setter of those drops them (kmu < 0: no valid gains).  Trajectories, duals and
// reference windows need no such care: the active set they produce is hashed and compared at every use.
static int drop_gains(altro_handle* h) {
  if (h->kmu && !h->debug_keep_gains) {
    hipLaunchKernelGGL(k_fill, grid_for((size_t)h->Bp), dim3(256), 0, h->stream, h->kmu, -1.0, (size_t)h->Bp);
    HIPCHK(h, hipGetLastError());
  }
  return ALTRO_OK;
}

static int check_ready(altro_handle* h) {
  if (!h->have_dyn) FAIL(h, ALTRO_ERR_STATE, "altro_batch_set_dynamics has not been called");
  if (!h->have_cost) FAIL(h, ALTRO_ERR_STATE, "altro_batch_set_tracking_cost has not been called");
  if (!h->have_ref) FAIL(h, ALTRO_ERR_STATE, "no reference trajectory (altro_batch_set_reference / altro_mpc_set_track)");
  return ALTRO_OK;
}

static int upload(altro_handle* h, const double* host, size_t count, size_t stage_off_elems = 0) {
  HIPCHK(h, hipMemcpyAsync(h->stage + stage_off_elems, host, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return ALTRO_OK;
}

// Nothing may propagate across the C boundary: every entry point runs its body inside guard(), which
// turns std::bad_alloc (the std::vector / std::string members of the handle) and anything else into
// ALTRO_ERR_INTERNAL.
template <class F>
static int32_t guard(altro_handle* h, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try { if (h) h->err = "out of host memory"; else g_create_err = "out of host memory"; } catch (...) {}
    return ALTRO_ERR_INTERNAL;
  } catch (const std::exception& e) {
    try { if (h) h->err = std::string("internal error: ") + e.what(); else g_create_err = e.what(); } catch (...) {}
    return ALTRO_ERR_INTERNAL;
  } catch (...) {
    return ALTRO_ERR_INTERNAL;
  }
}

static void apply_wide_switches(altro_wide::WideBackend* wb) {
  wb->debug_keep_gains = g_dbg.keep_gains != 0;
  if (g_dbg.wide_compact >= 0) wb->compact_np_max = g_dbg.wide_compact;
  if (g_dbg.wide_coop >= 0) wb->coop_mode = g_dbg.wide_coop != 0 ? 1 : 0;
  if (g_dbg.wide_static_mask >= 0) wb->static_mask = g_dbg.wide_static_mask;
}

// ------------------------------------------------------------------ C-ABI
extern "C" {

int32_t altro_debug_set(altro_handle* h, const char* key, int32_t value) {
  return guard(h, [&]() -> int32_t {
    if (!key) return ALTRO_ERR_INVALID_ARG;
    const std::string k(key);
    auto bad = [&](int32_t code, const char* msg) -> int32_t {
      if (h) h->err = msg; else g_create_err = msg;
      return code;
    };
    if (k == "keep_gains") {
#ifdef ALTRO_DEBUG
      if (h) { if (h->wide) h->wide->debug_keep_gains = value != 0; else h->debug_keep_gains = value != 0; }
      else g_dbg.keep_gains = value != 0;
      return ALTRO_OK;
#else
      if (value == 0) return ALTRO_OK;
      return bad(ALTRO_ERR_UNSUPPORTED, "keep_gains exists in -DALTRO_DEBUG builds of the library only");
#endif
    }
    // create-time switches: which backend, and the LDS carve-up of the one-wave-per-instance backend
    int* pre = k == "force_wide" ? &g_dbg.force_wide : k == "wide_compact" ? &g_dbg.wide_compact : k == "wide_coop" ? &g_dbg.wide_coop
             : k == "wide_static_mask" ? &g_dbg.wide_static_mask : nullptr;
    if (pre) {
      if (h) return bad(ALTRO_ERR_STATE, "this switch is read when a handle is created: pass a null handle before altro_batch_create");
      *pre = value;
      return ALTRO_OK;
    }
    DebugSwitches tmp;
    DebugSwitches& d = h ? tmp : g_dbg;
    if (h) { tmp.mate = h->mate; tmp.lone = h->lone; tmp.useqz = h->useqz; tmp.shadow = h->shadow; tmp.resync = h->resync; tmp.reuse = h->reuse; tmp.group = h->group;
             tmp.group_max_steps = h->group_max_steps; tmp.trace_wave = h->dbg_wave; }
    if (k == "no_lone") d.lone = value ? 0 : 1;
    else if (k == "no_qz_pass") d.useqz = value ? 0 : 1;
    else if (k == "no_shadow") d.shadow = value ? 0 : 1;
    else if (k == "no_resync") d.resync = value ? 0 : 1;
    else if (k == "no_mate_rank") d.mate = value ? 0 : 1;
    else if (k == "no_reuse") d.reuse = value ? 0 : 1;
    else if (k == "no_group") d.group = value ? 0 : 1;
    else if (k == "group_mode") { if (value < 0 || value > 4) return bad(ALTRO_ERR_INVALID_ARG, "group_mode is 0..4"); d.group = value; }
    else if (k == "group_max_steps") d.group_max_steps = value;
    else if (k == "trace_wave") d.trace_wave = value;
    else return bad(ALTRO_ERR_INVALID_ARG, "unknown switch");
    if (h && !h->wide) {
      const bool reuse_changed = h->reuse != tmp.reuse;
      h->mate = tmp.mate;
      h->lone = tmp.lone; h->useqz = tmp.useqz; h->shadow = tmp.shadow; h->resync = tmp.resync; h->reuse = tmp.reuse; h->group = tmp.group;
      h->group_max_steps = tmp.group_max_steps; h->dbg_wave = tmp.trace_wave;
      if (reuse_changed) { HIPCHK(h, hipSetDevice(h->device)); return drop_gains(h); }
    }
    return ALTRO_OK;
  });
}

int32_t altro_default_opts(altro_opts* o) {
  return guard(nullptr, [&]() -> int32_t {
    if (!o) return ALTRO_ERR_INVALID_ARG;
    o->cost_tolerance = 1e-4;
    o->cost_tolerance_intermediate = 1e-4;
    o->gradient_tolerance = 10.0;
    o->gradient_tolerance_intermediate = 1.0;
    o->constraint_tolerance = 1e-6;
    o->penalty_initial = NAN;
    o->penalty_scaling = NAN;
    o->penalty_max = 1e8;
    o->dual_max = 1e8;
    o->line_search_lower_bound = 1e-8;
    o->line_search_upper_bound = 10.0;
    o->max_cost_value = 1e8;
    o->max_state_value = 1e8;
    o->max_control_value = 1e8;
    o->bp_reg_initial = 0.0;
    o->bp_reg_increase_factor = 1.6;
    o->bp_reg_max = 1e8;
    o->bp_reg_min = 1e-8;
    o->bp_reg_fp = 10.0;
    o->iterations = 1000;
    o->iterations_inner = 300;
    o->iterations_outer = 30;
    o->iterations_linesearch = 20;
    o->dJ_counter_limit = 10;
    o->reset_duals = 1;
    o->reset_penalties = 1;
    o->bp_reg = 0;
    o->soc_second_order = 1;
    o->strict = 0;
    o->kickout_max_penalty = 0;
    o->projected_newton = 0;
    o->projected_newton_tolerance = 1e-3;
    o->active_set_tolerance_pn = 1e-3;
    o->rho_chol = 1e-2;
    o->rho_primal = 1e-8;
    o->r_threshold = 1.1;
    return ALTRO_OK;
  });
}

const char* altro_last_error(const altro_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int32_t altro_batch_create(const altro_dims* dims, const altro_opts* opts, int32_t device, altro_handle** out) {
  return guard(nullptr, [&]() -> int32_t {
    if (!dims || !out) { g_create_err = "null argument"; return ALTRO_ERR_INVALID_ARG; }
    *out = nullptr;
    if (dims->batch < 1 || dims->n < 1 || dims->m < 1 || dims->N < 3) { g_create_err = "bad dims"; return ALTRO_ERR_INVALID_ARG; }
    // (n, m) of the 16-lane kernel set run there; everything else up to n <= 64, m <= 32 runs on the
    // one-wave-per-instance MFMA kernel (altro_debug_set(NULL, "force_wide", 1) sends every size there: used by the tests)
    const bool use_wide = !supported_dims(dims->n, dims->m) || g_dbg.force_wide != 0;
    if (use_wide && !altro_wide::WideBackend::supports(dims->n, dims->m)) {
      g_create_err = "unsupported (n, m): the wide kernel holds n <= 64, m <= 32";
      return ALTRO_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
      g_create_err = std::string("no HIP device: ") + hipGetErrorString(e);
      return ALTRO_ERR_HIP;
    }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return ALTRO_ERR_INVALID_ARG; }
    if (use_wide) {
      altro_handle* hw = new (std::nothrow) altro_handle();
      altro_wide::WideBackend* wb = new (std::nothrow) altro_wide::WideBackend();
      if (!hw || !wb) { g_create_err = "out of host memory"; delete hw; delete wb; return ALTRO_ERR_INVALID_ARG; }
      struct WOwner {  // releases both on every path out of this block (exceptions included) unless handed over
        altro_handle* hw;
        altro_wide::WideBackend* wb;
        ~WOwner() { if (wb) { wb->destroy(); delete wb; } delete hw; }
      } wo{hw, wb};
      altro_opts o0;
      if (opts) o0 = *opts; else altro_default_opts(&o0);
      hw->d = *dims;
      hw->o = o0;
      hw->device = device;
      apply_wide_switches(wb);
      const int rc = wb->create(dims, &o0, device);
      if (rc) {
        g_create_err = wb->err;
        return rc;
      }
      hw->wide = wb;
      wo.hw = nullptr;
      wo.wb = nullptr;
      *out = hw;
      return ALTRO_OK;
    }
    altro_handle* h = new (std::nothrow) altro_handle();
    if (!h) { g_create_err = "out of host memory"; return ALTRO_ERR_INVALID_ARG; }
    // anything that throws below (std::vector::assign, std::string) unwinds through this guard: the handle, its stream,
    // events and every array allocated so far are released before guard() turns the exception into an error code
    struct Owner {
      altro_handle* p;
      ~Owner() { if (p) altro_batch_destroy(p); }
    } owner{h};
    h->d = *dims;
    if (opts) h->o = *opts; else altro_default_opts(&h->o);
    h->device = device;
    h->mate = g_dbg.mate;
    h->lone = g_dbg.lone; h->useqz = g_dbg.useqz; h->shadow = g_dbg.shadow; h->resync = g_dbg.resync; h->reuse = g_dbg.reuse; h->group = g_dbg.group;
    h->group_max_steps = g_dbg.group_max_steps; h->dbg_wave = g_dbg.trace_wave; h->debug_keep_gains = g_dbg.keep_gains != 0;
    h->Bp = (dims->batch + IPW - 1) / IPW * IPW;
    auto fail = [&](const char* what, hipError_t er) {
      g_create_err = std::string(what) + ": " + hipGetErrorString(er);
      return ALTRO_ERR_HIP;      // (the Owner releases the handle)
    };
  #define CCHK(call) do { hipError_t e2 = (call); if (e2 != hipSuccess) return fail(#call, e2); } while (0)
    CCHK(hipSetDevice(device));
    CCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CCHK(hipEventCreate(&h->ev0));
    CCHK(hipEventCreate(&h->ev1));
    CCHK(hipEventCreate(&h->bev0));
    CCHK(hipEventCreate(&h->bev1));
    h->ring.reset();
    const size_t Bp = h->Bp, N = dims->N, n = dims->n, m = dims->m;
    const size_t row = Bp * LW;
    // the kernels address every array with 32-bit element offsets
    if ((2 * N + 1) * row * sizeof(double) >= (1ull << 32) || N * Bp * m * LW * sizeof(double) >= (1ull << 32)) {
      g_create_err = "batch * N too large for one handle (arrays must stay below 4 GiB); split the batch";
      return ALTRO_ERR_UNSUPPORTED;
    }
    CCHK(hipMalloc(&h->Gcol, Bp * n * LW * sizeof(double)));
    CCHK(hipMalloc(&h->Grow, Bp * LW * LW * sizeof(double)));
    CCHK(hipMalloc(&h->fvec, row * sizeof(double)));
    CCHK(hipMalloc(&h->wd, LW * sizeof(double)));
    CCHK(hipMalloc(&h->wf, LW * sizeof(double)));
    CCHK(hipMalloc(&h->zmin, LW * sizeof(double)));
    CCHK(hipMalloc(&h->zmax, LW * sizeof(double)));
    CCHK(hipMalloc(&h->x0, row * sizeof(double)));
    // + one trash row at the end of each (stores of rows that sit out a phase land there)
    CCHK(hipMalloc(&h->Z, (2 * N + 1) * row * sizeof(double)));
    for (int j = 0; j < LW; ++j) h->bslot_h[j] = -1;
    h->nbp = 1;
    CCHK(hipMalloc(&h->Lb, (N + 1) * Bp * 2 * h->nbp * sizeof(double)));
    CCHK(hipMalloc(&h->bslot, LW * sizeof(int)));
    CCHK(hipMalloc(&h->Acon, N * LW * LW * sizeof(double)));
    CCHK(hipMalloc(&h->bcon, N * LW * sizeof(double)));
    CCHK(hipMalloc(&h->cmeta, N * LW * 4 * sizeof(int)));
    CCHK(hipMalloc(&h->ckn, LW * sizeof(int)));
    CCHK(hipMemsetAsync(h->ckn, 0, LW * sizeof(int), h->stream));
    CCHK(hipMalloc(&h->lanebuf, LW * sizeof(int)));
    CCHK(hipMalloc(&h->noise_w, LW * sizeof(double)));
    CCHK(hipMalloc(&h->noise_grp, LW * sizeof(int)));
    {
      std::vector<double> w(LW, 0.01);  // 1 % of ||x0||_inf (random_linear_problem.jl:129)
      std::vector<int> g(LW, 0);
      CCHK(hipMemcpyAsync(h->noise_w, w.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
      CCHK(hipMemcpyAsync(h->noise_grp, g.data(), LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
      CCHK(hipStreamSynchronize(h->stream));
    }
    CCHK(hipMalloc(&h->Lc, (N + 1) * row * sizeof(double)));
    h->Acon_h.assign(N * LW * LW, 0.0);
    h->bcon_h.assign(N * LW, 0.0);
    h->cmeta_h.assign(N * LW * 4, 0);
    for (size_t e = 0; e < N * LW; ++e) h->cmeta_h[4 * e + 2] = -1;
    CCHK(hipMemcpyAsync(h->Acon, h->Acon_h.data(), h->Acon_h.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CCHK(hipMemcpyAsync(h->bcon, h->bcon_h.data(), h->bcon_h.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CCHK(hipMemcpyAsync(h->cmeta, h->cmeta_h.data(), h->cmeta_h.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    CCHK(hipMemsetAsync(h->Lc, 0, (N + 1) * row * sizeof(double), h->stream));
    CCHK(hipMemcpyAsync(h->bslot, h->bslot_h, LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
    CCHK(hipMalloc(&h->mu, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->KD, N * Bp * m * LW * sizeof(double)));  // N-1 gain blocks + a trash slot
    CCHK(hipMalloc(&h->Qz, (N + 1) * row * sizeof(double)));
    CCHK(hipMalloc(&h->Dff, (N + 1) * row * sizeof(double)));
    CCHK(hipMemsetAsync(h->Dff, 0, (N + 1) * row * sizeof(double), h->stream));
    CCHK(hipMalloc(&h->ahash, row * sizeof(altro::ASet)));
    CCHK(hipMemsetAsync(h->ahash, 0, row * sizeof(altro::ASet), h->stream));
    CCHK(hipMalloc(&h->kmu, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->n_fo, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->pn_ran, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->pn_failed, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->pn_res, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->pn_dfail, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->pn_dres0, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->pn_dres, Bp * sizeof(double)));
    CCHK(hipMemsetAsync(h->pn_dfail, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->pn_dres0, 0, Bp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->pn_dres, 0, Bp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->pn_ran, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->pn_failed, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->pn_res, 0, Bp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->n_fo, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->Qz, 0, (N + 1) * row * sizeof(double), h->stream));
    CCHK(hipMalloc(&h->cur, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->perm, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->gscore, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->iters, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->iters_outer, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->status, Bp * sizeof(int)));
    CCHK(hipMalloc(&h->cost, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->cmax, Bp * sizeof(double)));
    CCHK(hipMalloc(&h->Jtrace, Bp * ALTRO_TRACE_LEN * sizeof(double)));
    CCHK(hipMalloc(&h->ctrace, Bp * ALTRO_TRACE_LEN * sizeof(double)));
    CCHK(hipMalloc(&h->atrace, Bp * ALTRO_TRACE_LEN * sizeof(double)));
    CCHK(hipMemsetAsync(h->atrace, 0, Bp * ALTRO_TRACE_LEN * sizeof(double), h->stream));
    CCHK(hipMalloc(&h->n_backward, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->n_rollout, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->wave_cycles, (Bp * 4 + 4096) * sizeof(long long)));
    CCHK(hipMalloc(&h->simd_tab, (size_t)65536 * 16 * sizeof(unsigned)));
    CCHK(hipMemsetAsync(h->simd_tab, 0, (size_t)65536 * 16 * sizeof(unsigned), h->stream));
    CCHK(hipMalloc(&h->n_solves, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->n_iters, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->n_ok, Bp * sizeof(long long)));
    CCHK(hipMalloc(&h->n_trials, Bp * sizeof(long long)));
    CCHK(hipMemsetAsync(h->n_trials, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMalloc(&h->n_gconf, Bp * sizeof(long long)));
    CCHK(hipMemsetAsync(h->n_gconf, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMalloc(&h->dzero, Bp * sizeof(int)));
    CCHK(hipMemsetAsync(h->dzero, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->n_solves, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->n_iters, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->n_ok, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->wave_cycles, 0, (Bp * 4 + 4096) * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->n_backward, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->n_rollout, 0, Bp * sizeof(long long), h->stream));
    CCHK(hipMemsetAsync(h->Z, 0, (2 * N + 1) * row * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->Lb, 0, (N + 1) * Bp * 2 * h->nbp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->KD, 0, N * Bp * m * LW * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->cur, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->iters, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->iters_outer, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->status, 0, Bp * sizeof(int), h->stream));
    CCHK(hipMemsetAsync(h->cost, 0, Bp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->cmax, 0, Bp * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->Jtrace, 0, Bp * ALTRO_TRACE_LEN * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->ctrace, 0, Bp * ALTRO_TRACE_LEN * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->x0, 0, row * sizeof(double), h->stream));
    CCHK(hipMemsetAsync(h->fvec, 0, row * sizeof(double), h->stream));
    {
      // no bounds until a BOX constraint is added
      std::vector<double> lo(LW, -INFINITY), hi(LW, INFINITY);
      CCHK(hipMemcpyAsync(h->zmin, lo.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
      CCHK(hipMemcpyAsync(h->zmax, hi.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
      CCHK(hipStreamSynchronize(h->stream));
    }
    hipLaunchKernelGGL(k_fill, grid_for(Bp), dim3(256), 0, h->stream, h->mu, 1.0, (size_t)Bp);
    hipLaunchKernelGGL(k_fill, grid_for(Bp), dim3(256), 0, h->stream, h->kmu, -1.0, (size_t)Bp);  // no gains yet
    CCHK(hipStreamSynchronize(h->stream));
  #undef CCHK
    owner.p = nullptr;
    *out = h;
    return ALTRO_OK;
  });
}

// release everything the 16-lane backend owns on the device (the handle itself stays)
static void free_dpp_backend(altro_handle* h) {
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  void** ptrs[] = {(void**)&h->Gcol, (void**)&h->Grow, (void**)&h->fvec, (void**)&h->wd, (void**)&h->wf, (void**)&h->zmin, (void**)&h->zmax,
                   (void**)&h->x0, (void**)&h->Zref, (void**)&h->Z, (void**)&h->Lb, (void**)&h->bslot, (void**)&h->Acon, (void**)&h->bcon,
                   (void**)&h->cmeta, (void**)&h->ckn, (void**)&h->Lc, (void**)&h->lanebuf, (void**)&h->noise_w, (void**)&h->noise_grp, (void**)&h->mu,
                   (void**)&h->KD, (void**)&h->noise, (void**)&h->cur, (void**)&h->iters, (void**)&h->iters_outer, (void**)&h->status,
                   (void**)&h->cost, (void**)&h->cmax, (void**)&h->Jtrace, (void**)&h->ctrace, (void**)&h->atrace, (void**)&h->stage,
                   (void**)&h->n_backward, (void**)&h->n_rollout, (void**)&h->wave_cycles, (void**)&h->simd_tab, (void**)&h->n_solves, (void**)&h->n_iters,
                   (void**)&h->n_ok, (void**)&h->n_trials, (void**)&h->Zsave, (void**)&h->n_gconf, (void**)&h->dzero, (void**)&h->Qz,
                   (void**)&h->Dff, (void**)&h->ahash, (void**)&h->kmu, (void**)&h->n_fo, (void**)&h->perm, (void**)&h->gscore,
                   (void**)&h->pn_ran, (void**)&h->pn_failed, (void**)&h->pn_res, (void**)&h->pn_dfail, (void**)&h->pn_dres0, (void**)&h->pn_dres, (void**)&h->pnE, (void**)&h->pndv, (void**)&h->pnLd, (void**)&h->pnLo,
                   (void**)&h->pnvec, (void**)&h->pntz, (void**)&h->pnnb, (void**)&h->pnnst, (void**)&h->pnrinfo};
  for (void** p : ptrs)
    if (*p) { hipFree(*p); *p = nullptr; }
  h->stage_bytes = 0;
  h->ring.destroy();
  if (h->bev0) { hipEventDestroy(h->bev0); h->bev0 = nullptr; }
  if (h->bev1) { hipEventDestroy(h->bev1); h->bev1 = nullptr; }
  if (h->ev0) { hipEventDestroy(h->ev0); h->ev0 = nullptr; }
  if (h->ev1) { hipEventDestroy(h->ev1); h->ev1 = nullptr; }
  if (h->stream) { hipStreamDestroy(h->stream); h->stream = nullptr; }
}

int32_t altro_batch_destroy(altro_handle* h) {
  if (!h) return ALTRO_OK;
  if (h->wide) {
    h->wide->destroy();
    delete h->wide;
    delete h;
    return ALTRO_OK;
  }
  free_dpp_backend(h);
  delete h;
  return ALTRO_OK;
}

// Per-knot (LTV) dynamics exist on the one-wave-per-instance kernel only.  A 16-lane handle on which nothing
// but create has happened moves there; the Julia model is fixed when ALTROSolver(prob, opts) is built
// (ALTROParams.jl:61,96), so set_dynamics is the first call of every harness.
static int migrate_to_wide(altro_handle* h) {
  if (h->have_cost || h->have_ref || h->have_dyn || h->ncon > 0 || h->timed)
    FAIL(h, ALTRO_ERR_UNSUPPORTED, "per-knot dynamics on an (n, m) of the 16-lane kernel set: call altro_batch_set_dynamics "
                                   "first after altro_batch_create (or set ALTRO_FORCE_WIDE=1)");
  altro_wide::WideBackend* wb = new (std::nothrow) altro_wide::WideBackend();
  if (!wb) FAIL(h, ALTRO_ERR_INTERNAL, "out of host memory");
  apply_wide_switches(wb);
  // the wide backend is created BEFORE the 16-lane one is released: if it cannot be (e.g. no device
  // memory for its arrays) the handle stays a working 16-lane handle and only this call fails
  const int rc = wb->create(&h->d, &h->o, h->device);
  if (rc) {
    h->err = wb->err;
    wb->destroy();
    delete wb;
    return rc;
  }
  if (h->have_x0) {  // an initial state uploaded before the model: carried over, not dropped
    std::vector<double> x((size_t)h->d.batch * h->d.n);
    int rcx = ensure_stage(h, x.size() * sizeof(double));
    if (!rcx) {
      hipLaunchKernelGGL(k_unpack_x0, grid_for((size_t)h->d.batch * LW), dim3(256), 0, h->stream, h->stage, h->x0, h->d.batch, h->d.n);
      if (hipMemcpyAsync(x.data(), h->stage, x.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipStreamSynchronize(h->stream) != hipSuccess) rcx = ALTRO_ERR_HIP;
    }
    if (!rcx) rcx = wb->set_initial_state(x.data());
    if (rcx) {
      h->err = rcx == ALTRO_ERR_HIP ? "copying the initial state to the wide backend failed" : wb->err;
      wb->destroy();
      delete wb;
      return rcx;
    }
  }
  free_dpp_backend(h);
  h->wide = wb;
  return ALTRO_OK;
}

int32_t altro_batch_set_dynamics(altro_handle* h, const double* A, const double* B, const double* f,
                                 int32_t per_knot, int32_t per_instance) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, set_dynamics(A, B, f, per_knot, per_instance));
    if (!h || !A || !B) return ALTRO_ERR_INVALID_ARG;
    if (per_knot) {
      const int rc = migrate_to_wide(h);
      if (rc) return rc;
      const int rc2 = h->wide->set_dynamics(A, B, f, per_knot, per_instance);
      if (rc2) h->err = h->wide->err;
      return rc2;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = h->d.n, m = h->d.m;
    const size_t nb = per_instance ? h->d.batch : 1;
    const size_t tot = nb * (n * n + n * m + n);
    int rc = ensure_stage(h, tot * sizeof(double));
    if (rc) return rc;
    if ((rc = upload(h, A, nb * n * n, 0))) return rc;
    if ((rc = upload(h, B, nb * n * m, nb * n * n))) return rc;
    if (f && (rc = upload(h, f, nb * n, nb * (n * n + n * m)))) return rc;
    hipLaunchKernelGGL(k_pack_dyn, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->stage,
                       h->stage + nb * n * n, f ? h->stage + nb * (n * n + n * m) : nullptr, h->Gcol, h->Grow, h->fvec,
                       h->d.batch, h->Bp, (int)n, (int)m, per_instance ? 1 : 0);
    HIPCHK(h, hipGetLastError());
    if (int rcd = drop_gains(h)) return rcd;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_dyn = true;
    h->dyn_per_instance = per_instance != 0;
    return ALTRO_OK;
  });
}

int32_t altro_batch_set_tracking_cost(altro_handle* h, const double* Qd, const double* Rd, const double* Qfd, double dt) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, set_tracking_cost(Qd, Rd, Qfd, dt));
    if (!h || !Qd || !Rd || !Qfd || !(dt > 0)) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const int n = h->d.n, m = h->d.m;
    std::vector<double> wd(LW, 0.0), wf(LW, 0.0);
    for (int j = 0; j < n; ++j) { wd[j] = dt * Qd[j]; wf[j] = Qfd[j]; }
    for (int j = 0; j < m; ++j) wd[n + j] = dt * Rd[j];
    HIPCHK(h, hipMemcpyAsync(h->wd, wd.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->wf, wf.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (int rcd = drop_gains(h)) return rcd;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dt = dt;
    h->have_cost = true;
    return ALTRO_OK;
  });
}

// Pack the recorded LINEAR / SOC constraints onto the 16 constraint-row lanes and fill the per-knot
// tables.  A constraint keeps the same lanes over its whole knot range (shift_fill moves duals
// along the knot axis); two constraints may share lanes when their ranges do not overlap (e.g. the
// goal at knot N-1 and the stage constraints on 0..N-2).  Every cone takes the first p lanes of an
// aligned quad, linear rows fill whatever lanes remain (also the spare lanes of a cone's quad).
// Redone whenever a constraint is added before the first solve.
static int pack_constraints(altro_handle* h) {
  if (!h->con_dirty) return ALTRO_OK;
  const int nz = h->d.n + h->d.m, N = h->d.N;
  // one table per instance as soon as any block carries per-instance data (grasp_mpc_helpers.jl:46-55 mutates each
  // problem's own per-knot tables); the lane assignment (cmeta) is common to the batch either way
  h->con_per_instance = false;
  for (const auto& cb : h->cons) h->con_per_instance = h->con_per_instance || cb.per_instance;
  const size_t ninst = h->con_per_instance ? (size_t)h->Bp : 1, B = h->d.batch;
  if (ninst * N * LW * LW * sizeof(double) >= (1ull << 32)) FAIL(h, ALTRO_ERR_UNSUPPORTED, "per-instance constraint tables of this batch exceed 4 GiB; split the batch");
  h->Acon_h.assign(ninst * N * LW * LW, 0.0);
  h->bcon_h.assign(ninst * N * LW, 0.0);
  std::fill(h->cmeta_h.begin(), h->cmeta_h.end(), 0);
  for (size_t e = 0; e < (size_t)N * LW; ++e) h->cmeta_h[4 * e + 2] = -1;
  std::vector<char> used((size_t)N * LW, 0);      // lane taken at knot k
  std::vector<char> quad_soc((size_t)N * 4, 0);   // quad holds a cone at knot k
  auto lane_free = [&](int lane, int k0, int k1) {
    for (int k = k0; k <= k1; ++k) if (used[(size_t)k * LW + lane]) return false;
    return true;
  };
  auto place = [&](altro_handle::ConBlock& cb, int r, int lane, int type, int pdim) {
    cb.lanes[r] = lane;
    const size_t nk = cb.per_knot ? (size_t)(cb.k1 - cb.k0 + 1) : 1;
    for (int k = cb.k0; k <= cb.k1; ++k) {
      const size_t e = (size_t)k * LW + lane;
      used[e] = 1;
      for (size_t ib = 0; ib < ninst; ++ib) {
        const size_t src = ib < B ? ib : B - 1;      // padded slots mirror the last instance
        const size_t blk = (cb.per_instance ? src * nk : 0) + (cb.per_knot ? (size_t)(k - cb.k0) : 0);
        const size_t et = ib * N * LW + e;
        for (int jj = 0; jj < nz; ++jj) h->Acon_h[et * LW + jj] = cb.A[(blk * cb.p + r) * nz + jj];
        h->bcon_h[et] = cb.b[blk * cb.p + r];
      }
      h->cmeta_h[4 * e + 0] = type;
      h->cmeta_h[4 * e + 1] = cb.k0;
      h->cmeta_h[4 * e + 2] = cb.k1;
      h->cmeta_h[4 * e + 3] = pdim;
    }
  };
  for (auto& cb : h->cons) {
    if (cb.kind != ALTRO_CON_SOC) continue;
    int q = -1;
    for (int c = 0; c < 4 && q < 0; ++c) {
      bool ok = true;
      for (int k = cb.k0; k <= cb.k1 && ok; ++k) ok = !quad_soc[(size_t)k * 4 + c];
      for (int r = 0; r < cb.p && ok; ++r) ok = lane_free(4 * c + r, cb.k0, cb.k1);
      if (ok) q = c;
    }
    if (q < 0) FAIL(h, ALTRO_ERR_UNSUPPORTED, "no free quad of constraint-row lanes for a second-order cone (4 cones per knot)");
    for (int k = cb.k0; k <= cb.k1; ++k) quad_soc[(size_t)k * 4 + q] = 1;
    for (int r = 0; r < cb.p; ++r) place(cb, r, 4 * q + r, 3, cb.p);
  }
  for (auto& cb : h->cons) {
    if (cb.kind != ALTRO_CON_LINEAR) continue;
    int lane = 0;
    for (int r = 0; r < cb.p; ++r) {
      while (lane < LW && !lane_free(lane, cb.k0, cb.k1)) ++lane;
      if (lane == LW) FAIL(h, ALTRO_ERR_UNSUPPORTED, "more than 16 constraint rows at one knot");
      place(cb, r, lane, cb.sense == ALTRO_SENSE_EQ ? 1 : 2, 0);
    }
  }
  // every lane of a quad carries the dimension of the cone the quad holds at that knot (a linear
  // row in a spare lane is excluded from the cone by pos >= p)
  for (int k = 0; k < N; ++k)
    for (int q = 0; q < 4; ++q) {
      int pdim = 0;
      for (int i = 0; i < 4; ++i) {
        const size_t e = (size_t)k * LW + 4 * q + i;
        if (h->cmeta_h[4 * e] == 3) pdim = h->cmeta_h[4 * e + 3];
      }
      for (int i = 0; i < 4; ++i) h->cmeta_h[4 * ((size_t)k * LW + 4 * q + i) + 3] = pdim;
    }
  h->ncrows = h->cons.empty() ? 0 : LW;
  if (h->Acon_h.size() != h->acon_elems) {  // the table changed shape (per-instance data arrived): reallocate
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->Acon) HIPCHK(h, hipFree(h->Acon));
    if (h->bcon) HIPCHK(h, hipFree(h->bcon));
    h->Acon = h->bcon = nullptr;
    HIPCHK(h, hipMalloc(&h->Acon, h->Acon_h.size() * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->bcon, h->bcon_h.size() * sizeof(double)));
    h->acon_elems = h->Acon_h.size();
  }
  HIPCHK(h, hipMemcpyAsync(h->Acon, h->Acon_h.data(), h->Acon_h.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->bcon, h->bcon_h.data(), h->bcon_h.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->cmeta, h->cmeta_h.data(), h->cmeta_h.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  // Time-invariant tables (rocket landing: every constraint has ONE block of data and its own lanes): the row a lane
  // holds is the same at every knot of its range, so the streaming sweeps load it once per sweep from one canonical
  // knot instead of once per knot (solve_dpp16.h trial_costs).  Decided on the packed tables themselves.
  {
    bool inv = !h->con_per_instance;
    int ckn[LW];
    for (int lane = 0; lane < LW; ++lane) {
      int first = -1;
      for (int k = 0; k < N && inv; ++k) {
        const size_t e = (size_t)k * LW + lane;
        if (h->cmeta_h[4 * e] == 0) continue;
        if (first < 0) { first = k; continue; }
        const size_t f = (size_t)first * LW + lane;
        for (int q = 0; q < 4; ++q) inv = inv && h->cmeta_h[4 * e + q] == h->cmeta_h[4 * f + q];
        inv = inv && h->bcon_h[e] == h->bcon_h[f];
        for (int jj = 0; jj < LW; ++jj) inv = inv && h->Acon_h[e * LW + jj] == h->Acon_h[f * LW + jj];
      }
      ckn[lane] = first < 0 ? 0 : first;
    }
    // a lane without rows must be empty at its canonical knot 0 too: true by construction (it is empty everywhere)
    h->con_inv = inv ? 1 : 0;
    HIPCHK(h, hipMemcpyAsync(h->ckn, ckn, LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->con_dirty = false;
  return ALTRO_OK;
}

int32_t altro_batch_add_constraint(altro_handle* h, int32_t kind, int32_t sense, int32_t k_first, int32_t k_last,
                                   int32_t p, const double* A, const double* b, const double* zmin, const double* zmax,
                                   int32_t per_knot, int32_t* con_id) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, add_constraint(kind, sense, k_first, k_last, p, A, b, zmin, zmax, per_knot, con_id));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    if (k_first < 0 || k_last >= h->d.N || k_last < k_first) FAIL(h, ALTRO_ERR_INVALID_ARG, "bad knot range");
    HIPCHK(h, hipSetDevice(h->device));
    const int nz = h->d.n + h->d.m;
    if (kind == ALTRO_CON_LINEAR || kind == ALTRO_CON_SOC) {
      if (!A || !b || p < 1) return ALTRO_ERR_INVALID_ARG;
      if (kind == ALTRO_CON_SOC && (p < 2 || p > 4)) FAIL(h, ALTRO_ERR_UNSUPPORTED, "second-order cones of dimension 2..4 are built");
      if (kind == ALTRO_CON_LINEAR && sense != ALTRO_SENSE_EQ && sense != ALTRO_SENSE_INEQ) return ALTRO_ERR_INVALID_ARG;
      if (h->con_locked) FAIL(h, ALTRO_ERR_STATE, "constraints must be added before the first solve");
      altro_handle::ConBlock cb;
      cb.id = h->ncon; cb.kind = kind; cb.sense = sense; cb.k0 = k_first; cb.k1 = k_last; cb.p = p;
      cb.per_knot = (per_knot & 1) ? 1 : 0;
      cb.per_instance = (per_knot & 2) ? 1 : 0;
      const size_t nblk = (cb.per_knot ? (size_t)(k_last - k_first + 1) : 1) * (cb.per_instance ? (size_t)h->d.batch : 1);
      cb.A.assign(A, A + nblk * p * nz);
      cb.b.assign(b, b + nblk * p);
      for (int r = 0; r < LW; ++r) cb.lanes[r] = -1;
      h->cons.push_back(cb);
      h->con_dirty = true;
      int rc = pack_constraints(h);
      if (rc) { h->cons.pop_back(); h->con_dirty = true; pack_constraints(h); return rc; }
      if (con_id) *con_id = h->ncon;
      h->ncon++;
      return ALTRO_OK;
    }
    if (kind != ALTRO_CON_BOX) return ALTRO_ERR_INVALID_ARG;
    if (h->box_id >= 0) FAIL(h, ALTRO_ERR_UNSUPPORTED, "one BOX constraint per problem");
    if (!zmin || !zmax) return ALTRO_ERR_INVALID_ARG;
    std::vector<double> lo(LW, -INFINITY), hi(LW, INFINITY);
    for (int j = 0; j < nz; ++j) { lo[j] = zmin[j]; hi[j] = zmax[j]; }
    HIPCHK(h, hipMemcpyAsync(h->zmin, lo.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->zmax, hi.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // compact dual rows: one slot per element with at least one finite bound
    int nb = 0;
    for (int j = 0; j < LW; ++j) h->bslot_h[j] = (j < nz && (std::isfinite(lo[j]) || std::isfinite(hi[j]))) ? nb++ : -1;
    h->nbp = nb > 0 ? nb : 1;
    HIPCHK(h, hipMemcpyAsync(h->bslot, h->bslot_h, LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(h->Lb));
    h->Lb = nullptr;
    {
      const size_t lbytes = (size_t)(h->d.N + 1) * h->Bp * 2 * h->nbp * sizeof(double);
      HIPCHK(h, hipMalloc(&h->Lb, lbytes));
      HIPCHK(h, hipMemsetAsync(h->Lb, 0, lbytes, h->stream));
    }
    if (int rcd = drop_gains(h)) return rcd;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->box_k0 = k_first;
    h->box_k1 = k_last;
    h->box_id = h->ncon++;
    if (con_id) *con_id = h->box_id;
    return ALTRO_OK;
  });
}

int32_t altro_batch_update_constraint_data(altro_handle* h, int32_t con_id, const double* A, const double* b) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, update_constraint_data(con_id, A, b));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const int nz = h->d.n + h->d.m;
    for (auto& cb : h->cons) {
      if (cb.id != con_id) continue;
      const size_t nblk = (cb.per_knot ? (size_t)(cb.k1 - cb.k0 + 1) : 1) * (cb.per_instance ? (size_t)h->d.batch : 1);
      if (A) cb.A.assign(A, A + nblk * cb.p * nz);
      if (b) cb.b.assign(b, b + nblk * cb.p);
      // same lanes, new coefficients: refresh the tables (the solver sees it at the next solve, as
      // the reference's in-place mutation does: grasp_mpc_helpers.jl:46-55)
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->con_dirty = true;
      return pack_constraints(h);
    }
    FAIL(h, ALTRO_ERR_INVALID_ARG, "unknown or non-affine constraint id");
  });
}

int32_t altro_batch_set_initial_state(altro_handle* h, const double* x0) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, set_initial_state(x0));
    if (!h || !x0) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t cnt = (size_t)h->d.batch * h->d.n;
    int rc = ensure_stage(h, cnt * sizeof(double));
    if (rc) return rc;
    if ((rc = upload(h, x0, cnt))) return rc;
    hipLaunchKernelGGL(k_pack_x0, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->stage, h->x0, h->d.batch,
                       h->Bp, h->d.n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_x0 = true;
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_initial_state(altro_handle* h, double* x0) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_initial_state(x0));
    if (!h || !x0) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t cnt = (size_t)h->d.batch * h->d.n;
    int rc = ensure_stage(h, cnt * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_x0, grid_for((size_t)h->d.batch * LW), dim3(256), 0, h->stream, h->stage, h->x0,
                       h->d.batch, h->d.n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(x0, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

static int set_ref_common(altro_handle* h, const double* Xref, const double* Uref, int Nt) {
  const size_t B = h->d.batch, n = h->d.n, m = h->d.m;
  const size_t cx = B * Nt * n, cu = B * (Nt - 1) * m;
  int rc = ensure_stage(h, (cx + cu) * sizeof(double));
  if (rc) return rc;
  if ((size_t)Nt * h->Bp * LW * sizeof(double) >= (1ull << 32)) FAIL(h, ALTRO_ERR_UNSUPPORTED, "reference trajectory too large for one handle (below 4 GiB)");
  if (h->Nt != Nt) {
    if (h->Zref) HIPCHK(h, hipFree(h->Zref));
    h->Zref = nullptr;
    HIPCHK(h, hipMalloc(&h->Zref, (size_t)Nt * h->Bp * LW * sizeof(double)));
    h->Nt = Nt;
  }
  if ((rc = upload(h, Xref, cx, 0))) return rc;
  if ((rc = upload(h, Uref, cu, cx))) return rc;
  hipLaunchKernelGGL(k_pack_ref, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->stage, h->stage + cx,
                     h->Zref, h->d.batch, h->Bp, Nt, (int)n, (int)m);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->kref = 0;
  h->have_ref = true;
  return ALTRO_OK;
}

int32_t altro_batch_set_reference(altro_handle* h, const double* Xref, const double* Uref) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, set_reference(Xref, Uref));
    if (!h || !Xref || !Uref) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    return set_ref_common(h, Xref, Uref, h->d.N);
  });
}

int32_t altro_batch_set_initial_trajectory(altro_handle* h, const double* X, const double* U) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, set_initial_trajectory(X, U));
    if (!h || !U) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t B = h->d.batch, N = h->d.N, n = h->d.n, m = h->d.m;
    const size_t cx = X ? B * N * n : 0, cu = B * (N - 1) * m;
    int rc = ensure_stage(h, (cx + cu) * sizeof(double));
    if (rc) return rc;
    if (X && (rc = upload(h, X, cx, 0))) return rc;
    if ((rc = upload(h, U, cu, cx))) return rc;
    const size_t plane = N * (size_t)LW;   // offset of plane 1 inside an instance's block of Z
    hipLaunchKernelGGL(k_pack_traj, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->stage, h->stage + cx, h->Z,
                       h->cur, plane, (int)B, h->Bp, (int)N, (int)n, (int)m, 1, X ? 1 : 0);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

int32_t altro_batch_shift_fill(altro_handle* h, int32_t primal, int32_t dual) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, shift_fill(primal, dual));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t plane = (size_t)h->d.N * LW;
    hipLaunchKernelGGL(k_shift, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->Z, h->cur, plane, h->Lb,
                       h->nbp, h->Bp, h->d.N, h->d.n, h->d.m, h->box_k0, h->box_k1, primal ? 1 : 0, dual ? 1 : 0, h->Lc,
                       h->cmeta, h->ncrows);
    HIPCHK(h, hipGetLastError());
    return ALTRO_OK;
  });
}

int32_t altro_batch_set_options(altro_handle* h, const altro_opts* o) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide && o) {
      h->wide->o = *o; h->wide->gains_valid = false; h->o = *o; return ALTRO_OK;
    }
    if (!h || !o) return ALTRO_ERR_INVALID_ARG;
    h->o = *o;
    HIPCHK(h, hipSetDevice(h->device));
    return drop_gains(h);
  });
}

// solve!(::ProjectedNewtonSolver) after the AL kernel of a plain solve (altro_opts.projected_newton).
// prepare_polish: every check and allocation the polish needs -- run BEFORE a launch takes its slot of the timing ring, so
// that a failure leaves no slot with a start event and no end event.
static int prepare_polish(altro_handle* h) {
  const size_t Bp = h->Bp, N = h->d.N;
  int nbounded = 0;
  for (int j = 0; j < LW; ++j) nbounded += h->bslot_h[j] >= 0 ? 1 : 0;
  const int bm = 2 * h->d.n + (h->box_k1 >= h->box_k0 ? 2 * nbounded : 0) + (h->ncrows > 0 ? LW : 0);
  if (bm > altro_pn::BMAX) FAIL(h, ALTRO_ERR_UNSUPPORTED, "projected_newton: more rows per knot than pn_polish.h holds");
  if (h->pn_bm != bm || !h->pnE) {
    void** ws[] = {(void**)&h->pnE, (void**)&h->pndv, (void**)&h->pnLd, (void**)&h->pnLo, (void**)&h->pnvec, (void**)&h->pntz,
                   (void**)&h->pnnb, (void**)&h->pnnst, (void**)&h->pnrinfo};
    for (void** q : ws) if (*q) { HIPCHK(h, hipFree(*q)); *q = nullptr; }
    HIPCHK(h, hipMalloc(&h->pnE, Bp * N * bm * LW * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pndv, Bp * N * bm * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pnLd, Bp * N * bm * bm * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pnLo, Bp * N * bm * bm * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pnvec, Bp * 6 * N * bm * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pntz, Bp * 3 * N * LW * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->pnnb, Bp * N * sizeof(int)));
    HIPCHK(h, hipMalloc(&h->pnnst, Bp * N * sizeof(int)));
    HIPCHK(h, hipMalloc(&h->pnrinfo, Bp * N * bm * sizeof(int)));
    h->pn_bm = bm;
  }
  return ALTRO_OK;
}

static int launch_polish(altro_handle* h) {
  const int bm = h->pn_bm;
  altro_pn::PnParams q{};
  q.B = h->d.batch; q.Bp = h->Bp; q.N = h->d.N; q.Nt = h->Nt; q.n = h->d.n; q.m = h->d.m; q.bm = bm;
  q.box_k0 = h->box_k0; q.box_k1 = h->box_k1; q.ncrows = h->ncrows;
  q.con_istride = h->con_per_instance ? (unsigned)(h->d.N * LW * LW) : 0u;
  q.Grow = h->Grow; q.fvec = h->fvec; q.wd = h->wd; q.wf = h->wf; q.zmin = h->zmin; q.zmax = h->zmax; q.x0 = h->x0;
  q.Acon = h->Acon; q.bcon = h->bcon; q.cmeta = h->cmeta;
  q.Z = h->Z; q.Zref = h->Zref; q.kref = h->kref; q.cur = h->cur; q.status = h->status; q.cost = h->cost; q.cmax = h->cmax;
  q.pn_ran = h->pn_ran; q.pn_failed = h->pn_failed; q.pn_res = h->pn_res;
  q.Lb = h->Lb; q.Lc = h->Lc; q.bslot = h->bslot; q.nbp = h->nbp;
  q.pn_dfail = h->pn_dfail; q.pn_dres0 = h->pn_dres0; q.pn_dres = h->pn_dres;
  q.E = h->pnE; q.dv = h->pndv; q.Ld = h->pnLd; q.Lo = h->pnLo; q.vec = h->pnvec; q.tz = h->pntz;
  q.nb = h->pnnb; q.nst = h->pnnst; q.rinfo = h->pnrinfo;
  q.o = h->o;
  hipLaunchKernelGGL(altro_pn::pn_kernel, dim3(h->d.batch), dim3(64), 0, h->stream, q);
  HIPCHK(h, hipGetLastError());
  return ALTRO_OK;
}

static int enqueue_solve(altro_handle* h, int first_step, int nsteps) {
  HIPCHK(h, hipSetDevice(h->device));
  int rc = check_ready(h);
  if (rc) return rc;
  if ((rc = pack_constraints(h))) return rc;
  h->con_locked = true;
  const int last_kref = nsteps > 0 ? first_step + nsteps : h->kref;
  if (last_kref + h->d.N > h->Nt) FAIL(h, ALTRO_ERR_STATE, "reference window runs past the end of the stored trajectory");
  // everything that can refuse the launch comes before it takes a slot of the timing ring
  if (h->o.projected_newton && (rc = prepare_polish(h))) return rc;
  if (!supported_dims(h->d.n, h->d.m)) FAIL(h, ALTRO_ERR_UNSUPPORTED, "no kernel built for this (n, m)");
  hipEvent_t h0, h1;
  HIPCHK(h, h->ring.next(&h0, &h1));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, hipEventRecord(h0, h->stream));
  if (h->o.projected_newton && nsteps > 0) {
    // solve!(::ALTROSolver) ends with the polish, and the next step's shift starts from what it left: the steps of a fused
    // launch run as nsteps pairs of (one-step solve kernel, polish kernel) on the stream, the polish over the step's window
    const int kref0 = h->kref;
    for (int s = 0; s < nsteps && !rc; ++s) {
      rc = launch_solve(h, first_step + s, 1);
      h->kref = first_step + s + 1;
      if (!rc) rc = launch_polish(h);
    }
    if (rc) h->kref = kref0;
  } else {
    rc = launch_solve(h, first_step, nsteps);
    if (!rc && h->o.projected_newton) rc = launch_polish(h);
  }
  // (a launch that failed after all -- a HIP error -- still closes its slot: every slot handed out has both events)
  HIPCHK(h, hipEventRecord(h1, h->stream));
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  if (nsteps > 0) h->kref = first_step + nsteps;
  return ALTRO_OK;
}

int32_t altro_batch_solve_async(altro_handle* h) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, enqueue(0, 0, 0));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    return enqueue_solve(h, 0, 0);
  });
}

int32_t altro_batch_synchronize(altro_handle* h) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, synchronize());
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

int32_t altro_batch_solve(altro_handle* h) {
  return guard(h, [&]() -> int32_t {
    int rc = altro_batch_solve_async(h);
    if (rc) return rc;
    return altro_batch_synchronize(h);
  });
}

static int get_traj(altro_handle* h, double* X, double* U) {
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = h->d.batch, N = h->d.N, n = h->d.n, m = h->d.m;
  const size_t cx = B * N * n, cu = B * (N - 1) * m;
  int rc = ensure_stage(h, (cx + cu) * sizeof(double));
  if (rc) return rc;
  const size_t plane = N * (size_t)LW;
  hipLaunchKernelGGL(k_unpack_traj, grid_for(B * LW), dim3(256), 0, h->stream, X ? h->stage : nullptr,
                     U ? h->stage + cx : nullptr, h->Z, h->cur, plane, (int)B, h->Bp, (int)N, (int)n, (int)m);
  HIPCHK(h, hipGetLastError());
  if (X) HIPCHK(h, hipMemcpyAsync(X, h->stage, cx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (U) HIPCHK(h, hipMemcpyAsync(U, h->stage + cx, cu * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ALTRO_OK;
}

int32_t altro_batch_get_states(altro_handle* h, double* X) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_planes(X, nullptr));
    if (!h || !X) return ALTRO_ERR_INVALID_ARG;
    return get_traj(h, X, nullptr);
  });
}

int32_t altro_batch_get_controls(altro_handle* h, double* U) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_planes(nullptr, U));
    if (!h || !U) return ALTRO_ERR_INVALID_ARG;
    return get_traj(h, nullptr, U);
  });
}

static int duals_xfer(altro_handle* h, int32_t con_id, double* lambda, int to_host) {
  HIPCHK(h, hipSetDevice(h->device));
  for (const auto& cb : h->cons) {
    if (cb.id != con_id) continue;
    const size_t nk = cb.k1 - cb.k0 + 1;
    const size_t cnt = (size_t)h->d.batch * nk * cb.p;
    int rc = ensure_stage(h, cnt * sizeof(double));
    if (rc) return rc;
    if (!to_host && (rc = upload(h, lambda, cnt))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->lanebuf, cb.lanes, LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_cduals, grid_for((size_t)h->d.batch * cb.p), dim3(256), 0, h->stream, h->stage, h->Lc,
                       h->d.batch, h->d.N, h->lanebuf, cb.p, cb.k0, cb.k1, to_host);
    HIPCHK(h, hipGetLastError());
    if (to_host) HIPCHK(h, hipMemcpyAsync(lambda, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  }
  if (con_id != h->box_id || h->box_id < 0) FAIL(h, ALTRO_ERR_INVALID_ARG, "unknown constraint id");
  const int nz = h->d.n + h->d.m;
  const size_t nk = h->box_k1 - h->box_k0 + 1;
  const size_t cnt = (size_t)h->d.batch * nk * 2 * nz;
  int rc = ensure_stage(h, cnt * sizeof(double));
  if (rc) return rc;
  if (!to_host && (rc = upload(h, lambda, cnt))) return rc;
  hipLaunchKernelGGL(k_duals, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->stage, h->Lb, h->bslot, h->nbp,
                     h->d.batch, h->Bp, h->d.N, nz, h->box_k0, h->box_k1, to_host);
  HIPCHK(h, hipGetLastError());
  if (to_host) HIPCHK(h, hipMemcpyAsync(lambda, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ALTRO_OK;
}

int32_t altro_batch_get_duals(altro_handle* h, int32_t con_id, double* lambda) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, duals(con_id, lambda, false));
    if (!h || !lambda) return ALTRO_ERR_INVALID_ARG;
    return duals_xfer(h, con_id, lambda, 1);
  });
}

int32_t altro_batch_set_duals(altro_handle* h, int32_t con_id, const double* lambda) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, duals(con_id, const_cast<double*>(lambda), true));
    if (!h || !lambda) return ALTRO_ERR_INVALID_ARG;
    return duals_xfer(h, con_id, const_cast<double*>(lambda), 0);
  });
}

int32_t altro_batch_get_stats(altro_handle* h, int32_t* iterations, int32_t* iterations_outer, int32_t* status,
                              double* cost, double* c_max, double* cost_trace, double* cmax_trace) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_stats(iterations, iterations_outer, status, cost, c_max, cost_trace, cmax_trace));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t B = h->d.batch;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (iterations) HIPCHK(h, hipMemcpy(iterations, h->iters, B * sizeof(int), hipMemcpyDeviceToHost));
    if (iterations_outer) HIPCHK(h, hipMemcpy(iterations_outer, h->iters_outer, B * sizeof(int), hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (cost) HIPCHK(h, hipMemcpy(cost, h->cost, B * sizeof(double), hipMemcpyDeviceToHost));
    if (c_max) HIPCHK(h, hipMemcpy(c_max, h->cmax, B * sizeof(double), hipMemcpyDeviceToHost));
    if (cost_trace) HIPCHK(h, hipMemcpy(cost_trace, h->Jtrace, B * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    if (cmax_trace) HIPCHK(h, hipMemcpy(cmax_trace, h->ctrace, B * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_alpha_trace(altro_handle* h, double* alpha_trace) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_alpha_trace(alpha_trace));
    if (!h || !alpha_trace) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(alpha_trace, h->atrace, (size_t)h->d.batch * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_gains(altro_handle* h, double* K, double* d) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, get_gains(K, d));
    if (!h || (!K && !d)) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t n = h->d.n, m = h->d.m, N = h->d.N, B = h->d.batch, Bp = h->Bp;
    std::vector<double> kd(N * Bp * m * LW);
    HIPCHK(h, hipMemcpy(kd.data(), h->KD, kd.size() * sizeof(double), hipMemcpyDeviceToHost));
    // an iteration confirmed by the costate sweep ran no backward pass: K is the previous pass's (the same
    // matrix: the active set was verified unchanged), its feedforward terms are zero (include/altro_batch.h, strict)
    std::vector<int> dz(Bp);
    HIPCHK(h, hipMemcpy(dz.data(), h->dzero, Bp * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> df((N + 1) * Bp * LW);
    HIPCHK(h, hipMemcpy(df.data(), h->Dff, df.size() * sizeof(double), hipMemcpyDeviceToHost));
    // device layout KD [instance][k][control a][lane]: state lane j holds K[a][j] (the control lanes carry the factors of
    // Quu); Dff [instance][k (N + 1 rows)][lane]: control lane n+a holds d[a]
    for (size_t b = 0; b < B; ++b)
      for (size_t k = 0; k + 1 < N; ++k)
        for (size_t a = 0; a < m; ++a) {
          const double* row = kd.data() + ((b * N + k) * m + a) * LW;
          if (K) for (size_t j = 0; j < n; ++j) K[((b * (N - 1) + k) * n + j) * m + a] = row[j];
          if (d) d[(b * (N - 1) + k) * m + a] = dz[b] ? 0.0 : df[(b * (N + 1) + k) * LW + n + a];
        }
    return ALTRO_OK;
  });
}

int32_t altro_batch_last_solve_ms(altro_handle* h, float* ms) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, last_solve_ms(ms));
    if (!h || !ms) return ALTRO_ERR_INVALID_ARG;
    if (!h->timed) FAIL(h, ALTRO_ERR_STATE, "no solve has been launched");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return ALTRO_OK;
  });
}

int32_t altro_batch_timing_reset(altro_handle* h) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, timing_reset());
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->ring.reset();
    HIPCHK(h, hipMemsetAsync(h->n_backward, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_rollout, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_solves, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_iters, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_ok, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_trials, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_gconf, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->n_fo, 0, h->Bp * sizeof(long long), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

int32_t altro_batch_timing_get(altro_handle* h, float* ms, int32_t capacity, int32_t* count) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, timing_get(ms, capacity, count));
    if (!h || !count) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int32_t n = (int32_t)h->ring.readable();
    *count = n;
    for (int32_t i = 0; ms && i < n && i < capacity; ++i) HIPCHK(h, h->ring.elapsed((size_t)i, &ms[i]));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_solve_counters(altro_handle* h, int64_t* solves, int64_t* iterations, int64_t* succeeded) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide) { long long* const src[3] = {h->wide->n_solves, h->wide->n_iters, h->wide->n_ok}; const int rc_ = h->wide->counters(src, solves, iterations, succeeded); if (rc_) h->err = h->wide->err; return rc_; }
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t B = h->d.batch;
    if (solves) HIPCHK(h, hipMemcpy(solves, h->n_solves, B * sizeof(long long), hipMemcpyDeviceToHost));
    if (iterations) HIPCHK(h, hipMemcpy(iterations, h->n_iters, B * sizeof(long long), hipMemcpyDeviceToHost));
    if (succeeded) HIPCHK(h, hipMemcpy(succeeded, h->n_ok, B * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_work_counters(altro_handle* h, int64_t* backward_passes, int64_t* rollouts, int64_t* trials) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide) { long long* const src[3] = {h->wide->n_backward, h->wide->n_rollout, h->wide->n_trials}; const int rc_ = h->wide->counters(src, backward_passes, rollouts, trials); if (rc_) h->err = h->wide->err; return rc_; }
    if (!h) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t B = h->d.batch;
    if (backward_passes) HIPCHK(h, hipMemcpy(backward_passes, h->n_backward, B * sizeof(long long), hipMemcpyDeviceToHost));
    if (rollouts) HIPCHK(h, hipMemcpy(rollouts, h->n_rollout, B * sizeof(long long), hipMemcpyDeviceToHost));
    if (trials) HIPCHK(h, hipMemcpy(trials, h->n_trials, B * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_reuse_counter(altro_handle* h, int64_t* reused) {
  return guard(h, [&]() -> int32_t {
    if (!h || !reused) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->wide) {
      HIPCHK(h, hipStreamSynchronize(h->wide->stream));
      HIPCHK(h, hipMemcpy(reused, h->wide->n_gs, (size_t)h->d.batch * sizeof(long long), hipMemcpyDeviceToHost));
      return ALTRO_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(reused, h->n_fo, (size_t)h->d.batch * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_polish_stats(altro_handle* h, int32_t* ran, int32_t* failed, double* residual) {
  return guard(h, [&]() -> int32_t {
    if (!h) return ALTRO_ERR_INVALID_ARG;
    const size_t B = h->d.batch;
    if (h->wide) { const int rc_ = h->wide->polish_stats(ran, failed, residual); if (rc_) h->err = h->wide->err; return rc_; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->o.projected_newton) {
      HIPCHK(h, hipMemsetAsync(h->pn_ran, 0, h->Bp * sizeof(int), h->stream));
      HIPCHK(h, hipMemsetAsync(h->pn_failed, 0, h->Bp * sizeof(int), h->stream));
      HIPCHK(h, hipMemsetAsync(h->pn_res, 0, h->Bp * sizeof(double), h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (ran) HIPCHK(h, hipMemcpy(ran, h->pn_ran, B * sizeof(int), hipMemcpyDeviceToHost));
    if (failed) HIPCHK(h, hipMemcpy(failed, h->pn_failed, B * sizeof(int), hipMemcpyDeviceToHost));
    if (residual) HIPCHK(h, hipMemcpy(residual, h->pn_res, B * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_polish_dual_residuals(altro_handle* h, double* before, double* after, int32_t* failed) {
  return guard(h, [&]() -> int32_t {
    if (!h) return ALTRO_ERR_INVALID_ARG;
    const size_t B = h->d.batch;
    if (h->wide) { const int rc_ = h->wide->polish_dual(before, after, failed); if (rc_) h->err = h->wide->err; return rc_; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->o.projected_newton) {
      if (before) std::memset(before, 0, B * sizeof(double));
      if (after) std::memset(after, 0, B * sizeof(double));
      if (failed) std::memset(failed, 0, B * sizeof(int32_t));
      return ALTRO_OK;
    }
    if (before) HIPCHK(h, hipMemcpy(before, h->pn_dres0, B * sizeof(double), hipMemcpyDeviceToHost));
    if (after) HIPCHK(h, hipMemcpy(after, h->pn_dres, B * sizeof(double), hipMemcpyDeviceToHost));
    if (failed) HIPCHK(h, hipMemcpy(failed, h->pn_dfail, B * sizeof(int), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_confirm_counter(altro_handle* h, int64_t* confirmed) {
  return guard(h, [&]() -> int32_t {
    if (!h || !confirmed) return ALTRO_ERR_INVALID_ARG;
    if (h->wide) {
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipStreamSynchronize(h->wide->stream));
      HIPCHK(h, hipMemcpy(confirmed, h->wide->n_gconf, (size_t)h->d.batch * sizeof(long long), hipMemcpyDeviceToHost));
      return ALTRO_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(confirmed, h->n_gconf, (size_t)h->d.batch * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_wave_cycles(altro_handle* h, int64_t* cycles, int32_t capacity, int32_t* count) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide) { if (count) *count = 0; return ALTRO_OK; }
    if (!h || !count) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // (diagnostic builds append a 4096-word per-turn trace of one wave, ALTRO_DEBUG_TRACE_WAVE: returned to callers that
    //  offer the room)
    const int32_t n = h->Bp / IPW * 16 + ((capacity >= h->Bp / IPW * 16 + 4096) ? 4096 : 0);
    *count = n;
    if (cycles) HIPCHK(h, hipMemcpy(cycles, h->wave_cycles, (size_t)(n < capacity ? n : capacity) * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  });
}

int32_t altro_mpc_set_track(altro_handle* h, const double* Xtrack, const double* Utrack, int32_t Nt) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, mpc_set_track(Xtrack, Utrack, Nt));
    if (!h || !Xtrack || !Utrack) return ALTRO_ERR_INVALID_ARG;
    if (Nt < h->d.N) FAIL(h, ALTRO_ERR_INVALID_ARG, "track shorter than the horizon");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = set_ref_common(h, Xtrack, Utrack, Nt);
    if (rc) return rc;
    // initial_trajectory!(prob, Z): the first window of the track (mpc.jl:19-20,45)
    HIPCHK(h, hipMemsetAsync(h->cur, 0, h->Bp * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_window_copy, grid_for((size_t)h->Bp * LW), dim3(256), 0, h->stream, h->Z, h->Zref, h->Bp, h->d.N, Nt);
    HIPCHK(h, hipGetLastError());
    // x0 <- the track's first state, through the pack path (its lanes >= n must be zero)
    {
      const size_t cnt = (size_t)h->d.batch * h->d.n;
      std::vector<double> x0(cnt);
      const size_t n = h->d.n;
      for (size_t b = 0; b < (size_t)h->d.batch; ++b)
        for (size_t j = 0; j < n; ++j) x0[b * n + j] = Xtrack[(b * Nt + 0) * n + j];
      HIPCHK(h, hipStreamSynchronize(h->stream));
      rc = altro_batch_set_initial_state(h, x0.data());
      if (rc) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

int32_t altro_mpc_set_noise(altro_handle* h, const double* noise, int32_t steps) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, mpc_set_noise(noise, steps));
    if (!h || !noise || steps < 1) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->noise) HIPCHK(h, hipFree(h->noise));
    h->noise = nullptr;
    const size_t cnt = (size_t)steps * h->d.batch * h->d.n;
    HIPCHK(h, hipMalloc(&h->noise, cnt * sizeof(double)));
    HIPCHK(h, hipMemcpy(h->noise, noise, cnt * sizeof(double), hipMemcpyHostToDevice));
    h->noise_steps = steps;
    return ALTRO_OK;
  });
}

int32_t altro_mpc_set_noise_model(altro_handle* h, int32_t mode, const double* weights, const int32_t* groups) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, mpc_set_noise_model(mode, weights, groups));
    if (!h || !weights || mode < 0 || mode > 2) return ALTRO_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<double> w(LW, 0.0);
    std::vector<int> g(LW, 0);
    for (int i = 0; i < h->d.n; ++i) {
      w[i] = weights[i];
      g[i] = groups ? groups[i] : 0;
      if (g[i] != 0 && g[i] != 1) FAIL(h, ALTRO_ERR_INVALID_ARG, "noise groups are 0 or 1");
    }
    HIPCHK(h, hipMemcpyAsync(h->noise_w, w.data(), LW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->noise_grp, g.data(), LW * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->noise_mode = mode;
    return ALTRO_OK;
  });
}

int32_t altro_mpc_set_shift(altro_handle* h, int32_t shift) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide) { h->wide->mpc_shift = shift ? 1 : 0; return ALTRO_OK; }
    if (!h) return ALTRO_ERR_INVALID_ARG;
    h->mpc_shift = shift ? 1 : 0;
    return ALTRO_OK;
  });
}

int32_t altro_mpc_run_async(altro_handle* h, int32_t first_step, int32_t nsteps) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, mpc_run(first_step, nsteps));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    if (nsteps < 1 || first_step < 0) FAIL(h, ALTRO_ERR_INVALID_ARG, "bad step range");
    if (h->noise && first_step + nsteps > h->noise_steps) FAIL(h, ALTRO_ERR_INVALID_ARG, "steps outside the uploaded noise");
    if (first_step + nsteps + h->d.N > h->Nt) FAIL(h, ALTRO_ERR_INVALID_ARG, "steps run past the end of the track");
    return enqueue_solve(h, first_step, nsteps);
  });
}

int32_t altro_mpc_step_async(altro_handle* h, int32_t step) { return altro_mpc_run_async(h, step, 1); }

int32_t altro_mpc_set_dynamics_track(altro_handle* h, const double* A, const double* B, const double* f, int32_t nblocks,
                                     int32_t step_stride, int32_t per_instance) {
  return guard(h, [&]() -> int32_t {
    if (!h || !A || !B) return ALTRO_ERR_INVALID_ARG;
    if (!h->wide) {
      const int rc = migrate_to_wide(h);
      if (rc) return rc;
    }
    WIDE_FWD(h, mpc_set_dynamics_track(A, B, f, nblocks, step_stride, per_instance));
    return ALTRO_ERR_STATE;
  });
}

int32_t altro_mpc_prepare_async(altro_handle* h, int32_t step) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, mpc_prepare(step));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    if (step < 0) FAIL(h, ALTRO_ERR_INVALID_ARG, "bad step");
    if (h->noise && step + 1 > h->noise_steps) FAIL(h, ALTRO_ERR_INVALID_ARG, "step outside the uploaded noise");
    if (step + 1 + h->d.N > h->Nt) FAIL(h, ALTRO_ERR_INVALID_ARG, "step runs past the end of the track");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready(h);
    if (rc) return rc;
    if ((rc = launch_solve(h, step, 1, 1))) return rc;  // the solve kernel's own plant step, nothing else
    h->kref = step + 1;
    return ALTRO_OK;
  });
}

int32_t altro_batch_benchmark_solve(altro_handle* h, int32_t samples, int32_t evals, float* sample_ms) {
  return guard(h, [&]() -> int32_t {
    WIDE_FWD(h, benchmark_solve(samples, evals, sample_ms));
    if (!h) return ALTRO_ERR_INVALID_ARG;
    if (samples < 1 || evals < 1) FAIL(h, ALTRO_ERR_INVALID_ARG, "samples and evals must be positive");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t plane = (size_t)h->d.N * LW;
    if (!h->Zsave) HIPCHK(h, hipMalloc(&h->Zsave, plane * h->Bp * sizeof(double)));
    const dim3 grid = grid_for((size_t)h->Bp * LW);
    // Z0 = deepcopy(get_trajectory(solver))
    hipLaunchKernelGGL(k_plane_copy, grid, dim3(256), 0, h->stream, h->Z, h->Zsave, h->cur, plane, h->Bp, h->d.N, 1);
    HIPCHK(h, hipGetLastError());
    auto one = [&]() -> int {  // initial_trajectory!(solver, Z0); solve!(solver)
      hipLaunchKernelGGL(k_plane_copy, grid, dim3(256), 0, h->stream, h->Z, h->Zsave, h->cur, plane, h->Bp, h->d.N, 0);
      // every evaluation recomputes its gains, as an evaluation of the reference's `@benchmark solve!` does (the gains of
      // the previous evaluation would otherwise serve the next one: same start, same active sets)
      if (int rcd = drop_gains(h)) return rcd;
      return enqueue_solve(h, 0, 0);
    };
    int rc = one();  // BenchmarkTools' warm-up evaluation
    if (rc) return rc;
    for (int32_t s = 0; s < samples; ++s) {
      HIPCHK(h, hipEventRecord(h->bev0, h->stream));
      for (int32_t e = 0; e < evals; ++e)
        if ((rc = one())) return rc;
      HIPCHK(h, hipEventRecord(h->bev1, h->stream));
      HIPCHK(h, hipEventSynchronize(h->bev1));
      float ms = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms, h->bev0, h->bev1));
      if (sample_ms) sample_ms[s] = ms / (float)evals;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ALTRO_OK;
  });
}

int32_t altro_batch_get_stream(altro_handle* h, void** stream) {
  return guard(h, [&]() -> int32_t {
    if (h && h->wide && stream) { *stream = (void*)h->wide->stream; return ALTRO_OK; }
    if (!h || !stream) return ALTRO_ERR_INVALID_ARG;
    *stream = (void*)h->stream;
    return ALTRO_OK;
  });
}

}  // extern "C"
