// wide_backend.h -- host side of the one-wave-per-instance kernel (solve_wide.h) behind the same
// C-ABI: altro_batch.hip forwards every entry point here when (n, m) is outside the 16-lane
// kernel set.  Device arrays use the ABI's own layouts, so transfers are plain copies.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/altro_batch.h"
#include "launch_ring.h"
#include "solve_wide.h"
#include "pn_wide.h"

namespace altro_wide {

#define WCHK(call)                                                    \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      err = std::string(#call) + ": " + hipGetErrorString(e_);        \
      return ALTRO_ERR_HIP;                                           \
    }                                                                 \
  } while (0)
#define WFAIL(code, msg) \
  do {                   \
    err = (msg);         \
    return (code);       \
  } while (0)

// dst[b][len] <- src[b][cur[b]][len]
__global__ void k_gather_plane(double* dst, const double* src, const int* cur, size_t len, int B) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * len) return;
  const size_t b = t / len, e = t - b * len;
  dst[t] = src[(b * 2 + cur[b]) * len + e];
}
// src[b][cur[b]][len] <- dst-layout host image
__global__ void k_scatter_plane(double* planes, const double* img, const int* cur, size_t len, int B) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * len) return;
  const size_t b = t / len, e = t - b * len;
  planes[(b * 2 + cur[b]) * len + e] = img[t];
}

struct WideBackend {
  altro_dims d{};
  altro_opts o{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  altro::LaunchRing ring;
  std::string err;
  // device
  double *A = nullptr, *Bm = nullptr, *f = nullptr, *wd = nullptr, *wf = nullptr, *zmin = nullptr, *zmax = nullptr;
  double *x0 = nullptr, *Xref = nullptr, *Uref = nullptr, *X = nullptr, *U = nullptr, *Lb = nullptr, *Lc = nullptr,
         *mu = nullptr, *Kg = nullptr, *dg = nullptr, *trash = nullptr, *AconT = nullptr, *bcon = nullptr, *stage = nullptr, *Qz = nullptr, *fac = nullptr;
  unsigned char* aset = nullptr;   // [B][3][N][64] exact active sets (solve_wide.h: Params::aset)
  unsigned* bwst = nullptr;   // [B][136] per instance: the state of the gain reuse between launches (solve_wide.h: bw_*)
  int coop_mode = -1, static_mask = 7;  // altro_debug_set "wide_coop", "wide_static_mask" (before create)
  int compact_np_max = 48;  // wide_compact: the LDS carve-up with Qux and K inside W, for padded state dimensions up to this
  bool debug_keep_gains = false;  // altro_debug_set "keep_gains" (-DALTRO_DEBUG builds only): stale gains are kept (exists to show that the tests notice them)
  // projected-Newton polish (pn_wide.h): per-instance results, and the workspace slots allocated by the first solve that asks
  int *pn_ran = nullptr, *pn_failed = nullptr, *pn_dfail = nullptr;
  double *pn_res = nullptr, *pn_dres0 = nullptr, *pn_dres = nullptr;
  double *pnE = nullptr, *pndv = nullptr, *pnLd = nullptr, *pnLo = nullptr, *pnvec = nullptr, *pntz = nullptr, *pnblk = nullptr;
  int *pnnb = nullptr, *pnnst = nullptr, *pnrinfo = nullptr;
  int pn_bm = 0, pn_slots = 0;
  bool gains_valid = false;   // nothing the stored gains depend on (model, cost, constraints, options) has changed since the last launch
  double *Xsave = nullptr, *Usave = nullptr;  // Z0 of benchmark_solve
  std::vector<hipEvent_t> bench_ev;
  int *cur = nullptr, *ctype = nullptr, *rowk0 = nullptr, *rowk1 = nullptr, *rowc0 = nullptr, *rowcp = nullptr, *iters = nullptr, *iters_outer = nullptr,
      *status = nullptr, *noise_grp = nullptr;
  double *cost = nullptr, *cmax = nullptr, *Jtrace = nullptr, *ctrace = nullptr, *atrace = nullptr, *noise = nullptr,
         *noise_w = nullptr;
  long long *n_backward = nullptr, *n_rollout = nullptr, *n_trials = nullptr, *n_solves = nullptr, *n_iters = nullptr,
            *n_ok = nullptr, *n_gconf = nullptr, *n_gs = nullptr;
  size_t stage_bytes = 0;
  int Nt = 0, kref = 0, noise_steps = 0, noise_mode = 0, mpc_shift = 1;
  int dyn_blocks = 1, dyn_step_stride = 0;
  bool ltv = false, dyn_per_instance = false, have_dyn = false, have_cost = false, have_ref = false;
  int box_k0 = 0, box_k1 = -1, box_id = -1;
  struct Block {
    int id, sense, k0, k1, p, per_knot, r0;
    bool soc = false;
    bool per_instance = false;
    std::vector<double> A, b;  // row-major p x nz blocks: [instance if per_instance][knot of the range if per_knot]
  };
  std::vector<Block> blocks;
  int Pn = 0, ncon = 0, ncone = 0;
  bool con_dirty = false, con_locked = false, con_per_instance = false;
  size_t acon_elems = (size_t)-1;

  int np() const { return pad16(d.n); }
  int mp() const { return pad16(d.m); }
  int nz() const { return d.n + d.m; }

  static bool supports(int n, int m) { return n >= 1 && m >= 1 && n <= kMaxN && m <= kMaxM; }

  int ensure_stage(size_t bytes) {
    if (bytes <= stage_bytes) return ALTRO_OK;
    if (stage) WCHK(hipFree(stage));
    stage = nullptr;
    stage_bytes = 0;
    WCHK(hipMalloc(&stage, bytes));
    stage_bytes = bytes;
    return ALTRO_OK;
  }

  template <typename Tp>
  int dalloc(Tp** p, size_t count, bool zero = true) {
    WCHK(hipMalloc(p, (count ? count : 1) * sizeof(Tp)));
    if (zero) WCHK(hipMemsetAsync(*p, 0, (count ? count : 1) * sizeof(Tp), stream));
    return ALTRO_OK;
  }

  int create(const altro_dims* dims, const altro_opts* opts, int dev) {
    d = *dims;
    o = *opts;
    device = dev;
    // (diagnostic switches -- compact_np_max: 0 = never, 32 / 48 = up to that padded n; coop_mode: cooperative blocks 0 = never,
    //  1 = every size with n or m > 16; static_mask: which uses of time-invariant constraint tables stay in LDS;
    //  debug_keep_gains -- are members set by the caller from altro_debug_set() before create(): nothing is read from the
    //  environment)
    WCHK(hipSetDevice(device));
    const Lds L = lds_layout(d.n, d.m, kMaxP);
    (void)L;
    WCHK(hipStreamCreate(&stream));
    WCHK(hipEventCreate(&ev0));
    WCHK(hipEventCreate(&ev1));
    ring.reset();
    bench_ev.reserve(2);
    const size_t B = d.batch, N = d.N, n = d.n, m = d.m, z = n + m;
    if (B * 3 * N * 64 >= (1ull << 32)) WFAIL(ALTRO_ERR_UNSUPPORTED, "batch x horizon too large for one handle (active-set planes are addressed with 32-bit offsets): split the batch");
    int rc;
#define DA_(p, c) if ((rc = dalloc(&p, (c)))) return rc
    DA_(wd, z); DA_(wf, n); DA_(zmin, z); DA_(zmax, z);
    DA_(x0, B * n); DA_(X, B * 2 * N * n); DA_(U, B * 2 * (N - 1) * m); DA_(cur, B);
    DA_(Lb, B * N * 2 * z); DA_(mu, B); DA_(Kg, B * (N - 1) * n * m); DA_(dg, B * (N - 1) * m); DA_(trash, B * 64); DA_(Qz, B * N * z); DA_(fac, m <= 16 ? B * N * wide_fac_size(m) : 1); DA_(bwst, B * 136); DA_(aset, B * 3 * N * 64);
    DA_(iters, B); DA_(iters_outer, B); DA_(status, B); DA_(cost, B); DA_(cmax, B);
    DA_(Jtrace, B * ALTRO_TRACE_LEN); DA_(ctrace, B * ALTRO_TRACE_LEN); DA_(atrace, B * ALTRO_TRACE_LEN);
    DA_(n_backward, B); DA_(n_rollout, B); DA_(n_trials, B); DA_(n_solves, B); DA_(n_iters, B); DA_(n_ok, B); DA_(n_gconf, B); DA_(n_gs, B);
    DA_(noise_w, kMaxN); DA_(noise_grp, kMaxN);
    DA_(Lc, 1); DA_(AconT, 1); DA_(bcon, 1); DA_(ctype, 1); DA_(rowk0, 1); DA_(rowk1, 1); DA_(rowc0, 1); DA_(rowcp, 1);
#undef DA_
    {
      std::vector<double> inf(z, INFINITY), ninf(z, -INFINITY), w(kMaxN, 0.01), m0(B, 1.0);
      WCHK(hipMemcpyAsync(zmax, inf.data(), z * sizeof(double), hipMemcpyHostToDevice, stream));
      WCHK(hipMemcpyAsync(zmin, ninf.data(), z * sizeof(double), hipMemcpyHostToDevice, stream));
      WCHK(hipMemcpyAsync(noise_w, w.data(), kMaxN * sizeof(double), hipMemcpyHostToDevice, stream));
      WCHK(hipMemcpyAsync(mu, m0.data(), B * sizeof(double), hipMemcpyHostToDevice, stream));
      WCHK(hipStreamSynchronize(stream));
    }
    return ALTRO_OK;
  }

  void destroy() {
    hipSetDevice(device);
    if (stream) hipStreamSynchronize(stream);
    void* ptrs[] = {A, Bm, f, wd, wf, zmin, zmax, x0, Xref, Uref, X, U, Lb, Lc, mu, Kg, dg, trash, AconT, bcon, stage, cur, ctype,
                    rowk0, rowk1, rowc0, rowcp, iters, iters_outer, status, noise_grp, cost, cmax, Jtrace, ctrace, atrace, noise, noise_w,
                    n_backward, n_rollout, n_trials, n_solves, n_iters, n_ok, Xsave, Usave, Qz, n_gconf, n_gs, fac, bwst, aset,
                    pn_ran, pn_failed, pn_dfail, pn_res, pn_dres0, pn_dres, pnE, pndv, pnLd, pnLo, pnvec, pntz, pnblk, pnnb, pnnst, pnrinfo};
    for (void* p : ptrs)
      if (p) hipFree(p);
    ring.destroy();
    for (hipEvent_t e : bench_ev) hipEventDestroy(e);
    bench_ev.clear();
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    if (stream) hipStreamDestroy(stream);
  }

  // blocks_per_instance knot blocks per instance (1: time-invariant)
  int upload_dynamics(const double* A_, const double* B_, const double* f_, size_t blocks_per_instance, int per_instance) {
    if (!A_ || !B_) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    const size_t n = d.n, m = d.m;
    const size_t blocks_ = (per_instance ? (size_t)d.batch : 1) * blocks_per_instance;
    for (double** p : {&A, &Bm, &f})
      if (*p) { WCHK(hipFree(*p)); *p = nullptr; }
    WCHK(hipMalloc(&A, blocks_ * n * n * sizeof(double)));
    WCHK(hipMalloc(&Bm, blocks_ * n * m * sizeof(double)));
    WCHK(hipMalloc(&f, blocks_ * n * sizeof(double)));
    WCHK(hipMemcpy(A, A_, blocks_ * n * n * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(Bm, B_, blocks_ * n * m * sizeof(double), hipMemcpyHostToDevice));
    if (f_) WCHK(hipMemcpy(f, f_, blocks_ * n * sizeof(double), hipMemcpyHostToDevice));
    else WCHK(hipMemset(f, 0, blocks_ * n * sizeof(double)));
    dyn_per_instance = per_instance != 0;
    have_dyn = true;
    return ALTRO_OK;
  }

  int set_dynamics(const double* A_, const double* B_, const double* f_, int per_knot, int per_instance) {
    gains_valid = false;
    const int rc = upload_dynamics(A_, B_, f_, per_knot ? (size_t)(d.N - 1) : 1, per_instance);
    if (rc) return rc;
    ltv = per_knot != 0;
    dyn_blocks = per_knot ? d.N - 1 : 1;
    dyn_step_stride = 0;
    return ALTRO_OK;
  }

  // altro_mpc_set_dynamics_track: see include/altro_batch.h
  int mpc_set_dynamics_track(const double* A_, const double* B_, const double* f_, int nblocks, int step_stride, int per_instance) {
    gains_valid = false;
    if (nblocks < d.N - 1 || (step_stride != 1 && step_stride != d.N - 1)) WFAIL(ALTRO_ERR_INVALID_ARG, "bad dynamics track shape");
    const int rc = upload_dynamics(A_, B_, f_, (size_t)nblocks, per_instance);
    if (rc) return rc;
    ltv = true;
    dyn_blocks = nblocks;
    dyn_step_stride = step_stride;
    return ALTRO_OK;
  }
  // last reference-window start the dynamics table covers
  bool dyn_covers(int kref_) const { return !ltv || (long long)kref_ * dyn_step_stride + (d.N - 1) <= (long long)dyn_blocks; }

  int set_tracking_cost(const double* Qd, const double* Rd, const double* Qfd, double dt) {
    gains_valid = false;
    if (!Qd || !Rd || !Qfd || !(dt > 0.0)) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    std::vector<double> w(nz());
    for (int i = 0; i < d.n; ++i) w[i] = dt * Qd[i];
    for (int i = 0; i < d.m; ++i) w[d.n + i] = dt * Rd[i];
    WCHK(hipMemcpy(wd, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(wf, Qfd, d.n * sizeof(double), hipMemcpyHostToDevice));
    have_cost = true;
    return ALTRO_OK;
  }

  int add_constraint(int kind, int sense, int k_first, int k_last, int p, const double* A_, const double* b_, const double* zmin_,
                     const double* zmax_, int per_knot, int* con_id) {
    WCHK(hipSetDevice(device));
    if (con_locked) WFAIL(ALTRO_ERR_STATE, "constraints must be added before the first solve");
    if (k_first < 0 || k_last < k_first || k_last > d.N - 1) WFAIL(ALTRO_ERR_INVALID_ARG, "bad knot range");
    if (kind == ALTRO_CON_BOX) {
      if (!zmin_ || !zmax_) return ALTRO_ERR_INVALID_ARG;
      if (box_id >= 0) WFAIL(ALTRO_ERR_UNSUPPORTED, "one BOX constraint per problem");
      WCHK(hipMemcpy(zmin, zmin_, nz() * sizeof(double), hipMemcpyHostToDevice));
      WCHK(hipMemcpy(zmax, zmax_, nz() * sizeof(double), hipMemcpyHostToDevice));
      box_k0 = k_first;
      box_k1 = k_last;
      box_id = ncon++;
      if (con_id) *con_id = box_id;
      return ALTRO_OK;
    }
    if ((kind != ALTRO_CON_LINEAR && kind != ALTRO_CON_SOC) || !A_ || !b_ || p < 1) return ALTRO_ERR_INVALID_ARG;
    if (kind == ALTRO_CON_SOC && (p < 2 || p > 4)) WFAIL(ALTRO_ERR_UNSUPPORTED, "second-order cones of dimension 2..4 only");
    if (kind == ALTRO_CON_LINEAR && sense != ALTRO_SENSE_EQ && sense != ALTRO_SENSE_INEQ) return ALTRO_ERR_INVALID_ARG;
    if (Pn + p > kMaxP) WFAIL(ALTRO_ERR_UNSUPPORTED, "more than 64 linear constraint rows");
    Block bl;
    bl.id = ncon++;
    bl.soc = kind == ALTRO_CON_SOC;
    bl.sense = sense; bl.k0 = k_first; bl.k1 = k_last; bl.p = p; bl.per_knot = (per_knot & 1) ? 1 : 0; bl.r0 = Pn;
    bl.per_instance = (per_knot & 2) != 0;
    const size_t nb = (bl.per_knot ? (size_t)(k_last - k_first + 1) : 1) * (bl.per_instance ? (size_t)d.batch : 1);
    bl.A.assign(A_, A_ + nb * p * nz());
    bl.b.assign(b_, b_ + nb * p);
    blocks.push_back(bl);
    Pn += p;
    con_dirty = true;
    if (con_id) *con_id = bl.id;
    return ALTRO_OK;
  }

  Block* find(int id) {
    for (auto& b : blocks)
      if (b.id == id) return &b;
    return nullptr;
  }

  int update_constraint_data(int con_id, const double* A_, const double* b_) {
    gains_valid = false;
    Block* bl = find(con_id);
    if (!bl) WFAIL(ALTRO_ERR_INVALID_ARG, "no such LINEAR constraint");
    const size_t nb = (bl->per_knot ? (size_t)(bl->k1 - bl->k0 + 1) : 1) * (bl->per_instance ? (size_t)d.batch : 1);
    if (A_) bl->A.assign(A_, A_ + nb * bl->p * nz());
    if (b_) bl->b.assign(b_, b_ + nb * bl->p);
    con_dirty = true;
    return ALTRO_OK;
  }

  // per-knot tables of the generic rows (transposed: AconT[k][j][r])
  int pack_constraints() {
    if (!con_dirty) return ALTRO_OK;
    WCHK(hipSetDevice(device));
    const size_t N = d.N, z = nz(), P = Pn;
    // one table per instance as soon as any block carries per-instance data (grasp_mpc_helpers.jl:46-55 mutates each
    // problem's own tables); otherwise one table shared by the batch
    con_per_instance = false;
    for (const auto& bl : blocks) con_per_instance = con_per_instance || bl.per_instance;
    const size_t ninst = con_per_instance ? (size_t)d.batch : 1;
    std::vector<double> At(ninst * N * z * P, 0.0), bc(ninst * N * P, 0.0);
    std::vector<int> ct(N * P, 0), k0(P, 0), k1(P, -1), c0(P, 0), cp(P, 0);
    ncone = 0;
    for (const auto& bl : blocks) ncone += bl.soc ? 1 : 0;
    for (const auto& bl : blocks)
      for (int r = 0; r < bl.p; ++r) {
        const int row = bl.r0 + r;
        k0[row] = bl.k0;
        k1[row] = bl.k1;
        c0[row] = bl.soc ? bl.r0 : 0;
        cp[row] = bl.soc ? bl.p : 0;
        const size_t nk = bl.per_knot ? (size_t)(bl.k1 - bl.k0 + 1) : 1;
        for (int k = bl.k0; k <= bl.k1; ++k) {
          ct[k * P + row] = bl.soc ? 3 : (bl.sense == ALTRO_SENSE_EQ ? 1 : 2);
          for (size_t ib = 0; ib < ninst; ++ib) {
            const size_t blk = (bl.per_instance ? ib * nk : 0) + (bl.per_knot ? (size_t)(k - bl.k0) : 0);
            bc[(ib * N + k) * P + row] = bl.b[blk * bl.p + r];
            for (size_t j = 0; j < z; ++j) At[((ib * N + k) * z + j) * P + row] = bl.A[(blk * bl.p + r) * z + j];
          }
        }
      }
    if (At.size() != acon_elems) {  // the table changed shape (a block with per-instance data arrived): reallocate
      for (double** p : {&AconT, &bcon})
        if (*p) { WCHK(hipFree(*p)); *p = nullptr; }
      WCHK(hipMalloc(&AconT, (At.size() ? At.size() : 1) * sizeof(double)));
      WCHK(hipMalloc(&bcon, (bc.size() ? bc.size() : 1) * sizeof(double)));
      acon_elems = At.size();
    }
    if (!con_locked) {
      for (void* p : {(void*)ctype, (void*)rowk0, (void*)rowk1, (void*)rowc0, (void*)rowcp, (void*)Lc})
        if (p) WCHK(hipFree(p));
      Lc = nullptr;
      ctype = rowk0 = rowk1 = rowc0 = rowcp = nullptr;
      WCHK(hipMalloc(&ctype, ct.size() * sizeof(int)));
      WCHK(hipMalloc(&rowk0, P * sizeof(int)));
      WCHK(hipMalloc(&rowk1, P * sizeof(int)));
      WCHK(hipMalloc(&rowc0, P * sizeof(int)));
      WCHK(hipMalloc(&rowcp, P * sizeof(int)));
      WCHK(hipMalloc(&Lc, (size_t)d.batch * N * P * sizeof(double)));
      WCHK(hipMemset(Lc, 0, (size_t)d.batch * N * P * sizeof(double)));
    }
    WCHK(hipMemcpy(AconT, At.data(), At.size() * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(bcon, bc.data(), bc.size() * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(ctype, ct.data(), ct.size() * sizeof(int), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(rowk0, k0.data(), P * sizeof(int), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(rowk1, k1.data(), P * sizeof(int), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(rowc0, c0.data(), P * sizeof(int), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(rowcp, cp.data(), P * sizeof(int), hipMemcpyHostToDevice));
    con_dirty = false;
    return ALTRO_OK;
  }

  int set_initial_state(const double* x) {
    if (!x) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    WCHK(hipMemcpyAsync(x0, x, (size_t)d.batch * d.n * sizeof(double), hipMemcpyHostToDevice, stream));
    WCHK(hipStreamSynchronize(stream));
    return ALTRO_OK;
  }
  int get_initial_state(double* x) {
    if (!x) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    WCHK(hipMemcpy(x, x0, (size_t)d.batch * d.n * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }

  int set_ref_common(const double* Xr, const double* Ur, int Nt_) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    if (Nt_ != Nt || !Xref) {
      if (Xref) WCHK(hipFree(Xref));
      if (Uref) WCHK(hipFree(Uref));
      Xref = Uref = nullptr;
      WCHK(hipMalloc(&Xref, (size_t)d.batch * Nt_ * d.n * sizeof(double)));
      WCHK(hipMalloc(&Uref, (size_t)d.batch * (Nt_ - 1) * d.m * sizeof(double)));
      Nt = Nt_;
    }
    WCHK(hipMemcpy(Xref, Xr, (size_t)d.batch * Nt * d.n * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(Uref, Ur, (size_t)d.batch * (Nt - 1) * d.m * sizeof(double), hipMemcpyHostToDevice));
    kref = 0;
    have_ref = true;
    return ALTRO_OK;
  }
  int set_reference(const double* Xr, const double* Ur) {
    if (!Xr || !Ur) return ALTRO_ERR_INVALID_ARG;
    return set_ref_common(Xr, Ur, d.N);
  }

  // plane cur[b] of X / U <- host image (instance-major)
  int put_planes(const double* Xh, const double* Uh) {
    const size_t B = d.batch, lx = (size_t)d.N * d.n, lu = (size_t)(d.N - 1) * d.m;
    int rc = ensure_stage(B * (lx + lu) * sizeof(double));
    if (rc) return rc;
    if (Xh) {
      WCHK(hipMemcpyAsync(stage, Xh, B * lx * sizeof(double), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_scatter_plane, dim3((unsigned)((B * lx + 255) / 256)), dim3(256), 0, stream, X, stage, cur, lx, (int)B);
    }
    if (Uh) {
      double* su = stage + B * lx;
      WCHK(hipMemcpyAsync(su, Uh, B * lu * sizeof(double), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_scatter_plane, dim3((unsigned)((B * lu + 255) / 256)), dim3(256), 0, stream, U, su, cur, lu, (int)B);
    }
    WCHK(hipStreamSynchronize(stream));
    return ALTRO_OK;
  }
  int set_initial_trajectory(const double* Xh, const double* Uh) {
    if (!Uh) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    return put_planes(Xh, Uh);
  }
  int get_planes(double* Xh, double* Uh) {
    WCHK(hipSetDevice(device));
    const size_t B = d.batch, lx = (size_t)d.N * d.n, lu = (size_t)(d.N - 1) * d.m;
    int rc = ensure_stage(B * (lx + lu) * sizeof(double));
    if (rc) return rc;
    if (Xh) {
      hipLaunchKernelGGL(k_gather_plane, dim3((unsigned)((B * lx + 255) / 256)), dim3(256), 0, stream, stage, X, cur, lx, (int)B);
      WCHK(hipMemcpyAsync(Xh, stage, B * lx * sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    if (Uh) {
      double* su = stage + B * lx;
      hipLaunchKernelGGL(k_gather_plane, dim3((unsigned)((B * lu + 255) / 256)), dim3(256), 0, stream, su, U, cur, lu, (int)B);
      WCHK(hipMemcpyAsync(Uh, su, B * lu * sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    WCHK(hipStreamSynchronize(stream));
    return ALTRO_OK;
  }

  Params params() const {
    Params p{};
    p.B = d.batch; p.n = d.n; p.m = d.m; p.N = d.N; p.Nt = Nt; p.np = np(); p.mp = mp(); p.Pn = Pn; p.Pp = pad4(Pn);
    p.ltv = ltv; p.dyn_per_instance = dyn_per_instance;
    p.A = A; p.Bm = Bm; p.f = f; p.wd = wd; p.wf = wf; p.zmin = zmin; p.zmax = zmax;
    p.box_k0 = box_k0; p.box_k1 = box_k1;
    p.AconT = AconT; p.bcon = bcon;
    p.con_istride = con_per_instance ? (size_t)d.N * nz() * Pn : 0;
    p.bcon_istride = con_per_instance ? (size_t)d.N * Pn : 0; p.ctype = ctype; p.rowk0 = rowk0; p.rowk1 = rowk1; p.rowc0 = rowc0; p.rowcp = rowcp; p.ncone = ncone;
    p.con_static = 7;
    for (const auto& bl : blocks) p.con_static = bl.per_knot ? 0 : p.con_static;
    p.con_static &= static_mask;
    p.x0 = x0; p.Xref = Xref; p.Uref = Uref; p.X = X; p.U = U; p.cur = cur; p.Lb = Lb; p.Lc = Lc; p.mu = mu; p.Kg = Kg; p.dg = dg; p.trash = trash;
    p.iters = iters; p.iters_outer = iters_outer; p.status = status; p.cost = cost; p.cmax = cmax;
    p.Jtrace = Jtrace; p.ctrace = ctrace; p.atrace = atrace;
    p.n_backward = n_backward; p.n_rollout = n_rollout; p.n_trials = n_trials; p.n_solves = n_solves; p.n_iters = n_iters; p.n_ok = n_ok; p.n_gconf = n_gconf; p.n_gs = n_gs; p.Qz = Qz; p.fac = fac; p.bwst = bwst; p.aset = aset; p.reuse_ok = (gains_valid || debug_keep_gains) ? 1 : 0;
    p.noise = noise; p.noise_w = noise_w; p.noise_grp = noise_grp; p.noise_mode = noise_mode; p.mpc_shift = mpc_shift;
    p.kref = kref;
    p.dyn_blocks = dyn_blocks; p.dyn_step_stride = dyn_step_stride;
    p.compact = compact();
    p.o = o;
    if (o.projected_newton) {  // solve!(::ALTROSolver): the AL stage only has to reach the polish's tolerance
      if (o.projected_newton_tolerance >= 0) p.o.constraint_tolerance = o.projected_newton_tolerance;
      else { p.o.constraint_tolerance = 0.0; p.o.kickout_max_penalty = 1; }
    }
    return p;
  }

  // solve!(::ProjectedNewtonSolver) after the AL kernel of a plain solve: every check and allocation BEFORE the launch takes
  // its slot of the timing ring
  int polish_prepare() {
    const size_t B = d.batch, N = d.N, z = nz();
    if (!pn_ran) {
      int rc;
      if ((rc = dalloc(&pn_ran, B)) || (rc = dalloc(&pn_failed, B)) || (rc = dalloc(&pn_dfail, B)) || (rc = dalloc(&pn_res, B)) ||
          (rc = dalloc(&pn_dres0, B)) || (rc = dalloc(&pn_dres, B))) return rc;
    }
    std::vector<double> lo(z), hi(z);
    WCHK(hipMemcpy(lo.data(), zmin, z * sizeof(double), hipMemcpyDeviceToHost));
    WCHK(hipMemcpy(hi.data(), zmax, z * sizeof(double), hipMemcpyDeviceToHost));
    int sides = 0;
    if (box_k1 >= box_k0)
      for (size_t j = 0; j < z; ++j) sides += (lo[j] > -1e300 ? 1 : 0) + (hi[j] < 1e300 ? 1 : 0);
    const int bm = 2 * d.n + sides + Pn;
    const int slots = (int)(B < 64 ? B : 64);
    if (bm != pn_bm || slots != pn_slots || !pnE) {
      void** ws[] = {(void**)&pnE, (void**)&pndv, (void**)&pnLd, (void**)&pnLo, (void**)&pnvec, (void**)&pntz, (void**)&pnblk,
                     (void**)&pnnb, (void**)&pnnst, (void**)&pnrinfo};
      for (void** q : ws) if (*q) { WCHK(hipFree(*q)); *q = nullptr; }
      const size_t S = slots, b = bm;
      WCHK(hipMalloc(&pnE, S * N * b * z * sizeof(double)));
      WCHK(hipMalloc(&pndv, S * N * b * sizeof(double)));
      WCHK(hipMalloc(&pnLd, S * N * b * b * sizeof(double)));
      WCHK(hipMalloc(&pnLo, S * N * b * b * sizeof(double)));
      WCHK(hipMalloc(&pnvec, S * 6 * N * b * sizeof(double)));
      WCHK(hipMalloc(&pntz, S * 3 * N * z * sizeof(double)));
      WCHK(hipMalloc(&pnblk, S * (2 * b * (b + 1) + 4 * b) * sizeof(double)));
      WCHK(hipMalloc(&pnnb, S * N * sizeof(int)));
      WCHK(hipMalloc(&pnnst, S * N * sizeof(int)));
      WCHK(hipMalloc(&pnrinfo, S * N * b * sizeof(int)));
      pn_bm = bm;
      pn_slots = slots;
    }
    return ALTRO_OK;
  }
  int polish_launch() {
    altro_pnw::WParams w{};
    w.P = params();
    w.P.o = o;   // the caller's tolerances (params() carries the AL stage's)
    w.bm = pn_bm; w.nslots = pn_slots;
    w.pn_ran = pn_ran; w.pn_failed = pn_failed; w.pn_dfail = pn_dfail; w.pn_res = pn_res; w.pn_dres0 = pn_dres0; w.pn_dres = pn_dres;
    w.E = pnE; w.dv = pndv; w.Ld = pnLd; w.Lo = pnLo; w.vec = pnvec; w.tz = pntz; w.blk = pnblk;
    w.nb = pnnb; w.nst = pnnst; w.rinfo = pnrinfo;
    hipLaunchKernelGGL(altro_pnw::pnw_kernel, dim3(pn_slots), dim3(64), 0, stream, w, o.constraint_tolerance);
    WCHK(hipGetLastError());
    return ALTRO_OK;
  }
  int polish_stats(int32_t* ran, int32_t* failed, double* residual) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    const size_t B = d.batch;
    const bool have = o.projected_newton && pn_ran;
    if (ran) { if (have) WCHK(hipMemcpy(ran, pn_ran, B * sizeof(int), hipMemcpyDeviceToHost)); else std::memset(ran, 0, B * sizeof(int32_t)); }
    if (failed) { if (have) WCHK(hipMemcpy(failed, pn_failed, B * sizeof(int), hipMemcpyDeviceToHost)); else std::memset(failed, 0, B * sizeof(int32_t)); }
    if (residual) { if (have) WCHK(hipMemcpy(residual, pn_res, B * sizeof(double), hipMemcpyDeviceToHost)); else std::memset(residual, 0, B * sizeof(double)); }
    return ALTRO_OK;
  }
  int polish_dual(double* before, double* after, int32_t* failed) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    const size_t B = d.batch;
    const bool have = o.projected_newton && pn_ran;
    if (before) { if (have) WCHK(hipMemcpy(before, pn_dres0, B * sizeof(double), hipMemcpyDeviceToHost)); else std::memset(before, 0, B * sizeof(double)); }
    if (after) { if (have) WCHK(hipMemcpy(after, pn_dres, B * sizeof(double), hipMemcpyDeviceToHost)); else std::memset(after, 0, B * sizeof(double)); }
    if (failed) { if (have) WCHK(hipMemcpy(failed, pn_dfail, B * sizeof(int), hipMemcpyDeviceToHost)); else std::memset(failed, 0, B * sizeof(int32_t)); }
    return ALTRO_OK;
  }

  int compact() const {  // only for one-wave blocks: the helper waves of a cooperative block read W while wave 0 writes Qux
    if (!wide_compact(d.n, d.m, ltv, compact_np_max)) return 0;
    return wide_block_threads(d.n, d.m, (size_t)lds_layout(d.n, d.m, Pn, 1).total * sizeof(double), coop_mode) == 64 ? 1 : 0;
  }
  size_t lds_bytes() const { return (size_t)lds_layout(d.n, d.m, Pn, compact()).total * sizeof(double); }

  int prepare_launch() {
    if (!have_dyn) WFAIL(ALTRO_ERR_STATE, "altro_batch_set_dynamics has not been called");
    if (!have_cost) WFAIL(ALTRO_ERR_STATE, "altro_batch_set_tracking_cost has not been called");
    if (!have_ref) WFAIL(ALTRO_ERR_STATE, "no reference trajectory (altro_batch_set_reference / altro_mpc_set_track)");
    int rc = pack_constraints();
    if (rc) return rc;
    con_locked = true;
    const size_t bytes = lds_bytes();
    if (bytes > 160 * 1024) WFAIL(ALTRO_ERR_UNSUPPORTED, "problem does not fit the 160 KB of LDS of one CU");
    WCHK(hipFuncSetAttribute((const void*)wide_kernel_for(d.n, d.m, Pn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    WCHK(hipFuncSetAttribute((const void*)wide_shift_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return ALTRO_OK;
  }

  int shift_fill(int primal, int dual) {
    WCHK(hipSetDevice(device));
    int rc = prepare_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wide_shift_kernel, dim3(d.batch), dim3(64), lds_bytes(), stream, params(), primal, dual);
    WCHK(hipGetLastError());
    return ALTRO_OK;
  }

  int enqueue(int mpc, int first_step, int nsteps) {
    WCHK(hipSetDevice(device));
    int rc = prepare_launch();
    if (rc) return rc;
    const int last_kref = mpc ? first_step + nsteps : kref;
    if (last_kref + d.N > Nt) WFAIL(ALTRO_ERR_STATE, "reference window runs past the end of the stored trajectory");
    if (mpc && ltv && dyn_step_stride == 0)
      WFAIL(ALTRO_ERR_UNSUPPORTED, "the device MPC loop over per-knot dynamics needs their table for every step: altro_mpc_set_dynamics_track");
    if (!dyn_covers(last_kref)) WFAIL(ALTRO_ERR_STATE, "the dynamics track ends before the last step's window");
    if (o.projected_newton && (rc = polish_prepare())) return rc;
    hipEvent_t h0, h1;
    WCHK(ring.next(&h0, &h1));
    WCHK(hipEventRecord(ev0, stream));
    WCHK(hipEventRecord(h0, stream));
    if (o.projected_newton && mpc) {
      // the steps of a fused launch as nsteps pairs of (one-step solve kernel, polish kernel): the next step's shift starts
      // from the polished trajectory and the projected multipliers, as after solve!(::ALTROSolver)
      const int kref0 = kref;
      for (int s = 0; s < nsteps && !rc; ++s) {
        hipLaunchKernelGGL(wide_kernel_for(d.n, d.m, Pn), dim3(d.batch), dim3(wide_block_threads(d.n, d.m, lds_bytes(), coop_mode)), lds_bytes(), stream, params(), mpc, first_step + s, 1);
        rc = hipGetLastError() == hipSuccess ? ALTRO_OK : ALTRO_ERR_HIP;
        if (rc) err = "launch of the solve kernel failed";
        kref = first_step + s + 1;
        if (!rc) rc = polish_launch();
      }
      if (rc) kref = kref0;
    } else {
      hipLaunchKernelGGL(wide_kernel_for(d.n, d.m, Pn), dim3(d.batch), dim3(wide_block_threads(d.n, d.m, lds_bytes(), coop_mode)), lds_bytes(), stream, params(), mpc, first_step, nsteps);
      rc = hipGetLastError() == hipSuccess ? ALTRO_OK : ALTRO_ERR_HIP;
      if (rc) err = "launch of the solve kernel failed";
      if (!rc && o.projected_newton) rc = polish_launch();
    }
    gains_valid = true;  // (until a setter changes something the stored gains depend on)
    WCHK(hipEventRecord(h1, stream));   // (every slot of the ring handed out has both events)
    if (rc) return rc;
    WCHK(hipEventRecord(ev1, stream));
    timed = true;
    if (mpc) kref = first_step + nsteps;
    return ALTRO_OK;
  }

  // altro_mpc_prepare_async: plant step + noise -> x0, reference window <- step + 1 (no shift, no solve)
  int mpc_prepare(int step) {
    if (step < 0) WFAIL(ALTRO_ERR_INVALID_ARG, "bad step");
    if (noise && step + 1 > noise_steps) WFAIL(ALTRO_ERR_INVALID_ARG, "step outside the uploaded noise");
    if (step + 1 + d.N > Nt) WFAIL(ALTRO_ERR_INVALID_ARG, "step runs past the end of the track");
    if (ltv && dyn_step_stride == 0) WFAIL(ALTRO_ERR_UNSUPPORTED, "the device plant step over per-knot dynamics needs altro_mpc_set_dynamics_track");
    if (!dyn_covers(step + 1)) WFAIL(ALTRO_ERR_STATE, "the dynamics track ends before this step's window");
    WCHK(hipSetDevice(device));
    int rc = prepare_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wide_kernel_for(d.n, d.m, Pn), dim3(d.batch), dim3(wide_block_threads(d.n, d.m, lds_bytes(), coop_mode)), lds_bytes(), stream, params(), 2, step, 1);
    WCHK(hipGetLastError());
    kref = step + 1;
    return ALTRO_OK;
  }

  // benchmark_solve!(solver; samples, evals): see include/altro_batch.h
  int benchmark_solve(int samples, int evals, float* sample_ms) {
    if (samples < 1 || evals < 1) WFAIL(ALTRO_ERR_INVALID_ARG, "samples and evals must be positive");
    WCHK(hipSetDevice(device));
    const size_t B = d.batch, lx = (size_t)d.N * d.n, lu = (size_t)(d.N - 1) * d.m;
    if (!Xsave) WCHK(hipMalloc(&Xsave, B * lx * sizeof(double)));
    if (!Usave) WCHK(hipMalloc(&Usave, B * lu * sizeof(double)));
    while (bench_ev.size() < 2) {
      hipEvent_t e;
      WCHK(hipEventCreate(&e));
      bench_ev.push_back(e);
    }
    const dim3 gx((unsigned)((B * lx + 255) / 256)), gu((unsigned)((B * lu + 255) / 256));
    hipLaunchKernelGGL(k_gather_plane, gx, dim3(256), 0, stream, Xsave, X, cur, lx, (int)B);  // Z0 = copy(get_trajectory(solver))
    hipLaunchKernelGGL(k_gather_plane, gu, dim3(256), 0, stream, Usave, U, cur, lu, (int)B);
    WCHK(hipGetLastError());
    auto one = [&]() -> int {  // initial_trajectory!(solver, Z0); solve!(solver)
      hipLaunchKernelGGL(k_scatter_plane, gx, dim3(256), 0, stream, X, Xsave, cur, lx, (int)B);
      hipLaunchKernelGGL(k_scatter_plane, gu, dim3(256), 0, stream, U, Usave, cur, lu, (int)B);
      gains_valid = false;   // every evaluation recomputes its gains, as the reference's `@benchmark solve!` does
      return enqueue(0, 0, 0);
    };
    int rc = one();  // BenchmarkTools' warm-up evaluation
    if (rc) return rc;
    for (int s_ = 0; s_ < samples; ++s_) {
      WCHK(hipEventRecord(bench_ev[0], stream));
      for (int e = 0; e < evals; ++e)
        if ((rc = one())) return rc;
      WCHK(hipEventRecord(bench_ev[1], stream));
      WCHK(hipEventSynchronize(bench_ev[1]));
      float ms = 0.f;
      WCHK(hipEventElapsedTime(&ms, bench_ev[0], bench_ev[1]));
      if (sample_ms) sample_ms[s_] = ms / (float)evals;
    }
    WCHK(hipStreamSynchronize(stream));
    return ALTRO_OK;
  }

  int synchronize() {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    return ALTRO_OK;
  }

  int duals(int con_id, double* lam, bool set) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    int rc = pack_constraints();
    if (rc) return rc;
    const size_t B = d.batch, N = d.N, z = nz();
    if (con_id == box_id && box_id >= 0) {
      const size_t nk = box_k1 - box_k0 + 1, w = nk * 2 * z * sizeof(double), pitch = N * 2 * z * sizeof(double);
      double* base = Lb + (size_t)box_k0 * 2 * z;
      if (set) WCHK(hipMemcpy2D(base, pitch, lam, w, w, B, hipMemcpyHostToDevice));
      else WCHK(hipMemcpy2D(lam, w, base, pitch, w, B, hipMemcpyDeviceToHost));
      return ALTRO_OK;
    }
    Block* bl = find(con_id);
    if (!bl) WFAIL(ALTRO_ERR_INVALID_ARG, "no such constraint");
    std::vector<double> all(B * N * Pn);
    WCHK(hipMemcpy(all.data(), Lc, all.size() * sizeof(double), hipMemcpyDeviceToHost));
    const size_t nk = bl->k1 - bl->k0 + 1;
    for (size_t b = 0; b < B; ++b)
      for (size_t kk = 0; kk < nk; ++kk)
        for (int r = 0; r < bl->p; ++r) {
          double& dv = all[(b * N + bl->k0 + kk) * Pn + bl->r0 + r];
          double& hv = lam[(b * nk + kk) * bl->p + r];
          if (set) dv = hv; else hv = dv;
        }
    if (set) WCHK(hipMemcpy(Lc, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
    return ALTRO_OK;
  }

  int get_stats(int32_t* it, int32_t* ito, int32_t* st, double* J, double* c, double* Jt, double* ct) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    const size_t B = d.batch;
    if (it) WCHK(hipMemcpy(it, iters, B * sizeof(int), hipMemcpyDeviceToHost));
    if (ito) WCHK(hipMemcpy(ito, iters_outer, B * sizeof(int), hipMemcpyDeviceToHost));
    if (st) WCHK(hipMemcpy(st, status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (J) WCHK(hipMemcpy(J, cost, B * sizeof(double), hipMemcpyDeviceToHost));
    if (c) WCHK(hipMemcpy(c, cmax, B * sizeof(double), hipMemcpyDeviceToHost));
    if (Jt) WCHK(hipMemcpy(Jt, Jtrace, B * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    if (ct) WCHK(hipMemcpy(ct, ctrace, B * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }
  int get_alpha_trace(double* a) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    WCHK(hipMemcpy(a, atrace, (size_t)d.batch * ALTRO_TRACE_LEN * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }
  int get_gains(double* K, double* dd) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    if (K) WCHK(hipMemcpy(K, Kg, (size_t)d.batch * (d.N - 1) * d.n * d.m * sizeof(double), hipMemcpyDeviceToHost));
    if (dd) WCHK(hipMemcpy(dd, dg, (size_t)d.batch * (d.N - 1) * d.m * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }
  int last_solve_ms(float* ms) {
    if (!timed) WFAIL(ALTRO_ERR_STATE, "no solve has been launched");
    WCHK(hipSetDevice(device));
    WCHK(hipEventSynchronize(ev1));
    WCHK(hipEventElapsedTime(ms, ev0, ev1));
    return ALTRO_OK;
  }
  int timing_reset() {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    ring.reset();
    const size_t B = d.batch;
    for (long long* p : {n_backward, n_rollout, n_trials, n_solves, n_iters, n_ok, n_gconf, n_gs}) WCHK(hipMemset(p, 0, B * sizeof(long long)));
    return ALTRO_OK;
  }
  int timing_get(float* ms, int capacity, int* count) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    const int nl = (int)ring.readable();
    if (count) *count = nl;
    for (int i = 0; i < nl && i < capacity && ms; ++i) WCHK(ring.elapsed((size_t)i, &ms[i]));
    return ALTRO_OK;
  }
  int counters(long long* const src[3], int64_t* a, int64_t* b, int64_t* c) {
    WCHK(hipSetDevice(device));
    WCHK(hipStreamSynchronize(stream));
    int64_t* dst[3] = {a, b, c};
    for (int i = 0; i < 3; ++i)
      if (dst[i]) WCHK(hipMemcpy(dst[i], src[i], (size_t)d.batch * sizeof(long long), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }

  int mpc_set_track(const double* Xt, const double* Ut, int Nt_) {
    if (!Xt || !Ut) return ALTRO_ERR_INVALID_ARG;
    if (Nt_ < d.N) WFAIL(ALTRO_ERR_INVALID_ARG, "track shorter than the horizon");
    int rc = set_ref_common(Xt, Ut, Nt_);
    if (rc) return rc;
    // initial_trajectory!(prob, Z): the first window of the track; x0 = its first knot (mpc.jl:19-20,45)
    const size_t B = d.batch, N = d.N, n = d.n, m = d.m;
    std::vector<double> Xw(B * N * n), Uw(B * (N - 1) * m), xs(B * n);
    for (size_t b = 0; b < B; ++b) {
      std::copy(Xt + b * Nt_ * n, Xt + b * Nt_ * n + N * n, Xw.begin() + b * N * n);
      std::copy(Ut + b * (Nt_ - 1) * m, Ut + b * (Nt_ - 1) * m + (N - 1) * m, Uw.begin() + b * (N - 1) * m);
      std::copy(Xt + b * Nt_ * n, Xt + b * Nt_ * n + n, xs.begin() + b * n);
    }
    if ((rc = put_planes(Xw.data(), Uw.data()))) return rc;
    return set_initial_state(xs.data());
  }
  int mpc_set_noise(const double* nzv, int steps) {
    if (!nzv || steps < 1) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    if (noise) WCHK(hipFree(noise));
    noise = nullptr;
    const size_t cnt = (size_t)steps * d.batch * d.n;
    WCHK(hipMalloc(&noise, cnt * sizeof(double)));
    WCHK(hipMemcpy(noise, nzv, cnt * sizeof(double), hipMemcpyHostToDevice));
    noise_steps = steps;
    return ALTRO_OK;
  }
  int mpc_set_noise_model(int mode, const double* w, const int32_t* g) {
    if (!w || mode < 0 || mode > 2) return ALTRO_ERR_INVALID_ARG;
    WCHK(hipSetDevice(device));
    std::vector<double> wv(kMaxN, 0.0);
    std::vector<int> gv(kMaxN, 0);
    for (int i = 0; i < d.n; ++i) {
      wv[i] = w[i];
      gv[i] = g ? g[i] : 0;
    }
    WCHK(hipMemcpy(noise_w, wv.data(), kMaxN * sizeof(double), hipMemcpyHostToDevice));
    WCHK(hipMemcpy(noise_grp, gv.data(), kMaxN * sizeof(int), hipMemcpyHostToDevice));
    noise_mode = mode;
    return ALTRO_OK;
  }
  int mpc_run(int first_step, int nsteps) {
    if (nsteps < 1 || first_step < 0) WFAIL(ALTRO_ERR_INVALID_ARG, "bad step range");
    if (noise && first_step + nsteps > noise_steps) WFAIL(ALTRO_ERR_INVALID_ARG, "steps outside the uploaded noise");
    if (first_step + nsteps + d.N > Nt) WFAIL(ALTRO_ERR_INVALID_ARG, "steps run past the end of the track");
    return enqueue(1, first_step, nsteps);
  }
};

#undef WCHK
#undef WFAIL

}  // namespace altro_wide
