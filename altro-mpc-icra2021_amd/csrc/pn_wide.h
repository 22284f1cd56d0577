// pn_wide.h -- the projected-Newton polish (pn_polish.h; Altro.jl solve!(::ProjectedNewtonSolver)) for the
// one-wave-per-instance backend: any n <= 64, m <= 32, per-knot dynamics, up to 64 generic rows.  Same algorithm, same
// oracle (oracle/altro_oracle.c projected_newton + multiplier_projection); PARITY WITH Altro.jl UNPINNED as there.
//
// Correctness first: the polish runs once per plain solve, on the instances whose AL stage ended above
// constraint_tolerance.  A block of 64 threads owns one WORKSPACE SLOT and walks the instances slot, slot + nslots, ...;
// every matrix block lives in HBM (a block row of S = D H^-1 D' has up to 2 n + active rows = ~200 entries at n = 64: two
// such blocks do not fit LDS), the serial structure is knot after knot as in pn_polish.h.  Data is read in the
// backend's own layouts (solve_wide.h Params): X [B][2][N][n], U [B][2][N-1][m], column-major dynamics blocks, the
// transposed row table AconT.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/altro_batch.h"
#include "solve_wide.h"

namespace altro_pnw {

constexpr int ZMAX = 96;  // n + m <= 64 + 32

struct WParams {
  altro_wide::Params P;
  int bm;                  // rows a block may hold
  int nslots;
  int *pn_ran, *pn_failed, *pn_dfail;        // [B]
  double *pn_res, *pn_dres0, *pn_dres;       // [B]
  // workspace, per slot
  double *E, *dv, *Ld, *Lo, *vec, *tz, *blk;  // E [N][bm][nz]; dv [N][bm]; Ld, Lo [N][bm][bm]; vec [6][N][bm]; tz [3][N][nz];
                                              // blk [2][bm][bm+1] + [4][bm]
  int *nb, *nst, *rinfo;                      // [N], [N], [N][bm]
};

__device__ __forceinline__ double wave_max(double v) {
  for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

struct PnW {
  const WParams& W;
  const altro_wide::Params& P;
  int inst = 0, tid, N, n, m, nz, bm, ld, kref;
  double *E, *dv, *Ld, *Lo, *lam, *res, *cor, *Sv, *dtr, *spare, *tz, *gz, *rz;
  int *nb, *nst, *rinfo;
  double *Lc, *Lp, *vv;
  bool unit = false;

  __device__ PnW(const WParams& w) : W(w), P(w.P) {
    tid = threadIdx.x;
    N = P.N; n = P.n; m = P.m; nz = n + m; bm = W.bm; ld = bm + 1; kref = P.kref;
    const size_t s = blockIdx.x;
    E = W.E + s * N * bm * nz;
    dv = W.dv + s * N * bm;
    Ld = W.Ld + s * N * bm * bm;
    Lo = W.Lo + s * N * bm * bm;
    double* v = W.vec + s * 6 * N * bm;
    lam = v; res = v + (size_t)N * bm; cor = v + 2 * (size_t)N * bm; Sv = v + 3 * (size_t)N * bm; dtr = v + 4 * (size_t)N * bm;
    spare = v + 5 * (size_t)N * bm;
    tz = W.tz + s * 3 * N * nz;
    gz = tz + (size_t)N * nz;
    rz = tz + 2 * (size_t)N * nz;
    nb = W.nb + s * N; nst = W.nst + s * N; rinfo = W.rinfo + s * N * bm;
    double* b = W.blk + s * (2 * (size_t)bm * ld + 4 * bm);
    Lc = b; Lp = b + (size_t)bm * ld; vv = b + 2 * (size_t)bm * ld;
  }

  // ---- the problem, in the backend's layouts
  __device__ __forceinline__ double zget(int pl, int k, int j) const {
    if (j < n) return P.X[(((size_t)inst * 2 + pl) * N + k) * n + j];
    return (k < N - 1) ? P.U[(((size_t)inst * 2 + pl) * (N - 1) + k) * m + (j - n)] : 0.0;
  }
  __device__ __forceinline__ void zset(int pl, int k, int j, double v) const {
    if (j < n) P.X[(((size_t)inst * 2 + pl) * N + k) * n + j] = v;
    else if (k < N - 1) P.U[(((size_t)inst * 2 + pl) * (N - 1) + k) * m + (j - n)] = v;
  }
  __device__ __forceinline__ double zref(int k, int j) const {
    if (j < n) return P.Xref[((size_t)inst * P.Nt + (kref + k)) * n + j];
    return (k < N - 1) ? P.Uref[((size_t)inst * (P.Nt - 1) + (kref + k)) * m + (j - n)] : 0.0;
  }
  __device__ __forceinline__ double hdiag(int k, int j) const { return (k < N - 1) ? P.wd[j] : (j < n ? P.wf[j] : 0.0); }
  __device__ __forceinline__ double hinv(int k, int j) const { return unit ? 1.0 : 1.0 / (hdiag(k, j) + P.o.rho_primal); }
  __device__ __forceinline__ size_t dynblk(int k) const {
    return (size_t)(P.dyn_per_instance ? inst : 0) * (P.ltv ? P.dyn_blocks : 1) + (P.ltv ? (size_t)kref * P.dyn_step_stride + k : 0);
  }
  __device__ __forceinline__ double G(int k, int i, int c) const {   // [A_k B_k][i][c]
    return c < n ? P.A[dynblk(k) * n * n + i + (size_t)n * c] : P.Bm[dynblk(k) * n * m + i + (size_t)n * (c - n)];
  }
  __device__ __forceinline__ double fdyn(int k, int i) const { return P.f[dynblk(k) * n + i]; }
  __device__ __forceinline__ double arow(int k, int r, int j) const {   // row r of knot k's table, column j
    return P.AconT[(size_t)inst * P.con_istride + ((size_t)k * nz + j) * P.Pn + r];
  }
  __device__ __forceinline__ double brow(int k, int r) const { return P.bcon[(size_t)inst * P.bcon_istride + (size_t)k * P.Pn + r]; }

  // Stage rows of knot k: code < 256: box side (upper: j, lower: 128 + j); 256 + r: generic row r (a cone by its first row)
  __device__ double pn_row(int code, int k, const double* z, double* Erow) const {
    const bool term = k == N - 1;
    const int ncol = term ? n : nz;
    if (Erow) for (int j = 0; j < nz; ++j) Erow[j] = 0.0;
    if (code < 256) {
      const int j = code & 127;
      if (code < 128) { if (Erow) Erow[j] = 1.0; return z[j] - P.zmax[j]; }
      if (Erow) Erow[j] = -1.0;
      return P.zmin[j] - z[j];
    }
    const int r0 = code - 256;
    auto val = [&](int r) {
      double acc = brow(k, r);
      for (int j = 0; j < ncol; ++j) acc += arow(k, r, j) * z[j];
      return acc;
    };
    if (P.ctype[(size_t)k * P.Pn + r0] != 3) {
      if (Erow) for (int j = 0; j < ncol; ++j) Erow[j] = arow(k, r0, j);
      return val(r0);
    }
    const int p = P.rowcp[r0], q = p - 1;
    double v[4], nv = 0.0;
    for (int r = 0; r < p; ++r) v[r] = val(r0 + r);
    for (int r = 0; r < q; ++r) nv += v[r] * v[r];
    nv = sqrt(nv);
    if (Erow) {
      for (int j = 0; j < ncol; ++j) {
        double g = -arow(k, r0 + q, j);
        if (nv > 0.0) for (int r = 0; r < q; ++r) g += v[r] / nv * arow(k, r0 + r, j);
        Erow[j] = g;
      }
    }
    return nv - v[q];
  }
  __device__ __forceinline__ bool row_on(int k, int r) const {
    const int t = P.ctype[(size_t)k * P.Pn + r];
    if (t == 0 || k < P.rowk0[r] || k > P.rowk1[r]) return false;
    return t != 3 || P.rowc0[r] == r;     // a cone is one row, coded by its first row
  }

  __device__ void load_z(int pl, int k, double* z) const {
    for (int j = 0; j < nz; ++j) z[j] = zget(pl, k, j);
  }

  __device__ double linearise_knot(int k, int pl) {
    const double tol = P.o.active_set_tolerance_pn;
    double z[ZMAX];
    load_z(pl, k, z);
    double* Ek = E + (size_t)k * bm * nz;
    double* dk = dv + (size_t)k * bm;
    int* ri = rinfo + (size_t)k * bm;
    int r = 0;
    if (k == 0) {
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < nz; ++j) Ek[(size_t)r * nz + j] = (j == i) ? 1.0 : 0.0;
        dk[r] = z[i] - P.x0[(size_t)inst * n + i];
        ++r;
      }
    }
    int ns = 0;
    if (k >= P.box_k0 && k <= P.box_k1) {
      const int lim = (k == N - 1) ? n : nz;
      for (int side = 0; side < 2; ++side)
        for (int j = 0; j < lim; ++j) {
          const bool has = side == 0 ? (P.zmax[j] < 1e300) : (P.zmin[j] > -1e300);
          if (!has) continue;
          const int code = side * 128 + j;
          const double v = pn_row(code, k, z, nullptr);
          if (!(v >= -tol) || r >= bm - n) continue;
          dk[r] = pn_row(code, k, z, Ek + (size_t)r * nz);
          ri[ns++] = code;
          ++r;
        }
    }
    for (int q = 0; q < P.Pn; ++q) {
      if (!row_on(k, q)) continue;
      const int code = 256 + q;
      const double v = pn_row(code, k, z, nullptr);
      const bool act = (P.ctype[(size_t)k * P.Pn + q] == 1) || (v >= -tol);
      if (!act || r >= bm - n) continue;
      dk[r] = pn_row(code, k, z, Ek + (size_t)r * nz);
      ri[ns++] = code;
      ++r;
    }
    nst[k] = ns;
    if (k < N - 1) {
      for (int i = 0; i < n; ++i) {
        double acc = fdyn(k, i);
        for (int c = 0; c < nz; ++c) {
          const double g = G(k, i, c);
          Ek[(size_t)r * nz + c] = g;
          acc += g * z[c];
        }
        dk[r] = acc - zget(pl, k + 1, i);
        ++r;
      }
    }
    nb[k] = r;
    double mx = 0.0;
    for (int q = 0; q < r; ++q) mx = fmax(mx, fabs(dk[q]));
    return mx;
  }

  __device__ double linearise(int pl) {
    double mx = 0.0;
    for (int k = tid; k < N; k += 64) mx = fmax(mx, linearise_knot(k, pl));
    __syncthreads();
    return wave_max(mx);
  }

  __device__ double values(int pl, double* out) {
    double mx = 0.0;
    for (int k = tid; k < N; k += 64) {
      double z[ZMAX];
      load_z(pl, k, z);
      double* dk = out + (size_t)k * bm;
      int r = 0;
      if (k == 0) for (int i = 0; i < n; ++i) dk[r++] = z[i] - P.x0[(size_t)inst * n + i];
      for (int q = 0; q < nst[k]; ++q) dk[r++] = pn_row(rinfo[(size_t)k * bm + q], k, z, nullptr);
      if (k < N - 1) {
        for (int i = 0; i < n; ++i) {
          double acc = fdyn(k, i);
          for (int c = 0; c < nz; ++c) acc += G(k, i, c) * z[c];
          dk[r++] = acc - zget(pl, k + 1, i);
        }
      }
      for (int q = 0; q < r; ++q) mx = fmax(mx, fabs(dk[q]));
    }
    __syncthreads();
    return wave_max(mx);
  }

  __device__ void apply_Dt(const double* v) {   // tz = H^-1 D' v
    for (int e = tid; e < N * nz; e += 64) {
      const int k = e / nz, j = e % nz;
      const double* Ek = E + (size_t)k * bm * nz;
      double acc = 0.0;
      for (int r = 0; r < nb[k]; ++r) acc += Ek[(size_t)r * nz + j] * v[(size_t)k * bm + r];
      if (k > 0 && j < n) acc -= v[(size_t)(k - 1) * bm + nb[k - 1] - n + j];
      tz[e] = acc * hinv(k, j);
    }
    __syncthreads();
  }
  __device__ void apply_D(const double* t, double* y) {   // y = D t
    for (int e = tid; e < N * bm; e += 64) {
      const int k = e / bm, r = e % bm;
      if (r >= nb[k]) continue;
      const double* Ek = E + (size_t)k * bm * nz;
      double acc = 0.0;
      for (int j = 0; j < nz; ++j) acc += Ek[(size_t)r * nz + j] * t[(size_t)k * nz + j];
      const int off = nb[k] - n;
      if (k < N - 1 && r >= off) acc -= t[(size_t)(k + 1) * nz + (r - off)];
      y[e] = acc;
    }
    __syncthreads();
  }
  __device__ void apply_S(const double* v, double* y) {
    apply_Dt(v);
    apply_D(tz, y);
  }

  __device__ bool factor() {   // S + rho_chol I = L L', block by block
    bool ok = true;
    for (int k = 0; k < N; ++k) {
      const int b = nb[k];
      const double* Ek = E + (size_t)k * bm * nz;
      for (int e = tid; e < b * b; e += 64) {
        const int r = e / b, c = e % b;
        double acc = 0.0;
        if (c <= r) {
          for (int j = 0; j < nz; ++j) acc += Ek[(size_t)r * nz + j] * hinv(k, j) * Ek[(size_t)c * nz + j];
          if (k < N - 1 && r == c && r >= b - n) acc += hinv(k + 1, r - (b - n));
          if (r == c) acc += P.o.rho_chol;
        }
        Lc[(size_t)r * ld + c] = acc;
      }
      __syncthreads();
      if (k > 0) {
        const int pb = nb[k - 1], poff = pb - n;
        for (int r = tid; r < b; r += 64) {
          double* lo = Lo + (size_t)k * bm * bm + (size_t)r * bm;
          for (int c = 0; c < pb; ++c) {
            double v = (c >= poff) ? -Ek[(size_t)r * nz + (c - poff)] * hinv(k, c - poff) : 0.0;
            for (int q = 0; q < c; ++q) v -= lo[q] * Lp[(size_t)c * ld + q];
            lo[c] = v / Lp[(size_t)c * ld + c];
          }
        }
        __syncthreads();
        for (int e = tid; e < b * b; e += 64) {
          const int r = e / b, c = e % b;
          if (c > r) continue;
          const double* lr = Lo + (size_t)k * bm * bm + (size_t)r * bm;
          const double* lc = Lo + (size_t)k * bm * bm + (size_t)c * bm;
          double acc = 0.0;
          for (int q = 0; q < pb; ++q) acc += lr[q] * lc[q];
          Lc[(size_t)r * ld + c] -= acc;
        }
        __syncthreads();
      }
      for (int c = 0; c < b; ++c) {
        double dd = Lc[(size_t)c * ld + c];
        for (int q = 0; q < c; ++q) dd -= Lc[(size_t)c * ld + q] * Lc[(size_t)c * ld + q];
        ok = ok && (dd > 0.0);
        const double piv = sqrt(dd > 0.0 ? dd : 1.0);
        __syncthreads();
        for (int r = c + 1 + tid; r < b; r += 64) {
          double v = Lc[(size_t)r * ld + c];
          for (int q = 0; q < c; ++q) v -= Lc[(size_t)r * ld + q] * Lc[(size_t)c * ld + q];
          Lc[(size_t)r * ld + c] = v / piv;
        }
        if (tid == 0) Lc[(size_t)c * ld + c] = piv;
        __syncthreads();
      }
      for (int e = tid; e < b * b; e += 64) {
        const int r = e / b, c = e % b;
        Ld[(size_t)k * bm * bm + (size_t)r * bm + c] = (c <= r) ? Lc[(size_t)r * ld + c] : 0.0;
        Lp[(size_t)r * ld + c] = Lc[(size_t)r * ld + c];
      }
      __syncthreads();
    }
    return ok;
  }

  __device__ void chol_solve(const double* bvec, double* x) {   // x = (L L')^-1 b
    for (int k = 0; k < N; ++k) {
      const int b = nb[k];
      const double* Lk = Ld + (size_t)k * bm * bm;
      for (int r = tid; r < b; r += 64) {
        double v = bvec[(size_t)k * bm + r];
        if (k > 0) {
          const double* lo = Lo + (size_t)k * bm * bm + (size_t)r * bm;
          for (int q = 0; q < nb[k - 1]; ++q) v -= lo[q] * x[(size_t)(k - 1) * bm + q];
        }
        vv[r] = v;
      }
      __syncthreads();
      for (int c = 0; c < b; ++c) {
        const double xc = vv[c] / Lk[(size_t)c * bm + c];
        __syncthreads();
        for (int r = c + 1 + tid; r < b; r += 64) vv[r] -= Lk[(size_t)r * bm + c] * xc;
        if (tid == 0) vv[c] = xc;
        __syncthreads();
      }
      for (int r = tid; r < b; r += 64) x[(size_t)k * bm + r] = vv[r];
      __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {
      const int b = nb[k];
      const double* Lk = Ld + (size_t)k * bm * bm;
      for (int r = tid; r < b; r += 64) {
        double v = x[(size_t)k * bm + r];
        if (k < N - 1) {
          const double* ln = Lo + (size_t)(k + 1) * bm * bm;
          for (int q = 0; q < nb[k + 1]; ++q) v -= ln[(size_t)q * bm + r] * x[(size_t)(k + 1) * bm + q];
        }
        vv[r] = v;
      }
      __syncthreads();
      for (int c = b - 1; c >= 0; --c) {
        const double xc = vv[c] / Lk[(size_t)c * bm + c];
        __syncthreads();
        for (int r = tid; r < c; r += 64) vv[r] -= Lk[(size_t)c * bm + r] * xc;
        if (tid == 0) vv[c] = xc;
        __syncthreads();
      }
      for (int r = tid; r < b; r += 64) x[(size_t)k * bm + r] = vv[r];
      __syncthreads();
    }
  }

  // reg_solve: A x = rhs with the factors of A + rho I, refined against A (apply_S under the metric in force)
  __device__ void reg_solve(const double* rhs, double* x) {
    chol_solve(rhs, x);
    for (int it = 0; it < 25; ++it) {
      apply_S(x, Sv);
      double rn = 0.0;
      for (int e = tid; e < N * bm; e += 64) {
        const double v = ((e % bm) < nb[e / bm]) ? rhs[e] - Sv[e] : 0.0;
        res[e] = v;
        rn = fmax(rn, fabs(v));
      }
      __syncthreads();
      rn = wave_max(rn);
      if (rn < 1e-8) break;
      chol_solve(res, spare);
      for (int e = tid; e < N * bm; e += 64)
        if ((e % bm) < nb[e / bm]) x[e] += spare[e];
      __syncthreads();
    }
  }

  __device__ void multiplier_projection(int cur) {   // pn_polish.h multiplier_projection, oracle multiplier_projection
    linearise(cur);
    unit = true;
    for (int e = tid; e < N * bm; e += 64) lam[e] = 0.0;
    for (int e = tid; e < N * nz; e += 64) {
      const int k = e / nz, j = e % nz;
      const bool live = j < n || k < N - 1;
      gz[e] = live ? hdiag(k, j) * (zget(cur, k, j) - zref(k, j)) : 0.0;
    }
    __syncthreads();
    for (int k = tid; k < N; k += 64) {
      const int base = (k == 0) ? n : 0;
      for (int q = 0; q < nst[k]; ++q) {
        const int code = rinfo[(size_t)k * bm + q];
        double l0 = 0.0;
        if (code < 256) l0 = P.Lb[(((size_t)inst * N + k) * 2 + (code >> 7)) * nz + (code & 127)];
        else if (P.ctype[(size_t)k * P.Pn + (code - 256)] != 3) l0 = P.Lc[((size_t)inst * N + k) * P.Pn + (code - 256)];
        lam[(size_t)k * bm + base + q] = l0;
      }
    }
    __syncthreads();
    apply_Dt(lam);
    double r0 = 0.0;
    for (int e = tid; e < N * nz; e += 64) {
      const double v = gz[e] + tz[e];
      rz[e] = v;
      r0 += v * v;
    }
    __syncthreads();
    r0 = wave_sum(r0);
    const bool ok = factor();
    if (ok) {
      apply_D(rz, dtr);
      reg_solve(dtr, cor);
      for (int e = tid; e < N * bm; e += 64)
        if ((e % bm) < nb[e / bm]) lam[e] -= cor[e];
      __syncthreads();
    }
    apply_Dt(lam);
    double r1 = 0.0;
    for (int e = tid; e < N * nz; e += 64) {
      const double v = gz[e] + tz[e];
      r1 += v * v;
    }
    r1 = wave_sum(r1);
    unit = false;
    if (tid == 0) {
      W.pn_dfail[inst] = ok ? 0 : 1;
      W.pn_dres0[inst] = sqrt(r0);
      W.pn_dres[inst] = sqrt(r1);
    }
  }

  __device__ void run(int instance, double ctol_user) {
    inst = instance;
    const altro_opts& o = P.o;
    const int cur = P.cur[inst];
    const bool need = (P.status[inst] <= ALTRO_SOLVE_SUCCEEDED) && (P.cmax[inst] > ctol_user);
    if (!need) {
      if (tid == 0) { W.pn_ran[inst] = 0; W.pn_failed[inst] = 0; W.pn_res[inst] = 0.0; W.pn_dfail[inst] = 0; W.pn_dres0[inst] = 0.0; W.pn_dres[inst] = 0.0; }
      return;
    }
    double viol = linearise(cur);
    bool failed = false;
    for (int outer = 0; outer <= 10 && viol > ctol_user; ++outer) {
      if (outer > 0) viol = linearise(cur);
      if (!factor()) { failed = true; break; }
      double viol_prev = viol;
      for (int refine = 0; refine < 10; ++refine) {
        reg_solve(dv, lam);
        apply_S(lam, Sv);  // tz = H^-1 D' lam: dz = -tz
        double alpha = 1.0, v_new = viol;
        for (int ls = 0;; ++ls) {
          for (int e = tid; e < N * nz; e += 64) zset(cur ^ 1, e / nz, e % nz, zget(cur, e / nz, e % nz) - alpha * tz[e]);
          __syncthreads();
          v_new = values(cur ^ 1, dtr);
          if (v_new < viol || ls >= 10) break;
          alpha *= 0.5;
        }
        for (int e = tid; e < N * nz; e += 64) zset(cur, e / nz, e % nz, zget(cur ^ 1, e / nz, e % nz));
        for (int e = tid; e < N * bm; e += 64) dv[e] = dtr[e];
        __syncthreads();
        viol = v_new;
        const double rate = log10(viol) / log10(viol_prev);
        viol_prev = viol;
        if (viol < ctol_user) break;
        if (rate < o.r_threshold) break;
      }
    }
    if (!failed) multiplier_projection(cur);
    else if (tid == 0) { W.pn_dfail[inst] = 1; W.pn_dres0[inst] = 0.0; W.pn_dres[inst] = 0.0; }
    __syncthreads();
    // objective (no AL terms) and violation of the problem's constraints at the polished trajectory
    double J = 0.0, cm = 0.0;
    for (int k = tid; k < N; k += 64) {
      const int lim = (k == N - 1) ? n : nz;
      double z[ZMAX];
      load_z(cur, k, z);
      for (int j = 0; j < lim; ++j) {
        const double e = z[j] - zref(k, j);
        J += 0.5 * hdiag(k, j) * e * e;
      }
      for (int j = lim; j < nz; ++j) z[j] = 0.0;
      if (k >= P.box_k0 && k <= P.box_k1)
        for (int j = 0; j < lim; ++j) {
          if (P.zmax[j] < 1e300) cm = fmax(cm, z[j] - P.zmax[j]);
          if (P.zmin[j] > -1e300) cm = fmax(cm, P.zmin[j] - z[j]);
        }
      for (int q = 0; q < P.Pn; ++q) {
        if (!row_on(k, q)) continue;
        const int t = P.ctype[(size_t)k * P.Pn + q];
        if (t == 3) {   // violation of a cone: ||Proj(v) - v||_inf (oracle con_violation)
          const int p = P.rowcp[q], qq = p - 1;
          double v[4], nv = 0.0;
          for (int r = 0; r < p; ++r) {
            double acc = brow(k, q + r);
            for (int j = 0; j < lim; ++j) acc += arow(k, q + r, j) * z[j];
            v[r] = acc;
          }
          for (int r = 0; r < qq; ++r) nv += v[r] * v[r];
          nv = sqrt(nv);
          const double tt = v[qq];
          if (nv <= tt) continue;
          if (nv <= -tt) { for (int r = 0; r < p; ++r) cm = fmax(cm, fabs(v[r])); continue; }
          const double c = 0.5 * (1.0 + tt / nv);
          for (int r = 0; r < qq; ++r) cm = fmax(cm, fabs(c * v[r] - v[r]));
          cm = fmax(cm, fabs(c * nv - tt));
        } else {
          const double v = pn_row(256 + q, k, z, nullptr);
          cm = fmax(cm, t == 1 ? fabs(v) : fmax(v, 0.0));
        }
      }
    }
    J = wave_sum(J);
    cm = wave_max(cm);
    if (tid == 0) {
      W.pn_ran[inst] = 1;
      W.pn_failed[inst] = failed ? 1 : 0;
      W.pn_res[inst] = viol;
      P.cost[inst] = J;
      P.cmax[inst] = cm;
      if (cm < ctol_user) P.status[inst] = ALTRO_SOLVE_SUCCEEDED;
    }
  }
};

// ctol_user: the caller's constraint_tolerance (P.o carries the AL stage's, i.e. projected_newton_tolerance)
__global__ void __launch_bounds__(64) pnw_kernel(WParams w, double ctol_user) {
  PnW s(w);
  for (int inst = blockIdx.x; inst < w.P.B; inst += gridDim.x) {
    s.run(inst, ctol_user);
    __syncthreads();
  }
}

}  // namespace altro_pnw
