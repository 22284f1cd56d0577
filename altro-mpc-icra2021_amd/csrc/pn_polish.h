// pn_polish.h -- projected-Newton polish (Altro.jl solve!(::ProjectedNewtonSolver); ALTRO, IROS 2019, Algorithm 4)
// for the 16-lane backend: SURVEY.md 8 f4.  Restated in oracle/altro_oracle.c (projected_newton), which is this
// file's parity oracle.  PARITY WITH Altro.jl IS UNPINNED: the reference stores no trajectory a polish produced (it is
// off in every live script, and skipped in the one old script that leaves it on).
//
// After the AL stage has stopped at projected_newton_tolerance, the primal trajectory z = (x_0, u_0, ..., x_{N-1}) is
// projected onto d(z) = 0 -- the initial condition, the dynamics defects and the ACTIVE rows of the constraints
// (equalities; inequality rows with c >= -active_set_tolerance_pn; a second-order cone through h = ||v|| - t) -- in the
// metric of the cost Hessian (diagonal here):
//     dz = -H^-1 D' (D H^-1 D')^-1 d
// S = D H^-1 D' is block tridiagonal over the knots (block k: [x_0 - x0 if k = 0; active stage rows; defect k]); it is
// factored as S + rho_chol I = L L' block by block and the solve is refined against S (Altro's reg_solve).
//
// Mapping: ONE wave per instance.  The serial structure is knot after knot (the block recursion); inside a knot the 64
// lanes share the rows / entries of the blocks.  The blocks of L stay in HBM (2 N bm^2 doubles per instance, allocated
// on the first solve that asks for the polish), the block in work sits in LDS.  This is a correctness-first kernel: the
// polish runs once per solve, on the instances whose AL stage ended above constraint_tolerance, and is not on the MPC
// hot path (projected_newton = false in every MPC script of the reference).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/altro_batch.h"

namespace altro_pn {

constexpr int LW = 16;
constexpr int BMAX = 56;  // rows a block may hold (2 n + active box sides + generic rows), checked on the host

struct PnParams {
  int B, Bp, N, n, m, bm;
  int Nt;                 // rows per instance of Zref
  int box_k0, box_k1, ncrows;
  unsigned con_istride;   // as SolveParams
  const double *Grow, *fvec, *wd, *wf, *zmin, *zmax, *x0;
  const double *Acon, *bcon;
  const int* cmeta;
  double* Z;              // [Bp][2 N + 1][16] (two planes of N rows + trash row per instance): plane cur is polished, plane cur^1 is the trial buffer
  const double* Zref;     // window start kref
  int kref;
  int* cur;
  int* status;
  double *cost, *cmax;
  int *pn_ran, *pn_failed;
  double* pn_res;
  // multiplier projection (the dual half of the polish): AL duals in, stationarity residuals out
  const double *Lb, *Lc;   // [Bp][N+1][2][nbp], [Bp][N+1][16]
  const int* bslot;        // [16]
  int nbp;
  int* pn_dfail;           // [Bp]
  double *pn_dres0, *pn_dres;   // [Bp] ||g + D' lam||_2 with the AL duals / with the projected multipliers
  // workspace, per instance
  double *E, *dv, *Ld, *Lo, *vec;  // E [N][bm][16]; dv [N][bm]; Ld, Lo [N][bm][bm]; vec [6][N][bm] (lam, res, cor, Sv, dtrial, spare)
  double* tz;                      // [3][N][16]: t of apply_S; the cost gradient g; g + D' lam
  int *nb, *nst, *rinfo;           // [N], [N], [N][bm] (row code: see pn_row)
  altro_opts o;
};

__device__ __forceinline__ double wave_max(double v) {
  for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
  return v;
}

struct Pn {
  const PnParams& P;
  int inst, tid, N, n, m, nz, bm;
  double *E, *dv, *Ld, *Lo, *lam, *res, *cor, *Sv, *dtr, *spare, *tz, *gz, *rz;
  bool unit = false;   // the metric H = I of the multiplier projection (D D' instead of D H^-1 D')
  int *nb, *nst, *rinfo;
  double* Lc;  // LDS: current block [bm][bm+1]
  double* Lp;  // LDS: second block
  double* vv;  // LDS: vectors [4][bm]

  __device__ Pn(const PnParams& p, double* lds) : P(p) {
    inst = blockIdx.x;
    tid = threadIdx.x;
    N = P.N; n = P.n; m = P.m; nz = n + m; bm = P.bm;
    const size_t i = (size_t)inst;
    E = P.E + i * N * bm * LW;
    dv = P.dv + i * N * bm;
    Ld = P.Ld + i * N * bm * bm;
    Lo = P.Lo + i * N * bm * bm;
    double* v = P.vec + i * 6 * N * bm;
    lam = v; res = v + (size_t)N * bm; cor = v + 2 * (size_t)N * bm; Sv = v + 3 * (size_t)N * bm; dtr = v + 4 * (size_t)N * bm;
    spare = v + 5 * (size_t)N * bm;
    tz = P.tz + i * 3 * N * LW;
    gz = tz + (size_t)N * LW;
    rz = tz + 2 * (size_t)N * LW;
    nb = P.nb + i * N; nst = P.nst + i * N; rinfo = P.rinfo + i * N * bm;
    Lc = lds;
    Lp = lds + BMAX * (BMAX + 1);
    vv = lds + 2 * BMAX * (BMAX + 1);
  }

  __device__ __forceinline__ const double* zrow(int plane, int k) const {
    return P.Z + ((size_t)inst * (2 * (size_t)N + 1) + (size_t)plane * N + k) * LW;
  }
  __device__ __forceinline__ double* zrow_w(int plane, int k) const {
    return P.Z + ((size_t)inst * (2 * (size_t)N + 1) + (size_t)plane * N + k) * LW;
  }
  __device__ __forceinline__ double hinv(int k, int j) const {
    if (unit) return 1.0;
    const double h = (k < N - 1) ? P.wd[j] : (j < n ? P.wf[j] : 0.0);
    return 1.0 / (h + P.o.rho_primal);
  }
  __device__ __forceinline__ double G(int i, int c) const { return P.Grow[((size_t)inst * LW + c) * LW + i]; }  // [A B][i][c]

  // Stage rows of knot k are coded as: 0 .. 2*16-1: box side (j upper: code j, lower: code 16 + j); 64 + lane: generic
  // row on constraint lane `lane` (a cone is coded by its first lane).  Value of row `code` at z (and its Jacobian row).
  __device__ double pn_row(int code, int k, const double* z, double* Erow) const {
    const bool term = k == N - 1;
    const int ncol = term ? n : nz;
    if (Erow) for (int j = 0; j < LW; ++j) Erow[j] = 0.0;
    if (code < 32) {
      const int j = code & 15;
      if (code < 16) { if (Erow) Erow[j] = 1.0; return z[j] - P.zmax[j]; }
      if (Erow) Erow[j] = -1.0;
      return P.zmin[j] - z[j];
    }
    const int lane = code - 64;
    const int* cm = P.cmeta + ((size_t)k * LW + lane) * 4;
    const size_t ab = (size_t)inst * P.con_istride + ((size_t)k * LW) * LW;
    const size_t bb = (size_t)inst * (P.con_istride / LW) + (size_t)k * LW;
    auto val = [&](int r) {
      double acc = P.bcon[bb + r];
      for (int j = 0; j < ncol; ++j) acc += P.Acon[ab + (size_t)r * LW + j] * z[j];
      return acc;
    };
    if (cm[0] != 3) {
      if (Erow) for (int j = 0; j < ncol; ++j) Erow[j] = P.Acon[ab + (size_t)lane * LW + j];
      return val(lane);
    }
    const int p = cm[3], q = p - 1;
    double v[4], nv = 0.0;
    for (int r = 0; r < p; ++r) v[r] = val(lane + r);
    for (int r = 0; r < q; ++r) nv += v[r] * v[r];
    nv = sqrt(nv);
    if (Erow) {
      for (int j = 0; j < ncol; ++j) {
        double g = -P.Acon[ab + (size_t)(lane + q) * LW + j];
        if (nv > 0.0) for (int r = 0; r < q; ++r) g += v[r] / nv * P.Acon[ab + (size_t)(lane + r) * LW + j];
        Erow[j] = g;
      }
    }
    return nv - v[q];
  }

  // active set + linearisation of knot k at plane pl (one thread per knot); returns max |d| of the knot
  __device__ double linearise_knot(int k, int pl) {
    const double tol = P.o.active_set_tolerance_pn;
    double z[LW], zn[LW];
    const double* zr = zrow(pl, k);
    for (int j = 0; j < LW; ++j) z[j] = (j < n || (j < nz && k < N - 1)) ? zr[j] : 0.0;
    double* Ek = E + (size_t)k * bm * LW;
    double* dk = dv + (size_t)k * bm;
    int* ri = rinfo + (size_t)k * bm;
    int r = 0;
    if (k == 0) {
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < LW; ++j) Ek[(size_t)r * LW + j] = (j == i) ? 1.0 : 0.0;
        dk[r] = z[i] - P.x0[(size_t)inst * LW + i];
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
          const int code = side * 16 + j;
          const double v = pn_row(code, k, z, nullptr);
          if (!(v >= -tol) || r >= bm - n) continue;
          dk[r] = pn_row(code, k, z, Ek + (size_t)r * LW);
          ri[ns++] = code;
          ++r;
        }
    }
    if (P.ncrows > 0) {
      for (int lane = 0; lane < LW; ++lane) {
        const int* cm = P.cmeta + ((size_t)k * LW + lane) * 4;
        if (cm[0] == 0 || k < cm[1] || k > cm[2]) continue;
        if (cm[0] == 3 && (lane & 3) != 0) continue;    // a cone is one row, coded by its first lane
        const int code = 64 + lane;
        const double v = pn_row(code, k, z, nullptr);
        const bool act = (cm[0] == 1) || (v >= -tol);
        if (!act || r >= bm - n) continue;
        dk[r] = pn_row(code, k, z, Ek + (size_t)r * LW);
        ri[ns++] = code;
        ++r;
      }
    }
    nst[k] = ns;
    if (k < N - 1) {
      const double* zr1 = zrow(pl, k + 1);
      for (int i = 0; i < n; ++i) zn[i] = zr1[i];
      for (int i = 0; i < n; ++i) {
        double acc = P.fvec[(size_t)inst * LW + i];
        for (int c = 0; c < nz; ++c) {
          const double g = G(i, c);
          Ek[(size_t)r * LW + c] = g;
          acc += g * z[c];
        }
        for (int c = nz; c < LW; ++c) Ek[(size_t)r * LW + c] = 0.0;
        dk[r] = acc - zn[i];
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

  // values of the same rows at plane pl into out; returns max |d|
  __device__ double values(int pl, double* out) {
    double mx = 0.0;
    for (int k = tid; k < N; k += 64) {
      double z[LW];
      const double* zr = zrow(pl, k);
      for (int j = 0; j < LW; ++j) z[j] = (j < n || (j < nz && k < N - 1)) ? zr[j] : 0.0;
      double* dk = out + (size_t)k * bm;
      int r = 0;
      if (k == 0) for (int i = 0; i < n; ++i) dk[r++] = z[i] - P.x0[(size_t)inst * LW + i];
      for (int q = 0; q < nst[k]; ++q) dk[r++] = pn_row(rinfo[(size_t)k * bm + q], k, z, nullptr);
      if (k < N - 1) {
        const double* zr1 = zrow(pl, k + 1);
        for (int i = 0; i < n; ++i) {
          double acc = P.fvec[(size_t)inst * LW + i];
          for (int c = 0; c < nz; ++c) acc += G(i, c) * z[c];
          dk[r++] = acc - zr1[i];
        }
      }
      for (int q = 0; q < r; ++q) mx = fmax(mx, fabs(dk[q]));
    }
    __syncthreads();
    return wave_max(mx);
  }

  // y = S v; also leaves t_k = H_k^-1 (E_k' v_k - [defect part of v_{k-1}]_x) in tz
  __device__ void apply_S(const double* v, double* y) {
    apply_Dt(v);
    apply_D(tz, y);
  }
  // tz = H^-1 D' v
  __device__ void apply_Dt(const double* v) {
    for (int e = tid; e < N * LW; e += 64) {
      const int k = e / LW, j = e % LW;
      double acc = 0.0;
      if (j < nz) {
        const double* Ek = E + (size_t)k * bm * LW;
        for (int r = 0; r < nb[k]; ++r) acc += Ek[(size_t)r * LW + j] * v[(size_t)k * bm + r];
        if (k > 0 && j < n) acc -= v[(size_t)(k - 1) * bm + nb[k - 1] - n + j];
        acc *= hinv(k, j);
      }
      tz[e] = acc;
    }
    __syncthreads();
  }
  // y = D t   (t: [N][16])
  __device__ void apply_D(const double* t, double* y) {
    for (int e = tid; e < N * bm; e += 64) {
      const int k = e / bm, r = e % bm;
      if (r >= nb[k]) continue;
      const double* Ek = E + (size_t)k * bm * LW;
      double acc = 0.0;
      for (int j = 0; j < nz; ++j) acc += Ek[(size_t)r * LW + j] * t[(size_t)k * LW + j];
      const int off = nb[k] - n;
      if (k < N - 1 && r >= off) acc -= t[(size_t)(k + 1) * LW + (r - off)];
      y[e] = acc;
    }
    __syncthreads();
  }

  __device__ __forceinline__ double wave_sum(double v) const {
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
  }

  // multiplier_projection! of Altro.jl's ProjectedNewtonSolver (oracle multiplier_projection): lam <- lam - (D D')^-1 D (g + D' lam)
  // at the polished trajectory, with the stationarity residual before and after.  The multipliers are not written back.
  __device__ void multiplier_projection(int cur) {
    linearise(cur);
    unit = true;
    for (int e = tid; e < N * bm; e += 64) lam[e] = 0.0;
    for (int e = tid; e < N * LW; e += 64) {   // g_k = H_k (z_k - zref_k)
      const int k = e / LW, j = e % LW;
      const bool live = j < n || (j < nz && k < N - 1);
      const double h = (k < N - 1) ? P.wd[j] : (j < n ? P.wf[j] : 0.0);
      gz[e] = live ? h * (zrow(cur, k)[j] - P.Zref[((size_t)inst * P.Nt + (size_t)(P.kref + k)) * LW + j]) : 0.0;
    }
    __syncthreads();
    for (int k = tid; k < N; k += 64) {        // lam0: the AL duals of the active box / linear rows
      const int base = (k == 0) ? n : 0;
      for (int q = 0; q < nst[k]; ++q) {
        const int code = rinfo[(size_t)k * bm + q];
        double l0 = 0.0;
        if (code < 32) {
          const int sl = P.bslot[code & 15];
          if (sl >= 0) l0 = P.Lb[(((size_t)inst * (N + 1) + k) * 2 + (code >> 4)) * P.nbp + sl];
        } else {
          const int lane = code - 64;
          if (P.cmeta[((size_t)k * LW + lane) * 4] != 3) l0 = P.Lc[((size_t)inst * (N + 1) + k) * LW + lane];
        }
        lam[(size_t)k * bm + base + q] = l0;
      }
    }
    __syncthreads();
    apply_Dt(lam);
    double r0 = 0.0;
    for (int e = tid; e < N * LW; e += 64) {
      const double v = gz[e] + tz[e];
      rz[e] = v;
      r0 += v * v;
    }
    __syncthreads();
    r0 = wave_sum(r0);
    const bool ok = factor();
    if (ok) {
      apply_D(rz, dtr);                        // rhs = D (g + D' lam0)
      chol_solve(dtr, cor);
      for (int it = 0; it < 25; ++it) {        // reg_solve against D D'
        apply_S(cor, Sv);
        double rn = 0.0;
        for (int e = tid; e < N * bm; e += 64) {
          const double v = ((e % bm) < nb[e / bm]) ? dtr[e] - Sv[e] : 0.0;
          res[e] = v;
          rn = fmax(rn, fabs(v));
        }
        __syncthreads();
        rn = wave_max(rn);
        if (rn < 1e-8) break;
        chol_solve(res, spare);
        for (int e = tid; e < N * bm; e += 64)
          if ((e % bm) < nb[e / bm]) cor[e] += spare[e];
        __syncthreads();
      }
      for (int e = tid; e < N * bm; e += 64)
        if ((e % bm) < nb[e / bm]) lam[e] -= cor[e];
      __syncthreads();
    }
    apply_Dt(lam);
    double r1 = 0.0;
    for (int e = tid; e < N * LW; e += 64) {
      const double v = gz[e] + tz[e];
      r1 += v * v;
    }
    r1 = wave_sum(r1);
    unit = false;
    if (tid == 0) {
      P.pn_dfail[inst] = ok ? 0 : 1;
      P.pn_dres0[inst] = sqrt(r0);
      P.pn_dres[inst] = sqrt(r1);
    }
  }

  // S + rho_chol I = L L'.  Returns false if a pivot is not positive.
  __device__ bool factor() {
    const int ld = BMAX + 1;
    bool ok = true;
    for (int k = 0; k < N; ++k) {
      const int b = nb[k];
      const double* Ek = E + (size_t)k * bm * LW;
      // S_kk into Lc
      for (int e = tid; e < b * b; e += 64) {
        const int r = e / b, c = e % b;
        double acc = 0.0;
        if (c <= r) {
          for (int j = 0; j < nz; ++j) acc += Ek[(size_t)r * LW + j] * hinv(k, j) * Ek[(size_t)c * LW + j];
          if (k < N - 1 && r == c && r >= b - n) acc += hinv(k + 1, r - (b - n));
          if (r == c) acc += P.o.rho_chol;
        }
        Lc[r * ld + c] = acc;
      }
      __syncthreads();
      if (k > 0) {
        // L_{k,k-1} = S_{k,k-1} L_{k-1,k-1}^-T, row r per thread; Lp holds L_{k-1,k-1}
        const int pb = nb[k - 1], poff = pb - n;
        for (int r = tid; r < b; r += 64) {
          double* lo = Lo + (size_t)k * bm * bm + (size_t)r * bm;
          for (int c = 0; c < pb; ++c) {
            double v = (c >= poff) ? -Ek[(size_t)r * LW + (c - poff)] * hinv(k, c - poff) : 0.0;
            for (int q = 0; q < c; ++q) v -= lo[q] * Lp[c * ld + q];
            lo[c] = v / Lp[c * ld + c];
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
          Lc[r * ld + c] -= acc;
        }
        __syncthreads();
      }
      // Cholesky of the block in LDS, column by column
      for (int c = 0; c < b; ++c) {
        double dd = Lc[c * ld + c];
        for (int q = 0; q < c; ++q) dd -= Lc[c * ld + q] * Lc[c * ld + q];
        ok = ok && (dd > 0.0);
        const double piv = sqrt(dd > 0.0 ? dd : 1.0);
        __syncthreads();
        for (int r = c + 1 + tid; r < b; r += 64) {
          double v = Lc[r * ld + c];
          for (int q = 0; q < c; ++q) v -= Lc[r * ld + q] * Lc[c * ld + q];
          Lc[r * ld + c] = v / piv;
        }
        if (tid == 0) Lc[c * ld + c] = piv;
        __syncthreads();
      }
      for (int e = tid; e < b * b; e += 64) {
        const int r = e / b, c = e % b;
        Ld[(size_t)k * bm * bm + (size_t)r * bm + c] = (c <= r) ? Lc[r * ld + c] : 0.0;
        Lp[r * ld + c] = Lc[r * ld + c];
      }
      __syncthreads();
    }
    return ok;
  }

  // x = (L L')^-1 b
  __device__ void chol_solve(const double* bvec, double* x) {
    const int ld = BMAX + 1;
    for (int k = 0; k < N; ++k) {  // forward
      const int b = nb[k];
      for (int e = tid; e < b * b; e += 64) Lc[(e / b) * ld + (e % b)] = Ld[(size_t)k * bm * bm + (size_t)(e / b) * bm + (e % b)];
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
        const double xc = vv[c] / Lc[c * ld + c];
        __syncthreads();
        for (int r = c + 1 + tid; r < b; r += 64) vv[r] -= Lc[r * ld + c] * xc;
        if (tid == 0) vv[c] = xc;
        __syncthreads();
      }
      for (int r = tid; r < b; r += 64) x[(size_t)k * bm + r] = vv[r];
      __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {  // backward
      const int b = nb[k];
      for (int e = tid; e < b * b; e += 64) Lc[(e / b) * ld + (e % b)] = Ld[(size_t)k * bm * bm + (size_t)(e / b) * bm + (e % b)];
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
        const double xc = vv[c] / Lc[c * ld + c];
        __syncthreads();
        for (int r = tid; r < c; r += 64) vv[r] -= Lc[c * ld + r] * xc;
        if (tid == 0) vv[c] = xc;
        __syncthreads();
      }
      for (int r = tid; r < b; r += 64) x[(size_t)k * bm + r] = vv[r];
      __syncthreads();
    }
  }

  __device__ void run() {
    const altro_opts& o = P.o;
    const int cur = P.cur[inst];
    const bool need = (P.status[inst] <= ALTRO_SOLVE_SUCCEEDED) && (P.cmax[inst] > o.constraint_tolerance);
    if (!need) {
      if (tid == 0) { P.pn_ran[inst] = 0; P.pn_failed[inst] = 0; P.pn_res[inst] = 0.0; P.pn_dfail[inst] = 0; P.pn_dres0[inst] = 0.0; P.pn_dres[inst] = 0.0; }
      return;
    }
    double viol = linearise(cur);
    bool failed = false;
    for (int outer = 0; outer <= 10 && viol > o.constraint_tolerance; ++outer) {
      if (outer > 0) viol = linearise(cur);
      if (!factor()) { failed = true; break; }
      double viol_prev = viol;
      for (int refine = 0; refine < 10; ++refine) {
        chol_solve(dv, lam);
        for (int it = 0; it < 25; ++it) {
          apply_S(lam, Sv);
          double rn = 0.0;
          for (int e = tid; e < N * bm; e += 64) {
            const int k = e / bm, r = e % bm;
            const double v = (r < nb[k]) ? dv[e] - Sv[e] : 0.0;
            res[e] = v;
            rn = fmax(rn, fabs(v));
          }
          __syncthreads();
          rn = wave_max(rn);
          if (rn < 1e-8) break;
          chol_solve(res, cor);
          for (int e = tid; e < N * bm; e += 64)
            if ((e % bm) < nb[e / bm]) lam[e] += cor[e];
          __syncthreads();
        }
        apply_S(lam, Sv);  // tz = H^-1 D' lam: dz = -tz
        double alpha = 1.0, v_new = viol;
        for (int ls = 0;; ++ls) {
          for (int e = tid; e < N * LW; e += 64) {
            const int k = e / LW, j = e % LW;
            const bool live = j < n || (j < nz && k < N - 1);
            zrow_w(cur ^ 1, k)[j] = live ? zrow(cur, k)[j] - alpha * tz[e] : 0.0;
          }
          __syncthreads();
          v_new = values(cur ^ 1, dtr);
          if (v_new < viol || ls >= 10) break;
          alpha *= 0.5;
        }
        for (int e = tid; e < N * LW; e += 64) zrow_w(cur, e / LW)[e % LW] = zrow(cur ^ 1, e / LW)[e % LW];
        for (int e = tid; e < N * bm; e += 64) dv[e] = dtr[e];
        __syncthreads();
        viol = v_new;
        const double rate = log10(viol) / log10(viol_prev);
        viol_prev = viol;
        if (viol < o.constraint_tolerance) break;
        if (rate < o.r_threshold) break;
      }
    }
    if (!failed) multiplier_projection(cur);
    else if (tid == 0) { P.pn_dfail[inst] = 1; P.pn_dres0[inst] = 0.0; P.pn_dres[inst] = 0.0; }
    __syncthreads();
    // objective (no AL terms) and violation of the problem's constraints at the polished trajectory
    double J = 0.0, cm = 0.0;
    for (int k = tid; k < N; k += 64) {
      const double* zr = zrow(cur, k);
      const double* rr = P.Zref + ((size_t)inst * P.Nt + (size_t)(P.kref + k)) * LW;
      const int lim = (k == N - 1) ? n : nz;
      for (int j = 0; j < lim; ++j) {
        const double e = zr[j] - rr[j];
        J += 0.5 * ((k < N - 1) ? P.wd[j] : P.wf[j]) * e * e;
      }
      double z[LW];
      for (int j = 0; j < LW; ++j) z[j] = (j < lim) ? zr[j] : 0.0;
      if (k >= P.box_k0 && k <= P.box_k1)
        for (int j = 0; j < lim; ++j) {
          if (P.zmax[j] < 1e300) cm = fmax(cm, z[j] - P.zmax[j]);
          if (P.zmin[j] > -1e300) cm = fmax(cm, P.zmin[j] - z[j]);
        }
      if (P.ncrows > 0)
        for (int lane = 0; lane < LW; ++lane) {
          const int* cmm = P.cmeta + ((size_t)k * LW + lane) * 4;
          if (cmm[0] == 0 || k < cmm[1] || k > cmm[2]) continue;
          if (cmm[0] == 3) {
            if (lane & 3) continue;
            // violation of a cone: ||Proj(v) - v||_inf (oracle con_violation)
            const int p = cmm[3], q = p - 1;
            double v[4], nv = 0.0;
            const size_t ab = (size_t)inst * P.con_istride + ((size_t)k * LW) * LW;
            const size_t bb = (size_t)inst * (P.con_istride / LW) + (size_t)k * LW;
            for (int r = 0; r < p; ++r) {
              double acc = P.bcon[bb + lane + r];
              for (int j = 0; j < lim; ++j) acc += P.Acon[ab + (size_t)(lane + r) * LW + j] * z[j];
              v[r] = acc;
            }
            for (int r = 0; r < q; ++r) nv += v[r] * v[r];
            nv = sqrt(nv);
            const double t = v[q];
            if (nv <= t) continue;
            if (nv <= -t) { for (int r = 0; r < p; ++r) cm = fmax(cm, fabs(v[r])); continue; }
            const double c = 0.5 * (1.0 + t / nv);
            for (int r = 0; r < q; ++r) cm = fmax(cm, fabs(c * v[r] - v[r]));
            cm = fmax(cm, fabs(c * nv - t));
          } else {
            const double v = pn_row(64 + lane, k, z, nullptr);
            cm = fmax(cm, cmm[0] == 1 ? fabs(v) : fmax(v, 0.0));
          }
        }
    }
    for (int s = 32; s >= 1; s >>= 1) J += __shfl_xor(J, s, 64);
    cm = wave_max(cm);
    if (tid == 0) {
      P.pn_ran[inst] = 1;
      P.pn_failed[inst] = failed ? 1 : 0;
      P.pn_res[inst] = viol;
      P.cost[inst] = J;
      P.cmax[inst] = cm;
      if (cm < o.constraint_tolerance) P.status[inst] = ALTRO_SOLVE_SUCCEEDED;
    }
  }
};

__global__ void __launch_bounds__(64) pn_kernel(PnParams p) {
  __shared__ double lds[2 * BMAX * (BMAX + 1) + 4 * BMAX];
  Pn s(p, lds);
  s.run();
}

}  // namespace altro_pn
