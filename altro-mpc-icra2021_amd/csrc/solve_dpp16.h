// solve_dpp16.h -- whole AL-iLQR solve (Altro.jl `solve!`) as ONE kernel for gfx950.
//
// Mapping (MI355X-first, see DESIGN.md "Kernel"):
//   * one 16-lane DPP row  = one MPC instance; one wave64 = 4 instances; one workgroup = 1 wave.
//   * lane j < NX owns column j of every n-column matrix of the Riccati recursion (S, A, S*A,
//     Qxx, Qux) and element j of every state vector; lane NX+a owns the column/element of
//     control a (B, S*B, Quu, Qu, u, box duals of u).  NX+NU <= 16.
//   * every matrix product is a sequence of `v_fmac_f64_dpp ... row_newbcast:k` (dpp_blocks.inc):
//     the broadcast operand comes out of a neighbour lane's register, the other operand and the
//     accumulator are the lane's own registers.  No LDS traffic and no shuffles in the products;
//     S, A, B stay in VGPRs for the whole backward pass.
//   * the serial structure of the solve (AL outer loop / iLQR iterations / line search) runs
//     inside the kernel with per-instance predicates; branches are wave-uniform (ballot), so
//     EXEC is all ones wherever a DPP instruction executes.
//
// Reference call sites served: solve!(altro) random_linear_problem.jl:113,161;
// algorithm restated from SURVEY.md Appendix A (rows P2-P9 of SURVEY 8a), the same restatement
// as oracle/altro_oracle.c, which is the parity oracle for this file.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/altro_batch.h"

namespace altro {

#include "dpp_blocks.inc"

constexpr int LW = 16;  // lanes per instance (one DPP row)
constexpr int IPW = 4;  // instances per wave

struct SolveParams {
  int B, Bp, N;
  int kref;              // first knot of the reference window inside Zref
  int box_k0, box_k1;    // knot range of the BOX constraint (box_k1 < box_k0: none)
  const double* Gcol;    // [Bp][NX][16]   Gcol[b][k][j] = [A B][k][j]
  const double* Grow;    // [Bp][16][16]   Grow[b][c][i] = [A B][i][c]
  const double* fvec;    // [Bp][16]       affine term (x lanes)
  const double* wd;      // [16] dt*Q (x lanes) | dt*R (u lanes)
  const double* wf;      // [16] Qf (x lanes) | 0
  const double* zmin;    // [16]
  const double* zmax;    // [16]
  const double* x0;      // [Bp][16]
  const double* Zref;    // [Nt][Bp][16]
  double* Z;             // [2][N][Bp][16]  ping-pong trajectories
  int* cur;              // [Bp] which plane of Z is current
  double* Lhi;           // [N][Bp][16] duals of z - zmax <= 0
  double* Llo;           // [N][Bp][16] duals of zmin - z <= 0
  double* mu;            // [Bp] box penalty (uniform over rows/knots)
  double* KD;            // [N-1][Bp][NU][16] gains: row a = K[a][0..NX-1], then d[a] in lanes >= NX
  int* iters;
  int* iters_outer;
  int* status;
  double* cost;
  double* cmax;
  double* Jtrace;        // [Bp][ALTRO_TRACE_LEN]
  double* ctrace;        // [Bp][ALTRO_TRACE_LEN]
  long long* n_backward; // [Bp] work counters (accumulated across launches)
  long long* n_rollout;  // [Bp]
  altro_opts o;
};

template <int K>
__device__ __forceinline__ double bcast(double v) {
  // v_mov_b64_dpp row_newbcast:K -- value of lane K of this lane's 16-lane row
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);
}

template <int I, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, E>(f);
  }
}

__device__ __forceinline__ double row_sum(double v) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  double one = 1.0;
  BlkCommon::ROWSUM(acc, v, one);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__device__ __forceinline__ double row_max(double v) {
  double m = v;
  sfor<0, 16>([&](auto k) { m = fmax(m, bcast<decltype(k)::value>(v)); });
  return m;
}

__device__ __forceinline__ bool row_any(bool p, int lane) {
  unsigned long long b = __ballot(p);
  return ((b >> (lane & 48)) & 0xFFFFull) != 0ull;
}

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// regularization_update! (Altro.jl iLQR) -- same arithmetic as oracle reg_update()
__device__ __forceinline__ void reg_update(double& rho, double& drho, const altro_opts& o, bool inc) {
  if (inc) {
    drho = fmax(drho * o.bp_reg_increase_factor, o.bp_reg_increase_factor);
    rho = fmax(rho * drho, o.bp_reg_min);
  } else {
    drho = fmin(drho / o.bp_reg_increase_factor, 1.0 / o.bp_reg_increase_factor);
    rho = rho * drho * ((rho * drho > o.bp_reg_min) ? 1.0 : 0.0);
  }
}

template <int NX, int NU>
struct Solver {
  static constexpr int NZ = NX + NU;
  static_assert(NZ <= LW, "packed kernel needs n + m <= 16");

  const SolveParams& P;
  int lane, j, inst;
  bool is_x, is_u;
  size_t rowoff;   // inst*16 + j
  size_t kstride;  // Bp*16
  double wd, wf, zmin, zmax;
  bool has_hi, has_lo;
  double x0;
  double mu;
  int cur;

  __device__ Solver(const SolveParams& p) : P(p) {
    lane = threadIdx.x & 63;
    j = lane & 15;
    inst = blockIdx.x * IPW + (lane >> 4);
    is_x = j < NX;
    is_u = (j >= NX) && (j < NZ);
    rowoff = (size_t)inst * LW + j;
    kstride = (size_t)P.Bp * LW;
    wd = P.wd[j];
    wf = P.wf[j];
    zmin = P.zmin[j];
    zmax = P.zmax[j];
    has_hi = zmax < 1e300;
    has_lo = zmin > -1e300;
    x0 = P.x0[rowoff];
    cur = P.cur[inst];
  }

  __device__ __forceinline__ size_t at(int k) const { return (size_t)k * kstride + rowoff; }
  __device__ __forceinline__ double* plane(int c) const { return P.Z + (size_t)c * P.N * kstride; }
  __device__ __forceinline__ bool box_at(int k) const { return k >= P.box_k0 && k <= P.box_k1; }

  // stage cost + box AL term of this lane's element; updates the lane's violation maximum.
  // (oracle total_cost / con_cost; TO.jl cost!  -- SURVEY A.2)
  __device__ __forceinline__ double lane_cost(double z, double zr, double w, double lhi, double llo,
                                              bool box_on, double& viol) const {
    double e = z - zr;
    double J = 0.5 * w * e * e;
    if (box_on) {
      double chi = z - zmax, clo = zmin - z;
      bool ahi = (chi >= 0.0) || (lhi > 0.0);
      bool alo = (clo >= 0.0) || (llo > 0.0);
      double Jhi = lhi * chi + (ahi ? 0.5 * mu * chi * chi : 0.0);
      double Jlo = llo * clo + (alo ? 0.5 * mu * clo * clo : 0.0);
      J += has_hi ? Jhi : 0.0;
      J += has_lo ? Jlo : 0.0;
      viol = fmax(viol, has_hi ? chi : 0.0);
      viol = fmax(viol, has_lo ? clo : 0.0);
    }
    return J;
  }

  struct RollOut {
    double J, cmax, grad_new, grad_old;
    bool limit;
  };

  // rollout!(solver[, alpha]) fused with cost!(obj, Z̄), max_violation and gradient_todorov!.
  //   OPEN : open-loop rollout of plane `cur` from x0 (iLQR initialize!), in place
  //   !OPEN: closed-loop rollout with gains KD and step alpha from plane cur into plane cur^1;
  //          stores are predicated on `store` (per instance)
  template <bool OPEN>
  __device__ RollOut rollout(double alpha, bool store) {
    const double* zs = plane(cur);
    double* zd = OPEN ? plane(cur) : plane(cur ^ 1);
    double grow[NZ];
    sfor<0, NZ>([&](auto c) {
      constexpr int C = decltype(c)::value;
      grow[C] = P.Grow[((size_t)inst * LW + C) * LW + j];
    });
    const double fv = P.fvec[rowoff];
    double xb = x0;
    double Jacc = 0.0, viol = 0.0, gnew = 0.0, gold = 0.0;
    bool limit = false;
    const int N = P.N;
    for (int k = 0; k < N - 1; ++k) {
      const double z = zs[at(k)];
      const double zr = P.Zref[at(P.kref + k)];
      const bool bx = box_at(k);
      double lhi = 0.0, llo = 0.0;
      if (bx) {
        lhi = P.Lhi[at(k)];
        llo = P.Llo[at(k)];
      }
      double zb;
      if constexpr (OPEN) {
        zb = is_x ? xb : z;
      } else {
        // u-lane NX+a reads row a of the gain block: K[a][0..NX-1] and d[a] (stored in lanes >= NX)
        double krow[NX];
        double dff = 0.0;
        const int ra = is_u ? (j - NX) : 0;
        const double* kd = P.KD + (((size_t)k * P.Bp + inst) * NU + ra) * LW;
        sfor<0, NX>([&](auto c) { krow[decltype(c)::value] = kd[decltype(c)::value]; });
        dff = kd[NX];
        double dx = is_x ? (xb - z) : 0.0;
        double acc[3] = {0.0, 0.0, 0.0};
        Blk<NX, NU>::KDX(acc, dx, krow);
        double du = (acc[0] + acc[1]) + acc[2];
        double ub = z + du + alpha * dff;
        zb = is_x ? xb : ub;
        double gn = is_u ? fabs(dff) / (fabs(ub) + 1.0) : 0.0;
        double go = is_u ? fabs(dff) / (fabs(z) + 1.0) : 0.0;
        gnew += row_max(gn);
        gold += row_max(go);
        if (store) zd[at(k)] = zb;
      }
      if constexpr (OPEN) {
        if (store) zd[at(k)] = zb;
      }
      Jacc += lane_cost(zb, zr, wd, lhi, llo, bx, viol);
      const double lim = is_x ? P.o.max_state_value : P.o.max_control_value;
      limit = limit || ((is_x || is_u) && !(fabs(zb) <= lim));
      double acc4[4] = {fv, 0.0, 0.0, 0.0};
      Blk<NX, NU>::GZ(acc4, zb, grow);
      xb = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    }
    {  // terminal knot: state only
      const int k = N - 1;
      const double zr = P.Zref[at(P.kref + k)];
      const bool bx = box_at(k);
      double lhi = 0.0, llo = 0.0;
      if (bx) {
        lhi = P.Lhi[at(k)];
        llo = P.Llo[at(k)];
      }
      double zb = is_x ? xb : 0.0;
      if (store) zd[at(k)] = zb;
      Jacc += lane_cost(zb, zr, wf, lhi, llo, bx && is_x, viol);
      limit = limit || (is_x && !(fabs(zb) <= P.o.max_state_value));
    }
    RollOut r;
    r.J = row_sum(Jacc);
    r.cmax = row_max(viol);
    r.grad_new = gnew / (double)(N - 1);
    r.grad_old = gold / (double)(N - 1);
    r.limit = row_any(limit, lane);
    return r;
  }

  // backwardpass! (SURVEY A.3 / oracle backward_pass): Riccati recursion over plane `cur`,
  // writes the gain blocks KD, returns dV and whether any Quu pivot was not positive.
  __device__ void backward(double rho, double& dV1, double& dV2, bool& fail, double* sm) {
    const double* zs = plane(cur);
    double g[NX];
    sfor<0, NX>([&](auto c) {
      constexpr int C = decltype(c)::value;
      g[C] = P.Gcol[((size_t)inst * NX + C) * LW + j];
    });
    const int N = P.N;
    // terminal expansion: S = Qf (+ box hessian), s = Qf (x - xr) (+ box gradient)
    double Sx[NX + 1];
    {
      const int k = N - 1;
      const double z = zs[at(k)];
      const double zr = P.Zref[at(P.kref + k)];
      double qz = wf * (z - zr), hz = wf;
      if (box_at(k)) {
        const double lhi = P.Lhi[at(k)], llo = P.Llo[at(k)];
        box_expand(z, lhi, llo, is_x, qz, hz);
      }
      sfor<0, NX>([&](auto c) {
        constexpr int C = decltype(c)::value;
        Sx[C] = (j == C) ? hz : 0.0;
      });
      Sx[NX] = is_x ? qz : 0.0;
    }
    dV1 = 0.0;
    dV2 = 0.0;
    fail = false;
    for (int k = N - 2; k >= 0; --k) {
      const double z = zs[at(k)];
      const double zr = P.Zref[at(P.kref + k)];
      double qz = wd * (z - zr), hz = wd;
      if (box_at(k)) {
        const double lhi = P.Lhi[at(k)], llo = P.Llo[at(k)];
        box_expand(z, lhi, llo, true, qz, hz);
      }
      // W = [S; s'] * G   (w[NX] = (G's)[lane])
      double w[NX + 1];
      sfor<0, NX + 1>([&](auto c) { w[decltype(c)::value] = 0.0; });
      Blk<NX, NU>::SG(w, Sx, g);
      // H = G' W + diag(lzz): x lanes hold [Qxx(:,j); Qux(:,j)], u lanes [Qxu(:,a); Quu(:,a)]
      double h[NZ];
      sfor<0, NZ>([&](auto c) {
        constexpr int C = decltype(c)::value;
        h[C] = (j == C) ? hz : 0.0;
      });
      Blk<NX, NU>::GtW(h, g, w);
      const double gz = qz + w[NX];  // Qx[j] on x lanes, Qu[a] on u lanes
      // gather Quu (lower triangle) and Qu to every lane
      double quu[NU][NU];
      double qu[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        qu[A] = bcast<NX + A>(gz);
        sfor<0, A + 1>([&](auto b) {
          constexpr int Bq = decltype(b)::value;
          quu[A][Bq] = bcast<NX + Bq>(h[NX + A]);
          quu[Bq][A] = quu[A][Bq];
        });
      });
      // Cholesky of Quu + rho I (redundantly on every lane)
      double L[NU][NU];
      double inv[NU];
      sfor<0, NU>([&](auto jc) {
        constexpr int Jc = decltype(jc)::value;
        double dd = quu[Jc][Jc] + rho;
        sfor<0, Jc>([&](auto kk) { dd -= L[Jc][decltype(kk)::value] * L[Jc][decltype(kk)::value]; });
        fail = fail || !(dd > 0.0);
        const double sq = sqrt(dd);
        L[Jc][Jc] = sq;
        inv[Jc] = 1.0 / sq;
        sfor<Jc + 1, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = quu[I][Jc];
          sfor<0, Jc>([&](auto kk) { v -= L[I][decltype(kk)::value] * L[Jc][decltype(kk)::value]; });
          L[I][Jc] = v * inv[Jc];
        });
      });
      // right-hand side: x lanes Qux(:,j), u lanes Qu  -> kd = -Quu_reg^{-1} rhs
      double r[NU], kd[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        r[A] = is_x ? h[NX + A] : qu[A];
      });
      {
        double y[NU];
        sfor<0, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = r[I];
          sfor<0, I>([&](auto kk) { v -= L[I][decltype(kk)::value] * y[decltype(kk)::value]; });
          y[I] = v * inv[I];
        });
        sfor<0, NU>([&](auto ir) {
          constexpr int I = NU - 1 - decltype(ir)::value;
          double v = y[I];
          sfor<I + 1, NU>([&](auto kk) { v -= L[decltype(kk)::value][I] * kd[decltype(kk)::value]; });
          kd[I] = v * inv[I];
        });
        sfor<0, NU>([&](auto a) { kd[decltype(a)::value] = -kd[decltype(a)::value]; });
      }
      // T = Quu*kd + rhs : x lanes (Quu K + Qux)(:,j), u lanes Quu d + Qu
      double T[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        double v = r[A];
        sfor<0, NU>([&](auto b) { v += quu[A][decltype(b)::value] * kd[decltype(b)::value]; });
        T[A] = v;
      });
      // d and (Quu d + Qu) to every lane (from the first u lane)
      double dd_[NU], Td[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        dd_[A] = bcast<NX>(kd[A]);
        Td[A] = bcast<NX>(T[A]);
      });
      // s = Qx + K'(Quu d + Qu) + Qux' d
      double snew = gz;
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        snew += kd[A] * Td[A] + r[A] * dd_[A];
      });
      // dV += (d'Qu, 1/2 d'Quu d)
      double t1 = 0.0, t2 = 0.0;
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        t1 += dd_[A] * qu[A];
        t2 += 0.5 * dd_[A] * (Td[A] - qu[A]);
      });
      dV1 += t1;
      dV2 += t2;
      // gains out: row a lanes 0..NX-1 = K[a][:], lanes >= NX = d[a]
      {
        double* kdp = P.KD + (((size_t)k * P.Bp + inst) * NU) * LW + j;
        sfor<0, NU>([&](auto a) {
          constexpr int A = decltype(a)::value;
          kdp[A * LW] = is_x ? kd[A] : dd_[A];
        });
      }
      // S = Qxx + K'(Quu K + Qux) + Qux'K   (in place on h[0..NX-1]), then S = (S + S')/2
      Blk<NX, NU>::CTG(h, kd, T, r);
      {
        double* my = sm + (size_t)(lane >> 4) * (LW * (LW + 1));
        sfor<0, NX>([&](auto c) {
          constexpr int C = decltype(c)::value;
          my[C * (LW + 1) + j] = h[C];
        });
        __syncthreads();
        sfor<0, NX>([&](auto c) {
          constexpr int C = decltype(c)::value;
          const double st = my[j * (LW + 1) + C];
          Sx[C] = is_x ? 0.5 * (h[C] + st) : 0.0;
        });
        __syncthreads();
      }
      Sx[NX] = is_x ? snew : 0.0;
    }
  }

  // gradient / Gauss-Newton hessian of the box AL term of this lane's element
  __device__ __forceinline__ void box_expand(double z, double lhi, double llo, bool on, double& qz,
                                             double& hz) const {
    const double chi = z - zmax, clo = zmin - z;
    const bool ahi = (chi >= 0.0) || (lhi > 0.0);
    const bool alo = (clo >= 0.0) || (llo > 0.0);
    const double ghi = lhi + (ahi ? mu * chi : 0.0);
    const double glo = llo + (alo ? mu * clo : 0.0);
    if (on && has_hi) {
      qz += ghi;
      hz += ahi ? mu : 0.0;
    }
    if (on && has_lo) {
      qz -= glo;
      hz += alo ? mu : 0.0;
    }
  }

  // dual_update! for the box rows of plane `cur` (penalty_update! is the caller's mu *= phi)
  __device__ void dual_update(bool upd) {
    const double* zs = plane(cur);
    const double dmax = P.o.dual_max;
    for (int k = P.box_k0; k <= P.box_k1; ++k) {
      const bool on = (k < P.N - 1) ? (is_x || is_u) : is_x;
      const double z = zs[at(k)];
      const double lhi = P.Lhi[at(k)], llo = P.Llo[at(k)];
      const double nhi = fmin(fmax(lhi + mu * (z - zmax), 0.0), dmax);
      const double nlo = fmin(fmax(llo + mu * (zmin - z), 0.0), dmax);
      if (upd && on && has_hi) P.Lhi[at(k)] = nhi;
      if (upd && on && has_lo) P.Llo[at(k)] = nlo;
    }
  }

  // solve!(::ALTROSolver) -> AL outer loop -> iLQR (oracle orc_solve)
  __device__ void solve(double* sm) {
    const altro_opts& o = P.o;
    const double mu0 = (o.penalty_initial != o.penalty_initial) ? 1.0 : o.penalty_initial;
    const double phi = (o.penalty_scaling != o.penalty_scaling) ? 10.0 : o.penalty_scaling;
    mu = o.reset_penalties ? mu0 : P.mu[inst];
    int status = ALTRO_UNSOLVED;
    int iters = 0, iters_outer = 0;
    int nbw = 0, nro = 0;  // work counters of this instance
    bool alive = true;
    double J = 0.0, cmax = 0.0;
    const bool has_con = P.box_k1 >= P.box_k0;
    const int n_outer = has_con ? o.iterations_outer : 1;

    for (int outer = 0; outer < n_outer; ++outer) {
      if (!wave_any(alive)) break;
      const bool last = (outer == n_outer - 1);
      const double cost_tol = (!last && has_con) ? o.cost_tolerance_intermediate : o.cost_tolerance;
      const double grad_tol = (!last && has_con) ? o.gradient_tolerance_intermediate : o.gradient_tolerance;
      // ---------------- iLQR solve
      double rho = o.bp_reg_initial, drho = 0.0;
      int dj_zero = 0;
      bool inner = alive;
      RollOut r0 = rollout<true>(0.0, inner);
      if (inner) nro++;
      double J_prev = r0.J;
      if (inner) {
        J = r0.J;
        cmax = r0.cmax;
      }
      if (inner && r0.limit) {
        status = ALTRO_STATE_LIMIT;
        J = __builtin_inf();
        cmax = __builtin_inf();
        inner = false;
      }
      for (int it = 0; it < o.iterations_inner; ++it) {
        if (!wave_any(inner)) break;
        double dV1, dV2;
        // backward pass (with regularisation restarts)
        while (true) {
          bool fail;
          backward(rho, dV1, dV2, fail, sm);
          if (inner) nbw++;
          fail = row_any(fail, lane) && inner;
          if (fail) {
            if (rho >= o.bp_reg_max) {
              status = ALTRO_NO_PROGRESS;
              inner = false;
            } else {
              reg_update(rho, drho, o, true);
            }
          }
          if (!wave_any(fail && inner)) {
            if (!fail) reg_update(rho, drho, o, false);
            break;
          }
        }
        // forward pass: line search on alpha
        double alpha = 1.0, zr = -1.0, Jn = __builtin_inf(), cm_n = cmax, g_n = 0.0;
        int ls = 0;
        bool searching = inner, accepted = false;
        while (true) {
          const bool failnow = searching && (ls > o.iterations_linesearch);
          if (failnow) {
            Jn = J_prev;
            cm_n = cmax;
            alpha = 0.0;
            reg_update(rho, drho, o, true);
            rho += o.bp_reg_fp;
            searching = false;
          }
          if (!wave_any(searching)) break;
          RollOut rr = rollout<false>(alpha, searching);
          if (searching) nro++;
          if (searching) {
            if (rr.limit) {
              ls++;
              alpha *= 0.5;
            } else {
              Jn = rr.J;
              const double expected = -alpha * (dV1 + alpha * dV2);
              zr = (expected > 0.0) ? (J_prev - Jn) / expected : -1.0;
              ls++;
              const bool again = ((zr <= o.line_search_lower_bound) || (zr > o.line_search_upper_bound)) &&
                                 (Jn >= J_prev);
              if (!again) {
                searching = false;
                accepted = true;
                cm_n = rr.cmax;
                g_n = rr.grad_new;
              } else {
                g_n = rr.grad_old;  // value used if the search ends in failure
                alpha *= 0.5;
              }
            }
          }
        }
        if (inner) {
          if (Jn > o.max_cost_value) {
            status = ALTRO_MAXIMUM_COST;
            J = Jn;
            inner = false;
          } else {
            if (accepted) cur ^= 1;  // copy_trajectories!
            cmax = cm_n;
            const double dJ = fabs(Jn - J_prev);
            J_prev = Jn;
            J = Jn;
            if (iters < ALTRO_TRACE_LEN && j == 0) {
              P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + iters] = J;
              P.ctrace[(size_t)inst * ALTRO_TRACE_LEN + iters] = cmax;
            }
            iters++;
            dj_zero = (dJ == 0.0) ? dj_zero + 1 : 0;
            if (dJ < cost_tol && g_n < grad_tol) {
              inner = false;
            } else if (iters >= o.iterations) {
              status = ALTRO_MAX_ITERATIONS;
              inner = false;
            } else if (dj_zero > o.dJ_counter_limit) {
              status = ALTRO_NO_PROGRESS;
              inner = false;
            }
          }
        }
      }
      // ---------------- AL outer update
      bool upd = false;
      if (alive) {
        if (has_con) iters_outer++;
        if (!has_con) {
          if (status == ALTRO_UNSOLVED) status = ALTRO_SOLVE_SUCCEEDED;
          cmax = 0.0;
          alive = false;
        } else if (status > ALTRO_SOLVE_SUCCEEDED) {
          alive = false;
        } else if (cmax < o.constraint_tolerance || mu >= o.penalty_max) {
          alive = false;
        } else if (last) {
          status = ALTRO_MAX_ITERATIONS_OUTER;
          alive = false;
        } else {
          upd = true;
        }
      }
      if (wave_any(upd)) {
        dual_update(upd);
        if (upd) mu = fmin(fmax(phi * mu, 0.0), o.penalty_max);
      }
    }
    if (has_con && status <= ALTRO_SOLVE_SUCCEEDED && cmax < o.constraint_tolerance)
      status = ALTRO_SOLVE_SUCCEEDED;
    if (j == 0) {
      P.iters[inst] = iters;
      P.iters_outer[inst] = iters_outer;
      P.status[inst] = status;
      P.cost[inst] = J;
      P.cmax[inst] = cmax;
      P.mu[inst] = mu;
      P.cur[inst] = cur;
      P.n_backward[inst] += nbw;
      P.n_rollout[inst] += nro;
    }
  }
};

template <int NX, int NU>
__global__ void __launch_bounds__(64, 2) solve_kernel(SolveParams p) {
  __shared__ double sm[IPW * LW * (LW + 1)];
  Solver<NX, NU> s(p);
  s.solve(sm);
}

}  // namespace altro
