// solve_dpp16.h -- whole AL-iLQR solve (Altro.jl `solve!`), and whole MPC steps around it, as
// ONE kernel for gfx950.
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
//   * HOT LOOPS ARE SINGLE BASIC BLOCKS.  Operands of later knots are requested several knots
//     ahead and consumed out of registers; hipcc's wait-count insertion can only emit counted
//     `s_waitcnt vmcnt(N)` for that if the loop body has no branches -- with any (even uniform)
//     branch in the body it falls back to `vmcnt(0)` at the first use and every knot pays a full
//     memory round trip (measured: 2.3 k cycles per rollout knot).  So inside the per-knot loops:
//     indices are clamped instead of guarded, per-knot conditions are selects, stores are
//     unconditional (into planes that are dead for the rows that do not need them).
//   * the serial structure of the solve (AL outer loop / iLQR iterations / line search) and of
//     the MPC loop around it (plant step, shift_fill, retarget, solve) runs inside the kernel as
//     a PER-ROW STATE MACHINE: each of the four rows of a wave walks through its own
//     (MPC step, AL outer iteration, iLQR iteration) sequence.  One turn of the wave loop runs,
//     for whichever rows need it: [plant step + solve setup] -> [open-loop rollout] ->
//     [backward pass, alpha = 1 rollout, line-search sweeps, convergence test] -> [dual update].
//     A row that converges starts its next MPC step in the next turn while its wave-mates keep
//     iterating, so a row never idles waiting for a slower neighbour (measured before this:
//     56 wave-iterations per 42 instance-iterations).  Branches are wave-uniform (ballot), so
//     EXEC is all ones wherever a DPP instruction executes; rows that sit out a phase have their
//     stores redirected to a trash slot.  Waves never synchronise with each other.
//   * row state (costs, tolerances, counters, phase) lives in LDS between phases, so that it
//     does not occupy VGPRs during the register-hungry backward pass.
//
// Reference call sites served: solve!(altro) random_linear_problem.jl:113,161; the MPC update
// sequence :121-139; algorithm restated from SURVEY.md Appendix A (rows P2-P12 of SURVEY 8a),
// the same restatement as oracle/altro_oracle.c, which is the parity oracle for this file.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/altro_batch.h"

// tuning knobs (defaults = the shipped configuration; overridable with -D for experiments)
#ifndef ALTRO_PD_OPEN
#define ALTRO_PD_OPEN 8    // knots of prefetch in the open-loop rollout (measured 2..8: 8 best once the loop is branch-free)
#endif
#ifndef ALTRO_PD_CLOSED
#define ALTRO_PD_CLOSED 4  // knots of prefetch in the closed-loop rollout (2..8 measured with the butterfly gain sums: 4 best)
#endif
#ifndef ALTRO_PD_ADJOINT
#define ALTRO_PD_ADJOINT 13  // knots per group of the costate sweep (one load per knot, two register sets: see adjoint())
#endif
#ifndef ALTRO_PD_FOSWEEP
#define ALTRO_PD_FOSWEEP 4  // knots of prefetch in the first-order sweep (2 + NU loads per knot)
#endif
#ifndef ALTRO_UN
#define ALTRO_UN 4           // knots per chunk in the streaming sweeps
#endif
#ifndef ALTRO_NA
#define ALTRO_NA 4           // line-search step sizes evaluated per streaming sweep
#endif
#ifndef ALTRO_PRIO_SERIAL
// Issue priority (s_setprio) of a wave while it is in a phase that is a serial dependency chain with few
// instructions (rollouts, gradient sweep); the backward pass runs at 0.  When the two waves of a SIMD are
// in different phases, the chain's next instruction issues as soon as it is ready and the backward pass
// of the other wave -- hundreds of independent FMAs per knot -- fills every other slot.
#define ALTRO_PRIO_SERIAL 1  // measured 0..3 on the headline: 1 is +1 %, 2 and 3 about the same, 0 = off
#endif
#ifndef ALTRO_PRIO_HARD
// Issue priority of a wave while one of its rows is in a solve that did not end with its second iteration (a
// "hard" solve: 3 % of them, 5-20 iterations).  Those rows are the launch's critical path -- the slowest wave
// takes twice the mean wave's time at 20 steps per launch -- and while such a wave still shares its SIMD it
// advances at 0.73 of the speed it reaches alone.  0 = off.
#define ALTRO_PRIO_HARD 2
#endif
#ifndef ALTRO_PRIO_HEAVY
// Issue priority of the waves in the upper half of a GROUPED launch (SolveParams::perm: instances sorted by the backward
// passes they are expected to need).  With two waves per SIMD the dispatcher pairs block i with block i + grid / 2, i.e. a
// pass-free wave with a pass-heavy one; the pass-free wave finishes its steps in a quarter of the launch either way, the
// pass-heavy one is the launch's critical path: while they share the SIMD it should issue first.  0 = off.
#define ALTRO_PRIO_HEAVY 0   // measured 2 and 3 over twelve windows: single windows -5 %, one +13 % (its slowest wave sat in the lower half), mean unchanged
#endif
#ifndef ALTRO_PRIO_LAG
#define ALTRO_PRIO_LAG 0  // measured 4 and 8 on the headline: within run-to-run noise of 0 (off)
#endif
#ifndef ALTRO_WAVES_PER_SIMD
#define ALTRO_WAVES_PER_SIMD 2  // register budget: 512 / this
#endif

namespace altro {

#include "dpp_blocks.inc"

constexpr int LW = 16;  // lanes per instance (one DPP row)
constexpr int IPW = 4;  // instances per wave

// The EXACT active set of one lane: two bits per knot (which sides of the lane's box entered the Hessian), 16 knots per
// word, N <= 128 knots.  It guards the gains kept in KD for reuse -- across solves and launches -- so it is compared
// bit for bit, not through a hash (rounds 2-3: 32-, then 64-bit hashes).  Kept in LDS while a launch runs (the sweeps
// OR their knots into it with ds_or_b32: no registers, the same two VALU instructions per knot the hash took) and in
// SolveParams::ahash between launches.  Longer horizons do not reuse gains (Solver::run, kvalid).
constexpr int ASET_WORDS = 8;
constexpr int ASET_MAXN = 16 * ASET_WORDS;
struct ASet {
  unsigned w[ASET_WORDS];
};

struct SolveParams {
  int B, Bp, N;
  int Nt;                // knots held by Zref (rows per instance)
  int kref;              // first knot of the reference window inside Zref (plain solve)
  int box_k0, box_k1;    // knot range of the BOX constraint (box_k1 < box_k0: none)
  int first_step;        // MPC mode: first step index
  int nsteps;            // MPC mode: steps to run in this launch; 0 = plain solve!()
  int prepare_only;      // MPC mode: plant step + new x0 only (altro_mpc_prepare_async): no shift, no solve
  const double* Gcol;    // [Bp][NX][16]   Gcol[b][k][j] = [A B][k][j]
  const double* Grow;    // [Bp][16][16]   Grow[b][c][i] = [A B][i][c]
  const double* fvec;    // [Bp][16]       affine term (x lanes)
  const double* wd;      // [16] dt*Q (x lanes) | dt*R (u lanes)
  const double* wf;      // [16] Qf (x lanes) | 0
  const double* zmin;    // [16]
  const double* zmax;    // [16]
  double* x0;            // [Bp][16]
  const double* Zref;    // [Bp][Nt][16]
  const double* noise;   // [steps][B][n] unit normals of the plant noise (may be null)
  const double* noise_w; // [16] per-state noise weight
  const int* noise_grp;  // [16] per-state norm group (0 or 1)
  int noise_mode;        // 0: w_i * ||x||_inf (random_linear_problem.jl:129); 1: w_i * ||x[group_i]||_2
                         // (simple_rocket.jl:65-71); 2: w_i (absolute, flexible_sat_mpc.jl:266)
  int mpc_shift;         // 1: shift_fill primal + dual at every MPC step (default); 0: keep (flexible_sat_mpc.jl:275-276)
  double* Z;             // [Bp][2 N + 1][16]  per instance: two ping-pong planes of N rows, then one trash row
  int* cur;              // [Bp] which plane of Z is current
  const int* perm;       // [Bp] wave slot -> instance (grouped MPC launches: altro_batch.hip k_group_score); null: identity
  double* Lb;            // [Bp][N+1][2][nbp] box duals of the nbp bounded elements of z: side 0 = duals of
                         // z - zmax <= 0, side 1 = duals of zmin - z <= 0  (knot N = trash row)
  const int* bslot;      // [16] slot of lane j's element among the bounded ones, -1 if unbounded
  int nbp;               // slots per side (>= 1)
  double* mu;            // [Bp] box penalty (uniform over rows/knots)
  int dbg_wave;          // -DALTRO_PHASE_STAMPS builds: the wave whose turns are traced behind the wave_cycles records
  int resync;            // 1: rows wait a turn to stay in step with their wave-mates (run(), phase A)
  int reuse;             // 1: gain reuse (fosweep) allowed; ALTRO_NO_REUSE=1 at create time switches it off
  int shadow;  // rows that sit a phase out take the identity of a row that takes part (Solver::shadow_enter)
  int lone;              // 1: a backward pass that only one row of a wave needs runs spread over the four DPP rows (backward_lone)
  int useqz;             // 1: a backward pass over a trajectory a rollout has just produced reads its expansion back from Qz (backward QV)
  // generic affine constraints (LINEAR eq/ineq, SOC): up to 16 constraint rows per knot, row r
  // on lane r, organised in 4 quads of 4 lanes; a quad is one cone (SOC of dimension <= 4, or
  // up to 4 independent equality / inequality rows).  Data is shared by all instances.
  // All tables are PER KNOT (time-varying data: grasp_problem.jl:35-67; lanes can be reused by
  // constraints whose knot ranges do not overlap):
  const double* Acon;    // [N][16][16] row-major: value_r = sum_j Acon[k][r][j] z_j + bcon[k][r]
  const double* bcon;    // [N][16]
  unsigned con_istride;  // 0: Acon / bcon are shared by the batch; N*16*16: one table per instance ([Bp][N][16][16], [Bp][N][16])
  const int* cmeta;      // [N][16][4] per lane: type (0 none, 1 EQ, 2 INEQ, 3 SOC), k0, k1 of the row's
                         // constraint, p (dimension of the cone this lane's quad holds at this knot;
                         // linear rows may use the quad's spare lanes)
  const int* ckn;        // [16] canonical knot of each constraint lane
  int con_inv;           // 1: the row a lane holds is the same at every knot of its range (time-invariant tables): the
                         // streaming sweeps load it once, from knot ckn[lane]
  double* Lc;            // [Bp][N+1][16] duals of the constraint rows (knot N = trash row)
  int ncrows;            // 0: no generic constraints
  double* Qz;            // [Bp][N+1][16] gradient of the AL cost at the trajectory the last alpha = 1 rollout produced
                         // (l_x on the state lanes, l_u on the control lanes; knot N = trash): input of the costate sweep
  double* KD;            // [Bp][N][NU][16] gains: row a = K[a][0..NX-1] in the x lanes; lane NX + b (b <= a) holds entry (a, b) of
                         // the factors of Quu = L D L' (1 / D_a on the diagonal, L below it); block N-1 = trash
  double* Dff;           // [Bp][N+1][16] feedforward terms: d[a] on lane NX + a (knot N = trash)
  ASet* ahash;           // [Bp][16] per lane: active set of the backward pass that left the gains in KD (kept between launches)
  double* kmu;           // [Bp] penalty of that pass; < 0: the gains in KD must not be reused
  long long* n_fo;       // [Bp] iterations that took their gains from memory (first-order sweep instead of a backward pass)
  int* iters;
  int* iters_outer;
  int* status;
  double* cost;
  double* cmax;
  double* Jtrace;        // [Bp][ALTRO_TRACE_LEN]
  double* ctrace;        // [Bp][ALTRO_TRACE_LEN]
  double* atrace;        // [Bp][ALTRO_TRACE_LEN] accepted line-search step (0 = search failed)
  long long* n_backward; // [Bp] work counters (accumulated across launches)
  long long* n_rollout;  // [Bp]
  long long* n_trials;   // [Bp] line-search trials evaluated by interpolation
  long long* n_solves;   // [Bp]
  long long* n_iters;    // [Bp]
  long long* n_ok;       // [Bp] solves that ended SOLVE_SUCCEEDED
  long long* n_gconf;    // [Bp] iterations confirmed by the costate sweep instead of a backward pass
  int* dzero;            // [Bp] 1: the feedforward terms of the last solve's last iteration are zero (costate-confirmed)
  unsigned* simd_tab;    // [65536][16] or null: per physical SIMD and wave slot, the work a resident wave has left (see Solver::mate_rank)
  long long* wave_cycles; // [Bp/4][16] shader cycles of the last launch, per wave (diagnostic):
                          // total, backward, closed rollouts, open rollouts, todorov, dual update,
                          // streaming line-search sweeps
  altro_opts o;
};

template <int K>
__device__ __forceinline__ double bcast(double v) {
  // v_mov_b64_dpp row_newbcast:K -- value of lane K of this lane's 16-lane row
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // two v_mov_b32_dpp: quad_perm 0x00..0xFF, row_ror:n 0x120 + n (within the 16-lane row)
  return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;

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

// cycle stamp for the per-phase diagnostic counters (one asm statement, fenced both sides).
// Compiled in only with -DALTRO_PHASE_STAMPS (diagnostic build): the accumulators cost
// registers in the production kernel.
#ifdef ALTRO_PHASE_STAMPS
#define ALTRO_STAMP(x) x
#else
#define ALTRO_STAMP(x)
#endif
__device__ __forceinline__ long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return (long long)t;
}

// Global accesses as uniform base (SGPR pair) + 32-bit per-lane BYTE offset: this is the pattern
// hipcc lowers to `global_load_dwordx2 v, v_off, s[base:base+1]` (one address VGPR instead of a
// 64-bit pointer per array and unroll slot).  Element index -> byte offset is computed in 32 bits
// on purpose; the host guarantees every array is smaller than 4 GiB.
__device__ __forceinline__ double ldg(const double* base, unsigned idx) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (idx << 3));
}
__device__ __forceinline__ void stg(double* base, unsigned idx, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (idx << 3)) = v;
}

// 1/x to full FP64 accuracy: v_rcp_f64 + two Newton steps (x > 0, normal range)
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}

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

// value of lane (lane & ~3) + Q of this lane's quad (4 consecutive lanes): two 32-bit DPP moves
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);  // quad_perm:[Q,Q,Q,Q]
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// Per-lane metadata of the constraint row this lane owns
struct ConMeta {
  int type, k0, k1, p, pos;  // pos = lane & 3
};
enum { CT_NONE = 0, CT_EQ = 1, CT_INEQ = 2, CT_SOC = 3 };

// Result of evaluating one constraint row (on its lane) at one knot
struct ConeEval {
  double cost;     // AL penalty contribution of this row
  double viol;     // violation contribution (max-reduced over rows)
  double g;        // d phi / d value_r
  double m[4];     // row `pos` of the quad's 4x4 Hessian weight M (phi_vv), columns q = 0..3
  double lam_new;  // dual update candidate
};

// AL terms of one constraint row (oracle con_cost / cost_expansion / dual update).
//   EQ:   phi = lam v + 1/2 mu v^2
//   INEQ: phi = lam v + 1/2 mu v^2 [v >= 0 or lam > 0]
//   SOC:  phi = (1/2mu)(|Proj(lam - mu v)|^2 - |lam|^2);  d phi/dv = -Proj(lb),  phi_vv = mu J_Proj(lb)
//         (with the projection-curvature term; Gauss-Newton only: mu J_Proj^2 -- same block shape)
// The quad's 4 values are gathered with DPP, so this must run with EXEC all ones.
template <bool HESS>
__device__ __forceinline__ ConeEval cone_eval(double v, double lam, double mu, const ConMeta& cm, bool act,
                                              double dual_max, bool second_order) {
  ConeEval e;
  const bool is_soc = cm.type == CT_SOC;
  // a linear row may sit in a spare lane of a cone's quad: it never takes part in the cone
  const bool row_on = act & (is_soc ? (cm.pos < cm.p) : (cm.type != CT_NONE));
  // ---- linear rows
  const bool a_in = (cm.type == CT_EQ) | (v >= 0.0) | (lam > 0.0);
  const double lin_g = lam + (a_in ? mu * v : 0.0);
  const double lin_m = a_in ? mu : 0.0;
  const double lin_cost = lam * v + (a_in ? 0.5 * mu * v * v : 0.0);
  const double lin_viol = (cm.type == CT_EQ) ? fabs(v) : fmax(v, 0.0);
  const double lin_lo = (cm.type == CT_EQ) ? -dual_max : 0.0;
  const double lin_new = fmin(fmax(lam + mu * v, lin_lo), dual_max);
  // ---- second-order cone: gather the quad
  const double lb = (row_on & is_soc) ? (lam - mu * v) : 0.0;
  const double vv = (row_on & is_soc) ? v : 0.0;
  double lq[4], vq[4];
  lq[0] = quad_bcast<0>(lb); lq[1] = quad_bcast<1>(lb); lq[2] = quad_bcast<2>(lb); lq[3] = quad_bcast<3>(lb);
  vq[0] = quad_bcast<0>(vv); vq[1] = quad_bcast<1>(vv); vq[2] = quad_bcast<2>(vv); vq[3] = quad_bcast<3>(vv);
  const int pt = cm.p - 1;  // index of the t row
  double n2 = 0.0, n2v = 0.0, t = 0.0, tv = 0.0;
  sfor<0, 4>([&](auto q) {
    constexpr int Q = decltype(q)::value;
    n2 += (Q < pt) ? lq[Q] * lq[Q] : 0.0;
    n2v += (Q < pt) ? vq[Q] * vq[Q] : 0.0;
    t = (Q == pt) ? lq[Q] : t;
    tv = (Q == pt) ? vq[Q] : tv;
  });
  const double nv = sqrt(n2), nvv = sqrt(n2v);
  const bool inside = nv <= t, polar = (!inside) & (nv <= -t);
  const bool bnd = !(inside | polar);
  const double nvs = bnd ? nv : 1.0;
  const double rn = 1.0 / nvs;
  const double c = 0.5 * (1.0 + t * rn);
  const bool is_t = cm.pos == pt, is_s = cm.pos < pt;
  const double pb = is_t ? c * nv : c * lb;               // boundary projection of this row
  const double proj = inside ? lb : (polar ? 0.0 : pb);
  const double soc_cost = (proj * proj - lam * lam) / (2.0 * mu);
  // violation: |Proj(v) - v| of this row
  const bool inside_v = nvv <= tv, polar_v = (!inside_v) & (nvv <= -tv);
  const double nvvs = (inside_v | polar_v) ? 1.0 : nvv;
  const double cv = 0.5 * (1.0 + tv / nvvs);
  const double pv = inside_v ? vv : (polar_v ? 0.0 : (is_t ? cv * nvv : cv * vv));
  const double soc_viol = fabs(pv - vv);
  e.cost = row_on ? (is_soc ? soc_cost : lin_cost) : 0.0;
  e.viol = row_on ? (is_soc ? soc_viol : lin_viol) : 0.0;
  e.g = row_on ? (is_soc ? -proj : lin_g) : 0.0;
  e.lam_new = is_soc ? proj : lin_new;
  if constexpr (HESS) {
    // row of M: linear rows diag; SOC boundary: [[a I + b s s', s/(2nv)], [s'/(2nv), 1/2]] * mu
    //   with curvature term (a, b) = (c, -t/(2 nv^3)); Gauss-Newton (a, b) = (c^2, (nv^2 - 2 t nv - t^2)/(4 nv^4))
    const double rn2 = rn * rn;
    const double a_ = second_order ? c : c * c;
    const double b_ = second_order ? (-0.5 * t * rn * rn2) : (0.25 * (n2 - 2.0 * t * nv - t * t) * rn2 * rn2);
    sfor<0, 4>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      const bool qs = Q < pt, qt = Q == pt;
      const double dlt = (cm.pos == Q) ? 1.0 : 0.0;
      double mb = 0.0;  // boundary case entry J[pos][Q] (or J^2)
      mb = (is_s & qs) ? (a_ * dlt + b_ * lb * lq[Q]) : mb;
      mb = (is_s & qt) ? (0.5 * lb * rn) : mb;
      mb = (is_t & qs) ? (0.5 * lq[Q] * rn) : mb;
      mb = (is_t & qt) ? 0.5 : mb;
      const double ms = inside ? dlt : (polar ? 0.0 : mb);
      const double ml = dlt * lin_m;
      e.m[Q] = row_on ? ((is_soc ? mu * ms : ml)) : 0.0;
    });
  }
  return e;
}

// Row state: everything the per-instance control flow of solve!() carries between phases.
// One per row, in LDS (all 16 lanes of a row read the same words: LDS broadcast).
enum { PH_STEP_BEGIN = 0, PH_OUTER_BEGIN = 1, PH_ITER = 2, PH_DONE = 3 };
struct RowState {
  double J, cmax, J_prev, rho, drho, mu, dV1, dV2, cost_tol, grad_tol;
  int phase, status, iters, iters_outer, outer, it, dj_zero, cur, kref, step, shift, last;
  int nbw, nro, nsolve, nit, nok, ntr;
  double kmu;     // penalty of the backward pass (rho == 0, every pivot positive) that left the gains in KD; < 0: none
  int qvalid;     // Qz (and the hash in qhash) describe the CURRENT trajectory: its last step was an accepted alpha = 1 rollout
  int gconf;      // the last iteration of the last solve was confirmed by the costate sweep (its d is exactly 0)
  int ngc;        // iterations confirmed by the costate sweep (work counter)
  int nfo;        // iterations that took their gains from memory (work counter)
};

template <int NX, int NU, bool CONES>
struct Solver {
  static constexpr int NZ = NX + NU;
  static_assert(NZ <= LW, "packed kernel needs n + m <= 16");

  const SolveParams& P;
  RowState* rs;  // this lane's row state (LDS)
  double* sm;    // this row's 16 x 17 transpose tile (LDS)
  bool hard_wave = false;  // wave-uniform: a row of this wave is in a hard solve (see ALTRO_PRIO_HARD)
  bool heavy_half = false; // wave-uniform: upper half of a grouped launch (see ALTRO_PRIO_HEAVY)
  unsigned* simd_row = nullptr;  // this wave's SIMD in P.simd_tab (16 wave slots), or null
  unsigned wave_slot = 0;
  int turns = 0;           // turns of the wave loop so far
  int n_lone = 0;          // backward passes of this launch that ran as backward_lone (diagnostic, wave_cycles[7])
  ASet* ah;    // this lane's active set of the backward pass that left the gains in KD (LDS; kept in P.ahash between launches)
  ASet* qhs;   // this lane's active set of the trajectory in Qz (LDS)
  ASet* atr;   // this lane's trash set: where a row that sits a phase out accumulates
  int lane, j, inst;
  bool is_x, is_u;
  unsigned rowoff;   // inst*16 + j          (element offsets are 32-bit: the host checks
                     //                       that every array stays below 2^32 bytes)
  unsigned lslot;    // this lane's slot in the compact dual rows (0 for unbounded lanes: dummy)
  bool bounded;
  ALTRO_STAMP(long long t_bw; long long t_rc; long long t_ro; long long t_td; long long t_du; long long t_ls; long long t_fo; long long t_aj; long long t_bl;
              long long c_bw; long long c_fo; long long c_aj; long long c_rc; long long c_ls; long long t_start;)

  struct LaneConst {
    double wd, wf, zmin, zmax;
    bool has_hi, has_lo;
  };

  __device__ Solver(const SolveParams& p, RowState* rows, double* tiles, ASet* hashes) : P(p) {
    lane = threadIdx.x & 63;
    ah = hashes + (threadIdx.x & 63);
    qhs = hashes + 64 + (threadIdx.x & 63);
    atr = hashes + 128 + (threadIdx.x & 63);
    j = lane & 15;
    inst = blockIdx.x * IPW + (lane >> 4);
    if (P.perm != nullptr) inst = P.perm[inst];
    heavy_half = (P.perm != nullptr) && (2u * blockIdx.x >= gridDim.x);
    if (P.simd_tab != nullptr) {
      // HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0
      const unsigned a = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
      const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 0xf;
      const unsigned key = (x << 12) | (((a >> 13) & 7) << 9) | (((a >> 12) & 1) << 8) | (((a >> 8) & 15) << 2) | ((a >> 4) & 3);
      simd_row = P.simd_tab + (size_t)key * 16;
      wave_slot = a & 15;
    }
    rs = rows + (lane >> 4);
    sm = tiles + (lane >> 4) * (LW * (LW + 1));
    is_x = j < NX;
    is_u = (j >= NX) && (j < NZ);
    rowoff = (unsigned)inst * LW + j;
    {
      const int sl = P.bslot[j];
      bounded = sl >= 0;
      lslot = bounded ? (unsigned)sl : 0u;
    }
    ALTRO_STAMP(t_bw = t_rc = t_ro = t_td = t_du = t_ls = t_fo = t_aj = t_bl = c_bw = c_fo = c_aj = c_rc = c_ls = 0;)
  }

  // ---- generic constraint rows (CONES): this lane owns constraint row j of every knot
  struct ConK {  // this lane's row of the knot's constraint table
    ConMeta cm;
    double arow[NZ];
    double brow;
  };
  __device__ __forceinline__ ConMeta con_meta(int k) const {
    ConMeta cm;
    const unsigned b = ((unsigned)k * LW + j) * 4;
    cm.type = P.cmeta[b + 0];
    cm.k0 = P.cmeta[b + 1];
    cm.k1 = P.cmeta[b + 2];
    cm.p = P.cmeta[b + 3];
    cm.pos = j & 3;
    return cm;
  }
  // row j of the knot's A (coefficients of z_0..z_NZ-1) for value = A z + b
  __device__ __forceinline__ void con_load(int k, ConK& c) const {
    c.cm = con_meta(k);
    const unsigned b = (unsigned)inst * P.con_istride + ((unsigned)k * LW + j) * LW;
    sfor<0, NZ>([&](auto q) { c.arow[decltype(q)::value] = ldg(P.Acon, b + decltype(q)::value); });
    c.brow = ldg(P.bcon, (unsigned)inst * (P.con_istride / LW) + (unsigned)k * LW + j);
  }
  // column j of the knot's A (coefficient of z_j in every constraint row)
  __device__ __forceinline__ void con_col(int k, double (&acol)[16]) const {
    const unsigned b = (unsigned)inst * P.con_istride + (unsigned)k * LW * LW + j;
    sfor<0, 16>([&](auto r) { acol[decltype(r)::value] = ldg(P.Acon, b + decltype(r)::value * LW); });
  }
  // constraint values of this lane's row for the knot vector z (one element per lane)
  __device__ __forceinline__ double con_value(double z, const double (&arow)[NZ], double brow) const {
    double acc4[4] = {brow, 0.0, 0.0, 0.0};
    Blk<NX, NU>::GZ(acc4, z, arow);
    return (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  }
  static __device__ __forceinline__ bool con_act(const ConMeta& cm, int k) {
    return (cm.type != CT_NONE) & (k >= cm.k0) & (k <= cm.k1);
  }

  __device__ __forceinline__ LaneConst consts() const {
    LaneConst c;
    c.wd = P.wd[j];
    c.wf = P.wf[j];
    c.zmin = P.zmin[j];
    c.zmax = P.zmax[j];
    c.has_hi = c.zmax < 1e300;
    c.has_lo = c.zmin > -1e300;
    return c;
  }

  // Everything lane-dependent a phase needs (a few dozen addresses: the Gcol columns, the dual rows, the gain rows) is
  // loop-invariant for the whole kernel; LLVM hoists all of it to the kernel's entry, where it does not fit the 256
  // registers, and every phase then opened with a chain of `scratch_load address; s_waitcnt vmcnt(0); global_load`
  // -- two dozen memory round trips in series before a rollout's first knot.  Making the lane's indices opaque at the
  // start of a phase keeps those computations (a handful of integer instructions) inside the phase.
  __device__ __forceinline__ void phase_begin() { asm volatile("" : "+v"(j), "+v"(rowoff), "+v"(lslot), "+v"(inst)); }

  // "these loads have landed": an empty asm that reads the registers, so that hipcc waits for them HERE (see backward())
  static __device__ __forceinline__ void landed_in(double& a) { asm volatile("" : "+v"(a)); }
  template <int M, int I = 0>
  static __device__ __forceinline__ void landed(double (&a)[M]) {
    if constexpr (I < M) {
      asm volatile("" : "+v"(a[I]));
      landed<M, I + 1>(a);
    }
  }

  // INSTANCE-MAJOR arrays (DESIGN.md "Data layout"): everything an instance owns in one array is contiguous, row after
  // row of 16 lanes, so a sweep over the knots walks one 128-byte line after the other (one DRAM page, one TLB entry per
  // array) and a row index becomes an address with a shift -- with the knots outermost every knot of every array sat in
  // its own megabyte and a per-lane row index cost a quarter-rate v_mul_lo_u32.
  //   Z    [Bp][2 N + 1][16]  rows 0..N-1 plane 0, N..2N-1 plane 1, row 2 N the trash row
  //   Zref [Bp][Nt][16]
  //   Lc, Qz, Dff [Bp][N + 1][16]  (row N = trash)
  __device__ __forceinline__ unsigned zat(int r) const { return ((unsigned)inst * (2u * (unsigned)P.N + 1u) + (unsigned)r) * LW + j; }
  __device__ __forceinline__ unsigned rat(int k) const { return ((unsigned)inst * (unsigned)P.Nt + (unsigned)k) * LW + j; }
  __device__ __forceinline__ unsigned qat(int k) const { return ((unsigned)inst * ((unsigned)P.N + 1u) + (unsigned)k) * LW + j; }
  // plane c of Z as an element offset inside the instance's block (per instance: cur differs between rows)
  __device__ __forceinline__ unsigned plane(int c) const { return (unsigned)c * (unsigned)P.N * LW; }
  __device__ __forceinline__ unsigned trash_z() const { return zat(2 * P.N); }
  // compact dual rows Lb [Bp][N + 1][2][nbp]: element offset of (knot k, side, this lane's slot); knot N is the trash row
  __device__ __forceinline__ unsigned lb_at(int k, int side) const {
    return (((unsigned)inst * ((unsigned)P.N + 1u) + (unsigned)k) * 2u + (unsigned)side) * (unsigned)P.nbp + lslot;
  }
  __device__ __forceinline__ unsigned trash_l(int side) const { return lb_at(P.N, side); }
  __device__ __forceinline__ bool box_at(int k) const { return k >= P.box_k0 && k <= P.box_k1; }
  __device__ __forceinline__ unsigned kd_at(int k, int row) const {  // KD [Bp][N][NU][16]
    return (((unsigned)inst * (unsigned)P.N + (unsigned)k) * NU + row) * LW + j;
  }
  static __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
  static __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

  // Who of the waves on this SIMD has more work left.  The launch ends with its slowest wave, and while two waves share a
  // SIMD the VALU is issue-bound: the one that issues first runs at the speed it would have alone, the other
  // gets the slots that are left.  Age decides between equal priorities, and age knows nothing of the work: so every turn a
  // wave publishes an estimate of what it has left -- the steps to go at the turns per step it needed so far -- in the slot of
  // its SIMD (HW_ID), reads its mates' and takes the high priorities if none of them has more.
  __device__ __forceinline__ bool mate_rank(int steps_left, int steps_done) const {
    // (measured over twelve 20-step windows, two boxes: 9.27 -> 9.00 ms and 9.29 -> 9.03 ms against the row's own state alone;
    //  the steps left alone as the estimate, and the priorities 0/0 or 0/1 against 3/3 instead of 0/1 against 2/3: the same)
    const unsigned mine = 1u + (unsigned)(steps_left * (turns + 1) * 16) / (unsigned)(steps_done + 1);
    if (lane == 0) __hip_atomic_store(simd_row + wave_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned other = 0u;
    if (lane < 16 && (unsigned)lane != wave_slot) other = __hip_atomic_load(simd_row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return !wave_any(other > mine);
  }
  __device__ __forceinline__ void mate_leave() const {
    if (simd_row != nullptr && lane == 0) __hip_atomic_store(simd_row + wave_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  __device__ __forceinline__ void prio_serial() const {
    if (ALTRO_PRIO_HARD > 0 && hard_wave) __builtin_amdgcn_s_setprio(ALTRO_PRIO_HARD < 3 ? ALTRO_PRIO_HARD + 1 : 3);
    else if (ALTRO_PRIO_HEAVY > 0 && heavy_half) __builtin_amdgcn_s_setprio(ALTRO_PRIO_HEAVY);
    else __builtin_amdgcn_s_setprio(ALTRO_PRIO_SERIAL);
  }
  __device__ __forceinline__ void prio_base() const {
    if (ALTRO_PRIO_HARD > 0 && hard_wave) __builtin_amdgcn_s_setprio(ALTRO_PRIO_HARD);
    else if (ALTRO_PRIO_HEAVY > 0 && heavy_half) __builtin_amdgcn_s_setprio(ALTRO_PRIO_HEAVY);
    else __builtin_amdgcn_s_setprio(0);
  }

  // The active set of one lane is built knot by knot, in whatever order a sweep walks them (ASet above): OR is order-free,
  // so the backward pass (descending), the rollouts (ascending) and the lone rollouts (four rows at once, each its own
  // knots, into the SAME set: hence the atomic) arrive at the same words.
  static __device__ __forceinline__ void aset_clear(ASet* t) {
    sfor<0, ASET_WORDS>([&](auto q) { t->w[decltype(q)::value] = 0u; });
  }
  static __device__ __forceinline__ void aset_add(ASet* t, unsigned code, int k) {
    atomicOr(&t->w[imin(k >> 4, ASET_WORDS - 1)], code << ((k & 15) * 2));
  }
  static __device__ __forceinline__ unsigned aset_get(const ASet* t, int k) {
    return (t->w[imin(k >> 4, ASET_WORDS - 1)] >> ((k & 15) * 2)) & 3u;
  }
  static __device__ __forceinline__ void aset_copy(ASet* d, const ASet* s_) {
    sfor<0, ASET_WORDS>([&](auto q) { d->w[decltype(q)::value] = s_->w[decltype(q)::value]; });
  }
  // Hessian diagonal of the lane's cost + box terms from the two bits of its active set: w + mu [upper side] + mu [lower side]
  // (the sum box_expand forms, in its order)
  static __device__ __forceinline__ double hz_of(double w, double mu, unsigned bits) {
    double hz = w;
    hz += (bits & 1u) ? mu : 0.0;
    hz += (bits & 2u) ? mu : 0.0;
    return hz;
  }
  static __device__ __forceinline__ bool aset_ne(const ASet* a, const ASet* b) {
    unsigned d = 0u;
    sfor<0, ASET_WORDS>([&](auto q) { d |= a->w[decltype(q)::value] ^ b->w[decltype(q)::value]; });
    return d != 0u;
  }

  // max over the NU control lanes of this row
  __device__ __forceinline__ double umax(double v) const {
    double m = bcast<NX>(v);
    sfor<1, NU>([&](auto a) { m = fmax(m, bcast<NX + decltype(a)::value>(v)); });
    return m;
  }

  // stage cost + box AL term of this lane's element; updates the lane's violation maximum.
  // Branch-free: box_on only selects.  (oracle total_cost / con_cost; TO.jl cost! -- SURVEY A.2)
  static __device__ __forceinline__ double lane_cost(const LaneConst& c, double mu, double z, double zr, double w,
                                                     double lhi, double llo, bool box_on, double& viol) {
    double qz;
    unsigned code;
    return lane_cost_grad<false>(c, mu, z, zr, w, lhi, llo, box_on, viol, qz, code);
  }
  // GRAD: also the gradient of the lane's AL cost term (what box_expand computes) and the 2-bit active-set code
  template <bool GRAD>
  static __device__ __forceinline__ double lane_cost_grad(const LaneConst& c, double mu, double z, double zr, double w,
                                                          double lhi, double llo, bool box_on, double& viol, double& qz,
                                                          unsigned& code) {
    // A side that is absent (no bound on this element, or a knot outside the box range) is switched off at its INPUTS --
    // value -1 (inside), dual 0 -- instead of at its four outputs: the side is then inactive, its cost and gradient terms
    // are exact zeros and its violation candidate is negative.  Three selects per side instead of five (the duals may
    // come in unselected); the arithmetic of a side that is present is unchanged.
    const double e = z - zr;
    double Jl = 0.5 * w * e * e;
    const bool bh = box_on & c.has_hi, bl = box_on & c.has_lo;
    const double chi = bh ? z - c.zmax : -1.0, clo = bl ? c.zmin - z : -1.0;
    const double lh = bh ? lhi : 0.0, ll = bl ? llo : 0.0;
    const bool ahi = (chi >= 0.0) | (lh > 0.0);
    const bool alo = (clo >= 0.0) | (ll > 0.0);
    const double phi = ahi ? mu * chi : 0.0, plo = alo ? mu * clo : 0.0;
    Jl += lh * chi + 0.5 * phi * chi;
    Jl += ll * clo + 0.5 * plo * clo;
    viol = fmax(viol, chi);   // viol >= 0 on entry
    viol = fmax(viol, clo);
    if constexpr (GRAD) {
      qz = w * e;
      qz += lh + phi;
      qz -= ll + plo;
      code = (ahi ? 1u : 0u) | (alo ? 2u : 0u);
    }
    return Jl;
  }

  struct RollOut {
    double J, cmax;
    bool limit;
    bool unchanged;  // the trial reproduced plane `cur` bit for bit (closed-loop rollouts only)
    bool tiny;       // every element moved by at most 1e-7 (1 + |z|)            (closed-loop rollouts only)
  };

  struct KnotIn {
    double z, zr, lhi, llo;
    double kcol[NU <= 4 ? 4 : NU];  // closed loop: x lane j holds K[:, j] (slot order: see rollout)
    double dff;                     // closed loop: u lane NX + a holds d[a]
    double lc, lcn;   // dual of this lane's constraint row at the knot and at the next one (CONES)
  };

  // rollout!(solver[, alpha]) fused with cost!(obj, Z̄) and max_violation.
  //   OPEN : open-loop rollout of plane `cur` from x0 (iLQR initialize!), in place.  Rows with
  //          `shift` read the controls and box duals one knot ahead and write them back in place:
  //          RD.shift_fill!(Z) + Altro.shift_fill!(conSet) folded into the same sweep
  //          (random_linear_problem.jl:136,139).  Rows with !take (they are in the middle of an
  //          inner loop) send their stores to the trash slot.
  //   !OPEN: closed-loop rollout with gains KD and step alpha = 1 from plane cur into plane
  //          cur^1 (dead storage for every row that is not iterating, so stores need no mask).
  //          storeq: rows whose gradient plane Qz this rollout refreshes (the rows that are searching).
  template <bool OPEN>
  __device__ RollOut rollout(bool take, bool shift, bool storeq = false) {
    phase_begin();
    prio_serial();  // latency-bound phase: see ALTRO_PRIO_SERIAL
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const int cur = rs->cur, kref = rs->kref;
    const unsigned zs = plane(cur);
    const unsigned zd = plane(OPEN ? cur : cur ^ 1);  // (closed loop: the plane the trial is written to)
    // Stores of rows that sit a phase out go to trash rows.  The select is made on the KNOT INDEX (one
    // v_cndmask); selecting between two computed addresses made hipcc emit divergent branches with
    // scratch reloads and s_waitcnt vmcnt(0) inside the loop, which serialises the prefetch ring.
    const bool wl = OPEN && shift && take && bounded;
    const int N = P.N;
    double grow[NZ];
    sfor<0, NZ>([&](auto c) {
      constexpr int C = decltype(c)::value;
      grow[C] = ldg(P.Grow, ((unsigned)inst * LW + C) * LW + j);
    });
    const double fv = ldg(P.fvec, rowoff);
    double xb = ldg(P.x0, rowoff);
    // The stage costs are summed in FOUR classes of knots (k mod 4), each in ascending order, and combined as
    // (J0 + J1) + (J2 + J3) + terminal: rollout_lone(), where DPP row r evaluates exactly class r, gives the same bits.
    double Jcls[4] = {0.0, 0.0, 0.0, 0.0}, Jterm = 0.0, viol = 0.0;
    bool limit = false, changed = false, big = false;
    ASet* const tq = storeq ? qhs : atr;      // the active set at the trajectory produced: rows that take part own it afterwards
    if constexpr (!CONES) aset_clear(tq);
    const int k1 = P.box_k1;
    const bool shl = OPEN && shift;           // per row
    const bool shu = shl && !is_x;            // controls are read one knot ahead
    const bool wr_l = OPEN && shift && take;  // shifted duals are written back
    const double dmax = P.o.dual_max;
    const bool so2 = P.o.soc_second_order != 0;
    // Pinned in a register: hipcc otherwise re-loads this select-of-two-kernel-arguments inside the
    // knot loop (a rematerialised global_load), and the in-order vmcnt wait on that load drains the
    // whole prefetch ring once or twice per group of knots.
    double lim = is_x ? P.o.max_state_value : P.o.max_control_value;
    asm volatile("" : "+v"(lim));

    // NU <= 4: slot s of lane l holds gain row A = ((s ^ (l & 3)) - NX) & 3, the order in which the
    // xor butterfly of `stage` leaves the sum of row A in lane NX + A (and d[A] in that lane's slot 0)
    unsigned kofs[4] = {0u, 0u, 0u, 0u};
    bool kval[4] = {false, false, false, false};
    if constexpr (!OPEN && NU <= 4) {
      sfor<0, 4>([&](auto c) {
        constexpr int S = decltype(c)::value;
        const int A = ((S ^ (j & 3)) - NX) & 3;
        kval[S] = is_x & (A < NU);
        kofs[S] = (unsigned)imin(A, NU - 1) * LW;
      });
    }

    // operands of stage knot k (k clamped to 0..N-2 by the callers)
    auto load = [&](int k, KnotIn& in, ConK& ck) {
      const int ku = imin(k + 1, N - 2);
      const int kl = imax(imin(k + 1, k1), 0);
      in.z = ldg(P.Z, zs + zat(shu ? ku : k));
      in.zr = ldg(P.Zref, rat(kref + k));
      const int kk = shl ? kl : k;
      in.lhi = ldg(P.Lb, lb_at(kk, 0));
      in.llo = ldg(P.Lb, lb_at(kk, 1));
      if constexpr (!OPEN) {
        if constexpr (NU <= 4) {
          sfor<0, 4>([&](auto c) { in.kcol[decltype(c)::value] = ldg(P.KD, kd_at(k, 0) + kofs[decltype(c)::value]); });
        } else {
          sfor<0, NU>([&](auto c) { in.kcol[decltype(c)::value] = ldg(P.KD, kd_at(k, decltype(c)::value)); });
        }
        in.dff = ldg(P.Dff, qat(k));
      }
      in.lc = 0.0;
      in.lcn = 0.0;
      if constexpr (CONES) {
        con_load(k, ck);
        // the shifted read (knot k + 1 while that is inside the row's own range) is chosen in `stage`:
        // an address that depends on the row's metadata would make this a dependent load, and the
        // in-order vmcnt wait in front of it drains the whole prefetch ring
        in.lc = ldg(P.Lc, qat(k));
        if constexpr (OPEN) in.lcn = ldg(P.Lc, qat(imin(k + 1, N - 1)));
      }
    };

    auto stage = [&](auto cls_, int k, const KnotIn& in, const ConK& ck) {
      // the callers pass the knot's position in its group of PD = 4 or 8: k mod 4 (conic kernels: one sum, as before)
      double& Jacc = Jcls[CONES ? 0 : (decltype(cls_)::value & 3)];
      const bool bx = box_at(k);
      const double lhi = bx ? in.lhi : 0.0, llo = bx ? in.llo : 0.0;
      double zb;
      if constexpr (OPEN) {
        zb = is_x ? xb : in.z;
        stg(P.Z, zat(take ? cur * N + k : 2 * N), zb);
        const int kl_ = wl ? k : P.N;  // knot N of Lb is the trash row
        stg(P.Lb, lb_at(kl_, 0), lhi);
        stg(P.Lb, lb_at(kl_, 1), llo);
      } else {
        // du = K dx: x lane j contributes K[:, j] dx_j; the NX-lane sums run as DPP FMAs
        const double dx = is_x ? (xb - in.z) : 0.0;
        double du;
        const double dff = in.dff;
        if constexpr (NU <= 4) {
          // four row sums over the x lanes in 15 VALU: lanes trade two slots with lane^1, one with
          // lane^2 (afterwards every lane of a quad holds the quad's part of row (l&3)-NX), then the
          // four quads are added by two rotations.  Lane NX + A ends with sum_j K[A][j] dx_j.
          double p[4];
          if constexpr (NU == 4 && NZ == LW) {
            // every slot is a gain row and every lane an element: dx is already 0 on the control lanes, whose slots hold
            // the (finite) factors of Quu, so the products there are exact zeros without a select
            sfor<0, 4>([&](auto c) { p[decltype(c)::value] = in.kcol[decltype(c)::value] * dx; });
          } else {
            sfor<0, 4>([&](auto c) { p[decltype(c)::value] = kval[decltype(c)::value] ? in.kcol[decltype(c)::value] * dx : 0.0; });
          }
          const double n0 = p[0] + dpp_mov<DPP_XOR1>(p[1]);
          const double n2 = p[2] + dpp_mov<DPP_XOR1>(p[3]);
          double q = n0 + dpp_mov<DPP_XOR2>(n2);
          q += dpp_mov<DPP_ROR8>(q);
          q += dpp_mov<DPP_ROR4>(q);
          du = q;
        } else {
          double prod[NU], acc[NU][3];
          sfor<0, NU>([&](auto a) {
            constexpr int A = decltype(a)::value;
            prod[A] = is_x ? in.kcol[A] * dx : 0.0;
            acc[A][0] = acc[A][1] = acc[A][2] = 0.0;
          });
          const double one = 1.0;
          Blk<NX, NU>::KDXT(acc, prod, one);
          du = (acc[0][0] + acc[0][1]) + acc[0][2];
          sfor<1, NU>([&](auto a) {
            constexpr int A = decltype(a)::value;
            const double da = (acc[A][0] + acc[A][1]) + acc[A][2];
            du = (j == NX + A) ? da : du;
          });
        }
        const double ub = in.z + du + dff;  // alpha = 1
        zb = is_x ? xb : ub;
        changed = changed | ((is_x | is_u) & (zb != in.z));
        big = big | ((is_x | is_u) & !(fabs(zb - in.z) <= 1e-7 * (1.0 + fabs(in.z))));
        stg(P.Z, zd + zat(k), zb);
      }
      if constexpr (!CONES) {
        double qz;
        unsigned code;
        Jacc += lane_cost_grad<true>(lc, mu, zb, in.zr, lc.wd, OPEN ? lhi : in.lhi, OPEN ? llo : in.llo, bx, viol, qz, code);
        aset_add(tq, code, k);
        stg(P.Qz, qat(storeq ? k : N), qz);
      } else {
        Jacc += lane_cost(lc, mu, zb, in.zr, lc.wd, lhi, llo, bx, viol);
      }
      if constexpr (CONES) {
        const bool act = con_act(ck.cm, k);
        const double v = con_value(zb, ck.arow, ck.brow);
        double lck = in.lc;
        if constexpr (OPEN) lck = (shl & (k + 1 <= ck.cm.k1)) ? in.lcn : in.lc;
        const ConeEval e = cone_eval<false>(v, act ? lck : 0.0, mu, ck.cm, act, dmax, so2);
        Jacc += e.cost;
        viol = fmax(viol, e.viol);
        if constexpr (OPEN) stg(P.Lc, qat((wr_l & act) ? k : P.N), lck);
      }
      limit = limit | ((is_x | is_u) & !(fabs(zb) <= lim));
      double acc4[4] = {fv, 0.0, 0.0, 0.0};
      Blk<NX, NU>::GZ(acc4, zb, grow);
      xb = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    };

    // terminal-knot operands (independent of the pipeline)
    const int kt = N - 1;
    const double t_zr = ldg(P.Zref, rat(kref + kt));
    const double t_lhi = ldg(P.Lb, lb_at(kt, 0)), t_llo = ldg(P.Lb, lb_at(kt, 1));
    double t_z = 0.0, t_lc = 0.0;
    ConK t_ck;
    if constexpr (!OPEN) t_z = ldg(P.Z, zs + zat(kt));
    if constexpr (CONES) {
      t_lc = ldg(P.Lc, qat(kt));
      con_load(kt, t_ck);
    }

    // software pipeline: operands of knot k+PD are requested right after knot k is consumed.
    // The main loop body is one basic block (PD stages, PD clamped loads).
    constexpr int PD = CONES ? 2 : (OPEN ? ALTRO_PD_OPEN : ALTRO_PD_CLOSED);  // cone tables cost registers
    KnotIn ring[PD];
    ConK cring[PD];
    sfor<0, PD>([&](auto u) {
      constexpr int U = decltype(u)::value;
      load(imin(U, N - 2), ring[U], cring[U]);
    });
    // The first fill of the ring is waited for here, once (see backward(): with these loads pending on the way into the
    // loop, hipcc's merged wait-count state asked every body for `vmcnt(PD * loads - 1)` at its first use -- right for the
    // first trip, half the ring's lead time on every later one, where the stores of the previous body are in flight too).
    sfor<0, PD>([&](auto u) {
      constexpr int U = decltype(u)::value;
      asm volatile("" : "+v"(ring[U].z), "+v"(ring[U].zr), "+v"(ring[U].lhi), "+v"(ring[U].llo));
      if constexpr (!OPEN) {
        landed(ring[U].kcol);
        asm volatile("" : "+v"(ring[U].dff));
      }
    });
    const int ngroups = (N - 1) / PD;
    int k = 0;
    for (int g = 0; g < ngroups; ++g, k += PD) {
      sfor<0, PD>([&](auto u) {
        constexpr int U = decltype(u)::value;
        stage(u, k + U, ring[U], cring[U]);
        load(imin(k + U + PD, N - 2), ring[U], cring[U]);
      });
    }
    {  // remaining (N-1) % PD stage knots
      const int rem = (N - 1) - k;
      sfor<0, PD>([&](auto u) {
        constexpr int U = decltype(u)::value;
        if (U < rem) stage(u, k + U, ring[U], cring[U]);
      });
    }
    {  // terminal knot: state only
      const bool bx = box_at(kt);
      const double zb = is_x ? xb : 0.0;
      if constexpr (OPEN) stg(P.Z, zat(take ? cur * N + kt : 2 * N), zb);
      else stg(P.Z, zd + zat(kt), zb);
      if constexpr (!CONES) {
        double qz;
        unsigned code;
        Jterm += lane_cost_grad<true>(lc, mu, zb, t_zr, lc.wf, bx ? t_lhi : 0.0, bx ? t_llo : 0.0, bx & is_x, viol, qz, code);
        aset_add(tq, code, kt);
        stg(P.Qz, qat(storeq ? kt : N), is_x ? qz : 0.0);
      } else {
        Jcls[0] += lane_cost(lc, mu, zb, t_zr, lc.wf, bx ? t_lhi : 0.0, bx ? t_llo : 0.0, bx & is_x, viol);
      }
      if constexpr (CONES) {
        const bool act = con_act(t_ck.cm, kt);
        const double v = con_value(zb, t_ck.arow, t_ck.brow);
        const ConeEval e = cone_eval<false>(v, act ? t_lc : 0.0, mu, t_ck.cm, act, dmax, so2);
        Jcls[0] += e.cost;
        viol = fmax(viol, e.viol);
      }
      limit = limit | (is_x & !(fabs(zb) <= P.o.max_state_value));
      if constexpr (!OPEN) {
        changed = changed | (is_x & (zb != t_z));
        big = big | (is_x & !(fabs(zb - t_z) <= 1e-7 * (1.0 + fabs(t_z))));
      }
    }
    RollOut r;
    if constexpr (CONES) r.J = row_sum(Jcls[0]);
    else r.J = ((row_sum(Jcls[0]) + row_sum(Jcls[1])) + (row_sum(Jcls[2]) + row_sum(Jcls[3]))) + row_sum(Jterm);
    r.cmax = row_max(viol);
    r.limit = row_any(limit, lane);
    r.unchanged = !row_any(changed, lane);
    r.tiny = !row_any(big, lane);
    prio_base();
    return r;
  }

  // Lone rollouts (box-only kernels).  When ONE row of the wave needs a rollout -- the tail of a launch, where single hard
  // instances walk their chains alone -- the other three rows used to execute it as shadows.  A rollout knot is ~45
  // instructions of recurrence (z_k -> x_{k+1}) and ~110 that hang off it (cost, gradient, violation, active-set hash,
  // stores): here all four rows run the recurrence of a group of four knots (they are clones of the lone row: same
  // operands, same results), then DPP row r does the rest for knot k0 + r alone.  Per knot ~90 (closed loop) and ~55
  // (open loop) instructions instead of 169 and 154.  Row r sums exactly class r of rollout()'s stage costs and the
  // classes are combined in rollout()'s order, so both forms return the same bits (test: ALTRO_NO_LONE=1).
  // The caller has pointed inst / rowoff / rs of all lanes at the lone row (lone_enter); take = storeq = true.
  template <bool OPEN>
  __device__ RollOut rollout_lone() {
    static_assert(!CONES, "box-only kernels");
    phase_begin();
    prio_serial();
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const int cur = rs->cur, kref = rs->kref;
    const bool shift = OPEN && (rs->shift != 0);
    const int rr = lane >> 4;
    const unsigned zs = plane(cur);
    const unsigned zd = OPEN ? plane(cur) : plane(cur ^ 1);
    const int N = P.N;
    double grow[NZ];
    sfor<0, NZ>([&](auto c) {
      constexpr int C = decltype(c)::value;
      grow[C] = ldg(P.Grow, ((unsigned)inst * LW + C) * LW + j);
    });
    const double fv = ldg(P.fvec, rowoff);
    double xb = ldg(P.x0, rowoff);
    double Jc = 0.0, Jterm = 0.0, viol = 0.0;
    bool limit = false, changed = false, big = false;
    ASet* const tq = qhs;                // (lone_enter has pointed it at the lone row's set: the four rows OR their knots into it)
    aset_clear(tq);
    const int k1 = P.box_k1;
    const bool shu = shift && !is_x;     // controls are read one knot ahead
    const bool wl = shift && bounded;    // shifted duals are written back
    double lim = is_x ? P.o.max_state_value : P.o.max_control_value;
    asm volatile("" : "+v"(lim));
    unsigned kofs[4] = {0u, 0u, 0u, 0u};
    bool kval[4] = {false, false, false, false};
    if constexpr (!OPEN && NU <= 4) {  // slot order of the gain rows: see rollout()
      sfor<0, 4>([&](auto c) {
        constexpr int S = decltype(c)::value;
        const int A = ((S ^ (j & 3)) - NX) & 3;
        kval[S] = is_x & (A < NU);
        kofs[S] = (unsigned)imin(A, NU - 1) * LW;
      });
    }
    constexpr int KS = (NU <= 4) ? 4 : NU;
    struct Rec {  // operands of the recurrence of four knots: the same for every row
      double z[4], kcol[OPEN ? 1 : 4][OPEN ? 1 : KS], dff[OPEN ? 1 : 4];
    };
    struct Cst {  // operands of the cost terms of this row's knot of the group
      double zr, lhi, llo;
    };
    auto load_rec = [&](int k0, Rec& r) {
      sfor<0, 4>([&](auto u) {
        constexpr int U = decltype(u)::value;
        const int ku = imin(k0 + U, N - 2);
        r.z[U] = ldg(P.Z, zs + zat(shu ? imin(ku + 1, N - 2) : ku));
        if constexpr (!OPEN) {
          if constexpr (NU <= 4) {
            sfor<0, 4>([&](auto c) { r.kcol[U][decltype(c)::value] = ldg(P.KD, kd_at(ku, 0) + kofs[decltype(c)::value]); });
          } else {
            sfor<0, NU>([&](auto c) { r.kcol[U][decltype(c)::value] = ldg(P.KD, kd_at(ku, decltype(c)::value)); });
          }
          r.dff[U] = ldg(P.Dff, qat(ku));
        }
      });
    };
    auto load_cst = [&](int k0, Cst& c) {
      const int kq = imin(k0 + rr, N - 2);
      c.zr = ldg(P.Zref, rat(kref + kq));
      const int kk = shift ? imax(imin(kq + 1, k1), 0) : kq;
      c.lhi = ldg(P.Lb, lb_at(kk, 0));
      c.llo = ldg(P.Lb, lb_at(kk, 1));
    };
    auto group = [&](int k0, const Rec& r, const Cst& c) {
      double zb[4];
      sfor<0, 4>([&](auto u) {
        constexpr int U = decltype(u)::value;
        const bool valid = k0 + U < N - 1;
        double zbu;
        if constexpr (OPEN) {
          zbu = is_x ? xb : r.z[U];
        } else {
          const double dx = is_x ? (xb - r.z[U]) : 0.0;
          double du;
          if constexpr (NU <= 4) {
            double p[4];
            if constexpr (NU == 4 && NZ == LW) {
              sfor<0, 4>([&](auto cc) { p[decltype(cc)::value] = r.kcol[U][decltype(cc)::value] * dx; });
            } else {
              sfor<0, 4>([&](auto cc) { p[decltype(cc)::value] = kval[decltype(cc)::value] ? r.kcol[U][decltype(cc)::value] * dx : 0.0; });
            }
            const double n0 = p[0] + dpp_mov<DPP_XOR1>(p[1]);
            const double n2 = p[2] + dpp_mov<DPP_XOR1>(p[3]);
            double q = n0 + dpp_mov<DPP_XOR2>(n2);
            q += dpp_mov<DPP_ROR8>(q);
            q += dpp_mov<DPP_ROR4>(q);
            du = q;
          } else {
            double prod[NU], acc[NU][3];
            sfor<0, NU>([&](auto a) {
              constexpr int A = decltype(a)::value;
              prod[A] = is_x ? r.kcol[U][A] * dx : 0.0;
              acc[A][0] = acc[A][1] = acc[A][2] = 0.0;
            });
            const double one = 1.0;
            Blk<NX, NU>::KDXT(acc, prod, one);
            du = (acc[0][0] + acc[0][1]) + acc[0][2];
            sfor<1, NU>([&](auto a) {
              constexpr int A = decltype(a)::value;
              const double da = (acc[A][0] + acc[A][1]) + acc[A][2];
              du = (j == NX + A) ? da : du;
            });
          }
          const double ub = r.z[U] + du + r.dff[U];  // alpha = 1
          zbu = is_x ? xb : ub;
        }
        zb[U] = zbu;
        double acc4[4] = {fv, 0.0, 0.0, 0.0};
        Blk<NX, NU>::GZ(acc4, zbu, grow);
        const double xn = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        xb = valid ? xn : xb;
      });
      // this row's knot of the group
      const int kq = k0 + rr;
      const bool valid = kq < N - 1;
      double zm = zb[0], zo = r.z[0];
      sfor<1, 4>([&](auto u) {
        constexpr int U = decltype(u)::value;
        zm = (rr == U) ? zb[U] : zm;
        zo = (rr == U) ? r.z[U] : zo;
      });
      const bool bx = box_at(kq) & valid;
      const double lhi = bx ? c.lhi : 0.0, llo = bx ? c.llo : 0.0;
      if constexpr (OPEN) {
        stg(P.Z, zat(valid ? cur * N + kq : 2 * N), zm);
        const int kl_ = (wl & valid) ? kq : P.N;  // knot N of Lb is the trash row
        stg(P.Lb, lb_at(kl_, 0), lhi);
        stg(P.Lb, lb_at(kl_, 1), llo);
      } else {
        stg(P.Z, zat(valid ? (cur ^ 1) * N + kq : 2 * N), zm);  // (a select of the knot index, not of two addresses: see rollout())
        changed = changed | (valid & (is_x | is_u) & (zm != zo));
        big = big | (valid & (is_x | is_u) & !(fabs(zm - zo) <= 1e-7 * (1.0 + fabs(zo))));
      }
      double qz;
      unsigned code;
      // an invalid knot (past the last stage knot) contributes exact zeros: weight 0, box off
      Jc += lane_cost_grad<true>(lc, mu, zm, c.zr, valid ? lc.wd : 0.0, lhi, llo, bx, viol, qz, code);
      aset_add(tq, valid ? code : 0u, kq);
      stg(P.Qz, qat(valid ? kq : N), qz);
      limit = limit | (valid & (is_x | is_u) & !(fabs(zm) <= lim));
    };

    // terminal-knot operands (independent of the pipeline)
    const int kt = N - 1;
    const double t_zr = ldg(P.Zref, rat(kref + kt));
    const double t_lhi = ldg(P.Lb, lb_at(kt, 0)), t_llo = ldg(P.Lb, lb_at(kt, 1));
    double t_z = 0.0;
    if constexpr (!OPEN) t_z = ldg(P.Z, zs + zat(kt));

    // two register sets: the operands of the next group are requested before the current one is consumed
    Rec ra, rb;
    Cst ca, cb;
    load_rec(0, ra);
    load_cst(0, ca);
    int k0 = 0;
    while (true) {
      load_rec(k0 + 4, rb);
      load_cst(k0 + 4, cb);
      group(k0, ra, ca);
      k0 += 4;
      if (k0 >= N - 1) break;
      load_rec(k0 + 4, ra);
      load_cst(k0 + 4, ca);
      group(k0, rb, cb);
      k0 += 4;
      if (k0 >= N - 1) break;
    }
    {  // terminal knot: state only (every row, redundantly)
      const bool bx = box_at(kt);
      const double zb = is_x ? xb : 0.0;
      if constexpr (OPEN) stg(P.Z, zat(cur * N + kt), zb);
      else stg(P.Z, zd + zat(kt), zb);
      double qz;
      unsigned code;
      Jterm += lane_cost_grad<true>(lc, mu, zb, t_zr, lc.wf, bx ? t_lhi : 0.0, bx ? t_llo : 0.0, bx & is_x, viol, qz, code);
      aset_add(tq, code, kt);   // (every row, the same bits: idempotent)
      stg(P.Qz, qat(kt), is_x ? qz : 0.0);
      limit = limit | (is_x & !(fabs(zb) <= P.o.max_state_value));
      if constexpr (!OPEN) {
        changed = changed | (is_x & (zb != t_z));
        big = big | (is_x & !(fabs(zb - t_z) <= 1e-7 * (1.0 + fabs(t_z))));
      }
    }
    RollOut r;
    r.J = rows_sum4(row_sum(Jc)) + row_sum(Jterm);
    r.cmax = rows_max4(row_max(viol));
    r.limit = wave_any(limit);
    r.unchanged = !wave_any(changed);
    r.tiny = !wave_any(big);
    prio_base();
    return r;
  }

  // gradient_todorov!: mean_k max_a |d_k,a| / (|u_k,a| + 1) on the current plane.  Evaluated
  // lazily: the reference only uses it in the convergence test, which also needs dJ < tol.
  __device__ double todorov() {
    phase_begin();
    prio_serial();
    const unsigned zs = plane(rs->cur);
    const int N = P.N;
    double acc = 0.0;
    constexpr int UN = ALTRO_UN;
    const int nch = (N - 1 + UN - 1) / UN;
    for (int c = 0; c < nch; ++c) {
      const int k0 = c * UN;
      double u[UN], d[UN];
      sfor<0, UN>([&](auto t) {
        constexpr int Tt = decltype(t)::value;
        const int k = imin(k0 + Tt, N - 2);
        u[Tt] = ldg(P.Z, zs + zat(k));
        d[Tt] = ldg(P.Dff, qat(k));
      });
      sfor<0, UN>([&](auto t) {
        constexpr int Tt = decltype(t)::value;
        const double v = fabs(d[Tt]) / (fabs(u[Tt]) + 1.0);
        const double m = umax(v);
        acc += (k0 + Tt < N - 1) ? m : 0.0;
      });
    }
    prio_base();
    return acc / (double)(N - 1);
  }

  // Lone streaming sweeps (conic kernels).  The sweeps below have no recurrence over the knots, so when only ONE row of
  // the wave needs one -- the stragglers of a conic launch walk thousands of line-searched iterations after their
  // wave-mates are done (tools/gpu_rocket_tail.py: 7050 iterations of one instance in 20 steps against a mean of 750) --
  // the wave's four DPP rows take every fourth chunk of knots of that one instance (the caller points inst / rowoff / rs
  // of all lanes at it) and the partial results are combined across the rows.
  static __device__ __forceinline__ double rows_sum4(double v) {
    double o[4];
    rows_gather(v, o);
    return (o[0] + o[1]) + (o[2] + o[3]);
  }
  static __device__ __forceinline__ double rows_max4(double v) {
    double o[4];
    rows_gather(v, o);
    return fmax(fmax(o[0], o[1]), fmax(o[2], o[3]));
  }

  // ---- line search on alpha < 1 without new rollouts ------------------------------------
  // Dynamics are linear/affine and the control law u + K dx + alpha d is affine, so the trial
  // trajectory is affine in alpha:  Z̄(alpha) = Z + alpha (Z̄(1) - Z)  (exact identity; alpha is a
  // power of two, so alpha*(Z̄(1)-Z) is exact in FP64 and the only rounding is the final add).
  // Plane cur^1 holds Z̄(1) from the alpha = 1 rollout.  trial_costs evaluates cost!, the
  // violation, the state/control limits and the "reproduces Z bit for bit" test for NA
  // consecutive step sizes alpha, alpha/2, ... in ONE streaming sweep (no gains, no serial chain).
  // (box-only kernels: a line search beyond alpha = 1 is rare -- 0.03 trials per solve on the headline -- and two trials
  // per sweep keep its registers out of the way: +1-2 %; the conic kernels need six trials per iteration and prefer four)
  static constexpr int NA = CONES ? ALTRO_NA : (ALTRO_NA < 2 ? ALTRO_NA : 2);
  struct Trials {
    double J[NA], cmax[NA];
    bool limit[NA], unchanged[NA];
  };

  template <bool LONE = false>
  __device__ void trial_costs(double alpha, Trials& T) {
    phase_begin();
    const int rr = LONE ? (lane >> 4) : 0;  // lone sweep: this DPP row takes the chunks rr, rr + 4, ...
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const int cur = rs->cur, kref = rs->kref;
    const unsigned zs = plane(cur);
    const unsigned z1 = plane(cur ^ 1);
    const int N = P.N;
    double Jacc[NA], viol[NA];
    bool lim[NA], chg[NA];
    double a[NA];
    sfor<0, NA>([&](auto t) {
      constexpr int Tt = decltype(t)::value;
      Jacc[Tt] = 0.0;
      viol[Tt] = 0.0;
      lim[Tt] = false;
      chg[Tt] = false;
      a[Tt] = alpha * (1.0 / (double)(1 << Tt));
    });
    const bool live = is_x | is_u;
    const double lm = is_x ? P.o.max_state_value : P.o.max_control_value;
    const double dmax = P.o.dual_max;
    const bool so2 = P.o.soc_second_order != 0;
    constexpr int UN = CONES ? 2 : ALTRO_UN;  // cone tables cost registers
    const int nch = LONE ? (N + 4 * UN - 1) / (4 * UN) : (N + UN - 1) / UN;
    // The cost of a trial is summed in FOUR classes of chunks (chunk index mod 4), each in ascending order, and the
    // classes are added as (J0 + J1) + (J2 + J3): the lone sweep, where DPP row r holds exactly class r, and the
    // row-parallel sweep then produce the same bits (the scheduling switches must not change results).
    double Jcls[NA][4];
    sfor<0, NA>([&](auto t) { sfor<0, 4>([&](auto q) { Jcls[decltype(t)::value][decltype(q)::value] = 0.0; }); });
    ConK ck0;  // time-invariant constraint tables: this lane's row, loaded once for the whole sweep
    const bool inv = CONES && (P.con_inv != 0);
    if constexpr (CONES) con_load(inv ? P.ckn[j] : 0, ck0);
    for (int c = 0; c < nch; ++c) {
      const int k0 = LONE ? (4 * c + rr) * UN : c * UN;
      const int cls = LONE ? 0 : (c & 3);
      sfor<0, NA>([&](auto t) { Jacc[decltype(t)::value] = 0.0; });
      double z[UN], zz1[UN], zr[UN], lhi[UN], llo[UN], lcq[UN];
      ConK ckq[UN];
      sfor<0, UN>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        const int k = imin(k0 + Q, N - 1);
        z[Q] = ldg(P.Z, zs + zat(k));
        zz1[Q] = ldg(P.Z, z1 + zat(k));
        zr[Q] = ldg(P.Zref, rat(kref + k));
        lhi[Q] = ldg(P.Lb, lb_at(k, 0));
        llo[Q] = ldg(P.Lb, lb_at(k, 1));
        lcq[Q] = 0.0;
        if constexpr (CONES) {
          lcq[Q] = ldg(P.Lc, qat(k));
          if (inv) ckq[Q] = ck0;
          else con_load(k, ckq[Q]);
        }
      });
      sfor<0, UN>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        const int k = k0 + Q;
        const bool valid = k < N;
        const bool term = (k >= N - 1);
        const bool bx = box_at(k) & valid;
        const double dz = zz1[Q] - z[Q];
        const double w = valid ? (term ? lc.wf : lc.wd) : 0.0;
        const bool on = (term ? is_x : live) & valid;
        const double lh = bx ? lhi[Q] : 0.0, ll = bx ? llo[Q] : 0.0;
        // constraint values are affine in alpha as well: v(alpha) = v(Z) + alpha (v(Z1) - v(Z))
        double cv0 = 0.0, cdv = 0.0;
        bool cact = false;
        if constexpr (CONES) {
          cact = con_act(ckq[Q].cm, k) & valid;
          cv0 = con_value(on ? z[Q] : 0.0, ckq[Q].arow, ckq[Q].brow);
          cdv = con_value(on ? zz1[Q] : 0.0, ckq[Q].arow, ckq[Q].brow) - cv0;
        }
        sfor<0, NA>([&](auto t) {
          constexpr int Tt = decltype(t)::value;
          const double zb = on ? __builtin_fma(a[Tt], dz, z[Q]) : 0.0;
          if constexpr (CONES) {
            const ConeEval e = cone_eval<false>(__builtin_fma(a[Tt], cdv, cv0), cact ? lcq[Q] : 0.0, mu, ckq[Q].cm, cact, dmax, so2);
            Jacc[Tt] += e.cost;
            viol[Tt] = fmax(viol[Tt], e.viol);
          }
          Jacc[Tt] += lane_cost(lc, mu, zb, on ? zr[Q] : 0.0, w, lh, ll, bx & on, viol[Tt]);
          lim[Tt] = lim[Tt] | (on & !(fabs(zb) <= lm));
          chg[Tt] = chg[Tt] | (on & (zb != z[Q]));
        });
      });
      sfor<0, NA>([&](auto t) {
        constexpr int Tt = decltype(t)::value;
        sfor<0, 4>([&](auto q) {
          constexpr int Q = decltype(q)::value;
          Jcls[Tt][Q] += (cls == Q) ? Jacc[Tt] : 0.0;
        });
      });
    }
    sfor<0, NA>([&](auto t) {
      constexpr int Tt = decltype(t)::value;
      T.J[Tt] = LONE ? rows_sum4(row_sum(Jcls[Tt][0]))
                     : (row_sum(Jcls[Tt][0]) + row_sum(Jcls[Tt][1])) + (row_sum(Jcls[Tt][2]) + row_sum(Jcls[Tt][3]));
      T.cmax[Tt] = LONE ? rows_max4(row_max(viol[Tt])) : row_max(viol[Tt]);
      T.limit[Tt] = LONE ? wave_any(lim[Tt]) : row_any(lim[Tt], lane);
      T.unchanged[Tt] = LONE ? !wave_any(chg[Tt]) : !row_any(chg[Tt], lane);
    });
  }

  // Z̄ <- Z + alpha (Z̄(1) - Z) in plane cur^1, for the rows flagged `doit`
  template <bool LONE = false>
  __device__ void interpolate(double alpha, bool doit) {
    phase_begin();
    const int rr = LONE ? (lane >> 4) : 0;
    const int cur = rs->cur;
    const unsigned zs = plane(cur);
    const unsigned z1 = plane(cur ^ 1);
    const int N = P.N;
    constexpr int UN = ALTRO_UN;
    const int nch = LONE ? (N + 4 * UN - 1) / (4 * UN) : (N + UN - 1) / UN;
    for (int c = 0; c < nch; ++c) {
      const int k0 = LONE ? (4 * c + rr) * UN : c * UN;
      double z[UN], zz1[UN];
      sfor<0, UN>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        const int k = imin(k0 + Q, N - 1);
        z[Q] = ldg(P.Z, zs + zat(k));
        zz1[Q] = ldg(P.Z, z1 + zat(k));
      });
      sfor<0, UN>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        const int k = k0 + Q;
        const bool on = (k >= N - 1) ? is_x : (is_x | is_u);
        const double zb = on ? __builtin_fma(alpha, zz1[Q] - z[Q], z[Q]) : 0.0;
        stg(P.Z, zat((doit & (k < N)) ? (cur ^ 1) * N + imin(k, N - 1) : 2 * N), zb);
      });
    }
  }

  // gradient / Gauss-Newton hessian of the box AL term of this lane's element (branch-free)
  // `code` (2 bits): which sides entered the Hessian -- the active set the gains of a backward pass depend on
  static __device__ __forceinline__ void box_expand(const LaneConst& c, double mu, double z, double lhi, double llo,
                                                    bool on, double& qz, double& hz, unsigned& code) {
    const double chi = z - c.zmax, clo = c.zmin - z;
    const bool ahi = (chi >= 0.0) | (lhi > 0.0);
    const bool alo = (clo >= 0.0) | (llo > 0.0);
    const double ghi = lhi + (ahi ? mu * chi : 0.0);
    const double glo = llo + (alo ? mu * clo : 0.0);
    const bool bh = on & c.has_hi, bl = on & c.has_lo;
    qz += bh ? ghi : 0.0;
    hz += (bh & ahi) ? mu : 0.0;
    qz -= bl ? glo : 0.0;
    hz += (bl & alo) ? mu : 0.0;
    code = ((bh & ahi) ? 1u : 0u) | ((bl & alo) ? 2u : 0u);
  }

  // backwardpass! (SURVEY A.3 / oracle backward_pass): Riccati recursion over plane `cur`,
  // writes the gain blocks KD, returns dV and whether any Quu pivot was not positive.
  //
  // Algebra used (exact identities of the reference's formulas, with Quu_reg = Quu + rho I and
  // Quu_reg K = -Qux, Quu_reg d = -Qu):
  //     Quu K + Qux = -rho K,  Quu d + Qu = -rho d
  //     S = Qxx + K'(Quu K + Qux) + Qux'K = Qxx + Qux'K - rho K'K
  //     s = Qx + K'(Quu d + Qu) + Qux'd  = Qx + Qux'd - rho K'd
  //     dV = (d'Qu, 1/2 d'Quu d)         = (d'Qu, -1/2 d'Qu - 1/2 rho d'd)
  // Quu_reg is factored as L D L' (no square roots; pivots D_j > 0 is the same PD test as
  // Cholesky's).  RHO = false is the rho == 0 instantiation (every convex run).
  // SYM: S <- (S + S')/2 after every knot as Altro.jl does (altro_opts.strict; costs 6 %, the asymmetry it
  // removes stays at rounding level because the closed loop is contracting -- DESIGN.md "The kernel").
  //
  // dtiny (out): every feedforward term of this pass is at rounding level, max_a |d_k,a| <= 1e-9 (1 + |u_k,a|) on
  // every control lane of every knot.  The step this pass proposes then moves nothing by more than ~1e-8: see the
  // "confirmation iteration" shortcut in run().
  // QV (box-only kernels, rho == 0): every row that takes part holds a trajectory a rollout has just produced -- then that
  // rollout has left, per knot, exactly what the expansion would recompute: the gradient of the cost and box terms in the
  // plane Qz (lane_cost_grad forms the sum box_expand forms, in its order) and the active set in qhs (two bits per knot:
  // the Hessian diagonal is w + mu [upper] + mu [lower]).  The pass then reads ONE row per knot (plus z, for the test of
  // the feedforward terms against |u|) instead of z, z_ref and both dual rows, and skips the expansion's ~35 instructions;
  // its own active set IS the trajectory's.  Same values, same operations on them: bit-identical to the full expansion
  // (tests: the lone and four-row passes, K fused steps against K launches, the oracle).
  template <bool RHO, bool SYM, bool QV = false>
  __device__ void backward(double& dV1, double& dV2, bool& fail, bool& dtiny, bool live) {
    static_assert(!QV || (!CONES && !RHO), "the Qz form of the pass: box-only kernels, no regularisation");
    phase_begin();
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const double rho = RHO ? rs->rho : 0.0;
    const int kref = rs->kref;
    const unsigned zs = plane(rs->cur);
    double g[NX];
    sfor<0, NX>([&](auto c) {
      constexpr int C = decltype(c)::value;
      g[C] = ldg(P.Gcol, ((unsigned)inst * NX + C) * LW + j);
    });
    const int N = P.N;
    const double dmax = P.o.dual_max;
    const bool so2 = P.o.soc_second_order != 0;
    // conic AL expansion at one knot: gradient A'g added to qz, Hessian A' M A added to hh.
    // Row j and column j of the knot's constraint table are loaded here (L2-resident, shared by
    // every instance).
    auto cone_expand = [&](int k, const ConK& ck, const double (&acol)[16], double z, double lam, double& qz, double (&hh)[NZ]) {
      const bool act = con_act(ck.cm, k);
      const double v = con_value(z, ck.arow, ck.brow);
      const ConeEval e = cone_eval<true>(v, act ? lam : 0.0, mu, ck.cm, act, dmax, so2);
      double a4[4] = {0.0, 0.0, 0.0, 0.0};
      BlkCommon::RS16(a4, e.g, acol);
      qz += (a4[0] + a4[1]) + (a4[2] + a4[3]);
      double y[16];
      sfor<0, 16>([&](auto r) { y[decltype(r)::value] = 0.0; });
      BlkCommon::YM(y, e.m, acol);
      Blk<NX, NU>::HC(hh, acol, y);
    };
    // terminal expansion: S = Qf (+ box / cone hessian), s = Qf (x - xr) (+ box / cone gradient)
    ASet* const ta = live ? ah : atr;   // the active set this pass sees, knot by knot
    if constexpr (QV) aset_copy(ta, qhs);
    else aset_clear(ta);
    double Sx[NX + 1];
    {
      const int k = N - 1;
      double qz, hz;
      double z = 0.0;
      if constexpr (QV) {
        qz = ldg(P.Qz, qat(k));
        hz = hz_of(lc.wf, mu, aset_get(qhs, k));
      } else {
        z = ldg(P.Z, zs + zat(k));
        const double zr = ldg(P.Zref, rat(kref + k));
        const double lhi = ldg(P.Lb, lb_at(k, 0)), llo = ldg(P.Lb, lb_at(k, 1));
        qz = lc.wf * (z - zr);
        hz = lc.wf;
        unsigned codeT;
        box_expand(lc, mu, z, lhi, llo, box_at(k) & is_x, qz, hz, codeT);
        aset_add(ta, codeT, k);
      }
      if constexpr (CONES) {
        double hT[NZ];
        sfor<0, NZ>([&](auto c) {
          constexpr int C = decltype(c)::value;
          hT[C] = (j == C) ? hz : 0.0;
        });
        ConK ckT;
        double acolT[16];
        con_load(k, ckT);
        con_col(k, acolT);
        cone_expand(k, ckT, acolT, is_x ? z : 0.0, ldg(P.Lc, qat(k)), qz, hT);
        sfor<0, NX>([&](auto c) { Sx[decltype(c)::value] = hT[decltype(c)::value]; });
      } else {
        sfor<0, NX>([&](auto c) {
          constexpr int C = decltype(c)::value;
          Sx[C] = (j == C) ? hz : 0.0;
        });
      }
      Sx[NX] = is_x ? qz : 0.0;
    }
    dV1 = 0.0;
    dV2 = 0.0;
    fail = false;
    bool dbig = false;
    double* my = sm;
    // operands of the knot about to be processed (loaded one knot ahead)
    double z = ldg(P.Z, zs + zat(N - 2)), zr = QV ? ldg(P.Qz, qat(N - 2)) : ldg(P.Zref, rat(kref + N - 2));   // QV: zr carries Qz
    double lhi = 0.0, llo = 0.0;
    if constexpr (!QV) {
      lhi = ldg(P.Lb, lb_at(N - 2, 0));
      llo = ldg(P.Lb, lb_at(N - 2, 1));
    }
    double lcc = 0.0;
    // the knot's constraint table (this lane's row and column, 26+ loads from L2) is requested one knot
    // ahead like the other operands: the conic kernels run one wave per SIMD, nothing else hides it
    ConK ckc;
    double acolc[16];
    if constexpr (CONES) {
      lcc = ldg(P.Lc, qat(N - 2));
      con_load(N - 2, ckc);
      con_col(N - 2, acolc);
    }
    // The operands loaded ahead are waited for HERE, once.  hipcc's wait-count pass merges the state of the loop's two
    // entries; with these loads still pending on the way in, the merged state made every iteration wait at its first use
    // of them for all but the 4 youngest memory operations (`s_waitcnt vmcnt(4)` right after the body's own four loads),
    // i.e. for the FIVE GAIN STORES of the previous knot: one store round trip per knot, the reason the pass slowed down
    // by 40-80 % whenever the memory system was busy.
    asm volatile("" : "+v"(z), "+v"(zr));
    if constexpr (!QV) asm volatile("" : "+v"(lhi), "+v"(llo));
    if constexpr (CONES) asm volatile("" : "+v"(lcc));
    for (int k = N - 2; k >= 0; --k) {  // body: one basic block
      const int km = imax(k - 1, 0);
      const double zn = ldg(P.Z, zs + zat(km));
      const double zrn = QV ? ldg(P.Qz, qat(km)) : ldg(P.Zref, rat(kref + km));
      double lhin = 0.0, llon = 0.0;
      if constexpr (!QV) {
        lhin = ldg(P.Lb, lb_at(km, 0));
        llon = ldg(P.Lb, lb_at(km, 1));
      }
      double lcn = 0.0;
      if constexpr (CONES) lcn = ldg(P.Lc, qat(km));
      double qz, hz;
      if constexpr (QV) {
        qz = zr;
        hz = hz_of(lc.wd, mu, aset_get(qhs, k));
      } else {
        qz = lc.wd * (z - zr);
        hz = lc.wd;
        unsigned code;
        box_expand(lc, mu, z, lhi, llo, box_at(k), qz, hz, code);
        aset_add(ta, code, k);
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
      if constexpr (CONES) {
        cone_expand(k, ckc, acolc, (is_x | is_u) ? z : 0.0, lcc, qz, h);
        con_load(km, ckc);  // consumed: refill for the next knot
        con_col(km, acolc);
      }
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
        });
      });
      // L D L' of Quu + rho I (redundantly on every lane); Ld[i][k] = L[i][k]*D[k]
      double L[NU][NU], Ld[NU][NU], dinv[NU];
      sfor<0, NU>([&](auto jc) {
        constexpr int Jc = decltype(jc)::value;
        double dd = RHO ? quu[Jc][Jc] + rho : quu[Jc][Jc];
        sfor<0, Jc>([&](auto kk) {
          constexpr int Kk = decltype(kk)::value;
          dd -= L[Jc][Kk] * Ld[Jc][Kk];
        });
        fail = fail | !(dd > 0.0);
        dinv[Jc] = rcp_nr(dd);
        sfor<Jc + 1, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = quu[I][Jc];
          sfor<0, Jc>([&](auto kk) {
            constexpr int Kk = decltype(kk)::value;
            v -= L[I][Kk] * Ld[Jc][Kk];
          });
          Ld[I][Jc] = v;
          L[I][Jc] = v * dinv[Jc];
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
          y[I] = v;
        });
        sfor<0, NU>([&](auto ir) {
          constexpr int I = NU - 1 - decltype(ir)::value;
          double v = y[I] * dinv[I];
          sfor<I + 1, NU>([&](auto kk) { v -= L[decltype(kk)::value][I] * kd[decltype(kk)::value]; });
          kd[I] = v;
        });
        sfor<0, NU>([&](auto a) { kd[decltype(a)::value] = -kd[decltype(a)::value]; });
      }
      {  // feedforward magnitude against the control it would change (u lanes hold every d[a] in kd[])
        double dm = fabs(kd[0]);
        sfor<1, NU>([&](auto a) { dm = fmax(dm, fabs(kd[decltype(a)::value])); });
        dbig = dbig | (is_u & !(dm <= 1e-9 * (1.0 + fabs(z))));
      }
      // d to every lane (from the first u lane)
      double dd_[NU];
      sfor<0, NU>([&](auto a) { dd_[decltype(a)::value] = bcast<NX>(kd[decltype(a)::value]); });
      // s = Qx + Qux'd - rho K'd ;  dV += (d'Qu, -1/2 d'Qu - 1/2 rho d'd)
      double snew = gz, t1 = 0.0, dtd = 0.0, ktd = 0.0;
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        snew += r[A] * dd_[A];
        t1 += dd_[A] * qu[A];
        if constexpr (RHO) {
          dtd += dd_[A] * dd_[A];
          ktd += kd[A] * dd_[A];
        }
      });
      if constexpr (RHO) snew -= rho * ktd;
      dV1 += t1;
      dV2 += RHO ? (-0.5 * t1 - 0.5 * rho * dtd) : (-0.5 * t1);
      store_gains(live ? k : N - 1, live ? k : N, kd, L, dinv);
      // S = Qxx + Qux'K - rho K'K   (in place on h[0..NX-1]), then S = (S + S')/2
      if constexpr (RHO) {
        double T[NU];
        sfor<0, NU>([&](auto a) { T[decltype(a)::value] = -rho * kd[decltype(a)::value]; });
        Blk<NX, NU>::CTG(h, kd, T, r);
      } else {
        Blk<NX, NU>::CTG0(h, kd, r);
      }
      // transpose through LDS.  The workgroup is ONE wave and each row has its own tile, so no
      // s_barrier is needed: LDS operations of a wave execute in issue order.  Lanes >= NX read
      // rows that were never written and carry garbage columns from here on; nothing ever reads
      // a column of a lane >= NX (the DPP broadcasts only source lanes < NX).
      if constexpr (SYM) {
        sfor<0, NX>([&](auto c) {
          constexpr int C = decltype(c)::value;
          my[C * (LW + 1) + j] = h[C];
        });
        __builtin_amdgcn_wave_barrier();
        sfor<0, NX>([&](auto c) {
          constexpr int C = decltype(c)::value;
          const double st = my[j * (LW + 1) + C];
          Sx[C] = 0.5 * (h[C] + st);
        });
        __builtin_amdgcn_wave_barrier();
      } else {
        sfor<0, NX>([&](auto c) { Sx[decltype(c)::value] = h[decltype(c)::value]; });
      }
      Sx[NX] = snew;
      z = zn;
      zr = zrn;
      lhi = lhin;
      llo = llon;
      lcc = lcn;
    }
    dtiny = !row_any(dbig, lane);
  }

  // Gains of one knot out (kk / kf: the knot, or the trash slots N-1 of KD and N of Dff for rows that sit the pass out).
  // Every lane stores its own column: x lane j the K[a][j]; u lane NX + b the factors of Quu the first-order sweep
  // needs to get d from Qu without a backward pass -- entry (a, b) of [1 / D on the diagonal, L below] in gain row a --
  // and, in Dff, lane NX + a the feedforward term d[a] (on the u lanes kd[] is the solve of Qu, i.e. d).
  __device__ __forceinline__ void store_gains(int kk, int kf, const double (&kd)[NU], const double (&L)[NU][NU],
                                              const double (&dinv)[NU]) const {
    double dl = kd[0];
    sfor<1, NU>([&](auto a) {
      constexpr int A = decltype(a)::value;
      dl = (j == NX + A) ? kd[A] : dl;
    });
    stg(P.Dff, qat(kf), dl);
    sfor<0, NU>([&](auto a) {
      constexpr int A = decltype(a)::value;
      double v = kd[A];
      sfor<0, A>([&](auto b) {
        constexpr int Bq = decltype(b)::value;
        v = (j == NX + Bq) ? L[A][Bq] : v;
      });
      v = (j == NX + A) ? dinv[A] : v;
      stg(P.KD, kd_at(kk, A), v);
    });
  }

  // First-order sweep (default mode, box-only problems): an iteration whose active set and penalty are those of the
  // backward pass that left the gains in KD -- in this solve or in an earlier one, in this launch or an earlier one --
  // does not need that pass again.  Inside a fixed active set the AL problem is LQ: K_k and Quu_k depend on the
  // dynamics, the weights, the penalty and WHICH rows are active, not on the trajectory, so they are what the pass
  // would recompute.  What does depend on the trajectory is first order:
  //     [Qx; Qu] = l_z + [A B]' s_{k+1},   d_k = -Quu^-1 Qu  (factors stored with the gains),   s_k = Qx + K_k' Qu,
  //     dV += (d'Qu, -1/2 d'Qu)
  // (identities of backward() with rho = 0).  l_z at the current trajectory was left in the plane Qz by the rollout that
  // produced it.  One 12-FMA product per knot on the critical path (s_{k+1} -> s_k); the solve for d hangs off it.
  // Writes d into Dff (rows with live), returns dV and the tiny-feedforward flag of backward().  54 % of the backward
  // passes of the headline workload are of this kind (tools/gpu_reuse_diag.py).
  __device__ void fosweep(bool live, double& dV1, double& dV2, bool& dtiny) {
    phase_begin();
    prio_serial();
    double g[NX];
    sfor<0, NX>([&](auto c) {
      constexpr int C = decltype(c)::value;
      g[C] = ldg(P.Gcol, ((unsigned)inst * NX + C) * LW + j);
    });
    const unsigned zs = plane(rs->cur);
    const int N = P.N;
    bool dbig = false;
    dV1 = 0.0;
    dV2 = 0.0;
    double sv = ldg(P.Qz, qat(N - 1));  // terminal knot: l_x on the state lanes, 0 elsewhere
    constexpr int PD = ALTRO_PD_FOSWEEP;
    struct In {
      double qz, z, kr[NU];
    };
    auto load = [&](int k, In& in) {
      in.qz = ldg(P.Qz, qat(k));
      in.z = ldg(P.Z, zs + zat(k));
      sfor<0, NU>([&](auto a) { in.kr[decltype(a)::value] = ldg(P.KD, kd_at(k, decltype(a)::value)); });
    };
    // two register sets of PD knots: the operands of the next group are requested before the current one is consumed (see
    // adjoint(): a reload-after-use ring has no lead time for the loads of a loop body's last knots)
    In ra[PD], rb[PD];
    int k = N - 2;
    sfor<0, PD>([&](auto u) { load(imax(k - decltype(u)::value, 0), ra[decltype(u)::value]); });
    // The feedforward terms of a group are stored one group LATER, after that group's operands have been waited for.  The
    // memory counter is in order: a store issued between the loads of the next group and the wait for them is waited for
    // as well -- issued a few instructions earlier, that was a full store round trip per group of PD knots.
    double dst_[PD];
    int kst = N;  // first knot of the pending stores (N: none yet -> the trash row)
    sfor<0, PD>([&](auto u) { dst_[decltype(u)::value] = 0.0; });
    auto flush = [&]() {
      sfor<0, PD>([&](auto u) {
        constexpr int U = decltype(u)::value;
        stg(P.Dff, qat((live & (kst - U >= 0) & (kst < N)) ? kst - U : N), dst_[U]);
      });
    };
    auto group = [&](In (&cur_)[PD], In (&nxt)[PD]) {
      sfor<0, PD>([&](auto u) { load(imax(k - PD - decltype(u)::value, 0), nxt[decltype(u)::value]); });
      landed_in(cur_[0].qz);   // the wait for this group's operands goes here, BEFORE the previous group's stores
      flush();
      kst = k;
      sfor<0, PD>([&](auto u) {
        constexpr int U = decltype(u)::value;
        const In& in = cur_[U];
        const bool valid = k - U >= 0;
        double acc4[4] = {in.qz, 0.0, 0.0, 0.0};
        Blk<NX, NU>::GTS(acc4, sv, g);
        const double gz = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);  // x lanes: Qx, u lanes: Qu
        double qu[NU];
        sfor<0, NU>([&](auto a) { qu[decltype(a)::value] = bcast<NX + decltype(a)::value>(gz); });
        // s_k = Qx + K' Qu (x lanes: kr[a] = K[a][j])
        double snew = gz;
        sfor<0, NU>([&](auto a) { snew += in.kr[decltype(a)::value] * qu[decltype(a)::value]; });
        // d = -(L D L')^-1 Qu, the factors from the u lanes of the gain rows (same arithmetic as backward())
        double y[NU], dk[NU];
        sfor<0, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = qu[I];
          sfor<0, I>([&](auto kk) {
            constexpr int Kk = decltype(kk)::value;
            v -= bcast<NX + Kk>(in.kr[I]) * y[Kk];
          });
          y[I] = v;
        });
        sfor<0, NU>([&](auto ir) {
          constexpr int I = NU - 1 - decltype(ir)::value;
          double v = y[I] * bcast<NX + I>(in.kr[I]);
          sfor<I + 1, NU>([&](auto kk) {
            constexpr int Kk = decltype(kk)::value;
            v -= bcast<NX + I>(in.kr[Kk]) * dk[Kk];
          });
          dk[I] = v;
        });
        double t1 = 0.0, dm = 0.0, dl = 0.0;
        sfor<0, NU>([&](auto a) {
          constexpr int A = decltype(a)::value;
          dk[A] = -dk[A];
          t1 += dk[A] * qu[A];
          dm = fmax(dm, fabs(dk[A]));
          dl = (j == NX + A) ? dk[A] : dl;
        });
        dbig = dbig | (valid & is_u & !(dm <= 1e-9 * (1.0 + fabs(in.z))));
        dV1 += valid ? t1 : 0.0;
        dV2 += valid ? -0.5 * t1 : 0.0;
        dst_[U] = dl;
        sv = valid ? (is_x ? snew : 0.0) : sv;
      });
      k -= PD;
    };
    while (k >= 0) {  // body: two groups
      group(ra, rb);
      if (k < 0) break;
      group(rb, ra);
    }
    flush();
    dtiny = !row_any(dbig, lane);
    prio_base();
  }

  // ---- lone-row backward pass ---------------------------------------------------------------------------
  // When exactly one row of the wave needs a backward pass (the tail of a launch, where single hard instances walk
  // their serial chains, and every turn of a wave whose rows are out of step), the wave's other three DPP rows
  // would execute the pass's 396 FMAs per knot on operands nobody reads.  backward_lone() spreads the ONE instance
  // over all four rows instead: row r owns rows r*RL .. r*RL + RL-1 of S, W = S [A B] and Qxx (row 0 also the
  // vector s), and rows NX + r*RQ.. of [Qux Quu]; lane j is column j in every row, as before.  The rows trade their
  // slices of W and of [Qux Quu] with v_permlane32_swap / v_permlane16_swap (gfx950), everything that is not a
  // product (box expansion, L D L', gain solves, dV) runs redundantly on all four.  153 product / exchange
  // instructions per knot instead of 396.  Every output element is the same chain of FMAs in the same order as in
  // backward<false, SYM>, so the two passes agree bit for bit (tests: ALTRO_NO_LONE=1 against the default).
  // The caller points inst / rowoff / rs / ah of ALL lanes at the lone row before the call.  rho == 0 only.
  static __device__ __forceinline__ void rows_gather(double v, double (&o)[4]) {
    // o[q] = v of the same lane of DPP row q
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);  // [0] = rows 0,1 | 0,1   [1] = rows 2,3 | 2,3
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const auto a0 = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false);  // [0] = row 0 everywhere, [1] = row 1
    const auto b0 = __builtin_amdgcn_permlane16_swap(b[0], b[0], false, false);
    const auto a1 = __builtin_amdgcn_permlane16_swap(a[1], a[1], false, false);  // [0] = row 2, [1] = row 3
    const auto b1 = __builtin_amdgcn_permlane16_swap(b[1], b[1], false, false);
    o[0] = __hiloint2double(b0[0], a0[0]);
    o[1] = __hiloint2double(b0[1], a0[1]);
    o[2] = __hiloint2double(b1[0], a1[0]);
    o[3] = __hiloint2double(b1[1], a1[1]);
  }
  static __device__ __forceinline__ double rows_gather0(double v) {  // v of the same lane of DPP row 0
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const auto a0 = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false);
    const auto b0 = __builtin_amdgcn_permlane16_swap(b[0], b[0], false, false);
    return __hiloint2double(b0[0], a0[0]);
  }
  static __device__ __forceinline__ double lane_gather(double v, int src_lane) {  // v of lane src_lane (ds_bpermute)
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }

  // Lone phases: all four DPP rows take the identity of row lrow (wave-uniform) for the duration of one phase.
  struct LoneCtx {
    int inst;
    unsigned rowoff;
    RowState* rs;
    double* sm;
    ASet* ah;
    ASet* qhs;
  };
  __device__ __forceinline__ LoneCtx lone_enter(int lrow) {
    LoneCtx c{inst, rowoff, rs, sm, ah, qhs};
    inst = __builtin_amdgcn_readlane(inst, lrow * LW);
    rowoff = (unsigned)inst * LW + j;
    rs = c.rs - (lane >> 4) + lrow;
    sm = c.sm - (lane >> 4) * (LW * (LW + 1)) + lrow * (LW * (LW + 1));
    ah = c.ah - lane + lrow * LW + j;
    qhs = c.qhs - lane + lrow * LW + j;
    return c;
  }
  __device__ __forceinline__ void lone_leave(const LoneCtx& c) {
    inst = c.inst;
    rowoff = c.rowoff;
    rs = c.rs;
    sm = c.sm;
    ah = c.ah;
    qhs = c.qhs;
  }
  // Shadow rows.  The four rows of a wave run every phase together, and a row that sits a phase out used to run it on its
  // own instance: real loads of operands nobody needs (a fifth of the kernel's HBM traffic).  It now takes the identity of a
  // row that does take part, exactly as in a lone phase: its loads hit the lines that row loads in the same instruction,
  // and what it stores is either that row's values over again (same inputs, same instructions) or goes to that row's
  // trash slots (its own flags still say "not mine").  Results cannot change: the caller reads a phase's outputs only for
  // the rows that asked for it.
  __device__ __forceinline__ LoneCtx shadow_enter(bool active) {
    LoneCtx c{inst, rowoff, rs, sm, ah, qhs};
    if (P.shadow == 0) return c;
    const unsigned long long bm = __ballot(active);
    if (bm == 0ull) return c;
    const int lrow = first_row(bm);  // wave-uniform: a row that takes part
    const int li = __builtin_amdgcn_readlane(inst, lrow * LW);
    inst = active ? inst : li;
    rowoff = active ? rowoff : (unsigned)li * LW + j;
    rs = active ? rs : c.rs - (lane >> 4) + lrow;
    sm = active ? sm : c.sm - (lane >> 4) * (LW * (LW + 1)) + lrow * (LW * (LW + 1));
    ah = active ? ah : c.ah - lane + lrow * LW + j;
    // (the trajectory's active set too: a shadow must compute what its row computes -- in the strict pass all four rows
    //  write the transpose tile of the row they shadow)
    qhs = active ? qhs : c.qhs - lane + lrow * LW + j;
    return c;
  }
  // the row of the wave (0..3) a wave-uniform ballot of a per-row flag names, and how many rows it names
  static __device__ __forceinline__ int rows_in(unsigned long long b) {
    return (int)((b & 1ull) + ((b >> 16) & 1ull) + ((b >> 32) & 1ull) + ((b >> 48) & 1ull));
  }
  static __device__ __forceinline__ int first_row(unsigned long long b) {
    return ((b >> 16) & 1ull) ? 1 : (((b >> 32) & 1ull) ? 2 : (((b >> 48) & 1ull) ? 3 : 0));
  }
  static __device__ __forceinline__ double row_value(double v, int lrow) {  // v of row lrow (any lane of it), wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lrow * LW);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lrow * LW);
    return __hiloint2double(hi, lo);
  }

  template <bool SYM, bool QV = false>   // QV: the expansion read back from Qz and the trajectory's active set (see backward())
  __device__ void backward_lone(double& dV1, double& dV2, bool& fail, bool& dtiny) {
    using BK = Blk<NX, NU>;
    constexpr int RL = BK::RL, RQ = BK::RQ;
    phase_begin();
    const int rr = lane >> 4;  // the quarter of the instance this DPP row owns
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const int kref = rs->kref;
    const unsigned zs = plane(rs->cur);
    // g: [A B] by columns as in backward(); gp: the columns this row's own output rows need as broadcast operands,
    // moved to lanes 0..: lane t < RL holds column rr*RL + t, lane RL + u holds control column NX + rr*RQ + u
    double g[NX], gp[NX];
    {
      const int pc = (j < RL) ? rr * RL + j : NX + rr * RQ + (j - RL);
      const bool pv = (j < RL) ? (rr * RL + j < NX) : ((j - RL < RQ) && (rr * RQ + (j - RL) < NU));
      sfor<0, NX>([&](auto c) {
        constexpr int C = decltype(c)::value;
        g[C] = ldg(P.Gcol, ((unsigned)inst * NX + C) * LW + j);
        const double v = ldg(P.Gcol, ((unsigned)inst * NX + C) * LW + (pv ? pc : 0));
        gp[C] = pv ? v : 0.0;
      });
    }
    const int N = P.N;
    // which global row of S / Qxx a local slot of this DPP row holds, and whether this lane is its diagonal
    bool diag_x[RL], own_x[RL], diag_u[RQ];
    sfor<0, RL>([&](auto t) {
      constexpr int Tt = decltype(t)::value;
      own_x[Tt] = rr * RL + Tt < NX;
      diag_x[Tt] = own_x[Tt] & (j == rr * RL + Tt);
    });
    sfor<0, RQ>([&](auto u) {
      constexpr int U = decltype(u)::value;
      diag_u[U] = (rr * RQ + U < NU) & (j == NX + rr * RQ + U);
    });
    const int psrc = (lane & 48) + ((rr * RL + j) & 15);  // lane whose [Qux] entry slot j of this row's S rows needs
    ASet* const ta = ah;   // (lone_enter has pointed it at the lone row's set; every DPP row ORs the same bits)
    if constexpr (QV) aset_copy(ta, qhs);
    else aset_clear(ta);
    double Sl[RL + 1];
    {
      const int k = N - 1;
      double qz, hz;
      if constexpr (QV) {
        qz = ldg(P.Qz, qat(k));
        hz = hz_of(lc.wf, mu, aset_get(qhs, k));
      } else {
        const double z = ldg(P.Z, zs + zat(k));
        const double zr = ldg(P.Zref, rat(kref + k));
        const double lhi = ldg(P.Lb, lb_at(k, 0)), llo = ldg(P.Lb, lb_at(k, 1));
        qz = lc.wf * (z - zr);
        hz = lc.wf;
        unsigned codeT;
        box_expand(lc, mu, z, lhi, llo, box_at(k) & is_x, qz, hz, codeT);
        aset_add(ta, codeT, k);
      }
      sfor<0, RL>([&](auto t) { Sl[decltype(t)::value] = diag_x[decltype(t)::value] ? hz : 0.0; });
      Sl[RL] = (is_x & (rr == 0)) ? qz : 0.0;
    }
    dV1 = 0.0;
    dV2 = 0.0;
    fail = false;
    bool dbig = false;
    double* my = sm;
    double z = ldg(P.Z, zs + zat(N - 2)), zr = QV ? ldg(P.Qz, qat(N - 2)) : ldg(P.Zref, rat(kref + N - 2));   // QV: zr carries Qz
    double lhi = 0.0, llo = 0.0;
    if constexpr (!QV) {
      lhi = ldg(P.Lb, lb_at(N - 2, 0));
      llo = ldg(P.Lb, lb_at(N - 2, 1));
    }
    asm volatile("" : "+v"(z), "+v"(zr));  // waited for once, outside the loop: see backward()
    if constexpr (!QV) asm volatile("" : "+v"(lhi), "+v"(llo));
    for (int k = N - 2; k >= 0; --k) {  // body: one basic block
      const int km = imax(k - 1, 0);
      const double zn = ldg(P.Z, zs + zat(km));
      const double zrn = QV ? ldg(P.Qz, qat(km)) : ldg(P.Zref, rat(kref + km));
      double lhin = 0.0, llon = 0.0;
      if constexpr (!QV) {
        lhin = ldg(P.Lb, lb_at(km, 0));
        llon = ldg(P.Lb, lb_at(km, 1));
      }
      double qz, hz;
      if constexpr (QV) {
        qz = zr;
        hz = hz_of(lc.wd, mu, aset_get(qhs, k));
      } else {
        qz = lc.wd * (z - zr);
        hz = lc.wd;
        unsigned code;
        box_expand(lc, mu, z, lhi, llo, box_at(k), qz, hz, code);
        aset_add(ta, code, k);
      }
      // this row's rows of W = [S; s'] G
      double wl[RL + 1];
      sfor<0, RL + 1>([&](auto t) { wl[decltype(t)::value] = 0.0; });
      BK::SGL(wl, Sl, g);
      // all rows of W to every DPP row
      double wa[NX + 1];
      sfor<0, RL>([&](auto t) {
        constexpr int Tt = decltype(t)::value;
        double o[4];
        rows_gather(wl[Tt], o);
        sfor<0, 4>([&](auto q) {
          constexpr int Q = decltype(q)::value;
          if constexpr (Q * RL + Tt < NX) wa[Q * RL + Tt] = o[Q];
        });
      });
      wa[NX] = rows_gather0(wl[RL]);
      // this row's rows of H = G' W + diag(lzz): RL rows of Qxx, RQ rows of [Qux Quu]
      double hl[RL + RQ];
      sfor<0, RL>([&](auto t) { hl[decltype(t)::value] = diag_x[decltype(t)::value] ? hz : 0.0; });
      sfor<0, RQ>([&](auto u) { hl[RL + decltype(u)::value] = diag_u[decltype(u)::value] ? hz : 0.0; });
      BK::GTWL(hl, gp, wa);
      const double gz = qz + wa[NX];  // Qx[j] on x lanes, Qu[a] on u lanes
      // rows of [Qux Quu] to every DPP row: hq[a] = what backward() calls h[NX + a]
      double hq[NU];
      sfor<0, RQ>([&](auto u) {
        constexpr int U = decltype(u)::value;
        double o[4];
        rows_gather(hl[RL + U], o);
        sfor<0, 4>([&](auto q) {
          constexpr int Q = decltype(q)::value;
          if constexpr (Q * RQ + U < NU) hq[Q * RQ + U] = o[Q];
        });
      });
      // the broadcast operand of S = Qxx + Qux'K on this row's rows: Qux[a][i] of row i = rr*RL + t sits on lane i
      // (requested here: the ds_bpermute round trip hides behind the factorisation)
      double rp[NU];
      sfor<0, NU>([&](auto a) { rp[decltype(a)::value] = lane_gather(hq[decltype(a)::value], psrc); });
      double quu[NU][NU];
      double qu[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        qu[A] = bcast<NX + A>(gz);
        sfor<0, A + 1>([&](auto b) {
          constexpr int Bq = decltype(b)::value;
          quu[A][Bq] = bcast<NX + Bq>(hq[A]);
        });
      });
      double L[NU][NU], Ld[NU][NU], dinv[NU];
      sfor<0, NU>([&](auto jc) {
        constexpr int Jc = decltype(jc)::value;
        double dd = quu[Jc][Jc];
        sfor<0, Jc>([&](auto kk) {
          constexpr int Kk = decltype(kk)::value;
          dd -= L[Jc][Kk] * Ld[Jc][Kk];
        });
        fail = fail | !(dd > 0.0);
        dinv[Jc] = rcp_nr(dd);
        sfor<Jc + 1, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = quu[I][Jc];
          sfor<0, Jc>([&](auto kk) {
            constexpr int Kk = decltype(kk)::value;
            v -= L[I][Kk] * Ld[Jc][Kk];
          });
          Ld[I][Jc] = v;
          L[I][Jc] = v * dinv[Jc];
        });
      });
      double r[NU], kd[NU];
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        r[A] = is_x ? hq[A] : qu[A];
      });
      {
        double y[NU];
        sfor<0, NU>([&](auto ii) {
          constexpr int I = decltype(ii)::value;
          double v = r[I];
          sfor<0, I>([&](auto kk) { v -= L[I][decltype(kk)::value] * y[decltype(kk)::value]; });
          y[I] = v;
        });
        sfor<0, NU>([&](auto ir) {
          constexpr int I = NU - 1 - decltype(ir)::value;
          double v = y[I] * dinv[I];
          sfor<I + 1, NU>([&](auto kk) { v -= L[decltype(kk)::value][I] * kd[decltype(kk)::value]; });
          kd[I] = v;
        });
        sfor<0, NU>([&](auto a) { kd[decltype(a)::value] = -kd[decltype(a)::value]; });
      }
      {
        double dm = fabs(kd[0]);
        sfor<1, NU>([&](auto a) { dm = fmax(dm, fabs(kd[decltype(a)::value])); });
        dbig = dbig | (is_u & !(dm <= 1e-9 * (1.0 + fabs(z))));
      }
      double dd_[NU];
      sfor<0, NU>([&](auto a) { dd_[decltype(a)::value] = bcast<NX>(kd[decltype(a)::value]); });
      double snew = gz, t1 = 0.0;
      sfor<0, NU>([&](auto a) {
        constexpr int A = decltype(a)::value;
        snew += r[A] * dd_[A];
        t1 += dd_[A] * qu[A];
      });
      dV1 += t1;
      dV2 += -0.5 * t1;
      store_gains((rr == 0) ? k : N - 1, (rr == 0) ? k : N, kd, L, dinv);  // one row stores; the others hit the trash slots
      BK::CTGL0(hl, kd, rp);  // S = Qxx + Qux'K on this row's rows
      if constexpr (SYM) {
        sfor<0, RL>([&](auto t) {
          constexpr int Tt = decltype(t)::value;
          my[(rr * RL + Tt) * (LW + 1) + j] = hl[Tt];  // rows >= NX of the tile are never read by a state lane
        });
        __builtin_amdgcn_wave_barrier();
        sfor<0, RL>([&](auto t) {
          constexpr int Tt = decltype(t)::value;
          const double st = my[j * (LW + 1) + ((rr * RL + Tt) & 15)];
          Sl[Tt] = 0.5 * (hl[Tt] + st);
        });
        __builtin_amdgcn_wave_barrier();
      } else {
        sfor<0, RL>([&](auto t) { Sl[decltype(t)::value] = hl[decltype(t)::value]; });
      }
      Sl[RL] = (rr == 0) ? snew : 0.0;
      z = zn;
      zr = zrn;
      lhi = lhin;
      llo = llon;
    }
    dtiny = !row_any(dbig, lane);
  }

  // Costate sweep (default mode, box-only problems): lambda_N = l_x(N), lambda_k = l_x(k) + A' lambda_{k+1},
  // g_k = l_u(k) + B' lambda_{k+1} -- the first-order part of the backward pass, one 12-FMA product per knot
  // instead of 396.  l_x, l_u at the current trajectory were left in the plane Qz by the alpha = 1 rollout that
  // produced it (one load per knot here).  By induction over the knots every feedforward term of the backward
  // pass vanishes iff every g_k does, and |d_k| <= |g_k|_2 / lambda_min(Quu) <= 2 |g_k|_inf / (dt R):
  // gtiny (out) says |g_k,a| <= 0.25e-9 dt R_a at every knot, i.e. |d| <= 0.5e-9.
  __device__ void adjoint(bool& gtiny) {
    phase_begin();
    prio_serial();
    double g[NX];
    sfor<0, NX>([&](auto c) {
      constexpr int C = decltype(c)::value;
      g[C] = ldg(P.Gcol, ((unsigned)inst * NX + C) * LW + j);
    });
    const double thr = 0.25e-9 * P.wd[j];
    const int N = P.N;
    bool gbig = false;
    double sv = ldg(P.Qz, qat(N - 1));  // terminal knot: l_x on the state lanes, 0 elsewhere
    // Two register sets of H knots each: the loads of group g + 1 are issued BEFORE group g is consumed.  A ring (reload slot
    // U right after knot U) looked like H knots of lead time but was not: hipcc places ONE s_waitcnt at the top of the loop
    // body that waits for every load of the ring except the youngest (vmcnt(1) with eight in flight), i.e. for loads issued a
    // knot ago -- a full memory round trip per group (22 k cycles per sweep of 49 knots where the issue is 8 k).  Here the
    // same conservative wait asks for exactly what is needed: everything but the H loads just issued.
    constexpr int H = ALTRO_PD_ADJOINT;
    double ra[H], rb[H];
    int k = N - 2;
    sfor<0, H>([&](auto u) { ra[decltype(u)::value] = ldg(P.Qz, qat(imax(k - decltype(u)::value, 0))); });
    auto group = [&](double (&cur_)[H], double (&nxt)[H]) {
      sfor<0, H>([&](auto u) { nxt[decltype(u)::value] = ldg(P.Qz, qat(imax(k - H - decltype(u)::value, 0))); });
      sfor<0, H>([&](auto u) {
        constexpr int U = decltype(u)::value;
        const bool valid = k - U >= 0;
        double acc4[4] = {cur_[U], 0.0, 0.0, 0.0};
        Blk<NX, NU>::GTS(acc4, sv, g);
        const double gz = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);  // x lanes: lambda_k, u lanes: g_k
        gbig = gbig | (valid & is_u & !(fabs(gz) <= thr));
        sv = valid ? (is_x ? gz : 0.0) : sv;
      });
      k -= H;
    };
    while (k >= 0) {  // body: two groups, one basic block each
      group(ra, rb);
      if (k < 0) break;
      group(rb, ra);
    }
    gtiny = !row_any(gbig, lane);
    prio_base();
  }

  // dual_update! for the box rows of plane `cur` (penalty_update! is the caller's mu *= phi)
  __device__ void dual_update(bool upd) {
    phase_begin();
    const LaneConst lc = consts();
    const double mu = rs->mu;
    const unsigned zs = plane(rs->cur);
    const double dmax = P.o.dual_max;
    for (int k = P.box_k0; k <= P.box_k1; ++k) {
      const bool on = (k < P.N - 1) ? (is_x | is_u) : is_x;
      const double z = ldg(P.Z, zs + zat(k));
      const double lhi = ldg(P.Lb, lb_at(k, 0)), llo = ldg(P.Lb, lb_at(k, 1));
      const double nhi = fmin(fmax(lhi + mu * (z - lc.zmax), 0.0), dmax);
      const double nlo = fmin(fmax(llo + mu * (lc.zmin - z), 0.0), dmax);
      stg(P.Lb, lb_at((upd & on & lc.has_hi) ? k : P.N, 0), nhi);
      stg(P.Lb, lb_at((upd & on & lc.has_lo) ? k : P.N, 1), nlo);
    }
    if constexpr (CONES) {  // dual_update! of the generic rows: eq / ineq clamp, SOC projection
      const bool so2 = P.o.soc_second_order != 0;
      for (int k = 0; k < P.N; ++k) {
        ConK ck;
        con_load(k, ck);
        const ConMeta& cm = ck.cm;
        const bool live = (k < P.N - 1) ? (is_x | is_u) : is_x;
        const double z = ldg(P.Z, zs + zat(k));
        const double lam = ldg(P.Lc, qat(k));
        const bool act = con_act(cm, k);
        const double v = con_value(live ? z : 0.0, ck.arow, ck.brow);
        const ConeEval e = cone_eval<false>(v, act ? lam : 0.0, mu, cm, act, dmax, so2);
        const bool row_on = act & ((cm.type == CT_SOC) ? (cm.pos < cm.p) : (cm.type != CT_NONE));
        stg(P.Lc, qat((upd & row_on) ? k : P.N), e.lam_new);
      }
    }
  }

  // Plant step of the MPC loop (random_linear_problem.jl:128-130) for the rows flagged `doit`:
  //   x0 <- A x_1 + B u_1 + f + randn(n) * ||x0||_inf / 100
  __device__ void plant_step(bool doit, int step) {
    phase_begin();
    double grow[NZ];
    sfor<0, NZ>([&](auto c) {
      constexpr int C = decltype(c)::value;
      grow[C] = ldg(P.Grow, ((unsigned)inst * LW + C) * LW + j);
    });
    const double z0 = ldg(P.Z, plane(rs->cur) + zat(0));
    double acc4[4] = {ldg(P.fvec, rowoff), 0.0, 0.0, 0.0};
    Blk<NX, NU>::GZ(acc4, z0, grow);
    const double xn = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    double nrm;
    const double wgt = P.noise_w[j];
    if (P.noise_mode == 0) {
      nrm = row_max(is_x ? fabs(xn) : 0.0);
    } else if (P.noise_mode == 2) {
      nrm = 1.0;
    } else {
      const int grp = P.noise_grp[j];
      const double sq = is_x ? xn * xn : 0.0;
      const double n0 = sqrt(row_sum(grp == 0 ? sq : 0.0)), n1 = sqrt(row_sum(grp == 1 ? sq : 0.0));
      nrm = grp == 0 ? n0 : n1;
    }
    double nz = 0.0;
    if (P.noise != nullptr && is_x && doit) {
      const int b = inst < P.B ? inst : P.B - 1;
      nz = P.noise[((size_t)step * P.B + b) * NX + j];
    }
    const double x0n = is_x ? xn + nz * nrm * wgt : 0.0;
    if (doit) stg(P.x0, rowoff, x0n);
  }

  // ------------------------------------------------------------------------------------------
  // The wave loop: solve!(::ALTROSolver) -> AL outer loop -> iLQR (oracle
  // orc_solve) and the MPC step sequence around it (random_linear_problem.jl:125-139,161), as a
  // per-row state machine.  mpc: nsteps consecutive MPC steps per row; else one plain solve!().
  __device__ void run(bool mpc, int first_step, int nsteps) {
    const altro_opts& o = P.o;
    const double mu0 = (o.penalty_initial != o.penalty_initial) ? 1.0 : o.penalty_initial;
    const double phi = (o.penalty_scaling != o.penalty_scaling) ? 10.0 : o.penalty_scaling;
    const bool has_con = (P.box_k1 >= P.box_k0) || (CONES && P.ncrows > 0);
    const int n_outer = has_con ? o.iterations_outer : 1;
    {  // row state init (every lane of the row writes the same words)
      RowState s;
      s.J = s.cmax = s.J_prev = s.rho = s.drho = s.dV1 = s.dV2 = s.cost_tol = s.grad_tol = 0.0;
      s.mu = P.mu[inst];
      s.phase = PH_STEP_BEGIN;
      s.status = ALTRO_UNSOLVED;
      s.iters = s.iters_outer = s.outer = s.it = s.dj_zero = s.shift = s.last = 0;
      s.cur = P.cur[inst];
      s.kref = P.kref;
      s.step = 0;
      s.nbw = s.nro = s.nsolve = s.nit = s.nok = s.ntr = 0;
      s.kmu = P.kmu[inst];
      s.qvalid = 0;
      s.gconf = P.dzero[inst];
      s.ngc = 0;
      s.nfo = 0;
      *rs = s;
      *ah = P.ahash[(unsigned)inst * LW + j];
      aset_clear(qhs);
    }
    __builtin_amdgcn_wave_barrier();

    while (true) {
      // ---------------- A. rows that begin a solve (one per MPC step, or the single plain solve)
      {
        const int ph = rs->phase;
        // Keeping the rows of a wave in step.  A warm MPC solve takes two turns (first iteration, then the confirmation);
        // a row that needed one more falls out of step with its wave-mates, and from then on EVERY turn of the wave
        // holds both kinds of work -- two backward passes, two open- and two closed-loop rollouts per step instead of
        // one (measured on the pass-heavy waves of a grouped launch: 50 passes in 20 steps).  So a row about to begin
        // a step waits one turn while a wave-mate is in the second turn of its solve: if that solve ends there (it
        // nearly always does) they begin the next step together.  A mate deep in a hard solve is not waited for.
        const bool mid = (ph == PH_ITER) && (rs->it == 1) && (rs->iters == 1);
        const bool hold = !CONES && mpc && (P.resync != 0) && wave_any(mid);
        const bool begin = (ph == PH_STEP_BEGIN) && !(hold && rs->step < nsteps);
        if (wave_any(begin)) {
          const int stp = rs->step;
          const bool go = begin && (stp < (mpc ? nsteps : 1));
          if (mpc && wave_any(go)) {
            const LoneCtx sh = shadow_enter(go);
            plant_step(go, first_step + stp);
            lone_leave(sh);
          }
          if (o.reset_duals && !P.prepare_only && wave_any(go)) {  // initialize!: lambda <- 0
            for (int k = P.box_k0; k <= P.box_k1; ++k) {
              stg(P.Lb, lb_at((go & bounded) ? k : P.N, 0), 0.0);
              stg(P.Lb, lb_at((go & bounded) ? k : P.N, 1), 0.0);
            }
            if constexpr (CONES) {
              for (int k = 0; k < P.N; ++k) stg(P.Lc, qat(go ? k : P.N), 0.0);
            }
          }
          if (begin) {
            if (go && !P.prepare_only) {
              if (mpc) rs->kref = first_step + stp + 1;  // update_trajectory!(obj, Z_track, k_mpc)
              if (o.reset_penalties) rs->mu = mu0;
              rs->status = ALTRO_UNSOLVED;
              rs->iters = 0;
              rs->iters_outer = 0;
              rs->outer = 0;
              rs->J = 0.0;
              rs->cmax = 0.0;
              rs->shift = (mpc && P.mpc_shift) ? 1 : 0;
              rs->gconf = 0;
              rs->phase = PH_OUTER_BEGIN;
            } else {
              rs->phase = PH_DONE;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (!wave_any(rs->phase != PH_DONE)) break;

      // ---------------- B. rows that begin an iLQR solve: open-loop rollout (initialize!)
      {
        const bool ob = rs->phase == PH_OUTER_BEGIN;
        if (wave_any(ob)) {
          ALTRO_STAMP(long long ts = stamp();)
          const bool ob_shift = ob && rs->shift != 0;
          const unsigned long long bmo = __ballot(ob);
          RollOut r0;
          bool lone_ro = false;
          if constexpr (!CONES) lone_ro = (P.lone != 0) && rows_in(bmo) == 1;
          if (lone_ro) {  // one row begins a solve: its rollout runs over all four DPP rows (rollout_lone)
            if constexpr (!CONES) {
              const LoneCtx ctx = lone_enter(first_row(bmo));
              r0 = rollout_lone<true>();
              lone_leave(ctx);
            }
          } else {
            const LoneCtx sh = shadow_enter(ob);
            r0 = rollout<true>(ob, ob_shift, ob);
            lone_leave(sh);
          }
          ALTRO_STAMP(t_ro += stamp() - ts;)
          if (ob) {
            const int outer = rs->outer;
            const bool last = (outer == n_outer - 1);
            rs->last = last ? 1 : 0;
            rs->cost_tol = (!last && has_con) ? o.cost_tolerance_intermediate : o.cost_tolerance;
            rs->grad_tol = (!last && has_con) ? o.gradient_tolerance_intermediate : o.gradient_tolerance;
            rs->rho = o.bp_reg_initial;
            rs->drho = 0.0;
            rs->dj_zero = 0;
            rs->it = 0;
            rs->qvalid = CONES ? 0 : 1;  // the open-loop rollout left l_z and the active-set hash of its trajectory
            rs->shift = 0;
            rs->nro += 1;
            rs->J_prev = r0.J;
            rs->J = r0.J;
            rs->cmax = r0.cmax;
            rs->phase = PH_ITER;
            if (r0.limit) {
              rs->status = ALTRO_STATE_LIMIT;
              rs->J = __builtin_inf();
              rs->cmax = __builtin_inf();
              rs->it = o.iterations_inner;  // skip the inner loop: falls through to the AL update
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();

      // ---------------- C. one iLQR iteration for the rows inside their inner loop
      bool upd = false;  // rows that need a dual update after this turn
      {
        // ... and for the rest of the launch once the wave trails the two-turns-per-step pace by ALTRO_PRIO_LAG turns:
        // its partner on the SIMD has that much slack, the launch does not
        turns++;
#ifdef ALTRO_PHASE_STAMPS
        if ((int)blockIdx.x == P.dbg_wave && turns < 4096 && (lane & 15) == 0) {
          // per turn and row: phase | it << 4 | iters << 12 | step << 20 | outer << 28
          long long* tr = P.wave_cycles + (size_t)gridDim.x * 16 + turns;
          const long long code = (long long)rs->phase | ((long long)(rs->it & 255) << 4) | ((long long)(rs->iters & 255) << 12) | ((long long)(rs->step & 255) << 20) | ((long long)(rs->outer & 15) << 28) |
                                 (((stamp() - t_start) >> 10) << 32);   // k-ticks since the wave started
          // four rows packed 16 bits apart would overflow: one word per row, interleaved
          P.wave_cycles[(size_t)gridDim.x * 16 + (size_t)(turns & 1023) * 4 + (lane >> 4)] = code;
          (void)tr;
        }
#endif
        const RowState* r0 = rs - (lane >> 4);
        int ms = r0[0].step;
        sfor<1, IPW>([&](auto q) { ms = imin(ms, r0[decltype(q)::value].step); });
        hard_wave = wave_any((rs->phase == PH_ITER) && (rs->iters >= 2)) || (ALTRO_PRIO_LAG > 0 && turns - 2 * ms >= ALTRO_PRIO_LAG + 2);
        if (simd_row != nullptr && mpc) hard_wave = mate_rank(nsteps - imin(ms, nsteps), ms);
      }
      prio_base();
      {
        bool inner = (rs->phase == PH_ITER) && (rs->it < o.iterations_inner) && (rs->status <= ALTRO_SOLVE_SUCCEEDED);
        bool inner_end = (rs->phase == PH_ITER) && !inner;  // loop exhausted / aborted before this turn
        if (wave_any(inner)) {
          double dV1 = 0.0, dV2 = 0.0;
          // Confirmation by the costate sweep (default mode, box-only kernels; altro_opts.strict = 1 never takes it).
          // From the second iteration of an inner solve on, the cheap first-order sweep adjoint() is tried first: if
          // the gradient of the AL cost along the trajectory is at rounding level AND the active set is the one the
          // previous backward pass saw, that pass's gains are exactly what the reference's confirmation pass would
          // recompute and its feedforward terms are zero to 0.5e-9 (1 + |u|): the iteration is booked as converged
          // without a backward pass, a rollout or a gradient sweep (see `confirm` below for what the reference does
          // in such an iteration).  Any other outcome falls through to the full iteration.
          bool gconf = false, fo = false, dtiny = false;
          if constexpr (!CONES) {
            // kvalid: the gains in memory ARE the ones a backward pass would compute now -- same active set (hash of the
            // current trajectory, left by the rollout that produced it, against the hash of the pass that wrote KD), same
            // penalty, no regularisation.  The problem is quadratic inside an active set, K does not depend on the iterate;
            // the pass may be one of an earlier solve or an earlier launch (gain reuse).
            const bool same = !row_any(aset_ne(qhs, ah), lane) && (P.N <= ASET_MAXN);
            const bool kvalid = !o.strict && inner && (rs->qvalid != 0) && same && (rs->rho == 0.0) && (rs->kmu == rs->mu);
            const bool tryg = kvalid && (rs->it >= 1) && (rs->grad_tol > 1e-8) && (rs->cost_tol > 1e-10 * (1.0 + fabs(rs->J_prev)));
            if (wave_any(tryg)) {
              bool gt;
              ALTRO_STAMP(long long ts = stamp();)
              const LoneCtx sh = shadow_enter(tryg);
              adjoint(gt);
              lone_leave(sh);
              ALTRO_STAMP(t_aj += stamp() - ts; c_aj++;)
              gconf = tryg && gt;
            }
            // ... and an iteration the costate sweep does not settle still skips its backward pass: fosweep() gets the
            // feedforward terms and dV for the gains in memory
            fo = kvalid && !gconf && (P.reuse != 0);
            if (wave_any(fo)) {
              double f1, f2;
              bool ft;
              ALTRO_STAMP(long long ts = stamp();)
              const LoneCtx sh = shadow_enter(fo);
              fosweep(fo, f1, f2, ft);
              lone_leave(sh);
              ALTRO_STAMP(t_fo += stamp() - ts; c_fo++;)
              if (fo) {
                dV1 = f1;
                dV2 = f2;
                dtiny = ft;
                rs->nfo += 1;
              }
            }
          }
          bool bwrow = inner && !gconf && !fo;  // rows that run the backward pass of this iteration
          if (bwrow) rs->nbw += 1;  // once per iteration (a pass restarted with more regularisation is the same iteration's pass)
          // backward pass (with regularisation restarts)
          while (wave_any(bwrow)) {
            bool fail;
            ALTRO_STAMP(long long ts = stamp();)
            const bool with_rho = wave_any(rs->rho != 0.0);
            const unsigned long long bm = __ballot(bwrow);
            const int nbwr = (int)((bm & 1ull) + ((bm >> 16) & 1ull) + ((bm >> 32) & 1ull) + ((bm >> 48) & 1ull));
            if (!CONES && P.lone && !with_rho && nbwr == 1) {
              // exactly one row needs the pass: all four DPP rows work on that row's instance (backward_lone)
              const LoneCtx ctx = lone_enter(first_row(bm));
              double a1, a2;
              bool dt;
              if constexpr (!CONES) {
                const bool useq = (P.useqz != 0) && (rs->qvalid != 0) && (P.N <= ASET_MAXN);   // (all lanes: the lone row's state)
                if (o.strict) { if (useq) backward_lone<true, true>(a1, a2, fail, dt); else backward_lone<true, false>(a1, a2, fail, dt); }
                else { if (useq) backward_lone<false, true>(a1, a2, fail, dt); else backward_lone<false, false>(a1, a2, fail, dt); }
              }
              lone_leave(ctx);
              if (bwrow) {
                dV1 = a1;
                dV2 = a2;
                dtiny = dt;
              }
              n_lone++;
            } else {
              double b1, b2;
              bool bt;
              const LoneCtx sh = shadow_enter(bwrow);
              // the Qz form of the pass: every row that takes part holds a trajectory a rollout has just produced
              bool useq = false;
              if constexpr (!CONES) useq = (P.useqz != 0) && !with_rho && (P.N <= ASET_MAXN) && !wave_any(bwrow && (rs->qvalid == 0));
              if (o.strict) {
                if (with_rho) backward<true, true>(b1, b2, fail, bt, bwrow);
                else if constexpr (!CONES) { if (useq) backward<false, true, true>(b1, b2, fail, bt, bwrow); else backward<false, true>(b1, b2, fail, bt, bwrow); }
                else backward<false, true>(b1, b2, fail, bt, bwrow);
              } else {
                if (with_rho) backward<true, false>(b1, b2, fail, bt, bwrow);
                else if constexpr (!CONES) { if (useq) backward<false, false, true>(b1, b2, fail, bt, bwrow); else backward<false, false>(b1, b2, fail, bt, bwrow); }
                else backward<false, false>(b1, b2, fail, bt, bwrow);
              }
              lone_leave(sh);
              if (bwrow) {
                dV1 = b1;
                dV2 = b2;
                dtiny = bt;
              }
            }
            ALTRO_STAMP(const long long te = stamp() - ts; if (nbwr == 1 && !CONES && P.lone && !with_rho) t_bl += te; else { t_bw += te; c_bw++; })
            fail = row_any(fail, lane) && bwrow;
            double rho = rs->rho, drho = rs->drho;
            if (bwrow) rs->kmu = (!fail && rho == 0.0) ? rs->mu : -1.0;
            if (fail) {
              if (rho >= o.bp_reg_max) {
                rs->status = ALTRO_NO_PROGRESS;
                inner = false;
                bwrow = false;
                inner_end = true;
              } else {
                reg_update(rho, drho, o, true);
              }
            }
            const bool again_bp = wave_any(fail && inner);
            if (!again_bp && !fail && bwrow) reg_update(rho, drho, o, false);
            rs->rho = rho;
            rs->drho = drho;
            __builtin_amdgcn_wave_barrier();
            if (!again_bp) break;
          }
          // forward pass: line search on alpha (forwardpass!).  Trial 0 (alpha = 1) is
          // the closed-loop rollout; later trials are evaluated by interpolation, NA per sweep.
          const double J_prev = rs->J_prev;
          double alpha = 1.0, zr = -1.0, Jn = __builtin_inf(), cm_n = rs->cmax;
          int ls = 0, ntr = 0;
          // Confirmation iterations (default mode only; altro_opts.strict = 1 runs them in full).  The LAST iteration
          // of nearly every warm solve only confirms convergence: the problem is quadratic inside an active set, so
          // the backward pass at the point the previous step reached returns feedforward terms at rounding level
          // (|d| ~ 1e-12).  The reference then rolls out Z + O(|d|), finds |dJ| ~ 1e-13 and stops -- with the step
          // accepted or after 20 fruitless halvings, whichever way the rounding of J falls.  When every |d_k,a| is
          // below 1e-9 (1 + |u_k,a|) that rollout, its line search and the Todorov sweep cannot change the outcome
          // (same status and iteration count, trajectory within ~1e-8, dJ and gradient far below the tolerances in
          // force), so the iteration is booked as converged on the trajectory it already holds.
          const bool confirm = gconf || (!o.strict && (bwrow || fo) && dtiny && (rs->grad_tol > 1e-8) &&
                                         (rs->cost_tol > 1e-10 * (1.0 + fabs(J_prev))));
          bool searching = inner && !confirm, accepted = false, need_interp = false, ls_failed = false;
          if (confirm) {
            Jn = J_prev;
            alpha = 1.0;
          }
          auto trial = [&](double a_t, double J_t, double cm_t, bool lim_t, bool unch_t, bool tiny_t) {
            if (lim_t) {
              ls++;
              alpha = 0.5 * a_t;
              return;
            }
            Jn = J_t;
            const double expected = -a_t * (dV1 + a_t * dV2);
            zr = (expected > 0.0) ? (J_prev - Jn) / expected : -1.0;
            ls++;
            const bool again = ((zr <= o.line_search_lower_bound) || (zr > o.line_search_upper_bound)) &&
                               (Jn >= J_prev);
            if (!again) {
              searching = false;
              accepted = true;
              cm_n = cm_t;
              alpha = a_t;
            } else {
              alpha = 0.5 * a_t;
              // Exact early-out: if this trial reproduced the current trajectory bit for bit, every
              // smaller alpha reproduces it too (round-to-nearest is monotone), J stays == J_prev,
              // and the reference loop would spin to iterations_linesearch and fail.  Jump there.
              if (unch_t) ls = o.iterations_linesearch + 1;
              // Pointless search: the full step moves no element by more than 1e-7 (1 + |z|) and the
              // quadratic model promises less than cost_tol / 1000.  Whatever a smaller alpha would
              // do, |dJ| stays below cost_tol and Z within 1e-7 of where it is, so the iteration ends
              // the same way (converged, same count) as after the reference's 20 more trials.
              // (altro_opts.strict = 1 switches this shortcut off: the search then runs its 20 halvings.)
              if (!o.strict && tiny_t && !(expected > 1e-3 * rs->cost_tol)) ls = o.iterations_linesearch + 1;
            }
          };
          if (wave_any(searching)) {
            ALTRO_STAMP(long long ts = stamp();)
            const unsigned long long bms = __ballot(searching);
            RollOut rr;
            bool lone_ro = false;
            if constexpr (!CONES) lone_ro = (P.lone != 0) && rows_in(bms) == 1;
            if (lone_ro) {
              if constexpr (!CONES) {
                const LoneCtx ctx = lone_enter(first_row(bms));
                rr = rollout_lone<false>();
                lone_leave(ctx);
              }
            } else {
              const LoneCtx sh = shadow_enter(searching);
              rr = rollout<false>(true, false, searching);
              lone_leave(sh);
            }
            ALTRO_STAMP(t_rc += stamp() - ts; c_rc++;)
            if (searching) {
              rs->nro += 1;
              trial(1.0, rr.J, rr.cmax, rr.limit, rr.unchanged, rr.tiny);
              rs->qvalid = (accepted && alpha == 1.0) ? 1 : 0;  // a smaller step (or none) leaves Qz describing a rejected trial
            }
          }
          while (true) {
            const bool failnow = searching && (ls > o.iterations_linesearch);
            if (failnow) {
              Jn = J_prev;
              cm_n = rs->cmax;
              alpha = 0.0;
              ls_failed = true;
              searching = false;
            }
            if (!wave_any(searching)) break;
            ALTRO_STAMP(long long ts = stamp();)
            Trials T;
            unsigned long long sb = __ballot(searching);
            if (P.lone && rows_in(sb) < IPW) {
              // fewer than four rows are searching: their sweeps run one after the other, each over all four DPP rows
              // (a quarter of the knots per row), which takes rows/4 of the time of one row-parallel sweep
              while (sb != 0ull) {
                const int lrow = first_row(sb);
                const double a_l = row_value(alpha, lrow);
                const LoneCtx ctx = lone_enter(lrow);
                Trials Tl;
                trial_costs<true>(a_l, Tl);
                lone_leave(ctx);
                if ((lane >> 4) == lrow) T = Tl;
                sb &= ~(0xFFFFull << (16 * lrow));
              }
            } else {
              trial_costs<false>(alpha, T);
            }
            ALTRO_STAMP(t_ls += stamp() - ts; c_ls++;)
            const double a0 = alpha;
            sfor<0, NA>([&](auto t) {
              constexpr int Tt = decltype(t)::value;
              if (searching && ls <= o.iterations_linesearch) {
                ntr++;
                trial(a0 * (1.0 / (double)(1 << Tt)), T.J[Tt], T.cmax[Tt], T.limit[Tt], T.unchanged[Tt], false);
                if (accepted && !searching) need_interp = true;
              }
            });
          }
          if (wave_any(need_interp)) {
            ALTRO_STAMP(long long ts = stamp();)
            unsigned long long ib = __ballot(need_interp);
            if (P.lone && rows_in(ib) < IPW) {
              while (ib != 0ull) {
                const int lrow = first_row(ib);
                const double a_l = row_value(alpha, lrow);
                const LoneCtx ctx = lone_enter(lrow);
                interpolate<true>(a_l, true);
                lone_leave(ctx);
                ib &= ~(0xFFFFull << (16 * lrow));
              }
            } else {
              interpolate<false>(alpha, need_interp);
            }
            ALTRO_STAMP(t_ls += stamp() - ts;)
          }
          // ---- bookkeeping of this iteration (record_iteration!, evaluate_convergence)
          bool cand = false;
          if (inner) {
            rs->ntr += ntr;
            rs->gconf = gconf ? 1 : 0;
            rs->ngc += gconf ? 1 : 0;
            if (ls_failed) {
              double rho = rs->rho, drho = rs->drho;
              reg_update(rho, drho, o, true);
              rho += o.bp_reg_fp;
              rs->rho = rho;
              rs->drho = drho;
            }
            if (Jn > o.max_cost_value) {
              rs->status = ALTRO_MAXIMUM_COST;
              rs->J = Jn;
              inner = false;
              inner_end = true;
            } else {
              if (accepted) rs->cur = rs->cur ^ 1;  // copy_trajectories!
              rs->cmax = cm_n;
              const double dJ = fabs(Jn - J_prev);
              rs->J_prev = Jn;
              rs->J = Jn;
              const int it_total = rs->iters;
              if (it_total < ALTRO_TRACE_LEN && j == 0) {
                P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + it_total] = Jn;
                P.ctrace[(size_t)inst * ALTRO_TRACE_LEN + it_total] = cm_n;
                P.atrace[(size_t)inst * ALTRO_TRACE_LEN + it_total] = alpha;
              }
              rs->iters = it_total + 1;
              rs->it += 1;
              rs->dj_zero = (dJ == 0.0) ? rs->dj_zero + 1 : 0;
              cand = dJ < rs->cost_tol;
            }
          }
          __builtin_amdgcn_wave_barrier();
          // evaluate_convergence: (0 <= dJ < cost_tol) && grad < grad_tol -- the Todorov gradient
          // is only evaluated for waves that hold a candidate
          double grad = confirm ? 0.0 : __builtin_inf();
          if (wave_any(cand && !confirm)) {
            ALTRO_STAMP(long long ts = stamp();)
            const LoneCtx sh = shadow_enter(cand && !confirm);
            grad = todorov();
            lone_leave(sh);
            ALTRO_STAMP(t_td += stamp() - ts;)
          }
          if (inner) {
            if (cand && grad < rs->grad_tol) {
              inner_end = true;
            } else if (rs->iters >= o.iterations) {
              rs->status = ALTRO_MAX_ITERATIONS;
              inner_end = true;
            } else if (rs->dj_zero > o.dJ_counter_limit) {
              rs->status = ALTRO_NO_PROGRESS;
              inner_end = true;
            } else if (rs->it >= o.iterations_inner) {
              inner_end = true;
            }
          }
        }
        // ---------------- D. AL outer update for rows whose inner loop has ended
        if (inner_end) {
          const int status = rs->status;
          const double cmax = rs->cmax;
          if (has_con) rs->iters_outer += 1;
          bool finished = false;
          if (!has_con) {
            if (status == ALTRO_UNSOLVED) rs->status = ALTRO_SOLVE_SUCCEEDED;
            rs->cmax = 0.0;
            finished = true;
          } else if (status > ALTRO_SOLVE_SUCCEEDED) {
            finished = true;
          } else if (cmax < o.constraint_tolerance || (o.kickout_max_penalty && rs->mu >= o.penalty_max)) {
            finished = true;
          } else if (rs->last) {
            rs->status = ALTRO_MAX_ITERATIONS_OUTER;
            finished = true;
          } else {
            upd = true;
          }
          if (finished) {
            if (has_con && rs->status <= ALTRO_SOLVE_SUCCEEDED && rs->cmax < o.constraint_tolerance)
              rs->status = ALTRO_SOLVE_SUCCEEDED;
            rs->nsolve += 1;
            rs->nit += rs->iters;
            rs->nok += (rs->status == ALTRO_SOLVE_SUCCEEDED) ? 1 : 0;
            rs->step += 1;
            rs->phase = PH_STEP_BEGIN;
          } else {
            rs->outer += 1;
            rs->phase = PH_OUTER_BEGIN;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (wave_any(upd)) {
        ALTRO_STAMP(long long ts = stamp();)
        const LoneCtx sh = shadow_enter(upd);
        dual_update(upd);
        lone_leave(sh);
        ALTRO_STAMP(t_du += stamp() - ts;)
        if (upd) rs->mu = fmin(fmax(phi * rs->mu, 0.0), o.penalty_max);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  __device__ void finish() {
    if (!P.prepare_only) P.ahash[(unsigned)inst * LW + j] = *ah;
    if (j == 0 && !P.prepare_only) {  // a prepare-only launch leaves the statistics of the last solve alone
      P.iters[inst] = rs->iters;
      P.iters_outer[inst] = rs->iters_outer;
      P.status[inst] = rs->status;
      P.cost[inst] = rs->J;
      P.cmax[inst] = rs->cmax;
      P.mu[inst] = rs->mu;
      P.cur[inst] = rs->cur;
      P.n_backward[inst] += rs->nbw;
      P.n_rollout[inst] += rs->nro;
      P.n_trials[inst] += rs->ntr;
      P.n_solves[inst] += rs->nsolve;
      P.n_iters[inst] += rs->nit;
      P.n_ok[inst] += rs->nok;
      P.n_gconf[inst] += rs->ngc;
      P.n_fo[inst] += rs->nfo;
      P.dzero[inst] = rs->gconf;
      P.kmu[inst] = rs->kmu;
    }
  }
};

// nsteps == 0: solve!(altro).  nsteps > 0: that many consecutive MPC steps of every instance,
// each in the reference's order (random_linear_problem.jl:125-139,161): plant step + noise -> x0;
// reference window <- step+1; shift_fill primal and dual; solve.
template <int NX, int NU, bool CONES>
// (NU > 4: the gain rows of the first-order sweep and of the closed-loop rollout outgrow 256 registers -- (6,6): 188 spilled
//  VGPRs at two waves per SIMD, none at one)
__global__ void __launch_bounds__(64, (CONES || NU > 4) ? 1 : ALTRO_WAVES_PER_SIMD) solve_kernel(SolveParams p) {
  __shared__ double tiles[IPW * LW * (LW + 1)];
  __shared__ RowState rows[IPW];
  __shared__ altro::ASet hashes[192];   // per lane: the pass's active set, the trajectory's, a trash set
  const long long t0 = __builtin_amdgcn_s_memtime();
  Solver<NX, NU, CONES> s(p, rows, tiles, hashes);
  ALTRO_STAMP(s.t_start = t0;)
  s.run(p.nsteps > 0, p.first_step, p.nsteps);
  s.mate_leave();
  s.finish();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    long long* wc = p.wave_cycles + (size_t)blockIdx.x * 16;
    wc[0] = t1 - t0;
    ALTRO_STAMP(wc[1] = s.t_bw; wc[2] = s.t_rc; wc[3] = s.t_ro; wc[4] = s.t_td; wc[5] = s.t_du; wc[6] = s.t_ls; wc[8] = s.t_bl; wc[9] = s.t_fo; wc[10] = s.t_aj;
                wc[11] = s.c_bw; wc[12] = s.c_fo; wc[13] = s.c_aj; wc[14] = s.c_rc; wc[15] = s.c_ls;)
    wc[7] = s.n_lone;
#ifdef ALTRO_PHASE_STAMPS
    {  // which SIMD the wave ran on (HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0): who shares a SIMD with whom
      const unsigned a = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
      const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 0xf;
      wc[4] = (long long)((x << 12) | (((a >> 13) & 7) << 9) | (((a >> 12) & 1) << 8) | (((a >> 8) & 15) << 2) | ((a >> 4) & 3));
    }
#endif
  }
}

}  // namespace altro
