// solve_wide.h -- AL-iLQR for problem sizes the 16-lane kernel (solve_dpp16.h) cannot hold:
// 16 < n + m, n <= 64, m <= 32 (quadruped n = m = 12 with per-knot affine dynamics; the
// state- and control-dimension sweeps of run_random_linear.jl:128-153; the (12,6) horizon sweep).
//
// Mapping: ONE wavefront = ONE MPC instance, sizes are runtime values.  The knot matrices live in
// LDS, zero-padded to multiples of 16, and the three Riccati contractions
//     W = S [A B],   [Qxx Qux' ; Qux Quu] = [A B]' W,   S += Qux' K
// run on the matrix cores as v_mfma_f64_16x16x4_f64 tiles (north_star: "MFMA only when n, m fill a
// tile").  Every product is phrased as C = AT' B with AT and B row-major in LDS, so both operand
// fetches are lane-consecutive ds_read_b64 (A-operand lane l holds AT[k0 + l/16][i0 + l%16]); the
// symmetric S serves as its own transpose.  Leading dimensions are odd, so the column accesses
// of the vector phases (a lane walking down a column) do not collide in the LDS banks.
// Vector work (cost expansion, Quu = L D L', the triangular solves for K and d, rollouts) is
// lane-per-element with wave reductions; rollouts read the dynamics from global memory in the
// ABI's own column-major layout (lane i = row i: coalesced) and trajectories, duals and gains
// use the ABI layouts directly, so this path needs no pack kernels.
//
// Semantics follow oracle/altro_oracle.c statement by statement; the line
// search uses true closed-loop rollouts for every alpha, as the reference does.  In the default mode
// (altro_opts.strict = 0) iterations that only confirm convergence are cut short as in the 16-lane kernels
// (ilqr(): confirmation iterations, line-search early-outs, and for n, m <= 16 the costate sweep).
//
// Three instantiation families:
//   * Solver<MC, SM = true>, n, m <= 16: every matrix of a knot is one 16 x 16 tile; compile-time leading dimensions,
//     the knot's five products chained in registers, L D L' per lane, rollouts and the costate sweep in one
//     16-lane DPP row (rollout_row, grad_pass, adjoint_row);
//   * Solver<MC, false>, one wave per block: runtime tile loops (gemm_tn);
//   * Solver<MC, false>, a cooperative block of four waves where LDS leaves room for one instance per CU (n >= 48):
//     waves 1..3 take strips of the large products on command (coop_exec / coop_helper).
// The lane index is made opaque at the start of every phase (phase_begin): without it LLVM hoists every
// lane-dependent address of the kernel to its entry and the knot loops reload them from scratch.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/altro_batch.h"

namespace altro_wide {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int kMaxN = 64, kMaxM = 32, kMaxP = 64;
// packed size of the factor the costate sweep reads back, for the control-size class of m (m <= 16)
__host__ __device__ inline int wide_fac_size(int m) {
  const int mc = m <= 4 ? 4 : m <= 8 ? 8 : m <= 12 ? 12 : 16;
  return mc * (mc + 1) / 2;
}

struct Params {
  int B, n, m, N, Nt, np, mp, Pn, Pp;
  int ltv, dyn_per_instance;
  const double *A, *Bm, *f;            // column-major blocks [inst or 1][N-1 or 1][n*n | n*m | n]
  const double *wd, *wf, *zmin, *zmax;  // [n+m] stage weights (x dt), [n] terminal, [n+m] box
  int box_k0, box_k1;
  const double* AconT;  // [N][n+m][Pn]: row r of knot k's table is column r
  const double* bcon;   // [N][Pn]
  size_t con_istride;   // 0: the tables are shared by all instances; N*(n+m)*Pn: one table per instance
  size_t bcon_istride;  // 0 or N*Pn
  const int* ctype;     // [N][Pn]: 0 none, 1 equality, 2 inequality, 3 row of a second-order cone
  const int *rowk0, *rowk1;  // [Pn] knot range of the constraint block that owns row r
  const int *rowc0, *rowcp;  // [Pn] second-order cones: first row and dimension (2..4) of the cone that owns row r, 0 if none
  double* x0;           // [B][n]
  const double *Xref, *Uref;  // [B][Nt][n], [B][Nt-1][m]
  double *X, *U;        // [B][2][N][n], [B][2][N-1][m]
  int* cur;             // [B]
  double *Lb, *Lc, *mu; // [B][N][2][n+m], [B][N][Pn], [B]
  double *Kg, *dg;      // [B][N-1][n][m] (m x n column-major), [B][N-1][m]
  double* trash;        // [B][64] sink for the stores of lanes that hold no element (keeps loop bodies branch-free)
  int *iters, *iters_outer, *status;
  double *cost, *cmax, *Jtrace, *ctrace, *atrace;
  long long *n_backward, *n_rollout, *n_trials, *n_solves, *n_iters, *n_ok, *n_gconf, *n_gs;
  double* Qz;           // [B][N][n+m] scratch of the costate sweep: gradient of the AL cost at every knot of plane cur
  unsigned* bwst;       // [B][136] gain-reuse state between launches: word 128 bw_ok, 129 bw_plain, 130-131 bw_mu, 132 the roles of the
                        // instance's three active-set planes (words 0..127: unused since the hashes went)
  unsigned char* aset;  // [B][3][N][64] EXACT active sets, one byte per knot and lane (box_code of x_t | box_code of u_t << 2 | row t
                        // active << 4): the plane of the last backward pass, of the trajectory in plane cur, of the trial in work
  int reuse_ok;         // 0: a setter has changed the model / cost / constraints / options since the last launch
  double* fac;          // [B][N][MC (MC + 1) / 2] (n, m <= 16) L D L' factor of Quu_k of the last backward pass: strictly lower
                        // triangle of L and 1 / D on the diagonal, row-major packed (costate sweep)
  const double *noise, *noise_w;
  const int* noise_grp;
  int noise_mode, mpc_shift;
  int ncone;  // number of second-order cones among the generic rows
  int con_static;  // nonzero: no constraint block has per-knot data, the row table is the same at every knot of a row's
                   // range and stays in LDS (bits: 1 rollouts, 2 row values of expansion / dual update, 4 expansion tables)
  int kref;
  // per-knot dynamics: every instance owns dyn_blocks knot blocks; window knot k of a solve whose reference
  // window starts at kref reads block kref * dyn_step_stride + k.  altro_batch_set_dynamics(per_knot) gives
  // dyn_blocks = N - 1, stride 0; altro_mpc_set_dynamics_track a long table with stride 1 (blocks indexed by
  // absolute knot, like the reference track) or N - 1 (one full table per MPC step).
  int dyn_blocks, dyn_step_stride;
  int compact;  // LDS carve-up with [Qux | Qu] and K inside W (wide_compact)
  altro_opts o;
};

__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }

// LDS carve-up (offsets in doubles)
struct Lds {
  int G, S, W, Hux, Kl, Huu, Ac, DA, vec, cmd, total;
  int ldg, lds, ldh, ldu;
};
// compact: [Qux | Qu] and K live in the two halves of W.  W = S [A B] is dead once the three products that read it have
// issued, and the one-wave backward pass runs Qux = B' W_A as the LAST of them (one strip: every read of W precedes the
// stores of the strip, and LDS instructions of a wave execute in order).  At n = 17..32, m <= 16 the carve-up falls
// from 51.6 KB to 39.0 KB: four one-wave blocks per CU instead of three.
__host__ __device__ inline bool wide_compact(int n, int m, int ltv, int np_max) {
  const int np = (n + 15) & ~15, mp = (m + 15) & ~15;
  return !(n <= 16 && m <= 16) && !ltv && mp == 16 && np >= 32 && np <= np_max && np <= 48;  // (n = 64 stays a cooperative block: its helper waves read W concurrently)
}
__host__ __device__ inline Lds lds_layout(int n, int m, int Pn, int compact = 0) {
  Lds L;
  const int np = pad16(n), mp = pad16(m), nzp = np + mp, Pp = pad4(Pn);
  L.ldg = nzp + 1;
  L.lds = np + 1;
  L.ldh = np + 17;
  L.ldu = mp + 1;
  int o = 0;
  L.G = o; o += np * L.ldg;
  L.S = o; o += np * L.lds;
  L.W = o; o += np * L.ldg;
  if (compact) {  // 2 mp ldh <= np ldg for mp = 16, np >= 32
    L.Hux = L.W;
    L.Kl = L.W + mp * L.ldh;
  } else {
    L.Hux = o; o += mp * L.ldh;
    L.Kl = o; o += mp * L.ldh;
  }
  L.Huu = o; o += mp * L.ldu;
  L.Ac = o; o += Pp * L.ldg;
  L.DA = o; o += Pp * L.ldg;
  L.vec = o; o += 6 * nzp + 2 * np + 8 * (Pp + 4);
  L.cmd = o; o += 32;  // command block of the cooperative products (CoopCmd)
  L.total = o;
  return L;
}

// One wave per block: LDS instructions of a wave execute in issue order, so a cross-lane hand-over
// through LDS only needs the COMPILER to keep the order.  __syncthreads() would also drain vmcnt(0),
// i.e. wait for every outstanding global store of the knot, three times per knot.
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// phase end inside the solver wave: every global store of the phase has completed and is visible to the other
// lanes (__syncthreads() minus the s_barrier: the helper waves of a cooperative block do not take part)
__device__ __forceinline__ void block_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ bool wave_any(bool f) { return __ballot(f) != 0ull; }
// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1
template <int B, int E, typename F>
__device__ __forceinline__ void wfor(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    wfor<B + 1, E>(f);
  }
}

// C[M x Nn] (ldc) = (ACC ? C : 0) + scale * sum_k AT[k][i] B[k][j];  M, Nn multiples of 16, K of 4.
// A strip of NT output tiles shares the A fragment; the fragments of k-step s+1 are requested before
// the MFMAs of step s issue, so the matrix pipe sees NT independent accumulator chains and the LDS
// latency is off the critical path (one tile at a time ran at ~250 cycles per MFMA instead of 64).
typedef __attribute__((address_space(3))) double lds_d;

// Built with -mllvm -amdgpu-mfma-vgpr-form=1 (see _lib.py): by default hipcc (ROCm 7.2) put the
// accumulators in AGPRs but carried them through the k loop in VGPRs, i.e. 64 v_accvgpr moves
// around the 4 MFMAs of every k-step.  Pointers are typed as LDS so the loads are ds_read.
template <bool ACC, int NT>
__device__ __forceinline__ void gemm_strip(lds_d* C, int ldc, const lds_d* ap, int lda, const lds_d* bp, int ldb, int K, double scale) {
  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
  double a = ap[0], b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b[t] = bp[16 * t];
  for (int k0 = 4; k0 <= K; k0 += 4) {
    double an, bn[NT];
    const int kk = k0 < K ? k0 : 0;  // the last step re-reads step 0 (harmless): the loop body stays branch-free
    an = ap[kk * lda];
#pragma unroll
    for (int t = 0; t < NT; ++t) bn[t] = bp[kk * ldb + 16 * t];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[t], 0, 0, 0);
    a = an;
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = bn[t];
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg; C already points at
  // (row lane >> 4, column lane & 15) of the strip
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lds_d* c = C + 4 * r * ldc + 16 * t;
      const double v = scale * acc[t][r];
      *c = ACC ? (*c + v) : v;
    }
}

// one strip of 16 output rows: Cl, Al, Bl already point at this lane's element of the strip's first tile
template <bool ACC>
__device__ __forceinline__ void gemm_rows(lds_d* Cl, int ldc, const lds_d* Al, int lda, const lds_d* Bl, int ldb, int Nn, int K, double scale) {
  // tiles of four share an A fragment; a single left-over tile rides with the last four (a lone tile is one
  // dependent MFMA chain, ~250 cycles per k-step against 5 x 32 for a strip of five: W = S [A B] at m <= 16 is
  // np/16 tiles of A and ONE of B)
  int j0 = 0;
  int t = Nn >> 4;
  for (; t >= 6 || t == 4; t -= 4, j0 += 64) gemm_strip<ACC, 4>(Cl + j0, ldc, Al, lda, Bl + j0, ldb, K, scale);
  if (t == 5) gemm_strip<ACC, 5>(Cl + j0, ldc, Al, lda, Bl + j0, ldb, K, scale);
  else if (t == 3) gemm_strip<ACC, 3>(Cl + j0, ldc, Al, lda, Bl + j0, ldb, K, scale);
  else if (t == 2) gemm_strip<ACC, 2>(Cl + j0, ldc, Al, lda, Bl + j0, ldb, K, scale);
  else if (t == 1) gemm_strip<ACC, 1>(Cl + j0, ldc, Al, lda, Bl + j0, ldb, K, scale);
}

template <bool ACC>
__device__ __forceinline__ void gemm_tn(double* C, int ldc, const double* AT, int lda, const double* Bq, int ldb, int M,
                                        int Nn, int K, double scale = 1.0) {
  const int l = threadIdx.x & 63, r16 = l & 15, q = l >> 4;
  lds_d* Cl = (lds_d*)C + q * ldc + r16;
  const lds_d* Al = (const lds_d*)AT + q * lda + r16;
  const lds_d* Bl = (const lds_d*)Bq + q * ldb + r16;
  for (int i0 = 0; i0 < M; i0 += 16) gemm_rows<ACC>(Cl + i0 * ldc, ldc, Al + i0, lda, Bl, ldb, Nn, K, scale);
}

// ---- cooperative products: a block of four waves per instance --------------------------------------------------
// With n >= 48 the LDS carve-up leaves room for ONE instance per CU, i.e. one wave on one of the CU's four SIMDs,
// and the knot's large products (W = S [A B], Qxx = A' W, S += Qux' K: 64 x 64 x 64 at n = 64) ran on that SIMD's
// matrix pipe alone.  Such problems are launched with 256 threads: wave 0 is the solver, waves 1..3 wait at a
// barrier for a command in LDS -- a list of products, or the symmetrisation of S -- take their share of the 16-row
// strips, and wait again.  Every command is two s_barrier for all four waves; wave 0 sends QUIT when it is done.
struct GemmDesc {
  int C, ldc, A, lda, B, ldb, M, Nn, K, acc, w0, wn;  // operands as offsets (doubles) from the LDS base; strip st runs on wave w0 + st % wn
  double scale;
};
struct CoopCmd {
  int op, ng;  // op: 0 quit, 1 products, 2 S <- (S + S')/2 (g[0].C, g[0].ldc, g[0].M = n)
  GemmDesc g[3];
};
constexpr int kCoopQuit = 0, kCoopGemm = 1, kCoopSym = 2;

__device__ __forceinline__ void coop_exec(double* lds, const CoopCmd* c, int wv, int nw) {
  const int l = threadIdx.x & 63, r16 = l & 15, q = l >> 4;
  if (c->op == kCoopGemm) {
    for (int g = 0; g < c->ng; ++g) {
      const GemmDesc d = c->g[g];
      lds_d* Cl = (lds_d*)lds + d.C + q * d.ldc + r16;
      const lds_d* Al = (const lds_d*)lds + d.A + q * d.lda + r16;
      const lds_d* Bl = (const lds_d*)lds + d.B + q * d.ldb + r16;
      for (int st = 0; 16 * st < d.M; ++st) {
        if (((d.w0 + st % d.wn) & (nw - 1)) != wv) continue;  // descriptors that accumulate into the same C give a strip the same wave
        if (d.acc) gemm_rows<true>(Cl + 16 * st * d.ldc, d.ldc, Al + 16 * st, d.lda, Bl, d.ldb, d.Nn, d.K, d.scale);
        else gemm_rows<false>(Cl + 16 * st * d.ldc, d.ldc, Al + 16 * st, d.lda, Bl, d.ldb, d.Nn, d.K, d.scale);
      }
    }
  } else if (c->op == kCoopSym) {
    lds_d* Sl = (lds_d*)lds + c->g[0].C;
    const int ld = c->g[0].ldc, nn = c->g[0].M;
    if (c->g[0].Nn == 64 && nw == 4) {
      // padded 64 x 64, four waves: every lane reads its 16 elements and their mirror images, all waves meet, then the averages
      // are written (a + b = b + a: an element and its mirror get the same bits as from the loop below; the pad is zero and
      // stays zero).  The loop below pays an LDS round trip in each of its 16 rows per wave.
      double own[16], mir[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = l + 64 * (wv + 4 * u), i = e >> 6, j = e & 63;
        own[u] = Sl[i * ld + j];
        mir[u] = Sl[j * ld + i];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = l + 64 * (wv + 4 * u), i = e >> 6, j = e & 63;
        Sl[i * ld + j] = 0.5 * (own[u] + mir[u]);
      }
      return;
    }
    for (int i = wv; i < nn; i += nw)
      for (int j = l; j < i; j += 64) {
        const double v = 0.5 * (Sl[i * ld + j] + Sl[j * ld + i]);
        Sl[i * ld + j] = v;
        Sl[j * ld + i] = v;
      }
  }
}

__device__ __forceinline__ void coop_barrier() { __syncthreads(); }

// waves 1..3 of a cooperative block
__device__ __forceinline__ void coop_helper(double* lds, int cmd_off) {
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const CoopCmd* slot = (const CoopCmd*)(lds + cmd_off);
  while (true) {
    __syncthreads();
    if (slot->op == kCoopQuit) break;
    coop_exec(lds, slot, wv, nw);
    __syncthreads();
  }
}

#ifdef ALTRO_WIDE_STAMPS
#define WSTAMP(x) x
__device__ __forceinline__ long long wstamp() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#else
#define WSTAMP(x)
#endif
#ifndef ALTRO_WIDE_WAVES_SMALL
// waves per SIMD the m <= 8 instantiations are compiled for.  Measured at 2 (256 registers): 1.1 KB of
// scratch per lane and 10-20 % slower than one wave with 512 registers (n = 16: 0.55 vs 0.60 M solves/s)
#define ALTRO_WIDE_WAVES_SMALL 1
#endif

// MC: control-size class the instantiation is compiled for.  4/8/12/16: m <= MC, Quu is factored in
// registers (factor_solve_lane<MC>); 0: m > 16, the LDS factorisation.  One kernel that switched
// between all of them at run time needed the registers of the largest (512, one wave per SIMD).
// SM: n <= 16 and m <= 16, every matrix of the knot is a single 16 x 16 tile.  The padded sizes and the leading
// dimensions of the LDS carve-up are then compile-time constants: the tile loops of the products collapse into
// straight-line MFMA chains with immediate LDS offsets, and a few dozen wave-uniform values leave the SGPR file
// (the generic kernel spills SGPRs into VGPR lanes -- ~500 v_readlane per knot).
// A register budget per phase (round 4).  The solver below used to be ONE inlined body of ~40 k instructions: the register
// allocation of such a function is global, and with more than 512 values alive somewhere in it the allocator spilled
// 220-670 VGPRs and over 1000 SGPRs -- into the hot loops of phases that need a fraction of the file on their own
// (loop-invariant columns of A reloaded from scratch in front of every FMA, `s_waitcnt vmcnt(0)` behind each).  The phases
// are now real functions (wide_phase<MC, SM, OP>, noinline): each builds its own Solver from the kernel's arguments (the
// pointers and strides are recomputed: ~100 instructions per call) and gets the few values of the solver that change
// (PhIn) as arguments; what it produces comes back in PhOut.  LDS contents (resident dynamics, parked blocks) and the
// arrays in HBM carry everything else.  ALTRO_WIDE_SPLIT=0 compiles the old single body (same arithmetic, same results).
#ifndef ALTRO_WIDE_SPLIT
#define ALTRO_WIDE_SPLIT 1
#endif

#ifndef ALTRO_WIDE_FACTOR_DPP
// control-size classes from this one on factor with L spread over the lanes (factor_solve_dpp), the smaller ones with L whole
// in every lane (factor_solve_lane).  Bit-identical; measured per 30 fused steps: MC = 16 at (30, 15): 389 -> 367 ms, MC = 12
// (quadruped): 422 -> 418, MC = 4 at n = 16: 69.4 -> 71.1 (the broadcasts cost more than the 20 FMAs they save).
#define ALTRO_WIDE_FACTOR_DPP 12
#endif


enum { PH_ROLL_OPEN = 0, PH_ROLL, PH_BACKWARD, PH_ADJ_FULL, PH_ADJ_CONF, PH_GRAD_ADJ_ROW, PH_DUAL, PH_SHIFT, PH_PLANT, PH_GRAD_ADJ_ROW_FULL };
struct PhIn {
  int cur, kref, i0, flags;
  double mu, rho, a;
  unsigned long long h;
};
struct PhOut {
  double a, b;
  unsigned long long h;
  int flags;
#ifdef ALTRO_WIDE_STAMPS
  long long t[5];
#endif
};
template <int MC, bool SM, int OP, int NPC, int PRC>
__device__ PhOut wide_phase(unsigned long long kp, PhIn in);
struct Resume {};

// NPC: the padded state dimension as a compile-time constant (32, 48, 64; 0 = read from Params).  What SM does for n <= 16 --
// leading dimensions, tile counts and trip counts of the product loops known to the compiler -- for the sizes of BASELINE
// configs[3]: measured at n = 32, m = 4, batch 8192, 30 fused steps: 207 -> 158 ms, bit-identical (round 4).
// PRC: the number of generic constraint rows as a compile-time fact (-1 = read from Params).  0 ("plain": box constraints only)
// -- the row tables, their products and the cone code fall away: n = 16: 75.8 -> 69.9 ms per 30 steps, n = 32: 158 -> 153;
// 16 (the quadruped's four friction pyramids) -- the row products and loops get their trip counts: 417 -> 393 ms (bit-identical).
template <int MC, bool SM, int NPC = 0, int PRC = -1>
struct Solver {
  static_assert(!(SM && NPC != 0), "SM fixes the padded size at 16");
  const Params& P;
  unsigned long long kp = 0ull;  // the kernel's argument segment (what a phase function rebuilds P from)
  int T;  // lane index; made opaque again at the start of every phase (phase_begin)
  const int inst, n, m, N, np, mp, nz, nzp, Pn, Pp;
  const Lds ly;
  double *G, *S, *W, *Hux, *Kl, *Huu, *Ac, *DA;
  double *zb, *dxv, *sv, *qz, *hz, *qv, *gr, *Dr, *cvv, *cll, *Hc;
  double* lds_base;
  int myc0 = 0, mycp = 0;
  // e = T + 64 t written as q * d + r without a division per element (an integer division is ~40
  // VALU instructions, and the element loops below ran one per 64 elements per knot): the lane's
  // start (T / d, T % d) and the step (64 / d, 64 % d) are worked out once per launch.
  struct Stride {
    int q0, r0, dq, dr, d;
  };
  struct Walk {
    int q, r;
  };
  Stride by_n, by_m, by_P;
  static __device__ __forceinline__ Stride make_stride(int T, int d) {
    Stride s;
    s.d = d > 0 ? d : 1;
    s.q0 = T / s.d;
    s.r0 = T - s.q0 * s.d;
    s.dq = 64 / s.d;
    s.dr = 64 - s.dq * s.d;
    return s;
  }
  static __device__ __forceinline__ Walk start(const Stride& s) { return Walk{s.q0, s.r0}; }
  static __device__ __forceinline__ void step(const Stride& s, Walk& w) {
    w.q += s.dq;
    w.r += s.dr;
    if (w.r >= s.d) {
      w.r -= s.d;
      ++w.q;
    }
  }
  double *Xi, *Ui, *Lbi, *Lci, *Kgi, *dgi, *x0i;
  const double *AconTi, *bconi;  // this instance's constraint tables (the shared ones unless per-instance data was given)
  const double *Xri, *Uri;
  int cur, kref;
  double mu, rho, drho;
  int dj_zero, status, iters, iters_outer;
  bool dtiny = false;  // backward(): every feedforward term of the pass is at rounding level, |d_k,a| <= 1e-9 (1 + |u_k,a|)
  // costate sweep (adjoint_row) and gain reuse: the active set of the last backward pass and that of the trajectory plane cur
  // holds; bw_plain: that pass ran without regularisation; qvalid: plane qp describes plane cur
  // (rounds 2-3: 32-, then 64-bit hashes per lane.  Round 4: the sets themselves, P.aset -- a byte per knot and lane in HBM, written
  // next to the trajectory by the rollouts and by the pass while it expands; the three planes of an instance change roles by
  // index: bwp = the pass's, qp = that of the trajectory in plane cur, ap = the trial's.  Compared byte for byte: sets_differ.)
  int bwp = 0, qp = 1, ap = 2;
  bool bw_plain = false, qvalid = false;
  bool bw_ok = false;  // Kg and fac hold a backward pass that succeeded (generic class: kept across the solves of a launch)
  double bw_mu = 0.0;  // the penalty it ran at
  long long ngs = 0;   // iterations that took their gains from memory (no backward pass)
  long long ngc = 0;
  long long nbw, nro, ntr;
  long long t_bw = 0, t_ro = 0, t_gemm = 0, t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_du = 0, t_sh = 0, t_run = 0, t_td = 0;
  // per-lane constants (lane T as state element T and as control element T), loaded once
  double cwx = 0.0, cwfx = 0.0, cxmax = __builtin_inf(), cxmin = -__builtin_inf(), cwu = 0.0, cumax = __builtin_inf(), cumin = -__builtin_inf();

  __device__ __forceinline__ Solver(const Params& p, double* lds)
      : P(p), T(threadIdx.x), inst(blockIdx.x), n(p.n), m(p.m), N(p.N), np(SM ? 16 : NPC ? NPC : p.np), mp(SM ? 16 : NPC ? 16 : p.mp), nz(p.n + p.m),
        nzp(SM ? 32 : NPC ? NPC + 16 : p.np + p.mp), Pn(PRC >= 0 ? PRC : p.Pn), Pp(PRC >= 0 ? ((PRC + 3) & ~3) : p.Pp), ly(lds_layout(SM ? 16 : NPC ? NPC : p.n, SM ? 16 : NPC ? 16 : p.m, PRC >= 0 ? PRC : p.Pn, SM ? 0 : p.compact)) {
    init(lds, true);
  }
  // a phase function's view of the same instance: same pointers, LDS as the kernel left it
  __device__ __forceinline__ Solver(const Params& p, double* lds, Resume)
      : P(p), T(threadIdx.x), inst(blockIdx.x), n(p.n), m(p.m), N(p.N), np(SM ? 16 : NPC ? NPC : p.np), mp(SM ? 16 : NPC ? 16 : p.mp), nz(p.n + p.m),
        nzp(SM ? 32 : NPC ? NPC + 16 : p.np + p.mp), Pn(PRC >= 0 ? PRC : p.Pn), Pp(PRC >= 0 ? ((PRC + 3) & ~3) : p.Pp), ly(lds_layout(SM ? 16 : NPC ? NPC : p.n, SM ? 16 : NPC ? 16 : p.m, PRC >= 0 ? PRC : p.Pn, SM ? 0 : p.compact)) {
    init(lds, false);
  }
  __device__ __forceinline__ void init(double* lds, bool fresh) {
    lds_base = lds;
    G = lds + ly.G; S = lds + ly.S; W = lds + ly.W; Hux = lds + ly.Hux; Kl = lds + ly.Kl; Huu = lds + ly.Huu;
    Ac = lds + ly.Ac; DA = lds + ly.DA;
    double* v = lds + ly.vec;
    zb = v; v += nzp; qz = v; v += nzp; hz = v; v += nzp; qv = v; v += nzp; v += 2 * nzp;
    dxv = v; v += np; sv = v; v += np; gr = v; v += Pp + 4; Dr = v; v += Pp + 4; cvv = v; v += Pp + 4; cll = v; v += Pp + 4;
    Hc = v;  // [Pp + 4][4] Hessian rows of the cone blocks
    by_n = make_stride(T, n);
    by_m = make_stride(T, m);
    by_P = make_stride(T, Pn);
    myc0 = (T < Pn) ? P.rowc0[T] : 0;
    mycp = (T < Pn) ? P.rowcp[T] : 0;
    const size_t b = inst;
    Xi = P.X + b * 2 * N * n; Ui = P.U + b * 2 * (N - 1) * m;
    Lbi = P.Lb + b * N * 2 * nz; Lci = P.Lc + b * N * (Pn > 0 ? Pn : 1);
    Kgi = P.Kg + b * (N - 1) * n * m; dgi = P.dg + b * (N - 1) * m;
    x0i = P.x0 + b * n;
    AconTi = P.AconT + b * P.con_istride;
    bconi = P.bcon + b * P.bcon_istride;
    Xri = P.Xref + b * P.Nt * n; Uri = P.Uref + b * (P.Nt - 1) * m;
    if (fresh)
      for (int e = T; e < ly.total; e += 64) lds[e] = 0.0;
    if (T < n) { cwx = P.wd[T]; cwfx = P.wf[T]; cxmax = P.zmax[T]; cxmin = P.zmin[T]; }
    if (T < m) { cwu = P.wd[n + T]; cumax = P.zmax[n + T]; cumin = P.zmin[n + T]; }
    if (fresh) block_sync();
  }

  struct RollOut {
    double J, cmax;
    bool limit;
    bool unchanged;  // closed-loop rollouts: the trial reproduced plane cur bit for bit
    bool tiny;       // closed-loop rollouts: no element moved by more than 1e-7 (1 + |z|)
  };

  __device__ __forceinline__ unsigned char* aset_plane(int pl) const { return P.aset + ((size_t)inst * 3 + (size_t)pl) * (size_t)N * 64; }
  // the trajectory's active set against the pass's, byte for byte
  __device__ __forceinline__ bool sets_differ() const {
    // (any byte anywhere: the planes are read as 16-byte chunks, 1 KB per load instruction of the wave)
    const uint4 *a = reinterpret_cast<const uint4*>(aset_plane(qp)), *b = reinterpret_cast<const uint4*>(aset_plane(bwp));
    unsigned diff = 0u;
    for (int c = T; c < N * 4; c += 64) {
      const uint4 x = a[c], y = b[c];
      diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    return wave_any(diff != 0u);
  }

  // ---- the phases as calls (ALTRO_WIDE_SPLIT) or inline ----
  __device__ __forceinline__ PhIn ph_in(double a = 0.0, int i0 = 0) const {
    PhIn in;
    in.cur = cur; in.kref = kref; in.i0 = i0; in.flags = dtiny ? 1 : 0;
    in.mu = mu; in.rho = rho; in.a = a; in.h = (unsigned long long)(bwp | (qp << 2) | (ap << 4));
    return in;
  }
  __device__ __forceinline__ void ph_stamps(const PhOut& o) {
#ifdef ALTRO_WIDE_STAMPS
    t_gemm += o.t[0]; t_a += o.t[1]; t_b += o.t[2]; t_c += o.t[3]; t_d += o.t[4];
#else
    (void)o;
#endif
  }
  __device__ __forceinline__ RollOut do_rollout(bool open, double alpha) {
    if constexpr (ALTRO_WIDE_SPLIT != 0) {
      const PhOut o = open ? wide_phase<MC, SM, PH_ROLL_OPEN, NPC, PRC>(kp, ph_in()) : wide_phase<MC, SM, PH_ROLL, NPC, PRC>(kp, ph_in(alpha));
      RollOut r;
      r.J = o.a; r.cmax = o.b;
      r.limit = (o.flags & 1) != 0; r.unchanged = (o.flags & 2) != 0; r.tiny = (o.flags & 4) != 0;
      return r;
    } else {
      return rollout(open, alpha);
    }
  }
  __device__ __forceinline__ bool do_backward(double& dV1, double& dV2) {
    if constexpr (ALTRO_WIDE_SPLIT != 0) {
      const PhOut o = wide_phase<MC, SM, PH_BACKWARD, NPC, PRC>(kp, ph_in());
      dV1 = o.a; dV2 = o.b; dtiny = (o.flags & 2) != 0;
      ph_stamps(o);
      return (o.flags & 1) != 0;
    } else {
      return backward(dV1, dV2);
    }
  }
  __device__ __forceinline__ bool do_adjoint_lds(bool full, double& dV1, double& dV2) {
    if constexpr (ALTRO_WIDE_SPLIT != 0) {
      const PhOut o = full ? wide_phase<MC, SM, PH_ADJ_FULL, NPC, PRC>(kp, ph_in()) : wide_phase<MC, SM, PH_ADJ_CONF, NPC, PRC>(kp, ph_in());
      dV1 = o.a; dV2 = o.b;
      return (o.flags & 1) != 0;
    } else {
      return adjoint_lds(full, dV1, dV2);
    }
  }
  __device__ __forceinline__ bool grad_adjoint_row() {
    if constexpr (SM) {
      if (Pn > 0) grad_pass<true>(); else grad_pass<false>();
      return P.ltv ? adjoint_row<true>() : adjoint_row<false>();
    } else {
      return false;
    }
  }
  // gain reuse in the n, m <= 16 instantiation with m <= 4: box constraints only, time-invariant dynamics, row rollouts.
  // Measured over 30 fused steps at batch 8192 (tools/debug/gpu_wide_ab.py): n = 16, m = 4: 81.4 -> 75.8 ms; with m <= 8 as
  // well, (12, 6): 121 -> 128 ms -- the pass of these sizes is a chain of register-resident tiles, and the sweep with six
  // columns of K per lane costs more than the share of passes it replaces.
  static constexpr bool kReuseRow = SM && MC == 4;
  __device__ __forceinline__ bool grad_adjoint_row_full(double& dV1, double& dV2) {
    if constexpr (kReuseRow) {
      grad_pass<false>();
      return adjoint_row<false, true>(&dV1, &dV2);
    } else {
      return false;
    }
  }
  __device__ __forceinline__ bool do_grad_adjoint_row_full(double& dV1, double& dV2) {
    if constexpr (ALTRO_WIDE_SPLIT != 0) {
      const PhOut o = wide_phase<MC, SM, PH_GRAD_ADJ_ROW_FULL, NPC, PRC>(kp, ph_in());
      dV1 = o.a; dV2 = o.b;
      return (o.flags & 1) != 0;
    } else {
      return grad_adjoint_row_full(dV1, dV2);
    }
  }
  __device__ __forceinline__ bool do_grad_adjoint_row() {
    if constexpr (ALTRO_WIDE_SPLIT != 0) return (wide_phase<MC, SM, PH_GRAD_ADJ_ROW, NPC, PRC>(kp, ph_in()).flags & 1) != 0;
    else return grad_adjoint_row();
  }
  __device__ __forceinline__ void do_dual_update() {
    if constexpr (ALTRO_WIDE_SPLIT != 0) (void)wide_phase<MC, SM, PH_DUAL, NPC, PRC>(kp, ph_in());
    else dual_update();
  }
  __device__ __forceinline__ void do_shift() {
    if constexpr (ALTRO_WIDE_SPLIT != 0) (void)wide_phase<MC, SM, PH_SHIFT, NPC, PRC>(kp, ph_in());
    else shift(true, true);
  }
  __device__ __forceinline__ void do_plant_step(int step) {
    if constexpr (ALTRO_WIDE_SPLIT != 0) (void)wide_phase<MC, SM, PH_PLANT, NPC, PRC>(kp, ph_in(0.0, step));
    else plant_step(step);
  }

  // Everything lane-dependent that a phase needs (dozens of addresses and offsets per phase) is loop-invariant for
  // the whole kernel, and LLVM hoists all of it to the kernel's entry: more values than the 512 registers hold, so
  // they went to scratch and every reload in a knot loop (followed by s_waitcnt vmcnt(0)) drained the operands in
  // flight.  Making the lane index opaque at the start of a phase keeps those computations inside the phase.
  __device__ __forceinline__ void phase_begin() { asm volatile("" : "+v"(T)); }

  // cooperative block (four waves, see coop_exec): wave 0 posts a command for the helper waves and takes its share
  __device__ __forceinline__ bool coop_on() const { return !SM && blockDim.x > 64; }
  __device__ __forceinline__ GemmDesc gdesc(bool acc, const double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int Nn,
                                            int K, double scale, int w0, int wn) const {
    GemmDesc d;
    d.C = (int)(C - lds_base); d.ldc = ldc;
    d.A = (int)(A - lds_base); d.lda = lda;
    d.B = (int)(B - lds_base); d.ldb = ldb;
    d.M = M; d.Nn = Nn; d.K = K; d.acc = acc ? 1 : 0; d.w0 = w0; d.wn = wn; d.scale = scale;
    return d;
  }
  __device__ __forceinline__ void coop_run(int op, int ng, const GemmDesc& g0, const GemmDesc& g1, const GemmDesc& g2) {
    CoopCmd* slot = (CoopCmd*)(lds_base + ly.cmd);
    if (T == 0) {
      slot->op = op;
      slot->ng = ng;
      slot->g[0] = g0;
      slot->g[1] = g1;
      slot->g[2] = g2;
    }
    coop_barrier();
    coop_exec(lds_base, slot, 0, (int)(blockDim.x >> 6));
    coop_barrier();
  }
  __device__ __forceinline__ void coop_quit() {
    if (coop_on()) {
      CoopCmd* slot = (CoopCmd*)(lds_base + ly.cmd);
      if (T == 0) slot->op = kCoopQuit;
      coop_barrier();
    }
  }

  __device__ __forceinline__ size_t dynblk(int k) const {
    return (size_t)(P.dyn_per_instance ? inst : 0) * (P.ltv ? P.dyn_blocks : 1) +
           (P.ltv ? (size_t)kref * P.dyn_step_stride + k : 0);
  }
  __device__ __forceinline__ const double* Ak(int k) const { return P.A + dynblk(k) * n * n; }
  __device__ __forceinline__ const double* Bk(int k) const { return P.Bm + dynblk(k) * n * m; }
  __device__ __forceinline__ const double* fk(int k) const { return P.f + dynblk(k) * n; }
  __device__ __forceinline__ bool box_at(int k) const { return k >= P.box_k0 && k <= P.box_k1; }
  __device__ __forceinline__ double* Xp(int pl) const { return Xi + (size_t)pl * N * n; }
  __device__ __forceinline__ double* Up(int pl) const { return Ui + (size_t)pl * (N - 1) * m; }

  // sum_j col[j * stride] * vec[j], j < cnt, with the global loads issued eight at a time (a plain
  // accumulate loop waits a full memory round trip per term at this occupancy).  vec is an LDS
  // vector that is zero beyond cnt up to the next multiple of 8.
  static __device__ __forceinline__ double dot_strided(const double* col, size_t stride, const double* vec, int cnt, double acc) {
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = col[(size_t)(j0 + u < cnt ? j0 + u : cnt - 1) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += a[u] * vec[j0 + u];
    }
    return acc;
  }

  // sum_{j < cnt} a[j * sa] * b[j * sb] for LDS operands, cnt a multiple of 8 (the padded tails are
  // zero): all sixteen reads of a chunk are issued before the first FMA needs one.
  static __device__ __forceinline__ double dot_lds(const double* a, int sa, const double* b, int sb, int cnt, double acc) {
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      double x[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x[u] = a[(j0 + u) * sa];
        y[u] = b[(j0 + u) * sb];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += x[u] * y[u];
    }
    return acc;
  }

  // x+ = A x + B u + f for row T of knot k's dynamics, z = [x; u] in zb (padded layout)
  __device__ __forceinline__ double next_state(int k) const {
    double acc = fk(k)[T];
    if (!P.ltv) {  // time-invariant: [A B] is resident in LDS (row T, odd leading dimension: no bank conflicts)
      return dot_lds(G + T * ly.ldg, 1, zb, 1, nzp, acc);  // pads of G and zb are zero
    }
    acc = dot_strided(Ak(k) + T, n, zb, n, acc);
    return dot_strided(Bk(k) + T, n, zb + np, m, acc);
  }

  // Per-knot dynamics one knot ahead (n, m <= 16).  A wave that asked for knot k's [A_k B_k f_k] when it needed them
  // waited for four dependent batches of strided loads per rollout knot and five per backward knot, ~2 us each from
  // HBM with one wave per SIMD and nothing else to run.  Instead the block of the NEXT knot is requested into nine
  // registers per lane while the current knot is processed and parked in LDS afterwards: the rollouts keep it in
  // global-memory order in W and Hux alternately (both are scratch outside the backward pass), the backward pass
  // writes it into G once the last product that reads G has issued.
  struct DynRegs {
    double a[4], b[4], f;
  };
  __device__ __forceinline__ bool dyn_ahead() const { return P.ltv && n <= 16 && m <= 16; }
  __device__ __forceinline__ DynRegs dyn_request(int k) const {
    const double *A_ = Ak(k), *B_ = Bk(k);
    const int nn = n * n, nm = n * m;
    DynRegs d;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = T + 64 * u;
      d.a[u] = A_[e < nn ? e : nn - 1];
      d.b[u] = B_[e < nm ? e : nm - 1];
    }
    d.f = fk(k)[T < n ? T : n - 1];
    return d;
  }
  __device__ __forceinline__ void dyn_park_linear(const DynRegs& d, lds_d* st) const {
    const int nn = n * n, nm = n * m;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = T + 64 * u;
      if (e < nn) st[e] = d.a[u];
      if (e < nm) st[nn + e] = d.b[u];
    }
    if (T < n) st[nn + nm + T] = d.f;
  }
  __device__ __forceinline__ void dyn_park_G(const DynRegs& d) {
    const int nn = n * n, nm = n * m;
    Walk w = start(by_n);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = T + 64 * u;
      if (e < nn) G[w.r * ly.ldg + w.q] = d.a[u];
      if (e < nm) G[w.r * ly.ldg + np + w.q] = d.b[u];
      step(by_n, w);
    }
  }
  // sum_j col[j * stride] * vec[j] over an LDS column, terms in the order of dot_strided
  static __device__ __forceinline__ double dot_lds_col(const lds_d* col, int stride, const lds_d* vec, int cnt, double acc) {
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = col[(j0 + u < cnt ? j0 + u : cnt - 1) * stride];
        b[u] = vec[j0 + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += a[u] * b[u];
    }
    return acc;
  }
  // next_state(k) from a block parked by dyn_park_linear
  __device__ __forceinline__ double next_state_parked(const lds_d* st) const {
    const int nn = n * n, nm = n * m;
    double acc = st[nn + nm + T];
    acc = dot_lds_col(st + T, n, (const lds_d*)zb, n, acc);
    return dot_lds_col(st + nn + T, n, (const lds_d*)zb + np, m, acc);
  }

  // AL cost of one box-bounded element
  static __device__ __forceinline__ double lane_cost(double w, double z, double zr, double zmx, double zmn, double lhi,
                                                     double llo, double mu, bool box_on, double& viol) {
    const double e = z - zr;
    double J = 0.5 * w * e * e;
    if (box_on && zmx < 1e300) {
      const double c = z - zmx;
      const bool a = (c >= 0.0) || (lhi > 0.0);
      J += lhi * c + (a ? 0.5 * mu * c * c : 0.0);
      viol = fmax(viol, c);
    }
    if (box_on && zmn > -1e300) {
      const double c = zmn - z;
      const bool a = (c >= 0.0) || (llo > 0.0);
      J += llo * c + (a ? 0.5 * mu * c * c : 0.0);
      viol = fmax(viol, c);
    }
    return J;
  }

  // value of generic row r at knot k from zb (padded layout: x at [0,n), u at [np, np+m))
  __device__ __forceinline__ double row_value(int k, int r, bool term) const {
    double v = bconi[(size_t)k * Pn + r];
    if ((P.con_static & 2) && !term) return dot_lds(Ac + r * ly.ldg, 1, zb, 1, nzp, v);  // table resident in LDS
    const double* At = AconTi + (size_t)k * nz * Pn + r;
    v = dot_strided(At, Pn, zb, n, v);
    if (!term) v = dot_strided(At + (size_t)n * Pn, Pn, zb + np, m, v);
    return v;
  }

  // time-invariant constraint data: the table lives in Ac for the whole launch (row r taken from the
  // first knot of its range); rows that are inactive at a knot get zero weights, so they do no harm
  __device__ __forceinline__ void build_static_Ac() {
    Walk w = start(by_P);
    for (int e = T; e < nz * Pn; e += 64, step(by_P, w)) {
      const int j = w.q, r = w.r;
      const int c = j < n ? j : np + (j - n);
      Ac[r * ly.ldg + c] = AconTi[((size_t)P.rowk0[r] * nz + j) * Pn + r];
    }
  }

  // ---- second-order cone rows (oracle soc_project / con_cost / cost_expansion) -------------
  // value v and dual lam of the whole cone are exchanged through LDS (cvv, cll); every lane of the
  // cone evaluates the same small dense block (dimension <= 4) and keeps its own row.
  struct Cone {
    double cost;      // (1/2mu)(||Pi(lam - mu v)||^2 - ||lam||^2), counted once per cone (on its first row)
    double viol;      // |Pi(v) - v| of this row
    double lam_new;   // Pi(lam - mu v) of this row
    double g;         // d phi / d v of this row
    double h[4];      // this row of the Hessian block  mu (JPi'JPi [+ curvature])
  };

  static __device__ __forceinline__ void soc_proj4(const double (&x)[4], int p, double (&out)[4], int& br, double& nv, double& t) {
    const int q = p - 1;
    double s2 = 0.0;
    t = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s2 += u < q ? x[u] * x[u] : 0.0;
      t = u == q ? x[u] : t;
    }
    nv = sqrt(s2);
    br = (nv <= t) ? 0 : ((nv <= -t) ? 1 : 2);
    const double cc = 0.5 * (1.0 + t / nv);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double b2 = u < q ? cc * x[u] : (u == q ? cc * nv : 0.0);
      out[u] = br == 0 ? (u < p ? x[u] : 0.0) : (br == 1 ? 0.0 : b2);
    }
  }

  template <bool HESS>
  __device__ __forceinline__ Cone cone_eval(int r0, int p, int pos) const {
    Cone c;
    double v[4], lam[4], lb[4], lp[4], pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = u < p;
      v[u] = in ? cvv[r0 + u] : 0.0;
      lam[u] = in ? cll[r0 + u] : 0.0;
      lb[u] = lam[u] - mu * v[u];
    }
    int br, brv;
    double nv, t, nvv, tv;
    soc_proj4(lb, p, lp, br, nv, t);
    soc_proj4(v, p, pv, brv, nvv, tv);
    double a2 = 0.0, l2 = 0.0;
    c.viol = 0.0;
    c.lam_new = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a2 += lp[u] * lp[u];
      l2 += lam[u] * lam[u];
      c.viol = u == pos ? fabs(pv[u] - v[u]) : c.viol;
      c.lam_new = u == pos ? lp[u] : c.lam_new;
    }
    c.cost = pos == 0 ? (a2 - l2) / (2.0 * mu) : 0.0;
    c.g = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) c.h[u] = 0.0;
    if constexpr (HESS) {
      const int q = p - 1;
      // Jacobian of the projection (symmetric)
      double Jp[4][4];
      const double cc = 0.5 * (1.0 + t / nv), i3 = 1.0 / (nv * nv * nv);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double e = 0.0;
          if (br == 0) e = (i == j && i < p) ? 1.0 : 0.0;
          else if (br == 2) {
            if (i < q && j < q) e = (i == j ? cc : 0.0) - 0.5 * t * lb[i] * lb[j] * i3;
            else if (i < q && j == q) e = 0.5 * lb[i] / nv;
            else if (i == q && j < q) e = 0.5 * lb[j] / nv;
            else if (i == q && j == q) e = 0.5;
          }
          Jp[i][j] = e;
        }
      double g[4], H[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += Jp[i][r] * lp[i];
        g[r] = -acc;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc += Jp[r][i] * Jp[r][j];
          H[i][j] = mu * acc;
        }
      if (P.o.soc_second_order != 0 && br == 2) {  // curvature of the projection contracted with y = Pi(lb)
        double yv = 0.0, yt = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          yv += u < q ? lp[u] * lb[u] : 0.0;
          yt = u == q ? lp[u] : yt;
        }
        const double i5 = i3 / (nv * nv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < q && j < q) {
              const double dcc = -0.5 * t * lb[j] * i3;
              const double term = dcc * lp[i] - 0.5 * t * ((i == j ? yv : 0.0) + lb[i] * lp[j]) * i3 + 1.5 * t * lb[i] * yv * lb[j] * i5 +
                                  yt * ((i == j ? 1.0 : 0.0) / (2.0 * nv) - lb[i] * lb[j] * 0.5 * i3);
              H[i][j] += mu * term;
            }
          if (i < q) {
            const double dt = mu * (lp[i] / (2.0 * nv) - lb[i] * yv * 0.5 * i3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              H[i][j] += (j == q) ? dt : 0.0;
              H[j][i] += (j == q) ? dt : 0.0;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c.g = r == pos ? g[r] : c.g;
#pragma unroll
        for (int u = 0; u < 4; ++u) c.h[u] = r == pos ? H[r][u] : c.h[u];
      }
    }
    return c;
  }


  // per-lane operands of one rollout knot, requested one knot ahead of their use
  struct KnotLd {
    double xs, us, dgv, xr, ur, lxh, lxl, luh, lul, lam, bc;
    double kp[16];  // K_k[T][0..15] for the control lane T (the rest of a wider K is read on demand)
    int ct;
  };

  // mode 0: closed-loop rollout (state of plane cur, gains); 1: open-loop rollout; 2: backward pass (state, no gains)
  __device__ __forceinline__ KnotLd load_knot(int k, bool term, int mode, const double* Xs, const double* Us) const {
    const bool open = mode == 1, need_k = mode == 0;
    KnotLd d;
    const bool bx = box_at(k);
    d.xs = d.us = d.dgv = d.xr = d.ur = d.lxh = d.lxl = d.luh = d.lul = d.lam = d.bc = 0.0;
    d.ct = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) d.kp[u] = 0.0;
    if (T < n) {
      if (!open) d.xs = Xs[(size_t)k * n + T];
      d.xr = Xri[(size_t)(kref + k) * n + T];
      if (bx) {
        d.lxh = Lbi[((size_t)k * 2 + 0) * nz + T];
        d.lxl = Lbi[((size_t)k * 2 + 1) * nz + T];
      }
    }
    if (!term && T < m) {
      d.us = Us[(size_t)k * m + T];
      d.ur = Uri[(size_t)(kref + k) * m + T];
      if (bx) {
        d.luh = Lbi[((size_t)k * 2 + 0) * nz + n + T];
        d.lul = Lbi[((size_t)k * 2 + 1) * nz + n + T];
      }
      if (need_k) {
        d.dgv = dgi[(size_t)k * m + T];
        const double* Kk = Kgi + (size_t)k * n * m + T;
#pragma unroll
        for (int u = 0; u < 16; ++u) d.kp[u] = Kk[(size_t)(u < n ? u : n - 1) * m];
      }
    }
    if (T < Pn) {
      d.ct = P.ctype[(size_t)k * Pn + T];
      d.lam = Lci[(size_t)k * Pn + T];
      d.bc = bconi[(size_t)k * Pn + T];
    }
    return d;
  }

  // cost!(obj, z_k) + max_violation of knot k; zb holds [x; u] of the knot (synchronised)
  __device__ __forceinline__ void eval_knot(int k, bool term, double xv, double uv, const KnotLd& d, double& J, double& viol) const {
    const bool bx = box_at(k);
    if (T < n) J += lane_cost(term ? cwfx : cwx, xv, d.xr, cxmax, cxmin, d.lxh, d.lxl, mu, bx, viol);
    if (!term && T < m) J += lane_cost(cwu, uv, d.ur, cumax, cumin, d.luh, d.lul, mu, bx, viol);
    double v = 0.0;
    if (T < Pn && d.ct != 0) {
      if ((P.con_static & 1) && !term) {
        v = dot_lds(Ac + T * ly.ldg, 1, zb, 1, nzp, d.bc);
      } else {
        const double* At = AconTi + (size_t)k * nz * Pn + T;
        v = dot_strided(At, Pn, zb, n, d.bc);
        if (!term) v = dot_strided(At + (size_t)n * Pn, Pn, zb + np, m, v);
      }
      if (d.ct != 3) {
        const bool eq = d.ct == 1;
        const bool act = eq || (v >= 0.0) || (d.lam > 0.0);
        J += d.lam * v + (act ? 0.5 * mu * v * v : 0.0);
        viol = fmax(viol, eq ? fabs(v) : v);
      }
    }
    if (P.ncone > 0) {
      if (T < Pn) {
        cvv[T] = v;
        cll[T] = d.lam;
      }
      wsync();
      if (T < Pn && d.ct == 3) {
        const Cone c = cone_eval<false>(myc0, mycp, T - myc0);
        J += c.cost;
        viol = fmax(viol, c.viol);
      }
      wsync();
    }
  }

  // branch-free AL cost of one element (selects only): same value as lane_cost
  static __device__ __forceinline__ double lane_cost_sel(double w, double z, double zr, double zmx, double zmn, double lhi, double llo,
                                                         double mu, bool on, bool box_on, double& viol) {
    const double e = z - zr;
    double J = 0.5 * w * e * e;
    const double chi = z - zmx, clo = zmn - z;
    const bool bh = on & box_on & (zmx < 1e300), bl = on & box_on & (zmn > -1e300);
    const bool ah = (chi >= 0.0) | (lhi > 0.0), al = (clo >= 0.0) | (llo > 0.0);
    J += bh ? lhi * chi + (ah ? 0.5 * mu * chi * chi : 0.0) : 0.0;
    J += bl ? llo * clo + (al ? 0.5 * mu * clo * clo : 0.0) : 0.0;
    viol = fmax(viol, bh ? chi : 0.0);
    viol = fmax(viol, bl ? clo : 0.0);
    return on ? J : 0.0;
  }

  // Rollout for the common problem class -- time-invariant dynamics, box constraints only -- written
  // without a single branch in the knot body (clamped unconditional loads, selects, LDS trash slot):
  // with branches hipcc waits vmcnt(0) at the first use after a join and the operands requested for
  // knot k+1 overlap with nothing (the generic body has ~60 of them).  CLOSED is a template argument.
  template <bool CLOSED>
  __device__ __forceinline__ RollOut rollout_simple(double alpha) {
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    double* Xd = CLOSED ? Xp(cur ^ 1) : Xp(cur);
    double* Ud = CLOSED ? Up(cur ^ 1) : Up(cur);
    const bool isx = T < n, isu = T < m;
    const int Tn = isx ? T : n - 1, Tm = isu ? T : m - 1;
    double* zslot = zb + (isx ? T : nzp);         // lanes without a state element write the trash word zb[nzp] (= qz[0], dead here)
    double* uslot = zb + (isu ? np + T : nzp);
    double* dslot = isx ? dxv + T : zb + nzp;
    double* gtrash = P.trash + (size_t)inst * 64 + T;
    const double* grow = G + Tn * ly.ldg;
    const double fT = fk(0)[Tn];
    double J = 0.0, viol = 0.0;
    bool lim = false, chg = false, big = false;
    unsigned char* const aq = aset_plane(ap) + T;  // active set of the trajectory produced (costate sweep, gain reuse)
    double xb = isx ? x0i[Tn] : 0.0;
    // K_k (m x n) travels as a whole block: requested a knot ahead with coalesced loads (n m / 64 per lane), parked in LDS
    // (the two halves of S, scratch outside the backward pass) at the end of the knot before, read from there by the m lanes
    // that form u_k.  Until round 4 the first sixteen columns came through registers and the rest was read from global memory
    // INSIDE the knot's chain -- (n - 16) / 8 exposed round trips per knot, half of a rollout knot at n = 32.  Same terms in the
    // same order.  (m > 16: the old path.)
    constexpr bool KLDS = MC > 0;
    constexpr int KR = MC > 0 ? MC : 1;
    const unsigned nm_ = (unsigned)(n * m);
    lds_d* const kbuf[2] = {(lds_d*)S, (lds_d*)S + nm_};
    lds_d* const ktrash = (lds_d*)zb + nzp;
    struct KB { double v[KR]; };
    auto k_req = [&](int kk) __attribute__((always_inline)) {
      const unsigned ku = kk < N - 1 ? kk : N - 2;
      KB r;
#pragma unroll
      for (int u = 0; u < KR; ++u) {
        const unsigned e = T + 64 * u;
        r.v[u] = (CLOSED && KLDS) ? ldg(Kgi, ku * nm_ + (e < nm_ ? e : nm_ - 1)) : 0.0;
      }
      return r;
    };
    auto k_put = [&](const KB& r, lds_d* st) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < KR; ++u) {
        const unsigned e = T + 64 * u;
        *(e < nm_ ? st + e : ktrash) = r.v[u];
      }
    };
    struct Ld { double xs, us, dgv, xr, ur, lxh, lxl, luh, lul, kp[KLDS ? 1 : 16]; };
    auto ld = [&](int k) {
      Ld d;
      const int ku = k < N - 1 ? k : N - 2;     // the terminal knot has no control: clamp, its control terms are switched off
      d.xs = Xs[(size_t)k * n + Tn];
      d.xr = Xri[(size_t)(kref + k) * n + Tn];
      d.lxh = Lbi[((size_t)k * 2 + 0) * nz + Tn];
      d.lxl = Lbi[((size_t)k * 2 + 1) * nz + Tn];
      d.us = Us[(size_t)ku * m + Tm];
      d.ur = Uri[(size_t)(kref + ku) * m + Tm];
      d.luh = Lbi[((size_t)ku * 2 + 0) * nz + n + Tm];
      d.lul = Lbi[((size_t)ku * 2 + 1) * nz + n + Tm];
      d.dgv = CLOSED ? dgi[(size_t)ku * m + Tm] : 0.0;
      if constexpr (KLDS) {
        d.kp[0] = 0.0;
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) d.kp[u] = CLOSED ? Kgi[(size_t)ku * n * m + (size_t)(u < n ? u : n - 1) * m + Tm] : 0.0;
      }
      return d;
    };
    Ld d = ld(0);
    if constexpr (CLOSED && KLDS) {
      const KB k0 = k_req(0);
      k_put(k0, kbuf[0]);
    }
    for (int k = 0; k < N - 1; ++k) {
      const Ld dn = ld(k + 1);
      KB kn;
      if constexpr (CLOSED && KLDS) kn = k_req(k + 1);
      const bool bx = box_at(k);
      *zslot = xb;
      if (CLOSED) *dslot = xb - d.xs;
      *(isx ? Xd + (size_t)k * n + T : gtrash) = xb;
      wsync();
      double acc = d.us;
      if (CLOSED) {
        acc += alpha * d.dgv;
        if constexpr (KLDS) {
          const lds_d* st = kbuf[k & 1];
          const lds_d* dv = (const lds_d*)dxv;
          for (int j0 = 0; j0 < np; j0 += 8) {
            double x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              x[u] = st[(unsigned)(j0 + u < n ? j0 + u : n - 1) * (unsigned)m + (unsigned)Tm];
              y[u] = dv[j0 + u];  // zero beyond n
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += x[u] * y[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += d.kp[u] * dxv[u];
          if (n > 16) acc = dot_strided(Kgi + (size_t)k * n * m + (size_t)16 * m + Tm, m, dxv + 16, n - 16, acc);
        }
        *(isu ? Ud + (size_t)k * m + T : gtrash) = acc;
      }
      const double uv = acc;
      *uslot = uv;
      if (CLOSED) {
        chg = chg | (isx & (xb != d.xs)) | (isu & (uv != d.us));
        big = big | (isx & !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs)))) | (isu & !(fabs(uv - d.us) <= 1e-7 * (1.0 + fabs(d.us))));
      }
      wsync();
      J += lane_cost_sel(cwx, xb, d.xr, cxmax, cxmin, d.lxh, d.lxl, mu, isx, bx, viol);
      J += lane_cost_sel(cwu, uv, d.ur, cumax, cumin, d.luh, d.lul, mu, isu, bx, viol);
      lim = lim | (isx & !(fabs(xb) <= P.o.max_state_value)) | (isu & !(fabs(uv) <= P.o.max_control_value));
      aq[(size_t)k * 64] = (unsigned char)(box_code(xb, cxmax, cxmin, d.lxh, d.lxl, isx & bx) | (box_code(uv, cumax, cumin, d.luh, d.lul, isu & bx) << 2));
      const double xn = dot_lds(grow, 1, zb, 1, nzp, fT);
      if constexpr (CLOSED && KLDS) k_put(kn, kbuf[(k + 1) & 1]);  // block k + 1 (its loads had the knot to arrive)
      wsync();
      xb = isx ? xn : 0.0;
      d = dn;
    }
    *zslot = xb;
    *(isx ? Xd + (size_t)(N - 1) * n + T : gtrash) = xb;
    wsync();
    J += lane_cost_sel(cwfx, xb, d.xr, cxmax, cxmin, d.lxh, d.lxl, mu, isx, box_at(N - 1), viol);
    lim = lim | (isx & !(fabs(xb) <= P.o.max_state_value));
    aq[(size_t)(N - 1) * 64] = (unsigned char)box_code(xb, cxmax, cxmin, d.lxh, d.lxl, isx & box_at(N - 1));
    if (CLOSED) {
      chg = chg | (isx & (xb != d.xs));
      big = big | (isx & !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs))));
    }
    block_sync();
    RollOut r;
    r.J = wave_sum(J);
    r.cmax = wave_max(viol);
    r.limit = wave_any(lim);
    r.unchanged = CLOSED && !wave_any(chg);
    r.tiny = CLOSED && !wave_any(big);
    return r;
  }

  // element J of v in this lane's 16-lane row (v_mov_b64_dpp row_newbcast:J)
  template <int J>
  static __device__ __forceinline__ double row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);
  }
  // acc + sum_{j < 16} row[j] * v_j in the order j = 0..15, v_j = element j of the row vector v
  template <int J>
  struct RowDot {
    static __device__ __forceinline__ double run(const double* row, double v, double acc) {
      if constexpr (J > 1) acc = RowDot<J - 1>::run(row, v, acc);
      return __builtin_fma(row[J - 1], row_bcast<J - 1>(v), acc);
    }
  };

  // Rollout for n, m <= 16 with at most 16 linear rows whose table is resident (no cones): the whole knot lives in
  // the first 16-lane row of the wave.  Lane T holds x_T and u_T, row T of K_k, of [A_k B_k] and of the constraint
  // table in registers; the matrix-vector products take their vector operand from the other lanes with DPP
  // row broadcasts, so a knot has no LDS hand-over and no barrier (the LDS version: three of each, ~8 k cycles per
  // knot against the ~70 FMAs of useful work).  The terms are accumulated in the order of the LDS version, the
  // results are bit-identical.  Lanes 16..63 carry zeros.  No branch in the knot body: every lane executes every
  // DPP instruction (a DPP read from a disabled lane returns 0).
  // uniform base (SGPR pair) + 32-bit per-lane element index: `global_load_dwordx2 v, v_off, s[base:base+1]`, one
  // address VGPR per access instead of a 64-bit pointer (every per-instance array is far below 4 GiB)
  static __device__ __forceinline__ double ldg(const double* base, unsigned idx) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (idx << 3));
  }
  static __device__ __forceinline__ void stg(double* base, unsigned idx, double v) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (idx << 3)) = v;
  }
  struct KRegs {
    double v[4];
  };

  template <bool CLOSED, bool LTV, bool ROWS>
  __device__ __forceinline__ RollOut rollout_row(double alpha) {
    // The lane index is made opaque here: everything lane-dependent below (a few dozen addresses) is then computed
    // inside this function.  Otherwise LLVM hoists it, with the lane-dependent invariants of every other phase, to
    // the top of the kernel -- more values than there are registers, and their reloads from scratch (each followed
    // by s_waitcnt vmcnt(0)) sat in this loop and drained the operands in flight at every knot.
    int t = T;
    asm volatile("" : "+v"(t));
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    double* Xd = CLOSED ? Xp(cur ^ 1) : Xp(cur);
    double* Ud = CLOSED ? Up(cur ^ 1) : Up(cur);
    const bool isx = t < n, isu = t < m, isr = ROWS && t < Pn, rows = ROWS;
    const unsigned Tn = isx ? t : n - 1, Tm = isu ? t : m - 1, Tr = isr ? t : 0;
    double* gtrash = P.trash + (size_t)inst * 64;  // [64]: lane t's sink is word t
    // per-knot operands of the generic rows; a problem without rows reads (and ignores) the trash line instead
    const int* ctp = rows ? P.ctype : (const int*)gtrash;
    const double* lcp = rows ? Lci : gtrash;
    const double* bcp = rows ? bconi : gtrash;
    const unsigned rstride = rows ? (unsigned)Pn : 0u;
    const int ldg_ = ly.ldg;
    const unsigned nn = n * n, nm = n * m;
    lds_d* const ltrash = (lds_d*)zb + nzp;  // LDS sink (qz[0], dead during rollouts)
    // Row t of the constraint table (ac) is read from LDS at every knot rather than held across the loop (registers:
    // a loop that spills reloads from scratch, each reload a vmcnt(0)); rows beyond Pn never count (on = false).
    const lds_d* acrow = (const lds_d*)Ac + (rows ? Tr * ldg_ : 0);
    const lds_d* abrow = (const lds_d*)G + Tn * ldg_;
    double ab[32];  // row t of [A B]: time-invariant dynamics keep it for the whole rollout, per-knot ones refill it
#pragma unroll
    for (int c = 0; c < 32; ++c) ab[c] = abrow[c];
    unsigned char* const aq = aset_plane(ap) + t;
    // closed-loop rollouts record the active set of the trajectory they produce; in the gain-reuse class the open-loop one
    // does too (the set of plane cur under the current duals, compared with the stored pass's at the start of a solve)
    constexpr bool CODES = CLOSED || (kReuseRow && !ROWS && !LTV);
    double fT = LTV ? 0.0 : fk(0)[Tn];
    double J = 0.0, viol = 0.0;
    bool lim = false, chg = false, big = false;
    double xb = isx ? x0i[Tn] : 0.0;
    struct Ld {
      double xs, us, dgv, xr, ur, lxh, lxl, luh, lul, lam, bc;
      int ct;
    };
    auto ld = [&](int kk) {
      Ld d;
      const unsigned k = kk < N - 1 ? kk : N - 1;
      const unsigned ku = k < (unsigned)(N - 1) ? k : N - 2;  // the terminal knot has no control: clamp, its control terms are switched off
      d.xs = ldg(Xs, k * n + Tn);
      d.xr = ldg(Xri, (kref + k) * n + Tn);
      d.lxh = ldg(Lbi, (k * 2 + 0) * nz + Tn);
      d.lxl = ldg(Lbi, (k * 2 + 1) * nz + Tn);
      d.us = ldg(Us, ku * m + Tm);
      d.ur = ldg(Uri, (kref + ku) * m + Tm);
      d.luh = ldg(Lbi, (ku * 2 + 0) * nz + n + Tm);
      d.lul = ldg(Lbi, (ku * 2 + 1) * nz + n + Tm);
      d.dgv = CLOSED ? ldg(dgi, ku * m + Tm) : 0.0;
      d.ct = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(ctp) + ((k * rstride + Tr) << 2));
      d.lam = ldg(lcp, k * rstride + Tr);
      d.bc = ldg(bcp, k * rstride + Tr);
      return d;
    };
    // K_k (m x n, n m <= 256 elements) and, for per-knot dynamics, [A_k B_k f_k] travel as whole blocks: four / nine
    // coalesced loads per lane, requested two knots ahead and parked in LDS a knot ahead (S and Huu alternate for K,
    // W and Hux for the dynamics: all four are scratch outside the backward pass); lane t then reads its rows.
    auto k_request = [&](int kk) {
      const unsigned ku = kk < N - 1 ? kk : N - 2;
      KRegs r;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        r.v[u] = CLOSED ? ldg(Kgi, ku * nm + (e < nm ? e : nm - 1)) : 0.0;
      }
      return r;
    };
    auto k_park = [&](const KRegs& r, lds_d* st) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        *(e < nm ? st + e : ltrash) = r.v[u];
      }
    };
    auto dyn_req = [&](int k) {
      const double *A_ = Ak(k), *B_ = Bk(k);
      DynRegs q;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        q.a[u] = ldg(A_, e < nn ? e : nn - 1);
        q.b[u] = ldg(B_, e < nm ? e : nm - 1);
      }
      q.f = ldg(fk(k), Tn);
      return q;
    };
    auto dyn_park = [&](const DynRegs& q, lds_d* st) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        *(e < nn ? st + e : ltrash) = q.a[u];
        *(e < nm ? st + nn + e : ltrash) = q.b[u];
      }
      *(isx ? st + nn + nm + t : ltrash) = q.f;
    };
    // The per-lane operands are requested THREE knots ahead: a knot is ~1.5 k cycles of arithmetic, a round trip to
    // HBM ~4 k, and the wave has the SIMD to itself -- with one knot of lookahead every knot waited for memory.
    // (Compile-time slots instead of the copy rotation below -- four knots per body -- were tried again in round 4 with the
    // rollout a function of its own: the body of the quadruped's variant still needs 256 + 256 registers and 763 scratch
    // operations, with or without a scheduling barrier per knot.)
    // The loop body is one basic block (no branch: hipcc waits vmcnt(0) after a join), so the waits are counted.
    lds_d* const park[2] = {(lds_d*)W, (lds_d*)Hux};
    lds_d* const kpark[2] = {(lds_d*)S, (lds_d*)Huu};
    DynRegs dq0 = {}, dq1 = {};
    KRegs kq0 = k_request(0), kq1 = k_request(1);
    k_park(kq0, kpark[0]);
    kq0 = kq1;
    if (LTV) {
      dq0 = dyn_req(0);
      dq1 = dyn_req(N > 2 ? 1 : 0);
      dyn_park(dq0, park[0]);
      dq0 = dq1;  // block 1, parked at the end of knot 0
    }
    Ld d = ld(0), d1 = ld(1), d2 = ld(2);
    wsync();
    for (int k = 0; k < N - 1; ++k) {
      const Ld d3 = ld(k + 3);
      kq1 = k_request(k + 2);
      double ac[32];
      if constexpr (ROWS) {
#pragma unroll
        for (int c = 0; c < 32; ++c) ac[c] = acrow[c];
      }
      double kp[16];
      if constexpr (CLOSED) {
        const lds_d* st = kpark[k & 1];
#pragma unroll
        for (int j = 0; j < 16; ++j) kp[j] = st[(j < n ? j : n - 1) * m + Tm];
      }
      if constexpr (LTV) {  // row t of this knot's block (parked a knot ago); block k + 2 on its way
        const lds_d* st = park[k & 1];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          ab[j] = st[Tn + n * (j < n ? j : n - 1)];
          ab[16 + j] = st[nn + Tn + n * (j < m ? j : m - 1)];
        }
        fT = st[nn + nm + Tn];
        dq1 = dyn_req(k + 2 < N - 1 ? k + 2 : N - 2);
      }
      const bool bx = box_at(k);
      *(isx ? Xd + (unsigned)k * n + t : gtrash + t) = xb;
      double uv = d.us;
      if (CLOSED) {
        const double dx = isx ? xb - d.xs : 0.0;
        uv += alpha * d.dgv;
        uv = RowDot<16>::run(kp, dx, uv);
        *(isu ? Ud + (unsigned)k * m + t : gtrash + t) = uv;
      }
      uv = isu ? uv : 0.0;
      if (CLOSED) {
        chg = chg | (isx & (xb != d.xs)) | (isu & (uv != d.us));
        big = big | (isx & !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs)))) | (isu & !(fabs(uv - d.us) <= 1e-7 * (1.0 + fabs(d.us))));
      }
      J += lane_cost_sel(cwx, xb, d.xr, cxmax, cxmin, d.lxh, d.lxl, mu, isx, bx, viol);
      J += lane_cost_sel(cwu, uv, d.ur, cumax, cumin, d.luh, d.lul, mu, isu, bx, viol);
      // generic rows: value, AL cost, violation
      double v = 0.0;
      if constexpr (ROWS) {
        v = RowDot<16>::run(ac, xb, d.bc);
        v = RowDot<16>::run(ac + 16, uv, v);
      }
      const bool on = isr & (d.ct != 0), eq = d.ct == 1;
      const bool act = eq | (v >= 0.0) | (d.lam > 0.0);
      {
        const double cj = d.lam * v + (act ? 0.5 * mu * v * v : 0.0);
        J += on ? cj : 0.0;
        viol = fmax(viol, on ? (eq ? fabs(v) : v) : 0.0);
      }
      lim = lim | (isx & !(fabs(xb) <= P.o.max_state_value)) | (isu & !(fabs(uv) <= P.o.max_control_value));
      if (CODES) {  // active-set code of the knot at the trajectory produced (compared with the backward pass's by the costate sweep)
        const unsigned code = box_code(xb, cxmax, cxmin, d.lxh, d.lxl, isx & bx) | (box_code(uv, cumax, cumin, d.luh, d.lul, isu & bx) << 2) |
                              ((on & act) ? 16u : 0u);
        aq[(unsigned)k * 64u] = (unsigned char)code;
      }
      double xn = RowDot<16>::run(ab, xb, fT);
      xn = RowDot<16>::run(ab + 16, uv, xn);
      k_park(kq0, kpark[(k & 1) ^ 1]);  // block k + 1
      kq0 = kq1;
      if constexpr (LTV) {
        dyn_park(dq0, park[(k & 1) ^ 1]);
        dq0 = dq1;
      }
      wsync();
      xb = isx ? xn : 0.0;
      d = d1;
      d1 = d2;
      d2 = d3;
    }
    *(isx ? Xd + (unsigned)(N - 1) * n + t : gtrash + t) = xb;
    double ac[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) ac[c] = ROWS ? (double)acrow[c] : 0.0;
    J += lane_cost_sel(cwfx, xb, d.xr, cxmax, cxmin, d.lxh, d.lxl, mu, isx, box_at(N - 1), viol);
    // rows of the terminal knot see the state only
    const double v = ROWS ? RowDot<16>::run(ac, xb, d.bc) : 0.0;
    const bool on = isr & (d.ct != 0), eq = d.ct == 1;
    const bool act = eq | (v >= 0.0) | (d.lam > 0.0);
    {
      const double cj = d.lam * v + (act ? 0.5 * mu * v * v : 0.0);
      J += on ? cj : 0.0;
      viol = fmax(viol, on ? (eq ? fabs(v) : v) : 0.0);
    }
    lim = lim | (isx & !(fabs(xb) <= P.o.max_state_value));
    if (CLOSED) {
      chg = chg | (isx & (xb != d.xs));
      big = big | (isx & !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs))));
    }
    if (CODES) {
      const bool bxT = box_at(N - 1);
      aq[(unsigned)(N - 1) * 64u] = (unsigned char)(box_code(xb, cxmax, cxmin, d.lxh, d.lxl, isx & bxT) | ((on & act) ? 16u : 0u));
    }
    block_sync();  // phase end: the trajectory written to global memory is read by other lanes next
    RollOut r;
    r.J = wave_sum(J);
    r.cmax = wave_max(viol);
    r.limit = wave_any(lim);
    r.unchanged = CLOSED && !wave_any(chg);
    r.tiny = CLOSED && !wave_any(big);
    return r;
  }

  // First half of the costate sweep: l_x, l_u of every knot of plane cur (tracking cost, box terms, A_c' g of the generic
  // rows: what expansion() computes) into the plane Qz.  No recursion here: the knots are independent, their
  // operands are requested three knots ahead.
  template <bool ROWS>
  __device__ __forceinline__ void grad_pass() {
    int t = T;
    asm volatile("" : "+v"(t));
    const bool isx = t < n, isu = t < m, isr = ROWS && t < Pn;
    const unsigned Tn = isx ? t : n - 1, Tm = isu ? t : m - 1, Tr = isr ? t : 0;
    const int ldg_ = ly.ldg;
    double* gtrash = P.trash + (size_t)inst * 64;
    double* Qzi = P.Qz + (size_t)inst * N * nz;
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    const int* ctp = ROWS ? P.ctype : (const int*)gtrash;
    const double* lcp = ROWS ? Lci : gtrash;
    const double* bcp = ROWS ? bconi : gtrash;
    const unsigned rstride = ROWS ? (unsigned)Pn : 0u;
    const lds_d* acrow = (const lds_d*)Ac + (ROWS ? Tr * ldg_ : 0);
    const lds_d* accol = (const lds_d*)Ac;
    const int lastrow = Pn > 0 ? Pn - 1 : 0;
    struct Qk {  // per-lane operands of a knot
      double xs, us, xr, ur, lxh, lxl, luh, lul, lam, bc;
      int ct;
    };
    auto ldq = [&](int kk) __attribute__((always_inline)) {
      const unsigned k = kk < N - 1 ? kk : N - 1;
      const unsigned ku = k < (unsigned)(N - 1) ? k : N - 2;
      Qk q;
      q.xs = ldg(Xs, k * n + Tn);
      q.xr = ldg(Xri, (kref + k) * n + Tn);
      q.lxh = ldg(Lbi, (k * 2 + 0) * nz + Tn);
      q.lxl = ldg(Lbi, (k * 2 + 1) * nz + Tn);
      q.us = ldg(Us, ku * m + Tm);
      q.ur = ldg(Uri, (kref + ku) * m + Tm);
      q.luh = ldg(Lbi, (ku * 2 + 0) * nz + n + Tm);
      q.lul = ldg(Lbi, (ku * 2 + 1) * nz + n + Tm);
      q.ct = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(ctp) + ((k * rstride + Tr) << 2));
      q.lam = ldg(lcp, k * rstride + Tr);
      q.bc = ldg(bcp, k * rstride + Tr);
      return q;
    };
    Qk qs[4];  // operands three knots ahead (a knot here is ~400 instructions, less than a memory round trip)
    qs[0] = ldq(0);
    qs[1] = ldq(1);
    qs[2] = ldq(2);
    auto knot = [&](auto uc, int k) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;
      const bool live = k < N, term = k >= N - 1;
      qs[(U + 3) & 3] = ldq(k + 3);
      const Qk& q = qs[U];
      const bool bx = box_at(k < N ? k : N - 1);
      const double x = isx ? q.xs : 0.0, u = (isu & !term) ? q.us : 0.0;
      double qx = (term ? cwfx : cwx) * (x - q.xr), qu = cwu * (u - q.ur);
      box_grad(x, cxmax, cxmin, q.lxh, q.lxl, isx & bx, qx);
      box_grad(u, cumax, cumin, q.luh, q.lul, isu & bx & !term, qu);
      if constexpr (ROWS) {
        double ac[32], acx[16], acu[16];
#pragma unroll
        for (int c = 0; c < 32; ++c) ac[c] = acrow[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = r < Pn ? r : lastrow;  // rows beyond Pn: g_r = 0
          acx[r] = accol[rr * ldg_ + Tn];
          acu[r] = accol[rr * ldg_ + 16 + Tm];
        }
        double v = RowDot<16>::run(ac, x, q.bc);
        v = RowDot<16>::run(ac + 16, u, v);  // u = 0 at the terminal knot: its rows see the state only
        const bool on = isr & (q.ct != 0), act = (q.ct == 1) | (v >= 0.0) | (q.lam > 0.0);
        const double g = on ? q.lam + (act ? mu * v : 0.0) : 0.0;
        qx = RowDot<16>::run(acx, g, qx);
        qu = RowDot<16>::run(acu, g, qu);
      }
      *((live & isx) ? Qzi + (unsigned)k * nz + t : gtrash + t) = qx;
      *((live & isu & !term) ? Qzi + (unsigned)k * nz + n + t : gtrash + t) = qu;
    };
    for (int k = 0; k < N; k += 4) {
      knot(std::integral_constant<int, 0>{}, k);
      knot(std::integral_constant<int, 1>{}, k + 1);
      knot(std::integral_constant<int, 2>{}, k + 2);
      knot(std::integral_constant<int, 3>{}, k + 3);
    }
    block_sync();  // Qz is read back by the recursion
  }

  // Costate sweep (default mode; the problems of the row rollouts): lambda_N = l_x(N), lambda_k = l_x(k) + A_k' lambda_{k+1},
  // g_k = l_u(k) + B_k' lambda_{k+1}, d_k = -Quu_k^-1 g_k -- the first-order part of the backward pass, ~400
  // instructions per knot instead of ~3000.  l_x, l_u at the current trajectory (tracking cost, box terms, A_c' g of
  // the generic rows) were written to Qz by grad_pass() just before; Quu_k = L D L' is the factor the last
  // backward pass stored (same active set = same Quu).  Inside a fixed active set the problem is quadratic, and by
  // induction over the knots every feedforward term of a new backward pass vanishes iff every d_k of this recursion
  // does.  Returns true if |d_k,a| <= 1e-9 (1 + |u_k,a|) at every knot -- the test of the confirmation iterations.
  // Lane T holds lambda_T, g_T; column T of A_k and of B_k in registers (time-invariant: loaded once; per-knot
  // dynamics: from the blocks parked in LDS, requested two knots ahead as in the rollouts); the factor of knot k is
  // parked in LDS a knot ahead and every lane solves for the whole d_k.
  // FULL (time-invariant dynamics, box constraints only, MC = 4): the first-order half of a whole iteration, as
  // adjoint_lds(full) for the larger sizes -- the gains K_k in memory stay (the caller has established that a backward pass
  // here would reproduce them), the costate runs s_k = Qx + K_k' Qu, the feedforward terms d_k = -Quu_k^-1 Qu are written
  // out and the expected decrease (dV1, dV2) is returned.
  template <bool LTV, bool FULL = false>
  __device__ __forceinline__ bool adjoint_row(double* dV1 = nullptr, double* dV2 = nullptr) {
    static_assert(!(LTV && FULL), "gain reuse: time-invariant dynamics only");
    constexpr int MP = MC > 0 ? MC : 4, FS = MP * (MP + 1) / 2;
    int t = T;
    asm volatile("" : "+v"(t));
    const bool isx = t < n, isu = t < m;
    const unsigned Tn = isx ? t : n - 1, Tm = isu ? t : m - 1;
    const int ldg_ = ly.ldg;
    const unsigned nn = n * n, nm = n * m;
    lds_d* const ltrash = (lds_d*)zb + nzp;
    const double* Qzi = P.Qz + (size_t)inst * N * nz;
    const double* faci = P.fac + (size_t)inst * N * FS;
    const double* Us = Up(cur);
    double ca[16], cb[16];  // column t of A, column t of B
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ca[i] = G[i * ldg_ + Tn];
      cb[i] = G[i * ldg_ + 16 + Tm];
    }
    auto dyn_req = [&](int kk) __attribute__((always_inline)) {
      const int k = kk > 0 ? kk : 0;
      const double *A_ = Ak(k), *B_ = Bk(k);
      DynRegs q;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        q.a[u] = ldg(A_, e < nn ? e : nn - 1);
        q.b[u] = ldg(B_, e < nm ? e : nm - 1);
      }
      q.f = 0.0;
      return q;
    };
    auto dyn_park = [&](const DynRegs& q, lds_d* st) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned e = t + 64 * u;
        *(e < nn ? st + e : ltrash) = q.a[u];
        *(e < nm ? st + nn + e : ltrash) = q.b[u];
      }
    };
    struct Fk {
      double v[3];  // FS <= 136 elements over 64 lanes
    };
    auto fac_req = [&](int kk) __attribute__((always_inline)) {
      const unsigned k = kk > 0 ? kk : 0;
      Fk f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const unsigned e = t + 64 * u;
        f.v[u] = ldg(faci, k * FS + (e < FS ? e : FS - 1));
      }
      return f;
    };
    auto fac_park = [&](const Fk& f, lds_d* st) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const unsigned e = t + 64 * u;
        *(e < FS ? st + e : ltrash) = f.v[u];
      }
    };
    struct Qk {
      double qx, qu, us;
      double kc[FULL ? MP : 1];  // FULL: column t of K_k
    };
    auto ldq = [&](int kk) __attribute__((always_inline)) {
      const unsigned k = kk > 0 ? kk : 0;
      Qk q;
      q.qx = ldg(Qzi, k * nz + Tn);
      q.qu = ldg(Qzi, k * nz + n + Tm);
      q.us = ldg(Us, k * m + Tm);
      if constexpr (FULL) {
#pragma unroll
        for (int a_ = 0; a_ < MP; ++a_) q.kc[a_] = ldg(Kgi, k * (unsigned)(n * m) + Tn * m + (a_ < m ? a_ : m - 1));
      } else {
        q.kc[0] = 0.0;
      }
      return q;
    };
    double dv1 = 0.0;
    double* gtr = P.trash + (size_t)inst * 64 + t;
    lds_d* const park[2] = {(lds_d*)W, (lds_d*)Hux};
    lds_d* const fpark[2] = {(lds_d*)S, (lds_d*)S + 136};
    DynRegs dq[2] = {};
    Fk fq[2];
    Qk qs[4];
    const int kt = N - 2;  // first knot of the sweep
    if (LTV) {
      dq[0] = dyn_req(kt);
      dyn_park(dq[0], park[kt & 1]);
      dq[1] = dyn_req(kt - 1);
    }
    fq[0] = fac_req(kt);
    fac_park(fq[0], fpark[kt & 1]);
    fq[1] = fac_req(kt - 1);
    qs[0] = ldq(kt);
    qs[1] = ldq(kt - 1);
    qs[2] = ldq(kt - 2);
    double lam = isx ? ldg(Qzi, (unsigned)(N - 1) * nz + Tn) : 0.0;
    bool dbig = false;
    wsync();
    auto knot = [&](auto uc, int k) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;  // position in the group of four: register slots are compile-time
      const bool live = k >= 0;
      qs[(U + 3) & 3] = ldq(k - 3);
      fq[U & 1] = fac_req(k - 2);  // block k is in LDS, block k - 1 waits in slot (U + 1) & 1
      if constexpr (LTV) {
        const lds_d* st = park[k & 1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          ca[i] = st[(i < n ? i : n - 1) + n * Tn];
          cb[i] = st[nn + (i < n ? i : n - 1) + n * Tm];
        }
        dq[U & 1] = dyn_req(k - 2);
      }
      const double gx = RowDot<16>::run(ca, lam, qs[U].qx);
      double gu = RowDot<16>::run(cb, lam, qs[U].qu);
      gu = isu ? gu : 0.0;
      // d_k = Quu_k^-1 g_k (the sign does not matter here), every lane for the whole vector
      double y[MP];
      gather_row<MP>(y, gu);
      double sx = gx;
      if constexpr (FULL) {  // s_k = Qx + K_k' Qu (gu is zero beyond m)
#pragma unroll
        for (int a_ = 0; a_ < MP; ++a_) sx += qs[U].kc[a_] * y[a_];
      }
      const lds_d* fk = fpark[k & 1];
#pragma unroll
      for (int j = 0; j < MP; ++j)          // forward: L y = g
#pragma unroll
        for (int i = j + 1; i < MP; ++i) y[i] -= fk[i * (i + 1) / 2 + j] * y[j];
#pragma unroll
      for (int r = 0; r < MP; ++r) y[r] *= fk[r * (r + 1) / 2 + r];  // 1 / D
#pragma unroll
      for (int j = MP - 1; j >= 0; --j)     // backward: L' d = y
#pragma unroll
        for (int i = 0; i < j; ++i) y[i] -= fk[j * (j + 1) / 2 + i] * y[j];
      double dsel = 0.0;
#pragma unroll
      for (int r = 0; r < MP; ++r) dsel = (t == r) ? y[r] : dsel;
      dbig = dbig | (live & isu & !(fabs(dsel) <= 1e-9 * (1.0 + fabs(qs[U].us))));
      if constexpr (FULL) {
        *((live & isu) ? dgi + (unsigned)(k > 0 ? k : 0) * m + t : gtr) = 0.0 - dsel;
        dv1 -= (live & isu) ? dsel * gu : 0.0;
      }
      lam = (live & isx) ? sx : lam;
      fac_park(fq[(U + 1) & 1], fpark[(k - 1) & 1]);
      if constexpr (LTV) dyn_park(dq[(U + 1) & 1], park[(k - 1) & 1]);
      wsync();
    };
    for (int k = kt; k >= 0; k -= 4) {
      knot(std::integral_constant<int, 0>{}, k);
      knot(std::integral_constant<int, 1>{}, k - 1);
      knot(std::integral_constant<int, 2>{}, k - 2);
      knot(std::integral_constant<int, 3>{}, k - 3);
    }
    if constexpr (FULL) {
      block_sync();  // the feedforward terms are read by the rollout
      *dV1 = wave_sum(dv1);
      *dV2 = -0.5 * *dV1;
    }
    return !wave_any(dbig);
  }

  // The costate sweep for the sizes of the generic instantiation (n > 16, m <= 16), time-invariant dynamics, box
  // constraints only -- the class of rollout_simple, i.e. the state-dimension sweep.  Same recursion and the same test
  // as adjoint_row, with the vectors in LDS: lambda_k (lane c: the c-th element, dot products down column c of the
  // resident [A B]), g_k handed to every lane through LDS for the per-lane solve with the stored factor.  O(n^2) per
  // knot against the O(n^3) of the backward pass it replaces: ~3 % of it at n = 64.
  // full = true: the first-order half of a whole iteration -- the gains K_k in memory stay (the caller has
  // established that a backward pass here would reproduce them), the costate runs s_k = Qx + K_k' Qu, the
  // feedforward terms d_k = -Quu_k^-1 Qu are written out and the expected decrease (dV1, dV2) is returned.
  __device__ __forceinline__ bool adjoint_lds(bool full, double& dV1, double& dV2) {
    constexpr int MP = MC > 0 ? MC : 4, FS = MP * (MP + 1) / 2;
    const bool isx = T < n, isu = T < m;
    const unsigned Tn = isx ? T : n - 1, Tm = isu ? T : m - 1;
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    const double* faci = P.fac + (size_t)inst * N * FS;
    lds_d* const ltrash = (lds_d*)zb + nzp;
    lds_d* const lamb[2] = {(lds_d*)sv, (lds_d*)dxv};      // lambda_{k+1} / lambda_k, zero beyond n
    lds_d* const fpark[2] = {(lds_d*)S, (lds_d*)S + 136};  // the factor of knot k / k - 1
    lds_d* const gvec = (lds_d*)qv;                         // g_k, zero beyond m
    struct Qk {
      double xs, us, xr, ur, lxh, lxl, luh, lul, f[3];
      double kc[MP];  // full: column Tn of K_k, requested with the knot's other operands (round 4: it was read inside the knot)
    };
    auto ldq = [&](int kk) __attribute__((always_inline)) {
      const unsigned k = kk > 0 ? kk : 0;
      Qk q;
#pragma unroll
      for (int a_ = 0; a_ < MP; ++a_) q.kc[a_] = full ? ldg(Kgi, k * (unsigned)(n * m) + Tn * m + (a_ < m ? a_ : m - 1)) : 0.0;
      q.xs = ldg(Xs, k * n + Tn);
      q.xr = ldg(Xri, (kref + k) * n + Tn);
      q.lxh = ldg(Lbi, (k * 2 + 0) * nz + Tn);
      q.lxl = ldg(Lbi, (k * 2 + 1) * nz + Tn);
      q.us = ldg(Us, k * m + Tm);
      q.ur = ldg(Uri, (kref + k) * m + Tm);
      q.luh = ldg(Lbi, (k * 2 + 0) * nz + n + Tm);
      q.lul = ldg(Lbi, (k * 2 + 1) * nz + n + Tm);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const unsigned e = T + 64 * u;
        q.f[u] = ldg(faci, k * FS + (e < FS ? e : FS - 1));
      }
      return q;
    };
    auto fpark_put = [&](const Qk& q, lds_d* st) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const unsigned e = T + 64 * u;
        *(e < FS ? st + e : ltrash) = q.f[u];
      }
    };
    // terminal knot: lambda_N = l_x(N)
    {
      const unsigned k = N - 1;
      const double x = ldg(Xs, k * n + Tn);
      double q = cwfx * (x - ldg(Xri, (kref + k) * n + Tn));
      box_grad(x, cxmax, cxmin, ldg(Lbi, (k * 2 + 0) * nz + Tn), ldg(Lbi, (k * 2 + 1) * nz + Tn), isx & box_at(N - 1), q);
      for (int c = T; c < np; c += 64) lamb[0][c] = (c < n) ? q : 0.0;
    }
    Qk qs[2];
    qs[0] = ldq(N - 2);
    fpark_put(qs[0], fpark[0]);
    bool dbig = false;
    double dv1 = 0.0;
    double* gtr = P.trash + (size_t)inst * 64 + T;
    wsync();
    auto knot = [&](auto uc, int k) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;  // parity of the position: slots and LDS buffers alternate
      const bool live = k >= 0;
      qs[U ^ 1] = ldq(k - 1);
      const Qk& q = qs[U];
      const bool bx = box_at(k > 0 ? k : 0);
      const double x = isx ? q.xs : 0.0, u = isu ? q.us : 0.0;
      double qx = cwx * (x - q.xr), qu = cwu * (u - q.ur);
      box_grad(x, cxmax, cxmin, q.lxh, q.lxl, isx & bx, qx);
      box_grad(u, cumax, cumin, q.luh, q.lul, isu & bx, qu);
      const double gx = dot_lds(G + Tn, ly.ldg, (const double*)lamb[U], 1, np, qx);        // l_x + A' lambda
      double gu = dot_lds(G + np + Tm, ly.ldg, (const double*)lamb[U], 1, np, qu);        // l_u + B' lambda
      gu = isu ? gu : 0.0;
      if (T < 16) gvec[T] = gu;
      fpark_put(qs[U ^ 1], fpark[U ^ 1]);
      wsync();
      {  // s_k = Qx + K_k' Qu (with d = 0, the confirmation test, the second term is at rounding level: it is left out)
        double sx = gx;
        if (full) {
          double gq[MP];
#pragma unroll
          for (int a_ = 0; a_ < MP; ++a_) gq[a_] = gvec[a_];  // zero beyond m
#pragma unroll
          for (int a_ = 0; a_ < MP; ++a_) sx += q.kc[a_] * gq[a_];
        }
        for (int c = T; c < np; c += 64) lamb[U ^ 1][c] = (c < n) ? sx : 0.0;
      }
      double y[MP];
#pragma unroll
      for (int r = 0; r < MP; ++r) y[r] = gvec[r];
      const lds_d* fk = fpark[U];
#pragma unroll
      for (int j = 0; j < MP; ++j)          // forward: L y = g
#pragma unroll
        for (int i = j + 1; i < MP; ++i) y[i] -= fk[i * (i + 1) / 2 + j] * y[j];
#pragma unroll
      for (int r = 0; r < MP; ++r) y[r] *= fk[r * (r + 1) / 2 + r];  // 1 / D
#pragma unroll
      for (int j = MP - 1; j >= 0; --j)     // backward: L' d = y
#pragma unroll
        for (int i = 0; i < j; ++i) y[i] -= fk[j * (j + 1) / 2 + i] * y[j];
      double dsel = 0.0;
#pragma unroll
      for (int r = 0; r < MP; ++r) dsel = (T == r) ? y[r] : dsel;
      dbig = dbig | (live & isu & !(fabs(dsel) <= 1e-9 * (1.0 + fabs(q.us))));
      if (full) {
        *((live & isu) ? dgi + (unsigned)(k > 0 ? k : 0) * m + T : gtr) = 0.0 - dsel;
        dv1 -= (live & isu) ? dsel * gu : 0.0;
      }
      wsync();  // gvec and the factor buffer of this knot are free again
    };
    for (int k = N - 2; k >= 0; k -= 2) {
      knot(std::integral_constant<int, 0>{}, k);
      knot(std::integral_constant<int, 1>{}, k - 1);
    }
    // (sv and dxv are scratch again: the backward pass and the rollouts rewrite them before reading)
    if (full) {
      block_sync();  // the feedforward terms are read by the rollout
      dV1 = wave_sum(dv1);
      dV2 = -0.5 * dV1;
    }
    return !wave_any(dbig);
  }

  // y[a] = element a of the row vector v (lanes a of this 16-lane row), a < MP
  template <int MP>
  static __device__ __forceinline__ void gather_row(double (&y)[MP], double v) {
    gather_row_at<MP, 0>(y, v);
  }
  template <int MP, int A>
  static __device__ __forceinline__ void gather_row_at(double (&y)[MP], double v) {
    if constexpr (A < MP) {
      y[A] = row_bcast<A>(v);
      gather_row_at<MP, A + 1>(y, v);
    }
  }

  __device__ __forceinline__ bool row_rollouts() const {
    return SM && P.ncone == 0 && Pn <= 16 && (Pn == 0 || (P.con_static & 1));
  }

  __device__ __forceinline__ RollOut rollout(bool open, double alpha) {
    phase_begin();
    if constexpr (SM) {
      if (row_rollouts()) {
        if (Pn > 0) {
          if (P.ltv) return open ? rollout_row<false, true, true>(0.0) : rollout_row<true, true, true>(alpha);
          return open ? rollout_row<false, false, true>(0.0) : rollout_row<true, false, true>(alpha);
        }
        if (P.ltv) return open ? rollout_row<false, true, false>(0.0) : rollout_row<true, true, false>(alpha);
        return open ? rollout_row<false, false, false>(0.0) : rollout_row<true, false, false>(alpha);
      }
    } else {
      if (Pn == 0 && !P.ltv) return open ? rollout_simple<false>(0.0) : rollout_simple<true>(alpha);
    }
    return rollout_generic(open, alpha);
  }

  // rollout!(solver[, alpha]): open loop in place on plane cur, or closed loop from plane cur into
  // plane cur^1 (oracle rollout_open / rollout_alpha), fused with cost! and max_violation.  The
  // per-lane operands of knot k+1 are requested before knot k is processed.
  __device__ __forceinline__ RollOut rollout_generic(bool open, double alpha) {
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    double* Xd = open ? Xp(cur) : Xp(cur ^ 1);
    double* Ud = open ? Up(cur) : Up(cur ^ 1);
    double J = 0.0, viol = 0.0;
    bool lim = false, chg = false, big = false;
    double xb = T < n ? x0i[T] : 0.0;
    KnotLd d = load_knot(0, false, open ? 1 : 0, Xs, Us);
    const bool ahead = dyn_ahead();
    lds_d* const park[2] = {(lds_d*)W, (lds_d*)Hux};
    DynRegs dq = {};
    if (ahead) {
      dq = dyn_request(0);
      dyn_park_linear(dq, park[0]);  // visible after the first barrier of the loop
    }
    for (int k = 0; k < N - 1; ++k) {
      const bool last = k == N - 2;
      const KnotLd dn = load_knot(k + 1, last, open ? 1 : 0, Xs, Us);   // knot N-1 is the terminal knot
      if (ahead) dq = dyn_request(last ? k : k + 1);
      if (T < n) {
        zb[T] = xb;
        if (!open) dxv[T] = xb - d.xs;
        Xd[(size_t)k * n + T] = xb;
      }
      wsync();
      double uv = 0.0;
      if (T < m) {
        double acc = d.us;
        if (!open) {
          acc += alpha * d.dgv;
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += d.kp[u] * dxv[u];   // dxv is zero beyond n
          if (n > 16) acc = dot_strided(Kgi + (size_t)k * n * m + (size_t)16 * m + T, m, dxv + 16, n - 16, acc);
          Ud[(size_t)k * m + T] = acc;
        }
        uv = acc;
        zb[np + T] = acc;
      }
      wsync();
      eval_knot(k, false, xb, uv, d, J, viol);
      lim = lim || (T < n && !(fabs(xb) <= P.o.max_state_value)) || (T < m && !(fabs(uv) <= P.o.max_control_value));
      if (!open) {
        chg = chg | (T < n && xb != d.xs) | (T < m && uv != d.us);
        big = big | (T < n && !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs)))) | (T < m && !(fabs(uv - d.us) <= 1e-7 * (1.0 + fabs(d.us))));
      }
      double xn = 0.0;
      if (T < n) xn = ahead ? next_state_parked(park[k & 1]) : next_state(k);
      if (ahead) dyn_park_linear(dq, park[(k & 1) ^ 1]);
      wsync();
      xb = xn;
      d = dn;
    }
    if (T < n) {
      zb[T] = xb;
      Xd[(size_t)(N - 1) * n + T] = xb;
    }
    wsync();
    eval_knot(N - 1, true, xb, 0.0, d, J, viol);
    lim = lim || (T < n && !(fabs(xb) <= P.o.max_state_value));
    if (!open) {
      chg = chg | (T < n && xb != d.xs);
      big = big | (T < n && !(fabs(xb - d.xs) <= 1e-7 * (1.0 + fabs(d.xs))));
    }
    block_sync();  // phase end: the trajectory written to global memory is read by other lanes next
    RollOut r;
    r.J = wave_sum(J);
    r.cmax = wave_max(viol);
    r.limit = wave_any(lim);
    r.unchanged = !open && !wave_any(chg);
    r.tiny = !open && !wave_any(big);
    return r;
  }

  __device__ __forceinline__ void load_dyn(int k) {
    const double *A_ = Ak(k), *B_ = Bk(k);
    Walk w = start(by_n);
    for (int e = T; e < n * n; e += 64, step(by_n, w)) G[w.r * ly.ldg + w.q] = A_[e];
    w = start(by_n);
    for (int e = T; e < n * m; e += 64, step(by_n, w)) G[w.r * ly.ldg + np + w.q] = B_[e];
  }

  static __device__ __forceinline__ void box_expand(double mu, double z, double zmx, double zmn, double lhi, double llo,
                                                    double& q, double& h) {
    if (zmx < 1e300) {
      const double c = z - zmx;
      const bool a = (c >= 0.0) || (lhi > 0.0);
      q += lhi + (a ? mu * c : 0.0);
      h += a ? mu : 0.0;
    }
    if (zmn > -1e300) {
      const double c = zmn - z;
      const bool a = (c >= 0.0) || (llo > 0.0);
      q -= llo + (a ? mu * c : 0.0);
      h += a ? mu : 0.0;
    }
  }

  // Active-set code of lane T at a knot (one byte of P.aset): bits 0-1 the box sides of x_T that enter the Hessian, bits 2-3
  // those of u_T, bit 4 generic row T active.  The backward pass (knots in descending order) and a rollout (ascending) write
  // the same byte for the same active set; the set of the gains in memory is compared with the sets of unrelated later
  // solves, launch after launch (gain reuse): byte for byte, not through a hash.
  static __device__ __forceinline__ unsigned box_code(double z, double zmx, double zmn, double lhi, double llo, bool on) {
    const bool bh = on & (zmx < 1e300), bl = on & (zmn > -1e300);
    const bool ah = ((z - zmx) >= 0.0) | (lhi > 0.0), al = ((zmn - z) >= 0.0) | (llo > 0.0);
    return ((bh & ah) ? 1u : 0u) | ((bl & al) ? 2u : 0u);
  }

  // box_expand with selects only: same values
  __device__ __forceinline__ void box_expand_sel(double z, double zmx, double zmn, double lhi, double llo, bool on, double& q, double& h) const {
    const double chi = z - zmx, clo = zmn - z;
    const bool bh = on & (zmx < 1e300), bl = on & (zmn > -1e300);
    const bool ah = (chi >= 0.0) | (lhi > 0.0), al = (clo >= 0.0) | (llo > 0.0);
    q += bh ? lhi + (ah ? mu * chi : 0.0) : 0.0;
    h += (bh & ah) ? mu : 0.0;
    q -= bl ? llo + (al ? mu * clo : 0.0) : 0.0;
    h += (bl & al) ? mu : 0.0;
  }

  // gradient of the box terms of one element (what box_expand adds to q), branch-free
  __device__ __forceinline__ void box_grad(double z, double zmx, double zmn, double lhi, double llo, bool on, double& q) const {
    const double chi = z - zmx, clo = zmn - z;
    const bool bh = on & (zmx < 1e300), bl = on & (zmn > -1e300);
    const bool ah = (chi >= 0.0) | (lhi > 0.0), al = (clo >= 0.0) | (llo > 0.0);
    q += bh ? lhi + (ah ? mu * chi : 0.0) : 0.0;
    q -= bl ? llo + (al ? mu * clo : 0.0) : 0.0;
  }

  // cost_expansion! at knot k of plane cur: gradient qz, Hessian diagonal hz (padded z layout), and
  // for the generic rows the tables Ac, DA = diag(I_mu) Ac with A'g already added to qz
  __device__ __forceinline__ void expansion(int k, bool term, const KnotLd& d, unsigned& code) {
    const bool bx = box_at(k);
    code = box_code(d.xs, cxmax, cxmin, d.lxh, d.lxl, bx & (T < n)) | (box_code(d.us, cumax, cumin, d.luh, d.lul, bx & (T < m) & !term) << 2);
    if constexpr (SM && MC >= 12) {  // (compiled into the m > 8 classes only: in the others its 64 transient registers pushed
      // the backward pass of the box-only sweeps -- n = 16, (12,6) -- into scratch, -3 % there)
      // n, m <= 16, at most 16 linear rows in a resident table, a stage knot: the whole expansion in the first DPP row,
      // lane T with x_T, u_T and row T of the table -- the row values A_c z + b and the gradient A_c' g take their vector
      // operand from the other lanes by row broadcast instead of three LDS hand-overs with a barrier each.  Terms in
      // the order of the LDS version below: bit-identical.  Branch-free (every lane executes the DPP instructions).
      if (!term && Pn > 0 && Pn <= 16 && da_on_the_fly()) {
        const bool isx = T < n, isu = T < m, isr = T < Pn;
        const int Tn = isx ? T : n - 1, Tm = isu ? T : m - 1, Tr = isr ? T : 0, lastrow = Pn - 1;
        const double x = isx ? d.xs : 0.0, u = isu ? d.us : 0.0;
        double qx = cwx * (x - d.xr), hx = cwx, qu = cwu * (u - d.ur), hu = cwu;
        box_expand_sel(x, cxmax, cxmin, d.lxh, d.lxl, isx & bx, qx, hx);
        box_expand_sel(u, cumax, cumin, d.luh, d.lul, isu & bx, qu, hu);
        const lds_d* acrow = (const lds_d*)Ac + Tr * ly.ldg;
        const lds_d* accol = (const lds_d*)Ac;
        double ac[32], acx[16], acu[16];
#pragma unroll
        for (int c = 0; c < 32; ++c) ac[c] = acrow[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = r < Pn ? r : lastrow;  // rows beyond Pn: g_r = 0
          acx[r] = accol[rr * ly.ldg + Tn];
          acu[r] = accol[rr * ly.ldg + 16 + Tm];
        }
        double v = RowDot<16>::run(ac, x, d.bc);
        v = RowDot<16>::run(ac + 16, u, v);
        const bool on = isr & (d.ct != 0), act = (d.ct == 1) | (v >= 0.0) | (d.lam > 0.0);
        const double g = on ? d.lam + (act ? mu * v : 0.0) : 0.0;
        const double D = (on & act) ? mu : 0.0;
        code |= (on & act) ? 16u : 0u;
        qx += RowDot<16>::run(acx, g, 0.0);
        qu += RowDot<16>::run(acu, g, 0.0);
        if (isx) {
          zb[T] = x;
          qz[T] = qx;
          hz[T] = hx;
        }
        if (isu) {
          zb[np + T] = u;
          qz[np + T] = qu;
          hz[np + T] = hu;
        }
        if (T < Pp) {
          gr[T] = g;
          Dr[T] = D;
        }
        wsync();
        return;
      }
    }
    if (T < n) {
      const double x = d.xs;
      zb[T] = x;
      const double w = term ? cwfx : cwx;
      double q = w * (x - d.xr), h = w;
      if (bx) box_expand(mu, x, cxmax, cxmin, d.lxh, d.lxl, q, h);
      qz[T] = q;
      hz[T] = h;
    }
    if (T < m) {
      double q = 0.0, h = 0.0, u = 0.0;
      if (!term) {
        u = d.us;
        const double w = cwu;
        q = w * (u - d.ur);
        h = w;
        if (bx) box_expand(mu, u, cumax, cumin, d.luh, d.lul, q, h);
      }
      zb[np + T] = u;
      qz[np + T] = q;
      hz[np + T] = h;
    }
    wsync();
    if (Pn > 0) {
      double v = 0.0, lam = 0.0;
      int ct = 0;
      if (T < Pp) {
        double g = 0.0, D = 0.0;
        if (T < Pn) {
          ct = d.ct;
          if (ct != 0) {
            v = row_value(k, T, term);
            lam = d.lam;
            if (ct != 3) {
              const bool act = (ct == 1) || (v >= 0.0) || (lam > 0.0);
              g = lam + (act ? mu * v : 0.0);
              D = act ? mu : 0.0;
              code |= act ? 16u : 0u;
            }
          }
        }
        gr[T] = g;
        Dr[T] = D;
        if (P.ncone > 0) {
          cvv[T] = v;
          cll[T] = lam;
        }
      }
      wsync();
      if (P.ncone > 0) {
        if (T < Pn) {
          double h4[4] = {0.0, 0.0, 0.0, 0.0};
          if (ct == 3) {
            const Cone c = cone_eval<true>(myc0, mycp, T - myc0);
            gr[T] = c.g;
#pragma unroll
            for (int u = 0; u < 4; ++u) h4[u] = c.h[u];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) Hc[T * 4 + u] = h4[u];
        }
        wsync();
      }
      const double* At = AconTi + (size_t)k * nz * Pn;
      const bool resident = (P.con_static & 4) && !term;
      Walk w = start(by_P);
      if (!(resident && da_on_the_fly())) {  // (otherwise the products scale the rows of Ac by Dr themselves)
        for (int e = T; e < nz * Pn; e += 64, step(by_P, w)) {
          const int j = w.q, r = w.r;
          const int c = j < n ? j : np + (j - n);
          const double a = resident ? Ac[r * ly.ldg + c] : ((term && j >= n) ? 0.0 : At[e]);
          if (!resident) Ac[r * ly.ldg + c] = a;
          DA[r * ly.ldg + c] = Dr[r] * a;
        }
        wsync();
      }
      if (P.ncone > 0) {  // cone rows: DA = H_block * A over the rows of the cone
        w = start(by_P);
        for (int e = T; e < nz * Pn; e += 64, step(by_P, w)) {
          const int j = w.q, r = w.r;
          const int cp = P.rowcp[r];
          if (cp > 0) {
            const int c0 = P.rowc0[r], c = j < n ? j : np + (j - n);
            double acc = 0.0;
            for (int q = 0; q < cp; ++q) acc += Hc[r * 4 + q] * Ac[(c0 + q) * ly.ldg + c];
            DA[r * ly.ldg + c] = acc;
          }
        }
        wsync();
      }
      for (int c = T; c < nzp; c += 64) {
        double acc = 0.0;
        if constexpr (SM) {  // four rows at a time, the eight reads ahead of the FMAs (gr is zero beyond Pn up to Pp)
          for (int r0 = 0; r0 < Pp; r0 += 4) {
            double a[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              a[u] = Ac[(r0 + u) * ly.ldg + c];
              g[u] = gr[r0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += a[u] * g[u];
          }
        } else {
          for (int r = 0; r < Pn; ++r) acc += Ac[r * ly.ldg + c] * gr[r];
        }
        qz[c] += acc;
      }
      wsync();
    }
  }

  // n, m <= 16, linear rows only, table resident in LDS: D A_c is never written out -- the A fragment of the products
  // A_c' (D A_c) is the B fragment scaled by the lane's D_r
  __device__ __forceinline__ bool da_on_the_fly() const { return SM && P.ncone == 0 && (P.con_static & 4); }

  // value of v on lane `lane` (wave-uniform index), to every lane: two v_readlane_b32
  static __device__ __forceinline__ double lane_bcast(double v, int lane) {
    const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
  }

  // Quu_reg = L D L' and K = -Quu_reg^-1 [Qux | Qu], every lane for itself (m <= MP <= 16): the lower triangle of
  // Quu_reg is read from LDS by all lanes (one broadcast read per element), factored redundantly in registers
  // (right-looking, in place) and lane c solves for column c of [Qux | Qu].  No cross-lane traffic, no branches, no
  // selects: the caller leaves the identity in rows >= m of Quu_reg (the pads are exact zeros, the diagonal is set
  // to 1) and zeros in rows >= m of [Qux | Qu], so the unrolled MP x MP code is exact for any m <= MP.  History: the LDS version walked dependent read-modify-write chains (40 k cycles per
  // knot at m = 12); a version that kept column b on lane b and handed the scalars round with v_readlane was
  // ~2800 instructions, 13 k cycles with one wave per SIMD; this one is ~900.  Returns true if a pivot is not positive.
  static __device__ __forceinline__ double rcp_nr(double x) {  // 1/x to full FP64 accuracy (x > 0, normal range)
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
  }
  // The same factorisation and the same solves -- element for element the same operands in the same order -- with L spread
  // over the lanes instead of whole in every lane: lane r of each 16-lane row holds row r of the triangle (MP doubles instead
  // of MP (MP + 1) / 2: 24 VGPRs against 156 at MP = 12), an element of another row arrives as the DPP operand of the FMA
  // that needs it (row_newbcast; written with plain operators so that hipcc contracts the products into the same FMAs as in
  // factor_solve_lane -- with explicit __builtin_fma the results differed in the last bits).  Right-looking step j: the pivot and f_c = L_cj come from lanes j and c, every lane updates
  // its own row: MP (MP - 1) / 2 FMAs per lane where the replicated form had MP (MP^2 - 1) / 6.  All four rows of the wave hold
  // the same copy, so a lane solves for its column (c0 + T) whatever row it sits in.
  template <int MP>
  __device__ __forceinline__ bool factor_solve_dpp(double* facout) {
    const int ldh = ly.ldh, ldu = ly.ldu;
    const lds_d* Hl = (const lds_d*)Huu;
    const int r = T & 15;
    double a[MP], inv[MP];
#pragma unroll
    for (int c = 0; c < MP; ++c) a[c] = Hl[(r < MP ? r : MP - 1) * ldu + c];  // row r (its part above the diagonal is never used)
    bool fail = false;
    wfor<0, MP>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const double dj = row_bcast<J>(a[J]);
      fail = fail | !(dj > 0.0);
      inv[J] = rcp_nr(dj);
      const double f = a[J] * inv[J];  // L[r][J] (meaningful for r > J)
      wfor<J + 1, MP>([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        a[Cc] -= a[J] * row_bcast<Cc>(f);   // a[r][c] -= a[r][J] f_c, f_c = L[c][J] from lane c
      });
      a[J] = (r > J) ? f : a[J];
    });
    if (fail) return true;  // wave-uniform: every lane saw the same pivots
    if (facout != nullptr) {  // the factor of this knot, for the costate sweep: lane r writes row r (1 / D_r on the diagonal)
      double diag = 0.0;
      wfor<0, MP>([&](auto jc) { diag = (r == decltype(jc)::value) ? inv[decltype(jc)::value] : diag; });
#pragma unroll
      for (int c = 0; c < MP; ++c)
        if (T < MP && c <= T) facout[T * (T + 1) / 2 + c] = (c == T) ? diag : a[c];
    }
    for (int c0 = 0; c0 <= np; c0 += 64) {
      const int c = c0 + T;
      const bool mine = (c < n) || (c == np);
      const lds_d* hc = (const lds_d*)Hux + (mine ? c : 0);
      double q[MP];
#pragma unroll
      for (int i = 0; i < MP; ++i) q[i] = hc[i * ldh];
      wfor<0, MP>([&](auto kc) {            // forward: L y = b  (L[i][k] = element k of lane i)
        constexpr int K = decltype(kc)::value;
        wfor<K + 1, MP>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          q[I] -= row_bcast<I>(a[K]) * q[K];
        });
      });
#pragma unroll
      for (int i = 0; i < MP; ++i) q[i] *= inv[i];
      wfor<0, MP>([&](auto kr) {            // backward: L' x = y  (L[k][i] = element i of lane k), k = MP - 1 .. 0
        constexpr int K = MP - 1 - decltype(kr)::value;
        wfor<0, K>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          q[I] -= row_bcast<K>(a[I]) * q[K];
        });
      });
      if (mine) {
#pragma unroll
        for (int i = 0; i < MP; ++i) Kl[i * ldh + c] = 0.0 - q[i];  // rows >= m: zero
      }
    }
    return false;
  }

  template <int MP>
  __device__ __forceinline__ bool factor_solve_lane(double* facout) {
    if constexpr (MP >= ALTRO_WIDE_FACTOR_DPP) return factor_solve_dpp<MP>(facout);
    const int ldh = ly.ldh, ldu = ly.ldu;
    const lds_d* Hl = (const lds_d*)Huu;
    double a[MP][MP], inv[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) a[i][j] = Hl[i * ldu + j];
    }
    bool fail = false;
#pragma unroll
    for (int j = 0; j < MP; ++j) {
      const double dj = a[j][j];
      fail = fail | !(dj > 0.0);
      inv[j] = rcp_nr(dj);
      double f[MP];
#pragma unroll
      for (int c = j + 1; c < MP; ++c) f[c] = a[c][j] * inv[j];  // L[c][j]
#pragma unroll
      for (int i = j + 1; i < MP; ++i)
#pragma unroll
        for (int c = j + 1; c <= i; ++c) a[i][c] -= a[i][j] * f[c];
#pragma unroll
      for (int c = j + 1; c < MP; ++c) a[c][j] = f[c];
    }
    if (fail) return true;  // wave-uniform: every lane saw the same pivots
    if (facout != nullptr && T == 0) {  // the factor of this knot, for the costate sweep (78 doubles at MP = 12, one lane)
#pragma unroll
      for (int i = 0; i < MP; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) facout[i * (i + 1) / 2 + j] = a[i][j];
        facout[i * (i + 1) / 2 + i] = inv[i];
      }
    }
    for (int c0 = 0; c0 <= np; c0 += 64) {
      const int c = c0 + T;
      const bool mine = (c < n) || (c == np);
      const lds_d* hc = (const lds_d*)Hux + (mine ? c : 0);
      double q[MP];
#pragma unroll
      for (int r = 0; r < MP; ++r) q[r] = hc[r * ldh];
#pragma unroll
      for (int k = 0; k < MP; ++k)          // forward: L y = b
#pragma unroll
        for (int i = k + 1; i < MP; ++i) q[i] -= a[i][k] * q[k];
#pragma unroll
      for (int r = 0; r < MP; ++r) q[r] *= inv[r];
#pragma unroll
      for (int k = MP - 1; k >= 0; --k)     // backward: L' x = y
#pragma unroll
        for (int i = 0; i < k; ++i) q[i] -= a[k][i] * q[k];
      if (mine) {
#pragma unroll
        for (int r = 0; r < MP; ++r) Kl[r * ldh + c] = 0.0 - q[r];  // rows >= m: zero
      }
    }
    return false;
  }

  // backwardpass! (oracle backward_pass).  Returns true if a pivot of Quu + rho I was not positive.
  __device__ __forceinline__ bool backward(double& dV1, double& dV2) {
    phase_begin();
    const int ldg = ly.ldg, lds = ly.lds, ldh = ly.ldh, ldu = ly.ldu;
    for (int e = T; e < np * lds; e += 64) S[e] = 0.0;
    wsync();
    // the active set this pass expands with (a failed pass leaves bw_ok false).  (Addressed from P.aset at every store: a
    // pointer kept across the knot loop sends hipcc 7.2 into "Illegal instruction detected" in the two-wave instantiations.)
    const unsigned aoff = (unsigned)(((size_t)inst * 3 + (size_t)bwp) * (size_t)N * 64 + (size_t)T);
    unsigned kcode = 0u;
    expansion(N - 1, true, load_knot(N - 1, true, 2, Xp(cur), Up(cur)), kcode);
    P.aset[aoff + (unsigned)(N - 1) * 64u] = (unsigned char)kcode;
    if (T < n) {
      S[T * lds + T] = hz[T];
      sv[T] = qz[T];
    }
    wsync();
    if (Pn > 0) gemm_tn<true>(S, lds, DA, ldg, Ac, ldg, np, np, Pp);
    if (Pn > 0 && P.con_static) {
      wsync();
      build_static_Ac();  // the terminal knot zeroed the control columns
    }
    dV1 = 0.0;
    dV2 = 0.0;
    wsync();
    KnotLd kd = load_knot(N - 2, false, 2, Xp(cur), Up(cur));
    bool dbig = false;
    const bool ahead = dyn_ahead();
    if (ahead) load_dyn(N - 2);
    for (int k = N - 2; k >= 0; --k) {
      WSTAMP(const long long b0 = wstamp();)
      const KnotLd kdn = load_knot(k > 0 ? k - 1 : 0, false, 2, Xp(cur), Up(cur));  // operands of the next knot, one knot ahead
      DynRegs dq = {};
      if (ahead) dq = dyn_request(k > 0 ? k - 1 : 0);
      else if (P.ltv) load_dyn(k);
      expansion(k, false, kd, kcode);  // ends with a barrier
      P.aset[aoff + (unsigned)k * 64u] = (unsigned char)kcode;
      const double us_k = kd.us;
      kd = kdn;
      WSTAMP(const long long b1 = wstamp(); t_a += b1 - b0;)
      WSTAMP(long long b3 = 0;)
      if constexpr (SM) {
        // n, m <= 16: every operand is one 16 x 16 tile and the whole chain of products stays in registers.  The C/D
        // layout of v_mfma_f64_16x16x4 (row = lane / 16 + 4 reg, column = lane % 16) is exactly the B fragment of
        // k-step reg, and the fragment of G that serves as B operand of W = S G serves as A operand of G' W: twelve
        // LDS reads feed all five products, W never touches LDS, Qxx waits in its accumulator for Qux' K.
        for (int c = T; c < nzp; c += 64) qv[c] = dot_lds(G + c, ldg, sv, 1, np, qz[c]);  // Q_z = l_z + [A B]' s
        WSTAMP(const long long tg = wstamp();)
        const int q4 = T >> 4, r16 = T & 15;
        const lds_d* Sl = (const lds_d*)S + q4 * lds + r16;
        const lds_d* Gl = (const lds_d*)G + q4 * ldg + r16;
        double sA[4], gA[4], gB[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sA[r] = Sl[4 * r * lds];
          gA[r] = Gl[4 * r * ldg];
          gB[r] = Gl[4 * r * ldg + 16];
        }
        if (ahead) dyn_park_G(dq);  // LDS operations of a wave execute in order: the reads above are ahead of these writes
        d4_t wA = {0.0, 0.0, 0.0, 0.0}, wB = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wA = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r], gA[r], wA, 0, 0, 0);  // W = S [A B]
          wB = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r], gB[r], wB, 0, 0, 0);
        }
        d4_t qxx = {0.0, 0.0, 0.0, 0.0}, qux = {0.0, 0.0, 0.0, 0.0}, quu = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qxx = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[r], wA[r], qxx, 0, 0, 0);  // A' S A
          qux = __builtin_amdgcn_mfma_f64_16x16x4f64(gB[r], wA[r], qux, 0, 0, 0);  // B' S A
          quu = __builtin_amdgcn_mfma_f64_16x16x4f64(gB[r], wB[r], quu, 0, 0, 0);  // B' S B
        }
        if (Pn > 0) {  // + A_c' diag(I_mu) A_c of the generic rows, same chains
          const lds_d* Dl = (const lds_d*)DA + q4 * ldg + r16;
          const lds_d* Al = (const lds_d*)Ac + q4 * ldg + r16;
          const bool fly = da_on_the_fly();
          for (int r = 0; 4 * r < Pp; ++r) {  // up to 64 rows
            const double ax = Al[4 * r * ldg], au = Al[4 * r * ldg + 16];
            double dx, du;
            if (fly) {
              const double dk = Dr[4 * r + q4];
              dx = dk * ax;
              du = dk * au;
            } else {
              dx = Dl[4 * r * ldg];
              du = Dl[4 * r * ldg + 16];
            }
            qxx = __builtin_amdgcn_mfma_f64_16x16x4f64(dx, ax, qxx, 0, 0, 0);
            qux = __builtin_amdgcn_mfma_f64_16x16x4f64(du, ax, qux, 0, 0, 0);
            quu = __builtin_amdgcn_mfma_f64_16x16x4f64(du, au, quu, 0, 0, 0);
          }
        }
        WSTAMP(t_gemm += wstamp() - tg;)
        {
          const double hx = hz[r16], hu = hz[np + r16] + (r16 < m ? rho : 1.0);  // diagonals: l_xx, l_uu + rho (bp_reg_type = :control); identity in the pad rows
          lds_d* Hq = (lds_d*)Hux + q4 * ldh + r16;
          lds_d* Uq = (lds_d*)Huu + q4 * ldu + r16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool dg = (q4 + 4 * r) == r16;
            qxx[r] += dg ? hx : 0.0;
            Hq[4 * r * ldh] = qux[r];
            Uq[4 * r * ldu] = quu[r] + (dg ? hu : 0.0);
          }
          if (T < 16) Hux[T * ldh + np] = qv[np + T];  // Qu rides as column np of Qux (zeros in the pad rows)
        }
        wsync();
        WSTAMP(const long long b2 = wstamp(); t_b += b2 - b1;)
        if (factor_solve_lane<MC>(P.fac + ((size_t)inst * N + k) * (MC * (MC + 1) / 2))) return true;
        wsync();
        WSTAMP(b3 = wstamp(); t_c += b3 - b2;)
        {  // dV = (d'Qu, 1/2 d'Quu d) with Quu d = -Qu - rho d
          double p1 = 0.0, p2 = 0.0;
          if (T < m) {
            const double d = Kl[T * ldh + np];
            p1 = d * Hux[T * ldh + np];
            p2 = d * d;
            dbig = dbig | !(fabs(d) <= 1e-9 * (1.0 + fabs(us_k)));
          }
          double t1 = 0.0, dd = 0.0;
          for (int a = 0; a < m; ++a) {
            t1 += lane_bcast(p1, a);
            dd += lane_bcast(p2, a);
          }
          dV1 += t1;
          dV2 += -0.5 * t1 - 0.5 * rho * dd;
        }
        // S = Qxx + Qux'K - rho K'K ; s = Qx + Qux'd - rho K'd
        {
          const lds_d* Hq = (const lds_d*)Hux + q4 * ldh + r16;
          const lds_d* Kq = (const lds_d*)Kl + q4 * ldh + r16;
          double hf[4], kf[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hf[r] = Hq[4 * r * ldh];
            kf[r] = Kq[4 * r * ldh];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) qxx = __builtin_amdgcn_mfma_f64_16x16x4f64(hf[r], kf[r], qxx, 0, 0, 0);
          if (rho != 0.0) {
            d4_t kk = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) kk = __builtin_amdgcn_mfma_f64_16x16x4f64(kf[r], kf[r], kk, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) qxx[r] -= rho * kk[r];
          }
        }
        if (T < n) {
          double acc = dot_lds(Hux + T, ldh, Kl + np, ldh, mp, qv[T]);  // rows >= m are zero
          if (rho != 0.0) acc -= rho * dot_lds(Kl + T, ldh, Kl + np, ldh, mp, 0.0);
          sv[T] = acc;
        }
        {  // S <- (S + S')/2 through LDS: every lane averages its four elements with their mirror images
          lds_d* Sq = (lds_d*)S + q4 * lds + r16;
          const lds_d* St = (const lds_d*)S + r16 * lds + q4;
#pragma unroll
          for (int r = 0; r < 4; ++r) Sq[4 * r * lds] = qxx[r];
          wsync();
          double tr[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) tr[r] = St[4 * r];
          wsync();
#pragma unroll
          for (int r = 0; r < 4; ++r) Sq[4 * r * lds] = 0.5 * (qxx[r] + tr[r]);
        }
      } else {
        // Q_z = l_z + [A B]' s
        for (int c = T; c < nzp; c += 64) qv[c] = dot_lds(G + c, ldg, sv, 1, np, qz[c]);  // rows >= n of G and sv are zero
        WSTAMP(const long long tg = wstamp();)
        const bool coop = coop_on();
        const GemmDesc g_none = {};
        if (coop) {
          // the strips of Qux and Quu (one each at m <= 16, and Quu a lone tile) on waves 0 and 1, those of Qxx on 2 and 3
          const int wq = mp > 16 ? 4 : 1;
          coop_run(kCoopGemm, 1, gdesc(false, W, ldg, S, lds, G, ldg, np, nzp, np, 1.0, 0, 4), g_none, g_none);  // W = S [A B]
          coop_run(kCoopGemm, 3, gdesc(false, Hux, ldh, G + np, ldg, W, ldg, mp, np, np, 1.0, 0, wq),          // Qux = B' S A
                   gdesc(false, Huu, ldu, G + np, ldg, W + np, ldg, mp, mp, np, 1.0, 1, wq),                   // Quu = B' S B
                   gdesc(false, S, lds, G, ldg, W, ldg, np, np, np, 1.0, mp > 16 ? 0 : 2, mp > 16 ? 4 : 2));   // Qxx = A' S A
        } else {
          gemm_tn<false>(W, ldg, S, lds, G, ldg, np, nzp, np);  // W = S [A B]
          wsync();
          gemm_tn<false>(Huu, ldu, G + np, ldg, W + np, ldg, mp, mp, np);  // Quu = B' S B
          gemm_tn<false>(S, lds, G, ldg, W, ldg, np, np, np);              // Qxx = A' S A  (S is free: W is complete)
          gemm_tn<false>(Hux, ldh, G + np, ldg, W, ldg, mp, np, np);       // Qux = B' S A: last, Hux may lie inside W (lds_layout)
          wsync();
        }
        if (ahead) dyn_park_G(dq);  // nothing reads G any more at this knot
        WSTAMP(t_gemm += wstamp() - tg;)
        if (T < n) S[T * lds + T] += hz[T];
        if (T < m) Huu[T * ldu + T] += hz[np + T];
        if (T < mp) Hux[T * ldh + np] = qv[np + T];  // Qu rides as column np of Qux (zeros in the pad rows)
        wsync();
        if (Pn > 0) {
          if (coop) {
            coop_run(kCoopGemm, 3, gdesc(true, Hux, ldh, DA + np, ldg, Ac, ldg, mp, np, Pp, 1.0, 0, 1),
                     gdesc(true, Huu, ldu, DA + np, ldg, Ac + np, ldg, mp, mp, Pp, 1.0, 1, 1), gdesc(true, S, lds, DA, ldg, Ac, ldg, np, np, Pp, 1.0, 0, 4));
          } else {
            gemm_tn<true>(S, lds, DA, ldg, Ac, ldg, np, np, Pp);
            gemm_tn<true>(Hux, ldh, DA + np, ldg, Ac, ldg, mp, np, Pp);
            gemm_tn<true>(Huu, ldu, DA + np, ldg, Ac + np, ldg, mp, mp, Pp);
            wsync();
          }
        }
        WSTAMP(const long long b2 = wstamp(); t_b += b2 - b1;)
        if constexpr (MC == 0)
          for (int e = T; e < mp * ldh; e += 64) Kl[e] = Hux[e];  // the LDS solve works in place on a copy of [Qux | Qu]
        if (T < m) Huu[T * ldu + T] += rho;  // bp_reg_type = :control
        if (MC > 0 && T >= m && T < mp) Huu[T * ldu + T] = 1.0;  // factor_solve_lane: identity in the pad rows
        wsync();
        if constexpr (MC > 0) {
          if (factor_solve_lane<MC>(P.fac + ((size_t)inst * N + k) * (MC * (MC + 1) / 2))) return true;
        } else {
          // Quu_reg = L D L' in place in LDS (unit L below the diagonal, D on it)
          for (int j = 0; j < m; ++j) {
            const double dj = Huu[j * ldu + j];
            if (!(dj > 0.0)) return true;  // wave-uniform
            if (T > j && T < m) Huu[T * ldu + j] *= 1.0 / dj;
            wsync();
            if (T > j && T < m) {
              const double li = Huu[T * ldu + j];
              for (int c = j + 1; c <= T; ++c) Huu[T * ldu + c] -= li * Huu[c * ldu + j] * dj;
            }
            wsync();
          }
          // K = -Quu_reg^-1 Qux, d = -Quu_reg^-1 Qu: one lane per column
          for (int c = T; c <= np; c += 64) {
            if (c < n || c == np) {
              for (int i = 0; i < m; ++i) {
                double v = Kl[i * ldh + c];
                for (int kk = 0; kk < i; ++kk) v -= Huu[i * ldu + kk] * Kl[kk * ldh + c];
                Kl[i * ldh + c] = v;
              }
              for (int i = 0; i < m; ++i) Kl[i * ldh + c] /= Huu[i * ldu + i];
              for (int i = m - 1; i >= 0; --i) {
                double v = Kl[i * ldh + c];
                for (int kk = i + 1; kk < m; ++kk) v -= Huu[kk * ldu + i] * Kl[kk * ldh + c];
                Kl[i * ldh + c] = v;
              }
              for (int i = 0; i < m; ++i) Kl[i * ldh + c] = -Kl[i * ldh + c];
            }
          }
        }
        wsync();
        WSTAMP(b3 = wstamp(); t_c += b3 - b2;)
        // dV = (d'Qu, 1/2 d'Quu d) with Quu d = -Qu - rho d
        {
          double p1 = 0.0, p2 = 0.0;
          if (T < m) {
            const double d = Kl[T * ldh + np];
            p1 = d * Hux[T * ldh + np];
            p2 = d * d;
            dbig = dbig | !(fabs(d) <= 1e-9 * (1.0 + fabs(us_k)));
          }
          double t1 = 0.0, dd = 0.0;
          for (int a = 0; a < m; ++a) {  // m terms in the oracle's order; v_readlane is far cheaper than a 6-step shuffle tree
            t1 += lane_bcast(p1, a);
            dd += lane_bcast(p2, a);
          }
          dV1 += t1;
          dV2 += -0.5 * t1 - 0.5 * rho * dd;
        }
        // S = Qxx + Qux'K - rho K'K ; s = Qx + Qux'd - rho K'd
        if (coop) {
          coop_run(kCoopGemm, rho != 0.0 ? 2 : 1, gdesc(true, S, lds, Hux, ldh, Kl, ldh, np, np, mp, 1.0, 0, 4),
                   gdesc(true, S, lds, Kl, ldh, Kl, ldh, np, np, mp, -rho, 0, 4), g_none);
        } else {
          gemm_tn<true>(S, lds, Hux, ldh, Kl, ldh, np, np, mp);
          if (rho != 0.0) gemm_tn<true>(S, lds, Kl, ldh, Kl, ldh, np, np, mp, -rho);
        }
        for (int c = T; c < n; c += 64) {
          double acc = dot_lds(Hux + c, ldh, Kl + np, ldh, mp, qv[c]);  // rows >= m are zero
          if (rho != 0.0) acc -= rho * dot_lds(Kl + c, ldh, Kl + np, ldh, mp, 0.0);
          sv[c] = acc;
        }
        wsync();
        if (coop) {
          coop_run(kCoopSym, 1, gdesc(false, S, lds, S, lds, S, lds, n, np, n, 1.0, 0, 4), g_none, g_none);  // S <- (S + S')/2 (Nn carries the padded size)
        } else if constexpr (NPC == 32 || NPC == 48) {
          // S <- (S + S')/2 without a branch: every lane reads its NPC^2 / 64 elements of the padded matrix and their mirror
          // images, then writes the averages (a + b = b + a: the element and its mirror get the same bits, as the loop below
          // gives them; the pad is zero and stays zero).  The loop below is a divergent branch with an LDS round trip in
          // each of its 16-36 iterations.
          constexpr int EPL = NPC * NPC / 64;
          lds_d* const Sl = (lds_d*)S;
          double own[EPL], mir[EPL];
#pragma unroll
          for (int u = 0; u < EPL; ++u) {
            const int e = T + 64 * u, i = e / NPC, j = e % NPC;
            own[u] = Sl[i * lds + j];
            mir[u] = Sl[j * lds + i];
          }
          wsync();
#pragma unroll
          for (int u = 0; u < EPL; ++u) {
            const int e = T + 64 * u, i = e / NPC, j = e % NPC;
            Sl[i * lds + j] = 0.5 * (own[u] + mir[u]);
          }
        } else {
          Walk w = start(by_n);
          for (int e = T; e < n * n; e += 64, step(by_n, w)) {  // S <- (S + S')/2
            const int i = w.q, j = w.r;
            if (i > j) {
              const double v = 0.5 * (S[i * lds + j] + S[j * lds + i]);
              S[i * lds + j] = v;
              S[j * lds + i] = v;
            }
          }
        }
      }
      double* Kk = Kgi + (size_t)k * n * m;
      Walk w = start(by_m);
      for (int e = T; e < n * m; e += 64, step(by_m, w)) Kk[e] = Kl[w.r * ldh + w.q];
      if (T < m) dgi[(size_t)k * m + T] = Kl[T * ldh + np];
      wsync();
      WSTAMP(t_d += wstamp() - b3;)
    }
    dtiny = !wave_any(dbig);
    return false;
  }

  __device__ __forceinline__ void reg_update(bool increase) {
    const altro_opts& o = P.o;
    if (increase) {
      drho = fmax(drho * o.bp_reg_increase_factor, o.bp_reg_increase_factor);
      rho = fmax(rho * drho, o.bp_reg_min);
    } else {
      drho = fmin(drho / o.bp_reg_increase_factor, 1.0 / o.bp_reg_increase_factor);
      rho = rho * drho * ((rho * drho > o.bp_reg_min) ? 1.0 : 0.0);
    }
  }

  // gradient_todorov! on plane cur
  __device__ __forceinline__ double todorov() const {
    const double* Us = Up(cur);
    double acc = 0.0;
    for (int k = 0; k < N - 1; ++k) {
      double v = 0.0;
      if (T < m) v = fabs(dgi[(size_t)k * m + T]) / (fabs(Us[(size_t)k * m + T]) + 1.0);
      acc += wave_max(v);
    }
    return acc / (double)(N - 1);
  }

  // solve!(::iLQRSolver) (oracle ilqr_solve); returns the final cost, cmax by reference
  __device__ __forceinline__ double ilqr(double cost_tol, double grad_tol, double& cmax) {
    const altro_opts& o = P.o;
    rho = o.bp_reg_initial;
    drho = 0.0;
    dj_zero = 0;
    WSTAMP(const long long ts0 = wstamp();)
    RollOut r0 = do_rollout(true, 0.0);
    WSTAMP(t_ro += wstamp() - ts0;)
    nro++;
    if (r0.limit) {
      status = ALTRO_STATE_LIMIT;
      cmax = __builtin_inf();
      return __builtin_inf();
    }
    double J_prev = r0.J, J = r0.J;
    cmax = r0.cmax;
    // Gain reuse (default mode; n > 16, box-only, time-invariant: a first-order pass is ~5 % of a backward pass there).
    // Inside a fixed active set and penalty the problem is LQ: K_k and Quu_k do not depend on the trajectory, so if the
    // pass whose gains and factors are in memory ran at this penalty, without regularisation, and saw the active set the
    // trajectory in plane cur has (hashes), a new pass would return the same K -- also across the MPC solves of a
    // launch (60-67 % of the headline-type solves end with the active set they started with).  Such an iteration
    // takes K from memory and its feedforward terms and expected decrease from adjoint_lds(full).
    constexpr bool kReuse = (MC > 0 && !SM) || kReuseRow;
    const bool reuse_class = kReuse && !o.strict && Pn == 0 && !P.ltv && (!SM || row_rollouts());
    if (reuse_class) {
      { const int t_ = qp; qp = ap; ap = t_; }  // the open-loop rollout's set of plane cur, with the current duals
      qvalid = true;
    } else {
      qvalid = false;  // the duals / the penalty may have changed since the hash was taken
      bw_plain = false;
    }
    for (int it = 0; it < o.iterations_inner; ++it) {
      double dV1 = 0.0, dV2 = 0.0;
      bool gave_up = false;
      bool swept = false;
      if constexpr (kReuse) {
        if (reuse_class && bw_ok && bw_plain && qvalid && rho == 0.0 && mu == bw_mu && !sets_differ()) {
          phase_begin();
          WSTAMP(const long long ts = wstamp();)
          if constexpr (SM) dtiny = do_grad_adjoint_row_full(dV1, dV2);
          else dtiny = do_adjoint_lds(true, dV1, dV2);
          WSTAMP(t_td += wstamp() - ts;)
          swept = true;
          ngs++;
        }
      }
      // Confirmation by the costate sweep (default mode; altro_opts.strict = 1 never takes it): from the second
      // iteration of an inner solve on, if the last backward pass ran without regularisation and the accepted step crossed
      // no active-set boundary (hashes of the pass and of the rollout), the first-order sweep decides whether a
      // backward pass here would return feedforward terms at rounding level.  If so the
      // iteration is booked as converged without that pass; the gains in memory are its K, the feedforward terms
      // are set to zero.
      bool gconf = false;
      if constexpr (MC > 0) {
        const bool can_sweep = SM ? row_rollouts() : (Pn == 0 && !P.ltv);
        const bool tryg = !swept && !(kReuse && reuse_class) && !o.strict && it >= 1 && bw_plain && qvalid && rho == 0.0 && can_sweep &&
                          !sets_differ() && (grad_tol > 1e-8) && (cost_tol > 1e-10 * (1.0 + fabs(J_prev)));
        if (tryg) {
          phase_begin();
          WSTAMP(const long long ts = wstamp();)
          if constexpr (SM) {
            gconf = do_grad_adjoint_row();
          } else {
            gconf = do_adjoint_lds(false, dV1, dV2);
          }
          WSTAMP(t_td += wstamp() - ts;)
          if (gconf) {
            ngc++;
            for (int k = 0; k < N - 1; ++k)
              if (T < m) dgi[(size_t)k * m + T] = 0.0;
            block_sync();
          }
        }
      }
      if (!gconf && !swept) nbw++;   // once per iteration: a pass restarted with more regularisation is the same iteration's pass
      while (!gconf && !swept) {  // regularisation restarts
        const bool plain = rho == 0.0;
        WSTAMP(const long long ts = wstamp();)
        const bool fail = do_backward(dV1, dV2);
        WSTAMP(t_bw += wstamp() - ts;)
        block_sync();  // phase end: gains written to global memory are read by other lanes in the rollout
        if (!fail) {
          bw_plain = plain;
          bw_ok = true;
          bw_mu = mu;
          break;
        }
        bw_ok = false;  // (a failed pass has overwritten part of the gains)
        if (rho >= o.bp_reg_max) { gave_up = true; break; }
        reg_update(true);
      }
      if (gave_up) { status = ALTRO_NO_PROGRESS; break; }
      if (!gconf && !swept) reg_update(false);
      // forwardpass!
      double alpha = 1.0, z = -1.0, cm = 0.0;
      J = __builtin_inf();
      int ls = 0;
      bool accepted = true;
      // Default-mode shortcuts, the ones of solve_dpp16.h (altro_opts.strict = 1 takes none of them).  Confirmation
      // iteration: every feedforward term of the backward pass is at rounding level, so the rollout, its line search
      // (20 fruitless halvings whenever the rounding of J falls the wrong way) and the Todorov sweep cannot change
      // the outcome -- the iteration is booked as converged on the trajectory it holds.
      const bool confirm = gconf || (!o.strict && dtiny && (grad_tol > 1e-8) && (cost_tol > 1e-10 * (1.0 + fabs(J_prev))));
      if (confirm && swept) {  // the confirmed iteration ran no backward pass: its feedforward terms are zero
        for (int k = 0; k < N - 1; ++k)
          if (T < m) dgi[(size_t)k * m + T] = 0.0;
        block_sync();
      }
      if (confirm) {
        J = J_prev;
        cm = cmax;
        accepted = false;
      }
      while (!confirm && (z <= o.line_search_lower_bound || z > o.line_search_upper_bound) && J >= J_prev) {
        if (ls > o.iterations_linesearch) {
          J = J_prev;
          cm = cmax;
          alpha = 0.0;
          accepted = false;
          reg_update(true);
          rho += o.bp_reg_fp;
          break;
        }
        WSTAMP(const long long ts = wstamp();)
        const RollOut r = do_rollout(false, alpha);
        WSTAMP(t_ro += wstamp() - ts;)
        if (ls == 0) nro++; else ntr++;
        if (r.limit) { ls++; alpha *= 0.5; continue; }
        J = r.J;
        cm = r.cmax;
        const double expected = -alpha * (dV1 + alpha * dV2);
        z = expected > 0.0 ? (J_prev - J) / expected : -1.0;
        ls++;
        alpha *= 0.5;
        // a trial that reproduced the trajectory bit for bit: every smaller step does too, the search would spin to
        // its limit and fail (exact, also in strict mode).  A trial that moved nothing by more than 1e-7 (1 + |z|)
        // while the model promises less than cost_tol / 1000: the iteration ends the same way whatever follows.
        if (r.unchanged || (!o.strict && r.tiny && !(expected > 1e-3 * cost_tol))) ls = o.iterations_linesearch + 1;
      }
      if (accepted) alpha *= 2.0;
      if (confirm) alpha = 1.0;
      if (!confirm) {  // the active-set hash of the trajectory now in plane cur: that of the accepted trial.  Only a full
        // step is followed by a confirmation sweep: inside an active set it lands on the minimiser of the quadratic model, a
        // damped one does not, and a sweep that does not confirm costs a quarter of the backward pass it fails to replace
        // (the gain-reuse class sweeps after any accepted step: there the pass always yields the iteration)
        // (the last trial run is the accepted one, if any is: its plane becomes the trajectory's)
        if (accepted) { const int t_ = qp; qp = ap; ap = t_; }
        if (!(kReuse && reuse_class)) qvalid = accepted && alpha == 1.0;
      }
      if (J > o.max_cost_value) { status = ALTRO_MAXIMUM_COST; break; }
      if (accepted) cur ^= 1;  // copy_trajectories!
      cmax = cm;
      const double dJ = fabs(J - J_prev);
      J_prev = J;
      if (iters < ALTRO_TRACE_LEN && T == 0) {
        P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + iters] = J;
        P.ctrace[(size_t)inst * ALTRO_TRACE_LEN + iters] = cm;
        P.atrace[(size_t)inst * ALTRO_TRACE_LEN + iters] = alpha;
      }
      iters++;
      dj_zero = (dJ == 0.0) ? dj_zero + 1 : 0;
      WSTAMP(const long long ttd = wstamp();)
      const bool conv = dJ < cost_tol && (confirm || todorov() < grad_tol);
      WSTAMP(t_td += wstamp() - ttd;)
      if (conv) break;
      if (iters >= o.iterations) { status = ALTRO_MAX_ITERATIONS; break; }
      if (dj_zero > o.dJ_counter_limit) { status = ALTRO_NO_PROGRESS; break; }
    }
    return J;
  }

  // dual_update! on plane cur (the penalty is scaled by the caller)
  __device__ __forceinline__ void dual_update() {
    phase_begin();
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    const double dmax = P.o.dual_max;
    for (int k = 0; k < N; ++k) {
      const bool term = k == N - 1;
      double xv = 0.0, uv = 0.0;
      if (T < n) { xv = Xs[(size_t)k * n + T]; zb[T] = xv; }
      if (T < m) { uv = term ? 0.0 : Us[(size_t)k * m + T]; zb[np + T] = uv; }
      wsync();
      if (box_at(k)) {
        for (int pass = 0; pass < 2; ++pass) {
          const bool on = pass == 0 ? (T < n) : (T < m && !term);
          if (!on) continue;
          const int j = pass == 0 ? T : n + T;
          const double z = pass == 0 ? xv : uv;
          const double zmx = pass == 0 ? cxmax : cumax, zmn = pass == 0 ? cxmin : cumin;
          if (zmx < 1e300) {
            double* l = Lbi + ((size_t)k * 2) * nz + j;
            *l = fmin(fmax(*l + mu * (z - zmx), 0.0), dmax);
          }
          if (zmn > -1e300) {
            double* l = Lbi + ((size_t)k * 2 + 1) * nz + j;
            *l = fmin(fmax(*l + mu * (zmn - z), 0.0), dmax);
          }
        }
      }
      int ct = 0;
      double rv = 0.0;
      if (T < Pn) {
        ct = P.ctype[(size_t)k * Pn + T];
        if (ct != 0) {
          double* l = Lci + (size_t)k * Pn + T;
          rv = row_value(k, T, term);
          if (P.ncone > 0) {
            cvv[T] = rv;
            cll[T] = *l;
          }
          if (ct != 3) *l = fmin(fmax(*l + mu * rv, ct == 1 ? -dmax : 0.0), dmax);
        }
      }
      if (P.ncone > 0) {
        wsync();
        if (ct == 3) Lci[(size_t)k * Pn + T] = cone_eval<false>(myc0, mycp, T - myc0).lam_new;
      }
      wsync();
    }
  }

  // one lane's column of a [K][stride] table moved one knot down: rows (k0, k1] -> [k0, k1).  The eight loads of a
  // chunk are issued before its first store; a plain copy loop waited a memory round trip per knot.
  static __device__ __forceinline__ void shift_column(double* col, size_t stride, int k0, int k1) {
    for (int k = k0; k < k1; k += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(k + 1 + u < k1 ? k + 1 + u : k1) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + u < k1) col[(size_t)(k + u) * stride] = v[u];
    }
  }

  // RD.shift_fill!(Z) and Altro.shift_fill!(conSet) on plane cur (oracle orc_shift_fill)
  __device__ __forceinline__ void shift(bool primal, bool dual) {
    phase_begin();
    if (primal) {
      if (T < n) shift_column(Xp(cur) + T, n, 0, N - 1);
      if (T < m) shift_column(Up(cur) + T, m, 0, N - 2);
    }
    if (dual) {
      for (int e = T; e < 2 * nz; e += 64) shift_column(Lbi + e, (size_t)2 * nz, P.box_k0, P.box_k1);
      if (T < Pn) shift_column(Lci + T, Pn, P.rowk0[T], P.rowk1[T]);
    }
    wsync();
  }

  // plant step of the device MPC loop: x0 <- A x_1 + B u_1 + f + noise (time-invariant dynamics only)
  __device__ __forceinline__ void plant_step(int step) {
    phase_begin();
    const double* Xs = Xp(cur);
    const double* Us = Up(cur);
    if (T < n) zb[T] = Xs[T];
    if (T < m) zb[np + T] = Us[T];
    wsync();
    double xn = 0.0;
    if (T < n) xn = next_state(0);
    double nrm = 1.0;
    if (P.noise_mode == 0) {
      nrm = wave_max(T < n ? fabs(xn) : 0.0);
    } else if (P.noise_mode == 1) {
      const int g = T < n ? P.noise_grp[T] : -1;
      const double n0 = sqrt(wave_sum(g == 0 ? xn * xn : 0.0)), n1 = sqrt(wave_sum(g == 1 ? xn * xn : 0.0));
      nrm = g == 0 ? n0 : n1;
    }
    if (T < n) {
      const double nzv = P.noise ? P.noise[((size_t)step * P.B + inst) * n + T] : 0.0;
      x0i[T] = xn + nzv * nrm * P.noise_w[T];
    }
    wsync();
  }

  // solve!(::ALTROSolver) (oracle orc_solve)
  __device__ __forceinline__ void solve_one() {
    const altro_opts& o = P.o;
    const double mu0 = (o.penalty_initial != o.penalty_initial) ? 1.0 : o.penalty_initial;
    const double phi = (o.penalty_scaling != o.penalty_scaling) ? 10.0 : o.penalty_scaling;
    const bool has_con = (P.box_k1 >= P.box_k0) || Pn > 0;
    if (o.reset_duals) {
      for (int e = T; e < N * 2 * nz; e += 64) Lbi[e] = 0.0;
      for (int e = T; e < N * Pn; e += 64) Lci[e] = 0.0;
    }
    if (o.reset_penalties) mu = mu0;
    status = ALTRO_UNSOLVED;
    iters = 0;
    iters_outer = 0;
    block_sync();  // the zeroed duals are read by other lanes
    double J = 0.0, cmax = 0.0;
    // one call site for ilqr(): the whole iLQR (rollouts, backward pass) is inlined into it
    const int nouter = has_con ? o.iterations_outer : 1;
    for (int jo = 0; jo < nouter; ++jo) {
      const bool last = jo == nouter - 1;
      J = ilqr(last ? o.cost_tolerance : o.cost_tolerance_intermediate,
               last ? o.gradient_tolerance : o.gradient_tolerance_intermediate, cmax);
      if (!has_con) {
        cmax = 0.0;
        if (status == ALTRO_UNSOLVED) status = ALTRO_SOLVE_SUCCEEDED;
        break;
      }
      iters_outer++;
      if (status > ALTRO_SOLVE_SUCCEEDED) break;
      if (cmax < o.constraint_tolerance || (o.kickout_max_penalty && mu >= o.penalty_max)) break;
      if (last) { status = ALTRO_MAX_ITERATIONS_OUTER; break; }
      WSTAMP(const long long tdu = wstamp();)
      do_dual_update();
      WSTAMP(t_du += wstamp() - tdu;)
      mu = fmin(fmax(phi * mu, 0.0), o.penalty_max);
    }
    if (has_con && status <= ALTRO_SOLVE_SUCCEEDED && cmax < o.constraint_tolerance) status = ALTRO_SOLVE_SUCCEEDED;
    if (T == 0) {
      P.cost[inst] = J;
      P.cmax[inst] = cmax;
      P.status[inst] = status;
      P.iters[inst] = iters;
      P.iters_outer[inst] = iters_outer;
    }
  }

  __device__ __forceinline__ void run(int mpc, int first_step, int nsteps) {
    cur = P.cur[inst];
    mu = P.mu[inst];
    kref = P.kref;
    nbw = nro = ntr = 0;
    {  // the gain-reuse state of the previous launch (one launch of K steps and K launches of one step decide alike)
      const unsigned* st = P.bwst + (size_t)inst * 136;
      {
        const unsigned roles = st[132];
        const int b_ = (int)(roles & 3u), q_ = (int)((roles >> 2) & 3u), a_ = (int)((roles >> 4) & 3u);
        if (((1 << b_) | (1 << q_) | (1 << a_)) == 7) { bwp = b_; qp = q_; ap = a_; }  // (zeroed state: the default roles)
      }
      bw_ok = P.reuse_ok != 0 && st[128] != 0u;
      bw_plain = st[129] != 0u;
      bw_mu = __hiloint2double((int)st[131], (int)st[130]);
    }
    long long nsolve = 0, nit = 0, nok = 0;
    if (!P.ltv) load_dyn(0);                                   // time-invariant dynamics stay resident in LDS
    if (Pn > 0 && P.con_static) build_static_Ac();           // and so does a time-invariant constraint table
    wsync();
    const int steps = mpc ? nsteps : 1;
    WSTAMP(const long long trun = wstamp();)
    for (int s = 0; s < steps; ++s) {
      if (mpc) {
        WSTAMP(const long long tsh = wstamp();)
        do_plant_step(first_step + s);
        kref = first_step + s + 1;  // update_trajectory!(obj, Z_track, k_mpc)
        if (mpc == 2) break;        // altro_mpc_prepare_async: new x0 only, no shift, no solve
        if (P.mpc_shift) do_shift();
        WSTAMP(t_sh += wstamp() - tsh;)
      }
      solve_one();
      nsolve++;
      nit += iters;
      nok += status == ALTRO_SOLVE_SUCCEEDED ? 1 : 0;
    }
    if (T == 0) {
      P.cur[inst] = cur;
      P.mu[inst] = mu;
      P.n_backward[inst] += nbw;
      P.n_rollout[inst] += nro;
#ifdef ALTRO_WIDE_STAMPS
      P.n_trials[inst] = t_gemm;  // diagnostic build: the three counters carry cycle counts
      t_run = wstamp() - trun;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 8] = (double)t_run;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 9] = (double)t_du;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 10] = (double)t_sh;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 11] = (double)t_td;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 12] = (double)t_a;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 13] = (double)t_b;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 14] = (double)t_c;
      P.Jtrace[(size_t)inst * ALTRO_TRACE_LEN + 15] = (double)t_d;
      P.n_backward[inst] = t_bw;
      P.n_rollout[inst] = t_ro;
#else
      P.n_trials[inst] += ntr;
#endif
      P.n_solves[inst] += nsolve;
      P.n_iters[inst] += nit;
      P.n_ok[inst] += nok;
      P.n_gconf[inst] += ngc;  // iterations booked as converged without a backward pass and a rollout
      P.n_gs[inst] += ngs;     // iterations that took their gains from memory (adjoint_lds(full) instead of a backward pass)
    }
    {
      unsigned* st = P.bwst + (size_t)inst * 136;
      if (T == 0) {
        st[132] = (unsigned)(bwp | (qp << 2) | (ap << 4));
        st[128] = bw_ok ? 1u : 0u;
        st[129] = bw_plain ? 1u : 0u;
        st[130] = (unsigned)__double2loint(bw_mu);
        st[131] = (unsigned)__double2hiint(bw_mu);
      }
    }
    coop_quit();  // releases the helper waves of a cooperative block: the ONLY exit of run(), reached on every path
  }
};

__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// One phase of the solver as a function of its own (see PhIn above).  The arguments arrive in VGPRs: what is wave-uniform is
// made scalar again first thing, so that addresses and loop bounds derived from it stay in the scalar unit.
template <int MC, bool SM, int OP, int NPC, int PRC>
__device__ __attribute__((noinline)) PhOut wide_phase(unsigned long long kp, PhIn in) {
  extern __shared__ double lds[];
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kp), hi = __builtin_amdgcn_readfirstlane((unsigned)(kp >> 32));
  typedef const __attribute__((address_space(4))) Params* KP;   // the kernel-argument segment: constant memory, scalar loads
  const Params& P = *(const Params*)(KP)(((unsigned long long)hi << 32) | (unsigned long long)lo);
  Solver<MC, SM, NPC, PRC> s(P, lds, Resume{});
  s.cur = __builtin_amdgcn_readfirstlane(in.cur);
  s.kref = __builtin_amdgcn_readfirstlane(in.kref);
  s.mu = uniform_f64(in.mu);
  s.rho = uniform_f64(in.rho);
  {
    const int roles = __builtin_amdgcn_readfirstlane((int)in.h);
    s.bwp = roles & 3; s.qp = (roles >> 2) & 3; s.ap = (roles >> 4) & 3;
  }
  s.dtiny = (__builtin_amdgcn_readfirstlane(in.flags) & 1) != 0;
  PhOut o;
  o.a = o.b = 0.0;
  o.h = 0ull;
  o.flags = 0;
  if constexpr (OP == PH_ROLL_OPEN || OP == PH_ROLL) {
    const typename Solver<MC, SM, NPC, PRC>::RollOut r = s.rollout(OP == PH_ROLL_OPEN, OP == PH_ROLL_OPEN ? 0.0 : uniform_f64(in.a));
    o.a = r.J; o.b = r.cmax;
    o.flags = (r.limit ? 1 : 0) | (r.unchanged ? 2 : 0) | (r.tiny ? 4 : 0);
  } else if constexpr (OP == PH_BACKWARD) {
    double d1 = 0.0, d2 = 0.0;
    const bool fail = s.backward(d1, d2);
    o.a = d1; o.b = d2;
    o.flags = (fail ? 1 : 0) | (s.dtiny ? 2 : 0);
  } else if constexpr (OP == PH_ADJ_FULL || OP == PH_ADJ_CONF) {
    double d1 = 0.0, d2 = 0.0;
    const bool ok = s.adjoint_lds(OP == PH_ADJ_FULL, d1, d2);
    o.a = d1; o.b = d2;
    o.flags = ok ? 1 : 0;
  } else if constexpr (OP == PH_GRAD_ADJ_ROW) {
    o.flags = s.grad_adjoint_row() ? 1 : 0;
  } else if constexpr (OP == PH_GRAD_ADJ_ROW_FULL) {
    double d1 = 0.0, d2 = 0.0;
    o.flags = s.grad_adjoint_row_full(d1, d2) ? 1 : 0;
    o.a = d1; o.b = d2;
  } else if constexpr (OP == PH_DUAL) {
    s.dual_update();
  } else if constexpr (OP == PH_SHIFT) {
    s.shift(true, true);
  } else if constexpr (OP == PH_PLANT) {
    s.plant_step(__builtin_amdgcn_readfirstlane(in.i0));
  }
#ifdef ALTRO_WIDE_STAMPS
  o.t[0] = s.t_gemm; o.t[1] = s.t_a; o.t[2] = s.t_b; o.t[3] = s.t_c; o.t[4] = s.t_d;
#endif
  return o;
}

// waves per SIMD the register allocator is held to, per control-size class
// n, m <= 16 with m <= 8: two waves per SIMD (256 registers each).  With the phases as functions of their own the spills that
// made this slower in round 3 stay out of the knot loops: n = 16, m = 4 at batch 8192: 97.7 -> 79.4 ms per 30 steps, (12, 6):
// 136.9 -> 122.1, (16, 8): 144.9 -> 129.9; at batch 1024 (one wave per SIMD either way) the same speed.  m = 9..16 (the
// quadruped): the triangle of factor_solve_lane<12> does not fit 256 registers, a wave is 2x slower there.
#ifndef ALTRO_WIDE_WAVES_SM
#define ALTRO_WIDE_WAVES_SM 2
#endif
#ifndef ALTRO_WIDE_WAVES_SM12
#define ALTRO_WIDE_WAVES_SM12 1   // m = 9 .. 16 (the quadruped): measured at 2 again in round 4 with L spread over the lanes: see DESIGN 3b
#endif
constexpr int wide_waves(int MC, bool SM) { return SM ? (MC <= 8 ? ALTRO_WIDE_WAVES_SM : ALTRO_WIDE_WAVES_SM12) : (MC == 4 || MC == 8) ? ALTRO_WIDE_WAVES_SMALL : 1; }

// threads per block: the n, m <= 16 instantiations are always one wave; the others may be launched as a cooperative
// block of four (wide_block_threads)
template <int MC, bool SM, int NPC = 0, int PRC = -1>
__global__ void __launch_bounds__(SM ? 64 : 256, wide_waves(MC, SM)) wide_kernel(Params P, int mpc, int first_step, int nsteps) {
  extern __shared__ double lds[];
  if (!SM && threadIdx.x >= 64) {  // helper waves: no solver state, only products on command
    coop_helper(lds, lds_layout(P.n, P.m, P.Pn, P.compact).cmd);
    return;
  }
  Solver<MC, SM, NPC, PRC> s(P, lds);
  s.kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();   // P is the first argument: offset 0
  s.run(mpc, first_step, nsteps);
}

// One wave per instance unless the LDS carve-up leaves room for a single instance per CU anyway (n >= 48 or so):
// then three more waves on the CU's other SIMDs share the large products.
// coop_mode: -1 = by size (the rule above), 0 = never, 1 = every size with n or m > 16 (diagnostic switch ALTRO_WIDE_COOP,
// read once when the solver is created)
inline int wide_block_threads(int n, int m, size_t lds_bytes, int coop_mode = -1) {
  if (coop_mode >= 0) return (coop_mode != 0 && !(n <= 16 && m <= 16)) ? 256 : 64;
  return (!(n <= 16 && m <= 16) && lds_bytes > 80 * 1024) ? 256 : 64;
}

// the separate shift_fill call of the fine-grained ABI (defined in the library's main translation unit only)
#ifndef ALTRO_WIDE_TU
__global__ void __launch_bounds__(64) wide_shift_kernel(Params P, int primal, int dual) {
  extern __shared__ double lds[];
  Solver<0, false> s(P, lds);
  s.cur = P.cur[s.inst];
  s.shift(primal != 0, dual != 0);
}
#endif

typedef void (*wide_kernel_t)(Params, int, int, int);
inline int wide_class(int m) { return m <= 4 ? 4 : m <= 8 ? 8 : m <= 12 ? 12 : m <= 16 ? 16 : 0; }

// Every instantiation of the library: X(translation unit, MC, SM, NPC, PRC).  The library is built from several translation
// units compiled side by side (_lib.build): altro_batch.hip declares all of these `extern`, wide_inst.hip defines the ones of
// the unit it is compiled for (-DALTRO_WIDE_TU=k).
#define ALTRO_WIDE_KERNELS(X)                                                                                   \
  X(0, 4, true, 0, -1) X(0, 4, true, 0, 0) X(0, 8, true, 0, -1) X(0, 4, false, 0, -1) X(0, 0, false, 0, -1)         \
  X(1, 12, true, 0, -1) X(1, 16, true, 0, -1) X(1, 8, false, 0, -1) X(1, 8, false, 32, -1) X(1, 12, true, 0, 16)    \
  X(2, 12, false, 0, -1) X(2, 16, false, 0, -1) X(2, 12, false, 32, -1) X(2, 16, false, 32, -1)                    \
  X(3, 4, false, 32, -1) X(3, 4, false, 48, -1) X(3, 4, false, 64, -1)                                            \
  X(4, 4, false, 32, 0) X(4, 4, false, 48, 0) X(4, 4, false, 64, 0)
constexpr int kWideTUs = 5;
#if !defined(ALTRO_DEV_HEADLINE_ONLY)
#if defined(ALTRO_WIDE_TU)
#define ALTRO_WIDE_DEFINE(tu, mc, sm, npc, pl) ALTRO_WIDE_DEFINE_##tu(mc, sm, npc, pl)
#define ALTRO_WIDE_EMIT(mc, sm, npc, pl) template __global__ void wide_kernel<mc, sm, npc, pl>(Params, int, int, int);
#define ALTRO_WIDE_SKIP(mc, sm, npc, pl)
#if ALTRO_WIDE_TU == 0
#define ALTRO_WIDE_DEFINE_0 ALTRO_WIDE_EMIT
#else
#define ALTRO_WIDE_DEFINE_0 ALTRO_WIDE_SKIP
#endif
#if ALTRO_WIDE_TU == 1
#define ALTRO_WIDE_DEFINE_1 ALTRO_WIDE_EMIT
#else
#define ALTRO_WIDE_DEFINE_1 ALTRO_WIDE_SKIP
#endif
#if ALTRO_WIDE_TU == 2
#define ALTRO_WIDE_DEFINE_2 ALTRO_WIDE_EMIT
#else
#define ALTRO_WIDE_DEFINE_2 ALTRO_WIDE_SKIP
#endif
#if ALTRO_WIDE_TU == 3
#define ALTRO_WIDE_DEFINE_3 ALTRO_WIDE_EMIT
#else
#define ALTRO_WIDE_DEFINE_3 ALTRO_WIDE_SKIP
#endif
#if ALTRO_WIDE_TU == 4
#define ALTRO_WIDE_DEFINE_4 ALTRO_WIDE_EMIT
#else
#define ALTRO_WIDE_DEFINE_4 ALTRO_WIDE_SKIP
#endif
ALTRO_WIDE_KERNELS(ALTRO_WIDE_DEFINE)
#elif defined(ALTRO_WIDE_EXTERN)
#define ALTRO_WIDE_DECLARE(tu, mc, sm, npc, pl) extern template __global__ void wide_kernel<mc, sm, npc, pl>(Params, int, int, int);
ALTRO_WIDE_KERNELS(ALTRO_WIDE_DECLARE)
#endif
#endif

// nrows: the number of generic constraint rows of the problem (0: box constraints only)
#ifndef ALTRO_WIDE_TU   // (naming a kernel instantiates it: the dispatch exists in the main translation unit only)
inline wide_kernel_t wide_kernel_for(int n, int m, int nrows) {
#ifdef ALTRO_DEV_HEADLINE_ONLY  // development builds (tools/build_stamps.sh -DALTRO_DEV_HEADLINE_ONLY): one small instantiation
#ifndef ALTRO_DEV_WIDE_KERNEL
#define ALTRO_DEV_WIDE_KERNEL wide_kernel<4, true>
#endif
  (void)nrows;
  return ALTRO_DEV_WIDE_KERNEL;  // e.g. '-DALTRO_DEV_WIDE_KERNEL=wide_kernel<4,false>' (30 s instead of 4 min)
#else
  const bool sm = n <= 16 && m <= 16, plain = nrows == 0;
  const int np = (n + 15) & ~15;
  switch (wide_class(m)) {
    case 4:   // the m <= 4 sweeps of the reference (state dimension 2 .. 64): padded n compile-time, and "no rows" where it is so
      if (sm) return plain ? wide_kernel<4, true, 0, 0> : wide_kernel<4, true>;
      if (np == 32) return plain ? wide_kernel<4, false, 32, 0> : wide_kernel<4, false, 32>;
      if (np == 48) return plain ? wide_kernel<4, false, 48, 0> : wide_kernel<4, false, 48>;
      if (np == 64) return plain ? wide_kernel<4, false, 64, 0> : wide_kernel<4, false, 64>;
      return wide_kernel<4, false>;
    // (m = 5 .. 16: the reference's control-dimension sweep runs at n = 30 -- padded 32; the quadruped has sixteen rows)
    case 8: return sm ? wide_kernel<8, true> : np == 32 ? wide_kernel<8, false, 32> : wide_kernel<8, false>;
    case 12: return sm ? (nrows == 16 ? wide_kernel<12, true, 0, 16> : wide_kernel<12, true>) : np == 32 ? wide_kernel<12, false, 32> : wide_kernel<12, false>;
    case 16: return sm ? wide_kernel<16, true> : np == 32 ? wide_kernel<16, false, 32> : wide_kernel<16, false>;
    default: return wide_kernel<0, false>;
  }
#endif
}
#endif

}  // namespace altro_wide
