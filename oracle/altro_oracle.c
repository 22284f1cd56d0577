/*
 * altro_oracle.c -- CPU restatement (FP64, single instance, single thread) of the
 * AL-iLQR solve performed by Altro.jl for the reference's benchmark problems.
 * TEST INFRASTRUCTURE ONLY -- see altro_oracle.h for the usage rule and parity status.
 *
 * The solver source is not under /root/reference (un-vendored Julia packages); each
 * function below cites the reference call site it serves and SURVEY.md Appendix A item it
 * restates.  "[PKG]" marks behaviour restated from the published ALTRO / ALTRO-C
 * algorithm and the public Altro.jl v0.2 sources rather than from a file in the reference.
 */
#include "altro_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXCON 16

typedef struct {
  int kind, sense, k0, k1, p, per_knot, nk;
  double* A;    /* [nkA][p][nz] row-major (LINEAR/SOC) */
  double* b;    /* [nkA][p] */
  double* zmin; /* BOX [nz] */
  double* zmax;
  double* lam;  /* [nk][p] */
  double* mu;   /* [nk][p] */
  double* c;    /* [nk][p] last evaluated values */
  double mu0, phi; /* per-constraint defaults (TO.ConstraintParams: 1.0, 10.0) [PKG] */
} con_t;

struct orc_solver {
  int n, m, N, nz;
  double dt;
  int ltv;
  double *A, *B, *f;           /* [(N-1) or 1] blocks, col-major */
  double *Qd, *Rd, *Qfd;
  double *Xref, *Uref;
  double *x0;
  double *X, *U, *Xb, *Ub;     /* current and candidate trajectories */
  double *K, *d;               /* gains: K[k] m*n col-major, d[k] m */
  /* expansions per knot */
  double *lxx, *luu, *lux, *lx, *lu; /* [N][n*n], [N][m*m], [N][m*n], [N][n], [N][m] */
  int ncon;
  con_t con[MAXCON];
  orc_opts opts;
  orc_stats stats;
  double rho, drho;
  int dJ_zero_counter;
  /* scratch */
  double *S, *s, *Qxx, *Quu, *Qux, *Qx, *Qu, *tmp_nn, *tmp_nm, *tmp_mm, *tmp_mn, *Quu_reg;
  /* diagnostic (orc_debug_pass_trace): highest knot whose second-order cost expansion differs from the previous pass's */
  int *dbg_buf, dbg_cap, dbg_n;
  double *dbg_prev;
  /* diagnostic (orc_debug_ls_trace): per rejected line-search trial, the fraction of the knots after which the partial
   * cost plus a lower bound of the rest already exceeds J_prev (suffix bound, crude bound) and J - J_prev */
  double *dbg_ls;
  int dbg_ls_cap, dbg_ls_n;
};

/* ---------------------------------------------------------------- options */
void orc_default_opts(orc_opts* o) {
  /* Altro.SolverOptions defaults, SURVEY.md Appendix A.1 [PKG] */
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
  o->kickout_max_penalty = 0;
  o->projected_newton = 0;
  o->projected_newton_tolerance = 1e-3;
  o->active_set_tolerance_pn = 1e-3;
  o->rho_chol = 1e-2;
  o->rho_primal = 1e-8;
  o->r_threshold = 1.1;
}

/* ---------------------------------------------------------------- lifecycle */
static double* dalloc(size_t k) { return (double*)calloc(k ? k : 1, sizeof(double)); }

/* ALTROSolver(prob, opts): reference random_linear_problem.jl:87 (P1) */
orc_solver* orc_create(int n, int m, int N, double dt) {
  orc_solver* s = (orc_solver*)calloc(1, sizeof(orc_solver));
  s->n = n; s->m = m; s->N = N; s->nz = n + m; s->dt = dt;
  s->A = dalloc((size_t)n * n); s->B = dalloc((size_t)n * m); s->f = dalloc(n);
  s->Qd = dalloc(n); s->Rd = dalloc(m); s->Qfd = dalloc(n);
  s->Xref = dalloc((size_t)N * n); s->Uref = dalloc((size_t)(N - 1) * m);
  s->x0 = dalloc(n);
  s->X = dalloc((size_t)N * n); s->U = dalloc((size_t)(N - 1) * m);
  s->Xb = dalloc((size_t)N * n); s->Ub = dalloc((size_t)(N - 1) * m);
  s->K = dalloc((size_t)(N - 1) * m * n); s->d = dalloc((size_t)(N - 1) * m);
  s->lxx = dalloc((size_t)N * n * n); s->luu = dalloc((size_t)N * m * m);
  s->lux = dalloc((size_t)N * m * n); s->lx = dalloc((size_t)N * n); s->lu = dalloc((size_t)N * m);
  s->S = dalloc((size_t)n * n); s->s = dalloc(n);
  s->Qxx = dalloc((size_t)n * n); s->Quu = dalloc((size_t)m * m); s->Qux = dalloc((size_t)m * n);
  s->Qx = dalloc(n); s->Qu = dalloc(m);
  s->tmp_nn = dalloc((size_t)n * n); s->tmp_nm = dalloc((size_t)n * m);
  s->tmp_mm = dalloc((size_t)m * m); s->tmp_mn = dalloc((size_t)m * n); s->Quu_reg = dalloc((size_t)m * m);
  orc_default_opts(&s->opts);
  return s;
}

void orc_destroy(orc_solver* s) {
  if (!s) return;
  free(s->A); free(s->B); free(s->f); free(s->Qd); free(s->Rd); free(s->Qfd);
  free(s->Xref); free(s->Uref); free(s->x0); free(s->X); free(s->U); free(s->Xb); free(s->Ub);
  free(s->K); free(s->d); free(s->lxx); free(s->luu); free(s->lux); free(s->lx); free(s->lu);
  free(s->S); free(s->s); free(s->Qxx); free(s->Quu); free(s->Qux); free(s->Qx); free(s->Qu);
  free(s->tmp_nn); free(s->tmp_nm); free(s->tmp_mm); free(s->tmp_mn); free(s->Quu_reg); free(s->dbg_prev);
  for (int i = 0; i < s->ncon; ++i) {
    con_t* c = &s->con[i];
    free(c->A); free(c->b); free(c->zmin); free(c->zmax); free(c->lam); free(c->mu); free(c->c);
  }
  free(s);
}

/* RD.LinearModel(A,B[,d]; dt[,times]) : random_linear_problem.jl:8, ALTROParams.jl:61 (P4) */
void orc_set_dynamics(orc_solver* s, const double* A, const double* B, const double* f, int per_knot) {
  int n = s->n, m = s->m;
  size_t nb = per_knot ? (size_t)(s->N - 1) : 1;
  free(s->A); free(s->B); free(s->f);
  s->A = dalloc(nb * n * n); s->B = dalloc(nb * n * m); s->f = dalloc(nb * n);
  memcpy(s->A, A, nb * n * n * sizeof(double));
  memcpy(s->B, B, nb * n * m * sizeof(double));
  if (f) memcpy(s->f, f, nb * n * sizeof(double));
  s->ltv = per_knot ? 1 : 0;
}

/* TO.TrackingObjective(Q,R,Z;Qf) / LQRObjective : mpc.jl:26-29 */
void orc_set_cost(orc_solver* s, const double* Qd, const double* Rd, const double* Qfd) {
  memcpy(s->Qd, Qd, s->n * sizeof(double));
  memcpy(s->Rd, Rd, s->m * sizeof(double));
  memcpy(s->Qfd, Qfd, s->n * sizeof(double));
}

/* TO.update_trajectory!(obj, Z_track, k) : random_linear_problem.jl:133 (P12) */
void orc_set_reference(orc_solver* s, const double* Xref, const double* Uref) {
  memcpy(s->Xref, Xref, (size_t)s->N * s->n * sizeof(double));
  memcpy(s->Uref, Uref, (size_t)(s->N - 1) * s->m * sizeof(double));
}

/* TO.set_initial_state! : random_linear_problem.jl:130 */
void orc_set_initial_state(orc_solver* s, const double* x0) {
  memcpy(s->x0, x0, s->n * sizeof(double));
}

/* initial_controls! : altro_solver.jl:71 */
void orc_set_controls(orc_solver* s, const double* U) {
  memcpy(s->U, U, (size_t)(s->N - 1) * s->m * sizeof(double));
}

void orc_set_opts(orc_solver* s, const orc_opts* o) { s->opts = *o; }

/* diagnostic: every backward pass from now on appends to buf (up to cap entries) the highest knot whose second-order
 * cost expansion (lxx, luu, lux: the active set and the penalties) differs from the previous pass's, -1 if none does.
 * Returns the number of passes recorded so far and restarts the count. */
/* diagnostic: rejected line-search trials from now on append 4 doubles each to buf (see dbg_rejected_trial); returns the
 * number recorded so far and restarts the count */
int orc_debug_ls_trace(orc_solver* s, double* buf, int cap) {
  int got = s->dbg_ls_n;
  s->dbg_ls = buf;
  s->dbg_ls_cap = cap;
  s->dbg_ls_n = 0;
  return got;
}
int orc_debug_pass_trace(orc_solver* s, int* buf, int cap) {
  int got = s->dbg_n;
  s->dbg_buf = buf;
  s->dbg_cap = cap;
  s->dbg_n = 0;
  return got;
}

/* add_constraint!(cons, con, inds) : random_linear_problem.jl:24 (C1-C8) */
int orc_add_constraint(orc_solver* s, int kind, int sense, int k_first, int k_last, int p,
                       const double* A, const double* b, const double* zmin, const double* zmax,
                       int per_knot) {
  if (s->ncon >= MAXCON) return -1;
  con_t* c = &s->con[s->ncon];
  memset(c, 0, sizeof(*c));
  int nz = s->nz;
  c->kind = kind; c->sense = sense; c->k0 = k_first; c->k1 = k_last;
  c->nk = k_last - k_first + 1;
  c->per_knot = per_knot;
  if (kind == ORC_BOX) {
    p = 2 * nz;
    c->zmin = dalloc(nz); c->zmax = dalloc(nz);
    memcpy(c->zmin, zmin, nz * sizeof(double));
    memcpy(c->zmax, zmax, nz * sizeof(double));
    c->sense = ORC_INEQ;
  } else {
    size_t nb = per_knot ? (size_t)c->nk : 1;
    c->A = dalloc(nb * p * nz); c->b = dalloc(nb * p);
    memcpy(c->A, A, nb * p * nz * sizeof(double));
    memcpy(c->b, b, nb * p * sizeof(double));
  }
  c->p = p;
  c->lam = dalloc((size_t)c->nk * p);
  c->mu = dalloc((size_t)c->nk * p);
  c->c = dalloc((size_t)c->nk * p);
  c->mu0 = 1.0; c->phi = 10.0;
  for (size_t i = 0; i < (size_t)c->nk * p; ++i) c->mu[i] = c->mu0;
  return s->ncon++;
}

/* in-place mutation of per-knot constraint data: grasp_mpc_helpers.jl:46-55 */
void orc_update_constraint_data(orc_solver* s, int ci, const double* A, const double* b) {
  con_t* c = &s->con[ci];
  size_t nb = c->per_knot ? (size_t)c->nk : 1;
  if (A) memcpy(c->A, A, nb * c->p * s->nz * sizeof(double));
  if (b) memcpy(c->b, b, nb * c->p * sizeof(double));
}

/* ---------------------------------------------------------------- small helpers */
static const double* Ak(const orc_solver* s, int k) { return s->A + (s->ltv ? (size_t)k * s->n * s->n : 0); }
static const double* Bk(const orc_solver* s, int k) { return s->B + (s->ltv ? (size_t)k * s->n * s->m : 0); }
static const double* fk(const orc_solver* s, int k) { return s->f + (s->ltv ? (size_t)k * s->n : 0); }

/* discrete_dynamics(PassThrough, LinearModel, z): x+ = A x + B u + f  (P4) */
static void dynamics(const orc_solver* s, int k, const double* x, const double* u, double* xn) {
  int n = s->n, m = s->m;
  const double *A = Ak(s, k), *B = Bk(s, k), *f = fk(s, k);
  for (int i = 0; i < n; ++i) {
    double acc = f[i];
    for (int j = 0; j < n; ++j) acc += A[i + n * j] * x[j];
    for (int j = 0; j < m; ++j) acc += B[i + n * j] * u[j];
    xn[i] = acc;
  }
}

void orc_plant_step(const orc_solver* s, double* xnext) { dynamics(s, 0, s->X, s->U, xnext); }

/* Euclidean projection onto the second-order cone {(v,t): ||v|| <= t}; SURVEY A.2.
 * Returns the branch taken: 0 inside, 1 polar (-> 0), 2 boundary. */
static int soc_project(const double* x, int p, double* out) {
  double nv = 0;
  for (int i = 0; i < p - 1; ++i) nv += x[i] * x[i];
  nv = sqrt(nv);
  double t = x[p - 1];
  if (nv <= t) { memcpy(out, x, p * sizeof(double)); return 0; }
  if (nv <= -t) { memset(out, 0, p * sizeof(double)); return 1; }
  double c = 0.5 * (1.0 + t / nv);
  for (int i = 0; i < p - 1; ++i) out[i] = c * x[i];
  out[p - 1] = c * nv;
  return 2;
}

/* constraint value c = A z + b at knot k (z = [x;u], u ignored at the terminal knot) */
static void con_eval(const orc_solver* s, const con_t* c, int k, const double* x, const double* u, double* out) {
  int n = s->n, m = s->m, nz = s->nz;
  int terminal = (k == s->N - 1);
  if (c->kind == ORC_BOX) {
    for (int j = 0; j < nz; ++j) {
      double z = j < n ? x[j] : (terminal ? 0.0 : u[j - n]);
      int live = j < n || !terminal;
      out[j] = (live && isfinite(c->zmax[j])) ? z - c->zmax[j] : -INFINITY;
      out[nz + j] = (live && isfinite(c->zmin[j])) ? c->zmin[j] - z : -INFINITY;
    }
    return;
  }
  size_t blk = c->per_knot ? (size_t)(k - c->k0) : 0;
  const double* A = c->A + blk * c->p * nz;
  const double* b = c->b + blk * c->p;
  for (int r = 0; r < c->p; ++r) {
    double acc = b[r];
    for (int j = 0; j < n; ++j) acc += A[r * nz + j] * x[j];
    if (!terminal) for (int j = 0; j < m; ++j) acc += A[r * nz + n + j] * u[j];
    out[r] = acc;
  }
}

/* AL penalty term for one constraint at one knot, SURVEY A.2 (P5):
 *   eq/ineq: lam'c + 1/2 c' I_mu c, active-set rule a = (c >= 0) | (lam > 0) for inequalities
 *   SOC:     (1/2mu)(||Proj(lam - mu c)||^2 - ||lam||^2)                           [PKG] */
static double con_cost(const con_t* c, const double* cv, const double* lam, const double* mu) {
  double J = 0;
  int p = c->p;
  if (c->kind == ORC_SOC) {
    double lb[64], lp[64];
    double m0 = mu[0];
    for (int r = 0; r < p; ++r) lb[r] = lam[r] - m0 * cv[r];
    soc_project(lb, p, lp);
    double a = 0, bq = 0;
    for (int r = 0; r < p; ++r) { a += lp[r] * lp[r]; bq += lam[r] * lam[r]; }
    return (a - bq) / (2.0 * m0);
  }
  for (int r = 0; r < p; ++r) {
    if (!(cv[r] > -INFINITY)) continue;
    int active = (c->sense == ORC_EQ) || (cv[r] >= 0.0) || (lam[r] > 0.0);
    J += lam[r] * cv[r] + (active ? 0.5 * mu[r] * cv[r] * cv[r] : 0.0);
  }
  return J;
}

/* violation of one constraint value (P8): eq |c|, ineq max(0,c), SOC ||Proj(c)-c||_inf [PKG] */
static double con_violation(const con_t* c, const double* cv) {
  double v = 0;
  int p = c->p;
  if (c->kind == ORC_SOC) {
    double pr[64];
    soc_project(cv, p, pr);
    for (int r = 0; r < p; ++r) { double e = fabs(pr[r] - cv[r]); if (e > v) v = e; }
    return v;
  }
  for (int r = 0; r < p; ++r) {
    if (!(cv[r] > -INFINITY)) continue;
    double e = (c->sense == ORC_EQ) ? fabs(cv[r]) : (cv[r] > 0 ? cv[r] : 0.0);
    if (e > v) v = e;
  }
  return v;
}

/* cost!(obj, Z): J = sum_k dt*l_k + l_N + AL terms; also refreshes stored constraint values
 * and returns c_max through *cmax.  (P5, P8) */
static double total_cost(orc_solver* s, const double* X, const double* U, double* cmax) {
  int n = s->n, m = s->m, N = s->N;
  double J = 0;
  for (int k = 0; k < N; ++k) {
    const double* x = X + (size_t)k * n;
    double l = 0;
    if (k < N - 1) {
      const double* u = U + (size_t)k * m;
      for (int i = 0; i < n; ++i) { double e = x[i] - s->Xref[(size_t)k * n + i]; l += 0.5 * s->Qd[i] * e * e; }
      for (int i = 0; i < m; ++i) { double e = u[i] - s->Uref[(size_t)k * m + i]; l += 0.5 * s->Rd[i] * e * e; }
      J += l * s->dt;
    } else {
      for (int i = 0; i < n; ++i) { double e = x[i] - s->Xref[(size_t)k * n + i]; l += 0.5 * s->Qfd[i] * e * e; }
      J += l;
    }
  }
  double vmax = 0;
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    for (int k = c->k0; k <= c->k1; ++k) {
      size_t off = (size_t)(k - c->k0) * c->p;
      con_eval(s, c, k, X + (size_t)k * n, U + (size_t)(k < N - 1 ? k : 0) * m, c->c + off);
      J += con_cost(c, c->c + off, c->lam + off, c->mu + off);
      double v = con_violation(c, c->c + off);
      if (v > vmax) vmax = v;
    }
  }
  if (cmax) *cmax = vmax;
  return J;
}

/* cost_expansion!: quadratic expansion of the AL objective at (X,U)  (P5) */
static void cost_expansion(orc_solver* s) {
  int n = s->n, m = s->m, N = s->N, nz = s->nz;
  memset(s->lxx, 0, (size_t)N * n * n * sizeof(double));
  memset(s->luu, 0, (size_t)N * m * m * sizeof(double));
  memset(s->lux, 0, (size_t)N * m * n * sizeof(double));
  for (int k = 0; k < N; ++k) {
    double w = (k < N - 1) ? s->dt : 1.0;
    const double* Qd = (k < N - 1) ? s->Qd : s->Qfd;
    for (int i = 0; i < n; ++i) {
      s->lxx[(size_t)k * n * n + i + n * i] = w * Qd[i];
      s->lx[(size_t)k * n + i] = w * Qd[i] * (s->X[(size_t)k * n + i] - s->Xref[(size_t)k * n + i]);
    }
    for (int i = 0; i < m; ++i) {
      if (k < N - 1) {
        s->luu[(size_t)k * m * m + i + m * i] = w * s->Rd[i];
        s->lu[(size_t)k * m + i] = w * s->Rd[i] * (s->U[(size_t)k * m + i] - s->Uref[(size_t)k * m + i]);
      } else {
        s->lu[(size_t)k * m + i] = 0;
      }
    }
  }
  double cv[64], g[64], H[64 * 64];
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    int p = c->p;
    for (int k = c->k0; k <= c->k1; ++k) {
      size_t off = (size_t)(k - c->k0) * p;
      const double* x = s->X + (size_t)k * n;
      const double* u = s->U + (size_t)(k < N - 1 ? k : 0) * m;
      const double *lam = c->lam + off, *mu = c->mu + off;
      int terminal = (k == N - 1);
      double* lxx = s->lxx + (size_t)k * n * n;
      double* luu = s->luu + (size_t)k * m * m;
      double* lux = s->lux + (size_t)k * m * n;
      double* lx = s->lx + (size_t)k * n;
      double* lu = s->lu + (size_t)k * m;
      if (c->kind == ORC_BOX) {
        double cb[2 * 64];
        con_eval(s, c, k, x, u, cb);
        for (int j = 0; j < nz; ++j) {
          if (j >= n && terminal) continue;
          double gj = 0, hj = 0;
          double chi = cb[j], clo = cb[nz + j];
          if (chi > -INFINITY) {
            int a = (chi >= 0.0) || (lam[j] > 0.0);
            gj += lam[j] + (a ? mu[j] * chi : 0.0);
            hj += a ? mu[j] : 0.0;
          }
          if (clo > -INFINITY) {
            int a = (clo >= 0.0) || (lam[nz + j] > 0.0);
            gj -= lam[nz + j] + (a ? mu[nz + j] * clo : 0.0);
            hj += a ? mu[nz + j] : 0.0;
          }
          if (j < n) { lx[j] += gj; lxx[j + n * j] += hj; }
          else { lu[j - n] += gj; luu[(j - n) + m * (j - n)] += hj; }
        }
        continue;
      }
      size_t blk = c->per_knot ? (size_t)(k - c->k0) : 0;
      const double* A = c->A + blk * p * nz;
      con_eval(s, c, k, x, u, cv);
      /* g (p): d phi / d c ; H (p x p): Gauss-Newton weight so that
       * grad_z = A' g, hess_z = A' H A */
      memset(H, 0, sizeof(double) * p * p);
      if (c->kind == ORC_LINEAR) {
        for (int r = 0; r < p; ++r) {
          int a = (c->sense == ORC_EQ) || (cv[r] >= 0.0) || (lam[r] > 0.0);
          g[r] = lam[r] + (a ? mu[r] * cv[r] : 0.0);
          H[r * p + r] = a ? mu[r] : 0.0;
        }
      } else { /* SOC: phi = (1/2mu)(||Pi(lb)||^2 - ||lam||^2), lb = lam - mu c          [PKG]
                 d phi/dc = -JPi' Pi(lb);  GN: mu JPi'JPi (+ mu * d/dlb[JPi' y]|_{y=Pi(lb)} term) */
        double lb[64], lp[64], Jp[64 * 64];
        double m0 = mu[0];
        for (int r = 0; r < p; ++r) lb[r] = lam[r] - m0 * cv[r];
        int br = soc_project(lb, p, lp);
        memset(Jp, 0, sizeof(double) * p * p);
        if (br == 0) {
          for (int r = 0; r < p; ++r) Jp[r * p + r] = 1.0;
        } else if (br == 2) {
          int q = p - 1;
          double nv = 0;
          for (int i = 0; i < q; ++i) nv += lb[i] * lb[i];
          nv = sqrt(nv);
          double t = lb[q];
          double cc = 0.5 * (1.0 + t / nv);
          /* d/dv [cc v] = cc I - (t/(2 nv^3)) v v' ; d/dt [cc v] = v/(2 nv)
             d/dv [cc nv] = v'/(2nv) (1 + t/nv) - t v'/(2 nv^2) = v'/(2 nv) ; d/dt = 1/2 */
          for (int i = 0; i < q; ++i) {
            for (int j = 0; j < q; ++j)
              Jp[i * p + j] = (i == j ? cc : 0.0) - 0.5 * t * lb[i] * lb[j] / (nv * nv * nv);
            Jp[i * p + q] = 0.5 * lb[i] / nv;
            Jp[q * p + i] = 0.5 * lb[i] / nv;
          }
          Jp[q * p + q] = 0.5;
        }
        for (int r = 0; r < p; ++r) {
          double acc = 0;
          for (int i = 0; i < p; ++i) acc += Jp[i * p + r] * lp[i];
          g[r] = -acc;
        }
        for (int i = 0; i < p; ++i)
          for (int j = 0; j < p; ++j) {
            double acc = 0;
            for (int r = 0; r < p; ++r) acc += Jp[r * p + i] * Jp[r * p + j];
            H[i * p + j] = m0 * acc;
          }
        if (s->opts.soc_second_order && br == 2) {
          /* curvature of the projection contracted with y = Pi(lb):  d/dlb [JPi(lb)' y]  (y fixed)
             With y = cc*[v; nv]:  JPi' y = cc*... ; closed form below (symmetric p x p). */
          int q = p - 1;
          double nv = 0;
          for (int i = 0; i < q; ++i) nv += lb[i] * lb[i];
          nv = sqrt(nv);
          double t = lb[q];
          const double* y = lp;
          double yv_dot_v = 0;
          for (int i = 0; i < q; ++i) yv_dot_v += y[i] * lb[i];
          double yt = y[q];
          /* (JPi' y)_v = cc y_v - (t/(2nv^3)) v (v'y_v) + yt v/(2nv);  (JPi' y)_t = (v'y_v)/(2nv) + yt/2 */
          for (int i = 0; i < q; ++i) {
            for (int j = 0; j < q; ++j) {
              double dcc_dvj = -0.5 * t * lb[j] / (nv * nv * nv);
              double term = dcc_dvj * y[i]
                  - 0.5 * t * ((i == j ? yv_dot_v : 0.0) + lb[i] * y[j]) / (nv * nv * nv)
                  + 1.5 * t * lb[i] * yv_dot_v * lb[j] / (nv * nv * nv * nv * nv)
                  + yt * ((i == j ? 1.0 : 0.0) / (2 * nv) - lb[i] * lb[j] / (2 * nv * nv * nv));
              H[i * p + j] += m0 * term;
            }
            double dt_term = y[i] / (2 * nv) - lb[i] * yv_dot_v / (2 * nv * nv * nv);
            H[i * p + q] += m0 * dt_term;
            H[q * p + i] += m0 * dt_term;
          }
        }
      }
      /* accumulate A' g and A' H A into the knot expansion */
      for (int j = 0; j < nz; ++j) {
        if (j >= n && terminal) continue;
        double acc = 0;
        for (int r = 0; r < p; ++r) acc += A[r * nz + j] * g[r];
        if (j < n) lx[j] += acc; else lu[j - n] += acc;
      }
      for (int i = 0; i < nz; ++i) {
        if (i >= n && terminal) continue;
        for (int j = 0; j < nz; ++j) {
          if (j >= n && terminal) continue;
          double acc = 0;
          for (int r = 0; r < p; ++r) {
            double hr = 0;
            for (int q2 = 0; q2 < p; ++q2) hr += H[r * p + q2] * A[q2 * nz + j];
            acc += A[r * nz + i] * hr;
          }
          if (acc == 0.0) continue;
          if (i < n && j < n) lxx[i + n * j] += acc;
          else if (i >= n && j >= n) luu[(i - n) + m * (j - n)] += acc;
          else if (i >= n && j < n) lux[(i - n) + m * j] += acc;
        }
      }
    }
  }
}

/* Cholesky factor (lower) of an m x m SPD col-major matrix in place; returns 0 on success */
static int chol_lower(double* M, int m) {
  for (int j = 0; j < m; ++j) {
    double d = M[j + m * j];
    for (int k = 0; k < j; ++k) d -= M[j + m * k] * M[j + m * k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    M[j + m * j] = d;
    for (int i = j + 1; i < m; ++i) {
      double v = M[i + m * j];
      for (int k = 0; k < j; ++k) v -= M[i + m * k] * M[j + m * k];
      M[i + m * j] = v / d;
    }
  }
  return 0;
}

static void chol_solve(const double* L, int m, double* rhs) {
  for (int i = 0; i < m; ++i) {
    double v = rhs[i];
    for (int k = 0; k < i; ++k) v -= L[i + m * k] * rhs[k];
    rhs[i] = v / L[i + m * i];
  }
  for (int i = m - 1; i >= 0; --i) {
    double v = rhs[i];
    for (int k = i + 1; k < m; ++k) v -= L[k + m * i] * rhs[k];
    rhs[i] = v / L[i + m * i];
  }
}

/* regularization_update! [PKG] */
static void reg_update(orc_solver* s, int increase) {
  const orc_opts* o = &s->opts;
  if (increase) {
    s->drho = fmax(s->drho * o->bp_reg_increase_factor, o->bp_reg_increase_factor);
    s->rho = fmax(s->rho * s->drho, o->bp_reg_min);
  } else {
    s->drho = fmin(s->drho / o->bp_reg_increase_factor, 1.0 / o->bp_reg_increase_factor);
    s->rho = s->rho * s->drho * (s->rho * s->drho > o->bp_reg_min ? 1.0 : 0.0);
  }
}

/* backwardpass!  (P6).  Returns 0 ok / 1 if Quu was not PD even after
 * regularisation reached bp_reg_max. */
static int backward_pass(orc_solver* s, double dV[2]) {
  int n = s->n, m = s->m, N = s->N;
  double *S = s->S, *sv = s->s;
  int restart;
  if (s->dbg_buf) {
    size_t per = (size_t)n * n + (size_t)m * m + (size_t)m * n;
    if (!s->dbg_prev) s->dbg_prev = dalloc((size_t)N * per);
    int kc = -1;
    for (int k = N - 1; k >= 0; --k) {
      double* pv = s->dbg_prev + (size_t)k * per;
      int same = !memcmp(pv, s->lxx + (size_t)k * n * n, sizeof(double) * n * n) &&
                 !memcmp(pv + n * n, s->luu + (size_t)k * m * m, sizeof(double) * m * m) &&
                 !memcmp(pv + n * n + m * m, s->lux + (size_t)k * m * n, sizeof(double) * m * n);
      if (!same && kc < 0) kc = k;
      memcpy(pv, s->lxx + (size_t)k * n * n, sizeof(double) * n * n);
      memcpy(pv + n * n, s->luu + (size_t)k * m * m, sizeof(double) * m * m);
      memcpy(pv + n * n + m * m, s->lux + (size_t)k * m * n, sizeof(double) * m * n);
    }
    if (s->dbg_n < s->dbg_cap) s->dbg_buf[s->dbg_n] = kc;
    s->dbg_n++;
  }
  do {
    restart = 0;
    memcpy(S, s->lxx + (size_t)(N - 1) * n * n, (size_t)n * n * sizeof(double));
    memcpy(sv, s->lx + (size_t)(N - 1) * n, n * sizeof(double));
    dV[0] = dV[1] = 0;
    for (int k = N - 2; k >= 0; --k) {
      const double *A = Ak(s, k), *B = Bk(s, k);
      double *Qxx = s->Qxx, *Quu = s->Quu, *Qux = s->Qux, *Qx = s->Qx, *Qu = s->Qu;
      double *SA = s->tmp_nn, *SB = s->tmp_nm;
      /* Qx = lx + A's ; Qu = lu + B's */
      for (int i = 0; i < n; ++i) {
        double acc = s->lx[(size_t)k * n + i];
        for (int j = 0; j < n; ++j) acc += A[j + n * i] * sv[j];
        Qx[i] = acc;
      }
      for (int i = 0; i < m; ++i) {
        double acc = s->lu[(size_t)k * m + i];
        for (int j = 0; j < n; ++j) acc += B[j + n * i] * sv[j];
        Qu[i] = acc;
      }
      /* SA = S A, SB = S B */
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          double acc = 0;
          for (int l = 0; l < n; ++l) acc += S[i + n * l] * A[l + n * j];
          SA[i + n * j] = acc;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
          double acc = 0;
          for (int l = 0; l < n; ++l) acc += S[i + n * l] * B[l + n * j];
          SB[i + n * j] = acc;
        }
      /* Qxx = lxx + A'SA ; Quu = luu + B'SB ; Qux = lux + B'SA */
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          double acc = s->lxx[(size_t)k * n * n + i + n * j];
          for (int l = 0; l < n; ++l) acc += A[l + n * i] * SA[l + n * j];
          Qxx[i + n * j] = acc;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) {
          double acc = s->luu[(size_t)k * m * m + i + m * j];
          for (int l = 0; l < n; ++l) acc += B[l + n * i] * SB[l + n * j];
          Quu[i + m * j] = acc;
        }
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
          double acc = s->lux[(size_t)k * m * n + i + m * j];
          for (int l = 0; l < n; ++l) acc += B[l + n * i] * SA[l + n * j];
          Qux[i + m * j] = acc;
        }
      /* regularisation (bp_reg_type = :control): Quu_reg = Quu + rho I */
      double* L = s->Quu_reg;
      memcpy(L, Quu, (size_t)m * m * sizeof(double));
      for (int i = 0; i < m; ++i) L[i + m * i] += s->rho;
      if (chol_lower(L, m)) {
        if (s->rho >= s->opts.bp_reg_max) return 1;
        reg_update(s, 1);
        restart = 1;
        break;
      }
      /* K = -Quu_reg^{-1} Qux ; d = -Quu_reg^{-1} Qu */
      double* K = s->K + (size_t)k * m * n;
      double* d = s->d + (size_t)k * m;
      for (int j = 0; j < n; ++j) {
        double col[64];
        for (int i = 0; i < m; ++i) col[i] = Qux[i + m * j];
        chol_solve(L, m, col);
        for (int i = 0; i < m; ++i) K[i + m * j] = -col[i];
      }
      {
        double col[64];
        for (int i = 0; i < m; ++i) col[i] = Qu[i];
        chol_solve(L, m, col);
        for (int i = 0; i < m; ++i) d[i] = -col[i];
      }
      /* cost-to-go with the un-regularised Quu:
         s = Qx + K'Quu d + K'Qu + Qux'd ; S = Qxx + K'Quu K + K'Qux + Qux'K ; S = (S+S')/2 */
      double* QuuK = s->tmp_mn; /* m x n */
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
          double acc = 0;
          for (int l = 0; l < m; ++l) acc += Quu[i + m * l] * K[l + m * j];
          QuuK[i + m * j] = acc;
        }
      double Quud[64];
      for (int i = 0; i < m; ++i) {
        double acc = 0;
        for (int l = 0; l < m; ++l) acc += Quu[i + m * l] * d[l];
        Quud[i] = acc;
      }
      for (int i = 0; i < n; ++i) {
        double acc = Qx[i];
        for (int l = 0; l < m; ++l) acc += K[l + m * i] * (Quud[l] + Qu[l]) + Qux[l + m * i] * d[l];
        sv[i] = acc;
      }
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          double acc = Qxx[i + n * j];
          for (int l = 0; l < m; ++l)
            acc += K[l + m * i] * (QuuK[l + m * j] + Qux[l + m * j]) + Qux[l + m * i] * K[l + m * j];
          S[i + n * j] = acc;
        }
      for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) {
          double v = 0.5 * (S[i + n * j] + S[j + n * i]);
          S[i + n * j] = v; S[j + n * i] = v;
        }
      double t1 = 0, t2 = 0;
      for (int i = 0; i < m; ++i) { t1 += d[i] * Qu[i]; t2 += 0.5 * d[i] * Quud[i]; }
      dV[0] += t1; dV[1] += t2;
    }
  } while (restart);
  reg_update(s, 0);
  return 0;
}

/* rollout!(solver, alpha): closed-loop rollout into (Xb,Ub); returns 0 if a state/control
 * limit was hit (P7) */
static int rollout_alpha(orc_solver* s, double alpha) {
  int n = s->n, m = s->m, N = s->N;
  memcpy(s->Xb, s->x0, n * sizeof(double));
  for (int k = 0; k < N - 1; ++k) {
    const double* K = s->K + (size_t)k * m * n;
    const double* d = s->d + (size_t)k * m;
    const double* x = s->X + (size_t)k * n;
    const double* xb = s->Xb + (size_t)k * n;
    double* ub = s->Ub + (size_t)k * m;
    for (int i = 0; i < m; ++i) {
      double acc = s->U[(size_t)k * m + i] + alpha * d[i];
      for (int j = 0; j < n; ++j) acc += K[i + m * j] * (xb[j] - x[j]);
      ub[i] = acc;
    }
    dynamics(s, k, xb, ub, s->Xb + (size_t)(k + 1) * n);
    double mx = 0, mu_ = 0;
    for (int i = 0; i < n; ++i) { double a = fabs(s->Xb[(size_t)(k + 1) * n + i]); if (!(a <= mx)) mx = a; }
    for (int i = 0; i < m; ++i) { double a = fabs(ub[i]); if (!(a <= mu_)) mu_ = a; }
    if (!(mx <= s->opts.max_state_value) || !(mu_ <= s->opts.max_control_value)) return 0;
  }
  return 1;
}

/* open-loop rollout!(solver) from x0 with the current controls */
static int rollout_open(orc_solver* s) {
  int n = s->n, m = s->m, N = s->N;
  memcpy(s->X, s->x0, n * sizeof(double));
  for (int k = 0; k < N - 1; ++k) {
    dynamics(s, k, s->X + (size_t)k * n, s->U + (size_t)k * m, s->X + (size_t)(k + 1) * n);
    for (int i = 0; i < n; ++i)
      if (!(fabs(s->X[(size_t)(k + 1) * n + i]) <= s->opts.max_state_value)) return 0;
  }
  return 1;
}

/* diagnostic only: where would a sweep over the knots know that this trial is rejected (J >= J_prev)?  Each AL term is
 * bounded below by -|lam|^2 / (2 mu), the stage cost by 0. */
static void dbg_rejected_trial(orc_solver* s, const double* X, const double* U, double J, double J_prev) {
  int n = s->n, m = s->m, N = s->N;
  if (!s->dbg_ls || s->dbg_ls_n >= s->dbg_ls_cap) { s->dbg_ls_n++; return; }
  double* ck = dalloc(N); double* lb = dalloc(N);
  for (int k = 0; k < N; ++k) {
    const double* x = X + (size_t)k * n;
    double l = 0;
    if (k < N - 1) {
      const double* u = U + (size_t)k * m;
      for (int i = 0; i < n; ++i) { double e = x[i] - s->Xref[(size_t)k * n + i]; l += 0.5 * s->Qd[i] * e * e; }
      for (int i = 0; i < m; ++i) { double e = u[i] - s->Uref[(size_t)k * m + i]; l += 0.5 * s->Rd[i] * e * e; }
      ck[k] = l * s->dt;
    } else {
      for (int i = 0; i < n; ++i) { double e = x[i] - s->Xref[(size_t)k * n + i]; l += 0.5 * s->Qfd[i] * e * e; }
      ck[k] = l;
    }
  }
  double cv[2 * 64];
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    int rows = c->p;
    for (int k = c->k0; k <= c->k1; ++k) {
      size_t off = (size_t)(k - c->k0) * rows;
      con_eval(s, c, k, X + (size_t)k * n, U + (size_t)(k < N - 1 ? k : 0) * m, cv);
      ck[k] += con_cost(c, cv, c->lam + off, c->mu + off);
      for (int r = 0; r < rows; ++r) lb[k] -= c->lam[off + r] * c->lam[off + r] / (2.0 * c->mu[off + (c->kind == ORC_SOC ? 0 : r)]);
    }
  }
  double all = 0, suf = 0;
  for (int k = 0; k < N; ++k) all += lb[k];
  double part = 0; int x_suf = N, x_crude = N;
  suf = all;
  for (int k = 0; k < N; ++k) {
    part += ck[k]; suf -= lb[k];
    if (x_suf == N && part + suf > J_prev) x_suf = k + 1;
    if (x_crude == N && part + all > J_prev) x_crude = k + 1;
  }
  /* the same sweep from the last knot down (o[3]) */
  int x_rev = N;
  part = 0; suf = all;
  for (int k = N - 1; k >= 0; --k) {
    part += ck[k]; suf -= lb[k];
    if (x_rev == N && part + suf > J_prev) x_rev = N - k;
  }
  double* o = s->dbg_ls + 4 * (size_t)s->dbg_ls_n++;
  o[0] = (double)x_suf / N; o[1] = (double)x_crude / N; o[2] = J - J_prev; o[3] = (double)x_rev / N;
  free(ck); free(lb);
}

/* forwardpass!  (P7).  Line search on alpha = 1, 1/2, ...  [PKG] */
static double forward_pass(orc_solver* s, const double dV[2], double J_prev, double* alpha_out, double* cmax_out) {
  const orc_opts* o = &s->opts;
  double J = INFINITY, alpha = 1.0, z = -1.0, expected = 0.0, cmax = 0.0;
  int iter = 0;
  while ((z <= o->line_search_lower_bound || z > o->line_search_upper_bound) && J >= J_prev) {
    if (iter > o->iterations_linesearch) {
      /* failed: keep the current trajectory, bump regularisation */
      memcpy(s->Xb, s->X, (size_t)s->N * s->n * sizeof(double));
      memcpy(s->Ub, s->U, (size_t)(s->N - 1) * s->m * sizeof(double));
      J = total_cost(s, s->Xb, s->Ub, &cmax);
      z = 0; alpha = 0.0; expected = 0.0;
      reg_update(s, 1);
      s->rho += o->bp_reg_fp;
      break;
    }
    if (!rollout_alpha(s, alpha)) { iter++; alpha /= 2.0; continue; }
    J = total_cost(s, s->Xb, s->Ub, &cmax);
    if (s->dbg_ls && J >= J_prev) dbg_rejected_trial(s, s->Xb, s->Ub, J, J_prev);
    expected = -alpha * (dV[0] + alpha * dV[1]);
    z = expected > 0.0 ? (J_prev - J) / expected : -1.0;
    iter++;
    alpha /= 2.0;
  }
  *alpha_out = 2.0 * alpha;
  *cmax_out = cmax;
  (void)expected;
  return J;
}

/* gradient_todorov!: mean_k max_i |d_k,i| / (|u_k,i| + 1) [PKG] */
static double gradient_todorov(const orc_solver* s) {
  int m = s->m, N = s->N;
  double acc = 0;
  for (int k = 0; k < N - 1; ++k) {
    double mx = 0;
    for (int i = 0; i < m; ++i) {
      double v = fabs(s->d[(size_t)k * m + i]) / (fabs(s->U[(size_t)k * m + i]) + 1.0);
      if (v > mx) mx = v;
    }
    acc += mx;
  }
  return acc / (N - 1);
}

/* solve!(::iLQRSolver)  (P3).  Returns the final cost; *cmax = violation of Z. */
static double ilqr_solve(orc_solver* s, double cost_tol, double grad_tol, double* cmax) {
  const orc_opts* o = &s->opts;
  orc_stats* st = &s->stats;
  s->rho = o->bp_reg_initial; s->drho = 0.0; s->dJ_zero_counter = 0;
  if (!rollout_open(s)) { st->status = ORC_STATE_LIMIT; *cmax = INFINITY; return INFINITY; }
  double J_prev = total_cost(s, s->X, s->U, cmax);
  double J = J_prev;
  for (int i = 0; i < o->iterations_inner; ++i) {
    double dV[2], alpha, cm;
    cost_expansion(s);
    if (backward_pass(s, dV)) { st->status = ORC_NO_PROGRESS; break; }
    J = forward_pass(s, dV, J_prev, &alpha, &cm);
    if (J > o->max_cost_value) { st->status = ORC_MAXIMUM_COST; break; }
    /* copy_trajectories! */
    memcpy(s->X, s->Xb, (size_t)s->N * s->n * sizeof(double));
    memcpy(s->U, s->Ub, (size_t)(s->N - 1) * s->m * sizeof(double));
    *cmax = cm;
    double dJ = fabs(J - J_prev);
    J_prev = J;
    double grad = gradient_todorov(s);
    /* record_iteration! */
    int it = st->iterations;
    if (it < ORC_TRACE_MAX) { st->J[it] = J; st->dJ[it] = dJ; st->grad[it] = grad; st->alpha[it] = alpha; st->cmax_it[it] = cm; }
    st->iterations = it + 1;
    if (dJ == 0.0) s->dJ_zero_counter++; else s->dJ_zero_counter = 0;
    /* evaluate_convergence: (0 <= dJ < cost_tol) && grad < grad_tol.  The non-strict lower
     * bound is pinned by the reference's stored statistics (horizon_comp.jld2 :iter, median 2,
     * max 5, every solve SOLVE_SUCCEEDED): when iteration 1 lands exactly on the optimum of the
     * AL sub-problem, iteration 2 has dJ == 0, and a strict test would end in NO_PROGRESS. */
    if (dJ < cost_tol && grad < grad_tol) break;
    if (st->iterations >= o->iterations) { st->status = ORC_MAX_ITERATIONS; break; }
    if (s->dJ_zero_counter > o->dJ_counter_limit) { st->status = ORC_NO_PROGRESS; break; }
  }
  return J;
}

/* dual_update! + penalty_update!  (P9) */
static void dual_penalty_update(orc_solver* s) {
  const orc_opts* o = &s->opts;
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    int p = c->p;
    double phi = isnan(o->penalty_scaling) ? c->phi : o->penalty_scaling;
    for (int kk = 0; kk < c->nk; ++kk) {
      double *lam = c->lam + (size_t)kk * p, *mu = c->mu + (size_t)kk * p, *cv = c->c + (size_t)kk * p;
      if (c->kind == ORC_SOC) {
        double lb[64];
        for (int r = 0; r < p; ++r) lb[r] = lam[r] - mu[0] * cv[r];
        soc_project(lb, p, lam);
      } else {
        for (int r = 0; r < p; ++r) {
          if (!(cv[r] > -INFINITY)) continue;
          double v = lam[r] + mu[r] * cv[r];
          double lo = (c->sense == ORC_EQ) ? -o->dual_max : 0.0;
          lam[r] = fmin(fmax(v, lo), o->dual_max);
        }
      }
      for (int r = 0; r < p; ++r) mu[r] = fmin(fmax(phi * mu[r], 0.0), o->penalty_max);
    }
  }
}

static double penalty_max_now(const orc_solver* s) {
  double v = 0;
  for (int ci = 0; ci < s->ncon; ++ci) {
    const con_t* c = &s->con[ci];
    for (size_t i = 0; i < (size_t)c->nk * c->p; ++i) if (c->mu[i] > v) v = c->mu[i];
  }
  return v;
}

/* ---------------------------------------------------------------- projected-Newton polish
 * solve!(::ProjectedNewtonSolver) of Altro.jl [PKG]; ALTRO (Howell, Jackson, Manchester, IROS 2019) Algorithm 4.
 * PARITY UNPINNED: no output of the polish is stored in the reference.
 *
 * Primal variables z = (x_0, u_0, ..., x_{N-1}); constraints d(z) = 0: the initial condition x_0 - x0, the dynamics
 * defects A x_k + B u_k + f - x_{k+1}, and the ACTIVE rows of the problem's constraints (equalities; inequality rows
 * with c >= -active_set_tolerance_pn; a second-order cone (v, t) through h = ||v|| - t with the same test).  One
 * projection step is the minimum-norm correction in the metric of the cost Hessian H (diagonal here, + rho_primal):
 *     dz = -H^-1 D' (D H^-1 D')^-1 d,      S = D H^-1 D'
 * S is block tridiagonal over the knots (block k: [initial condition if k = 0; active stage rows of knot k; defect k]),
 * factored as S + rho_chol I = L L' block by block; reg_solve refines the solution against S itself.  A step is
 * accepted if it lowers ||d||_inf (else halved, at most 10 times); the same factors serve further steps while
 * log(viol)/log(viol_prev) stays above r_threshold; the active set and the linearisation are renewed up to 10 times. */
typedef struct {
  int bmax;        /* rows a block can hold */
  int* nb;         /* [N] rows of block k */
  int* nst;        /* [N] of which stage rows (after the n initial-condition rows of block 0) */
  int* rcon;       /* [N][bmax] constraint index of a stage row */
  int* rrow;       /* [N][bmax] row inside that constraint (SOC: 0) */
  double* E;       /* [N][bmax][nz]  Jacobian of block k wrt z_k */
  double* dv;      /* [N][bmax] values */
  double* Ld;      /* [N][bmax][bmax] diagonal Cholesky blocks (lower) */
  double* Lo;      /* [N][bmax][bmax] L_{k,k-1} (rows of block k, columns of block k-1) */
  double* hinv;    /* [N][nz] 1 / (H + rho_primal) */
} pn_ws;

static double* pn_hdiag(const orc_solver* s, int k, double* h) {
  int n = s->n, m = s->m;
  for (int i = 0; i < n; ++i) h[i] = ((k < s->N - 1) ? s->dt * s->Qd[i] : s->Qfd[i]) + s->opts.rho_primal;
  for (int i = 0; i < m; ++i) h[n + i] = ((k < s->N - 1) ? s->dt * s->Rd[i] : 0.0) + s->opts.rho_primal;
  return h;
}

/* value (and, if E != NULL, the Jacobian row wrt z_k) of stage row (ci, r) at knot k */
static double pn_row(const orc_solver* s, int ci, int r, int k, const double* x, const double* u, double* E) {
  const con_t* c = &s->con[ci];
  int n = s->n, nz = s->nz, terminal = (k == s->N - 1);
  double cv[64];
  con_eval(s, c, k, x, u, cv);
  if (E) memset(E, 0, nz * sizeof(double));
  if (c->kind == ORC_BOX) {
    int j = r % nz;
    if (E) E[j] = (r < nz) ? 1.0 : -1.0;
    return cv[r];
  }
  size_t blk = c->per_knot ? (size_t)(k - c->k0) : 0;
  const double* A = c->A + blk * c->p * nz;
  int ncol = terminal ? n : nz;
  if (c->kind == ORC_LINEAR) {
    if (E) for (int j = 0; j < ncol; ++j) E[j] = A[r * nz + j];
    return cv[r];
  }
  /* SOC: h = ||v|| - t,  grad = (v/||v||)' A_v - A_t */
  int q = c->p - 1;
  double nv = 0;
  for (int i = 0; i < q; ++i) nv += cv[i] * cv[i];
  nv = sqrt(nv);
  if (E) {
    for (int j = 0; j < ncol; ++j) {
      double g = -A[q * nz + j];
      if (nv > 0) for (int i = 0; i < q; ++i) g += cv[i] / nv * A[i * nz + j];
      E[j] = g;
    }
  }
  return nv - cv[q];
}

/* active set + linearisation at (X, U): fills nb, nst, rcon, rrow, E, dv; returns ||d||_inf */
static double pn_linearise(const orc_solver* s, pn_ws* w, const double* X, const double* U) {
  int n = s->n, m = s->m, N = s->N, nz = s->nz, bm = w->bmax;
  double tol = s->opts.active_set_tolerance_pn, viol = 0;
  for (int k = 0; k < N; ++k) {
    const double* x = X + (size_t)k * n;
    const double* u = U + (size_t)(k < N - 1 ? k : 0) * m;
    double* E = w->E + (size_t)k * bm * nz;
    double* dv = w->dv + (size_t)k * bm;
    int nb = 0;
    memset(E, 0, (size_t)bm * nz * sizeof(double));
    if (k == 0) {
      for (int i = 0; i < n; ++i) { E[(size_t)nb * nz + i] = 1.0; dv[nb] = x[i] - s->x0[i]; nb++; }
    }
    int nst = 0;
    for (int ci = 0; ci < s->ncon; ++ci) {
      const con_t* c = &s->con[ci];
      if (k < c->k0 || k > c->k1) continue;
      int rows = (c->kind == ORC_SOC) ? 1 : c->p;
      for (int r = 0; r < rows; ++r) {
        double v = pn_row(s, ci, r, k, x, u, NULL);
        if (!(v > -INFINITY)) continue;
        int act = (c->kind != ORC_SOC && c->sense == ORC_EQ) || (v >= -tol);
        if (!act) continue;
        dv[nb] = pn_row(s, ci, r, k, x, u, E + (size_t)nb * nz);
        w->rcon[(size_t)k * bm + nst] = ci;
        w->rrow[(size_t)k * bm + nst] = r;
        nb++; nst++;
      }
    }
    w->nst[k] = nst;
    if (k < N - 1) {
      const double* A = s->A + (s->ltv ? (size_t)k * n * n : 0);
      const double* B = s->B + (s->ltv ? (size_t)k * n * m : 0);
      double xn[64];
      dynamics(s, k, x, u, xn);
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) E[(size_t)nb * nz + j] = A[i + n * j];
        for (int j = 0; j < m; ++j) E[(size_t)nb * nz + n + j] = B[i + n * j];
        dv[nb] = xn[i] - X[(size_t)(k + 1) * n + i];
        nb++;
      }
    }
    w->nb[k] = nb;
    for (int r = 0; r < nb; ++r) if (fabs(dv[r]) > viol) viol = fabs(dv[r]);
  }
  return viol;
}

/* values of the SAME rows at another trajectory (line search); returns ||d||_inf */
static double pn_values(const orc_solver* s, const pn_ws* w, const double* X, const double* U, double* dv_out) {
  int n = s->n, m = s->m, N = s->N, bm = w->bmax;
  double viol = 0;
  for (int k = 0; k < N; ++k) {
    const double* x = X + (size_t)k * n;
    const double* u = U + (size_t)(k < N - 1 ? k : 0) * m;
    double* dv = dv_out + (size_t)k * bm;
    int nb = 0;
    if (k == 0) for (int i = 0; i < n; ++i) dv[nb++] = x[i] - s->x0[i];
    for (int q = 0; q < w->nst[k]; ++q) dv[nb++] = pn_row(s, w->rcon[(size_t)k * bm + q], w->rrow[(size_t)k * bm + q], k, x, u, NULL);
    if (k < N - 1) {
      double xn[64];
      dynamics(s, k, x, u, xn);
      for (int i = 0; i < n; ++i) dv[nb++] = xn[i] - X[(size_t)(k + 1) * n + i];
    }
    for (int r = 0; r < nb; ++r) if (fabs(dv[r]) > viol) viol = fabs(dv[r]);
  }
  return viol;
}

/* y = S v (block tridiagonal, applied through D and H^-1): t_k = H_k^-1 (E_k' v_k - [defect part of v_{k-1}]_x),
 * y_k = E_k t_k - [t_{k+1}]_x on the defect rows */
static void pn_apply_S(const orc_solver* s, const pn_ws* w, const double* v, double* y, double* tz /* [N][nz] */) {
  int n = s->n, N = s->N, nz = s->nz, bm = w->bmax;
  for (int k = 0; k < N; ++k) {
    const double* E = w->E + (size_t)k * bm * nz;
    double* t = tz + (size_t)k * nz;
    for (int j = 0; j < nz; ++j) {
      double acc = 0;
      for (int r = 0; r < w->nb[k]; ++r) acc += E[(size_t)r * nz + j] * v[(size_t)k * bm + r];
      t[j] = acc;
    }
    if (k > 0) {
      int off = w->nb[k - 1] - n;  /* defect rows of block k-1 are its last n rows */
      for (int i = 0; i < n; ++i) t[i] -= v[(size_t)(k - 1) * bm + off + i];
    }
    for (int j = 0; j < nz; ++j) t[j] *= w->hinv[(size_t)k * nz + j];
  }
  for (int k = 0; k < N; ++k) {
    const double* E = w->E + (size_t)k * bm * nz;
    for (int r = 0; r < w->nb[k]; ++r) {
      double acc = 0;
      for (int j = 0; j < nz; ++j) acc += E[(size_t)r * nz + j] * tz[(size_t)k * nz + j];
      y[(size_t)k * bm + r] = acc;
    }
    if (k < N - 1) {
      int off = w->nb[k] - n;
      for (int i = 0; i < n; ++i) y[(size_t)k * bm + off + i] -= tz[(size_t)(k + 1) * nz + i];
    }
  }
}

/* S + rho_chol I = L L', block by block.  Returns 0 on success. */
static int pn_factor(const orc_solver* s, pn_ws* w) {
  int n = s->n, N = s->N, nz = s->nz, bm = w->bmax;
  double rho = s->opts.rho_chol;
  for (int k = 0; k < N; ++k) {
    int nb = w->nb[k];
    const double* E = w->E + (size_t)k * bm * nz;
    double* Ld = w->Ld + (size_t)k * bm * bm;
    double* Lo = w->Lo + (size_t)k * bm * bm;
    /* S_kk = E H^-1 E' (+ H_x,k+1^-1 on the defect rows) + rho I */
    for (int r = 0; r < nb; ++r)
      for (int c = 0; c <= r; ++c) {
        double acc = 0;
        for (int j = 0; j < nz; ++j) acc += E[(size_t)r * nz + j] * w->hinv[(size_t)k * nz + j] * E[(size_t)c * nz + j];
        Ld[(size_t)r * bm + c] = acc;
      }
    if (k < N - 1) {
      int off = nb - n;
      for (int i = 0; i < n; ++i) Ld[(size_t)(off + i) * bm + off + i] += w->hinv[(size_t)(k + 1) * nz + i];
    }
    for (int r = 0; r < nb; ++r) Ld[(size_t)r * bm + r] += rho;
    if (k > 0) {
      /* S_{k,k-1}[r][c]: c a defect row i of block k-1:  -E_k[r][i] / h_k[i];  L_{k,k-1} = S_{k,k-1} L_{k-1,k-1}^-T */
      int pb = w->nb[k - 1], poff = pb - n;
      const double* Lp = w->Ld + (size_t)(k - 1) * bm * bm;
      for (int r = 0; r < nb; ++r) {
        for (int c = 0; c < pb; ++c) {
          double v = (c >= poff) ? -E[(size_t)r * nz + (c - poff)] * w->hinv[(size_t)k * nz + (c - poff)] : 0.0;
          for (int q = 0; q < c; ++q) v -= Lo[(size_t)r * bm + q] * Lp[(size_t)c * bm + q];
          Lo[(size_t)r * bm + c] = v / Lp[(size_t)c * bm + c];
        }
      }
      for (int r = 0; r < nb; ++r)
        for (int c = 0; c <= r; ++c) {
          double acc = 0;
          for (int q = 0; q < pb; ++q) acc += Lo[(size_t)r * bm + q] * Lo[(size_t)c * bm + q];
          Ld[(size_t)r * bm + c] -= acc;
        }
    }
    for (int c = 0; c < nb; ++c) {  /* Cholesky of the block, in place (lower) */
      double dd = Ld[(size_t)c * bm + c];
      for (int q = 0; q < c; ++q) dd -= Ld[(size_t)c * bm + q] * Ld[(size_t)c * bm + q];
      if (!(dd > 0.0)) return 1;
      dd = sqrt(dd);
      Ld[(size_t)c * bm + c] = dd;
      for (int r = c + 1; r < nb; ++r) {
        double v = Ld[(size_t)r * bm + c];
        for (int q = 0; q < c; ++q) v -= Ld[(size_t)r * bm + q] * Ld[(size_t)c * bm + q];
        Ld[(size_t)r * bm + c] = v / dd;
      }
    }
  }
  return 0;
}

/* x = (L L')^-1 b */
static void pn_chol_solve(const orc_solver* s, const pn_ws* w, const double* b, double* x) {
  int N = s->N, bm = w->bmax;
  for (int k = 0; k < N; ++k) {  /* forward */
    int nb = w->nb[k];
    const double* Ld = w->Ld + (size_t)k * bm * bm;
    const double* Lo = w->Lo + (size_t)k * bm * bm;
    for (int r = 0; r < nb; ++r) {
      double v = b[(size_t)k * bm + r];
      if (k > 0) for (int q = 0; q < w->nb[k - 1]; ++q) v -= Lo[(size_t)r * bm + q] * x[(size_t)(k - 1) * bm + q];
      for (int q = 0; q < r; ++q) v -= Ld[(size_t)r * bm + q] * x[(size_t)k * bm + q];
      x[(size_t)k * bm + r] = v / Ld[(size_t)r * bm + r];
    }
  }
  for (int k = N - 1; k >= 0; --k) {  /* backward */
    int nb = w->nb[k];
    const double* Ld = w->Ld + (size_t)k * bm * bm;
    for (int r = nb - 1; r >= 0; --r) {
      double v = x[(size_t)k * bm + r];
      if (k < N - 1) {
        const double* Ln = w->Lo + (size_t)(k + 1) * bm * bm;
        for (int q = 0; q < w->nb[k + 1]; ++q) v -= Ln[(size_t)q * bm + r] * x[(size_t)(k + 1) * bm + q];
      }
      for (int q = r + 1; q < nb; ++q) v -= Ld[(size_t)q * bm + r] * x[(size_t)k * bm + q];
      x[(size_t)k * bm + r] = v / Ld[(size_t)r * bm + r];
    }
  }
}

/* multiplier_projection! of Altro.jl's ProjectedNewtonSolver [PKG] (ALTRO, IROS 2019, section IV-B): the least-squares
 * multipliers of the polish's active rows D (initial condition, active stage rows, dynamics defects) at the polished
 * trajectory,
 *     lam <- lam - (D D')^-1 D (g + D' lam),        g = gradient of the cost (no AL terms),
 * and the stationarity residual ||g + D' lam||_2 before and after.  D D' has the block structure of S with the metric
 * H = I, so the same block factorisation serves (pn_factor with hinv = 1; reg_solve refinement against D D' itself).
 * lam starts from the AL duals of the active box / linear rows and from zero for cones (the polish sees a cone as the
 * one row ||v|| - t), the initial condition and the dynamics; the projected multipliers do not depend on that start
 * when D D' is regular, only the "before" residual does.  They are NOT written back into the AL duals (Altro's own
 * polish keeps its multiplier vector to itself; PARITY UNPINNED as for the primal half): the two residuals are the
 * output, stats.pn_dual_residual0 / pn_dual_residual.  Called with w linearised at the final (X, U). */
static void pn_Dt(const orc_solver* s, const pn_ws* w, const double* v, double* tz) {   /* tz = D' v  (metric-free) */
  int n = s->n, N = s->N, nz = s->nz, bm = w->bmax;
  for (int k = 0; k < N; ++k) {
    const double* E = w->E + (size_t)k * bm * nz;
    double* t = tz + (size_t)k * nz;
    for (int j = 0; j < nz; ++j) {
      double acc = 0;
      for (int r = 0; r < w->nb[k]; ++r) acc += E[(size_t)r * nz + j] * v[(size_t)k * bm + r];
      t[j] = acc;
    }
    if (k > 0) {
      int off = w->nb[k - 1] - n;
      for (int i = 0; i < n; ++i) t[i] -= v[(size_t)(k - 1) * bm + off + i];
    }
  }
}

static void pn_D(const orc_solver* s, const pn_ws* w, const double* tz, double* y) {   /* y = D tz */
  int n = s->n, N = s->N, nz = s->nz, bm = w->bmax;
  for (int k = 0; k < N; ++k) {
    const double* E = w->E + (size_t)k * bm * nz;
    for (int r = 0; r < w->nb[k]; ++r) {
      double acc = 0;
      for (int j = 0; j < nz; ++j) acc += E[(size_t)r * nz + j] * tz[(size_t)k * nz + j];
      y[(size_t)k * bm + r] = acc;
    }
    if (k < N - 1) {
      int off = w->nb[k] - n;
      for (int i = 0; i < n; ++i) y[(size_t)k * bm + off + i] -= tz[(size_t)(k + 1) * nz + i];
    }
  }
}

static int multiplier_projection(orc_solver* s, pn_ws* w, double* res0_out, double* res_out) {
  int n = s->n, m = s->m, N = s->N, nz = s->nz, bm = w->bmax;
  double *lam = dalloc((size_t)N * bm), *rhs = dalloc((size_t)N * bm), *del = dalloc((size_t)N * bm), *cor = dalloc((size_t)N * bm);
  double *Av = dalloc((size_t)N * bm), *g = dalloc((size_t)N * nz), *tz = dalloc((size_t)N * nz), *r0 = dalloc((size_t)N * nz);
  for (size_t i = 0; i < (size_t)N * nz; ++i) w->hinv[i] = 1.0;          /* the metric of D D' */
  /* cost gradient (diagonal tracking cost): g_k = H_k (z_k - zref_k) */
  for (int k = 0; k < N; ++k) {
    for (int i = 0; i < n; ++i)
      g[(size_t)k * nz + i] = ((k < N - 1) ? s->dt * s->Qd[i] : s->Qfd[i]) * (s->X[(size_t)k * n + i] - s->Xref[(size_t)k * n + i]);
    for (int i = 0; i < m; ++i)
      g[(size_t)k * nz + n + i] = (k < N - 1) ? s->dt * s->Rd[i] * (s->U[(size_t)k * m + i] - s->Uref[(size_t)k * m + i]) : 0.0;
  }
  /* lam0: AL duals of the active box / linear rows */
  for (int k = 0; k < N; ++k) {
    int base = (k == 0) ? n : 0;
    for (int q = 0; q < w->nst[k]; ++q) {
      const con_t* c = &s->con[w->rcon[(size_t)k * bm + q]];
      int r = w->rrow[(size_t)k * bm + q];
      lam[(size_t)k * bm + base + q] = (c->kind == ORC_SOC) ? 0.0 : c->lam[(size_t)(k - c->k0) * c->p + r];
    }
  }
  pn_Dt(s, w, lam, tz);
  double res0 = 0;
  for (size_t i = 0; i < (size_t)N * nz; ++i) { r0[i] = g[i] + tz[i]; res0 += r0[i] * r0[i]; }
  int rc = pn_factor(s, w);
  if (!rc) {
    pn_D(s, w, r0, rhs);
    pn_chol_solve(s, w, rhs, del);
    for (int it = 0; it < 25; ++it) {                                      /* reg_solve against D D' */
      pn_Dt(s, w, del, tz);
      pn_D(s, w, tz, Av);
      double rn = 0;
      for (int k = 0; k < N; ++k)
        for (int r = 0; r < w->nb[k]; ++r) {
          double e = rhs[(size_t)k * bm + r] - Av[(size_t)k * bm + r];
          Av[(size_t)k * bm + r] = e;
          if (fabs(e) > rn) rn = fabs(e);
        }
      if (rn < 1e-8) break;
      pn_chol_solve(s, w, Av, cor);
      for (int k = 0; k < N; ++k) for (int r = 0; r < w->nb[k]; ++r) del[(size_t)k * bm + r] += cor[(size_t)k * bm + r];
    }
    for (int k = 0; k < N; ++k) for (int r = 0; r < w->nb[k]; ++r) lam[(size_t)k * bm + r] -= del[(size_t)k * bm + r];
  }
  pn_Dt(s, w, lam, tz);
  double res = 0;
  for (size_t i = 0; i < (size_t)N * nz; ++i) { double e = g[i] + tz[i]; res += e * e; }
  *res0_out = sqrt(res0);
  *res_out = sqrt(res);
  free(lam); free(rhs); free(del); free(cor); free(Av); free(g); free(tz); free(r0);
  return rc;
}

static int projected_newton(orc_solver* s, double* viol_out) {
  const orc_opts* o = &s->opts;
  int n = s->n, m = s->m, N = s->N, nz = s->nz;
  int pm = 0;
  for (int ci = 0; ci < s->ncon; ++ci) pm += (s->con[ci].kind == ORC_SOC) ? 1 : s->con[ci].p;
  pn_ws w;
  w.bmax = 2 * n + pm;
  int bm = w.bmax;
  w.nb = (int*)calloc(N, sizeof(int)); w.nst = (int*)calloc(N, sizeof(int));
  w.rcon = (int*)calloc((size_t)N * bm, sizeof(int)); w.rrow = (int*)calloc((size_t)N * bm, sizeof(int));
  w.E = dalloc((size_t)N * bm * nz); w.dv = dalloc((size_t)N * bm);
  w.Ld = dalloc((size_t)N * bm * bm); w.Lo = dalloc((size_t)N * bm * bm); w.hinv = dalloc((size_t)N * nz);
  double *lam = dalloc((size_t)N * bm), *res = dalloc((size_t)N * bm), *cor = dalloc((size_t)N * bm), *Sv = dalloc((size_t)N * bm);
  double *tz = dalloc((size_t)N * nz), *dz = dalloc((size_t)N * nz), *dtrial = dalloc((size_t)N * bm);
  for (int k = 0; k < N; ++k) {
    double h[128];
    pn_hdiag(s, k, h);
    for (int j = 0; j < nz; ++j) w.hinv[(size_t)k * nz + j] = 1.0 / h[j];
  }
  double viol = pn_linearise(s, &w, s->X, s->U);
  int rc = 0;
  for (int outer = 0; outer <= 10 && viol > o->constraint_tolerance; ++outer) {   /* projection_solve! */
    if (outer > 0) viol = pn_linearise(s, &w, s->X, s->U);
    if (pn_factor(s, &w)) { rc = 1; break; }
    double viol_prev = viol;
    for (int refine = 0; refine < 10; ++refine) {                                   /* _projection_solve! */
      /* reg_solve: S lam = d with the factors of S + rho I, refined against S (tol 1e-8, at most 25 rounds) */
      pn_chol_solve(s, &w, w.dv, lam);
      for (int it = 0; it < 25; ++it) {
        pn_apply_S(s, &w, lam, Sv, tz);
        double rn = 0;
        for (int k = 0; k < N; ++k)
          for (int r = 0; r < w.nb[k]; ++r) {
            double e = w.dv[(size_t)k * bm + r] - Sv[(size_t)k * bm + r];
            res[(size_t)k * bm + r] = e;
            if (fabs(e) > rn) rn = fabs(e);
          }
        if (rn < 1e-8) break;
        pn_chol_solve(s, &w, res, cor);
        for (int k = 0; k < N; ++k) for (int r = 0; r < w.nb[k]; ++r) lam[(size_t)k * bm + r] += cor[(size_t)k * bm + r];
      }
      /* dz = -H^-1 D' lam */
      pn_apply_S(s, &w, lam, Sv, tz);
      for (size_t i = 0; i < (size_t)N * nz; ++i) dz[i] = -tz[i];
      /* _projection_linesearch! */
      double alpha = 1.0, v_new = viol;
      for (int ls = 0;; ++ls) {
        for (int k = 0; k < N; ++k) {
          for (int i = 0; i < n; ++i) s->Xb[(size_t)k * n + i] = s->X[(size_t)k * n + i] + alpha * dz[(size_t)k * nz + i];
          if (k < N - 1) for (int i = 0; i < m; ++i) s->Ub[(size_t)k * m + i] = s->U[(size_t)k * m + i] + alpha * dz[(size_t)k * nz + n + i];
        }
        v_new = pn_values(s, &w, s->Xb, s->Ub, dtrial);
        if (v_new < viol || ls >= 10) break;
        alpha *= 0.5;
      }
      memcpy(s->X, s->Xb, (size_t)N * n * sizeof(double));
      memcpy(s->U, s->Ub, (size_t)(N - 1) * m * sizeof(double));
      memcpy(w.dv, dtrial, (size_t)N * bm * sizeof(double));
      viol = v_new;
      double rate = log10(viol) / log10(viol_prev);
      viol_prev = viol;
      if (viol < o->constraint_tolerance) break;
      if (rate < o->r_threshold) break;
    }
  }
  *viol_out = viol;
  if (!rc) {   /* the dual half: multipliers of the active rows at the polished trajectory */
    pn_linearise(s, &w, s->X, s->U);
    double r0 = 0, r1 = 0;
    s->stats.pn_dual_failed = multiplier_projection(s, &w, &r0, &r1);
    s->stats.pn_dual_residual0 = r0;
    s->stats.pn_dual_residual = r1;
  }
  free(w.nb); free(w.nst); free(w.rcon); free(w.rrow); free(w.E); free(w.dv); free(w.Ld); free(w.Lo); free(w.hinv);
  free(lam); free(res); free(cor); free(Sv); free(tz); free(dz); free(dtrial);
  return rc;
}

/* objective (no AL terms) and violation of the problem's constraints at (X, U) as they are (not rolled out) */
static double objective_and_violation(orc_solver* s, double* cmax) {
  int n = s->n, m = s->m, N = s->N;
  double J = 0, vmax = 0;
  for (int k = 0; k < N; ++k) {
    const double* x = s->X + (size_t)k * n;
    double l = 0;
    const double* Qd = (k < N - 1) ? s->Qd : s->Qfd;
    for (int i = 0; i < n; ++i) { double e = x[i] - s->Xref[(size_t)k * n + i]; l += 0.5 * Qd[i] * e * e; }
    if (k < N - 1) for (int i = 0; i < m; ++i) { double e = s->U[(size_t)k * m + i] - s->Uref[(size_t)k * m + i]; l += 0.5 * s->Rd[i] * e * e; }
    J += (k < N - 1) ? l * s->dt : l;
  }
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    for (int k = c->k0; k <= c->k1; ++k) {
      size_t off = (size_t)(k - c->k0) * c->p;
      con_eval(s, c, k, s->X + (size_t)k * n, s->U + (size_t)(k < N - 1 ? k : 0) * m, c->c + off);
      double v = con_violation(c, c->c + off);
      if (v > vmax) vmax = v;
    }
  }
  *cmax = vmax;
  return J;
}

/* solve!(::ALTROSolver) -> solve!(::AugmentedLagrangianSolver)  (P2).
 * projected_newton is false in every MPC benchmark of the reference
 * (run_random_linear.jl:48) and is not restated. */
void orc_solve(orc_solver* s) {
  orc_opts opts_al = s->opts;         /* solve!(::ALTROSolver): with the polish on, the AL stage only has to reach its tolerance */
  const orc_opts user = s->opts;
  if (user.projected_newton) {
    if (user.projected_newton_tolerance >= 0) opts_al.constraint_tolerance = user.projected_newton_tolerance;
    else { opts_al.constraint_tolerance = 0; opts_al.kickout_max_penalty = 1; }
  }
  s->opts = opts_al;
  const orc_opts* o = &s->opts;
  orc_stats* st = &s->stats;
  memset(st, 0, sizeof(*st));
  st->status = ORC_UNSOLVED;
  /* initialize!/reset!: duals and penalties */
  for (int ci = 0; ci < s->ncon; ++ci) {
    con_t* c = &s->con[ci];
    size_t tot = (size_t)c->nk * c->p;
    if (o->reset_duals) memset(c->lam, 0, tot * sizeof(double));
    if (o->reset_penalties) {
      double mu0 = isnan(o->penalty_initial) ? c->mu0 : o->penalty_initial;
      for (size_t i = 0; i < tot; ++i) c->mu[i] = mu0;
    }
  }
  double cmax = 0, J = 0;
  if (s->ncon == 0) {
    J = ilqr_solve(s, o->cost_tolerance, o->gradient_tolerance, &cmax);
    st->cost = J; st->c_max = 0;
    if (st->status == ORC_UNSOLVED) st->status = ORC_SOLVE_SUCCEEDED;
    s->opts = user;
    return;
  }
  for (int j = 0; j < o->iterations_outer; ++j) {
    /* set_tolerances!: intermediate tolerances except on the last allowed outer iteration */
    double ct = (j != o->iterations_outer - 1) ? o->cost_tolerance_intermediate : o->cost_tolerance;
    double gt = (j != o->iterations_outer - 1) ? o->gradient_tolerance_intermediate : o->gradient_tolerance;
    J = ilqr_solve(s, ct, gt, &cmax);
    int jo = st->iterations_outer;
    if (jo < 64) { st->c_max_outer[jo] = cmax; st->penalty_max_outer[jo] = penalty_max_now(s); }
    st->iterations_outer = jo + 1;
    if (st->status > ORC_SOLVE_SUCCEEDED) break;
    /* evaluate_convergence(::AugmentedLagrangianSolver) */
    if (cmax < o->constraint_tolerance || (o->kickout_max_penalty && penalty_max_now(s) >= o->penalty_max)) break;
    if (j == o->iterations_outer - 1) { st->status = ORC_MAX_ITERATIONS_OUTER; break; }
    dual_penalty_update(s);
  }
  st->cost = J; st->c_max = cmax;
  s->opts = user;
  if (st->status <= ORC_SOLVE_SUCCEEDED && user.projected_newton && cmax > user.constraint_tolerance) {
    double dviol;
    st->pn_ran = 1;
    st->pn_failed = projected_newton(s, &dviol);
    st->pn_residual = dviol;                  /* ||d||_inf of the active rows, the initial condition and the dynamics defects */
    st->cost = objective_and_violation(s, &cmax);
    st->c_max = cmax;
  }
  if (st->status <= ORC_SOLVE_SUCCEEDED && cmax < user.constraint_tolerance) st->status = ORC_SOLVE_SUCCEEDED;
}

/* RD.shift_fill!(Z) (P11) and Altro.shift_fill!(conSet) (P10): shift by one knot, repeat the
 * last entry.  [PKG] */
void orc_shift_fill(orc_solver* s, int primal, int dual) {
  int n = s->n, m = s->m, N = s->N;
  if (primal) {
    memmove(s->X, s->X + n, (size_t)(N - 1) * n * sizeof(double));
    if (N > 2) memmove(s->U, s->U + m, (size_t)(N - 2) * m * sizeof(double));
  }
  if (dual) {
    for (int ci = 0; ci < s->ncon; ++ci) {
      con_t* c = &s->con[ci];
      if (c->nk > 1) {
        memmove(c->lam, c->lam + c->p, (size_t)(c->nk - 1) * c->p * sizeof(double));
        memmove(c->mu, c->mu + c->p, (size_t)(c->nk - 1) * c->p * sizeof(double));
      }
    }
  }
}

/* ---------------------------------------------------------------- accessors (P13) */
const double* orc_states(const orc_solver* s) { return s->X; }
const double* orc_controls(const orc_solver* s) { return s->U; }
const double* orc_gain_K(const orc_solver* s) { return s->K; }
const double* orc_gain_d(const orc_solver* s) { return s->d; }
const orc_stats* orc_get_stats(const orc_solver* s) { return &s->stats; }
int orc_num_duals(const orc_solver* s, int con) { return s->con[con].nk * s->con[con].p; }
const double* orc_duals(const orc_solver* s, int con) { return s->con[con].lam; }
const double* orc_penalties(const orc_solver* s, int con) { return s->con[con].mu; }
void orc_set_duals(orc_solver* s, int con, const double* lam) {
  memcpy(s->con[con].lam, lam, (size_t)orc_num_duals(s, con) * sizeof(double));
}

double orc_cost(orc_solver* s) {
  rollout_open(s);
  return total_cost(s, s->X, s->U, NULL);
}

double orc_max_violation(orc_solver* s) {
  double cm = 0;
  rollout_open(s);
  total_cost(s, s->X, s->U, &cm);
  return cm;
}
