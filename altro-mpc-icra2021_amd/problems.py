"""Synthetic workload generators restating the reference's benchmark problems (host side,
numpy only).  Julia's MersenneTwister stream is not reproducible outside Julia, so every
instance draws from its own counter-based Philox stream keyed by (seed, instance): instance i
is identical whatever the batch size or the rank that generates it.

Reference files restated here:
  benchmarks/random_linear_mpc/random_linear.jl:26-41      genA / genB / gendiscrete
  benchmarks/random_linear_mpc/random_linear_problem.jl:5-32  gen_random_linear (u_bnd = 3, dt = 0.1)
  benchmarks/random_linear_mpc/run_random_linear.jl:29-39  gen_trajectory
  benchmarks/mpc.jl:11-47                                  gen_tracking_problem (Q=10, R=0.1, Qf=10)
"""
from dataclasses import dataclass

import numpy as np


def instance_rng(seed, instance):
    return np.random.Generator(np.random.Philox(key=(int(seed) << 32) + int(instance)))


def gendiscrete(n, m, rng, tol=1e-4):
    """random_linear.jl:35-41: A = Q diag(v) Q', v = randn(n)/(||randn||_inf + tol); B = randn(n,m)."""
    v = rng.standard_normal(n)
    v = v / (np.abs(v).max() + tol)
    X = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(X)
    A = Q @ np.diag(v) @ Q.T
    Bm = rng.standard_normal((n, m))
    return A, Bm


@dataclass
class RandomLinearBatch:
    """B independent random-linear tracking-MPC problems (BASELINE configs 1, 2, 4)."""
    n: int
    m: int
    N: int            # MPC horizon (knot points)
    dt: float
    A: np.ndarray     # (B, n, n)
    Bm: np.ndarray    # (B, n, m)
    Xtrack: np.ndarray  # (B, Nt, n)   long reference trajectory (gen_trajectory)
    Utrack: np.ndarray  # (B, Nt-1, m)
    noise: np.ndarray   # (S, B, n) unit normals for the 1 % plant noise, one row per MPC step
    u_bnd: float = 3.0
    Qk: float = 10.0
    Rk: float = 0.1
    Qfk: float = 10.0

    @property
    def batch(self):
        return self.A.shape[0]

    @property
    def Nt(self):
        return self.Xtrack.shape[1]

    def window(self, k):
        """TO.update_trajectory!(obj, Z_track, k) with 0-based k: reference window k..k+N-1."""
        return self.Xtrack[:, k:k + self.N], self.Utrack[:, k:k + self.N - 1]


def gen_random_linear_batch(batch, n=12, m=4, N=50, steps=100, dt=0.1, seed=1, first_instance=0,
                            u_bnd=3.0):
    """gen_trajectory + gen_tracking_problem for `batch` instances starting at global instance
    index `first_instance` (used to shard one logical batch over ranks)."""
    Nt = N + steps + 1
    A = np.empty((batch, n, n))
    Bm = np.empty((batch, n, m))
    U = np.empty((batch, Nt - 1, m))
    noise = np.empty((steps, batch, n))
    for b in range(batch):
        rng = instance_rng(seed, first_instance + b)
        A[b], Bm[b] = gendiscrete(n, m, rng)
        U[b] = rng.standard_normal((Nt - 1, m))
        noise[:, b] = rng.standard_normal((steps, n))
    X = np.zeros((batch, Nt, n))
    for k in range(Nt - 1):  # x_{k+1} = A x_k + B u_k from x_1 = 0 (run_random_linear.jl:33-35)
        X[:, k + 1] = np.einsum("bij,bj->bi", A, X[:, k]) + np.einsum("bij,bj->bi", Bm, U[:, k])
    return RandomLinearBatch(n=n, m=m, N=N, dt=dt, A=A, Bm=Bm, Xtrack=X, Utrack=U, noise=noise,
                             u_bnd=u_bnd)


# ---------------------------------------------------------------------------------------------
# flexible spacecraft: reference benchmarks/flexible_satellite/flexible_sat_mpc.jl
FLEXSAT_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-4,
                    penalty_initial=100.0, penalty_scaling=100.0)
"""set_options! of flexible_sat_mpc.jl:250-257 at tol = 1e-4 (reset_duals stays at its default, true)"""


def flexsat_model(dt=0.5):
    """Rigid hub with reaction wheels and three flexible modes, x = [MRP(3); omega(3); eta(3);
    eta_dot(3)], u = wheel torques (generate_AB, flexible_sat_mpc.jl:72-130), discretised with a
    zero-order hold by one matrix exponential of [[A, B], [0, 0]] dt (c2d, :59-70)."""
    from scipy.linalg import expm
    inertia = np.diag([1.0, 2.0, 3.0])
    coupling = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-0.7, 0.1, 0.1]])      # angular momentum coupling
    wn = 2.0 * np.pi * np.array([0.05, 0.2, 0.125])                               # modal frequencies
    zeta = np.full(3, 1e-3)
    Kmod, Cmod = np.diag(wn ** 2), np.diag(2.0 * zeta * wn)
    T = np.linalg.inv(inertia - coupling.T @ coupling)
    Ac = np.zeros((12, 12))
    Ac[0:3, 3:6] = 0.25 * np.eye(3)                      # MRP kinematics about the origin
    Ac[3:6, 6:9] = T @ coupling.T @ Kmod
    Ac[3:6, 9:12] = T @ coupling.T @ Cmod
    Ac[6:9, 9:12] = np.eye(3)
    Ac[9:12, 6:9] = -Kmod - coupling @ T @ coupling.T @ Kmod
    Ac[9:12, 9:12] = -Cmod - coupling @ T @ coupling.T @ Cmod
    Bc = np.zeros((12, 3))
    Bc[3:6] = -T
    Bc[9:12] = coupling @ T
    M = np.zeros((15, 15))
    M[:12, :12], M[:12, 12:] = Ac * dt, Bc * dt
    E = expm(M)
    return E[:12, :12].copy(), E[:12, 12:].copy()


def gen_flexsat_batch(batch, steps=45, N=80, seed=2, first_instance=0):
    """run_flexsat_mpc (:133-296) for a batch: every instance has the same plant, its own initial
    attitude error (the reference starts from MRP = 0.1 on each axis; instance 0 does too) and its
    own noise stream.  LQR to the origin = tracking an all-zero trajectory; |u| <= 0.01; cost dt 0.1.
    `noise` are unit normals: the plant adds 0.0002 * noise (:266), absolute.  Returns (batch, x0)."""
    A, Bm = flexsat_model()
    Nt = N + steps + 1
    x0 = np.zeros((batch, 12))
    noise = np.empty((steps, batch, 12))
    for b in range(batch):
        rng = instance_rng(seed, first_instance + b)
        x0[b, :3] = 0.1 if first_instance + b == 0 else 0.1 * (1.0 + 0.3 * rng.standard_normal(3))
        noise[:, b] = rng.standard_normal((steps, 12))
    pb = RandomLinearBatch(n=12, m=3, N=N, dt=0.1, A=np.tile(A, (batch, 1, 1)), Bm=np.tile(Bm, (batch, 1, 1)),
                           Xtrack=np.zeros((batch, Nt, 12)), Utrack=np.zeros((batch, Nt - 1, 3)), noise=noise,
                           u_bnd=0.01)
    return pb, x0


# ---------------------------------------------------------------------------------------------
# rocket landing (BASELINE config 3): reference benchmarks/rocket_landing/rocket_landing_problem.jl
BOX, LINEAR, SOC = 0, 1, 2
EQ, INEQ = 0, 1


def rocket_model(mass, grav, dt):
    """RocketModel (:17-40) with omega_planet = 0, discretised exactly (RD.Exponential):
    x = [r; v], x+ = A x + B u + f."""
    I3, Z3 = np.eye(3), np.zeros((3, 3))
    A = np.block([[I3, dt * I3], [Z3, I3]])
    Bm = np.vstack([0.5 * dt * dt / mass * I3, dt / mass * I3])
    g = np.asarray(grav, dtype=float)
    f = np.concatenate([0.5 * dt * dt * g, dt * g])
    return A, Bm, f


@dataclass
class ConstraintSpec:
    """One add_constraint! call as data: value = A z + b on knots k_first..k_last (0-based,
    inclusive); kind BOX / LINEAR / SOC, sense EQ / INEQ for LINEAR."""
    kind: int
    sense: int
    k_first: int
    k_last: int
    A: np.ndarray = None
    b: np.ndarray = None
    zmin: np.ndarray = None
    zmax: np.ndarray = None


@dataclass
class RocketProblemData:
    n: int
    m: int
    N: int
    dt: float
    A: np.ndarray
    Bm: np.ndarray
    f: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    Qf: np.ndarray
    xf: np.ndarray
    x0: np.ndarray
    U0: np.ndarray
    constraints: list


def gen_rocket_problem(N=101, tf=10.0, x0=(4.0, 2.0, 20.0, -3.0, 2.0, -5.0), Qk=1e-2, Qfk=100.0, Rk=1e-1,
                       gravity=(0.0, 0.0, -9.81), mass=10.0, perWeightMax=2.0, theta_thrust_max=7.0,
                       theta_glideslope=60.0, glide_recover_k=8, include_goal=True, include_thrust_angle=True,
                       include_glideslope=True):
    """RocketProblem (:44-186).  NormConstraint2's value [A y; c'y] carries zero rows for the
    components A does not select (:136-141, :155-164); they are dropped here (same cone)."""
    n, m = 6, 3
    dt = tf / (N - 1)
    A, Bm, f = rocket_model(mass, gravity, dt)
    cons = []
    nz = n + m
    if include_goal:  # GoalConstraint(xf) at N (:96)
        Ag = np.hstack([np.eye(n), np.zeros((n, m))])
        cons.append(ConstraintSpec(LINEAR, EQ, N - 1, N - 1, A=Ag, b=np.zeros(n)))
    u_bnd = mass * abs(gravity[2]) * perWeightMax  # :121
    At = np.zeros((4, nz))
    At[0, n + 0] = At[1, n + 1] = At[2, n + 2] = 1.0
    cons.append(ConstraintSpec(SOC, 0, 0, N - 2, A=At, b=np.array([0.0, 0.0, 0.0, u_bnd])))  # :123-124
    if include_thrust_angle:  # || (ux, uy) || <= tan(theta) uz   (:134-144)
        a = np.tan(np.deg2rad(theta_thrust_max))
        Aa = np.zeros((3, nz))
        Aa[0, n + 0] = Aa[1, n + 1] = 1.0
        Aa[2, n + 2] = a
        cons.append(ConstraintSpec(SOC, 0, 0, N - 2, A=Aa, b=np.zeros(3)))
    if include_glideslope:  # || (x, y) || <= tan(theta) z on knots glide_recover_k..N-1 (1-based) (:153-167)
        a = np.tan(np.deg2rad(theta_glideslope))
        Agl = np.zeros((3, nz))
        Agl[0, 0] = Agl[1, 1] = 1.0
        Agl[2, 2] = a
        cons.append(ConstraintSpec(SOC, 0, glide_recover_k - 1, N - 2, A=Agl, b=np.zeros(3)))
    U0 = np.tile(-mass * np.asarray(gravity, dtype=float), (N - 1, 1))  # hover (:181-183)
    return RocketProblemData(n=n, m=m, N=N, dt=dt, A=A, Bm=Bm, f=f, Q=np.full(n, Qk), R=np.full(m, Rk),
                             Qf=np.full(n, Qfk), xf=np.zeros(n), x0=np.asarray(x0, dtype=float), U0=U0,
                             constraints=cons)


# ---------------------------------------------------------------------------------------------
# grasp optimisation: reference benchmarks/grasp_optimization/src/{grasp_model,grasp_problem,utils}.jl
def _rot3(th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def _skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


@dataclass
class GraspProblemData:
    n: int
    m: int
    N: int
    dt: float
    A: np.ndarray
    Bm: np.ndarray
    f: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    Qf: np.ndarray
    xf: np.ndarray
    x0: np.ndarray
    U0: np.ndarray
    constraints: list   # ConstraintSpec with per-knot A (nk, p, nz) and b (nk, p)
    theta: np.ndarray
    p: list             # contact points p[i][k]
    v: list             # inward normals v[i][k]


def gen_grasp_problem(N=61, tf=6.0, x0=(0.0, 3.0, 3.0, 0.0, 0.0, 0.0), mu=0.5, mass=0.2, f_max=3.0,
                      theta0=0.0, thetaf=np.pi / 4, thetad0=0.0, thetadf=0.15):
    """GraspProblem (grasp_problem.jl:1-107) for the SquareObject of grasp_model.jl:4-92.
    Two fingers hold a rotating block: per-knot torque balance (equality), max normal force
    (inequality) and two friction cones ||(I - v v')F_i|| <= mu v'F_i (second-order cones)."""
    n = m = 6
    dt = tf / (N - 1)
    g = np.array([0.0, 0.0, -9.81])
    I3, Z3 = np.eye(3), np.zeros((3, 3))
    A = np.block([[I3, dt * I3], [Z3, I3]])                                   # grasp_model.jl:74-92
    Bm = np.vstack([0.5 * dt * dt / mass * np.hstack([I3, I3]), dt / mass * np.hstack([I3, I3])])
    f = np.concatenate([0.5 * dt * dt * g, dt * g])
    # cubic orientation trajectory (utils.jl:23-31, grasp_model.jl:33-41)
    t0 = 0.0
    M = np.array([[t0**3, t0**2, t0, 1], [tf**3, tf**2, tf, 1], [3 * t0**2, 2 * t0, 1, 0], [3 * tf**2, 2 * tf, 1, 0]])
    c = np.linalg.solve(M, np.array([theta0, thetaf, thetad0, thetadf]))
    ts = np.arange(N) * dt
    theta = c[0] * ts**3 + c[1] * ts**2 + c[2] * ts + c[3]
    thetadd = 6 * c[0] * ts + 2 * c[1]
    p10, v10 = np.array([0.0, -1.0, 0.0]), np.array([0.0, 1.0, 0.0])        # grasp_model.jl:44-49 (last assignments)
    p20, v20 = np.array([0.0, 1.0, 0.0]), np.array([0.0, -1.0, 0.0])
    p = [[_rot3(th) @ p10 for th in theta], [_rot3(th) @ p20 for th in theta]]
    v = [[_rot3(th) @ v10 for th in theta], [_rot3(th) @ v20 for th in theta]]
    nz = n + m
    nk = N - 1
    At = np.zeros((nk, 3, nz)); bt = np.zeros((nk, 3))
    Ag = np.zeros((nk, 2, nz)); bg = np.full((nk, 2), -f_max)
    Af = [np.zeros((nk, 4, nz)), np.zeros((nk, 4, nz))]
    for k in range(nk):
        At[k, :, n:n + 3] = _skew(p[0][k]); At[k, :, n + 3:] = _skew(p[1][k])   # torque balance (:35-38)
        bt[k] = -np.array([thetadd[k], 0.0, 0.0])
        Ag[k, 0, n:n + 3] = v[0][k]; Ag[k, 1, n + 3:] = v[1][k]                 # max normal force (:41-49)
        for i in range(2):                                                       # friction cones (:52-67)
            vv = v[i][k]
            Af[i][k, :3, n + 3 * i:n + 3 * i + 3] = np.eye(3) - np.outer(vv, vv)
            Af[i][k, 3, n + 3 * i:n + 3 * i + 3] = mu * vv
    xf = np.zeros(n)
    cons = [
        ConstraintSpec(LINEAR, EQ, N - 1, N - 1, A=np.hstack([np.eye(n), np.zeros((n, m))]), b=-xf),   # goal (:29-30)
        ConstraintSpec(LINEAR, EQ, 0, N - 2, A=At, b=bt),
        ConstraintSpec(LINEAR, INEQ, 0, N - 2, A=Ag, b=bg),
        ConstraintSpec(SOC, 0, 0, N - 2, A=Af[0], b=np.zeros((nk, 4))),
        ConstraintSpec(SOC, 0, 0, N - 2, A=Af[1], b=np.zeros((nk, 4))),
    ]
    u0 = np.array([0.0, -1.5, mass * 9.81 / 2, 0.0, 1.5, mass * 9.81 / 2])      # :101-104
    return GraspProblemData(n=n, m=m, N=N, dt=dt, A=A, Bm=Bm, f=f, Q=np.full(n, 1e-3), R=np.full(m, 1.0),
                            Qf=np.full(n, 10.0), xf=xf, x0=np.asarray(x0, dtype=float),
                            U0=np.tile(u0, (N - 1, 1)), constraints=cons, theta=theta, p=p, v=v)


# ---------------------------------------------------------------------------------------------
# quadruped contact-switching MPC (BASELINE config 5): reference benchmarks/quadruped/Woofer/MPCControl
QUADRUPED_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-3, constraint_tolerance=1e-4,
                      penalty_initial=10.0, penalty_scaling=100.0, reset_duals=0)
"""SolverOptions of Structs/ALTROParams.jl:86-95"""


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _mrp_rotation(p):
    """rotation matrix of a Modified Rodrigues Parameter vector"""
    s = float(p @ p)
    P = _skew(p)
    return np.eye(3) + (8.0 * P @ P + 4.0 * (1.0 - s) * P) / (1.0 + s) ** 2


def _mrp_kinematics(p, w):
    s = float(p @ p)
    return 0.25 * ((1.0 - s) * np.eye(3) + 2.0 * _skew(p) + 2.0 * np.outer(p, p)) @ w


def quadruped_continuous_dynamics(x, u, feet, contacts, inertia, mass):
    """Single rigid body with four point feet (NonLinearContinuousDynamics,
    linearized_dynamics.jl:1-38): x = [position; MRP attitude; velocity; body angular velocity],
    u = four world-frame foot forces; a swing foot (contact 0) exerts nothing."""
    p, phi, v, w = x[0:3], x[3:6], x[6:9], x[9:12]
    R = _mrp_rotation(phi)
    force = np.array([0.0, 0.0, -9.81])
    torque = np.zeros(3)
    for i in range(4):
        fi = u[3 * i:3 * i + 3]
        force = force + contacts[i] * fi / mass
        rb = R.T @ (feet[i] - p)                      # foot in the body frame
        torque = torque + contacts[i] * _skew(rb) @ (R.T @ fi)
    wdot = np.linalg.solve(inertia, -_skew(w) @ inertia @ w + torque)
    return np.concatenate([v, _mrp_kinematics(phi, w), force, wdot])


def quadruped_linearize(xr, ur, feet, contacts, inertia, mass, dt, eps=1e-6):
    """A_k, B_k, d_k of update_dynamics_matrices! (altro_solver.jl:5-37): Jacobians of the
    continuous dynamics at (x_ref, u_ref) (central differences here, ForwardDiff there), affine
    remainder d = f(x_ref,u_ref) - A x_ref - B u_ref, forward-Euler discretisation.  (The reference's
    rollouts use a shortcut of this model, linearized_dynamics.jl:69-96; the solvers here roll out the
    full affine model.)"""
    f0 = quadruped_continuous_dynamics(xr, ur, feet, contacts, inertia, mass)
    Ac = np.zeros((12, 12))
    Bc = np.zeros((12, 12))
    for j in range(12):
        e = np.zeros(12)
        e[j] = eps
        Ac[:, j] = (quadruped_continuous_dynamics(xr + e, ur, feet, contacts, inertia, mass) -
                    quadruped_continuous_dynamics(xr - e, ur, feet, contacts, inertia, mass)) / (2 * eps)
        Bc[:, j] = (quadruped_continuous_dynamics(xr, ur + e, feet, contacts, inertia, mass) -
                    quadruped_continuous_dynamics(xr, ur - e, feet, contacts, inertia, mass)) / (2 * eps)
    dc = f0 - Ac @ xr - Bc @ ur
    return np.eye(12) + Ac * dt, Bc * dt, dc * dt


def trot_contacts(t, stance_time=0.2, swing_time=0.2):
    """trot(): four phases [all feet | diagonal pair A | all feet | diagonal pair B]
    (Structs/GaitParams.jl:38-49, MPC.yaml:2-5)"""
    phases = np.array([[1, 1, 1, 1], [1, 0, 0, 1], [1, 1, 1, 1], [0, 1, 1, 0]], dtype=float)
    times = np.array([stance_time, swing_time, stance_time, swing_time])
    tt = t % times.sum()
    k = int(np.searchsorted(np.cumsum(times), tt, side="right"))
    return phases[min(k, 3)]


@dataclass
class QuadrupedData:
    n: int
    m: int
    N: int
    dt: float
    Q: np.ndarray
    R: np.ndarray
    x_des: np.ndarray
    u_hover: np.ndarray
    feet: np.ndarray          # (4, 3) world-frame foot positions of the nominal stance
    inertia: np.ndarray
    mass: float
    mu: float
    fz_max: float
    constraints: list         # ConstraintSpec: 4 friction pyramids (LINEAR ineq) + the f_z box

    def dynamics(self, t0, x_ref=None):
        """per-knot (A, B, d) for a horizon starting at time t0 (contact schedule of the trot)"""
        xr = self.x_des if x_ref is None else x_ref
        A = np.zeros((self.N - 1, 12, 12))
        Bm = np.zeros((self.N - 1, 12, 12))
        d = np.zeros((self.N - 1, 12))
        for k in range(self.N - 1):
            c = trot_contacts(t0 + k * self.dt)
            A[k], Bm[k], d[k] = quadruped_linearize(xr, np.zeros(12), self.feet, c, self.inertia, self.mass, self.dt)
        return A, Bm, d


def gen_quadruped_problem(N=15, dt=0.03, vx=0.0, linearized_friction=True):
    """AltroParams (Structs/ALTROParams.jl:32-108) with MPC.yaml's weights: n = m = 12, LQR about
    x_des with u_ref = 0, per leg the linearised friction pyramid |f_x|, |f_y| <= mu f_z (4 rows,
    LinearizedFrictionConstraint.jl:14-25) on knots 1..N-1, 0 <= f_z <= 133 on every control.
    linearized_friction=False gives the cone variant instead (FrictionConstraint.jl:1-8: NormConstraint2
    with A = diag(1,1,0), c = mu e_z, i.e. (f_x, f_y, 0, mu f_z) in the second-order cone)."""
    n = m = 12
    q = np.array([1.0, 1.0, 500.0, 5000.0, 5000.0, 1000.0, 500.0, 1000.0, 1000.0, 500.0, 500.0, 100.0])
    r = np.tile([1.0, 1.0, 0.001], 4)
    mass = 3.0 + 4 * 1.033 + 8 * 0.070                       # sprung mass (Config.jl:56)
    inertia = np.diag([0.025, 0.854, 0.897])
    hx, hy, abd, h = 0.230, 0.109, 0.064, 0.28
    feet = np.array([[hx, hy + abd, 0.0], [hx, -hy - abd, 0.0], [-hx, hy + abd, 0.0], [-hx, -hy - abd, 0.0]])
    x_des = np.zeros(12)
    x_des[2] = h
    x_des[6] = vx
    mu, fz_max = 0.5, 133.0
    cons = []
    for leg in range(4):
        A = np.zeros((4, n + m))
        ix, iy, iz = n + 3 * leg, n + 3 * leg + 1, n + 3 * leg + 2
        if not linearized_friction:
            A[0, ix], A[1, iy], A[3, iz] = 1.0, 1.0, mu
            cons.append(ConstraintSpec(SOC, 0, 0, N - 2, A=A, b=np.zeros(4)))
            continue
        A[0, ix], A[0, iz] = 1.0, -mu
        A[1, ix], A[1, iz] = -1.0, -mu
        A[2, iy], A[2, iz] = 1.0, -mu
        A[3, iy], A[3, iz] = -1.0, -mu
        cons.append(ConstraintSpec(LINEAR, INEQ, 0, N - 2, A=A, b=np.zeros(4)))
    zmin = np.full(n + m, -np.inf)
    zmax = np.full(n + m, np.inf)
    for leg in range(4):
        zmin[n + 3 * leg + 2], zmax[n + 3 * leg + 2] = 0.0, fz_max
    cons.append(ConstraintSpec(BOX, INEQ, 0, N - 1, zmin=zmin, zmax=zmax))
    u_hover = np.tile([0.0, 0.0, 9.81 * mass / 4.0], 4)
    return QuadrupedData(n=n, m=m, N=N, dt=dt, Q=q, R=r, x_des=x_des, u_hover=u_hover, feet=feet, inertia=inertia,
                         mass=mass, mu=mu, fz_max=fz_max, constraints=cons)


@dataclass
class QuadrupedBatch:
    """A batch of quadruped MPC loops (BASELINE configs[4]): per instance a gait phase, an initial state, the per-knot
    affine dynamics of every tick of the loop and the plant noise.  Instance i draws from its own stream keyed by
    (seed, first_instance + i), so the shards of a multi-rank run are slices of one global batch."""
    qp: QuadrupedData
    t0: np.ndarray       # (B,) gait phase at tick 0
    x0: np.ndarray       # (B, 12)
    A: np.ndarray        # (B, T, 12, 12) blocks of absolute knots 0..T-1 (tick r's window reads r .. r + N - 2)
    Bm: np.ndarray       # (B, T, 12, 12)
    d: np.ndarray        # (B, T, 12)
    noise: np.ndarray    # (steps, B, 12) unit normals
    first_instance: int


def gen_quadruped_batch(batch, N=40, steps=20, seed=17, first_instance=0, linearized_friction=True):
    """altro_solver.jl:44-88 driven as a closed loop: trot gait at a random phase, start near the stance pose."""
    qp = gen_quadruped_problem(N=N, linearized_friction=linearized_friction)
    T = steps + N
    sx = np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
    t0 = np.zeros(batch)
    x0 = np.zeros((batch, 12))
    noise = np.zeros((steps, batch, 12))
    A, Bm, d = np.zeros((batch, T, 12, 12)), np.zeros((batch, T, 12, 12)), np.zeros((batch, T, 12))
    cache = {}
    for b in range(batch):
        rng = instance_rng(seed, first_instance + b)
        t0[b] = rng.uniform(0.0, 0.8)
        x0[b] = qp.x_des + rng.standard_normal(12) * sx
        noise[:, b] = rng.standard_normal((steps, 12))
        for t in range(T):
            c = tuple(trot_contacts(t0[b] + t * qp.dt))
            if c not in cache:
                cache[c] = quadruped_linearize(qp.x_des, np.zeros(12), qp.feet, np.array(c), qp.inertia, qp.mass, qp.dt)
            A[b, t], Bm[b, t], d[b, t] = cache[c]
    return QuadrupedBatch(qp=qp, t0=t0, x0=x0, A=A, Bm=Bm, d=d, noise=noise, first_instance=first_instance)
