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
