#!/usr/bin/env python3
"""Headline benchmark: MPC solves/sec, batched AL-iLQR run to tolerance, random_linear_mpc
n=12 m=4 N=50 (BASELINE.json), batch 8192 per GPU.

A "step" is one MPC step of the whole batch in the reference's order (plant step + 1 % noise,
retarget the tracking cost, primal and dual shift_fill, warm-started solve;
random_linear_problem.jl:121-161).  Every input of the timed region (long reference
trajectories, noise samples, dynamics) is resident in HBM before the clock starts.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; instances are
sharded over ranks (weak scaling: 8192 per GPU), there is no data-path collective, and the
only RCCL traffic is one all_gather of the first controls after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

N_STATE, N_CTRL, N_KNOT = 12, 4, 50
BATCH_PER_GPU = 8192
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector (= matrix) peak, datasheet; see DESIGN.md
HBM_PEAK_GBS = 8000.0


def flops_backward(n, m, N):
    """SURVEY.md 8(d)"""
    return (N - 1) * (4 * n**3 + 8 * n**2 * m + 6 * n * m**2 + m**3 / 3 + 2 * n**2 + 8 * n * m + 4 * m**2)


def flops_forward(n, m, N):
    return (N - 1) * (2 * n**2 + 4 * n * m + 2 * (n + m)) + N * 3 * (n + m)


def flops_costate(n, m, N):
    """first-order costate sweep of a confirmed iteration (include/altro_batch.h, strict): per knot
    lambda_k = l_x + A' lambda_{k+1}, g_k = l_u + B' lambda_{k+1} and the box terms"""
    return (N - 1) * (2 * n * (n + m) + 8 * (n + m))


def bytes_solve(n, m, N, p):
    return 8 * ((n * n + n * m) + n + (N * n + (N - 1) * m) + (N - 1) * m + 2 * (N - 1) * p
                + (N * n + (N - 1) * m) + 2 * (N - 1) * p + 12)


def measured_traffic(batch, steps, warmup, spl, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command line; FETCH_SIZE doubled per MI355X_MICROARCH.md and tools/probes/fetch_calib.hip).
    Only returned when a committed measurement exists for exactly this configuration: the file holds
    one entry per profiled command line (the driver's `--steps 20 --warmup 5` and the default)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in (t.get("entries") or [t]):
            if (e.get("batch") == batch and e.get("steps") == steps and e.get("steps_per_launch") == spl and
                    e.get("warmup", warmup) == warmup and world == 1):
                best = e["hbm_bytes_per_launch"]      # later rounds override earlier ones
    return best


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_worker(args):
    """Run whole MPC loops of the oracle for ~budget seconds on one core."""
    first, count, steps, budget = args
    import altro_amd_loader  # noqa: F401
    import altro_mpc_icra2021_amd as altro
    import oracle_py
    from helpers import make_oracle, mpc_update
    pb = altro.problems.gen_random_linear_batch(count, n=N_STATE, m=N_CTRL, N=N_KNOT, steps=steps, seed=1,
                                                first_instance=first)
    solves, t_solve = 0, 0.0
    t_end = time.perf_counter() + budget
    for b in range(count):
        o = make_oracle(oracle_py, pb, b)
        o.solve()
        for i in range(steps):
            mpc_update(o, pb, b, i)
            t0 = time.perf_counter()
            o.solve()
            t_solve += time.perf_counter() - t0
            solves += 1
        if time.perf_counter() > t_end:
            break
    return solves, t_solve


PUBLISHED_1THREAD_MS = 0.868
"""the reference's published median of one warm ALTRO MPC solve, random-linear n=12 m=6 N=51, one thread of
an unnamed CPU (figures/horizon_comp.tikz:11; BASELINE.md): the sanity anchor for the 1-thread figure below"""


def cpu_baseline(budget_s=12.0):
    """CPU restatement (the oracle, kind 'port') timed on this host's cores, same workload,
    same options, one warm-started solve per MPC step (SURVEY 6.2 caveat).  One process per core
    this job may use (affinity mask and cgroup quota; the count is reported).  Run BEFORE the GPU
    is touched (process pool forks)."""
    import multiprocessing as mp
    cores = host_cores()
    steps = 40
    per = 1500  # instances offered to each worker; it stops when the time budget is spent
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(100000 + w * per, per, steps, budget_s) for w in range(cores)])
    wall = time.perf_counter() - t0
    solves = sum(r[0] for r in res)
    t_solve = sum(r[1] for r in res)
    return {
        "value": solves / (t_solve / cores),          # solves/s with all `cores` busy (solve time only)
        "unit": "solves/s",
        "cores": cores,
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "per_core": solves / t_solve,
        "ms_per_solve_1core": 1e3 * t_solve / solves,
        "published_ms_per_solve_1thread": PUBLISHED_1THREAD_MS,
        "published_note": "reference, n=12 m=6 N=51, unnamed CPU, Julia 1.4 (figures/horizon_comp.tikz:11); "
                          "timed there over repeated solves from converged duals (benchmark_solve!)",
        "sample": f"{solves} warm-started MPC solves (random_linear n=12 m=4 N=50, same options), "
                  f"{cores} processes x whole MPC loops of {steps} steps, {wall:.1f} s wall",
    }


def flops_first_order(n, m, N):
    """first-order sweep of an iteration that takes its gains from memory (include/altro_batch.h,
    altro_batch_get_reuse_counter): per knot [Qx; Qu] = l_z + [A B]' s, d = -(L D L')^-1 Qu, s = Qx + K' Qu, dV"""
    return (N - 1) * (2 * n * (n + m) + 2 * n * m + 2 * m * m + 6 * m)


def secondary_traffic(key, steps=None, whole=False):
    """HBM bytes of the timed launch of a secondary line from the committed rocprofv3 PMC passes
    (profiles/rNN_secondary_kernels.json, tools/profile_secondary.sh), or None.  key: (config name, kernel substring
    [, {launch configuration}: the sweep points of one kernel differ in grid and block size]);
    only a profile of a launch with the same number of fused steps counts."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_secondary_kernels.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        ent = t.get(key[0]) or {}
        if steps is not None and not any(bl.get("steps") == steps for bl in ent.get("bench_lines", [])):
            continue
        for rec in ent.get("launches", []):
            if key[1] in rec.get("kernel", "") and "hbm_bytes" in rec and all(rec.get(k) == v for k, v in (key[2] if len(key) > 2 else {}).items()):
                best = rec if whole else rec["hbm_bytes"]
    return best


def mfma_fraction(rec):
    """FP64 MFMA utilisation of a profiled launch: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop / launch time / FP64 matrix peak
    (the counter and the launch time both come from the committed rocprofv3 passes of the same command line)"""
    if not rec or "SQ_INSTS_VALU_MFMA_MOPS_F64" not in rec or not rec.get("timed_launch_ms"):
        return None
    return rec["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / (rec["timed_launch_ms"] * 1e-3) / (FP64_PEAK_TFLOPS * 1e12)


class _Solo:
    """the rank group of a single process (no torch): what parallel.RankGroup does with world = 1"""
    rank, world = 0, 1

    def barrier(self):
        pass

    def max_over_ranks(self, x):
        return float(x)

    def sum_over_ranks(self, v):
        return int(v)

    def gather(self, U1, status):
        return np.asarray(U1), np.asarray(status)

    def timed(self, fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def gather_checked(self, U1, status):
        return np.asarray(U1), np.asarray(status)


def _secondary_line(name, workload, kernel, bound, n, m, N, B, K, W, mp, altro, extra=None, traffic_key=None, grp=None):
    """Run W warm-up + K timed MPC steps (one fused launch) of a secondary BASELINE config on every rank's shard and
    build its JSON line: same metric and bracket as the headline (barrier + synchronise on both sides, MAX over ranks,
    whole-job solves / that time), roofline from HIP events on rank 0's stream and its measured pass counts."""
    grp = grp or _Solo()
    for i in range(W):
        mp.step(i)
    altro.timing_reset(mp.solver)          # synchronises the library's stream
    def run():
        mp.run_async(K, first=W)
        mp.synchronize()

    dt = grp.timed(run)
    # the results an MPC consumer reads each tick: the only collective, after the timed region
    allU, allS = grp.gather_checked(altro.controls(mp.solver)[:, 0].copy(), altro.stats(mp.solver).status)
    assert allU.shape == (grp.world * B, m), allU.shape
    ms = altro.timing_get(mp.solver)
    nb, nr, ntr = altro.work_counters(mp.solver)
    nsol, nit, nok = altro.solve_counters(mp.solver)
    ngc = altro.confirm_counter(mp.solver)
    nfo = altro.reuse_counter(mp.solver)
    assert len(ms) == 1, len(ms)
    # roofline.achieved = SURVEY 8(d) flops_solve with the measured iteration and trial counts (as on the headline line);
    # roofline.executed = the passes the kernel actually ran: every iteration is a backward pass, a first-order sweep
    # with the gains in memory, or a costate-sweep confirmation (iterations = nb + nfo + ngc)
    flops_exec = (nb.sum() * flops_backward(n, m, N) + nr.sum() * flops_forward(n, m, N) + ngc.sum() * flops_costate(n, m, N) +
                  nfo.sum() * flops_first_order(n, m, N))
    flops = nit.sum() * flops_backward(n, m, N) + (nit.sum() + ntr.sum() + B * K) * flops_forward(n, m, N)
    avg_ms = float(ms.mean())
    achieved = flops / len(ms) / (avg_ms * 1e-3) / 1e12
    executed = flops_exec / len(ms) / (avg_ms * 1e-3) / 1e12
    out = {"metric": "MPC solves/sec (batched iLQR to tol), " + name, "value": grp.world * B * K / dt, "unit": "solves/s", "n_gpus": grp.world,
           "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": workload, "batch_per_gpu": B, "global_batch": grp.world * B,
                                                          "parallelism": "instances sharded over %d GPU(s), no data-path collective" % grp.world},
           "roofline": {"bound": bound, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                        "traffic": secondary_traffic(traffic_key, K) if traffic_key else None,
                        "mfma_frac": mfma_fraction(secondary_traffic(traffic_key, K, whole=True)) if traffic_key else None,
                        "kernel": kernel, "avg_launch_ms": avg_ms,
                        "launches": int(len(ms)),
                        "executed": {"achieved": executed, "frac": executed / FP64_PEAK_TFLOPS},
                        "note": "achieved: SURVEY 8(d) flops_solve (measured iterations and trials x the base Riccati / rollout formulas; "
                                "constraint-expansion flops are not counted) -- an algorithmic rate, NOT hardware utilisation: iterations "
                                "that take their gains from memory or are confirmed by the costate sweep execute no backward pass; "
                                "executed: the passes the kernel ran; traffic: HBM bytes of this launch from the committed PMC passes or null; "
                                "mfma_frac: FP64 MFMA utilisation of the same profiled launch (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 / time / peak) or null"},
           "cpu_baseline": None, "solve_succeeded_frac": float(nok.sum() / max(1, nsol.sum())),
           "iterations_mean": float(nit.sum() / max(1, nsol.sum())), "backward_passes_per_solve": float(nb.sum() / (B * K)),
           "gains_from_memory_iterations_per_solve": float(nfo.sum() / (B * K)),
           "costate_confirmed_iterations_per_solve": float(ngc.sum() / (B * K)),
           "rollouts_per_solve": float(nr.sum() / (B * K))}
    if extra:
        out.update(extra)
    return out


def _next_window(mp, altro, K, first, B):
    """The K steps after a line's own, same options: a launch lasts as long as its slowest instance, and single solves of a
    few hundred iterations (0.05 % of them, the oracle walks the same counts) fall into some windows and not into others"""
    altro.timing_reset(mp.solver)
    t0 = time.perf_counter()
    mp.run_async(K, first=first)
    mp.synchronize()
    dt = time.perf_counter() - t0
    nsol, nit, nok = altro.solve_counters(mp.solver)
    return {"value": B * K / dt, "unit": "solves/s", "steps": K, "ms_per_step": 1e3 * dt / K, "solve_succeeded_frac": float(nok.sum() / max(1, nsol.sum())),
            "iterations_mean": float(nit.sum() / max(1, nsol.sum())), "iterations_max_per_instance": int(nit.max()),
            "note": "the next K steps of the same closed loops, same options"}


def _capped(mp, altro, cap, K, W, B):
    """The same closed loops once more with Altro's `iterations` option (the total iLQR iterations one solve may take,
    default 1000) set to `cap`, as an MPC deployment with a tick deadline would: a solve that hits it reports
    MAX_ITERATIONS and the loop goes on from the trajectory it holds.  Without it a launch lasts as long as its slowest
    instance (single solves of several hundred iterations at the penalty cap: tools/gpu_rocket_tail.py)."""
    altro.set_options(mp.solver, iterations=cap)
    altro.timing_reset(mp.solver)
    t0 = time.perf_counter()
    mp.run_async(K, first=mp.i)
    mp.synchronize()
    dt = time.perf_counter() - t0
    nsol, nit, nok = altro.solve_counters(mp.solver)
    return {"iterations_option": cap, "value": B * K / dt, "unit": "solves/s", "steps": K, "ms_per_step": 1e3 * dt / K,
            "solve_succeeded_frac": float(nok.sum() / max(1, nsol.sum())), "iterations_mean": float(nit.sum() / max(1, nsol.sum())),
            "note": "the K steps after the line's own, Altro option iterations = %d; not the reference's configuration (its scripts leave the default 1000)" % cap}


def secondary_configs(which, K, W, emit=None, state_dims=(8, 16, 32, 48, 64), grp=None, device=0):
    """BASELINE configs[2..4] at their per-GPU sizes.  `--config <name>` prints them as lines of their own (under
    torch.distributed.run every rank owns a shard of the config's instances: configs[3] "65536 sharded over 8" and
    configs[4] "16384 over 8" are 8192 and 2048 instances per GPU); the default run carries short single-GPU versions
    (<= 10 steps) inside the headline line's `secondary` list."""
    import altro_amd_loader  # noqa: F401
    import altro_mpc_icra2021_amd as altro
    P, api, mpcm = altro.problems, altro, altro.mpc
    grp = grp or _Solo()
    shard0 = altro.parallel.shard_first_instance    # global index of this rank's first instance
    lines = []

    def done(d):
        lines.append(d)
        if emit and grp.rank == 0:
            emit(d)

    if which in ("rocket", "all"):   # configs[2]: rocket landing, second-order cones, N_mpc = 100, batch 4096
        B, Nm, Nt, dt = 4096, 100, 301, 0.05
        K2 = min(K, (Nt - Nm - 1 - W) // 2)
        rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
        f0 = shard0(grp.rank, B)
        rngs = [P.instance_rng(1, f0 + b) for b in range(B)]     # one stream per global instance: shards = slices
        x0 = np.tile(rp.x0, (B, 1)) + np.stack([r.standard_normal(6) for r in rngs]) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
        nz_rocket = np.stack([r.standard_normal((W + 2 * K2, 6)) for r in rngs], axis=1)
        cold = api.ALTROSolver(mpcm.constrained_problem(rp, x0), api.SolverOptions(**altro.benchmarks.ROCKET_COLD_OPTS), device)
        api.solve(cold)
        Xt, Ut = api.states(cold), api.controls(cold)
        cold.close()
        tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
        tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
        prob = mpcm.constrained_problem(tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy())
        mp = mpcm.TrackMPC(prob, api.SolverOptions(**altro.benchmarks.ROCKET_MPC_OPTS), Xt, Ut, nz_rocket,
                           (np.array([1e-3] * 3 + [1e-2] * 3), np.array([0, 0, 0, 1, 1, 1])), device=device)
        mp.initial_solve()
        line = _secondary_line("rocket_landing (SOC thrust cone) N=100", "rocket_landing N_mpc=100 batch=4096 on 1 GPU (BASELINE configs[2])",
                               "altro::solve_kernel<6,3,true>", "valu_fp64", 6, 3, Nm, B, K2, W, mp, altro, traffic_key=("rocket", "solve_kernel"), grp=grp)
        if grp.world == 1:
            line["with_iteration_cap"] = _capped(mp, altro, 100, K2, W, B)
        done(line)
        mp.solver.close()
    if which in ("state_dim", "all"):   # configs[3]: state-dimension sweep, m = 4, N = 50, 8192 instances per GPU
        for n in state_dims:
            B = 8192                      # configs[3]: 65536 instances over 8 GPUs
            K3 = min(K, 10 if n <= 16 else 5)
            pb = P.gen_random_linear_batch(B, n=n, m=4, N=50, steps=K3 + W, seed=10, first_instance=shard0(grp.rank, B))
            mp = mpcm.BatchMPC(pb, device=device)
            mp.initial_solve()
            kern = "altro::solve_kernel<8,4>" if n == 8 else "altro_wide::wide_kernel<4>"
            # launch configuration of the point in the committed profile: one wave per four instances (n = 8), per instance
            # (n <= 48) or a cooperative block of four waves per instance (wide_block_threads)
            wg = 256 if n > 48 else 64
            # (every sweep point has an instantiation of its own since the padded n became a template argument)
            tk = ("state_dim", "solve_kernel<8, 4" if n == 8 else "wide_kernel<4, true" if n <= 16 else "wide_kernel<4, false, %d" % ((n + 15) // 16 * 16),
                  {"grid_size": (B // 4 * 64) if n == 8 else B * wg, "workgroup_size": wg, "point": 0})
            done(_secondary_line("random_linear_mpc n=%d m=4 N=50" % n, "state_dim sweep point n=%d m=4 N=50 batch=%d on 1 GPU (BASELINE configs[3])" % (n, B),
                                 kern, "valu_fp64" if n == 8 else "mfma", n, 4, 50, B, K3, W, mp, altro, traffic_key=tk, grp=grp))
            mp.solver.close()
    if which in ("quadruped", "all"):   # configs[4]: quadruped contact-switching MPC, N = 40, 2048 instances per GPU, LTV loop on device
        B, N = 2048, 40                   # configs[4]: 16384 instances over 8 GPUs
        K4 = min(K, 20)
        qb = P.gen_quadruped_batch(B, N=N, steps=W + 3 * K4, seed=17, first_instance=shard0(grp.rank, B))
        qp, x0, A, Bm, d = qb.qp, qb.x0, qb.A, qb.Bm, qb.d
        Nt = W + 3 * K4 + N + 1
        prob = mpcm.quadruped_problem(qp, x0, A[:, :N - 1], Bm[:, :N - 1], d[:, :N - 1])
        mp = mpcm.TrackMPC(prob, api.SolverOptions(**P.QUADRUPED_OPTS), np.tile(qp.x_des, (B, Nt, 1)), np.zeros((B, Nt - 1, 12)),
                           qb.noise, (np.full(12, 1e-3),), device=device)
        api.set_dynamics_track(mp.solver, A, Bm, d, step_stride=1)
        api.initial_controls(mp.solver, np.tile(qp.u_hover, (B, N - 1, 1)))
        mp.initial_solve()
        line = _secondary_line("quadruped contact-switching MPC N=40", "quadruped N=40 batch=2048 on 1 GPU, per-knot dynamics resident on the device (BASELINE configs[4])",
                               "altro_wide::wide_kernel<12>", "valu_fp64+mfma", 12, 12, N, B, K4, W, mp, altro, traffic_key=("quadruped", "wide_kernel"), grp=grp)
        if grp.world == 1:
            line["next_window"] = _next_window(mp, altro, K4, W + K4, B)
            line["with_iteration_cap"] = _capped(mp, altro, 50, K4, W, B)
        done(line)
        mp.solver.close()
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (the profiling scripts use this)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="K-step regions timed back to back; `value` is the FIRST (W warm-up steps, then exactly K steps), the others are reported beside it")
    ap.add_argument("--config", default="headline", choices=["headline", "rocket", "state_dim", "quadruped", "all"],
                    help="headline (BASELINE configs[1], the driver's line) or a secondary config: extra JSON lines, 1 GPU only")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed GPU activity right before the timed region (elementwise torch kernels on a scratch buffer): the "
                         "W warm-up steps are single launches with host work between them and leave the clocks wherever the idle "
                         "GPU had them; 0 = none")
    ap.add_argument("--steps-per-launch", type=int, default=0,
                    help="MPC steps per kernel launch in the timed region (0 = all K in one launch)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))

    if a.config != "headline":
        grp = None
        if world > 1:   # one rank per GPU, each owns a shard of the config's instances (same bracket as the headline)
            import altro_amd_loader  # noqa: F401
            import altro_mpc_icra2021_amd as altro
            grp = altro.parallel.RankGroup("nccl")
        secondary_configs(a.config, a.steps, a.warmup, emit=lambda d: print(json.dumps(d), flush=True), grp=grp, device=local_rank)
        if grp is not None:
            grp.close()
        return

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch
    import altro_amd_loader  # noqa: F401
    import altro_mpc_icra2021_amd as altro

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the solver has no CPU path")
    # process-group plumbing of the timed region (tests/test_parallel_gloo.py runs the same class over gloo)
    grp = altro.parallel.RankGroup("nccl")
    assert (grp.rank, grp.world) == (rank, world)

    B, K, W, R = a.batch, a.steps, a.warmup, max(1, a.repeats)
    pb = altro.problems.gen_random_linear_batch(B, n=N_STATE, m=N_CTRL, N=N_KNOT, steps=W + K * R, seed=1,
                                                first_instance=altro.parallel.shard_first_instance(rank, B))
    mp = altro.mpc.BatchMPC(pb, device=local_rank)
    mp.initial_solve()
    for i in range(W):
        mp.step(i)
    altro.timing_reset(mp.solver)

    spl = a.steps_per_launch if a.steps_per_launch > 0 else K
    launches_per_region = (K + spl - 1) // spl
    assert launches_per_region * R <= 1024, "more launches than the library's timing ring holds (launch_ring.h CAP)"

    def region(first):
        """exactly K MPC steps of every instance, bracketed by barrier + synchronize on both sides; max over ranks"""
        def run():
            i = first
            while i < first + K:
                n = min(spl, first + K - i)
                mp.run_async(n, first=i)   # n consecutive MPC steps of every instance in one launch
                i += n
            mp.synchronize()
        return grp.timed(run)

    def preheat(ms_):
        """keep the GPU busy for ms_ milliseconds with work that touches nothing of the solver"""
        if ms_ <= 0:
            return
        dev = torch.device("cuda", local_rank)
        x = torch.ones(1 << 24, device=dev, dtype=torch.float64)
        t_end = time.perf_counter() + ms_ * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(50):
                x.mul_(1.0000001)
            torch.cuda.synchronize(dev)
        del x

    preheat(a.preheat_ms)
    dt = region(W)                     # THE timed region: W warm-up steps have run, now exactly K steps

    st = altro.stats(mp.solver)
    ms = altro.timing_get(mp.solver)
    assert len(ms) == launches_per_region, (len(ms), launches_per_region)
    nb, nr, ntr = altro.work_counters(mp.solver)
    nsol, nit, nok = altro.solve_counters(mp.solver)
    ngc = altro.confirm_counter(mp.solver)
    nfo = altro.reuse_counter(mp.solver)
    assert int(nsol.sum()) == B * K, (int(nsol.sum()), B * K)
    assert int(nb.sum() + nfo.sum() + ngc.sum()) == int(nit.sum())    # every iteration is exactly one of the three kinds
    ok = int(nok.sum())

    # final gather of the first controls + status (what an MPC consumer reads each tick): the only
    # collective of the run, after the timed region
    U1 = altro.controls(mp.solver)[:, 0].copy()
    allU, allS = grp.gather_checked(U1, st.status)   # asserts that rank r's shard sits at rows [r B, (r + 1) B)
    assert allU.shape == (world * B, N_CTRL)
    ok = grp.sum_over_ranks(ok)

    # the same K-step region R - 1 more times, back to back (later steps of the same closed loops): the spread of the number
    rep_wall = [dt]
    for r in range(1, R):
        rep_wall.append(region(W + r * K))
    ms_all = altro.timing_get(mp.solver)
    assert len(ms_all) == launches_per_region * R, (len(ms_all), launches_per_region, R)
    rep_kernel = [float(ms_all[r * launches_per_region:(r + 1) * launches_per_region].sum()) for r in range(R)]

    secondary = None
    if rank == 0 and world == 1 and not a.no_secondary:
        mp.solver.close()
        secondary = secondary_configs("all", min(K, 10), min(W, 3), state_dims=(8, 16, 32, 48, 64), device=local_rank)

    if rank == 0:
        n, m, N = N_STATE, N_CTRL, N_KNOT
        solves = world * B * K
        value = solves / dt
        # roofline of the dominant kernel (solve_kernel).  `achieved` follows SURVEY 8(d) to the letter ("the single
        # source for builder and judge"): flops_solve = sum over the MEASURED inner iterations of [flops_backward +
        # trials x flops_forward] + flops_forward(initial rollout), summed over the solves of one launch -- the
        # arithmetic of the reference's algorithm for the iteration counts the kernel reports (they equal the
        # oracle's, tests/).  In the default mode most of those iterations do NOT execute a backward pass (gains taken
        # from memory, or the costate sweep confirms convergence); what the kernel actually executed is reported
        # next to it as roofline.executed (measured pass counts x the formulas of each pass), never in its place.
        flops_exec = (nb.sum() * flops_backward(n, m, N) + nr.sum() * flops_forward(n, m, N) +
                      ngc.sum() * flops_costate(n, m, N) + nfo.sum() * flops_first_order(n, m, N)) / len(ms)
        flops_launch = (nit.sum() * flops_backward(n, m, N) + (nit.sum() + ntr.sum() + B * K) * flops_forward(n, m, N)) / len(ms)
        avg_ms = float(ms.mean())
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12
        bytes_launch = B * K * bytes_solve(n, m, N, 2 * m) / len(ms)
        rep_rate = sorted(solves / w for w in rep_wall)
        out = {
            "metric": "MPC solves/sec (batched iLQR to tol), random_linear_mpc n=12 m=4 N=50",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "preheat_ms": a.preheat_ms,
            "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "random_linear_mpc n=12 m=4 N=50 batch=%d per GPU (BASELINE configs[1])" % B,
                       "batch_per_gpu": B, "global_batch": world * B,
                       "options": "tol 1e-4, penalty_initial 1000, penalty_scaling 100, reset_duals=false "
                                  "(run_random_linear.jl:41-49)",
                       "parallelism": "instances sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": measured_traffic(B, K, W, spl, world),
                         "kernel": "altro::solve_kernel<12,4>", "avg_launch_ms": avg_ms, "launches": int(len(ms)),
                         "note": "FP64 VALU (v_fmac_f64_dpp) kernel, no MFMA: the compute roof is the FP64 vector peak "
                                 "(= the FP64 matrix peak on MI355X); traffic = measured HBM bytes per launch "
                                 "(committed rocprofv3 PMC passes of this command line) or null",
                         "algorithmic_note": "achieved = SURVEY 8(d) flops_solve (measured inner iterations and trials x the "
                                             "Riccati / rollout formulas) per launch / avg launch duration",
                         "executed": {"achieved": flops_exec / (avg_ms * 1e-3) / 1e12,
                                      "frac": flops_exec / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                      "note": "flops of the passes the kernel ran (backward passes, rollouts, first-order and costate "
                                              "sweeps): iterations whose active set and penalty are those of the gains in memory, and "
                                              "iterations that only confirm convergence, run no backward pass in the default mode "
                                              "(altro_opts.strict = 1 runs them all)"},
                         "hbm_algorithmic_GBps": bytes_launch / (avg_ms * 1e-3) / 1e9,
                         "hbm_frac": bytes_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "cpu_baseline": cpu,
            "repeats": {"regions": R, "steps_each": K, "wall_ms": [1e3 * w for w in rep_wall], "kernel_ms": rep_kernel,
                        "solves_per_s_median": rep_rate[len(rep_rate) // 2], "solves_per_s_min": rep_rate[0],
                        "solves_per_s_max": rep_rate[-1],
                        "note": "`value` is region 0 (W warm-up steps, then exactly K steps); regions 1.. are the next K steps of the "
                                "same closed loops, each bracketed the same way"},
            "steps_per_launch": spl,
            "solve_succeeded_frac": ok / (world * B * K),
            "iterations_mean": float(nit.sum() / (B * K)),
            "iterations_hist_last_step": np.bincount(st.iterations).tolist(),
            "interp_trials_per_solve": float(ntr.sum() / (B * K)),
            "backward_passes_per_solve": float(nb.sum() / (B * K)),
            "gains_from_memory_iterations_per_solve": float(nfo.sum() / (B * K)),
            "rollouts_per_solve": float(nr.sum() / (B * K)),
            "costate_confirmed_iterations_per_solve": float(ngc.sum() / (B * K)),
            "secondary": secondary,
        }
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
