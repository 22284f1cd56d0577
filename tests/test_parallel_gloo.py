"""N > 1 path on CPU: two gloo ranks each own a shard of the batch (no data-path collective),
solve it, and all_gather first controls + status; the gathered result must equal the
single-process batch.  The solver on CPU ranks is the oracle (this is a test; the product
path has no CPU solver) -- what is under test is the sharding and the gather."""
import os
import socket
import sys
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "oracle"), os.path.join(_ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)   # also needed in the spawned ranks, which do not run conftest.py
import altro_amd_loader  # noqa: E402,F401
import altro_mpc_icra2021_amd as altro  # noqa: E402
from helpers import make_oracle, mpc_update  # noqa: E402

B_PER_RANK, STEPS = 3, 2


def solve_shard(first, count):
    import oracle_py
    pb = altro.problems.gen_random_linear_batch(count, n=6, m=3, N=15, steps=STEPS, seed=9, first_instance=first)
    U1 = np.zeros((count, pb.m))
    st = np.zeros(count, dtype=np.int64)
    for b in range(count):
        o = make_oracle(oracle_py, pb, b)
        o.solve()
        for i in range(STEPS):
            mpc_update(o, pb, b, i)
            s = o.solve()
        U1[b] = o.controls()[0]
        st[b] = s.status
    return U1, st


def _worker(rank, world, port, q):
    """bench.py's timed-region sequence over gloo, through the same RankGroup class bench.py uses on RCCL:
    init from the torch.distributed.run environment, barrier, work, barrier, MAX of the wall time over
    ranks, SUM of a counter, the final gather, close."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    grp = altro.parallel.RankGroup("gloo")
    assert (grp.rank, grp.world) == (rank, world) and dist.is_initialized()
    first = altro.parallel.shard_first_instance(grp.rank, B_PER_RANK)
    grp.barrier()
    t0 = time.perf_counter()
    U1, st = solve_shard(first, B_PER_RANK)
    if rank == 1:
        time.sleep(0.3)                      # the slower rank sets the job's time
    grp.barrier()
    dt_local = time.perf_counter() - t0
    dt = grp.max_over_ranks(dt_local if rank == 1 else 0.0)
    ok = grp.sum_over_ranks(int((st == altro.SOLVE_SUCCEEDED).sum()))
    allU, allS = grp.gather(U1, st)
    if rank == 0:
        q.put((allU, allS, dt, ok))
    grp.close()
    assert not dist.is_initialized()


def test_two_rank_shards_equal_single_batch(oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join(
        [_ROOT, os.path.join(_ROOT, "oracle"), os.path.join(_ROOT, "tests"), os.environ.get("PYTHONPATH", "")])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    while q.empty():
        assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died"
        assert time.time() - t0 < 120, "ranks timed out"
        time.sleep(0.1)
    allU, allS, dt, ok = q.get()
    assert dt >= 0.3 and ok == 2 * B_PER_RANK      # MAX over ranks came from rank 1; the counter was summed
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    Uref, Sref = solve_shard(0, 2 * B_PER_RANK)
    assert allU.shape == (2 * B_PER_RANK, 3)
    assert np.array_equal(allU, Uref)
    assert np.array_equal(allS, Sref)
    assert np.all(allS == altro.SOLVE_SUCCEEDED)


def _quad_shard(first, count, N=8):
    """tick 0 of the quadruped MPC loop (BASELINE configs[4]) for a shard of instances, solved by the oracle"""
    import oracle_py
    from helpers import quadruped_oracle
    P = altro.problems
    qb = P.gen_quadruped_batch(count, N=N, steps=1, seed=17, first_instance=first)
    U1 = np.zeros((count, 12))
    st = np.zeros(count, dtype=np.int64)
    for b in range(count):
        o = quadruped_oracle(oracle_py, qb.qp, qb.x0[b], qb.A[b, :N - 1], qb.Bm[b, :N - 1], qb.d[b, :N - 1], P.QUADRUPED_OPTS)
        s = o.solve()
        U1[b] = o.controls()[0]
        st[b] = s.status
    return U1, st


def _state_dim_shard(first, count):
    """one warm step of a state-dimension sweep point (BASELINE configs[3]; a small n here) for a shard of instances"""
    import oracle_py
    pb = altro.problems.gen_random_linear_batch(count, n=5, m=2, N=11, steps=1, seed=10, first_instance=first)
    U1 = np.zeros((count, pb.m))
    st = np.zeros(count, dtype=np.int64)
    for b in range(count):
        o = make_oracle(oracle_py, pb, b)
        o.solve()
        mpc_update(o, pb, b, 0)
        s = o.solve()
        U1[b] = o.controls()[0]
        st[b] = s.status
    return U1, st


def _secondary_worker(rank, world, port, q):
    """bench.py --config state_dim|quadruped under torch.distributed.run: the shard of each config's instances that
    belongs to this rank (parallel.shard_first_instance), the timed bracket (RankGroup.timed) and the one gather
    afterwards (RankGroup.gather_checked) -- the methods bench.py calls, here over gloo"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    grp = altro.parallel.RankGroup("gloo")
    out = {}
    for name, fn, B in (("state_dim", _state_dim_shard, 3), ("quadruped", _quad_shard, 2)):
        first = altro.parallel.shard_first_instance(grp.rank, B)
        res = {}

        def run():
            res["U1"], res["st"] = fn(first, B)
            if rank == 0:
                time.sleep(0.2)          # the slower rank sets the job's time

        dt = grp.timed(run)
        allU, allS = grp.gather_checked(res["U1"], res["st"])
        out[name] = (allU, allS, dt)
    if rank == 0:
        q.put(out)
    grp.close()


def test_secondary_configs_shard_over_two_ranks(oracle):
    """BASELINE configs[3] and [4] are defined as shards over 8 GPUs: their multi-rank path on two gloo ranks"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join(
        [_ROOT, os.path.join(_ROOT, "oracle"), os.path.join(_ROOT, "tests"), os.environ.get("PYTHONPATH", "")])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_secondary_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    while q.empty():
        assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died"
        assert time.time() - t0 < 240, "ranks timed out"
        time.sleep(0.1)
    out = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for name, fn, B in (("state_dim", _state_dim_shard, 3), ("quadruped", _quad_shard, 2)):
        allU, allS, dt = out[name]
        Uref, Sref = fn(0, 2 * B)            # the single-process batch of the same global instances
        assert dt >= 0.2
        assert np.array_equal(allU, Uref) and np.array_equal(allS, Sref), name
        assert np.all(allS == altro.SOLVE_SUCCEEDED), name


def test_quadruped_shard_generation_matches_global_batch():
    P = altro.problems
    full = P.gen_quadruped_batch(4, N=6, steps=2, seed=17)
    part = P.gen_quadruped_batch(2, N=6, steps=2, seed=17, first_instance=altro.parallel.shard_first_instance(1, 2))
    for f in ("t0", "x0", "A", "Bm", "d"):
        assert np.array_equal(getattr(full, f)[2:], getattr(part, f)), f
    assert np.array_equal(full.noise[:, 2:], part.noise)


def test_shard_generation_matches_global_batch():
    full = altro.problems.gen_random_linear_batch(6, n=6, m=3, N=9, steps=2, seed=4)
    part = altro.problems.gen_random_linear_batch(3, n=6, m=3, N=9, steps=2, seed=4,
                                                  first_instance=altro.parallel.shard_first_instance(1, 3))
    assert np.array_equal(full.A[3:], part.A) and np.array_equal(full.Xtrack[3:], part.Xtrack)
    assert np.array_equal(full.noise[:, 3:], part.noise)
