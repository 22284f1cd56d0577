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


def test_shard_generation_matches_global_batch():
    full = altro.problems.gen_random_linear_batch(6, n=6, m=3, N=9, steps=2, seed=4)
    part = altro.problems.gen_random_linear_batch(3, n=6, m=3, N=9, steps=2, seed=4,
                                                  first_instance=altro.parallel.shard_first_instance(1, 3))
    assert np.array_equal(full.A[3:], part.A) and np.array_equal(full.Xtrack[3:], part.Xtrack)
    assert np.array_equal(full.noise[:, 3:], part.noise)
