"""Multi-GPU plumbing.  Instances are independent MPC problems, so the batch is sharded over
ranks with no data-path collective; the only exchange is a gather of the results
an MPC consumer reads each tick (first controls + status), done with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests)."""
import numpy as np


def shard_first_instance(rank, batch_per_rank):
    """Global index of the first instance of `rank`'s shard (weak scaling: fixed work per rank).
    Instance streams are keyed by global index (problems.instance_rng), so the union of the
    shards is exactly the single-process batch of world*batch_per_rank instances."""
    return int(rank) * int(batch_per_rank)


def gather_results(U1, status, device=None):
    """all_gather of the first controls (B, m) and status (B,) of every rank.
    Returns (world*B, m) and (world*B,) numpy arrays on every rank; without an initialised
    process group it returns its inputs."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(U1), np.asarray(status)
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    u = torch.from_numpy(np.ascontiguousarray(U1, dtype=np.float64)).to(dev)
    s = torch.from_numpy(np.ascontiguousarray(status, dtype=np.int64)).to(dev)
    # concatenated-along-dim-0 output: the form both gloo and nccl accept
    ug = torch.empty((world * u.shape[0],) + tuple(u.shape[1:]), dtype=u.dtype, device=dev)
    sg = torch.empty((world * s.shape[0],), dtype=s.dtype, device=dev)
    dist.all_gather_into_tensor(ug, u)
    dist.all_gather_into_tensor(sg, s)
    return ug.reshape(-1, u.shape[-1]).cpu().numpy(), sg.reshape(-1).cpu().numpy()


class RankGroup:
    """The process-group side of bench.py's timed region, in one place so that the CPU tests run the very
    code the driver launches on 8 GPUs (there with backend "nccl" = RCCL, in tests/ with "gloo"):
    one process per GPU started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment); barrier + device synchronise on both sides of the timed region; MAX over ranks
    of the wall time; SUM over ranks of counters; one all_gather of the results afterwards."""

    def __init__(self, backend="nccl", device=None, rank=None, world=None):
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.cuda = backend == "nccl"
        self.device = device if device is not None else (torch.device("cuda", self.local_rank) if self.cuda else torch.device("cpu"))
        if self.cuda:
            torch.cuda.set_device(self.local_rank)
        self.owns_group = False
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool
            kw = {"device_id": self.device} if self.cuda else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.owns_group = True

    def barrier(self):
        """barrier + device synchronise: brackets the timed region on both sides"""
        if self.world > 1:
            self.dist.barrier()
        if self.cuda:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(t)
        return int(t.item())

    def gather(self, U1, status):
        return gather_results(U1, status, device=self.device)

    def timed(self, fn):
        """The timed region of every bench line: barrier + device synchronise, fn(), barrier + device synchronise;
        returns the MAX over ranks of the wall time.  fn must leave its own device work finished (the solver
        synchronises its HIP stream itself: it does not run on torch's current stream)."""
        import time
        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def gather_checked(self, U1, status):
        """The one collective of a run, after the timed region: all_gather of the first controls and the status of
        every rank's shard.  Rank r's rows must sit at [r B, (r + 1) B) of the result (a mis-ordered gather shows)."""
        U1, status = np.asarray(U1), np.asarray(status)
        B = U1.shape[0]
        allU, allS = self.gather(U1, status)
        assert allU.shape == (self.world * B,) + U1.shape[1:] and allS.shape == (self.world * B,), (allU.shape, allS.shape)
        assert np.array_equal(allU[self.rank * B:(self.rank + 1) * B], U1)
        assert np.array_equal(allS[self.rank * B:(self.rank + 1) * B], status)
        return allU, allS

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            if self.owns_group:
                self.dist.destroy_process_group()
