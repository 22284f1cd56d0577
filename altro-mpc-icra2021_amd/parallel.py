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
