"""Sharding a Monte-Carlo ensemble of arenas over the GPUs of one node.

The reference runs its Monte-Carlo games one after another in a Python loop
(py/MonteCarlo.py:23-45); the runs are independent, so here they are the batch dimension and
are split into contiguous arena ranges, one per rank (one process per GPU).  Nothing is
exchanged while stepping; the only collective is one all-gather of the final states
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np


def partition(n_arenas, world_size):
    """Contiguous [lo, hi) per rank; the first n_arenas % world_size ranks hold one arena more."""
    base, extra = divmod(int(n_arenas), int(world_size))
    bounds = [0]
    for r in range(world_size):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def shard_columns(array, lo, hi):
    """Slice the arena axis (always the last one) of an input array."""
    return np.ascontiguousarray(array[..., lo:hi])


def all_gather_states(local_state, n_arenas, group=None):
    """Every rank contributes its (S, b_r) block and receives the full (S, n_arenas) array.

    local_state: torch tensor on the rank's device (cuda for nccl, cpu for gloo), shape (S, b_r)
    with b_r this rank's shard width under partition().  Shards may differ by one arena, so blocks
    are padded to the widest shard for a single all_gather_into_tensor and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = partition(n_arenas, world)
    width = max(hi - lo for lo, hi in parts)
    S = local_state.shape[0]
    send = torch.zeros((S, width), dtype=local_state.dtype, device=local_state.device)
    send[:, :local_state.shape[1]] = local_state
    recv = torch.empty((world, S, width), dtype=local_state.dtype, device=local_state.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    out = torch.empty((S, n_arenas), dtype=local_state.dtype, device=local_state.device)
    for r, (lo, hi) in enumerate(parts):
        out[:, lo:hi] = recv[r, :, :hi - lo]
    return out


class ShardedEnsemble:
    """One rank's view of an ensemble sharded over the ranks of a torch.distributed job.

    Every rank constructs it with the same total size; it owns the contiguous arena range
    partition(n_total, world)[rank] on GPU `device` (default LOCAL_RANK), slices full-size inputs
    (arena axis last) to that range, steps only its shard, and `gather_states()` returns the full
    (S, n_total) array on every rank -- the job's only collective.
    """

    def __init__(self, n_total, n_per_team=2, device=None, group=None):
        import os
        import torch.distributed as dist
        from . import BatchedGameSim
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = int(n_total)
        self.lo, self.hi = partition(self.n_total, self.world)[self.rank]
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.sim = BatchedGameSim(self.hi - self.lo, n_per_team, dev)

    def shard(self, array):
        array = np.asarray(array)
        return shard_columns(array, self.lo, self.hi) if array.shape[-1] == self.n_total else array

    def reset(self, dt, winning_score, x0_ball, *rest, **kw):
        return self.sim.reset(dt, winning_score, self.shard(x0_ball), *rest, **kw)

    def run_table(self, table, **kw):
        return self.sim.run_table(self.shard(table), **kw)

    def gather_states(self, on_device=True):
        """(S, n_total) torch tensor on every rank: cuda (RCCL) if on_device else cpu (gloo)."""
        import torch
        if self.world == 1:
            return torch.from_numpy(self.sim.state())
        if on_device:
            t = torch.empty((self.sim.state_size, self.hi - self.lo), dtype=torch.float64,
                            device=torch.device("cuda", self.sim.device))
            self.sim.copy_state_to(t.data_ptr())
            self.sim.sync()
        else:
            t = torch.from_numpy(self.sim.state())
        return all_gather_states(t, self.n_total, self.group)

    def close(self):
        self.sim.close()
