"""Multi-GPU layout of the env batch.

Environments are independent, so the batch is split contiguously over ranks (one process per GPU) and no
collective is needed inside ``step``.  The only exchange is one all-gather of the per-env episode returns at the
end of every episode (RCCL over xGMI with the ``nccl`` backend; ``gloo`` in CPU tests).  Per-env seeds are
functions of the GLOBAL env id, so results do not depend on the number of ranks.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of ``range(total)`` owned by ``rank`` (first ``total % world`` ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def global_env_ids(total: int, rank: int, world: int):
    s, e = shard_range(total, rank, world)
    return list(range(s, e))


class EpisodeReturnGatherer:
    """Accumulates per-env rewards of the local shard and all-gathers the episode returns when the episode ends
    (the logged quantity of the reference's rollout is built from them: algorithm.py:509-510)."""

    def __init__(self, local_envs: int, device, distributed: bool, group=None):
        import torch

        self._torch = torch
        self.local_envs = int(local_envs)
        self.device = device
        self.distributed = bool(distributed)
        self.group = group
        self.returns = torch.zeros(self.local_envs, dtype=torch.float32, device=device)
        self.last_global_returns = None
        if self.distributed:
            import torch.distributed as dist

            self._dist = dist
            self.world = dist.get_world_size(group)
            self._out = torch.empty(self.world * self.local_envs, dtype=torch.float32, device=device)

        self._attached = None

    def attach(self, env):
        """Let ``env`` (a ``BatchedAOEnv``) add each step's rewards into ``self.returns`` inside its own last kernel; ``add`` then
        has nothing left to do."""
        env.accumulate_returns(self.returns)
        self._attached = env

    def detach(self):
        if self._attached is not None:
            self._attached.accumulate_returns(None)
            self._attached = None

    def start_episode(self):
        self.returns.zero_()

    def add(self, reward):
        if self._attached is None:
            self.returns += reward

    def finish_episode(self):
        """Returns the [world * local_envs] tensor of episode returns ordered by global env id."""
        if self.distributed:
            self._dist.all_gather_into_tensor(self._out, self.returns, group=self.group)
            self.last_global_returns = self._out
        else:
            self.last_global_returns = self.returns.clone()
        return self.last_global_returns
