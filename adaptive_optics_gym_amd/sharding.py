"""Multi-GPU layout of the env batch.

Environments are independent, so the batch is split contiguously over ranks (one process per GPU) and no
collective is needed inside ``step``.  The only exchange is one all-gather of the per-env episode returns at the
end of every episode (RCCL over xGMI with the ``nccl`` backend; ``gloo`` in CPU tests).  Per-env seeds are
functions of the GLOBAL env id, so results do not depend on the number of ranks.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of ``range(total)`` owned by ``rank`` (first ``total % world`` ranks get one extra).  Pass the result to
    ``BatchedAOEnv(..., global_env_offset=start, total_envs=total)`` and ``EpisodeReturnGatherer(..., total_envs=total)``."""
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

    def __init__(self, local_envs: int, device, distributed: bool, group=None, total_envs=None):
        """``total_envs`` (optional): size of the global batch when it does not divide evenly over the ranks
        (``shard_range`` shards); the all-gather then runs on shards padded to ``ceil(total / world)`` and the result is
        trimmed back to ``total`` entries in global-env order."""
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
            self.total_envs = int(total_envs) if total_envs is not None else self.world * self.local_envs
            self.padded = -(-self.total_envs // self.world)     # ceil: every rank contributes the same number of slots
            rank = dist.get_rank(group)
            if self.local_envs > self.padded or shard_range(self.total_envs, rank, self.world)[1] - \
                    shard_range(self.total_envs, rank, self.world)[0] != self.local_envs:
                raise ValueError(f"EpisodeReturnGatherer: rank {rank} holds {self.local_envs} envs, which is not its "
                                 f"shard_range share of {self.total_envs} envs over {self.world} ranks")
            self._out = torch.empty(self.world * self.padded, dtype=torch.float32, device=device)
            self._stage = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
            self._send = self.returns if self.padded == self.local_envs else torch.zeros(self.padded, dtype=torch.float32, device=device)
            # positions of the real entries inside the padded gather, in global-env order
            keep = []
            for r in range(self.world):
                s_, e_ = shard_range(self.total_envs, r, self.world)
                keep.extend(range(r * self.padded, r * self.padded + (e_ - s_)))
            self._keep = None if len(keep) == self.world * self.padded else torch.tensor(keep, dtype=torch.long, device=device)

        self._attached = None
        self._timing = None      # time_collective(): [(start event, end event)] or host seconds

    def time_collective(self, enable=True):
        """Bracket every ``finish_episode`` exchange with device events (host clock for staged gloo rehearsals) so that a scaling run can
        report what the collective cost: ``collective_ms()`` -> (total milliseconds, exchanges) since the last call."""
        self._timing = [] if enable else None

    def collective_ms(self):
        if not self._timing:
            return 0.0, 0
        tot = 0.0
        for a, b in self._timing:
            if b is None:
                tot += a * 1e3
            else:
                b.synchronize()
                tot += a.elapsed_time(b)
        n = len(self._timing)
        self._timing = []
        return tot, n

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
            timed = self._timing is not None
            if self._send is not self.returns:
                self._send[: self.local_envs].copy_(self.returns)
            if self._stage:     # gloo does not all-gather device tensors: single-GPU rehearsals stage through the host
                import time

                t0 = time.perf_counter()
                out = self._torch.empty(self._out.shape, dtype=self._out.dtype)
                self._dist.all_gather_into_tensor(out, self._send.cpu(), group=self.group)
                self._out.copy_(out)
                if timed:
                    self._timing.append((time.perf_counter() - t0, None))
            else:
                ev = None
                if timed and self._torch.device(self.device).type == "cuda":
                    ev = (self._torch.cuda.Event(enable_timing=True), self._torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                self._dist.all_gather_into_tensor(self._out, self._send, group=self.group)
                if ev:
                    ev[1].record()
                    self._timing.append(ev)
            self.last_global_returns = self._out if self._keep is None else self._out.index_select(0, self._keep)
        else:
            self.last_global_returns = self.returns.clone()
        return self.last_global_returns


def gather_transitions(transitions, distributed: bool, group=None):
    """Optional exchange for a CENTRAL replay buffer: all-gather the five flat transition tensors of
    ``rollout.replay_transitions`` over the ranks (one collective per tensor, rank-major order; ~0.3 KB per env-step).  Every rank
    must contribute the same number of transitions (equal shards: the lock-step episodes guarantee equal step counts).  Identity
    when not distributed.  bool tensors travel as uint8 (RCCL has no bool type)."""
    if not distributed:
        return tuple(transitions)
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = []
    for t in transitions:
        is_bool = t.dtype == torch.bool
        send = (t.to(torch.uint8) if is_bool else t).contiguous()
        stage = send.device.type == "cuda" and dist.get_backend(group) == "gloo"   # single-GPU rehearsals: gloo moves host tensors
        src = send.cpu() if stage else send
        recv = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(recv, src, group=group)
        if stage:
            recv = recv.to(send.device)
        out.append(recv.to(torch.bool) if is_bool else recv)
    return tuple(out)
