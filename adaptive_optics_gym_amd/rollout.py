"""Batched data collection over ``BatchedAOEnv`` — the device-resident counterpart of the reference's
``ALGORITHM.rollout`` (``algorithm.py:216-296``) and of the policy query it makes per step (``network.py:62-69``).

What is mirrored, and what is not:

* episode structure: ``reset()``, then ``timesteps_per_episode`` steps, break on ``done`` (``algorithm.py:238-276``) — here
  all envs run in lock-step because ``done`` depends only on the step counter (``AO_env.py:147``);
* the action is ``mean + N(0, 0.5 I)`` (``cov_var = 0.5``, ``algorithm.py:107-108``) from a 3-hidden-layer ReLU MLP whose
  dropout (p = 0.5) stays ACTIVE while acting (``network.py:39,48-55``; the reference never calls ``eval()``);
* batch layout ``(T*E, ...)`` per env (``algorithm.py:219-226``) gains a leading env axis: ``[T*E, B, ...]``;
* the logged scalar is ``mean(sum(ep_rew)) / T`` (``algorithm.py:509-510``), here over all envs of all ranks via one
  all-gather of episode returns per episode (``sharding.EpisodeReturnGatherer``).

The learners (SAC/DDPG/PPO updates, replay buffer) are outside the env hot path and are not rebuilt; the tensors returned
here are what they consume.
"""
from __future__ import annotations

import math


def make_actor(state_dim: int, act_dim: int, hidden_dim: int, init_w: float = 3e-3, device=None):
    """An MLP with the reference Actor's shape and initialisation ranges (``network.py:17-39``)."""
    import torch
    from torch import nn

    class Actor(nn.Module):
        def __init__(self):
            super().__init__()
            dims = [(state_dim, hidden_dim), (hidden_dim, hidden_dim), (hidden_dim, hidden_dim)]
            self.hidden = nn.ModuleList(nn.Linear(i, o) for i, o in dims)
            for layer, (i, _) in zip(self.hidden, dims):
                bound = 1.0 / math.sqrt(i)
                nn.init.uniform_(layer.weight, -bound, bound)
                nn.init.uniform_(layer.bias, -bound, bound)
            self.out = nn.Linear(hidden_dim, act_dim)
            nn.init.uniform_(self.out.weight, -init_w, init_w)
            nn.init.uniform_(self.out.bias, -init_w, init_w)
            self.dropout = nn.Dropout(0.5)

        def forward(self, obs):
            x = obs.to(torch.float32)
            for layer in self.hidden:
                x = self.dropout(torch.relu(layer(x)))
            return self.out(x)

    actor = Actor()
    actor.train()  # dropout stays on while acting, like the reference
    return actor.to(device) if device is not None else actor


def sample_action(mean, cov_var: float = 0.5, generator=None):
    """``MultivariateNormal(mean, cov_var * I).rsample()`` and its log-probability, batched."""
    import torch

    std = math.sqrt(cov_var)
    eps = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
    action = mean + std * eps
    k = mean.shape[-1]
    log_prob = -0.5 * (eps * eps).sum(-1) - 0.5 * k * math.log(2 * math.pi * cov_var)
    return action, log_prob


def rollout(env, actor, episodes: int = 1, cov_var: float = 0.5, gatherer=None, generator=None):
    """Collect ``episodes`` lock-step episodes from ``env`` (a ``BatchedAOEnv``).

    Returns a dict of device tensors shaped ``[T*E, B, ...]`` (obs, act, log_prob, rew, next_obs, done), ``ep_returns``
    ``[E, B_global]`` (gathered over ranks when ``gatherer`` is distributed) and ``avg_ep_rew`` (the reference's logged
    scalar)."""
    import torch

    from .sharding import EpisodeReturnGatherer

    T = int(env.max_steps)
    B = env.num_envs
    if gatherer is None:
        gatherer = EpisodeReturnGatherer(B, env.device, False)
    keys = ("obs", "act", "log_prob", "rew", "next_obs", "done")
    buf = {k: [] for k in keys}
    ep_returns = []
    with torch.no_grad():
        for _ in range(episodes):
            obs, _ = env.reset()
            gatherer.start_episode()
            for _t in range(T):
                mean = actor(obs)
                action, log_prob = sample_action(mean, cov_var, generator)
                next_obs, rew, done, _, _ = env.step(action)
                gatherer.add(rew)
                for k, v in zip(keys, (obs, action, log_prob, rew, next_obs, done)):
                    buf[k].append(v)
                obs = next_obs
                # lock-step: done is identical for every env (AO_env.py:147), so no host sync is needed to break
            ep_returns.append(gatherer.finish_episode().clone())
    out = {k: torch.stack(v) for k, v in buf.items()}
    out["ep_returns"] = torch.stack(ep_returns)
    out["avg_ep_rew"] = float(out["ep_returns"].mean().item()) / T
    return out
