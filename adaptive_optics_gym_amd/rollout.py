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
here are what they consume: ``replay_transitions`` flattens them into the five arrays the reference's replay buffer stores and
samples (``replay_buffer.py:21-34``: state, action, reward, next_state, done), ``sharding.gather_transitions`` all-gathers them over
the ranks for a central buffer.
"""
from __future__ import annotations

import math
import weakref


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


def actor_layers(actor):
    """The four ``nn.Linear`` of a policy module in forward order, or None.  Two structures are recognised by ATTRIBUTE NAME: the reference's own
    ``Actor`` (``network.py:17-39``: ``layer1a``, ``layer2a``, ``layer3a``, ``outputa``) — so ``rollout(env, reference_actor)`` takes the fused
    kernel like any other — and ``make_actor``'s (``hidden`` = three layers, ``out``)."""
    names = ("layer1a", "layer2a", "layer3a", "outputa")
    if all(hasattr(actor, n) for n in names):
        return [getattr(actor, n) for n in names]
    if hasattr(actor, "hidden") and hasattr(actor, "out") and len(list(actor.hidden)) == 3:
        return list(actor.hidden) + [actor.out]
    return None


class OrnsteinUhlenbeckNoise:
    """DDPG's exploration noise (``network.py:259-274``; added to the action at ``algorithm.py:258-259``) for a batch of envs on the device:
    ``state += theta (mu - state) + sigma N(0, I)``; ``sample`` returns the state itself.  One independent process per env
    (the reference has one env, hence one process); ``reset`` puts every state back to ``mu``.  Like the reference's it is NOT reset between
    episodes unless the caller does so."""

    def __init__(self, num_envs: int, action_dim: int, mu: float, theta: float, sigma: float, device=None, generator=None):
        import torch

        self.mu, self.theta, self.sigma = float(mu), float(theta), float(sigma)
        self.generator = generator
        self.state = torch.full((int(num_envs), int(action_dim)), self.mu, dtype=torch.float32, device=device)

    def reset(self):
        self.state.fill_(self.mu)

    def sample(self):
        import torch

        eps = torch.randn(self.state.shape, dtype=self.state.dtype, device=self.state.device, generator=self.generator)
        self.state += self.theta * (self.mu - self.state) + self.sigma * eps
        return self.state


# rollout()'s DeviceActor per actor module.  Kept here, NOT on the module: a DeviceActor holds the ctypes library handle, which can be
# neither deep-copied (target networks) nor pickled (torch.save of the module).  Weak keys, and DeviceActor only holds a weak
# reference back to its module, so neither keeps the other alive.
_DEVICE_ACTORS = weakref.WeakKeyDictionary()


class DeviceActor:
    """The policy query (``Actor.forward`` + ``get_action``, network.py:48-69) as ONE library launch (``aog_actor_act``): the
    weights of a torch module built by ``make_actor`` or of the reference's own ``Actor`` (``actor_layers``: recognised by attribute name)
    are read in place on every call, so a learner may keep updating them.  Dropout masks / Gaussian noise come from the
    library's Philox streams keyed by (seed, call counter, GLOBAL env id, layer, unit), not from torch's generator.  Keep ONE
    instance alive for the whole training run: its call counter is what makes every query draw fresh masks and noise
    (``rollout`` caches it per actor module for that reason).  ``env_id_base`` = global id of obs row 0 (multi-GPU: the
    env's ``global_env_offset``), so ranks explore with independent noise and a split batch reproduces the unsplit one."""

    def __init__(self, actor, seed: int = 0, dropout_p: float = 0.5, env_id_base: int = 0):
        import ctypes as C

        from . import _lib

        self._C, self._lib_mod = C, _lib
        if actor_layers(actor) is None:
            raise ValueError("DeviceActor: the module exposes neither layer1a / layer2a / layer3a / outputa (the reference's Actor, network.py:17-39) "
                             "nor hidden (3 x nn.Linear) + out")
        self.lib = _lib.load()
        self._actor_ref = weakref.ref(actor)
        self.seed = int(seed)
        self.dropout_p = float(dropout_p)
        self.env_id_base = int(env_id_base)
        self.calls = 0

    @property
    def actor(self):
        a = self._actor_ref()
        if a is None:
            raise RuntimeError("DeviceActor: the actor module it was built for has been garbage-collected")
        return a

    def __call__(self, obs, cov_var: float = 0.5, out=None):
        """obs [B, S] float16 or float32 on the GPU -> (action [B, A] float32, log_prob [B] float32, mean [B, A])."""
        import torch

        C, _lib = self._C, self._lib_mod
        layers = actor_layers(self.actor)
        for layer in layers:
            if layer.weight.dtype != torch.float32 or not layer.weight.is_contiguous() or not layer.weight.is_cuda:
                raise ValueError("DeviceActor needs contiguous float32 CUDA weights")
        if obs.dtype not in (torch.float16, torch.float32) or not obs.is_contiguous():
            raise ValueError("obs must be a contiguous float16 or float32 tensor")
        B, S = obs.shape
        H, A = layers[0].weight.shape[0], layers[3].weight.shape[0]
        if layers[0].weight.shape[1] != S:
            raise ValueError("obs width does not match the actor's first layer")
        if out is None:
            action = torch.empty((B, A), dtype=torch.float32, device=obs.device)
            log_prob = torch.empty((B,), dtype=torch.float32, device=obs.device)
            mean = torch.empty((B, A), dtype=torch.float32, device=obs.device)
        else:
            action, log_prob, mean = out
        net = _lib.AogActor(B, S, H, A, self.env_id_base, 0, *[C.c_void_p(t.data_ptr()) for layer in layers for t in (layer.weight, layer.bias)],
                            self.dropout_p, float(cov_var), self.seed, self.calls)
        self.calls += 1
        _lib.check(self.lib.aog_actor_act(C.byref(net), obs.device.index or 0, C.c_void_p(obs.data_ptr()), int(obs.dtype == torch.float16),
                                          C.c_void_p(mean.data_ptr()), C.c_void_p(action.data_ptr()), C.c_void_p(log_prob.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
        return action, log_prob, mean


def rollout(env, actor, episodes: int = 1, cov_var: float = 0.5, gatherer=None, generator=None, actor_impl: str = "auto", seed: int = 0,
            dev_actor=None, lookahead: bool = False, policy: str = "actor", ou_noise=None):
    """Collect ``episodes`` lock-step episodes from ``env`` (a ``BatchedAOEnv``).

    ``actor_impl``: "hip" = the fused policy-query kernel (``DeviceActor``), "torch" = the module's own forward +
    ``sample_action``, "auto" = "hip" for CUDA modules with the ``make_actor`` structure.  The ``DeviceActor`` (and with it the
    call counter of its random streams) persists across calls: pass one in as ``dev_actor``, or let this function cache it (in a
    module-level weak dictionary keyed by the actor: the caller's module stays deep-copyable and picklable) — a training loop that calls ``rollout`` once per iteration (algorithm.py:156) then explores with fresh
    dropout masks and noise in every iteration, like the reference's torch generator does.  Returns a dict of device tensors
    shaped ``[T*E, B, ...]`` (obs, act, log_prob, rew, next_obs, done), ``ep_returns`` ``[E, B_global]`` (gathered over ranks
    when ``gatherer`` is distributed), ``avg_ep_rew`` (the reference's logged scalar) and ``batch_lens`` (``algorithm.py:226,283``: a numpy
    array of T*E zeros whose first E entries hold the length of each episode — always T here: ``done`` depends on the step counter only).

    ``policy="shack"`` (``algorithm.py:252-253``, the 'SHACK' algorithm): the action of every step comes from ``env.SH_step()`` (the
    Shack-Hartmann integrator on the device; needs an env built with ``SH_operation=True``), ``actor`` may be None and ``log_prob`` holds the
    reference's constant 1.  ``ou_noise`` (``algorithm.py:258-259``, DDPG): an ``OrnsteinUhlenbeckNoise`` whose sample is added to every
    action before the env sees it (and before it is stored, like the reference's in-place ``action +=``)."""
    import numpy as np
    import torch

    from .sharding import EpisodeReturnGatherer

    T = int(env.max_steps)
    B = env.num_envs
    if gatherer is None:
        gatherer = EpisodeReturnGatherer(B, env.device, False)
    if policy not in ("actor", "shack"):
        raise ValueError("policy must be 'actor' or 'shack'")
    shack = policy == "shack"
    if shack:
        if not getattr(env, "SH_operation", False):
            raise ValueError("policy='shack' needs an env created with SH_operation=True (AO_env.py:115-116, 254)")
        actor_impl = "none"
    elif actor_impl == "auto":
        actor_impl = "hip" if actor_layers(actor) is not None and next(actor.parameters()).is_cuda else "torch"
    if actor_impl != "hip":
        dev_actor = None
    elif dev_actor is None:
        base = int(getattr(env, "global_env_offset", 0))
        dev_actor = _DEVICE_ACTORS.get(actor)
        if dev_actor is None or dev_actor.seed != int(seed) or dev_actor.env_id_base != base or dev_actor.actor is not actor:
            dev_actor = DeviceActor(actor, seed=seed, env_id_base=base)
            _DEVICE_ACTORS[actor] = dev_actor
    import inspect

    step_takes_out = "out" in inspect.signature(env.step).parameters
    # lookahead=True (dynamic atmosphere on the device stream): the next step's wind extrusion is launched on the library's own stream
    # beside this step's epilogue and the policy query — bit-identical results; the screens are off limits between two steps of an
    # episode, which this loop never touches.  Off by default: on ROCm 7.2 the two cross-stream event hand-offs it needs cost what the
    # overlap saves.
    looking_ahead = bool(env.lookahead(True)) if (lookahead and callable(getattr(env, "lookahead", None))) else False
    n = T * episodes
    out = None
    ep_returns = []
    with torch.no_grad():
        i = 0
        for _ in range(episodes):
            obs, _ = env.reset()
            gatherer.start_episode()
            i0 = i
            for _t in range(T):
                if out is None:   # buffers are laid out once the shapes are known: [T*E, B, ...], written in place
                    S = obs.shape[1]
                    A = int(env.num_modes) if shack else int(list(actor.parameters())[-1].shape[0])   # width of the output layer's bias
                    dev = obs.device
                    out = {"obs": torch.empty((n, B, S), dtype=obs.dtype, device=dev), "act": torch.empty((n, B, A), dtype=torch.float32, device=dev),
                           "log_prob": torch.empty((n, B), dtype=torch.float32, device=dev), "rew": torch.empty((n, B), dtype=torch.float32, device=dev),
                           "next_obs": torch.empty((n, B, S), dtype=obs.dtype, device=dev), "done": torch.empty((n, B), dtype=torch.bool, device=dev)}
                    mean_buf = torch.empty((B, A), dtype=torch.float32, device=dev)
                if shack:
                    sh_act, _ = env.SH_step()                       # (actuators [B, A] float64, the reference's constant log-probability 1)
                    out["act"][i].copy_(sh_act)
                    out["log_prob"][i].fill_(1.0)
                    action = out["act"][i]
                elif dev_actor is not None:
                    action, _, _ = dev_actor(obs, cov_var, out=(out["act"][i], out["log_prob"][i], mean_buf))
                else:
                    action, log_prob = sample_action(actor(obs), cov_var, generator)
                    out["act"][i].copy_(action)
                    out["log_prob"][i].copy_(log_prob)
                if ou_noise is not None:
                    out["act"][i].add_(ou_noise.sample())           # algorithm.py:258-259
                    action = out["act"][i]
                if step_takes_out:   # the env writes the transition straight into this step's slices
                    next_obs = env.step(action, out=(out["next_obs"][i], out["rew"][i], out["done"][i]))[0]
                else:
                    next_obs, rew, done, _, _ = env.step(action)
                    out["rew"][i].copy_(rew)
                    out["next_obs"][i].copy_(next_obs)
                    out["done"][i].copy_(done)
                if _t == 0:
                    out["obs"][i].copy_(obs)
                obs = next_obs
                i += 1
                # lock-step: done is identical for every env (AO_env.py:147), so no host sync is needed to break
            if T > 1:
                out["obs"][i0 + 1:i0 + T].copy_(out["next_obs"][i0:i0 + T - 1])   # obs of step t+1 = next_obs of step t
            gatherer.add(out["rew"][i0:i0 + T].sum(0))
            ep_returns.append(gatherer.finish_episode().clone())
    if looking_ahead:
        env.lookahead(False)
    out["ep_returns"] = torch.stack(ep_returns)
    out["avg_ep_rew"] = float(out["ep_returns"].mean().item()) / T
    lens = np.zeros(n)
    lens[:episodes] = T
    out["batch_lens"] = lens
    return out


def replay_transitions(batch, episode_major: bool = False):
    """The hand-off to a replay buffer (``ReplayBuffer.add(state, action, reward, next_state, done)``, ``replay_buffer.py:21-25``;
    ``sample_batch`` stacks exactly these five, ``:30-34``): the ``[T*E, B, ...]`` tensors of ``rollout`` as five flat device tensors
    ``(state [n, o*o], action [n, A], reward [n], next_state [n, o*o], done [n])``, n = T*E*B.

    Order: step-major (all envs of step 0, then step 1, ...) — the order a vectorised collector appends in; ``episode_major=True``
    gives env-major order instead (each env's steps contiguous: the order ``B`` single-env collectors appending one after the other
    would produce, ``algorithm.py:238-276``).  Views where the layout allows, no host copy."""
    keys = ("obs", "act", "rew", "next_obs", "done")
    out = []
    for k in keys:
        t = batch[k]
        if episode_major:
            t = t.transpose(0, 1)
        out.append(t.reshape((t.shape[0] * t.shape[1],) + tuple(t.shape[2:])))
    return tuple(out)
