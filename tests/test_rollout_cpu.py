"""Host logic of the batched rollout harness (no GPU): a stand-in env with the BatchedAOEnv interface on CPU tensors."""
import math

import torch
from torch.distributions import MultivariateNormal

from adaptive_optics_gym_amd.rollout import make_actor, rollout, sample_action


class FakeEnv:
    def __init__(self, B, obs_n, A, T):
        self.num_envs, self.max_steps, self.device = B, T, torch.device("cpu")
        self.obs_n, self.A, self.t = obs_n, A, 0

    def reset(self):
        self.t = 0
        return torch.zeros(self.num_envs, self.obs_n, dtype=torch.float16), {}

    def step(self, a):
        self.t += 1
        rew = -a.abs().mean(dim=1) - self.t
        done = torch.full((self.num_envs,), self.t == self.max_steps)
        obs = torch.full((self.num_envs, self.obs_n), float(self.t), dtype=torch.float16)
        return obs, rew, done, torch.zeros(self.num_envs, dtype=torch.bool), {}


def test_actor_shape_init_and_active_dropout():
    torch.manual_seed(0)
    actor = make_actor(4, 64, 150)
    assert [tuple(l.weight.shape) for l in actor.hidden] == [(150, 4), (150, 150), (150, 150)]
    assert tuple(actor.out.weight.shape) == (64, 150)
    assert actor.hidden[0].weight.abs().max() <= 0.5 and actor.hidden[1].weight.abs().max() <= 1 / math.sqrt(150)
    assert actor.out.weight.abs().max() <= 3e-3
    x = torch.ones(3, 4)
    assert not torch.equal(actor(x), actor(x))      # dropout is live while acting (network.py:48-55)


def test_sample_action_matches_multivariate_normal_logprob():
    torch.manual_seed(1)
    mean = torch.randn(5, 6)
    a, lp = sample_action(mean, 0.5)
    ref = MultivariateNormal(mean, torch.diag(torch.full((6,), 0.5))).log_prob(a)
    torch.testing.assert_close(lp, ref, rtol=1e-5, atol=1e-5)
    big = sample_action(torch.zeros(20000, 4), 0.5)[0]
    assert abs(float(big.var()) - 0.5) < 0.02


def test_rollout_layout_and_logged_scalar():
    torch.manual_seed(2)
    B, T, E = 3, 4, 2
    env = FakeEnv(B, 4, 6, T)
    out = rollout(env, make_actor(4, 6, 16), episodes=E)
    assert out["obs"].shape == (T * E, B, 4) and out["act"].shape == (T * E, B, 6)
    assert out["rew"].shape == (T * E, B) and out["done"].shape == (T * E, B) and out["log_prob"].shape == (T * E, B)
    assert out["done"][T - 1].all() and not out["done"][:T - 1].any()
    assert torch.equal(out["next_obs"][0], out["obs"][1])
    ep = out["rew"].reshape(E, T, B).sum(1)
    torch.testing.assert_close(out["ep_returns"], ep)
    assert abs(out["avg_ep_rew"] - float(ep.mean()) / T) < 1e-6   # algorithm.py:509-510


def test_replay_transitions_match_per_env_collection():
    """The replay hand-off (replay_buffer.py:21-34): flattened (state, action, reward, next_state, done) of the batched rollout equal
    what B single-env collectors appending one transition per step would have stored."""
    from adaptive_optics_gym_amd.rollout import replay_transitions

    torch.manual_seed(5)
    B, T, E = 3, 4, 2
    env = FakeEnv(B, 4, 6, T)
    out = rollout(env, make_actor(4, 6, 16), episodes=E)
    st, ac, rw, ns, dn = replay_transitions(out)
    n = T * E * B
    assert st.shape == (n, 4) and ac.shape == (n, 6) and rw.shape == (n,) and ns.shape == (n, 4) and dn.shape == (n,)
    # step-major: transition (step i, env b) sits at i * B + b
    for i in (0, 3, 5):
        for b in range(B):
            k = i * B + b
            assert torch.equal(st[k], out["obs"][i, b]) and torch.equal(ac[k], out["act"][i, b]) and torch.equal(ns[k], out["next_obs"][i, b])
            assert rw[k] == out["rew"][i, b] and dn[k] == out["done"][i, b]
    # env-major: each env's trajectory contiguous, in time order (what a single-env loop appends, algorithm.py:238-276)
    st2, ac2, rw2, ns2, dn2 = replay_transitions(out, episode_major=True)
    for b in range(B):
        torch.testing.assert_close(rw2[b * T * E:(b + 1) * T * E], out["rew"][:, b])
        assert torch.equal(ns2[b * T * E:(b + 1) * T * E - 1][: T - 1], st2[b * T * E + 1:b * T * E + T])   # next_state of t = state of t + 1
    assert int(dn.sum()) == E * B


class _PicklableActor(torch.nn.Module):
    """Module-level class (picklable) with the structure DeviceActor reads: ``hidden`` = three Linear layers, ``out``."""

    def __init__(self):
        super().__init__()
        self.hidden = torch.nn.ModuleList([torch.nn.Linear(4, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)])
        self.out = torch.nn.Linear(8, 3)

    def forward(self, x):
        for layer in self.hidden:
            x = torch.relu(layer(x.float()))
        return self.out(x)


def test_device_actor_cache_leaves_the_module_copyable_and_picklable():
    """rollout() keeps its DeviceActor (which holds the ctypes library handle) in a weak dictionary beside the module, not on it:
    target-network deep copies and torch.save of the module keep working after a rollout, and nothing outlives the module."""
    import copy
    import gc
    import io
    import pickle

    from adaptive_optics_gym_amd import rollout as ro

    actor = _PicklableActor()
    da = ro.DeviceActor(actor, seed=3)          # loads libaogym.so (no GPU call)
    ro._DEVICE_ACTORS[actor] = da               # what rollout(actor_impl="hip") does
    assert ro._DEVICE_ACTORS.get(actor) is da and da.actor is actor
    assert not any("aog" in k for k in vars(actor))          # nothing was attached to the caller's module
    twin = copy.deepcopy(actor)
    assert twin is not actor and ro._DEVICE_ACTORS.get(twin) is None
    pickle.loads(pickle.dumps(actor))
    torch.save(actor, io.BytesIO())
    n_before = len(ro._DEVICE_ACTORS)
    del actor, twin
    gc.collect()
    assert len(ro._DEVICE_ACTORS) == n_before - 1            # the cache entry went with the module
    try:
        da.actor
        raise AssertionError("expected RuntimeError")
    except RuntimeError:
        pass


class ReferenceNamedActor(torch.nn.Module):
    """A stand-in carrying the attribute names of the reference's Actor (network.py:17-39: layer1a, layer2a, layer3a, outputa, dropout); the
    reference's own file cannot be imported here (its imports need gym), and only the names matter to ``actor_layers``."""

    def __init__(self, s, a, h):
        super().__init__()
        self.layer1a, self.layer2a, self.layer3a = torch.nn.Linear(s, h), torch.nn.Linear(h, h), torch.nn.Linear(h, h)
        self.outputa = torch.nn.Linear(h, a)
        self.dropout = torch.nn.Dropout(0.5)

    def forward(self, state):
        x = state.to(torch.float32)
        for layer in (self.layer1a, self.layer2a, self.layer3a):
            x = self.dropout(torch.relu(layer(x)))
        return self.outputa(x)


def test_actor_layers_recognises_the_reference_actor_by_attribute_name():
    from adaptive_optics_gym_amd.rollout import actor_layers

    ref = ReferenceNamedActor(4, 6, 16)
    assert [l is m for l, m in zip(actor_layers(ref), (ref.layer1a, ref.layer2a, ref.layer3a, ref.outputa))] == [True] * 4
    mine = make_actor(4, 6, 16)
    assert actor_layers(mine)[3] is mine.out and len(actor_layers(mine)) == 4
    assert actor_layers(torch.nn.Linear(4, 6)) is None
    out = rollout(FakeEnv(3, 4, 6, 4), ref, episodes=1)          # CPU module: the torch path, same result layout
    assert out["act"].shape == (4, 3, 6)


def test_rollout_returns_batch_lens_like_the_reference():
    out = rollout(FakeEnv(2, 4, 6, 5), make_actor(4, 6, 8), episodes=3)
    lens = out["batch_lens"]                     # algorithm.py:226,283: zeros(T * E) with the first E entries = episode lengths
    assert lens.shape == (15,) and (lens[:3] == 5).all() and (lens[3:] == 0).all()


class FakeShackEnv(FakeEnv):
    SH_operation = True

    def __init__(self, B, obs_n, A, T):
        super().__init__(B, obs_n, A, T)
        self.num_modes = A
        self.sh_calls = 0

    def SH_step(self):
        self.sh_calls += 1
        return torch.full((self.num_envs, self.A), 1e-7 * self.sh_calls, dtype=torch.float64), torch.tensor([1])


def test_rollout_shack_policy_drives_sh_step():
    env = FakeShackEnv(3, 25, 5, 4)
    out = rollout(env, None, episodes=2, policy="shack")          # algorithm.py:252-253
    assert env.sh_calls == 8 and out["act"].shape == (8, 3, 5)
    torch.testing.assert_close(out["act"][:, 0, 0], torch.arange(1, 9, dtype=torch.float32) * 1e-7)
    assert (out["log_prob"] == 1).all()
    try:
        rollout(FakeEnv(3, 4, 6, 4), None, policy="shack")
        assert False, "an env without SH_operation must be refused"
    except ValueError:
        pass


def test_ornstein_uhlenbeck_noise_update_rule_and_use_in_rollout():
    from adaptive_optics_gym_amd.rollout import OrnsteinUhlenbeckNoise

    g = torch.Generator().manual_seed(3)
    ou = OrnsteinUhlenbeckNoise(4, 6, mu=0.1, theta=0.15, sigma=0.2, generator=g)
    assert torch.equal(ou.state, torch.full((4, 6), 0.1))
    g2 = torch.Generator().manual_seed(3)
    x = torch.full((4, 6), 0.1)
    for _ in range(3):                           # network.py:270-273: dx = theta (mu - x) + sigma randn; x += dx; return x
        x = x + 0.15 * (0.1 - x) + 0.2 * torch.randn((4, 6), generator=g2)
        torch.testing.assert_close(ou.sample(), x)
    ou.reset()
    assert torch.equal(ou.state, torch.full((4, 6), 0.1))
    # in the rollout the sample is added to the action the env sees and to the stored action (algorithm.py:258-259)
    torch.manual_seed(5)
    actor = make_actor(4, 6, 8)
    seen = []

    class Spy(FakeEnv):
        def step(self, a):
            seen.append(a.clone())
            return super().step(a)

    ou = OrnsteinUhlenbeckNoise(4, 6, 0.0, 0.15, 0.2, generator=torch.Generator().manual_seed(7))
    out = rollout(Spy(4, 4, 6, 3), actor, episodes=1, ou_noise=ou)
    assert all(torch.equal(seen[t], out["act"][t]) for t in range(3))
