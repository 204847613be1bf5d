"""world_size-2 gloo test of the only exchange on the multi-GPU path: the all-gather of episode returns."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard_range(total_envs, rank, world)
    g = EpisodeReturnGatherer(e - s, torch.device("cpu"), True, total_envs=total_envs)
    out = []
    for ep in range(2):
        g.start_episode()
        for t in range(steps):
            ids = torch.arange(s, e, dtype=torch.float32)
            g.add(-(ids + 1) * (t + 1) * (ep + 1))       # reward is a function of the GLOBAL env id
        out.append(g.finish_episode().clone())
    if rank == 0:
        q.put([o.tolist() for o in out])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [16, 17])   # 17: uneven shards (9 + 8) -> padded all-gather, trimmed back to global-env order
def test_episode_return_allgather_world2(total):
    world, steps = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tri = steps * (steps + 1) / 2
    for ep, vals in enumerate(got):
        expect = [-(i + 1) * tri * (ep + 1) for i in range(total)]
        assert vals == pytest.approx(expect)
    # the gathered order IS the single-process order: one gatherer over all envs fed the same per-env rewards
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    one = EpisodeReturnGatherer(total, torch.device("cpu"), False)
    for ep, vals in enumerate(got):
        one.start_episode()
        for t in range(steps):
            one.add(-(torch.arange(total, dtype=torch.float32) + 1) * (t + 1) * (ep + 1))
        assert one.finish_episode().tolist() == vals


def test_shard_ranges_tile_the_batch():
    from adaptive_optics_gym_amd.sharding import global_env_ids, shard_range

    for total, world in ((8192, 8), (17, 2), (5, 8), (1000, 3)):
        ids = [i for r in range(world) for i in global_env_ids(total, r, world)]
        assert ids == list(range(total))
        sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_single_process_gatherer_matches():
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    g = EpisodeReturnGatherer(4, torch.device("cpu"), False)
    g.start_episode()
    g.add(torch.tensor([1.0, 2.0, 3.0, 4.0]))
    g.add(torch.tensor([1.0, 1.0, 1.0, 1.0]))
    assert g.finish_episode().tolist() == [2.0, 3.0, 4.0, 5.0]


def _worker_transitions(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from adaptive_optics_gym_amd.sharding import gather_transitions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 6
    base = rank * n
    tr = (torch.arange(base, base + n, dtype=torch.float16).reshape(n, 1).repeat(1, 4), torch.full((n, 3), float(rank)),
          torch.arange(base, base + n, dtype=torch.float32), torch.zeros((n, 4), dtype=torch.float16), torch.arange(n) % 2 == rank)
    got = gather_transitions(tr, True)
    if rank == 0:
        q.put([g.tolist() for g in got] + [[str(g.dtype) for g in got]])
    dist.barrier()
    dist.destroy_process_group()


def test_transition_allgather_world2():
    """Optional exchange for a central replay buffer: rank-major all-gather of the five transition tensors; dtypes kept."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_transitions, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    state, action, reward, nxt, done, dtypes = got
    assert reward == [float(i) for i in range(12)]
    assert [row[0] for row in state] == [float(i) for i in range(12)] and len(state[0]) == 4
    assert [row[0] for row in action] == [0.0] * 6 + [1.0] * 6
    assert done == [i % 2 == 0 for i in range(6)] + [i % 2 == 1 for i in range(6)]
    assert dtypes == ["torch.float16", "torch.float32", "torch.float32", "torch.float16", "torch.bool"]
    from adaptive_optics_gym_amd.sharding import gather_transitions

    same = gather_transitions((torch.ones(2, 2),), False)
    assert same[0].shape == (2, 2)
