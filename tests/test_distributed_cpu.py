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
    g = EpisodeReturnGatherer(e - s, torch.device("cpu"), True)
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


def test_episode_return_allgather_world2():
    world, total, steps = 2, 16, 5
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


def test_single_process_gatherer_matches():
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    g = EpisodeReturnGatherer(4, torch.device("cpu"), False)
    g.start_episode()
    g.add(torch.tensor([1.0, 2.0, 3.0, 4.0]))
    g.add(torch.tensor([1.0, 1.0, 1.0, 1.0]))
    assert g.finish_episode().tolist() == [2.0, 3.0, 4.0, 5.0]
