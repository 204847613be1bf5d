#!/usr/bin/env python3
"""Headline benchmark: batched env-steps/s of the AOEnv.step() hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], per GPU): 1024 envs, quasi_static, 256x256 pupil, act_type=num_actuators
act_dim=64, obs_dim=2, strehl_ratio reward, 30-step episodes; synthetic von Karman screens (r0 = 0.20 m at
2.2 um, L0 = 10 m) synthesised on the device, actions ~ N(0, 0.5 I) resident in HBM.  A "step" is one
``BatchedAOEnv.step`` of all envs; every 30 steps the episode ends: ``reset()`` and one all-gather of the
per-env episode returns (RCCL when N > 1).  Envs are sharded across ranks with no data-path collective
(weak scaling: per-GPU batch fixed).

Rank 0 prints ONE JSON line (see the task contract): value = total env-steps / max-over-ranks wall time,
plus ``roofline`` (dominant kernel, timed live with HIP events on its own stream) and ``cpu_baseline`` (the
float64 numpy restatement of the reference's literal dataflow, timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(batch_per_gpu=1024, n_pupil=256, act_dim=64, obs_dim=2, atm_type="quasi_static", atm_fried=0.20,
                act_type="num_actuators", rew_type="strehl_ratio", timesteps_per_episode=30)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
SPINUP_STEPS = 300         # steps (spin-up + warm-up) before the timed region: see main()
PROFILE_EVERY = 8          # HIP events around one block of 8 launches of the fused kernel in 8 inside the timed region


def algorithmic_per_step(n_pupil, act_dim, obs_dim, batch, n_ap):
    """compulsory HBM bytes and flops of one env-step of the fused, collapsed dataflow."""
    K = obs_dim ** 2 + 4
    n2 = n_pupil * n_pupil
    bytes_ = 4 * n2 + 4 * act_dim + 4 * obs_dim ** 2 + 2 * obs_dim ** 2 + 9 + 4 * n2 * (act_dim + 2 * K) / batch
    flops = n_ap * (2 * act_dim + 8 * (obs_dim ** 2 + 3) + 10)
    return bytes_, flops


def cpu_baseline(budget_s=12.0):
    """Time the CPU oracle (literal HCIPy dataflow restated in numpy float64) on the same single-env shape."""
    import numpy as np

    from oracle.ao_env_oracle import AOEnvOracle

    w = WORKLOAD
    N = w["n_pupil"]
    rng = np.random.RandomState(0)
    from scipy.ndimage import gaussian_filter

    screen = gaussian_filter(rng.randn(N, N), 8.0)
    screen = screen / screen.std() * 3e-6
    env = AOEnvOracle(atm_type=w["atm_type"], atm_fried=w["atm_fried"], act_type=w["act_type"], act_dim=w["act_dim"],
                      obs_dim=w["obs_dim"], rew_type=w["rew_type"], timesteps_per_episode=w["timesteps_per_episode"],
                      num_pupil_pixels=N, screen=screen.ravel(), verbose=False)
    env.reset()
    a = rng.randn(w["act_dim"]).astype(np.float32)
    for _ in range(3):
        env.step(a)
    n, t0 = 0, time.perf_counter()
    while True:
        _, _, done, _, _ = env.step(a)
        n += 1
        if done:
            env.reset()
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        pass
    return {"value": n / dt, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} single-env steps (N=256, A=64, o=2, float64 numpy restatement of the literal HCIPy dataflow, "
                      f"not HCIPy itself) in {dt:.1f} s; host has {os.cpu_count()} logical cores"}


def strehl_check(env, screens_dev, torch):
    """Strehl / obs error of the device path vs the CPU oracle on the first 2 envs (same screens, same action)."""
    import numpy as np

    from oracle.ao_env_oracle import AOEnvOracle

    w = WORKLOAD
    a = torch.randn((env.num_envs, w["act_dim"]), device=env.device, generator=torch.Generator(env.device).manual_seed(5))
    _, _, _, _, info = env.step(a)
    out = {"strehl_abs_err": 0.0, "obs_rel_err": 0.0}
    for b in range(2):
        ref = AOEnvOracle(atm_type=w["atm_type"], atm_fried=w["atm_fried"], act_type=w["act_type"], act_dim=w["act_dim"],
                          obs_dim=w["obs_dim"], rew_type=w["rew_type"], timesteps_per_episode=w["timesteps_per_episode"],
                          num_pupil_pixels=w["n_pupil"], screen=screens_dev[b].double().cpu().numpy().ravel(), verbose=False)
        ref.reset()
        ref.step(a[b].cpu().numpy())
        out["strehl_abs_err"] = max(out["strehl_abs_err"], abs(float(info["strehl"][b]) - ref.last_strehl))
        out["obs_rel_err"] = max(out["obs_rel_err"],
                                 float(np.max(np.abs(info["obs_raw"][b].cpu().numpy() / ref.last_obs_raw - 1))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 episodes timed after 10 of warm-up (~0.15 s of device time).  A process's first few hundred steps run ~5 % slower
    # than steady state (tools/fixed_overhead.py; independent of event timing and of what the device did before), so short runs
    # under-report: 300 steps after 30 give ~13.0 M env-steps/s, 1500 after 300 ~13.7 M
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--kernel", default="auto", choices=["auto", "mfma", "valu"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("AOG_FORCE_DIST") == "1"   # the latter: rehearse the RCCL path with one rank
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch
    from adaptive_optics_gym_amd.params import OpticalParams
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    w = WORKLOAD
    B = w["batch_per_gpu"]
    p = OpticalParams(num_pupil_pixels=w["n_pupil"])
    # The handle first (seconds of host-side table building with the GPU idle), the synthetic input screens after it: the device then
    # goes from half a second of transform work straight into the warm-up instead of waking from idle inside the timed region
    # (the first ~25 ms after an idle spell run ~5 % slow).
    env = BatchedAOEnv(B, device, atm_type=w["atm_type"], atm_fried=w["atm_fried"], act_type=w["act_type"],
                       act_dim=w["act_dim"], obs_dim=w["obs_dim"], rew_type=w["rew_type"],
                       timesteps_per_episode=w["timesteps_per_episode"], num_pupil_pixels=w["n_pupil"],
                       screens=torch.zeros((B, w["n_pupil"], w["n_pupil"]), dtype=torch.float32, device=device),
                       kernel=args.kernel, verbose=False)
    gen = torch.Generator(device).manual_seed(1234 + rank)       # global env id = rank*B + e lives in the seed offset
    screens = screens_torch(B, p.num_pupil_pixels, p.pupil_pixel, cn_squared_from_fried_parameter(w["atm_fried"], p.wavelength_sci),
                            p.outer_scale, device, gen, oversampling=16)
    env.set_screens(screens)
    T = w["timesteps_per_episode"]
    agen = torch.Generator(device).manual_seed(10 + rank)        # main.py:155 seed; cov 0.5 I (algorithm.py:107)
    actions = torch.randn((T, B, w["act_dim"]), device=device, generator=agen) * (0.5 ** 0.5)
    gather = EpisodeReturnGatherer(B, device, distributed)
    gather.attach(env)                                            # episode returns accumulate inside the step's epilogue kernel

    def run(n_steps):
        t = 0
        env.reset()
        gather.start_episode()
        for i in range(n_steps):
            _, rew, _, _, _ = env.step(actions[t])
            gather.add(rew)
            t += 1
            if t == T:                                            # lock-step episode end (AO_env.py:147)
                gather.finish_episode()                           # all-gather of per-env episode returns
                env.reset()
                gather.start_episode()
                t = 0

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    env.profile(True, every=PROFILE_EVERY)   # switched on ahead of the warm-up: the first timed launches of a process pay ~1 ms of runtime set-up
    # Device spin-up (reported as config.spinup_steps): a process's first few hundred steps run ~5 % slower than steady state (device
    # clocks; tools/fixed_overhead.py).  With a caller-chosen warm-up shorter than that, the difference is run here, ahead of the
    # W warm-up steps, so that the K timed steps measure the steady state a long-running job sees.
    spinup = max(0, SPINUP_STEPS - args.warmup)
    if spinup:
        run(spinup)
    run(args.warmup)
    fence()
    env.profile_read()                       # discard the warm-up's samples; timing stays on
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    kernel_ms, launches = env.profile_read()
    env.profile(False)
    if distributed:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        bytes_step, flops_step = algorithmic_per_step(w["n_pupil"], w["act_dim"], w["obs_dim"], B, env.tables.n_ap)
        if launches == 0 or kernel_ms <= 0:
            raise SystemExit("bench.py: no fused-kernel launch was timed inside the measured region")
        k_s = kernel_ms * 1e-3
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ach_tf = flops_step * B / k_s / 1e12
        ach_gbs = bytes_step * B / k_s / 1e9
        result = {
            "metric": "env_steps_per_sec", "value": world * B * args.steps / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch=1024 envs/GPU, quasi_static, 256x256 pupil, act_type=num_actuators "
                                   "act_dim=64, obs_dim=2, strehl_ratio, 30-step episodes with reset + all-gather of returns",
                       "batch_per_gpu": B, "global_batch": world * B, "n_pupil": w["n_pupil"], "act_dim": w["act_dim"],
                       "obs_dim": w["obs_dim"], "kernel": {1: "valu", 2: "mfma"}.get(env.info.kernel, "ref"), "spinup_steps": spinup,
                       "parallelism": f"envs sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_fused_mfma" if os.environ.get("AOG_TABLES_MFMA") == "0" else "k_fused_tab",
                         "kernel_ms": kernel_ms, "launches_timed": launches, "bytes_per_env_step": bytes_step,
                         "timed_every": PROFILE_EVERY,
                         "note": "algorithmic bytes (SURVEY.md 8d: 282,913 B per env-step) x 1024 envs per launch / mean "
                                 "HIP-event duration of the fused kernel over the timed region (one block of 8 launches in 8 carries "
                                 "the two event records: they hold the stream ~6 us, which would otherwise be in every step); "
                                 "traffic = PMC (2*FETCH_SIZE + WRITE_SIZE) KiB per "
                                 "launch from profiles/traffic_latest.json; 6.29 TB/s is the measured copy ceiling"},
            "roofline_fp32": {"bound": "valu", "achieved": ach_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach_tf / FP32_PEAK_TFLOPS, "flops_per_env_step": flops_step,
                              "note": "SURVEY.md 8d algorithmic flops priced at the fp32 vector/MFMA peak; the surface "
                                      "contraction (2*A*n_ap of them) and the table sums actually run as split-f16 MFMAs, so "
                                      "this fraction overstates fp32 pipe use — the kernel is bound by the per-CU L2-served fill "
                                      "rate of its operands and its sin/cos work, then by HBM"},
        }
        if not args.no_parity:
            result["parity"] = strehl_check(env, screens, torch)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
