#!/usr/bin/env python3
"""Headline benchmark: batched env-steps/s of the AOEnv.step() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Started WITHOUT torchrun (``WORLD_SIZE`` unset) and ``--gpus N > 1`` this script spawns its N ranks itself, as child
processes created before anything in the parent touches the GPU; started under torchrun it checks ``WORLD_SIZE == --gpus``
and exits non-zero otherwise.  One rank per GPU, RCCL (``nccl`` backend) for the one collective.

Workloads (BASELINE.json ``configs``; per GPU, weak scaling — the global batch is ``N x`` the per-GPU batch):

  --config 2 (default, the headline line)  configs[1]: 1024 envs, quasi_static, 256x256 pupil, num_actuators A=64, o=2,
             strehl_ratio, 30-step episodes.  step = ``BatchedAOEnv.step``; every 30 steps ``reset()`` + all-gather of returns.
  --config 3  configs[2]: 4096 envs, semi_dynamic (requested atm_vel=10 is coerced to 0 like the reference), r0=0.15, o=5,
             20-step episodes; every reset regenerates all screens on the device (two-band synthesis, oversampling 16).
  --config 4  configs[3]'s per-GPU shard: 1024 envs, dynamic v=10 m/s (random direction per env), o=2, SAC-style rollout:
             policy query (fused actor kernel, hidden 150) + step + in-place transition writes, 30-step episodes.
  --config 5  configs[4]: 2048 envs, 512x512 pupil, zernike A=20, o=5, smf_ssim, SH_operation=True; step = ``SH_step`` + ``step``.

Synthetic inputs: von Karman screens synthesised inside the library (Philox keyed by the GLOBAL env id = rank * batch + e, seed
1234), actions ~ N(0, 0.5 I) for the global batch from torch seed 10 (main.py:155), sliced per rank — resident in HBM before the
timed region.  Rank 0 prints ONE JSON line: value = total env-steps / max-over-ranks wall time, ``roofline`` (the fused kernel,
timed live with HIP events on its own stream) and ``cpu_baseline`` (the float64 numpy restatement of the reference's literal
dataflow, timed on this box's host cores in the three modes SURVEY.md §8d prescribes).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    2: dict(name="configs[1]", batch_per_gpu=1024, n_pupil=256, act_dim=64, obs_dim=2, atm_type="quasi_static", atm_vel=0, atm_fried=0.20,
            act_type="num_actuators", rew_type="strehl_ratio", timesteps_per_episode=30, SH_operation=False, rollout=False,
            text="batch=1024 envs/GPU, quasi_static, 256x256 pupil, act_type=num_actuators act_dim=64, obs_dim=2, strehl_ratio, "
                 "30-step episodes with reset + all-gather of returns"),
    3: dict(name="configs[2]", batch_per_gpu=4096, n_pupil=256, act_dim=64, obs_dim=5, atm_type="semi_dynamic", atm_vel=10, atm_fried=0.15,
            act_type="num_actuators", rew_type="strehl_ratio", timesteps_per_episode=20, SH_operation=False, rollout=False,
            text="batch=4096 envs/GPU, semi_dynamic (atm_vel=10 coerced to 0 like the reference) atm_fried=0.15, 256x256 pupil, 64 actuators, "
                 "obs_dim=5, 20-step episodes; every reset regenerates all screens on the device (two-band von Karman synthesis: hcipy's 16x-oversampled "
                 "grid below 2 cycles per pupil diameter + the (2N)^2 grid above; same covariance on the pupil to 2e-5 of the variance)"),
    4: dict(name="configs[3] per-GPU shard", batch_per_gpu=1024, n_pupil=256, act_dim=64, obs_dim=2, atm_type="dynamic", atm_vel=10, atm_fried=0.15,
            act_type="num_actuators", rew_type="strehl_ratio", timesteps_per_episode=30, SH_operation=False, rollout=True,
            text="batch=1024 envs/GPU (8192 over 8 GPUs), dynamic atmosphere v=10 m/s random direction per env, 256x256 pupil, 64 actuators, "
                 "obs_dim=2, SAC-style rollout (fused policy kernel, hidden 150, dropout on, N(0, 0.5 I) exploration) writing the "
                 "transition buffers in place, 30-step episodes, all-gather of returns per episode"),
    5: dict(name="configs[4]", batch_per_gpu=2048, n_pupil=512, act_dim=20, obs_dim=5, atm_type="quasi_static", atm_vel=0, atm_fried=0.15,
            act_type="zernike", rew_type="smf_ssim", timesteps_per_episode=20, SH_operation=True, rollout=False,
            text="batch=2048 envs/GPU, 512x512 pupil, act_type=zernike act_dim=20, obs_dim=5, smf_ssim, SH_operation=True: every step is "
                 "SH_step (Shack-Hartmann sensor image, photon noise, slopes, leaky integrator on the device) + step"),
}

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense f16/bf16 matrix peak (256 CUs x 4096 flop/clk x 2.4 GHz)
F64_MFMA_PEAK_TFLOPS = 78.6     # v_mfma_f64_16x16x4_f64: 256 CUs x 4 SIMDs x 2048 flop / 64 clk x 2.4 GHz (tools/microbench/mfma_f64.hip: 77 measured)
ISSUE_CYCLES_PEAK = 256 * 4 * 2.4e9   # SIMD issue cycles / s.  Issue-cycle model of a vector-bound kernel (MI355X_MICROARCH.md, row
                                      # 'vector-instruction ISSUE cost'): 8 cycles per transcendental and per matrix instruction, 4 per other
SPINUP_STEPS = 300           # steps (spin-up + warm-up) before the timed region: see main()
PROFILE_EVERY = 8            # HIP events around one block of 8 launches of the fused kernel in 8 inside the timed region


def algorithmic_per_step(n_pupil, act_dim, obs_dim, batch, n_ap):
    """compulsory HBM bytes and flops of one env-step of the fused, collapsed dataflow (contract figure), and
    the bytes the aperture-packed layout really has to move (4 n_ap instead of 4 N^2 per screen; shared tables amortised)."""
    K = obs_dim ** 2 + 4
    n2 = n_pupil * n_pupil
    io = 4 * act_dim + 4 * obs_dim ** 2 + 2 * obs_dim ** 2 + 9
    bytes_ = 4 * n2 + io + 4 * n2 * (act_dim + 2 * K) / batch
    layout = 4 * n_ap + io + 4 * n_ap * (act_dim + 2 * K) / batch
    flops = n_ap * (2 * act_dim + 8 * (obs_dim ** 2 + 3) + 10)
    return bytes_, flops, layout


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the ONLY part of this file that touches oracle/): float64 numpy restatement of the literal HCIPy dataflow
# ---------------------------------------------------------------------------------------------------------------------------------
_CPU_WORKER = r"""
import os, sys, time, json
sys.path.insert(0, {root!r})
import numpy as np
from scipy.ndimage import gaussian_filter
from oracle.ao_env_oracle import AOEnvOracle
w = {w!r}
N = w["n_pupil"]
rng = np.random.RandomState({seed})
screen = gaussian_filter(rng.randn(N, N), 8.0)
screen = screen / screen.std() * 3e-6
env = AOEnvOracle(atm_type="quasi_static", atm_fried=w["atm_fried"], act_type=w["act_type"], act_dim=w["act_dim"], obs_dim=w["obs_dim"],
                  rew_type=w["rew_type"], timesteps_per_episode=w["timesteps_per_episode"], num_pupil_pixels=N, screen=screen.ravel(),
                  rng=rng, verbose=False)
env.reset()
a = rng.randn(w["act_dim"]).astype(np.float32)
for _ in range(3):
    env.step(a)
sys.stdout.write("ready\n"); sys.stdout.flush()
sys.stdin.readline()                      # all workers start their timed loop together
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < {budget}:
    _, _, done, _, _ = env.step(a)
    n += 1
    if done:
        env.reset()
dt = time.perf_counter() - t0
threads = 1
try:
    from threadpoolctl import threadpool_info
    threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
except Exception:
    pass
print(json.dumps({{"n": n, "dt": dt, "threads": threads}}))
"""


def _host_cpus():
    """(logical cores the OS reports, cores this process may actually use: affinity mask and cgroup quota)."""
    total = os.cpu_count() or 1
    usable = total
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            usable = min(usable, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    # a 1-GPU box of this pool gives a job 16 host cores whatever the OS reports: the "share" figure never starts more workers than that
    # unless told to; `usable` = min(affinity mask, cgroup quota) is what SURVEY.md section 8d(iii) asks for and is reported beside it
    cap = int(os.environ.get("AOG_CPU_WORKERS", "16"))
    return total, max(1, usable), max(1, min(usable, cap))


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _run_cpu_workers(w, procs, threads, budget):
    """``procs`` worker processes, each stepping its own single-env oracle for ``budget`` seconds with ``threads`` BLAS threads
    (None = library default).  Returns (aggregate env-steps/s, BLAS threads a worker really used, total steps)."""
    env = dict(os.environ)
    if threads is not None:
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "BLIS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
            env[k] = str(threads)
    env["HIP_VISIBLE_DEVICES"] = ""          # the workers are CPU-only
    env.pop("LD_PRELOAD", None)              # (a profiler's preloaded library stays with the GPU process)
    ws = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER.format(root=ROOT, w={k: w[k] for k in (
        "n_pupil", "atm_fried", "act_type", "act_dim", "obs_dim", "rew_type", "timesteps_per_episode")}, seed=i, budget=budget)],
        stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env) for i in range(procs)]
    for p in ws:
        assert p.stdout.readline().strip() == "ready"
    for p in ws:
        p.stdin.write("go\n")
        p.stdin.flush()
    res = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ws]
    return sum(r["n"] / r["dt"] for r in res), max(r["threads"] for r in res), sum(r["n"] for r in res)


def cpu_baseline(w, budget_s=8.0):
    """(i) 1 process x 1 BLAS thread, (ii) 1 process x default threads, (iii) one single-thread process per usable
    host core.  ``value`` is (iii), the fair "all host cores" number; the other two are in ``modes``."""
    total, usable, share = _host_cpus()
    one, _, n1 = _run_cpu_workers(w, 1, 1, budget_s)
    dflt, dthreads, n2 = _run_cpu_workers(w, 1, None, budget_s)
    shr, _, n3 = _run_cpu_workers(w, share, 1, budget_s)
    # one single-thread worker per core of min(affinity, cgroup).  Each worker holds ~0.5 GB (mode matrices,
    # propagator tables): bounded by the memory the box has free; same budget, so the default run stays within minutes
    allp, nall = usable, None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                allp = max(1, min(allp, int(int(ln.split()[1]) * 1024 * 0.5 / 0.6e9)))
    except Exception:
        pass
    if allp > share:
        allc, _, nall = _run_cpu_workers(w, allp, 1, budget_s)
    else:
        allc, allp, nall = shr, share, n3
    import numpy as np

    modes = {"1proc_1thread": one, "1proc_default_threads": dflt, "default_threads": int(dthreads), f"{share}procs_1thread": shr}
    modes[f"{allp}procs_1thread"] = allc
    # BASELINE.md section 4 / configs[0] (the reference's own CPU-runnable case): 1 env, quasi_static, zernike act_dim = 6, obs_dim = 2,
    # strehl_ratio, 30-step episodes, at N = 128 and at the reference's N = 240 — 1 process, 1 BLAS thread, a few seconds each
    c1 = dict(w, act_type="zernike", act_dim=6, obs_dim=2, rew_type="strehl_ratio", timesteps_per_episode=30, atm_fried=0.20)
    for n_c1 in (128, 240):
        rate, _, n_done = _run_cpu_workers(dict(c1, n_pupil=n_c1), 1, 1, max(2.0, budget_s / 2))
        modes[f"configs0_n{n_c1}_1proc_1thread"] = rate
    return {"value": allc, "unit": "env-steps/s", "cores": int(allp), "kind": "port", "modes": modes,
            "cpu_model": _cpu_model(), "os_cpu_count": int(total), "usable_cores": int(usable), "share_cores": int(share), "numpy": np.__version__,
            "sample": f"single-env steps of the float64 numpy restatement of the literal HCIPy dataflow (not HCIPy itself) at N={w['n_pupil']}, "
                      f"A={w['act_dim']}, o={w['obs_dim']}, {w['rew_type']}: {n1} steps in {budget_s:.0f} s (1 process, 1 BLAS thread), {n2} steps in "
                      f"{budget_s:.0f} s (1 process, {dthreads} BLAS threads), {n3} steps in {budget_s:.0f} s ({share} processes x 1 thread = the host "
                      f"share of one GPU of this pool), {nall} steps in {budget_s:.0f} s ({allp} processes x 1 thread = min(affinity, cgroup quota"
                      f"{', memory' if allp < usable else ''}) = `value`; the OS reports {total} logical cores)"}


def parity_check(env, w, actions, torch):
    """Strehl / obs error of the device path vs the CPU oracle on 2 envs at full size (same screens, same action).  Quasi-static
    handles only (the screens are read back from the handle).  ``obs_rel_err`` is the tests' measure (relative, with the absolute floor
    of 1e-3 x the vector's peak below which an element is held to an absolute bound: tests/test_gpu_parity.py::_assert_obs_close);
    ``obs_rel_err_pure`` is the worst |device - oracle| / |oracle| with no floor at all, ``obs_floor_fraction`` the share of elements
    that sit below the floor (where the two measures differ)."""
    import numpy as np

    from oracle.ao_env_oracle import AOEnvOracle

    env.reset()
    _, _, _, _, info = env.step(actions)
    out = {"strehl_abs_err": 0.0, "obs_rel_err": 0.0, "obs_rel_err_pure": 0.0, "obs_floor_fraction": 0.0, "envs": 2}
    below, count = 0, 0
    for b in (0, env.num_envs - 1):
        screen = env.get_screens(b, 1)[0].cpu().numpy().ravel()      # the stored screen, hcipy's unit (phase * lambda)
        ref = AOEnvOracle(atm_type="quasi_static", atm_fried=w["atm_fried"], act_type=w["act_type"], act_dim=w["act_dim"], obs_dim=w["obs_dim"],
                          rew_type=w["rew_type"], timesteps_per_episode=w["timesteps_per_episode"], num_pupil_pixels=w["n_pupil"],
                          screen=screen, verbose=False)
        ref.reset()
        ref.step(actions[b].cpu().numpy())
        if w["rew_type"] == "strehl_ratio":
            out["strehl_abs_err"] = max(out["strehl_abs_err"], abs(float(info["strehl"][b]) - ref.last_strehl))
        o, r = info["obs_raw"][b].double().cpu().numpy(), ref.last_obs_raw
        out["obs_rel_err"] = max(out["obs_rel_err"], float(np.max(np.abs(o - r) / np.maximum(np.abs(r), 1e-3 * r.max()))))
        out["obs_rel_err_pure"] = max(out["obs_rel_err_pure"], float(np.max(np.abs(o - r) / np.abs(r))))
        below += int(np.sum(np.abs(r) < 1e-3 * r.max()))
        count += int(r.size)
    out["obs_floor_fraction"] = below / max(1, count)
    return out


def roofline_dominant(env, w, kernels, steps_range):
    """Rooflines of the kernels that dominate the workloads other than config 2, from HIP events taken inside the timed region (every
    launch of these kernels is bracketed on its stream while profiling is on: ``aog_profile_read_kernel``).
      screen synthesis passes (config 3)  instruction issue: issue cycles per env from the tracked PMC summary (a citation, like
                                          ``roofline.traffic``) / measured time, against 1024 SIMDs x 2.4 GHz
      screen packing (config 3)           HBM: 4 N^2 read + 4 n_ap written per env
      extrusion (config 4)                float64 matrix cores: 2 N (nz + N) flop per one-pixel shift of one env (new = A z + B n)
      Shack-Hartmann passes (config 5)    HBM: the layout's bytes per env (separable two-pass form: 20 N^2; three-pass form: 68 N^2)"""
    import numpy as np

    N, B = w["n_pupil"], env.num_envs
    out = {}
    cite = {}
    try:
        cite = json.load(open(os.path.join(ROOT, "profiles", "reset_pmc_latest.json")))
    except Exception:
        pass
    for name, kname in (("screen_rows", "k_screen2_rows"), ("screen_cols", "k_screen2_cols")):
        if name in kernels:
            ms, n = kernels[name]
            e = {"kernel": kname, "bound": "valu_issue", "ms": ms, "launches": n, "us_per_env": ms * 1e3 / B}
            c = cite.get(kname)
            if c and c.get("issue_cycles") and cite.get("envs_per_launch"):
                cyc = c["issue_cycles"] / cite["envs_per_launch"] * B
                e.update(achieved=cyc / (ms * 1e-3), peak=ISSUE_CYCLES_PEAK, unit="SIMD issue cycles/s", frac=cyc / (ms * 1e-3) / ISSUE_CYCLES_PEAK,
                         note="issue cycles per env from profiles/reset_pmc_latest.json (PMC citation), time from this run's HIP events")
            out[name] = e
    if "pack" in kernels:
        ms, n = kernels["pack"]
        by = (4.0 * N * N + 4.0 * env.info.n_ap_padded) * B
        out["pack"] = {"kernel": "k_screen_means + k_pack_tiles", "bound": "hbm", "ms": ms, "launches": n, "achieved": by / (ms * 1e-3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "algorithmic bytes: 4 N^2 read + 4 n_ap written per env (the kernels read the screen twice: mean, then conversion)"}
    if "extrude" in kernels and getattr(env, "_layer", None) is not None:
        from adaptive_optics_gym_amd.atmosphere_host import integer_shifts

        ms, n = kernels["extrude"]
        shifts = 0
        int8_mfma = [0.0, 0.0]   # v_mfma_i32_32x32x32_i8 of the x and of the y phase (int8 composite form), summed over the timed steps
        f64_flop = 0.0      # flops of the same products in plain float64 arithmetic
        n_steps = max(1, steps_range[1] - steps_range[0])
        i8 = getattr(env, "extrusion_kmax", 0) > 0
        Np = -(-N // 64) * 64
        for t in range(steps_range[0], steps_range[1]):
            sh = np.abs(integer_shifts(env.velocity_vectors, t * env.delta_t, (t + 1) * env.delta_t, env.params.pupil_pixel))
            shifts += int(sh.sum())
            if i8:
                for phase, axis in ((0, 1), (1, 0)):          # x shifts use the horizontal operator
                    # envs packed into 64-env tiles, most shifts first; a tile runs the row blocks of its first env's shift count; the rows of
                    # shift j read KsA_j steps of stencil + j Np / 32 of normals, padded to a multiple of four 32-deep steps
                    ks = np.sort(sh[:, phase][sh[:, phase] > 0])[::-1]
                    for t0 in range(0, ks.size, 64):
                        for j in range(1, int(ks[t0]) + 1):
                            steps = -(-(-(-env.extrusion_union[axis][j - 1] // 32) + j * (Np // 32)) // 4) * 4
                            int8_mfma[phase] += 19.0 * (Np // 32) * steps * 2
                    for k in range(1, env.extrusion_kmax + 1):
                        cnt = int((sh[:, phase] == k).sum())
                        f64_flop += 2.0 * cnt * sum(N * (env.extrusion_union[axis][j - 1] + j * N) for j in range(1, k + 1))
        nz = int(max(env._layer["stencil_vertical"].size, env._layer["stencil_horizontal"].size))
        if i8:
            ahead = not os.environ.get("AOG_X8_NO_PLAN_AHEAD") and not os.environ.get("AOG_X8_NO_PHASE_AHEAD")
            # 32 x 32 x 32 multiply-adds = 65 536 integer operations per instruction; with the x phase run ahead only the y phase is inside `ms`
            ops = (int8_mfma[1] if ahead else sum(int8_mfma)) / n_steps * 65536.0
            out["extrude"] = {"kernel": "k_x8_prepare + k_x8_product of the y phase (plan and x phase: ahead, on a side stream beside the previous step's fused kernel)" if ahead
                              else "k_x8_plan + k_x8_prepare x 2 + k_x8_product x 2", "bound": "mfma_i8", "ms": ms, "launches": n,
                              "total_tops_per_step": sum(int8_mfma) / n_steps * 65536.0 / 1e12,
                              "achieved": ops / (ms * 1e-3) / 1e12, "peak": 2 * F16_MFMA_PEAK_TFLOPS, "unit": "TOP/s",
                              "frac": ops / (ms * 1e-3) / 1e12 / (2 * F16_MFMA_PEAK_TFLOPS),
                              "equivalent_float64_tflops": f64_flop / n_steps / (ms * 1e-3) / 1e12 * (ops / max(1.0, sum(int8_mfma) / n_steps * 65536.0)),
                              "shifts_per_env_step": shifts / max(1, B * n_steps),
                              "note": "int8 composite extrusion: digit products as issued (19 v_mfma_i32_32x32x32_i8 per 32 x 32 tile and 32-deep step, "
                                      "partly filled env tiles included) against the dense int8 matrix peak (2 x the f16 peak); ms = mean duration of the "
                                      "sampled steps' in-line launches together (HIP events around one block of 8 steps in 8; by default the y phase, "
                                      "including any wait for the x phase made ahead; AOG_X8_NO_PLAN_AHEAD=1 puts all five launches in line and in `ms`); "
                                      "achieved counts the products of the launches inside ms; equivalent_float64_tflops = "
                                      "the same composite products counted as float64 flops (float64 matrix peak: 78.6)"}
        else:
            flop = 2.0 * N * (nz + N) * shifts / n_steps          # per step (= per launch), mean over the timed region
            out["extrude"] = {"kernel": "k_extrude16_split", "bound": "mfma_f64", "ms": ms, "launches": n, "achieved": flop / (ms * 1e-3) / 1e12,
                              "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                              "shifts_per_env_step": shifts / max(1, B * n_steps),
                              "note": "2 N (nz + N) flop per one-pixel shift of one env, shifts recomputed on the host for the timed steps (mean per "
                                      "step); ms = mean duration of the sampled launches (HIP events around one block of 8 steps in 8)"}
    if all(k in kernels for k in ("sh_rows_fwd", "sh_cols")):
        three = "sh_rows_inv" in kernels                      # (transfer functions that do not factorise keep the three-pass form)
        names = ("sh_rows_fwd", "sh_cols", "sh_rows_inv") if three else ("sh_rows_fwd", "sh_cols")
        ms = sum(kernels[k][0] for k in names)
        by = (68.0 if three else 20.0) * N * N * B
        issue = None
        try:   # PMC citation (profiles/sh_pmc_latest.json: instruction counts at the same pupil size): the separable passes are bound by vector issue
            c = json.load(open(os.path.join(ROOT, "profiles", "sh_pmc_latest.json")))
            if not three and c.get("n_pupil") == N and c.get("envs_per_launch"):
                cyc = sum(c[k]["issue_cycles"] for k in ("k_sh_rows_sep", "k_sh_cols_sep")) / c["envs_per_launch"] * B
                issue = {"achieved": cyc / (ms * 1e-3), "peak": ISSUE_CYCLES_PEAK, "unit": "SIMD issue cycles/s", "frac": cyc / (ms * 1e-3) / ISSUE_CYCLES_PEAK,
                         "note": "issue cycles per env from profiles/sh_pmc_latest.json (PMC citation), time from this run's HIP events"}
        except Exception:
            issue = None
        out["shack_hartmann"] = {"kernel": "k_sh_rows_fwd + k_sh_cols + k_sh_rows_inv" if three else "k_sh_rows_sep + k_sh_cols_sep", "bound": "hbm", "ms": ms,
                                 "valu_issue": issue,
                                 "launches": kernels["sh_cols"][1],
                                 "per_pass_ms": {k: kernels[k][0] for k in ("sh_field", "sh_rows_fwd", "sh_cols", "sh_rows_inv") if k in kernels},
                                 "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": ("three-pass layout bytes per env: phase grid 4 N^2 read, F1T and GT (2N x N complex64 each) written and read once" if three else
                                          "separable two-pass layout bytes per env: phase grid 4 N^2 read, the N x N complex64 intermediate written and read once")}
    return out


def fused_at_b4096(env, w, device, torch, steps=120):
    """The fused kernel at config 2's shape on a working set the Infinity Cache cannot hold (B = 4096: screens + tables 0.9 GB = 3.5 x 256 MiB;
    every launch re-reads all of it from HBM): its roofline fraction there, next to the headline's B = 1024 figure (whose 0.29 GB working set
    is partly served on-die).  A second handle with the first one's host tables; 120 causal steps, every fused launch timed."""
    from adaptive_optics_gym_amd import BatchedAOEnv

    B4 = 4096
    T = w["timesteps_per_episode"]
    e4 = BatchedAOEnv(B4, device, atm_type=w["atm_type"], atm_vel=w["atm_vel"], atm_fried=w["atm_fried"], act_type=w["act_type"],
                      act_dim=w["act_dim"], obs_dim=w["obs_dim"], rew_type=w["rew_type"], timesteps_per_episode=T, num_pupil_pixels=w["n_pupil"],
                      seed=1234, screen_source="device", screen_oversampling=16, verbose=False, tables=env.tables)
    e4.persistent_outputs(True)
    a4 = torch.randn((T, B4, w["act_dim"]), device=device, generator=torch.Generator(device).manual_seed(11)) * (0.5 ** 0.5)
    e4.reset()
    for t in range(T):
        e4.step(a4[t])
    e4.reset()
    torch.cuda.synchronize()
    e4.profile(True, every=1, block=8)
    t0 = time.perf_counter()
    for i in range(steps):
        e4.step(a4[i % T])
        if (i + 1) % T == 0:
            e4.reset()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = e4.profile_read()
    e4.profile(False)
    by, _, lay = algorithmic_per_step(w["n_pupil"], w["act_dim"], w["obs_dim"], B4, e4.tables.n_ap)
    out = {"batch": B4, "kernel_ms": ms, "launches_timed": n, "achieved": by * B4 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": by * B4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_layout": lay * B4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "env_steps_per_sec": B4 * steps / dt, "bytes_per_env_step": by,
           "note": "same kernel, 0.9 GB working set (3.5 x the Infinity Cache): every timed launch streams it from HBM"}
    e4.close()
    del e4
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """--gpus N without a launcher: start the N ranks as children (this process has not touched the GPU and never will)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live and not rc:            # poll every rank: a failed one must not leave its siblings waiting in a collective until it times out
        time.sleep(0.05)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    if rc:
        for p in live:
            p.terminate()
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: a rank exited with status {rc}; the other ranks were stopped", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 episodes timed after 10 of warm-up (~0.15 s of device time at config 2).  A process's first few hundred steps run
    # ~5 % slower than steady state (device clocks), so short runs under-report: see SPINUP_STEPS
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "mfma", "valu"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU env batch (tests / rehearsals only; the JSON line reports it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-spinup", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the second timed region (aog_step_pipelined: `value_pipelined`) of configs 2 and 3")
    ap.add_argument("--no-b4096", action="store_true", help="config 2: skip the HBM-resident (B = 4096) measurement of the fused kernel")
    ap.add_argument("--extrusion", default="auto", choices=["auto", "f64"], help="config 4: 'f64' = the float64 round kernels (validation form) instead of the int8 composite form")
    ap.add_argument("--lookahead", action="store_true", help="config 4: launch each step's wind extrusion one step ahead on the library's own stream")
    args = ap.parse_args()
    w = dict(WORKLOADS[args.config])
    if args.batch:
        w["batch_per_gpu"] = args.batch
    if args.steps is None:
        args.steps = {2: 1500, 3: 200, 4: 300, 5: 40}[args.config]
    if args.warmup is None:
        args.warmup = {2: 300, 3: 40, 4: 60, 5: 10}[args.config]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} != WORLD_SIZE {world} (launch one rank per GPU, or drop the launcher and let "
                         f"bench.py spawn them)")

    import torch
    import torch.distributed as dist

    distributed = world > 1 or os.environ.get("AOG_FORCE_DIST") == "1"   # the latter: rehearse the RCCL path with one rank
    share = os.environ.get("AOG_BENCH_SHARE_GPU") == "1"                 # single-GPU rehearsal of --gpus N: every rank on card 0, gloo
    if not share and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) are visible")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))             # (AOG_FORCE_DIST=1 without a launcher: a one-rank group)
        os.environ.setdefault("WORLD_SIZE", str(world))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    B = w["batch_per_gpu"]
    total = world * B
    T = w["timesteps_per_episode"]
    env = BatchedAOEnv(B, device, atm_type=w["atm_type"], atm_vel=w["atm_vel"], atm_fried=w["atm_fried"], act_type=w["act_type"],
                       act_dim=w["act_dim"], obs_dim=w["obs_dim"], rew_type=w["rew_type"], timesteps_per_episode=T,
                       SH_operation=w["SH_operation"], num_pupil_pixels=w["n_pupil"], seed=1234, screen_source="device",
                       screen_oversampling=16, kernel=args.kernel, verbose=False, global_env_offset=rank * B, total_envs=total, extrusion=args.extrusion)
    # actions of the GLOBAL batch from one seed (main.py:155; cov 0.5 I, algorithm.py:107), this rank's slice kept
    agen = torch.Generator(device).manual_seed(10)
    actions = (torch.randn((T, total, w["act_dim"]), device=device, generator=agen) * (0.5 ** 0.5))[:, rank * B:(rank + 1) * B].contiguous()
    gather = EpisodeReturnGatherer(B, device, distributed, total_envs=total)
    gather.attach(env)                                            # episode returns accumulate inside the step's epilogue kernel

    if w["rollout"]:
        from adaptive_optics_gym_amd.rollout import DeviceActor, make_actor

        env.lookahead(args.lookahead)   # opt-in (rollout(lookahead=True)): the next step's extrusion beside this step's epilogue + policy query
        torch.manual_seed(10)
        actor = make_actor(w["obs_dim"] ** 2, w["act_dim"], 150, device=device)        # SAC actor, hidden 150 (main.py:170)
        dev_actor = DeviceActor(actor, seed=10, env_id_base=rank * B)
        n_o = w["obs_dim"] ** 2
        buf = {"obs": torch.empty((T + 1, B, n_o), dtype=torch.float16, device=device), "act": torch.empty((T, B, w["act_dim"]), device=device),
               "log_prob": torch.empty((T, B), device=device), "rew": torch.empty((T, B), device=device),
               "done": torch.empty((T, B), dtype=torch.bool, device=device), "mean": torch.empty((B, w["act_dim"]), device=device)}

    state = {"t": 0, "obs": None, "resets": 0}
    # `value` is measured with plain, causal aog_step (the loop every caller of the reference runs: algorithm.py:256-262).
    # aog_step_pipelined (the NEXT action handed over with the current one; bit-identical, one launch less per step, usable only by
    # open-loop callers) is timed in a second region of the same length and reported as `value_pipelined`.
    plain_capable = not w["rollout"] and not w["SH_operation"]
    if not w["rollout"]:
        env.persistent_outputs(True)   # a step's outputs live in one block of the env (nothing here keeps them past the next step): no allocation,
                                       # no new views per step — the host's cost per step drops from ~22 to ~10 us

    def start_episode():
        obs, _ = env.reset()
        gather.start_episode()
        state["t"] = 0
        state["resets"] += 1
        if w["rollout"]:
            buf["obs"][0].copy_(obs)

    def run(n_steps, pipeline=False):
        for i_run in range(n_steps):
            t = state["t"]
            if w["rollout"]:          # algorithm.py:242-270: policy query, env.step, transition stored (here: written in place)
                a, _, _ = dev_actor(buf["obs"][t], 0.5, out=(buf["act"][t], buf["log_prob"][t], buf["mean"]))
                env.step(a, out=(buf["obs"][t + 1], buf["rew"][t], buf["done"][t]))
            elif w["SH_operation"]:   # algorithm.py:253 + :262
                a, _ = env.SH_step()
                env.step(a)
            elif pipeline:
                # the actions are synthetic and known ahead: the next one is handed over with the current one (aog_step_pipelined: same
                # results bit for bit, the prologue of step t + 1 rides in the launch of step t's epilogue); an episode's last step ends the sequence
                # (and so does the last step of this call: every timed window is a self-contained sequence)
                env.step(actions[t], next_actions=actions[t + 1] if (t + 1 < T and i_run + 1 < n_steps) else None)
            else:
                env.step(actions[t])
            state["t"] = t + 1
            if t + 1 == T:                                        # lock-step episode end (AO_env.py:147)
                gather.finish_episode()                           # all-gather of per-env episode returns
                start_episode()

    # config 5 (the 'SHACK' policy, algorithm.py:252-253) runs through the rollout harness itself whenever the timed steps are whole episodes
    # (the default 40 and the driver's 20 are, T = 20): rollout(policy="shack") = reset, T x (SH_step, step), the episode's all-gather
    harness = w["SH_operation"] and not w["rollout"] and args.steps % T == 0
    if harness:
        from adaptive_optics_gym_amd.rollout import rollout as rollout_harness

        def run(n_steps, pipeline=False):   # (whole episodes: a warm-up that is not is rounded up)
            n_ep = (n_steps + T - 1) // T
            if n_ep:
                rollout_harness(env, None, episodes=n_ep, policy="shack", gatherer=gather)
                state["resets"] += n_ep
            return n_ep * T

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # a window of a few dozen steps (the driver's 20) times ONE block of 2 launches: 8 timed launches of 20 held the stream for 48 us = 3.8 %
    prof_block, prof_every = (8, PROFILE_EVERY) if args.steps >= 64 else (2, max(1, args.steps // 2))   # (the library times the MIDDLE block of each period)
    env.profile(True, every=prof_every, block=prof_block)   # switched on ahead of the warm-up: the first timed launches of a process pay ~1 ms of runtime set-up
    # Device spin-up (reported as config.spinup_steps): a process's first few hundred steps run ~5 % slower than steady state (device
    # clocks).  With a caller-chosen warm-up shorter than that, the difference is run here, ahead of the W warm-up steps, so that the K
    # timed steps measure the steady state a long-running job sees.  Config 2 only (the other configs' steps are 5-100x longer).
    spinup = max(0, SPINUP_STEPS - args.warmup) if (args.config == 2 and not args.no_spinup) else 0
    # A timed window shorter than an episode (the driver's --steps 20) holds no episode end: 300 = 10 episodes of 30 come before it.  Forcing
    # one into it was tried (the window then straddled a reset): it charges one reset — a full pass of the fused kernel for the observation
    # reset() returns — per 20 steps instead of per 30, and read 9 % BELOW the steady state of the same box (14.7 M against 16.1 M over 1500
    # steps with their 50 resets); without it the short window still reads below the long run (pipeline fill + closing synchronisation).
    start_episode()
    if spinup:
        run(spinup)
        fence()      # the first burst of launches after a LONG asynchronous run costs the host ~55 us per step instead of ~30 (measured:
                     # tools/spin_probe.py) — the W warm-up steps take that, not the K timed ones
    warm_run = run(args.warmup)
    warm_run = args.warmup if warm_run is None else warm_run
    fence()
    env.profile_read()                       # discard the warm-up's samples; timing stays on
    gather.time_collective(True)
    t_first = env.timestep                   # (python-side step counter: the extrusion's shift count of the timed region is recomputed from it)
    resets0 = state["resets"]
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    dt_local = dt
    resets_timed = state["resets"] - resets0
    steps_range = (t_first, env.timestep)
    kernel_ms, launches = env.profile_read()
    other_kernels = env.profile_kernels()    # {name: (mean ms, launches)}: reset / extrusion / Shack-Hartmann kernels timed in the same region
    coll_ms, coll_n = gather.collective_ms()
    status = env.device_status()
    # second region, same length: the pipelined call (open-loop callers only)
    dt_pipe, kernel_ms_pipe = None, None
    if plain_capable and not args.no_pipeline:
        fence()
        t0 = time.perf_counter()
        run(args.steps, pipeline=True)
        fence()
        dt_pipe = time.perf_counter() - t0
        kernel_ms_pipe, _ = env.profile_read()
        gather.collective_ms()
    env.profile(False)
    status |= env.device_status()
    rank_wall = [dt_local]
    if distributed:
        tmax = torch.tensor([dt, dt_pipe if dt_pipe is not None else 0.0], dtype=torch.float64, device="cpu" if share else device)
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)         # per-rank wall times: a bad scaling number can be read from one run
        rank_wall = [float(x[0]) for x in every]
        dt = max(rank_wall)
        if dt_pipe is not None:
            dt_pipe = max(float(x[1]) for x in every)
    if status != 0:
        raise SystemExit(f"bench.py: device status {status} (an inter-workgroup wait timed out): results invalid")

    if rank == 0:
        n_ap = env.tables.n_ap
        bytes_step, flops_step, layout_step = algorithmic_per_step(w["n_pupil"], w["act_dim"], w["obs_dim"], B, n_ap)
        if launches == 0 or kernel_ms <= 0:
            raise SystemExit("bench.py: no fused-kernel launch was timed inside the measured region")
        k_s = kernel_ms * 1e-3
        traffic, pmc = None, {}
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile) and args.config == 2 and not args.batch:
            try:
                pmc = json.load(open(tfile))
                traffic = pmc.get("hbm_bytes_per_launch")
            except Exception:
                traffic, pmc = None, {}
        ach_gbs = bytes_step * B / k_s / 1e9
        lay_gbs = layout_step * B / k_s / 1e9
        # matrix work of the fused kernel as issued: split-f16 contractions, 3 products per operand pair
        #   phase: 3 x (2 A_pad) flop per (pixel, env); tables: 2 (cos, sin) x 3 x 2 x 32 rows per (pixel, env)
        a_pad = env.info.n_modes_padded
        n_pix_pad = env.info.n_ap_padded
        mfma_flops = (3 * 2 * a_pad + 2 * 3 * 2 * 32) * n_pix_pad * env.info.num_envs_padded
        result = {
            "metric": "env_steps_per_sec", "value": world * B * args.steps / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "warmup_effective": warm_run + spinup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 storage, split-f16 MFMA, f32/f64 accumulate", "data": "synthetic",
            "config": {"workload": f"{w['name']}: {w['text']}", "batch_per_gpu": B, "global_batch": total, "n_pupil": w["n_pupil"],
                       "act_dim": w["act_dim"], "obs_dim": w["obs_dim"], "atm_type": w["atm_type"],
                       "kernel": {1: "valu", 2: "mfma"}.get(env.info.kernel, "ref"), "spinup_steps": spinup,
                       "stepping": "rollout(policy='shack') harness: reset, T x (SH_step, step), all-gather of returns" if harness else "aog_step" + ("" if plain_capable else " inside the policy / Shack-Hartmann loop"),
                       "timed_window_resets": resets_timed,   # episode ends (reset + all-gather of returns) inside the K timed steps
                       "collective_backend": (dist.get_backend() if distributed else "none (single process)"),
                       "lookahead": bool(w["rollout"] and args.lookahead),
                       "parallelism": f"envs sharded over {world} GPU(s) by global env id, no data-path collective; one all-gather of "
                                      f"episode returns per episode"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_fused_tab" if env.info.kernel == 2 else "k_fused_valu",
                         "kernel_ms": kernel_ms, "launches_timed": launches, "bytes_per_env_step": bytes_step,
                         "layout_bytes_per_env_step": layout_step, "achieved_layout": lay_gbs, "frac_layout": lay_gbs / HBM_PEAK_GBS,
                         "timed_every": prof_every, "timed_block": prof_block,
                         "f16_mfma": {"achieved": mfma_flops / k_s / 1e12, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": mfma_flops / k_s / 1e12 / F16_MFMA_PEAK_TFLOPS,
                                      "note": "matrix flops as issued (split-f16: 3 products per operand pair, padded tiles)"},
                         "valu_issue": None,   # filled below when the PMC citation is there
                         "note": "achieved = algorithmic bytes (SURVEY.md 8d: 4 N^2 + ... per env-step) x envs per launch / mean HIP-event "
                                 "duration of the fused kernel over the timed region (one block of timed_block launches in timed_every carries the two "
                                 "event records); achieved_layout = the same with the bytes the aperture-packed layout must move (4 n_ap per "
                                 "screen); traffic = PMC (2*FETCH_SIZE + WRITE_SIZE) per launch from profiles/traffic_latest.json; "
                                 "6.29 TB/s is the measured copy ceiling"},
        }
        result["timing"] = {"rank_wall_s": rank_wall, "collective_ms_total": coll_ms, "collectives": coll_n,
                            "note": "rank_wall_s: every rank's own wall time of the K timed steps (value uses the maximum); collective_ms_total: "
                                    "device time of this rank's all-gathers of episode returns inside them"}
        if dt_pipe is not None:
            result["value_pipelined"] = world * B * args.steps / dt_pipe
            result["ms_per_step_pipelined"] = dt_pipe / args.steps * 1e3
            result["pipelined_note"] = ("aog_step_pipelined over a second region of the same K steps: the next (synthetic, known) action is handed over "
                                        "with the current one, bit-identical results, one launch less per step; NOT usable by a policy in the loop, "
                                        "hence not `value`")
            if kernel_ms_pipe:
                result["roofline"]["kernel_ms_pipelined_region"] = kernel_ms_pipe
        if pmc.get("valu_insts_per_launch") and pmc.get("trans_insts_per_launch") is not None:
            tr_, mf_ = pmc["trans_insts_per_launch"], pmc.get("mfma_insts_per_launch", 0.0)
            cyc = 8 * tr_ + 8 * mf_ + 4 * (pmc["valu_insts_per_launch"] - tr_ - mf_)
            result["roofline"]["valu_issue"] = {
                "achieved": cyc / k_s, "peak": ISSUE_CYCLES_PEAK, "unit": "SIMD issue cycles/s", "frac": cyc / k_s / ISSUE_CYCLES_PEAK,
                "note": "issue cycles per launch = 8 x transcendental + 8 x matrix + 4 x other vector instructions (SQ_INSTS_VALU, "
                        "SQ_INSTS_VALU_TRANS_F32, SQ_INSTS_MFMA per launch: PMC citation from profiles/traffic_latest.json, not measured in "
                        "this run) / this run's kernel time, against 1024 SIMDs x 2.4 GHz; the binding resource of this kernel"}
        if args.config == 2 and not args.batch and world == 1 and not args.no_b4096:
            result["roofline"]["hbm_resident_b4096"] = fused_at_b4096(env, w, device, torch)
        dom = roofline_dominant(env, w, other_kernels, steps_range)
        if dom:
            result["roofline_dominant"] = dom
        if not args.no_parity and w["atm_type"] == "quasi_static" and not w["SH_operation"]:
            result["parity"] = parity_check(env, w, actions[0], torch)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
