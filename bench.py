#!/usr/bin/env python3
"""bench.py — env-steps/s of the HIP TRON env path on MI355X.

One "step" = one launch of the fused step + observation-encode + autoreset kernel over
every env of this rank: both players of each env move once (i.i.d. uniform actions drawn
in-kernel from Philox-4x32-10, as BASELINE.md §3 specifies), both players' int8 code-plane
observations are written, finished games are replaced by fresh ones.  Workload =
BASELINE.json configs[2]'s env side: 65 536 parallel 24x24 envs per GPU, mode=None.

Multi-GPU: envs are independent, so each rank owns its own 65 536 envs and its own Philox
stream (weak scaling, no data-path collective).  Launch as the driver does:
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "deep-q-learning_tron_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

N_ENVS = 65536          # per GPU (BASELINE.json configs[2] / [3])
WIDTH = 24
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes_per_env_step(width):
    """fused step+encode, int8 state, int8 code planes for both players:
    read G + 16, write 2G + 16  =>  3G + 32 bytes per env-step."""
    g = (width + 2) * (width + 2)
    return 3 * g + 32


def pmc_traffic(envs, width, obs, mode):
    """HBM bytes per STEP of the step kernel from the committed rocprofv3 PMC passes
    (profiles/r*_summary.json, made by scripts/pmc_summary.py: FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE, KiB -> bytes).  Only valid for the exact workload it was taken on."""
    import glob
    if (envs, width, obs, mode) != (N_ENVS, WIDTH, "codes", "none"):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    summ = json.load(open(files[-1]))
    for name, k in summ["kernels"].items():
        if "k_obs_roll" in name:                   # one launch = steps_per_launch steps of every env
            return k["hbm_bytes_per_launch"] / summ.get("steps_per_launch", 1), os.path.relpath(files[-1], ROOT)
    return None, None


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box
    shows all of the host's CPUs in the mask but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, int(os.environ.get("TRON_CPU_BASELINE_THREADS", "16")))


def cpu_baseline(width, budget_s=14.0):
    """The CPU oracle (C restatement of the reference, `kind: port`), same unit of work: step + both
    observations + autoreset, i.i.d. uniform actions.  Timed on one host core and on all the cores
    this process may use (one env per OpenMP iteration); `value` is the all-cores rate."""
    import oracle
    cores = host_cores()
    rates = {}
    for threads in ([1, cores] if cores > 1 else [1]):
        n = 4096 * (1 if threads == 1 else max(1, min(16, threads // 2)))
        oracle.set_threads(threads)
        ref = oracle.VecOracle(n, width, seed=0x5EED)
        ref.reset_all()
        ref.step(autoreset=True)                   # warm
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget_s / 2:
            ref.step(autoreset=True)
            steps += 1
        dt = time.perf_counter() - t0
        rates[threads] = (n * steps / dt, steps, n, dt)
    oracle.set_threads(1)
    v, steps, n, dt = rates[cores]
    return {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_1core": rates[1][0],
            "sample": f"{steps} steps x {n} envs {width}x{width}, mode=None, autoreset, Philox actions, "
                      f"{dt:.1f} s on {cores} host threads (oracle/libtron_oracle.so, OpenMP over envs); "
                      f"1 core: {rates[1][0] / 1e6:.2f} M env-steps/s over {rates[1][3]:.1f} s"}


def dqn_bench(args):
    """DDQN.train on VecTron + device replay: env-steps/s with the policy in the loop and transitions/s
    consumed by learn() (SURVEY.md §8(d) metric 2).  One rank per GPU; gradients all-reduced over RCCL."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        dist.init_process_group(os.environ.get("TRON_DIST_BACKEND", "nccl"))
    import DDQN
    envs, width = args.envs, args.width
    DDQN.train(n_envs=envs, width=width, steps=max(args.warmup, 4), learn_every=2, batch_size=args.batch,
               capacity=1 << 20, log_every=0)                                  # warm-up (MIOpen find, allocator)
    out = DDQN.train(n_envs=envs, width=width, steps=args.steps, learn_every=2, batch_size=args.batch,
                     capacity=1 << 20, log_every=0)
    if rank == 0:
        print(json.dumps({
            "metric": "dqn-transitions/sec", "value": out["learned_transitions_per_s"], "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "env_steps_per_s_with_policy": out["env_steps_per_s"],
            "transitions_pushed_per_s": out["transitions_pushed"] / out["seconds"],
            "config": {"workload": f"{envs} parallel {width}x{width} self-play envs per GPU, DDQN + target net, "
                                   f"1M-slot HBM replay, learn batch {args.batch} every 2 env-steps, eps-greedy "
                                   f"policy = the 7-conv CNN on f32 planes"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)     # multiples of the rollout's 64 steps per launch
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65536; --dqn: 4096)")
    ap.add_argument("--width", type=int, default=None, help="board side (default 24; --dqn: 10)")
    ap.add_argument("--obs", default="codes", choices=["codes", "planes3", "planes4"])
    ap.add_argument("--mode", default="none", choices=["none", "ice", "temper"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--incremental", action="store_true",
                    help="secondary variant: in-place observation update (only touched cells and restarted boards "
                         "are written); reported under its own label, not comparable with the default line")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "nonreversing"],
                    help="synthetic policy: i.i.d. uniform over the 4 headings (headline), or uniform over the 3 "
                         "that do not reverse the last move (longer episodes; secondary line, SURVEY.md 8(d))")
    ap.add_argument("--dqn", action="store_true",
                    help="secondary metric: DQN transitions/s of the batched DDQN trainer (BASELINE configs[1] by "
                         "default: 4096 envs 10x10; use --envs/--width for others)")
    ap.add_argument("--batch", type=int, default=4096, help="--dqn: learn batch")
    args = ap.parse_args()
    if args.dqn:
        args.envs = 4096 if args.envs is None else args.envs
        args.width = 10 if args.width is None else args.width
        return dqn_bench(args)
    args.envs = N_ENVS if args.envs is None else args.envs
    args.width = WIDTH if args.width is None else args.width

    import torch
    import torch.distributed as dist
    from tron.vec import VecTron

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # one rank per GPU; TRON_DIST_BACKEND=gloo lets several ranks share a GPU to rehearse this path
    backend = os.environ.get("TRON_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    env = VecTron(args.envs, args.width, mode=None if args.mode == "none" else args.mode, seed=0x5EED, rank=rank,
                  obs_format=args.obs, incremental=args.incremental)
    env.reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The K steps go out through tron_rollout_random: K dependent launches of the fused kernel (in-kernel
    # Philox actions, autoreset) on a side stream — the launch loop is native, not Python.
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())  # the reset above ran on the current stream; torch's side streams
    torch.cuda.synchronize()                       # do not order themselves behind it
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        if args.incremental:                       # the in-place variant has no rollout entry point: launch loop
            step = env.step_fn(autoreset=True, nonreversing=args.actions == "nonreversing")

            def run(k):
                for _ in range(k):
                    step()
        else:
            def run(k, per_step=False):
                env.rollout_random(k, nonreversing=args.actions == "nonreversing", per_step_launches=per_step)
        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        ev0.record()                               # same stream the kernels are launched on
        run(args.steps)
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        # for reference, outside the measured job: the same K steps as one launch per step (what a caller
        # that supplies actions every step gets)
        per_step_ms = None
        if not args.incremental and not os.environ.get("TRON_ROLL_PER_STEP"):
            ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(min(args.warmup, 16), True)
            ev2.record()
            run(args.steps, True)
            ev3.record()
            torch.cuda.synchronize()
            per_step_ms = ev2.elapsed_time(ev3) / args.steps
    # launches in the timed region: the rollout is persistent (<= 64 steps per launch of k_obs_roll / k_tile_roll,
    # include/tron_hip.h TRON_ROLLOUT_CHUNK); the incremental variant launches once per step
    persistent = not args.incremental and args.steps > 1 and not os.environ.get("TRON_ROLL_PER_STEP")
    chunk = int(os.environ.get("TRON_ROLL_CHUNK", "64")) if persistent else 1
    n_launches = (args.steps + chunk - 1) // chunk
    step_ms = ev0.elapsed_time(ev1) / args.steps   # per step, HIP events on the launch stream
    kern_ms = ev0.elapsed_time(ev1) / n_launches   # avg launch of the step kernel

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    if rank == 0:
        total_env_steps = args.envs * world * args.steps
        b_alg = alg_bytes_per_env_step(args.width)
        if args.obs != "codes":                    # f32 planes: 2 players x C planes x 4 B per cell
            g = (args.width + 2) ** 2
            b_alg = g + 32 + 2 * (3 if args.obs == "planes3" else 4) * g * 4
        achieved = b_alg * args.envs / (step_ms * 1e-3) / 1e9
        hbm_bytes, hbm_src = (pmc_traffic(args.envs, args.width, args.obs, args.mode)
                              if persistent and env.obs_is_state else (None, None))
        if args.incremental:
            # bytes this variant needs per env-step: state words + outputs (~70 B), 2 cells read, 8 written,
            # and both planes (2G) for the ~36 % of envs that restart under random play
            g = (args.width + 2) ** 2
            b_alg = 80 + int(0.36 * 2 * g)
            achieved = b_alg * args.envs / (step_ms * 1e-3) / 1e9
            hbm_bytes = hbm_src = None
        out = {
            "metric": "env-steps/sec" + (" (incremental observation update)" if args.incremental else "") +
                      (" (non-reversing uniform actions)" if args.actions == "nonreversing" else ""),
            "value": total_env_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": f"{args.envs} parallel {args.width}x{args.width} TRON envs per GPU, "
                                   f"mode={args.mode}, {'non-reversing ' if args.actions == 'nonreversing' else ''}random actions "
                                   f"(in-kernel Philox), autoreset, "
                                   f"obs={args.obs} for both players",
                       "envs_per_gpu": args.envs, "grid": f"{args.width}x{args.width}",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if hbm_bytes is None else hbm_bytes / (step_ms * 1e-3) / 1e9,
                         "traffic_bytes_per_step": hbm_bytes, "traffic_source": hbm_src,
                         "kernel": ("k_inc (in-place update: touched cells + restarted boards only)" if args.incremental
                                    else "k_obs / k_tile (one launch per step)" if not persistent
                                    else "k_obs_roll (persistent rollout of the observation-is-state step, int8 codes)"
                                    if env.obs_is_state else "k_tile_roll (persistent rollout, board-owning layout)"),
                         "kernel_ms": kern_ms, "launches": n_launches, "steps_per_launch": args.steps / n_launches,
                         "alg_bytes_per_env_step": b_alg,
                         "alg_bytes_per_launch": b_alg * args.envs * args.steps / n_launches},
        }
        if per_step_ms is not None:
            out["per_step_launches"] = {"ms_per_step": per_step_ms, "value": args.envs * world / (per_step_ms * 1e-3),
                                        "frac": b_alg * args.envs / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "note": "same K steps, one kernel launch per step (k_obs / k_tile), this rank"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.width)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
