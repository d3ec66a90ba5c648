#!/usr/bin/env python3
"""bench.py — env-steps/s of the HIP TRON env path on MI355X (+ DQN transitions/s beside it).

One "step" = one pass of the fused step + observation-encode + autoreset path over every env of this
rank: both players of each env move once (i.i.d. uniform actions drawn in-kernel from Philox-4x32-10, as
BASELINE.md §3 specifies), both players' int8 code-plane observations are written, finished games are
replaced by fresh ones.  Workload = BASELINE.json configs[2]'s env side: 65 536 parallel 24x24 envs per
GPU, mode=None.

The timed region (K steps, bracketed by barrier + synchronize, MAX over ranks) is repeated R times
(--repeats, default 5); `value` is the median repeat, min / max are in the line.  Order: W warm-up steps, the
K steps timed as they are then (`cold_start`: the chip was idle before the warm-up), --settle-steps further
untimed steps (the same launches, back to back), then the R timed regions of the headline.  Further records ride
on the same line: `per_step_launches` (the same K steps as one kernel launch per step — what a caller that
supplies actions every step gets) and `dqn` (DDQN.train at BASELINE configs[1]: env-steps/s with the
epsilon-greedy CNN policy in the loop and transitions/s consumed by learn()).

Multi-GPU: envs are independent, so each rank owns its own 65 536 envs and its own Philox stream (weak
scaling, no data-path collective); the DQN record adds the one real exchange, the gradient all-reduce.
Launch as the driver does:
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
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
HBM_COPY_GBS = 6290.0   # the same guide's measured float4 copy rate: what a pure stream achieves
F32_MATRIX_PEAK_TFLOPS = 157.3   # fp32-in MFMA = fp32 vector peak (guide, chip-level parameters)
F16_MATRIX_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak (guide); the split-f16 convolutions spend 3 f16 MFMAs per f32 one
# the reference's own Python `Game.step` loop, one thread, survey container (the reference
# cannot travel to the GPU box, so these are recorded figures, not re-timed here)
REFERENCE_PYTHON_RECORDED = {"10x10": 3.5e3, "24x24": 0.9e3, "32x32": 0.5e3,
                             "source": "BASELINE.md §2 (8 vCPU Xeon 2.1 GHz, 1 Python thread, i.i.d. uniform actions)"}


def alg_bytes_per_env_step(width):
    """fused step+encode, int8 state, int8 code planes for both players:
    read G + 16, write 2G + 16  =>  3G + 32 bytes per env-step."""
    g = (width + 2) * (width + 2)
    return 3 * g + 32


def env_kernel_source_sha():
    """sha256 (first 16 hex digits) of the env kernels' sources: a committed PMC record names the sources it was taken on,
    so a record of an older kernel is never quoted for the current one."""
    import hashlib
    h = hashlib.sha256()
    for f in ("tron_env.hip", "tron_device.hpp"):
        h.update(open(os.path.join(PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(envs, width, obs, mode, variant, steps_per_launch):
    """HBM bytes per STEP of the persistent rollout kernel from the committed rocprofv3 PMC passes
    (profiles/r*_rollout_pmc.json, made by scripts/rollout_pmc.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE /
    --kernel-trace runs of `bench.py --only-rollout`: every dispatch of the timed class is the same launch; FETCH_SIZE x2 per
    the gfx950 correction, KiB -> bytes).  Returned only for the exact workload, variant ("plain" / "resident") and
    steps-per-launch the passes were taken on (a persistent launch of another length has another L2 / Infinity Cache
    carry-over between its steps) AND only when the record names the CURRENT kernel sources."""
    import glob
    import re
    paths = glob.glob(os.path.join(ROOT, "profiles", "r*_rollout_pmc*.json"))
    paths.sort(key=lambda q: int(re.match(r"r(\d+)_", os.path.basename(q)).group(1)), reverse=True)
    sha = env_kernel_source_sha()
    stale = None
    for path in paths:
        summ = json.load(open(path))
        for rec in summ.get("records", []):
            if (rec["envs"], rec["width"], rec["obs"], rec["mode"], rec["variant"], int(rec["steps_per_launch"])) != \
                    (envs, width, obs, mode, variant, int(steps_per_launch)):
                continue
            if summ.get("env_kernel_source_sha16") != sha:
                stale = stale or f"{os.path.relpath(path, ROOT)} is of other kernel sources ({summ.get('env_kernel_source_sha16')} != {sha}): not quoted"
                continue
            return rec["hbm_bytes_per_launch"] / rec["steps_per_launch"], os.path.relpath(path, ROOT)
    return None, stale


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
            "reference_python_recorded": dict(REFERENCE_PYTHON_RECORDED,
                                              value=REFERENCE_PYTHON_RECORDED.get(f"{width}x{width}")),
            "sample": f"{steps} steps x {n} envs {width}x{width}, mode=None, autoreset, Philox actions, "
                      f"{dt:.1f} s on {cores} host threads (oracle/libtron_oracle.so, OpenMP over envs); "
                      f"1 core: {rates[1][0] / 1e6:.2f} M env-steps/s over {rates[1][3]:.1f} s"}


def net_forward_flops(width, in_channels=3):
    """FLOPs (2 x MACs) of one DQNNet.Net forward per sample (Net/DQNNet.py:33-63): 36.1 MFLOP at 12x12."""
    from Net.DQNNet import conv7_side
    s = width + 2
    px = s * s
    f = 2 * px * (32 * in_channels * 9 + 2 * 32 * 32 * 9 + 64 * 32 * 9 + 2 * 64 * 64 * 9)
    o = conv7_side(s)
    f += 2 * o * o * 64 * 64 * 49
    f += 2 * (64 * o * o * 256 + 256 * 128 + 128 * 64 + 64 * 4)
    return f


def max_over_ranks(x, world):
    """MAX of a host float over all ranks (gloo rehearsal: through a CPU tensor)."""
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dqn_record(envs, width, steps, warmup, batch, repeats, world, rank):
    """DDQN.train on VecTron + the HBM replay ring (BASELINE configs[1] by default): env-steps/s with the
    epsilon-greedy policy in the loop and transitions/s consumed by learn() (SURVEY.md §8(d) metric 2).
    One rank per GPU; every learn step all-reduces the flattened gradient when world > 1."""
    import torch
    import torch.distributed as dist
    import DDQN
    out = DDQN.train(n_envs=envs, width=width, steps=max(warmup, 4), learn_every=2, batch_size=batch,
                     capacity=1 << 20, log_every=0)                            # warm-up (MIOpen find, allocator)
    brain = out["brain"]
    runs = []
    for _ in range(repeats):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        o = DDQN.train(n_envs=envs, width=width, steps=steps, learn_every=2, batch_size=batch, capacity=1 << 20,
                       log_every=0, brain=brain)
        runs.append((max_over_ranks(o["seconds"], world), o))
    runs.sort(key=lambda x: x[0])
    sec, o = runs[len(runs) // 2]
    # the policy alone: greedy CNN forward over 2N observations + env step, no learning (what evaluation /
    # rating sweeps run, play.py:72-98) — csrc/tron_conv.hip trunk, conv1 straight from the int8 codes
    from tron.vec import VecTron
    env = VecTron(envs, width, seed=0x5EED, rank=rank, obs_format="codes")
    S = width + 2
    codes = env.reset().reshape(2 * envs, S, S)
    pol = []
    for rep in range(repeats + 1):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            a = brain.qnetwork_local.infer(codes, codes=True, greedy=True).reshape(envs, 2)
            codes = env.step(a)[0].reshape(2 * envs, S, S)
        torch.cuda.synchronize()
        if rep:
            pol.append(max_over_ranks(time.perf_counter() - t0, world))
    env.close()
    pol_s = statistics.median(pol)
    learned = o["learn_steps"] * batch * world
    f_fwd = net_forward_flops(width)
    # the learner alone, saturated: learn steps back to back on the warm ring (sample as int8 codes -> forward + backward
    # of the local net, two target forwards, Adam, soft update, gradient all-reduce when world > 1) — what learn() can
    # consume when the env side is not what paces it (the paced loop above learns `batch` of every
    # 2 x envs x learn_every pushed transitions)
    sat_steps = max(10, 2 * steps)
    for _ in range(3):
        brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
    sat = []
    for _ in range(repeats):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(sat_steps):
            brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
        torch.cuda.synchronize()
        sat.append(max_over_ranks(time.perf_counter() - t0, world))
    sat_s = statistics.median(sat)
    sat_flops = f_fwd * batch * (3 + 2) * sat_steps * world
    learner_saturated = {
        "metric": "dqn-transitions/sec, learn() back to back on a warm replay ring (no env stepping in between)",
        "value": sat_steps * batch * world / sat_s, "unit": "transitions/s", "learn_steps": sat_steps,
        "ms_per_learn_step": sat_s / sat_steps * 1e3, "learn_batch": batch, "replay_slots_filled": len(brain.memory),
        "roofline": {"bound": "mfma", "unit": "TFLOP/s (f32-equivalent)", "peak": F16_MATRIX_PEAK_TFLOPS / 3,
                     "achieved": sat_flops / sat_s / 1e12, "frac": sat_flops / sat_s / 1e12 / (F16_MATRIX_PEAK_TFLOPS / 3),
                     "flops_per_learn_step": f_fwd * batch * 5,
                     "note": "forward + backward of the local net on s (3 x forward), forward-only local and target on s' "
                             "(weight-stationary chain from the int8 codes)"}}
    # per iteration of the loop: the policy forward on 2N observations (eval, no grad); per learn step on a
    # batch B: forward + backward of the local net on s (~3x forward), forward-only local and target on s'
    flops = f_fwd * (2 * envs * steps + o["learn_steps"] * batch * (3 + 2)) * world
    return {"metric": "dqn-transitions/sec", "value": learned / sec, "unit": "transitions/s",
            "env_steps_per_s_with_policy": envs * steps * world / sec,
            "transitions_pushed_per_s": 2 * envs * steps * world / sec,
            "policy_rollout": {"metric": "env-steps/sec with the greedy CNN policy in the loop, no learning",
                               "value": envs * steps * world / pol_s,
                               "tflops": f_fwd * 2 * envs * steps * world / pol_s / 1e12,
                               "frac_of_f16x3_peak": f_fwd * 2 * envs * steps * world / pol_s / 1e12 / (F16_MATRIX_PEAK_TFLOPS / 3)},
            "steps": steps, "repeats": repeats, "seconds_min_med_max": [runs[0][0], sec, runs[-1][0]],
            "learn_batch": batch, "learn_every_env_steps": 2, "dtype": "f32",
            "replay_ratio": {"learned_per_pushed": batch / (2.0 * envs * 2),
                             "note": f"the paced loop learns {batch} of the {2 * envs * 2} transitions pushed per 2 env-steps; "
                                     "the reference learns 64 per 4 pushed (DDQN.py:73-88)"},
            "learner_saturated": learner_saturated,
            "policy_path": brain.qnetwork_local.infer_path(codes, codes=True),
            "config": {"workload": f"{envs} parallel {width}x{width} self-play envs per GPU, DDQN + target net, "
                                   f"1M-slot HBM replay, learn batch {batch} every 2 env-steps, eps-greedy policy = "
                                   f"the 7-conv CNN (Net/DQNNet.py) on the int8 observations",
                       "parallelism": f"env-shard + replay-shard x{world}, gradient all-reduce per learn step"},
            "roofline": {"bound": "mfma", "unit": "TFLOP/s (f32-equivalent)", "peak": F16_MATRIX_PEAK_TFLOPS / 3,
                         "achieved": flops / sec / 1e12, "frac": flops / sec / 1e12 / (F16_MATRIX_PEAK_TFLOPS / 3),
                         "flops_forward_per_sample": f_fwd,
                         "f32_matrix_peak": F32_MATRIX_PEAK_TFLOPS,
                         "vs_f32_matrix_peak": flops / sec / 1e12 / F32_MATRIX_PEAK_TFLOPS,
                         "note": "f32 results like the reference (Q within 1e-5).  Every product of the CNN runs on the f16 matrix "
                                 "cores as three MFMAs per f32 product (v = hi + lo 2^-11: csrc/tron_conv_f16.hip, "
                                 "tron_conv_wgrad.hip, tron_head.hip), so the ceiling is the dense f16 peak / 3 (`peak`); "
                                 "f32_matrix_peak is what exact-f32 MFMA arithmetic could reach (the loop runs above it).  "
                                 "Whole-loop time incl. env step, replay push / sample, optimizer; forward and input gradient of "
                                 "the learner's four linear layers are f32 library GEMMs (their weight gradients: "
                                 "tron_linear_wgrad; conv7: tron_gemm_f16x3 on its dense form at 12x12, tron_pool_conv7 at "
                                 "26x26); per-kernel rows: profiles/r04_learn_*_kernel_rows.txt, r04_infer_*"}}


class _RandomModel:
    """The stub policy of BASELINE.md §2's main_loop row: model.act(...) -> a uniform action, no network."""

    def __init__(self, seed=0x5EED):
        import random
        self.rng = random.Random(seed)

    def act(self, x, env=None):
        return self.rng.randrange(4)


def config1_record(games, width=10):
    """BASELINE configs[0] / SURVEY 8(d) c1: single games of the scalar facade — tron.util.make_game + Game.main_loop(model,
    pop=pop_up), two random-action players — every step a one-env launch of the HIP step kernel plus the facade's host
    round trips (state pull, map snapshot, pop_up per player).  Plumbing, not throughput: reported so that the N = 1 case
    has a measured line next to the reference's recorded 1.1 k env-steps/s."""
    import torch
    from tron import util
    model = _RandomModel()
    for _ in range(3):                                                     # warm-up: library load, first launches
        util.make_game(True, True, width=width).main_loop(model, pop=util.pop_up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = wins1 = wins2 = 0
    for _ in range(games):
        g = util.make_game(True, True, width=width)
        g.main_loop(model, pop=util.pop_up)
        steps += len(g.history) - 1
        wins1 += g.winner == 1
        wins2 += g.winner == 2
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": "env-steps/sec, scalar Game facade (BASELINE configs[0]: single 10x10 game, 2 random-action players, main_loop)",
            "value": steps / dt, "unit": "env-steps/s", "n_gpus": 1, "games": games, "env_steps": steps, "seconds": dt,
            "mean_episode_steps": steps / games, "wins_p1_p2_draw": [int(wins1), int(wins2), int(games - wins1 - wins2)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "reference_python_recorded": {"value": 1.1e3, "source": "BASELINE.md §2, Game.main_loop(stub random-action model, pop=pop_up), 10x10"},
            "config": {"workload": f"{games} sequential {width}x{width} games of tron.game.Game (one env per launch, host-side "
                                   f"history / Map mirrors as the reference keeps them), stub random-action model, pop=pop_up",
                       "parallelism": "none (N = 1 plumbing case)"},
            "note": "latency-bound by construction: each step is one tiny kernel launch plus ~6 device-to-host copies of a few hundred bytes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)     # multiples of the rollout's 64 steps per launch
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = median")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65536; --dqn: 4096)")
    ap.add_argument("--width", type=int, default=None, help="board side (default 24; --dqn: 10)")
    ap.add_argument("--obs", default="codes", choices=["codes", "planes3", "planes4"])
    ap.add_argument("--mode", default="none", choices=["none", "ice", "temper"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dqn", action="store_true", help="skip the DQN transitions/s record")
    ap.add_argument("--incremental", action="store_true",
                    help="secondary variant: in-place observation update (only touched cells and restarted boards "
                         "are written); reported under its own label, not comparable with the default line")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "nonreversing"],
                    help="synthetic policy: i.i.d. uniform over the 4 headings (headline), or uniform over the 3 "
                         "that do not reverse the last move (longer episodes; secondary line, SURVEY.md 8(d))")
    ap.add_argument("--dqn", action="store_true",
                    help="print the DQN record as its own line instead (BASELINE configs[1] by default: 4096 envs "
                         "10x10; use --envs/--width/--batch/--dqn-steps for others)")
    ap.add_argument("--batch", type=int, default=4096, help="DQN record: learn batch")
    ap.add_argument("--dqn-steps", type=int, default=40, help="DQN record: env steps per timed region")
    ap.add_argument("--acktr", action="store_true", help="print the ACKTR record instead (BASELINE configs[4]: 16 384 envs 32x32)")
    ap.add_argument("--acktr-iterations", type=int, default=5, help="ACKTR record: timed iterations (5 = 10 updates = one period of the eigendecompositions, Tf = 10)")
    ap.add_argument("--dqn3-steps", type=int, default=8, help="config-3 DQN record (65 536 envs x 24x24): env steps per timed region")
    ap.add_argument("--settle-steps", type=int, default=6400,
                    help="untimed steps (in launches of --steps) run after the --warmup steps and before the timed regions, so that "
                         "the timed K steps see the clocks of a chip under load: from an idle chip the same launch is 5-12 %% slower "
                         "for the first tens of milliseconds (scripts/roll_fixed_cost.sh).  The K steps timed right after --warmup "
                         "alone are reported beside the headline as `cold_start`; 0 = skip (the headline is then the cold figure)")
    ap.add_argument("--only-rollout", action="store_true",
                    help="profiling runs: the warm-up and the timed persistent-rollout launches only (no per-step, two-stream "
                         "or resident passes, no temper / DQN / CPU records), so that every dispatch of the step kernel in a "
                         "rocprofv3 pass is the same launch")
    ap.add_argument("--rollout-variant", default="plain", choices=["plain", "resident"],
                    help="what the timed region launches: the persistent rollout re-reading the boards every step (headline) or "
                         "keeping them resident in LDS (own byte model); 'resident' is for --only-rollout profiling runs")
    ap.add_argument("--config1", action="store_true",
                    help="BASELINE configs[0]: single 10x10 games of the scalar Game facade, two random-action players, "
                         "Game.main_loop(model, pop=pop_up) — the plumbing case (SURVEY 8(d) c1); prints its own line")
    ap.add_argument("--dqn-envs", type=int, default=4096)
    ap.add_argument("--dqn-width", type=int, default=10)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config1:
        if rank == 0:
            print(json.dumps(config1_record(max(20, min(args.steps, 2000)))), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.acktr:
        # BASELINE configs[4]: the ACKTR.py path at 32x32 boards, 16 384 envs, K-FAC natural-gradient step on PyTorch-ROCm
        import ACKTR
        envs, width = args.envs or 16384, args.width or 32
        torch.cuda.reset_peak_memory_stats()
        ACKTR.train(n_envs=envs, width=width, model="mul", reward="3", iterations=1, acktr=True, log_every=0)   # warm-up (MIOpen find)
        o = ACKTR.train(n_envs=envs, width=width, model="mul", reward="3", iterations=args.acktr_iterations, acktr=True, log_every=0)
        job_seconds = max_over_ranks(o["seconds"], world)    # one net over all ranks (gradients and K-FAC factor samples all-reduced per update)
        if rank == 0:
            from Net import activations, fused, kfac
            net = o["brain"].actor_critic
            conv = net.conv2.module if hasattr(net.conv2, "module") else net.conv2
            side = width + 2
            paths = {
                "trunk_convolutions": ("csrc/tron_conv_ws_train.hip (the updates: conv2 .. conv6 forward, input gradient and weight gradient on the "
                                       "weight-stationary chain, one autograd node; the rollouts' gradient-free forwards on the same kernels); csrc/tron_conv_f16.hip (conv1)"
                                       if activations.ac_trunk_px_supported(torch.empty(1, 3, side, side, device="cuda"),
                                                                            [getattr(net, f"conv{i}").module.weight if hasattr(getattr(net, f"conv{i}"), "module")
                                                                             else getattr(net, f"conv{i}").weight for i in range(1, 7)])
                                       else "csrc/tron_conv_f16.hip (forward, input gradient); weight gradient csrc/tron_conv_wgrad*.hip"),
                "conv7": ("csrc/tron_head.hip (tron_conv7_fwd / _bwd)" if (side // 2 in (13, 17) and activations._use_pool_conv7_cl
                                                                              and fused.default_math == "f16x3") else "MIOpen"),
                "kfac_factors": "csrc/tron_kfac_px.hip (3x3 input factors from one haloed PX16 window, no patch matrix) + csrc/tron_kfac.hip Gram kernels" if kfac.use_gram else "extract_patches + library GEMM",
                "bias_residual_activation": "in the hooked convolutions' own kernel (epilogue: activation and pre-activation; backward: mish', bias sums and the gradient's scale in one pass), K-FAC's hooks fed by hand (Net/kfac.py::SplitBias)",
                "fisher_pass": "statistics only" if o["brain"].fisher_stats_only else "full backward",
                "eigendecompositions": "torch.linalg.eigh (rocSOLVER)"}
            print(json.dumps({
                "paths": paths,
                "metric": "env-steps/sec (ACKTR trainer: 5-step A2C rollouts of both players + two K-FAC updates per iteration)",
                "value": o["env_steps"] * world / job_seconds, "unit": "env-steps/s", "n_gpus": world, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "iterations": o["iterations"], "seconds": o["seconds"],
                "kfac_update_seconds_per_iteration": o["update_seconds"] / o["iterations"],
                "rollout_seconds_per_iteration": o["rollout_seconds"] / o["iterations"],
                "samples_per_update": 5 * envs, "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
                "config": {"workload": f"{envs} parallel {width}x{width} self-play envs (temper mode), Mulnet actor-critic, "
                                       f"ACKTR: Fisher statistics every update (Ts = 1), eigendecompositions every tenth (Tf = 10, kfac.py:107-110,217: "
                                       f"a run starts at update 0: {2 * o['iterations']} timed updates hold {(2 * o['iterations'] + 9) // 10} round(s) of them), "
                                       f"micro-batches of 16 384 (69 GB of HBM at its peak)", "parallelism": f"env-shard x{world}" + ("; one net: gradients and K-FAC factor samples "
                                       "averaged over the ranks per update" if world > 1 else "")}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.dqn:
        rec = dqn_record(args.envs or args.dqn_envs, args.width or args.dqn_width, args.dqn_steps if args.steps == 320
                         else args.steps, min(args.warmup, 8), args.batch, max(1, min(args.repeats, 3)), world, rank)
        if rank == 0:
            rec.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                        "data": "synthetic"})
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    args.envs = N_ENVS if args.envs is None else args.envs
    args.width = WIDTH if args.width is None else args.width
    from tron.vec import VecTron

    env = VecTron(args.envs, args.width, mode=None if args.mode == "none" else args.mode, seed=0x5EED, rank=rank,
                  obs_format=args.obs, incremental=args.incremental)
    env.reset()

    # The K steps go out through tron_rollout_random on a side stream — the launch loop is native, not Python.
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())  # the reset above ran on the current stream; torch's side streams
    torch.cuda.synchronize()                       # do not order themselves behind it
    nonrev = args.actions == "nonreversing"
    walls, dev_ms = [], []
    per_step_ms, two_stream_ms, resident_ms = [], [], []
    obs_is_state_cfg = args.obs == "codes" and args.width % 2 == 0      # (every mode: the sliding modes keep their slide tiles in a log)
    with torch.cuda.stream(side):
        if args.incremental:                       # the in-place variant has no rollout entry point: launch loop
            step = env.step_fn(autoreset=True, nonreversing=nonrev)

            def run(k, per_step=False):
                for _ in range(k):
                    step()
        else:
            main_resident = args.rollout_variant == "resident"

            def run(k, per_step=False, two=False, resident=main_resident):
                env.rollout_random(k, nonreversing=nonrev, per_step_launches=per_step, two_streams=two, resident=resident)
        run(args.warmup)

        def timed_regions(n):
            w, d = [], []
            for _ in range(n):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                barrier()
                t0 = time.perf_counter()
                ev0.record()                       # same stream the kernels are launched on
                run(args.steps)
                ev1.record()
                barrier()
                w.append(max_over_ranks(time.perf_counter() - t0, world))
                d.append(ev0.elapsed_time(ev1))
            return w, d
        cold_walls, cold_ms, settled = None, None, 0
        if args.settle_steps > args.warmup:
            cold_walls, cold_ms = timed_regions(max(1, min(args.repeats, 3)))
            while settled < args.settle_steps:     # the same launches as the timed ones, back to back, untimed
                run(args.steps)
                settled += args.steps
        walls, dev_ms = timed_regions(max(1, args.repeats))
        # the same K steps as one launch per step (what a caller that supplies actions every step gets)
        if not args.incremental and not args.only_rollout and not os.environ.get("TRON_ROLL_PER_STEP"):
            run(min(args.warmup, 16), True)
            for _ in range(max(1, args.repeats)):
                ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev2.record()
                run(args.steps, True)
                ev3.record()
                torch.cuda.synchronize()
                per_step_ms.append(ev2.elapsed_time(ev3) / args.steps)
            # ... and as two half-batches on two streams (a caller that pipelines the halves against its policy)
            run(min(args.warmup, 16), False, True)
            for _ in range(max(1, args.repeats)):
                ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev2.record()
                run(args.steps, False, True)
                ev3.record()
                torch.cuda.synchronize()
                two_stream_ms.append(ev2.elapsed_time(ev3) / args.steps)
            # ... and the persistent rollout with the boards resident in LDS between steps (a different traffic
            # contract: 2G + 32 bytes per env-step; reported under its own label)
            if obs_is_state_cfg:
                run(args.warmup, False, False, True)
                for _ in range(max(1, args.repeats)):
                    ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev2.record()
                    run(args.steps, False, False, True)
                    ev3.record()
                    torch.cuda.synchronize()
                    resident_ms.append(ev2.elapsed_time(ev3) / args.steps)
    # launches in the timed region: the rollout is persistent (<= 64 steps per launch of k_obs_roll / k_tile_roll,
    # include/tron_hip.h TRON_ROLLOUT_CHUNK); the incremental variant launches once per step
    persistent = not args.incremental and args.steps > 1 and not os.environ.get("TRON_ROLL_PER_STEP")
    chunk = int(os.environ.get("TRON_ROLL_CHUNK", "64")) if persistent else 1
    n_launches = (args.steps + chunk - 1) // chunk
    wall = statistics.median(walls)
    ev_ms = statistics.median(dev_ms)              # HIP events on the launch stream, median repeat
    step_ms = ev_ms / args.steps
    kern_ms = ev_ms / n_launches                   # avg launch of the step kernel

    # a sliding mode beside the headline ("temper" as a secondary row): the same K steps on k_obs_roll_slide
    temper = None
    if args.mode == "none" and args.obs == "codes" and not args.incremental and not args.only_rollout and not nonrev:
        tenv = VecTron(args.envs, args.width, mode="temper", seed=0x5EED, rank=rank, obs_format="codes")
        tenv.reset()
        t_ms = []
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            tenv.rollout_random(args.warmup)
            for _ in range(max(1, min(args.repeats, 3))):
                ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                barrier()
                ev2.record()
                tenv.rollout_random(args.steps)
                ev3.record()
                barrier()
                t_ms.append(max_over_ranks(ev2.elapsed_time(ev3), world) / args.steps)
        tenv.close()
        del tenv
        tm = statistics.median(t_ms)
        t_ach = alg_bytes_per_env_step(args.width) * args.envs / (tm * 1e-3) / 1e9
        temper = {"metric": "env-steps/sec, mode='temper' (slides: game.py:96-108,163-178; in-kernel Philox uniforms)",
                  "value": args.envs * world / (tm * 1e-3), "ms_per_step": tm, "ms_per_step_min_max": [min(t_ms), max(t_ms)],
                  "roofline": {"bound": "hbm", "achieved": t_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": t_ach / HBM_PEAK_GBS,
                               "frac_of_achievable": t_ach / HBM_COPY_GBS, "alg_bytes_per_env_step": alg_bytes_per_env_step(args.width),
                               "kernel": ("k_obs_roll_slide (persistent rollout of the observation-is-state step; slide tiles in a per-env log)"
                                          if args.width % 2 == 0 else "k_tile_roll (persistent rollout, board-owning layout)")}}

    dqn = dqn3 = None
    if not args.no_dqn and not args.incremental and not args.only_rollout:
        env.close()
        del env
        env = None
        # A single process lets the headline survive a trainer-side failure (the record then carries the error).  With one
        # rank per GPU that is not an option: the rank that raised has left the per-learn-step all-reduce and its peers
        # would block there — the job ends non-zero instead (DDQN.abort_job; never a line that looks like a result).
        def guarded(*a):
            try:
                return dqn_record(*a)
            except Exception as e:
                if world > 1:
                    import DDQN
                    DDQN.abort_job(e)
                return {"error": f"{type(e).__name__}: {e}"}
        dqn = guarded(args.dqn_envs, args.dqn_width, args.dqn_steps, 6, args.batch, 3, world, rank)
        torch.cuda.empty_cache()                   # BASELINE configs[2]: 65 536 envs x 24x24, 1M-slot replay
        dqn3 = guarded(N_ENVS, WIDTH, args.dqn3_steps, 2, args.batch, 3, world, rank)
        torch.cuda.empty_cache()

    if rank == 0:
        total_env_steps = args.envs * world * args.steps
        b_alg = alg_bytes_per_env_step(args.width)
        if args.obs != "codes":                    # f32 planes: 2 players x C planes x 4 B per cell
            g = (args.width + 2) ** 2
            b_alg = g + 32 + 2 * (3 if args.obs == "planes3" else 4) * g * 4
        obs_is_state = args.obs == "codes" and args.width % 2 == 0
        if args.incremental:
            # bytes this variant needs per env-step: state words + outputs (~70 B), 2 cells read, 8 written,
            # and both planes (2G) for the ~36 % of envs that restart under random play
            g = (args.width + 2) ** 2
            b_alg = 80 + int(0.36 * 2 * g)
        achieved = b_alg * args.envs / (step_ms * 1e-3) / 1e9
        hbm_bytes, hbm_src = (pmc_traffic(args.envs, args.width, args.obs, args.mode, args.rollout_variant, args.steps / n_launches)
                              if persistent and obs_is_state else (None, None))
        out = {
            "metric": "env-steps/sec" + (" (incremental observation update)" if args.incremental else "") +
                      (" (non-reversing uniform actions)" if nonrev else ""),
            "value": total_env_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settled,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "repeats": len(walls),
            "value_min_med_max": [total_env_steps / max(walls), total_env_steps / wall, total_env_steps / min(walls)],
            "config": {"workload": f"{args.envs} parallel {args.width}x{args.width} TRON envs per GPU, "
                                   f"mode={args.mode}, {'non-reversing ' if nonrev else ''}random actions "
                                   f"(in-kernel Philox), autoreset, obs={args.obs} for both players; synthetic "
                                   f"persistent rollout (no consumer reads the intermediate observations)",
                       "envs_per_gpu": args.envs, "grid": f"{args.width}x{args.width}",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "achievable_peak": HBM_COPY_GBS, "frac_of_achievable": achieved / HBM_COPY_GBS,
                         "traffic": None if hbm_bytes is None else hbm_bytes / (step_ms * 1e-3) / 1e9,
                         "traffic_bytes_per_step": hbm_bytes, "traffic_source": hbm_src if hbm_bytes is not None else None,
                         "traffic_note": ("PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs) of this "
                                          "workload at this steps-per-launch on these kernel sources, committed under profiles/"
                                          if hbm_bytes is not None else
                                          (hbm_src or "no committed PMC pass for this workload / steps-per-launch")),
                         "kernel": ("k_inc (in-place update: touched cells + restarted boards only)" if args.incremental
                                    else "k_obs / k_tile (one launch per step)" if not persistent
                                    else ("k_obs_roll (persistent rollout of the observation-is-state step, int8 codes)" if args.mode == "none"
                                          else "k_obs_roll_slide (persistent rollout of the observation-is-state step, int8 codes; slide tiles in a per-env log)")
                                    if obs_is_state else "k_tile_roll (persistent rollout, board-owning layout)"),
                         "kernel_ms": kern_ms, "kernel_ms_min_max": [min(dev_ms) / n_launches, max(dev_ms) / n_launches],
                         "launches": n_launches, "steps_per_launch": args.steps / n_launches,
                         "alg_bytes_per_env_step": b_alg,
                         "alg_bytes_per_launch": b_alg * args.envs * args.steps / n_launches},
        }
        if cold_walls:
            cw, cm = statistics.median(cold_walls), statistics.median(cold_ms)
            c_ach = b_alg * args.envs / (cm / args.steps * 1e-3) / 1e9
            out["cold_start"] = {
                "metric": "env-steps/sec, the same K steps timed right after the --warmup steps alone (chip idle before them)",
                "value": total_env_steps / cw, "ms_per_step": cw / args.steps * 1e3, "repeats": len(cold_walls),
                "kernel_ms": cm / n_launches, "kernel_ms_min_max": [min(cold_ms) / n_launches, max(cold_ms) / n_launches],
                "roofline": {"bound": "hbm", "achieved": c_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": c_ach / HBM_PEAK_GBS},
                "note": f"the headline's timed regions follow {settled} further untimed steps (the same launches back to back): "
                        "the step kernel is HBM-bound and the same launch runs 5-12 % slower on a chip that was idle tens of "
                        "milliseconds earlier (profiles/r04_roll_fixed_cost.txt); a trainer steps continuously"}
        if per_step_ms:
            ps = statistics.median(per_step_ms)
            ps_ach = b_alg * args.envs / (ps * 1e-3) / 1e9
            out["per_step_launches"] = {
                "metric": "env-steps/sec, one kernel launch per step (policy-in-the-loop callers)",
                "ms_per_step": ps, "ms_per_step_min_max": [min(per_step_ms), max(per_step_ms)],
                "value": args.envs * world / (ps * 1e-3),
                "roofline": {"bound": "hbm", "achieved": ps_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ps_ach / HBM_PEAK_GBS, "achievable_peak": HBM_COPY_GBS,
                             "frac_of_achievable": ps_ach / HBM_COPY_GBS,
                             "kernel": "k_obs (observation-is-state step)" if obs_is_state else "k_tile"},
                "frac": ps_ach / HBM_PEAK_GBS,
                "note": "same K steps, this rank's HIP events, after the measured job"}
            if two_stream_ms:
                ts = statistics.median(two_stream_ms)
                ts_ach = b_alg * args.envs / (ts * 1e-3) / 1e9
                out["per_step_launches"]["two_streams"] = {
                    "metric": "env-steps/sec, one launch per step and per half of the envs, halves on two streams",
                    "ms_per_step": ts, "ms_per_step_min_max": [min(two_stream_ms), max(two_stream_ms)],
                    "value": args.envs * world / (ts * 1e-3), "achieved": ts_ach, "frac": ts_ach / HBM_PEAK_GBS,
                    "frac_of_achievable": ts_ach / HBM_COPY_GBS}
        if resident_ms:
            rs_ms = statistics.median(resident_ms)
            g = (args.width + 2) ** 2
            rs_ach = (2 * g + 32) * args.envs / (rs_ms * 1e-3) / 1e9
            out["resident_rollout"] = {
                "metric": "env-steps/sec (persistent rollout, boards resident in LDS between the steps of a launch)",
                "value": args.envs * world / (rs_ms * 1e-3), "ms_per_step": rs_ms,
                "ms_per_step_min_max": [min(resident_ms), max(resident_ms)],
                "roofline": {"bound": "hbm", "alg_bytes_per_env_step": 2 * g + 32,
                             "note": "both observation planes are still written every step (2G + 32 B); the G-byte state "
                                     "read per step is gone because the state never leaves the chip inside a launch",
                             "achieved": rs_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rs_ach / HBM_PEAK_GBS,
                             "achievable_peak": 6100.0, "frac_of_achievable": rs_ach / 6100.0,
                             "achievable_note": "the guide's plain-store rate 6.0-6.2 TB/s",
                             "traffic_bytes_per_step": pmc_traffic(args.envs, args.width, args.obs, args.mode, "resident",
                                                                   min(args.steps, chunk))[0],
                             "kernel": "k_obs_roll with TRON_ROLLOUT_RESIDENT"}}
        if temper is not None:
            out["temper"] = temper
        if dqn is not None:
            out["dqn"] = dqn
        if dqn3 is not None:
            out["dqn_config3"] = dqn3
        if world == 1 and not args.no_cpu_baseline and not args.only_rollout:
            out["cpu_baseline"] = cpu_baseline(args.width)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:                     # one rank per GPU: a rank that fails ends the job, non-zero (DDQN.abort_job)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not (isinstance(e, SystemExit) and not e.code):
            import DDQN
            DDQN.abort_job(e)
        raise
