"""The multi-rank code paths executed for real, on ONE GPU: two fresh child processes (started before they
touch the GPU; TRON_DIST_BACKEND=gloo because RCCL refuses two ranks on one device) run

  (i)  `bench.py --gpus 2` exactly as the driver launches it (RANK / WORLD_SIZE / MASTER_* in the env):
       process group, barrier, per-rank env shard + Philox stream, MAX-over-ranks timing, one JSON line;
  (ii) tests/dist_rehearsal_worker.py: env shards checked against the oracle per rank, then DDQN.train with
       the gradient all-reduce of every learn step.

Everything but the transport (gloo through host memory instead of RCCL over xGMI) is the code an 8-GPU
node runs.  Reference: independent envs per worker, ACKTR.py:183,285-289; SURVEY.md §8(e)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _spawn(world, argv, extra_env=None, timeout=600):
    port = 29500 + (os.getpid() * 7 + len(argv[0])) % 3000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TRON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                   # the exact PIDs started above
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    return outs


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    outs = _spawn(2, ["bench.py", "--gpus", "2", "--steps", "64", "--warmup", "8", "--envs", "16384", "--no-dqn",
                      "--repeats", "2"])
    lines = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][1].splitlines() if ln.startswith("{")]   # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 64
    assert rec["config"]["envs_per_gpu"] == 16384 and rec["config"]["parallelism"] == "env-shard x2"
    # whole-job value = units of BOTH ranks over the max-over-ranks wall
    assert abs(rec["value"] - 2 * 16384 * 64 / (rec["ms_per_step"] * 1e-3 * 64)) <= 1e-6 * rec["value"]
    out_dir = os.environ.get("TRON_REHEARSAL_OUT")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        open(os.path.join(out_dir, "dist_rehearsal_bench.json"), "w").write(lines[0] + "\n")


@pytest.mark.timeout(900)
def test_driver_invocation_two_ranks_with_the_dqn_records():
    """The driver's own multi-rank command line (`bench.py --gpus N --steps 20 --warmup 5`: headline, cold / settled regions, temper,
    and the two DDQN records with their per-learn-step gradient all-reduce, deferred update and the one-launch optimizer) with two
    ranks on one GPU: both ranks exit 0, rank 0 prints the one line, no record carries an error."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    outs = _spawn(2, ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"], timeout=800)
    lines = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][1].splitlines() if ln.startswith("{")]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["settle_steps"] >= 6400
    assert "cold_start" in rec and "temper" in rec and "cpu_baseline" not in rec          # (the CPU baseline is an N = 1 record)
    for k in ("dqn", "dqn_config3"):
        assert "error" not in rec[k] and rec[k]["value"] > 0 and rec[k]["learner_saturated"]["value"] > 0
        assert rec[k]["config"]["parallelism"].startswith("env-shard + replay-shard x2")


@pytest.mark.timeout(900)
def test_trainer_two_ranks_on_one_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(2, [os.path.join("tests", "dist_rehearsal_worker.py"), str(tmp_path)])
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in (0, 1))
    # rank-own Philox streams: each shard equals the oracle on key (seed, rank); the shards differ
    assert bool(r0["oracle_equal"]) and bool(r1["oracle_equal"])
    assert not np.array_equal(r0["obs"], r1["obs"])
    # replicated parameters: same averaged gradient + same Adam step on every rank => identical weights
    assert int(r0["learn_steps"]) == int(r1["learn_steps"]) == 6
    # learn(defer=True) + finish_learn() through the side-stream all-reduce == learn(defer=False), on both ranks
    assert bool(r0["deferred_equal"]) and bool(r1["deferred_equal"]), (r0["deferred_diag"], r1["deferred_diag"])
    assert np.array_equal(r0["local"], r1["local"]) and np.array_equal(r0["target"], r1["target"])
    # ACKTR: one net on both ranks (averaged gradients and K-FAC factor samples => identical weights and factors), different games
    assert np.array_equal(r0["acktr_weights"], r1["acktr_weights"]) and np.array_equal(r0["acktr_factors"], r1["acktr_factors"])
    assert bool(r0["acktr_finite"]) and not np.array_equal(r0["acktr_actions"], r1["acktr_actions"])
    import DDQN
    torch.manual_seed(0x5EED)
    init = DDQN.Agent(10, 3, device="cpu", make_memory=False)
    w0 = torch.cat([p.detach().reshape(-1) for p in init.qnetwork_local.parameters()]).numpy()
    assert w0.shape == r0["local"].shape and np.abs(w0 - r0["local"]).max() > 1e-4      # and they were trained
    assert int(r0["env_steps"]) == 2 * 512 * 12 and int(r0["world"]) == 2
    out_dir = os.environ.get("TRON_REHEARSAL_OUT")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        json.dump({"ranks": 2, "backend": "gloo (two ranks sharing one MI355X)", "learn_steps": 6,
                   "weights_identical_across_ranks": True, "shards_equal_oracle_per_rank": True,
                   "max_weight_change_from_init": float(np.abs(w0 - r0["local"]).max())},
                  open(os.path.join(out_dir, "dist_rehearsal_trainer.json"), "w"))
