"""One rank of the multi-GPU rehearsal (started by tests/test_gpu_dist.py as a fresh child process with
RANK / WORLD_SIZE / MASTER_* set, before anything touches the GPU).  TEST INFRASTRUCTURE: it imports the
oracle to check this rank's env shard.

What a rank does — exactly what it does on an 8-GPU node, except that TRON_DIST_BACKEND=gloo lets the
ranks share one GPU:
  1. joins the process group;
  2. builds its env shard VecTron(seed, rank=RANK), steps it with in-kernel Philox actions and checks it
     against the oracle on Philox key (seed, stream=RANK) — rank-own random streams;
  3. runs DDQN.train (own envs, own replay shard, gradients averaged across ranks each learn step);
  4. runs ACKTR.train the same way (gradients and K-FAC factor samples averaged over the ranks);
  5. writes its observations and its network weights to <out>/rank<r>.npz for the parent to compare.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_dir):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TRON_DIST_BACKEND", "nccl")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend)
    import oracle
    import DDQN
    from tron.vec import VecTron

    N, W, K, seed = 512, 10, 6, 0x5EED
    env = VecTron(N, W, seed=seed, rank=rank, obs_format="codes")
    ref = oracle.VecOracle(N, W, seed=seed, stream=rank)
    env.reset()
    ref.reset_all()
    same = True
    for _ in range(K):
        obs, r, d, w = env.step()
        o, dd, ww, rr = ref.step(autoreset=True)
        same &= bool(np.array_equal(obs.cpu().numpy().reshape(N, 2, -1), o) and np.array_equal(d.cpu().numpy(), dd))
    obs_np = env.obs.cpu().numpy().copy()
    env.close()

    out = DDQN.train(n_envs=N, width=W, steps=12, learn_every=2, batch_size=64, capacity=1 << 14, log_every=0, seed=seed)
    brain = out["brain"]
    # the side-stream branch of learn(defer=True) (world > 1: the all-reduce on its own stream, the update applied
    # by finish_learn) against learn(defer=False) on the same batches — same parameters, on every rank.
    import copy
    brain.finish_learn()
    mem, brain.memory, brain._side = brain.memory, None, None            # (the ring's handle and the stream are not copyable; both agents share the ring)
    twin = copy.deepcopy(brain)
    brain.memory = mem
    twin.optimizer = type(brain.optimizer)(twin.qnetwork_local.parameters())
    twin.optimizer.load_state_dict(copy.deepcopy(brain.optimizer.state_dict()))   # (load_state_dict keeps tensors that already match: without the copy both optimizers would update ONE set of moments)
    twin.memory = brain.memory
    twin._side = None
    for net in (brain.qnetwork_local, twin.qnetwork_local):
        net.dropout.p = 0.0
    deferred_equal, diag = True, []
    for k in range(3):
        batch = brain.memory.sample_codes()
        l1 = brain.learn(batch, DDQN.GAMMA, defer=True)
        assert brain._pending and brain._pending_on_side                  # the branch under test
        brain.finish_learn()
        l2 = twin.learn(batch, DDQN.GAMMA, defer=False)
        deferred_equal &= bool(torch.equal(l1, l2))
        diag.append((float(l1), float(l2), max(float((p - q).abs().max()) for p, q in
                                               zip(brain.qnetwork_local.parameters(), twin.qnetwork_local.parameters()))))
    deferred_equal &= all(bool(torch.equal(p, q)) for p, q in zip(brain.qnetwork_local.parameters(), twin.qnetwork_local.parameters()))
    print("deferred vs plain (loss, loss, max |dparam|):", diag, flush=True)
    flat = torch.cat([p.detach().reshape(-1) for p in brain.qnetwork_local.parameters()]).cpu().numpy()
    tflat = torch.cat([p.detach().reshape(-1) for p in brain.qnetwork_target.parameters()]).cpu().numpy()
    # each rank's replay shard holds its own transitions
    brain.memory.sample()
    idx = brain.memory.memory.last_indices(64).cpu().numpy()
    # 5. the ACKTR trainer the same way: own env shard, one net, gradients and K-FAC factor samples averaged over the ranks
    import ACKTR
    acktr_games = []
    ao = ACKTR.train(n_envs=64, width=10, model="mul", reward="3", iterations=2, acktr=True, seed=seed, micro_batch=128,
                     trace=lambda kind, it, *a: acktr_games.append(a[1].cpu().numpy().copy()) if kind == "step" else None)
    aflat = torch.cat([p.detach().reshape(-1) for p in ao["brain"].actor_critic.parameters()]).cpu().numpy()
    opt = ao["brain"].optimizer
    afac = torch.cat([opt.m_aa[m].reshape(-1) for m in opt.modules] + [opt.m_gg[m].reshape(-1) for m in opt.modules]).cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), obs=obs_np, acktr_weights=aflat, acktr_factors=afac, acktr_actions=np.stack(acktr_games),
             acktr_finite=bool(np.isfinite(aflat).all()), oracle_equal=same, local=flat, target=tflat,
             learn_steps=out["learn_steps"], deferred_equal=deferred_equal, deferred_diag=np.array(diag), games=out["games"], env_steps=out["env_steps"], world=world, idx=idx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
