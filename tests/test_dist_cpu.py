"""world_size-2 gloo test of the only collective on the path: the flattened gradient
all-reduce of the DDQN learn step (DDQN.average_gradients).  Averaged per-rank gradients of
two half batches must equal the single-process gradient of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG, ROOT


def _worker(rank, world, port, q):
    for p in (ROOT, PKG, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import DDQN
    torch.manual_seed(0)
    agent = DDQN.Agent(10, 3, device="cpu", make_memory=False)
    agent.qnetwork_local.dropout.p = 0.0
    rs = np.random.RandomState(5)
    B = 8
    s = torch.from_numpy(rs.rand(B, 3, 12, 12).astype(np.float32))
    a = torch.from_numpy(rs.randint(0, 4, (B, 1)))
    y = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
    half = slice(rank * B // world, (rank + 1) * B // world)
    loss = torch.nn.functional.mse_loss(agent.qnetwork_local(s[half]).gather(1, a[half]), y[half])
    agent.optimizer.zero_grad()
    loss.backward()
    DDQN.average_gradients(agent.qnetwork_local)
    got = torch.cat([p.grad.reshape(-1) for p in agent.qnetwork_local.parameters()])
    # single-process reference on the whole batch
    agent.optimizer.zero_grad()
    torch.nn.functional.mse_loss(agent.qnetwork_local(s).gather(1, a), y).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in agent.qnetwork_local.parameters()])
    q.put((rank, float((got - ref).abs().max()), float(ref.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_allreduce_equals_full_batch_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, scale in res:
        assert err <= 1e-6 * max(scale, 1.0), (rank, err, scale)


def _batch(k, B=8):
    rs = np.random.RandomState(100 + k)
    s = torch.from_numpy(rs.rand(B, 3, 12, 12).astype(np.float32))
    a = torch.from_numpy(rs.randint(0, 4, (B, 1)))
    r = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
    s2 = torch.from_numpy(rs.rand(B, 3, 12, 12).astype(np.float32))
    d = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
    return s, a, r, s2, d


def _deferred_run(agent, K, rows):
    """The order DDQN.train defines for world > 1 (DDQN.py here, `train`): policy forward, THEN the pending update of the
    previous learn step, then the next learn step with its update deferred.  Returns the policy's Q-values per step (they
    show which weights it acted on) and the final parameters."""
    import DDQN
    probe = _batch(999)[0]
    qs = []
    for k in range(K):
        qs.append(agent.qnetwork_local.infer(probe).clone())
        agent.finish_learn()
        agent.learn(tuple(t[rows] for t in _batch(k)), DDQN.GAMMA, defer=True)
    agent.finish_learn()
    flat = torch.cat([p.detach().reshape(-1) for p in agent.qnetwork_local.parameters()])
    tflat = torch.cat([p.detach().reshape(-1) for p in agent.qnetwork_target.parameters()])
    return torch.stack(qs), flat, tflat


def _make_agent():
    import DDQN
    torch.manual_seed(0)
    agent = DDQN.Agent(10, 3, device="cpu", make_memory=False)
    agent.qnetwork_local.dropout.p = 0.0
    agent.qnetwork_target.dropout.p = 0.0
    return agent


def _deferred_worker(rank, world, port, q):
    for p in (ROOT, PKG, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 8
    qs, flat, tflat = _deferred_run(_make_agent(), 4, slice(rank * B // world, (rank + 1) * B // world))
    q.put((rank, qs.numpy(), flat.numpy(), tflat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_deferred_learn_steps_equal_single_process_with_the_same_update_order():
    """VERDICT r03 weak #3 / ADVICE: with world > 1 the update of learn step k is applied after the policy forward of the
    next env step.  Two gloo ranks, each learning on its half of every batch with defer=True, must (i) agree with each other
    bit for bit, (ii) equal ONE process that runs the same order on the whole batches (averaged half-batch gradients = the
    whole-batch gradient of the mean loss), policy Q-values per step included — i.e. the policy of step k saw the weights
    of learn step k - 2's update — and (iii) differ from the plain order's policy outputs, so the test really sees the
    staleness it pins."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)                                     # replicated parameters stay identical
    import DDQN
    qs1, flat1, tflat1 = _deferred_run(_make_agent(), 4, slice(0, 8))   # one process, whole batches, the same order
    # (Adam normalises the step: a 1e-7 difference between averaged and whole-batch gradients moves a weight by a few 1e-7;
    #  a wrong update order moves it by the step size, 1e-3)
    assert np.abs(res[0][2] - flat1.numpy()).max() <= 2e-5
    assert np.abs(res[0][3] - tflat1.numpy()).max() <= 2e-5
    assert np.abs(res[0][1] - qs1.numpy()).max() <= 1e-5
    # the plain order (update applied inside learn) acts on newer weights from step 1 on, and ends at the same parameters
    # only because the probe forward does not feed back into the batches
    plain = _make_agent()
    probe = _batch(999)[0]
    qs_plain = []
    for k in range(4):
        qs_plain.append(plain.qnetwork_local.infer(probe).clone())
        plain.learn(_batch(k), DDQN.GAMMA)
    qs_plain = torch.stack(qs_plain).numpy()
    assert np.allclose(qs_plain[0], res[0][1][0], atol=1e-6) and np.allclose(qs_plain[0], res[0][1][1], atol=1e-6)
    assert np.abs(qs_plain[1] - res[0][1][1]).max() > 1e-5             # plain: step 1 already sees update 0; deferred: not yet
    assert np.allclose(qs_plain[1], res[0][1][2], atol=1e-5)           # deferred policy of step k = plain policy of step k - 1


def _failing_worker(rank, world, port):
    for p in (ROOT, PKG, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import DDQN
    agent = _make_agent()

    @DDQN.fails_the_job
    def loop():
        for k in range(1000):                                           # (never gets there: rank 1 fails in step 2)
            if rank == 1 and k == 2:
                real = agent.qnetwork_local.forward

                def boom(x):
                    raise RuntimeError("injected failure inside learn() on rank 1")
                agent.qnetwork_local.forward = boom
            agent.learn(tuple(t[rank * 4:(rank + 1) * 4] for t in _batch(k)), DDQN.GAMMA, defer=True)
    loop()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failing_rank_takes_the_job_down():
    """an exception on one rank inside learn() must end the job non-zero instead of leaving the
    peers blocked in the gradient all-reduce.  Rank 1 raises in its third learn step; rank 0 is then alone in the
    collective — both processes must be gone, non-zero, within seconds (DDQN.fails_the_job -> abort_job)."""
    import time
    world = 2
    ctx = mp.get_context("spawn")
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(max(1.0, 90 - (time.time() - t0)))
        codes = [p.exitcode for p in procs]
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()                                                # the exact processes started above
    assert all(c is not None and c != 0 for c in codes), codes
    assert time.time() - t0 < 90


def test_env_shards_use_distinct_philox_streams():
    """Host-side sharding rule: rank r owns its own envs and Philox key (seed, r) — checked on
    the oracle's Philox (the HIP path is compared with it bit-for-bit in the gpu tests)."""
    import oracle
    a = oracle.philox([0, 0, 0, 0], [0x5EED, 0])
    b = oracle.philox([0, 0, 0, 0], [0x5EED, 1])
    assert not np.array_equal(a, b)


# ---- ACKTR with one rank per GPU: the update of the global batch (gradients and K-FAC factor samples averaged over the ranks) ----
def _acktr_rollouts(T, N, W, lo, hi, seed=9):
    """Envs [lo, hi) of a fixed synthetic rollout of N envs: a rank's share, or (0, N) for the single process."""
    import ACKTR
    rs = np.random.RandomState(seed)
    S = W + 2
    obs = rs.rand(T + 1, N, 3, S, S).astype(np.float32)
    acts = rs.randint(0, 4, (T, N, 1))
    probs = rs.rand(T, N, 2).astype(np.float32)
    rets = rs.randn(T + 1, N, 1).astype(np.float32)
    noise = rs.randn(T, N, 1).astype(np.float32)
    r = ACKTR.RolloutStorage(T, hi - lo, 3, W, 2, "cpu")
    r.observations.copy_(torch.from_numpy(obs[:, lo:hi]))
    r.actions.copy_(torch.from_numpy(acts[:, lo:hi]))
    r.probs.copy_(torch.from_numpy(probs[:, lo:hi]))
    r.returns.copy_(torch.from_numpy(rets[:, lo:hi]))
    return r, torch.from_numpy(noise[:, lo:hi].reshape(-1, 1).copy())


def _acktr_update(rollouts, noise, micro_batch, updates=2):
    """`updates` ACKTR updates of a fresh, seeded Mulnet on `rollouts` with the Fisher noise given (Brain.update draws it with
    torch.randn: patched for the call).  Returns the parameters and the running factors."""
    import ACKTR
    from Net.ACNet import Mulnet
    torch.manual_seed(4)
    net = Mulnet(10)
    net.dropout.p = 0.0
    brain = ACKTR.Brain(net, acktr=True, device="cpu")
    real = torch.randn
    torch.randn = lambda *shape, **kw: noise.clone() if tuple(shape) == tuple(noise.shape) else real(*shape, **kw)
    try:
        for _ in range(updates):
            brain.update(rollouts, micro_batch)
    finally:
        torch.randn = real
    opt = brain.optimizer
    return ([p.detach().clone() for p in net.parameters()], [opt.m_aa[m].clone() for m in opt.modules], [opt.m_gg[m].clone() for m in opt.modules])


def _acktr_worker(rank, world, port, q, micro_batch):
    for p in (ROOT, PKG, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    T, N, W = 2, 8, 10
    share = N // world
    r, noise = _acktr_rollouts(T, N, W, rank * share, (rank + 1) * share)
    params, aa, gg = _acktr_update(r, noise, micro_batch)
    q.put((rank, [p.numpy() for p in params], [a.numpy() for a in aa], [g.numpy() for g in gg]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("micro_batch", [None, 4])
def test_acktr_update_over_two_ranks_equals_the_global_batch(micro_batch):
    """Two gloo ranks, each with half the envs of one rollout, run two ACKTR updates (ACKTR.py:88-159 with kfac.py:156-254): their
    weights and running Kronecker factors equal each other's and those of one process updating on the whole rollout — the
    gradient and every factor sample are means over the ranks.  micro_batch: the accumulate path (factor sums reduced once per
    update) and the immediate one (every hook's sample reduced)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 + (3 if micro_batch else 0)) % 2000
    procs = [ctx.Process(target=_acktr_worker, args=(r, world, port, q, micro_batch)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for p in (ROOT, PKG, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    r, noise = _acktr_rollouts(2, 8, 10, 0, 8)
    want = _acktr_update(r, noise, None if micro_batch is None else 2 * micro_batch)
    for kind in range(3):
        for a, b, w in zip(res[0][1 + kind], res[1][1 + kind], want[kind]):
            assert np.array_equal(a, b)                                   # the ranks hold the same numbers
            w = w.numpy()
            assert np.abs(a - w).max() <= 2e-4 * max(1.0, np.abs(w).max()), (kind, np.abs(a - w).max(), np.abs(w).max())
