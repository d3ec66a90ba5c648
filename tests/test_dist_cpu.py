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


def test_env_shards_use_distinct_philox_streams():
    """Host-side sharding rule: rank r owns its own envs and Philox key (seed, r) — checked on
    the oracle's Philox (the HIP path is compared with it bit-for-bit in the gpu tests)."""
    import oracle
    a = oracle.philox([0, 0, 0, 0], [0x5EED, 0])
    b = oracle.philox([0, 0, 0, 0], [0x5EED, 1])
    assert not np.array_equal(a, b)
