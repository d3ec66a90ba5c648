"""The learner fed with int8 observation codes instead of f32 planes (DDQN.py:191-200 hands over what it stored; the ring
stores codes): tron_replay_sample_codes against tron_replay_sample, Net.forward_codes against Net.forward on the planes
(values and gradients), and a whole Agent.learn() step either way."""
import copy

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import config  # noqa: F401


def _codes(B, S, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    return vals[torch.randint(0, 6, (B, S, S), device="cuda", generator=gen)]


@pytest.mark.parametrize("S,batch", [(12, 64), (12, 4096), (26, 257), (11, 33)])
def test_sample_codes_is_sample_without_the_expansion(S, batch):
    """Two rings with the same seed and contents draw the same slots; the codes rows are the planes rows' preimage."""
    import tron.vec as tv
    n = 5000
    rings = [tv.DeviceReplay(6000, S * S, seed=9, rank=1) for _ in range(2)]
    s, s2 = _codes(n, S, 1), _codes(n, S, 2)
    a = (torch.arange(n, device="cuda") % 4).to(torch.int8)
    r = torch.arange(n, device="cuda", dtype=torch.float32)
    d = (torch.arange(n, device="cuda") % 7 == 0).to(torch.int8)
    for rb in rings:
        rb.add(s, a, r, s2, d)
    for _ in range(2):
        ps, pa, pr, ps2, pd = rings[0].sample(batch, channels=3, side=S)
        cs, ca, cr, cs2, cd = rings[1].sample_codes(batch, side=S)
        assert torch.equal(rings[0].last_indices(batch), rings[1].last_indices(batch))
        assert cs.dtype == torch.int8 and cs.shape == (batch, S, S)
        assert torch.equal(tv.pop_up_planes(cs), ps) and torch.equal(tv.pop_up_planes(cs2), ps2)
        assert torch.equal(ca, pa) and torch.equal(cr, pr) and torch.equal(cd, pd)
        idx = rings[1].last_indices(batch)
        assert torch.equal(cs, s[idx]) and torch.equal(cs2, s2[idx])
    from tron import _native as nat
    with pytest.raises(nat.TronNativeError):
        rings[1].sample_codes(6001, side=S)


@pytest.mark.parametrize("W,cin,B", [(10, 3, 50), (10, 4, 129), (24, 3, 9)])
def test_forward_codes_equals_forward_on_planes(W, cin, B):
    """The same network two ways: from the int8 codes (the learner's path: `_TrunkPX`, conv1 as a table sum, PX16 images between
    the layers) and from f32 planes (`_TrunkHIP`, the layer kernels).  Two f32-grade evaluations of one expression: equal to
    rounding, for the values and for every parameter's gradient (each is held to float64 on its own in test_gpu_trunk_px.py /
    test_gpu_trunk_node.py)."""
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W + cin)
    S = W + 2
    net = Net(cin, W).cuda()
    net.dropout.p = 0.0
    codes = _codes(B, S, B)
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full_like(planes[:, :1], 5.0)], 1)
    q_c = net.forward_codes(codes, 5.0)
    q_c.square().mean().backward()
    g_c = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    q_p = net(planes)
    q_p.square().mean().backward()
    assert (q_c - q_p).abs().max().item() < 1e-5
    for a, p in zip(g_c, net.parameters()):
        assert (a - p.grad).abs().max().item() <= 2e-5 * max(p.grad.abs().max().item(), 1e-12)


def test_learn_on_codes_equals_learn_on_planes():
    """Agent.learn (DDQN.py:115-151) with the batch as codes — conv1 from the codes, both target forwards on the
    weight-stationary chain, the body as `_BodyPX` — against the same step fed with f32 planes (the layer kernels): loss within
    1e-5, updated weights to rounding (Adam's first step is lr * sign-like: 1e-3 +- a few 1e-6 where two f32-grade gradients differ)."""
    import DDQN
    from tron.vec import pop_up_planes
    torch.manual_seed(3)
    B, W = 512, 10
    base = DDQN.Agent(W, 3, device="cuda", make_memory=False)
    base.qnetwork_local.dropout.p = 0.0
    agents = [base, copy.deepcopy(base)]
    agents[1].optimizer = torch.optim.Adam(agents[1].qnetwork_local.parameters())
    s, s2 = _codes(B, W + 2, 1), _codes(B, W + 2, 2)
    a = torch.randint(0, 4, (B, 1), device="cuda")
    r = torch.randn(B, 1, device="cuda")
    d = (torch.rand(B, 1, device="cuda") < 0.3).float()
    loss_c = agents[0].learn((s, a, r, s2, d), DDQN.GAMMA)
    loss_p = agents[1].learn((pop_up_planes(s), a, r, pop_up_planes(s2), d), DDQN.GAMMA)
    assert abs(float(loss_c) - float(loss_p)) < 1e-5 * max(1.0, abs(float(loss_p)))
    for pc, pp in zip(agents[0].qnetwork_local.parameters(), agents[1].qnetwork_local.parameters()):
        assert (pc - pp).abs().max().item() < 5e-6


def test_split_push_equals_push():
    """add_states() + add(None, ...) (tron_replay_push_states: the state rows written ahead of the step that overwrites the
    observation buffer) fills the ring exactly like add(state, ...), also across the ring's wrap."""
    import tron.vec as tv
    S, cap, n = 12, 1000, 384
    rings = [tv.DeviceReplay(cap, S * S, seed=4) for _ in range(2)]
    for k in range(4):                                                   # 1 536 rows through 1 000 slots: wraps
        s, s2 = _codes(n, S, 10 + k), _codes(n, S, 20 + k)
        a = ((torch.arange(n, device="cuda") + k) % 4).to(torch.int8)
        r = torch.arange(n, device="cuda", dtype=torch.float32) + 1000 * k
        d = ((torch.arange(n, device="cuda") + k) % 5 == 0).to(torch.int8)
        rings[0].add(s, a, r, s2, d)
        rings[1].add_states(s)
        rings[1].add(None, a, r, s2, d)
        assert len(rings[0]) == len(rings[1])
    for _ in range(2):
        x = rings[0].sample_codes(777, side=S)
        y = rings[1].sample_codes(777, side=S)
        assert all(torch.equal(p, q) for p, q in zip(x, y))


def test_flat_gradient_buffer_and_deferred_update():
    """Agent.flat_grads: every .grad is a view of one buffer and stays one across learn steps; learn(defer=True) +
    finish_learn() in a single process is the plain learn()."""
    import DDQN
    torch.manual_seed(5)
    B, W = 256, 10
    a1 = DDQN.Agent(W, 3, device="cuda", make_memory=False)
    a1.qnetwork_local.dropout.p = 0.0
    a1.force_flat_grads = True                 # (what one-rank-per-GPU runs use; a single process lets autograd assign the gradients)
    a2 = copy.deepcopy(a1)
    a2.optimizer = type(a1.optimizer)(a2.qnetwork_local.parameters())      # (as Agent builds it)
    for step in range(3):
        s, s2 = _codes(B, W + 2, step), _codes(B, W + 2, 50 + step)
        act = torch.randint(0, 4, (B, 1), device="cuda")
        r = torch.randn(B, 1, device="cuda")
        d = (torch.rand(B, 1, device="cuda") < 0.3).float()
        l1 = a1.learn((s, act, r, s2, d), DDQN.GAMMA)
        l2 = a2.learn((s, act, r, s2, d), DDQN.GAMMA, defer=True)
        a2.finish_learn()
        assert torch.equal(l1, l2)
        flat = a1.qnetwork_local._tron_flat_grads
        assert all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in a1.qnetwork_local.parameters())
        assert torch.equal(torch.cat([p.grad.reshape(-1) for p in a1.qnetwork_local.parameters()]), flat)
    for p, q in zip(a1.qnetwork_local.parameters(), a2.qnetwork_local.parameters()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("W,p", [(10, 0.2), (24, 0.2), (10, 0.0)])
def test_tail_mlp_node_is_the_layer_by_layer_graph(W, p, monkeypatch):
    """fc1 .. actor2 with mish and dropout (DQNNet.py:55-63) as one autograd node (`_TailMLP`) against the same layers as
    separate nodes: the same kernels and the same dropout draws, so outputs and every gradient agree bit for bit."""
    from Net import activations
    from Net.DQNNet import Net
    torch.manual_seed(W)
    net = Net(3, W).cuda().train()
    net.dropout.p = p
    B = 512
    x0 = torch.randn(B, net.flat, device="cuda")
    gq = torch.randn(B, 4, device="cuda")
    res = []
    for fused_tail in (True, False):
        if not fused_tail:
            monkeypatch.setattr(activations, "tail_mlp_supported", lambda net, x: False)
        x = x0.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        torch.manual_seed(123)
        q = net._after_conv7(x)
        assert (type(q.grad_fn).__name__ == "_TailMLPBackward") == fused_tail
        q.backward(gq)
        res.append([q.detach(), x.grad] + [l.grad.clone() for m in (net.fc1, net.fc2, net.actor1, net.actor2) for l in (m.weight, m.bias)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    with torch.no_grad():                          # eval / no-grad calls keep the modules
        net.eval()
        assert net._after_conv7(x0).grad_fn is None


def test_adam_soft_is_adam_then_soft_update():
    """DDQN.AdamSoft (one launch: Adam's step of every tensor + the target net's soft update, DDQN.py:52,149-165) against
    torch.optim.Adam + the reference's soft-update formula over five steps: parameters with gradients that are views of one flat
    buffer at odd offsets, one parameter without a gradient (no Adam step, soft update still applied), more tensors than one
    launch holds; the state interchanges with optim.Adam's."""
    import DDQN
    torch.manual_seed(11)
    shapes = [(3, 5), (64, 64, 3, 3), (7,), (1023,), (1025,), (256, 577)] + [(k + 1,) for k in range(34)]
    ps = [torch.nn.Parameter(torch.randn(*sh, device="cuda")) for sh in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    tp = [torch.randn_like(p) for p in ps]
    tq = [t.clone() for t in tp]
    mine, ref = DDQN.AdamSoft(ps), torch.optim.Adam(qs)
    flat = torch.zeros(sum(p.numel() for p in ps) + 1, device="cuda")
    tau, skip = 0.05, 2
    for step in range(5):
        flat.normal_()
        off = 1                                                           # (odd offsets: the views are not 16-byte aligned)
        for k, (p, q) in enumerate(zip(ps, qs)):
            if k == skip:
                p.grad = q.grad = None
            else:
                p.grad = flat[off:off + p.numel()].view_as(p)
                q.grad = p.grad.clone() * (1.0 + step)                    # (scaled below as well: magnitudes change from step to step)
                p.grad = p.grad * (1.0 + step)
            off += p.numel()
        assert mine.step(targets=tp, tau=tau) is True
        ref.step()
        with torch.no_grad():
            for t, q in zip(tq, qs):
                t.copy_(tau * q + (1 - tau) * t)
        for k, (p, q, t, u) in enumerate(zip(ps, qs, tp, tq)):
            assert (p - q).abs().max().item() <= 2e-7 * max(1.0, q.abs().max().item()) * (step + 1), (step, k)
            assert (t - u).abs().max().item() <= 2e-7 * max(1.0, u.abs().max().item()) * (step + 1), (step, k)
    assert torch.equal(ps[skip], qs[skip]) and ps[skip] not in mine.state
    sd = mine.state_dict()
    assert all(float(v["step"]) == 5.0 for v in sd["state"].values())
    ref2 = torch.optim.Adam(qs)
    ref2.load_state_dict(sd)                                              # torch's optimizer takes AdamSoft's state ...
    mine2 = DDQN.AdamSoft(ps)
    mine2.load_state_dict(ref.state_dict())                               # ... and the other way round
    for k, p in enumerate(ps):
        if k != skip:
            assert (mine2.state[p]["exp_avg"] - mine.state[p]["exp_avg"]).abs().max().item() < 1e-6


@pytest.mark.parametrize("contents", [True, False])
def test_full_checkpoint_restores_the_replay_ring(tmp_path, contents):
    """save_checkpoint(replay="contents" / "cursor") + load_checkpoint (optimizer, epsilon, replay head on top
    of the .bak state-dict; DDQN.py:326 saves the target net only): a resumed agent draws the SAME next batches from the
    same ring contents, pushes to the same slots, continues the exploration draw sequence, and its next learn step equals
    the original's bit for bit.  The ring has wrapped (head != size) when it is saved."""
    import DDQN
    torch.manual_seed(9)
    W, S, cap, n = 10, 12, 1500, 512
    a = DDQN.Agent(W, 3, device="cuda", buffer_size=cap, batch_size=256, seed=77, rank=1)
    a.qnetwork_local.dropout.p = 0.0

    def push(agent, k):
        s, s2 = _codes(n, S, 10 + k), _codes(n, S, 20 + k)
        act = ((torch.arange(n, device="cuda") + k) % 4).to(torch.int8)
        r = torch.arange(n, device="cuda", dtype=torch.float32) + 1000 * k
        d = ((torch.arange(n, device="cuda") + k) % 5 == 0).to(torch.int8)
        agent.memory.add_batch(s, act, r, s2, d)

    for k in range(4):                                                   # 2 048 rows through 1 500 slots: wrapped
        push(a, k)
    eps = torch.tensor([0.5], device="cuda")
    for _ in range(3):
        a.act_batch(_codes(64, S, 99), eps, codes=True)
        a.learn(a.memory.sample_codes(), DDQN.GAMMA)
    head, size, calls = a.memory.memory.cursor()
    assert size == cap and head == 2048 % cap and calls == 3
    path = str(tmp_path / "full.ckpt")
    DDQN.save_checkpoint(path, a, epsilon=0.25, counters={"games": 11}, replay="contents" if contents else "cursor")
    b = DDQN.Agent(W, 3, device="cuda", buffer_size=cap, batch_size=256, seed=77, rank=1)    # a resumed run: the same launch arguments
    b.qnetwork_local.dropout.p = 0.0                                     # (the sampler's Philox key is the ring's (seed, rank))
    if not contents:                                                     # a cursor-only checkpoint needs the transitions from elsewhere:
        for k in range(4):                                               # (here: replayed) — into a ring that holds fewer it is skipped
            push(b, k)
    e, counters = DDQN.load_checkpoint(path, b)
    assert e == 0.25 and counters == {"games": 11} and b._eps_calls == a._eps_calls == 3 and b._eps_seed == 77
    assert any(not torch.equal(p, q) for p, q in zip(DDQN.Agent(W, 3, device="cuda", make_memory=False).qnetwork_local.parameters(),
                                                     b.qnetwork_local.parameters()))
    assert b.memory.memory.cursor() == (head, size, calls)
    x, y = a.memory.sample_codes(), b.memory.sample_codes()              # the same permutation (call counter) of the same contents
    assert all(torch.equal(p, q) for p, q in zip(x, y))
    push(a, 7), push(b, 7)                                               # lands in the same slots
    assert a.memory.memory.cursor() == b.memory.memory.cursor()
    sa, sb = a.memory.memory.state_dict(), b.memory.memory.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in ("states", "next_states", "actions", "rewards", "dones"))
    la, lb = a.learn(a.memory.sample_codes(), DDQN.GAMMA), b.learn(b.memory.sample_codes(), DDQN.GAMMA)
    assert torch.equal(la, lb)
    for p, q in zip(a.qnetwork_local.parameters(), b.qnetwork_local.parameters()):
        assert torch.equal(p, q)
    obs = _codes(64, S, 123)
    assert torch.equal(a.act_batch(obs, eps, codes=True), b.act_batch(obs, eps, codes=True))
    # a cursor that claims more than a fresh ring holds is not applied
    c = DDQN.Agent(W, 3, device="cuda", buffer_size=cap, batch_size=256, seed=1)
    if not contents:
        DDQN.load_checkpoint(path, c)
        assert c.memory.memory.cursor()[:2] == (0, 0)
    L = a.memory.memory._lib
    assert L.tron_replay_set_cursor(a.memory.memory._h, cap, 0, 0) != 0 and L.tron_replay_set_cursor(a.memory.memory._h, 5, 3, 0) != 0
