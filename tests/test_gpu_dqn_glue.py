"""csrc/tron_dqn.hip — the trainer's small steps around the network as one launch each — against the PyTorch expressions they
replace: the Double-DQN loss and its gradient (DDQN.py:129-146), the epsilon-greedy mix (DDQN.py:105-110), the epsilon
schedule (DDQN.py:313-315)."""
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import config  # noqa: F401


@pytest.mark.parametrize("B", [1, 7, 64, 4096, 5000])
def test_td_loss_matches_the_composed_form(B):
    import DDQN
    torch.manual_seed(B)
    q = torch.randn(B, 4, device="cuda", requires_grad=True)
    a = torch.randint(0, 4, (B, 1), device="cuda")
    r = torch.randn(B, 1, device="cuda")
    d = (torch.rand(B, 1, device="cuda") < 0.3).float()
    ql, qt = torch.randn(B, 4, device="cuda"), torch.randn(B, 4, device="cuda")
    assert DDQN._td_fusable(q, a, r, d, ql, qt)
    loss = DDQN._TDLoss.apply(q, a, r, d, ql, qt, 0.95)
    (3.0 * loss).backward()
    q2 = q.detach().clone().requires_grad_(True)
    labels = r + (0.95 * qt.gather(1, ql.max(1)[1].unsqueeze(1)) * (1 - d))
    want = F.mse_loss(q2.gather(1, a), labels)
    (3.0 * want).backward()
    assert abs(loss.item() - want.item()) <= 1e-6 * max(1.0, abs(want.item()))
    assert torch.allclose(q.grad, q2.grad, rtol=1e-6, atol=1e-9)
    loss2 = DDQN._TDLoss.apply(q.detach(), a, r, d, ql, qt, 0.95)
    assert torch.equal(loss2, loss.detach())                       # fixed-order sum


def test_eps_greedy_mix():
    import DDQN
    brain = DDQN.Agent(10, 3, device="cuda", make_memory=False, seed=11)
    n = 1 << 16
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    obs = vals[torch.randint(0, 6, (n, 12, 12), device="cuda")]
    greedy = brain.qnetwork_local.infer(obs, codes=True, greedy=True)
    eps = torch.zeros(1, device="cuda")
    assert torch.equal(brain.act_batch(obs, eps, codes=True), greedy)                    # epsilon 0: the arg-max
    eps.fill_(1.0)
    a1 = brain.act_batch(obs, eps, codes=True)
    a2 = brain.act_batch(obs, eps, codes=True)
    assert a1.dtype == torch.int8 and int(a1.min()) == 0 and int(a1.max()) == 3 and not torch.equal(a1, a2)   # a new draw per call
    counts = torch.bincount(a1.long(), minlength=4).float() / n
    assert (counts - 0.25).abs().max().item() < 0.01
    eps.fill_(0.3)
    a3 = brain.act_batch(obs, eps, codes=True)
    changed = (a3 != greedy).float().mean().item()                  # explores 30 % of the time, a different action 3/4 of those
    assert abs(changed - 0.3 * 0.75) < 0.01
    # the same agent state gives the same draws
    b1, b2 = (DDQN.Agent(10, 3, device="cuda", make_memory=False, seed=5) for _ in range(2))
    b2.qnetwork_local.load_state_dict(b1.qnetwork_local.state_dict())
    assert torch.equal(b1.act_batch(obs[:999], eps, codes=True), b2.act_batch(obs[:999], eps, codes=True))


def test_eps_schedule_equals_the_tensor_expressions():
    import DDQN
    from tron import _native as nat
    torch.manual_seed(0)
    n, eps0 = 4096, 0.5
    left = DDQN._decays_left(eps0)
    sched = torch.tensor([0, 0, 0, left], dtype=torch.int64, device="cuda")
    eps64, eps32 = torch.zeros(1, dtype=torch.float64, device="cuda"), torch.zeros(1, device="cuda")
    games = cycles = decays = 0
    for step in range(40):
        done = (torch.rand(n, device="cuda") < (0.9 if step % 7 == 0 else 0.05)).to(torch.int8)
        nat.check(nat.lib().tron_eps_schedule(nat.ptr(done), n, nat.ptr(sched), DDQN.GAME_CYCLE, eps0, DDQN.DECAY_RATE, nat.ptr(eps64),
                                              nat.ptr(eps32), nat.stream_ptr()), "tron_eps_schedule")
        games += int(done.sum())
        new_cycles = games // DDQN.GAME_CYCLE
        decays = min(decays + new_cycles - cycles, left)
        cycles = new_cycles
        assert sched.tolist() == [games, cycles, decays, left]
        want = eps0 * DDQN.DECAY_RATE ** decays
        assert abs(eps64.item() - want) < 1e-12 and abs(eps32.item() - want) < 1e-7
    assert decays == left or decays > 0


@pytest.mark.parametrize("n", [1, 5, 4096, 1_000_003, 37_000_000])
@pytest.mark.parametrize("magnitude", [1.0, 3e-9, 7e5, 0.0])
def test_absmax_pow2(n, magnitude):
    """tron_absmax_pow2: max |x| exactly, and the power of two that puts it in [2^15, 2^16) (1 for an all-zero tensor)."""
    from Net.kfac import _pow2_scale
    from tron import _native as nat
    torch.manual_seed(n)
    x = torch.randn(n, device="cuda") * magnitude
    out = torch.zeros(4, device="cuda")
    nat.check(nat.lib().tron_absmax_pow2(nat.ptr(x), n, 16, nat.ptr(out), nat.stream_ptr()), "tron_absmax_pow2")
    want = x.abs().max().item()
    assert out[1].item() == want
    s = out[0].item()
    if want == 0.0:
        assert s == 1.0
    else:
        assert 2.0 ** 15 <= want * s < 2.0 ** 16 and abs(torch.log2(torch.tensor(s)).item() - round(torch.log2(torch.tensor(s)).item())) == 0.0
    assert _pow2_scale(x).item() == s


@pytest.mark.parametrize("B,O,I", [(4096, 256, 576), (4096, 128, 256), (4096, 64, 128), (4096, 4, 64), (300, 256, 3136), (257, 70, 65), (1000, 1, 1)])
def test_linear_wgrad_matches_float64(B, O, I):
    """tron_linear_wgrad (an nn.Linear layer's weight and bias gradient with the batch split over workgroups) against float64;
    through Net/activations.py::linear against the plain module."""
    from Net.activations import linear
    torch.manual_seed(B + O + I)
    lin = torch.nn.Linear(I, O).cuda()
    x = torch.randn(B, I, device="cuda", requires_grad=True)
    gy = torch.randn(B, O, device="cuda") * 1e-3
    y = linear(lin, x)
    assert torch.equal(y, lin(x))
    y.backward(gy)
    want_w, want_b, want_x = gy.double().t() @ x.detach().double(), gy.double().sum(0), gy.double() @ lin.weight.detach().double()
    for got, want in ((lin.weight.grad, want_w), (lin.bias.grad, want_b), (x.grad, want_x)):
        assert (got.double() - want).abs().max().item() / want.abs().max().item() < 3e-6
    g1 = lin.weight.grad.clone()
    lin.weight.grad = None
    linear(lin, x).backward(gy)
    assert torch.equal(lin.weight.grad, g1)                                # fixed-order sums
