"""GPU parity of the one-launch inference pass of the actor / critic MLP (``cusrl_mlp2_forward``, csrc/mlp_forward.hip) through
the C ABI: against a float64 evaluation of the same stack (fp32 bar: 1e-5 of the accumulated magnitude, written at the assert),
its sampling epilogue bit for bit against ``cusrl_normal_sample_logp`` given the same mean, and the Actor / Value modules with
the pass on and off (same generator stream, same leaves)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from cusrl_amd import ops as _ops

    return _ops


def _stack(K, H1, H2, A, seed, bias3=True):
    g = torch.Generator().manual_seed(seed)
    make = lambda *shape, scale: (torch.randn(*shape, generator=g) * scale).to(DEV)  # noqa: E731
    return (make(H1, K, scale=K ** -0.5), make(H1, scale=0.3), make(H2, H1, scale=H1 ** -0.5), make(H2, scale=0.3),
            make(A, H2, scale=H2 ** -0.5), make(A, scale=0.3) if bias3 else None)


def _reference(x, layers):
    """float64, plus the magnitude the fp32 sums accumulate: |w3| (|w2| relu(...) + |b2|) + |b3| — the yardstick of the bound."""
    w1, b1, w2, b2, w3, b3 = [None if t is None else t.double().cpu() for t in layers]
    x = x.double().cpu()
    h1 = torch.relu(x @ w1.T + b1)
    h2 = torch.relu(h1 @ w2.T + b2)
    out = h2 @ w3.T + (0 if b3 is None else b3)
    m1 = x.abs() @ w1.abs().T + b1.abs()
    m2 = m1 @ w2.abs().T + b2.abs()
    mag = m2 @ w3.abs().T + (0 if b3 is None else b3.abs())
    return out, mag


@pytest.mark.parametrize("rows,K,H1,H2,A", [
    (4096, 48, 256, 128, 12),   # BASELINE config 2: acting
    (4096, 48, 256, 128, 1),    # ... the value head
    (37, 20, 128, 64, 5),       # ragged rows, K not a multiple of 16, A not a multiple of 4
    (1, 4, 128, 128, 16),       # one row, one k-quad
    (10000, 64, 256, 64, 3),    # more tiles than workgroups: weights walk the tiles in registers
    (24 * 4096, 48, 256, 128, 12),  # the statistics pass over the whole buffer
])
def test_mlp2_forward_matches_float64(ops, rows, K, H1, H2, A):
    layers = _stack(K, H1, H2, A, seed=rows + K)
    x = torch.randn(rows, K, generator=torch.Generator().manual_seed(7)).to(DEV)
    assert ops.mlp2_forward_supported(x, layers)
    out = ops.mlp2_forward(x, layers)
    want, mag = _reference(x, layers)
    err = (out.double().cpu() - want).abs()
    # fp32 sums of <= 256 + 128 + 64 terms: 1e-5 of the accumulated magnitude (achieved: ~2e-7)
    assert float((err / mag).max()) < 1e-5, float((err / mag).max())
    # and against the library's own fp32 chain (what the unfused path computes): same bound
    w1, b1, w2, b2, w3, b3 = layers
    lib = torch.relu(torch.relu(x @ w1.T + b1) @ w2.T + b2) @ w3.T + b3
    assert float(((out - lib).abs().double().cpu() / mag).max()) < 1e-5


def test_mlp2_forward_without_head_bias_and_nonfinite_rows(ops):
    layers = _stack(48, 256, 128, 12, seed=3, bias3=False)
    x = torch.randn(64, 48, generator=torch.Generator().manual_seed(1)).to(DEV)
    x[5, 3] = float("nan")
    x[9, 0] = float("inf")
    out = ops.mlp2_forward(x, layers)
    want, mag = _reference(x, layers)
    good = torch.ones(64, dtype=torch.bool)
    good[5] = good[9] = False
    assert float(((out.double().cpu() - want).abs() / mag)[good].max()) < 1e-5
    assert not torch.isfinite(out[5]).all() and not torch.isfinite(out[9]).all()  # a poisoned row stays poisoned, nothing else is


@pytest.mark.parametrize("rows,A", [(4096, 12), (100, 5), (8192, 16)])
def test_mlp2_sampling_epilogue_is_normal_sample_logp(ops, rows, A):
    """Given the mean the pass itself produced, action / log-prob / repeated std are BIT-identical to the stand-alone sampling
    launch (same expressions, same summation order over the action dims)."""
    layers = _stack(48, 256, 128, A, seed=11)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(rows, 48, generator=g).to(DEV)
    eps = torch.randn(rows, A, generator=g).to(DEV)
    std = (torch.rand(A, generator=g) * 0.9 + 0.1).to(DEV)
    action, logp, mean, repeated = ops.mlp2_forward(x, layers, std=std, eps=eps)
    assert torch.equal(mean, ops.mlp2_forward(x, layers))  # the epilogue does not change the pass
    want_action, want_logp, want_repeated = ops.normal_sample_logp(mean, std, eps, repeat_std=True)
    assert torch.equal(action, want_action) and torch.equal(logp, want_logp) and torch.equal(repeated, want_repeated)
    assert logp.shape == (rows, 1)


def test_mlp2_forward_refuses_what_it_does_not_take(ops):
    from cusrl_amd import _native

    lib = _native.lib()
    assert not lib.cusrl_mlp2_forward_supported(48, 96, 128, 12)    # hidden widths outside the instantiated set
    assert not lib.cusrl_mlp2_forward_supported(50, 256, 128, 12)   # K % 4
    assert not lib.cusrl_mlp2_forward_supported(48, 256, 128, 17)   # more than one output tile
    layers = _stack(48, 256, 128, 12, seed=1)
    x = torch.randn(8, 48, device=DEV)
    buf = torch.empty(8, 12, device=DEV)
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    args = lambda **kw: (P(kw.get("x", x)), 8, 48, P(layers[0]), P(layers[1]), 256, P(layers[2]), P(layers[3]), 128, P(layers[4]),  # noqa: E731
                         P(layers[5]), 12, kw.get("out", P(buf)), kw.get("std"), kw.get("eps"), kw.get("action"), kw.get("logp"),
                         None, None)
    assert lib.cusrl_mlp2_forward(*args()) == 0
    assert lib.cusrl_mlp2_forward(*args(out=None)) == -1           # no sampling epilogue and nowhere to store
    assert lib.cusrl_mlp2_forward(*args(eps=P(buf))) == -1         # eps without std / action / logp
    assert lib.cusrl_mlp2_forward(*args(std=P(buf))) == -1         # epilogue operands without eps
    assert lib.cusrl_mlp2_forward(*args(x=torch.empty(8 * 48 + 1, device=DEV)[1:])) == -3  # an input that is not 16-byte aligned
    torch.cuda.synchronize()


def _agent_parts(seed, fused):
    import cusrl_amd as cusrl

    cusrl.set_global_seed(seed)
    factory = cusrl.preset.PpoAgentFactory()
    spec_env = cusrl.testing.SyntheticEnvironment(512, 48, 12, device=DEV)
    agent = factory.from_environment(spec_env)
    agent.actor.fused_inference = fused
    agent.critic.fused_inference = fused
    return agent, spec_env


def test_actor_and_value_modules_take_the_fused_pass_without_grad_only():
    from cusrl_amd import _native

    agent, env = _agent_parts(3, True)
    plain, _ = _agent_parts(3, False)
    plain.load_state_dict(agent.state_dict())
    observation = torch.randn(512, 48, device=DEV)
    before = _native.launch_counts.get("cusrl_mlp2_forward", 0)
    with torch.no_grad():
        torch.manual_seed(9)
        dist_f, (action_f, logp_f), _ = agent.actor.explore(observation)
        torch.manual_seed(9)
        dist_p, (action_p, logp_p), _ = plain.actor.explore(observation)
        value_f, value_p = agent.critic.evaluate(observation), plain.critic.evaluate(observation)
        params_f, _ = agent.actor(observation)
        params_p, _ = plain.actor(observation)
    assert _native.launch_counts.get("cusrl_mlp2_forward", 0) == before + 3  # explore, evaluate, forward of the fused agent
    # same eps (same generator call), means within the fp32 bound of a 256-term sum, so actions / log-probs follow
    for a, b in ((dist_f["mean"], dist_p["mean"]), (action_f, action_p), (value_f, value_p), (params_f["mean"], params_p["mean"])):
        assert a.shape == b.shape and float((a - b).abs().max()) < 1e-5 * max(1.0, float(b.abs().max()))
    assert torch.equal(dist_f["std"], dist_p["std"]) and torch.equal(params_f["std"], params_p["std"])
    assert float((logp_f - logp_p).abs().max()) < 1e-4 and logp_f.shape == logp_p.shape == (512, 1)
    assert "backbone.output" not in agent.actor.intermediate_repr and "backbone.output" in plain.actor.intermediate_repr
    # with autograd on, the modules run their differentiable path (and leave the latent)
    out, _ = agent.actor(observation)
    assert out["mean"].requires_grad and "backbone.output" in agent.actor.intermediate_repr
    assert agent.critic.evaluate(observation).requires_grad
