"""Correctness soak of the CAPTURED minibatch step (GPU): every replay's flat gradient against an independent evaluation.

What must hold is the reference's ``_train_step`` (cusrl/template/actor_critic.py:302-320): the gradient the optimizer sees
is d(value_loss + surrogate_loss + entropy_loss [+ further objectives]) / d(parameters) on THIS minibatch at THESE parameters.
Round 4's suite compared captured steps only with other forms of this repository or at 8 envs x 16 steps; a defect that put
foreign words into two bias-gradient slots of replayed steps at 1024-row minibatches went through it (DESIGN.md section 5:
memset nodes of replayed hipGraphs).  Here, for {stock, AMP, RND, split, hook-by-hook} compositions x {1024, 4096, 24 576}-row
minibatches, the step is captured and then replayed >= 64 times on fresh index slices, and after EVERY replay

* EVERY window of the flat gradient buffer — the actor's, the critic's and (round 6) the AMP discriminator's / the RND
  predictor's — is recomputed with plain torch autograd in float64 from the parameters the step started from and the rows its
  index slice names (plain indexing of the buffer leaves; formulas as in cusrl/hook/on_policy/ppo.py:10-18, value.py:121-137,
  nn/module/distribution.py:207-213, hook/auxiliary/amp.py:137-168, rnd.py:77-82) and held to 1e-5: a weight gradient of that
  tensor's largest reference entry (a discriminator weight: of the summed largest entries of the three loss parts that cancel
  in it), a bias / std-vector gradient — a plain sum over the rows — element by element of the summed MAGNITUDES of its own
  terms (round 6; round 5 measured biases against their layer's weight gradients);
* no bias-gradient window may come back bit-identical to the previous replay's (a stale slot), every word is finite;
* the captured graph itself is checked: no memset node, no ATen ``reduce_kernel`` (cusrl_graph_census).

These replays go one graph per minibatch step (``CUSRL_EPOCH_GRAPHS=0``) so that every step's gradient can be looked at; the
default form since round 6 — one graph per epoch, gathers one step ahead — is held to this one bit for bit
(``test_epoch_graphs_change_no_bit``).  The recurrent (never captured) step has its own float64 soak in tests/test_recurrent.py.

(The check style follows the reference's cusrl_test/_helpers.py:76-94: run the real loop, inspect what it produced.)
"""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SIZES = {  # minibatch rows -> (envs, horizon, minibatches, epochs, obs, act, iterations after the capture)
    1024: (256, 8, 2, 2, 12, 4, 16),
    4096: (1024, 8, 2, 2, 12, 4, 16),
    24576: (4096, 24, 4, 2, 48, 12, 8),
}


@pytest.fixture(scope="module")
def cusrl():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    import cusrl_amd

    cusrl_amd.config.set_device(DEV)
    return cusrl_amd


def _probe_hook(cusrl, weight):
    class PolicyTermsProbe(cusrl.Hook):
        """A user-written objective with plain torch reductions (what brings ATen's split reduce_kernel into a captured step)."""

        def objective(self, metadata, batch):
            ratio, entropy = batch["action_prob_ratio"], batch["curr_entropy"]
            penalty = (ratio - 1.2).square().mean() + 0.1 * (batch["action_logp_ratio"] + 0.3).square().mean() - 0.05 * entropy.mean()
            # ... and a column sum of a WIDE matrix over the minibatch: the shape ATen reduces in its split form (staging buffer +
            # semaphore zeroed by hipMemsetAsync), i.e. the memset nodes `_Capture.capture` must replace
            penalty = penalty + WIDE_WEIGHT * _wide_column_sums(batch["curr_action_dist"]["mean"]).square().mean()
            return {"probe_loss": weight * (penalty + 0.01 * batch["curr_action_logp"].mean())}

    return PolicyTermsProbe()


PROBE_WEIGHT = 0.5
WIDE_WEIGHT = 1e-3


def _wide_column_sums(mean):
    """``[rows, A] -> [A * 64]``: every action column spread over 64 fixed weights, summed over the rows."""
    spread = torch.linspace(0.5, 1.5, 64, dtype=mean.dtype, device=mean.device)
    return (mean.unsqueeze(-1) * spread).flatten(1).sum(0) / mean.shape[0]


def _factory(cusrl, kind, T, minibatches, epochs):
    common = dict(num_steps_per_update=T, sampler_epochs=epochs, sampler_mini_batches=minibatches, compile=True,
                  optimizer_kwargs={"capturable": True, "fused": True})
    if kind == "amp":
        k = 6
        dataset = torch.randn(4096, 2 * k, device=DEV)
        # amp_batch_size=None: the discriminator is trained on every row of the minibatch (no torch.randint subsample inside the
        # captured objective whose indices an outside evaluation could not know) — its gradients are checkable replay by replay
        return cusrl.preset.AmpAgentFactory(amp_dataset_source=dataset, amp_state_indices=slice(k), extrinsic_reward_scale=0.5,
                                            amp_reward_scale=2.0, amp_batch_size=None, **common)
    factory = cusrl.preset.PpoAgentFactory(**common)
    if kind == "split":
        factory = factory.to_underlying()
        factory.register_hook(_probe_hook(cusrl, PROBE_WEIGHT), after="entropy_loss")
    if kind == "rnd":
        factory = factory.to_underlying()
        factory.register_hook(cusrl.hook.RandomNetworkDistillation(cusrl.Mlp.Factory([32, 16]), output_dim=8, reward_scale=0.1),
                              before="value_computation")
    return factory


AMP, RND = "hook.adversarial_motion_prior.discriminator", "hook.random_network_distillation.predictor"


def _reference_gradients(agent, names, params_before, indices, kind):
    """float64 autograd of the step's objective on the rows ``indices`` names, at ``params_before`` (dict name -> fp32 tensor):
    ``(gradients by parameter name, per-element sums of |terms| for the parameters whose gradient is a plain sum over the rows
    — biases, the std vector — , clip margin)``."""
    p = {name: value.double().requires_grad_(True) for name, value in params_before.items()
         if name.startswith(("actor.", "critic.", AMP + ".", RND + "."))}
    rows = lambda key: agent.buffer.storage[key].flatten(0, 1)[indices].double()  # noqa: E731
    rows32 = lambda key: agent.buffer.storage[key].flatten(0, 1)[indices]  # noqa: E731
    obs, action, old_logp = rows("observation"), rows("action"), rows("action_logp")
    advantage, ret = rows("advantage"), rows("return")
    summed: dict[str, torch.Tensor] = {}  # bias / std name -> the [rows, n] tensor whose row gradients add up to its gradient
    part_scales: dict[str, float] = {}  # weight name -> sum of the largest entries of the loss parts that cancel in its gradient

    def mlp(prefix, x, x32, hidden=(0, 2)):
        """float64 values and float64 autograd, but each ReLU's on / off decision is the fp32 one: a pre-activation within fp32
        rounding of zero (~0.4 of the 10^6 units of a replay) would otherwise flip between the two precisions and move a
        weight-gradient row by ~1 / rows of its largest entry — a property of the kink, not of the step.  The fp32 decisions
        come from an eager re-execution of the very library call the step's forward makes (same shapes, same arguments:
        the libraries are bit-reproducible call to call, scripts/gemm_determinism.py)."""
        for i in hidden:
            weight, bias = f"{prefix}.layers.{i}.weight", f"{prefix}.layers.{i}.bias"
            x32 = torch._addmm_activation(params_before[bias], x32, params_before[weight].t())  # + ReLU in the GEMM epilogue
            z = summed[bias] = x @ p[weight].t() + p[bias]
            x = torch.where(x32 > 0, z, torch.zeros((), dtype=torch.float64, device=x.device))
        return x

    def head(weight, bias, x):
        z = summed[bias] = x @ p[weight].t() + p[bias]
        return z

    mean = head("actor.distribution.mean_head.weight", "actor.distribution.mean_head.bias",
                mlp("actor.backbone", obs, rows32("observation")))
    std = summed["actor.distribution.std.param"] = p["actor.distribution.std.param"].expand_as(mean)  # identity bijector (the preset's)
    logp = (-((action - mean) ** 2) / (2 * std**2) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1, keepdim=True)
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + std.log()).sum(-1, keepdim=True)
    logp_ratio = logp - old_logp
    ratio = logp_ratio.exp()
    hooks = agent.hook
    surrogate, value_hook, entropy_hook = hooks["ppo_surrogate_loss"], hooks["value_loss"], hooks["entropy_loss"]
    lo, hi = float(torch.tensor(1.0 - surrogate.clip_ratio, dtype=torch.float32)), float(torch.tensor(1.0 + surrogate.clip_ratio, dtype=torch.float32))
    loss = -torch.min(advantage * ratio, advantage * ratio.clamp(lo, hi)).mean() * surrogate.weight
    value = head("critic.value_head.weight", "critic.value_head.bias", mlp("critic.backbone", obs, rows32("observation")))
    if value_hook.loss_clip is None:
        loss = loss + (value - ret).square().mean() * value_hook.weight
    else:
        old = rows("value")
        clipped = old + (value - old).clamp(-value_hook.loss_clip, value_hook.loss_clip)
        loss = loss + torch.max((value - ret).square(), (clipped - ret).square()).mean() * value_hook.weight
    loss = loss - entropy.mean() * entropy_hook.weight
    if kind == "split":
        penalty = (ratio - 1.2).square().mean() + 0.1 * (logp_ratio + 0.3).square().mean() - 0.05 * entropy.mean()
        penalty = penalty + WIDE_WEIGHT * _wide_column_sums(mean).square().mean()
        loss = loss + PROBE_WEIGHT * (penalty + 0.01 * logp.mean())
    if kind == "amp":
        # AdversarialMotionPrior.objective (cusrl/hook/auxiliary/amp.py:137-168) on every row of the minibatch: BCE of the agent's
        # logits against 0 and the expert's against 1, halved, and the gradient penalty E ||d logit / d expert||^2 — both times
        # loss_weight.  The hook evaluates the discriminator on the JOINT [agent; expert] batch: so do the fp32 ReLU decisions.
        hook = hooks["adversarial_motion_prior"]
        agent_rows, expert_rows = rows("agent_transition"), rows("expert_transition").requires_grad_(True)
        count = agent_rows.shape[0]
        joint = mlp(AMP, torch.cat((agent_rows, expert_rows)), torch.cat((rows32("agent_transition"), rows32("expert_transition"))))
        logit = head(f"{AMP}.layers.4.weight", f"{AMP}.layers.4.bias", joint)
        agent_logit, expert_logit = logit[:count], logit[count:]
        softplus = torch.nn.functional.softplus
        (slope,) = torch.autograd.grad(expert_logit.sum(), expert_rows, create_graph=True)
        # BCEWithLogits against 0 (agent) / 1 (expert), halved; the penalty
        parts = [hook.loss_weight * softplus(agent_logit).mean() / 2, hook.loss_weight * softplus(-expert_logit).mean() / 2,
                 hook.loss_weight * hook.grad_penalty_weight * slope.square().sum(-1).mean()]
        loss = loss + (parts[0] + parts[1]) + parts[2]
        # The three parts pull the discriminator's weights in opposite directions and cancel near its equilibrium (the gradient
        # shrinks as training goes on, its fp32 summation error does not): a weight gradient's yardstick is the sum of the
        # parts' largest entries — the magnitudes that were added up — not the largest entry of what is left of them.
        weights = [name for name in p if name.startswith(AMP + ".") and name.endswith(".weight")]
        for part in parts:
            for name, grad in zip(weights, torch.autograd.grad(part, [p[name] for name in weights], retain_graph=True, allow_unused=True)):
                if grad is not None:
                    part_scales[name] = part_scales.get(name, 0.0) + float(grad.abs().max())
    if kind == "rnd":
        # RandomNetworkDistillation.objective (cusrl/hook/auxiliary/rnd.py:77-82): MSE of the predictor against the frozen target
        next_obs, next_obs32 = rows("next_observation"), rows32("next_observation")
        target32 = next_obs32
        prefix = RND[: -len("predictor")] + "target"
        for i in (0, 2):
            target32 = torch._addmm_activation(params_before[f"{prefix}.layers.{i}.bias"], target32, params_before[f"{prefix}.layers.{i}.weight"].t())
        target = torch.nn.functional.linear(target32, params_before[f"{prefix}.layers.4.weight"], params_before[f"{prefix}.layers.4.bias"]).double()
        prediction = head(f"{RND}.layers.4.weight", f"{RND}.layers.4.bias", mlp(RND, next_obs, next_obs32))
        loss = loss + (prediction - target).square().mean()
    wanted = [name for name, value in p.items() if value.requires_grad and not name.startswith(RND[: -len("predictor")] + "target")]
    grads = torch.autograd.grad(loss, [p[name] for name in wanted] + list(summed.values()), allow_unused=True)
    by_name = {name: (grad if grad is not None else torch.zeros_like(p[name])) for name, grad in zip(wanted, grads)}
    # a gradient that is a plain sum over the rows: the sum of the MAGNITUDES of its terms, per element (its own yardstick)
    magnitudes = {name: terms.abs().sum(0) for name, terms in zip(summed, grads[len(wanted):]) if terms is not None}
    # rows whose ratio sits within fp32 noise of a clip bound may legitimately fall on either side (oracle.ppo_loss_f64)
    margin = torch.minimum((ratio - lo).abs(), (ratio - hi).abs()).min().detach()
    return by_name, magnitudes, float(margin), part_scales


@pytest.mark.parametrize("rows", [1024, 4096, 24576])
@pytest.mark.parametrize("kind", ["stock", "amp", "rnd", "split", "hook_by_hook"])
def test_every_replay_of_a_captured_step_matches_float64_autograd(cusrl, kind, rows, gradient_parity, monkeypatch):
    from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
    from cusrl_amd.template import graphs

    # one graph per minibatch step, so that every replay's gradient can be looked at (the default since round 6 replays a whole
    # epoch's steps from one graph; that form is held to THIS one bit for bit: test_epoch_graphs_change_no_bit below)
    monkeypatch.setenv("CUSRL_EPOCH_GRAPHS", "0")

    N, T, minibatches, epochs, obs_dim, act_dim, soak_iterations = SIZES[rows]
    assert N * T // minibatches == rows
    cusrl.set_global_seed(33)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=obs_dim, action_dim=act_dim, device=DEV)
    warm = 3  # iteration 0: eager warm-up of every step, 1: capture, 2: first pure replays
    trainer = cusrl.Trainer(env, _factory(cusrl, kind, T, minibatches, epochs), num_iterations=warm + soak_iterations, verbose=False)
    agent = trainer.agent
    assert FusedPpoObjective.mode(agent.hook) == ("split" if kind == "split" else "fused")
    if kind == "amp":
        assert agent.hook["adversarial_motion_prior"].batch_size is None
    if kind == "hook_by_hook":
        # the reference's own op chains (Normal.log_prob / exp / min / mean ..., differentiated by autograd op by op) inside the
        # captured step: the composition with the most ATen reductions — at >= 1024 rows their split form, i.e. the memset
        # nodes that `_Capture.capture` replaces — and therefore the one that exercises the replacement end to end
        agent.fuse_objective = False
    flat = agent.flat_gradients
    names = {id(p): name for name, p in agent.named_parameters()}
    windows = [(names[id(p)], offset, p.numel(), tuple(p.shape)) for p, offset in zip(flat.params, flat.offsets)]
    assert any(name.startswith("critic.") for name, *_ in windows) and any(name.startswith("actor.") for name, *_ in windows)
    record = {"replays": 0, "worst": {}, "previous": None, "near_clip": 0}
    original = graphs.GraphedTrainStep.run

    def run(self, metadata, indices, *args, **kwargs):
        before = {name: p.detach().clone() for name, p in agent.named_parameters()}
        was_captured = self.state == 2
        original(self, metadata, indices, *args, **kwargs)
        if not (was_captured and self.state == 2):
            return
        torch.cuda.synchronize()
        grads = flat.buffer.clone()
        assert torch.isfinite(grads).all(), f"replay {record['replays']}: non-finite words in the flat gradient buffer"
        reference, magnitudes, margin, part_scales = _reference_gradients(agent, names, before, indices, kind)
        record["near_clip"] += margin < 1e-6
        for name, offset, numel, shape in windows:
            mine = grads[offset : offset + numel]
            if name in reference:
                want = reference[name].reshape(-1)
                if name in magnitudes:
                    # a bias / std-vector gradient is a plain sum of `rows` signed terms that cancels to ~1e-3 of their absolute
                    # sum: each ELEMENT is held to 1e-5 of the sum of the magnitudes of ITS OWN terms — what fp32 summation
                    # can be asked for, and tight enough that a wrong value-head bias gradient cannot hide behind its layer's
                    # weight gradients (round 5's yardstick was the layer's largest weight-gradient entry)
                    error = float(((mine.double() - want).abs() / magnitudes[name].reshape(-1).clamp_min(1e-30)).max())
                else:
                    scale = max(float(want.abs().max()), part_scales.get(name, 0.0), 1e-30)
                    error = float((mine.double() - want).abs().max()) / scale
                if margin >= 1e-6:  # (the recorded worst errors are those of the replays held to 1e-5)
                    record["worst"][name] = max(record["worst"].get(name, 0.0), error)
                # A row whose ratio lies within 1e-6 of a clip bound may fall on the other side in fp32, and then its WHOLE
                # contribution is there or not: a gradient entry is a sum of `rows` signed terms of random sign, so one term is
                # ~1 / sqrt(rows) of it, not 1 / rows (seen: 2.7e-2 at 24 576 rows).  Such a replay (a handful per thousand)
                # is only held to the gross bound that any corrupted word would still break; they are counted below.
                tight = 1e-5  # (a discriminator weight: of the summed largest entries of the parts that cancel in it, see above)
                bound = tight if margin >= 1e-6 else 0.1
                assert error <= bound, (f"replay {record['replays']} ({kind}, {rows} rows): {name} off by {error:.3e} of its largest entry; "
                                        f"first words {mine[:4].tolist()} vs {want[:4].tolist()}")
            else:
                record["unchecked"] = record.get("unchecked", set()) | {name}
            if name.endswith(".bias") and record["previous"] is not None:
                assert not torch.equal(mine, record["previous"][offset : offset + numel]), \
                    f"replay {record['replays']}: {name} came back bit-identical to the previous replay (stale slot)"
        record["previous"] = grads
        record["replays"] += 1

    graphs.GraphedTrainStep.run = run
    try:
        trainer.run_training_loop()
    finally:
        graphs.GraphedTrainStep.run = original
    torch.cuda.synchronize()
    assert record["replays"] >= 64, record["replays"]
    assert record["near_clip"] <= 3, record["near_clip"]  # (the replays held to the gross bound only)
    # EVERY window of the flat buffer was recomputed: the actor's, the critic's and (amp / rnd) the discriminator's / predictor's
    assert not record.get("unchecked"), record["unchecked"]
    steps = list(agent._graphed_steps.values())
    assert steps and all(step.state == 2 for step in steps)
    for step in steps:  # the structural rule behind the fix
        census = step.forward_backward.census
        assert census["memset"] == 0, census
        reduces = [n for n in census["names"] if "reduce_kernel" in n]
        if kind in ("stock", "amp", "rnd"):  # (split / hook-by-hook: torch's .mean() calls are ATen reductions — allowed, their memset nodes replaced)
            assert not reduces, reduces[:3]
        if kind == "hook_by_hook":
            assert reduces  # torch's own op chains
        if kind == "split":  # the probe hook's wide column sum is a split ATen reduction: its memset node(s) were replaced
            assert reduces and census.get("memset_replaced", 0) > 0, (len(reduces), census.get("memset_replaced"))
    for name, error in record["worst"].items():
        # (recorded; the bound in force was this one unless a replay had a ratio within 1e-6 of a clip bound)
        gradient_parity(f"captured_step_soak[{kind},{rows},{name}]", [1.0 + error], [1.0], 1e-5)
    print(f"captured-step soak {kind} {rows} rows: {record['replays']} replays, worst error "
          f"{max(record['worst'].values()):.2e} of a tensor's largest entry ({record['near_clip']} replays with a ratio within 1e-6 of a clip bound)")


@pytest.mark.parametrize("concurrent", [False, True])
@pytest.mark.parametrize("kind", ["stock", "amp", "rnd", "split", "hook_by_hook"])
def test_two_seeded_runs_of_the_captured_loop_are_bit_identical(cusrl, kind, concurrent):
    """Same seed, same process, 8 iterations at 1024-row minibatches, single-stream and with the critic on its branch stream
    (split compositions are single-stream by construction): every parameter and every buffer leaf bit-identical.  This is
    the symptom the round-4 defect was found by (scripts/debug_amp_identity.py) — in BOTH stream layouts now."""
    N, T, minibatches, epochs = 256, 8, 2, 2
    finals = []
    for _ in range(2):
        cusrl.set_global_seed(21)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=12, action_dim=4, device=DEV)
        trainer = cusrl.Trainer(env, _factory(cusrl, kind, T, minibatches, epochs), num_iterations=8, verbose=False)
        trainer.agent.concurrent_critic = concurrent
        if kind == "hook_by_hook":
            trainer.agent.fuse_objective = False
        trainer.run_training_loop()
        torch.cuda.synchronize()
        agent = trainer.agent
        assert all(step.state == 2 for step in agent._graphed_steps.values())
        state = {f"param/{name}": p.detach().clone() for name, p in agent.named_parameters()}
        state.update({f"buffer/{key}": leaf.clone() for key, leaf in agent.buffer.storage.items()})
        state["grad"] = agent.flat_gradients.buffer.clone()
        finals.append(state)
    differing = [key for key in finals[0] if not torch.equal(finals[0][key], finals[1][key])]
    assert not differing, differing[:8]


@pytest.mark.parametrize("rows", [1024, 24576])
@pytest.mark.parametrize("kind", ["stock", "amp", "rnd", "split"])
def test_epoch_graphs_change_no_bit(cusrl, kind, rows, monkeypatch):
    """The default captured form (round 6): one hipGraph per EPOCH whose step bodies read their index slices in place and whose
    gathers run one step ahead on a second stream (template/graphs.py GraphedEpochs).  It is the step-by-step form above —
    float64-checked replay by replay — with another issue order and nothing else: same seed, 8 iterations, every parameter,
    every buffer leaf and the last flat gradient bit-identical between the two; and the epoch graphs really ran."""
    N, T, minibatches, epochs, obs_dim, act_dim, _ = SIZES[rows]
    finals = []
    for epoch_graphs in ("update", "1", "0"):
        monkeypatch.setenv("CUSRL_EPOCH_GRAPHS", epoch_graphs)
        cusrl.set_global_seed(57)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=obs_dim, action_dim=act_dim, device=DEV)
        trainer = cusrl.Trainer(env, _factory(cusrl, kind, T, minibatches, epochs), num_iterations=8, verbose=False)
        trainer.run_training_loop()
        torch.cuda.synchronize()
        agent = trainer.agent
        graphed = agent._graphed_epochs
        if epoch_graphs != "0" and not any(h.objective_draws_random for h in agent.hook if h.active):
            # (AMP draws random numbers inside its objective: the sampler then keeps the reference's interleaving of draws, no
            # up-front permutations, and the update steps graph by graph)
            # (one graph per update by default, one per epoch with CUSRL_EPOCH_GRAPHS=1)
            assert graphed is not None and graphed.replays >= 4 and len(graphed.epochs) in (1, epochs), (graphed and graphed.replays)
            for entry in graphed.epochs.values():
                assert entry["capture"].census["memset"] == 0
        if epoch_graphs == "0":
            assert graphed is None or graphed.replays == 0
        state = {f"param/{name}": p.detach().clone() for name, p in agent.named_parameters()}
        state.update({f"buffer/{key}": leaf.clone() for key, leaf in agent.buffer.storage.items()})
        state["grad"] = agent.flat_gradients.buffer.clone()
        state["metrics"] = {}
        finals.append(state)
    for other in finals[1:]:
        differing = [key for key in finals[0] if key != "metrics" and not torch.equal(finals[0][key], other[key])]
        assert not differing, differing[:8]
