"""The captured env step (template/graphs.py GraphedRolloutStep) against the host-driven rollout loop (GPU).

A `capturable` env (template/environment.py) is driven with fixed-shape resets spliced by a device-side count; under
``compile=True`` a whole env step — act, env.step, episode statistics, post_step hooks, buffer push, resets — replays
from ONE hipGraph.  Both loops issue the same kernels on the same operands and consume torch's generator identically
(the reference loop is cusrl/template/trainer.py:296-321), so from the same seed every buffer leaf, every parameter and
the episode statistics must be bit-identical.
"""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cusrl():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    import cusrl_amd

    cusrl_amd.config.set_device(DEV)
    return cusrl_amd


def _factory(cusrl, kind, T):
    common = dict(num_steps_per_update=T, sampler_epochs=2, sampler_mini_batches=2, compile=True,
                  optimizer_kwargs={"capturable": True, "fused": True})
    if kind == "continuous":
        return cusrl.preset.PpoAgentFactory(**common)
    if kind == "discrete_obs_norm":  # BASELINE config 1's shape: categorical policy + observation normalisation
        return cusrl.preset.PpoAgentFactory(action_space_type="discrete", normalize_observation=True,
                                            actor_hidden_dims=(64, 64), critic_hidden_dims=(64, 64), activation_fn="Tanh", **common)
    if kind == "amp":  # BASELINE config 5's per-step hook: discriminator forward + style reward inside the step
        k = 6
        dataset = torch.randn(4096, 2 * k, device=DEV)
        return cusrl.preset.AmpAgentFactory(amp_dataset_source=dataset, amp_state_indices=slice(k), extrinsic_reward_scale=0.5,
                                            amp_reward_scale=2.0, **common)
    raise AssertionError(kind)


def _run(cusrl, kind, capture, iterations=6, N=256, T=8):
    cusrl.set_global_seed(21)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=12, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, _factory(cusrl, kind, T), num_iterations=iterations, verbose=False)
    trainer.capture_rollout = capture
    trainer.run_training_loop()
    return trainer


@pytest.mark.parametrize("kind", ["continuous", "discrete_obs_norm", "amp"])
def test_captured_rollout_is_bit_identical_to_the_host_driven_loop(cusrl, kind):
    from cusrl_amd import _native

    T = 8
    host = _run(cusrl, kind, capture=False, T=T)
    assert host._graphed_rollout is None and host._static_resets
    appends = lambda: sum(_native.launch_counts.get(k, 0) for k in ("cusrl_buffer_push", "cusrl_step_epilogue_push"))  # noqa: E731
    pushes_before, fused_before = appends(), _native.launch_counts.get("cusrl_step_epilogue_push", 0)
    captured = _run(cusrl, kind, capture=True, T=T)
    graphed = captured._graphed_rollout
    assert graphed is not None and graphed.captured == T, (graphed and graphed.captured)
    # iteration 0 host-driven, 1 eager step bodies, 2 one capture per step (+ the replay behind each), 3 the whole rollout
    # captured as ONE graph (every step was warm), 4-5 one replay per rollout
    assert graphed.replays == 0 and len(graphed.rollouts) == 1 and graphed.rollout_replays == 2
    # third part of round 6: the env draws nothing from torch's generator (`generator_free`), so the exploration noise of a whole
    # rollout — the T draws its act steps would issue — is drawn AHEAD of the rollout's graph on a side stream (iterations 3-5);
    # the same calls in the same order: everything below stays bit-identical to the host-driven loop.  A categorical policy's act
    # step is not the fused explore pass: its draws stay inside the step.
    assert captured.environment.generator_free
    # ... and AdversarialMotionPrior draws a step's expert transitions inside the step (`step_draws_random`): drawing the noise
    # ahead would change the order of the generator's consumers
    assert graphed.noise_draws == (T * 3 if kind == "continuous" else 0)
    # ... and a replay issues no C call at all: the push entry point was only called in iterations 0-3
    assert appends() - pushes_before == T * 4
    # when no post_step hook touches the transition on the device (the continuous preset), the captured step's epilogue and
    # append are ONE launch: iterations 1-3 (iteration 0 is host-driven: two launches)
    fused = _native.launch_counts.get("cusrl_step_epilogue_push", 0) - fused_before
    # (the first captured-path step re-plans the append — its transition lists the fields in another order than the
    # host-driven loop's — and takes two launches)
    assert (T * 3 - 1 <= fused <= T * 3) if kind == "continuous" else fused == 0, fused
    a, b = host.agent, captured.agent
    assert set(a.buffer.storage) == set(b.buffer.storage)
    for key in a.buffer.storage:
        assert torch.equal(a.buffer.storage[key], b.buffer.storage[key]), key
    for (name, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(p, q), name
    for attr in ("episode_rew", "episode_len", "rew_buffer", "len_buffer"):
        assert torch.equal(getattr(host.stats, attr), getattr(captured.stats, attr)), attr
    assert host.stats.num_episodes == captured.stats.num_episodes and host.stats.num_episodes > 0
    assert host.stats.total_steps == captured.stats.total_steps == 6 * T * 256
    assert a.buffer["terminated"].any() and a.buffer["truncated"].any()  # resets were exercised
    for key, value in host.last_info.items():
        if key.startswith("Perf/"):
            continue
        assert captured.last_info[key] == pytest.approx(value, rel=1e-6, abs=1e-7), key
    assert captured.last_info["Perf/environment_time"] > 0 and captured.last_info["Perf/agent_time"] > 0


def test_per_step_graphs_alone_are_bit_identical_too(cusrl):
    """The same comparison with the whole-rollout graph switched off (what runs when a rollout is not exactly one pass over
    the buffer, or a hook decides ``should_update`` itself): T replays per rollout."""
    T = 8
    host = _run(cusrl, "continuous", capture=False, T=T)
    import os

    os.environ["CUSRL_WHOLE_ROLLOUT_GRAPH"] = "0"
    try:
        captured = _run(cusrl, "continuous", capture=True, T=T)
    finally:
        del os.environ["CUSRL_WHOLE_ROLLOUT_GRAPH"]
    graphed = captured._graphed_rollout
    assert graphed.captured == T and graphed.replays == T * 3 and not graphed.rollouts
    for key in host.agent.buffer.storage:
        assert torch.equal(host.agent.buffer.storage[key], captured.agent.buffer.storage[key]), key
    for p, q in zip(host.agent.parameters(), captured.agent.parameters()):
        assert torch.equal(p, q)


def test_static_reset_rows_reach_only_the_finished_envs_in_order(cusrl):
    """``Trainer._splice_static``: row k of the reset tensors goes to env ``indices[k]`` for k < count (read on the device);
    stale index entries past the count leave their envs untouched (environment.py:365-379 semantics)."""
    observation = torch.arange(8 * 3, dtype=torch.float32, device=DEV).view(8, 3).clone()
    state = -observation.clone()
    before_obs, before_state = observation.clone(), state.clone()
    indices = torch.tensor([1, 4, 6, 0, 0, 0, 0, 0], dtype=torch.int64, device=DEV)
    count = torch.tensor([3], dtype=torch.int32, device=DEV)
    init_obs = 100 + torch.arange(8 * 3, dtype=torch.float32, device=DEV).view(8, 3)
    init_state = 200 + torch.arange(8 * 3, dtype=torch.float32, device=DEV).view(8, 3)
    cusrl.Trainer._splice_static(observation, state, indices, count, init_obs, init_state)
    expect_obs, expect_state = before_obs.clone(), before_state.clone()
    expect_obs[[1, 4, 6]] = init_obs[:3]
    expect_state[[1, 4, 6]] = init_state[:3]
    assert torch.equal(observation, expect_obs) and torch.equal(state, expect_state)
    count.zero_()
    cusrl.Trainer._splice_static(observation, None, indices, count, init_obs + 1, None)
    assert torch.equal(observation, expect_obs)


def test_splice_rows_is_the_reset_scatter_plus_the_copy(cusrl):
    """``cusrl_splice_rows`` (what a captured step runs instead of scatter + copy): every dtype width, rows of any size,
    no finished env, every env finished."""
    from cusrl_amd import ops

    generator = torch.Generator(device=DEV).manual_seed(3)
    for N, shape, dtype in ((4096, (48,), torch.float32), (257, (5,), torch.float32), (64, (3,), torch.uint8), (1000, (6,), torch.float16),
                            (33, (2, 4), torch.float64)):
        for rate in (0.1, 0.0, 1.0):
            src = (torch.rand((N,) + shape, device=DEV, generator=generator) * 100).to(dtype)
            init = (torch.rand((N,) + shape, device=DEV, generator=generator) * 100 + 200).to(dtype)
            done = torch.rand(N, 1, device=DEV, generator=generator) < rate
            finished = done.squeeze(-1).nonzero().squeeze(-1)
            indices = torch.zeros(N, dtype=torch.int64, device=DEV)
            indices[: finished.numel()] = finished
            indices[finished.numel():] = torch.randint(0, N, (N - finished.numel(),), device=DEV, generator=generator)  # stale ids
            count = torch.tensor([finished.numel()], dtype=torch.int32, device=DEV)
            dst = torch.full_like(src, 7)
            ops.splice_rows(src, init, indices, count, done, dst)
            expect = src.clone()
            expect[finished] = init[: finished.numel()]
            assert torch.equal(dst, expect), (N, shape, dtype, rate)
    with pytest.raises(ValueError, match="alias"):
        ops.splice_rows(src, init, indices, count, done, src)


def test_a_user_hook_with_per_step_python_state_keeps_the_loop_host_driven(cusrl):
    """A hook from outside the package that overrides ``post_step`` may keep Python state per step: the captured step is
    only taken when it opts in (``rollout_capture_safe``)."""
    seen = []

    class Recorder(cusrl.Hook):
        def post_step(self, transition):
            seen.append(float(transition["reward"].sum()))

    class Shaper(cusrl.Hook):
        rollout_capture_safe = True

        def post_step(self, transition):
            transition["reward"].mul_(0.5)

    for hook_type, expect_capture in ((Recorder, False), (Shaper, True)):
        cusrl.set_global_seed(5)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=64, observation_dim=12, action_dim=4, device=DEV)
        factory = _factory(cusrl, "continuous", 4).to_underlying()
        factory.register_hook(hook_type(), index=0)
        trainer = cusrl.Trainer(env, factory, num_iterations=4, verbose=False)
        trainer.run_training_loop()
        graphed = trainer._graphed_rollout
        if expect_capture:
            assert graphed is not None and graphed.captured == 4 and len(graphed.rollouts) == 1  # iteration 3: whole rollout
        else:
            assert graphed is None or graphed.captured == 0
            assert len(seen) == 4 * 4


def test_a_non_capturable_env_keeps_the_reference_reset_path(cusrl):
    cusrl.set_global_seed(5)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=64, observation_dim=12, action_dim=4, device=DEV, capturable=False)
    trainer = cusrl.Trainer(env, _factory(cusrl, "continuous", 4), num_iterations=3, verbose=False)
    assert not trainer._static_resets
    trainer.run_training_loop()
    assert trainer._graphed_rollout is None and trainer.stats.num_episodes > 0


def test_a_changed_hook_attribute_recaptures_the_step_graphs(cusrl):
    cusrl.set_global_seed(5)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=64, observation_dim=12, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, _factory(cusrl, "continuous", 4), num_iterations=4, verbose=False)
    trainer.run_training_loop()
    graphed = trainer._graphed_rollout
    assert graphed.captured == 4
    trainer.agent.hook["entropy_loss"].update_attribute("weight", 0.02)  # part of the capture signature
    trainer.num_iterations += 1
    observation, state, _ = env.reset()
    trainer._rollout_and_update(observation, state)
    assert graphed.captured == 4 and all(torch.isfinite(p).all() for p in trainer.agent.parameters())


def test_fused_synthetic_env_step_draws_the_documented_distributions(cusrl):
    """cusrl_synthetic_env_step (ONE launch per env step): N(0, 1) observations / rewards / reset rows, Bernoulli flags at the
    configured rates, fresh numbers every step — also when the step is replayed from a hipGraph (the launch advances its own
    device-side step counter) — and the same stream for the same seed."""
    from cusrl_amd import _native

    def build():
        cusrl.set_global_seed(11)
        return cusrl.testing.SyntheticEnvironment(65536, 48, 12, device=DEV, terminate_prob=0.02, truncate_prob=0.01)

    env = build()
    assert env.fused
    action = torch.zeros(65536, 12, device=DEV)
    before = _native.launch_counts.get("cusrl_synthetic_env_step", 0)
    steps = [env.step(action) for _ in range(3)]
    assert _native.launch_counts["cusrl_synthetic_env_step"] - before == 3
    for next_obs, state, reward, terminated, truncated, info in steps:
        assert state is None and next_obs.shape == (65536, 48) and reward.shape == (65536, 1)
        assert terminated.dtype == torch.bool and terminated.shape == (65536, 1) == truncated.shape
        assert abs(next_obs.mean().item()) < 0.005 and abs(next_obs.var().item() - 1.0) < 0.01
        assert abs(reward.mean().item()) < 0.02 and abs(reward.var().item() - 1.0) < 0.03
        assert abs(terminated.float().mean().item() - 0.02) < 0.003 and abs(truncated.float().mean().item() - 0.01) < 0.002
        # neighbouring elements / channels are uncorrelated
        assert abs((next_obs[:, 0] * next_obs[:, 1]).mean().item()) < 0.02 and abs((next_obs[1:, 0] * next_obs[:-1, 0]).mean().item()) < 0.02
    assert not torch.equal(steps[0][0], steps[1][0]) and not torch.equal(steps[1][0], steps[2][0])
    rows, _, _ = env.reset_static(torch.zeros(65536, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV))
    assert rows.shape == (65536, 48) and abs(rows.var().item() - 1.0) < 0.01 and not torch.equal(rows, steps[2][0])
    again = build()
    assert torch.equal(again.step(action)[0], steps[0][0])  # same seed, same stream
    # replayed from a graph: every replay is a new step
    stream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        env.step(action)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        captured = env.step(action)[0]
    graph.replay()
    torch.cuda.synchronize()
    first = captured.clone()
    graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(first, captured) and abs(captured.var().item() - 1.0) < 0.01


class _ListLogger:
    """Collects every ``log(info, iteration)`` call (the trainer's logger protocol)."""

    def __init__(self):
        self.entries = []

    def log(self, info, iteration):
        self.entries.append((iteration, dict(info)))

    def save_checkpoint(self, checkpoint, iteration):
        self.entries.append(("checkpoint", iteration))


@pytest.mark.parametrize("kind", ["continuous", "amp"])
def test_pipelined_logging_reports_every_iteration_once_with_the_same_values(cusrl, kind):
    """Round 6: the trainer reads an iteration's log AFTER the next rollout has been launched (`Trainer.pipeline_logs`).  Against
    the unpipelined loop from the same seed: the same iterations in the same order, every value of every log equal — the timing
    keys aside —, checkpoints behind the log of their iteration, parameters bit-identical."""
    runs = {}
    for pipelined in (True, False):
        cusrl.set_global_seed(33)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=256, observation_dim=12, action_dim=4, device=DEV)
        logger = _ListLogger()
        trainer = cusrl.Trainer(env, _factory(cusrl, kind, 8), logger_factory=lambda logger=logger: logger, num_iterations=9,
                                checkpoint_interval=4, verbose=False)
        trainer.pipeline_logs = pipelined
        trainer.run_training_loop()
        assert trainer._pending_log is None
        runs[pipelined] = (logger.entries, [p.detach().clone() for p in trainer.agent.parameters()], trainer)
    piped, plain = runs[True][0], runs[False][0]
    order = [entry[0] if entry[0] != "checkpoint" else ("checkpoint", entry[1]) for entry in piped]
    assert order == [entry[0] if entry[0] != "checkpoint" else ("checkpoint", entry[1]) for entry in plain]
    assert order == [("checkpoint", 0), 1, 2, 3, 4, ("checkpoint", 4), 5, 6, 7, 8, ("checkpoint", 8), 9, ("checkpoint", 9)]
    assert runs[True][2]._graphed_rollout is not None and runs[True][2]._graphed_rollout.rollout_replays > 0  # (it did pipeline)
    for (it_a, a), (it_b, b) in zip([e for e in piped if e[0] != "checkpoint"], [e for e in plain if e[0] != "checkpoint"]):
        assert it_a == it_b and a.keys() == b.keys()
        for key in a:
            if key.startswith("Perf/") and key != "Perf/environment_step":
                assert a[key] > 0
            else:
                assert a[key] == b[key], (it_a, key)
    for p, q in zip(runs[True][1], runs[False][1]):
        assert torch.equal(p, q)


def test_direct_agent_update_still_returns_plain_floats(cusrl):
    """``agent.update()`` hands out a staged summary only to a trainer that asked for it (`deferred_summary`)."""
    cusrl.set_global_seed(5)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=64, observation_dim=12, action_dim=4, device=DEV)
    agent = _factory(cusrl, "continuous", 4).from_environment(env)
    observation, state, _ = env.reset()
    for _ in range(4):
        action = agent.act(observation, state)
        observation, state, reward, terminated, truncated, info = env.step(action)
        ready = agent.step(observation, reward, terminated, truncated, state, **info)
    assert ready
    summary = agent.update()
    assert isinstance(summary, dict) and summary and all(isinstance(v, float) for v in summary.values())
