"""Every BASELINE.json config at its per-GPU workload (VERDICT r01 "configs_untested"), each asserting that the HIP
entry points — not a torch-op form — produced the result (``_native.launch_counts``).

  config 1  MountainCar-v0 `ppo` kwargs, 8 envs, discrete / Tanh / obs-norm / T=16 / 4x4 minibatches: full update
            replayed against a trace recorded from the reference (golden ``update_trace_config1.npz``)
  config 2  4096 envs x obs 48 x act 12 (also tests/test_agent_gpu.py::test_full_size_iteration_...)
  config 3  8192 envs per GPU (65 536 over 8 ranks): the same full-size checks at 8192
  config 4  GRU recurrent PPO, 16 384 envs, T=24, BPTT minibatches of 4096 env columns
  config 5  RND + AMP hooks together, 4096 envs per GPU
"""

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cusrl():
    import cusrl_amd

    cusrl_amd.config.set_device(DEV)
    return cusrl_amd


def host(t):
    return t.detach().cpu().numpy()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


class Launches:
    """Difference of the C-ABI launch census across a block of code."""

    def __enter__(self):
        from cusrl_amd import _native

        self._counts = _native.launch_counts
        self._before = dict(self._counts)
        return self

    def __exit__(self, *exc):
        self.delta = {k: v - self._before.get(k, 0) for k, v in self._counts.items() if v != self._before.get(k, 0)}

    def __getitem__(self, name):
        return self.delta.get(name, 0)


# ------------------------------------------------------------------------------------------------ config 1
MOUNTAIN_CAR_KWARGS = dict(  # cusrl/zoo/gym/classic_control.py:65-80
    num_steps_per_update=16, actor_hidden_dims=(64, 64), critic_hidden_dims=(64, 64), activation_fn="Tanh",
    action_space_type="discrete", lr=3e-4, sampler_epochs=4, sampler_mini_batches=4, orthogonal_init=False,
    normalize_observation=True, gae_gamma=0.99, gae_lamda=0.98, entropy_loss_weight=0.0, max_grad_norm=0.5,
)


@pytest.mark.parametrize("mode", ["fused", "hook_by_hook", "hipgraph"])
def test_config1_mountain_car_update_replays_the_reference(cusrl, golden, mode, gradient_parity):
    g = golden("update_trace_config1")
    overrides = {"compile": True} if mode == "hipgraph" else {}
    underlying = cusrl.preset.PpoAgentFactory(**MOUNTAIN_CAR_KWARGS, device=DEV, **overrides).to_underlying()
    trace = {k: [] for k in ("objectives", "indices", "grads_unclipped", "grads", "params_after")}

    class Capture(cusrl.Hook):
        def __init__(self, where):
            super().__init__()
            self.where = where
            self.name_(f"capture_{where}")

        def pre_optim(self, optimizer):
            flat = torch.cat([p.grad.reshape(-1) for group in optimizer.param_groups for p in group["params"]])
            trace["grads_unclipped" if self.where == "pre" else "grads"].append(flat.clone())

        def post_optim(self):
            if self.where == "post":
                trace["params_after"].append(torch.cat([p.detach().reshape(-1) for _, p in self.agent.named_parameters()]))

        def post_objective(self, metadata, batch):
            if self.where == "post":
                trace["indices"].append(batch["flat_index"].squeeze(-1).clone())

    underlying.register_hook(Capture("pre"), before="gradient_clipping")
    underlying.register_hook(Capture("post"), after="gradient_clipping")
    agent = underlying(cusrl.EnvironmentSpec(2, 3, num_instances=8, device=DEV))
    from cusrl_amd.hook.on_policy.fused import FusedPpoObjective

    assert FusedPpoObjective.eligible(agent.hook)  # the categorical objective has a fused form too
    agent.fuse_objective = mode != "hook_by_hook"
    named = dict(agent.named_parameters())
    assert list(named) == [str(n) for n in g["param_names"]]
    with torch.no_grad():
        for name, param in named.items():
            param.copy_(dev(g[f"param0/{name}"]))
    agent.sampler.permutation_device = "cpu"  # the CPU reference's mt19937 stream on the GPU buffer
    agent.hook["on_policy_statistics"].sampler.permutation_device = "cpu"
    leaves = {str(k): g[f"buffer_in/{k}"] for k in g["buffer_keys"]}
    for t in range(16):
        step = {k: dev(v[t]) for k, v in leaves.items() if not k.startswith("action_dist.")}
        step["action_dist"] = {"logits": dev(leaves["action_dist.logits"][t])}
        agent.buffer.push(step)
    assert agent.buffer.full and agent.buffer.cursor == 0
    original = agent.hook.objective

    def wrapped(metadata, batch):
        result = original(metadata, batch)
        if result["value_loss"] is not None:  # None while a step is being captured: deferred finalize (ops.DeferredLoss)
            trace["objectives"].append(torch.stack([result["value_loss"], result["surrogate_loss"], result["entropy_loss"]]).detach())
        return result

    agent.hook.objective = wrapped
    torch.manual_seed(99)
    with Launches() as launched:
        metrics = agent.update()
    # which reference train steps the Python-side trace holds: all 16 when eager; under compile=True only epoch 0 (eager
    # warm-up) and epoch 1 (capture) execute Python, and the tensors taken during the capture live in the graph's pool, so
    # after the replays of epochs 2-3 they show the LAST epoch's values — i.e. reference steps 0-3 and 12-15
    steps = list(range(16)) if mode != "hipgraph" else [0, 1, 2, 3, 12, 13, 14, 15]
    assert len(trace["indices"]) == len(steps)
    assert np.array_equal(host(torch.stack(trace["indices"])), g["indices"][steps]), "minibatch permutations differ"
    for key in ("next_value", "advantage", "return"):
        np.testing.assert_allclose(host(agent.buffer[key]), g[f"buffer_out/{key}"], rtol=1e-5, atol=2e-6)
    seen = len(trace["objectives"])  # under compile=True: the eager warm-up epoch (captured steps defer their loss values)
    assert seen == (16 if mode != "hipgraph" else 4)
    np.testing.assert_allclose(host(torch.stack(trace["objectives"])), g["objectives"][steps[:seen]], rtol=2e-5, atol=1e-6)
    kept = [(row, steps.index(int(step))) for row, step in enumerate(g["kept_steps"]) if int(step) in steps]
    rows = [row for row, _ in kept]
    pick = lambda name: host(torch.stack([trace[name][i] for _, i in kept]))  # noqa: E731
    clipped = g["grads_unclipped" if agent.flat_optimizer is not None else "grads"]
    for (row, _), raw, after in zip(kept, pick("grads_unclipped"), pick("grads")):  # 1e-5 of each step's largest entry
        gradient_parity(f"config1_trace.grads_unclipped[{mode},{row}]", raw, g["grads_unclipped"][row], 1e-5)
        gradient_parity(f"config1_trace.grads[{mode},{row}]", after, clipped[row], 1e-5)
    np.testing.assert_allclose(pick("params_after"), g["params_after"][rows], rtol=1e-4, atol=2e-6)
    final = torch.cat([p.detach().reshape(-1) for _, p in agent.named_parameters()])
    np.testing.assert_allclose(host(final), g["params_after"][-1], rtol=1e-4, atol=2e-6)  # end state of all 16 steps
    ref = dict(zip((str(k) for k in g["metric_keys"]), g["metric_vals"]))
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/entropy_loss", "Agent/entropy", "Agent/value",
                "Agent/grad_norm/default", "Agent/ratio", "Agent/kl_divergence", "Agent/importance_weighted_advantage"):
        np.testing.assert_allclose(metrics[key], ref[key], rtol=1e-3, atol=1e-5, err_msg=key)
    # the HIP entry points carried it
    assert launched["cusrl_next_value"] == 1 and launched["cusrl_gae"] == 1 and launched["cusrl_normalize_from_partials"] == 1
    assert launched["cusrl_categorical_policy_stats"] == 1
    gathers = launched["cusrl_gather_rows"] + launched["cusrl_gather_rows_packed"]
    if mode == "fused":
        assert launched["cusrl_ppo_loss_categorical_fwd_bwd"] == 16 and gathers >= 16
    elif mode == "hook_by_hook":
        assert launched["cusrl_ppo_loss_categorical_fwd_bwd"] == 0 and gathers >= 16
    else:  # 4 minibatch slots: one eager warm-up + one capture each, the other 8 steps are replays
        assert launched["cusrl_ppo_loss_categorical_fwd_bwd"] == 8


def test_config1_rollout_with_observation_normalisation_runs_on_hip(cusrl):
    """8 envs of MountainCar's shapes through Trainer: the observation statistics / normalisation kernels run every step."""
    cusrl.set_global_seed(4)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=8, observation_dim=2, action_dim=3, device=DEV)
    trainer = cusrl.Trainer(env, cusrl.preset.PpoAgentFactory(**MOUNTAIN_CAR_KWARGS), num_iterations=2, verbose=False)
    with Launches() as launched:
        trainer.run_training_loop()
    assert launched["cusrl_masked_col_stats"] >= 2 * 16 and launched["cusrl_rms_merge"] >= 2 * 16
    assert launched["cusrl_rms_normalize"] >= 2 * 2 * 16 and launched["cusrl_buffer_push"] == 2 * 16
    assert launched["cusrl_ppo_loss_categorical_fwd_bwd"] == 2 * 16
    assert launched["cusrl_categorical_sample_logp"] == 2 * 16  # one draw launch per env step
    buffer = trainer.agent.buffer
    assert {"original_observation", "original_next_observation"} <= set(buffer.storage) and buffer["action"].shape == (16, 8, 3)
    assert torch.equal(buffer["action"].sum(-1), torch.ones(16, 8, device=DEV))  # one-hot actions
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/kl_divergence"):
        assert np.isfinite(trainer.last_info[key])


def test_config1_compiled_captures_the_discrete_act_step_and_trains_like_eager(cusrl):
    """compile=True on the MountainCar shape: the act step of the one-hot categorical policy (observation normalisation,
    actor, the HIP draw) replays from a hipGraph, the minibatch steps from theirs, and four iterations end at the
    parameters of the eager run (same generator stream: the race variables are drawn by torch inside the capture)."""
    finals = []
    for compile_ in (False, True):
        cusrl.set_global_seed(9)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=8, observation_dim=2, action_dim=3, device=DEV)
        factory = cusrl.preset.PpoAgentFactory(**MOUNTAIN_CAR_KWARGS, compile=compile_,
                                               optimizer_kwargs={"capturable": True, "fused": True})
        trainer = cusrl.Trainer(env, factory, num_iterations=4, verbose=False)
        trainer.run_training_loop()
        agent = trainer.agent
        if compile_:
            assert agent._graphed_act.state == 2, "the discrete act step was not captured"
            assert agent._graphed_steps and all(step.state == 2 for step in agent._graphed_steps.values())
        assert torch.equal(agent.buffer["action"].sum(-1), torch.ones(16, 8, device=DEV))
        finals.append(torch.cat([p.detach().reshape(-1) for p in agent.parameters()]))
    assert torch.allclose(finals[0], finals[1], rtol=1e-4, atol=1e-5), (finals[0] - finals[1]).abs().max()


# ------------------------------------------------------------------------------------------------ configs 2 / 3
@pytest.mark.parametrize("num_envs", [4096, 8192])
def test_config2_and_3_full_size_iteration(cusrl, num_envs):
    """One iteration of the `ppo` preset at 4096 (config 2) and 8192 (config 3's per-GPU share) envs: pre_update on the
    real buffer vs the oracle at full size — return bit-exact, normalised advantage 1e-5 — and every slot is gathered
    exactly once per epoch.  These buffers (25 / 50 MB of sampled leaves) sit in L2 + Infinity Cache, where the plain
    per-leaf gather is the faster form: config 2 runs it as the product does (no record, no pack launch); config 3's share
    is run with the record forced on (threshold 0) so the packed path is exercised at a BASELINE size too."""
    cusrl.set_global_seed(1)
    env = cusrl.testing.SyntheticEnvironment(num_envs, 48, 12, device=DEV)
    trainer = cusrl.Trainer(env, cusrl.preset.PpoAgentFactory(), num_iterations=1, verbose=False)
    record = num_envs == 8192
    if record:
        trainer.agent.buffer.record_threshold_bytes = 0
    with Launches() as launched:
        trainer.run_training_loop()
    assert launched["cusrl_buffer_push"] == 24 and launched["cusrl_gae"] == 1 and launched["cusrl_ppo_loss_fwd_bwd"] == 20
    assert launched["cusrl_step_epilogue"] == 24 and launched["cusrl_episode_stats"] == 0  # one launch per env step
    assert launched["cusrl_policy_stats"] == 1
    if record:
        # (the update-time leaves fill the record's last chunk on their own: the pack stores it whole)
        assert launched["cusrl_gather_rows_packed"] >= 20 and launched["cusrl_pack_rows_owned"] + launched["cusrl_pack_rows"] >= 1
    else:
        assert launched["cusrl_gather_rows"] >= 20 and launched["cusrl_gather_rows_packed"] == 0
        assert launched["cusrl_pack_rows"] == launched["cusrl_pack_rows_owned"] == 0
        assert trainer.agent.buffer._pack is None
    buffer = trainer.agent.buffer
    buffer.record_threshold_bytes = 0  # the sampling checks below go through the record at both sizes
    S = 24 * num_envs
    h = {k: host(buffer[k]) for k in ("reward", "value", "next_value", "done", "advantage", "return", "terminated", "truncated")}
    adv, ret = oracle.gae(h["reward"], h["done"], h["value"], h["next_value"], 0.99, 0.95)
    assert np.array_equal(ret, h["return"])                                                  # bit-exact
    var, mean = oracle.var_mean(adv)
    np.testing.assert_allclose(oracle.normalize(adv, mean, var), h["advantage"], rtol=1e-5, atol=1e-5)
    keep = ~(h["terminated"] | h["truncated"])[:-1]
    assert np.array_equal(h["next_value"][:-1][keep], h["value"][1:][keep])
    assert not h["next_value"][h["terminated"] & ~h["truncated"]].any()
    seen = torch.zeros(S, dtype=torch.int32, device=DEV)
    buffer["slot"] = torch.arange(S, device=DEV).view(24, num_envs, 1)
    sampler = cusrl.MiniBatchSampler(2, 4)
    for metadata, batch in sampler(buffer):
        slots = batch["slot"].squeeze(-1)
        seen[slots] += 1
        assert batch["observation"].shape == (S // 4, 48) and batch["done"].dtype == torch.bool
        assert torch.equal(batch["reward"], buffer["reward"].flatten(0, 1)[slots])           # through the packed record
        assert torch.equal(batch["done"], buffer["done"].flatten(0, 1)[slots])
    assert bool((seen == 2).all())
    # the next pass of the same consumer finds a record holding exactly the fields it read (observation 192 B, the int64
    # slot as two 4-byte entries, reward, done = 205 B -> 256 B: two memory lines per sampled slot) and gets the same rows
    assert sampler.hot_fields >= {"slot", "observation", "reward", "done"}
    seen.zero_()
    for metadata, batch in sampler(buffer):
        assert set(buffer._pack.leaves) == {"observation", "slot", "reward", "done"} and buffer._pack.record_bytes == 256
        slots = batch["slot"].squeeze(-1)
        seen[slots] += 1
        assert torch.equal(batch["observation"], buffer.storage["observation"].flatten(0, 1)[slots])
        assert torch.equal(batch["reward"], buffer.storage["reward"].flatten(0, 1)[slots])
        assert torch.equal(batch["done"], buffer.storage["done"].flatten(0, 1)[slots])
    assert bool((seen == 2).all())


# ------------------------------------------------------------------------------------------------ config 4
def test_config4_recurrent_gru_16384_envs(cusrl):
    """`RecurrentPpoAgentFactory(rnn_type="GRU")` defaults (2 x 256 both nets) on 16 384 envs, T = 24, one iteration:
    memory leaves [24, 16384, 512], temporal minibatches of 4096 env columns, done-split layout vs the oracle's numpy
    restatement at full size, return bit-exact vs the oracle."""
    from cusrl_amd.nn import recurrent as R

    cusrl.set_global_seed(3)
    N, T = 16384, 24
    env = cusrl.testing.SyntheticEnvironment(N, 48, 12, device=DEV)
    shapes = {}

    class Probe(cusrl.Hook):
        def objective(self, metadata, batch):
            if not shapes:
                assert metadata["temporal"] is True and metadata["total_mini_batches"] == 4
                for key in ("observation", "actor_memory", "critic_memory", "done", "advantage"):
                    shapes[key] = tuple(batch[key].shape)
                shapes["done_tensor"] = batch["done"].clone()
                shapes["observation_tensor"] = batch["observation"].clone()

    factory = cusrl.preset.RecurrentPpoAgentFactory(rnn_type="GRU", sampler_epochs=1, optimizer_kwargs={"fused": True}).to_underlying()
    factory.register_hook(Probe())
    trainer = cusrl.Trainer(env, factory, num_iterations=1, verbose=False)
    with Launches() as launched:
        trainer.run_training_loop()
    buffer = trainer.agent.buffer
    for key in ("actor_memory", "critic_memory", "next_critic_memory"):
        assert buffer[key].shape == (T, N, 512), key
    assert shapes["observation"] == (T, N // 4, 48) and shapes["actor_memory"] == (T, N // 4, 512)
    assert shapes["critic_memory"] == (T, N // 4, 512) and shapes["done"] == (T, N // 4, 1)
    # 4 minibatches x (actor + critic) + the statistics pass: every sequence forward goes through the layout kernels
    assert launched["cusrl_sequence_layout"] >= 9 and launched["cusrl_sequence_count"] >= 9
    assert launched["cusrl_gather_rows"] + launched["cusrl_gather_rows_packed"] >= 4 and launched["cusrl_gae"] == 1
    h = {k: host(buffer[k]) for k in ("reward", "value", "next_value", "done")}
    _, ret = oracle.gae(h["reward"], h["done"], h["value"], h["next_value"], 0.99, 0.95)
    assert np.array_equal(ret, host(buffer["return"]))
    # layout of the first minibatch's [24, 4096] done flags: HIP vs the numpy restatement, bit-exact at full size
    done = shapes["done_tensor"]
    layout = R.compute_sequence_layout(done)
    dest, num_sequences = oracle.sequence_layout(host(done))
    assert layout.num_sequences == num_sequences and np.array_equal(host(layout.dest), dest)
    assert np.array_equal(host(layout.lengths), oracle.sequence_lengths(host(done)))
    padded, mask = R.split_and_pad_sequences(shapes["observation_tensor"], done, layout)
    assert padded.shape == (T, num_sequences, 48) and int(mask.sum()) == T * (N // 4)
    assert torch.equal(R.unpad_and_merge_sequences(padded, layout), shapes["observation_tensor"])
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/kl_divergence"):
        assert np.isfinite(trainer.last_info[key]), key


# ------------------------------------------------------------------------------------------------ config 5
def test_config5_rnd_and_amp_together_4096_envs(cusrl):
    """RND (cusrl_test/integration/test_agent_state_dict.py:6-19 wiring) AND AMP (preset/amp.py:12-53) on 4096 envs: both
    reward epilogues run as HIP launches, the buffer's reward leaf equals a host recomputation of
    extrinsic * scale + style reward (per step) + RND bonus (at pre_update), and the update trains."""
    cusrl.set_global_seed(8)
    N, T, k = 4096, 24, 6
    env = cusrl.testing.SyntheticEnvironment(N, 48, 12, device=DEV)
    dataset = torch.randn(100_000, 2 * k, device=DEV)
    factory = cusrl.preset.AmpAgentFactory(amp_dataset_source=dataset, amp_state_indices=slice(k), extrinsic_reward_scale=0.5,
                                           amp_reward_scale=2.0).to_underlying()
    factory.register_hook(cusrl.hook.RandomNetworkDistillation(module_factory=cusrl.Mlp.Factory(hidden_dims=[128, 64]),
                                                               output_dim=16, reward_scale=0.1), before="value_computation")
    names = [hook.name for hook in factory.hooks]
    assert names.index("random_network_distillation") < names.index("value_computation")
    assert names.index("reward_shaping") < names.index("adversarial_motion_prior") < names.index("value_computation")
    recorded = {"extrinsic": [], "style": []}

    class Tap(cusrl.Hook):  # sits right after AMP: sees the reward after shaping + style bonus of every env step
        def post_step(self, transition):
            recorded["style"].append(transition["reward"].clone())

    class Raw(cusrl.Hook):  # first hook: the env's own reward
        def post_step(self, transition):
            recorded["extrinsic"].append(transition["reward"].clone())

    factory.register_hook(Raw(), index=0)
    factory.register_hook(Tap(), after="adversarial_motion_prior")
    trainer = cusrl.Trainer(env, factory, num_iterations=1, verbose=False)
    agent = trainer.agent
    rnd, amp = agent.hook["random_network_distillation"], agent.hook["adversarial_motion_prior"]
    snapshot = {}

    class BeforeUpdate(cusrl.Hook):
        def pre_update(self, buffer):  # registered before RND: the buffer as the rollout left it
            snapshot["reward"] = buffer["reward"].clone()
            snapshot["next_observation"] = buffer["next_observation"].clone()
            snapshot["predictor"] = {n: p.detach().clone() for n, p in rnd.predictor.named_parameters()}

    before = BeforeUpdate()
    before.pre_init(agent)
    original_pre_update = agent.hook.pre_update

    def pre_update(buffer):
        before.pre_update(buffer)
        original_pre_update(buffer)
        snapshot["reward_after"] = buffer["reward"].clone()

    agent.hook.pre_update = pre_update
    with Launches() as launched:
        trainer.run_training_loop()
    # per env step: ONE launch for the transition assembly + dataset rows + both statistics updates + both normalisations,
    # one for the style reward and its recorded mean, one for the reward shaping (scale 0.5); RND's bonus once per update
    assert launched["cusrl_amp_prepare"] == T and launched["cusrl_amp_style_reward_mean"] == T
    assert launched["cusrl_reward_shaping"] == T and launched["cusrl_rnd_reward"] == 1
    assert launched["cusrl_mse_loss_fwd_bwd"] > 0  # RND's objective: squared error forward + backward in one pass
    assert amp.transition_rms.count == 2 * N * T  # agent + expert rows of every step went into the running statistics
    assert {"agent_transition", "expert_transition"} <= set(agent.buffer.storage)
    assert agent.buffer["agent_transition"].shape == (T, N, 2 * k)
    # per-step: reward = extrinsic * 0.5 + style, style >= 0 bounded by -log(1e-4) * scale
    extrinsic, shaped = torch.stack(recorded["extrinsic"]), torch.stack(recorded["style"])
    style = shaped - 0.5 * extrinsic
    assert float(style.min()) > -1e-5 and float(style.max()) <= 2.0 * 9.2104 + 1e-3
    assert torch.equal(snapshot["reward"], shaped)  # what the rollout pushed
    # pre_update: reward += 0.1 * mean((target - predictor)(next_observation)^2), recomputed on the host in fp64
    with torch.no_grad():
        flat = snapshot["next_observation"].reshape(-1, 48)
        live = {n: p.detach().clone() for n, p in rnd.predictor.named_parameters()}
        for n, p in rnd.predictor.named_parameters():
            p.data.copy_(snapshot["predictor"][n])
        bonus = 0.1 * (rnd.target(flat).double() - rnd.predictor(flat).double()).square().mean(-1, keepdim=True)
        for n, p in rnd.predictor.named_parameters():
            p.data.copy_(live[n])
    expect = snapshot["reward"].double() + bonus.view(T, N, 1)
    torch.testing.assert_close(snapshot["reward_after"].double(), expect, rtol=1e-5, atol=1e-6)
    info = trainer.last_info
    for key in ("Agent/rnd_loss", "Agent/amp_discrimination_loss", "Agent/amp_grad_penalty_loss", "Agent/rnd_reward",
                "Agent/amp_reward", "Agent/value_loss", "Agent/surrogate_loss"):
        assert key in info and np.isfinite(info[key]), key
