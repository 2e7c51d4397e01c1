"""RND and AMP against values recorded from the reference hooks (golden ``aux_rewards.npz``) and
the reference's own integration tests (cusrl_test/hook/auxiliary/test_rnd.py, test_amp.py)."""

from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cusrl_amd as cusrl


def fake_agent(device, records):
    return SimpleNamespace(state_dim=10, device=torch.device(device), setup_module=lambda m: m.to(device),
                           to_tensor=lambda v: torch.as_tensor(v, device=device),
                           record=lambda **kw: records.update({k: v.clone() for k, v in kw.items()}),
                           environment_spec=SimpleNamespace(demonstration_sampler=None))


def load(module, g, prefix):
    module.load_state_dict({k: torch.from_numpy(g[f"{prefix}/{k}"].copy()) for k in module.state_dict()})


def run_rnd(g, device):
    records = {}
    rnd = cusrl.hook.RandomNetworkDistillation(cusrl.Mlp.Factory([12, 8]), output_dim=5, reward_scale=0.25, state_indices=slice(2, 9))
    rnd.pre_init(fake_agent(device, records))
    rnd.init()
    load(rnd.target, g, "rnd_target"), load(rnd.predictor, g, "rnd_predictor")
    buffer = {"next_observation": torch.from_numpy(g["rnd_next_observation"].copy()).to(device),
              "reward": torch.from_numpy(g["rnd_reward_in"].copy()).to(device)}
    rnd.pre_update(buffer)
    np.testing.assert_allclose(buffer["reward"].cpu().numpy(), g["rnd_reward_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(records["rnd_reward"].cpu().numpy(), g["rnd_reward_metric"], rtol=1e-5, atol=1e-7)
    loss = rnd.objective({}, {"next_observation": buffer["next_observation"].flatten(0, 1)})["rnd_loss"]
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["rnd_loss"], rtol=1e-5)
    grad = torch.cat([p.grad.reshape(-1) for p in rnd.predictor.parameters()])
    np.testing.assert_allclose(grad.cpu().numpy(), g["rnd_grad"], rtol=1e-4, atol=1e-7)
    assert all(not p.requires_grad for p in rnd.target.parameters())


def run_amp(g, device):
    records = {}
    amp = cusrl.hook.AdversarialMotionPrior(cusrl.Mlp.Factory([16, 8]), dataset_source=g["amp_dataset"].copy(), state_indices=slice(1, 5),
                                            batch_size=None, reward_scale=0.5, loss_weight=2.0, grad_penalty_weight=5.0)
    amp.pre_init(fake_agent(device, records))
    amp.init()
    load(amp.discriminator, g, "amp_discriminator")
    dev = lambda name: torch.from_numpy(g[name].copy()).to(device)  # noqa: E731
    for t in range(int(g["amp_steps"])):
        tr = {"observation": dev(f"amp_obs_{t}"), "next_observation": dev(f"amp_next_obs_{t}"), "reward": dev(f"amp_reward_in_{t}")}
        # the reference draws expert rows with torch.randint from the CPU generator: replay its picks on any device
        torch.manual_seed(100 + t)
        picks = torch.randint(50, (7,))
        amp._sample_demonstration = lambda n, _p=picks: amp.dataset[_p.to(amp.dataset.device)]
        amp.post_step(tr)
        np.testing.assert_allclose(tr["agent_transition"].cpu().numpy(), g[f"amp_agent_transition_{t}"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(tr["expert_transition"].cpu().numpy(), g[f"amp_expert_transition_{t}"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(tr["reward"].cpu().numpy(), g[f"amp_reward_out_{t}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(amp.transition_rms.mean.cpu().numpy(), g[f"amp_rms_mean_{t}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(amp.transition_rms.var.cpu().numpy(), g[f"amp_rms_var_{t}"], rtol=1e-5, atol=1e-6)
    losses = amp.objective({}, {"agent_transition": tr["agent_transition"].clone(), "expert_transition": tr["expert_transition"].clone()})
    sum(losses.values()).backward()
    np.testing.assert_allclose(losses["amp_discrimination_loss"].item(), g["amp_discrimination_loss"], rtol=1e-5)
    np.testing.assert_allclose(losses["amp_grad_penalty_loss"].item(), g["amp_grad_penalty_loss"], rtol=1e-4)
    grad = torch.cat([p.grad.reshape(-1) for p in amp.discriminator.parameters()])
    np.testing.assert_allclose(grad.cpu().numpy(), g["amp_grad"], rtol=1e-3, atol=1e-6)


def test_rnd_and_amp_hook_logic_matches_reference_on_host(golden):
    run_rnd(golden("aux_rewards"), "cpu")
    run_amp(golden("aux_rewards"), "cpu")


@pytest.mark.gpu
def test_rnd_and_amp_with_hip_reward_epilogues_match_reference(golden):
    run_rnd(golden("aux_rewards"), "cuda:0")
    run_amp(golden("aux_rewards"), "cuda:0")


@pytest.mark.gpu
def test_host_forms_are_refused_in_gpu_tests(golden):
    """In a process that runs ``gpu`` tests the gate in front of the hooks' torch-op (host) forms is SHUT (tests/conftest.py):
    each of the four places where a CPU tensor could otherwise quietly take the reference's op chain raises — the same
    behaviour a user's process gets (cusrl_amd/utils/misc.py host_form)."""
    import os

    assert os.environ.get("CUSRL_HOST_FORMS") != "1"
    g = golden("aux_rewards")
    with pytest.raises(RuntimeError, match="RandomNetworkDistillation.pre_update received CPU tensors"):
        run_rnd(g, "cpu")
    with pytest.raises(RuntimeError, match="received CPU tensors"):  # (its running statistics come first: RunningMeanStd)
        run_amp(g, "cpu")
    shaping = cusrl.hook.RewardShaping(scale=2.0, shift=0.5)
    with pytest.raises(RuntimeError, match="RewardShaping.post_step received CPU tensors"):
        shaping.post_step({"reward": torch.ones(4, 1)})
    from cusrl_amd.nn.rms import mean_var_count

    with pytest.raises(RuntimeError, match="RunningMeanStd .mean_var_count. received CPU tensors"):
        mean_var_count(torch.ones(4, 3))
    # AMP's own site, reached with the statistics switched off the host path: the style reward of CPU logits
    amp = cusrl.hook.AdversarialMotionPrior(cusrl.Mlp.Factory([16, 8]), dataset_source=g["amp_dataset"].copy(), state_indices=slice(1, 5),
                                            batch_size=None, reward_scale=0.5, loss_weight=2.0, grad_penalty_weight=5.0)
    amp.pre_init(fake_agent("cpu", {}))
    amp.init()
    amp.transition_rms.update = lambda *a, **k: None
    amp.transition_rms.normalize = lambda x: x
    amp._sample_demonstration = lambda n: amp.dataset[:7]
    with pytest.raises(RuntimeError, match="AdversarialMotionPrior.post_step received CPU tensors"):
        amp.post_step({"observation": torch.from_numpy(g["amp_obs_0"].copy()), "next_observation": torch.from_numpy(g["amp_next_obs_0"].copy()),
                       "reward": torch.from_numpy(g["amp_reward_in_0"].copy())})


@pytest.mark.gpu
def test_reward_epilogue_kernels_vs_formula():
    from cusrl_amd import ops

    dev = "cuda:0"
    target, prediction = torch.randn(24 * 4096, 16, device=dev), torch.randn(24 * 4096, 16, device=dev)
    reward = torch.randn(24, 4096, 1, device=dev)
    expect = reward + 0.1 * (target - prediction).square().mean(-1, keepdim=True).view(24, 4096, 1)
    bonus = ops.rnd_reward_(reward, target, prediction, 0.1)
    torch.testing.assert_close(reward, expect, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bonus, expect - (expect - bonus), rtol=1e-5, atol=1e-6)
    logit = torch.randn(4096, 1, device=dev) * 6  # reaches the 1e-4 clamp on the positive side
    reward = torch.randn(4096, 1, device=dev)
    expect = reward + 2.0 * -torch.log(torch.clamp(1 - 1 / (1 + torch.exp(-logit)), min=1e-4))
    ops.amp_style_reward_(reward, logit, 2.0)
    torch.testing.assert_close(reward, expect, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("with_state", [False, True])
def test_rnd_in_the_ppo_preset(with_state):  # cusrl_test/hook/auxiliary/test_rnd.py:9-26
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(0)
    factory = cusrl.preset.PpoAgentFactory(num_steps_per_update=8, sampler_epochs=2, sampler_mini_batches=2).to_underlying()
    factory.register_hook(cusrl.hook.RandomNetworkDistillation(cusrl.Mlp.Factory([128, 128]), output_dim=16, reward_scale=0.1),
                          before="value_computation")
    env = cusrl.testing.DummyTorchEnvironment(num_instances=8, observation_dim=16, action_dim=8,
                                              state_dim=24 if with_state else None, device="cuda:0")
    trainer = cusrl.Trainer(env, factory, num_iterations=3, verbose=False)
    trainer.run_training_loop()
    info = trainer.last_info
    assert np.isfinite(info["Agent/rnd_loss"]) and info["Agent/rnd_reward"] > 0 and np.isfinite(info["Agent/value_loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("compile_", [False, True])
def test_amp_preset(compile_):  # cusrl_test/hook/auxiliary/test_amp.py:9-18
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(0)
    factory = cusrl.preset.AmpAgentFactory(amp_dataset_source=partial(torch.randn, 100, 16), amp_state_indices=slice(16, 24),
                                           num_steps_per_update=8, sampler_epochs=2, sampler_mini_batches=2, compile=compile_,
                                           optimizer_kwargs={"capturable": True, "fused": True})
    env = cusrl.testing.DummyTorchEnvironment(num_instances=8, observation_dim=16, action_dim=8, state_dim=24, device="cuda:0")
    trainer = cusrl.Trainer(env, factory, num_iterations=4, verbose=False)
    trainer.run_training_loop()
    info = trainer.last_info
    for key in ("Agent/amp_discrimination_loss", "Agent/amp_grad_penalty_loss", "Agent/amp_reward", "Agent/surrogate_loss"):
        assert np.isfinite(info[key]), key
    assert {"agent_transition", "expert_transition"} <= set(trainer.agent.buffer.storage)


# ------------------------------------------------------------------------------------------------ round 4: one-launch forms
def test_amp_prepare_restatement_reproduces_the_reference_hook(golden):
    """oracle.amp_prepare / oracle.amp_style_reward against what the reference's AdversarialMotionPrior.post_step recorded
    (aux_rewards.npz): normalised transitions, running statistics and shaped rewards of every step."""
    import oracle

    g = golden("aux_rewards")
    dataset, columns = g["amp_dataset"], np.arange(1, 5)
    mean, var, count = np.zeros(8, np.float32), np.ones(8, np.float32), 0
    for t in range(int(g["amp_steps"])):
        torch.manual_seed(100 + t)
        picks = torch.randint(50, (7,)).numpy()
        agent, expert, mean, var, std, count = oracle.amp_prepare(g[f"amp_obs_{t}"], g[f"amp_next_obs_{t}"], columns, dataset, picks,
                                                                  mean, var, count)
        np.testing.assert_allclose(agent, g[f"amp_agent_transition_{t}"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(expert, g[f"amp_expert_transition_{t}"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(mean, g[f"amp_rms_mean_{t}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var, g[f"amp_rms_var_{t}"], rtol=1e-5, atol=1e-6)


def test_reward_shaping_restatement_matches_the_reference_hook_formula():
    import oracle

    reward = torch.randn(5, 7, 1)
    expect = reward.clone().mul_(1.7).add_(-0.3).clamp_(min=-1.0, max=None)
    assert np.array_equal(oracle.reward_shaping(reward.numpy(), 1.7, -0.3, -1.0, None), expect.numpy())
    expect = reward.clone().mul_(0.5).add_(0.25).clamp_(min=None, max=0.4)
    assert np.array_equal(oracle.reward_shaping(reward.numpy(), 0.5, 0.25, None, 0.4), expect.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,state_dim,columns", [(7, 10, slice(1, 5)), (4096, 48, slice(6)), (300, 16, [15, 2, 7])])
def test_amp_prepare_kernel_vs_oracle(rows, state_dim, columns):
    """cusrl_amp_prepare — assembly, dataset rows, two statistics updates, two normalisations in ONE launch — against the
    numpy restatement over three consecutive steps (the running statistics carry over)."""
    import oracle
    from cusrl_amd import ops
    from cusrl_amd.nn.rms import RunningMeanStd

    dev = "cuda:0"
    rng = np.random.default_rng(rows)
    picked = np.arange(state_dim)[columns]
    C = 2 * len(picked)
    dataset = rng.standard_normal((1000, C)).astype(np.float32) * 3 + 1
    rms = RunningMeanStd(C).to(dev)
    mean, var, count = np.zeros(C, np.float32), np.ones(C, np.float32), 0
    prefix = len(picked) if np.array_equal(picked, np.arange(len(picked))) else None
    cols = None if prefix is not None else torch.as_tensor(picked, dtype=torch.int32, device=dev)
    for step in range(3):
        state, nxt = rng.standard_normal((rows, state_dim)).astype(np.float32), rng.standard_normal((rows, state_dim)).astype(np.float32)
        picks = rng.integers(0, 1000, rows)
        agent, expert = ops.amp_prepare(rms, state=torch.from_numpy(state).to(dev), next_state=torch.from_numpy(nxt).to(dev), columns=cols,
                                        width=prefix, dataset=torch.from_numpy(dataset).to(dev), indices=torch.from_numpy(picks).to(dev))
        ref_agent, ref_expert, mean, var, std, count = oracle.amp_prepare(state, nxt, picked, dataset, picks, mean, var, count)
        np.testing.assert_allclose(agent.cpu().numpy(), ref_agent, rtol=1e-5, atol=2e-5)  # 1e-5 rel fp32
        np.testing.assert_allclose(expert.cpu().numpy(), ref_expert, rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(rms.mean.cpu().numpy(), mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rms.var.cpu().numpy(), var, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rms.std.cpu().numpy(), std, rtol=1e-5, atol=1e-6)
        assert rms.count == count


@pytest.mark.gpu
def test_reward_shaping_style_mean_and_mse_kernels_vs_oracle():
    import oracle
    from cusrl_amd import ops

    dev = "cuda:0"
    reward = torch.randn(4096, 1, device=dev)
    for args in ((1.7, -0.3, -1.0, None), (0.5, 0.25, None, 0.4), (1.0, 0.0, -0.2, 0.2), (2.0, 1.0, None, None)):
        got = ops.reward_shaping_(reward.clone(), *args)
        assert np.array_equal(got.cpu().numpy(), oracle.reward_shaping(reward.cpu().numpy(), *args)), args  # bit-exact
    hook = cusrl.hook.RewardShaping(scale=1.7, shift=-0.3, lower_bound=-1.0)
    transition = {"reward": reward.clone()}
    hook.post_step(transition)
    assert np.array_equal(transition["reward"].cpu().numpy(), oracle.reward_shaping(reward.cpu().numpy(), 1.7, -0.3, -1.0, None))
    logit = torch.randn(4096, 1, device=dev) * 6
    base = torch.randn(4096, 1, device=dev)
    target = base.clone()
    bonus, mean = ops.amp_style_reward_mean_(target, logit, 2.0)
    expect = oracle.amp_style_reward(logit.cpu().numpy(), 2.0)
    np.testing.assert_allclose(bonus.cpu().numpy(), expect, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(target.cpu().numpy(), base.cpu().numpy() + expect, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(mean.item(), expect.astype(np.float64).mean(), rtol=1e-5)
    for shape in ((24576, 16), (1000, 5), (3,)):
        prediction, tgt = torch.randn(*shape, device=dev), torch.randn(*shape, device=dev)
        loss, grad = ops.mse_loss_fwd_bwd(prediction, tgt)
        ref_loss, ref_grad = oracle.mse_loss(prediction.cpu().numpy(), tgt.cpu().numpy())
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
        np.testing.assert_allclose(grad.cpu().numpy(), ref_grad, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_bce_pair_and_sumsq_kernels_vs_torch_formulas():
    from cusrl_amd import ops

    dev = "cuda:0"
    for rows in (1, 512, 5000):
        logit = (torch.randn(2 * rows, 1, device=dev) * 4).requires_grad_()
        target = torch.cat((torch.zeros(rows, 1, device=dev), torch.ones(rows, 1, device=dev)))
        ref = torch.nn.functional.binary_cross_entropy_with_logits(logit.double(), target.double()) * 2.5
        (ref_grad,) = torch.autograd.grad(ref, logit)
        loss, grad = ops.bce_pair_fwd_bwd(logit.detach(), 2.5)
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)  # 1e-5 rel fp32
        np.testing.assert_allclose(grad.cpu().numpy(), ref_grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    for shape in ((512, 12), (7,), (300000,)):
        x = torch.randn(*shape, device=dev)
        loss, grad = ops.sumsq_fwd_bwd(x, 0.3, 0.6)
        np.testing.assert_allclose(loss.item(), 0.3 * x.double().square().sum().item(), rtol=1e-5)
        np.testing.assert_allclose(grad.cpu().numpy(), (0.6 * x).cpu().numpy(), rtol=1e-6)
