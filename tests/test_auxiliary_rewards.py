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
