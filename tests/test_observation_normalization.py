"""ObservationNormalization / RunningMeanStd against a rollout recorded from the reference hook
(golden ``obs_norm.npz``): normalised observations, running mean / var / count after every step — 1e-5 relative."""

from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cusrl_amd as cusrl


def replay(golden, device):
    g = golden("obs_norm")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        N, C, S, steps, max_count = (int(v) for v in g[p + "params"])
        with_state = S > 0
        spec = cusrl.EnvironmentSpec(C, 2, state_dim=S if with_state else None, num_instances=N, device=device)
        agent = SimpleNamespace(environment_spec=spec, observation_dim=C, state_dim=S if with_state else C,
                                has_state=with_state, inference_mode=False, setup_module=lambda m: m.to(device),
                                to_tensor=lambda v: torch.as_tensor(v, device=device))
        hook = cusrl.hook.ObservationNormalization(max_count=None if max_count < 0 else max_count)
        hook.pre_init(agent)
        hook.init()
        dev = lambda name: torch.from_numpy(g[p + name].copy()).to(device)  # noqa: E731
        for t in range(steps):
            tr = {"observation": dev(f"obs_in_{t}")}
            if with_state:
                tr["state"] = dev(f"state_in_{t}")
            with torch.no_grad():
                hook.pre_act(tr)
            np.testing.assert_allclose(tr["observation"].cpu().numpy(), g[p + f"obs_out_{t}"], rtol=1e-5, atol=2e-5)
            assert torch.equal(tr["original_observation"], dev(f"obs_in_{t}"))
            tr.update(next_observation=dev(f"next_in_{t}"), done=dev(f"done_{t}"))
            if with_state:
                np.testing.assert_allclose(tr["state"].cpu().numpy(), g[p + f"state_out_{t}"], rtol=1e-5, atol=2e-5)
                tr["next_state"] = dev(f"next_state_in_{t}")
            with torch.no_grad():
                hook.post_step(tr)
            np.testing.assert_allclose(tr["next_observation"].cpu().numpy(), g[p + f"next_out_{t}"], rtol=1e-5, atol=2e-5)
            np.testing.assert_allclose(hook.observation_rms.mean.cpu().numpy(), g[p + f"mean_{t}"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(hook.observation_rms.var.cpu().numpy(), g[p + f"var_{t}"], rtol=1e-5, atol=1e-6)
            assert hook.observation_rms.count == int(g[p + f"count_{t}"])
            if with_state:
                np.testing.assert_allclose(tr["next_state"].cpu().numpy(), g[p + f"next_state_out_{t}"], rtol=1e-5, atol=2e-5)
                np.testing.assert_allclose(hook.state_rms.mean.cpu().numpy(), g[p + f"state_mean_{t}"], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(hook.state_rms.var.cpu().numpy(), g[p + f"state_var_{t}"], rtol=1e-5, atol=1e-6)


def test_host_side_module_matches_reference_rollout(golden):
    replay(golden, "cpu")


@pytest.mark.gpu
def test_hip_path_matches_reference_rollout(golden):
    assert torch.cuda.is_available()
    replay(golden, "cuda:0")


def test_running_mean_std_validation_and_state():
    with pytest.raises(ValueError, match="clamp"):
        cusrl.nn.RunningMeanStd(3, clamp=0.0)
    with pytest.raises(ValueError, match="max_count"):
        cusrl.nn.RunningMeanStd(3, max_count=0)
    with pytest.raises(ValueError, match="must not overlap"):
        cusrl.nn.RunningMeanStd(4, groups=[[0, 1], [1, 2]])
    with pytest.raises(ValueError, match="must not overlap with 'groups'"):
        cusrl.nn.RunningMeanStd(4, groups=[[0, 1]], excluded_indices=[1])
    rms = cusrl.nn.RunningMeanStd(4, groups=[[0, 1]], excluded_indices=[3])
    x = torch.tensor([[1.0, 3.0, 5.0, 100.0], [3.0, 5.0, 7.0, -100.0]])
    rms.update(x)
    assert rms.count == 2
    assert torch.allclose(rms.mean, torch.tensor([3.0, 3.0, 6.0, 0.0]))  # grouped channels share, excluded stays (0, 1)
    assert torch.allclose(rms.var[3], torch.tensor(1.0)) and torch.allclose(rms.var[0], rms.var[1])
    state = rms.state_dict()
    other = cusrl.nn.RunningMeanStd(4)
    other.load_state_dict(state)
    assert other.count == 2 and torch.equal(other.mean, rms.mean)
    assert torch.allclose(rms.unnormalize(rms.normalize(x))[:, :3], x[:, :3], atol=1e-4)  # channel 3 is clamped at +-10


@pytest.mark.gpu
def test_ppo_preset_with_observation_normalization_runs_eager_and_graphed():
    cusrl.config.set_device("cuda:0")
    finals = []
    for compile_ in (False, True):
        cusrl.set_global_seed(21)
        env = cusrl.testing.DummyTorchEnvironment(num_instances=32, observation_dim=10, action_dim=3, device="cuda:0")
        factory = cusrl.preset.PpoAgentFactory(num_steps_per_update=6, sampler_epochs=2, sampler_mini_batches=2,
                                               normalize_observation=True, compile=compile_,
                                               optimizer_kwargs={"capturable": True, "fused": True})
        trainer = cusrl.Trainer(env, factory, num_iterations=4, verbose=False)
        trainer.run_training_loop()
        agent = trainer.agent
        assert "original_observation" in agent.buffer and "original_next_observation" in agent.buffer
        rms = agent.hook["observation_normalization"].observation_rms
        finals.append((rms.mean.clone(), rms.var.clone(), rms.count, torch.cat([p.detach().reshape(-1) for p in agent.parameters()])))
        assert np.isfinite(trainer.last_info["Agent/value_loss"])
    assert finals[0][2] == finals[1][2] > 0
    assert torch.allclose(finals[0][0], finals[1][0], rtol=1e-4, atol=1e-5)
    assert torch.allclose(finals[0][1], finals[1][1], rtol=1e-4, atol=1e-5)
    assert torch.allclose(finals[0][3], finals[1][3], rtol=1e-3, atol=1e-4)
