#!/usr/bin/env python3
"""Run a few iterations of one BASELINE.json config on cuda:0 at its per-GPU workload and print timing lines (not the
headline bench — that is bench.py on config 2).

    python scripts/run_config.py config1            # MountainCar `ppo` kwargs, 8 envs, discrete / Tanh / obs-norm
    python scripts/run_config.py config2 --compile  # 4096 envs, MLP (the bench workload)
    python scripts/run_config.py config3 --compile  # 8192 envs per GPU (65 536 over 8 ranks)
    python scripts/run_config.py config4            # GRU recurrent PPO, 16 384 envs, BPTT minibatches
    python scripts/run_config.py config5            # RND + AMP hooks, 4096 envs per GPU
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402
from cusrl_amd import _native  # noqa: E402

MOUNTAIN_CAR = dict(num_steps_per_update=16, actor_hidden_dims=(64, 64), critic_hidden_dims=(64, 64), activation_fn="Tanh",
                    action_space_type="discrete", lr=3e-4, sampler_epochs=4, sampler_mini_batches=4, orthogonal_init=False,
                    normalize_observation=True, gae_gamma=0.99, gae_lamda=0.98, entropy_loss_weight=0.0, max_grad_norm=0.5)


def build(name: str, envs: int | None, compile_: bool):
    extra = {"capturable": True, "fused": True}
    if name == "config1":
        env = cusrl.testing.DummyTorchEnvironment(envs or 8, 2, 3, device="cuda:0")
        return env, cusrl.preset.PpoAgentFactory(**MOUNTAIN_CAR, compile=compile_)
    if name in ("config2", "config3"):
        env = cusrl.testing.SyntheticEnvironment(envs or (4096 if name == "config2" else 8192), 48, 12, device="cuda:0")
        return env, cusrl.preset.PpoAgentFactory(compile=compile_, optimizer_kwargs=extra)
    if name == "config4":
        env = cusrl.testing.SyntheticEnvironment(envs or 16384, 48, 12, device="cuda:0")
        return env, cusrl.preset.RecurrentPpoAgentFactory(rnn_type="GRU", optimizer_kwargs=extra)
    if name == "config5":
        env = cusrl.testing.SyntheticEnvironment(envs or 4096, 48, 12, device="cuda:0")
        k = 6
        factory = cusrl.preset.AmpAgentFactory(amp_dataset_source=torch.randn(100_000, 2 * k, device="cuda:0"),
                                               amp_state_indices=slice(k), compile=compile_).to_underlying()
        factory.register_hook(cusrl.hook.RandomNetworkDistillation(module_factory=cusrl.Mlp.Factory(hidden_dims=[128, 64]),
                                                                   output_dim=16, reward_scale=0.1), before="value_computation")
        return env, factory
    raise ValueError(name)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("config", choices=["config1", "config2", "config3", "config4", "config5"])
    parser.add_argument("--envs", type=int, default=None)
    parser.add_argument("--iterations", type=int, default=6)
    parser.add_argument("--compile", action="store_true")
    parser.add_argument("--phases", action="store_true",
                        help="also print agent.update()'s share (two device synchronisations per iteration: the host can no "
                             "longer run ahead of the device, small configs read ~1 ms slower)")
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env, factory = build(args.config, args.envs, args.compile)
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    horizon = trainer.agent.num_steps_per_update
    observation, state, _ = env.reset()
    print(f"== {args.config}: {env.num_instances} envs x {horizon} steps, compile={args.compile}", flush=True)
    # where an iteration goes: agent.update() bracketed by a synchronised host clock (the rollout is the rest)
    update_time = [0.0]
    original_update = trainer.agent.update

    def timed_update():
        torch.cuda.synchronize()
        start = time.perf_counter()
        result = original_update()
        torch.cuda.synchronize()
        update_time[0] = time.perf_counter() - start
        return result

    if args.phases:
        trainer.agent.update = timed_update
    for i in range(args.iterations):
        before = dict(_native.launch_counts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        info = trainer.last_info
        launches = sum(v - before.get(k, 0) for k, v in _native.launch_counts.items())
        share = f" (update {update_time[0] * 1e3:7.2f})" if args.phases else ""
        print(f"iteration {i}: {dt * 1e3:8.2f} ms{share}  {env.num_instances * horizon / dt / 1e6:6.3f} M env-steps/s  "
              f"value_loss={info['Agent/value_loss']:.4f} kl={info['Agent/kl_divergence']:.2e} "
              f"hip_entry_calls={launches} mem={torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
    census = {k: v for k, v in sorted(_native.launch_counts.items())}
    print("C-ABI calls (whole run, Python-side; graph replays excluded):", census, flush=True)


if __name__ == "__main__":
    main()
