"""Reference-equivalent rollout + PPO update as plain PyTorch ops — TEST INFRASTRUCTURE / CPU BASELINE ONLY.

This is the torch fp32 restatement of the WHOLE per-iteration path of the reference's `ppo` preset, written as one
procedural class: the same op sequence the reference executes (per-leaf ``storage[cursor] = value`` appends,
the T-1 step Python GAE loop, ``var_mean`` normalisation, ``randperm`` + per-leaf ``flatten(0,1)[idx]`` gathers,
per-hook loss formulas under autograd, flat-norm clipping, Adam, the post-update statistics pass).  It exists to
(1) time the reference's CPU path on the GPU box's host cores (``bench.py`` ``cpu_baseline``, kind "port") and
(2) cross-check the HIP path end to end.  It is pinned against the real reference by
``tests/test_oracle_golden.py::test_torch_port_replays_reference_update`` (golden ``update_trace.npz``).

Reference lines restated: cusrl/template/actor_critic.py:227-320 (act/step/update/_train_step),
template/buffer.py:124-162, sampler/mini_batch_sampler.py:52-89, hook/on_policy/value.py:56-82,121-137,
gae.py:8-20,85-110, advantage.py:108-115, common.py:29-43, ppo.py:10-18,50-55,82-84,
gradient_clipping.py:58-76, stats.py:29-40, preset/ppo.py:37-65 (hook order).
Never imported by cusrl_amd.
"""

from __future__ import annotations

import math

import torch

from cusrl_amd.nn import Actor, Mlp, NormalDist, Value  # plain torch.nn modules (not part of the HIP hot path)


class TorchPpo:
    def __init__(self, observation_dim, action_dim, num_envs, *, num_steps_per_update=24, hidden=(256, 128), lr=2e-4,
                 epochs=5, mini_batches=4, gamma=0.99, lamda=0.95, lamda_value=None, clip=0.2, value_clip=None,
                 w_sur=1.0, w_val=0.5, w_ent=0.01, max_grad_norm=1.0, device="cpu", orthogonal_init=True):
        self.device = torch.device(device)
        self.T, self.N = num_steps_per_update, num_envs
        self.epochs, self.mini_batches = epochs, mini_batches
        self.gamma, self.lamda, self.lamda_value = gamma, lamda, lamda_value
        self.clip, self.value_clip = clip, value_clip
        self.w_sur, self.w_val, self.w_ent, self.max_grad_norm = w_sur, w_val, w_ent, max_grad_norm
        backbone = lambda: Mlp.Factory(hidden_dims=hidden, activation_fn="ReLU", ends_with_activation=True)  # noqa: E731
        self.actor = Actor.Factory(backbone(), NormalDist.Factory())(observation_dim, action_dim)
        self.critic = Value.Factory(backbone())(observation_dim, 1)
        if orthogonal_init:  # hook/control/initialization.py:75-83
            for net in (self.actor, self.critic):
                for m in net.modules():
                    if isinstance(m, torch.nn.Linear):
                        torch.nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
                        torch.nn.init.zeros_(m.bias)
            torch.nn.init.orthogonal_(self.actor.distribution.mean_head.weight, gain=math.sqrt(2) * 0.1)
            torch.nn.init.zeros_(self.actor.distribution.mean_head.bias)
        self.actor.to(self.device), self.critic.to(self.device)
        self.params = [p for net in (self.actor, self.critic) for p in net.parameters()]
        self.optimizer = torch.optim.Adam(self.params, lr=lr)
        self.storage: dict[str, torch.Tensor] = {}
        self.cursor = 0
        self.transition: dict[str, torch.Tensor] = {}
        self.trace: dict[str, list] | None = None

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def act(self, observation):
        observation = observation.clone()
        dist, (action, logp), _ = self.actor.explore(observation)
        value = self.critic.evaluate(observation)
        self.transition = {"observation": observation, "action_dist.mean": dist["mean"], "action_dist.std": dist["std"],
                           "action": action, "action_logp": logp, "value": value}
        return action

    @torch.no_grad()
    def step(self, next_observation, reward, terminated, truncated, **extra):
        tr = self.transition
        tr.update(next_observation=next_observation, reward=reward, terminated=terminated, truncated=truncated, **extra)
        tr["done"] = terminated | truncated
        for key, value in tr.items():  # buffer.py:134-146: one indexed copy per leaf
            if key not in self.storage:
                self.storage[key] = value.new_zeros(self.T, *value.shape)
            self.storage[key][self.cursor] = value
        self.cursor += 1
        if self.cursor == self.T:
            self.cursor = 0
            return True
        return False

    # ------------------------------------------------------------------ update
    @staticmethod
    def _gae(reward, done, value, next_value, gamma, lamda):
        not_done = done.logical_not()
        advantage = reward + next_value * gamma - value
        for t in range(advantage.size(0) - 2, -1, -1):
            advantage[t] += not_done[t] * (gamma * lamda) * advantage[t + 1]
        return advantage

    @torch.no_grad()
    def pre_update(self):
        s = self.storage
        value = s["value"]
        next_value = s.setdefault("next_value", torch.zeros_like(value))
        terminated, truncated = s["terminated"].squeeze(-1), s["truncated"].squeeze(-1)
        next_value[:-1] = value[1:]
        next_value[-1] = self.critic.evaluate(s["next_observation"][-1])
        next_value[terminated] = value.new_zeros(value.size(-1))
        if truncated.any():
            next_value[truncated] = self.critic.evaluate(s["next_observation"][truncated])
        s["advantage"] = self._gae(s["reward"], s["done"], value, next_value, self.gamma, self.lamda)
        s["return"] = value + (s["advantage"] if self.lamda_value is None else
                               self._gae(s["reward"], s["done"], value, next_value, self.gamma, self.lamda_value))
        var, mean = torch.var_mean(s["advantage"], dim=(0, 1))
        s["advantage"].sub_(mean).div_((var + 1e-8).sqrt())

    def _losses(self, batch):
        dist, _ = self.actor(batch["observation"])
        curr_value = self.critic.evaluate(batch["observation"])
        if self.value_clip is None:
            value_loss = torch.nn.functional.mse_loss(batch["return"], curr_value)
        else:
            clipped = batch["value"] + (curr_value - batch["value"]).clamp(-self.value_clip, self.value_clip)
            value_loss = torch.max((curr_value - batch["return"]).square(), (clipped - batch["return"]).square()).mean()
        logp = self.actor.compute_logp(dist, batch["action"])
        entropy = self.actor.compute_entropy(dist)
        ratio = (logp - batch["action_logp"]).exp()
        adv = batch["advantage"]
        surrogate = -torch.min(adv * ratio, adv * ratio.clamp(1.0 - self.clip, 1.0 + self.clip)).mean()
        return {"value_loss": value_loss * self.w_val, "surrogate_loss": surrogate * self.w_sur,
                "entropy_loss": -entropy.mean() * self.w_ent}

    def update(self):
        self.pre_update()
        S = self.T * self.N
        flat = {k: v.flatten(0, 1) for k, v in self.storage.items()}
        epoch_indices = torch.randperm(S, device=self.device)
        size = S // self.mini_batches
        last = {}
        for epoch in range(self.epochs):
            if epoch > 0:
                torch.randperm(S, device=self.device, out=epoch_indices)
            for j in range(self.mini_batches):
                idx = epoch_indices[j * size:(j + 1) * size]
                batch = {k: v[idx] for k, v in flat.items()}  # every stored leaf, like Buffer.sample
                objectives = self._losses(batch)
                loss = sum(objectives.values())
                self.optimizer.zero_grad()
                loss.backward()
                if self.trace is not None:
                    self.trace["indices"].append(idx.clone())
                    self.trace["objectives"].append(torch.stack([v.detach() for v in objectives.values()]))
                    self.trace["grads_unclipped"].append(torch.cat([p.grad.reshape(-1) for p in self.params]))
                if self.max_grad_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
                if self.trace is not None:
                    self.trace["grads"].append(torch.cat([p.grad.reshape(-1) for p in self.params]))
                self.optimizer.step()
                if self.trace is not None:
                    self.trace["params_after"].append(torch.cat([p.detach().reshape(-1) for p in self.params]))
                last = {k: v.detach() for k, v in objectives.items()}
        # OnPolicyStatistics.post_update with AutoMiniBatchSampler(): one more permutation + full gather + actor pass
        with torch.no_grad():
            idx = torch.randperm(S, device=self.device)
            batch = {k: v[idx] for k, v in flat.items()}
            dist, _ = self.actor(batch["observation"])
            old = {"mean": batch["action_dist.mean"], "std": batch["action_dist.std"]}
            last["kl_divergence"] = self.actor.compute_kl_div(old, dist).mean()
            ratio = (self.actor.compute_logp(dist, batch["action"]) - batch["action_logp"]).exp()
            last["importance_weighted_advantage"] = (batch["advantage"] * ratio).mean()
            last["action_std"] = dist["std"].mean()
        return last


def run_iterations(agent: TorchPpo, env, iterations: int, observation=None):
    """Rollout + update loop with the reference trainer's per-step host work (trainer.py:296-321)."""
    if observation is None:
        observation, _, _ = env.reset()
    for _ in range(iterations):
        while True:
            action = agent.act(observation)
            next_observation, _, reward, terminated, truncated, _ = env.step(action)
            ready = agent.step(next_observation, reward, terminated, truncated)
            done = (terminated | truncated).squeeze(-1).nonzero().reshape(-1).tolist()
            if done:
                init_observation, _, _ = env.reset(indices=done)
                next_observation[done] = init_observation
            observation = next_observation
            if ready:
                break
        agent.update()
    return observation
