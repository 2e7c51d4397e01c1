#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference (chengruiz/cusrl) on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU
box); the resulting ``*.npz`` files are data — inputs and expected outputs —
and are committed under ``tests/golden/``.  Nothing of the reference (source or
bytecode) is copied: the reference package is imported from where it lies, with
four non-arithmetic third-party modules stubbed: ``gymnasium``,
``git``, ``tyro`` and ``objprint``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every fixture records ``torch_version`` because the permutation stream and the
reduction order are properties of the installed torch build.
"""

from __future__ import annotations

import os
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REFERENCE = Path(os.environ.get("CUSRL_REFERENCE", "/root/reference"))


def import_reference():
    for name in (
        "gymnasium",
        "gymnasium.envs",
        "gymnasium.envs.registration",
        "gymnasium.spaces",
        "gymnasium.vector",
        "gymnasium.wrappers",
        "git",
        "tyro",
        "tyro.constructors",
        "tyro.conf",
        "tyro.extras",
    ):
        sys.modules.setdefault(name, MagicMock())
    objprint = types.ModuleType("objprint")
    objprint.add_objprint = lambda *a, **k: (lambda cls: cls)
    objprint.objstr = repr
    objprint.op = print
    sys.modules.setdefault("objprint", objprint)
    sys.path.insert(0, str(REFERENCE))
    import cusrl  # noqa: PLC0415

    return cusrl


def np_(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().copy()


META = {"torch_version": np.array(torch.__version__)}


# --------------------------------------------------------------------------- GAE
def make_gae(cusrl):
    """Rows a4/a5: GAE scan, return, advantage normalisation (gae.py:8-110, advantage.py:108-115)."""
    from cusrl.hook.on_policy.gae import _generalized_advantage_estimation  # noqa: PLC0415

    out = dict(META)
    cases = []
    gen = torch.Generator().manual_seed(1234)
    idx = 0
    for T, N, D in [(24, 64, 1), (24, 64, 2), (7, 33, 3), (1, 5, 1), (2, 1, 1)]:
        for gamma, lamda, lamda_value in [(0.99, 0.95, None), (0.99, 0.95, 0.995), (0.5, 1.0, 0.0)]:
            reward = torch.randn(T, N, D, generator=gen)
            value = torch.randn(T, N, D, generator=gen)
            next_value = torch.randn(T, N, D, generator=gen)
            done = torch.rand(T, N, 1, generator=gen) < 0.05
            hook = cusrl.hook.GeneralizedAdvantageEstimation(gamma=gamma, lamda=lamda, lamda_value=lamda_value)
            data = {"reward": reward, "done": done, "value": value, "next_value": next_value}
            hook.pre_update(data)
            advantage = data["advantage"].clone()
            ret = data["return"].clone()
            norm = cusrl.hook.AdvantageNormalization(synchronize=False)
            normalized = advantage.clone()
            var, mean = torch.var_mean(advantage, dim=(0, 1))
            if T * N > 1:
                norm.normalize_(normalized)
            # sanity: functional form agrees with the hook
            assert torch.equal(
                advantage, _generalized_advantage_estimation(reward, done, value, next_value, gamma, lamda)
            )
            p = f"c{idx}_"
            out[p + "reward"] = np_(reward)
            out[p + "value"] = np_(value)
            out[p + "next_value"] = np_(next_value)
            out[p + "done"] = np_(done)
            out[p + "advantage"] = np_(advantage)
            out[p + "return"] = np_(ret)
            out[p + "mean"] = np_(mean)
            out[p + "var"] = np_(var)
            out[p + "normalized"] = np_(normalized)
            out[p + "params"] = np.array([gamma, lamda, -1.0 if lamda_value is None else lamda_value], dtype=np.float64)
            cases.append(idx)
            idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "gae.npz", **out)
    print("gae.npz:", idx, "cases")


# ------------------------------------------------------------------- next_value
def make_next_value(cusrl):
    """Row a3: ValueComputation.pre_update (value.py:56-82) with a closed-form critic."""
    from types import SimpleNamespace  # noqa: PLC0415
    from contextlib import nullcontext  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(77)
    idx = 0

    class Critic:
        # value(state) = 0.25 * sum(state) + 0.5 * state[0], per value channel d scaled by (d+1)
        def __init__(self, D):
            self.D = D

        def evaluate(self, state, memory=None):
            base = 0.25 * state.sum(-1, keepdim=True) + 0.5 * state[..., :1]
            return torch.cat([base * (d + 1) for d in range(self.D)], dim=-1)

    for T, N, D, O in [(24, 64, 1, 5), (8, 16, 2, 3), (1, 4, 1, 2), (5, 7, 1, 4)]:
        for term_p, trunc_p, bootstrap, term_value in [
            (0.05, 0.05, True, 0.0),
            (0.1, 0.0, True, 0.0),
            (0.05, 0.1, False, 0.0),
            (0.2, 0.2, True, -1.5),
        ]:
            value = torch.randn(T, N, D, generator=gen)
            next_obs = torch.randn(T, N, O, generator=gen)
            terminated = torch.rand(T, N, 1, generator=gen) < term_p
            truncated = torch.rand(T, N, 1, generator=gen) < trunc_p
            buffer = cusrl.Buffer(capacity=T, parallelism=N, device="cpu")
            buffer["value"] = value.clone()
            buffer["next_observation"] = next_obs.clone()
            buffer["terminated"] = terminated.clone()
            buffer["truncated"] = truncated.clone()
            hook = cusrl.hook.ValueComputation(termination_value=term_value, bootstrap_truncated_states=bootstrap)
            hook.agent = SimpleNamespace(critic=Critic(D), autocast=nullcontext)
            hook._critic_memory = None
            hook.pre_update(buffer)
            p = f"c{idx}_"
            out[p + "value"] = np_(value)
            out[p + "next_observation"] = np_(next_obs)
            out[p + "terminated"] = np_(terminated)
            out[p + "truncated"] = np_(truncated)
            out[p + "next_value"] = np_(buffer["next_value"])
            out[p + "params"] = np.array([term_value, float(bootstrap)], dtype=np.float64)
            idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "next_value.npz", **out)
    print("next_value.npz:", idx, "cases")


# --------------------------------------------------------------------- randperm
def make_randperm(cusrl):
    """Row a7: the permutation stream of MiniBatchSampler (mini_batch_sampler.py:56,68).

    Recorded through the reference sampler itself: a buffer holding its own flat
    sample index is sampled for 3 epochs, so the gathered values ARE the indices.
    """
    out = dict(META)
    idx = 0
    for seed, T, N, mbs in [(0, 4, 4, 2), (0, 24, 4096, 4), (42, 1, 4096, 1), (7, 3, 5, 4)]:
        buffer = cusrl.Buffer(capacity=T, parallelism=N, device="cpu")
        flat = torch.arange(T * N, dtype=torch.int64).reshape(T, N, 1)
        for t in range(T):
            buffer.push({"flat_index": flat[t]})
        torch.manual_seed(seed)
        sampler = cusrl.MiniBatchSampler(num_epochs=3, num_mini_batches=mbs)
        epochs = [[] for _ in range(3)]
        for metadata, batch in sampler(buffer):
            epochs[metadata["epoch_index"]].append(batch["flat_index"].squeeze(-1))
        p = f"c{idx}_"
        out[p + "params"] = np.array([seed, T, N, mbs], dtype=np.int64)
        out[p + "indices"] = np.stack([np_(torch.cat(e)) for e in epochs])  # [3, mbs * (S // mbs)]
        # the raw stream, for the C restatement of torch's CPU randperm
        torch.manual_seed(seed)
        first = torch.randperm(T * N)
        second = torch.randperm(T * N, out=first.clone())
        out[p + "raw0"] = np_(first)
        out[p + "raw1"] = np_(second)
        idx += 1
    # temporal sampler: permutes env ids (mini_batch_sampler.py:110-114)
    T, N = 3, 10
    buffer = cusrl.Buffer(capacity=T, parallelism=N, device="cpu")
    for t in range(T):
        env = torch.arange(N, dtype=torch.float32).reshape(N, 1)
        buffer.push({"observation": env * 100 + t, "actor_memory": env * 100 + t + 0.5})
    torch.manual_seed(3)
    got = []
    for metadata, batch in cusrl.AutoMiniBatchSampler(num_epochs=2, num_mini_batches=3)(buffer):
        assert metadata["temporal"] is True
        got.append(np_(batch["observation"]))
    out["temporal_obs"] = np.stack(got)  # [6, T, 3, 1]
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "randperm.npz", **out)
    print("randperm.npz:", idx, "cases")


# ----------------------------------------------------------------------- losses
def make_losses(cusrl):
    """Rows a9-a12: surrogate / value / entropy losses and Normal log-prob with their gradients."""
    from cusrl.hook.on_policy.ppo import _ppo_surrogate_loss  # noqa: PLC0415
    from cusrl.hook.on_policy.value import _clipped_value_loss  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(2024)
    idx = 0
    dist = cusrl.NormalDist(4, 3)  # only its stateless compute_* methods are used
    for B, A, D, clip, vclip, w in [
        (257, 12, 1, 0.2, None, (1.0, 0.5, 0.01)),
        (64, 3, 1, 0.1, 0.2, (2.0, 1.0, 0.0)),
        (1000, 12, 2, 0.3, 0.5, (1.0, 0.25, 0.05)),
        (1, 1, 1, 0.2, None, (1.0, 0.5, 0.01)),
    ]:
        w_sur, w_val, w_ent = w
        mean = torch.randn(B, A, generator=gen).requires_grad_()
        std = (torch.rand(B, A, generator=gen) * 0.9 + 0.1).requires_grad_()
        old_mean = mean.detach() + 0.1 * torch.randn(B, A, generator=gen)
        old_std = std.detach() * (1 + 0.05 * torch.randn(B, A, generator=gen)).abs()
        action = old_mean + old_std * torch.randn(B, A, generator=gen)
        old_logp = dist.compute_logp({"mean": old_mean, "std": old_std}, action)
        advantage = torch.randn(B, 1, generator=gen)
        ret = torch.randn(B, D, generator=gen)
        old_value = ret + 0.3 * torch.randn(B, D, generator=gen)
        curr_value = (old_value + 0.3 * torch.randn(B, D, generator=gen)).requires_grad_()

        params = {"mean": mean, "std": std}
        logp = dist.compute_logp(params, action)
        entropy = dist.compute_entropy(params)
        kl = dist.compute_kl_div({"mean": old_mean, "std": old_std}, params)
        logp_ratio = logp - old_logp
        ratio = logp_ratio.exp()
        surrogate = _ppo_surrogate_loss(advantage, ratio, clip) * w_sur
        if vclip is None:
            value_loss = torch.nn.functional.mse_loss(ret, curr_value) * w_val
        else:
            value_loss = _clipped_value_loss(old_value, curr_value, ret, vclip) * w_val
        entropy_loss = -entropy.mean() * w_ent
        # the reference sums objectives as a Python left fold in hook order (actor_critic.py:309)
        loss = sum({"value_loss": value_loss, "surrogate_loss": surrogate, "entropy_loss": entropy_loss}.values())
        loss.backward()
        p = f"c{idx}_"
        for k, v in dict(
            mean=mean, std=std, old_mean=old_mean, old_std=old_std, action=action, old_logp=old_logp,
            advantage=advantage, ret=ret, old_value=old_value, curr_value=curr_value, logp=logp,
            entropy=entropy, kl=kl, logp_ratio=logp_ratio, ratio=ratio, surrogate=surrogate,
            value_loss=value_loss, entropy_loss=entropy_loss, loss=loss, d_mean=mean.grad,
            d_std=std.grad, d_value=curr_value.grad,
        ).items():
            out[p + k] = np_(v)
        out[p + "params"] = np.array([clip, -1.0 if vclip is None else vclip, w_sur, w_val, w_ent], dtype=np.float64)
        idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "losses.npz", **out)
    print("losses.npz:", idx, "cases")


def make_categorical_losses(cusrl):
    """Rows a9-a12 for discrete action spaces (BASELINE config 1): ``OneHotCategoricalDist`` log-prob / entropy / KL
    (nn/module/distribution.py:332-366) under the same surrogate / value / entropy losses, with gradients wrt logits."""
    from cusrl.hook.on_policy.ppo import _ppo_surrogate_loss  # noqa: PLC0415
    from cusrl.hook.on_policy.value import _clipped_value_loss  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(2025)
    idx = 0
    dist = cusrl.OneHotCategoricalDist(4, 3)  # only its stateless compute_* methods are used
    for B, A, D, clip, vclip, w in [
        (64, 3, 1, 0.2, None, (1.0, 0.5, 0.0)),       # MountainCar-v0: 3 actions, 4 x 4 minibatches of 8 envs x 16 steps... x2
        (257, 6, 1, 0.2, None, (1.0, 0.5, 0.01)),
        (1000, 18, 2, 0.3, 0.5, (2.0, 0.25, 0.05)),
        (1, 2, 1, 0.2, None, (1.0, 0.5, 0.01)),
        (300, 40, 1, 0.1, 0.2, (1.0, 1.0, 0.02)),
    ]:
        w_sur, w_val, w_ent = w
        logits = (2.0 * torch.randn(B, A, generator=gen)).requires_grad_()
        old_logits = logits.detach() + 0.2 * torch.randn(B, A, generator=gen)
        taken = torch.multinomial(torch.softmax(old_logits, -1), 1, generator=gen).squeeze(-1)
        action = torch.nn.functional.one_hot(taken, A).float()
        old_logp = dist.compute_logp({"logits": old_logits}, action)
        advantage = torch.randn(B, 1, generator=gen)
        ret = torch.randn(B, D, generator=gen)
        old_value = ret + 0.3 * torch.randn(B, D, generator=gen)
        curr_value = (old_value + 0.3 * torch.randn(B, D, generator=gen)).requires_grad_()
        params = {"logits": logits}
        logp = dist.compute_logp(params, action)
        entropy = dist.compute_entropy(params)
        kl = dist.compute_kl_div({"logits": old_logits}, params)
        logp_ratio = logp - old_logp
        ratio = logp_ratio.exp()
        surrogate = _ppo_surrogate_loss(advantage, ratio, clip) * w_sur
        if vclip is None:
            value_loss = torch.nn.functional.mse_loss(ret, curr_value) * w_val
        else:
            value_loss = _clipped_value_loss(old_value, curr_value, ret, vclip) * w_val
        entropy_loss = -entropy.mean() * w_ent
        loss = sum({"value_loss": value_loss, "surrogate_loss": surrogate, "entropy_loss": entropy_loss}.values())
        loss.backward()
        p = f"c{idx}_"
        for k, v in dict(
            logits=logits, old_logits=old_logits, action=action, old_logp=old_logp, advantage=advantage, ret=ret,
            old_value=old_value, curr_value=curr_value, logp=logp, entropy=entropy, kl=kl, logp_ratio=logp_ratio, ratio=ratio,
            surrogate=surrogate, value_loss=value_loss, entropy_loss=entropy_loss, loss=loss, d_logits=logits.grad,
            d_value=curr_value.grad,
        ).items():
            out[p + k] = np_(v)
        out[p + "params"] = np.array([clip, -1.0 if vclip is None else vclip, w_sur, w_val, w_ent], dtype=np.float64)
        idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "categorical_losses.npz", **out)
    print("categorical_losses.npz:", idx, "cases")


# -------------------------------------------------------------- merge mean/var
def make_merge(cusrl):
    """Row a6: distributed.reduce_mean_var_ (distributed.py:175-183) with gather_stack replaced by given stacks."""
    from cusrl.utils import distributed  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(5)
    idx = 0
    orig_conf, orig_gather = distributed.configure_distributed, distributed.gather_stack
    try:
        for W, D in [(2, 1), (8, 1), (8, 3), (4, 2)]:
            means = torch.randn(W, D, generator=gen)
            vars_ = torch.rand(W, D, generator=gen) + 0.1
            stack = torch.cat((means, vars_), dim=-1)
            distributed.configure_distributed = lambda *a, **k: True
            distributed.gather_stack = lambda tensor, _s=stack: _s
            mean, var = means[0].clone(), vars_[0].clone()
            distributed.reduce_mean_var_(mean, var)
            p = f"c{idx}_"
            out[p + "means"] = np_(means)
            out[p + "vars"] = np_(vars_)
            out[p + "mean"] = np_(mean)
            out[p + "var"] = np_(var)
            idx += 1
    finally:
        distributed.configure_distributed, distributed.gather_stack = orig_conf, orig_gather
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "merge_mean_var.npz", **out)
    print("merge_mean_var.npz:", idx, "cases")


# ----------------------------------------------------------------- update trace
def make_update_trace(cusrl):
    """Golden (4): one rollout + one full ``agent.update()`` of the `ppo` preset on an 8-env x 16-obs x 8-act
    dummy task — buffer-in, per-minibatch indices, losses, flat gradient, post-step parameters."""
    from cusrl.testing.environment import DummyTorchEnvironment  # noqa: PLC0415
    from cusrl.template.environment import get_done_indices  # noqa: PLC0415

    cusrl.config.set_device("cpu")
    out = dict(META)
    for tag, factory_kwargs, mini_batch_wise_kl in [
        ("a", dict(num_steps_per_update=6, sampler_epochs=2, sampler_mini_batches=3), None),
        ("b", dict(num_steps_per_update=5, sampler_epochs=2, sampler_mini_batches=2, gae_lamda_value=0.98,
                   value_loss_clip=0.2), None),
        # MiniBatchWiseLRSchedule right behind the preparation hook: the learning rate moves between minibatch steps
        ("c", dict(num_steps_per_update=6, sampler_epochs=4, sampler_mini_batches=2), 3e-5),
    ]:
        torch.manual_seed(11)
        env = DummyTorchEnvironment(num_instances=8, observation_dim=16, action_dim=8, reward_dim=1)
        factory = cusrl.preset.PpoAgentFactory(actor_hidden_dims=(32, 16), critic_hidden_dims=(32, 16), **factory_kwargs)
        underlying = factory.to_underlying()

        trace = {"objectives": [], "indices": [], "grads_unclipped": [], "grads": [], "params_after": [], "lrs": [], "kls": []}

        class Capture(cusrl.Hook):
            def __init__(self, where):
                super().__init__()
                self.where = where
                self.name_(f"capture_{where}")

            def pre_optim(self, optimizer):
                flat = torch.cat([p.grad.reshape(-1) for g in optimizer.param_groups for p in g["params"]])
                trace["grads_unclipped" if self.where == "pre" else "grads"].append(np_(flat))
                if self.where == "pre":
                    trace["lrs"].append([group["lr"] for group in optimizer.param_groups])

            def post_optim(self):
                if self.where == "post":
                    flat = torch.cat([p.detach().reshape(-1) for _, p in self.agent.named_parameters()])
                    trace["params_after"].append(np_(flat))

            def objective(self, metadata, batch):
                if self.where == "post":
                    trace["indices"].append(np_(batch["flat_index"].squeeze(-1)))
                    if "kl_divergence" in batch:
                        trace["kls"].append(batch["kl_divergence"].mean().item())

        if mini_batch_wise_kl is not None:
            underlying.register_hook(cusrl.hook.MiniBatchWiseLRSchedule(mini_batch_wise_kl), after="on_policy_preparation")
        underlying.register_hook(Capture("pre"), before="gradient_clipping")
        underlying.register_hook(Capture("post"), after="gradient_clipping")
        agent = underlying(env.spec)

        state0 = {n: np_(p) for n, p in agent.named_parameters()}
        orig_objective = agent.hook.objective

        def wrapped(metadata, batch, _o=orig_objective):
            res = _o(metadata, batch)
            trace["objectives"].append(np.array([res["value_loss"].item(), res["surrogate_loss"].item(),
                                                 res["entropy_loss"].item()], dtype=np.float32))
            return res

        agent.hook.objective = wrapped

        observation, state, _ = env.reset()
        T = factory.num_steps_per_update
        step = 0
        while True:
            action = agent.act(observation, state)
            observation, state, reward, terminated, truncated, _ = env.step(action)
            flat_index = (torch.arange(8) + step * 8).reshape(8, 1)
            ready = agent.step(observation, reward, terminated, truncated, state, flat_index=flat_index)
            step += 1
            if ready:
                break
        buffer_in = {k: np_(v) for k, v in agent.buffer.storage.items()}
        torch.manual_seed(99)  # generator state at the update boundary
        metrics = agent.update()
        buffer_out = {k: np_(agent.buffer.storage[k]) for k in ("next_value", "advantage", "return")}

        p = tag + "_"
        out[p + "factory_keys"] = np.array(list(factory_kwargs.keys()))
        out[p + "factory_vals"] = np.array([-1.0 if v is None else float(v) for v in factory_kwargs.values()])
        for k, v in state0.items():
            out[p + "param0/" + k] = v
        out[p + "param_names"] = np.array(list(state0.keys()))
        for k, v in buffer_in.items():
            out[p + "buffer_in/" + k] = v
        out[p + "buffer_keys"] = np.array(list(buffer_in.keys()))
        for k, v in buffer_out.items():
            out[p + "buffer_out/" + k] = v
        out[p + "objectives"] = np.stack(trace["objectives"])
        out[p + "indices"] = np.stack(trace["indices"])
        out[p + "grads_unclipped"] = np.stack(trace["grads_unclipped"])
        out[p + "grads"] = np.stack(trace["grads"])
        out[p + "params_after"] = np.stack(trace["params_after"])
        out[p + "lrs"] = np.asarray(trace["lrs"], dtype=np.float64)
        if mini_batch_wise_kl is not None:
            kls = np.asarray(trace["kls"])
            out[p + "mini_batch_wise_kl"], out[p + "kls"] = np.float64(mini_batch_wise_kl), kls
            edges = np.array([mini_batch_wise_kl / 2.0, mini_batch_wise_kl * 2.0])
            margin = np.abs(kls[:, None] / edges[None] - 1.0).min()
            print(f"  minibatch KLs {kls}, lr {out[p + 'lrs'][:, 0]}, closest decision edge {margin:.1%} away")
            assert margin > 0.1, "a KL this close to a threshold edge would make the replay flip on rounding"
        out[p + "metric_keys"] = np.array(list(metrics.keys()))
        out[p + "metric_vals"] = np.array(list(metrics.values()), dtype=np.float64)
        print(f"update trace {tag}: {len(trace['objectives'])} train steps, buffer leaves {list(buffer_in)}")
    np.savez_compressed(HERE / "update_trace.npz", **out)


MOUNTAIN_CAR_KWARGS = dict(  # cusrl/zoo/gym/classic_control.py:65-80 (the `ppo` registration of MountainCar-v0)
    num_steps_per_update=16, actor_hidden_dims=(64, 64), critic_hidden_dims=(64, 64), activation_fn="Tanh",
    action_space_type="discrete", lr=3e-4, sampler_epochs=4, sampler_mini_batches=4, orthogonal_init=False,
    normalize_observation=True, gae_gamma=0.99, gae_lamda=0.98, entropy_loss_weight=0.0, max_grad_norm=0.5,
)


def make_update_trace_config1(cusrl):
    """BASELINE config 1 (MountainCar-v0 `ppo` preset, 8 vectorised envs): one rollout + one full ``agent.update()`` with
    exactly the zoo's agent kwargs — discrete 3-way policy, Tanh (64, 64), observation normalisation, T = 16, 4 x 4
    minibatches — on a random-tensor env of MountainCar's shapes (observation 2, 3 one-hot actions; gymnasium and with
    it the car dynamics are absent from the image, and the update path never sees the dynamics anyway)."""
    from cusrl.testing.environment import DummyTorchEnvironment  # noqa: PLC0415

    cusrl.config.set_device("cpu")
    out = dict(META)
    torch.manual_seed(21)
    env = DummyTorchEnvironment(num_instances=8, observation_dim=2, action_dim=3, reward_dim=1)
    underlying = cusrl.preset.PpoAgentFactory(**MOUNTAIN_CAR_KWARGS).to_underlying()
    trace = {"objectives": [], "indices": [], "grads_unclipped": [], "grads": [], "params_after": []}

    class Capture(cusrl.Hook):
        def __init__(self, where):
            super().__init__()
            self.where = where
            self.name_(f"capture_{where}")

        def pre_optim(self, optimizer):
            flat = torch.cat([p.grad.reshape(-1) for g in optimizer.param_groups for p in g["params"]])
            trace["grads_unclipped" if self.where == "pre" else "grads"].append(np_(flat))

        def post_optim(self):
            if self.where == "post":
                trace["params_after"].append(np_(torch.cat([p.detach().reshape(-1) for _, p in self.agent.named_parameters()])))

        def objective(self, metadata, batch):
            if self.where == "post":
                trace["indices"].append(np_(batch["flat_index"].squeeze(-1)))

    underlying.register_hook(Capture("pre"), before="gradient_clipping")
    underlying.register_hook(Capture("post"), after="gradient_clipping")
    agent = underlying(env.spec)
    state0 = {n: np_(p) for n, p in agent.named_parameters()}
    orig_objective = agent.hook.objective

    def wrapped(metadata, batch, _o=orig_objective):
        res = _o(metadata, batch)
        trace["objectives"].append(np.array([res["value_loss"].item(), res["surrogate_loss"].item(),
                                             res["entropy_loss"].item()], dtype=np.float32))
        return res

    agent.hook.objective = wrapped
    observation, state, _ = env.reset()
    step = 0
    while True:
        action = agent.act(observation, state)
        observation, state, reward, terminated, truncated, _ = env.step(action)
        flat_index = (torch.arange(8) + step * 8).reshape(8, 1)
        ready = agent.step(observation, reward, terminated, truncated, state, flat_index=flat_index)
        step += 1
        if ready:
            break
    buffer_in = {k: np_(v) for k, v in agent.buffer.storage.items()}
    torch.manual_seed(99)
    metrics = agent.update()
    for k, v in state0.items():
        out["param0/" + k] = v
    out["param_names"] = np.array(list(state0.keys()))
    for k, v in buffer_in.items():
        out["buffer_in/" + k] = v
    out["buffer_keys"] = np.array(list(buffer_in.keys()))
    for k in ("next_value", "advantage", "return"):
        out["buffer_out/" + k] = np_(agent.buffer.storage[k])
    for k in ("objectives", "indices"):
        out[k] = np.stack(trace[k])
    kept = [0, 5, len(trace["grads"]) - 1]  # the wide per-step vectors: first, one in the middle, last train step
    out["kept_steps"] = np.array(kept)
    for k in ("grads_unclipped", "grads", "params_after"):
        out[k] = np.stack([trace[k][i] for i in kept])
    out["metric_keys"] = np.array(list(metrics.keys()))
    out["metric_vals"] = np.array(list(metrics.values()), dtype=np.float64)
    rms = agent.hook["observation_normalization"].observation_rms
    out["rms_mean"], out["rms_var"], out["rms_count"] = np_(rms.mean), np_(rms.var), np.array(rms.count)
    np.savez_compressed(HERE / "update_trace_config1.npz", **out)
    print(f"update trace config 1: {len(trace['objectives'])} train steps, buffer leaves {list(buffer_in)}")


# ------------------------------------------------------------------------------- observation normalisation
def make_obs_norm(cusrl):
    """ObservationNormalization pre_act / post_step over a short rollout
    (hook/mdp/observation.py:159-246, nn/layer/rms.py:136-206, nn/utils/normalization.py:15-93)."""
    from types import SimpleNamespace  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(31)
    idx = 0
    for N, C, S, with_state, max_count in [(64, 48, 0, False, None), (16, 5, 7, True, None), (8, 3, 0, False, 40)]:
        spec = cusrl.EnvironmentSpec(C, 2, state_dim=S if with_state else None, num_instances=N)
        agent = SimpleNamespace(environment_spec=spec, observation_dim=C, state_dim=S if with_state else C,
                                has_state=with_state, inference_mode=False, setup_module=lambda m: m,
                                to_tensor=torch.as_tensor)
        hook = cusrl.hook.ObservationNormalization(max_count=max_count)
        hook.pre_init(agent)
        hook.init()
        steps = 6
        p = f"c{idx}_"
        out[p + "params"] = np.array([N, C, S if with_state else 0, steps, -1 if max_count is None else max_count])
        observation = torch.randn(N, C, generator=gen) * 3 + 1
        state = torch.randn(N, S, generator=gen) * 0.5 - 2 if with_state else None
        for t in range(steps):
            tr = {"observation": observation.clone()}
            if with_state:
                tr["state"] = state.clone()
            hook.pre_act(tr)
            out[p + f"obs_in_{t}"] = np_(observation)
            out[p + f"obs_out_{t}"] = np_(tr["observation"])
            if with_state:
                out[p + f"state_in_{t}"] = np_(state)
                out[p + f"state_out_{t}"] = np_(tr["state"])
            next_observation = torch.randn(N, C, generator=gen) * (3 + t) + 1
            next_state = torch.randn(N, S, generator=gen) * 0.5 - 2 if with_state else None
            done = torch.rand(N, 1, generator=gen) < 0.3
            tr.update(next_observation=next_observation.clone(), done=done)
            if with_state:
                tr["next_state"] = next_state.clone()
            hook.post_step(tr)
            out[p + f"next_in_{t}"] = np_(next_observation)
            out[p + f"next_out_{t}"] = np_(tr["next_observation"])
            out[p + f"done_{t}"] = np_(done)
            out[p + f"mean_{t}"] = np_(hook.observation_rms.mean)
            out[p + f"var_{t}"] = np_(hook.observation_rms.var)
            out[p + f"count_{t}"] = np.array(hook.observation_rms.count)
            if with_state:
                out[p + f"next_state_in_{t}"] = np_(next_state)
                out[p + f"next_state_out_{t}"] = np_(tr["next_state"])
                out[p + f"state_mean_{t}"] = np_(hook.state_rms.mean)
                out[p + f"state_var_{t}"] = np_(hook.state_rms.var)
            # reset finished envs with fresh observations (what the trainer does)
            fresh = torch.randn(N, C, generator=gen) * 0.1
            observation = torch.where(done, fresh, next_observation)
            if with_state:
                state = torch.where(done, torch.randn(N, S, generator=gen), next_state)
        idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "obs_norm.npz", **out)
    print("obs_norm.npz:", idx, "cases")


# ------------------------------------------------------------------------------------------- recurrent BPTT
def make_recurrent(cusrl):
    """done-split sequence packing (nn/utils/recurrent.py:63-272) and the GRU / LSTM
    ``Rnn`` wrapper on temporal minibatches (nn/module/rnn.py:237-299)."""
    from cusrl.nn.utils import recurrent as R  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(404)
    idx = 0
    for L, N, C, H, p_done in [(6, 5, 3, 4, 0.3), (24, 16, 8, 6, 0.05), (3, 1, 2, 2, 0.5), (8, 7, 4, 4, 0.0), (5, 4, 2, 3, 1.0)]:
        x = torch.randn(L, N, C, generator=gen)
        done = torch.rand(L, N, 1, generator=gen) < p_done
        memory = torch.randn(N, H, generator=gen)
        padded, mask = R.split_and_pad_sequences(x, done)
        assert torch.equal(R.unpad_and_merge_sequences(padded, mask), x)
        p = f"c{idx}_"
        out[p + "x"], out[p + "done"], out[p + "memory"] = np_(x), np_(done), np_(memory)
        out[p + "padded"], out[p + "mask"] = np_(padded), np_(mask)
        out[p + "scattered"] = np_(R.scatter_memory(memory, done))
        out[p + "sequence_lengths"] = np_(R.compute_sequence_lengths(done))
        out[p + "sequence_indices"] = np_(R.compute_sequence_indices(done))
        idx += 1
    out["num_cases"] = np.array(idx)
    # Rnn wrapper: single step, sequence without done, sequence with done (zero-initialised segments)
    for kind in ("GRU", "LSTM"):
        torch.manual_seed(9)
        rnn = cusrl.Rnn.Factory(kind, hidden_size=8, num_layers=2)(5)
        L, N = 7, 6
        x = torch.randn(L, N, 5, generator=gen)
        done = torch.rand(L, N, 1, generator=gen) < 0.2
        p = kind.lower() + "_"
        for k, v in rnn.state_dict().items():
            out[p + "param/" + k] = np_(v)
        out[p + "param_names"] = np.array(list(rnn.state_dict().keys()))
        out[p + "x"], out[p + "done"] = np_(x), np_(done)
        with torch.no_grad():
            memory = None
            stepwise = []
            memories = []
            for t in range(L):  # rollout: one step at a time, memory reset where done
                flat = (lambda m: np_(torch.cat([m["hidden"], m["cell"]], -1))) if kind == "LSTM" else np_
                memories.append(np.zeros((N, 16 * (2 if kind == "LSTM" else 1)), np.float32) if memory is None else flat(memory))
                y, memory = rnn(x[t], memory=memory, sequential=False)
                stepwise.append(np_(y))
                rnn.reset_memory(memory, done[t])
            out[p + "stepwise"] = np.stack(stepwise)
            out[p + "memories"] = np.stack(memories)
            initial = None
            y_seq, _ = rnn(x, memory=initial, done=done)
            out[p + "sequence_with_done"] = np_(y_seq)
            y_plain, last = rnn(x, memory=None)
            out[p + "sequence_plain"] = np_(y_plain)
    np.savez_compressed(HERE / "recurrent.npz", **out)
    print("recurrent.npz:", idx, "cases + GRU/LSTM wrappers")


def make_recurrent_packed(cusrl):
    """SURVEY.md §8f rank 1, second half: ``gather_memory`` (nn/utils/recurrent.py:124-157), the per-slot time indices
    (``compute_cumulative_timesteps`` / ``compute_reverse_cumulative_timesteps`` ``:28-32,95-99``) and the
    ``pack_sequence=True`` path of the ``Rnn`` wrapper that recovers every env's final recurrent state
    (nn/module/rnn.py:273-291), as in cusrl_test/nn/module/test_rnn.py:88-128."""
    from cusrl.nn.utils import recurrent as R  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(405)
    idx = 0
    for L, N, H, p_done in [(6, 5, 4, 0.3), (24, 16, 6, 0.05), (3, 1, 2, 0.5), (8, 7, 4, 0.0), (5, 4, 3, 1.0), (16, 33, 8, 0.25)]:
        done = torch.rand(L, N, 1, generator=gen) < p_done
        num_sequences = int(R.compute_sequence_lengths(done).numel())
        scattered = torch.randn(num_sequences, H, generator=gen)
        p = f"c{idx}_"
        out[p + "done"], out[p + "scattered"] = np_(done), np_(scattered)
        out[p + "gathered"] = np_(R.gather_memory(scattered, done))
        nested = {"hidden": scattered, "cell": scattered * 2.0}
        out[p + "gathered_cell"] = np_(R.gather_memory(nested, done)["cell"])
        out[p + "cumulative_timesteps"] = np_(R.compute_cumulative_timesteps(done))
        out[p + "reverse_cumulative_timesteps"] = np_(R.compute_reverse_cumulative_timesteps(done))
        out[p + "cumulative_sequence_lengths"] = np_(R.compute_cumulative_sequence_lengths(done))
        idx += 1
    out["num_cases"] = np.array(idx)
    for kind in ("RNN", "GRU", "LSTM"):
        torch.manual_seed(19)
        rnn = cusrl.Rnn.Factory(kind, hidden_size=8, num_layers=2)(5)
        L, N = 12, 9
        warmup = torch.randn(4, N, 5, generator=gen)
        x = torch.randn(L, N, 5, generator=gen)
        done = torch.rand(L, N, 1, generator=gen) > 0.75
        p = kind.lower() + "_"
        for k, v in rnn.state_dict().items():
            out[p + "param/" + k] = np_(v)
        out[p + "param_names"] = np.array(list(rnn.state_dict().keys()))
        out[p + "warmup"], out[p + "x"], out[p + "done"] = np_(warmup), np_(x), np_(done)
        flat = (lambda m: np_(torch.cat([m["hidden"], m["cell"]], -1))) if kind == "LSTM" else np_
        with torch.no_grad():
            _, initial = rnn(warmup)
            out[p + "initial_memory"] = flat(initial)
            clone = (lambda m: {k: v.clone() for k, v in m.items()}) if kind == "LSTM" else torch.clone
            y, memory = rnn(x, memory=clone(initial), done=done, pack_sequence=True)
            out[p + "packed_output"], out[p + "packed_memory"] = np_(y), flat(memory)
            y_unpacked, none = rnn(x, memory=clone(initial), done=done)
            assert none is None
            out[p + "unpacked_output"] = np_(y_unpacked)
    np.savez_compressed(HERE / "recurrent_packed.npz", **out)
    print("recurrent_packed.npz:", idx, "cases + RNN/GRU/LSTM packed forward")


def make_random_sampler(cusrl):
    """SURVEY.md §8f rank 4 (sampler part): ``RandomSampler`` / ``TemporalRandomSampler`` / ``AutoRandomSampler``
    (sampler/random_sampler.py:18-138) on partially filled and wrapped ring buffers; the CPU generator stream
    (``torch.randint``) is part of what is pinned."""
    from cusrl.sampler.random_sampler import AutoRandomSampler, RandomSampler, TemporalRandomSampler  # noqa: PLC0415

    out = dict(META)
    idx = 0
    for capacity, parallelism, pushes, memory in [(4, 1, 3, True), (4, 2, 6, True), (8, 5, 8, False), (6, 3, 15, True), (5, 7, 2, False)]:
        buffer = cusrl.Buffer(capacity=capacity, parallelism=parallelism, device="cpu")
        steps = []
        for step in range(pushes):
            base = torch.arange(parallelism, dtype=torch.float32).unsqueeze(-1) * 100.0 + step
            data = {"observation": torch.cat([base, -base, base * 0.5], dim=-1), "flag": (base.long() % 3 == 0)}
            if memory:
                data["actor_memory"] = base + 1000.0
            buffer.push(data)
            steps.append(data)
        p = f"c{idx}_"
        out[p + "shape"] = np.array([capacity, parallelism, pushes, int(memory)])
        for key in steps[0]:
            out[p + "push/" + key] = np.stack([np_(s[key]) for s in steps])
        cases = [("random", RandomSampler(num_batches=2, batch_size=11)),
                 ("temporal_full", TemporalRandomSampler(num_batches=2, batch_size=4)),
                 ("temporal_2", TemporalRandomSampler(num_batches=3, batch_size=5, sequence_len=2)),
                 ("auto", AutoRandomSampler(num_batches=2, batch_size=6, sequence_len=3))]
        for tag, sampler in cases:
            torch.manual_seed(1000 + idx)
            for j, (metadata, batch) in enumerate(sampler(buffer)):
                q = f"{p}{tag}_{j}_"
                out[q + "temporal"] = np.array(metadata["temporal"])
                for key, value in batch.items():
                    out[q + key] = np_(value)
        idx += 1
    out["num_cases"] = np.array(idx)
    np.savez_compressed(HERE / "random_sampler.npz", **out)
    print("random_sampler.npz:", idx, "buffers x 4 samplers")


# ------------------------------------------------------------------------------------------------ RND / AMP
def make_aux_rewards(cusrl):
    """RandomNetworkDistillation.pre_update / objective (hook/auxiliary/rnd.py:55-81) and
    AdversarialMotionPrior.post_step / objective (hook/auxiliary/amp.py:112-160) with fixed network weights."""
    from types import SimpleNamespace  # noqa: PLC0415

    out = dict(META)
    gen = torch.Generator().manual_seed(77)
    records = {}
    agent = SimpleNamespace(state_dim=10, device=torch.device("cpu"), setup_module=lambda m: m, to_tensor=torch.as_tensor,
                            record=lambda **kw: records.update({k: v.clone() for k, v in kw.items()}),
                            environment_spec=SimpleNamespace(demonstration_sampler=None))
    # --- RND
    torch.manual_seed(3)
    rnd = cusrl.hook.RandomNetworkDistillation(cusrl.Mlp.Factory([12, 8]), output_dim=5, reward_scale=0.25, state_indices=slice(2, 9))
    rnd.pre_init(agent)
    rnd.init()
    for name, module in (("target", rnd.target), ("predictor", rnd.predictor)):
        for k, v in module.state_dict().items():
            out[f"rnd_{name}/{k}"] = np_(v)
    next_obs = torch.randn(6, 9, 10, generator=gen)
    reward = torch.randn(6, 9, 1, generator=gen)
    buffer = {"next_observation": next_obs.clone(), "reward": reward.clone()}
    rnd.pre_update(buffer)
    out["rnd_next_observation"], out["rnd_reward_in"], out["rnd_reward_out"] = np_(next_obs), np_(reward), np_(buffer["reward"])
    out["rnd_reward_metric"] = np_(records["rnd_reward"])
    batch = {"next_observation": next_obs.flatten(0, 1)}
    loss = rnd.objective({}, batch)["rnd_loss"]
    loss.backward()
    out["rnd_loss"] = np_(loss)
    out["rnd_grad"] = np_(torch.cat([p.grad.reshape(-1) for p in rnd.predictor.parameters()]))
    # --- AMP
    torch.manual_seed(4)
    dataset = torch.randn(50, 8, generator=gen)
    amp = cusrl.hook.AdversarialMotionPrior(cusrl.Mlp.Factory([16, 8]), dataset_source=dataset.clone(), state_indices=slice(1, 5),
                                            batch_size=None, reward_scale=0.5, loss_weight=2.0, grad_penalty_weight=5.0)
    amp.pre_init(agent)
    amp.init()
    for k, v in amp.discriminator.state_dict().items():
        out[f"amp_discriminator/{k}"] = np_(v)
    out["amp_dataset"] = np_(dataset)
    steps = 3
    for t in range(steps):
        obs, nobs = torch.randn(7, 10, generator=gen), torch.randn(7, 10, generator=gen)
        rew = torch.randn(7, 1, generator=gen)
        torch.manual_seed(100 + t)  # pins torch.randint inside _sample_demonstration
        tr = {"observation": obs.clone(), "next_observation": nobs.clone(), "reward": rew.clone()}
        amp.post_step(tr)
        out[f"amp_obs_{t}"], out[f"amp_next_obs_{t}"], out[f"amp_reward_in_{t}"] = np_(obs), np_(nobs), np_(rew)
        out[f"amp_reward_out_{t}"] = np_(tr["reward"])
        out[f"amp_agent_transition_{t}"], out[f"amp_expert_transition_{t}"] = np_(tr["agent_transition"]), np_(tr["expert_transition"])
        out[f"amp_rms_mean_{t}"], out[f"amp_rms_var_{t}"] = np_(amp.transition_rms.mean), np_(amp.transition_rms.var)
    batch = {"agent_transition": tr["agent_transition"].clone(), "expert_transition": tr["expert_transition"].clone()}
    losses = amp.objective({}, batch)
    total = sum(losses.values())
    total.backward()
    out["amp_discrimination_loss"], out["amp_grad_penalty_loss"] = np_(losses["amp_discrimination_loss"]), np_(losses["amp_grad_penalty_loss"])
    out["amp_grad"] = np_(torch.cat([p.grad.reshape(-1) for p in amp.discriminator.parameters()]))
    out["amp_steps"] = np.array(steps)
    np.savez_compressed(HERE / "aux_rewards.npz", **out)
    print("aux_rewards.npz: RND + AMP")


# ------------------------------------------------------------------------------------------------ lr schedules
class ScheduleProbe:
    """The slice of an agent the KL-driven LR schedules touch (shared by the generator and tests/test_host_logic.py)."""

    def __init__(self):
        self.optimizer = types.SimpleNamespace(param_groups=[
            {"lr": 2e-4, "param_names": ["actor.backbone.0.weight", "actor.distribution.std"]},
            {"lr": 1e-3, "param_names": ["critic.backbone.0.weight"]},
        ])
        self.metrics = {}
        self.iteration = 0
        self.recorded: list[dict] = []
        self.loads = 0
        self.hook: list = []

    def record(self, **kwargs):
        self.recorded.append(kwargs)

    def state_dict(self):
        return {"marker": self.iteration}

    def load_state_dict(self, state):
        self.loads += 1

    def run(self, hook, kls, schedule_first: bool):
        hook.agent = self
        hook.post_init()
        rows = []
        for i, kl in enumerate(kls):
            self.iteration = i
            if schedule_first:
                hook.apply_schedule(i)
            hook.pre_update(None)
            self.metrics["kl_divergence"] = types.SimpleNamespace(mean=torch.tensor(kl, dtype=torch.float32))
            self.recorded.clear()
            hook.post_update()
            merged = {k: v for item in self.recorded for k, v in item.items()}
            rows.append([self.optimizer.param_groups[0]["lr"], self.optimizer.param_groups[1]["lr"],
                         merged.get("lr_scale", np.nan), merged.get("update_rejected", np.nan), float(self.loads)])
        return np.asarray(rows, dtype=np.float64)


    def run_mini_batch_wise(self, hook, preparation, kls, mini_batches: int):
        """MiniBatchWiseLRSchedule: ``mini_batches`` objective() calls per iteration, each on a [5, 1] KL column whose
        mean is the listed value; one row per call + one after post_update()."""
        self.hook = [preparation]
        hook.agent = self
        hook.post_init()
        rows = []
        spread = torch.tensor([[0.5], [1.5], [1.0], [0.25], [1.75]], dtype=torch.float32)
        for i in range(0, len(kls), mini_batches):
            self.iteration = i // mini_batches
            hook.apply_schedule(self.iteration)
            hook.pre_update(None)
            for kl in kls[i:i + mini_batches]:
                self.recorded.clear()
                assert hook.objective({}, {"kl_divergence": spread * kl}) is None
                merged = {k: v for item in self.recorded for k, v in item.items()}
                rows.append([self.optimizer.param_groups[0]["lr"], self.optimizer.param_groups[1]["lr"],
                             merged.get("lr_scale", np.nan), float(preparation.calculate_kl_divergence), float(self.loads)])
            self.recorded.clear()
            hook.post_update()
            rows.append([self.optimizer.param_groups[0]["lr"], self.optimizer.param_groups[1]["lr"],
                         float(len(self.recorded)), np.nan, float(self.loads)])
        return np.asarray(rows, dtype=np.float64)


MINI_BATCH_WISE_CASES = {
    "mini_batch_wise": dict(desired_kl_divergence=0.01),
    "mini_batch_wise_warmup": dict(desired_kl_divergence=0.008, threshold=1.5, scale_factor=1.2, warmup_iterations=2,
                                   initial_scale=0.5),
}
SCHEDULE_CASES = {
    "adaptive": ("AdaptiveLRSchedule", dict(desired_kl_divergence=0.01), False),
    "adaptive_all_maxkl": ("AdaptiveLRSchedule", dict(desired_kl_divergence=0.02, max_kl_divergence=0.05, scale_all_params=True,
                                                      threshold=0.7, scale_factor=0.3), False),
    "adaptive_warmup": ("AdaptiveLRSchedule", dict(desired_kl_divergence=0.01, warmup_iterations=4, initial_scale=0.25), True),
    "threshold": ("ThresholdLRSchedule", dict(desired_kl_divergence=0.01), False),
    "threshold_maxkl": ("ThresholdLRSchedule", dict(desired_kl_divergence=0.01, max_kl_divergence=0.03, threshold=1.5,
                                                    scale_factor=1.25, scale_all_params=True), False),
}
SCHEDULE_KLS = [0.012, 0.031, 0.004, 0.0, 0.0105, 0.06, 0.02, 0.0007, 0.011, 0.25, 0.009, 0.0031, 0.018, 0.04]


def make_lr_schedule(cusrl):
    out = {"torch_version": np.array(torch.__version__), "kls": np.asarray(SCHEDULE_KLS)}
    for tag, (cls_name, kwargs, schedule_first) in SCHEDULE_CASES.items():
        out[tag] = ScheduleProbe().run(getattr(cusrl.hook, cls_name)(**kwargs), SCHEDULE_KLS, schedule_first)
    for tag, kwargs in MINI_BATCH_WISE_CASES.items():
        out[tag] = ScheduleProbe().run_mini_batch_wise(cusrl.hook.MiniBatchWiseLRSchedule(**kwargs),
                                                       cusrl.hook.OnPolicyPreparation(), SCHEDULE_KLS + SCHEDULE_KLS[:2], 4)
    np.savez_compressed(HERE / "lr_schedule.npz", **out)
    print("lr_schedule.npz", {k: v.shape for k, v in out.items() if k not in ("torch_version",)})


MAKERS = ("gae", "next_value", "randperm", "losses", "merge", "update_trace", "obs_norm", "recurrent", "aux_rewards",
          "lr_schedule", "recurrent_packed", "random_sampler", "categorical_losses", "update_trace_config1")


def main():
    """``make_golden.py`` regenerates every fixture; ``make_golden.py NAME...`` only the named ones (see MAKERS)."""
    cusrl = import_reference()
    cusrl.config.set_device("cpu")
    selected = sys.argv[1:] or MAKERS
    unknown = [name for name in selected if name not in MAKERS]
    assert not unknown, f"unknown fixtures {unknown}; choose from {MAKERS}"
    for name in selected:
        globals()[f"make_{name}"](cusrl)
    leaked = list(REFERENCE.rglob("__pycache__"))
    assert not leaked, f"bytecode leaked into the reference tree: {leaked[:3]}"


if __name__ == "__main__":
    main()
