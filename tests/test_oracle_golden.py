"""Pin the CPU oracle: every function of oracle/cusrl_oracle.c against (i) golden vectors produced by
running the reference (tests/golden/make_golden.py) and (ii) the reference's own known-answer tests
(values quoted from cusrl_test/hook/on_policy/test_gae.py:8-31, test_ppo.py:8-32)."""

import numpy as np
import pytest

import oracle


def cases(npz):
    return range(int(npz["num_cases"]))


def test_gae_bit_exact_vs_reference(golden):
    g = golden("gae")
    for i in cases(g):
        p = f"c{i}_"
        gamma, lamda, lv = g[p + "params"]
        adv, ret = oracle.gae(g[p + "reward"], g[p + "done"], g[p + "value"], g[p + "next_value"], gamma, lamda,
                              None if lv < 0 else lv)
        assert np.array_equal(adv, g[p + "advantage"]), f"case {i}: advantage not bit-exact"
        assert np.array_equal(ret, g[p + "return"]), f"case {i}: return not bit-exact"


def test_gae_known_answers_from_reference_tests():
    # test_gae.py:8-16
    reward = np.ones((3, 1, 1), np.float32)
    done = np.array([False, True, False]).reshape(3, 1, 1)
    zeros = np.zeros_like(reward)
    adv, _ = oracle.gae(reward, done, zeros, zeros, 0.5, 1.0)
    assert np.allclose(adv.ravel(), [1.5, 1.0, 1.0])
    # test_gae.py:19-31
    adv, ret = oracle.gae(np.array([1.0, 2.0]).reshape(2, 1, 1), np.zeros((2, 1, 1), bool),
                          np.array([0.5, 1.0]).reshape(2, 1, 1), np.array([1.0, 0.0]).reshape(2, 1, 1), 0.5, 1.0, 0.0)
    assert np.allclose(adv.ravel(), [1.5, 1.0])
    assert np.allclose(ret.ravel(), [1.5, 2.0])


def test_advantage_normalisation_vs_reference(golden):
    g = golden("gae")
    for i in cases(g):
        p = f"c{i}_"
        adv = g[p + "advantage"]
        if adv.shape[0] * adv.shape[1] < 2:
            continue
        var, mean = oracle.var_mean(adv)
        np.testing.assert_allclose(mean, g[p + "mean"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(var, g[p + "var"], rtol=1e-5)
        # torch's var_mean is alignment dependent at the last ulp (the reference recomputes the statistics on
        # its own copy), so even with the recorded statistics the apply step is compared by tolerance
        np.testing.assert_allclose(oracle.normalize(adv, g[p + "mean"], g[p + "var"]), g[p + "normalized"],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(oracle.normalize(adv, mean, var), g[p + "normalized"], rtol=1e-5, atol=1e-6)


def test_next_value_vs_reference(golden):
    g = golden("next_value")

    def critic(state, D):
        base = np.float32(0.25) * state.sum(-1, keepdims=True, dtype=np.float32) + np.float32(0.5) * state[..., :1]
        return np.concatenate([base * np.float32(d + 1) for d in range(D)], -1).astype(np.float32)

    for i in cases(g):
        p = f"c{i}_"
        value, nobs = g[p + "value"], g[p + "next_observation"]
        term, trunc = g[p + "terminated"], g[p + "truncated"]
        term_value, bootstrap = g[p + "params"]
        D = value.shape[-1]
        tv = critic(nobs[trunc.squeeze(-1)], D)
        out, k = oracle.next_value(value, term, trunc, critic(nobs[-1], D), tv, bool(bootstrap), term_value)
        assert k == int(trunc.sum())
        np.testing.assert_allclose(out, g[p + "next_value"], rtol=1e-6, atol=1e-6)
        # everything that is not a critic output is a pure copy -> bit-exact
        copied = np.ones(value.shape[:2], bool)
        copied[-1] = False
        if bootstrap:
            copied &= ~trunc.squeeze(-1)
        assert np.array_equal(out[copied], g[p + "next_value"][copied])


def test_randperm_stream_bit_exact(golden):
    g = golden("randperm")
    for i in cases(g):
        p = f"c{i}_"
        seed, T, N, mbs = (int(v) for v in g[p + "params"])
        gen = oracle.Mt19937(seed)
        assert np.array_equal(gen.randperm(T * N), g[p + "raw0"])
        assert np.array_equal(gen.randperm(T * N), g[p + "raw1"])
        batches = oracle.mini_batch_indices(seed, T * N, 3, mbs)
        for e in range(3):
            assert np.array_equal(np.concatenate(batches[e]), g[p + "indices"][e])
    # first values observed in SURVEY.md §8c (torch 2.10.0 CPU)
    assert oracle.Mt19937(0).randperm(16)[:8].tolist() == [12, 10, 9, 6, 11, 8, 13, 5]
    assert oracle.Mt19937(0).randperm(98304)[:4].tolist() == [2732, 3934, 72341, 64776]
    assert oracle.Mt19937(42).randperm(4096)[:4].tolist() == [3174, 3363, 876, 1219]


def test_randperm_matches_installed_torch():
    torch = pytest.importorskip("torch")
    for seed, n in [(0, 1), (1, 2), (5, 1000), (123456789, 4097)]:
        torch.manual_seed(seed)
        a = torch.randperm(n)
        b = torch.randperm(n)
        gen = oracle.Mt19937(seed)
        assert np.array_equal(gen.randperm(n), a.numpy())
        assert np.array_equal(gen.randperm(n), b.numpy())


def test_temporal_gather_vs_reference(golden):
    g = golden("randperm")
    T, N = 3, 10
    env = np.arange(N, dtype=np.float32).reshape(N, 1)
    storage = np.stack([env * 100 + t for t in range(T)])
    batches = oracle.mini_batch_indices(3, N, 2, 3)
    got = [oracle.gather_rows(storage, idx, temporal=True) for epoch in batches for idx in epoch]
    assert np.array_equal(np.stack(got), g["temporal_obs"])


def test_gather_rows_is_flat_indexing():
    rng = np.random.default_rng(0)
    storage = rng.standard_normal((5, 7, 3)).astype(np.float32)
    idx = rng.permutation(35)[:20]
    assert np.array_equal(oracle.gather_rows(storage, idx), storage.reshape(35, 3)[idx])
    flags = rng.random((5, 7, 1)) < 0.5
    assert np.array_equal(oracle.gather_rows(flags, idx), flags.reshape(35, 1)[idx])
    assert oracle.gather_rows(storage, np.zeros(0, np.int64)).shape == (0, 3)


def test_buffer_push_copies_one_step():
    storage = np.zeros((3, 2, 4), np.float32)
    step = np.arange(8, dtype=np.float32).reshape(2, 4)
    oracle.buffer_push(step, storage, 1)
    assert np.array_equal(storage[1], step) and not storage[0].any() and not storage[2].any()


def test_merge_mean_var_vs_reference(golden):
    g = golden("merge_mean_var")
    for i in cases(g):
        p = f"c{i}_"
        mean, var = oracle.merge_mean_var(g[p + "means"], g[p + "vars"])
        np.testing.assert_allclose(mean, g[p + "mean"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(var, g[p + "var"], rtol=1e-6)


def test_losses_and_gradients_vs_reference(golden):
    g = golden("losses")
    for i in cases(g):
        p = f"c{i}_"
        clip, vclip, w_sur, w_val, w_ent = g[p + "params"]
        out = oracle.ppo_loss(g[p + "advantage"], g[p + "old_logp"], g[p + "action"], g[p + "mean"], g[p + "std"],
                              g[p + "ret"], g[p + "curr_value"], g[p + "old_value"], clip=clip,
                              value_clip=None if vclip < 0 else vclip, w_sur=w_sur, w_val=w_val, w_ent=w_ent)
        np.testing.assert_allclose(out["logp"], g[p + "logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["entropy"], g[p + "entropy"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out["ratio"], g[p + "ratio"], rtol=2e-5)
        np.testing.assert_allclose(out["losses"][0], g[p + "value_loss"], rtol=1e-5)
        np.testing.assert_allclose(out["losses"][1], g[p + "surrogate"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(out["losses"][2], g[p + "entropy_loss"], rtol=1e-5, atol=1e-8)
        scale = 1.0 / g[p + "mean"].shape[0]
        np.testing.assert_allclose(out["d_mean"], g[p + "d_mean"], rtol=1e-4, atol=1e-6 * scale)
        np.testing.assert_allclose(out["d_std"], g[p + "d_std"], rtol=1e-4, atol=1e-5 * scale)
        np.testing.assert_allclose(out["d_value"], g[p + "d_value"], rtol=1e-5, atol=1e-7 * scale)
        lp, en = oracle.normal_logp_entropy(g[p + "action"], g[p + "mean"], g[p + "std"])
        np.testing.assert_allclose(lp, g[p + "logp"], rtol=1e-5, atol=1e-5)
        kl = oracle.normal_kl(g[p + "old_mean"], g[p + "old_std"], g[p + "mean"], g[p + "std"])
        np.testing.assert_allclose(kl, g[p + "kl"], rtol=1e-4, atol=1e-5)


def test_loss_known_answers_from_reference_tests():
    # test_ppo.py:8-14 — A=[1,-2], ratio=[1.5,0.5], eps=0.2 -> 0.2.  Build ratio through logp: one action dim,
    # std=1, action=mean -> logp = -log(sqrt(2pi)); old_logp = logp - log(ratio).
    mean = np.zeros((2, 1), np.float32)
    std = np.ones((2, 1), np.float32)
    logp = np.float32(-np.log(np.sqrt(2 * np.pi)))
    old_logp = (logp - np.log(np.array([[1.5], [0.5]]))).astype(np.float32)
    out = oracle.ppo_loss(np.array([[1.0], [-2.0]]), old_logp, mean, mean, std, np.zeros((2, 1)), np.zeros((2, 1)),
                          clip=0.2, w_sur=1.0, w_val=0.5, w_ent=0.5)
    assert out["losses"][1] == pytest.approx(0.2, rel=1e-5)
    # test_ppo.py:28-32 form: entropy loss = -mean(entropy) * w
    assert out["losses"][2] == pytest.approx(-0.5 * (0.5 + 0.5 * np.log(2 * np.pi)), rel=1e-6)


# ------------------------------------------------------------------------------------------------ torch port
def _load_port(g, tag):
    import torch

    from oracle.torch_ppo import TorchPpo

    kw = dict(zip((str(k) for k in g[tag + "_factory_keys"]), g[tag + "_factory_vals"]))
    port = TorchPpo(
        16, 8, 8, num_steps_per_update=int(kw["num_steps_per_update"]), hidden=(32, 16), epochs=int(kw["sampler_epochs"]),
        mini_batches=int(kw["sampler_mini_batches"]),
        lamda_value=kw.get("gae_lamda_value"), value_clip=kw.get("value_loss_clip"),
    )
    named = dict(list(port.actor.named_parameters(prefix="actor")) + list(port.critic.named_parameters(prefix="critic")))
    assert list(named) == [str(n) for n in g[tag + "_param_names"]]
    with torch.no_grad():
        for name, param in named.items():
            param.copy_(torch.from_numpy(g[f"{tag}_param0/{name}"]))
    for key in g[tag + "_buffer_keys"]:
        port.storage[str(key)] = torch.from_numpy(g[f"{tag}_buffer_in/{key}"].copy())
    return port


@pytest.mark.parametrize("tag", ["a", "b"])
def test_torch_port_replays_reference_update(golden, tag):
    """The CPU baseline path (oracle/torch_ppo.py) reproduces one full reference ``agent.update()``:
    permutations bit-exact, losses / gradients / parameters within 1e-5."""
    torch = pytest.importorskip("torch")
    g = golden("update_trace")
    port = _load_port(g, tag)
    port.trace = {k: [] for k in ("indices", "objectives", "grads_unclipped", "grads", "params_after")}
    torch.manual_seed(99)
    metrics = port.update()
    assert np.array_equal(torch.stack(port.trace["indices"]).numpy(), g[tag + "_indices"])  # bit-exact permutations
    for key in ("next_value", "advantage", "return"):
        np.testing.assert_allclose(port.storage[key].numpy(), g[f"{tag}_buffer_out/{key}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(torch.stack(port.trace["objectives"]).numpy(), g[tag + "_objectives"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(port.trace["grads_unclipped"]).numpy(), g[tag + "_grads_unclipped"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(torch.stack(port.trace["grads"]).numpy(), g[tag + "_grads"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(torch.stack(port.trace["params_after"]).numpy(), g[tag + "_params_after"], rtol=1e-5, atol=1e-6)
    ref = dict(zip((str(k) for k in g[tag + "_metric_keys"]), g[tag + "_metric_vals"]))
    np.testing.assert_allclose(metrics["kl_divergence"].item(), ref["Agent/kl_divergence"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(metrics["action_std"].item(), ref["Agent/action_std"], rtol=1e-5)


@pytest.mark.parametrize("max_norm", [0.5, 1e6])
def test_clip_grad_norm_matches_torch_utility(max_norm):
    """The reference's GradientClipping calls torch.nn.utils.clip_grad_norm_ (gradient_clipping.py:67-83)."""
    import torch

    rng = np.random.default_rng(7)
    grad = rng.standard_normal(92569).astype(np.float32) * 0.01
    param = torch.nn.Parameter(torch.zeros(grad.size))
    param.grad = torch.from_numpy(grad.copy())
    total = torch.nn.utils.clip_grad_norm_([param], max_norm)
    clipped, norm = oracle.clip_grad_norm(grad, max_norm)
    np.testing.assert_allclose(norm, total.item(), rtol=1e-6)
    np.testing.assert_allclose(clipped, param.grad.numpy(), rtol=1e-6)


@pytest.mark.parametrize("cls_name,weight_decay", [("Adam", 0.0), ("Adam", 0.01), ("AdamW", 0.01)])
def test_adam_step_matches_torch_optimizer(cls_name, weight_decay):
    """cusrl/preset/ppo.py builds torch.optim.AdamW / Adam; the oracle restates one step of it."""
    import torch

    rng = np.random.default_rng(11)
    param = rng.standard_normal(1000).astype(np.float32)
    reference = torch.nn.Parameter(torch.from_numpy(param.copy()))
    optimizer = getattr(torch.optim, cls_name)([reference], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    exp_avg, exp_avg_sq, step = np.zeros_like(param), np.zeros_like(param), 0
    for _ in range(4):
        grad = rng.standard_normal(1000).astype(np.float32) * 0.1
        reference.grad = torch.from_numpy(grad.copy())
        optimizer.step()
        param, exp_avg, exp_avg_sq, step = oracle.adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=2e-4, weight_decay=weight_decay,
                                                            decoupled=cls_name == "AdamW")
        np.testing.assert_allclose(param, reference.detach().numpy(), rtol=1e-6, atol=1e-7)
    state = optimizer.state[reference]
    np.testing.assert_allclose(exp_avg, state["exp_avg"].numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(exp_avg_sq, state["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-10)


def test_policy_stats_matches_torch_distributions():
    """OnPolicyStatistics (stats.py:28-40) goes through torch.distributions for Normal policies."""
    import torch
    from torch.distributions import Normal, kl_divergence

    rng = np.random.default_rng(5)
    B, A = 257, 6
    mp, mq = rng.standard_normal((B, A)).astype(np.float32), rng.standard_normal((B, A)).astype(np.float32)
    sp, sq = (rng.random((B, A)) + 0.5).astype(np.float32), (rng.random((B, A)) + 0.5).astype(np.float32)
    action = (mp + sp * rng.standard_normal((B, A))).astype(np.float32)
    advantage = rng.standard_normal((B, 1)).astype(np.float32)
    p, q = Normal(torch.from_numpy(mp), torch.from_numpy(sp)), Normal(torch.from_numpy(mq), torch.from_numpy(sq))
    old_logp = p.log_prob(torch.from_numpy(action)).sum(-1, keepdim=True)
    kl = kl_divergence(p, q).sum(-1, keepdim=True).mean().item()
    iw = (torch.from_numpy(advantage) * (q.log_prob(torch.from_numpy(action)).sum(-1, keepdim=True) - old_logp).exp()).mean().item()
    got = oracle.policy_stats(mp, sp, mq, sq, action, old_logp.numpy(), advantage)
    np.testing.assert_allclose(got, (kl, iw, float(sq.mean())), rtol=2e-5)


def test_gather_memory_restatement_vs_reference(golden):
    """oracle.gather_memory (numpy restatement of nn/utils/recurrent.py:124-157) against the reference's outputs."""
    g = golden("recurrent_packed")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        assert np.array_equal(oracle.gather_memory(g[p + "scattered"], g[p + "done"]), g[p + "gathered"]), i


def test_categorical_objective_restatement_vs_reference(golden):
    """oracle.categorical_ppo_loss against losses and autograd gradients recorded from the reference's
    OneHotCategoricalDist + PPO hooks (golden ``categorical_losses.npz``); 1e-5 relative fp32."""
    g = golden("categorical_losses")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        clip, vclip, w_sur, w_val, w_ent = g[p + "params"]
        ref = oracle.categorical_ppo_loss(g[p + "advantage"], g[p + "old_logp"], g[p + "action"], g[p + "logits"], g[p + "ret"],
                                          g[p + "curr_value"], g[p + "old_value"], clip=clip, value_clip=None if vclip < 0 else vclip,
                                          w_sur=w_sur, w_val=w_val, w_ent=w_ent)
        B = g[p + "logits"].shape[0]
        np.testing.assert_allclose(ref["logp"], g[p + "logp"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ref["entropy"], g[p + "entropy"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ref["ratio"], g[p + "ratio"], rtol=2e-5)
        np.testing.assert_allclose(ref["losses"], [g[p + "value_loss"], g[p + "surrogate"], g[p + "entropy_loss"]], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ref["d_logits"], g[p + "d_logits"], rtol=1e-4, atol=2e-7 / B)
        np.testing.assert_allclose(ref["d_value"], g[p + "d_value"], rtol=1e-5, atol=1e-7 / B)


def test_sequence_layout_restatement_vs_reference(golden):
    """oracle.sequence_lengths / sequence_layout against the reference's split_and_pad_sequences outputs."""
    g = golden("recurrent")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        x, done, padded, mask = g[p + "x"], g[p + "done"], g[p + "padded"], g[p + "mask"]
        assert np.array_equal(oracle.sequence_lengths(done), g[p + "sequence_lengths"])
        dest, num_sequences = oracle.sequence_layout(done)
        assert num_sequences == padded.shape[1]
        assert np.array_equal(padded.reshape(-1, x.shape[-1])[dest], x.reshape(-1, x.shape[-1]))
        valid = np.zeros(mask.size, bool)
        valid[dest] = True
        assert np.array_equal(valid.reshape(mask.shape), mask)


def test_categorical_draw_restatement_is_the_exponential_race():
    """oracle.categorical_sample (checker of cusrl_categorical_sample_logp): the winner of p_j / q_j, one-hot, with the
    log-softmax of the taken category — and, fed Exp(1) race variables, it draws category j with probability p_j."""
    rng = np.random.default_rng(5)
    logits = rng.standard_normal((4000, 5)) * 1.5
    noise = rng.exponential(1.0, logits.shape)
    taken, action, logp, margin = oracle.categorical_sample(logits, noise)
    p = np.exp(logits - logits.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    assert np.array_equal(taken, (p / noise).argmax(-1)) and np.array_equal(action.argmax(-1), taken)
    assert np.array_equal(action.sum(-1), np.ones(len(taken))) and (margin >= 1.0).all()
    np.testing.assert_allclose(logp[:, 0], np.log(p[np.arange(len(taken)), taken]), rtol=1e-5, atol=1e-6)
    row = np.array([2.0, 0.5, -1.0, -3.0])
    draws, _, _, _ = oracle.categorical_sample(np.tile(row, (200000, 1)), rng.exponential(1.0, (200000, 4)))
    freq = np.bincount(draws, minlength=4) / draws.size
    expect = np.exp(row) / np.exp(row).sum()
    np.testing.assert_allclose(freq, expect, atol=4e-3)


def test_policy_terms_restatement_vs_reference(golden):
    """oracle.policy_terms_f64 / categorical_terms_f64 against the reference: the four batch entries OnPolicyPreparation
    leaves (recorded in losses.npz / categorical_losses.npz), and their vector-Jacobian products — fed the gradients the PPO
    objective sends into `ratio` and `entropy`, they must reproduce the reference's recorded autograd gradients d_mean / d_std /
    d_logits (1e-5 of the tensor's largest entry)."""
    g = golden("losses")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        clip, vclip, w_sur, w_val, w_ent = g[p + "params"]
        B = g[p + "mean"].shape[0]
        adv, ratio = g[p + "advantage"].astype(np.float64).reshape(-1), g[p + "ratio"].astype(np.float64).reshape(-1)
        lo, hi = np.float64(np.float32(1.0 - clip)), np.float64(np.float32(1.0 + clip))
        s1, s2 = adv * ratio, adv * np.clip(ratio, lo, hi)
        inside = (ratio >= lo) & (ratio <= hi)
        d_ratio = np.where(s1 < s2, adv, np.where(s1 > s2, np.where(inside, adv, 0.0), 0.5 * adv + np.where(inside, 0.5 * adv, 0.0)))
        out = oracle.policy_terms_f64(g[p + "mean"], g[p + "std"], g[p + "action"], g[p + "old_logp"],
                                      g_ratio=(-w_sur / B) * d_ratio, g_entropy=np.full(B, -w_ent / B))
        np.testing.assert_allclose(out["logp"], g[p + "logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["entropy"], g[p + "entropy"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out["logp_ratio"], g[p + "logp_ratio"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["ratio"], g[p + "ratio"], rtol=2e-5)
        assert oracle.gradient_error(out["d_mean"], g[p + "d_mean"]) <= 1e-5
        assert oracle.gradient_error(out["d_std"], g[p + "d_std"]) <= 1e-5
    g = golden("categorical_losses")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        clip, vclip, w_sur, w_val, w_ent = g[p + "params"]
        B = g[p + "logits"].shape[0]
        adv, ratio = g[p + "advantage"].astype(np.float64).reshape(-1), g[p + "ratio"].astype(np.float64).reshape(-1)
        lo, hi = np.float64(np.float32(1.0 - clip)), np.float64(np.float32(1.0 + clip))
        s1, s2 = adv * ratio, adv * np.clip(ratio, lo, hi)
        inside = (ratio >= lo) & (ratio <= hi)
        d_ratio = np.where(s1 < s2, adv, np.where(s1 > s2, np.where(inside, adv, 0.0), 0.5 * adv + np.where(inside, 0.5 * adv, 0.0)))
        out = oracle.categorical_terms_f64(g[p + "logits"], g[p + "action"], g[p + "old_logp"],
                                           g_ratio=(-w_sur / B) * d_ratio, g_entropy=np.full(B, -w_ent / B))
        np.testing.assert_allclose(out["logp"], g[p + "logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["entropy"], g[p + "entropy"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out["ratio"], g[p + "ratio"], rtol=2e-5)
        assert oracle.gradient_error(out["d_logits"], g[p + "d_logits"]) <= 1e-5
