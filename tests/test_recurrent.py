"""Recurrent BPTT path: done-split sequence packing and the GRU / LSTM wrapper against vectors
recorded from the reference (golden ``recurrent.npz``): layouts and copies bit-exact, RNN outputs 1e-5."""

import numpy as np
import pytest
import torch

import cusrl_amd as cusrl

DEV = "cuda:0"


def load_rnn(g, kind, device):
    rnn = cusrl.Rnn.Factory(kind, hidden_size=8, num_layers=2)(5)
    names = [str(n) for n in g[f"{kind.lower()}_param_names"]]
    assert list(rnn.state_dict().keys()) == names
    rnn.load_state_dict({n: torch.from_numpy(g[f"{kind.lower()}_param/{n}"].copy()) for n in names})
    return rnn.to(device)


@pytest.mark.parametrize("kind", ["GRU", "LSTM"])
def test_rollout_stepping_and_plain_sequences_match_reference(golden, kind):
    """Host-side module logic (memory layout [N, layers*hidden], reset at done) — runs on CPU."""
    g = golden("recurrent")
    p = kind.lower() + "_"
    rnn = load_rnn(g, kind, "cpu")
    x, done = torch.from_numpy(g[p + "x"].copy()), torch.from_numpy(g[p + "done"].copy())
    with torch.no_grad():
        memory = None
        for t in range(x.size(0)):
            if memory is not None:
                flat = torch.cat([memory["hidden"], memory["cell"]], -1) if kind == "LSTM" else memory
                np.testing.assert_allclose(flat.numpy(), g[p + "memories"][t], rtol=1e-5, atol=1e-6)
            y, memory = rnn(x[t], memory=memory, sequential=False)
            np.testing.assert_allclose(y.numpy(), g[p + "stepwise"][t], rtol=1e-5, atol=1e-6)
            rnn.reset_memory(memory, done[t])
        y, _ = rnn(x, memory=None)
        np.testing.assert_allclose(y.numpy(), g[p + "sequence_plain"], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError, match="only when 'sequential'"):
        rnn(x[0], done=done[0], sequential=False)


@pytest.mark.gpu
def test_sequence_packing_bit_exact_vs_reference(golden):
    from cusrl_amd.nn import recurrent as R

    g = golden("recurrent")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        x, done = torch.from_numpy(g[p + "x"].copy()).to(DEV), torch.from_numpy(g[p + "done"].copy()).to(DEV)
        memory = torch.from_numpy(g[p + "memory"].copy()).to(DEV)
        padded, mask = R.split_and_pad_sequences(x, done)
        assert np.array_equal(padded.cpu().numpy(), g[p + "padded"]), f"case {i}"
        assert np.array_equal(mask.cpu().numpy(), g[p + "mask"])
        assert torch.equal(R.unpad_and_merge_sequences(padded, mask), x)
        bare_mask = mask.clone()  # no cached layout: rebuilt from the mask by ordered compaction
        assert torch.equal(R.unpad_and_merge_sequences(padded, bare_mask), x)
        assert np.array_equal(R.scatter_memory(memory, done).cpu().numpy(), g[p + "scattered"])
        assert np.array_equal(R.compute_sequence_lengths(done).cpu().numpy(), g[p + "sequence_lengths"])
        assert np.array_equal(R.compute_sequence_indices(done).cpu().numpy(), g[p + "sequence_indices"])
    # nested (LSTM) memories and wide rows
    done = torch.rand(24, 300, 1, device=DEV) < 0.04
    mem = {"hidden": torch.randn(300, 512, device=DEV), "cell": torch.randn(300, 512, device=DEV)}
    layout = R.compute_sequence_layout(done)
    out = R.scatter_memory(mem, done, layout)
    assert out["hidden"].shape == (layout.num_sequences, 512)
    assert torch.equal(out["cell"][layout.first_seq], mem["cell"]) and int((out["cell"].abs().sum(-1) > 0).sum()) == 300
    x = torch.randn(24, 300, 48, device=DEV)
    padded, mask = R.split_and_pad_sequences(x, done, layout)
    assert torch.equal(R.unpad_and_merge_sequences(padded, layout), x) and int(mask.sum()) == 24 * 300
    assert not padded.transpose(0, 1)[~mask.transpose(0, 1)].any()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["GRU", "LSTM"])
def test_sequences_with_episode_boundaries_match_reference(golden, kind):
    g = golden("recurrent")
    p = kind.lower() + "_"
    rnn = load_rnn(g, kind, DEV)
    x, done = torch.from_numpy(g[p + "x"].copy()).to(DEV), torch.from_numpy(g[p + "done"].copy()).to(DEV)
    with torch.no_grad():
        y, memory = rnn(x, memory=None, done=done)
    assert memory is None
    np.testing.assert_allclose(y.cpu().numpy(), g[p + "sequence_with_done"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(y.cpu().numpy(), g[p + "stepwise"], rtol=1e-5, atol=2e-6)  # == the rollout's own outputs
    # BPTT through the scatter / gather: gradients equal those of the step-by-step evaluation with memory resets
    x.requires_grad_(True)
    rnn(x, memory=None, done=done)[0].square().sum().backward()
    packed = [p.grad.clone() for p in rnn.parameters()] + [x.grad.clone()]
    for p in rnn.parameters():
        p.grad = None
    x.grad = None
    memory, outputs = None, []
    for t in range(x.size(0)):
        y, memory = rnn(x[t], memory=memory, sequential=False)
        outputs.append(y)
        keep = (~done[t]).float()
        memory = {k: v * keep for k, v in memory.items()} if kind == "LSTM" else memory * keep
    torch.stack(outputs).square().sum().backward()
    for got, want in zip(packed, [p.grad for p in rnn.parameters()] + [x.grad]):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rnn_type", ["GRU", "LSTM"])
def test_recurrent_ppo_preset_trains(rnn_type):
    cusrl.config.set_device(DEV)
    cusrl.set_global_seed(2)

    class Consistency(cusrl.Hook):
        checked = 0

        def objective(self, metadata, batch):
            # cusrl_test/_helpers.py:76-94: the policy re-evaluated on the first minibatch reproduces the rollout's means
            if metadata["epoch_index"] == 0 and metadata["mini_batch_index"] == 0:
                assert metadata["temporal"] is True and batch["observation"].dim() == 3
                error = (batch["curr_action_dist"]["mean"] - batch["action_dist"]["mean"]).abs().max()
                assert error < 1e-4, f"Max error {error}"
                Consistency.checked += 1

    factory = cusrl.preset.RecurrentPpoAgentFactory(
        rnn_type=rnn_type, actor_hidden_size=32, critic_hidden_size=32, num_steps_per_update=8, sampler_epochs=1,
        sampler_mini_batches=1).to_underlying()
    factory.register_hook(Consistency())
    env = cusrl.testing.DummyTorchEnvironment(num_instances=16, observation_dim=10, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, factory, num_iterations=3, verbose=False)
    trainer.run_training_loop()
    assert Consistency.checked == 3
    keys = set(trainer.agent.buffer.storage)
    assert {"actor_memory", "critic_memory", "next_critic_memory"} <= {k.split(".")[0] for k in keys}
    assert np.isfinite(trainer.last_info["Agent/value_loss"]) and np.isfinite(trainer.last_info["Agent/kl_divergence"])


@pytest.mark.gpu
def test_gather_memory_and_time_indices_bit_exact_vs_reference(golden):
    """cusrl_gather_memory + the layout kernel's per-sequence outputs against the reference's recurrent.py:28-157."""
    from cusrl_amd import _native
    from cusrl_amd.nn import recurrent as R

    g = golden("recurrent_packed")
    before = _native.launch_counts.get("cusrl_gather_memory", 0)
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        done = torch.from_numpy(g[p + "done"].copy()).to(DEV)
        scattered = torch.from_numpy(g[p + "scattered"].copy()).to(DEV)
        layout = R.compute_sequence_layout(done)
        assert np.array_equal(R.gather_memory(scattered, done, layout).cpu().numpy(), g[p + "gathered"]), f"case {i}"
        nested = R.gather_memory({"hidden": scattered, "cell": scattered * 2.0}, done)
        assert np.array_equal(nested["cell"].cpu().numpy(), g[p + "gathered_cell"])
        assert np.array_equal(R.compute_cumulative_timesteps(done).cpu().numpy(), g[p + "cumulative_timesteps"])
        assert np.array_equal(R.compute_reverse_cumulative_timesteps(done).cpu().numpy(), g[p + "reverse_cumulative_timesteps"])
        assert np.array_equal(R.compute_cumulative_sequence_lengths(done).cpu().numpy(), g[p + "cumulative_sequence_lengths"])
        # scatter followed by gather returns the stored memory of every env that did not finish at the last step
        memory = torch.randn(done.shape[1], 7, device=DEV)
        roundtrip = R.gather_memory(R.scatter_memory(memory, done, layout), done, layout)
        single = (layout.lengths.numel() == done.shape[1])  # no boundary inside the batch: first == last sequence
        if single:
            assert torch.equal(roundtrip, memory * (~done[-1]).float())
    assert _native.launch_counts.get("cusrl_gather_memory", 0) - before >= 3 * int(g["num_cases"])
    # config-4 row width: [Ns, 512] states of 16 384 envs
    done = torch.rand(24, 16384, 1, device=DEV) < 0.02
    layout = R.compute_sequence_layout(done)
    states = torch.randn(layout.num_sequences, 512, device=DEV)
    out = R.gather_memory(states, done, layout)
    expect = states[layout.last_seq] * (~done[-1]).float()
    assert torch.equal(out, expect)
    assert int(layout.lengths.sum()) == 24 * 16384 and int(layout.lengths.max()) <= 24


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNN", "GRU", "LSTM"])
def test_packed_sequences_recover_the_final_memory(golden, kind):
    """``pack_sequence=True`` (rnn.py:273-291): outputs and the per-env final state vs the reference's own run."""
    g = golden("recurrent_packed")
    p = kind.lower() + "_"
    rnn = cusrl.Rnn.Factory(kind, hidden_size=8, num_layers=2)(5)
    names = [str(n) for n in g[p + "param_names"]]
    assert list(rnn.state_dict().keys()) == names
    rnn.load_state_dict({n: torch.from_numpy(g[p + "param/" + n].copy()) for n in names})
    rnn = rnn.to(DEV)
    x, done = torch.from_numpy(g[p + "x"].copy()).to(DEV), torch.from_numpy(g[p + "done"].copy()).to(DEV)
    warmup = torch.from_numpy(g[p + "warmup"].copy()).to(DEV)

    def flat(memory):
        return torch.cat([memory["hidden"], memory["cell"]], -1) if kind == "LSTM" else memory

    with torch.no_grad():
        _, initial = rnn(warmup)
        np.testing.assert_allclose(flat(initial).cpu().numpy(), g[p + "initial_memory"], rtol=1e-5, atol=2e-6)
        clone = {k: v.clone() for k, v in initial.items()} if kind == "LSTM" else initial.clone()
        y, memory = rnn(x, memory=clone, done=done, pack_sequence=True)
    np.testing.assert_allclose(y.cpu().numpy(), g[p + "packed_output"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(flat(memory).cpu().numpy(), g[p + "packed_memory"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(y.cpu().numpy(), g[p + "unpacked_output"], rtol=1e-5, atol=2e-6)
    # the step-by-step rollout ends in the same state (cusrl_test/nn/module/test_rnn.py:88-128)
    with torch.no_grad():
        step_memory = {k: v.clone() for k, v in initial.items()} if kind == "LSTM" else initial.clone()
        for t in range(x.size(0)):
            _, step_memory = rnn(x[t], memory=step_memory, sequential=False)
            rnn.reset_memory(step_memory, done[t])
    torch.testing.assert_close(flat(memory), flat(step_memory), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError, match="Packed RNN input must be 3D"):
        rnn(torch.randn(4, 3, 2, 5, device=DEV), done=torch.zeros(4, 3, 1, dtype=torch.bool, device=DEV), pack_sequence=True)


# ------------------------------------------------------------------------------------------------ GRU as GEMMs + HIP gate passes
def _golden_gru_weights(g):
    return [tuple(g[f"gru_param/rnn.{name}_l{layer}"] for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for layer in (0, 1)]


def test_oracle_gru_is_pinned_to_the_reference_rollout(golden):
    """oracle.gru_sequence (the numpy restatement the GPU tests check the HIP gate passes with) against outputs and
    memories recorded from the reference's Rnn wrapper: the plain sequence, and the step-by-step rollout with resets."""
    import oracle

    g = golden("recurrent")
    weights = _golden_gru_weights(g)
    x, done = g["gru_x"], g["gru_done"]
    out, _ = oracle.gru_sequence(x, None, weights)
    np.testing.assert_allclose(out, g["gru_sequence_plain"], rtol=1e-5, atol=1e-6)
    h = np.zeros((2, x.shape[1], 8), np.float32)
    for t in range(x.shape[0]):
        np.testing.assert_allclose(np.concatenate([h[0], h[1]], -1), g["gru_memories"][t], rtol=1e-5, atol=1e-6)
        y, h = oracle.gru_sequence(x[t:t + 1], h, weights)
        np.testing.assert_allclose(y[0], g["gru_stepwise"][t], rtol=1e-5, atol=1e-6)
        h = h * (~done[t]).astype(np.float32)[None]
    # lengths: the state freezes and the output is zero from each sequence's end on
    lengths = np.array([7, 3, 1, 5, 7, 2])
    out_l, h_l = oracle.gru_sequence(x, None, weights, lengths)
    for b, n in enumerate(lengths):
        ref_out, ref_h = oracle.gru_sequence(x[:n, b:b + 1], None, weights)
        np.testing.assert_allclose(out_l[:n, b], ref_out[:, 0], rtol=1e-6, atol=1e-7)
        assert not out_l[n:, b].any()
        np.testing.assert_allclose(h_l[:, b], ref_h[:, 0], rtol=1e-6, atol=1e-7)


def test_oracle_lstm_is_pinned_to_the_reference_rollout(golden):
    import oracle

    g = golden("recurrent")
    weights = [tuple(g[f"lstm_param/rnn.{name}_l{layer}"] for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for layer in (0, 1)]
    x, done = g["lstm_x"], g["lstm_done"]
    out, _ = oracle.lstm_sequence(x, None, weights)
    np.testing.assert_allclose(out, g["lstm_sequence_plain"], rtol=1e-5, atol=1e-6)
    h = c = np.zeros((2, x.shape[1], 8), np.float32)
    for t in range(x.shape[0]):
        flat = np.concatenate([h[0], h[1], c[0], c[1]], -1)
        np.testing.assert_allclose(flat, g["lstm_memories"][t], rtol=1e-5, atol=1e-6)
        y, (h, c) = oracle.lstm_sequence(x[t:t + 1], (h, c), weights)
        np.testing.assert_allclose(y[0], g["lstm_stepwise"][t], rtol=1e-5, atol=1e-6)
        keep = (~done[t]).astype(np.float32)[None]
        h, c = h * keep, c * keep


def test_oracle_rnn_is_pinned_to_the_reference_packed_run(golden):
    """oracle.rnn_sequence against the reference's own vanilla-RNN run (recurrent_packed.npz): the warm-up's final state,
    every output of the done-split sequence (== stepping with resets), and the per-env final memory."""
    import oracle

    g = golden("recurrent_packed")
    weights = [tuple(g[f"rnn_param/rnn.{name}_l{layer}"] for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for layer in (0, 1)]
    _, h = oracle.rnn_sequence(g["rnn_warmup"], None, weights)
    np.testing.assert_allclose(np.concatenate([h[0], h[1]], -1), g["rnn_initial_memory"], rtol=1e-5, atol=1e-6)
    x, done = g["rnn_x"], g["rnn_done"]
    for t in range(x.shape[0]):
        y, h = oracle.rnn_sequence(x[t:t + 1], h, weights)
        np.testing.assert_allclose(y[0], g["rnn_packed_output"][t], rtol=1e-5, atol=2e-6)
        h = h * (~done[t]).astype(np.float32)[None]
    np.testing.assert_allclose(np.concatenate([h[0], h[1]], -1), g["rnn_packed_memory"], rtol=1e-5, atol=2e-6)


def test_length_plan_is_the_packed_sequence_schedule():
    """The host-side plan of the fused cores (sort by length, per-step prefix sizes) against torch's PackedSequence."""
    from cusrl_amd.nn.gru import _LengthPlan

    gen = torch.Generator().manual_seed(3)
    for L, B in ((7, 6), (24, 300), (1, 5), (5, 1)):
        lengths = torch.randint(1, L + 1, (B,), generator=gen)
        plan = _LengthPlan(lengths, L + 2)  # two steps past the longest sequence: nothing runs there
        x = torch.randn(L, B, 3, generator=gen)
        packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, enforce_sorted=False)
        sizes = packed.batch_sizes.tolist()
        assert plan.sizes == sizes + [0] * (L + 2 - len(sizes))
        assert torch.equal(lengths[plan.order], lengths[packed.sorted_indices])  # ties may be ordered differently
        assert torch.equal(plan.inverse[plan.order], torch.arange(B))
        assert torch.equal(plan.unsort(plan.sort(x, 1), 1), x)


def _twin_grus(I, H, layers, bias):
    from cusrl_amd.nn.rnn import _Gru

    torch.manual_seed(I * 1000 + H)
    plain = torch.nn.GRU(I, H, layers, bias=bias).to(DEV)  # MIOpen
    fused = _Gru(I, H, layers, bias=bias).to(DEV)
    fused.load_state_dict(plain.state_dict())
    return plain, fused


@pytest.mark.gpu
@pytest.mark.parametrize("L,B,I,H,layers,bias", [(24, 300, 48, 256, 2, True), (1, 4096, 48, 256, 2, True), (5, 7, 3, 5, 1, True),
                                                 (6, 65, 16, 32, 2, False), (24, 5500, 48, 256, 1, True)])
def test_fused_gru_matches_nn_gru_forward_and_backward(L, B, I, H, layers, bias):
    """The GEMM + HIP-gate-pass GRU against torch.nn.GRU (MIOpen) with the same parameters: outputs, final state and every
    gradient (parameters, input, initial state); both 16-byte-lane and scalar layouts (H = 5), with and without biases."""
    from cusrl_amd import _native

    plain, fused = _twin_grus(I, H, layers, bias)
    x = torch.randn(L, B, I, device=DEV)
    h0 = torch.randn(B, layers * H, device=DEV) * 0.5
    w_out, w_last = torch.randn(L, B, H, device=DEV), torch.randn(layers, B, H, device=DEV)
    results = []
    before = dict(_native.launch_counts)
    for module in (plain, fused):
        xi, hi = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        if module is fused:
            out, last = module(xi, hi)
            last = last.reshape(B, layers, H).transpose(0, 1)
        else:
            out, last = module(xi, hi.reshape(B, layers, H).transpose(0, 1).contiguous())
        ((out * w_out).sum() + (last * w_last).sum()).backward()
        results.append([out.detach(), last.detach(), xi.grad, hi.grad] + [p.grad for p in module.parameters()])
    assert _native.launch_counts["cusrl_gru_gates_fwd"] - before.get("cusrl_gru_gates_fwd", 0) == L * layers
    backward = sum(_native.launch_counts.get(k, 0) - before.get(k, 0) for k in ("cusrl_gru_gates_bwd", "cusrl_gru_gates_bwd_bias"))
    assert backward == L * layers  # (with biases and a layout that tiles a block: the pass that also leaves the bias-gradient sums)
    names = ["out", "h_n", "d_x", "d_h0"] + [n for n, _ in plain.named_parameters()]
    for name, want, got in zip(names, *results):
        scale = float(want.abs().max()) + 1e-6
        assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0) + 1e-4 * scale * (name not in ("out", "h_n")), \
            (name, float((got - want).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("L,B,H", [(9, 40, 32), (24, 700, 256), (4, 3, 5)])
def test_fused_gru_with_lengths_matches_the_packed_sequence(L, B, H):
    """``lengths`` on the device = what nn.GRU returns for the PackedSequence of the same batch: padded outputs zero, the
    state of every sequence taken at its own last step, gradients included — and the oracle agrees."""
    import oracle

    plain, fused = _twin_grus(6, H, 2, True)
    gen = torch.Generator().manual_seed(L * B)
    lengths = torch.randint(1, L + 1, (B,), generator=gen)
    lengths[0] = L
    x = torch.randn(L, B, 6, device=DEV)
    h0 = torch.randn(B, 2 * H, device=DEV) * 0.3
    w_out, w_last = torch.randn(L, B, H, device=DEV), torch.randn(2, B, H, device=DEV)
    xi = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xi, lengths, enforce_sorted=False)
    packed_out, want_last = plain(packed, h0.reshape(B, 2, H).transpose(0, 1).contiguous())
    want_out, _ = torch.nn.utils.rnn.pad_packed_sequence(packed_out, total_length=L)
    ((want_out * w_out).sum() + (want_last * w_last).sum()).backward()
    want = [want_out.detach(), want_last.detach(), xi.grad] + [p.grad for p in plain.parameters()]
    xj = x.clone().requires_grad_(True)
    out, last = fused(xj, h0, lengths=lengths.to(DEV))
    last = last.reshape(B, 2, H).transpose(0, 1)
    ((out * w_out).sum() + (last * w_last).sum()).backward()
    got = [out.detach(), last.detach(), xj.grad] + [p.grad for p in fused.parameters()]
    for i, (a, b) in enumerate(zip(want, got)):
        scale = max(float(a.abs().max()), 1.0)
        assert float((a - b).abs().max()) <= (2e-5 if i < 2 else 2e-4) * scale, (i, float((a - b).abs().max()), scale)
    mask = torch.arange(L)[:, None] >= lengths[None]
    assert not out.detach().cpu()[mask].any()
    if B <= 40:
        weights = [tuple(getattr(plain, f"{n}_l{layer}").detach().cpu().numpy() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                   for layer in (0, 1)]
        ref_out, ref_last = oracle.gru_sequence(x.cpu().numpy(), h0.reshape(B, 2, H).transpose(0, 1).cpu().numpy(), weights,
                                                lengths.numpy())
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(last.detach().cpu().numpy(), ref_last, rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_fused_gru_refuses_a_second_backward_and_falls_back_where_it_must():
    from cusrl_amd.nn.gru import gru_supported

    plain, fused = _twin_grus(4, 8, 1, True)
    x = torch.randn(3, 5, 4, device=DEV, requires_grad=True)
    out, _ = fused(x)
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="ONE backward pass"):
        out.sum().backward()
    assert not gru_supported(fused, x.double()) and not gru_supported(fused, x.cpu())
    packed = torch.nn.utils.rnn.pack_padded_sequence(x.detach(), torch.tensor([3, 2, 2, 1, 1]))
    assert not gru_supported(fused, packed)
    packed_out, _ = fused(packed)  # a PackedSequence still goes to MIOpen
    assert isinstance(packed_out, torch.nn.utils.rnn.PackedSequence)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not gru_supported(fused, x)


def _twin_lstms(I, H, layers, bias):
    from cusrl_amd.nn.rnn import _Lstm

    torch.manual_seed(I * 1000 + H + 1)
    plain = torch.nn.LSTM(I, H, layers, bias=bias).to(DEV)  # MIOpen
    fused = _Lstm(I, H, layers, bias=bias).to(DEV)
    fused.load_state_dict(plain.state_dict())
    return plain, fused


@pytest.mark.gpu
@pytest.mark.parametrize("L,B,I,H,layers,bias,with_lengths", [(24, 300, 48, 256, 2, True, False), (1, 4096, 48, 256, 2, True, False),
                                                              (5, 7, 3, 5, 1, True, True), (6, 65, 16, 32, 2, False, False),
                                                              (24, 700, 48, 256, 2, True, True)])
def test_fused_lstm_matches_nn_lstm_forward_and_backward(L, B, I, H, layers, bias, with_lengths):
    """The GEMM + HIP-gate-pass LSTM against torch.nn.LSTM (MIOpen): outputs, both final states and every gradient, in the
    padded form and (``with_lengths``) against the PackedSequence of the same batch; oracle.lstm_sequence on the small one."""
    import oracle
    from cusrl_amd import _native

    plain, fused = _twin_lstms(I, H, layers, bias)
    x = torch.randn(L, B, I, device=DEV)
    h0, c0 = torch.randn(B, layers * H, device=DEV) * 0.5, torch.randn(B, layers * H, device=DEV) * 0.5
    lengths = None
    if with_lengths:
        lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(B))
        lengths[0] = L
    w_out, w_h, w_c = torch.randn(L, B, H, device=DEV), torch.randn(layers, B, H, device=DEV), torch.randn(layers, B, H, device=DEV)
    to_layers = lambda m: m.reshape(B, layers, H).transpose(0, 1).contiguous()  # noqa: E731
    before = dict(_native.launch_counts)
    results = []
    for module in (plain, fused):
        xi, hi, ci = (v.clone().requires_grad_(True) for v in (x, h0, c0))
        if module is fused:
            out, memory = module(xi, {"hidden": hi, "cell": ci}, lengths=None if lengths is None else lengths.to(DEV))
            hn, cn = to_layers(memory["hidden"]), to_layers(memory["cell"])
        elif lengths is None:
            out, (hn, cn) = module(xi, (to_layers(hi), to_layers(ci)))
        else:
            packed = torch.nn.utils.rnn.pack_padded_sequence(xi, lengths, enforce_sorted=False)
            packed_out, (hn, cn) = module(packed, (to_layers(hi), to_layers(ci)))
            out, _ = torch.nn.utils.rnn.pad_packed_sequence(packed_out, total_length=L)
        ((out * w_out).sum() + (hn * w_h).sum() + (cn * w_c).sum()).backward()
        results.append([out.detach(), hn.detach(), cn.detach(), xi.grad, hi.grad, ci.grad] + [p.grad for p in module.parameters()])
    assert _native.launch_counts["cusrl_lstm_gates_fwd"] - before.get("cusrl_lstm_gates_fwd", 0) == L * layers
    assert _native.launch_counts["cusrl_lstm_gates_bwd"] - before.get("cusrl_lstm_gates_bwd", 0) == L * layers
    names = ["out", "h_n", "c_n", "d_x", "d_h0", "d_c0"] + [n for n, _ in plain.named_parameters()]
    for i, (name, want, got) in enumerate(zip(names, *results)):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got - want).abs().max()) <= (2e-5 if i < 3 else 2e-4) * scale, (name, float((got - want).abs().max()), scale)
    if B <= 7:
        weights = [tuple(getattr(plain, f"{n}_l{layer}").detach().cpu().numpy() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                   for layer in range(layers)]
        ref_out, (ref_h, ref_c) = oracle.lstm_sequence(x.cpu().numpy(), (to_layers(h0).cpu().numpy(), to_layers(c0).cpu().numpy()),
                                                       weights, None if lengths is None else lengths.numpy())
        np.testing.assert_allclose(results[1][0].cpu().numpy(), ref_out, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(results[1][1].cpu().numpy(), ref_h, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(results[1][2].cpu().numpy(), ref_c, rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("nonlinearity", ["tanh", "relu"])
@pytest.mark.parametrize("L,B,I,H,layers,bias,with_lengths", [(24, 300, 48, 256, 2, True, False), (5, 7, 3, 5, 1, True, True),
                                                              (6, 65, 16, 32, 2, False, True)])
def test_fused_rnn_matches_nn_rnn_forward_and_backward(nonlinearity, L, B, I, H, layers, bias, with_lengths):
    """The GEMM + HIP-cell vanilla RNN against torch.nn.RNN (MIOpen), tanh and relu, padded and packed forms."""
    import oracle
    from cusrl_amd import _native
    from cusrl_amd.nn.rnn import _VanillaRnn

    torch.manual_seed(L * B + H)
    plain = torch.nn.RNN(I, H, layers, nonlinearity=nonlinearity, bias=bias).to(DEV)
    fused = _VanillaRnn(I, H, layers, nonlinearity=nonlinearity, bias=bias).to(DEV)
    fused.load_state_dict(plain.state_dict())
    x, h0 = torch.randn(L, B, I, device=DEV), torch.randn(B, layers * H, device=DEV) * 0.5
    lengths = None
    if with_lengths:
        lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(B))
        lengths[0] = L
    w_out, w_h = torch.randn(L, B, H, device=DEV), torch.randn(layers, B, H, device=DEV)
    to_layers = lambda m: m.reshape(B, layers, H).transpose(0, 1).contiguous()  # noqa: E731
    before = dict(_native.launch_counts)
    results = []
    for module in (plain, fused):
        xi, hi = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        if module is fused:
            out, memory = module(xi, hi, lengths=None if lengths is None else lengths.to(DEV))
            hn = to_layers(memory)
        elif lengths is None:
            out, hn = module(xi, to_layers(hi))
        else:
            packed_out, hn = module(torch.nn.utils.rnn.pack_padded_sequence(xi, lengths, enforce_sorted=False), to_layers(hi))
            out, _ = torch.nn.utils.rnn.pad_packed_sequence(packed_out, total_length=L)
        ((out * w_out).sum() + (hn * w_h).sum()).backward()
        results.append([out.detach(), hn.detach(), xi.grad, hi.grad] + [p.grad for p in module.parameters()])
    assert _native.launch_counts["cusrl_rnn_cell_fwd"] - before.get("cusrl_rnn_cell_fwd", 0) == L * layers
    assert _native.launch_counts["cusrl_rnn_cell_bwd"] - before.get("cusrl_rnn_cell_bwd", 0) == L * layers
    for i, (want, got) in enumerate(zip(*results)):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got - want).abs().max()) <= (2e-5 if i < 2 else 2e-4) * scale, (i, float((got - want).abs().max()), scale)
    if B <= 7:
        weights = [tuple(getattr(plain, f"{n}_l{layer}").detach().cpu().numpy() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                   for layer in range(layers)]
        ref_out, ref_h = oracle.rnn_sequence(x.cpu().numpy(), to_layers(h0).cpu().numpy(), weights,
                                             None if lengths is None else lengths.numpy(), relu=nonlinearity == "relu")
        np.testing.assert_allclose(results[1][0].cpu().numpy(), ref_out, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(results[1][1].cpu().numpy(), ref_h, rtol=1e-5, atol=2e-6)


def _gru_float64(x, done, memory0, params, prefix, layers, hidden):
    """A stacked GRU over ``x [T, n, C]`` in float64 with torch.nn.GRU's cell equations (gate rows r, z, n), one env column at a
    time step: the memory ``[n, layers * hidden]`` of an env restarts from zero behind every step at which its episode ended
    (cusrl/nn/module/rnn.py:206-253: the done-split layout gives every later segment a zero memory).  Returns the top layer's
    outputs ``[T, n, hidden]``."""
    T, n = x.shape[:2]
    state = [memory0[:, k * hidden:(k + 1) * hidden] for k in range(layers)]
    outputs = []
    for t in range(T):
        below = x[t]
        for k in range(layers):
            w_ih, w_hh = params[f"{prefix}.weight_ih_l{k}"], params[f"{prefix}.weight_hh_l{k}"]
            gi = below @ w_ih.t() + params[f"{prefix}.bias_ih_l{k}"]
            gh = state[k] @ w_hh.t() + params[f"{prefix}.bias_hh_l{k}"]
            r = torch.sigmoid(gi[:, :hidden] + gh[:, :hidden])
            z = torch.sigmoid(gi[:, hidden:2 * hidden] + gh[:, hidden:2 * hidden])
            cand = torch.tanh(gi[:, 2 * hidden:] + r * gh[:, 2 * hidden:])
            below = state[k] = (1 - z) * cand + z * state[k]
        outputs.append(below)
        keep = (~done[t]).to(x.dtype)  # [n, 1]
        state = [s * keep for s in state]
    return torch.stack(outputs)


@pytest.mark.gpu
def test_recurrent_minibatch_steps_match_float64_autograd(gradient_parity):
    """The BPTT minibatch step of a GRU agent (BASELINE config 4's path: temporal minibatches, done-split sequences, the fused GRU
    cores, the one-launch objective, the flat gradient assembly) against an INDEPENDENT float64 evaluation, on every optimizer
    step of 10 iterations (>= 32 steps): the flat gradient buffer behind each step is d(value + surrogate + entropy loss) /
    d(parameters) at the parameters the step started from, recomputed here with a hand-written float64 GRU (torch.nn.GRU's
    equations, memories restarting at episode ends), the formulas of cusrl/hook/on_policy/ppo.py:10-18, value.py:121-137,
    nn/module/distribution.py:207-213 and plain autograd.  Weights: 1e-5 of the tensor's largest entry; biases / the std vector:
    1e-5 of the summed magnitudes of each element's own per-row terms.  (Recurrent minibatch steps are not captured — dynamic
    sequence counts — so this is a correctness soak of the eager path, the recurrent twin of tests/test_captured_step_soak.py.)"""
    import math

    from cusrl_amd.template.actor_critic import ActorCritic

    cusrl.config.set_device(DEV)
    cusrl.set_global_seed(9)
    N, T, H, actor_layers, critic_layers = 64, 8, 32, 2, 1
    factory = cusrl.preset.RecurrentPpoAgentFactory(
        rnn_type="GRU", actor_num_layers=actor_layers, actor_hidden_size=H, critic_num_layers=critic_layers, critic_hidden_size=H,
        num_steps_per_update=T, sampler_epochs=2, sampler_mini_batches=2, optimizer_kwargs={"capturable": True, "fused": True})
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=10, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, factory, num_iterations=10, verbose=False)
    agent = trainer.agent
    flat = agent.flat_gradients
    assert flat is not None and agent.flat_optimizer is not None
    names = {id(p): name for name, p in agent.named_parameters()}
    windows = [(names[id(p)], offset, p.numel()) for p, offset in zip(flat.params, flat.offsets)]
    record = {"steps": 0, "worst": {}, "segments": 0, "near_clip": 0}
    original = ActorCritic._train_step

    def train_step(self, metadata, batch):
        assert metadata["temporal"] is True
        before = {name: p.detach().double().clone().requires_grad_(True) for name, p in self.named_parameters()}
        f64 = lambda key: batch[key].double()  # noqa: E731
        obs, action, old_logp, advantage, ret = f64("observation"), f64("action"), f64("action_logp"), f64("advantage"), f64("return")
        done = batch["done"].clone()
        actor_memory, critic_memory = batch["actor_memory"][0].double(), batch["critic_memory"][0].double()
        record["segments"] += int(done[:-1].sum())
        original(self, metadata, batch)
        torch.cuda.synchronize()
        mine = flat.buffer.clone()
        summed = {}
        latent = _gru_float64(obs, done, actor_memory, before, "actor.backbone.rnn", actor_layers, H)
        mean = summed["actor.distribution.mean_head.bias"] = (
            latent @ before["actor.distribution.mean_head.weight"].t() + before["actor.distribution.mean_head.bias"])
        std = summed["actor.distribution.std.param"] = before["actor.distribution.std.param"].expand_as(mean)
        logp = (-((action - mean) ** 2) / (2 * std**2) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1, keepdim=True)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + std.log()).sum(-1, keepdim=True)
        ratio = (logp - old_logp).exp()
        hooks = self.hook
        surrogate, value_hook, entropy_hook = hooks["ppo_surrogate_loss"], hooks["value_loss"], hooks["entropy_loss"]
        lo = float(torch.tensor(1.0 - surrogate.clip_ratio, dtype=torch.float32))
        hi = float(torch.tensor(1.0 + surrogate.clip_ratio, dtype=torch.float32))
        loss = -torch.min(advantage * ratio, advantage * ratio.clamp(lo, hi)).mean() * surrogate.weight
        value_latent = _gru_float64(obs, done, critic_memory, before, "critic.backbone.rnn", critic_layers, H)
        value = summed["critic.value_head.bias"] = value_latent @ before["critic.value_head.weight"].t() + before["critic.value_head.bias"]
        loss = loss + (value - ret).square().mean() * value_hook.weight - entropy.mean() * entropy_hook.weight
        wanted = list(before)
        grads = torch.autograd.grad(loss, [before[name] for name in wanted] + list(summed.values()), allow_unused=True)
        reference = {name: (g if g is not None else torch.zeros_like(before[name])) for name, g in zip(wanted, grads)}
        magnitudes = {name: terms.abs().sum((0, 1)) for name, terms in zip(summed, grads[len(wanted):])}
        margin = float(torch.minimum((ratio - lo).abs(), (ratio - hi).abs()).min().detach())
        record["near_clip"] += margin < 1e-6
        for name, offset, numel in windows:
            got, want = mine[offset:offset + numel].double(), reference[name].reshape(-1)
            if name in magnitudes:
                error = float(((got - want).abs() / magnitudes[name].reshape(-1).clamp_min(1e-30)).max())
            else:
                error = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
            bound = 1e-5 if margin >= 1e-6 else 0.1  # (a ratio within fp32 noise of a clip bound may fall on either side)
            assert error <= bound, f"step {record['steps']}: {name} off by {error:.3e}"
            if margin >= 1e-6:
                record["worst"][name] = max(record["worst"].get(name, 0.0), error)
        record["steps"] += 1

    ActorCritic._train_step = train_step
    try:
        trainer.run_training_loop()
    finally:
        ActorCritic._train_step = original
    assert record["steps"] >= 32, record["steps"]
    assert record["segments"] > 0  # episodes ended inside the minibatches: the memory restarts were exercised
    assert record["near_clip"] <= 2
    for name, error in record["worst"].items():
        gradient_parity(f"recurrent_step_soak[{name}]", [1.0 + error], [1.0], 1e-5)
    print(f"recurrent step soak: {record['steps']} optimizer steps, {record['segments']} episode ends inside minibatches, worst "
          f"error {max(record['worst'].values()):.2e}")
