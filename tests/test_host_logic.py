"""CPU-side checks (no GPU, no compute through the HIP library): the C ABI loads and exports every declared symbol,
host-side validation mirrors the reference's error behaviour, the plugin surface composes like the reference's, and the
hot path refuses to run anywhere but on the GPU."""

import os
import re
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cusrl_amd as cusrl
from cusrl_amd import _native

ROOT = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_symbol_declared_in_the_header():
    header = (ROOT / "include" / "cusrl_hip.h").read_text()
    declared = set(re.findall(r"\b(cusrl_[a-z0-9_]+)\s*\(", header))
    declared -= {"cusrl_field_t"}
    lib = _native.lib()
    for symbol in declared:
        assert hasattr(lib, symbol), f"{symbol} declared in include/cusrl_hip.h but not exported"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    assert lib.cusrl_abi_version() == _native.ABI_VERSION
    assert b"invalid" in lib.cusrl_error_string(-1) and lib.cusrl_error_string(0) == b"success"


def test_size_helpers_and_argument_validation_without_a_gpu():
    lib = _native.lib()
    assert lib.cusrl_flag_blocks(0) == 0 and lib.cusrl_flag_blocks(1) == 1 and lib.cusrl_flag_blocks(4097) == 2
    assert lib.cusrl_gae_num_partials(24, 4096, 1) == 64  # one row per 64-column wave block (small-rollout shape)
    # blocks of one loss launch: 252 rows per block at A = 12 (21 rows x 4 waves x 3 rounds), 256 for A = 16, 128 for
    # A = 32, 256 for the categorical / row-wise forms; the workspace bound covers the smallest of them
    assert lib.cusrl_ppo_loss_blocks(24576, 12) == 98 and lib.cusrl_ppo_loss_blocks(24576, 16) == 96
    assert lib.cusrl_ppo_loss_blocks(24576, 32) == 192 and lib.cusrl_ppo_loss_blocks(24576, 0) == 96
    assert lib.cusrl_ppo_loss_blocks(24576, 7) == 96 and lib.cusrl_ppo_loss_blocks(0, 12) == 0
    assert lib.cusrl_ppo_loss_num_partials(24576) == 192 and lib.cusrl_ppo_loss_std_partial_rows(24576) == 192 + 2
    assert lib.cusrl_col_stats_num_partials(98304, 1) == 24
    # invalid arguments are rejected on the host before any launch
    assert lib.cusrl_gae(None, None, None, None, None, None, None, 2, 2, 1, 0.9, 0.9, -1.0, None) == -1
    assert lib.cusrl_buffer_push(None, 99, 0, 1, None) == -1
    table = (_native.Field * 30)()
    assert lib.cusrl_buffer_push(table, 30, 0, 1, None) == -2


def test_every_negative_return_code_of_the_c_abi_is_reachable_without_a_launch():
    """CUSRL_E_INVALID (-1), CUSRL_E_TOO_MANY (-2), CUSRL_E_UNSUPPORTED (-3), CUSRL_E_COMM (-4): the argument checks sit on
    the host in front of every launch, so they can be exercised with never-dereferenced placeholder addresses."""
    lib = _native.lib()
    p = 0x1000  # a non-null, 16-byte aligned placeholder: these calls must return before touching it
    odd = p + 4
    # cusrl_gae: more value channels than one block reduces
    assert lib.cusrl_gae(p, p, p, p, p, p, None, 4, 8, 257, 0.99, 0.95, -1.0, None) == -3
    assert lib.cusrl_gae(p, p, p, p, p, p, None, -1, 8, 1, 0.99, 0.95, -1.0, None) == -1
    # cusrl_gather_rows: too many leaves, missing indices, bad geometry
    fields = (_native.Field * 30)()
    for f in fields:
        f.src, f.dst, f.row_bytes = p, p, 4
    assert lib.cusrl_gather_rows(fields, 30, p, 8, 2, 4, 0, None) == -2
    assert lib.cusrl_gather_rows(fields, 3, None, 8, 2, 4, 0, None) == -1
    assert lib.cusrl_gather_rows(fields, 3, p, 8, 0, 4, 0, None) == -1
    fields[1].src = None
    assert lib.cusrl_gather_rows(fields, 3, p, 8, 2, 4, 0, None) == -1
    # cusrl_ppo_loss_fwd_bwd: std vector (std_rows = 1) needs the 16-byte chunk layout (A % 4 == 0, aligned pointers)
    loss = lambda **kw: lib.cusrl_ppo_loss_fwd_bwd(  # noqa: E731
        p, p, p, p, kw.get("std", p), p, p, kw.get("old_value", None), kw.get("B", 64), kw.get("A", 12), 1, 0.2,
        kw.get("value_clip", -1.0), 1.0, 0.5, 0.01, p, None, None, None, None, p, p, p, p, kw.get("std_rows", 64),
        kw.get("std_partials", p), kw.get("flags", 0), None)
    assert loss(std_rows=1, A=7) == -3                      # A not a multiple of 4
    assert loss(std_rows=1, std=odd) == -3                  # misaligned std vector
    assert loss(std_rows=1, std_partials=None) == -1        # d_std wanted but no workspace for its column sums
    assert loss(std_rows=3) == -1                           # std rows must be B or 1
    assert loss(value_clip=0.2) == -1                       # clipped form without the old value
    assert loss(B=0) == -1
    assert loss(std_rows=1, std_partials=None, flags=1) == -1  # CUSRL_LOSS_DEFER with a std vector leaves block rows there
    # categorical objective, record tables, statistics, sequence layout, scatter, window indices
    assert lib.cusrl_ppo_loss_categorical_fwd_bwd(p, p, p, None, p, p, None, 8, 3, 1, 0.2, -1.0, 1.0, 0.5, 0.0, p, None, None, None,
                                                  None, p, p, p, 0, None) == -1
    assert lib.cusrl_normalize_from_partials(p, p, 4, 10, 1e-8, 10, 300, p, p, None) == -3
    assert lib.cusrl_stats_finalize(None, 4, 1, 10, p, p, None) == -1
    assert lib.cusrl_sequence_count(p, 0, 4, p, p, p, None) == -1
    assert lib.cusrl_sequence_layout(p, 4, 4, p, p, 0, p, p, p, None, None, None) == -1
    assert lib.cusrl_gather_memory(None, p, p, p, 4, 64, None) == -1
    assert lib.cusrl_scatter_rows(p, None, p, 4, 16, None, None) == -1
    assert lib.cusrl_window_indices(p, p, p, 4, 2, 8, 3, 8, None) == -1   # cursor outside the ring
    assert lib.cusrl_adam_step(p, p, p, p, p, p, 100, 0.9, 0.999, 1e-8, 0.0, 0, 0, None, 0, -1.0, None, None, None, None) == -1
    assert lib.cusrl_adam_step(odd, p, p, p, p, p, 100, 0.9, 0.999, 1e-8, 0.0, 0, 0, None, 0, -1.0, None, None, p, None) == -3
    assert lib.cusrl_categorical_sample_logp(p, None, p, p, 8, 3, None) == -1
    assert lib.cusrl_categorical_sample_logp(p, p, p, p, 8, 0, None) == -1
    assert lib.cusrl_categorical_sample_logp(None, None, None, None, 0, 3, None) == 0   # an empty step is not an error
    # recurrent cores: geometry and missing operands
    assert lib.cusrl_gru_gates_fwd(p, p, None, p, p, None, 0, 4, 0, None) == -1            # H = 0
    assert lib.cusrl_gru_gates_fwd(p, None, None, p, p, None, 0, 4, 8, None) == -1         # no recurrent projection
    assert lib.cusrl_gru_gates_bwd(p, p, None, p, None, None, None, 0, 4, 8, None) == -1   # no state gradient
    assert lib.cusrl_gru_gates_fwd(None, None, None, None, None, None, 3, 0, 8, None) == 0  # an empty batch is not an error
    assert lib.cusrl_lstm_gates_fwd(p, p, None, p, None, p, None, None, 0, 4, 8, None) == -1  # no cell state
    assert lib.cusrl_lstm_gates_bwd(p, p, p, None, p, p, None, -1, 4, 8, None) == -1        # negative step
    assert lib.cusrl_lstm_gates_bwd(None, None, None, None, None, None, None, 0, 0, 8, None) == 0
    assert lib.cusrl_rnn_cell_fwd(p, p, None, None, p, None, 0, 4, 8, 0, None) == -1        # no state
    assert lib.cusrl_rnn_cell_bwd(p, p, None, p, None, 0, 4, 0, 1, None) == -1              # H = 0
    assert lib.cusrl_rnn_cell_bwd(None, None, None, None, None, 0, 0, 8, 1, None) == 0
    assert lib.cusrl_narrow_linear_supported(128, 12) == 1 and lib.cusrl_narrow_linear_supported(100, 12) == 0
    assert lib.cusrl_narrow_linear_supported(128, 17) == 0
    # communicator: argument errors; the text of every code
    assert lib.cusrl_allreduce_mean(p, 4, None, None) in (-1, -4) and lib.cusrl_comm_create(None, 1, 0, None) in (-1, -4)
    for code, word in ((-1, b"invalid"), (-2, b"too many"), (-3, b"not supported"), (-4, b"RCCL")):
        assert word in lib.cusrl_error_string(code)
    from cusrl_amd._native import NativeError, check

    with pytest.raises(NativeError, match="code -3"):
        check(-3, "cusrl_gae")


def test_hot_path_refuses_cpu_tensors():
    from cusrl_amd import ops

    x = torch.zeros(2, 3, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gae(x, x, x, torch.zeros(2, 3, 1, dtype=torch.bool), 0.99, 0.95, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.normalize_(x, torch.zeros(1), torch.ones(1))
    buffer = cusrl.Buffer(capacity=2, parallelism=1, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        buffer.push({"terminated": torch.tensor([[True]])})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cusrl.hook.GeneralizedAdvantageEstimation().pre_update(
            {"reward": x, "value": x, "next_value": x, "done": torch.zeros(2, 3, 1, dtype=torch.bool)})


# ------------------------------------------------------------------------------------------------ Buffer (test_buffer.py)
def test_buffer_setitem_accepts_numpy_arrays():
    buffer = cusrl.Buffer(capacity=3, parallelism=2, device="cpu")
    buffer["observation"] = np.arange(6, dtype=np.float32).reshape(3, 2, 1)
    observation = buffer["observation"]
    assert isinstance(observation, torch.Tensor) and observation.shape == (3, 2, 1) and observation.device.type == "cpu"
    assert "observation" in buffer and list(buffer) == ["observation"] and len(buffer) == 1
    del buffer["observation"]
    with pytest.raises(KeyError):
        del buffer["observation"]


def test_buffer_validation_messages():
    buffer = cusrl.Buffer(capacity=3, parallelism=2, device="cpu")
    with pytest.raises(ValueError, match=r"\[parallelism, \.\.\.\]"):
        buffer.push({"observation": torch.tensor([1.0])})
    with pytest.raises(ValueError, match="Parallelism mismatch"):
        buffer.push({"observation": torch.zeros(3, 1)})
    with pytest.raises(ValueError, match=r"\[capacity, parallelism, \.\.\.\]"):
        buffer["observation"] = torch.tensor([1.0, 2.0, 3.0])
    with pytest.raises(ValueError, match="Capacity mismatch"):
        buffer["observation"] = torch.zeros(2, 2, 1)
    buffer["nested"] = {"a": torch.zeros(3, 2, 1), "b": (torch.zeros(3, 2, 2),)}
    assert set(buffer.storage) == {"nested.a", "nested.b.0"}
    with pytest.raises(ValueError, match="Schema mismatch"):
        buffer["nested"] = {"a": torch.zeros(3, 2, 1)}
    buffer.resize(5)
    assert buffer.capacity == 5 and not buffer.storage and buffer.cursor == 0 and not buffer.full


# ------------------------------------------------------------------------------------------------ samplers
def test_mini_batch_sampler_validation():
    with pytest.raises(ValueError, match="'num_epochs' must be positive"):
        cusrl.MiniBatchSampler(num_epochs=0)
    with pytest.raises(ValueError, match="'num_mini_batches' must be positive"):
        cusrl.MiniBatchSampler(num_mini_batches=0)
    with pytest.raises(ValueError, match="'num_mini_batches' values must be positive"):
        cusrl.MiniBatchSampler(num_epochs=2, num_mini_batches=[1, 0])
    with pytest.raises(ValueError, match="length"):
        cusrl.MiniBatchSampler(num_epochs=2, num_mini_batches=[1])
    buffer = cusrl.Buffer(capacity=2, parallelism=1, device="cpu")
    buffer["observation"] = torch.zeros(2, 1, 1)
    with pytest.raises(RuntimeError, match="full buffer"):
        next(iter(cusrl.MiniBatchSampler()(buffer)))
    metadata, batch = next(iter(cusrl.Sampler()(buffer)))
    assert metadata == {} and batch["observation"] is buffer.storage["observation"]


# ------------------------------------------------------------------------------------------------ hooks
@pytest.mark.parametrize("kwargs", [{"gamma": -0.1}, {"gamma": 1.0}, {"lamda": -0.1}, {"lamda": 1.1}, {"lamda_value": 1.1}])
def test_gae_validates_discount_parameters(kwargs):
    with pytest.raises(ValueError):
        cusrl.hook.GeneralizedAdvantageEstimation(**kwargs)


@pytest.mark.parametrize("factory", [
    lambda: cusrl.hook.PpoSurrogateLoss(clip_ratio=0.0),
    lambda: cusrl.hook.PpoSurrogateLoss(weight=-1.0),
    lambda: cusrl.hook.EntropyLoss(weight=-1.0),
    lambda: cusrl.hook.ValueLoss(weight=0.0),
    lambda: cusrl.hook.ValueLoss(loss_clip=0.0),
    lambda: cusrl.hook.GradientClipping(max_grad_norm=-1.0),
    lambda: cusrl.hook.GradientClipping(groups={"": 1.0}),
    lambda: cusrl.hook.AdvantageReduction(reduction="max"),
])
def test_hooks_validate_configuration(factory):
    with pytest.raises(ValueError):
        factory()


def test_hook_names_mutables_and_composite():
    hook = cusrl.hook.GeneralizedAdvantageEstimation()
    assert hook.name == "generalized_advantage_estimation" and hook.training_only and hook.active
    hook.update_attribute("gamma", 0.5)
    assert hook.gamma == 0.5
    with pytest.raises(ValueError, match="not mutable"):
        hook.update_attribute("recompute", True)
    assert cusrl.hook.PpoSurrogateLoss().name == "ppo_surrogate_loss"
    from cusrl_amd.template.hook import HookComposite

    with pytest.raises(RuntimeError, match="already exists"):
        HookComposite([cusrl.hook.EntropyLoss(), cusrl.hook.EntropyLoss()])
    with pytest.raises(TypeError):
        HookComposite([object()])
    composite = HookComposite([cusrl.hook.EntropyLoss(), cusrl.hook.ValueLoss().active_(False)])
    composite.pre_init(SimpleNamespace(inference_mode=True))
    assert [h.name for h in composite.active_hooks()] == []  # training-only skipped in inference, inactive skipped
    composite.agent.inference_mode = False
    assert [h.name for h in composite.active_hooks()] == ["entropy_loss"]
    assert composite["value_loss"].weight == 0.5


def test_advantage_reduction_is_plain_tensor_math():
    hook = cusrl.hook.AdvantageReduction(reduction="sum", weight=(1.0, 2.0))
    hook.agent = SimpleNamespace(to_tensor=lambda value: torch.as_tensor(value, dtype=torch.float32))
    hook.init()
    batch = {"advantage": torch.tensor([[1.0, 2.0], [3.0, 4.0]])}
    hook.objective({}, batch)
    assert torch.allclose(batch["advantage"], torch.tensor([[5.0], [11.0]]))
    hook.update_attribute("weight", (0.5, 0.5))
    batch = {"advantage": torch.tensor([[2.0, 6.0]])}
    hook.objective({}, batch)
    assert torch.allclose(batch["advantage"], torch.tensor([[4.0]]))


def test_ppo_preset_hook_order_and_register_hook():
    factory = cusrl.preset.PpoAgentFactory().to_underlying()
    assert [h.name for h in factory.hooks] == [
        "module_initialization", "value_computation", "generalized_advantage_estimation", "advantage_normalization",
        "value_loss", "on_policy_preparation", "ppo_surrogate_loss", "entropy_loss", "gradient_clipping",
        "on_policy_statistics",
    ]

    class Probe(cusrl.Hook):
        pass

    factory.register_hook(Probe().name_("a"), before="value_loss")
    factory.register_hook(Probe().name_("b"), after="entropy_loss")
    factory.register_hook(Probe().name_("c"))
    names = [h.name for h in factory.hooks]
    assert names.index("a") == names.index("value_loss") - 1 and names.index("b") == names.index("entropy_loss") + 1
    assert names[-1] == "c" and factory.hooks.gradient_clipping.max_grad_norm == 1.0
    with pytest.raises(ValueError, match="Only one of"):
        factory.register_hook(Probe(), index=0, before="a")
    with pytest.raises(ValueError, match="No hook named"):
        factory.get_hook("missing")
    assert cusrl.preset.PpoAgentFactory().sampler_epochs == 5 and cusrl.preset.PpoAgentFactory().num_steps_per_update == 24


def test_agent_builds_on_cpu_with_reference_parameter_layout():
    spec = cusrl.EnvironmentSpec(48, 12, num_instances=4, device="cpu")
    agent = cusrl.preset.PpoAgentFactory(device="cpu").to_underlying()(spec)
    names = [n for n, _ in agent.named_parameters()]
    assert names[:3] == ["actor.backbone.layers.0.weight", "actor.backbone.layers.0.bias", "actor.backbone.layers.2.weight"]
    assert names[-2:] == ["critic.value_head.weight", "critic.value_head.bias"] and len(names) == 13
    assert sum(p.numel() for p in agent.parameters()) == 92569  # 13 tensors, 92 569 parameters
    assert agent.flat_gradients.packed().numel() == 92569 and agent.flat_gradients.intact()
    assert agent.flat_gradients.buffer.numel() == 92572 and all(o % 4 == 0 for o in agent.flat_gradients.offsets)  # 16-byte windows
    groups = agent.optimizer.param_groups
    assert groups[0]["param_names"] == names
    # orthogonal init: zero biases, small policy head
    assert not agent.actor.backbone.layers[0].bias.any()
    assert agent.actor.distribution.mean_head.weight.norm() < agent.actor.backbone.layers[2].weight.norm()
    with pytest.raises(TypeError, match="'terminated' must have dtype bool"):
        agent.act(torch.zeros(4, 48))
        agent.step(torch.zeros(4, 48), torch.zeros(4, 1), torch.zeros(4, 1), torch.zeros(4, 1, dtype=torch.bool))


def test_optimizer_factory_groups_and_names():
    model = torch.nn.ModuleDict({"actor": torch.nn.Linear(2, 2), "critic": torch.nn.Linear(2, 1)})
    factory = cusrl.OptimizerFactory("Adam", defaults={"lr": 1e-3}, group_overrides=[("critic", {"lr": 1e-2})])
    optimizer = factory(model.named_parameters())
    by_lr = {g["lr"]: g["param_names"] for g in optimizer.param_groups}
    assert by_lr[1e-2] == ["critic.weight", "critic.bias"] and by_lr[1e-3] == ["actor.weight", "actor.bias"]
    with pytest.raises(ValueError, match="No trainable parameters"):
        cusrl.OptimizerFactory("Adam", param_filter="nothing")(model.named_parameters())
    with pytest.raises(ValueError, match="not assigned"):
        cusrl.template.build_optimizer(cusrl.OptimizerFactory("Adam", param_filter="actor"), model.named_parameters())


# ------------------------------------------------------------------------------------------------ utils
def test_nest_roundtrip():
    from cusrl_amd.utils.nest import flatten_nested, get_schema, iterate_nested, map_nested, reconstruct_nested

    data = {"a": 1, "b": {"c": [10, 20], "d": 30}, "e": (40,)}
    assert list(iterate_nested(data)) == [("a", 1), ("b.c.0", 10), ("b.c.1", 20), ("b.d", 30), ("e.0", 40)]
    assert get_schema({"a": 1, "b": {"c": 2}}) == {"a": "a", "b": {"c": "b.c"}}
    assert get_schema([10, 20, {"key": 30}]) == ["0", "1", {"key": "2.key"}]
    assert reconstruct_nested(flatten_nested(data), get_schema(data)) == data
    assert reconstruct_nested({"a": 10, "b.c": 20, "b.d": 30}, {"a": "a", "b": ("b.c", "b.d")}) == {"a": 10, "b": (20, 30)}
    assert map_nested(lambda v: v * 2, data)["b"]["c"] == [20, 40]
    assert dict(iterate_nested({"x": {"y": 1}}, "p")) == {"p.x.y": 1}


def test_metrics_weighted_mean():
    metrics = cusrl.utils.Metrics()
    metrics.record(loss=torch.tensor([1.0, 3.0]))       # mean 2 over 2 samples
    metrics.record({"loss": torch.tensor([7.0, 7.0, 7.0])}, skipped=None)  # mean 7 over 3 samples
    assert metrics.summary("Agent") == {"Agent/loss": pytest.approx(5.0)}
    metrics.clear()
    assert metrics.summary() == {}


def test_metrics_read_staged_device_sums_and_lazy_tensors_in_one_pass():
    """Round 6: what captured graphs accumulate on the device reaches the metrics through ``pending`` sources (snapshot + reset,
    batched per dtype) and ``defer``-red tensors of any length — resolved together with the queued scalars by ONE read."""
    metrics = cusrl.utils.Metrics()
    accumulator = torch.tensor([6.0, 9.0, 0.0, 0.0])   # two tapped metrics summed over 3 replays
    rows = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    calls = []

    class Capture:
        def stage_metrics(self, m):
            calls.append("staged")
            return [(accumulator[:2], accumulator, lambda sums: [m.add_resolved(name, total * 1, 1 * 3) for name, total in zip(("a", "b"), sums)]),
                    (rows, rows, lambda values: m.add_resolved("rows", sum(values), 1))]

    source = Capture()
    metrics.pending(source)
    metrics.pending(source)  # registered once
    metrics.defer(torch.tensor([[2.0, 4.0]]), lambda values: metrics.add_resolved("lazy", sum(values), 2))
    metrics.record(plain=torch.tensor([5.0, 7.0]))
    summary = metrics.summary("Agent")
    assert calls == ["staged"]
    assert summary == {"Agent/plain": pytest.approx(6.0), "Agent/a": pytest.approx(2.0), "Agent/b": pytest.approx(3.0),
                       "Agent/rows": pytest.approx(10.0), "Agent/lazy": pytest.approx(3.0)}
    assert float(accumulator.abs().sum()) == 0.0 and float(rows.abs().sum()) == 0.0  # snapshot taken, sources reset
    metrics.clear()
    assert metrics.summary() == {}


def test_precision_resolves_the_autocast_argument():
    """``Agent(autocast=...)``: False / None = fp32 without autocast, True = fp16 autocast (needs a GradScaler), a dtype or its
    name = autocast to it (cusrl/template/agent.py:101-109)."""
    from cusrl_amd.template.agent import Precision

    assert Precision.resolve(False) == (torch.float32, False) and Precision.resolve(None) == (torch.float32, False)
    assert Precision.resolve(True) == (torch.float16, True) and Precision.resolve(True).needs_grad_scaler
    assert Precision.resolve("bfloat16") == (torch.bfloat16, True) and not Precision.resolve("torch.bf16").needs_grad_scaler
    assert Precision.resolve(torch.float16).needs_grad_scaler and not Precision.resolve("fp32").needs_grad_scaler
    with pytest.raises(ValueError, match="Unknown autocast dtype"):
        Precision.resolve("float8")


def test_objectives_hand_out_the_branch_root_as_a_root_of_its_own():
    """``Objectives.terms()`` (round 6): the fused total, the non-fused terms, and — when the value term was evaluated by its own
    launch on the critic's stream — that term as one more root, marked as the branch (``Roots.branch``)."""
    from cusrl_amd.template.hook import Objectives, Roots

    total, value, extra = (torch.tensor(float(v), requires_grad=True) for v in (1.0, 2.0, 3.0))
    objectives = Objectives(value_loss=None, surrogate_loss=None, entropy_loss=None, extra_loss=extra)
    objectives.total, objectives.fused_keys = total, ("value_loss", "surrogate_loss", "entropy_loss")
    plain = objectives.terms()
    assert isinstance(plain, Roots) and plain.branch is None and [t is x for t, x in zip(plain, (total, extra))] == [True, True]
    objectives.branch_root = (value, "stream")
    roots = objectives.terms()
    assert len(roots) == 3 and roots[-1] is value and roots.branch == (value, "stream")
    unfused = Objectives(a=total, b=None, c=extra)
    assert [t is x for t, x in zip(unfused.terms(), (total, extra))] == [True, True] and len(unfused.terms()) == 2  # None entries skipped


def test_options_of_the_c_abi_without_a_gpu(monkeypatch):
    """``cusrl_set_option`` / ``cusrl_get_option`` are host functions: keys, accepted values, error codes — and the host's one-time
    translation of the A/B scripts' CUSRL_* variables (the library itself reads no environment variable in a launch entry point)."""
    from cusrl_amd import _native

    lib = _native.lib()
    assert lib.cusrl_set_option(b"unknown", 1) == -1 and lib.cusrl_set_option(b"gae_block", 64) == -3
    for key, value in (("gae_policy", 8), ("gae_block", 256), ("loss_policy", 1), ("push_policy", 2), ("colsum_rows", 128),
                       ("head_rows", 96), ("gru_bias_rows", 32)):
        _native.set_option(key, value)
        assert _native.get_option(key) == value
    monkeypatch.setenv("CUSRL_GAE_POLICY", "5")
    monkeypatch.setenv("CUSRL_LOSS_POLICY", "1")
    monkeypatch.setenv("CUSRL_PUSH_POLICY", "bogus")  # (an unknown value meant "the kernel's own rule" to the library, too)
    for key in ("gae_policy", "loss_policy", "push_policy"):
        _native.set_option(key, 0)
    _native._options_from_environment()
    assert (_native.get_option("gae_policy"), _native.get_option("loss_policy"), _native.get_option("push_policy")) == (6, 2, 0)
    for key in _native._ENVIRONMENT_OPTIONS.values():
        _native.set_option(key[0], 0)
    sources = "".join(path.read_text() for path in (ROOT / "cusrl_amd" / "csrc").iterdir())
    assert sources.count("getenv(") == 1 and 'getenv("CUSRL_RCCL_LIBRARY")' in sources


def test_size_helpers_of_the_round_6_entry_points():
    from cusrl_amd import _native

    import ctypes

    lib = _native.lib()
    assert lib.cusrl_value_loss_blocks(24576, 1) == 24 and lib.cusrl_value_loss_blocks(1000, 3) == 3 and lib.cusrl_value_loss_blocks(0, 1) == 0
    assert lib.cusrl_input_layer_supported(48, 256) and lib.cusrl_input_layer_supported(12, 64)
    assert not lib.cusrl_input_layer_supported(64, 256) and not lib.cusrl_input_layer_supported(48, 96)
    assert lib.cusrl_input_layer_row_blocks(24576, 256) == 64 and lib.cusrl_input_layer_row_blocks(8, 256) == 1
    p = ctypes.c_void_p(16)
    assert lib.cusrl_value_loss_fwd_bwd(p, p, None, 8, 1, 0.2, 0.5, p, None, p, 0, None) == -1   # clipped form without the old value
    assert lib.cusrl_input_layer_bwd(p, None, p, 8, 50, 256, p, p, None) == -3                     # K not a multiple of 4
    assert lib.cusrl_ppo_loss_fwd_bwd(p, p, p, p, p, None, None, None, 8, 4, 0, 0.2, -1.0, 1.0, 0.5, 0.0, None, None, None, None, None,
                                      None, None, None, None, 8, None, 0, None) == -1             # D = 0 still needs losses / partials


def test_timer_sections_accumulate():
    timer = cusrl.utils.Timer("cpu")
    with timer.record("agent"):
        pass
    with timer.record("agent"):
        pass
    assert timer["agent"] >= 0.0
    with pytest.raises(RuntimeError, match="has not been started"):
        timer.stop("missing")
    timer.start("x")
    with pytest.raises(RuntimeError, match="already been started"):
        timer.start("x")


def test_distributed_helpers_are_local_no_ops_in_a_single_process():
    from cusrl_amd.utils import distributed

    assert not distributed.enabled() and distributed.world_size() == 1 and distributed.rank() == 0
    mean, var = torch.tensor([1.0]), torch.tensor([2.0])
    assert distributed.reduce_mean_var_(mean, var) == (mean, var)
    assert distributed.gather_stack(mean).shape == (1, 1)
    assert distributed.average_dict({"a": 1.0}) == {"a": 1.0}
    assert distributed.gather_obj("x") == ["x"]
    distributed.barrier()


def test_set_global_seed_is_rank_offset_and_reproducible():
    cusrl.set_global_seed(42)
    a = torch.randperm(8)
    cusrl.set_global_seed(42)
    assert torch.equal(a, torch.randperm(8)) and cusrl.config.seed == 42


@pytest.mark.parametrize("tag", ["adaptive", "adaptive_all_maxkl", "adaptive_warmup", "threshold", "threshold_maxkl"])
def test_kl_driven_lr_schedules_replay_reference_trajectories(tag):
    """Learning rates, recorded lr_scale / update_rejected and roll-backs over a KL sequence, as recorded from the
    reference's AdaptiveLRSchedule / ThresholdLRSchedule (tests/golden/lr_schedule.npz, make_golden.py)."""
    import sys
    from pathlib import Path

    import cusrl_amd

    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from make_golden import SCHEDULE_CASES, ScheduleProbe

    golden = np.load(Path(__file__).resolve().parent / "golden" / "lr_schedule.npz")
    cls_name, kwargs, schedule_first = SCHEDULE_CASES[tag]
    got = ScheduleProbe().run(getattr(cusrl_amd.hook, cls_name)(**kwargs), list(golden["kls"]), schedule_first)
    np.testing.assert_allclose(got, golden[tag], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("tag", ["mini_batch_wise", "mini_batch_wise_warmup"])
def test_mini_batch_wise_lr_schedule_replays_the_reference_trajectory(tag):
    """MiniBatchWiseLRSchedule (lr_schedule.py:242-296): learning rates and recorded lr_scale after every minibatch's
    objective() call, warm-up iterations ignoring the KL, post_update() recording nothing, and post_init() switching on
    OnPolicyPreparation.calculate_kl_divergence — against the trajectory recorded from the reference."""
    import sys
    from pathlib import Path

    import cusrl_amd

    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from make_golden import MINI_BATCH_WISE_CASES, ScheduleProbe

    golden = np.load(Path(__file__).resolve().parent / "golden" / "lr_schedule.npz")
    kls = list(golden["kls"]) + list(golden["kls"][:2])
    hook = cusrl_amd.hook.MiniBatchWiseLRSchedule(**MINI_BATCH_WISE_CASES[tag])
    preparation = cusrl_amd.hook.OnPolicyPreparation()
    assert not preparation.calculate_kl_divergence
    got = ScheduleProbe().run_mini_batch_wise(hook, preparation, kls, 4)
    np.testing.assert_allclose(got, golden[tag], rtol=1e-12, atol=0, equal_nan=True)
    assert hook.eager_phases() == ("objective",) and hook.scale_all_params


def test_an_agent_with_a_host_reading_hook_keeps_its_minibatch_steps_out_of_capture():
    import cusrl_amd
    from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
    from cusrl_amd.template.graphs import eager_phases

    spec = cusrl_amd.EnvironmentSpec(6, 3, num_instances=4, device="cpu")
    factory = cusrl_amd.preset.PpoAgentFactory(device="cpu").to_underlying()
    assert eager_phases(factory(spec)) == set()
    factory.register_hook(cusrl_amd.hook.MiniBatchWiseLRSchedule(), after="on_policy_preparation")
    agent = factory(spec)
    assert eager_phases(agent) == {"objective"}
    assert agent.hook["on_policy_preparation"].calculate_kl_divergence
    order = [type(h).__name__ for h in agent.hook]
    assert order.index("MiniBatchWiseLRSchedule") == order.index("OnPolicyPreparation") + 1
    # the schedule neither reads nor differentiates the policy terms: the fused objective stays available on a GPU
    agent.device = torch.device("cuda")
    assert FusedPpoObjective.eligible(agent.hook)


def test_ppo_preset_includes_the_adaptive_lr_schedule_when_asked():
    import cusrl_amd

    suite = cusrl_amd.preset.ppo_hook_suite(desired_kl_divergence=0.01, max_kl_divergence=0.05)
    assert type(suite[-1]) is cusrl_amd.hook.AdaptiveLRSchedule and suite[-1].max_kl_divergence == 0.05
    assert not any(isinstance(h, cusrl_amd.hook.AdaptiveLRSchedule) for h in cusrl_amd.preset.ppo_hook_suite())


def test_affinity_cpu_list_parsing_and_noop_without_topology(monkeypatch):
    from cusrl_amd.utils import affinity

    assert affinity._parse_cpu_list("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._parse_cpu_list("") == []
    monkeypatch.setattr(affinity, "device_local_cpus", lambda index: [])
    before = os.sched_getaffinity(0)
    assert affinity.pin_host_thread(0) == [] and os.sched_getaffinity(0) == before  # unknown topology: left alone
    cpus = sorted(before)
    if len(cpus) >= 2:
        monkeypatch.setattr(affinity, "device_local_cpus", lambda index: cpus)
        try:
            assert affinity.pin_host_thread(0, cores=1, slot=1) == [cpus[1]] and os.sched_getaffinity(0) == {cpus[1]}
        finally:
            os.sched_setaffinity(0, before)


def test_timer_sampled_sections_scale_to_all_occurrences():
    import time

    import cusrl_amd as cusrl

    timer = cusrl.utils.Timer("cpu")
    for _ in range(16):
        with timer.record("step", every=4):
            time.sleep(0.002)
    with timer.record("step"):  # an always-timed section under the same name adds on top
        time.sleep(0.004)
    assert timer._timed[("step", 4)] == 4 and timer._seen[("step", 4)] == 16
    assert 0.032 * 0.8 + 0.004 <= timer["step"] <= 0.032 * 2.5 + 0.02
    timer.clear()
    assert timer["step"] == 0.0


def test_flat_gradient_assembly_builds_one_piece_per_parameter(monkeypatch):
    """Host logic of FlatGradients.assemble: plain gradients, split-GEMM slabs, deferred column windows, unused
    parameters and a parameter that got both a gradient and slabs (used twice) — checked against a numpy reduction."""
    import torch

    from cusrl_amd import ops
    from cusrl_amd.utils.distributed import FlatGradients

    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2)),
              torch.nn.Parameter(torch.zeros(6)), torch.nn.Parameter(torch.zeros(3))]
    flat = FlatGradients(torch.optim.SGD(params, lr=0.1))
    captured = {}

    def fake_assemble(pieces, buffer, want_sumsq=False):  # numpy restatement of cusrl_assemble_gradients
        captured["pieces"], captured["want_sumsq"] = pieces, want_sumsq
        for src, offset, numel, splits in pieces:
            if isinstance(src, ops.DeferredColumns):
                rows_ = src.partials.reshape(-1)[: src.splits * src.row_stride].view(src.splits, src.row_stride)
                buffer[offset : offset + numel] = rows_[:, src.column : src.column + src.numel].sum(0)
            elif src is None or splits == 0:
                buffer[offset : offset + numel] = 0
            else:
                buffer[offset : offset + numel] = src.reshape(splits, numel).sum(0)

    monkeypatch.setattr(ops, "assemble_gradients", fake_assemble)
    plain = torch.randn(4, 3)
    slabs = torch.randn(7, 5)                      # [S, numel] slabs for params[1]
    rows = torch.randn(9, 20)                      # partial rows; params[2] = window [4, 8), params[3] = window [10, 16)
    twice_grad, twice_slabs = torch.randn(3), torch.randn(2, 3)
    sink = {params[1].data_ptr(): slabs,
            params[2].data_ptr(): ops.DeferredColumns(rows, 9, 20, 4, 4),
            params[4].data_ptr(): twice_slabs}
    flat.buffer.fill_(float("nan"))
    flat.assemble([plain, None, None, None, twice_grad], sink)
    assert not sink and len(captured["pieces"]) == 5
    assert captured["want_sumsq"] and flat.take_sumsq() is None  # single process: asked for; the stand-in returned none
    want = torch.cat([plain.reshape(-1), slabs.sum(0), rows[:, 4:8].sum(0), torch.zeros(6), twice_grad + twice_slabs.sum(0)])
    torch.testing.assert_close(flat.packed(), want)
    assert [piece[3] for piece in captured["pieces"]] == [1, 7, 9, 0, 1]
    with pytest.raises(RuntimeError, match="not optimizer parameters"):
        flat.assemble([plain, None, None, None, twice_grad], {12345: slabs})


def test_flat_gradient_assembly_by_window_for_the_split_backward(monkeypatch):
    """``FlatGradients.assemble(subset=...)`` (the per-network split of the backward, ActorCritic._backward): only the subset's
    windows are written, slabs of parameters outside the subset (the shared loss node hands the std vector's over in both passes)
    are dropped instead of rejected, ``absent`` accumulates over the windows, ``window()`` spans consecutive parameters with
    their alignment padding."""
    import torch

    from cusrl_amd import ops
    from cusrl_amd.utils.distributed import FlatGradients

    torch.manual_seed(1)
    params = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2)),
              torch.nn.Parameter(torch.zeros(6))]
    flat = FlatGradients(torch.optim.SGD(params, lr=0.1))
    assert flat.offsets == [0, 12, 20, 24] and flat.buffer.numel() == 32  # 16-byte windows: 12, 5 -> 8, 4, 6 -> 8

    def fake_assemble(pieces, buffer, want_sumsq=False):
        assert not want_sumsq  # a window's squared norm is not the buffer's
        for src, offset, numel, splits in pieces:
            buffer[offset : offset + numel] = 0 if src is None or splits == 0 else src.reshape(splits, numel).sum(0)

    monkeypatch.setattr(ops, "assemble_gradients", fake_assemble)
    flat.buffer.fill_(7.0)
    flat.absent = []
    critic = [2, 3]
    g2 = torch.randn(2, 2)
    foreign = {params[1].data_ptr(): torch.randn(3, 5)}  # a slab of a parameter outside the subset: dropped
    flat.assemble([g2, None], foreign, subset=critic)
    assert not foreign and flat.absent == [3]
    assert torch.equal(flat.buffer[20:24], g2.reshape(-1)) and float(flat.buffer[24:30].abs().sum()) == 0.0
    assert torch.equal(flat.buffer[:20], torch.full((20,), 7.0))  # the other windows untouched
    g0, slabs1 = torch.randn(4, 3), torch.randn(3, 5)
    flat.assemble([g0, None], {params[1].data_ptr(): slabs1}, subset=[0, 1])
    assert flat.absent == [3]
    torch.testing.assert_close(flat.buffer[:12], g0.reshape(-1))
    torch.testing.assert_close(flat.buffer[12:17], slabs1.sum(0))
    assert flat.window(critic).data_ptr() == flat.buffer[20:].data_ptr() and flat.window(critic).numel() == 12
    assert flat.window([0, 1]).numel() == 20
    with pytest.raises(ValueError, match="consecutive"):
        flat.window([0, 2])
    # (round 6) a slab keyed by something that is NO optimizer parameter at all is still an error, subset or not
    with pytest.raises(RuntimeError, match="not optimizer parameters"):
        flat.assemble([g2, None], {123456789: torch.randn(3, 5)}, subset=critic)


def test_graph_census_families():
    """scripts/graph_census.py sorts mangled kernel names into the families the captured-step rule is stated in."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("graph_census", Path(__file__).resolve().parent.parent / "scripts" / "graph_census.py")
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    assert module.family("_ZN2at6native13reduce_kernelILi512ELi1ENS0_8ReduceOpIfNS0_7MeanOpsIffffEEjfLi4ELi4EEEEEvT1_") == "aten_reduce"
    assert module.family("_ZN5cusrl13gather_kernelENS_10GatherArgsEPKcPKlllli") == "cusrl"
    assert module.family("Cijk_Alik_Bljk_SB_MT64x64x32_MI16x16x4x1") == "gemm"
    assert module.family("_ZN2at6native29vectorized_elementwise_kernelILi4EZZZNS0_16tanh_kernel_cudaE") == "aten_other"
    census = [{"region": "step", "kernel": 3, "memcpy": 0, "memset": 1, "names": ["_ZN2at6native13reduce_kernelIx", "_ZN5cusrl1aE", "Cijk_x"]}]
    import io

    totals = module.summarize(census, out=io.StringIO())
    assert totals == {"memset": 1, "aten_reduce": 1, "graphs": 1}


def test_lazy_batch_is_a_dict_that_gathers_on_first_access():
    """LazyBatch (template/buffer.py) against a stand-in buffer: which fields are fetched when, dict protocol, expiry."""
    from cusrl_amd.template.buffer import LazyBatch

    class FakeBuffer:
        schema = {"observation": "observation", "action_dist": {"mean": "action_dist.mean"}, "reward": "reward", "done": "done"}

        def __init__(self):
            self.calls = []

        def gather(self, indices, temporal=False, fields=None):
            self.calls.append(tuple(fields))
            return {name: (name, len(self.calls)) for name in fields}

    buffer, hot = FakeBuffer(), set()
    first = LazyBatch(buffer, None, False, hot)                # empty hot set: everything at once, like the reference
    assert buffer.calls == [("observation", "action_dist", "reward", "done")] and not first._pending
    assert first["reward"] == ("reward", 1) and first.get("done") == ("done", 1) and hot == {"reward", "done"}
    batch = LazyBatch(buffer, None, False, hot)                # now only the hot fields are prefetched
    assert buffer.calls[-1] == ("reward", "done") and list(batch._pending) == ["observation", "action_dist"]
    assert "observation" in batch and len(batch) == 4 and "next_observation" not in batch
    assert batch.get("missing", 5) == 5 and len(buffer.calls) == 2
    assert batch["observation"] == ("observation", 3) and "observation" in hot     # fetched alone, and learned
    batch["curr_value"] = 1.0
    batch["action_dist"] = "overwritten"                       # an overwritten field is never fetched
    assert len(buffer.calls) == 3 and batch["action_dist"] == "overwritten"
    assert set(batch) == {"reward", "done", "observation", "curr_value", "action_dist"}
    assert dict(**batch)["curr_value"] == 1.0
    third = LazyBatch(buffer, None, False, hot)
    assert set(dict.keys(third)) == {"reward", "done", "observation"}
    del third["action_dist"]
    assert "action_dist" not in third and len(third) == 3
    fourth = LazyBatch(buffer, None, False, hot)
    assert fourth.pop("action_dist") == ("action_dist", len(buffer.calls)) and "action_dist" not in fourth
    fifth = LazyBatch(buffer, None, False, hot)
    fifth.expire()
    assert fifth["reward"][0] == "reward"                      # fields already read stay valid
    try:
        fifth["action_dist"]
    except RuntimeError as error:
        assert "never read" in str(error)
    else:
        raise AssertionError("an expired batch must not fetch")
    try:
        fifth["nope"]
    except KeyError:
        pass
    else:
        raise AssertionError
    # update() / |= / | go through __setitem__: an overwritten pending field must never be re-fetched over the user's value
    sixth = LazyBatch(buffer, None, False, hot)
    calls = len(buffer.calls)
    sixth.update(action_dist="mine", extra=3)
    sixth |= {"observation": "also mine"}
    assert "action_dist" not in sixth._pending and len(sixth) == 5 and len(buffer.calls) == calls
    everything = dict(sixth.items())                                       # whole-batch access: nothing left to fetch over them
    assert everything["action_dist"] == "mine" and everything["observation"] == "also mine" and everything["extra"] == 3
    assert len(buffer.calls) == calls
    merged = sixth | {"reward": "override"}
    assert type(merged) is dict and merged["reward"] == "override" and sixth["reward"][0] == "reward"


def test_replayed_steps_repeat_their_host_side_effects():
    """What a hipGraph replay of an env step leaves to Python (template/graphs.py GraphedRolloutStep): the buffer cursor
    (``Buffer.replay_push``), the hooks' host halves (``Hook.on_replay``) and the update cadence (``ActorCritic.replay_step``)
    — exercised on CPU objects, no launch involved."""
    from cusrl_amd.hook.on_policy.value import ValueComputation
    from cusrl_amd.template.buffer import Buffer

    buffer = Buffer(3, 2, device="cpu")
    with pytest.raises(RuntimeError, match="no steady-state append"):
        buffer.replay_push()
    buffer._push_plan = (("reward",), (), (), (), None, (), ("reward",))  # as a steady-state plan would leave it
    buffer.set_derived("reward", "by-product")
    for expect_cursor, expect_full in ((1, False), (2, False), (0, True)):
        buffer.replay_push()
        assert buffer.cursor == expect_cursor and buffer.full == expect_full
    assert buffer.take_derived("reward") is None
    hook = ValueComputation()
    hook.agent = SimpleNamespace(critic=SimpleNamespace(is_recurrent=False), inference_mode=False, device=torch.device("cpu"),
                                 hook=[hook])
    hook.defer_value = True
    hook.on_replay("act")
    assert not hook._value_pending
    hook.on_replay("step")
    assert hook._value_pending                                              # pre_update must run the deferred critic pass
    hook.defer_value = False
    hook._value_pending = False
    hook.on_replay("step")
    assert not hook._value_pending


def test_tuned_gemm_selection_file_is_well_formed_and_inert_without_a_gpu():
    """cusrl_amd/tuned_gemms_gfx950.csv (scripts/tune_gemms.py): TunableOp's validators first, then one
    (operator, shape, kernel, milliseconds) row per GEMM shape of configs 2 and 3; nothing is loaded in a process
    without a GPU."""
    from cusrl_amd.utils import tuning

    rows = [line.split(",") for line in tuning.TUNED_GEMMS_FILE.read_text().splitlines() if line]
    validators = {row[1]: row[2] for row in rows if row[0] == "Validator"}
    assert {"PT_VERSION", "HIP_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= set(validators)
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    entries = [row for row in rows if row[0] != "Validator"]
    assert len(entries) >= 20 and all(len(row) == 4 and float(row[3]) > 0 for row in entries)
    shapes = {row[1] for row in entries}
    assert "tn_256_24576_48_ld_48_48_256" in shapes and "tn_128_24576_256_ld_256_256_128" in shapes  # the minibatch step's layers
    assert all(row[0].split("_")[0] in ("GemmTunableOp", "GemmAndBiasTunableOp", "GemmStridedBatchedTunableOp") for row in entries)
    if not torch.cuda.is_available():
        assert tuning.enable_tuned_gemms() is False
    # the programmatic opt-out the agent consults (a process-wide TunableOp setting must be refusable)
    from cusrl_amd.template import actor_critic

    assert actor_critic.CONFIG is cusrl.config  # `cusrl.config.tuned_gemms = False` before building an agent
    assert actor_critic.CONFIG.tuned_gemms is (os.environ.get("CUSRL_TUNED_GEMMS", "1") != "0")


def test_tracked_metadata_records_what_hooks_read():
    """graphs.TrackedMetadata: the metadata dict of a captured minibatch step remembers which keys were read, so that
    ActorCritic.update can key its step captures on their values (no stale capture-time branch)."""
    from cusrl_amd.template.graphs import TrackedMetadata

    reads: set[str] = set()
    metadata = TrackedMetadata({"epoch_index": 2, "mini_batch_index": 1, "temporal": False, "total_epochs": 5}, reads)
    assert not reads and len(metadata) == 4 and not reads           # constructing and sizing it reads nothing
    assert metadata["epoch_index"] == 2 and metadata.get("missing", 7) == 7 and "temporal" in metadata
    assert reads == {"epoch_index", "missing", "temporal"}
    assert dict(metadata.items())["total_epochs"] == 5 and reads >= {"total_epochs", "mini_batch_index"}
    other: set[str] = set()
    assert sorted(TrackedMetadata({"a": 1, "b": 2}, other)) == ["a", "b"] and other == {"a", "b"}   # iteration reads all


def test_empty_cuda_cache_hook_and_preset_fields():
    """hook.EmptyCudaCache (cusrl/hook/control/empty_cuda_cache.py:8-13) and where the presets put it (preset/ppo.py:35,64,243)."""
    with pytest.raises(ValueError, match="min_reserved_fraction"):
        cusrl.hook.EmptyCudaCache(min_reserved_fraction=1.5)
    hook = cusrl.hook.EmptyCudaCache()
    hook.agent = SimpleNamespace(device=torch.device("cpu"))
    hook.post_update()                                                # no GPU in this process: a no-op, like the reference
    assert hook.releases == 0 and hook.name == "empty_cuda_cache"
    names = lambda hooks: [h.name for h in hooks]  # noqa: E731
    assert names(cusrl.preset.ppo_hook_suite(empty_cuda_cache=True))[-1] == "empty_cuda_cache"
    assert "empty_cuda_cache" not in names(cusrl.preset.ppo_hook_suite())
    assert names(cusrl.preset.RecurrentPpoAgentFactory(rnn_type="GRU").to_underlying().hooks)[-1] == "empty_cuda_cache"
    assert "empty_cuda_cache" not in names(cusrl.preset.RecurrentPpoAgentFactory(empty_cuda_cache=False).to_underlying().hooks)
    assert "empty_cuda_cache" not in names(cusrl.preset.PpoAgentFactory().to_underlying().hooks)


def test_capturable_environment_protocol_defaults():
    """template/environment.py: an env is not capturable unless it says so, and the fixed-shape reset is its own to write;
    the synthetic env only advertises the protocol on a GPU."""
    from cusrl_amd.template.environment import Environment

    class Plain(Environment):
        def reset(self, *, indices=None, randomize_episode_progress=False):
            return torch.zeros(self.num_instances, self.observation_dim), None, {}

        def step(self, action):
            raise NotImplementedError

    env = Plain(3, 2, num_instances=4)
    assert env.capturable is False
    with pytest.raises(NotImplementedError, match="capturable reset protocol"):
        env.reset_static(torch.zeros(4, dtype=torch.int64), torch.zeros(1, dtype=torch.int32))
    synthetic = cusrl.testing.SyntheticEnvironment(4, 3, 2, device="cpu")
    assert synthetic.capturable is False                              # device-side protocol: CPU envs keep the reference loop
    observation, state, _ = synthetic.reset_static(torch.zeros(4, dtype=torch.int64), torch.zeros(1, dtype=torch.int32))
    assert observation.shape == (4, 3) and state is None
    _, _, reward, terminated, truncated, _ = synthetic.step(torch.zeros(4, 2))
    assert reward.shape == (4, 1) and terminated.dtype == torch.bool and terminated.shape == truncated.shape == (4, 1)
    assert terminated.is_contiguous() and truncated.is_contiguous()


@pytest.mark.parametrize("hidden", [[32, 16], [24], [16, 16, 8]])
def test_amp_closed_form_objective_matches_the_autograd_double_backward(hidden):
    """AdversarialMotionPrior.objective (cusrl/hook/auxiliary/amp.py:135-154; gradient penalty cusrl/nn/layer/loss.py:
    10-56): for a Linear / ReLU discriminator both terms and every parameter gradient in closed form equal the autograd
    evaluation (create_graph double backward) to float64 rounding, for one, two and three hidden layers and non-unit
    upstream gradients; any other discriminator keeps the autograd path."""
    from types import SimpleNamespace

    import cusrl_amd as cusrl
    from cusrl_amd.hook.auxiliary.amp import AdversarialMotionPrior

    def make(factory):
        hook = AdversarialMotionPrior(factory, dataset_source=torch.randn(1000, 12, dtype=torch.float64), batch_size=None,
                                      loss_weight=0.7, grad_penalty_weight=5.0)
        hook.agent = SimpleNamespace(device=torch.device("cpu"), to_tensor=torch.as_tensor, environment_spec=None)
        hook.register_module = lambda name, module: setattr(hook, name, module)
        hook.init()
        hook.discriminator.double()
        return hook

    torch.manual_seed(0)
    hook = make(cusrl.Mlp.Factory(hidden_dims=hidden))
    assert hook._relu_stack() is not None
    batch = {"agent_transition": torch.randn(64, 12, dtype=torch.float64), "expert_transition": torch.randn(64, 12, dtype=torch.float64)}
    results = {}
    for closed in (True, False):
        hook.closed_form_objective = closed
        hook.discriminator.zero_grad()
        terms = hook.objective({}, {name: value.clone() for name, value in batch.items()})
        assert list(terms) == ["amp_discrimination_loss", "amp_grad_penalty_loss"]
        (terms["amp_discrimination_loss"] * 1.3 + terms["amp_grad_penalty_loss"] * 0.9).backward()
        results[closed] = [t.detach().clone() for t in terms.values()] + [p.grad.clone() for p in hook.discriminator.parameters()]
    for got, want in zip(results[True], results[False]):
        assert (got - want).abs().max().item() <= 1e-12 * max(want.abs().max().item(), 1e-30)
    # a discriminator the closed form does not cover keeps the autograd path
    assert make(cusrl.Mlp.Factory(hidden_dims=hidden, activation_fn="Tanh"))._relu_stack() is None
    assert make(cusrl.Mlp.Factory(hidden_dims=hidden, dropout=0.1))._relu_stack() is None


def test_merge_of_tuned_gemm_selections_keeps_shipped_entries_and_drops_data_dependent_shapes(tmp_path, monkeypatch):
    """scripts/merge_tuned_gemms.py: shipped entries win unless --replace, validator lines must agree, and shapes whose
    row count depends on the data (not a multiple of 256, nor a multiple of 8 up to 512) never reach the shipped file."""
    import importlib.util
    import sys as _sys

    spec = importlib.util.spec_from_file_location("merge_tuned_gemms", ROOT / "scripts" / "merge_tuned_gemms.py")
    merge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(merge)
    validators = "Validator,PT_VERSION,2.10.0\nValidator,HIP_VERSION,700\n"
    shipped = tmp_path / "shipped.csv"
    shipped.write_text(validators + "GemmTunableOp_float_NN,nn_1_4096_128_ld_1_128_1,Gemm_Rocblas_1,0.01\n")
    fresh = tmp_path / "fresh.csv"
    fresh.write_text(validators
                     + "GemmTunableOp_float_NN,nn_1_4096_128_ld_1_128_1,Gemm_Rocblas_2,0.02\n"      # already shipped
                     + "GemmTunableOp_float_TN,tn_768_5504_256_ld_256_256_768,Gemm_Rocblas_3,0.02\n"  # 5504 = 21.5 x 256: dropped
                     + "GemmTunableOp_float_TN,tn_768_5632_256_ld_256_256_768,Gemm_Rocblas_4,0.02\n"  # 22 x 256: kept
                     + "GemmAndBiasTunableOp_float_TN,tn_256_512_12_ld_12_12_256,Default,0.01\n"      # the AMP batch: kept
                     + "GemmAndBiasTunableOp_float_TN,tn_256_983_48_ld_48_48_256,Default,0.01\n")     # truncated-env count: dropped
    monkeypatch.setattr(merge, "SHIPPED", shipped)
    monkeypatch.setattr(_sys, "argv", ["merge_tuned_gemms.py", str(fresh)])
    merge.main()
    lines = shipped.read_text().splitlines()
    assert lines[:2] == validators.splitlines()
    entries = {tuple(line.split(",")[:2]): line.split(",")[2] for line in lines[2:]}
    assert entries == {("GemmTunableOp_float_NN", "nn_1_4096_128_ld_1_128_1"): "Gemm_Rocblas_1",
                       ("GemmTunableOp_float_TN", "tn_768_5632_256_ld_256_256_768"): "Gemm_Rocblas_4",
                       ("GemmAndBiasTunableOp_float_TN", "tn_256_512_12_ld_12_12_256"): "Default"}
    monkeypatch.setattr(_sys, "argv", ["merge_tuned_gemms.py", str(fresh), "--replace"])
    merge.main()
    assert "Gemm_Rocblas_2" in shipped.read_text()
    other = tmp_path / "other.csv"
    other.write_text("Validator,PT_VERSION,2.11.0\nValidator,HIP_VERSION,700\n")
    monkeypatch.setattr(_sys, "argv", ["merge_tuned_gemms.py", str(other)])
    with pytest.raises(SystemExit):
        merge.main()


def test_recurrent_step_gemm_rows_are_bucketed_to_256_and_capped_by_the_batch():
    """nn/gru.py::_gemm_rows — the per-time-step recurrent GEMM covers the running sequences rounded up to a multiple of
    256 (never more than the batch), so its shape does not depend on where episodes ended."""
    from cusrl_amd.nn.gru import _gemm_rows

    assert [_gemm_rows(n, 5536) for n in (1, 255, 256, 257, 4100, 5376, 5377, 5536)] == [256, 256, 256, 512, 4352, 5376, 5536, 5536]
    assert _gemm_rows(7, 7) == 7 and _gemm_rows(100, 100) == 100


def test_critic_stream_branch_default_is_chosen_per_composition(monkeypatch):
    """GraphedTrainStep._critic_branch: forced by ``agent.concurrent_critic``; unset, the branch only where it measured
    faster — the stock fused composition at >= 4096-row minibatches (profiles/r05/stream_ab.txt); launch-bound small
    minibatches, RND / AMP chains in the same step and split compositions take one stream."""
    from types import SimpleNamespace

    from cusrl_amd.hook.auxiliary import RandomNetworkDistillation
    from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
    from cusrl_amd.template.graphs import GraphedTrainStep

    mode = {"value": "fused"}
    monkeypatch.setattr(FusedPpoObjective, "mode", staticmethod(lambda composite: mode["value"]))
    step = GraphedTrainStep.__new__(GraphedTrainStep)
    step.agent = SimpleNamespace(concurrent_critic=None, hook=[], buffer=SimpleNamespace(capacity=8))
    step.static_indices, step.temporal = torch.zeros(32, dtype=torch.int64), False  # a launch-bound 32-row minibatch
    assert not step._critic_branch()
    step.static_indices = torch.zeros(4096, dtype=torch.int64)
    assert step._critic_branch()
    step.static_indices, step.temporal = torch.zeros(512, dtype=torch.int64), True  # 512 env columns x 8 steps
    assert step._critic_branch()
    rnd = RandomNetworkDistillation.__new__(RandomNetworkDistillation)
    rnd._active = True
    step.agent.hook = [rnd]  # an auxiliary objective chain in the same step: one stream
    assert not step._critic_branch()
    rnd._active = False
    assert step._critic_branch()
    mode["value"] = "split"  # further objective hooks read curr_value on the main stream: one stream by construction
    assert not step._critic_branch()
    mode["value"] = None
    assert not step._critic_branch()
    mode["value"] = "fused"
    step.agent.concurrent_critic = False
    assert not step._critic_branch()
    step.agent.concurrent_critic, mode["value"] = True, None
    assert step._critic_branch()


def test_every_script_and_package_module_compiles():
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    files = sorted(root.glob("scripts/*.py")) + sorted(root.glob("cusrl_amd/**/*.py")) + [root / "bench.py", root / "__graft_entry__.py"]
    assert len(files) > 40
    for file in files:
        compile(file.read_text(), str(file), "exec")  # SyntaxError names the file


def test_staged_summary_keeps_its_own_entries_while_the_store_moves_on():
    """Round 6: ``Metrics.staged_summary`` issues the copy and returns; the store may be cleared and recorded into again before
    ``resolve()`` — deferred callbacks record into the STAGED read, not into whatever the store holds by then."""
    metrics = cusrl.utils.Metrics()
    metrics.record(loss=torch.tensor([1.0, 3.0]))
    metrics.defer(torch.tensor([4.0, 6.0]), lambda values: metrics.add_resolved("late", sum(values), 2))
    staged = metrics.staged_summary("Agent")
    metrics.clear()
    metrics.record(loss=torch.tensor([100.0]))              # the next iteration's recording
    assert staged.resolve() == {"Agent/loss": pytest.approx(2.0), "Agent/late": pytest.approx(5.0)}
    assert staged.resolve() is staged.resolve()             # resolved once
    assert metrics.summary("Agent") == {"Agent/loss": pytest.approx(100.0)}   # nothing of the staged read leaked into the store


def test_timer_detach_hands_the_sections_over_and_starts_from_zero():
    from cusrl_amd.utils.timing import Timer

    timer = Timer("cpu")
    with timer.record("agent"):
        pass
    for _ in range(3):
        with timer.record("environment", 2):
            pass
    frozen = timer.detach()
    assert frozen["agent"] >= 0.0 and frozen["environment"] >= 0.0 and set(frozen._seen) == {("environment", 2)}
    assert not timer._total and not timer._seen and timer["agent"] == 0.0
    with timer.record("agent"):
        with pytest.raises(RuntimeError, match="still open"):
            timer.detach()


def test_environment_stats_freeze_snapshots_the_rollout_and_resets_the_step_accumulators():
    from cusrl_amd.template.trainer import EnvironmentStats

    stats = EnvironmentStats(num_envs=4, reward_dim=1, buffer_size=8)
    for step in range(3):
        stats.track_step(torch.full((4, 1), float(step + 1)))
    stats.track_episode(torch.tensor([1, 3]))
    frame = stats.freeze()
    assert (frame.num_steps, frame.total_steps) == (3, 12) and stats.num_steps == 0 and stats.total_steps == 12
    length, reward, step_reward = frame.means
    assert length == pytest.approx(3.0) and reward == pytest.approx(6.0) and step_reward == pytest.approx(2.0)
    assert float(stats.reward.abs().sum()) == 0.0
    # the device form's decode, from a host list shaped like `_snapshot()`: count, reward sum, the ring of rewards, of lengths
    from cusrl_amd.template.trainer import StatsFrame

    decoded = StatsFrame(num_steps=2, total_steps=8, num_envs=4, reward_dim=1, buffer_size=3)
    decoded.receive([2.0, 16.0, 5.0, 7.0, 0.0, 10.0, 20.0, 0.0])
    assert decoded.means == (pytest.approx(15.0), pytest.approx(6.0), pytest.approx(2.0))


def test_trainer_flushes_a_pending_log_before_last_info_answers():
    """The trainer's loop with a stand-in agent on the CPU (the product agent only runs on a GPU): a host-driven rollout logs right
    away; a log left pending — what a captured rollout does — is written by ``flush()`` / by whoever asks for ``last_info`` first,
    under the iteration it belongs to; the training loop leaves nothing pending and checkpoints behind the log."""
    from cusrl_amd.template.agent import Agent, AgentFactory

    class CountingAgent(Agent):
        def act(self, observation, state=None):
            return torch.zeros(observation.shape[0], 2)

        def step(self, next_observation, reward, terminated, truncated, next_state=None, **kwargs):
            return super().step(next_observation, reward, terminated, truncated, next_state, **kwargs)

        def update(self):
            self.metrics.record(updates=torch.tensor(float(self.iteration + 1)))
            return super().update()

    class Factory(AgentFactory):
        def __call__(self, environment_spec):
            return CountingAgent(environment_spec, num_steps_per_update=4, name="Agent", device="cpu")

    class Logger:
        def __init__(self):
            self.entries = []

        def log(self, info, iteration):
            self.entries.append((iteration, info["Agent/updates"], info["Perf/environment_step"]))

        def save_checkpoint(self, checkpoint, iteration):
            self.entries.append(("checkpoint", iteration))

    env = cusrl.testing.DummyTorchEnvironment(num_instances=8, observation_dim=6, action_dim=2, device="cpu")
    logger = Logger()
    trainer = cusrl.Trainer(env, Factory(num_steps_per_update=4), logger_factory=lambda: logger, num_iterations=3, checkpoint_interval=2,
                            verbose=False)
    observation, state, _ = env.reset()
    observation, state = trainer._rollout_and_update(observation, state)
    assert trainer._pending_log is None and logger.entries == [(1, 1.0, 32)]   # a host-driven rollout logs right away
    assert trainer.last_info["Agent/updates"] == 1.0 and trainer.agent.deferred_summary is False
    trainer._pending_log = ({"Agent/updates": 9.0}, trainer.stats.freeze(), trainer.timer.detach(), 7)
    assert trainer.last_info["Agent/updates"] == 9.0 and trainer._pending_log is None and logger.entries[-1][:2] == (8, 9.0)
    logger.entries.clear()
    trainer.iteration = 0
    trainer.agent.iteration = 0
    trainer.run_training_loop()
    assert logger.entries == [("checkpoint", 0), (1, 1.0, 64), (2, 2.0, 96), ("checkpoint", 2), (3, 3.0, 128), ("checkpoint", 3)]


# ------------------------------------------------------------------------------------------------ third part of round 6
def test_packed_mean_var_is_the_finalize_row_itself_or_a_cat():
    """What a cross-rank merge gathers is ``mean | var`` as one row: the two halves of one allocation are handed over as they are
    (no launch); anything else is concatenated."""
    from cusrl_amd import ops

    row = torch.arange(6, dtype=torch.float32)
    mean, var = row[:3], row[3:]
    packed = ops.packed_mean_var(mean, var)
    assert packed.data_ptr() == row.data_ptr() and torch.equal(packed, row)
    for other_mean, other_var in ((mean.clone(), var), (mean, var.clone()), (row[1:4], row[3:]), (row[::2], row[1::2])):
        packed = ops.packed_mean_var(other_mean, other_var)
        assert torch.equal(packed, torch.cat((other_mean, other_var))) and packed.data_ptr() != row.data_ptr()


def test_ahead_of_time_noise_draws_are_opt_in_contracts():
    """The exploration noise of a rollout may only be drawn ahead of it when nothing else consumes torch's generator inside the
    rollout: an env says so (``generator_free`` — off by default, and off for the synthetic env wherever its fused step does not
    run), a hook that draws inside an env step says the opposite (``step_draws_random``: AdversarialMotionPrior)."""
    from cusrl_amd.hook.auxiliary import AdversarialMotionPrior, RandomNetworkDistillation
    from cusrl_amd.template.environment import Environment
    from cusrl_amd.template.hook import Hook

    assert Environment.generator_free is False and Hook.step_draws_random is False
    assert AdversarialMotionPrior.step_draws_random is True and RandomNetworkDistillation.step_draws_random is False
    env = cusrl.testing.SyntheticEnvironment(8, 4, 2, device="cpu")
    assert not env.fused and not env.generator_free  # (no GPU: torch-generator form)
    actor = cusrl.Actor.Factory(cusrl.Mlp.Factory((16, 16)), cusrl.NormalDist.Factory())(4, 2)
    assert actor.pending_noise is None and actor.noise_shape is None


def test_side_stream_module_names_its_switches():
    """``utils/streams.py`` needs a GPU to do anything; what a CPU process can hold is that it imports and documents its switches."""
    from cusrl_amd.utils import streams

    assert callable(streams.side_stream) and callable(streams.runs_beside)
    integration = (ROOT / "INTEGRATION.md").read_text()
    for switch in ("CUSRL_SIDE_STREAM_PROBE", "CUSRL_SIDE_STREAM_PRIORITY", "CUSRL_NORMED_MAIN_FIRST", "CUSRL_STEP_MAIN_FIRST",
                   "CUSRL_PREDRAW_NOISE", "CUSRL_TWO_WINDOW_STEP"):
        assert switch in integration, switch
