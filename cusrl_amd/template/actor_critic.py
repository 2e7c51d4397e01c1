"""Actor-critic agent: rollout into the HBM buffer, then minibatch PPO-style updates driven by hooks
(counterpart of cusrl/template/actor_critic.py:23-320; the ONNX/JIT export part is out of scope).

Differences that matter on MI355X (semantics unchanged):
* ``buffer.push`` is one HIP launch per env step; minibatches come from one gather launch (template/buffer.py);
* gradients live in ONE flat fp32 buffer (``FlatGradients``): ``zero_grad`` is a single memset, the per-step
  data-parallel all-reduce (RCCL over xGMI) needs no pack / unpack copies, and norm clipping is one reduction;
* the loss is ``Objectives.loss()`` — the fused kernel's pre-summed total when the stock PPO hooks are fused,
  otherwise the reference's left-fold ``sum(objectives.values())``.
"""

from __future__ import annotations

import os

from collections.abc import Iterable, Mapping
from dataclasses import dataclass
from typing import Any

import torch

from cusrl_amd.nn.actor import Actor, Value
from cusrl_amd.nn.module import collect_split_weight_grads, register_unit_gradient
from cusrl_amd.template.agent import Agent, AgentFactory, preserve_io_format
from cusrl_amd.template.buffer import Buffer, Sampler
from cusrl_amd.template.environment import EnvironmentSpec
from cusrl_amd.template.hook import Hook, HookComposite
from cusrl_amd.template.optimizer import OptimizerFactory, build_optimizer
from cusrl_amd.utils.config import CONFIG
from cusrl_amd.utils.distributed import FlatGradients, broadcast_parameters, reduce_gradients

__all__ = ["ActorCritic", "ActorCriticFactory", "HookList"]


def _hashable(value):
    """A metadata value as part of a step-graph key: hashable as it is, else its frozen form, else its repr."""
    from cusrl_amd.template.graphs import _freeze

    try:
        hash(value)
        return value
    except TypeError:
        frozen = _freeze(value)
        try:
            hash(frozen)
            return frozen
        except TypeError:
            return repr(value)


class HookList(list):
    """List of hooks with by-name attribute access (actor_critic.py:23-62)."""

    def to_dict(self):
        return {hook.name: hook for hook in self}

    @classmethod
    def from_dict(cls, data: dict[str, Hook]) -> "HookList":
        return cls(hook.name_(name) for name, hook in data.items())

    def __getattr__(self, name: str) -> Any:
        for hook in self:
            if hook.name == name:
                return hook
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    @classmethod
    def coerce(cls, data: Any) -> "HookList":
        if isinstance(data, dict):
            return cls.from_dict(data)
        if isinstance(data, (list, tuple)):
            return cls(data)
        raise TypeError(f"Unsupported hooks payload: {type(data)!r}")


@dataclass(kw_only=True)
class ActorCriticFactory(AgentFactory):
    actor_factory: Any
    critic_factory: Any
    optimizer_factory: OptimizerFactory | Mapping[str, OptimizerFactory]
    sampler: Sampler
    hooks: list

    def __post_init__(self):
        self.hooks = HookList.coerce(self.hooks)

    def __call__(self, environment_spec: EnvironmentSpec) -> "ActorCritic":
        return ActorCritic(
            environment_spec=environment_spec, actor_factory=self.actor_factory, critic_factory=self.critic_factory,
            optimizer_factory=self.optimizer_factory, sampler=self.sampler, hooks=self.hooks,
            num_steps_per_update=self.num_steps_per_update, name=self.name, device=self.device, compile=self.compile,
            autocast=self.autocast,
        )

    def register_hook(self, hook: Hook, index: int | None = None, before: str | None = None, after: str | None = None):
        """Insert ``hook`` at ``index``, or relative to a named hook; append by default (``:97-136``)."""
        if (index is not None) + (before is not None) + (after is not None) > 1:
            raise ValueError("Only one of index, before, or after can be specified")
        if before is not None:
            index = self.get_hook_index(before)
        elif after is not None:
            index = self.get_hook_index(after) + 1
        elif index is None:
            index = len(self.hooks)
        self.hooks.insert(index, hook)
        return self

    def get_hook(self, hook_name: str) -> Hook:
        return self.hooks[self.get_hook_index(hook_name)]

    def get_hook_index(self, hook_name: str) -> int:
        for i, hook in enumerate(self.hooks):
            if hook.name == hook_name:
                return i
        raise ValueError(f"No hook named '{hook_name}' is registered")


class ActorCritic(Agent):
    Factory = ActorCriticFactory
    MODULES = ["actor", "critic", "hook"]
    STATEFULS = ["optimizer", "grad_scaler"]

    def __init__(self, environment_spec: EnvironmentSpec, actor_factory, critic_factory, optimizer_factory,
                 sampler: Sampler, hooks: Iterable[Hook], num_steps_per_update: int, name: str = "Agent",
                 device=None, compile: bool | str = False, autocast=False):
        super().__init__(environment_spec, num_steps_per_update, name, device, compile, autocast)
        self.value_dim = environment_spec.reward_dim
        self.buffer_capacity = num_steps_per_update
        self.actor_factory, self.critic_factory, self.optimizer_factory = actor_factory, critic_factory, optimizer_factory
        self.fuse_objective = True
        self._fused_objective = None

        self.hook = HookComposite(hooks)
        self.hook.pre_init(self)
        self.actor: Actor = actor_factory(self.observation_dim, self.action_dim)
        action_aware = getattr(critic_factory, "action_aware", False)
        self.critic: Value = critic_factory(self.state_dim + self.action_dim * action_aware, self.value_dim)
        self.buffer = Buffer(self.buffer_capacity, self.parallelism, device=self.device)
        self.sampler = sampler
        self.grad_scaler = torch.GradScaler(device=self.device.type, enabled=self.grad_scaler_enabled)
        self.actor_memory = None
        self.hook.init()

        if self.device.type == "cuda" and CONFIG.tuned_gemms:
            from cusrl_amd.utils.tuning import enable_tuned_gemms

            enable_tuned_gemms()  # measured rocBLAS / hipBLASLt kernel choice for this workload's GEMM shapes
        self.actor = self.setup_module(self.actor)
        self.critic = self.setup_module(self.critic)
        self.optimizer = build_optimizer(optimizer_factory, self.named_parameters())
        self._graphed_act = None
        self._graphed_steps: dict[tuple, Any] = {}
        self._graphed_epochs = None
        self._branch_tail = None  # a callable GraphedEpochs wants issued at the tail of the running step's critic branch
        # set by GraphedEpochs around the step bodies of a whole-update / whole-epoch capture: a step may leave its two streams
        # unjoined (they meet behind the LAST body); `_batch_on_branch`: the running step's rows were gathered on the critic's stream
        self._unjoined_steps = False
        self._batch_on_branch = False
        self._while_waiting = None  # host work to issue while a pre_update hook waits for the device (run_while_waiting)
        self._minibatches_done = None  # event behind the last minibatch step of the previous update (its index rows may be redrawn)
        self._graph_key_reads = 0
        self._graph_budget_warned = False
        self._metadata_reads: set[str] = set()  # metadata keys hooks read inside captured steps (graphs.TrackedMetadata)
        if self.compile:
            # `compile=True` = hipGraph replay of the act step and of every minibatch step (template/graphs.py)
            if self.device.type != "cuda":
                raise RuntimeError("compile=True captures hipGraphs and needs a GPU device")
            if any(not group.get("capturable", False) for group in self.optimizer.param_groups):
                raise ValueError("compile=True needs a graph-capturable optimizer, e.g. Adam(capturable=True, fused=True)")
            from cusrl_amd.template.graphs import GraphedAct

            self._graph_stream = torch.cuda.Stream(device=self.device)
            self._graph_pool = torch.cuda.graph_pool_handle()
            # second branch of the captured minibatch step (critic forward / backward, hook/on_policy/value.py)
            self._branch_stream = torch.cuda.Stream(device=self.device)
            # True / False force it; None (default, CUSRL_CONCURRENT_CRITIC unset) = per composition, where it measured faster
            # (GraphedTrainStep._critic_branch)
            forced = os.environ.get("CUSRL_CONCURRENT_CRITIC")
            self.concurrent_critic = None if forced is None else forced != "0"
            # captured minibatch steps run the fused objective without its one-block finalize launch (ops.DeferredLoss)
            self.defer_loss_finalize = os.environ.get("CUSRL_DEFER_LOSS_FINALIZE", "1") != "0"
            self._graphed_act = GraphedAct(self)
        self.flat_gradients: FlatGradients | None = None
        self._split_plan = False  # per-network split of the backward: not looked at yet (None = does not apply)
        self._networks = False  # critic / other parameter windows: not looked at yet (None = they share parameters)
        self._unit_grads: dict[tuple, torch.Tensor] = {}
        # the value term of the stock composition as its own launch + root on the critic's branch of a captured step (A/B switch)
        self._separate_value_term = os.environ.get("CUSRL_SEPARATE_VALUE_TERM", "1") != "0"
        if isinstance(self.optimizer, torch.optim.Optimizer) and not self.grad_scaler_enabled:
            self.flat_gradients = FlatGradients(self.optimizer)
        self.flat_optimizer = None
        if self.flat_gradients is not None and self.device.type == "cuda" and all(
                group.get("fused") for group in self.optimizer.param_groups):
            # a fused Adam / AdamW was asked for: step it as ONE HIP launch over flat buffers (utils/flat_optimizer.py)
            from cusrl_amd.utils.flat_optimizer import FlatAdam

            if FlatAdam.eligible(self.optimizer, self.flat_gradients):
                self.flat_optimizer = FlatAdam(self.optimizer, self.flat_gradients)
                self.flat_optimizer.metrics = self.metrics
        self._set_training_mode(False)
        self.hook.post_init()
        broadcast_parameters(self.parameters())
        self.hook.apply_schedule(0)

    def _save_transition(self, *, _clone: bool = True, **fields):
        """Store non-None fields as device tensors.  ``_clone`` keeps the reference's defensive copy
        (agent.py:257-261) for values that must survive an ``env.step`` before they are pushed (observation, state,
        recurrent memory); values pushed into the buffer right away are stored as they are."""
        device, transition = self.device, self.transition
        for key, value in fields.items():
            if value is None:
                continue
            if not _clone and type(value) is torch.Tensor and value.device == device:
                transition[key] = value  # the common case in the rollout loop: already a tensor where it belongs
                continue
            try:
                self.transition[key] = self.to_nested_tensor(value) if _clone else self._as_nested_tensor(value)
            except Exception as error:
                raise ValueError(f"Failed to convert transition field '{key}' to a tensor") from error

    def _as_nested_tensor(self, value):
        if isinstance(value, (tuple, list)):
            return tuple(self._as_nested_tensor(v) for v in value)
        if isinstance(value, Mapping):
            return {k: self._as_nested_tensor(v) for k, v in value.items()}
        return torch.as_tensor(value, device=self.device)

    @torch.no_grad()
    @preserve_io_format
    def act(self, observation, state=None):
        if self._graphed_act is not None:
            observation_t = torch.as_tensor(observation, device=self.device)
            state_t = None if state is None else torch.as_tensor(state, device=self.device)
            if self._graphed_act.supported(observation_t, state_t):
                return self._graphed_act.run(observation_t, state_t)
        self.transition.clear()
        self._save_transition(observation=observation, state=state)
        self.hook.pre_act(self.transition)
        with self.autocast():
            action_dist, (action, action_logp), next_memory = self.actor.explore(
                self.transition["observation"], memory=self.actor_memory, deterministic=self.deterministic,
                backbone_kwargs={"sequential": False},
            )
        self._save_transition(actor_memory=self.actor_memory)
        self.transition.update(action_dist=action_dist, action=action, action_logp=action_logp)
        self.actor_memory = next_memory
        self.hook.post_act(self.transition)
        return self.transition["action"]

    @torch.no_grad()
    def step(self, next_observation, reward, terminated, truncated, next_state=None, **kwargs) -> bool:
        self._save_transition(_clone=False, next_observation=next_observation, next_state=next_state, reward=reward,
                              terminated=terminated, truncated=truncated, **kwargs)
        transition = self.transition
        if transition["terminated"].dtype != torch.bool:
            raise TypeError("'terminated' must have dtype bool")
        if transition["truncated"].dtype != torch.bool:
            raise TypeError("'truncated' must have dtype bool")
        supplied = kwargs.get("done")  # extension: the trainer's fused step epilogue already formed terminated | truncated
        if not (type(supplied) is torch.Tensor and supplied.dtype == torch.bool and supplied.shape == transition["terminated"].shape
                and supplied.device == transition["terminated"].device):
            transition["done"] = transition["terminated"] | transition["truncated"]
        self.hook.post_step(transition)
        if not self.inference_mode:
            self.buffer.push(transition)  # a1: every leaf of the transition in one HIP launch
        self.actor.reset_memory(self.actor_memory, transition["done"])
        ready = super().step(next_observation, reward, terminated, truncated, next_state, **kwargs)
        return ready and self.hook.should_update(transition)

    def replay_step(self) -> bool:
        """Host half of :meth:`step` for an env step whose device work was replayed from a hipGraph
        (template/graphs.py GraphedRolloutStep): hooks' host effects, buffer cursor, update cadence."""
        self.hook.on_replay("step")
        if not self.inference_mode:
            self.buffer.replay_push()
        ready = Agent.step(self, None, None, None, None)
        return ready and self.hook.should_update(self.transition)

    def _steps_draw_random(self) -> bool:
        """Does anything between two permutation draws consume torch's generator (a hook's objective, a dropout layer)?"""
        if any(hook.active and hook.objective_draws_random for hook in self.hook):
            return True
        modules = [self.actor, self.critic] + [m for hook in self.hook for m in hook._modules.values() if m is not None]
        return any(isinstance(layer, torch.nn.modules.dropout._DropoutNd) and layer.p > 0
                   for module in modules for layer in module.modules())

    def _check_sampler_prefetch(self):
        if hasattr(self.sampler, "prefetch") and getattr(self, "_sampler_prefetch_checked", None) is not self.sampler:
            self._sampler_prefetch_checked = self.sampler
            if self._steps_draw_random():
                self.sampler.prefetch = False  # keep the reference's interleaving of permutation and in-step draws

    def update(self):
        self._check_sampler_prefetch()
        # The first epoch's permutation depends on nothing pre_update computes: drawn NOW (side stream), it runs under
        # pre_update's kernels instead of between them and the first minibatch step.  Only when no hook this package does not
        # know could draw from the generator inside pre_update (the reference draws the permutation behind it,
        # cusrl/sampler/mini_batch_sampler.py:56): none of this package's hooks does.
        early = self._draw_epochs(prepare=False) if self._draws_early() else None
        if early is not None:
            from cusrl_amd.template.graphs import epoch_graphs_mode

            if epoch_graphs_mode() == "update":
                # the whole-update graph needs EVERY epoch's permutation before it starts: the remaining draws are issued while
                # the host has nothing to do — ValueComputation.pre_update waits for the truncated count of a region it has
                # just launched (`run_while_waiting`) — and run under that region on the side stream
                def while_waiting():
                    early.draw(len(early) - 1)
                    if self._graphed_epochs is not None:  # ... and the host half of the update graph's launch (GraphedEpochs.prepare)
                        self._graphed_epochs.prepare(early)

                self._while_waiting = while_waiting
        try:
            self.hook.pre_update(self.buffer)  # a3-a6: next_value, GAE, advantage normalisation
        finally:
            self._while_waiting = None
        with self._training_mode():
            # (recurrent networks: dynamic sequence counts, not capturable; a hook's collective / host read-back stays out of capture)
            graphed = self._graphable()
            if graphed:
                from cusrl_amd.template.graphs import GraphedTrainStep

                # every epoch's permutation drawn up front on the draw-ahead stream (same generator calls, same order):
                # once every step replays from its own graph, a whole epoch's steps replay from ONE graph that reads its
                # index slices in place and gathers each step's rows while the step before it runs (template/graphs.py
                # GraphedEpochs); until then — and whenever a condition does not hold — the steps run graph by graph over
                # the very same permutations
                from cusrl_amd.template.graphs import GraphedEpochs

                drawn = early if early is not None else self._draw_epochs()
                if drawn is not None and self._graphed_epochs is None:
                    self._graphed_epochs = GraphedEpochs(self)
                steps = () if drawn is not None and self._graphed_epochs.run(drawn) else (
                    self._iter_drawn(drawn) if drawn is not None else self.sampler.iter_indices(self.buffer))
                for metadata, indices in steps:
                    key = self._step_key(metadata, indices.numel())
                    if (step := self._graphed_steps.get(key)) is None:
                        if self._graph_key_reads != len(self._metadata_reads):
                            # the set of metadata keys hooks read has grown: graphs keyed on the shorter signature can
                            # never be looked up again — release them (and their static buffers)
                            self._graph_key_reads = len(self._metadata_reads)
                            width = len(key)
                            for stale in [k for k in self._graphed_steps if len(k) != width]:
                                self._graphed_steps.pop(stale).flush_metrics()
                        step = self._graphed_steps[key] = GraphedTrainStep(self, key[0], key[1])
                        budget = self.sampler.num_epochs * (self.sampler.num_mini_batches if isinstance(self.sampler.num_mini_batches, int)
                                                            else max(self.sampler.num_mini_batches)) if hasattr(self.sampler, "num_epochs") else 0
                        if budget and len(self._graphed_steps) > budget and not self._graph_budget_warned:
                            self._graph_budget_warned = True
                            self.warn(f"{len(self._graphed_steps)} minibatch-step graphs for {budget} steps per update: a hook reads "
                                      f"metadata whose values keep changing ({sorted(self._metadata_reads)}); every distinct value is "
                                      "its own capture")
                    step.run(metadata, indices)
                if drawn is not None:
                    if self._minibatches_done is None:
                        self._minibatches_done = torch.cuda.Event()
                    self._minibatches_done.record(torch.cuda.current_stream())
                # what the replays accumulated on the device (metric taps, loss sums): handed to the metrics, read in ONE host
                # copy together with everything else this update recorded (Agent.update -> Metrics.summary)
                for step in self._graphed_steps.values():
                    step.flush_metrics()
                if self._graphed_epochs is not None:
                    self._graphed_epochs.flush_metrics()
            else:
                for metadata, batch in self.sampler(self.buffer):  # a7/a8
                    self._train_step(metadata, batch)
        self.hook.post_update()
        self.hook.apply_schedule(self.iteration + 1)
        return super().update()

    def run_while_waiting(self) -> None:
        """Called by a hook right before it blocks on a device result it has just launched the work for: host work that depends
        on neither (here: drawing the later epochs' permutations) is issued now instead of behind the wait."""
        pending, self._while_waiting = self._while_waiting, None
        if pending is not None:
            pending()

    def _graphable(self) -> bool:
        """Does this update's minibatch loop go through captured steps (``compile=True``, an index-yielding sampler, feed-forward
        networks, no hook that keeps its objective phase out of capture)?"""
        if not (self.compile and hasattr(self.sampler, "iter_indices")) or self.actor.is_recurrent or self.critic.is_recurrent:
            return False
        from cusrl_amd.template.graphs import eager_phases

        return "objective" not in eager_phases(self)

    def _draws_early(self) -> bool:
        return self._graphable() and all(type(hook).__module__.startswith("cusrl_amd.") for hook in self.hook if hook.active)

    def _draw_epochs(self, prepare: bool = True):
        """The sampler's up-front permutations (``DrawnEpochs``) when whole-epoch graphs are on and the sampler offers them."""
        from cusrl_amd.template.graphs import epoch_graphs_enabled

        if not (epoch_graphs_enabled() and hasattr(self.sampler, "draw_epochs")):
            return None
        # (the draw only has to wait for the previous update's readers of the index rows — not for whatever the main stream has
        # been given since)
        return self.sampler.draw_epochs(self.buffer, after=self._minibatches_done, prepare=prepare)

    def _step_key(self, metadata, numel: int) -> tuple:
        """Key of the captured minibatch step that serves ``metadata``: slot, sampling form, batch size and the value of every
        metadata key a hook ever read (steps whose values differ are different captures)."""
        key = (metadata["mini_batch_index"], metadata["temporal"], numel)
        if self._metadata_reads:
            key += tuple((name, _hashable(metadata.get(name))) for name in sorted(self._metadata_reads)
                         if name not in ("mini_batch_index", "temporal"))
        return key

    @staticmethod
    def _iter_drawn(drawn):
        """``(metadata, index slice)`` of every minibatch of permutations drawn by ``sampler.draw_epochs`` — each epoch after
        its permutation's event."""
        for epoch, row in enumerate(drawn.plan):
            drawn.wait(epoch)
            for metadata, lo, hi in row:
                yield dict(metadata), drawn.permutations[epoch, lo:hi]

    def _zero_grad(self):
        if self.flat_optimizer is not None:
            self.flat_optimizer.discard_pending_clip()  # a clip deferred for a step that never ran must not leak into this one
        if self.flat_gradients is None:
            self.optimizer.zero_grad()
        elif not self.flat_gradients.intact():
            self.flat_gradients.attach()

    def _backward(self, loss):
        """Gradients of ``loss`` into ``p.grad``.  With the flat gradient buffer the per-parameter gradients — and the
        unsummed slabs of the split-batch weight-gradient GEMMs — are written into it by ONE kernel (no memset, no 13
        accumulate launches, no per-layer sum(0)); otherwise this is the
        reference's ``scaled_loss.backward()`` (actor_critic.py:311-312)."""
        flat = self.flat_gradients
        roots = list(loss) if isinstance(loss, (list, tuple)) else [loss]  # several roots = the summands of the loss
        if len(roots) > 1:  # a constant summand (a hook returning a plain number) changes no gradient
            roots = [t for t in roots if isinstance(t, torch.Tensor) and t.requires_grad] or roots[:1]
        if flat is None:
            total = roots[0]
            for term in roots[1:]:
                total = total + term
            self.grad_scaler.scale(total).backward()
            return
        units = [self._unit_gradient(term) for term in roots]
        plan = self._split_backward_plan()
        # a summand evaluated on the critic's stream (hook/on_policy/value.py: the value term of the stock composition)
        branch_root = getattr(loss, "branch", None)
        networks = self._network_windows() if branch_root is not None else None
        if branch_root is not None and (networks is None or len(roots) != len(loss)):
            # not differentiable network by network (shared parameters, or a summand was dropped above): join first
            torch.cuda.current_stream().wait_stream(branch_root[1])
            branch_root = None
        if plan is None and branch_root is not None:
            # The critic's backward where its forward and its loss ran, the other summands' on the main stream, both issued from
            # here back to back: neither pass waits for the other (the engine would order a one-pass backward behind the stream
            # this call is made from), the two streams meet ONCE, in front of the assembly.
            value_root, branch = branch_root
            critic_ids, other_ids = networks
            position = next(i for i, term in enumerate(roots) if term is value_root)
            others = [term for i, term in enumerate(roots) if i != position]
            other_units = [unit for i, unit in enumerate(units) if i != position]
            ranges = self._two_window_ranges() if self._unjoined_steps else None
            if ranges is not None:
                # Round 6, inside a whole-update / whole-epoch graph: each network's window of the flat buffer is assembled on the
                # stream its backward ran on, and the streams do NOT meet — the optimizer steps the two windows where they are
                # (FlatAdam.step: each behind the other window's assembly, the clipping coefficient needs both).  The rows of the
                # NEXT minibatch step are gathered in front of the critic's assembly, so the one event the main stream waits for
                # covers them too.
                main = torch.cuda.current_stream()
                flat.absent = []
                # (several ranks: the rows would be of the un-averaged gradients — reduce_gradients averages the buffer behind
                # both assemblies, the step launches measure its norm themselves)
                from cusrl_amd.utils.config import configure_distributed

                multi_rank = configure_distributed()
                with torch.cuda.stream(branch):
                    with collect_split_weight_grads() as critic_slabs:
                        critic_grads = torch.autograd.grad([value_root], [flat.params[i] for i in critic_ids],
                                                           grad_outputs=[units[position]], allow_unused=True)
                    tail, self._branch_tail = self._branch_tail, None
                    if tail is not None:
                        tail()
                    branch_sumsq = flat.assemble(critic_grads, critic_slabs, subset=critic_ids, want_sumsq=not multi_rank)
                    branch_assembled = torch.cuda.Event()
                    branch_assembled.record(branch)
                with collect_split_weight_grads() as split_slabs:
                    other_grads = torch.autograd.grad(others, [flat.params[i] for i in other_ids], grad_outputs=other_units,
                                                      allow_unused=True)
                main_sumsq = flat.assemble(other_grads, split_slabs, subset=other_ids, want_sumsq=not multi_rank)
                main_assembled = torch.cuda.Event()
                main_assembled.record(main)
                main_range, branch_range = ranges
                # (the rows in parameter order: summed as ONE assembly's rows would be — the same norm to the bit)
                sumsq = (main_sumsq, branch_sumsq) if main_range[0] < branch_range[0] else (branch_sumsq, main_sumsq)
                flat.split_tail = {"branch": branch, "branch_assembled": branch_assembled, "main_assembled": main_assembled,
                                   "sumsq": sumsq, "main_range": main_range, "branch_range": branch_range, "reduce": multi_rank}
                return
            with torch.cuda.stream(branch):
                with collect_split_weight_grads() as critic_slabs:
                    critic_grads = torch.autograd.grad([value_root], [flat.params[i] for i in critic_ids],
                                                       grad_outputs=[units[position]], allow_unused=True)
                # the critic's branch ends before the actor's: work that depends on neither — the gather of the NEXT minibatch
                # step's rows (template/graphs.py GraphedEpochs) — rides at its tail
                tail, self._branch_tail = self._branch_tail, None
                if tail is not None:
                    tail()
            with collect_split_weight_grads() as split_slabs:
                other_grads = torch.autograd.grad(others, [flat.params[i] for i in other_ids], grad_outputs=other_units,
                                                  allow_unused=True)
            torch.cuda.current_stream().wait_stream(branch)  # the step's one join
            grads: list = [None] * len(flat.params)
            for i, grad in zip(critic_ids, critic_grads):
                grads[i] = grad
            for i, grad in zip(other_ids, other_grads):
                grads[i] = grad
            split_slabs.update(critic_slabs)
            flat.assemble(grads, split_slabs)
            return
        if plan is None:
            with collect_split_weight_grads() as split_slabs:
                grads = torch.autograd.grad(roots, flat.params, grad_outputs=units, allow_unused=True)
            flat.assemble(grads, split_slabs)
            return
        # Per-network split (CONFIG.split_gradient_allreduce; cusrl/utils/distributed.py:145-172 reduces once, behind the
        # whole backward): the critic first — its window is assembled and averaged on the branch stream through the second
        # communicator while the actor's backward, issued right behind, runs on the main stream.  Same autograd nodes, same
        # kernels, same operands as the one-pass backward; the loss node they share is evaluated by both passes.
        from cusrl_amd.utils import distributed

        critic_ids, other_ids, windows = plan
        main, branch = torch.cuda.current_stream(), self._branch_stream
        # collectives inside the backward need a route that may be enqueued here: eager, or capturable (the C ABI)
        inline = not torch.cuda.is_current_stream_capturing() or distributed.native_comm() is not None
        flat.absent, flat.split_windows = [], windows
        # The critic's pass and its window's assembly run where the critic's autograd nodes run: on the branch stream when the
        # step evaluated the critic there (GraphedTrainStep._critic_branch), else on the main stream — gradients are consumed on
        # the stream whose allocator pool they came from.  Only the all-reduce, which touches nothing but the persistent flat
        # buffer, always goes to the branch stream.
        on_branch = getattr(self, "_critic_backward_stream", None) is branch
        critic_roots, critic_units, other_roots, other_units = roots, units, roots, units
        if branch_root is not None:  # the value term is a root of its own (evaluated on `branch`): each pass takes its summands
            position = next(i for i, term in enumerate(roots) if term is branch_root[0])
            critic_roots, critic_units = [roots[position]], [units[position]]
            other_roots = [term for i, term in enumerate(roots) if i != position]
            other_units = [unit for i, unit in enumerate(units) if i != position]
            on_branch = True

        def critic_pass():
            with collect_split_weight_grads() as slabs:
                grads = torch.autograd.grad(critic_roots, [flat.params[i] for i in critic_ids], grad_outputs=critic_units,
                                            allow_unused=True, retain_graph=branch_root is None)
            flat.assemble(grads, slabs, subset=critic_ids)

        # Two collectives in flight on two streams need two communicators: without the second one (it could not be created, or
        # the route is torch.distributed's) the critic's window is averaged on the MAIN stream behind the join below — one
        # communicator is only ever used from one stream at a time.
        on_branch_comm = inline and distributed.branch_comm() is not None

        def critic_reduce():
            if on_branch_comm:
                distributed.branch_comm().allreduce_mean_(windows[0])

        if on_branch:
            branch.wait_stream(main)
            with torch.cuda.stream(branch):
                critic_pass()
                critic_reduce()
        else:
            critic_pass()
            branch.wait_stream(main)
            with torch.cuda.stream(branch):
                critic_reduce()
        with collect_split_weight_grads() as slabs:
            grads = torch.autograd.grad(other_roots, [flat.params[i] for i in other_ids], grad_outputs=other_units, allow_unused=True)
        flat.assemble(grads, slabs, subset=other_ids)
        if inline:
            for window in windows[1:]:
                distributed.reduce_mean_(window)
        main.wait_stream(branch)
        if inline and not on_branch_comm:
            distributed.reduce_mean_(windows[0])
        flat.reduced = inline

    def _unit_gradient(self, term: torch.Tensor) -> torch.Tensor:
        """The persistent, registered ones-scalar of ``term``'s dtype and device (no ones_like per step; custom backwards
        recognise a registered scalar by identity, nn/module.py) — one per (dtype, device), for every root of a step."""
        key = (term.dtype, term.device)
        unit = self._unit_grads.get(key)
        if unit is None:
            unit = self._unit_grads[key] = register_unit_gradient(torch.ones((), dtype=term.dtype, device=term.device))
        return unit

    def _network_windows(self):
        """``(critic parameter indices, the others' indices)`` of the flat gradient buffer when the critic shares no parameter
        with anything else the optimizer steps (then a summand that reaches the critic alone can be differentiated on its own),
        else None.  Computed once."""
        if self._networks is not False:
            return self._networks
        self._networks = None
        flat = self.flat_gradients
        if flat is None:
            return None
        critic = {id(p) for p in self.critic.parameters()}
        outside = {id(p) for p in self.actor.parameters()} | {id(p) for p in self.hook.parameters()}
        critic_ids = [i for i, p in enumerate(flat.params) if id(p) in critic]
        if critic_ids and not (critic & outside):
            self._networks = (critic_ids, [i for i, p in enumerate(flat.params) if id(p) not in critic])
        return self._networks

    def _two_window_ranges(self):
        """``(element range of the others' window, of the critic's window)`` of the flat buffers when a minibatch step may leave
        its two streams unjoined (``_backward`` / ``FlatAdam.step``): one process, the flat Adam step, critic and others each one
        run of consecutive parameters, nobody but the stock gradient clipping between backward and step (a ``pre_optim`` of
        another hook could read gradients of the window that lives on the other stream); else None.  Computed once."""
        cached = getattr(self, "_two_windows", False)
        if cached is not False:
            return cached
        self._two_windows = None
        from cusrl_amd.hook.on_policy.gradient_clipping import GradientClipping
        from cusrl_amd.template.hook import Hook
        from cusrl_amd.utils.config import configure_distributed

        networks, flat = self._network_windows(), self.flat_gradients
        if networks is None or flat is None or self.flat_optimizer is None:
            return None
        if configure_distributed():
            # several ranks: the step stays unjoined when the all-reduce can be captured where it belongs — ONE collective over
            # the whole buffer through the C-ABI communicator, on the main stream behind both assemblies (reduce_gradients) —
            # and the step launches measure the averaged gradients' norm themselves (cusrl_adam_step_normed)
            from cusrl_amd.utils import distributed
            from cusrl_amd.utils.config import CONFIG

            if CONFIG.split_gradient_allreduce or distributed.native_comm() is None:
                return None
        if os.environ.get("CUSRL_TWO_WINDOW_STEP", "1") == "0":  # A/B switch
            return None
        for hook in self.hook:
            stock = type(hook).pre_optim is Hook.pre_optim and type(hook).post_optim is Hook.post_optim
            if hook._active and not stock and not (isinstance(hook, GradientClipping) and not hook.groups
                                                   and type(hook).post_optim is Hook.post_optim):
                return None
        try:
            self._two_windows = (flat.element_range(networks[1]), flat.element_range(networks[0]))
        except ValueError:
            self._two_windows = None
        return self._two_windows

    @property
    def separate_value_root(self) -> bool:
        """May ``ValueLoss`` evaluate its term by its own launch on the critic's stream (a root of its own in ``_backward``)?"""
        return self._separate_value_term and self._network_windows() is not None

    def _split_backward_plan(self):
        """``(critic parameter indices, the others' indices, [critic window, other windows ...])`` of the flat gradient buffer
        when the per-network split of the backward applies — a multi-rank job with ``CONFIG.split_gradient_allreduce``, a
        GPU agent with a branch stream, critic parameters forming one run of the buffer — else None.  Computed once."""
        if self._split_plan is not False:
            return self._split_plan
        self._split_plan = None
        from cusrl_amd.utils.config import CONFIG, configure_distributed

        flat = self.flat_gradients
        if (flat is None or not CONFIG.split_gradient_allreduce or not configure_distributed() or self.device.type != "cuda"
                or getattr(self, "_branch_stream", None) is None):
            return None
        critic = {id(p) for p in self.critic.parameters()}
        if critic & {id(p) for p in self.actor.parameters()}:
            return None  # shared parameters: one pass
        critic_ids = [i for i, p in enumerate(flat.params) if id(p) in critic]
        other_ids = [i for i, p in enumerate(flat.params) if id(p) not in critic]
        if not critic_ids or not other_ids or critic_ids != list(range(critic_ids[0], critic_ids[-1] + 1)):
            return None
        runs, run = [], [other_ids[0]]
        for i in other_ids[1:]:
            if i == run[-1] + 1:
                run.append(i)
            else:
                runs.append(run)
                run = [i]
        runs.append(run)
        self._split_plan = (critic_ids, other_ids, [flat.window(critic_ids)] + [flat.window(r) for r in runs])
        return self._split_plan

    def _train_step(self, metadata: dict[str, Any], batch: dict[str, Any]):
        self.actor.clear_intermediate_repr()
        self.critic.clear_intermediate_repr()
        self.hook.pre_objective(metadata, batch)
        with self.autocast():
            objectives = self.hook.objective(metadata, batch)  # a9-a13
        if objectives is not None:
            # with the flat gradient buffer the summands are differentiated as separate roots (no additions launched)
            if hasattr(objectives, "terms") and self.flat_gradients is not None:
                loss = objectives.terms()
            else:
                loss = objectives.loss() if hasattr(objectives, "loss") else sum(objectives.values())
            self._zero_grad()
            self._backward(loss)
            self.grad_scaler.unscale_(self.optimizer)
            reduce_gradients(self.optimizer, self.flat_gradients)  # a14
            self.hook.pre_optim(self.optimizer)
            self.grad_scaler.step(self.optimizer)
            self.grad_scaler.update()
            self.hook.post_optim()
            self.record(**objectives)
        self.hook.post_objective(metadata, batch)

    def load_state_dict(self, state_dict: dict[str, Any]):
        super().load_state_dict(state_dict)
        if self.flat_optimizer is not None:  # torch swapped the optimizer's state tensors: fold them back in
            self.flat_optimizer.adopt_state()

    def set_iteration(self, iteration: int):
        if iteration != self.iteration:
            super().set_iteration(iteration)
            self.hook.apply_schedule(self.iteration)

    def resize_buffer(self, capacity: int):
        if self.buffer_capacity != capacity:
            self.buffer_capacity = capacity
            self.buffer.resize(capacity)
