"""hipGraph capture of the launch-bound loops (what ``compile=True`` means in cusrl_amd).

The reference offers ``compile=True`` = ``torch.compile`` of actor / critic / hook objective
(cusrl/template/actor_critic.py:217-220, hook.py:396-399).  On MI355X these loops are chains of dependent launches of
a few µs each (a dependent kernel costs ≥ 1.5 µs from a graph, several times that when Python issues it), so instead of a
tracing compiler the same knob captures them into HIP graphs and replays them:

* :class:`GraphedRolloutStep` — one per buffer cursor: a WHOLE env step of a capturable env — ``pre_act`` hooks →
  ``actor.explore`` → ``post_act`` hooks → ``env.step`` → fused step epilogue (done flag, episode statistics, ordered
  finished-env ids + device-side count) → ``post_step`` hooks → buffer push → fixed-shape resets spliced by that count →
  the next act input.  No host read on the path; once every step is captured, the whole rollout is ONE graph.
* :class:`GraphedAct` — ``pre_act`` → ``actor.explore`` (sampling included: torch's graph-safe Philox state) →
  ``post_act`` for envs that are not capturable (the env step then stays host-driven).
* :class:`GraphedTrainStep` — one per minibatch slot (and per value of every ``metadata`` key a hook reads,
  :class:`TrackedMetadata`).  The gather of the fields the step's hooks read (``LazyBatch``; plain leaves while the buffer is
  cache-resident, the per-slot record beyond) is NOT part of the graph (round 6): it depends on nothing the previous step
  computes, so it runs ahead on the agent's gather stream into the step's persistent batch tensors while that step is still
  in its backward.  The graph: critic forward → value term → critic backward on a second stream ‖ actor forward →
  surrogate + entropy terms → actor backward (ONE fork, ONE join; loss values as running block sums without a finalize
  launch, ``ops.DeferredLoss``) → one-launch gradient assembly (+ the squared-norm partials of the clip) → gradient
  all-reduce → Adam.
  One process, or RCCL through the C ABI (``cusrl_allreduce_mean`` enqueued on the step's stream — the default route of
  an RCCL job): ONE graph; torch.distributed's collectives (gloo, or the fallback route): two graphs with the eager
  all-reduce between them.
* :class:`GraphedRegion` — the shape-static parts of ``agent.update()`` outside the minibatch loop (critic pass +
  ``next_value`` + compaction, the truncated-slot bootstrap per power-of-two bucket, the statistics pass).

Protocol per graph: first use runs eagerly on the capture stream (warm-up of rocBLAS workspaces and allocator,
and it IS the real step), second use captures and replays, later uses replay; a graph is only replayed while the
host-side values it froze (``capture_signature``: hook mutables, inference / deterministic flags; buffer layout) still
hold.  Hooks must be capture-safe: no host synchronisation and no Python side effects other than ``agent.record``
inside the captured phases — true for every stock hook; a hook that needs more declares the phase eager
(``Hook.eager_phases``), repeats its host half in ``Hook.on_replay``, or — for user hooks overriding ``post_step`` —
simply does not opt in (``rollout_capture_safe``).  ``compile=False`` (the default) runs everything eagerly.
"""

from __future__ import annotations

import gc
import os
from typing import Any

import torch

from cusrl_amd.utils.metrics import MetricTap

__all__ = ["GraphedAct", "GraphedEpochs", "GraphedRegion", "GraphedRolloutStep", "GraphedTrainStep", "capture_signature",
           "collective_phases", "eager_phases"]


def _freeze(value):
    """Hashable snapshot of a mutable hook attribute (floats, tuples, None, small tensors by identity)."""
    if isinstance(value, (list, tuple)):
        return tuple(_freeze(v) for v in value)
    if isinstance(value, dict):
        return tuple(sorted((k, _freeze(v)) for k, v in value.items()))
    try:
        hash(value)
        return value
    except TypeError:
        return id(value)


def capture_signature(agent) -> tuple:
    """Everything a captured region reads on the HOST while it is being captured and would therefore freeze:
    every hook's registered mutable attributes and activity flag, ``agent.deterministic`` and the inference flag.
    A graph is only replayed while the signature it was captured under still holds; a change (``update_attribute``
    through a schedule, ``ObservationNormalization.freeze()``, ``set_inference_mode``) sends the graph back to its
    eager warm-up + re-capture."""
    parts = [agent.deterministic, agent.inference_mode]
    for hook in agent.hook:
        parts.append((hook.name, hook._active, tuple((name, _freeze(getattr(hook, name, None))) for name in sorted(hook._mutable))))
    return tuple(parts)


class TrackedMetadata(dict):
    """The ``metadata`` dict a captured minibatch step hands to the hooks: reads are recorded in ``reads`` (a set the
    agent owns).  A captured step freezes whatever Python did with a value it read — a hook branching on
    ``metadata["epoch_index"]`` would silently keep the capture-time branch on every replay — so ``ActorCritic.update``
    makes every key that was ever read part of the key its step graphs are looked up under: a step whose read values
    differ gets (and replays) its own capture."""

    __slots__ = ("reads",)

    def __init__(self, data, reads: set):
        super().__init__(data)
        self.reads = reads

    def __getitem__(self, key):
        self.reads.add(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self.reads.add(key)
        return dict.get(self, key, default)

    def __contains__(self, key):
        self.reads.add(key)
        return dict.__contains__(self, key)

    def _all(self):
        self.reads.update(dict.keys(self))

    def __iter__(self):
        self._all()
        return dict.__iter__(self)

    def keys(self):
        self._all()
        return dict.keys(self)

    def values(self):
        self._all()
        return dict.values(self)

    def items(self):
        self._all()
        return dict.items(self)

    def copy(self):
        self._all()
        return dict(self)


def collective_phases(agent) -> set[str]:
    """Phases (``"act"`` / ``"objective"``) in which some active hook issues a cross-rank collective.  Collectives stay
    outside captured regions (module docstring), so with several ranks such a phase runs eagerly."""
    from cusrl_amd.utils import distributed

    if not distributed.enabled():
        return set()
    phases: set[str] = set()
    for hook in agent.hook:
        if hook._active:
            phases |= set(hook.collective_phases())
    return phases


def eager_phases(agent) -> set[str]:
    """Phases that must not be captured for this agent: those with a hook-issued collective (above) and those a hook
    declares eager itself (``Hook.eager_phases`` — a host read-back, e.g. MiniBatchWiseLRSchedule)."""
    phases = collective_phases(agent)
    for hook in agent.hook:
        if hook._active:
            phases |= set(hook.eager_phases())
    return phases


class _Capture:
    """A captured region + the metric taps recorded while capturing it."""

    def __init__(self, agent):
        self.agent = agent
        self.graph: torch.cuda.CUDAGraph | None = None
        self.tap_names: list[str] = []
        self.tap_counts: list[int] = []
        self.accumulator: torch.Tensor | None = None
        self.replays = 0
        self.census: dict | None = None

    MAX_TAPS = 256  # (a whole-rollout graph taps every env step's metrics)
    censuses: list[dict] = []  # node census of the most recent captures of the process (tests, scripts/graph_census.py)

    def capture(self, fn, stream: torch.cuda.Stream, pool=None):
        for values, _, record in self.stage_metrics(self.agent.metrics):  # an earlier capture's sums: its accumulator goes away
            self.agent.metrics.defer(values.clone(), record)
        # persistent (allocated outside the capture, so replays do not re-zero it); taps accumulate into it in-graph
        self.accumulator = torch.zeros(self.MAX_TAPS, dtype=torch.float32, device=self.agent.device)
        tap = MetricTap(self.accumulator)
        self.agent.metrics.tap(tap)
        graph = torch.cuda.CUDAGraph(keep_graph=True)  # kept for the node census below; instantiated right behind it
        # Python's cyclic collector must not run inside the capture: finalisers of unrelated garbage (an old agent's
        # pinned host buffers, events, graphs) issue stream operations that are illegal while capturing and abort
        # the process.  Collect now, keep the collector off for the duration of the capture.
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        # In a multi-process job other threads of this process are alive during the capture — torch.distributed's
        # watchdog polling events, RCCL's proxy threads — and in the default "global" mode any of their HIP calls that is
        # not capture-safe invalidates OUR capture.  The capturing thread's own discipline is what matters here.
        from cusrl_amd.utils import distributed

        mode = os.environ.get("CUSRL_CAPTURE_ERROR_MODE") or ("thread_local" if distributed.enabled() else "global")
        try:
            with torch.cuda.graph(graph, stream=stream, pool=pool, capture_error_mode=mode):
                result = fn()
                if tap.values:
                    if len(tap.values) > self.MAX_TAPS:
                        raise RuntimeError(f"more than {self.MAX_TAPS} metrics recorded inside one captured phase")
                    from cusrl_amd import ops

                    # one launch per run of slots whose producer did not add its value itself (torch: stack + add_); the stock
                    # step's only tap — the gradient norm — is added by the Adam launch: no launch here
                    pending = [i for i in range(len(tap.values)) if i not in tap.produced]
                    while pending:
                        run = [pending.pop(0)]
                        while pending and pending[0] == run[-1] + 1:
                            run.append(pending.pop(0))
                        ops.accumulate_scalars_(self.accumulator[run[0]:], [tap.values[i] for i in run])
        finally:
            if gc_was_enabled:
                gc.enable()
            self.agent.metrics.tap(None)
        self.tap_names, self.tap_counts = tap.names, tap.counts
        from cusrl_amd import ops

        self.census = ops.graph_census(graph)
        if self.census["memset"] and os.environ.get("CUSRL_GRAPH_MEMSETS", "replace") != "keep":
            # This stack does not replay memset nodes reliably (scripts/probe_aten_reduce_capture.py) and
            # ATen's split reductions — any `.sum()` / `.mean()` over >= ~1024 rows a hook issues — zero their semaphores
            # with one: every memset node becomes a fill-kernel node with the same edges before the graph is instantiated.
            replaced = ops.graph_replace_memsets(graph)
            self.census = ops.graph_census(graph)
            self.census["memset_replaced"] = replaced
            if self.census["memset"]:
                raise RuntimeError(f"{self.census['memset']} memset nodes left in a captured region after the replacement")
        self.census["region"] = getattr(fn, "__qualname__", "region")
        _Capture.censuses.append(self.census)
        del _Capture.censuses[:-512]
        graph.instantiate()
        self.graph = graph
        return result

    def replay(self):
        self.graph.replay()
        self.replays += 1

    def flush_metrics(self):
        """Hand the accumulated taps over to the agent's metrics: they are read — together with every other capture's, in ONE host
        copy — when the metrics are (``Metrics._stage_pending``); nothing is launched or synchronised here."""
        if self.accumulator is None or self.replays == 0 or not self.tap_names:
            self.replays = 0
            return
        self.agent.metrics.pending(self)

    def stage_metrics(self, metrics):
        """``[(values, tensor to reset, callback)]`` for :meth:`Metrics._stage_pending`: this capture's running sums as of now."""
        if self.accumulator is None or self.replays == 0 or not self.tap_names:
            self.replays = 0
            return []
        names, counts, replays = list(self.tap_names), list(self.tap_counts), self.replays
        self.replays = 0

        def record(sums):
            for name, count, total in zip(names, counts, sums):
                metrics.add_resolved(name, total * count, count * replays)

        return [(self.accumulator[: len(names)], self.accumulator, record)]


class GraphedTrainStep:
    def __init__(self, agent, slot: int, temporal: bool):
        self.agent = agent
        self.slot = slot
        self.temporal = temporal
        self.state = 0  # 0: cold, 1: warmed, 2: captured
        self.stream = agent._graph_stream
        self.static_indices: torch.Tensor | None = None
        self.metadata: dict[str, Any] | None = None
        self.forward_backward = _Capture(agent)
        self.optimize = _Capture(agent)
        self.carry: dict[str, Any] = {}
        # one process: nothing has to run between backward and the optimizer step, so the whole minibatch step is ONE
        # graph; with several ranks the gradient all-reduce (RCCL, eager) sits between two graphs
        from cusrl_amd.utils.distributed import configure_distributed, native_comm

        # ... unless the collectives go through the C ABI (RCCL kernels enqueued on the step's stream): then the
        # all-reduce is captured as a node of the one graph
        self.graph_collectives = configure_distributed() and native_comm() is not None
        self.single_graph = not configure_distributed() or self.graph_collectives
        self.signature: tuple | None = None
        # fields of the batch the step reads: learned by the eager warm-up (which gathers everything), so that the
        # captured gather moves only those leaves (LazyBatch, template/buffer.py)
        self.hot_fields: set[str] = set()
        # running block sums of the fused objective (ops.DeferredLoss): the captured step runs the loss kernel without its
        # finalize launch; created by the eager warm-up, read and reset by flush_metrics
        self.deferred_loss = None
        # fields of the step's minibatch that somebody gathered AHEAD of the step (a whole-epoch graph forks the next step's
        # gather off while this step runs, GraphedEpochs): the step's LazyBatch takes them as they are
        self.preloaded: dict[str, Any] | None = None
        # replays of this step's body from a whole-epoch graph (GraphedEpochs): counted here so that the running loss sums
        # are divided by the right number of evaluations
        self.extra_replays = 0

    def eligible(self) -> bool:
        """False when an objective-phase hook synchronises across ranks (e.g. minibatch-wise advantage normalisation
        with ``synchronize``) or reads a value back to the host: that step runs eagerly instead of baking an RCCL call
        or a stale host decision into a graph."""
        return "objective" not in eager_phases(self.agent)

    def _critic_branch(self) -> bool:
        """The critic as a second stream-branch of the captured step: forced by ``agent.concurrent_critic`` (True / False,
        ``CUSRL_CONCURRENT_CRITIC``), else only where it measured faster (profiles/r05/stream_ab.txt, profiles/r04/configs/
        config5_concurrent_critic_ab.txt) — the stock fused composition at a minibatch of >= 4096 rows: config 2 7.57 -> 6.96 ms
        and config 3 12.14 -> 11.71 ms per iteration WITH the branch; config 5 (RND + AMP chains in the same step)
        16.9 -> 15.7 ms and config 1 (32-row minibatches, launch-bound) 4.25 -> 3.87 ms WITHOUT it.

        History: round 4 withdrew this per-composition choice because the single-stream form was not bit-reproducible.  That
        was not a property of the stream layout: replayed hipGraphs do not execute their MEMSET nodes reliably on this stack,
        and below 4096 rows the bias gradients came from ATen's split ``sum`` whose semaphore is zeroed by one
        (scripts/probe_aten_reduce_capture.py).  No captured region contains a memset node any more
        (``_Capture.capture``), and both layouts are held to a float64 evaluation of every replay
        (tests/test_captured_step_soak.py)."""
        agent = self.agent
        if agent.concurrent_critic is not None:
            return bool(agent.concurrent_critic)
        from cusrl_amd.hook.auxiliary import AdversarialMotionPrior, RandomNetworkDistillation
        from cusrl_amd.hook.on_policy.fused import FusedPpoObjective

        if FusedPpoObjective.mode(agent.hook) != "fused":
            return False
        if any(hook._active and isinstance(hook, (AdversarialMotionPrior, RandomNetworkDistillation)) for hook in agent.hook):
            return False
        rows = 0 if self.static_indices is None else self.static_indices.numel() * (agent.buffer.capacity if self.temporal else 1)
        return rows >= 4096

    def _whole_step(self):
        self._phase_a()
        if self.graph_collectives:
            from cusrl_amd.utils.distributed import reduce_gradients

            reduce_gradients(self.agent.optimizer, self.agent.flat_gradients)  # a14, captured with the step
        self._phase_b()

    # the two phases, written once and used for the eager warm-up, the capture and (implicitly) the replays
    def _phase_a(self):
        agent = self.agent
        batch = agent.buffer.gather_lazy(self.static_indices, self.temporal, self.hot_fields, preloaded=self.preloaded)
        agent.actor.clear_intermediate_repr()
        agent.critic.clear_intermediate_repr()
        agent.hook.pre_objective(self.metadata, batch)
        agent._critic_stream = agent._branch_stream if self._critic_branch() else None
        agent._critic_backward_stream = agent._critic_stream  # (where the critic's autograd nodes will run: _backward's split route)
        agent._deferred_loss_owner = self if agent.defer_loss_finalize else None
        try:
            with agent.autocast():
                objectives = agent.hook.objective(self.metadata, batch)
        finally:
            agent._critic_stream = None
            agent._deferred_loss_owner = None
        if objectives is not None:
            loss = objectives.terms() if agent.flat_gradients is not None else objectives.loss()
            agent._zero_grad()
            agent._backward(loss)
            agent.grad_scaler.unscale_(agent.optimizer)
        self.carry = {"batch": batch, "objectives": objectives}

    def _phase_b(self):
        agent = self.agent
        batch, objectives = self.carry["batch"], self.carry["objectives"]
        if objectives is not None:
            agent.hook.pre_optim(agent.optimizer)
            agent.grad_scaler.step(agent.optimizer)
            agent.grad_scaler.update()
            agent.hook.post_optim()
            agent.record(**objectives)
        agent.hook.post_objective(self.metadata, batch)

    def run(self, metadata: dict[str, Any], indices: torch.Tensor):
        """One minibatch step on the index slice ``indices`` (copied into the step's static index buffer)."""
        from cusrl_amd.utils.distributed import reduce_gradients

        agent = self.agent
        if self.static_indices is None or self.static_indices.shape != indices.shape:
            self.static_indices = torch.empty_like(indices)
            self.state = 0
        self.static_indices.copy_(indices)
        self.metadata = TrackedMetadata(metadata, agent._metadata_reads)
        # the captured gather reads the per-slot record: keep it current (flag check); once the warm-up has learned which
        # fields the step reads, the record holds exactly those (two memory lines per sampled slot)
        agent.buffer.prepare_sampling(self.hot_fields)
        if agent.flat_optimizer is not None:
            agent.flat_optimizer.refresh()  # learning-rate changes reach the captured step through device memory
        signature = (capture_signature(agent), agent.buffer.layout_version)
        if self.state == 2 and signature != self.signature:
            for values, reset, record in self.stage_metrics(agent.metrics):  # (now: the flags below describe the NEXT capture)
                agent.metrics.defer(values.clone(), record)
                reset.zero_()
            if self.deferred_loss is not None:
                self.deferred_loss.armed = self.deferred_loss.value_armed = False  # the new capture records its own launches
            self.state = 1  # a host-side value the capture froze has changed: capture again (the warm-up is still valid)
        self.signature = signature
        if self.state == 0:  # eager on the capture stream: warms rocBLAS / allocator and performs this real step
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._phase_a()
                reduce_gradients(agent.optimizer, agent.flat_gradients)
                self._phase_b()
            torch.cuda.current_stream().wait_stream(self.stream)
            self.carry = {}
            self.state = 1
            return
        if self.state == 1:
            if self.single_graph:
                self.forward_backward.capture(self._whole_step, self.stream, pool=agent._graph_pool)
            else:
                self.forward_backward.capture(self._phase_a, self.stream, pool=agent._graph_pool)
                self.optimize.capture(self._phase_b, self.stream, pool=agent._graph_pool)
            self.carry = {}
            self.state = 2
        self.forward_backward.replay()
        if not self.single_graph:
            reduce_gradients(agent.optimizer, agent.flat_gradients)
            self.optimize.replay()

    def flush_metrics(self, deferred: list | None = None):
        """Hand what the replays accumulated on the device — the captures' metric taps and the objective's running loss sums —
        over to the agent's metrics; read with everything else in one host copy (``Metrics._stage_pending``)."""
        self.agent.metrics.pending(self)

    def stage_metrics(self, metrics):
        replays = self.forward_backward.replays + self.extra_replays
        self.extra_replays = 0
        entries = self.forward_backward.stage_metrics(metrics) + self.optimize.stage_metrics(metrics)
        if self.deferred_loss is not None and replays > 0 and (staged := self.deferred_loss.stage()) is not None:
            values, decode = staged

            def record(host, decode=decode, replays=replays):
                for name, (total, count) in decode(host).items():  # total = sum over the replays of the step's mean
                    metrics.add_resolved(name, total * count, count * replays)

            entries.append((values, values, record))
        return entries


def epoch_graphs_mode() -> str:
    """``"update"`` (default since round 6): ALL minibatch steps of an update replay from one hipGraph; ``"epoch"``
    (``CUSRL_EPOCH_GRAPHS=1``): one graph per epoch; ``"off"`` (``=0``): one graph per minibatch step.  Round 4 measured the epoch
    form neutral (profiles/r04/bench_epoch_graphs_ab.txt) because the step-by-step loop was device-bound then; with one fork and
    one join per step and the gather off the critical path the device needs less time per step than the HOST needs to issue one
    (the replay of a two-branch 22-node graph costs the host ~80 us, the Python around it as much again:
    profiles/r06/experiments/), and what the device gained was lost to replays that arrive late."""
    value = os.environ.get("CUSRL_EPOCH_GRAPHS", "update")
    return {"0": "off", "1": "epoch", "epoch": "epoch"}.get(value, "update")


def epoch_graphs_enabled() -> bool:
    return epoch_graphs_mode() != "off"


class GraphedEpochs:
    """The minibatch steps of ONE epoch back to back as one hipGraph — built on top of the per-step graphs once every step
    of an update replays from its own graph, like the whole-rollout graph on top of the per-step env graphs.

    What it removes per minibatch step is what sits BETWEEN two step graphs — the copy of the index slice into the step's
    static buffer (the slices are read in place from the sampler's persistent ``[E, S]`` permutation buffer,
    ``MiniBatchSampler.draw_epochs``), a replay boundary and ~150 us of host work — and what it adds (round 6) is the
    gather AHEAD of the step: the rows of step k + 1 depend on nothing step k computes, so body k forks their gather off to
    the agent's gather stream before it starts and body k + 1 joins it — inside one graph a fork and a join are edges, not
    host calls.  The gathered leaves live in two persistent sets of batch tensors used in turn (the set body k reads was
    written while body k - 1 ran; the one gather k + 1 writes was last read by body k - 1).  An epoch's graph may start
    as soon as its permutation's event has fired; the later epochs' permutations keep being drawn on the side stream
    meanwhile.  One graph per epoch (not one per update): a captured graph cannot wait for an event recorded outside of it.

    Only with ONE graph per step (single process or the C-ABI collectives) and steps that are all captured under the
    current signature; anything else keeps stepping graph by graph."""

    def __init__(self, agent):
        self.agent = agent
        self.stream: torch.cuda.Stream = agent._graph_stream
        self.epochs: dict[tuple, dict] = {}
        self.signature: tuple | None = None
        self.mode = epoch_graphs_mode()
        self.enabled = self.mode != "off"
        self.replays = 0
        # the gather of the next step's rows ahead of that step.  CUSRL_PREFETCH_GATHER: "tail" (default) — at the tail of the
        # running step's critic branch, which ends before the actor's (no third stream; compositions without the branch issue it
        # behind the step); "side" — forked to a stream of its own at the start of the running step; "0" — inside the step
        self.prefetch = os.environ.get("CUSRL_PREFETCH_GATHER", "tail")
        self.gather_stream = torch.cuda.Stream(device=agent.device) if self.prefetch == "side" else None
        self.stores: dict[tuple, dict[str, torch.Tensor]] = {}

    def _steps_of(self, plan_row, permutations, epoch):
        """The (warm, captured) GraphedTrainStep of every minibatch of this epoch, or None if one is missing."""
        agent, found = self.agent, []
        for metadata, lo, hi in plan_row:
            step = agent._graphed_steps.get(agent._step_key(metadata, hi - lo))
            if step is None or step.state != 2 or not step.single_graph or step.signature != self.signature:
                return None
            found.append((step, metadata, permutations[epoch, lo:hi]))
        return found

    def prepare(self, drawn) -> None:
        """The host half of :meth:`run` that depends on nothing ``pre_update`` computes — the capture signature, the lookup of every
        step's graph, the grouping — done AHEAD of it: the agent calls this while a ``pre_update`` hook waits for the device
        (``ActorCritic.run_while_waiting``), so that between the truncated count's arrival and the update graph's launch the host
        only has the launches themselves left (~100 us of dictionary walks otherwise sat there, with the device running dry).
        Valid for the ``run`` of the same ``drawn`` within the same ``agent.update()``."""
        self._prepared = None
        if not self.enabled:
            return
        agent = self.agent
        permutations, plan = drawn.permutations, drawn.plan
        signature = (capture_signature(agent), agent.buffer.layout_version)
        if signature != self.signature:
            return  # (a change of signature flushes and clears: left to `run`)
        rows = [self._steps_of(plan[epoch], permutations, epoch) for epoch in range(len(plan))]
        if any(row is None for row in rows):
            return
        self._prepared = (drawn, signature, rows)

    def run(self, drawn) -> bool:
        """One update from per-epoch graphs; False = conditions not met (the caller steps graph by graph — and must do so
        WITHOUT consuming the generator again: it iterates the same ``drawn`` permutations)."""
        agent = self.agent
        if not self.enabled:
            return False
        permutations, plan = drawn.permutations, drawn.plan
        prepared, self._prepared = getattr(self, "_prepared", None), None
        if prepared is not None and prepared[0] is drawn and prepared[1][1] == agent.buffer.layout_version:
            _, signature, rows = prepared
        else:
            signature = (capture_signature(agent), agent.buffer.layout_version)
            if signature != self.signature:
                self.flush_metrics()
                self.epochs.clear()
                self.signature = signature
            rows = [self._steps_of(plan[epoch], permutations, epoch) for epoch in range(len(plan))]
            if any(row is None for row in rows):
                return False
        hot = set().union(*(step.hot_fields for row in rows for step, _, _ in row))
        agent.buffer.prepare_sampling(hot)
        if agent.flat_optimizer is not None:
            agent.flat_optimizer.refresh()  # learning-rate changes reach the captured steps through device memory
        if self.mode == "update":
            # ONE graph for the whole update: a replay boundary costs the device the time the host needs to enqueue the next
            # graph's first packets (the first step of every epoch graph ran with its second branch tens of microseconds late),
            # and with every permutation at hand the gather ahead of a step also crosses the epoch boundaries.  All permutations
            # must have been drawn: they are issued while the host waits for pre_update's truncated count
            # (ActorCritic._while_waiting), i.e. they run under pre_update's kernels.
            groups = [(len(rows) - 1, [entry for row in rows for entry in row])]
        else:
            groups = list(enumerate(rows))
        for last_epoch, row in groups:
            key = (last_epoch, len(groups), permutations.data_ptr(), tuple(id(step) for step, _, _ in row))
            entry = self.epochs.get(key)
            drawn.wait(last_epoch)  # the permutations these steps read have been drawn
            if entry is None:
                self._allocate([row])
                entry = self.epochs[key] = {"capture": _Capture(agent)}
                entry["capture"].capture(lambda row=row: self._body(row), self.stream, pool=agent._graph_pool)
            entry["capture"].replay()
            # the next epoch's permutation: issued behind this epoch's launch, so that the draw's dozen small launches run
            # under these steps (the host is free now) instead of in front of the update's first step
            drawn.draw(last_epoch + 1)
            for step, _, _ in row:
                step.extra_replays += 1
        self.replays += 1
        return True

    def _gather(self, step, indices, parity: int):
        """The fields ``step`` is known to read, for the rows ``indices`` names, into the persistent batch tensors of ``parity``."""
        buffer = self.agent.buffer
        names = tuple(name for name in buffer.schema if name in step.hot_fields)
        store = self.stores.setdefault((parity, indices.numel(), step.temporal, names), {})
        return buffer.gather(indices, step.temporal, fields=names, out=store)

    def _body(self, row):
        agent = self.agent
        main, side = torch.cuda.current_stream(), self.gather_stream
        ahead_of_step = self.prefetch in ("side", "tail")
        # (the persistent batch tensors exist by now: `_allocate` ran outside the capture)
        ahead = self._gather(row[0][0], row[0][2], 0) if ahead_of_step else None
        on_branch = branch_open = False
        for k, (step, metadata, indices) in enumerate(row):
            saved = step.static_indices, step.metadata, step.preloaded
            step.static_indices = indices  # read in place
            step.metadata = TrackedMetadata(metadata, agent._metadata_reads)
            step.preloaded = ahead
            following: list = []
            if ahead_of_step and k + 1 < len(row):
                # the next step's rows depend on the buffer and the permutation only
                fetch = lambda k=k: following.append(self._gather(row[k + 1][0], row[k + 1][2], (k + 1) % 2))  # noqa: E731
                if side is not None:  # fork: on a stream of their own while this step runs
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        fetch()
                else:  # at the tail of this step's critic branch (ActorCritic._backward calls it there, on that stream)
                    agent._branch_tail = fetch
            optimizer = agent.flat_optimizer
            two_window_before = optimizer.two_window_steps if optimizer is not None else 0
            agent._unjoined_steps = self.prefetch == "tail"  # (the step may leave its streams unjoined: ActorCritic._backward)
            agent._batch_on_branch = on_branch
            try:
                step._whole_step()
            finally:
                step.static_indices, step.metadata, step.preloaded = saved
                pending, agent._branch_tail = agent._branch_tail, None
                agent._unjoined_steps = agent._batch_on_branch = False
            step.carry = {}
            unjoined = optimizer is not None and optimizer.two_window_steps != two_window_before
            if agent.flat_gradients is not None and agent.flat_gradients.split_tail is not None:
                # (a backward left its streams unjoined and nobody stepped the optimizer: meet here)
                main.wait_stream(agent.flat_gradients.split_tail["branch"])
                agent.flat_gradients.split_tail, unjoined = None, False
            if pending is not None:
                pending()  # no critic branch in this composition: behind the step, on its stream
            elif side is not None and following:
                main.wait_stream(side)  # join, in front of the step that reads them
            ahead = following[0] if following else None
            # the next step's rows were gathered at the tail of this step's critic branch, which then stepped the critic's window
            # itself: the next step's critic forward need not meet the main stream (hook/on_policy/value.py)
            on_branch = unjoined and pending is None and side is None and bool(following)
            branch_open = unjoined
        if branch_open:
            main.wait_stream(agent._branch_stream)  # the streams of the last body meet before the capture ends

    def _allocate(self, rows):
        """The persistent batch tensors of both parities, created OUTSIDE the capture (one throw-away gather per set: an
        allocation made while capturing would belong to the graph's private pool)."""
        if self.prefetch not in ("side", "tail"):
            return
        for row in rows:
            for k, (step, _, indices) in enumerate(row):
                self._gather(step, indices, k % 2)

    def flush_metrics(self):
        for entry in self.epochs.values():
            entry["capture"].flush_metrics()


class GraphedRegion:
    """A shape-static, argument-free slice of ``agent.update()`` outside the minibatch loop (the critic pass over the
    whole buffer in front of GAE, the statistics pass behind the last epoch) replayed from a hipGraph.  Such a slice is
    a dozen-plus dependent launches whose device time is far below their host launch time; one replay costs one launch.

    ``fn`` must touch persistent memory only — buffer storages, parameters, scratch tensors the caller owns — issue no
    host read-back and record no metrics; what it returns stays valid until the next :meth:`run`.  Protocol as for the
    other captures: eager on the capture stream the first time (that run warms rocBLAS and the allocator and IS that
    call), captured on the second, replayed afterwards, re-captured when anything the capture froze on the host changes
    (hook mutables, buffer layout, the caller's ``extra`` key)."""

    def __init__(self, agent, fn):
        self.agent, self.fn = agent, fn
        self.capture = _Capture(agent)
        self.stream: torch.cuda.Stream = agent._graph_stream
        self.state = 0
        self.signature: tuple | None = None
        self.outputs = None

    def run(self, *extra):
        agent = self.agent
        signature = (capture_signature(agent), agent.buffer.layout_version, extra)
        if self.state == 2 and signature != self.signature:
            self.state = 1
        self.signature = signature
        if self.state == 2:
            self.capture.replay()
            return self.outputs
        if self.state == 0:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                outputs = self.fn()
            torch.cuda.current_stream().wait_stream(self.stream)
            self.state = 1
            return outputs
        self.outputs = self.capture.capture(self.fn, self.stream, pool=agent._graph_pool)
        self.state = 2
        self.capture.replay()
        return self.outputs


class GraphedAct:
    """``pre_act`` → ``actor.explore`` → ``post_act`` as one replay; the observation goes through a static buffer
    (this copy replaces the defensive clone the reference makes of every observation, agent.py:257-261)."""

    def __init__(self, agent):
        self.agent = agent
        self.state = 0
        self.stream = agent._graph_stream
        self.static_observation: torch.Tensor | None = None
        self.static_state: torch.Tensor | None = None
        self.capture = _Capture(agent)
        self.transition: dict[str, Any] = {}
        self.signature: tuple | None = None

    def _body(self):
        agent = self.agent
        transition = agent.transition
        transition.clear()
        transition["observation"] = self.static_observation
        if self.static_state is not None:
            transition["state"] = self.static_state
        agent.hook.pre_act(transition)
        with agent.autocast():
            action_dist, (action, action_logp), _ = agent.actor.explore(
                transition["observation"], memory=None, deterministic=agent.deterministic,
                backbone_kwargs={"sequential": False},
            )
        transition.update(action_dist=action_dist, action=action, action_logp=action_logp)
        agent.hook.post_act(transition)

    def supported(self, observation, state) -> bool:
        agent = self.agent
        return (isinstance(observation, torch.Tensor) and observation.is_cuda and not agent.actor.is_recurrent
                and not agent.critic.is_recurrent and not agent.inference_mode
                and getattr(agent.actor.distribution, "capture_safe", True)
                and "act" not in eager_phases(agent))

    def run(self, observation: torch.Tensor, state: torch.Tensor | None) -> torch.Tensor:
        agent = self.agent
        if self.static_observation is None or self.static_observation.shape != observation.shape:
            self.static_observation = torch.empty_like(observation)
            self.static_state = None if state is None else torch.empty_like(state)
            self.state = 0
        self.static_observation.copy_(observation)
        if state is not None:
            self.static_state.copy_(state)
        signature = capture_signature(agent)
        if self.state == 2 and signature != self.signature:
            self.capture.flush_metrics()
            self.state = 1
        self.signature = signature
        if self.state == 2:
            self.capture.replay()
            agent.transition.clear()
            agent.transition.update(self.transition)
            agent.hook.on_replay("act")
            return agent.transition["action"]
        if self.state == 0:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._body()
            torch.cuda.current_stream().wait_stream(self.stream)
            self.state = 1
            return agent.transition["action"]
        self.capture.capture(self._body, self.stream, pool=agent._graph_pool)
        self.transition = dict(agent.transition)
        self.state = 2
        self.capture.replay()
        return agent.transition["action"]


class GraphedRolloutStep:
    """ONE env step of the rollout loop (cusrl/template/trainer.py:296-321) as one hipGraph replay:

        pre_act hooks -> actor.explore -> post_act hooks            (what GraphedAct replays on its own)
        env.step(action)                                            (a `capturable` env: shape-static device work)
        cusrl_step_epilogue: done flag, episode statistics, ordered finished-env ids + their count ON THE DEVICE
        agent.step: post_step hooks, cusrl_buffer_push at this graph's cursor
        env.reset_static(ids, count) -> cusrl_scatter_rows(count)   (reset rows spliced in by a device-side count)
        the spliced observation -> the act input of the next step

    Nothing on this path is read by the host: the trainer issues the replays back to back and the update's graphs behind
    them, the device never waits for Python.  One graph per (buffer cursor, statistics parity) — both are baked into
    kernel arguments — captured under the usual protocol (first use eager on the capture stream, second use capture,
    later uses replay; re-capture when the host-side signature changes).  What a replay skips on the HOST is replayed
    explicitly: the agent's step counter, the buffer cursor, the statistics counters, ``Hook.on_replay("step")``.
    Once every step of a rollout replays from its own graph, the T bodies are captured once more back to back as ONE
    graph (:meth:`run_rollout`): one launch and one pair of generator-state fills per rollout instead of T (a replayed
    graph costs the device >= 1.5 us per dependent node AND per replay boundary); the per-step graphs stay as the path
    for rollouts that are not exactly one pass over the buffer.

    Requirements (checked by :meth:`supported`): ``compile=True`` agent with a capturable act step, an env with
    ``capturable = True`` and no autoreset, device-resident episode statistics, fp32 rewards / bool flags of the shapes
    the fused epilogue takes, and no active hook that declares the ``"step"`` phase eager.  Anything else keeps the
    host-driven loop."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.agent = trainer.agent
        self.stream: torch.cuda.Stream = self.agent._graph_stream
        self.steps: dict[tuple, dict] = {}
        self.signature: tuple | None = None
        # persistent device state shared by every step graph
        env, device = trainer.environment, self.agent.device
        n = env.num_instances
        self.done = torch.zeros(n, 1, dtype=torch.bool, device=device)
        self.indices = torch.zeros(n, dtype=torch.int64, device=device)  # zeros: stale entries must be valid env ids
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.static_observation: torch.Tensor | None = None
        self.static_state: torch.Tensor | None = None
        self.replays = 0
        # ... and, once every step of a rollout has its graph, the WHOLE rollout as one graph (run_rollout)
        self.rollouts: dict[tuple, dict] = {}
        self.rollout_replays = 0
        self.whole_rollouts = os.environ.get("CUSRL_WHOLE_ROLLOUT_GRAPH", "1") != "0"
        self.fuse_epilogue_push = os.environ.get("CUSRL_FUSE_EPILOGUE_PUSH", "1") != "0"  # A/B switch
        # the exploration noise of a whole rollout drawn AHEAD of it (an env that leaves torch's generator alone: the T draws are
        # the generator's only consumers inside the rollout — issued before its launch, on a side stream, they run while the
        # previous update does, and the captured env step is one launch shorter); CUSRL_PREDRAW_NOISE=0: drawn inside the step
        self.predraw_noise = os.environ.get("CUSRL_PREDRAW_NOISE", "1") != "0"
        self._noise: torch.Tensor | None = None
        self._noise_stream: torch.cuda.Stream | None = None
        self._noise_ready: torch.cuda.Event | None = None
        self._noise_read: torch.cuda.Event | None = None
        self.noise_draws = 0

    # ------------------------------------------------------------------ eligibility
    def supported(self, observation, state) -> bool:
        trainer, agent = self.trainer, self.agent
        env = trainer.environment
        act = agent._graphed_act
        if act is None or not getattr(env, "capturable", False) or env.spec.autoreset or not trainer.stats.on_device:
            return False
        if not trainer._static_resets or trainer._epilogue_ok is not True:
            return False
        if not act.supported(observation, state) or "step" in eager_phases(agent):
            return False
        from cusrl_amd.template.hook import Hook

        for hook in agent.hook:
            # post_step / should_update of a hook this package does not know may keep Python state per step (a list of
            # rewards, a counter): such a hook opts in with `rollout_capture_safe = True`, otherwise the loop stays
            # host-driven (pre_act / post_act are already part of the compile=True contract, see GraphedAct)
            custom = type(hook).post_step is not Hook.post_step or type(hook).should_update is not Hook.should_update
            if (hook._active and custom and not type(hook).__module__.startswith("cusrl_amd.")
                    and not getattr(hook, "rollout_capture_safe", False)):
                return False
        return agent.buffer._push_plan is not None  # the steady-state append (one launch, nothing allocated)

    # ------------------------------------------------------------------ the step, written once
    @torch.no_grad()  # like Agent.act / Agent.step, whose decorated entry points this body bypasses
    def _body(self):
        trainer, agent = self.trainer, self.agent
        env, stats, act = trainer.environment, trainer.stats, agent._graphed_act
        act._body()  # reads act.static_observation / static_state
        transition = agent.transition
        next_observation, next_state, reward, terminated, truncated, info = env.step(transition["action"])
        if self.fuse_epilogue_push and not agent.inference_mode and all(h.post_step_device_free for h in agent.hook if h._active):
            # no hook touches the transition on the device between the epilogue and the append: ONE launch for both,
            # issued by buffer.push (which falls back to two launches whenever it cannot take the steady-state path)
            agent.buffer.pending_epilogue = stats.defer_fused(reward, terminated, truncated, self.done, self.indices, self.count)
        else:
            stats.track_fused(reward, terminated, truncated, self.done, self.indices, self.count)
        self.ready = agent.step(next_observation, reward, terminated, truncated, next_state, **{**info, "done": self.done})
        if agent.buffer.pending_epilogue is not None:  # nobody appended (a hook suppressed the push): issue it now
            pending, agent.buffer.pending_epilogue = agent.buffer.pending_epilogue, None
            pending.launch()
        init_observation, init_state, _ = env.reset_static(self.indices, self.count)
        # the next act input = the env's next observation with the reset rows spliced in, one launch (the host-driven
        # loop does the same in two: the in-place row scatter of Trainer._splice_static, then GraphedAct's copy)
        self._advance_input(act.static_observation, next_observation, init_observation)
        if act.static_state is not None:
            self._advance_input(act.static_state, next_state, init_state)

    def _advance_input(self, static, following, init):
        from cusrl_amd import ops

        if (following.dtype == static.dtype and init.dtype == static.dtype and following.shape == static.shape
                and init.shape == static.shape and following.is_contiguous() and init.is_contiguous()):
            ops.splice_rows(following, init, self.indices, self.count, self.done, static)
        else:  # layouts the fused launch does not take: scatter in place, then copy
            ops.scatter_rows(init, self.indices, following.unsqueeze(0), self.count)
            static.copy_(following)

    def _replay_host_effects(self) -> bool:
        """The Python-side effects of one step that a replay does not perform."""
        trainer, agent = self.trainer, self.agent
        trainer.stats.count_step()
        ready = agent.replay_step()
        return ready

    # ------------------------------------------------------------------ driver
    def begin(self):
        """Once per rollout: compare what the captures froze on the host (hook mutables, inference / deterministic flags,
        buffer layout) with the current values; a change sends every step graph back to capture."""
        agent = self.agent
        signature = (capture_signature(agent), agent.buffer.layout_version)
        if signature != self.signature:
            self.flush_metrics()
            for entry in self.steps.values():
                entry["state"] = min(entry["state"], 1)
            self.rollouts.clear()
            self.signature = signature
            agent.actor.noise_shape = None  # (set again by the step bodies that take the fused explore pass under these flags)

    def run(self, observation, state):
        """One env step; returns ``(next_observation, next_state, ready)`` (the static act inputs of the next step)."""
        agent, trainer = self.agent, self.trainer
        act = agent._graphed_act
        if act.static_observation is None or act.static_observation.shape != observation.shape:
            act.static_observation = torch.empty_like(observation)
            act.static_state = None if state is None else torch.empty_like(state)
            act.state = 0
            self.steps.clear()
        if observation is not act.static_observation:
            act.static_observation.copy_(observation)
            if state is not None:
                act.static_state.copy_(state)
        if self.signature is None:
            self.begin()
        key = (agent.buffer.cursor, trainer.stats._parity)
        entry = self.steps.get(key)
        if entry is None:
            entry = self.steps[key] = {"state": 0, "capture": _Capture(agent), "transition": None}
        if entry["state"] == 2:
            entry["capture"].replay()
            agent.transition.clear()
            agent.transition.update(entry["transition"])
            agent.hook.on_replay("act")
            ready = self._replay_host_effects()
            self.replays += 1
            return act.static_observation, act.static_state, ready
        if entry["state"] == 0:  # eager on the capture stream: the real step, and the warm-up of this key
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._body()
            torch.cuda.current_stream().wait_stream(self.stream)
            entry["state"] = 1
            return act.static_observation, act.static_state, self.ready
        # capture: the body's host effects happen once here (they belong to this very step), the replay right behind
        # the capture performs the device work
        entry["capture"].capture(self._body, self.stream, pool=agent._graph_pool)
        entry["transition"] = dict(agent.transition)
        entry["state"] = 2
        entry["capture"].replay()
        return act.static_observation, act.static_state, self.ready

    # ------------------------------------------------------------------ the whole rollout as ONE graph
    def _rollout_length(self) -> int | None:
        """T when this rollout is exactly one pass over the buffer — starts at cursor 0 with a fresh step counter, the
        update comes after ``capacity`` steps, no hook decides ``should_update`` itself — and every one of its steps
        already replays from its own graph under the current signature (so every body is warm); else None."""
        agent, trainer = self.agent, self.trainer
        buffer = agent.buffer
        T = buffer.capacity
        if not self.whole_rollouts or buffer.cursor != 0 or agent.step_index != 0 or agent.num_steps_per_update != T:
            return None
        from cusrl_amd.template.hook import Hook

        if any(hook._active and type(hook).should_update is not Hook.should_update for hook in agent.hook):
            return None
        parity = trainer.stats._parity
        taps = 0
        for t in range(T):
            entry = self.steps.get((t, parity ^ (t & 1)))
            if entry is None or entry["state"] != 2:
                return None
            taps += len(entry["capture"].tap_names)
        if taps > _Capture.MAX_TAPS:
            # one accumulator slot per recorded metric and env step: a rollout that taps more than the accumulator holds keeps
            # its per-step graphs (finding out inside the capture would be too late: the bodies' host effects — buffer
            # cursor, counters — cannot be undone)
            return None
        return T

    def _rollout_body(self, steps: int, noise: torch.Tensor | None = None):
        actor = self.agent.actor
        for t in range(steps):
            if noise is not None:
                actor.pending_noise = noise[t]
            self._body()
            if actor.pending_noise is not None:
                actor.pending_noise = None
                raise RuntimeError("a captured env step did not take the exploration noise drawn ahead for it")

    def _noise_plan(self, steps: int) -> torch.Tensor | None:
        """The persistent ``[T, N, A]`` noise rows of a whole rollout when they may be drawn ahead of it: an env that draws nothing
        from torch's generator (``generator_free``), a stochastic policy whose act step is the fused explore pass (the step bodies
        of the current signature recorded its noise shape), hooks of this package only and none of them drawing inside an env step
        (``Hook.step_draws_random``: AdversarialMotionPrior samples a step's expert transitions in ``post_step``)."""
        agent, env = self.agent, self.trainer.environment
        actor = agent.actor
        shape = getattr(actor, "noise_shape", None)
        if (not self.predraw_noise or not getattr(env, "generator_free", False) or agent.deterministic or shape is None
                or not all(type(hook).__module__.startswith("cusrl_amd.") and not (hook._active and hook.step_draws_random)
                           for hook in agent.hook)):
            return None
        if self._noise is None or tuple(self._noise.shape) != (steps, *shape):
            self._noise = torch.empty((steps, *shape), dtype=torch.float32, device=agent.device)
            self._noise_ready, self._noise_read = torch.cuda.Event(), None
            if self._noise_stream is None:
                from cusrl_amd.utils.streams import side_stream

                self._noise_stream = side_stream(agent.device)
        return self._noise

    def _draw_noise(self, noise: torch.Tensor):
        """The T ``normal_()`` calls of this rollout's act steps — the same calls, in the same order, on tensors of the same
        shape as the steps themselves would issue (nn/actor.py) — on the side stream; the current stream waits for them."""
        main, side = torch.cuda.current_stream(), self._noise_stream
        if self._noise_read is not None:
            side.wait_event(self._noise_read)  # (the previous rollout has read its rows)
        with torch.cuda.stream(side):
            for t in range(noise.shape[0]):
                noise[t].normal_()
            self._noise_ready.record(side)
        main.wait_event(self._noise_ready)
        self.noise_draws += noise.shape[0]

    def run_rollout(self, observation, state):
        """The whole rollout from one replay — the T step bodies captured back to back into ONE graph: one launch, one
        pair of generator-state fills and no replay boundaries instead of T of each.  Returns ``(observation, state)`` for
        the next rollout, or None when the conditions of :meth:`_rollout_length` do not hold (the caller then steps)."""
        steps = self._rollout_length()
        if steps is None:
            return None
        agent, trainer = self.agent, self.trainer
        act = agent._graphed_act
        if observation is not act.static_observation:
            act.static_observation.copy_(observation)
            if state is not None:
                act.static_state.copy_(state)
        noise = self._noise_plan(steps)
        key = (trainer.stats._parity, steps, None if noise is None else noise.data_ptr())
        entry = self.rollouts.get(key)
        if noise is not None:
            self._draw_noise(noise)
        if entry is None:
            # capture: the bodies' host effects (cursor, counters, hooks' host halves) happen here, once, for this very
            # rollout; the replay right behind the capture performs its device work
            entry = self.rollouts[key] = {"capture": _Capture(agent), "transition": None}
            entry["capture"].capture(lambda: self._rollout_body(steps, noise), self.stream, pool=agent._graph_pool)
            entry["transition"] = dict(agent.transition)
            entry["capture"].replay()
            ready = self.ready
        else:
            entry["capture"].replay()
            agent.transition.clear()
            agent.transition.update(entry["transition"])
            ready = False
            for _ in range(steps):
                agent.hook.on_replay("act")
                ready = self._replay_host_effects()
            self.rollout_replays += 1
        if noise is not None:
            if self._noise_read is None:
                self._noise_read = torch.cuda.Event()
            self._noise_read.record(torch.cuda.current_stream())
        if not ready:
            raise RuntimeError("a whole-rollout graph ended without the agent asking for an update")
        return act.static_observation, act.static_state

    def flush_metrics(self):
        for entry in self.steps.values():
            entry["capture"].flush_metrics()
        for entry in self.rollouts.values():
            entry["capture"].flush_metrics()

    @property
    def captured(self) -> int:
        return sum(1 for entry in self.steps.values() if entry["state"] == 2)
