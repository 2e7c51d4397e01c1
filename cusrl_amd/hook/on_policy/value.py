"""Critic bootstrap target and value loss (replaces cusrl/hook/on_policy/value.py:14-144).

``ValueComputation.pre_update`` builds ``next_value`` with ONE HIP launch for the shift / last-step / terminated
logic and an ordered on-device compaction of the truncated slots; only the critic GEMMs stay in torch.  The single
host read (the truncated count) replaces the reference's ``truncated.any()`` sync (value.py:71).
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from cusrl_amd import ops
from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
from cusrl_amd.template.buffer import Buffer
from cusrl_amd.template.hook import Hook
from cusrl_amd.utils.misc import get_first
from cusrl_amd.utils.nest import map_nested

__all__ = ["ValueComputation", "ValueLoss"]


class ValueComputation(Hook):
    """``defer_value`` (extension; ``None`` = automatic): a feed-forward critic does not have to run inside every env
    step — its parameters do not change during the rollout and the buffer keeps exactly the states it would be fed —
    so the ``value`` field is filled at ``pre_update`` by ONE critic pass over the whole ``[T*N]`` buffer instead of T
    passes over ``[N]`` rows (three launch-bound GEMMs fewer on every env step's critical path; same numbers up to
    GEMM summation order).  Recurrent critics carry memory from step to step and keep the reference's per-step form.
    While deferred, ``transition`` carries no ``value`` / ``critic_memory`` / ``next_critic_memory`` during the rollout;
    the automatic mode therefore only defers when all active hooks are stock ones (a user hook may read them)."""

    def __init__(self, *, termination_value: float = 0.0, bootstrap_truncated_states: bool = True,
                 defer_value: bool | None = None):
        super().__init__()
        self.termination_value = termination_value
        self.bootstrap_truncated_states = bootstrap_truncated_states
        self.defer_value = defer_value
        self._critic_memory = None
        self._value_pending = False
        self._auto_defer: bool | None = None
        self._replay_scratch: dict | None = None

    def init(self):
        if self.agent.environment_spec.final_state_is_missing:
            self.bootstrap_truncated_states = False

    def _deferred(self) -> bool:
        critic = self.agent.critic
        if getattr(critic, "is_recurrent", False) or self.agent.inference_mode:
            return False
        if self.defer_value is not None:
            return bool(self.defer_value)
        # automatic: only when every active hook is one of this package's (none of them reads transition["value"]
        # during the rollout).  A user-defined hook may read it in post_act / post_step as the reference allows, so
        # its presence keeps the per-step critic pass and the reference's transition contract.
        if self._auto_defer is None:
            self._auto_defer = self.agent.device.type == "cuda" and all(
                type(hook).__module__.startswith("cusrl_amd.") for hook in self.agent.hook if hook.active)
        return self._auto_defer

    def post_act(self, transition):
        if self._deferred():
            return
        state = get_first(transition, "state", "observation")
        with self.agent.autocast():
            value, next_memory = self.agent.critic(state, memory=self._critic_memory, sequential=False)
        transition["value"] = value
        transition["critic_memory"] = self._critic_memory
        transition["next_critic_memory"] = next_memory
        self._critic_memory = next_memory

    @property
    def post_step_device_free(self) -> bool:
        return self._deferred()  # deferred: post_step only sets a host flag

    def post_step(self, transition):
        if self._deferred():  # (post_act is replayed from a hipGraph under compile=True; this hook always runs)
            self._value_pending = True
            return
        self.agent.critic.reset_memory(self._critic_memory, transition["done"])

    def on_replay(self, phase):
        if phase == "step" and self._deferred():  # the host half of post_step
            self._value_pending = True

    @torch.no_grad()
    def pre_update(self, buffer: Buffer):
        if self._value_pending and self._replayable(buffer):
            self._value_pending = False
            return self._pre_update_replayed(buffer)
        critic = self.agent.critic
        if self._value_pending:  # deferred: the whole rollout's values from one critic pass
            self._value_pending = False
            state: Tensor = get_first(buffer, "state", "observation")
            with self.agent.autocast():
                flat_value = critic.evaluate(state.flatten(0, 1))
            stacked = flat_value.float().view(*state.shape[:2], -1)
            buffer.field("value", stacked).copy_(stacked)
        value: Tensor = buffer["value"]
        next_value = buffer.field("next_value", value)
        next_state: Tensor = get_first(buffer, "next_state", "next_observation")
        terminated, truncated = buffer["terminated"], buffer["truncated"]
        T, N = value.shape[:2]

        with self.agent.autocast():
            last_value = critic.evaluate(next_state[-1], memory=self._critic_memory)
        counts = ops.next_value(
            value, terminated, truncated, last_value.float(), self.termination_value,
            truncated_uses_own_value=not self.bootstrap_truncated_states, out=next_value,
        )
        if not self.bootstrap_truncated_states:
            return
        slots, count = ops.compact_flags(truncated, counts)
        k = int(count.item())  # the one device->host read of pre_update
        if k == 0:
            return
        slots = slots[:k]
        (truncated_next_state,) = ops.gather_rows([next_state], slots, T, N)
        next_memory = buffer.get("next_critic_memory")
        if next_memory is not None:
            next_memory = map_nested(lambda m: ops.gather_rows([m], slots, T, N)[0], next_memory)
        with self.agent.autocast():
            truncated_next_value = critic.evaluate(truncated_next_state, memory=next_memory)
        ops.scatter_rows(truncated_next_value.float(), slots, next_value)

    # ------------------------------------------------------------------ compile=True: the same work from two hipGraphs
    # Eagerly the block above is ~25 dependent launches (two critic passes, the shift kernel, the compaction, a gather,
    # a scatter) whose device time is a quarter of their host launch time.  With a deferred feed-forward critic every
    # shape in it is static except the number k of truncated slots, so it becomes
    #   head  : compaction of the truncated flags (k stored straight into pinned host memory), value pass over [T*N],
    #           last_value, next_value
    #   tail_b: gather of the first b slots, critic on b rows, scatter limited to k ON THE DEVICE (b = a power-of-two
    #           capacity >= 2k that only ever grows; slots past k are stale but valid rows, computed and dropped)
    # The host polls k (no stream synchronisation) — it is there a few microseconds into the head — and picks the bucket.
    _MIN_BUCKET = 256

    def _replayable(self, buffer: Buffer) -> bool:
        agent = self.agent
        return (bool(getattr(agent, "compile", False)) and getattr(agent, "_graph_stream", None) is not None
                and "value" in buffer.storage and "next_value" in buffer.storage  # created by the first, eager pass
                and buffer.get("next_critic_memory") is None and self._critic_memory is None)

    def _pre_update_replayed(self, buffer: Buffer):
        from cusrl_amd.template.graphs import GraphedRegion

        state: Tensor = get_first(buffer, "state", "observation")
        value = buffer.field("value", buffer.storage["value"])  # host-side bookkeeping of a write to these fields
        next_value = buffer.field("next_value", value)
        next_state: Tensor = get_first(buffer, "next_state", "next_observation")
        terminated, truncated = buffer["terminated"], buffer["truncated"]
        T, N = value.shape[:2]
        scratch = self._replay_scratch
        key = (buffer.layout_version, state.data_ptr(), next_state.data_ptr(), T, N)
        if scratch is None or scratch["key"] != key:  # the regions below close over these very tensors
            # TWO pinned counters used in turn: the host may run up to one iteration ahead of the device (the trainer does not
            # wait for an update to finish before it issues the next rollout and this hook), and the tail of iteration i reads
            # ITS counter on the device when it executes — `arm()` of iteration i + 1 must not touch that one.  The host cannot
            # get two iterations ahead: `wait()` below returns only once the device has reached this iteration's head.
            scratch = self._replay_scratch = {"key": key, "slots": torch.zeros(T * N, dtype=torch.int64, device=value.device),
                                              "counters": (ops.HostCounter(), ops.HostCounter()), "parity": 0, "heads": {}, "tails": {}}
        parity = scratch["parity"] = scratch["parity"] ^ 1
        slots, counter = scratch["slots"], scratch["counters"][parity]
        compaction = scratch.get("compaction")
        if compaction is None:
            from cusrl_amd import _native

            blocks = max(int(_native.lib().cusrl_flag_blocks(T * N)), 1)
            compaction = scratch["compaction"] = {"n": T * N, "device": truncated.device, "indices": slots,
                                                  "counts": torch.empty(blocks, dtype=torch.int32, device=truncated.device)}
        critic, autocast = self.agent.critic, self.agent.autocast

        def head():
            if self.bootstrap_truncated_states:
                # FIRST: the truncated slots and their number k depend on the buffer's flags alone, and k is the one value the
                # host waits for (it picks the tail's capacity) — published a few microseconds into the region, the host issues
                # the tail, the hooks behind this one and the update's graph while the critic pass below keeps the device busy
                # (up to round 5 the compaction came last and took the shift kernel's block counts: the device then sat idle
                # for as long as the host needed to issue all of that)
                ops.compact_flags(truncated, None, counter.tensor, scratch=compaction)
            with autocast():
                flat_value = critic.evaluate(state.flatten(0, 1))
                last_value = critic.evaluate(next_state[-1])
            value.copy_(flat_value.float().view(T, N, -1))
            ops.next_value(value, terminated, truncated, last_value.float(), self.termination_value,
                           truncated_uses_own_value=not self.bootstrap_truncated_states, out=next_value)

        if parity not in scratch["heads"]:  # (one captured head per counter: its address is baked into the compaction launch)
            scratch["heads"][parity] = GraphedRegion(self.agent, head)
        counter.arm()
        scratch["heads"][parity].run(self.bootstrap_truncated_states, self.termination_value)
        if not self.bootstrap_truncated_states:
            return
        if (idle := getattr(self.agent, "run_while_waiting", None)) is not None:
            idle()  # (the region above is running: whatever the agent wants issued meanwhile goes out before the host blocks)
        k = counter.wait()
        if k == 0:
            return
        bucket = scratch.get("bucket", 0)
        if k > bucket:  # capacities only grow, with headroom: a count hovering around a power of two must not re-capture
            bucket = scratch["bucket"] = min(max(self._MIN_BUCKET, 1 << (2 * k - 1).bit_length()), T * N)

        def tail():
            (rows,) = ops.gather_rows([next_state], slots[:bucket], T, N)
            with autocast():
                bootstrap = critic.evaluate(rows)
            ops.scatter_rows(bootstrap.float(), slots[:bucket], next_value, counter.tensor)

        region = scratch["tails"].get((bucket, parity))
        if region is None:
            region = scratch["tails"][(bucket, parity)] = GraphedRegion(self.agent, tail)
        region.run(bucket)


def _clipped_value_loss(value: Tensor, curr_value: Tensor, return_: Tensor, loss_clip: float) -> Tensor:
    clipped = value + (curr_value - value).clamp(-loss_clip, loss_clip)
    return torch.max((curr_value - return_).square(), (clipped - return_).square()).mean()


class ValueLoss(Hook):
    def __init__(self, weight: float = 0.5, loss_clip: float | None = None):
        if weight <= 0:
            raise ValueError("'weight' must be positive")
        if loss_clip is not None and loss_clip <= 0:
            raise ValueError("'loss_clip' must be positive or None")
        super().__init__()
        self.weight: float = weight
        self.loss_clip: float | None = loss_clip
        self.register_mutable("weight")
        self.register_mutable("loss_clip")

    def objective(self, metadata, batch):
        state = get_first(batch, "state", "observation")
        memory, done = batch.get("critic_memory"), batch["done"]
        fused = FusedPpoObjective.current(self)
        # (split mode: a further hook may read `curr_value` on the main stream right behind this one — no second stream)
        branch = getattr(self.agent, "_critic_stream", None) if fused is not None and not fused.split else None
        if branch is not None:
            # inside a minibatch step that is (being) captured: the critic's forward — and, because autograd replays
            # every node on the stream its forward ran on, its backward — goes to a second stream.  The two networks
            # share nothing until their losses are summed, so the captured graph gets two independent branches whose
            # latency-bound small kernels fill the gaps of the other branch's GEMM tails.
            main = torch.cuda.current_stream()
            separate = getattr(self.agent, "separate_value_root", False) and fused.unit_grad
            if separate:  # (read on the main stream, in front of the fork: a lazy batch gathers a field where it is first touched)
                ret, old_value = batch["return"], (batch["value"] if self.loss_clip is not None else None)
            if not (separate and getattr(self.agent, "_batch_on_branch", False)):
                branch.wait_stream(main)  # the gather of `state` was issued on `main`
            # (else: a step inside a whole-update graph whose predecessor left the streams unjoined — the critic's parameters were
            # stepped, and this step's rows gathered, on `branch` itself: it carries on without meeting the main stream)
            with torch.cuda.stream(branch):
                curr_value = self.agent.critic.evaluate(state, memory=memory, done=done)
                if separate:
                    # round 6: the value term right here, on the critic's stream, as its own root of the backward — the
                    # branch runs critic forward -> value term -> critic backward and meets the actor's stream ONCE, in
                    # front of the gradient assembly (ActorCritic._backward), instead of joining for the one-launch
                    # objective and forking again for the backward
                    terms = fused.evaluate_value(curr_value, old_value, ret, self.weight, self.loss_clip, branch)
            curr_value.record_stream(main)
            if separate:
                batch["curr_value"] = curr_value
                return terms
            fused.join(branch)  # the one-launch objective waits for the branch (FusedPpoObjective.resolve)
        else:
            curr_value = self.agent.critic.evaluate(state, memory=memory, done=done)
        batch["curr_value"] = curr_value
        if fused is not None:
            # the behaviour-policy value is only read by the clipped form: not touching it keeps the leaf out of the
            # lazy minibatch (and of the per-slot record, which then fits 256 bytes for the `ppo` buffer)
            old_value = batch["value"] if self.loss_clip is not None else None
            return fused.add_value(curr_value, old_value, batch["return"], self.weight, self.loss_clip)
        if self.loss_clip is None:
            loss = nn.functional.mse_loss(batch["return"], curr_value)
        else:
            loss = _clipped_value_loss(batch["value"], curr_value, batch["return"], self.loss_clip)
        return {"value_loss": loss * self.weight}

    def post_objective(self, metadata, batch):
        curr_value: Tensor = batch["curr_value"]
        if (reduced := batch.get("_fused_metrics")) is not None:
            if not reduced.get("deferred") and "value" in reduced:  # (captured step: read once per update from the kernel's running sums)
                self.agent.metrics.record_reduced("value", *reduced["value"])
        else:
            self.agent.record(value=curr_value.sum(dim=-1))
        if (dim := curr_value.size(-1)) != 1:
            with torch.no_grad():
                self.agent.record(**{f"value.{i}": curr_value[..., i] for i in range(dim)})
