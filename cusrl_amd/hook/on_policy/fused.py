"""One-launch PPO objective behind the per-hook plugin API.

The reference evaluates the PPO objective as four independent hooks — ``ValueLoss`` (value.py:121-137),
``OnPolicyPreparation`` (common.py:29-43), ``PpoSurrogateLoss`` (ppo.py:50-55), ``EntropyLoss`` (ppo.py:82-84) —
each a chain of small torch ops, summed in ``ActorCritic._train_step`` (actor_critic.py:309) and differentiated
by autograd op by op.  On MI355X that is ~40 launch-bound kernels per minibatch over 24 576 x 12 floats.

Here the four hooks keep their names, constructors and outputs, but when the composition is the stock one
they only *register* their term with a per-step :class:`FusedPpoObjective`; once the last hook has run, one
``cusrl_ppo_loss_fwd_bwd`` launch produces the three weighted losses, the per-sample log-prob / entropy /
ratios the other hooks and metrics read, AND d(loss)/d(mean, std, value) — autograd then continues from the
actor / critic heads.

A composition that contains the stock four plus FURTHER hooks defining ``objective`` ("split" mode, round 4) keeps the
one launch for the stock terms and additionally publishes the policy terms the reference's ``OnPolicyPreparation`` leaves
in the batch — ``curr_action_logp``, ``curr_entropy``, ``action_logp_ratio``, ``action_prob_ratio`` — as DIFFERENTIABLE
tensors right when that hook runs, from one more HIP launch (``cusrl_policy_terms_fwd``; its backward,
``cusrl_policy_terms_bwd``, is only launched if some hook's loss actually reaches them).  A hook that REPLACES one of these
batch entries takes the stock term reading it (surrogate / entropy) out of the fused launch: that term is then evaluated
from the replaced tensor with the reference's formula.  Only compositions without the stock four in order, non-Gaussian /
non-categorical policies and CPU agents run hook by hook.
"""

from __future__ import annotations

from typing import Any

import torch

from cusrl_amd import ops

__all__ = ["FusedPpoObjective"]


def _side_outputs(out, deferred, like):
    """(total, losses) of a launch: the kernel's own scalars, or — deferred finalize — placeholders nobody reads: the
    backward runs with a unit gradient and the values reach the metrics through ``ops.DeferredLoss``."""
    if deferred is None:
        losses = out["losses"]
        return losses[6], losses  # total = (value + surrogate) + entropy, summed by the kernel
    return torch.empty((), dtype=torch.float32, device=like.device), torch.empty(0, dtype=torch.float32, device=like.device)


class _FusedPpoFunction(torch.autograd.Function):
    """total = value_loss + surrogate_loss + entropy_loss; gradients precomputed by the forward kernel.
    ``curr_value = ret = None``: the launch carries no value term (:class:`_ValueTermFunction` evaluated it on the critic's
    stream): total = surrogate_loss + entropy_loss."""

    @staticmethod
    def forward(ctx, mean, std, curr_value, advantage, old_logp, action, ret, old_value, clip, value_clip,
                w_sur, w_val, w_ent, unit_grad, deferred):
        out = ops.ppo_loss_fwd_bwd(
            advantage, old_logp, action, mean, std, ret, curr_value, old_value,
            clip=clip, value_clip=value_clip, w_sur=w_sur, w_val=w_val, w_ent=w_ent, want_grads=True, deferred=deferred,
        )
        d_std = out["d_std"]
        ctx.deferred_std = None
        saved = [out["d_mean"]]
        if isinstance(d_std, ops.DeferredColumns):
            # the blocks' column sums of d_std: `std` is the parameter itself here, so they go straight to the flat
            # gradient assembly (backward) instead of through a reduction launch
            ctx.deferred_std, ctx.std_key = d_std, std.data_ptr()
        else:
            saved.append(d_std)
        ctx.has_value = curr_value is not None
        if ctx.has_value:
            saved.append(out["d_value"])
        ctx.save_for_backward(*saved)
        # the five side outputs never receive a gradient; without this autograd would materialise a zero tensor for
        # each of them on every backward (5 fill launches per minibatch)
        ctx.set_materialize_grads(False)
        ctx.unit_grad = unit_grad
        ctx.shapes = (mean.shape, std.shape, curr_value.shape if ctx.has_value else None)
        total, losses = _side_outputs(out, deferred, mean)
        side = (losses, out["logp"], out["entropy"], out["logp_ratio"], out["ratio"])
        ctx.mark_non_differentiable(*side)
        return (total, *side)

    @staticmethod
    def backward(ctx, grad_total, *_unused):
        if grad_total is None:
            return (None,) * 15
        shapes = ctx.shapes
        from cusrl_amd.nn import module as nn_module

        saved = list(ctx.saved_tensors)
        d_mean = saved.pop(0)
        d_value = saved.pop() if ctx.has_value else None
        if ctx.deferred_std is not None:
            if not nn_module.is_unit_gradient(grad_total):
                raise RuntimeError("the deferred-finalize form of the fused PPO objective is differentiated with the agent's unit "
                                   "gradient only (captured steps); something rescaled its loss")
            d_std = nn_module._hand_over(nn_module._split_grad_sink, ctx.std_key, ctx.deferred_std)
            return (d_mean.view(shapes[0]), None if d_std is None else d_std.view(shapes[1]),
                    None if d_value is None else d_value.view(shapes[2]), *([None] * 12))
        d_std = saved.pop()
        if not (ctx.unit_grad and nn_module.is_unit_gradient(grad_total)):  # GradScaler, or a caller that rescales the loss
            d_mean, d_std = d_mean * grad_total, d_std * grad_total
            d_value = None if d_value is None else d_value * grad_total
        return (d_mean.view(shapes[0]), d_std.view(shapes[1]), None if d_value is None else d_value.view(shapes[2]),
                *([None] * 12))


class _ValueTermFunction(torch.autograd.Function):
    """value_loss alone (value.py:85-89,121-137), one launch on the CURRENT stream — the critic's branch of a captured minibatch
    step: critic forward -> this -> critic backward never meets the actor's stream inside the step.  ``losses`` =
    (weighted value loss, mean value) — absent (empty) in the deferred-finalize form, like the one-launch objective's."""

    @staticmethod
    def forward(ctx, curr_value, ret, old_value, weight, value_clip, unit_grad, deferred):
        out = ops.value_loss_fwd_bwd(ret, curr_value, old_value, value_clip=value_clip, w_val=weight, deferred=deferred)
        ctx.save_for_backward(out["d_value"])
        ctx.set_materialize_grads(False)
        ctx.unit_grad, ctx.shape = unit_grad, curr_value.shape
        if deferred is None:
            losses = out["losses"]
            total = losses[0]
        else:  # placeholders nobody reads: unit-gradient backward, values through ops.DeferredLoss
            total = torch.empty((), dtype=torch.float32, device=curr_value.device)
            losses = torch.empty(0, dtype=torch.float32, device=curr_value.device)
        ctx.mark_non_differentiable(losses)
        return total, losses

    @staticmethod
    def backward(ctx, grad_total, _unused=None):
        if grad_total is None:
            return (None,) * 7
        from cusrl_amd.nn.module import is_unit_gradient

        (d_value,) = ctx.saved_tensors
        if not (ctx.unit_grad and is_unit_gradient(grad_total)):
            d_value = d_value * grad_total
        return (d_value.view(ctx.shape), *([None] * 6))


class _FusedCategoricalPpoFunction(torch.autograd.Function):
    """The same objective for a one-hot categorical policy: gradients wrt ``logits`` and ``curr_value``."""

    @staticmethod
    def forward(ctx, logits, curr_value, advantage, old_logp, action, ret, old_value, clip, value_clip, w_sur, w_val, w_ent,
                unit_grad, deferred):
        out = ops.ppo_loss_categorical_fwd_bwd(
            advantage, old_logp, action, logits, ret, curr_value, old_value,
            clip=clip, value_clip=value_clip, w_sur=w_sur, w_val=w_val, w_ent=w_ent, want_grads=True, deferred=deferred,
        )
        ctx.has_value = curr_value is not None
        ctx.save_for_backward(out["d_logits"], *((out["d_value"],) if ctx.has_value else ()))
        ctx.set_materialize_grads(False)
        ctx.unit_grad = unit_grad
        ctx.shapes = (logits.shape, curr_value.shape if ctx.has_value else None)
        total, losses = _side_outputs(out, deferred, logits)
        side = (losses, out["logp"], out["entropy"], out["logp_ratio"], out["ratio"])
        ctx.mark_non_differentiable(*side)
        return (total, *side)

    @staticmethod
    def backward(ctx, grad_total, *_unused):
        if grad_total is None:
            return (None,) * 14
        from cusrl_amd.nn.module import is_unit_gradient

        d_logits, *rest = ctx.saved_tensors
        d_value = rest[0] if rest else None
        if not (ctx.unit_grad and is_unit_gradient(grad_total)):
            d_logits = d_logits * grad_total
            d_value = None if d_value is None else d_value * grad_total
        return (d_logits.view(ctx.shapes[0]), None if d_value is None else d_value.view(ctx.shapes[1]), *([None] * 12))


class _PolicyTermsFunction(torch.autograd.Function):
    """(logp, entropy, logp_ratio, prob_ratio) of a Gaussian policy: one launch forward, one backward."""

    @staticmethod
    def forward(ctx, mean, std, action, old_logp):
        outs = ops.policy_terms_fwd(mean, std, action, old_logp)
        ctx.save_for_backward(mean, std, action, outs[3])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_logp, g_entropy, g_logp_ratio, g_ratio):
        if g_logp is None and g_entropy is None and g_logp_ratio is None and g_ratio is None:
            return None, None, None, None
        mean, std, action, ratio = ctx.saved_tensors
        d_mean, d_std = ops.policy_terms_bwd(mean, std, action, ratio, g_logp, g_entropy, g_logp_ratio, g_ratio)
        return d_mean, d_std, None, None


class _CategoricalTermsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, action, old_logp):
        outs = ops.categorical_terms_fwd(logits, action, old_logp)
        ctx.save_for_backward(logits, action, outs[3])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_logp, g_entropy, g_logp_ratio, g_ratio):
        if g_logp is None and g_entropy is None and g_logp_ratio is None and g_ratio is None:
            return None, None, None
        logits, action, ratio = ctx.saved_tensors
        return ops.categorical_terms_bwd(logits, action, ratio, g_logp, g_entropy, g_logp_ratio, g_ratio), None, None


_TERM_KEYS = ("curr_action_logp", "curr_entropy", "action_logp_ratio", "action_prob_ratio")


def _overrides(hook, method: str) -> bool:
    from cusrl_amd.template.hook import Hook

    return getattr(type(hook), method) is not getattr(Hook, method)


class FusedPpoObjective:
    """Collects the terms of one minibatch step; ``resolve`` turns them into real tensors."""

    def __init__(self, unit_grad: bool, owner=None, split: bool = False):
        self.unit_grad = unit_grad
        self.owner = owner  # the GraphedTrainStep this objective is evaluated for, if any (deferred loss finalize)
        # further hooks define `objective`: the policy terms are published as differentiable tensors when
        # OnPolicyPreparation runs (self.terms), the stock terms still share the one fused launch
        self.split = split
        self.terms: dict[str, torch.Tensor] = {}
        self.value: tuple | None = None
        self.policy: tuple | None = None
        self.surrogate: tuple | None = None
        self.entropy: float | None = None
        self.pending_streams: list = []
        # (total, losses, stream) of the value term when it was evaluated by its own launch on the critic's stream
        self.value_root: tuple | None = None
        self.value_dim = 1

    # ------------------------------------------------------------------ arming
    @staticmethod
    def eligible(composite) -> bool:
        """Exactly the stock composition (one launch, nothing else): see :meth:`mode`."""
        return FusedPpoObjective.mode(composite) == "fused"

    @staticmethod
    def mode(composite) -> str | None:
        """``"fused"``: exactly one each of the four term hooks (exact types, in the reference's order), a Gaussian or
        one-hot categorical policy on a GPU, and no other active hook that defines ``objective``;  ``"split"``: the same
        with further ``objective`` hooks (they get differentiable policy terms from ``cusrl_policy_terms_fwd``);
        ``None``: the hooks evaluate their terms one by one."""
        from cusrl_amd.hook.auxiliary import AdversarialMotionPrior, RandomNetworkDistillation
        from cusrl_amd.hook.mdp.observation import ObservationNormalization
        from cusrl_amd.hook.on_policy.advantage import AdvantageNormalization, AdvantageReduction
        from cusrl_amd.hook.on_policy.common import OnPolicyPreparation
        from cusrl_amd.hook.on_policy.gae import GeneralizedAdvantageEstimation
        from cusrl_amd.hook.on_policy.lr_schedule import MiniBatchWiseLRSchedule
        from cusrl_amd.hook.on_policy.ppo import EntropyLoss, PpoSurrogateLoss
        from cusrl_amd.hook.on_policy.value import ValueLoss

        agent = composite.agent
        distribution = getattr(getattr(agent, "actor", None), "distribution", None)
        supported = getattr(distribution, "is_normal", False) or getattr(distribution, "is_categorical", False)
        if not supported or agent.device.type != "cuda":
            return None
        terms = (ValueLoss, OnPolicyPreparation, PpoSurrogateLoss, EntropyLoss)
        # hooks whose objective neither reads nor differentiates the policy terms: they keep fusion available
        passive = (GeneralizedAdvantageEstimation, AdvantageNormalization, AdvantageReduction, ObservationNormalization,
                   RandomNetworkDistillation, AdversarialMotionPrior, MiniBatchWiseLRSchedule)
        order, extra = [], False
        for hook in composite:
            if not hook.active:
                continue
            if type(hook) in terms:
                order.append(type(hook))
            elif _overrides(hook, "objective") and type(hook) not in passive:
                extra = True
        if tuple(order) != terms:
            return None
        return "split" if extra else "fused"

    @classmethod
    def arm(cls, composite, batch) -> "FusedPpoObjective | None":
        agent = composite.agent
        if getattr(agent, "inference_mode", False) or not torch.is_grad_enabled():
            return None
        if not getattr(agent, "fuse_objective", True):
            return None
        key = tuple(hook.active for hook in composite)
        cached = getattr(composite, "_fusion_cache", None)
        if cached is None or cached[0] != key:
            cached = composite._fusion_cache = (key, cls.mode(composite))
        if cached[1] is None:
            return None
        context = cls(unit_grad=not agent.grad_scaler_enabled, owner=getattr(agent, "_deferred_loss_owner", None),
                      split=cached[1] == "split")
        agent._fused_objective = context
        return context

    @staticmethod
    def disarm(composite):
        composite.agent._fused_objective = None

    @staticmethod
    def current(hook) -> "FusedPpoObjective | None":
        return getattr(getattr(hook, "agent", None), "_fused_objective", None)

    # ------------------------------------------------------------------ term registration (called by the hooks)
    def add_value(self, curr_value, old_value, ret, weight: float, loss_clip: float | None):
        self.value = (curr_value, old_value, ret, weight, loss_clip)
        return {"value_loss": None}

    def evaluate_value(self, curr_value, old_value, ret, weight: float, loss_clip: float | None, stream):
        """The value term NOW, on the current stream (``stream``: the critic's branch) instead of inside the one-launch
        objective: its own root of the step's backward (``Objectives.terms().branch``), so the critic's forward, loss and
        backward form ONE branch of the captured step.  Only in the plain fused mode with a unit-gradient backward."""
        D, B = ret.shape[-1], ret.numel() // max(ret.shape[-1], 1)
        self.value_dim = D
        deferred = self._deferred_value(ret.device, B, D)
        total, losses = _ValueTermFunction.apply(curr_value, ret, old_value, weight, loss_clip, self.unit_grad, deferred)
        self.value = (None, None, None, weight, loss_clip)
        self.value_root = (total, losses, stream, deferred is not None)
        return {"value_loss": None}

    def _deferred_value(self, device, B: int, D: int):
        """The step's :class:`ops.DeferredLoss` for the value term's launch: same conditions as :meth:`_deferred`; the rows are
        created (outside any capture) by whichever of the two launches of the eager warm-up comes first."""
        owner = self.owner
        if owner is None or not self.unit_grad:
            return None
        current = owner.deferred_loss
        if current is None or (current.B, current.D) != (B, D) or not torch.cuda.is_current_stream_capturing():
            return None
        return None if current.blocks > ops.DeferredLoss.MAX_BLOCKS else current

    def join(self, stream):
        """A term was produced on another stream: the loss launch waits for it."""
        self.pending_streams.append(stream)

    def add_policy(self, action_dist, action, old_logp, batch=None):
        self.policy = (action_dist, action, old_logp)
        if not self.split or batch is None:
            return
        # what common.py:38-41 leaves in the batch, differentiable, for the further objective hooks of this composition
        if "logits" in action_dist:
            outs = _CategoricalTermsFunction.apply(action_dist["logits"].float(), action, old_logp)
        else:
            outs = _PolicyTermsFunction.apply(action_dist["mean"].float(), self._std_operand(action_dist["std"]), action, old_logp)
        self.terms = dict(zip(_TERM_KEYS, outs))
        batch.update(self.terms)

    @staticmethod
    def _std_operand(std):
        """The ``[A]`` vector a state-independent std is a broadcast view of (autograd continues from the vector), else
        the matrix itself."""
        row_vector = getattr(std, "_cusrl_row_vector", None)
        if row_vector is not None and std.dim() >= 2 and std.stride(-2) == 0 and row_vector.numel() <= 64:
            return row_vector.float()
        return std.float()

    def owns(self, batch, *keys: str) -> bool:
        """True while the batch entries a stock term reads are still the tensors this objective published (fused mode: they
        do not exist yet — nothing can have replaced them)."""
        return all(batch.get(key) is self.terms.get(key) for key in keys) if self.split else True

    def drop_surrogate(self):
        """A further hook replaced ``action_prob_ratio``: PpoSurrogateLoss evaluates its term from the batch itself."""
        self.surrogate = (None, 0.2, 0.0)

    def drop_entropy(self):
        self.entropy = 0.0

    def add_surrogate(self, advantage, clip_ratio: float, weight: float):
        self.surrogate = (advantage, clip_ratio, weight)
        return {"surrogate_loss": None}

    def add_entropy(self, weight: float):
        self.entropy = weight
        return {"entropy_loss": None}

    # ------------------------------------------------------------------ the launch
    def resolve(self, objectives, batch: dict[str, Any]):
        if any(term is None for term in (self.value, self.policy, self.surrogate, self.entropy)):
            raise RuntimeError("fused PPO objective armed but a term hook did not report; this is a bug")
        curr_value, old_value, ret, w_val, value_clip = self.value  # (all None but the weight: evaluate_value ran the term)
        action_dist, action, old_logp = self.policy
        advantage, clip, w_sur = self.surrogate
        if advantage is None:  # surrogate term dropped (split mode): zero weight, any [B, 1] tensor serves as operand
            advantage = torch.zeros_like(old_logp)
        for stream in self.pending_streams:
            torch.cuda.current_stream().wait_stream(stream)
        self.pending_streams.clear()
        if "logits" in action_dist:  # one-hot categorical policy (discrete action space)
            logits = action_dist["logits"].float()
            deferred = self._deferred(logits.device, logits.shape[-1], self._value_dim(ret), logits.numel() // logits.shape[-1], True, None)
            total, losses, logp, entropy, logp_ratio, ratio = _FusedCategoricalPpoFunction.apply(
                logits, curr_value, advantage, old_logp, action, ret, old_value,
                clip, value_clip, w_sur, w_val, self.entropy, self.unit_grad, deferred,
            )
            self._publish(objectives, batch, advantage, total, losses, logp, entropy, logp_ratio, ratio, deferred, self)
            return
        std = action_dist["std"]
        row_vector = getattr(std, "_cusrl_row_vector", None)
        if row_vector is not None and std.dim() == 2 and std.stride(0) == 0 and ops.ppo_loss_accepts_std_vector(std.shape[-1]):
            std = row_vector  # the [A] vector the batch view repeats: broadcast inside the kernel, d_std comes back as [A]
        mean = action_dist["mean"]
        deferred = self._deferred(mean.device, mean.shape[-1], self._value_dim(ret), mean.numel() // mean.shape[-1], False, std)
        total, losses, logp, entropy, logp_ratio, ratio = _FusedPpoFunction.apply(
            mean, std, curr_value, advantage, old_logp, action, ret, old_value,
            clip, value_clip, w_sur, w_val, self.entropy, self.unit_grad, deferred,
        )
        self._publish(objectives, batch, advantage, total, losses, logp, entropy, logp_ratio, ratio, deferred, self)

    def _value_dim(self, ret) -> int:
        return self.value_dim if ret is None else ret.shape[-1]

    def _deferred(self, device, A: int, D: int, B: int, categorical: bool, std):
        """The :class:`ops.DeferredLoss` of the captured minibatch step this objective belongs to, or None.  Taken only
        while the step is being CAPTURED (its eager warm-up creates the rows, outside any capture), with a unit-gradient
        backward, fp32 inputs, few enough blocks — and, for a std vector, only when that vector is the parameter itself
        (identity bijector: its block column sums then go straight into the parameter's gradient slot)."""
        owner = self.owner
        if owner is None or not self.unit_grad:
            return None
        if std is not None and std.dim() == 1 and not (isinstance(std, torch.nn.Parameter) and std.is_leaf):
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        current = owner.deferred_loss
        if current is None or (current.B, current.A, current.D) != (B, A, D):
            if capturing:
                return None
            current = owner.deferred_loss = ops.DeferredLoss(B, A, D, device, categorical)
        if not capturing or current.blocks > ops.DeferredLoss.MAX_BLOCKS:
            return None
        return current

    @staticmethod
    def _publish(objectives, batch, advantage, total, losses, logp, entropy, logp_ratio, ratio, deferred=None, context=None):
        split = context is not None and context.split
        if not split:  # (split mode: the batch already holds these, differentiable, since OnPolicyPreparation ran)
            batch["curr_action_logp"] = logp
            batch["curr_entropy"] = entropy
            batch["action_logp_ratio"] = logp_ratio
            batch["action_prob_ratio"] = ratio
        dropped = set()
        if split and context.surrogate[0] is None:
            dropped.add("surrogate_loss")
        if split and context.entropy == 0.0 and "entropy_loss" in objectives and objectives["entropy_loss"] is not None:
            dropped.add("entropy_loss")
        fused_keys = tuple(k for k in ("value_loss", "surrogate_loss", "entropy_loss") if k not in dropped)
        objectives.total = total
        objectives.fused_keys = fused_keys
        value_root = context.value_root if context is not None else None
        if value_root is not None:  # the value term is a root of its own, living on the critic's stream
            objectives.branch_root = (value_root[0], value_root[2])
        if deferred is not None:
            # captured step without a finalize launch: the loss values exist as running block sums (ops.DeferredLoss),
            # read once per update by GraphedTrainStep.flush_metrics; agent.record skips the None entries
            batch["_fused_metrics"] = {"deferred": True}
            for key in fused_keys:
                objectives[key] = None
            return
        value_loss, surrogate_loss, entropy_loss, mean_abs_ratio, mean_entropy, mean_value, _ = losses.unbind(0)
        # the means the hooks record after every minibatch, already reduced by the kernel (no extra launches)
        rows = advantage.numel()
        batch["_fused_metrics"] = {"ratio": (mean_abs_ratio, rows), "entropy": (mean_entropy, rows), "value": (mean_value, rows)}
        if value_root is not None:
            if value_root[3]:  # its launch deferred the finalize although this one could not: its sums reach the metrics alone
                value_loss = None
                del batch["_fused_metrics"]["value"]
            else:  # (produced on the critic's stream; read by the metrics after the step's join)
                value_loss, mean_value = value_root[1].unbind(0)
                batch["_fused_metrics"]["value"] = (mean_value, rows)
        for key, value in (("value_loss", value_loss), ("surrogate_loss", surrogate_loss), ("entropy_loss", entropy_loss)):
            if key in fused_keys:
                objectives[key] = value
