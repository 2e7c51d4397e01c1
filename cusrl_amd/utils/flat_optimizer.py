"""torch.optim.Adam / AdamW stepped as ONE HIP launch over flat buffers.

The optimizer object stays the user's ``torch.optim.Adam`` (same ``param_groups``, same ``state_dict`` layout, so
checkpoints and LR schedules keep working), but

* every parameter's storage is re-pointed into one contiguous fp32 buffer (values preserved),
* ``exp_avg`` / ``exp_avg_sq`` of every parameter are views of two more flat buffers and all ``step`` counters are
  one shared device scalar,
* ``optimizer.step`` becomes ``cusrl_adam_step`` over ``(params, FlatGradients.buffer, exp_avg, exp_avg_sq)``.

``GradientClipping`` cooperates: instead of scaling the gradients itself it leaves the block partials of the squared
norm with :meth:`FlatAdam.defer_clip` and the step applies ``min(max_norm / (norm + 1e-6), 1)`` while streaming the
gradient — "clip + step" is 2 launches instead of 8 (torch: norm, add, reciprocal, mul, clamp, mul, foreach-add,
fused-adam), and the learning rate is read from device memory so hipGraph replays follow LR schedules.
"""

from __future__ import annotations

import functools
import os

import torch

from cusrl_amd import ops
from cusrl_amd.utils.distributed import FlatGradients

__all__ = ["FlatAdam"]

_SUPPORTED = (torch.optim.Adam, torch.optim.AdamW)
_OWN_NORM = "own"  # a pending clip without squared-norm rows: the step launch measures the norm of the flat buffer itself
_GROUP_KEYS = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "decoupled_weight_decay")


class FlatAdam:
    @staticmethod
    def eligible(optimizer, flat_gradients: FlatGradients | None) -> bool:
        """Exactly Adam / AdamW with a single parameter group, plain (non-amsgrad, non-differentiable)
        fp32 parameters on the GPU whose gradients already live in ``flat_gradients``."""
        if type(optimizer) not in _SUPPORTED or flat_gradients is None:
            return False
        groups = optimizer.param_groups
        if len(groups) != 1:  # schedules may move the groups' learning rates apart later on
            return False
        first = groups[0]
        if any(group.get("amsgrad") or group.get("differentiable") for group in groups):
            return False
        if any(isinstance(group["lr"], torch.Tensor) and group["lr"].numel() != 1 for group in groups):
            return False
        if any(any(group.get(key) != first.get(key) for key in _GROUP_KEYS) for group in groups[1:]):
            return False
        params = [p for group in groups for p in group["params"] if p.requires_grad]
        if len(params) != len(flat_gradients.params) or any(a is not b for a, b in zip(params, flat_gradients.params)):
            return False
        return all(p.is_cuda and p.dtype == torch.float32 for p in params)

    def __init__(self, optimizer, flat_gradients: FlatGradients):
        self.optimizer, self.gradients = optimizer, flat_gradients
        self.params = flat_gradients.params
        device = self.params[0].device
        total = flat_gradients.buffer.numel()
        flat = lambda: torch.zeros(total, dtype=torch.float32, device=device)  # noqa: E731
        self.param_buffer, self.exp_avg, self.exp_avg_sq = flat(), flat(), flat()
        self.step_count = torch.zeros(1, dtype=torch.float32, device=device)
        self.lr = torch.zeros(1, dtype=torch.float32, device=device)
        self._lr_value: float | None = None
        self.ticket = torch.zeros(1, dtype=torch.int32, device=device)
        # the two-window step (ActorCritic._backward left the streams unjoined, FlatGradients.split_tail): the critic's window is
        # stepped on the critic's stream with a counter and a ticket of its own; every launch keeps the two counters equal
        self.branch_step_count = torch.zeros(1, dtype=torch.float32, device=device)
        self.branch_ticket = torch.zeros(1, dtype=torch.int32, device=device)
        self.two_window_steps = 0
        # workspaces of cusrl_adam_step_normed, one per window (allocated here: never inside a capture)
        self._norm_workspaces = (ops.adam_norm_workspace(device), ops.adam_norm_workspace(device))
        self._pending_clip: tuple[torch.Tensor, float | None, torch.Tensor] | None = None
        self.metrics = None  # the agent's Metrics (set by the agent): a captured step's tap may take the norm straight from the launch
        self._views: list[tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
        with torch.no_grad():
            for p, offset in zip(self.params, flat_gradients.offsets):  # the same 16-byte-aligned windows as the gradients
                window = slice(offset, offset + p.numel())
                view = self.param_buffer[window].view_as(p)
                view.copy_(p)
                p.data = view  # same values, storage now inside the flat buffer
                self._views.append((view, self.exp_avg[window].view_as(p), self.exp_avg_sq[window].view_as(p)))
        self.adopt_state()
        optimizer.step = self.step  # GradScaler(enabled=False).step(optimizer) and direct calls land here

    # ------------------------------------------------------------------ state aliasing
    def adopt_state(self):
        """(Re-)alias ``optimizer.state`` into the flat buffers, keeping whatever values it currently holds — call
        after ``optimizer.load_state_dict`` (torch replaces the state tensors there)."""
        state = self.optimizer.state
        steps = [float(state[p]["step"]) for p in self.params if p in state and "step" in state[p]]
        with torch.no_grad():
            self.step_count.fill_(max(steps) if steps else 0.0)
            self.branch_step_count.fill_(max(steps) if steps else 0.0)
            for p, (_, exp_avg, exp_avg_sq) in zip(self.params, self._views):
                held = state.get(p, {})
                if "exp_avg" in held and held["exp_avg"] is not exp_avg:
                    exp_avg.copy_(held["exp_avg"])
                    exp_avg_sq.copy_(held["exp_avg_sq"])
                state[p] = {"step": self.step_count[0], "exp_avg": exp_avg, "exp_avg_sq": exp_avg_sq}

    def intact(self) -> bool:
        state = self.optimizer.state
        return self.gradients.intact() and all(
            p.data_ptr() == view.data_ptr() and state.get(p, {}).get("exp_avg") is exp_avg
            for p, (view, exp_avg, _) in zip(self.params, self._views))

    # ------------------------------------------------------------------ the step
    def defer_clip(self, max_norm: float | None) -> torch.Tensor:
        """Take over gradient clipping: launches only the squared-norm partials now; the coefficient is applied by
        the next :meth:`step`.  Returns the device scalar that will hold the pre-clip norm after that step."""
        norm = torch.empty(1, dtype=torch.float32, device=self.lr.device)  # one per step: metrics keep a reference
        tail = self.gradients.split_tail
        if tail is not None and tail.get("reduce"):
            raise RuntimeError("FlatAdam: the windows of an unjoined multi-rank step reached the clipping without having been "
                               "averaged over the ranks (reduce_gradients comes between the backward and pre_optim)")
        if tail is not None:
            # two unjoined window assemblies: their rows, in parameter order (summed as one array by both launches) — or, when
            # the gradients were averaged over the ranks behind the assemblies, the norm each step launch measures itself
            self._pending_clip = (_OWN_NORM if tail.get("reduced") else tail["sumsq"], max_norm, norm)
            return norm[0]
        partials = self.gradients.take_sumsq()  # left behind by the gradient assembly when nothing touched them since
        if partials is None:
            # (several ranks: the all-reduce came in between; or something edited the gradients) — the step launch measures the
            # norm itself (cusrl_adam_step_normed) instead of a squared-norm launch in front of it
            partials = _OWN_NORM
        self._pending_clip = (partials, max_norm, norm)
        return norm[0]

    def discard_pending_clip(self):
        self._pending_clip = None

    def _sync_lr(self, group) -> None:
        lr = group["lr"]
        if isinstance(lr, torch.Tensor):
            if lr.data_ptr() != self.lr.data_ptr():
                self.lr.copy_(lr.reshape(1))
        elif lr != self._lr_value:  # host-side schedules: one tiny fill when the value changes, none otherwise
            self.lr.fill_(float(lr))
            self._lr_value = float(lr)

    def refresh(self):
        """Host-side bookkeeping that must happen OUTSIDE a captured graph before it replays (learning rate)."""
        self._sync_lr(self.optimizer.param_groups[0])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.intact():
            raise RuntimeError("FlatAdam: parameters, gradients or optimizer state were re-allocated behind its back "
                               "(module.to(), optimizer.load_state_dict()): call adopt_state() or rebuild the agent")
        group = self.optimizer.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self._sync_lr(group)
        partials, max_norm, norm = self._pending_clip if self._pending_clip is not None else (None, None, None)
        self._pending_clip = None
        decoupled = bool(group.get("decoupled_weight_decay", False)) or isinstance(self.optimizer, torch.optim.AdamW)
        # torch.optim skips parameters whose gradient is None (no momentum drift, no weight decay); the one-launch step
        # covers the whole flat buffer, so the windows of such parameters are put back afterwards.  (Their step counter
        # is the shared one: a parameter that is unused for a while and then used again sees a bias correction that is
        # ahead of torch's per-parameter counter.)
        kept = [(view, view.clone(), m, m.clone(), v, v.clone())
                for view, m, v in (self._views[i] for i in self.gradients.absent)]
        # a captured step whose metric tap holds this step's norm: the launch adds it to the tap's running sum itself
        tap = getattr(self.metrics, "_tap", None) if norm is not None else None
        slot = tap.slot_of(norm) if tap is not None else None
        hyper = dict(betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"], decoupled=decoupled,
                     maximize=bool(group.get("maximize", False)), max_norm=max_norm)
        tail, self.gradients.split_tail = self.gradients.split_tail, None
        if tail is not None and not kept:
            # The backward left the critic's window assembled on the critic's stream and the others' on this one, unjoined: each
            # window is stepped where its gradients are, behind the OTHER window's assembly (an event edge — the clipping
            # coefficient needs both windows' rows).  No join, no fork: the critic's next forward follows its own step on its own
            # stream, and a stream only ever waits for the other's assembly — a fork / join pair per minibatch step costs
            # ~18 us on this stack, two late-bound event edges ~11 (scripts/probe_graph_fork.py).
            main, branch = torch.cuda.current_stream(), tail["branch"]
            (lo, hi), (blo, bhi) = tail["main_range"], tail["branch_range"]
            if partials is _OWN_NORM:
                # several ranks (reduce_gradients averaged the whole buffer on this stream, behind the critic's assembly): both
                # launches measure the norm of the WHOLE averaged buffer themselves — same split, same norm to the bit
                work_main, work_branch = self._norm_workspaces
                stepper = lambda work: functools.partial(ops.adam_step_normed, norm_grad=self.gradients.buffer, workspace=work)  # noqa: E731
                step_main, step_branch = stepper(work_main), stepper(work_branch)
            else:
                pair = partials if isinstance(partials, tuple) else (partials, None)
                step_main = step_branch = functools.partial(ops.adam_step_window, clip_partials=pair)

            def critic_window():
                with torch.cuda.stream(branch):
                    branch.wait_event(tail["main_assembled"])
                    step_branch(self.param_buffer[blo:bhi], self.gradients.buffer[blo:bhi], self.exp_avg[blo:bhi],
                                self.exp_avg_sq[blo:bhi], self.branch_step_count, self.lr, self.branch_ticket, **hyper)

            def main_window():
                if not tail.get("main_joined"):  # (the all-reduce of a multi-rank step already waited for the critic's assembly)
                    main.wait_event(tail["branch_assembled"])
                step_main(self.param_buffer[lo:hi], self.gradients.buffer[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                          self.step_count, self.lr, self.ticket, norm_out=norm, norm_accumulator=slot, **hyper)

            # Which launch is CAPTURED first decides which chain the graph's executor keeps on the hardware queue of the node in
            # front of them (it follows a node's first edge): behind the all-reduce of a multi-rank step that must be the main
            # stream's — the longer chain, the actor's — so that its step launch follows the collective without a queue hop.
            # (a single process: measured neutral, profiles/r06/experiments/step_order_single_process_ab.txt — left as it was)
            main_first = (tail.get("reduced") and os.environ.get("CUSRL_NORMED_MAIN_FIRST", "1") != "0"
                          or os.environ.get("CUSRL_STEP_MAIN_FIRST") == "1")
            if main_first:
                main_window(), critic_window()
            else:
                critic_window(), main_window()
            self.two_window_steps += 1
            return loss
        if tail is not None:  # (parameters without a gradient are put back below: one launch over everything, behind a join)
            torch.cuda.current_stream().wait_stream(tail["branch"])
        if partials is _OWN_NORM:
            ops.adam_step_normed(self.param_buffer, self.gradients.buffer, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                                 self.ticket, norm_grad=self.gradients.buffer, workspace=self._norm_workspaces[0], norm_out=norm,
                                 norm_accumulator=slot, step_mirror=self.branch_step_count, **hyper)
        else:
            pair = partials if isinstance(partials, tuple) else (partials, None)
            ops.adam_step_window(self.param_buffer, self.gradients.buffer, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                                 self.ticket, clip_partials=pair, norm_out=norm, norm_accumulator=slot,
                                 step_mirror=self.branch_step_count, **hyper)
        for view, view0, m, m0, v, v0 in kept:
            view.copy_(view0), m.copy_(m0), v.copy_(v0)
        return loss
