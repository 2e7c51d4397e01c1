"""Outer training loop (counterpart of cusrl/template/trainer.py:33-416): rollout until the agent asks for an
update, update, log.  Checkpointing goes through an optional logger object with ``save_checkpoint``; version
dumps, trial folders and console boxes of the reference are out of scope."""

from __future__ import annotations

import os
from collections.abc import Callable, Iterable, Mapping
from typing import Any

import torch

from cusrl_amd.template.agent import Agent, AgentFactory
from cusrl_amd.template.environment import Environment, get_done_indices, update_observation_and_state
from cusrl_amd.utils import distributed
from cusrl_amd.utils.timing import Timer

__all__ = ["EnvironmentStats", "Trainer", "TrainerHook"]


class EnvironmentStats:
    """Per-env episode return / length accumulators and a ring of the last finished episodes (``:33-113``).

    On a GPU env everything is device-resident and ``track(reward, done)`` is ONE HIP launch
    (``cusrl_episode_stats``) with no host round trip; on CPU / NumPy envs it is the reference's torch-op form.
    """

    def __init__(self, num_envs: int, reward_dim: int = 1, buffer_size: int = 100, device=None):
        self.num_envs, self.reward_dim, self.buffer_size = num_envs, reward_dim, buffer_size
        self.device = torch.device("cpu" if device is None else device)
        self.on_device = self.device.type == "cuda"
        zeros = lambda *shape, **kw: torch.zeros(shape, device=self.device, **kw)  # noqa: E731
        self.episode_rew, self.episode_len = zeros(num_envs, reward_dim), zeros(num_envs, 1)
        self.rew_buffer, self.len_buffer = zeros(buffer_size, reward_dim), zeros(buffer_size, 1)
        self.reward = zeros(reward_dim)
        self.total_steps = self.num_steps = 0
        self._num_episodes = 0
        if self.on_device:
            # one block of 8-byte slots: [episode counter x 2 (uint64, double-buffered: see cusrl_episode_stats) | reward sums
            # (fp64)] — it travels to the host as ONE piece of the metrics' staged read, the counters as bit patterns
            self._block = zeros(2 + reward_dim, dtype=torch.float64)
            self._episodes_dev = self._block[:2].view(torch.int64)
            self._parity = 0
            self._reward_sum = self._block[2:]
            self._frame: StatsFrame | None = None

    # ---- device path: one launch per env step
    def track(self, reward: torch.Tensor, done: torch.Tensor):
        from cusrl_amd import ops

        self.total_steps += self.num_envs
        self.num_steps += 1
        ops.episode_stats(reward, done, self.episode_rew, self.episode_len, self.rew_buffer, self.len_buffer,
                          self._episodes_dev, self._reward_sum, self._parity)
        self._parity ^= 1

    def track_fused(self, reward, terminated, truncated, done_out, indices_out, count_out):
        """``cusrl_step_epilogue``: ``done = terminated | truncated``, the statistics of :meth:`track` and the ordered
        finished-env indices + count — one launch for what the reference does in ``ActorCritic.step`` (the OR),
        ``track_step`` / ``track_episode`` and ``get_done_indices``."""
        from cusrl_amd import ops

        ops.step_epilogue(reward, terminated, truncated, done_out, self.episode_rew, self.episode_len, self.rew_buffer,
                          self.len_buffer, self._episodes_dev, self._reward_sum, indices_out, count_out, self._parity)
        self.count_step()

    def defer_fused(self, reward, terminated, truncated, done_out, indices_out, count_out):
        """:meth:`track_fused` whose launch is handed to the buffer append of the same env step (one launch for both,
        ``cusrl_step_epilogue_push``): returns the pending epilogue for ``Buffer.pending_epilogue``; counters as usual."""
        from cusrl_amd import ops

        pending = ops.PendingStepEpilogue(reward, terminated, truncated, done_out, self.episode_rew, self.episode_len, self.rew_buffer,
                                          self.len_buffer, self._episodes_dev, self._reward_sum, indices_out, count_out, self._parity)
        self.count_step()
        return pending

    def count_step(self):
        """Host counters of one fused step (all a hipGraph replay of :meth:`track_fused` leaves to do)."""
        self.total_steps += self.num_envs
        self.num_steps += 1
        self._parity ^= 1

    @property
    def num_episodes(self) -> int:
        return int(self._episodes_dev[self._parity].item()) if self.on_device else self._num_episodes

    # ---- host path (reference form)
    def track_step(self, reward):
        reward = torch.as_tensor(reward, device=self.device)
        self.total_steps += self.num_envs
        self.episode_rew += reward
        self.episode_len += 1
        self.reward += reward.mean(dim=0)
        self.num_steps += 1

    def track_episode(self, indices):
        rew, length = self.episode_rew[indices], self.episode_len[indices]
        count = rew.size(0)
        slot = (torch.arange(count, device=self.device) + self._num_episodes) % self.buffer_size
        self.rew_buffer[slot], self.len_buffer[slot] = rew, length
        self._num_episodes += count
        self.episode_rew[indices] = 0.0
        self.episode_len[indices] = 0.0

    def clear_step_info(self):
        if self.on_device:
            self._reward_sum.zero_()
        else:
            self.reward.zero_()
        self.num_steps = 0

    def freeze(self, metrics=None) -> "StatsFrame":
        """Close the statistics of the rollout that has just been issued: a :class:`StatsFrame` that will give its three means
        and carries its step counters, with the per-rollout accumulators back at zero before the next rollout touches them.
        With device-resident statistics and a ``metrics`` store NOTHING is launched here: the store takes the values along — and
        zeroes the reward sums — when it is next read (``Metrics.pending`` -> :meth:`stage_metrics`: the one snapshot / one
        reset / one host copy per dtype that every captured graph's running sums share), which must happen before the next
        rollout is issued (``agent.update()`` reads the store when it closes; ``Trainer`` checks).  Otherwise the values are
        read right here (a host round trip on a GPU, none on the CPU)."""
        frame = StatsFrame(self.num_steps, self.total_steps, self.num_envs, self.reward_dim, self.buffer_size)
        if self.on_device and metrics is not None:
            frame.parity, self._frame = self._parity, frame
            metrics.pending(self)
            self.num_steps = 0
            return frame
        if self.on_device:
            frame.receive(self._snapshot().tolist())
        else:
            frame.means = (self.mean_episode_length, self.mean_episode_reward, self.mean_step_reward)
        self.clear_step_info()
        return frame

    def stage_metrics(self, metrics):
        """``[(values, tensor to reset or None, callback)]`` for ``Metrics._stage_pending``: the 8-byte block (reset: the reward
        sums), the two rings of finished episodes (no reset) — handed to the frame :meth:`freeze` left waiting."""
        frame, self._frame = self._frame, None
        if frame is None:
            return []
        parts: dict[str, list] = {}

        def collect(name):
            def keep(values):
                parts[name] = values
                if len(parts) == 3:
                    import struct

                    block = parts["block"]
                    episodes = struct.unpack("<q", struct.pack("<d", block[frame.parity]))[0]  # the uint64 counter's bits
                    frame.receive([float(episodes), *block[2:], *parts["rewards"], *parts["lengths"]])
            return keep

        frame.staged = True
        return [(self._block, self._reward_sum, collect("block")), (self.rew_buffer.reshape(-1), None, collect("rewards")),
                (self.len_buffer.reshape(-1), None, collect("lengths"))]

    def close_frame(self, frame: "StatsFrame") -> None:
        """A frame nobody staged (the agent's ``update()`` did not read its metrics): read it now, synchronously — the values
        must leave before the next rollout moves them."""
        if self.on_device and not frame.staged and frame.means is None:
            self._frame = None
            frame.parity = None
            frame.receive(self._snapshot().tolist())
            self._reward_sum.zero_()

    def _snapshot(self) -> torch.Tensor:
        return torch.cat((self._episodes_dev[self._parity].reshape(1).double(), self._reward_sum,
                          self.rew_buffer.reshape(-1).double(), self.len_buffer.reshape(-1).double()))

    def means(self):
        """``(mean_episode_length, mean_episode_reward, mean_step_reward)`` — what the three properties below give — from ONE
        host copy when the statistics live on the device (the properties cost a reduction launch and a synchronising ``item()``
        each, the episode count two more)."""
        if not self.on_device:
            return self.mean_episode_length, self.mean_episode_reward, self.mean_step_reward
        frame = StatsFrame(self.num_steps, self.total_steps, self.num_envs, self.reward_dim, self.buffer_size)
        frame.receive(self._snapshot().tolist())
        return frame.means

    @property
    def mean_step_reward(self):
        if self.on_device:
            total = (self._reward_sum / self.num_envs).float()
        else:
            total = self.reward
        mean = total / self.num_steps if self.num_steps else total
        return tuple(mean.tolist()) if self.reward_dim > 1 else mean.item()

    @property
    def mean_episode_reward(self):
        count = min(self.num_episodes, self.buffer_size)
        if count == 0:
            return 0.0 if self.reward_dim == 1 else (0.0,) * self.reward_dim
        mean = self.rew_buffer[:count].mean(dim=0)
        return tuple(mean.tolist()) if self.reward_dim > 1 else mean.item()

    @property
    def mean_episode_length(self) -> float:
        count = min(self.num_episodes, self.buffer_size)
        return 0.0 if count == 0 else self.len_buffer[:count].mean().item()

    def state_dict(self) -> dict:
        return {"num_episodes": self.num_episodes, "total_steps": self.total_steps}

    def load_state_dict(self, state_dict: dict):
        if state_dict:
            self.total_steps = state_dict["total_steps"]


class StatsFrame:
    """The episode statistics of ONE rollout as its log will report them (``EnvironmentStats.freeze``): the host counters at the
    moment the rollout was closed and — once ``receive`` has been given the device snapshot — the three means."""

    __slots__ = ("num_steps", "total_steps", "num_envs", "reward_dim", "buffer_size", "means", "parity", "staged")

    def __init__(self, num_steps: int, total_steps: int, num_envs: int, reward_dim: int, buffer_size: int):
        self.num_steps, self.total_steps, self.num_envs = num_steps, total_steps, num_envs
        self.reward_dim, self.buffer_size = reward_dim, buffer_size
        self.means = None
        self.parity, self.staged = None, False  # device statistics: which episode counter is current; taken along by a staged read

    def receive(self, flat) -> None:
        """``flat`` = host values of ``EnvironmentStats._snapshot()``: episode count, reward sums, the ring of finished episodes."""
        import numpy as np

        D, R, num_steps = self.reward_dim, self.buffer_size, self.num_steps
        count = min(int(flat[0]), R)
        step = np.asarray(flat[1 : 1 + D], dtype=np.float64) / self.num_envs
        step = (step.astype(np.float32) / np.float32(num_steps)) if num_steps else step.astype(np.float32)
        rewards = np.asarray(flat[1 + D : 1 + D + R * D], dtype=np.float32).reshape(R, D)
        lengths = np.asarray(flat[1 + D + R * D :], dtype=np.float32)
        if count == 0:
            episode_length, episode_reward = 0.0, np.zeros(D, dtype=np.float32)
        else:
            episode_length, episode_reward = float(lengths[:count].mean()), rewards[:count].mean(axis=0)
        scalar = (lambda v: float(v[0])) if D == 1 else (lambda v: tuple(float(x) for x in v))
        self.means = (episode_length, scalar(episode_reward), scalar(step))


class TrainerHook:
    trainer: "Trainer"

    def init(self, trainer: "Trainer"):
        self.trainer = trainer

    def pre_log_info(self, info: dict[str, float]): ...

    def post_update(self): ...


class Trainer:
    Hook = TrainerHook

    def __init__(self, environment: Environment | Callable[[], Environment], agent_factory: AgentFactory,
                 logger_factory: Callable[[], Any] | None = None, num_iterations: int = 1000,
                 init_iteration: int | None = None, checkpoint_interval: int = 50, checkpoint_path: str | None = None,
                 trial_metadata: Mapping[str, Any] | None = None, verbose: bool = True, hooks: Iterable[TrainerHook] = (),
                 pin_host_thread: bool = False):
        self.logger = None if logger_factory is None else logger_factory()
        self.environment = environment if isinstance(environment, Environment) else environment()
        self.agent: Agent = agent_factory.from_environment(self.environment)
        self.hooks = tuple(hooks)
        for hook in self.hooks:
            hook.init(self)
        self.stats = EnvironmentStats(self.environment.num_instances, self.environment.spec.reward_dim,
                                      device=self.environment.spec.device)
        self.verbose = verbose and distributed.is_main_process()
        self.iteration = 0
        if checkpoint_path is not None:
            checkpoint = torch.load(checkpoint_path, map_location=self.agent.device)
            self.agent.load_state_dict(checkpoint["agent"])
            self.environment.load_state_dict(checkpoint["environment"])
            self.stats.load_state_dict(checkpoint.get("stats", {}))
            self.iteration = checkpoint["iteration"]
        if init_iteration is not None:
            self.iteration = init_iteration
        self.agent.set_iteration(self.iteration)
        self.num_iterations = num_iterations
        self.checkpoint_interval = checkpoint_interval
        self.trial_metadata = dict(trial_metadata or {})
        self.timer = Timer(self.agent.device)
        # Perf/*_time come from HIP events; on a GPU every 8th env step is bracketed and scaled (utils/timing.py)
        self.timer_sampling = 8 if self.agent.device.type == "cuda" else 1
        self._last_info: dict[str, float] = {}
        self._pending_log: tuple | None = None
        # the log of an iteration is read, averaged and written AFTER the next rollout has been launched (A/B switch)
        self.pipeline_logs = os.environ.get("CUSRL_PIPELINE_LOGS", "1") != "0"
        # "late" (A/B; measured 0.12 ms SLOWER): the pending log is written behind the update's launches instead of between the
        # rollout's launch and the update's — the host then issues the next rollout while the update still runs
        self._late_flush = os.environ.get("CUSRL_PIPELINE_LOGS") == "late"
        self.host_thread_cpus: list[int] = []
        if pin_host_thread and self.agent.device.type == "cuda":  # extension: NUMA-local placement of the driving thread
            from cusrl_amd.utils.affinity import pin_host_thread as pin

            index = self.agent.device.index if self.agent.device.index is not None else torch.cuda.current_device()
            self.host_thread_cpus = pin(index, slot=distributed.local_rank())
        self._done_counter = None
        self._done_scratch: dict = {}
        self._last_checkpoint_iteration: int | None = None
        # capturable envs (template/environment.py): fixed-shape resets spliced by a device-side count — no host read
        # per step — and, under compile=True, whole env steps replayed from hipGraphs (template/graphs.py)
        env = self.environment
        self._static_resets = bool(getattr(env, "capturable", False)) and self.stats.on_device and not env.spec.autoreset
        self._static_scratch: dict = {}
        self._epilogue_ok: bool | None = None
        self._graphed_rollout = None
        self._rollout_split: tuple[float, float] | None = None
        self.capture_rollout = os.environ.get("CUSRL_CAPTURE_ROLLOUT", "1") != "0"  # A/B switch of the captured env step

    def run_training_loop(self):
        """Checkpoints as the reference does (trainer.py:280-294): once before the loop, every ``checkpoint_interval``
        iterations, and once more after the loop when the last iteration was not saved."""
        self._save_checkpoint()
        try:
            with self.timer.record("environment"):
                observation, state, _ = self.environment.reset(randomize_episode_progress=True)
            while self.iteration < self.num_iterations:
                observation, state = self._rollout_and_update(observation, state)
                self.iteration += 1
                if self.iteration % self.checkpoint_interval == 0:
                    self.flush()
                    self._save_checkpoint()
            self.flush()
            if self.iteration != self._last_checkpoint_iteration:
                self._save_checkpoint()
        finally:
            self._pending_log = None
            self.environment.close()

    def _rollout_and_update(self, observation, state):
        agent, timer = self.agent, self.timer
        graphed = self._rollout_graphs(observation, state)
        if graphed is not None:
            observation, state = self._rollout_captured(graphed, observation, state)
        else:
            observation, state = self._rollout_eager(observation, state)
        if graphed is not None and self._pending_log is not None and not self._late_flush:
            # The log of the PREVIOUS iteration, left pending: read, averaged and written now — with this rollout already enqueued
            # behind that iteration's update, the device goes from one into the other without waiting for the host to read and
            # print.  (Two other placements measured slower on one box, 4.75-4.80 ms this way: behind this iteration's update
            # launches — the host then issues the next rollout and its side-stream draws while the update still runs, 4.88-4.91
            # — and here but with the coming update's permutation draws issued in front of the read, 4.86-4.88: work parked on
            # another hardware queue behind an event costs the running graph time.)
            self.flush()
        metrics = getattr(agent, "metrics", None)
        frame = self.stats.freeze(metrics)  # (device statistics: read with the update's metrics, one host copy for both)
        # Pipelined logging: only around captured rollouts (a host-driven rollout keeps the device waiting anyway), without
        # trainer hooks (their `post_update` follows the log and may read anything) and for agents that stage their summary.
        pipelined = self.pipeline_logs and graphed is not None and not self.hooks and metrics is not None
        if metrics is not None:
            agent.deferred_summary = pipelined
        with timer.record("agent"):
            agent_info = agent.update()
        self.stats.close_frame(frame)  # (no-op when the update's staged read took the statistics along)
        self.flush()  # (a log still pending here: `CUSRL_PIPELINE_LOGS=late` — written behind this iteration's launches, A/B)
        self._pending_log = (agent_info, frame, timer.detach(), self.iteration)
        if not pipelined:
            self.flush()
        return observation, state

    def flush(self) -> None:
        """Write the log an iteration left pending (no-op when there is none): ``_rollout_and_update`` calls it behind the next
        rollout's launch; the training loop before a checkpoint and at its end; ``last_info`` before it answers."""
        pending, self._pending_log = self._pending_log, None
        if pending is None:
            return
        agent_info, frame, timer, iteration = pending
        if hasattr(agent_info, "resolve"):  # a staged summary (utils/metrics.py): wait for its copy, not for the stream
            agent_info = agent_info.resolve()
        self._log_info(agent_info, frame, timer, iteration)
        for hook in self.hooks:
            hook.post_update()

    @property
    def last_info(self) -> dict[str, float]:
        self.flush()
        return self._last_info

    def _rollout_graphs(self, observation, state):
        """The captured-step driver when this rollout can go through it (template/graphs.py GraphedRolloutStep): a
        ``compile=True`` agent, a capturable env, and one host-driven iteration behind us (it allocates the buffer,
        plans the steady-state push, proves that the fused step epilogue takes this env's outputs and measures how a
        step's time splits between agent and environment)."""
        agent = self.agent
        if not (self.capture_rollout and self._static_resets and getattr(agent, "_graphed_act", None) is not None and self._epilogue_ok
                and self._rollout_split is not None and isinstance(observation, torch.Tensor)):
            return None
        if self._graphed_rollout is None:
            from cusrl_amd.template.graphs import GraphedRolloutStep

            self._graphed_rollout = GraphedRolloutStep(self)
        graphed = self._graphed_rollout
        observation_t = observation if observation.device == agent.device else observation.to(agent.device)
        return graphed if graphed.supported(observation_t, state) else None

    def _rollout_captured(self, graphed, observation, state):
        timer = self.timer
        graphed.begin()
        with timer.record("rollout"):
            whole = graphed.run_rollout(observation, state)  # one graph for all steps once every step is captured
            while whole is None:
                observation, state, ready = graphed.run(observation, state)
                if ready:
                    break
            if whole is not None:
                observation, state = whole
        graphed.flush_metrics()
        return observation, state

    def _rollout_eager(self, observation, state):
        agent, env, timer, stats = self.agent, self.environment, self.timer, self.stats
        every = self.timer_sampling  # the four sections of an env step are bracketed on every k-th step only
        while True:
            with timer.record("agent", every):
                action = agent.act(observation, state)
            fused = False
            with timer.record("environment", every):
                next_observation, next_state, reward, terminated, truncated, info = env.step(action)
                if not stats.on_device:
                    stats.track_step(reward)
                elif self._epilogue_applies(reward, terminated, truncated, info):
                    # ONE launch right behind env.step: done flag, episode statistics, ordered finished-env indices and
                    # their count (into pinned host memory).  agent.step's host work (hooks, push) runs while it
                    # executes, so the count is already there when the resets need it — no blocking read-back.
                    fused = True
                    done = torch.empty_like(terminated)
                    if self._static_resets:
                        indices, count = self._static_buffers(terminated)
                        stats.track_fused(reward, terminated, truncated, done, indices, count)
                    else:
                        if self._done_counter is None:
                            from cusrl_amd import ops

                            self._done_counter = ops.HostCounter()
                        indices = self._done_scratch.get("epilogue")
                        if indices is None or indices.numel() != terminated.numel() or indices.device != terminated.device:
                            indices = self._done_scratch["epilogue"] = torch.empty(
                                terminated.numel(), dtype=torch.int64, device=terminated.device)
                        stats.track_fused(reward, terminated, truncated, done, indices, self._done_counter.arm())
                    info = {**info, "done": done}
            with timer.record("agent", every):
                ready = agent.step(next_observation, reward, terminated, truncated, next_state, **info)
            with timer.record("environment", every):
                if fused and self._static_resets:
                    # rows for every index slot, spliced by the kernel up to the device-side count: no host read
                    init_observation, init_state, _ = env.reset_static(indices, count)
                    self._splice_static(next_observation, next_state, indices, count, init_observation, init_state)
                elif fused:
                    if not env.spec.autoreset:
                        done_indices = indices[: self._done_counter.wait()]
                        if done_indices.numel():
                            init_observation, init_state, _ = env.reset(indices=done_indices)
                            next_observation, next_state = self._splice_resets(
                                next_observation, next_state, done_indices, init_observation, init_state)
                elif stats.on_device:
                    # device-resident bookkeeping: one launch, and a host round trip only if the env needs indices
                    done = agent.transition.get("done")
                    if not isinstance(done, torch.Tensor) or done.device != stats.device:
                        done = torch.as_tensor(terminated, device=stats.device) | torch.as_tensor(truncated, device=stats.device)
                    stats.track(torch.as_tensor(reward, device=stats.device), done)
                    if not env.spec.autoreset:
                        # index tensor (allowed by the contract).  A staged copy of the flags to pinned host memory
                        # + event wait + host scan was measured 4 % SLOWER per iteration than this blocking nonzero.
                        done_indices = self._done_indices(done)
                        if done_indices.numel():
                            init_observation, init_state, _ = env.reset(indices=done_indices)
                            next_observation, next_state = self._splice_resets(
                                next_observation, next_state, done_indices, init_observation, init_state)
                elif done_indices := get_done_indices(terminated, truncated):
                    if not env.spec.autoreset:
                        init_observation, init_state, _ = env.reset(indices=done_indices)
                        next_observation, next_state = update_observation_and_state(
                            next_observation, next_state, done_indices, init_observation, init_state)
                    stats.track_episode(done_indices)
            observation, state = next_observation, next_state
            if ready:
                break
        if self._static_resets and getattr(agent, "_graphed_act", None) is not None and self._rollout_split is None:
            # how an env step's device time splits between the agent and the environment: a captured step is one graph
            # whose inside cannot be bracketed, so Perf/agent_time / Perf/environment_time of captured rollouts are
            # the bracketed rollout time in these proportions (one event synchronisation, once)
            agent_time, env_time = timer["agent"], timer["environment"]
            total = agent_time + env_time
            self._rollout_split = (agent_time / total, env_time / total) if total > 0 else (0.5, 0.5)
        return observation, state

    def _static_buffers(self, terminated):
        scratch = self._static_scratch
        if scratch.get("n") != terminated.numel() or scratch.get("device") != terminated.device:
            scratch.update(n=terminated.numel(), device=terminated.device,
                           indices=torch.zeros(terminated.numel(), dtype=torch.int64, device=terminated.device),
                           count=torch.zeros(1, dtype=torch.int32, device=terminated.device))
        return scratch["indices"], scratch["count"]

    @staticmethod
    def _splice_static(observation, state, indices, count, init_observation, init_state):
        """``update_observation_and_state`` (environment.py:365-379) for the capturable reset protocol: row ``k`` of the
        reset tensors goes to env ``indices[k]`` for ``k < count``, with ``count`` read by the kernel."""
        from cusrl_amd import ops

        ops.scatter_rows(init_observation, indices, observation.unsqueeze(0), count)
        if state is not None:
            ops.scatter_rows(init_state, indices, state.unsqueeze(0), count)

    def _epilogue_applies(self, reward, terminated, truncated, info) -> bool:
        ok = self._epilogue_applies_now(reward, terminated, truncated, info)
        self._epilogue_ok = ok if self._epilogue_ok is None else (self._epilogue_ok and ok)
        return ok

    def _epilogue_applies_now(self, reward, terminated, truncated, info) -> bool:
        """Device tensors of the shapes the fused step epilogue takes ([N, D] fp32 reward, [N, 1] bool flags)."""
        stats = self.stats
        limit = getattr(self, "_epilogue_limit", None)
        if limit is None:
            from cusrl_amd import _native

            limit = self._epilogue_limit = int(_native.lib().cusrl_step_epilogue_max_envs())
        return (isinstance(reward, torch.Tensor) and isinstance(terminated, torch.Tensor) and isinstance(truncated, torch.Tensor)
                and reward.device == stats.device and terminated.device == stats.device and truncated.device == stats.device
                and reward.dtype == torch.float32 and terminated.dtype == torch.bool and truncated.dtype == torch.bool
                and reward.shape == (stats.num_envs, stats.reward_dim)  # the kernel indexes the [N, D] accumulators with it
                and terminated.shape == (stats.num_envs, 1) and truncated.shape == terminated.shape
                and reward.is_contiguous() and terminated.is_contiguous() and truncated.is_contiguous()
                and stats.num_envs <= limit and "done" not in info)

    @staticmethod
    def _splice_resets(observation, state, indices, init_observation, init_state):
        """``update_observation_and_state`` (environment.py:365-379) with the row scatter kernel when everything is a
        matching device tensor; torch's ``index_put_`` otherwise."""
        from cusrl_amd import ops

        def spliceable(dst, src):
            return (isinstance(dst, torch.Tensor) and isinstance(src, torch.Tensor) and dst.is_cuda and src.is_cuda
                    and dst.dtype == src.dtype and dst.is_contiguous() and dst.dim() >= 2 and src.shape[1:] == dst.shape[1:]
                    and src.shape[0] == indices.numel())

        if (init_observation.shape != observation.shape and spliceable(observation, init_observation)
                and (state is None or spliceable(state, init_state))):
            ops.assign_rows(observation, indices, init_observation)
            if state is not None:
                ops.assign_rows(state, indices, init_state)
            return observation, state
        return update_observation_and_state(observation, state, indices, init_observation, init_state)

    def _done_indices(self, done: torch.Tensor) -> torch.Tensor:
        """``done.squeeze(-1).nonzero().squeeze(-1)`` (environment.py get_done_indices) on the device: ordered stream
        compaction in two launches whose count lands in pinned host memory that the host polls — instead of torch's
        four-kernel nonzero + copy + stream synchronisation on the step's critical path (the host needs this number
        before it can issue the resets and the next act)."""
        from cusrl_amd import ops

        if done.dtype != torch.bool or not done.is_contiguous():
            return done.reshape(-1).nonzero().reshape(-1)
        if self._done_counter is None:
            self._done_counter = ops.HostCounter()
        # the index buffer is reused every step: its consumers (env.reset, the observation patch) are enqueued before
        # the next step's compaction overwrites it
        indices, _ = ops.compact_flags(done, count_out=self._done_counter.arm(), scratch=self._done_scratch)
        return indices[: self._done_counter.wait()]

    def _save_checkpoint(self):
        if self.logger is None or not distributed.is_main_process():
            return
        self.logger.save_checkpoint(
            {"agent": self.agent.state_dict(), "environment": self.environment.state_dict(),
             "iteration": self.iteration, "stats": self.stats.state_dict()}, iteration=self.iteration)
        self._last_checkpoint_iteration = self.iteration

    def _log_info(self, info: dict[str, float], frame: StatsFrame, timer: Timer, iteration: int):
        """The log of iteration ``iteration`` (trainer.py:370-416) from what that iteration left behind: its agent metrics, the
        frame of its rollout's statistics and the timer sections recorded during it."""
        for key, value in self.environment.get_metrics().items():
            info[f"Environment/{key}"] = value
        info["Metric/episode_length"], episode_reward, step_reward = frame.means
        if isinstance(episode_reward, tuple):
            info.update({f"Metric/episode_reward.{i}": v for i, v in enumerate(episode_reward)})
            info.update({f"Metric/reward.{i}": v for i, v in enumerate(step_reward)})
        else:
            info["Metric/episode_reward"], info["Metric/reward"] = episode_reward, step_reward
        rollout = timer["rollout"]  # captured rollouts: one bracket, split as measured on the host-driven iteration
        split = self._rollout_split or (0.5, 0.5)
        info["Perf/environment_time"] = timer["environment"] + rollout * split[1]
        info["Perf/agent_time"] = timer["agent"] + rollout * split[0]
        info = distributed.average_dict(info)
        world = distributed.world_size()
        steps = frame.num_steps * self.environment.num_instances * world
        info["Perf/environment_step"] = frame.total_steps * world
        info["Perf/environment_fps"] = steps / max(info["Perf/environment_time"], 1e-12)
        info["Perf/agent_fps"] = steps / max(info["Perf/agent_time"], 1e-12)
        for hook in self.hooks:
            hook.pre_log_info(info)
        if self.logger is not None:
            self.logger.log(info, iteration + 1)
        if self.verbose:
            print(f"[iteration {iteration + 1}/{self.num_iterations}] episode_len={info['Metric/episode_length']:.2f} "
                  f"reward={info['Metric/reward'] if 'Metric/reward' in info else float('nan'):.4f} "
                  f"env_time={info['Perf/environment_time']:.4f}s agent_time={info['Perf/agent_time']:.4f}s")
        self._last_info = info
