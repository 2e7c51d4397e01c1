#!/usr/bin/env python3
"""Where do two runs of the same seeded training diverge?  (Round 4's open defect — single words of bias-gradient slots of
replayed steps — root-caused in round 5: memset nodes of replayed hipGraphs; logs in
profiles/r04/amp_identity/ and profiles/r05/defect/.  The shipped tree no longer shows it; to REPRODUCE it put the memset nodes
back: ``CUSRL_WIDE_LINEAR_MIN_ROWS=4096 CUSRL_GRAPH_MEMSETS=keep CUSRL_CONCURRENT_CRITIC=0 python scripts/debug_amp_identity.py``.)

Two host-driven runs of the tests/test_captured_rollout.py workload (N = 256, T = 8, 2 epochs x 2 minibatches) in one process;
the flat gradient buffer, the parameters and the index slice are recorded behind every minibatch step and compared.

    CUSRL_CONCURRENT_CRITIC=0 python scripts/debug_amp_identity.py          # AMP composition, single-stream steps
    CUSRL_CONCURRENT_CRITIC=0 DEBUG_KIND=continuous DEBUG_ITERATIONS=8 ...   # the stock composition needs a few more iterations
    DEBUG_INGRAPH=1 ...            # + snapshots inside the captured step: behind the assembly, and of its deferred sources
    DEBUG_DETERMINISTIC=1 ...      # torch.use_deterministic_algorithms(True, warn_only=True): empty tensors are NaN-filled,
                                   # rocBLAS atomics off; =2: the same without the NaN fill
    DEBUG_COMPILE=0 ...            # no hipGraphs at all
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402

DEV = "cuda:0"
if os.environ.get("DEBUG_DETERMINISTIC") in ("1", "2"):
    torch.use_deterministic_algorithms(True, warn_only=True)  # (also: rocBLAS atomics mode "not allowed")
    if os.environ["DEBUG_DETERMINISTIC"] == "2":  # ... without NaN-filling every torch.empty (isolates the atomics mode)
        torch.utils.deterministic.fill_uninitialized_memory = False
cusrl.config.set_device(DEV)


def factory(T):
    if os.environ.get("DEBUG_KIND") == "continuous":
        return cusrl.preset.PpoAgentFactory(num_steps_per_update=T, sampler_epochs=2, sampler_mini_batches=2,
                                            compile=os.environ.get("DEBUG_COMPILE", "1") != "0",
                                            optimizer_kwargs={"capturable": True, "fused": True})
    k = 6
    dataset = torch.randn(4096, 2 * k, device=DEV)
    return cusrl.preset.AmpAgentFactory(amp_dataset_source=dataset, amp_state_indices=slice(k), extrinsic_reward_scale=0.5,
                                        amp_reward_scale=2.0, num_steps_per_update=T, sampler_epochs=2, sampler_mini_batches=2,
                                        compile=os.environ.get("DEBUG_COMPILE", "1") != "0", optimizer_kwargs={"capturable": True, "fused": True})


def run(capture, iterations, T=8, N=256):
    cusrl.set_global_seed(21)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=12, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, factory(T), num_iterations=iterations, verbose=False)
    trainer.capture_rollout = capture
    trainer.run_training_loop()
    torch.cuda.synchronize()
    return trainer


def state(trainer):
    agent = trainer.agent
    out = {f"param/{n}": p.detach().clone() for n, p in agent.named_parameters()}
    out.update({f"buffer/{k}": v.clone() for k, v in agent.buffer.storage.items()})
    for hook in agent.hook:
        for name, module in getattr(hook, "named_modules", lambda: [])():
            pass
        for attr in ("state_rms", "_state_rms", "rms", "transition_rms"):
            m = getattr(hook, attr, None)
            if m is not None and hasattr(m, "mean"):
                out[f"{type(hook).__name__}.{attr}.mean"] = m.mean.clone()
                out[f"{type(hook).__name__}.{attr}.var"] = m.var.clone()
                out[f"{type(hook).__name__}.{attr}.count"] = m._count.clone()
    for i, group in enumerate(agent.optimizer.state.values()):
        for k, v in group.items():
            if torch.is_tensor(v):
                out[f"optim/{i}/{k}"] = v.clone()
    out["cuda_rng"] = torch.cuda.get_rng_state(0)[-16:].clone()
    return out


def poison():
    """Fill the caching allocator's free blocks with NaN so that a read of uninitialised memory shows."""
    blocks = [torch.full((1 << 26,), float("nan"), device=DEV) for _ in range(8)]
    small = [torch.full((n,), float("nan"), device=DEV) for n in (16, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20) for _ in range(64)]
    del blocks, small
    torch.cuda.synchronize()


def summarize(tag, a, b):
    bad = [k for k in a if k in b and a[k].shape == b[k].shape and not torch.equal(a[k], b[k])]
    nan = [k for k in b if b[k].is_floating_point() and torch.isnan(b[k]).any()]
    groups = {}
    for k in bad:
        groups[k.split("/")[0]] = groups.get(k.split("/")[0], 0) + 1
    print(f"{tag}: differ {groups} buffers {[k for k in bad if k.startswith('buffer/')]} other {[k for k in bad if not k.startswith(('buffer/', 'param/', 'optim/'))]}; NaN {nan[:6]}", flush=True)


class Snap(cusrl.Trainer.Hook):
    def __init__(self):
        self.snaps = []

    def post_update(self):
        torch.cuda.synchronize()
        self.snaps.append(state(self.trainer))


def run_snap(capture, iterations, T=8, N=256):
    cusrl.set_global_seed(21)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=12, action_dim=4, device=DEV)
    hook = Snap()
    trainer = cusrl.Trainer(env, factory(T), num_iterations=iterations, verbose=False, hooks=[hook])
    trainer.capture_rollout = capture
    trainer.run_training_loop()
    return hook.snaps


from cusrl_amd.template import graphs as _graphs  # noqa: E402

RECORD = []
_orig_run = _graphs.GraphedTrainStep.run


def _wrapped(self, metadata, indices, *a, **k):
    out = _orig_run(self, metadata, indices, *a, **k)
    torch.cuda.synchronize()
    agent = self.agent
    carry = getattr(self, "carry", None) or {}
    RECORD.append({"state": self.state, "grad": agent.flat_gradients.buffer.clone(), "indices": indices.clone(),
                   "params": torch.cat([p.detach().reshape(-1) for p in agent.flat_gradients.params])})
    return out


_graphs.GraphedTrainStep.run = _wrapped

# DEBUG_INGRAPH=1: persistent snapshots taken INSIDE the step (captured with it): the flat buffer right behind the assembly
# launch, and every deferred partial buffer the assembly is about to reduce — tells whether the foreign words are already in
# the assembly's sources, appear in its output, or are written behind it (Adam, taps).  Also prints, once per captured step,
# the address range of every deferred partial buffer.
if os.environ.get("DEBUG_INGRAPH") == "1":
    from cusrl_amd import ops as _ops
    from cusrl_amd.utils import distributed as _dist

    _orig_assemble = _dist.FlatGradients.assemble

    def _assemble(self, grads, split_slabs=None):
        snaps = self.__dict__.setdefault("_debug_sources", {})
        capturing = torch.cuda.is_current_stream_capturing()
        for key, slabs in (split_slabs or {}).items():
            if isinstance(slabs, _ops.DeferredColumns):
                tag = (key, slabs.column)
                if tag not in snaps:
                    if capturing:
                        continue  # (created by the eager warm-up; a shape first seen while capturing is skipped)
                    snaps[tag] = torch.empty_like(slabs.partials)
                if snaps[tag].shape == slabs.partials.shape:
                    snaps[tag].copy_(slabs.partials)
                if capturing:
                    lo = slabs.partials.data_ptr()
                    print(f"      capture: partials of param@{key:#x} column {slabs.column}: [{lo:#x}, {lo + slabs.partials.numel() * 4:#x}) "
                          f"{tuple(slabs.partials.shape)} splits {slabs.splits}", flush=True)
        _orig_assemble(self, grads, split_slabs)
        if "_debug_after_assemble" not in self.__dict__:
            if capturing:
                return
            self._debug_after_assemble = torch.empty_like(self.buffer)
        self._debug_after_assemble.copy_(self.buffer)

    _dist.FlatGradients.assemble = _assemble
    _prev_wrapped = _graphs.GraphedTrainStep.run

    def _wrapped_ingraph(self, metadata, indices, *a, **k):
        out = _prev_wrapped(self, metadata, indices, *a, **k)
        flat = self.agent.flat_gradients
        RECORD[-1]["after_assemble"] = flat.__dict__.get("_debug_after_assemble", flat.buffer).clone()
        RECORD[-1]["sources"] = {tag: t.clone() for tag, t in flat.__dict__.get("_debug_sources", {}).items()}
        return out

    _graphs.GraphedTrainStep.run = _wrapped_ingraph


def run_rec(n):
    RECORD.clear()
    cusrl.set_global_seed(21)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=256, observation_dim=12, action_dim=4, device=DEV)
    trainer = cusrl.Trainer(env, factory(8), num_iterations=n, verbose=False)
    trainer.capture_rollout = False
    trainer.run_training_loop()
    flat = trainer.agent.flat_gradients
    names = {id(p): n for n, p in trainer.agent.named_parameters()}
    segments = []
    offset = 0
    for p in flat.params:
        segments.append((names.get(id(p), "?"), offset, offset + p.numel()))
        offset += p.numel()
    return list(RECORD), segments


n_it = int(os.environ.get('DEBUG_ITERATIONS', '4'))
a, seg = run_rec(n_it)
b, _ = run_rec(n_it)
print(f"{len(a)} steps recorded; DEBUG_DETERMINISTIC={os.environ.get('DEBUG_DETERMINISTIC')}")
for i, (x, y) in enumerate(zip(a, b)):
    same_idx = torch.equal(x["indices"], y["indices"])
    same_grad = torch.equal(x["grad"], y["grad"])
    same_par = torch.equal(x["params"], y["params"])
    if not (same_idx and same_grad and same_par):
        print(f"step {i}: graph state after {x['state']}/{y['state']} indices equal {same_idx} grads equal {same_grad} params equal {same_par}")
    if not same_grad:
        for name, s0, e0 in seg:
            d = (x["grad"][s0:e0].double() - y["grad"][s0:e0].double()).abs()
            if d.max() > 0:
                xs, ys = x["grad"][s0:e0], y["grad"][s0:e0]
                print(f"     {name}: max abs diff {d.max().item():.3e} ({int((d > 0).sum())}/{d.numel()}); run1 norm {xs.norm().item():.4e} run2 norm {ys.norm().item():.4e}")
                for j in range(max(0, i - 3), i):
                    print(f"         vs step {j}: run1==run1[{j}] {torch.equal(xs, a[j]['grad'][s0:e0])}, run2==run2[{j}] {torch.equal(ys, b[j]['grad'][s0:e0])}, "
                          f"run1==run2[{j}] {torch.equal(xs, b[j]['grad'][s0:e0])}; norms at {j}: {a[j]['grad'][s0:e0].norm().item():.4e}")
                print(f"         run1 head {xs[:6].tolist()}")
                print(f"         run2 head {ys[:6].tolist()}")
                if "after_assemble" in x:
                    xa, ya = x["after_assemble"][s0:e0], y["after_assemble"][s0:e0]
                    print(f"         right behind the assembly: run1 equals its end-of-step value {torch.equal(xa, xs)}, run2 {torch.equal(ya, ys)}, "
                          f"run1 == run2 {torch.equal(xa, ya)}; heads {xa[:3].tolist()} / {ya[:3].tolist()}")
        for tag in x.get("sources", {}):
            xs_, ys_ = x["sources"][tag], y["sources"].get(tag)
            if ys_ is not None and xs_.shape == ys_.shape and not torch.equal(xs_, ys_):
                where = (xs_ != ys_).nonzero()
                print(f"     deferred partials of param@{tag[0]:#x} column {tag[1]} differ in {where.shape[0]} words, first at {where[0].tolist()}: "
                      f"{xs_[tuple(where[0])].item()!r} / {ys_[tuple(where[0])].item()!r}")
        break
else:
    print("all recorded steps bit-identical between the two runs")

sys.path.insert(0, str(Path(__file__).resolve().parent))
from graph_census import summarize  # noqa: E402

summarize(_graphs._Capture.censuses)
