import sys, time
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import cusrl_amd as cusrl
from cusrl_amd.utils.affinity import pin_host_thread
cusrl.config.set_device("cuda:0")
pin_host_thread(0)
cusrl.set_global_seed(42)
env = cusrl.testing.SyntheticEnvironment(4096, 48, 12, device="cuda:0")
factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"capturable": True, "fused": True})
trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
agent = trainer.agent
obs, state, _ = env.reset()
for _ in range(3):
    obs, state = trainer._rollout_and_update(obs, state)
spent = defaultdict(float); calls = defaultdict(int)
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); spent[label] += time.perf_counter() - t0; calls[label] += 1; return r
    setattr(obj, name, w)
wrap(agent, "_save_transition", "step/_save_transition")
wrap(agent.hook, "post_step", "step/hook.post_step")
wrap(agent.buffer, "push", "step/buffer.push")
wrap(agent.actor, "reset_memory", "step/actor.reset_memory")
wrap(agent.hook, "should_update", "step/hook.should_update")
wrap(agent, "step", "agent.step total")
wrap(agent, "act", "agent.act total")
wrap(agent._graphed_act, "run", "act/graphed.run")
wrap(env, "step", "env.step")
wrap(trainer.stats, "track", "stats.track")
wrap(trainer, "_done_indices", "done_indices")
wrap(env, "reset", "env.reset")
n_iter = 6
torch.cuda.synchronize(); t0 = time.perf_counter()
upd = 0.0
orig_update = agent.update
def timed_update():
    global upd
    torch.cuda.synchronize(); a = time.perf_counter(); r = orig_update(); torch.cuda.synchronize(); upd += time.perf_counter() - a; return r
agent.update = timed_update
for _ in range(n_iter):
    obs, state = trainer._rollout_and_update(obs, state)
torch.cuda.synchronize(); total = time.perf_counter() - t0
steps = n_iter * 24
print(f"per step (rollout only): {(total - upd) / steps * 1e6:.1f} us; update {upd / n_iter * 1e3:.2f} ms")
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / steps * 1e6:7.1f} us/step ({calls[k] // n_iter} calls/iter)")
