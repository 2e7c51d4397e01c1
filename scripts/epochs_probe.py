import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import cusrl_amd as cusrl
cusrl.config.set_device("cuda:0")
for epochs in (1, 5, 9):
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(4096, 48, 12, device="cuda:0")
    factory = cusrl.preset.PpoAgentFactory(compile=True, sampler_epochs=epochs, optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent
    obs, state, _ = env.reset()
    times = []
    update = agent.update
    def timed(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = update(*a, **k)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        return r
    agent.update = timed
    for i in range(14):
        obs, state = trainer._rollout_and_update(obs, state)
    t = sorted(times[4:])
    print(f"epochs {epochs}: update median {t[len(t)//2]*1e3:.3f} ms  min {t[0]*1e3:.3f} ms")
