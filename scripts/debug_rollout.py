"""Find the first divergence between the host-driven and the captured rollout (debug aid)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import cusrl_amd as cusrl

DEV = "cuda:0"
cusrl.config.set_device(DEV)


def run(capture, iterations=6, N=256, T=8, compile_=True):
    cusrl.set_global_seed(21)
    env = cusrl.testing.DummyTorchEnvironment(num_instances=N, observation_dim=12, action_dim=4, device=DEV)
    factory = cusrl.preset.PpoAgentFactory(num_steps_per_update=T, sampler_epochs=2, sampler_mini_batches=2, compile=compile_,
                                           optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=iterations, verbose=False)
    trainer.capture_rollout = capture
    snaps = []

    class Snap(cusrl.Trainer.Hook):
        def post_update(self):
            agent = trainer.agent
            torch.cuda.synchronize()
            snaps.append(({k: v.clone() for k, v in agent.buffer.storage.items()},
                          torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone(),
                          torch.cuda.get_rng_state(DEV).clone()))

    trainer.hooks = (Snap(),)
    trainer.hooks[0].init(trainer)
    trainer.run_training_loop()
    return snaps


a = run(False)
b = run(True)
for it, ((ba, pa, ra), (bb, pb, rb)) in enumerate(zip(a, b)):
    print(f"iteration {it}: params equal={torch.equal(pa, pb)} maxdiff={(pa - pb).abs().max().item():.3e} rng equal={torch.equal(ra, rb)}")
    for key in ba:
        if not torch.equal(ba[key], bb[key]):
            x, y = ba[key].float(), bb[key].float()
            bad_t = [(t, int((x[t] != y[t]).sum())) for t in range(x.shape[0]) if not torch.equal(x[t], y[t])]
            print(f"   {key}: maxdiff={(x - y).abs().max().item():.3e} mismatching (t, count): {bad_t}")
