"""One RCCL rank (torchrun, backend nccl == RCCL) running the ppo preset: exercises parameter broadcast, per-step flat
gradient all-reduce between graph replays, advantage-statistics all-gather + HIP merge, metric all_gather_object."""

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402
from cusrl_amd.utils import distributed  # noqa: E402


def main(out_path: str, compile_: str):
    assert distributed.enabled()
    cusrl.utils.configure_distributed()
    assert torch.distributed.get_backend() == "nccl"
    cusrl.set_global_seed(5)
    env = cusrl.testing.SyntheticEnvironment(256, 20, 6)
    factory = cusrl.preset.PpoAgentFactory(num_steps_per_update=8, sampler_epochs=3, sampler_mini_batches=2,
                                           compile=compile_ == "1")
    trainer = cusrl.Trainer(env, factory, num_iterations=3, verbose=False)
    trainer.run_training_loop()
    mean, var = torch.tensor([1.0], device="cuda"), torch.tensor([4.0], device="cuda")
    distributed.reduce_mean_var_(mean, var)  # world of one: unchanged, but through all_gather + the HIP merge kernel
    info = {k: v for k, v in trainer.last_info.items() if k.startswith("Agent/")}
    Path(out_path).write_text(json.dumps({"info": info, "mean": mean.item(), "var": var.item(),
                                          "world": distributed.world_size()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
