from cusrl_amd.sampler.mini_batch_sampler import AutoMiniBatchSampler, MiniBatchSampler, TemporalMiniBatchSampler
from cusrl_amd.sampler.random_sampler import AutoRandomSampler, RandomSampler, TemporalRandomSampler

__all__ = ["AutoMiniBatchSampler", "AutoRandomSampler", "MiniBatchSampler", "RandomSampler", "TemporalMiniBatchSampler",
           "TemporalRandomSampler"]
