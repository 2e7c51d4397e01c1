from cusrl_amd.sampler.mini_batch_sampler import AutoMiniBatchSampler, MiniBatchSampler, TemporalMiniBatchSampler

__all__ = ["AutoMiniBatchSampler", "MiniBatchSampler", "TemporalMiniBatchSampler"]
