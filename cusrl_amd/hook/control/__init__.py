from cusrl_amd.hook.control.empty_cuda_cache import EmptyCudaCache
from cusrl_amd.hook.control.initialization import ModuleInitialization

__all__ = ["EmptyCudaCache", "ModuleInitialization"]
