from cusrl_amd.hook.control.initialization import ModuleInitialization

__all__ = ["ModuleInitialization"]
