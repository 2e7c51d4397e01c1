from cusrl_amd.testing.environment import DummyTorchEnvironment, SyntheticEnvironment

__all__ = ["DummyTorchEnvironment", "SyntheticEnvironment"]
