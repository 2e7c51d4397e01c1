from cusrl_amd.hook.mdp.observation import ObservationNormalization

__all__ = ["ObservationNormalization"]
