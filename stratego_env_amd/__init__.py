"""stratego_env_amd: MI355X-native vectorised Stratego environment (hot path of JBLanier/stratego_env)."""
from .enums import ObservationModes, ObservationComponents, GameVersions, SP  # noqa: F401
from .config import VARIANTS, Variant, get_variant  # noqa: F401


def __getattr__(name):
    # the reference package exports these two at the top level (stratego_env/__init__.py:1-2); resolved lazily so that
    # `import stratego_env_amd` stays free of torch for the host-only helpers
    if name in ('StrategoMultiAgentEnv', 'SPATIAL_STRATEGO_ENV'):
        from . import multiagent_env
        return getattr(multiagent_env, name)
    raise AttributeError("module 'stratego_env_amd' has no attribute %r" % name)
