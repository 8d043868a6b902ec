"""stratego_env_amd: MI355X-native vectorised Stratego environment (hot path of JBLanier/stratego_env)."""
from .enums import ObservationModes, ObservationComponents, GameVersions, SP  # noqa: F401
from .config import VARIANTS, Variant, get_variant  # noqa: F401
