"""String-valued enums that are part of the drop-in API (reference: stratego_env/game/enums.py:4-29).

Observation dict keys are the *values* of ObservationComponents; players are the ints 1 and -1.
"""
from enum import Enum


class ObservationModes(Enum):
    PARTIALLY_OBSERVABLE = 'partially_observable'
    FULLY_OBSERVABLE = 'fully_observable'
    BOTH_OBSERVATIONS = 'both_observations'


class ObservationComponents(Enum):
    PARTIAL_OBSERVATION = 'partial_observation'
    FULL_OBSERVATION = 'full_observation'
    VALID_ACTIONS_MASK = 'valid_actions_mask'
    INTERNAL_STATE = 'internal_state'


class GameVersions(Enum):
    STANDARD = 'standard'
    SHORT_STANDARD = 'short_standard'
    MEDIUM_STANDARD = 'medium_standard'
    STANDARD2 = 'standard2'
    BARRAGE = 'barrage'
    SHORT_BARRAGE = 'short_barrage'
    OCTA_BARRAGE = 'octa_barrage'
    MEDIUM = 'medium'
    TINY = 'tiny'
    MICRO = 'micro'
    FIVES = 'fives'


class SP(Enum):
    """Piece codes (reference: stratego_procedural_impl.py:145-163)."""
    NOPIECE = 0
    SPY = 1
    SCOUT = 2
    MINER = 3
    SERGEANT = 4
    LIEUTENANT = 5
    CAPTAIN = 6
    MAJOR = 7
    COLONEL = 8
    GENERAL = 9
    MARSHALL = 10
    FLAG = 11
    BOMB = 12
    UNKNOWN = 13
