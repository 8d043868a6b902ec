"""Enums of the drop-in API.  Member names and values are the reference's (stratego_env/game/enums.py:4-29; piece codes
stratego_procedural_impl.py:145-163) because they are the interface: observation dicts are keyed by the *values* of
ObservationComponents, `env_config['version']` takes a GameVersions member, `piece_amounts` is keyed by SP.
Every string-valued member's value is its lower-cased name; piece codes count up from NOPIECE = 0.
"""
from enum import Enum


def _lowercase_enum(name, members):
    return Enum(name, {m: m.lower() for m in members.split()}, module=__name__)


ObservationModes = _lowercase_enum('ObservationModes', 'PARTIALLY_OBSERVABLE FULLY_OBSERVABLE BOTH_OBSERVATIONS')

ObservationComponents = _lowercase_enum('ObservationComponents',
                                        'PARTIAL_OBSERVATION FULL_OBSERVATION VALID_ACTIONS_MASK INTERNAL_STATE')

GameVersions = _lowercase_enum('GameVersions', 'STANDARD SHORT_STANDARD MEDIUM_STANDARD STANDARD2 BARRAGE SHORT_BARRAGE '
                                               'OCTA_BARRAGE MEDIUM TINY MICRO FIVES')

# Stratego pieces: ranks 1 (spy) .. 10 (marshal), then flag, bomb, and the "unknown" marker of the partially-observable layers
SP = Enum('SP', 'NOPIECE SPY SCOUT MINER SERGEANT LIEUTENANT CAPTAIN MAJOR COLONEL GENERAL MARSHALL FLAG BOMB UNKNOWN',
          start=0, module=__name__)
