"""The reference's setup helpers (stratego_env/game/util.py) under their own names, for callers that build start states
themselves.  `game_version_config` is one of the dicts of stratego_env_amd.config (`BARRAGE_STRATEGO_CONFIG`, ...,
`VERSION_CONFIGS[version]`) or any dict with the reference's keys.  States are built by
StrategoProceduralEnv.create_initial_state, i.e. on the device.
"""
import random

import numpy as np

from . import setups
from .enums import GameVersions
from .procedural_env import StrategoProceduralEnv

_STANDARD_FAMILY = (GameVersions.STANDARD.value, GameVersions.SHORT_STANDARD.value, GameVersions.MEDIUM_STANDARD.value)
_BARRAGE_FAMILY = (GameVersions.BARRAGE.value, GameVersions.SHORT_BARRAGE.value)


def _obstacle_map(cfg):
    m = np.zeros((cfg['rows'], cfg['columns']), dtype=np.int64)
    for loc in cfg['obstacle_locations']:
        m[tuple(loc)] = 1
    return m


def _codes(setup):
    """A Gravon setup as a string of letters (util.py:154-180) or already as piece codes."""
    return setups.codes_from_string(setup) if isinstance(setup, str) else np.asarray(setup, dtype=np.uint8)


def _create_random_initial_piece_map(game_version_config):                                           # util.py:13-30
    cfg = game_version_config
    m = np.zeros((cfg['rows'], cfg['columns']), dtype=np.int64)
    cells = [(r, c) for r in range(cfg['initial_state_usable_rows']) for c in range(cfg['columns'])]
    random.shuffle(cells)
    k = 0
    for piece_type, amount in cfg['piece_amounts'].items():
        for _ in range(amount):
            m[cells[k]] = getattr(piece_type, 'value', piece_type)
            k += 1
    return m


def get_random_initial_state_fn(base_env, game_version_config):                                      # util.py:33-53
    def random_initial_state():
        maps = [_create_random_initial_piece_map(game_version_config) for _ in range(2)]
        return base_env.create_initial_state(obstacle_map=_obstacle_map(game_version_config),
                                             player_1_initial_piece_map=maps[0], player_2_initial_piece_map=maps[1],
                                             max_turns=game_version_config['max_turns'])
    return random_initial_state


def create_initial_positions_from_human_data(player1_string, player2_string, game_version_config):   # util.py:241-275
    """-> int64 [2, rows, columns]: both players' own-side piece maps (the net effect of the reference's flips)."""
    cfg = game_version_config
    m1, m2 = setups.own_side_maps(_codes(player1_string), _codes(player2_string), cfg['rows'], cfg['columns'],
                                  cfg['initial_state_usable_rows'])
    return np.asarray([m1, m2])


def create_game_from_data(player1_string, player2_string, game_version_config, procedural_env=None):  # util.py:278-298
    cfg = game_version_config
    if procedural_env is None:
        procedural_env = StrategoProceduralEnv(cfg['rows'], cfg['columns'])
    piece_maps = create_initial_positions_from_human_data(player1_string, player2_string, cfg)
    return procedural_env.create_initial_state(obstacle_map=_obstacle_map(cfg), player_1_initial_piece_map=piece_maps[0],
                                               player_2_initial_piece_map=piece_maps[1], max_turns=int(cfg['max_turns']))


def get_random_human_init_fn(game_version, game_version_config, procedural_env=None):                # util.py:301-319
    if isinstance(game_version, GameVersions):
        game_version = game_version.value
    if game_version in _STANDARD_FAMILY:
        table = setups.load_setup_table('standard')
    elif game_version in _BARRAGE_FAMILY:
        table = setups.load_setup_table('barrage')
    else:
        raise ValueError("Human inits not supported with {} game version".format(game_version))

    def random_human_init():
        i1 = int(np.random.choice(table.shape[0]))            # np.random.choice(HUMAN_INITS), twice (util.py:313-314)
        i2 = int(np.random.choice(table.shape[0]))
        return create_game_from_data(table[i1], table[i2], game_version_config, procedural_env=procedural_env)
    return random_human_init


def get_random_curriculum_init_fn(inits_path, max_turns):                                            # util.py:372-387
    from .multiagent_env import load_curriculum_start_states
    states, winners = load_curriculum_start_states(inits_path)

    def random_human_init():
        offset = np.random.randint(low=0, high=len(states))
        state = np.squeeze(np.asarray(states[offset])).astype(np.int64)
        winner = int(np.squeeze(winners[offset]))
        state[5, 0, 0] = 0                 # StateData.TURN_COUNT
        state[5, 1, 0] = max_turns         # StateData.MAX_TURNS
        return state, winner
    return random_human_init
