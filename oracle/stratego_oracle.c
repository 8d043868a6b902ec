/*
 * stratego_oracle.c -- CPU restatement of the reference's env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (stratego_env_amd/, the
 * C-ABI library, the HIP kernels) may include, link, import or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
 * the checker -- never as the thing measured or shipped.
 *
 * It follows the reference's algorithm function by function on the reference's own
 * state layout (int64[34][R][C], absolute coordinates).  Every function cites the
 * reference file:line it restates (paths relative to /root/reference):
 *   impl  = stratego_env/game/stratego_procedural_impl.py
 *   penv  = stratego_env/game/stratego_procedural_env.py
 *   maenv = stratego_env/stratego_multiagent_env.py
 *   util  = stratego_env/game/util.py
 *
 * Parity is PINNED: tools/oracle/check_oracle_vs_reference.py runs this file against
 * the imported reference in the build container (full games of every variant, garbage
 * actions included), and tests/golden/ holds vectors generated from the reference by
 * tools/oracle/gen_golden.py which tests/test_oracle_golden.py replays on any box.
 *
 * Integer semantics: the reference computes on Python ints / np.int64 with floor
 * division and non-negative modulo; fdiv()/fmod_() below reproduce that for the
 * out-of-board coordinates that garbage actions produce.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SO_EXPORT __attribute__((visibility("default")))

typedef int64_t i64;

/* ---- impl:69-163 state schema ------------------------------------------------------ */
enum {
    L_P1_PIECES = 0, L_P2_PIECES = 1, L_OBSTACLES = 2, L_P1_PO = 3, L_P2_PO = 4, L_DATA = 5,
    L_P1_RECENT = 6, L_P2_RECENT = 7, L_P1_CAP_START = 8, L_P1_CAP_END = 20, L_P2_CAP_START = 20,
    L_P2_CAP_END = 32, L_P1_STILL = 32, L_P2_STILL = 33, NUM_STATE_LAYERS = 34
};
enum { SP_NOPIECE = 0, SP_SPY = 1, SP_SCOUT = 2, SP_MINER = 3, SP_MARSHALL = 10, SP_FLAG = 11, SP_BOMB = 12, SP_UNKNOWN = 13 };
enum { RM_NODATA = 0, RM_JUST_CAME_FROM = 1, RM_JUST_ARRIVED = -1, RM_NEXT_DOUBLE_BACK_ILLEGAL = -2, RM_CANT_DOUBLE_BACK = -3 };
#define PO_OBS_LAYERS 67 /* impl:1332 */
#define FO_OBS_LAYERS 79 /* impl:1227 */

/* python floor division / modulo */
static i64 fdiv(i64 a, i64 b) { i64 q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }
static i64 fmod_(i64 a, i64 b) { i64 m = a % b; if (m != 0 && ((m < 0) != (b < 0))) m += b; return m; }
static i64 iabs(i64 a) { return a < 0 ? -a : a; }
static i64 isign(i64 a) { return (a > 0) - (a < 0); }

#define AT(state, l, r, c) ((state)[((l) * R + (r)) * C + (c)])
/* impl:136-142 StateData slices */
#define TURN_COUNT(s) AT(s, L_DATA, 0, 0)
#define GAME_OVER(s) AT(s, L_DATA, 0, 1)
#define WINNER(s) AT(s, L_DATA, 0, 2)
#define MAX_TURNS(s) AT(s, L_DATA, 1, 0)
#define ENDING_INVALID(s) AT(s, L_DATA, 1, 1)

/* impl:172-208 */
static i64 player_index(i64 player) { return fdiv(player - 1, -2); }
static i64 player_po_index(i64 player) { return player == 1 ? L_P1_PO : L_P2_PO; }
static i64 player_moves_index(i64 player) { return player == 1 ? L_P1_RECENT : L_P2_RECENT; }
static i64 player_still_index(i64 player) { return player == 1 ? L_P1_STILL : L_P2_STILL; }
static i64 player_captured_layer(i64 player, i64 piece_type) { return player == 1 ? 7 + piece_type : 19 + piece_type; }

/* impl:166-169 */
SO_EXPORT i64 so_mpa(i64 R, i64 C) { return R + C; }
/* impl:252-254 */
SO_EXPORT i64 so_action_size(i64 R, i64 C) { return R * C * so_mpa(R, C) + 1; }
/* impl:257-259 */
SO_EXPORT i64 so_spatial_channels(i64 R, i64 C) { return (R - 1) * 2 + (C - 1) * 2 + 1; }

/* impl:262-277 */
SO_EXPORT i64 so_action_1d_from_positions(i64 R, i64 C, i64 sr, i64 sc, i64 er, i64 ec) {
    i64 off = (er != sr) ? er : R + ec;
    return ((sr * C) + sc) * so_mpa(R, C) + off;
}

/* impl:280-311 ; returns -1 for a diagonal move (assert False) or start==end (ValueError) */
SO_EXPORT int so_action_spatial_from_positions(i64 R, i64 C, i64 sr, i64 sc, i64 er, i64 ec, i64 out[3]) {
    i64 col_dist = ec - sc, row_dist = er - sr, off;
    if (!(col_dist == 0 || row_dist == 0)) return -1;
    if (row_dist > 0) off = 0;
    else if (row_dist < 0) off = R - 1;
    else if (col_dist > 0) off = 2 * (R - 1);
    else if (col_dist < 0) off = 2 * (R - 1) + (C - 1);
    else return -1;
    out[0] = sr; out[1] = sc; out[2] = off + iabs(row_dist + col_dist) - 1;
    return 0;
}

/* impl:314-335 */
SO_EXPORT void so_action_positions_from_spatial(i64 R, i64 C, i64 r, i64 c, i64 ch, i64 out[4]) {
    i64 mr = R - 1, mc = C - 1, er, ec;
    if (ch < mr) { er = r + (ch + 1); ec = c; }
    else if (ch < 2 * mr) { er = r - ((ch - mr) + 1); ec = c; }
    else if (ch < 2 * mr + mc) { er = r; ec = c + ((ch - 2 * mr) + 1); }
    else { er = r; ec = c - ((ch - (2 * mr + mc)) + 1); }
    out[0] = r; out[1] = c; out[2] = er; out[3] = ec;
}

/* impl:338-347 */
SO_EXPORT i64 so_action_1d_from_spatial(i64 R, i64 C, i64 r, i64 c, i64 ch) {
    i64 p[4];
    so_action_positions_from_spatial(R, C, r, c, ch, p);
    return so_action_1d_from_positions(R, C, p[0], p[1], p[2], p[3]);
}

/* impl:350-383 ; returns -1 (ValueError) for the no-op index */
SO_EXPORT int so_action_positions_from_1d(i64 R, i64 C, i64 idx, i64 out[4]) {
    i64 mpa = so_mpa(R, C);
    if (idx == so_action_size(R, C) - 1) return -1;
    i64 sr = fdiv(fdiv(idx, mpa), C), sc = fmod_(fdiv(idx, mpa), C), off = fmod_(idx, mpa), er, ec;
    if (off >= R) { ec = off - R; er = sr; } else { er = off; ec = sc; }
    out[0] = sr; out[1] = sc; out[2] = er; out[3] = ec;
    return 0;
}

/* impl:386-396 */
SO_EXPORT int so_action_spatial_from_1d(i64 R, i64 C, i64 idx, i64 out[3]) {
    i64 p[4];
    if (so_action_positions_from_1d(R, C, idx, p)) return -1;
    return so_action_spatial_from_positions(R, C, p[0], p[1], p[2], p[3], out);
}

/* impl:211-249 ; maps are in each player's OWN-side coordinates, p2's is rotated 180 degrees */
SO_EXPORT void so_create_initial_state(i64 R, i64 C, const i64 *obstacles, const i64 *p1_map, const i64 *p2_map,
                                       i64 max_turns, i64 *state) {
    memset(state, 0, sizeof(i64) * NUM_STATE_LAYERS * R * C);
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) {
            i64 a = p1_map[r * C + c], b = p2_map[(R - 1 - r) * C + (C - 1 - c)];
            AT(state, L_P1_PIECES, r, c) = a;
            AT(state, L_P2_PIECES, r, c) = b;
            AT(state, L_P1_PO, r, c) = a != SP_NOPIECE ? SP_UNKNOWN : a;
            AT(state, L_P2_PO, r, c) = b != SP_NOPIECE ? SP_UNKNOWN : b;
            AT(state, L_P1_STILL, r, c) = a != SP_NOPIECE ? 1 : a;
            AT(state, L_P2_STILL, r, c) = b != SP_NOPIECE ? 1 : b;
            AT(state, L_OBSTACLES, r, c) = obstacles[r * C + c];
        }
    MAX_TURNS(state) = max_turns;
}

/* shared enumeration of impl:399-517 (spatial=1) and impl:520-642 (spatial=0) */
static void valid_moves(i64 R, i64 C, const i64 *state, i64 player, int spatial, i64 *mask) {
    i64 K = so_spatial_channels(R, C), AS = so_action_size(R, C);
    i64 n = spatial ? R * C * K : AS;
    memset(mask, 0, sizeof(i64) * n);
    i64 own = player_index(player), enemy = player_index(-player), rec = player_moves_index(player);
    int no_moves = 1;
    if (!GAME_OVER(state)) {
        for (i64 sr = 0; sr < R; sr++)
            for (i64 sc = 0; sc < C; sc++) {
                i64 t = AT(state, own, sr, sc);
                if (t == 0 || t == SP_FLAG || t == SP_BOMB) continue;
                if (t == SP_SCOUT) {
                    /* vertical rays then horizontal rays (impl:427-490) */
                    for (int pass = 0; pass < 4; pass++) {
                        i64 dr = pass == 0 ? 1 : pass == 1 ? -1 : 0, dc = pass == 2 ? 1 : pass == 3 ? -1 : 0;
                        i64 er = sr, ec = sc;
                        for (;;) {
                            er += dr; ec += dc;
                            if (er >= R || er < 0 || ec >= C || ec < 0 || AT(state, L_OBSTACLES, er, ec) != 0 ||
                                AT(state, own, er, ec) != 0)
                                break;
                            if (AT(state, rec, sr, sc) == RM_CANT_DOUBLE_BACK && AT(state, rec, er, ec) == RM_JUST_CAME_FROM &&
                                AT(state, enemy, er, ec) == 0)
                                continue; /* vetoed cell is skipped, the ray goes on (impl:439-445) */
                            if (spatial) {
                                i64 si[3];
                                so_action_spatial_from_positions(R, C, sr, sc, er, ec, si);
                                mask[(si[0] * C + si[1]) * K + si[2]] = 1;
                            } else
                                mask[so_action_1d_from_positions(R, C, sr, sc, er, ec)] = 1;
                            no_moves = 0;
                            if (AT(state, enemy, er, ec) != 0) break;
                        }
                    }
                } else {
                    const i64 d[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}}; /* impl:494-495 */
                    for (int k = 0; k < 4; k++) {
                        i64 er = sr + d[k][0], ec = sc + d[k][1];
                        if (ec >= C || er >= R || ec < 0 || er < 0 || AT(state, L_OBSTACLES, er, ec) != 0 ||
                            AT(state, own, er, ec) != 0)
                            continue;
                        if (AT(state, rec, sr, sc) == RM_CANT_DOUBLE_BACK && AT(state, rec, er, ec) == RM_JUST_CAME_FROM &&
                            AT(state, enemy, er, ec) == 0)
                            continue;
                        if (spatial) {
                            i64 si[3];
                            so_action_spatial_from_positions(R, C, sr, sc, er, ec, si);
                            mask[(si[0] * C + si[1]) * K + si[2]] = 1;
                        } else
                            mask[so_action_1d_from_positions(R, C, sr, sc, er, ec)] = 1;
                        no_moves = 0;
                    }
                }
            }
    }
    if (no_moves) {
        if (spatial) mask[K - 1] = 1; /* valid_moves_mask[0, 0, -1] (impl:514-515) */
        else mask[AS - 1] = 1;        /* impl:639-640 */
    }
}

/* impl:399-517 */
SO_EXPORT void so_valid_moves_spatial(i64 R, i64 C, const i64 *state, i64 player, i64 *mask) {
    valid_moves(R, C, state, player, 1, mask);
}
/* impl:520-642 */
SO_EXPORT void so_valid_moves_1d(i64 R, i64 C, const i64 *state, i64 player, i64 *mask) {
    valid_moves(R, C, state, player, 0, mask);
}

static void copy_flipped(i64 R, i64 C, const i64 *src, i64 sl, i64 *dst, i64 dl) {
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) AT(dst, dl, r, c) = AT(src, sl, R - 1 - r, C - 1 - c);
}

/* impl:645-675 */
SO_EXPORT void so_state_from_player_perspective(i64 R, i64 C, const i64 *state, i64 player, i64 *out) {
    memcpy(out, state, sizeof(i64) * NUM_STATE_LAYERS * R * C);
    if (player == 1) return;
    copy_flipped(R, C, state, L_P2_PIECES, out, L_P1_PIECES);
    copy_flipped(R, C, state, L_P1_PIECES, out, L_P2_PIECES);
    copy_flipped(R, C, state, L_OBSTACLES, out, L_OBSTACLES);
    copy_flipped(R, C, state, L_P2_PO, out, L_P1_PO);
    copy_flipped(R, C, state, L_P1_PO, out, L_P2_PO);
    copy_flipped(R, C, state, L_P2_RECENT, out, L_P1_RECENT);
    copy_flipped(R, C, state, L_P1_RECENT, out, L_P2_RECENT);
    copy_flipped(R, C, state, L_P2_STILL, out, L_P1_STILL);
    copy_flipped(R, C, state, L_P1_STILL, out, L_P2_STILL);
    for (i64 k = 0; k < 12; k++) {
        copy_flipped(R, C, state, L_P2_CAP_START + k, out, L_P1_CAP_START + k);
        copy_flipped(R, C, state, L_P1_CAP_START + k, out, L_P2_CAP_START + k);
    }
}

/* impl:678-695 */
SO_EXPORT void so_action_positions_from_player_perspective(i64 R, i64 C, i64 player, const i64 in[4], i64 out[4]) {
    if (player == 1) { memcpy(out, in, 4 * sizeof(i64)); return; }
    out[0] = (R - 1) - in[0]; out[2] = (R - 1) - in[2];
    out[1] = (C - 1) - in[1]; out[3] = (C - 1) - in[3];
}

/* impl:698-720 */
SO_EXPORT i64 so_action_1d_from_player_perspective(i64 R, i64 C, i64 idx, i64 player) {
    if (player == 1) return idx;
    if (idx == so_action_size(R, C) - 1) return idx;
    i64 p[4], f[4];
    so_action_positions_from_1d(R, C, idx, p);
    so_action_positions_from_player_perspective(R, C, player, p, f);
    return so_action_1d_from_positions(R, C, f[0], f[1], f[2], f[3]);
}

/* impl:723-798 */
SO_EXPORT int so_is_move_valid_by_position(i64 R, i64 C, const i64 *state, i64 player, i64 sr, i64 sc, i64 er, i64 ec,
                                           int allow_osc) {
    i64 own = player_index(player), enemy = player_index(-player), rec = player_moves_index(player);
    if (GAME_OVER(state)) return 0;
    if (sc < 0 || sc >= C || sr < 0 || sr >= R || AT(state, L_OBSTACLES, sr, sc) != 0) return 0;
    if (ec < 0 || ec >= C || er < 0 || er >= R || AT(state, L_OBSTACLES, er, ec) != 0) return 0;
    i64 t = AT(state, own, sr, sc);
    if (t == 0 || t == SP_FLAG || t == SP_BOMB) return 0;
    if (AT(state, own, er, ec) != 0) return 0;
    if (er != sr && ec != sc) return 0;
    if (AT(state, rec, sr, sc) == RM_CANT_DOUBLE_BACK && AT(state, rec, er, ec) == RM_JUST_CAME_FROM &&
        AT(state, enemy, er, ec) == 0 && !allow_osc)
        return 0;
    if (t == SP_SCOUT) {
        if (er != sr) {
            i64 d = isign(er - sr);
            for (i64 r = sr + d; r != er; r += d)
                if (AT(state, own, r, ec) != 0 || AT(state, enemy, r, ec) != 0 || AT(state, L_OBSTACLES, r, ec)) return 0;
        } else {
            i64 d = isign(ec - sc);
            if (d != 0) /* range(step=0) cannot occur: start==end was rejected by the owned-piece test */
                for (i64 c = sc + d; c != ec; c += d)
                    if (AT(state, own, er, c) != 0 || AT(state, enemy, er, c) != 0 || AT(state, L_OBSTACLES, er, c)) return 0;
        }
    } else {
        if (iabs(er - sr) > 1 || iabs(ec - sc) > 1) return 0;
    }
    return 1;
}

/* impl:801-831 */
SO_EXPORT int so_is_move_valid_by_1d(i64 R, i64 C, const i64 *state, i64 player, i64 idx, int allow_osc) {
    i64 AS = so_action_size(R, C);
    if (idx == AS - 1) {
        i64 *m = (i64 *)malloc(sizeof(i64) * AS);
        so_valid_moves_1d(R, C, state, player, m);
        int ok = m[AS - 1] == 1;
        free(m);
        return ok;
    }
    i64 p[4];
    so_action_positions_from_1d(R, C, idx, p);
    return so_is_move_valid_by_position(R, C, state, player, p[0], p[1], p[2], p[3], allow_osc);
}

/* impl:834-842 ; float32 under Numba */
SO_EXPORT float so_game_ended(i64 R, i64 C, const i64 *state, i64 player) {
    if (GAME_OVER(state)) {
        i64 w = WINNER(state);
        if (w == 0) return 1e-4f;
        return (float)(w * player);
    }
    return 0.0f;
}

/* impl:845-849 */
SO_EXPORT int so_game_result_is_invalid(i64 R, i64 C, const i64 *state) {
    if (GAME_OVER(state)) return ENDING_INVALID(state) != 0;
    return 0;
}

/* impl:894-1045 ; returns 0, or -1 where the reference raises ValueError (invalid move) */
SO_EXPORT int so_next_state(i64 R, i64 C, const i64 *state, i64 player, i64 idx, int allow_osc, i64 *ns) {
    i64 AS = so_action_size(R, C);
    if (!so_is_move_valid_by_1d(R, C, state, player, idx, allow_osc)) return -1;
    memcpy(ns, state, sizeof(i64) * NUM_STATE_LAYERS * R * C);
    if (GAME_OVER(ns)) return 0; /* impl:907-909 */
    TURN_COUNT(ns) = TURN_COUNT(ns) + 1;
    if (idx == AS - 1) { /* impl:916-920: no-op loses, no max-turn check */
        GAME_OVER(ns) = 1;
        WINNER(ns) = -player;
        return 0;
    }
    i64 p[4];
    so_action_positions_from_1d(R, C, idx, p);
    i64 sr = p[0], sc = p[1], er = p[2], ec = p[3];
    i64 own = player_index(player), enemy = player_index(-player);
    i64 own_po = player_po_index(player), enemy_po = player_po_index(-player);
    i64 own_still = player_still_index(player), enemy_still = player_still_index(-player);

    AT(ns, own_still, sr, sc) = 0; /* impl:939-941 */
    AT(ns, own_still, er, ec) = 0;
    AT(ns, enemy_still, er, ec) = 0;

    i64 moved = AT(ns, own, sr, sc), moved_po = AT(ns, own_po, sr, sc), dest = AT(ns, enemy, er, ec);
    AT(ns, own, sr, sc) = SP_NOPIECE; /* impl:950-951 */
    AT(ns, own_po, sr, sc) = SP_NOPIECE;

    int wins = 0, tied = 0;
    if (dest == SP_NOPIECE) { /* impl:955-964 */
        AT(ns, own, er, ec) = moved;
        if (iabs(er - sr) > 1 || iabs(ec - sc) > 1) AT(ns, own_po, er, ec) = SP_SCOUT;
        else AT(ns, own_po, er, ec) = moved_po;
    } else { /* impl:966-995 */
        if (moved == SP_MINER && dest == SP_BOMB) wins = 1;
        else if (moved == SP_SPY && dest == SP_MARSHALL) wins = 1;
        else if (dest == SP_FLAG) { GAME_OVER(ns) = 1; WINNER(ns) = player; wins = 1; }
        else if (dest != SP_BOMB) {
            if (moved == dest) tied = 1;
            else if (moved > dest) wins = 1;
        }
        if (tied || wins) { AT(ns, enemy, er, ec) = SP_NOPIECE; AT(ns, enemy_po, er, ec) = SP_NOPIECE; }
        if (wins) { AT(ns, own, er, ec) = moved; AT(ns, own_po, er, ec) = moved; }
        if (!wins && !tied) AT(ns, enemy_po, er, ec) = dest;
    }
    if (dest != SP_NOPIECE) { /* impl:999-1009 */
        if (!wins) { i64 l = player_captured_layer(player, moved); AT(ns, l, er, ec) = AT(ns, l, er, ec) + 1; }
        if (wins || tied) { i64 l = player_captured_layer(-player, dest); AT(ns, l, er, ec) = AT(ns, l, er, ec) + 1; }
    }
    { /* impl:1013-1028 */
        i64 ml = player_moves_index(player);
        i64 old_end = AT(ns, ml, er, ec), old_start = AT(ns, ml, sr, sc);
        for (i64 r = 0; r < R; r++)
            for (i64 c = 0; c < C; c++) AT(ns, ml, r, c) = 0;
        if (dest == SP_NOPIECE) {
            AT(ns, ml, sr, sc) = RM_JUST_CAME_FROM;
            if (old_end == RM_JUST_CAME_FROM) {
                if (old_start == RM_NEXT_DOUBLE_BACK_ILLEGAL) AT(ns, ml, er, ec) = RM_CANT_DOUBLE_BACK;
                else AT(ns, ml, er, ec) = RM_NEXT_DOUBLE_BACK_ILLEGAL;
            } else
                AT(ns, ml, er, ec) = RM_JUST_ARRIVED;
        }
    }
    { /* impl:1031-1036 */
        i64 *m = (i64 *)malloc(sizeof(i64) * AS);
        so_valid_moves_1d(R, C, ns, -player, m);
        if (m[AS - 1] == 1) { GAME_OVER(ns) = 1; WINNER(ns) = player; }
        free(m);
    }
    if (TURN_COUNT(ns) >= MAX_TURNS(ns) && !GAME_OVER(ns)) { /* impl:1040-1043 */
        GAME_OVER(ns) = 1;
        ENDING_INVALID(ns) = 1;
    }
    return 0;
}

/* impl:1335-1397 ; raw (un-normalised) float32 (R,C,67), perspective flip applied inside like the reference */
SO_EXPORT void so_po_obs_extended(i64 R, i64 C, const i64 *state_in, i64 player, float *obs) {
    i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    so_state_from_player_perspective(R, C, state_in, player, state);
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) {
            float *o = obs + (r * C + c) * PO_OBS_LAYERS;
            for (i64 t = 1; t < 13; t++) o[0 + t - 1] = AT(state, L_P1_PIECES, r, c) == t ? 1.0f : 0.0f;
            for (i64 t = 1; t < 14; t++) o[12 + t - 1] = AT(state, L_P1_PO, r, c) == t ? 1.0f : 0.0f;
            for (i64 t = 1; t < 14; t++) o[25 + t - 1] = AT(state, L_P2_PO, r, c) == t ? 1.0f : 0.0f;
            o[38] = (float)AT(state, L_OBSTACLES, r, c);
            o[39] = (float)AT(state, L_P1_RECENT, r, c);
            o[40] = (float)AT(state, L_P2_RECENT, r, c);
            for (i64 k = 0; k < 12; k++) o[41 + k] = (float)AT(state, L_P1_CAP_START + k, r, c);
            for (i64 k = 0; k < 12; k++) o[53 + k] = (float)AT(state, L_P2_CAP_START + k, r, c);
            o[65] = (float)AT(state, L_P1_STILL, r, c);
            o[66] = (float)AT(state, L_P2_STILL, r, c);
        }
    free(state);
}

/* impl:1230-1303 ; raw float32 (R,C,79) */
SO_EXPORT void so_fo_obs_extended(i64 R, i64 C, const i64 *state_in, i64 player, float *obs) {
    i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    so_state_from_player_perspective(R, C, state_in, player, state);
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) {
            float *o = obs + (r * C + c) * FO_OBS_LAYERS;
            for (i64 t = 1; t < 13; t++) o[0 + t - 1] = AT(state, L_P1_PIECES, r, c) == t ? 1.0f : 0.0f;
            for (i64 t = 1; t < 13; t++) o[12 + t - 1] = AT(state, L_P2_PIECES, r, c) == t ? 1.0f : 0.0f;
            for (i64 t = 1; t < 14; t++) o[24 + t - 1] = AT(state, L_P1_PO, r, c) == t ? 1.0f : 0.0f;
            for (i64 t = 1; t < 14; t++) o[37 + t - 1] = AT(state, L_P2_PO, r, c) == t ? 1.0f : 0.0f;
            o[50] = (float)AT(state, L_OBSTACLES, r, c);
            o[51] = (float)AT(state, L_P1_RECENT, r, c);
            o[52] = (float)AT(state, L_P2_RECENT, r, c);
            for (i64 k = 0; k < 12; k++) o[53 + k] = (float)AT(state, L_P1_CAP_START + k, r, c);
            for (i64 k = 0; k < 12; k++) o[65 + k] = (float)AT(state, L_P2_CAP_START + k, r, c);
            o[77] = (float)AT(state, L_P1_STILL, r, c);
            o[78] = (float)AT(state, L_P2_STILL, r, c);
        }
    free(state);
}

/* maenv:261-313 highs/lows and maenv:388-391 ranges/mids ; piece_amounts indexed by piece type 0..12 */
SO_EXPORT void so_p_obs_norm_constants(const i64 *piece_amounts, float *mids, float *ranges) {
    float hi[PO_OBS_LAYERS], lo[PO_OBS_LAYERS];
    for (int i = 0; i < 38; i++) { hi[i] = 1; lo[i] = -1; }
    hi[38] = 1; lo[38] = -1;
    hi[39] = hi[40] = (float)RM_JUST_CAME_FROM;
    lo[39] = lo[40] = (float)RM_CANT_DOUBLE_BACK;
    for (int i = 41; i < 65; i++) { hi[i] = 8; lo[i] = 0; }
    hi[65] = hi[66] = 1; lo[65] = lo[66] = -1;
    for (int t = 1; t <= 12; t++)
        if (piece_amounts[t] > 1) { hi[41 + t - 1] = (float)piece_amounts[t]; hi[53 + t - 1] = (float)piece_amounts[t]; }
    for (int i = 0; i < PO_OBS_LAYERS; i++) {
        ranges[i] = (hi[i] - lo[i]) / 2.0f;
        mids[i] = (hi[i] + lo[i]) / 2.0f;
    }
}

/* maenv:202-258 + maenv:393-396 (fully observable, extended channels) */
SO_EXPORT void so_f_obs_norm_constants(const i64 *piece_amounts, float *mids, float *ranges) {
    float hi[FO_OBS_LAYERS], lo[FO_OBS_LAYERS];
    for (int i = 0; i < 50; i++) { hi[i] = 1; lo[i] = -1; }
    hi[50] = 1; lo[50] = -1;
    hi[51] = hi[52] = (float)RM_JUST_CAME_FROM;
    lo[51] = lo[52] = (float)RM_CANT_DOUBLE_BACK;
    for (int i = 53; i < 77; i++) { hi[i] = 8; lo[i] = 0; }
    hi[77] = hi[78] = 1; lo[77] = lo[78] = -1;
    for (int t = 1; t <= 12; t++)
        if (piece_amounts[t] > 1) { hi[53 + t - 1] = (float)piece_amounts[t]; hi[65 + t - 1] = (float)piece_amounts[t]; }
    for (int i = 0; i < FO_OBS_LAYERS; i++) {
        ranges[i] = (hi[i] - lo[i]) / 2.0f;
        mids[i] = (hi[i] + lo[i]) / 2.0f;
    }
}

/* ---- obs_channel_mode='original' (deprecated 32/33-layer observations holding piece VALUES, not one-hots) ---- */
#define PO_OBS_LAYERS_ORIG 32 /* impl:1148 */
#define FO_OBS_LAYERS_ORIG 33 /* impl:1070 */

/* impl:1153-1197 ; raw float32 (R,C,32) */
SO_EXPORT void so_po_obs_original(i64 R, i64 C, const i64 *state_in, i64 player, float *obs) {
    i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    so_state_from_player_perspective(R, C, state_in, player, state);
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) {
            float *o = obs + (r * C + c) * PO_OBS_LAYERS_ORIG;
            o[0] = (float)AT(state, L_P1_PIECES, r, c);
            o[1] = (float)AT(state, L_P1_PO, r, c);
            o[2] = (float)AT(state, L_P2_PO, r, c);
            o[3] = (float)AT(state, L_OBSTACLES, r, c);
            o[4] = (float)AT(state, L_P1_RECENT, r, c);
            o[5] = (float)AT(state, L_P2_RECENT, r, c);
            for (i64 k = 0; k < 12; k++) o[6 + k] = (float)AT(state, L_P1_CAP_START + k, r, c);
            for (i64 k = 0; k < 12; k++) o[18 + k] = (float)AT(state, L_P2_CAP_START + k, r, c);
            o[30] = (float)AT(state, L_P1_STILL, r, c);
            o[31] = (float)AT(state, L_P2_STILL, r, c);
        }
    free(state);
}

/* impl:1075-1123 ; raw float32 (R,C,33) */
SO_EXPORT void so_fo_obs_original(i64 R, i64 C, const i64 *state_in, i64 player, float *obs) {
    i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    so_state_from_player_perspective(R, C, state_in, player, state);
    for (i64 r = 0; r < R; r++)
        for (i64 c = 0; c < C; c++) {
            float *o = obs + (r * C + c) * FO_OBS_LAYERS_ORIG;
            o[0] = (float)AT(state, L_P1_PIECES, r, c);
            o[1] = (float)AT(state, L_P2_PIECES, r, c);
            o[2] = (float)AT(state, L_OBSTACLES, r, c);
            o[3] = (float)AT(state, L_P1_RECENT, r, c);
            o[4] = (float)AT(state, L_P2_RECENT, r, c);
            o[5] = (float)AT(state, L_P1_PO, r, c);
            o[6] = (float)AT(state, L_P2_PO, r, c);
            for (i64 k = 0; k < 12; k++) o[7 + k] = (float)AT(state, L_P1_CAP_START + k, r, c);
            for (i64 k = 0; k < 12; k++) o[19 + k] = (float)AT(state, L_P2_CAP_START + k, r, c);
            o[31] = (float)AT(state, L_P1_STILL, r, c);
            o[32] = (float)AT(state, L_P2_STILL, r, c);
        }
    free(state);
}

/* maenv:146-199 (_get_partially_observable_max_and_min_vals) + maenv:388-391 */
SO_EXPORT void so_p_obs_norm_constants_original(const i64 *piece_amounts, float *mids, float *ranges) {
    float hi[PO_OBS_LAYERS_ORIG], lo[PO_OBS_LAYERS_ORIG];
    hi[0] = 12; lo[0] = 0;                   /* SP.BOMB .. SP.NOPIECE */
    hi[1] = hi[2] = 13; lo[1] = lo[2] = 0;   /* SP.UNKNOWN .. SP.NOPIECE */
    hi[3] = 2; lo[3] = 0;
    hi[4] = hi[5] = (float)RM_JUST_CAME_FROM;
    lo[4] = lo[5] = (float)RM_CANT_DOUBLE_BACK;
    for (int i = 6; i < 32; i++) { hi[i] = 2; lo[i] = 0; }
    for (int t = 1; t <= 12; t++)
        if (piece_amounts[t] > 1) { hi[6 + t - 1] = (float)piece_amounts[t]; hi[18 + t - 1] = (float)piece_amounts[t]; }
    for (int i = 0; i < PO_OBS_LAYERS_ORIG; i++) {
        ranges[i] = (hi[i] - lo[i]) / 2.0f;
        mids[i] = (hi[i] + lo[i]) / 2.0f;
    }
}

/* maenv:87-143 (_get_fully_observable_max_and_min_vals) + maenv:393-396 */
SO_EXPORT void so_f_obs_norm_constants_original(const i64 *piece_amounts, float *mids, float *ranges) {
    float hi[FO_OBS_LAYERS_ORIG], lo[FO_OBS_LAYERS_ORIG];
    hi[0] = hi[1] = 12; lo[0] = lo[1] = 0;
    hi[2] = 2; lo[2] = 0;
    hi[3] = hi[4] = (float)RM_JUST_CAME_FROM;
    lo[3] = lo[4] = (float)RM_CANT_DOUBLE_BACK;
    hi[5] = hi[6] = 13; lo[5] = lo[6] = 0;
    for (int i = 7; i < 33; i++) { hi[i] = 2; lo[i] = 0; }
    for (int t = 1; t <= 12; t++)
        if (piece_amounts[t] > 1) { hi[7 + t - 1] = (float)piece_amounts[t]; hi[19 + t - 1] = (float)piece_amounts[t]; }
    for (int i = 0; i < FO_OBS_LAYERS_ORIG; i++) {
        ranges[i] = (hi[i] - lo[i]) / 2.0f;
        mids[i] = (hi[i] + lo[i]) / 2.0f;
    }
}

/* maenv:499-508 : (obs - mids) / ranges in float32, broadcast over cells */
SO_EXPORT void so_normalize_obs(i64 n_cells, i64 n_layers, const float *mids, const float *ranges, float *obs) {
    for (i64 i = 0; i < n_cells; i++)
        for (i64 l = 0; l < n_layers; l++) {
            /* two IEEE float32 roundings; the Makefile builds with -ffp-contract=off and no fast-math */
            float d = obs[i * n_layers + l] - mids[l];
            obs[i * n_layers + l] = d / ranges[l];
        }
}

/* ------------------------------------------------------------------------------------------
 * Env-level restatement of StrategoMultiAgentEnv._get_current_obs / step (maenv:447-497, 659-828)
 * for observation_mode=PARTIALLY_OBSERVABLE, obs_channel_mode='extended', no GUI/bot.
 * ------------------------------------------------------------------------------------------ */

/* maenv:447-475 : mask (R,C,K) as uint8 (reference dtype int64, values 0/1) and normalised partial obs */
SO_EXPORT void so_env_current_obs2(i64 R, i64 C, const i64 *state, i64 player, const float *mids, const float *ranges,
                                   const float *f_mids, const float *f_ranges, uint8_t *mask_u8, float *p_obs, float *f_obs);
SO_EXPORT void so_env_current_obs(i64 R, i64 C, const i64 *state, i64 player, const float *mids, const float *ranges,
                                  uint8_t *mask_u8, float *p_obs) {
    so_env_current_obs2(R, C, state, player, mids, ranges, 0, 0, mask_u8, p_obs, 0);
}
/* maenv:447-497 with observation_mode BOTH / FULLY_OBSERVABLE: f_obs (R,C,79) normalised (maenv:477-492) */
SO_EXPORT void so_env_current_obs3(i64 R, i64 C, const i64 *state, i64 player, int original, const float *mids, const float *ranges,
                                   const float *f_mids, const float *f_ranges, uint8_t *mask_u8, float *p_obs, float *f_obs);
SO_EXPORT void so_env_current_obs2(i64 R, i64 C, const i64 *state, i64 player, const float *mids, const float *ranges,
                                   const float *f_mids, const float *f_ranges, uint8_t *mask_u8, float *p_obs, float *f_obs) {
    so_env_current_obs3(R, C, state, player, 0, mids, ranges, f_mids, f_ranges, mask_u8, p_obs, f_obs);
}
/* same with obs_channel_mode selectable: original != 0 => the 32/33-layer observations (maenv:460-468, 479-486) */
SO_EXPORT void so_env_current_obs3(i64 R, i64 C, const i64 *state, i64 player, int original, const float *mids, const float *ranges,
                                   const float *f_mids, const float *f_ranges, uint8_t *mask_u8, float *p_obs, float *f_obs) {
    i64 K = so_spatial_channels(R, C), n = R * C * K;
    i64 *pp = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    so_state_from_player_perspective(R, C, state, player, pp); /* maenv:452 */
    if (mask_u8) {
        i64 *m = (i64 *)malloc(sizeof(i64) * n);
        so_valid_moves_spatial(R, C, pp, 1, m); /* maenv:454 */
        for (i64 i = 0; i < n; i++) mask_u8[i] = (uint8_t)m[i];
        free(m);
    }
    if (p_obs) {
        if (original) so_po_obs_original(R, C, pp, 1, p_obs); /* maenv:465-467 */
        else so_po_obs_extended(R, C, pp, 1, p_obs);          /* maenv:461-463 */
        so_normalize_obs(R * C, original ? PO_OBS_LAYERS_ORIG : PO_OBS_LAYERS, mids, ranges, p_obs); /* maenv:471 */
    }
    if (f_obs) {
        if (original) so_fo_obs_original(R, C, pp, 1, f_obs); /* maenv:484-486 */
        else so_fo_obs_extended(R, C, pp, 1, f_obs);          /* maenv:480-482 */
        so_normalize_obs(R * C, original ? FO_OBS_LAYERS_ORIG : FO_OBS_LAYERS, f_mids, f_ranges, f_obs); /* maenv:488 */
    }
    free(pp);
}

/* result record of one env.step() */
typedef struct {
    int32_t error;          /* 1 = the reference raises (ValueError from unravel_index or invalid move) */
    int32_t done;           /* dones["__all__"] */
    int32_t next_player;    /* self.player after the step (+1/-1) */
    int32_t ending_invalid; /* infos[p]['game_result_was_invalid'] */
    float reward_p1;        /* rewards[1] (terminal) or 0 */
    float reward_m1;        /* rewards[-1] (terminal) or 0 */
} so_step_result;

/* maenv:659-828 ; spatial flat action of the current player; state/player updated in place on success.
 * Non-terminal: obs/mask of the next mover go to slot 0.  Terminal: slot 0 = player +1, slot 1 = player -1
 * (maenv:772-773).  mask slots are R*C*K bytes, obs slots R*C*67 floats. */
SO_EXPORT void so_env_step2(i64 R, i64 C, i64 *state, i64 *player, i64 action, int penalize_ties, const float *mids,
                            const float *ranges, const float *f_mids, const float *f_ranges, uint8_t *mask_out, float *obs_out,
                            float *fobs_out, so_step_result *res);
SO_EXPORT void so_env_step(i64 R, i64 C, i64 *state, i64 *player, i64 action, int penalize_ties, const float *mids,
                           const float *ranges, uint8_t *mask_out, float *obs_out, so_step_result *res) {
    so_env_step2(R, C, state, player, action, penalize_ties, mids, ranges, 0, 0, mask_out, obs_out, 0, res);
}
/* same, also producing the fully-observable observation (slots of R*C*79 floats) when fobs_out != NULL */
SO_EXPORT void so_env_step3(i64 R, i64 C, i64 *state, i64 *player, i64 action, int penalize_ties, int original, const float *mids,
                            const float *ranges, const float *f_mids, const float *f_ranges, uint8_t *mask_out, float *obs_out,
                            float *fobs_out, so_step_result *res);
SO_EXPORT void so_env_step2(i64 R, i64 C, i64 *state, i64 *player, i64 action, int penalize_ties, const float *mids,
                            const float *ranges, const float *f_mids, const float *f_ranges, uint8_t *mask_out, float *obs_out,
                            float *fobs_out, so_step_result *res) {
    so_env_step3(R, C, state, player, action, penalize_ties, 0, mids, ranges, f_mids, f_ranges, mask_out, obs_out, fobs_out, res);
}
/* same with obs_channel_mode selectable (original != 0: slots of R*C*32 / R*C*33 floats) */
SO_EXPORT void so_env_step3(i64 R, i64 C, i64 *state, i64 *player, i64 action, int penalize_ties, int original, const float *mids,
                            const float *ranges, const float *f_mids, const float *f_ranges, uint8_t *mask_out, float *obs_out,
                            float *fobs_out, so_step_result *res) {
    i64 K = so_spatial_channels(R, C), NA = R * C * K;
    memset(res, 0, sizeof(*res));
    res->next_player = (int32_t)*player;
    if (action < 0 || action >= NA) { res->error = 1; return; } /* np.unravel_index raises (maenv:685) */
    i64 cell = action / K, ch = action % K;
    i64 idx = so_action_1d_from_spatial(R, C, cell / C, cell % C, ch);       /* maenv:686 */
    idx = so_action_1d_from_player_perspective(R, C, idx, *player);          /* maenv:689 */
    i64 *ns = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
    if (so_next_state(R, C, state, *player, idx, 0, ns)) { free(ns); res->error = 1; return; } /* maenv:691 */
    memcpy(state, ns, sizeof(i64) * NUM_STATE_LAYERS * R * C);
    free(ns);
    *player = -*player; /* penv:153 */
    res->next_player = (int32_t)*player;
    float reward = so_game_ended(R, C, state, *player); /* maenv:699 */
    if (reward == 0) {                                  /* maenv:767-770 */
        so_env_current_obs3(R, C, state, *player, original, mids, ranges, f_mids, f_ranges, mask_out, obs_out, fobs_out);
        return;
    }
    res->done = 1; /* maenv:772-805 */
    so_env_current_obs3(R, C, state, 1, original, mids, ranges, f_mids, f_ranges, mask_out, obs_out, fobs_out);
    so_env_current_obs3(R, C, state, -1, original, mids, ranges, f_mids, f_ranges, mask_out ? mask_out + NA : 0,
                        obs_out ? obs_out + R * C * (original ? PO_OBS_LAYERS_ORIG : PO_OBS_LAYERS) : 0,
                        fobs_out ? fobs_out + R * C * (original ? FO_OBS_LAYERS_ORIG : FO_OBS_LAYERS) : 0);
    int tied;
    if (so_game_result_is_invalid(R, C, state)) {
        res->ending_invalid = 1;
        res->reward_p1 = 0; res->reward_m1 = 0;
        tied = 1;
    } else {
        res->reward_p1 = so_game_ended(R, C, state, 1);
        res->reward_m1 = so_game_ended(R, C, state, -1);
        tied = !(res->reward_p1 == 1 || res->reward_p1 == -1);
    }
    if (penalize_ties && tied) { res->reward_p1 = -0.5f; res->reward_m1 = -0.5f; }
}

/* ------------------------------------------------------------------------------------------
 * Synthetic-rollout harness: the build-defined counter RNG, setup sampling and
 * "k-th valid action" rule, restated here independently of the HIP kernels so that the two can
 * be compared trajectory by trajectory.  Not reference code: the reference has no such harness.
 * ------------------------------------------------------------------------------------------ */
static uint64_t sm_fin(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
/* R(seed, global env id g, game number j, stream, counter t) */
SO_EXPORT uint64_t so_rng(uint64_t seed, uint64_t g, uint64_t j, uint32_t stream, uint32_t t) {
    uint64_t h = sm_fin(seed + 0x9E3779B97F4A7C15ull * (g + 1));
    uint64_t ctr = ((uint64_t)stream << 32) | t;
    return sm_fin(h ^ (j * 0xD1B54A32D192ED03ull + ctr * 0x8CB92BA72F3D8DD7ull + 0x2545F4914F6CDD1Dull));
}
/* bounded draw: high 32 bits scaled into [0, n) */
SO_EXPORT uint32_t so_rng_below(uint64_t r, uint32_t n) { return (uint32_t)(((r >> 32) * (uint64_t)n) >> 32); }

enum { STREAM_SETUP = 0, STREAM_ACTION = 1, STREAM_SHUFFLE_P1 = 2, STREAM_SHUFFLE_P2 = 3 };

typedef struct {
    i64 rows, cols, max_turns, usable_rows;
    i64 piece_amounts[13];   /* by piece type, [0] unused */
    const uint8_t *obstacles; /* rows*cols bytes */
    const uint8_t *setups;    /* n_setups x (usable_rows*cols) piece codes in Gravon string order, or NULL */
    i64 n_setups;
} so_variant;

/* Own-side piece maps for game j of env g.  With a setup table: util:241-275 net effect
 * (SURVEY A3): p1_map[r][c] = s1[(U-1-r)*C + c], p2_map[r][c] = s2[(U-1-r)*C + (C-1-c)].
 * Without: Fisher-Yates over the usable back cells, pieces placed in piece-type order
 * (util:13-30 with the build's RNG in place of random.shuffle). */
SO_EXPORT void so_sample_setup(const so_variant *v, uint64_t seed, uint64_t g, uint64_t j, i64 *p1_map, i64 *p2_map) {
    i64 R = v->rows, C = v->cols, U = v->usable_rows;
    memset(p1_map, 0, sizeof(i64) * R * C);
    memset(p2_map, 0, sizeof(i64) * R * C);
    if (v->setups) {
        uint32_t i1 = so_rng_below(so_rng(seed, g, j, STREAM_SETUP, 0), (uint32_t)v->n_setups);
        uint32_t i2 = so_rng_below(so_rng(seed, g, j, STREAM_SETUP, 1), (uint32_t)v->n_setups);
        const uint8_t *s1 = v->setups + (i64)i1 * U * C, *s2 = v->setups + (i64)i2 * U * C;
        for (i64 r = 0; r < U; r++)
            for (i64 c = 0; c < C; c++) {
                p1_map[r * C + c] = s1[(U - 1 - r) * C + c];
                p2_map[r * C + c] = s2[(U - 1 - r) * C + (C - 1 - c)];
            }
        return;
    }
    for (int pl = 0; pl < 2; pl++) {
        i64 n = U * C, *map = pl ? p2_map : p1_map;
        i64 *loc = (i64 *)malloc(sizeof(i64) * n);
        for (i64 i = 0; i < n; i++) loc[i] = i;
        for (i64 i = n - 1; i > 0; i--) {
            uint32_t k = so_rng_below(so_rng(seed, g, j, pl ? STREAM_SHUFFLE_P2 : STREAM_SHUFFLE_P1, (uint32_t)i), (uint32_t)(i + 1));
            i64 t = loc[i]; loc[i] = loc[k]; loc[k] = t;
        }
        i64 at = 0;
        for (int t = 1; t <= 12; t++)
            for (i64 k = 0; k < v->piece_amounts[t]; k++) map[loc[at++]] = t;
        free(loc);
    }
}

SO_EXPORT void so_reset_env(const so_variant *v, uint64_t seed, uint64_t g, uint64_t j, i64 *state) {
    i64 R = v->rows, C = v->cols;
    i64 *m1 = (i64 *)malloc(sizeof(i64) * R * C * 3), *m2 = m1 + R * C, *ob = m2 + R * C;
    so_sample_setup(v, seed, g, j, m1, m2);
    for (i64 i = 0; i < R * C; i++) ob[i] = v->obstacles[i];
    so_create_initial_state(R, C, ob, m1, m2, v->max_turns, state);
    free(m1);
}

/* k-th set byte of a mask in ascending flat index order */
SO_EXPORT i64 so_kth_valid(const uint8_t *mask, i64 n, i64 k) {
    for (i64 i = 0; i < n; i++)
        if (mask[i]) { if (k == 0) return i; k--; }
    return -1;
}
SO_EXPORT i64 so_sample_action(const uint8_t *mask, i64 n, uint64_t seed, uint64_t g, uint64_t j, uint32_t turn) {
    i64 nv = 0;
    for (i64 i = 0; i < n; i++) nv += mask[i] != 0;
    return so_kth_valid(mask, n, so_rng_below(so_rng(seed, g, j, STREAM_ACTION, turn), (uint32_t)nv));
}

static uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}
SO_EXPORT uint64_t so_fnv1a(uint64_t h, const void *p, int64_t n) { return fnv1a(h, p, (size_t)n); }

/* Play `n_steps` batched steps of envs [g0, g0+n_envs) with auto-reset, the way bench.py's GPU loop
 * does.  Per env: running FNV-1a digest over each step's (mask bytes, obs bytes, rewards, done,
 * next player) of the post-step (post-auto-reset) observation; counters of steps and finished games.
 * Returns total env steps executed.  threads: OpenMP threads over envs (<=1: serial). */
SO_EXPORT i64 so_rollout(const so_variant *v, uint64_t seed, i64 g0, i64 n_envs, i64 n_steps, int threads,
                         uint64_t *digests, i64 *games_finished) {
    i64 R = v->rows, C = v->cols, K = so_spatial_channels(R, C), NA = R * C * K, NO = R * C * PO_OBS_LAYERS;
    float mids[PO_OBS_LAYERS], ranges[PO_OBS_LAYERS];
    so_p_obs_norm_constants(v->piece_amounts, mids, ranges);
    i64 total = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) reduction(+ : total) schedule(dynamic, 16)
#endif
    for (i64 e = 0; e < n_envs; e++) {
        uint64_t g = (uint64_t)(g0 + e), j = 0, dig = 0xCBF29CE484222325ull;
        i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
        uint8_t *mask = (uint8_t *)malloc(2 * NA);
        float *obs = (float *)malloc(sizeof(float) * 2 * NO);
        i64 player = 1, fin = 0;
        so_reset_env(v, seed, g, j, state);
        so_env_current_obs(R, C, state, player, mids, ranges, mask, obs);
        for (i64 s = 0; s < n_steps; s++) {
            i64 a = so_sample_action(mask, NA, seed, g, j, (uint32_t)TURN_COUNT(state));
            so_step_result res;
            so_env_step(R, C, state, &player, a, 0, mids, ranges, mask, obs, &res);
            total++;
            if (res.done) {
                fin++; j++;
                so_reset_env(v, seed, g, j, state);
                player = 1;
                so_env_current_obs(R, C, state, player, mids, ranges, mask, obs);
            }
            int32_t tail[4] = {res.done, (int32_t)player, res.ending_invalid, res.error};
            dig = fnv1a(dig, mask, NA);
            dig = fnv1a(dig, obs, sizeof(float) * NO);
            dig = fnv1a(dig, &res.reward_p1, 4);
            dig = fnv1a(dig, &res.reward_m1, 4);
            dig = fnv1a(dig, tail, sizeof(tail));
        }
        if (digests) digests[e] = dig;
        if (games_finished) games_finished[e] = fin;
        free(state); free(mask); free(obs);
    }
    return total;
}

/* so_rollout for the checks that cannot read back every step (bench.py's verification of the envs it has just timed, the
 * full-size GPU tests that play hundreds of untested warm-up steps first):
 *   skip            the rolling digest covers steps skip .. n_steps-1 only (skip = 0: so_rollout's digest);
 *   flags bit 0     BOTH_OBSERVATIONS: the fully-observable observation is digested right after the partial one;
 *   last_digests    digest of the LAST step's outputs alone (FNV offset basis -> mask, obs, [fobs,] rewards, tail);
 *   final_states    int64 [n_envs][34][R][C] after the last step (post-auto-reset);
 *   final_info      int32 [n_envs][4] = {turn count, game number, game_over, current player} (sgx_get_env_info's record).
 * Every output pointer may be NULL.  Harness code, not reference code. */
SO_EXPORT i64 so_rollout_ex(const so_variant *v, uint64_t seed, i64 g0, i64 n_envs, i64 n_steps, i64 skip, int flags, int threads,
                            uint64_t *digests, i64 *games_finished, uint64_t *last_digests, i64 *final_states, int32_t *final_info) {
    i64 R = v->rows, C = v->cols, K = so_spatial_channels(R, C), NA = R * C * K, NO = R * C * PO_OBS_LAYERS, NF = R * C * FO_OBS_LAYERS;
    const int both = flags & 1;
    float mids[PO_OBS_LAYERS], ranges[PO_OBS_LAYERS], f_mids[FO_OBS_LAYERS], f_ranges[FO_OBS_LAYERS];
    so_p_obs_norm_constants(v->piece_amounts, mids, ranges);
    so_f_obs_norm_constants(v->piece_amounts, f_mids, f_ranges);
    i64 total = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) reduction(+ : total) schedule(dynamic, 16)
#endif
    for (i64 e = 0; e < n_envs; e++) {
        uint64_t g = (uint64_t)(g0 + e), j = 0, dig = 0xCBF29CE484222325ull, last = 0xCBF29CE484222325ull;
        i64 *state = (i64 *)malloc(sizeof(i64) * NUM_STATE_LAYERS * R * C);
        uint8_t *mask = (uint8_t *)malloc(2 * NA);
        float *obs = (float *)malloc(sizeof(float) * 2 * NO);
        float *fobs = both ? (float *)malloc(sizeof(float) * 2 * NF) : 0;
        i64 player = 1, fin = 0;
        so_reset_env(v, seed, g, j, state);
        so_env_current_obs3(R, C, state, player, 0, mids, ranges, f_mids, f_ranges, mask, obs, fobs);
        for (i64 s = 0; s < n_steps; s++) {
            i64 a = so_sample_action(mask, NA, seed, g, j, (uint32_t)TURN_COUNT(state));
            so_step_result res;
            so_env_step3(R, C, state, &player, a, 0, 0, mids, ranges, f_mids, f_ranges, mask, obs, fobs, &res);
            total++;
            if (res.done) {
                fin++; j++;
                so_reset_env(v, seed, g, j, state);
                player = 1;
                so_env_current_obs3(R, C, state, player, 0, mids, ranges, f_mids, f_ranges, mask, obs, fobs);
            }
            int32_t tail[4] = {res.done, (int32_t)player, res.ending_invalid, res.error};
            uint64_t one = 0xCBF29CE484222325ull;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 0 && s < skip) continue;
                uint64_t d = pass == 0 ? dig : one;
                d = fnv1a(d, mask, NA);
                d = fnv1a(d, obs, sizeof(float) * NO);
                if (both) d = fnv1a(d, fobs, sizeof(float) * NF);
                d = fnv1a(d, &res.reward_p1, 4);
                d = fnv1a(d, &res.reward_m1, 4);
                d = fnv1a(d, tail, sizeof(tail));
                if (pass == 0) dig = d; else one = d;
            }
            last = one;
        }
        if (digests) digests[e] = dig;
        if (games_finished) games_finished[e] = fin;
        if (last_digests) last_digests[e] = last;
        if (final_states) memcpy(final_states + e * NUM_STATE_LAYERS * R * C, state, sizeof(i64) * NUM_STATE_LAYERS * R * C);
        if (final_info) {
            final_info[4 * e + 0] = (int32_t)TURN_COUNT(state);
            final_info[4 * e + 1] = (int32_t)j;
            final_info[4 * e + 2] = (int32_t)state[5 * R * C + 1];
            final_info[4 * e + 3] = (int32_t)player;
        }
        free(state); free(mask); free(obs); free(fobs);
    }
    return total;
}
