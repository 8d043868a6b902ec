// sgx_lane.h -- lane-per-game logic for boards of at most 16 cells (Micro 3x4, Tiny 4x4): one game per LANE, 64 games per wave
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip (one translation unit).  The functions in this header are plain
// scalar code on one game's registers -- no LDS, no cross-lane operation -- and compile for the host as well: tests/lane_harness.cpp
// builds them with g++ and tests/test_lane_logic_cpu.py plays them against the CPU oracle (test infrastructure only; the product has
// no CPU path).  The kernel that wraps them (staging, cooperative emission) is in sgx_lane_kernel.h.
//
// Why.  With one wave per four 3x4 games (sgx_step.h, Geo::LPG = 16) a step costs 291 VALU instructions per game and a launch of
// 65,536 Micro games keeps every SIMD's VALU busy for ~31 us: the toy boards ran at the instruction-issue limit, not at the memory
// rate.  Here a game's boards are nibble-packed in registers -- 16 cells x 4 bit = one 64-bit word per board -- and the rules are
// bitboard arithmetic on 16-bit cell masks, the same instruction stream for 64 games at once.
//
// Reference functions reproduced (paths relative to /root/reference/stratego_env/game): stratego_procedural_impl.py (impl)
//   decode            maenv:684-689 -> impl:314-347, 698-720, 350-383      lane_decode
//   validity          impl:723-831                                         lane_apply
//   move application  impl:894-1028                                        lane_apply
//   valid moves       impl:399-517                                         lane_gen_moves (bitboards, mover's perspective)
//   endings           impl:1031-1043                                       lane_finish
//   random setups     util.py:13-53 (counter RNG, sgx_layout.h)               lane_sample_boards
#pragma once

#ifndef SGX_HD
#define SGX_HD __host__ __device__ __forceinline__
#endif

namespace {

// One game in registers.  pc / po: true / partially-observable piece code per ABSOLUTE cell, one nibble each (cell i at bits
// 4i .. 4i+3; impl layers 0/1 and 3/4); still: never-moved flag per absolute cell, one bit each (layers 32/33); the scalars of the
// packed record (sgx_layout.h).  The capture-event list stays in memory (LDS on the device): `ev`.
struct LaneGame {
    uint64_t pc[2], po[2];
    uint32_t still[2];
    int turn, flags, max_turns, game_no, n_events, rp0, rp1;
};

SGX_HD int lg_nib(uint64_t x, int i) { return (int)((x >> (4 * i)) & 15u); }
SGX_HD uint64_t lg_set(uint64_t x, int i, int v) { return (x & ~((uint64_t)15u << (4 * i))) | ((uint64_t)(unsigned)v << (4 * i)); }

// bit i of the result = nibble i of w is non-zero (8 nibbles of a 32-bit word)
SGX_HD uint32_t lg_nz8(uint32_t w) {
    uint32_t t = w | (w >> 2);
    t = (t | (t >> 1)) & 0x11111111u;
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}
SGX_HD uint32_t lg_nz16(uint64_t x) { return lg_nz8((uint32_t)x) | (lg_nz8((uint32_t)(x >> 32)) << 8); }
// bit i = nibble i of x equals v
SGX_HD uint32_t lg_eq16(uint64_t x, int v) { return 0xFFFFu & ~lg_nz16(x ^ ((uint64_t)(unsigned)v * 0x1111111111111111ull)); }

// the low `n` bits of x in reverse order (n <= 16): perspective of player -1 = the board turned by 180 degrees (impl:645-675)
SGX_HD uint32_t lg_rev(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x) >> (32 - n);
#else
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((x >> i) & 1u) << (n - 1 - i);
    return r;
#endif
}
SGX_HD int lg_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// python-style floor division / modulo by a positive constant (the reference's garbage-action chain, see sgx_step.h)
SGX_HD int lg_fdiv(int a, int b) { int q = a / b; return (a % b < 0) ? q - 1 : q; }
SGX_HD int lg_fmod(int a, int b) { int m = a % b; return m < 0 ? m + b : m; }

// ---------------------------------------------------------------------------------------------------------------------
// Valid moves of player index qi in ITS perspective as bitboards: V[ch] bit p = perspective cell p may play channel ch
// (ch < K - 1: the move channels in the reference's order -- +r by 1..R-1, -r, +c by 1..C-1, -c, impl:285-311).  Returns the
// number of valid moves.  impl:399-517: blocked-by = off board, obstacle, own piece; scouts walk on over empty cells and stop
// after an enemy; the two-square veto removes ONE destination and the scout walks past it (impl:439-445).
// ---------------------------------------------------------------------------------------------------------------------
template <class G>
SGX_HD int lane_gen_moves(const LaneGame &g, int qi, uint32_t obst_abs, bool game_over, uint32_t (&V)[G::K - 1]) {
    constexpr int R = G::R, C = G::C, RC = G::RC;
    constexpr uint32_t ALL = (1u << RC) - 1u;
    uint32_t first_col = 0, last_col = 0;
    for (int r = 0; r < R; ++r) { first_col |= 1u << (r * C); last_col |= 1u << (r * C + C - 1); }
    const uint64_t mine = qi ? g.pc[1] : g.pc[0], theirs = qi ? g.pc[0] : g.pc[1];     // (selects, not g.pc[qi]: a run-time index into
    uint32_t own = lg_nz16(mine), enemy = lg_nz16(theirs), obst = obst_abs;            //  the register struct would put it in scratch memory)
    uint32_t movable = own & ~lg_eq16(mine, SP_FLAG) & ~lg_eq16(mine, SP_BOMB), scouts = lg_eq16(mine, SP_SCOUT);
    if (qi) { own = lg_rev(own, RC); enemy = lg_rev(enemy, RC); obst = lg_rev(obst, RC); movable = lg_rev(movable, RC); scouts = lg_rev(scouts, RC); }
    if (game_over) movable = 0;
    // the two-square veto: recent[start] == -3, recent[end] == 1, no enemy on `end` (the mover's two recent-move pairs)
    const int rp = qi ? g.rp1 : g.rp0;
    const int pa = rp & 0xFFFF, pb = (rp >> 16) & 0xFFFF;
    const int ca = G::pair_code(pa), cb = G::pair_code(pb);
    int vs = -1, ve = -1;
    if (ca == -3 && cb == 1) { vs = G::pair_cell(pa); ve = G::pair_cell(pb); }
    if (cb == -3 && ca == 1) { vs = G::pair_cell(pb); ve = G::pair_cell(pa); }
    if (vs >= RC || ve >= RC) vs = -1;                                   // (imported records: never index past the board)
    if (vs >= 0 && qi) { vs = RC - 1 - vs; ve = RC - 1 - ve; }
    int vdir = -1, vdist = 0;
    if (vs >= 0 && !((enemy >> ve) & 1u)) {
        const int rs = vs / C, cs = vs - rs * C, re = ve / C, ce = ve - re * C;
        if (cs == ce && re != rs) { vdir = re > rs ? 0 : 1; vdist = re > rs ? re - rs : rs - re; }
        else if (rs == re && ce != cs) { vdir = ce > cs ? 2 : 3; vdist = ce > cs ? ce - cs : cs - ce; }
    }
    const uint32_t T = ~own & ~obst & ALL, E = T & ~enemy;
    int total = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t t = T, e = E, clear = ALL;
        const int maxk = d < 2 ? R - 1 : C - 1;
        const int ch0 = d == 0 ? 0 : d == 1 ? R - 1 : d == 2 ? 2 * (R - 1) : 2 * (R - 1) + (C - 1);
#pragma unroll
        for (int k = 1; k <= maxk; ++k) {
            // start cells whose cell k steps away in direction d is in T / E (no wrap around the board's sides)
            if (d == 0) { t >>= C; e >>= C; }
            else if (d == 1) { t = (t << C) & ALL; e = (e << C) & ALL; }
            else if (d == 2) { t = (t >> 1) & ~last_col; e = (e >> 1) & ~last_col; }
            else { t = (t << 1) & ~first_col & ALL; e = (e << 1) & ~first_col & ALL; }
            uint32_t ok = clear & t & (k == 1 ? movable : (movable & scouts));
            if (vdir == d && vdist == k) ok &= ~(1u << vs);
            V[ch0 + k - 1] = ok;
            total += lg_popc(ok);
            clear &= e;
        }
    }
    return total;
}

// ---------------------------------------------------------------------------------------------------------------------
// Action decode (maenv:684-689): what the mover passed -> absolute (sr, sc, er, ec), or the no-op, or "not even decodable".
// The garbage cases go through the reference's own chain with Python floor-division semantics (sgx_step.h has the long form).
// ---------------------------------------------------------------------------------------------------------------------
struct LaneMove {
    int sr, sc, er, ec;
    bool valid, noop;
};
template <class G>
SGX_HD LaneMove lane_decode(int a, int4 pos, int flags, int player) {
    constexpr int R = G::R, C = G::C, K = G::K, NA = G::NA, MPA = G::MPA, AS = G::AS;
    LaneMove m{0, 0, 0, 0, true, false};
    if (flags & SGX_STEP_ACTIONS_POSITIONS) {
        m.sr = pos.x; m.sc = pos.y; m.er = pos.z; m.ec = pos.w;
    } else if (flags & SGX_STEP_ACTIONS_1D) {
        if (a == AS - 1) m.noop = true;
        else {
            const int q = lg_fdiv(a, MPA), off = lg_fmod(a, MPA);
            m.sr = lg_fdiv(q, C); m.sc = lg_fmod(q, C);
            if (off >= R) { m.ec = off - R; m.er = m.sr; } else { m.er = off; m.ec = m.sc; }
        }
    } else if (a < 0 || a >= NA) {
        m.valid = false;                                                     // np.unravel_index raises
    } else {
        const int cell = a / K, ch = a - cell * K;
        int sr = cell / C, sc = cell - sr * C, er, ec;
        if (ch < R - 1) { er = sr + ch + 1; ec = sc; }                       // impl:322-324
        else if (ch < 2 * (R - 1)) { er = sr - (ch - (R - 1) + 1); ec = sc; }
        else if (ch < 2 * (R - 1) + (C - 1)) { er = sr; ec = sc + (ch - 2 * (R - 1) + 1); }
        else { er = sr; ec = sc - (ch - (2 * (R - 1) + (C - 1)) + 1); }      // also the no-op channel
        if (ch < K - 1 && er >= 0 && er < R && ec >= 0 && ec < C) {
            if (player == -1) { sr = R - 1 - sr; sc = C - 1 - sc; er = R - 1 - er; ec = C - 1 - ec; }
        } else {
            int idx = (sr * C + sc) * MPA + ((er != sr) ? er : R + ec);     // impl:268-277
            if (player == -1 && idx != AS - 1) {                             // impl:698-720
                const int q = lg_fdiv(idx, MPA), off = lg_fmod(idx, MPA);
                int r0 = lg_fdiv(q, C), c0 = lg_fmod(q, C), r1, c1;
                if (off >= R) { c1 = off - R; r1 = r0; } else { r1 = off; c1 = c0; }
                r0 = R - 1 - r0; r1 = R - 1 - r1; c0 = C - 1 - c0; c1 = C - 1 - c1;
                idx = (r0 * C + c0) * MPA + ((r1 != r0) ? r1 : R + c1);
            }
            if (idx == AS - 1) m.noop = true;                                // impl:809-814
            else {                                                           // impl:369-383
                const int q = lg_fdiv(idx, MPA), off = lg_fmod(idx, MPA);
                sr = lg_fdiv(q, C); sc = lg_fmod(q, C);
                if (off >= R) { ec = off - R; er = sr; } else { er = off; ec = sc; }
            }
        }
        m.sr = sr; m.sc = sc; m.er = er; m.ec = ec;
    }
    return m;
}

// One more captured piece on (layer, cell) `key`: the count of its event goes up, or a new event is appended (sgx_step.h:
// add_capture).  `ev`: the game's event list.
template <class G>
SGX_HD int lane_add_capture(uint16_t *ev, int n_events, int max_events, int key, bool multi = false) {
    bool found = false;
    for (int i = 0; i < G::EVL_MAX; ++i) {
        if (i < n_events && (int)(ev[i] & G::EV_KEY_MASK) == key) {
            const bool room = (int)(ev[i] >> G::EV_COUNT_SHIFT) < EV_COUNT_MAX - 1;
            if (room) ev[i] = (uint16_t)(ev[i] + (1 << G::EV_COUNT_SHIFT));
            if (room || !multi) found = true;             // (multi: a full event does not count -- the capture opens another one)
        }
    }
    if (!found && n_events < max_events) { ev[n_events] = (uint16_t)key; n_events += 1; }
    return n_events;
}

// ---------------------------------------------------------------------------------------------------------------------
// Validity + move application (impl:723-831, 894-1028).  `combat`: the 16 x 16 outcome table (sgx_layout.h: COMBAT_*);
// `mover_has_moves`: only looked at for the no-op (legal iff the mover has no move).  Returns true if the action was applied
// (the state then is the successor and the mover has changed in g.flags' F_PLAYER_M1 -- NOT yet: the caller flips the player).
// ---------------------------------------------------------------------------------------------------------------------
struct LaneApplied {
    bool applied, noop;
};
template <class G, class CombatPtr>
SGX_HD LaneApplied lane_apply(LaneGame &g, uint16_t *ev, const LaneMove &m, int player, uint32_t obst_abs, CombatPtr combat, int max_events,
                              int step_flags, bool mover_has_moves, bool multi = false) {
    constexpr int R = G::R, C = G::C;
    const int pi = player == 1 ? 0 : 1;
    const bool over = (g.flags & F_OVER) != 0;
    if (!m.valid) return LaneApplied{false, false};
    if (m.noop) {
        // legal only if the mover has no move (or the game is over); finished games stay unchanged (impl:916-920)
        if (over) return LaneApplied{true, true};
        if (mover_has_moves) return LaneApplied{false, true};
        g.turn += 1;
        g.flags |= F_OVER | (player == 1 ? F_WIN_M1 : F_WIN_P1);
        return LaneApplied{true, true};
    }
    const int sr = m.sr, sc = m.sc, er = m.er, ec = m.ec;
    const bool s_in = !(sc < 0 || sc >= C || sr < 0 || sr >= R), e_in = !(ec < 0 || ec >= C || er < 0 || er >= R);
    const int s = s_in ? sr * C + sc : 0, e = e_in ? er * C + ec : 0;
    // the mover's / the opponent's boards by select (a run-time index into the register struct would put it in scratch memory)
    uint64_t own_pc = pi ? g.pc[1] : g.pc[0], en_pc = pi ? g.pc[0] : g.pc[1], own_po = pi ? g.po[1] : g.po[0], en_po = pi ? g.po[0] : g.po[1];
    uint32_t own_still = pi ? g.still[1] : g.still[0], en_still = pi ? g.still[0] : g.still[1];
    const int t = lg_nib(own_pc, s), own_e = lg_nib(own_pc, e), dest = lg_nib(en_pc, e), moved_po = lg_nib(own_po, s);
    // the mover's recent-move codes at s and e, from its two pairs
    const int rp = pi ? g.rp1 : g.rp0;
    const int pa = rp & 0xFFFF, pb = (rp >> 16) & 0xFFFF;
    int old_start = 0, old_end = 0;
    if (G::pair_code(pa) != 0) { if (G::pair_cell(pa) == s) old_start = G::pair_code(pa); if (G::pair_cell(pa) == e) old_end = G::pair_code(pa); }
    if (G::pair_code(pb) != 0) { if (G::pair_cell(pb) == s) old_start = G::pair_code(pb); if (G::pair_cell(pb) == e) old_end = G::pair_code(pb); }
    bool valid = !over;
    if (!s_in || ((obst_abs >> s) & 1u)) valid = false;
    if (!e_in || ((obst_abs >> e) & 1u)) valid = false;
    if (t == 0 || t == SP_FLAG || t == SP_BOMB) valid = false;
    if (own_e != 0) valid = false;
    if (er != sr && ec != sc) valid = false;
    if (old_start == -3 && old_end == 1 && dest == 0 && !(step_flags & SGX_STEP_ALLOW_OSCILLATION)) valid = false;   // impl:771-777
    if (valid) {
        const int dist = (er != sr) ? (er > sr ? er - sr : sr - er) : (ec > sc ? ec - sc : sc - ec);
        if (t == SP_SCOUT) {
            const int stepc = (er != sr) ? ((er > sr) ? C : -C) : ((ec > sc) ? 1 : -1);
            const uint32_t occ = lg_nz16(g.pc[0]) | lg_nz16(g.pc[1]) | obst_abs;
            for (int k = 1; k < (R > C ? R : C); ++k)
                if (k < dist && ((occ >> (s + k * stepc)) & 1u)) valid = false;
        } else if (dist > 1) valid = false;
    }
    if (!valid) return LaneApplied{false, false};
    // ---- _get_next_state (impl:905-1028)
    const int moved = t;
    g.turn += 1;
    bool wins = false, tied = false;
    if (dest != 0) {
        const int outcome = (int)combat[16 * moved + dest];
        wins = outcome >= COMBAT_WIN;
        tied = outcome == COMBAT_TIE;
        if (outcome == COMBAT_WIN_FLAG) g.flags |= F_OVER | (player == 1 ? F_WIN_P1 : F_WIN_M1);
    }
    own_still &= ~((1u << s) | (1u << e));                                   // impl:939-941
    en_still &= ~(1u << e);
    own_pc = lg_set(own_pc, s, 0);                                           // impl:950-951
    own_po = lg_set(own_po, s, 0);
    int new_rp = 0;
    if (dest == 0) {
        const bool far = (er > sr ? er - sr : sr - er) > 1 || (ec > sc ? ec - sc : sc - ec) > 1;
        own_pc = lg_set(own_pc, e, moved);
        own_po = lg_set(own_po, e, far ? SP_SCOUT : moved_po);               // impl:960-964
        const int code = old_end == 1 ? (old_start == -2 ? -3 : -2) : -1;     // impl:1019-1026
        new_rp = G::make_pair(s, 1) | (G::make_pair(e, code) << 16);
    } else {
        if (tied || wins) { en_pc = lg_set(en_pc, e, 0); en_po = lg_set(en_po, e, 0); }
        if (wins) { own_pc = lg_set(own_pc, e, moved); own_po = lg_set(own_po, e, moved); }
        if (!wins && !tied) en_po = lg_set(en_po, e, dest);
        // captured counts (impl:999-1009): the attacker's own layer unless it won, the defender's if it lost or tied
        if (!wins) g.n_events = lane_add_capture<G>(ev, g.n_events, max_events, ((12 * pi + moved - 1) << G::CELL_BITS) | e, multi);
        if (wins || tied) g.n_events = lane_add_capture<G>(ev, g.n_events, max_events, ((12 * (1 - pi) + dest - 1) << G::CELL_BITS) | e, multi);
    }
    g.pc[0] = pi ? en_pc : own_pc; g.pc[1] = pi ? own_pc : en_pc;
    g.po[0] = pi ? en_po : own_po; g.po[1] = pi ? own_po : en_po;
    g.still[0] = pi ? en_still : own_still; g.still[1] = pi ? own_still : en_still;
    if (pi) g.rp1 = new_rp; else g.rp0 = new_rp;                             // an attack wipes the mover's layer
    return LaneApplied{true, false};
}

// Opponent-stuck and max-turn endings after an applied move (impl:1031-1043).  nvalid = valid moves of the NEXT mover on the new
// state.  Returns true if the game is over now.
SGX_HD bool lane_finish(LaneGame &g, const LaneApplied &ap, int mover, int nvalid) {
    bool over = (g.flags & F_OVER) != 0;
    if (ap.applied && !ap.noop) {
        if (nvalid == 0) { over = true; g.flags = (g.flags & ~(F_WIN_P1 | F_WIN_M1)) | F_OVER | (mover == 1 ? F_WIN_P1 : F_WIN_M1); }
        if (g.turn >= g.max_turns && !over) { over = true; g.flags |= F_OVER | F_END_INVALID; }
    }
    return over;
}

// ---------------------------------------------------------------------------------------------------------------------
// The valid-actions mask of the game as BYTES, uint8 [R][C][K] in the mover's perspective, produced four at a time in ascending flat
// index order: store(j, dword) receives bytes 4j .. 4j+3 (NA = RC * K is a multiple of 4 on the boards this kernel plays).  While
// the bytes go by, the k-th set one is found (the fused sampler: k-th valid action in ascending flat index order, maenv:830-834
// with the counter RNG): returns its flat index.  `noop`: no move at all -- only [0, 0, K-1] is set (impl:514-515).
// ---------------------------------------------------------------------------------------------------------------------
template <class G, class Store>
SGX_HD int lane_emit_mask(const uint32_t (&V)[G::K - 1], bool noop, int k, Store store) {
    constexpr int K = G::K, NA = G::NA;
    static_assert(NA % 4 == 0, "lane kernel: boards with a multiple of 4 cells");
    int cnt = 0, before = 0, jf = -1;
    uint32_t df = 0;
#pragma unroll
    for (int j = 0; j < NA / 4; ++j) {
        uint32_t d = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int a = 4 * j + b, cell = a / K, ch = a - cell * K;
            uint32_t bit;
            if (ch == K - 1) bit = (cell == 0 && noop) ? 1u : 0u;
            else bit = (V[ch] >> cell) & 1u;
            d |= bit << (8 * b);
        }
        store(j, d);
        const int c2 = cnt + lg_popc(d);
        if (jf < 0 && c2 > k) { jf = j; df = d; before = cnt; }
        cnt = c2;
    }
    int r = k - before, pos = 0;
    bool found = false;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool set = ((df >> (8 * b)) & 1u) != 0;
        if (set && !found && r == 0) { pos = b; found = true; }
        if (set) r -= 1;
    }
    return jf < 0 ? K - 1 : 4 * jf + pos;
}

// ---------------------------------------------------------------------------------------------------------------------
// A fresh game (sample_boards of sgx_setup.h for one lane): Gravon table rows, or Fisher-Yates of the usable back cells with the
// counter RNG.  The permutation of at most 8 cells is a nibble array in one register.
// ---------------------------------------------------------------------------------------------------------------------
template <class G>
SGX_HD void lane_sample_boards(LaneGame &g, const uint8_t *setups, int n_setups, int usable_rows, const int32_t *piece_counts, uint64_t seed,
                               uint64_t gid, uint64_t j) {
    constexpr int C = G::C, RC = G::RC;
    const int U = usable_rows, n = U * C;
    g.pc[0] = g.pc[1] = g.po[0] = g.po[1] = 0;
    g.still[0] = g.still[1] = 0;
    if (setups) {
        const uint32_t i1 = rng_below(sgx_rng(seed, gid, j, STREAM_SETUP, 0), (uint32_t)n_setups);
        const uint32_t i2 = rng_below(sgx_rng(seed, gid, j, STREAM_SETUP, 1), (uint32_t)n_setups);
        const uint8_t *s1 = setups + (int64_t)i1 * n, *s2 = setups + (int64_t)i2 * n;
        for (int x = 0; x < n; ++x) {
            const int r = x / C, c = x - r * C;
            const int t1 = s1[(U - 1 - r) * C + c], t2 = s2[x];
            g.pc[0] = lg_set(g.pc[0], r * C + c, t1);
            g.pc[1] = lg_set(g.pc[1], RC - n + x, t2);
        }
    } else {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint64_t loc = 0xFEDCBA9876543210ull;                                // loc[i] = i
            for (int i = n - 1; i > 0; --i) {
                const int k = (int)rng_below(sgx_rng(seed, gid, j, pl ? STREAM_SHUFFLE_P2 : STREAM_SHUFFLE_P1, (uint32_t)i), (uint32_t)(i + 1));
                const int a = lg_nib(loc, i), b = lg_nib(loc, k);
                loc = lg_set(lg_set(loc, i, b), k, a);
            }
            int at = 0;
            for (int t = 1; t <= 12; ++t)
                for (int q = 0; q < piece_counts[t - 1]; ++q) {
                    const int own_cell = lg_nib(loc, at++);                      // own-side (r, c), r < U
                    g.pc[pl] = lg_set(g.pc[pl], pl ? RC - 1 - own_cell : own_cell, t);
                }
        }
    }
    // partially-observable layer = UNKNOWN and never-moved = 1 wherever a piece stands (impl:231-244)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const uint32_t occ = lg_nz16(g.pc[pl]);
        g.still[pl] = occ;
        uint64_t po = 0;
        for (int i = 0; i < RC; ++i) po |= (uint64_t)(((occ >> i) & 1u) ? SP_UNKNOWN : 0) << (4 * i);
        g.po[pl] = po;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Record bytes <-> registers.  Four dense byte boards (S bytes each) -> nibble words; the never-moved bitmaps and the scalars are
// read where they lie.  `rec` = the record image (16-byte aligned, the packed layout of sgx_layout.h).
// ---------------------------------------------------------------------------------------------------------------------
SGX_HD uint32_t lg_pack4(uint32_t x) {                     // 4 bytes (each < 16) -> 4 nibbles
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0xFFFFu;
}
SGX_HD uint32_t lg_unpack4(uint32_t n) {                   // 4 nibbles -> 4 bytes
    n = (n | (n << 8)) & 0x00FF00FFu;
    return (n | (n << 4)) & 0x0F0F0F0Fu;
}
template <class G>
SGX_HD void lane_load(LaneGame &g, const uint8_t *rec) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rec);
    uint64_t b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t x = 0;
#pragma unroll
        for (int d = 0; d < G::S / 4; ++d) x |= (uint64_t)lg_pack4(w[k * (G::S / 4) + d]) << (16 * d);
        b[k] = x;
    }
    g.pc[0] = b[B_PIECES]; g.pc[1] = b[B_PIECES + 1]; g.po[0] = b[B_PO]; g.po[1] = b[B_PO + 1];
    g.still[0] = w[G::ST_OFF / 4] & ((1u << G::RC) - 1u);
    g.still[1] = w[(G::ST_OFF + G::SB) / 4] & ((1u << G::RC) - 1u);
    const int32_t *sc = reinterpret_cast<const int32_t *>(rec + G::SC_OFF);
    g.turn = sc[0]; g.flags = sc[1]; g.max_turns = sc[2]; g.game_no = sc[3];
    g.n_events = sc[4] < (int)G::EVL_MAX ? sc[4] : (int)G::EVL_MAX;
    g.rp0 = sc[5]; g.rp1 = sc[6];
}
template <class G>
SGX_HD void lane_store(const LaneGame &g, uint8_t *rec) {
    uint32_t *w = reinterpret_cast<uint32_t *>(rec);
    const uint64_t b[4] = {g.pc[0], g.pc[1], g.po[0], g.po[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int d = 0; d < G::S / 4; ++d) w[k * (G::S / 4) + d] = lg_unpack4((uint32_t)(b[k] >> (16 * d)) & 0xFFFFu);
    w[G::ST_OFF / 4] = g.still[0];
    w[(G::ST_OFF + G::SB) / 4] = g.still[1];
    int32_t *sc = reinterpret_cast<int32_t *>(rec + G::SC_OFF);
    sc[0] = g.turn; sc[1] = g.flags; sc[2] = g.max_turns; sc[3] = g.game_no;
    sc[4] = g.n_events; sc[5] = g.rp0; sc[6] = g.rp1; sc[7] = 0;
}

}  // namespace
