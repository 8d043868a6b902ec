// lane_harness.hip -- TEST INFRASTRUCTURE: the lane-per-game logic of stratego_env_amd/csrc/sgx_lane.h compiled for the HOST
// (hipcc --offload-host-only), so that tests/test_lane_logic_cpu.py can play it against the CPU oracle step by step without a GPU.
// Nothing in the product loads this library; the device kernel that wraps the same functions is tested on the GPU (-m gpu).
// One entry point per board size: an env.step() on the reference's int64 [34,R,C] state, like so_env_step3 of the oracle.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <string.h>

#include "stratego_mi355x.h"
#include "sgx_layout.h"
#include "sgx_lane.h"

namespace {

template <class G>
void from_state(const int64_t *st, int player, LaneGame &g, uint16_t *ev, uint32_t &obst) {
    constexpr int RC = G::RC, C = G::C;
    memset(&g, 0, sizeof(g));
    obst = 0;
    for (int i = 0; i < RC; ++i) {
        g.pc[0] = lg_set(g.pc[0], i, (int)st[0 * RC + i]);
        g.pc[1] = lg_set(g.pc[1], i, (int)st[1 * RC + i]);
        g.po[0] = lg_set(g.po[0], i, (int)st[3 * RC + i]);
        g.po[1] = lg_set(g.po[1], i, (int)st[4 * RC + i]);
        if (st[2 * RC + i]) obst |= 1u << i;
        if (st[32 * RC + i]) g.still[0] |= 1u << i;
        if (st[33 * RC + i]) g.still[1] |= 1u << i;
    }
    g.turn = (int)st[5 * RC + 0];
    g.flags = (st[5 * RC + 1] ? F_OVER : 0) | (st[5 * RC + 2] > 0 ? F_WIN_P1 : st[5 * RC + 2] < 0 ? F_WIN_M1 : 0) | (st[5 * RC + C + 1] ? F_END_INVALID : 0) |
              (player < 0 ? F_PLAYER_M1 : 0);
    g.max_turns = (int)st[5 * RC + C];
    for (int pl = 0; pl < 2; ++pl) {
        int found = 0, pair = 0;
        for (int i = 0; i < RC && found < 2; ++i)
            if (st[(6 + pl) * RC + i]) { pair |= G::make_pair(i, (int)st[(6 + pl) * RC + i]) << (16 * found); ++found; }
        if (pl) g.rp1 = pair; else g.rp0 = pair;
    }
    for (int key = 0; key < 24; ++key)
        for (int i = 0; i < RC; ++i)
            if (st[(8 + key) * RC + i] > 0 && g.n_events < (int)G::EVL_MAX)
                ev[g.n_events++] = (uint16_t)((((int)st[(8 + key) * RC + i] - 1) << G::EV_COUNT_SHIFT) | (key << G::CELL_BITS) | i);
}

template <class G>
void to_state(const LaneGame &g, const uint16_t *ev, uint32_t obst, int64_t *st) {
    constexpr int RC = G::RC, C = G::C;
    memset(st, 0, sizeof(int64_t) * 34 * RC);
    for (int i = 0; i < RC; ++i) {
        st[0 * RC + i] = lg_nib(g.pc[0], i); st[1 * RC + i] = lg_nib(g.pc[1], i);
        st[3 * RC + i] = lg_nib(g.po[0], i); st[4 * RC + i] = lg_nib(g.po[1], i);
        st[2 * RC + i] = (obst >> i) & 1u;
        st[32 * RC + i] = (g.still[0] >> i) & 1u; st[33 * RC + i] = (g.still[1] >> i) & 1u;
    }
    st[5 * RC + 0] = g.turn;
    st[5 * RC + 1] = (g.flags & F_OVER) ? 1 : 0;
    st[5 * RC + 2] = (g.flags & F_WIN_P1) ? 1 : (g.flags & F_WIN_M1) ? -1 : 0;
    st[5 * RC + C] = g.max_turns;
    st[5 * RC + C + 1] = (g.flags & F_END_INVALID) ? 1 : 0;
    for (int pl = 0; pl < 2; ++pl) {
        const int rp = pl ? g.rp1 : g.rp0;
        for (int k = 0; k < 2; ++k) {
            const int pr = (rp >> (16 * k)) & 0xFFFF;
            if (G::pair_code(pr) != 0) st[(6 + pl) * RC + G::pair_cell(pr)] = G::pair_code(pr);
        }
    }
    for (int i = 0; i < g.n_events; ++i) {
        const int e = ev[i], key = (e >> G::CELL_BITS) & 31, cell = e & G::CELL_MASK;
        st[(8 + key) * RC + cell] = (e >> G::EV_COUNT_SHIFT) + 1;
    }
}

// One env.step() (mode 0) or observation (mode 1) on `state` / `*player` (both updated), the way the device kernel composes the lane
// functions.  out_flags: bit 0 invalid action, bit 1 done, bit 2 ending invalid; mask: uint8 [NA] of the next mover; kth: the
// flat index of the k-th valid action of that mask for k = k_sample; rewards for +1 / -1.  The record round trip (lane_store ->
// lane_load on the packed layout) is part of every call.
template <class G>
int step(int64_t *state, int *player_io, int action, const int32_t *pos, int step_flags, int mode, int max_events, int k_sample, uint8_t *mask_out,
         int *nvalid_out, int *kth_out, float *rewards, int *out_flags) {
    LaneGame g0;
    alignas(16) uint8_t rec[512];
    memset(rec, 0, sizeof(rec));
    uint16_t *ev = reinterpret_cast<uint16_t *>(rec + G::EVL_OFF);
    uint32_t obst;
    from_state<G>(state, *player_io, g0, ev, obst);
    lane_store<G>(g0, rec);
    LaneGame g;
    lane_load<G>(g, rec);                                   // through the packed record, as the kernel sees a game
    uint8_t combat[256];
    for (int a = 0; a < 16; ++a)
        for (int d = 0; d < 16; ++d) combat[16 * a + d] = (uint8_t)combat_outcome(a, d);
    int player = (g.flags & F_PLAYER_M1) ? -1 : 1;
    const int mover = player;
    LaneApplied ap{false, false};
    bool invalid = false;
    uint32_t V[G::K - 1];
    if (mode == 0) {
        const LaneMove m = lane_decode<G>(action, make_int4(pos[0], pos[1], pos[2], pos[3]), step_flags, player);
        bool has_moves = false;
        if (m.valid && m.noop && !(g.flags & F_OVER)) has_moves = lane_gen_moves<G>(g, player == 1 ? 0 : 1, obst, false, V) != 0;
        ap = lane_apply<G>(g, ev, m, player, obst, combat, max_events, step_flags, has_moves);
        if (ap.applied) player = -player; else invalid = true;
    }
    const int qi = player == 1 ? 0 : 1;
    int nvalid = lane_gen_moves<G>(g, qi, obst, (g.flags & F_OVER) != 0, V);
    const bool over = lane_finish(g, ap, mover, nvalid);
    if (over && nvalid != 0) { for (int c = 0; c < G::K - 1; ++c) V[c] = 0; nvalid = 0; }
    g.flags = (g.flags & ~F_PLAYER_M1) | (player == -1 ? F_PLAYER_M1 : 0);
    const bool end_invalid = over && (g.flags & F_END_INVALID);
    rewards[0] = rewards[1] = 0.f;
    if (over && !end_invalid) {
        const int w = (g.flags & F_WIN_P1) ? 1 : (g.flags & F_WIN_M1) ? -1 : 0;
        rewards[0] = w == 0 ? 1e-4f : (float)w;
        rewards[1] = w == 0 ? 1e-4f : (float)-w;
    }
    *out_flags = (invalid ? 1 : 0) | (over ? 2 : 0) | (end_invalid ? 4 : 0);
    uint32_t *mw = reinterpret_cast<uint32_t *>(mask_out);
    *kth_out = lane_emit_mask<G>(V, nvalid == 0, k_sample, [&](int j, uint32_t d) { mw[j] = d; });
    *nvalid_out = nvalid;
    lane_store<G>(g, rec);
    LaneGame g2;
    lane_load<G>(g2, rec);
    to_state<G>(g2, ev, obst, state);
    *player_io = player;
    return 0;
}

template <class G>
int sample(const int32_t *piece_counts, int usable_rows, uint64_t seed, uint64_t gid, uint64_t j, int max_turns, uint32_t obst, int64_t *state) {
    LaneGame g;
    memset(&g, 0, sizeof(g));
    uint16_t ev[16] = {0};
    lane_sample_boards<G>(g, nullptr, 0, usable_rows, piece_counts, seed, gid, j);
    g.max_turns = max_turns;
    to_state<G>(g, ev, obst, state);
    return 0;
}

}  // namespace

#define LH_API extern "C" __attribute__((visibility("default")))
#define LH_GEOMETRIES(X) X(3, 4) X(4, 4) X(4, 3) X(3, 3) X(3, 5)

LH_API int lh_step(int R, int C, int64_t *state, int *player_io, int action, const int32_t *pos, int step_flags, int mode, int max_events,
                   int k_sample, uint8_t *mask_out, int *nvalid_out, int *kth_out, float *rewards, int *out_flags) {
#define LH_CASE(r, c) \
    if (R == r && C == c) return step<Geo<r, c>>(state, player_io, action, pos, step_flags, mode, max_events, k_sample, mask_out, nvalid_out, kth_out, rewards, out_flags);
    LH_CASE(3, 4) LH_CASE(4, 4) LH_CASE(4, 3)
#undef LH_CASE
    return -1;
}

LH_API int lh_sample(int R, int C, const int32_t *piece_counts, int usable_rows, uint64_t seed, uint64_t gid, uint64_t j, int max_turns,
                     uint32_t obst, int64_t *state) {
#define LH_CASE(r, c) \
    if (R == r && C == c) return sample<Geo<r, c>>(piece_counts, usable_rows, seed, gid, j, max_turns, obst, state);
    LH_CASE(3, 4) LH_CASE(4, 4) LH_CASE(4, 3)
#undef LH_CASE
    return -1;
}
