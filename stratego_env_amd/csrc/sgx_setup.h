// sgx_setup.h -- fresh games in LDS (explicit maps, Gravon table, random placement) and small integer helpers
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Fresh game into LDS boards: _create_initial_state (impl:211-249) from own-side maps (explicit, from the
// human-setup table: util:241-275 net effect, or random back-row placement: util:13-30)
// ---------------------------------------------------------------------------------------------
template <class G, int NB>
__device__ void clear_boards(Lds<G, NB> &L, int lane) {
    const int4 z = make_int4(0, 0, 0, 0);
    for (int i = lane; i < G::LDS_BOARDS_BYTES / 16; i += G::LPG) reinterpret_cast<int4 *>(&L.b[0][0])[i] = z;
}

// place code `t` of player index pi at absolute cell
template <class G, int NB>
__device__ inline void place(Lds<G, NB> &L, int pi, int cell, int t) {
    L.b[B_PIECES + pi][cell] = (int8_t)t;
    L.b[B_PO + pi][cell] = t ? SP_UNKNOWN : 0;
    L.b[B_STILL + pi][cell] = t ? 1 : 0;
}

template <class G, int NB, class KP>
__device__ void sample_boards(Lds<G, NB> &L, const KP &P, uint64_t g, uint64_t j, int lane) {
    constexpr int C = G::C, RC = G::RC;
    const int U = P.usable_rows, n = U * C;
    clear_boards(L, lane);
    wave_sync<G>();
    if (P.setups) {
        const uint32_t i1 = rng_below(sgx_rng(P.seed, g, j, STREAM_SETUP, 0), (uint32_t)P.n_setups);
        const uint32_t i2 = rng_below(sgx_rng(P.seed, g, j, STREAM_SETUP, 1), (uint32_t)P.n_setups);
        const uint8_t *s1 = P.setups + (int64_t)i1 * n, *s2 = P.setups + (int64_t)i2 * n;
        for (int x = lane; x < n; x += G::LPG) {
            const int r = x / C, c = x - r * C;
            place(L, 0, r * C + c, s1[(U - 1 - r) * C + c]);           // p1 own-side row r = string row U-1-r
            place(L, 1, RC - n + x, s2[x]);                            // absolute rows R-U.. = string rows 0..
        }
    } else if (lane < 2) {
        // the two players' Fisher-Yates shuffles are independent (own RNG stream, own boards): lane 0 places player +1,
        // lane 1 player -1, each in its own half of the scratch (L.plist: 2n <= cells <= CNT_PAD)
        const int pl = lane;
        typename G::cell_t *loc = L.plist + pl * n;
        for (int i = 0; i < n; ++i) loc[i] = (typename G::cell_t)i;
        for (int i = n - 1; i > 0; --i) {
            const uint32_t k = rng_below(sgx_rng(P.seed, g, j, pl ? STREAM_SHUFFLE_P2 : STREAM_SHUFFLE_P1, (uint32_t)i), (uint32_t)(i + 1));
            const typename G::cell_t t = loc[i]; loc[i] = loc[k]; loc[k] = t;
        }
        int at = 0;
        for (int t = 1; t <= 12; ++t)
            for (int q = 0; q < P.piece_counts[t - 1]; ++q) {
                const int own_cell = loc[at++];                        // own-side (r, c), r < U
                place(L, pl, pl ? RC - 1 - own_cell : own_cell, t);    // p2 map rotated 180 degrees (impl:221)
            }
    }
    wave_sync<G>();
}

// python-style floor division / modulo by a positive constant
__device__ inline int fdiv_(int a, int b) { int q = a / b; return (a % b < 0) ? q - 1 : q; }
__device__ inline int fmod_(int a, int b) { int m = a % b; return m < 0 ? m + b : m; }

}  // namespace
