// sgx_mask.h -- valid-actions mask: ray-per-lane generation, byte emission (perspective and state coordinates), k-th valid action
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Valid-actions mask of player index qi in qi's perspective, as BITS in L.mbits (bit a = flat action a) with
// per-perspective-cell counts in L.cnt.  Returns the number of valid moves (0 => the no-op bit was set).
// impl:399-517.  Work distribution: movable pieces are compacted, then every lane walks ONE ray
// (piece, direction); a Barrage position (<= 7 movable pieces) is a single 64-lane pass of <= 9 steps.
// ---------------------------------------------------------------------------------------------
#ifndef SGX_GENMASK_INLINE
#define SGX_GENMASK_INLINE __forceinline__
#endif
// finished game: only the no-op bit [0,0,K-1] (impl:414, 514-515)
template <class G, int NB>
__device__ __forceinline__ void mask_noop_only(Lds<G, NB> &L, int lane) {
    const int4 z = make_int4(0, 0, 0, 0);
    for (int i = lane; i < G::MB_WORDS / 4; i += G::LPG) reinterpret_cast<int4 *>(L.mbits)[i] = z;
    for (int i = lane; i < G::CNT_PAD / 4; i += G::LPG) reinterpret_cast<int *>(L.cnt)[i] = 0;
    wave_sync<G>();
    if (lane == 0) { L.mbits[(G::K - 1) >> 5] = 1u << ((G::K - 1) & 31); L.cnt[0] = 1; }
    wave_sync<G>();
}

// SGX_ABLATE (diagnostic builds only, tools/ablate_logic.py): parts of the game logic are SKIPPED by the bits of SGX_MAP="0,<bits>" so that
// their cost can be read off the launch time of state-preserving observe launches.  Results are wrong by construction.
#ifdef SGX_ABLATE
#define SGX_ABLATED(P_, bit) (((P_) >> (bit)) & 1)
#else
#define SGX_ABLATED(P_, bit) 0
#endif
// ANY = true: the caller only asks WHETHER player qi has a move (launches that emit neither mask nor next action: search expansion, the
// legality of a no-op): no bits, no counts -- every (piece, direction) lane looks at its first cell, and only a scout whose first cell is
// the vetoed one (impl:439-445: the ray goes on past it) walks further.  Returns 1 / 0; L.mbits and L.cnt are left alone.
template <class G, int NB, bool ANY = false>
__device__ SGX_GENMASK_INLINE int gen_mask(Lds<G, NB> &L, int qi, bool game_over, int lane, int ablate = 0) {
    constexpr int R = G::R, C = G::C, RC = G::RC, K = G::K;
    if (SGX_ABLATED(ablate, 1)) return 1;                       // no mask generation at all
    constexpr int OCC_OWN = 1, OCC_ENEMY = 2, OCC_OBST = 4, OCC_CAME_FROM = 8;
    const int8_t *own = L.b[B_PIECES + qi], *enemy = L.b[B_PIECES + 1 - qi], *rec = L.b[B_RECENT + qi], *obst = L.b[B_OBST];
    if constexpr (!ANY) {
        const int4 z = make_int4(0, 0, 0, 0);
        for (int i = lane; i < G::MB_WORDS / 4; i += G::LPG) reinterpret_cast<int4 *>(L.mbits)[i] = z;
        for (int i = lane; i < G::CNT_PAD / 4; i += G::LPG) reinterpret_cast<int *>(L.cnt)[i] = 0;
    }
    int total = 0;
    if (!game_over) {
        // pass 1: combined occupancy byte per cell (one LDS read per ray step) + compaction of movable pieces
        int npieces = 0;
#pragma unroll
        for (int cc = 0; cc < G::CPL; ++cc) {
            const int i = lane + G::LPG * cc;
            bool movable = false;
            if (i < RC) {
                const int t = own[i];
                movable = t != 0 && t != SP_FLAG && t != SP_BOMB;
                L.occ[i] = (uint8_t)((t != 0 ? OCC_OWN : 0) | (enemy[i] != 0 ? OCC_ENEMY : 0) | (obst[i] != 0 ? OCC_OBST : 0) |
                                     (rec[i] == 1 ? OCC_CAME_FROM : 0));
            }
            const unsigned long long bm = gballot<G>(movable);
            if (movable) L.plist[npieces + __popcll(bm & ((1ull << lane) - 1ull))] = (typename G::cell_t)i;
            npieces += __popcll(bm);
        }
        wave_sync<G>();
        if (SGX_ABLATED(ablate, 4)) return npieces;                  // pass 1 only
        // pass 2: one ray per lane, perspective direction order +r, -r, +c, -c (impl:427-490 / 494-495)
        const int sgn = qi ? -1 : 1;  // perspective +r is absolute -r for player -1 (impl:678-695)
        const int nrays = 4 * npieces;
        int mine = 0;
        for (int j0 = 0; j0 < nrays; j0 += G::LPG) {
            const int j = j0 + lane;
            const bool act = j < nrays;
            const int i = act ? L.plist[j >> 2] : 0, d = j & 3;
            const int t = own[i];
            const int r = i / C, c = i - r * C;
            const bool pinned = rec[i] == -3;  // JUST_ARRIVED_AND_CANT_DOUBLE_BACK
            const int pcell = qi ? RC - 1 - i : i;
            const int avail = d == 0 ? (qi ? r : R - 1 - r) : d == 1 ? (qi ? R - 1 - r : r) : d == 2 ? (qi ? c : C - 1 - c) : (qi ? C - 1 - c : c);
            const int delta = d == 0 ? sgn * C : d == 1 ? -sgn * C : d == 2 ? sgn : -sgn;
            const int bit0 = pcell * K + (d == 0 ? 0 : d == 1 ? R - 1 : d == 2 ? 2 * (R - 1) : 2 * (R - 1) + (C - 1)) - 1;
            int lim = act ? (t == SP_SCOUT ? avail : min(avail, 1)) : 0;
            int e = i, n = 0;
            for (int k = 1; k < (R > C ? R : C); ++k) {
                if (SGX_ABLATED(ablate, 0)) break;                   // no ray walk
                if (!__any(k <= lim)) break;
                if (k <= lim) {
                    e += delta;
                    const int v = L.occ[e];
                    if (v & (OCC_OWN | OCC_OBST)) {
                        lim = 0;                                                  // blocked: the ray stops
                    } else {
                        // two-square veto: this cell is skipped but the ray goes on (impl:439-445)
                        if (!(pinned && (v & OCC_CAME_FROM) && !(v & OCC_ENEMY))) {
                            if constexpr (!ANY) {
                                const int bit = bit0 + k;
                                atomicOr(&L.mbits[bit >> 5], 1u << (bit & 31));
                            }
                            ++n;
                        }
                        if (v & OCC_ENEMY) lim = 0;                               // an attacked piece ends the ray
                    }
                }
                if constexpr (ANY)
                    if (gballot<G>(n > 0) != 0ull) return 1;                      // somebody has a move: that is all that was asked
            }
            if constexpr (!ANY) {
                n = quad_sum(n);                                                  // moves of the piece = its 4 rays
                if (act && d == 0) { L.cnt[pcell] = (uint8_t)n; mine += n; }
            }
        }
        if constexpr (ANY) return 0;
        total = uni<G>(glane<G>(gscan_incl<G>(mine), G::LPG - 1));
    }
    if constexpr (ANY) return 0;                                                   // (a finished game)
    if (total == 0 && lane == 0) {
        L.mbits[(K - 1) >> 5] = 1u << ((K - 1) & 31);  // valid_moves_mask[0, 0, -1] (impl:514-515); mbits was just zeroed
        L.cnt[0] = 1;
    }
    wave_sync<G>();
    return total;
}

// "does player qi have a move at all?" (gen_mask<..., ANY = true>)
template <class G, int NB>
__device__ SGX_GENMASK_INLINE int gen_any(Lds<G, NB> &L, int qi, bool game_over, int lane) {
    return gen_mask<G, NB, true>(L, qi, game_over, lane);
}

// 4 mask bits -> 4 mask bytes
__device__ inline uint32_t expand4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }
// `n` (<= 32) mask bits starting at bit position p
template <class G, int NB>
__device__ inline uint32_t mask_bits(const Lds<G, NB> &L, int p, int n) {
    const unsigned long long w = (unsigned long long)L.mbits[p >> 5] | ((unsigned long long)L.mbits[(p >> 5) + 1] << 32);
    return (uint32_t)(w >> (p & 31)) & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
}

// LDS mask bits -> global uint8 [R,C,K].  An env's mask starts at env*NA bytes: 4-byte but not 16-byte aligned at 10x10
// (3700 = 4 mod 16), byte aligned when NA is odd (5x5, 15x15); lanes own 16-byte-aligned chunks of the global range, so
// whole chunks leave as one 16-byte store per lane and only the partial first / last chunks go out as dwords (bytes when
// the base is not 4-byte aligned).
// (Non-temporal stores for the mask measured no different from plain ones, in place and in a ring of output sets, on 10x10, 8x8 and
// 3x4: profiles/r04_mask_nt_ab.log; the experiment was removed.)
template <class G, int STRIDE = G::LPG, int NB>
__device__ void emit_mask(const Lds<G, NB> &L, uint8_t *__restrict__ dst, int lane) {
    const int A = (int)(reinterpret_cast<uintptr_t>(dst) & 15);
    const int nchunks = (A + G::NA + 15) >> 4;
    uint8_t *gbase = dst - A;                       // 16-byte aligned
    const int shift = (int)((reinterpret_cast<uintptr_t>(gbase) >> 4) & (G::LPG - 1));   // start the sweep on a 1 KiB boundary
    for (int c0 = -shift; c0 < nchunks; c0 += STRIDE) {
        const int c = c0 + lane;
        if (c < 0 || c >= nchunks) continue;
        const int lo = 16 * c - A;                  // first mask byte of this chunk
        if (lo >= 0 && lo + 16 <= G::NA) {
            const uint32_t b16 = mask_bits(L, lo, 16);
            i32x4 q4 = {(int)expand4(b16 & 15), (int)expand4((b16 >> 4) & 15), (int)expand4((b16 >> 8) & 15), (int)expand4(b16 >> 12)};
            reinterpret_cast<i32x4 *>(gbase)[c] = q4;
        } else if constexpr (G::NA % 4 == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int o = lo + 4 * w;
                if (o >= 0 && o < G::NA) *reinterpret_cast<uint32_t *>(dst + o) = expand4(mask_bits(L, o, 4));
            }
        } else {
            for (int o = max(lo, 0); o < min(lo + 16, (int)G::NA); ++o) dst[o] = (uint8_t)((L.mbits[o >> 5] >> (o & 31)) & 1u);
        }
    }
}

// The same mask in another index space: byte i of the output = mask bit src(i) (src(i) < 0: always 0).  Used for the
// functional operator API, whose masks are indexed in the coordinates of the given STATE rather than in the mover's
// perspective: the 1-D encoding (impl:520-642) and the spatial encoding for player -1 (impl:399-517 on an unflipped state).
// 16-byte chunks of the address range like emit_mask; the source index is computed per byte (a handful of integer ops).
template <class G, int NB, class F>
__device__ __forceinline__ void emit_mask_mapped_inl(const Lds<G, NB> &L, uint8_t *__restrict__ dst, int n_bytes, F src, int lane) {
    const int A = (int)(reinterpret_cast<uintptr_t>(dst) & 15);
    const int nchunks = (A + n_bytes + 15) >> 4;
    uint8_t *gbase = dst - A;
    for (int c = lane; c < nchunks; c += G::LPG) {
        const int lo = 16 * c - A;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = lo + j;
            if (i >= 0 && i < n_bytes) {
                const int b = src(i);
                if (b >= 0) w[j >> 2] |= ((L.mbits[b >> 5] >> (b & 31)) & 1u) << (8 * (j & 3));
            }
        }
        if (lo >= 0 && lo + 16 <= n_bytes) {
            i32x4 q4 = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
            reinterpret_cast<i32x4 *>(gbase)[c] = q4;
        } else {
#pragma unroll                                   // (a rolled loop would index w[] dynamically: scratch memory for the whole kernel)
            for (int j = 0; j < 16; ++j)
                if (lo + j >= 0 && lo + j < n_bytes) dst[lo + j] = (uint8_t)((w[j >> 2] >> (8 * (j & 3))) & 1u);
        }
    }
}

// The same, inlining left to the compiler: in the observing MAPPED kernel the forced inline costs 352 B of scratch; in the small
// no-observation kind the compiler otherwise keeps it as a call with a stack frame (env_step picks).
template <class G, int NB, class F>
__device__ void emit_mask_mapped(const Lds<G, NB> &L, uint8_t *__restrict__ dst, int n_bytes, F src, int lane) {
    emit_mask_mapped_inl(L, dst, n_bytes, src, lane);
}

// perspective channel of the straight move (sr,sc)->(er,ec) given in the mover's perspective (impl:280-311)
template <class G>
__device__ inline int channel_of(int sr, int sc, int er, int ec) {
    constexpr int R = G::R, C = G::C;
    const int dr = er - sr, dc = ec - sc;
    return dr > 0 ? dr - 1 : dr < 0 ? (R - 1) + (-dr - 1) : dc > 0 ? 2 * (R - 1) + dc - 1 : 2 * (R - 1) + (C - 1) + (-dc - 1);
}
// absolute 1-D action index (impl:262-277) -> bit of the perspective mask of player index qi
template <class G>
struct Src1D {
    int qi;
    __device__ int operator()(int i) const {
        constexpr int R = G::R, C = G::C, K = G::K, MPA = G::MPA;
        if (i == G::AS - 1) return K - 1;                                  // the no-op: [0,0,K-1] in any coordinates
        const int q = i / MPA, off = i - q * MPA;
        int sr = q / C, sc = q - sr * C, er, ec;
        if (off >= R) { er = sr; ec = off - R; } else { er = off; ec = sc; }
        if (er == sr && ec == sc) return -1;                              // the encoding's null moves
        if (qi) { sr = R - 1 - sr; sc = C - 1 - sc; er = R - 1 - er; ec = C - 1 - ec; }
        return (sr * C + sc) * K + channel_of<G>(sr, sc, er, ec);
    }
};
// flat spatial index in the STATE's coordinates -> bit of player -1's perspective mask (cells and directions turn by 180 degrees;
// the no-op stays at [0,0,K-1])
template <class G>
struct SrcSpatialFlipped {
    __device__ int operator()(int i) const {
        constexpr int R = G::R, C = G::C, K = G::K, RC = G::RC;
        const int cell = i / K, ch = i - cell * K;
        if (ch == K - 1) return cell == 0 ? K - 1 : -1;
        const int pch = ch < R - 1 ? ch + (R - 1) : ch < 2 * (R - 1) ? ch - (R - 1) : ch < 2 * (R - 1) + (C - 1) ? ch + (C - 1) : ch - (C - 1);
        return (RC - 1 - cell) * K + pch;
    }
};

// k-th (0-based) valid action in ascending flat index order, from L.mbits / L.cnt
template <class G, int NB>
__device__ int kth_valid(const Lds<G, NB> &L, int k, int lane) {
    constexpr int K = G::K;
    int cell = 0, before = 0, run = 0;
    bool found = false;
#pragma unroll
    for (int cc = 0; cc < G::CPL; ++cc) {
        const int c0 = L.cnt[lane + G::LPG * cc];
        const int incl = gscan_incl<G>(c0);
        const unsigned long long hit = gballot<G>(run + incl > k);
        if (!found && hit) {
            const int l = __ffsll((long long)hit) - 1;
            cell = G::LPG * cc + l;
            before = run + glane<G>(incl - c0, l);
            found = true;
        }
        run += glane<G>(incl, G::LPG - 1);
    }
    cell = uni<G>(cell);
    int kk = uni<G>(k - before);
    if constexpr (K <= G::LPG) {
        const int p = cell * K + (lane < K ? lane : 0);
        unsigned long long bits = gballot<G>(lane < K && ((L.mbits[p >> 5] >> (p & 31)) & 1u) != 0);
        for (int i = 0; i < kk; ++i) bits &= bits - 1;
        const int ch = __ffsll((long long)bits) - 1;
        return cell * K + ch;
    } else {
        // long thin boards (R + C > 33): the cell's K channels in slices of LPG
        int ch = 0;
        for (int c0 = 0; c0 < K; c0 += G::LPG) {
            const bool in = c0 + lane < K;
            const int p = cell * K + (in ? c0 + lane : 0);
            unsigned long long bits = gballot<G>(in && ((L.mbits[p >> 5] >> (p & 31)) & 1u) != 0);
            const int n = __popcll(bits);
            if (kk < n) {
                for (int i = 0; i < kk; ++i) bits &= bits - 1;
                ch = c0 + __ffsll((long long)bits) - 1;
                break;
            }
            kk -= n;
        }
        return cell * K + ch;
    }
}

}  // namespace
