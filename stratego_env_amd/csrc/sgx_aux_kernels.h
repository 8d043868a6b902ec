// sgx_aux_kernels.h -- reset, standalone sampler, export / import of reference-layout states, bookkeeping kernels
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// reset kernel: explicit own-side maps or sampled setups
// ---------------------------------------------------------------------------------------------
struct ResetParams {
    KParams k;
    const uint8_t *select;
    const int8_t *p1_maps, *p2_maps;
};

template <int R_, int C_>
__global__ __launch_bounds__(64) void reset_kernel(const ResetParams P) {
    using G = Geo<R_, C_>;
    constexpr int RC = G::RC;
    __shared__ Lds<G> L;
    const int lane = threadIdx.x;
    const int64_t env = blockIdx.x;
    if (env >= P.k.n_envs || lane >= G::LPG) return;     // helpers work on the LPG lanes of one game
    if (P.select && P.select[env] == 0) return;
    int game_no;
    if (P.p1_maps) {
        clear_boards(L, lane);
        wave_sync<G>();
        const int8_t *m1 = P.p1_maps + env * (int64_t)RC, *m2 = P.p2_maps + env * (int64_t)RC;
        for (int i = lane; i < RC; i += G::LPG) {
            place(L, 0, i, m1[i]);
            place(L, 1, i, m2[RC - 1 - i]);  // p2 map rotated 180 degrees (impl:221)
        }
        wave_sync<G>();
        game_no = 0;
    } else {
        game_no = uni<G>(rec_scal<G>(P.k.boards, P.k.rec_bytes, env)[0].w) + 1;
        sample_boards(L, P.k, (uint64_t)(P.k.env_id_offset + env), (uint64_t)game_no, lane);
    }
    write_record(L, P.k.boards + env * (int64_t)P.k.rec_bytes, P.k.rec_bytes, make_int4(0, 0, P.k.max_turns, game_no),
                 make_int4(0, 0, 0, 0), 0, lane);
}

// ---------------------------------------------------------------------------------------------
// standalone sampler: k-th set byte of each env's mask (maenv:830-834 with the counter RNG)
// ---------------------------------------------------------------------------------------------
template <int R_, int C_>
__global__ __launch_bounds__(64) void sample_kernel(const KParams P, const uint8_t *__restrict__ mask, int32_t *__restrict__ actions) {
    using G = Geo<R_, C_>;
    constexpr int RC = G::RC, K = G::K, NA = G::NA;
    __shared__ Lds<G> L;
    const int lane = threadIdx.x;
    const int64_t env = blockIdx.x;
    if (env >= P.n_envs || lane >= G::LPG) return;       // helpers work on the LPG lanes of one game
    const uint8_t *m = mask + env * (int64_t)NA;
    for (int i = lane; i < G::MB_WORDS; i += G::LPG) L.mbits[i] = 0;
    wave_sync<G>();
    {   // mask bytes -> bits in LDS, read as 16-byte chunks of the address range (the mirror image of emit_mask)
        const int A = (int)(reinterpret_cast<uintptr_t>(m) & 15);
        const int nchunks = (A + NA + 15) >> 4;
        const uint8_t *gbase = m - A;
        for (int c = lane; c < nchunks; c += G::LPG) {
            const int lo = 16 * c - A;
            if (lo >= 0 && lo + 16 <= NA) {
                const uint4 v = reinterpret_cast<const uint4 *>(gbase)[c];
                uint32_t bits = 0;
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t nz = ((((w4[j] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w4[j]) >> 7) & 0x01010101u;   // byte != 0
                    bits |= ((nz & 1u) | ((nz >> 7) & 2u) | ((nz >> 14) & 4u) | ((nz >> 21) & 8u)) << (4 * j);
                }
                if (bits) {
                    atomicOr(&L.mbits[lo >> 5], bits << (lo & 31));
                    if ((lo & 31) > 16) atomicOr(&L.mbits[(lo >> 5) + 1], bits >> (32 - (lo & 31)));
                }
            } else {
                for (int o = max(lo, 0); o < min(lo + 16, (int)NA); ++o)
                    if (m[o] != 0) atomicOr(&L.mbits[o >> 5], 1u << (o & 31));
            }
        }
    }
    wave_sync<G>();
    int mine = 0;
#pragma unroll
    for (int cc = 0; cc < G::CPL; ++cc) {
        const int cell = lane + G::LPG * cc;
        int n = 0;
        if (cell < RC) {
#pragma unroll
            for (int o = 0; o < K; o += 32) n += __popc(mask_bits(L, cell * K + o, K - o < 32 ? K - o : 32));
        }
        L.cnt[cell] = (uint8_t)n;
        mine += n;
    }
#pragma unroll
    for (int o = G::LPG / 2; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    wave_sync<G>();
    const int total = uni<G>(mine);
    const int4 sc = rec_scal<G>(P.boards, P.rec_bytes, env)[0];
    int na = -1;
    if (total > 0) {
        const uint32_t k = rng_below(sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)sc.w, STREAM_ACTION, (uint32_t)sc.x), (uint32_t)total);
        na = kth_valid(L, (int)k, lane);
    }
    if (lane == 0) actions[env] = na;
}

// ---------------------------------------------------------------------------------------------
// export / import in the reference's int64 [N,34,R,C] layout (impl:16-60); 256 threads per state
// ---------------------------------------------------------------------------------------------
template <class G, int NT_ = 256>
struct StateIO {
    static constexpr int RC = G::RC, NT = NT_, NX = SGX_STATE_LAYERS * RC, NQ = NX / 2;
    static constexpr int IMG = (G::ST_OFF + G::TAIL_BYTES + 127) & ~127;   // >= rec_bytes of any piece set on this board (BIG: + the dense recent boards)
    static constexpr int DENSE = (NX + 15) & ~15;
    // one entry per (layer, cell): a byte normally; the general-state variant counts up to 32,768 captures on a cell
    using dense_t = std::conditional_t<G::BIG, int16_t, int8_t>;
    using cap_t = std::conditional_t<G::BIG, uint16_t, uint8_t>;
    static constexpr int NE = 24 * RC, PER = (NE + NT - 1) / NT, NEP = (NE + 3) & ~3;
    static constexpr int ITER = (NQ + NT - 1) / NT;
};
// LDS of one state: the record image, and scratch of the import (count table, recent-move codes, ...) that the export reuses as
// its byte-per-(layer, cell) table
template <class G, int NT_ = 256>
struct alignas(16) StateLds {
    using IO = StateIO<G, NT_>;
    alignas(16) uint8_t img[IO::IMG];
    alignas(16) typename IO::dense_t dense[IO::DENSE]; // export: [layer][cell]; import: cap[] at 8 * RC, recent[] at 6 * RC
    int scan[IO::NT / 64];
    int scal[5];       // layer 5: turn count, game over, winner, max turns, ending invalid
    int pairs[2];
    int altered;       // the state held something the packed record cannot carry: reported through sanitised[]
};

// Record image (W.img) -> the 34 int64 layers of `env`.  The image is expanded in LDS to one BYTE per (layer, cell) -- the boards,
// the obstacle map, never-moved bits, recent-move pairs and capture events scattered by many lanes at once -- and the layers then
// stream out as 16-byte stores (two int64 per lane, every wave's store instruction one aligned KiB; non-temporal when the batch
// does not fit the Infinity Cache).  (Round 2 computed every element from the record in global memory and patched the
// recent-move and captured layers with one thread's read-modify-writes to global memory: 4.0 TB/s.)
template <class G, int NT_>
__device__ __forceinline__ void export_from_image(const KParams &P, StateLds<G, NT_> &W, int64_t *__restrict__ out, int8_t *__restrict__ player_out,
                                                  const int64_t env, const int tid, const int nt) {
    using IO = StateIO<G, NT_>;
    constexpr int RC = G::RC, C = G::C, S = G::S, NT = IO::NT, NX = IO::NX, NQ = IO::NQ;
    using dense_t = typename IO::dense_t;
    const uint8_t *img = W.img;
    dense_t *dense = W.dense;
    for (int i = tid; i < IO::DENSE * (int)sizeof(dense_t) / 4; i += NT) reinterpret_cast<int *>(dense)[i] = 0;
    __syncthreads();
    const int4 sc = reinterpret_cast<const int4 *>(img + G::SC_OFF)[0], sc2 = reinterpret_cast<const int4 *>(img + G::SC_OFF)[1];
    for (int i = tid; i < RC; i += NT) {
        dense[0 * RC + i] = (int8_t)img[(B_PIECES + 0) * S + i];
        dense[1 * RC + i] = (int8_t)img[(B_PIECES + 1) * S + i];
        dense[2 * RC + i] = (int8_t)P.tab->obstacles[i];
        dense[3 * RC + i] = (int8_t)img[(B_PO + 0) * S + i];
        dense[4 * RC + i] = (int8_t)img[(B_PO + 1) * S + i];
        const uint32_t *stb = reinterpret_cast<const uint32_t *>(img + G::ST_OFF);
        dense[32 * RC + i] = (int8_t)((stb[i >> 5] >> (i & 31)) & 1u);
        dense[33 * RC + i] = (int8_t)((stb[G::SB / 4 + (i >> 5)] >> (i & 31)) & 1u);
        if constexpr (G::BIG) {
            dense[6 * RC + i] = (int8_t)img[G::RECB_OFF + i];
            dense[7 * RC + i] = (int8_t)img[G::RECB_OFF + G::S_PAD + i];
        }
    }
    if (!G::BIG && tid < 4) {
        const int pr = (((tid >> 1) ? sc2.z : sc2.y) >> (16 * (tid & 1))) & 0xFFFF;
        if (pr >> G::CELL_BITS) dense[(6 + (tid >> 1)) * RC + G::pair_cell(pr)] = (int8_t)G::pair_code(pr);
    }
    {   // one event per (layer, cell) with captured pieces: no two lanes write the same byte
        const typename G::ev_t *ev = reinterpret_cast<const typename G::ev_t *>(img + G::EVL_OFF);
        const int n_events = min(sc2.x, (int)G::EVL_MAX);
        for (int i = tid; i < n_events; i += NT) {
            const int e = (int)ev[i], key = (e >> G::CELL_BITS) & 31, cell = e & G::CELL_MASK;
            // (a variant with more than 8 pieces of a type may hold several events of one key: their counts add up -- 32-bit LDS atomics on
            //  the word that holds the entry; sums stay below the entry's width: <= 127 pieces of a type, 32,768 in a general-state image)
            if (key < 24 && cell < RC) {
                const int idx = (8 + key) * RC + cell, cnt = (e >> G::EV_COUNT_SHIFT) + 1;
                constexpr int PERW = 4 / (int)sizeof(dense_t);
                atomicAdd(reinterpret_cast<int *>(dense) + idx / PERW, cnt << (8 * (int)sizeof(dense_t) * (idx % PERW)));
            }
        }
    }
    __syncthreads();
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 *o = reinterpret_cast<i64x2 *>(out + env * (int64_t)NX);                 // NX * 8 is a multiple of 16
    const int w = (sc.y & F_WIN_P1) ? 1 : (sc.y & F_WIN_M1) ? -1 : 0;
    // like the observation stream (sgx_obs.h): every wave's store instruction covers one 1 KiB-aligned KiB (a state is 272 * RC bytes:
    // only 16-byte aligned), and the first / last 128-byte line, shared with the neighbouring states, goes through L2
    const int m0 = (int)((reinterpret_cast<uintptr_t>(o) >> 4) & 63), l0 = (int)((reinterpret_cast<uintptr_t>(o) >> 4) & 7);
    const int first_line = l0 ? 0 : -1, last_line = ((l0 + NQ) & 7) ? (NQ - 1 + l0) >> 3 : -1;
    for (int q = tid - m0; q < NQ; q += NT) {
        if (q < 0) continue;
        i64x2 v = {(long long)dense[2 * q], (long long)dense[2 * q + 1]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int x = 2 * q + j - 5 * RC;                                          // index inside layer 5
            if (x >= 0 && x < RC) {
                long long d = 0;
                if (x == 0) d = sc.x;                                    // TURN_COUNT  [5,0,0]
                else if (x == 1) d = (sc.y & F_OVER) ? 1 : 0;            // GAME_OVER   [5,0,1]
                else if (x == 2) d = w;                                  // WINNER      [5,0,2]
                else if (x == C) d = sc.z;                               // MAX_TURNS   [5,1,0]
                else if (x == C + 1) d = (sc.y & F_END_INVALID) ? 1 : 0; // ENDING_INVALID [5,1,1]
                if (j == 0) v.x = d; else v.y = d;
            }
        }
        const bool edge = ((q + l0) >> 3) == first_line || ((q + l0) >> 3) == last_line;
        if (nt && !edge) __builtin_nontemporal_store(v, &o[q]);
        else o[q] = v;
    }
    if (tid == 0 && player_out) player_out[env] = (sc.y & F_PLAYER_M1) ? -1 : 1;
}

// The 34 int64 layers of `env` -> record image (W.img).  Reachable states only: at most two non-zero recent-move cells per player
// (impl:1013-1028), at most max_events (layer, cell) pairs with captured pieces and at most 8 of them on one pair; anything beyond
// that cannot come from play and is dropped (and reported through sanitised[]).  All of the state's 16-byte loads are issued first
// (two int64 per lane per load, fully coalesced, non-temporal), the values are scattered into the LDS image of the record (dense
// boards, never-moved BITS by atomicOr, recent-move codes and captured counts as bytes), the capture-event list is laid out with a
// block-wide prefix sum and the recent-move pairs by one wave per player with ballots.  Ends with a __syncthreads(): the image is
// complete for every thread.  (Round 2 read 8 bytes per lane and finished with one thread walking both recent-move layers and
// five dependent global loads: 3.0-3.3 TB/s.)
template <class G, int NT_>
__device__ __forceinline__ void import_to_image(const KParams &P, StateLds<G, NT_> &W, const int64_t *__restrict__ in, const int8_t *__restrict__ player_in,
                                                uint8_t *__restrict__ sanitised, const int64_t env, const int tid) {
    using IO = StateIO<G, NT_>;
    constexpr int RC = G::RC, C = G::C, S = G::S, NT = IO::NT, NX = IO::NX, NQ = IO::NQ, NE = IO::NE, PER = IO::PER, ITER = IO::ITER;
    using cap_t = typename IO::cap_t;
    uint8_t *img = W.img;
    cap_t *cap = reinterpret_cast<cap_t *>(W.dense) + 8 * RC;                 // [24][RC] captured counts
    typename IO::dense_t *recent = W.dense + 6 * RC;                          // [2][RC] recent-move codes
    const int lane = tid & 63, wave = tid >> 6;
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const i64x2 *s = reinterpret_cast<const i64x2 *>(in + env * (int64_t)NX);
    // all of the state's loads in flight at once -- up to 16 per lane; bigger boards (15 x 15 needs 15, 32 x 32 would need 68
    // = 272 VGPRs) go through them in batches of 8
    // (fewer loads per batch = fewer VGPRs = 8 instead of 7 waves per SIMD in the fused kernel: 14 / 7 / 5 loads per lane measured
    // 653 / 650 / 647 us per 65,536 Barrage states in one process -- nothing)
    constexpr int BATCH = ITER <= 16 ? ITER : 8;
    i64x2 rawv[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
        const int q = tid + k * NT;
        rawv[k] = q < NQ ? __builtin_nontemporal_load(&s[q]) : i64x2{0, 0};
    }
    if (tid == 0) W.altered = 0;     // (ordered before every `altered = 1` below by the __syncthreads() that follows the clears)
    for (int i = tid; i < IO::IMG / 4; i += NT) reinterpret_cast<uint32_t *>(img)[i] = 0;
    __syncthreads();
    bool bad = false;
    for (int k0 = 0; k0 < ITER; k0 += BATCH) {
    if (k0 > 0) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int q = tid + (k0 + k) * NT;
            rawv[k] = q < NQ ? __builtin_nontemporal_load(&s[q]) : i64x2{0, 0};
        }
    }
#pragma unroll
    for (int kk = 0; kk < BATCH; ++kk) {
        const int k = k0 + kk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int x = 2 * (tid + k * NT) + j;
            if (x >= NX) continue;
            const int l = x / RC, cell = x - l * RC;
            const int64_t raw = j ? rawv[kk].y : rawv[kk].x;
            bool ok = true;
            if (l == 5) {                                                     // the scalars (other cells of the layer are not looked at)
                const int which = cell == 0 ? 0 : cell == 1 ? 1 : cell == 2 ? 2 : cell == C ? 3 : cell == C + 1 ? 4 : -1;
                if (which >= 0) W.scal[which] = which == 2 ? (raw > 0 ? 1 : raw < 0 ? -1 : 0) : (which == 0 || which == 3) ? (int)raw : (raw != 0);
            } else if (l == 2) ok = raw == (int64_t)P.tab->obstacles[cell];   // obstacles are the variant's (a per-handle constant)
            else if (l < 2 || l == 3 || l == 4) {                             // legal range of the layer; anything else -> 0
                const int b = l < 2 ? B_PIECES + l : B_PO + (l - 3), hi = l < 2 ? SP_BOMB : SP_UNKNOWN;
                ok = raw >= 0 && raw <= hi;
                img[b * S + cell] = (uint8_t)(ok ? (int)raw : 0);
            } else if (l == 6 || l == 7) {
                ok = raw >= -3 && raw <= 1;
                recent[(l - 6) * RC + cell] = (int8_t)(ok ? (int)raw : 0);
                if constexpr (G::BIG) img[G::RECB_OFF + (l - 6) * G::S_PAD + cell] = (uint8_t)(int8_t)(ok ? (int)raw : 0);   // dense in the image
            }
            else if (l < 32) {
                // (multi_ev: a count beyond 8 is laid out as several events of the same key, 8 + 8 + ... + rest)
                const int cmax = (!G::BIG && P.multi_ev) ? 127 : (int)G::COUNT_MAX;
                ok = raw >= 0 && raw <= cmax;
                cap[(l - 8) * RC + cell] = (cap_t)(raw <= 0 ? 0 : (raw > cmax ? cmax : (int)raw));
            }
            else {
                ok = raw == 0 || raw == 1;
                if (raw == 1) atomicOr(reinterpret_cast<uint32_t *>(img + G::ST_OFF) + (l - 32) * (G::SB / 4) + (cell >> 5), 1u << (cell & 31));
            }
            bad = bad || !ok;
        }
    }
    }
    if (bad) W.altered = 1;
    __syncthreads();
    // capture events (one per non-zero count) in (layer, cell) order: thread t owns entries [t*PER, (t+1)*PER) of the count table
    const bool split = !G::BIG && P.multi_ev;                               // counts beyond 8 become several events of one key
    int cnt = 0;
    for (int k = 0; k < PER; ++k) {
        const int e = tid * PER + k;
        if (e < NE) cnt += split ? ((int)cap[e] + (int)G::COUNT_MAX - 1) / (int)G::COUNT_MAX : (cap[e] != 0);
    }
    // block-wide inclusive scan: DPP scan inside each wave, then the four wave totals through LDS
    const int incl = gscan_incl<Geo<10, 10>>(cnt);                            // (any one-game-per-wave geometry: a 64-lane scan)
    if (lane == 63) W.scan[wave] = incl;
    // the recent-move pairs of player `wave` (waves 0 and 1): the first two non-zero codes in cell order; more cannot come from play
    if (wave < 2) {
        int found = 0, total = 0, pair = 0;
        for (int c0 = 0; c0 < RC; c0 += 64) {
            const int cell = c0 + lane;
            unsigned long long bal = __ballot(cell < RC && recent[wave * RC + (cell < RC ? cell : 0)] != 0);
            total += __popcll(bal);
            while (bal && found < 2) {
                const int cb = c0 + __ffsll((long long)bal) - 1;
                pair |= G::make_pair(cb, (int)recent[wave * RC + cb]) << (16 * found);
                ++found;
                bal &= bal - 1;
            }
        }
        if (lane == 0) { W.pairs[wave] = pair; if (total > 2 && !G::BIG) W.altered = 1; }
    }
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += W.scan[w];
    int total_events = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) total_events += W.scan[w];
    {
        int at = wave_base + incl - cnt;
        typename G::ev_t *ev = reinterpret_cast<typename G::ev_t *>(img + G::EVL_OFF);
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            if (e < NE && cap[e] != 0) {
                int left = cap[e];
                do {
                    const int c = (split && left > (int)G::COUNT_MAX) ? (int)G::COUNT_MAX : left;
                    if (at < max_events_of<G>(P)) ev[at] = (typename G::ev_t)(((c - 1) << G::EV_COUNT_SHIFT) | ((e / RC) << G::CELL_BITS) | (e % RC));
                    ++at;
                    left -= c;
                } while (split && left > 0);
            }
        }
    }
    if (tid == 0) {
        int flags = 0;
        if (W.scal[1] != 0) flags |= F_OVER;
        if (W.scal[2] > 0) flags |= F_WIN_P1; else if (W.scal[2] < 0) flags |= F_WIN_M1;
        if (W.scal[4] != 0) flags |= F_END_INVALID;
        if (player_in && player_in[env] < 0) flags |= F_PLAYER_M1;
        bool dropped = total_events > max_events_of<G>(P);
        if (player_in && player_in[env] != 1 && player_in[env] != -1) dropped = true;
        if (sanitised) sanitised[env] = (W.altered || dropped) ? 1 : 0;
        if ((W.altered || dropped) && P.flag_count) P.flag_list[atomicAdd(P.flag_count, 1)] = (int32_t)env;     // (first pass of sgx_step_states)
        const int old_game = rec_scal<G>(P.boards, P.rec_bytes, env)[0].w;
        int4 *scg = reinterpret_cast<int4 *>(img + G::SC_OFF);
        scg[0] = make_int4(W.scal[0], flags, W.scal[3], old_game < 0 ? 0 : old_game);
        scg[1] = make_int4(min(total_events, max_events_of<G>(P)), W.pairs[0], W.pairs[1], 0);
    }
    __syncthreads();
}

template <int R_, int C_>
__global__ __launch_bounds__(256) void export_kernel(const KParams P, int64_t *__restrict__ out, int8_t *__restrict__ player_out, const int nt) {
    using G = Geo<R_, C_>;
    __shared__ StateLds<G> W;
    const int tid = threadIdx.x;
    const int64_t env = P.env_first + group_of_block(P);
    if (env >= P.n_envs) return;
    const int8_t *rec = P.boards + env * (int64_t)P.rec_bytes;
    for (int i = tid; i < P.rec_bytes / 16; i += 256) reinterpret_cast<int4 *>(W.img)[i] = reinterpret_cast<const int4 *>(rec)[i];
    export_from_image(P, W, out, player_out, env, tid, nt);                  // (its first barrier orders the staging above)
}

template <int R_, int C_>
__global__ __launch_bounds__(256) void import_kernel(const KParams P, const int64_t *__restrict__ in, const int8_t *__restrict__ player_in,
                                                     uint8_t *__restrict__ sanitised) {
    using G = Geo<R_, C_>;
    __shared__ StateLds<G> W;
    const int tid = threadIdx.x;
    const int64_t env = P.env_first + group_of_block(P);
    if (env >= P.n_envs) return;
    import_to_image(P, W, in, player_in, sanitised, env, tid);
    int8_t *rec = P.boards + env * (int64_t)P.rec_bytes;
    for (int i = tid; i < P.rec_bytes / 16; i += 256) reinterpret_cast<int4 *>(rec)[i] = reinterpret_cast<const int4 *>(W.img)[i];
}

// sgx_step_states: import -> env.step() -> export of one state by one 128- or 192-thread block (states_threads), the record never leaving LDS in between.
// The step is one wave's work (sgx_step.h: env_step, ~10 us of dependent LDS round trips); the other wave waits at the barrier while
// the CU's other blocks keep the memory pipes busy with their loads and stores, so the step costs no launch of its own (88 us per
// 65,536 states as a separate launch between a 300 us import and a 330 us export).  Small blocks = many of them per CU = many
// steps in flight beside the streaming.  OBS = false: no observation is rendered (get_next_state, is_move_valid_*, masks): no code
// buffer and no templates in LDS (the step runs as the value-channel kind, whose Lds has no code buffer -- the game logic does
// not depend on the observation kind), 16 blocks per CU instead of 11.  Partial-observation 'extended' kind only.
// Threads per state: 128 / 192 / 256 measured 654 / 628 / 636 us per 65,536 Barrage states in one process (tools/states_ab.py).
// (get_next_state / is_move_valid_*: three waves per state; the variants whose ONE stepping wave also emits a mask or an observation
// want more blocks per CU instead: 1-D masks 463 us with 128 threads, 592 us with 192)
// The workgroup-shared observation tables of a step of kind KIND (the staging of game_kernel_body, sgx_step.h, for a block of `nthreads`)
template <class G, int KIND>
__device__ inline void stage_kind_tables(const KParams &P, uint8_t *shared, int tid, int nthreads) {
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool FULL = ObsKind<KIND>::FULL, ORIG = ObsKind<KIND>::ORIG;
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    if constexpr (ORIG) {
        float *lut_s = reinterpret_cast<float *>(shared);
        const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0)]);
        for (int i = tid; i < LUT_DWORDS / 4; i += nthreads) reinterpret_cast<f32x4 *>(lut_s)[i] = lsrc[i];
        build_quad_table<G, PS>(reinterpret_cast<uint32_t *>(lut_s + LUT_DWORDS), tid, nthreads);
        if constexpr (FULL) {
            const f32x4 *fsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0) + 1]);
            for (int i = tid; i < LUT_DWORDS / 4; i += nthreads) reinterpret_cast<f32x4 *>(lut_s + OBS_TAB_DWORDS)[i] = fsrc[i];
            build_quad_table<G, FS>(reinterpret_cast<uint32_t *>(lut_s + OBS_TAB_DWORDS + LUT_DWORDS), tid, nthreads);
        }
    } else {
        constexpr int NP = tmpl_lds_bytes<G, KIND>(false), NF = FULL ? tmpl_lds_bytes<G, KIND>(true) : 0;
        const int4 *tp = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0]);
        for (int i = tid; i < NP / 16; i += nthreads) reinterpret_cast<int4 *>(shared)[i] = tp[i];
        if constexpr (FULL) {
            const int4 *tf = reinterpret_cast<const int4 *>(P.tab->tmpl[(raw ? 2 : 0) + 1]);
            for (int i = tid; i < NF / 16; i += nthreads) reinterpret_cast<int4 *>(shared + NP)[i] = tf[i];
        }
        const int4 *ct = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0]);
        for (int i = tid; i < CODETAB_BYTES / 16; i += nthreads) reinterpret_cast<int4 *>(shared + NP + NF)[i] = ct[i];
    }
}

template <bool MAPPED, bool OBS>
constexpr int states_threads() { return (!MAPPED && !OBS) ? 192 : 128; }
// VAR = 1: the second pass of sgx_step_states over the states the first pass had to alter (sanitised[env] != 0; every other block leaves at
// once): the same import -> env_step -> export on the general-state variant of the geometry (Geo<R, C, 1>: dense recent-move boards, an
// event for every (layer, cell) pair, counts to 32,768), whose record image exists only here in LDS; the outputs of the first pass are
// overwritten and the flag is cleared unless the state holds values outside its layers' ranges.
// KINDX >= 0 (general-state pass only): the observation kind of the step (sgx_layout.h: ObsKind; 1 = also the 79-channel observation, 2 / 3 =
// 'original' channels) instead of the partial 'extended' one -- the first pass of those kinds is the three-launch path on packed records.
template <int R_, int C_, bool MAPPED, bool OBS, int VAR = 0, int KINDX = -1>
__global__ __launch_bounds__((states_threads<MAPPED, OBS>())) void states_kernel(const KParams P, const int64_t *__restrict__ in, const int8_t *__restrict__ player_in,
                                                     uint8_t *__restrict__ sanitised, int64_t *__restrict__ out, int8_t *__restrict__ player_out, const int nt,
                                                     const int32_t *__restrict__ redo_list, const int32_t *__restrict__ redo_count) {
    using G = Geo<R_, C_, VAR>;
    static_assert(G::LPG == 64, "one game per wave");
    static_assert(KINDX < 0 || (VAR == 1 && OBS && !MAPPED && KINDX >= 1 && KINDX <= 3), "other kinds: general-state pass, observing, unmapped");
    constexpr int NT = states_threads<MAPPED, OBS>(), KIND = KINDX >= 0 ? KINDX : (OBS ? 0 : 2);
    __shared__ StateLds<G, NT> W;
    __shared__ Lds<G, ObsKind<KIND>::NIB_CH> L;
    __shared__ alignas(16) uint8_t shared[OBS ? shared_table_bytes<G, KIND>() : 16];
    __shared__ alignas(16) uint8_t obst_s[G::OBST_BYTES + COMBAT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    // first pass: one block per state (mapped onto the XCDs like the step's workgroups); general-state pass: a small persistent grid that
    // walks the list of the states the first pass had to alter (usually empty: the launch costs a few microseconds, not one block per state)
    int64_t env = 0;
    int n_redo = 0, k_redo = (int)blockIdx.x;
    if constexpr (G::BIG) {
        n_redo = *redo_count;
        if (k_redo >= n_redo) return;
    } else {
        env = P.env_first + group_of_block(P);
        if (env >= P.n_envs) return;
    }
    {   // the step's workgroup-shared tables (game_kernel_body): default-code templates, code table, obstacle map, combat outcomes
        if constexpr (KINDX >= 0) {
            stage_kind_tables<G, KIND>(P, shared, tid, NT);
        } else if constexpr (OBS) {
            const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
            constexpr int NP = tmpl_lds_bytes<G, 0>(false);
            const int4 *tp = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0]);
            for (int i = tid; i < NP / 16; i += NT) reinterpret_cast<int4 *>(shared)[i] = tp[i];
            const int4 *ct = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0]);
            for (int i = tid; i < CODETAB_BYTES / 16; i += NT) reinterpret_cast<int4 *>(shared + NP)[i] = ct[i];
        }
        for (int i = tid; i < G::S / 4; i += NT) reinterpret_cast<int *>(obst_s)[i] = reinterpret_cast<const int *>(P.tab->obstacles)[i];
        for (int i = tid; i < COMBAT_BYTES / 4; i += NT) reinterpret_cast<int *>(obst_s + G::OBST_BYTES)[i] = reinterpret_cast<const int *>(P.tab->combat)[i];
    }
    for (;;) {
        if constexpr (G::BIG) env = redo_list[k_redo];
        import_to_image(P, W, in, player_in, sanitised, env, tid);              // (ends with a barrier: image and tables are in place)
        if (tid < 64) {
            const GameInput gin = load_game_from<G>(P, reinterpret_cast<const int4 *>(W.img), env, lane);
            env_step<R_, C_, KIND, MAPPED, false, VAR>(P, L, shared, obst_s, env, lane, gin, reinterpret_cast<int8_t *>(W.img));
        }
        __syncthreads();
        if constexpr (!G::BIG) {                                                   // (a general-state image does not fit the handle's packed records)
            int8_t *rec = P.boards + env * (int64_t)P.rec_bytes;                   // the handle keeps the successor, like after sgx_step
            for (int i = tid; i < P.rec_bytes / 16; i += NT) reinterpret_cast<int4 *>(rec)[i] = reinterpret_cast<const int4 *>(W.img)[i];
        }
        if (out) export_from_image(P, W, out, player_out, env, tid, nt);
        if constexpr (!G::BIG) break;
        k_redo += (int)gridDim.x;
        if (k_redo >= n_redo) break;
        __syncthreads();                                                           // (the next state reuses W and L)
    }
}

// ---------------------------------------------------------------------------------------------
// sgx_decode_obs / sgx_decode_mask: compact outputs (SGX_STEP_COMPACT_OBS / _MASK) -> the contract's float32 observation / uint8 mask.
// The same lanes-per-game geometry and the very emit_codes / patch_uncoded / emit_mask functions of the step kernel, fed from the compact
// records instead of a freshly built code buffer: the result is byte-identical to what the step would have written.
// ---------------------------------------------------------------------------------------------
template <int R_, int C_>
__global__ __launch_bounds__((64 * Geo<R_, C_>::WPB)) void decode_obs_kernel(const uint8_t *__restrict__ compact, int stride, int capacity, float *__restrict__ obs,
                                                                           int64_t n_envs, int nt) {
    using G = Geo<R_, C_>;
    using Spec = PartialObs;
    __shared__ Lds<G> LW[G::WPB * G::GPW];
    const int lane = threadIdx.x & (G::LPG - 1), slot = threadIdx.x / G::LPG;
    const int64_t env = blockIdx.x * (int64_t)(G::WPB * G::GPW) + slot;
    if (env >= n_envs) return;
    Lds<G> &L = LW[slot];
    const int n_unc = load_compact<G, Spec>(L, compact + env * (int64_t)stride, capacity, lane);
    float *dst = obs + env * (int64_t)(G::RC * Spec::NCH);
    if constexpr (G::RC % 4 == 0) {
        if (n_unc == 0) {
            if (nt) emit_codes<G, Spec, false, true>(L, dst, lane);
            else emit_codes<G, Spec, false, false>(L, dst, lane);
        } else {
            if (nt) emit_codes<G, Spec, true, true>(L, dst, lane);
            else emit_codes<G, Spec, true, false>(L, dst, lane);
            patch_uncoded<G, Spec>(L, dst, n_unc, lane);
        }
    } else {
        if (nt) emit_codes<G, Spec, false, true>(L, dst, lane);
        else emit_codes<G, Spec, false, false>(L, dst, lane);
        if (n_unc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            patch_uncoded_floats(L, dst, n_unc, lane);
        }
    }
}

template <int R_, int C_>
__global__ __launch_bounds__((64 * Geo<R_, C_>::WPB)) void decode_mask_kernel(const uint32_t *__restrict__ bits, uint8_t *__restrict__ mask, int64_t n_envs) {
    using G = Geo<R_, C_>;
    __shared__ Lds<G> LW[G::WPB * G::GPW];
    const int lane = threadIdx.x & (G::LPG - 1), slot = threadIdx.x / G::LPG;
    const int64_t env = blockIdx.x * (int64_t)(G::WPB * G::GPW) + slot;
    if (env >= n_envs) return;
    Lds<G> &L = LW[slot];
    const int4 *src = reinterpret_cast<const int4 *>(bits + env * (int64_t)G::MB_WORDS);
    for (int i = lane; i < G::MB_WORDS / 4; i += G::LPG) reinterpret_cast<int4 *>(L.mbits)[i] = src[i];
    wave_sync<G>();
    emit_mask(L, mask + env * (int64_t)G::NA, lane);
}

// sgx_copy_envs: packed records between two handles of the same variant, one wave per record
__global__ __launch_bounds__(256) void copy_records_kernel(int8_t *__restrict__ dst, const int32_t *__restrict__ dst_idx, const int8_t *__restrict__ src,
                                                           const int32_t *__restrict__ src_idx, int rec_bytes, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)(blockDim.x / 64) + threadIdx.x / 64;
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const int4 *s = reinterpret_cast<const int4 *>(src + (int64_t)(src_idx ? src_idx[i] : i) * rec_bytes);
    int4 *d = reinterpret_cast<int4 *>(dst + (int64_t)(dst_idx ? dst_idx[i] : i) * rec_bytes);
    for (int q = lane; q < rec_bytes / 16; q += 64) d[q] = s[q];
}

__global__ void info_kernel(const int8_t *__restrict__ boards, int rec_bytes, int sc_off, int32_t *__restrict__ out, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 s = *reinterpret_cast<const int4 *>(boards + i * (int64_t)rec_bytes + sc_off);
    reinterpret_cast<int4 *>(out)[i] = make_int4(s.x, s.w, (s.y & F_OVER) ? 1 : 0, (s.y & F_PLAYER_M1) ? -1 : 1);
}

__global__ void init_scal_kernel(int8_t *boards, int rec_bytes, int sc_off, int64_t n, int max_turns) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        int4 *sc = reinterpret_cast<int4 *>(boards + i * (int64_t)rec_bytes + sc_off);
        sc[0] = make_int4(0, 0, max_turns, -1);
        sc[1] = make_int4(0, 0, 0, 0);
    }
}

}  // namespace
