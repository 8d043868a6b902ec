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
// export / import in the reference's int64 [N,34,R,C] layout (impl:16-60)
// ---------------------------------------------------------------------------------------------
template <int R_, int C_>
__global__ void export_kernel(const KParams P, int64_t *__restrict__ out, int8_t *__restrict__ player_out) {
    using G = Geo<R_, C_>;
    constexpr int RC = G::RC, C = G::C, S = G::S;
    const int64_t env = blockIdx.x;
    if (env >= P.n_envs) return;
    const int8_t *rec = P.boards + env * (int64_t)P.rec_bytes;
    const int4 sc = rec_scal<G>(P.boards, P.rec_bytes, env)[0], sc2 = rec_scal<G>(P.boards, P.rec_bytes, env)[1];
    int64_t *o = out + env * (int64_t)(SGX_STATE_LAYERS * RC);
    for (int x = threadIdx.x; x < SGX_STATE_LAYERS * RC; x += blockDim.x) {
        const int l = x / RC, cell = x - l * RC;
        int64_t v = 0;
        if (l == 0 || l == 1) v = rec[(B_PIECES + l) * S + cell];
        else if (l == 2) v = P.tab->obstacles[cell];
        else if (l == 3 || l == 4) v = rec[(B_PO + l - 3) * S + cell];
        else if (l == 32 || l == 33)
            v = (reinterpret_cast<const uint32_t *>(rec + G::ST_OFF)[(l - 32) * (G::SB / 4) + (cell >> 5)] >> (cell & 31)) & 1u;
        else if (l == 5) {
            const int w = (sc.y & F_WIN_P1) ? 1 : (sc.y & F_WIN_M1) ? -1 : 0;
            if (cell == 0) v = sc.x;                               // TURN_COUNT  [5,0,0]
            else if (cell == 1) v = (sc.y & F_OVER) ? 1 : 0;       // GAME_OVER   [5,0,1]
            else if (cell == 2) v = w;                             // WINNER      [5,0,2]
            else if (cell == C) v = sc.z;                          // MAX_TURNS   [5,1,0]
            else if (cell == C + 1) v = (sc.y & F_END_INVALID) ? 1 : 0;  // ENDING_INVALID [5,1,1]
        }
        o[x] = v;   // recent-moves and captured layers start at 0 and are filled below
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int pl = 0; pl < 2; ++pl)
            for (int h = 0; h < 2; ++h) {
                const int pr = ((pl ? sc2.z : sc2.y) >> (16 * h)) & 0xFFFF;
                if (pr >> 8) o[(6 + pl) * RC + (pr & 0xFF)] = (int8_t)(pr >> 8);
            }
        const uint16_t *ev = reinterpret_cast<const uint16_t *>(rec + G::EVL_OFF);
        for (int i = 0; i < sc2.x; ++i) o[(8 + ((ev[i] >> 8) & 31)) * RC + (ev[i] & 0xFF)] += (ev[i] >> EV_COUNT_SHIFT) + 1;
        if (player_out) player_out[env] = (sc.y & F_PLAYER_M1) ? -1 : 1;
    }
}

// Reachable states only: at most two non-zero recent-move cells per player (impl:1013-1028), at most max_events (layer, cell)
// pairs with captured pieces and at most 8 of them on one pair; anything beyond that cannot come from play and is dropped.
// One 256-thread block per state: a single coalesced pass over the 34 int64 layers scatters them into LDS (dense boards
// straight into the record image, never-moved flags, recent-move codes and captured counts as bytes), the capture-event list
// is laid out with a block-wide prefix sum, and the finished record leaves as whole 128-byte lines.  (The first version
// walked the 24 captured layers with one thread: 4.6 ms per 65,536 states against 0.44 ms for the export.)
template <int R_, int C_>
__global__ __launch_bounds__(256) void import_kernel(const KParams P, const int64_t *__restrict__ in, const int8_t *__restrict__ player_in,
                                                     uint8_t *__restrict__ sanitised) {
    using G = Geo<R_, C_>;
    constexpr int RC = G::RC, C = G::C, S = G::S, NT = 256;
    constexpr int IMG = (G::EVL_OFF + 2 * G::EVL_MAX + 127) & ~127;      // >= rec_bytes of any piece set on this board
    constexpr int NE = 24 * RC, PER = (NE + NT - 1) / NT;
    __shared__ alignas(16) uint8_t img[IMG];
    __shared__ uint8_t cap[NE];
    __shared__ int8_t recent[2 * RC];
    __shared__ uint8_t still[2 * RC];
    __shared__ int scan[NT / 64];
    __shared__ int altered;       // the state held something the packed record cannot carry: reported through sanitised[]
    const int tid = threadIdx.x;
    const int64_t env = blockIdx.x;
    if (env >= P.n_envs) return;
    if (tid == 0) altered = 0;     // (ordered before every `altered = 1` below by the __syncthreads() that follows the img clear)
    int8_t *rec = P.boards + env * (int64_t)P.rec_bytes;
    const int64_t *s = in + env * (int64_t)(SGX_STATE_LAYERS * RC);
    // all of the thread's loads first (the scatter below is branchy, the compiler would otherwise wait for each load in turn:
    // 13 dependent round trips per block made the kernel latency-bound at 2.4 TB/s)
    constexpr int NX = SGX_STATE_LAYERS * RC, ITER = (NX + NT - 1) / NT;
    int64_t rawv[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int x = tid + k * NT;
        rawv[k] = x < NX ? s[x] : 0;
    }
    for (int i = tid; i < IMG / 4; i += NT) reinterpret_cast<uint32_t *>(img)[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int x = tid + k * NT;
        if (x >= NX) continue;
        const int l = x / RC, cell = x - l * RC;
        if (l == 5) continue;                                         // scalars below
        const int64_t raw = rawv[k];
        bool ok = true;
        if (l == 2) ok = raw == (int64_t)P.tab->obstacles[cell];      // obstacles are the variant's (a per-handle constant)
        else if (l < 2 || l == 3 || l == 4) {                         // legal range of the layer; anything else -> 0
            const int b = l < 2 ? B_PIECES + l : B_PO + (l - 3), hi = l < 2 ? SP_BOMB : SP_UNKNOWN;
            ok = raw >= 0 && raw <= hi;
            img[b * S + cell] = (uint8_t)(ok ? (int)raw : 0);
        } else if (l == 6 || l == 7) { ok = raw >= -3 && raw <= 1; recent[(l - 6) * RC + cell] = (int8_t)(ok ? (int)raw : 0); }
        else if (l < 32) { ok = raw >= 0 && raw <= EV_COUNT_MAX; cap[(l - 8) * RC + cell] = (uint8_t)(raw <= 0 ? 0 : (raw > EV_COUNT_MAX ? EV_COUNT_MAX : (int)raw)); }
        else { ok = raw == 0 || raw == 1; still[(l - 32) * RC + cell] = raw == 1 ? 1 : 0; }
        if (!ok) altered = 1;
    }
    __syncthreads();
    for (int w = tid; w < 2 * (G::SB / 4); w += NT) {                  // never-moved bitmaps from layers 32/33
        const int pl = w / (G::SB / 4), w0 = w - pl * (G::SB / 4);
        uint32_t bits = 0;
        for (int k = 0; k < 32; ++k) {
            const int cell = 32 * w0 + k;
            if (cell < RC && still[pl * RC + cell]) bits |= 1u << k;
        }
        reinterpret_cast<uint32_t *>(img + G::ST_OFF)[w] = bits;
    }
    // capture events (one per non-zero count) in (layer, cell) order: thread t owns entries [t*PER, (t+1)*PER) of the count table
    int cnt = 0;
    for (int k = 0; k < PER; ++k) {
        const int e = tid * PER + k;
        if (e < NE) cnt += cap[e] != 0;
    }
    // block-wide inclusive scan: shuffle scan inside each wave, then the four wave totals through LDS
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) scan[tid >> 6] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < (tid >> 6); ++w) wave_base += scan[w];
    const int total_events = scan[0] + scan[1] + scan[2] + scan[3];
    {
        int at = wave_base + incl - cnt;
        uint16_t *ev = reinterpret_cast<uint16_t *>(img + G::EVL_OFF);
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            if (e < NE && cap[e] != 0) {
                if (at < P.max_events) ev[at] = (uint16_t)(((cap[e] - 1) << EV_COUNT_SHIFT) | ((e / RC) << 8) | (e % RC));
                ++at;
            }
        }
    }
    if (tid == 0) {
        const int64_t *d = s + 5 * RC;
        int flags = 0;
        if (d[1] != 0) flags |= F_OVER;
        if (d[2] > 0) flags |= F_WIN_P1; else if (d[2] < 0) flags |= F_WIN_M1;
        if (d[C + 1] != 0) flags |= F_END_INVALID;
        if (player_in && player_in[env] < 0) flags |= F_PLAYER_M1;
        int pairs[2] = {0, 0};
        bool dropped = total_events > P.max_events;
        for (int pl = 0; pl < 2; ++pl) {
            int k = 0;
            for (int cell = 0; cell < RC; ++cell) {
                const int code = recent[pl * RC + cell];
                if (code == 0) continue;
                if (k < 2) pairs[pl] |= (cell | ((code & 0xFF) << 8)) << (16 * k);
                else dropped = true;                                  // more than two recent-move cells cannot come from play
                ++k;
            }
        }
        if (player_in && player_in[env] != 1 && player_in[env] != -1) dropped = true;
        if (sanitised) sanitised[env] = (altered || dropped) ? 1 : 0;
        const int old_game = rec_scal<G>(P.boards, P.rec_bytes, env)[0].w;
        int4 *scg = reinterpret_cast<int4 *>(img + G::SC_OFF);
        scg[0] = make_int4((int)d[0], flags, (int)d[C], old_game < 0 ? 0 : old_game);
        scg[1] = make_int4(min(total_events, P.max_events), pairs[0], pairs[1], 0);
    }
    __syncthreads();
    for (int i = tid; i < P.rec_bytes / 16; i += NT) reinterpret_cast<int4 *>(rec)[i] = reinterpret_cast<const int4 *>(img)[i];
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
