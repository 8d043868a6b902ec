// sgx_obs.h -- observation rendering: 4-bit code buffer + line-aligned emission ('extended' channels), LUT emission ('original')
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Observation render: float32 [R][C][NCH], perspective of player index qi
// (partial: impl:1335-1397, full: impl:1230-1303; normalisation maenv:499-508)
// ---------------------------------------------------------------------------------------------
#ifndef SGX_OBS_UNROLL
#define SGX_OBS_UNROLL 4
#endif

// ---- 'extended' channels (67 / 79): rendered from 4-bit codes -------------------------------------------------------------
// 41 of the 67 partial channels (53 of 79) are 0 / 1 indicators of one board byte (piece one-hots, obstacle, never-moved), and
// the others are almost always at their default (captured counts 0, recent-move code 0).  The first version looked every float
// up through (board byte, LUT row): 9 LDS + 27 VALU instructions per 16-byte store, the kernel's largest phase on every board
// size.  Now: (1) the game's code buffer L.nib is filled from the variant's template of channel defaults (whole 16-byte copies);
// (2) the lane owning a cell ORs the code of 1.0 into the <= 8 indicator entries that are set; (3) a quad of the output is ONE
// 16-bit LDS read and four v_cvt_off_f32_i4 (x 4) -- 1-2 LDS + ~18 VALU per store; (4) the few entries that are neither default
// nor 1.0 (a captured count >= 1, a non-zero recent-move code) are written afterwards as single floats (emit_obs_patches),
// after the wave has waited for its bulk stores.

// 4-bit code -> float: v_cvt_off_f32_i4 gives sext(code) / 16
__device__ inline float code_to_float(unsigned x) { return __builtin_amdgcn_cvt_off_f32_i4((int)x) * 4.0f; }

template <class G, int NB>
__device__ inline void set_code_one(Lds<G, NB> &L, int entry) {
    atomicOr(reinterpret_cast<unsigned int *>(L.nib) + (entry >> 3), (unsigned)NIB_ONE << ((entry & 7) * 4));
}
// replaces the default code `was` of an entry by `now`
template <class G, int NB>
__device__ inline void swap_code(Lds<G, NB> &L, int entry, int was, int now) {
    atomicXor(reinterpret_cast<unsigned int *>(L.nib) + (entry >> 3), (unsigned)(was ^ now) << ((entry & 7) * 4));
}

// entry (cell, channel, LUT value index) of capture event / recent-move pair number i (events first, then the four pairs) in
// Spec's observation from perspective qi; false for an empty pair
template <class G, class Spec, int NB>
__device__ inline bool special_entry(const Lds<G, NB> &L, int i, int n_events, int rp0, int rp1, int qi, int &entry, int &ch, int &v, int &tab_idx) {
    const typename G::ev_t *evl = reinterpret_cast<const typename G::ev_t *>(L.tail + 2 * G::SB + 32);
    int cell;
    if (i < n_events) {
        const int ev = (int)evl[i], b = (ev >> G::CELL_BITS) & 31, pi = b >= 12 ? 1 : 0, t = b - 12 * pi;
        cell = ev & G::CELL_MASK;
        ch = Spec::CAP0 + (pi == qi ? 0 : 12) + t;
        v = (ev >> G::EV_COUNT_SHIFT) + 1;
        tab_idx = 16 * t + v;
    } else if constexpr (G::BIG) {                // general-state image: every cell of the two dense recent-move boards
        const int k = i - n_events, pl = k >= G::RC ? 1 : 0;
        cell = k - pl * G::RC;
        const int code = L.b[B_RECENT + pl][cell];
        if (code == 0) return false;
        ch = Spec::REC0 + (pl == qi ? 0 : 1);
        v = code + 3;
        tab_idx = CODETAB_REC + v;
    } else {
        const int k = i - n_events, pl = k >> 1, pr = (((pl ? rp1 : rp0) >> (16 * (k & 1))) & 0xFFFF);
        const int code = G::pair_code(pr);
        if (code == 0) return false;
        cell = G::pair_cell(pr);
        ch = Spec::REC0 + (pl == qi ? 0 : 1);
        v = code + 3;
        tab_idx = CODETAB_REC + v;
    }
    entry = (qi ? G::RC - 1 - cell : cell) * Spec::NCH + ch;
    return true;
}

// Events of one (layer, cell) key when a variant has more than 8 pieces of a type (KParams::multi_ev): event i speaks for its key if no
// earlier event has it; its count then is the sum over all events of the key.  Returns false for a later duplicate.
template <class G, int NB>
__device__ inline bool merged_event(const Lds<G, NB> &L, int i, int n_events, int &v) {
    const typename G::ev_t *evl = reinterpret_cast<const typename G::ev_t *>(L.tail + 2 * G::SB + 32);
    const int key = (int)(evl[i] & G::EV_KEY_MASK);
    for (int j = 0; j < i; ++j)
        if ((int)(evl[j] & G::EV_KEY_MASK) == key) return false;
    for (int j = i + 1; j < n_events; ++j)
        if ((int)(evl[j] & G::EV_KEY_MASK) == key) v += (int)(evl[j] >> G::EV_COUNT_SHIFT) + 1;
    return true;
}

// Fills L.nib with the codes of Spec's observation from player index qi's perspective.  tmpl = the variant's default codes,
// codetab = codes of captured counts / recent-move codes (both workgroup-shared LDS copies), glut = this kind's LUT in global
// memory.  Returns the number of entries whose value has no code (L.unc_entry / L.unc_val; the same for every lane of the game).
template <class G, class Spec, int NB, class PC>     // PC: pointer to the variant's piece counts (KParams::piece_counts, possibly in the kernel-argument address space)
__device__ inline int build_codes(Lds<G, NB> &L, const uint8_t *tmpl, const uint8_t *codetab, const float *__restrict__ glut, int qi, int n_events,
                                  int rp0, int rp1, int lane, PC piece_counts, bool raw = false, bool multi = false) {
    constexpr int RC = G::RC, NCH = Spec::NCH, NBYTES = ((RC * NCH + 1) / 2 + 15) & ~15;
    static_assert(NBYTES <= Lds<G, NB>::NIB_BYTES, "code buffer too small for this observation kind");
    for (int i = lane; i < NBYTES / 16; i += G::LPG) reinterpret_cast<int4 *>(L.nib)[i] = reinterpret_cast<const int4 *>(tmpl)[i];
    wave_sync<G>();
#pragma unroll
    for (int cc = 0; cc < G::CPL; ++cc) {
        const int i = lane + G::LPG * cc;
        if (i < RC) {
            const int base = (qi ? RC - 1 - i : i) * NCH;
            const int own = L.b[B_PIECES + qi][i], en = L.b[B_PIECES + 1 - qi][i];
            const int own_po = L.b[B_PO + qi][i], en_po = L.b[B_PO + 1 - qi][i];
            if (own) set_code_one(L, base + Spec::OWN0 + own - 1);
            if constexpr (Spec::ENEMY0 >= 0)
                if (en) set_code_one(L, base + Spec::ENEMY0 + en - 1);
            if (own_po) set_code_one(L, base + Spec::OWN_PO0 + own_po - 1);
            if (en_po) set_code_one(L, base + Spec::ENEMY_PO0 + en_po - 1);
            if (L.b[B_OBST][i]) set_code_one(L, base + Spec::OBST);
            if (L.b[B_STILL + qi][i]) set_code_one(L, base + Spec::STILL0);
            if (L.b[B_STILL + 1 - qi][i]) set_code_one(L, base + Spec::STILL0 + 1);
        }
    }
    // captured counts / recent-move codes: their code replaces the channel default; a value without a code gets CODE_ESC and
    // goes on the game's list of uncoded entries (float from the LUT in global memory: a few lanes, once per step)
    int n_unc = 0;
    for (int i0 = 0; i0 < n_events + G::SPECIAL_EXTRA; i0 += G::LPG) {
        const int i = i0 + lane;
        int entry = 0, ch, v, ti;
        bool unc = false;
        float val = 0.f;
        bool have = i < n_events + G::SPECIAL_EXTRA && special_entry<G, Spec>(L, i, n_events, rp0, rp1, qi, entry, ch, v, ti);
        if (multi && have && i < n_events) {                       // (variants with more than 8 pieces of a type only)
            have = merged_event(L, i, n_events, v);
            ti = (ti & ~15) + (v < LUT_STRIDE ? v : 0);
        }
        if (have) {
            int now, was;
            if ((G::BIG || multi) && i < n_events && v >= LUT_STRIDE) {
                // a count beyond the 16-entry LUT (general states only): the reference's arithmetic itself, (x - mid) / range in float32 with
                // mid = range = hi / 2, hi = the type's piece count if > 1 else 8 (maenv:288-298, 506-508); raw observations hold the count
                const int t = (ch - Spec::CAP0) % 12, pc = piece_counts[t];
                const float hi = pc > 1 ? (float)pc : 8.0f, half = hi / 2.0f;
                val = raw ? (float)v : __fdiv_rn(__fsub_rn((float)v, half), half);
                now = CODE_NONE;
                was = codetab[16 * t];
            } else {
                now = codetab[ti];
                was = codetab[i < n_events ? (ti & ~15) : CODETAB_REC + 3];   // the default: count 0 / code 0
                if (now == CODE_NONE) val = glut[lut_row(ch) + v];
            }
            if (now == CODE_NONE) { unc = true; now = CODE_ESC; }
            swap_code(L, entry, was, now);
        }
        const unsigned long long bal = gballot<G>(unc);
        if (unc) {
            const int pos = n_unc + __popcll(bal & ((1ull << lane) - 1ull));
            L.unc_entry[pos] = (typename G::entry_t)entry;
            L.unc_val[pos] = val;
        }
        n_unc += __popcll(bal);
    }
    wave_sync<G>();
    return n_unc;
}

// Compact observation (SGX_STEP_COMPACT_OBS): the game's code buffer itself instead of the floats it decodes to -- NIB_BYTES of 4-bit codes,
// a 16-byte header {n_unc, 0, 0, 0}, then n_unc x {uint32 entry, float value} for the entries whose value has no code (CODE_ESC in
// the buffer).  sgx_decode_obs expands it with the very emit_codes / patch path below, so the float32 result is byte-identical to what
// the step would have written.  `dst` is 16-byte aligned (the record stride is a multiple of 128).
template <class G, class Spec, int NB>
__device__ inline void store_compact(const Lds<G, NB> &L, uint8_t *__restrict__ dst, int n_unc, int lane) {
    constexpr int NBYTES = ((G::RC * Spec::NCH + 1) / 2 + 15) & ~15;
    for (int i = lane; i < NBYTES / 16; i += G::LPG) reinterpret_cast<int4 *>(dst)[i] = reinterpret_cast<const int4 *>(L.nib)[i];
    if (lane == 0) *reinterpret_cast<int4 *>(dst + NBYTES) = make_int4(n_unc, 0, 0, 0);
    uint2 *ent = reinterpret_cast<uint2 *>(dst + NBYTES + 16);
    for (int k = lane; k < n_unc; k += G::LPG) ent[k] = make_uint2((uint32_t)L.unc_entry[k], __float_as_uint(L.unc_val[k]));
}
// the reverse: compact record -> L.nib / L.unc_*; returns n_unc
template <class G, class Spec, int NB>
__device__ inline int load_compact(Lds<G, NB> &L, const uint8_t *__restrict__ src, int capacity, int lane) {
    constexpr int NBYTES = ((G::RC * Spec::NCH + 1) / 2 + 15) & ~15;
    for (int i = lane; i < NBYTES / 16; i += G::LPG) reinterpret_cast<int4 *>(L.nib)[i] = reinterpret_cast<const int4 *>(src)[i];
    int n_unc = reinterpret_cast<const int4 *>(src + NBYTES)->x;
    n_unc = n_unc < 0 ? 0 : (n_unc > capacity ? capacity : n_unc);
    const uint2 *ent = reinterpret_cast<const uint2 *>(src + NBYTES + 16);
    for (int k = lane; k < n_unc; k += G::LPG) {
        const uint2 e = ent[k];
        L.unc_entry[k] = (typename G::entry_t)e.x;
        L.unc_val[k] = __uint_as_float(e.y);
    }
    wave_sync<G>();
    return n_unc;
}

// L.nib -> global.  The observation is written in 1 KiB chunks aligned to 1 KiB ADDRESS boundaries (whole 128-byte lines per
// store instruction; chunking by cell group left two partial lines per store and ran 1.5x slower in the store-pattern probe).
// CHECKED (games with uncoded entries, 4-aligned boards): a quad that holds a CODE_ESC entry is not stored here -- patch_uncoded
// writes it whole, so no address is written twice and nothing has to be waited for.
// NT: lines this wave writes whole leave as non-temporal stores -- chosen per launch, see KParams::nt_stores.
// STRIDE: lanes that share the sweep -- the game's LPG lanes, or every thread of the workgroup (`lane` = thread index) when a whole
// workgroup emits one game (single_kernel).
template <class G, class Spec, bool CHECKED, bool NT, int STRIDE = G::LPG, int NB>
__device__ inline void emit_codes(const Lds<G, NB> &L, float *__restrict__ dst, int lane) {
    constexpr int RC = G::RC, NCH = Spec::NCH;
    const uint16_t *n16 = reinterpret_cast<const uint16_t *>(L.nib);
    if constexpr (RC % 4 == 0) {
        constexpr int NQ = (RC / 4) * NCH;                                           // quads (16 B) of one observation
        const int m0 = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & (G::LPG - 1));  // quads past a chunk boundary
        f32x4 *base = reinterpret_cast<f32x4 *>(dst);
        // NT: lines written whole by this wave leave as non-temporal stores (they are never read back by this kernel and need no
        // merging: 325.8 -> 281.1 us per launch of 65,536 Barrage games, 137 -> 117 us on 6x6, 244 -> 198 us on 8x8, in-process A/B).
        // The first and the last 128-byte line of a game's observation are shared with the neighbouring games (other waves): those
        // go through L2 so that the halves merge (everything non-temporal: 316 us).  It pays when the launch's observations do
        // not fit the 256 MiB Infinity Cache and costs 5-20 % when they do (8,192 Barrage games 41 vs 50 us, 65,536 Micro games
        // 42 vs 47 us; 16,384 Barrage games 102 vs 87 us, 131,072 Micro games 100 vs 82 us): the host decides per launch.
        const int l0 = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7);               // quads past a 128-byte line
        const int first_line = l0 ? 0 : -1, last_line = ((l0 + NQ) & 7) ? (NQ - 1 + l0) >> 3 : -1;   // lines counted from dst - 16 * l0
        auto sweep = [&](const int q0) __attribute__((always_inline)) {
            const int q = q0 + lane;
            const bool in = (unsigned)q < (unsigned)NQ;
            const unsigned x = n16[in ? q : 0];
            f32x4 o = {code_to_float(x), code_to_float(x >> 4), code_to_float(x >> 8), code_to_float(x >> 12)};
            bool esc = false;
            if constexpr (CHECKED) {
                const unsigned y = x ^ (0x1111u * CODE_ESC);                           // a CODE_ESC nibble becomes 0
                esc = ((y - 0x1111u) & ~y & 0x8888u) != 0;                            // some nibble of y is 0
            }
            if constexpr (NT) {
                bool edge = ((q + l0) >> 3) == first_line || ((q + l0) >> 3) == last_line;
                if constexpr (CHECKED) {
                    // a line with a left-out quad is completed by patch_uncoded's 16-byte store: both parts go through L2, which merges them
                    // into one whole-line write-back; non-temporal, the line reached memory as two partial writes (Standard 300 steps into
                    // the games, most envs with captured miners / majors / bombs, same buffers: 262,144 games 1,387 us against 1,328 us with
                    // plain stores everywhere, tools/nt_ab.py).  A line = 8 consecutive lanes of the sweep.
                    const unsigned line_esc = (unsigned)(__ballot(in && esc) >> (__lane_id() & ~7)) & 0xFFu;
                    edge = edge || line_esc != 0;
                }
                if (in && !esc && edge) base[q] = o;
                if (in && !esc && !edge) __builtin_nontemporal_store(o, &base[q]);
            } else {
                if (in && !esc) base[q] = o;
            }
        };
        if constexpr (CHECKED && NT) {
            // (a ballot per iteration: a convergent operation, which the optimiser cannot unroll under a run-time trip count -- asking for
            // it only produced a -Wpass-failed warning per instantiation)
            for (int q0 = -m0; q0 < NQ; q0 += STRIDE) sweep(q0);
        } else {
#pragma unroll SGX_OBS_UNROLL
            for (int q0 = -m0; q0 < NQ; q0 += STRIDE) sweep(q0);
        }
    } else {
        static_assert(!CHECKED, "odd boards patch single floats after a wait (patch_uncoded_floats)");
        static_assert(STRIDE == G::LPG, "odd boards are emitted by the game's own lanes");
        // odd cell counts (5x5, 15x15): an env's observation is only 4-byte aligned.  Lanes own the 16-byte slots of the
        // ADDRESS range; slot k holds floats 4k-a .. 4k-a+3 (a = floats past a 16-byte boundary), i.e. 16 code bits that start
        // (4-a) nibbles into halfword k-1: two halfword reads and a shift.  Whole slots leave as one 16-byte store, the partial
        // first / last slot as dwords.
        constexpr int NF = RC * NCH;
        const int a = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
        float *base = dst - a;
        const int nslots = (a + NF + 3) >> 2;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(base) >> 4) & (G::LPG - 1));
        const int sh = a ? (4 - a) * 4 : 0, back = a ? 1 : 0;
        const int l0 = (int)((reinterpret_cast<uintptr_t>(base) >> 4) & 7), last_line = (nslots - 1 + l0) >> 3;   // (both edge lines may be shared)
#pragma unroll 2
        for (int k0 = -m0; k0 < nslots; k0 += G::LPG) {
            const int k = k0 + lane;
            const bool slot_in = k >= 0 && k < nslots;
            const int kk = slot_in ? k : 0;
            // halfword -1 is L.nib_lead (its value never reaches a stored float)
            const unsigned w = (unsigned)n16[kk - back] | ((unsigned)n16[kk - back + 1] << 16);
            const unsigned x = w >> sh;
            const float o[4] = {code_to_float(x), code_to_float(x >> 4), code_to_float(x >> 8), code_to_float(x >> 12)};
            const int f0 = 4 * kk - a;
            if (slot_in && f0 >= 0 && f0 + 3 < NF) {
                f32x4 qv = {o[0], o[1], o[2], o[3]};
                const int line = (k + l0) >> 3;
                // (sending the lines that hold an uncoded entry through L2, like the 4-aligned boards do, measured 11 % SLOWER here:
                // 15x15, 32,768 games 250 steps in, 447 -> 497 us)
                if (NT && line != 0 && line != last_line) __builtin_nontemporal_store(qv, &reinterpret_cast<f32x4 *>(base)[k]);
                else reinterpret_cast<f32x4 *>(base)[k] = qv;
            } else if (slot_in) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((unsigned)(f0 + j) < (unsigned)NF) base[4 * k + j] = o[j];
            }
        }
    }
}

// 4-aligned boards: every quad that holds an uncoded entry, written whole (codes of the other entries + the floats of ALL uncoded
// entries in it; two lanes that own entries of the same quad store the same 16 bytes).
template <class G, class Spec, int NB>
__device__ inline void patch_uncoded(const Lds<G, NB> &L, float *__restrict__ dst, int n_unc, int lane) {
    static_assert(G::RC % 4 == 0, "");
    const uint16_t *n16 = reinterpret_cast<const uint16_t *>(L.nib);
    for (int k = lane; k < n_unc; k += G::LPG) {
        const int q = L.unc_entry[k] >> 2;
        const unsigned x = n16[q];
        float o[4] = {code_to_float(x), code_to_float(x >> 4), code_to_float(x >> 8), code_to_float(x >> 12)};
        for (int m = 0; m < n_unc; ++m) {
            const int em = L.unc_entry[m];
            if ((em >> 2) == q) {
                const float vm = L.unc_val[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (em & 3) == j ? vm : o[j];
            }
        }
        f32x4 qv = {o[0], o[1], o[2], o[3]};
        reinterpret_cast<f32x4 *>(dst)[q] = qv;
    }
}
// odd boards: the uncoded entries as single floats; the caller has waited for the bulk stores (s_waitcnt vmcnt(0)): these
// stores hit addresses other lanes have just written
template <class G, int NB>
__device__ inline void patch_uncoded_floats(const Lds<G, NB> &L, float *__restrict__ dst, int n_unc, int lane) {
    for (int k = lane; k < n_unc; k += G::LPG) dst[L.unc_entry[k]] = L.unc_val[k];
}

// 'original' kinds: the LUT path leaves the captured-count channels at their default; every capture event is written as a single
// float afterwards (`lut` = this kind's LUT in global memory).  The caller has waited for the bulk stores of `dst`.
template <class G, class Spec, int NB, class PC>
__device__ inline void emit_obs_patches(const Lds<G, NB> &L, const float *__restrict__ lut, int qi, float *__restrict__ dst, int n_events, int lane,
                                        bool raw, bool multi, PC piece_counts) {
    static_assert(!Spec::CODES, "");
    for (int i = lane; i < n_events; i += G::LPG) {
        int entry, ch, v, ti;
        if (special_entry<G, Spec>(L, i, n_events, 0, 0, qi, entry, ch, v, ti)) {
            if (multi && !merged_event(L, i, n_events, v)) continue;
            float val;
            if (v < LUT_STRIDE) val = lut[lut_row(ch) + v];
            else {
                // beyond the 16-entry LUT (more than 8 pieces of a type only): the reference's arithmetic, (x - mid) / range with
                // mid = range = hi / 2, hi = the type's piece count if > 1 else 2 in the 'original' channel modes (maenv:87-199)
                const int pc = piece_counts[(ch - Spec::CAP0) % 12];
                const float half = (pc > 1 ? (float)pc : 2.0f) / 2.0f;
                val = raw ? (float)v : __fdiv_rn(__fsub_rn((float)v, half), half);
            }
            dst[entry] = val;
        }
    }
}

// ---- 'original' channels (32 / 33 value channels): every float through (board byte, LUT row) -----------------------------------
// quad table entry of (perspective qi, quad qd, element j): LDS byte offset of the source board at the first 4-cell group
// (low 16 bits) and LUT index base (high 16 bits); built once per workgroup (build_quad_table)
template <class G, class Spec>
__device__ inline void build_quad_table(uint32_t *qtab, int tid, int nthreads) {
    constexpr int RC = G::RC, S = G::S, NCH = Spec::NCH;
    if constexpr (RC % 4 != 0) {
        // odd cell counts (5x5, 15x15): per-channel table instead -- entry (qi, ch) = LDS byte offset of the source board (low 16
        // bits) and LUT index base (high 16 bits); emit_obs_lut adds the cell
        for (int i = tid; i < 2 * NCH; i += nthreads) {
            const int qi = i / NCH, ch = i - qi * NCH;
            qtab[i] = (uint32_t)(Spec::board(ch, qi) * S) | ((uint32_t)(lut_row(ch) + Spec::bias(ch)) << 16);
        }
        return;
    }
    for (int i = tid; i < 2 * NCH * 4; i += nthreads) {
        const int qi = i / (NCH * 4), r = i - qi * (NCH * 4), f = r;            // f = 4*qd + j : float index inside a 4-cell group
        const int rc = f / NCH, ch = f - rc * NCH;
        const uint32_t boff = (uint32_t)(Spec::board(ch, qi) * S + (qi ? RC - 1 - rc : rc));
        const uint32_t lrow = (uint32_t)(lut_row(ch) + Spec::bias(ch));
        qtab[i] = boff | (lrow << 16);
    }
}

// `tab` = this observation kind's LUT followed by its quad table (workgroup-shared LDS).  Same chunking as emit_codes; a lane's
// quad changes every iteration, hence the quad table.  The captured-count channels read the all-zero board (their default);
// emit_obs_patches writes the counts that are not zero.
template <class G, class Spec, int NB>
__device__ void emit_obs_lut(const Lds<G, NB> &L, const float *tab, int qi, float *__restrict__ dst, int lane) {
    constexpr int RC = G::RC, NCH = Spec::NCH;
    const int8_t *bb = &L.b[0][0];
    const float *lut = tab;
    if constexpr (RC % 4 == 0) {
        constexpr int NQ = (RC / 4) * NCH;                                           // quads (16 B) of one observation
        const uint4 *qtab = reinterpret_cast<const uint4 *>(tab + LUT_DWORDS) + qi * NCH;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & (G::LPG - 1));  // quads past a 1 KiB boundary (LPG = 64)
        f32x4 *base = reinterpret_cast<f32x4 *>(dst);
        const int gstep = qi ? -4 : 4;
#pragma unroll SGX_OBS_UNROLL
        for (int q0 = -m0; q0 < NQ; q0 += G::LPG) {
            const int q = q0 + lane;
            const bool in = (unsigned)q < (unsigned)NQ;
            const int qq = in ? q : 0, g = qq / NCH, qd = qq - g * NCH;
            const uint4 e = qtab[qd];
            const int g4 = g * gstep;
            // board bytes are legal by construction (reset, move application, sanitised import): no clamp on the LUT index
            f32x4 o;
            o.x = lut[(e.x >> 16) + bb[(e.x & 0xFFFF) + g4]];
            o.y = lut[(e.y >> 16) + bb[(e.y & 0xFFFF) + g4]];
            o.z = lut[(e.z >> 16) + bb[(e.z & 0xFFFF) + g4]];
            o.w = lut[(e.w >> 16) + bb[(e.w & 0xFFFF) + g4]];
            if (in) base[q] = o;
        }
    } else {
        constexpr int NF = RC * NCH;
        const uint32_t *ctab = reinterpret_cast<const uint32_t *>(tab + LUT_DWORDS) + qi * NCH;
        const int a = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);             // floats past a 16-byte boundary
        float *base = dst - a;
        const int nslots = (a + NF + 3) >> 2;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(base) >> 4) & (G::LPG - 1));
#pragma unroll 2
        for (int k0 = -m0; k0 < nslots; k0 += G::LPG) {
            const int k = k0 + lane;
            const bool slot_in = k >= 0 && k < nslots;
            const int f0 = 4 * (slot_in ? k : 0) - a;
            float o[4];
            bool in[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = f0 + j;
                in[j] = slot_in && (unsigned)f < (unsigned)NF;
                const int ff = in[j] ? f : 0, pcell = ff / NCH, ch = ff - pcell * NCH;
                const uint32_t e = ctab[ch];
                o[j] = lut[(e >> 16) + bb[(e & 0xFFFF) + (qi ? RC - 1 - pcell : pcell)]];
            }
            if (in[0] && in[3]) {
                f32x4 q = {o[0], o[1], o[2], o[3]};
                reinterpret_cast<f32x4 *>(base)[k] = q;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (in[j]) base[4 * k + j] = o[j];
            }
        }
    }
}

}  // namespace
