// sgx_step.h -- the step / observe kernels: one env.step() per game
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// The step kernel: env.step() of N games (maenv:659-828), one wave per game
// ---------------------------------------------------------------------------------------------
// workgroup-shared tables: 'extended' kinds -- default-code templates (+ the small code table); 'original' kinds -- LUT + quad table
template <class G, int KIND>
constexpr int tmpl_bytes(bool full) { return ((G::RC * (full ? FOBS_CH : OBS_CH) + 1) / 2 + 15) & ~15; }
// bytes of the template(s) the workgroup holds in LDS: none on WIDE boards, which read the templates from global memory (a 32 x 32
// board's two templates would take 74 KB of the CU's 160 KB)
template <class G, int KIND>
constexpr int tmpl_lds_bytes(bool full) { return G::WIDE ? 0 : tmpl_bytes<G, KIND>(full); }
template <class G, int KIND>
constexpr int shared_table_bytes() {
    constexpr bool FULL = (KIND & 1) != 0, ORIG = (KIND & 2) != 0;
    if (ObsKind<KIND>::NOOBS) return 16;                       // no observation: no templates, no LUTs
    if (ORIG) return 4 * (OBS_TAB_DWORDS + (FULL ? FOBS_TAB_DWORDS : 0));
    return tmpl_lds_bytes<G, KIND>(false) + (FULL ? tmpl_lds_bytes<G, KIND>(true) : 0) + CODETAB_BYTES;
}

// waves per SIMD this geometry can reach: LDS per workgroup = WPB game regions + the shared LUT, 160 KiB per CU
template <class G, int KIND, bool MAPPED = false>
constexpr int waves_per_simd() {
    constexpr int per_wg = G::WPB * G::GPW * (int)sizeof(Lds<G, ObsKind<KIND>::NIB_CH>) + shared_table_bytes<G, KIND>() + G::OBST_BYTES + COMBAT_BYTES;
    constexpr int wgs = (160 * 1024) / per_wg;
    constexpr int w = wgs * G::WPB / 4;
    // toy boards are latency-bound (tiny per-game work): 8 waves/SIMD measured +7 %; on 10x10 forcing 64 VGPRs spills
    // (the MAPPED instantiation's index computations take ~70 VGPRs: with the no-observation kind's small LDS footprint a promise of 8
    //  waves would cap it at 64 and spill; 6 leaves it 80)
    // (two games per wave: per-lane values replace wave-uniform ones -- 79 VGPRs in the 10x10 steps loop -- and 6 waves are 12 games per SIMD)
    constexpr int want = (G::HALF || (MAPPED && ObsKind<KIND>::NOOBS)) ? 6 : (G::RC <= 64 ? 8 : SGX_MIN_WAVES);
    return w > want ? want : (w < 1 ? 1 : w);
}

// One more captured piece on layer / cell `key` ((12 * pi + type - 1) << CELL_BITS | cell): the count of its event goes up, or a new event
// is appended.  Returns the new number of events.
// multi (KParams::multi_ev): a full event of the key does not count as found -- the capture opens another event with the same key
template <class G, int NB>
__device__ inline int add_capture(Lds<G, NB> &L, int n_events, int max_events, int key, int lane, bool multi = false) {
    using ev_t = typename G::ev_t;
    ev_t *evl = reinterpret_cast<ev_t *>(L.tail + 2 * G::SB + 32);
    bool found = false;
    for (int i0 = 0; i0 < n_events; i0 += G::LPG) {                    // (wave-uniform bound for a 64-lane game)
        const int i = i0 + lane;
        const bool match = i < n_events && (int)(evl[i] & G::EV_KEY_MASK) == key;
        const bool room = match && (int)(evl[i] >> G::EV_COUNT_SHIFT) < G::COUNT_MAX - 1;
        if (room) evl[i] = (ev_t)(evl[i] + (1 << G::EV_COUNT_SHIFT));
        found = found || gballot<G>(multi ? room : match) != 0ull;
    }
    if (!found && n_events < max_events) {
        if (lane == 0) evl[n_events] = (ev_t)key;
        n_events += 1;
    }
    wave_sync<G>();
    return n_events;
}

// Builds the record image from ST_OFF on in L.tail (never-moved bitmaps from the LDS still boards, the two scalar int4s,
// the event list already kept in L.tail) and writes the whole record to HBM as 16-byte stores over whole 128-byte lines.
template <class G, int NB>
__device__ inline void write_record(Lds<G, NB> &L, int8_t *rec_g, int rec_bytes, int4 sc0, int4 sc1, int n_events, int lane) {
    constexpr int S = G::S, RC = G::RC;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int w = 0; w < G::SB / 8; ++w) {
            unsigned long long m = 0;
#pragma unroll
            for (int h = 0; h < G::GPW; ++h) {          // a game's ballot covers LPG cells
                const int i = 64 * w + G::LPG * h + lane;
                if (64 * w + G::LPG * h < RC) m |= gballot<G>(i < RC && L.b[B_STILL + pi][i < RC ? i : 0] != 0) << (G::LPG * h);
            }
            if (lane == 0) reinterpret_cast<unsigned long long *>(L.tail + pi * G::SB)[w] = m;
        }
    if (lane == 0) {
        reinterpret_cast<int4 *>(L.tail + 2 * G::SB)[0] = sc0;
        reinterpret_cast<int4 *>(L.tail + 2 * G::SB)[1] = sc1;
    }
    if constexpr (G::BIG) {                          // the recent-move boards stay dense in a general-state record image
        for (int i = lane; i < G::S_PAD; i += G::LPG) {
            L.tail[G::RECB_OFF - G::ST_OFF + i] = i < S ? (uint8_t)L.b[B_RECENT][i] : 0;
            L.tail[G::RECB_OFF - G::ST_OFF + G::S_PAD + i] = i < S ? (uint8_t)L.b[B_RECENT + 1][i] : 0;
        }
        for (int i = n_events + lane; i < G::EVL_MAX; i += G::LPG) reinterpret_cast<typename G::ev_t *>(L.tail + 2 * G::SB + 32)[i] = 0;
    }
    wave_sync<G>();
    const int4 *bsrc = reinterpret_cast<const int4 *>(&L.b[0][0]), *tsrc = reinterpret_cast<const int4 *>(L.tail);
    int4 *dst = reinterpret_cast<int4 *>(rec_g);
    const int n_tail_q = G::BIG ? G::TAIL_BYTES / 16 : (2 * G::SB + 32 + G::EV_BYTES * n_events + 15) >> 4;  // tail int4s that carry data
    for (int i = lane; i < rec_bytes / 16; i += G::LPG) {
        int4 v = make_int4(0, 0, 0, 0);
        if (i < G::ST_OFF / 16) {
            v = bsrc[i];
            const int keep = STORED_BOARDS * S - 16 * i;                       // bytes of this int4 that belong to the stored boards
            if (keep < 16) { if (keep <= 12) v.w = 0; if (keep <= 8) v.z = 0; if (keep <= 4) v.y = 0; if (keep <= 0) v.x = 0; }
        } else if (i < G::ST_OFF / 16 + n_tail_q) {
            v = tsrc[i - G::ST_OFF / 16];
        }
        dst[i] = v;        // (non-temporal: no change)
    }
}

// Everything one game's step reads from global memory: the record (one or two int4 per lane) and the action.
struct GameInput {
    int4 rq0, rq1, pos_raw;
    int a_raw;
    const int4 *src;      // WIDE boards (a record of more than two int4 per lane): staged straight from here in env_step
};
// (src: the game's record -- global memory, or the LDS image of sgx_step_states' fused kernel)
template <class G, class KP>
__device__ __forceinline__ GameInput load_game_from(const KP &P, const int4 *src, const int64_t env, const int lane) {
    constexpr int Q_BOARDS = G::ST_OFF / 16, Q_REC = Q_BOARDS + G::TAIL_BYTES / 16, NLOAD = (Q_REC + G::LPG - 1) / G::LPG;
    static_assert(G::TAIL_BYTES % 16 == 0 && (NLOAD <= 2 || G::WIDE || G::BIG), "record image must fit two int4 per lane");
    const int4 zero4 = make_int4(0, 0, 0, 0);
    GameInput in{zero4, zero4, zero4, 0, src};
    const int nq = G::BIG ? Q_REC : min(P.rec_bytes >> 4, Q_REC);
    if constexpr (NLOAD <= 2) {
        if (lane < nq) in.rq0 = src[lane];
        if constexpr (NLOAD > 1)
            if (lane + G::LPG < nq) in.rq1 = src[lane + G::LPG];
    }
    if (P.mode == 0) {
        if (P.io.flags & SGX_STEP_ACTIONS_POSITIONS) in.pos_raw = reinterpret_cast<const int4 *>(P.io.actions_dev)[env];
        else in.a_raw = P.io.actions_dev[env];
    }
    return in;
}
template <class G, bool MAPPED, class KP>
__device__ __forceinline__ GameInput load_game(const KP &P, const int64_t env, const int lane) {
    if (env >= P.n_envs) { const int4 zero4 = make_int4(0, 0, 0, 0); return GameInput{zero4, zero4, zero4, 0, nullptr}; }
    const int8_t *rec_src = P.boards + env * (int64_t)P.rec_bytes;
    if constexpr (MAPPED)        // sgx_expand: the game comes from another handle's records (functional-API instantiation only)
        if (P.src_boards) rec_src = P.src_boards + (int64_t)(P.src_index ? P.src_index[env] : env) * (int64_t)P.rec_bytes;
    return load_game_from<G>(P, reinterpret_cast<const int4 *>(rec_src), env, lane);
}

// The compact observation of one game (SGX_STEP_COMPACT_OBS): build the code buffer exactly as render() does, then store it as it is.
// One call site (env_step), outside render(): see there.
template <class G, class Spec, int NB, class KP>
__device__ __forceinline__ void compact_obs(const KP &P, float *obs_dev, Lds<G, NB> &L, const uint8_t *shared, const uint8_t *codetab, const float *glut, bool raw, int qi,
                                         int n_events, int rp0, int rp1, int64_t env, int lane) {
    const uint8_t *tmpl = shared;
    if constexpr (G::WIDE) tmpl = P.tab->tmpl[raw ? 2 : 0];
    const int n_unc = build_codes<G, Spec>(L, tmpl, codetab, glut, qi, n_events, rp0, rp1, lane, P.piece_counts, raw, P.multi_ev != 0);
    store_compact<G, Spec>(L, reinterpret_cast<uint8_t *>(obs_dev) + env * (int64_t)P.compact_stride, n_unc, lane);
}

// What the emission of the next mover's mask and observations needs from the step (SPLIT instantiation: single_kernel)
struct StepOut {
    int qi, n_events, rp0, rp1, n_unc;
};

// What a game carries from one step of a multi-step launch (steps_kernel) to the next while its boards stay in LDS: the record's scalars, the
// action the step drew, and whether the record in HBM is stale.
struct StepCarry {
    int turn, flags, max_turns, game_no, n_events, rp0, rp1, na;
    bool dirty;
    float *obs, *fobs;        // this step's output tensors (the output set of an sgx_step_ring / the slot of an sgx_step_traj): the caller sets them before every step
    uint8_t *mask;
    int slot;                 // ... and the slot number (per-slot results and the action log of sgx_step_traj: KParams::traj_*)
};

// One game's env.step() by one wave (called with the wave's private LDS region).
// `shared` = the workgroup's tables (shared_table_bytes): templates + code table, or LUTs + quad tables
// SPLIT: the next mover's mask and observations are NOT emitted here -- the caller does it with the whole workgroup from the LDS
// state this function leaves behind and the scalars in *so.
// PERSIST_ (steps_kernel): the call is one of several consecutive steps of the same game by the same wave.  Only the FIRST stages the record;
// later ones find the boards -- dense, never-moved bytes, recent-move codes, the event list -- where the step before left them in LDS,
// take the scalars and the action from *carry, and only the LAST writes the record back (if any step changed it).
template <int R_, int C_, int KIND, bool MAPPED, bool SPLIT = false, int VAR = 0, int PERSIST_ = 0, class KP = KParams>
__device__ __forceinline__ void env_step(const KP &P, Lds<Geo<R_, C_, VAR>, ObsKind<KIND>::NIB_CH> &L, const uint8_t *shared, const uint8_t *obst_s,
                                         const int64_t env, const int lane, const GameInput &in, int8_t *rec_out = nullptr, StepOut *so = nullptr,
                                         StepCarry *carry = nullptr, const bool last = true) {
    constexpr bool PERSIST = PERSIST_ != 0, first = PERSIST_ != 2;      // PERSIST_: 0 = a launch of its own, 1 = the first step of a multi-step launch, 2 = a later one
    using G = Geo<R_, C_, VAR>;
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool ORIG = ObsKind<KIND>::ORIG, FULL = ObsKind<KIND>::FULL, COMPACT = (KIND & 4) != 0, NOOBS = ObsKind<KIND>::NOOBS;
    const uint8_t *combat_s = obst_s + G::OBST_BYTES;           // the combat table follows the obstacle map in the shared LDS
    constexpr int R = G::R, C = G::C, RC = G::RC, S = G::S, K = G::K, NA = G::NA, MPA = G::MPA, AS = G::AS;
    STAMP(0);
    // the step's output tensors: the launch's, or -- one of several steps of a multi-step launch -- this step's output set
    float *const io_obs = PERSIST ? carry->obs : P.io.obs_dev, *const io_fobs = PERSIST ? carry->fobs : P.io.fobs_dev;
    uint8_t *const io_mask = PERSIST ? carry->mask : P.io.mask_dev;
    // where this step's results go: env, or -- a multi-step launch into a trajectory buffer that keeps every step's results -- the env's
    // place in this step's slot
    int64_t renv = env;
    if constexpr (PERSIST) renv = env + (int64_t)carry->slot * P.traj_res_envs;

    int8_t *rec_g = rec_out ? rec_out : P.boards + env * (int64_t)P.rec_bytes;   // (rec_out: the LDS record image of sgx_step_states)
    // ---- stage.  Every global read of the step has been issued up front (load_game) -- the whole record (a few 128-byte lines)
    //      as one or two int4 per lane, and the action -- so the wave pays ONE memory round trip, which overlaps the staging of
    //      the workgroup's shared tables.  The scalars, never-moved bitmaps and
    //      capture events are then read from the LDS image of the record (L.tail has the record's layout from ST_OFF on).
    constexpr int Q_BOARDS = G::ST_OFF / 16, Q_REC = Q_BOARDS + G::TAIL_BYTES / 16, NLOAD = (Q_REC + G::LPG - 1) / G::LPG;
    const int4 rq0 = in.rq0, rq1 = in.rq1;
    const int a_raw = (PERSIST && !first) ? carry->na : in.a_raw;
    const int4 pos_raw = in.pos_raw;
    if (!PERSIST || first)
    {   // (the loads may still be in flight) clear the recent-move boards and the zero board, copy the obstacle map (shared per
        // workgroup).  (The never-moved boards are written cell by cell below; captured counts are never dense.)
        int4 *dst = reinterpret_cast<int4 *>(&L.b[0][0]);
        for (int i = lane; i < S / 4; i += G::LPG) {
            reinterpret_cast<int *>(L.b[B_RECENT])[i] = 0;
            reinterpret_cast<int *>(L.b[B_RECENT + 1])[i] = 0;
            if constexpr (ORIG) reinterpret_cast<int *>(L.b[B_ZERO])[i] = 0;
            reinterpret_cast<int *>(L.b[B_OBST])[i] = reinterpret_cast<const int *>(obst_s)[i];
        }
        int4 *tl = reinterpret_cast<int4 *>(L.tail);
        if constexpr (NLOAD <= 2) {
            if (lane < Q_BOARDS) dst[lane] = rq0;
            else if (lane < Q_REC) tl[lane - Q_BOARDS] = rq0;
            if constexpr (NLOAD > 1) {
                if (lane + G::LPG < Q_BOARDS) dst[lane + G::LPG] = rq1;
                else if (lane + G::LPG < Q_REC) tl[lane + G::LPG - Q_BOARDS] = rq1;
            }
        } else {                                              // WIDE boards: the record is several KiB, staged in a loop
            const int nq = G::BIG ? Q_REC : min(P.rec_bytes >> 4, Q_REC);
            for (int i = lane; i < Q_REC; i += G::LPG) {
                const int4 v = i < nq ? in.src[i] : make_int4(0, 0, 0, 0);
                if (i < Q_BOARDS) dst[i] = v;
                else tl[i - Q_BOARDS] = v;
            }
        }
    }
    wave_sync<G>();
    int turn, flags, game_no, max_turns, n_events, rp0, rp1;   // (rp0 / rp1: recent-move pairs of player +1 / -1 -- two named scalars: a
                                                               //  runtime-indexed array would live in scratch memory)
    if (PERSIST && !first) {
        turn = carry->turn; flags = carry->flags; game_no = carry->game_no; max_turns = carry->max_turns;
        n_events = carry->n_events; rp0 = carry->rp0; rp1 = carry->rp1;
    } else {
        const int4 sc = reinterpret_cast<const int4 *>(L.tail + 2 * G::SB)[0], sc2 = reinterpret_cast<const int4 *>(L.tail + 2 * G::SB)[1];
        turn = uni<G>(sc.x); flags = uni<G>(sc.y); game_no = uni<G>(sc.w);
        max_turns = uni<G>(sc.z);
        n_events = min(uni<G>(sc2.x), (int)G::EVL_MAX);
        rp0 = uni<G>(sc2.y); rp1 = uni<G>(sc2.z);
    }
    if (!SGX_ABLATED(P.map_arg, 2) && (!PERSIST || first))
    {   // ---- rebuild the derived boards: never-moved bitmaps, recent-move pairs (capture events stay a list)
        const uint32_t *stb = reinterpret_cast<const uint32_t *>(L.tail);
#pragma unroll
        for (int cc = 0; cc < G::CPL; ++cc) {
            const int i = lane + G::LPG * cc;
            if (i < RC) {
                L.b[B_STILL][i] = (int8_t)((stb[i >> 5] >> (i & 31)) & 1u);
                L.b[B_STILL + 1][i] = (int8_t)((stb[G::SB / 4 + (i >> 5)] >> (i & 31)) & 1u);
            }
        }
        if constexpr (G::BIG) {                      // general-state image: the recent-move boards are dense
            for (int i = lane; i < S; i += G::LPG) {
                L.b[B_RECENT][i] = (int8_t)L.tail[G::RECB_OFF - G::ST_OFF + i];
                L.b[B_RECENT + 1][i] = (int8_t)L.tail[G::RECB_OFF - G::ST_OFF + G::S_PAD + i];
            }
        } else if (lane < 4) {
            const int pr = (((lane >> 1) ? rp1 : rp0) >> (16 * (lane & 1))) & 0xFFFF;
            if (pr >> G::CELL_BITS) L.b[B_RECENT + (lane >> 1)][G::pair_cell(pr)] = (int8_t)G::pair_code(pr);
        }
    }
    wave_sync<G>();
    STAMP(1);   // state staged

    int player = (flags & F_PLAYER_M1) ? -1 : 1;
    bool over = (flags & F_OVER) != 0;
    bool applied = false, invalid_action = false, noop_path = false;
    int mover = player;

    if (P.mode == 0) {
        // ------------------------------------------------------------------------------------------
        // decode (maenv:684-689): flat spatial index -> positions -> 1-D index -> absolute 1-D index
        // ------------------------------------------------------------------------------------------
        const int a = uni<G>(a_raw);
        int sr = 0, sc_ = 0, er = 0, ec = 0;
        bool valid = true;
        if (P.io.flags & SGX_STEP_ACTIONS_POSITIONS) {
            // is_move_valid_by_position (penv:87-92): actions_dev is int32 [N][4] = (start_r, start_c, end_r, end_c), absolute
            sr = uni<G>(pos_raw.x); sc_ = uni<G>(pos_raw.y); er = uni<G>(pos_raw.z); ec = uni<G>(pos_raw.w);
        } else if (P.io.flags & SGX_STEP_ACTIONS_1D) {
            // functional API (penv:148-155): the action already is an absolute-coordinate 1-D index (impl:262-277)
            if (a == AS - 1) {
                noop_path = true;
            } else {                                                                         // impl:369-383
                const int q = fdiv_(a, MPA), off = fmod_(a, MPA);
                sr = fdiv_(q, C); sc_ = fmod_(q, C);
                if (off >= R) { ec = off - R; er = sr; } else { er = off; ec = sc_; }
            }
        } else if (a < 0 || a >= NA) {
            valid = false;  // np.unravel_index raises
        } else {
            const int cell = a / K, ch = a - cell * K;
            sr = cell / C; sc_ = cell - sr * C;
            if (ch < R - 1) { er = sr + ch + 1; ec = sc_; }                                   // impl:322-324
            else if (ch < 2 * (R - 1)) { er = sr - (ch - (R - 1) + 1); ec = sc_; }
            else if (ch < 2 * (R - 1) + (C - 1)) { er = sr; ec = sc_ + (ch - 2 * (R - 1) + 1); }
            else { er = sr; ec = sc_ - (ch - (2 * (R - 1) + (C - 1)) + 1); }                   // also the no-op channel
            if (ch < K - 1 && er >= 0 && er < R && ec >= 0 && ec < C) {
                // A straight move that stays on the board: its 1-D index (impl:268-277) decodes back to the same four
                // coordinates (impl:369-383) and the perspective flip through the 1-D index (impl:698-720) is the flip of the
                // coordinates -- the reference's chain below, which exists for the garbage cases, is the identity here.
                if (player == -1) { sr = R - 1 - sr; sc_ = C - 1 - sc_; er = R - 1 - er; ec = C - 1 - ec; }
            } else {
                int idx = (sr * C + sc_) * MPA + ((er != sr) ? er : R + ec);                 // impl:268-277
                if (player == -1 && idx != AS - 1) {                                         // impl:698-720
                    const int q = fdiv_(idx, MPA), off = fmod_(idx, MPA);
                    int r0 = fdiv_(q, C), c0 = fmod_(q, C), r1, c1;
                    if (off >= R) { c1 = off - R; r1 = r0; } else { r1 = off; c1 = c0; }
                    r0 = R - 1 - r0; r1 = R - 1 - r1; c0 = C - 1 - c0; c1 = C - 1 - c1;
                    idx = (r0 * C + c0) * MPA + ((r1 != r0) ? r1 : R + c1);
                }
                if (idx == AS - 1) {
                    noop_path = true;                                                        // impl:809-814
                } else {                                                                     // impl:369-383
                    const int q = fdiv_(idx, MPA), off = fmod_(idx, MPA);
                    sr = fdiv_(q, C); sc_ = fmod_(q, C);
                    if (off >= R) { ec = off - R; er = sr; } else { er = off; ec = sc_; }
                }
            }
        }
        const int pi = player == 1 ? 0 : 1;
        int8_t *own = L.b[B_PIECES + pi], *enemy = L.b[B_PIECES + 1 - pi];
        int8_t *own_po = L.b[B_PO + pi], *enemy_po = L.b[B_PO + 1 - pi];
        int8_t *own_still = L.b[B_STILL + pi], *enemy_still = L.b[B_STILL + 1 - pi];
        int8_t *recent = L.b[B_RECENT + pi];
        const int8_t *obst = L.b[B_OBST];

        if (valid && noop_path) {
            // no-op is legal only if the mover has no move (or the game is over); finished games stay unchanged
            if (!over) {
                const int nmoves = gen_any(L, pi, false, lane);          // (only whether there is one)
                if (nmoves != 0) valid = false;
                else { turn += 1; over = true; flags |= F_OVER | (player == 1 ? F_WIN_M1 : F_WIN_P1); }  // impl:916-920
            }
        } else if (valid) {
            // ---- _is_move_valid_by_position (impl:723-798).  All board bytes the checks (and the move) need are read
            //      up front from clamped cell indices, so the wave pays one LDS round trip instead of ten dependent ones.
            const bool s_in = !(sc_ < 0 || sc_ >= C || sr < 0 || sr >= R), e_in = !(ec < 0 || ec >= C || er < 0 || er >= R);
            const int s = s_in ? sr * C + sc_ : 0, e = e_in ? er * C + ec : 0;
            const int v_obst_s = obst[s], v_obst_e = obst[e], v_own_s = own[s], v_own_e = own[e], v_en_e = enemy[e];
            const int v_rec_s = recent[s], v_rec_e = recent[e], v_po_s = own_po[s];
            const int obst_s = uni<G>(v_obst_s), obst_e = uni<G>(v_obst_e), t = uni<G>(v_own_s), own_e = uni<G>(v_own_e);
            const int dest = uni<G>(v_en_e), old_start = uni<G>(v_rec_s), old_end = uni<G>(v_rec_e), moved_po = uni<G>(v_po_s);
            if (over) valid = false;
            if (!s_in || obst_s != 0) valid = false;
            if (!e_in || obst_e != 0) valid = false;
            if (t == 0 || t == SP_FLAG || t == SP_BOMB) valid = false;
            if (own_e != 0) valid = false;
            if (er != sr && ec != sc_) valid = false;
            if (old_start == -3 && old_end == 1 && dest == 0 && !(P.io.flags & SGX_STEP_ALLOW_OSCILLATION)) valid = false;   // impl:771-777
            if (valid) {
                const int dist = (er != sr) ? abs(er - sr) : abs(ec - sc_);
                if (t == SP_SCOUT) {
                    const int stepc = (er != sr) ? ((er > sr) ? C : -C) : ((ec > sc_) ? 1 : -1);
                    bool blk = false;        // lanes check the intermediate cells (a board side can exceed LPG + 1 cells: loop)
                    for (int k = lane + 1; k < dist; k += G::LPG) { const int m = s + k * stepc; blk = blk || own[m] != 0 || enemy[m] != 0 || obst[m] != 0; }
                    if (gballot<G>(blk) != 0ull) valid = false;
                } else if (dist > 1) valid = false;
            }
            if (valid) {
                // ---- _get_next_state (impl:905-1028)
                const int moved = t;
                turn += 1;
                bool wins = false, tied = false;
                if (dest != 0) {
                    // branch-free combat: one read of the 16 x 16 outcome table (the first-match chain of impl:968-982, tabulated
                    // on the host: miner > bomb, spy > marshal when attacking, anything > flag, bomb > the rest, else by rank)
                    const int outcome = uni<G>((int)combat_s[16 * moved + dest]);
                    wins = outcome >= COMBAT_WIN;
                    tied = outcome == COMBAT_TIE;
                    if (outcome == COMBAT_WIN_FLAG) { over = true; flags |= F_OVER | (player == 1 ? F_WIN_P1 : F_WIN_M1); }
                }
                wave_sync<G>();
                // clear the mover's recent-moves board (np.zeros_like, impl:1014)
                for (int i = lane; i < S / 4; i += G::LPG) reinterpret_cast<int *>(recent)[i] = 0;
                wave_sync<G>();
                if (lane == 0) {
                    own_still[s] = 0; own_still[e] = 0; enemy_still[e] = 0;  // impl:939-941
                    own[s] = 0; own_po[s] = 0;                               // impl:950-951
                    if (dest == 0) {
                        own[e] = (int8_t)moved;
                        const int far = (abs(er - sr) > 1 || abs(ec - sc_) > 1);
                        own_po[e] = (int8_t)(far ? SP_SCOUT : moved_po);      // impl:960-964
                        recent[s] = 1;                                       // impl:1019-1026
                        recent[e] = (int8_t)(old_end == 1 ? (old_start == -2 ? -3 : -2) : -1);
                    } else {
                        if (tied || wins) { enemy[e] = 0; enemy_po[e] = 0; }
                        if (wins) { own[e] = (int8_t)moved; own_po[e] = (int8_t)moved; }
                        if (!wins && !tied) enemy_po[e] = (int8_t)dest;
                    }
                }
                if (dest != 0) {
                    // captured counts (impl:999-1009): the attacker's own layer unless it won, the defender's if it lost or tied
                    if (!wins) n_events = add_capture(L, n_events, max_events_of<G>(P), ((12 * pi + moved - 1) << G::CELL_BITS) | e, lane, P.multi_ev != 0);
                    if (wins || tied) n_events = add_capture(L, n_events, max_events_of<G>(P), ((12 * (1 - pi) + dest - 1) << G::CELL_BITS) | e, lane, P.multi_ev != 0);
                    if (pi) rp1 = 0; else rp0 = 0;                                       // an attack wipes the mover's layer
                } else {
                    const int code = old_end == 1 ? (old_start == -2 ? -3 : -2) : -1;
                    const int pr = G::make_pair(s, 1) | (G::make_pair(e, code) << 16);
                    if (pi) rp1 = pr; else rp0 = pr;
                }
                wave_sync<G>();
            }
        }
        if (valid) { applied = true; player = -player; } else invalid_action = true;
    }

    STAMP(2);   // move applied
    // ---- next mover's mask; opponent-stuck and max-turn endings (impl:1031-1043)
    int qi = player == 1 ? 0 : 1;
    // Launches of the no-observation kind that want neither the mask nor a next action (search expansion, sgx_expand; logic-only steps)
    // only need to know WHETHER the next mover has a move (the opponent-stuck ending): no mask bits, no counts, one cell per ray.
    const bool want_bits = !NOOBS || SPLIT || io_mask != nullptr || (P.mode == 0 && P.io.next_actions_dev != nullptr);
#ifdef SGX_ABLATE
    if (SGX_ABLATED(P.map_arg, 3)) return;                              // staging only
    int nvalid = gen_mask(L, qi, over, lane, P.map_arg);
#else
    int nvalid;
    if constexpr (NOOBS) nvalid = want_bits ? gen_mask(L, qi, over, lane) : gen_any(L, qi, over, lane);
    else nvalid = gen_mask(L, qi, over, lane);
#endif
    bool ended_now = false;
    if (applied && !noop_path) {
        const bool was_over = over;
        if (nvalid == 0) { over = true; flags = (flags & ~(F_WIN_P1 | F_WIN_M1)) | F_OVER | (mover == 1 ? F_WIN_P1 : F_WIN_M1); }
        if (turn >= max_turns && !over) { over = true; flags |= F_OVER | F_END_INVALID; }
        if (over && !was_over && nvalid != 0) { if (want_bits) mask_noop_only(L, lane); nvalid = 0; }  // finished: the no-op only
        ended_now = over;
    } else if (applied && noop_path) {
        ended_now = over;
        if (nvalid != 0) { if (want_bits) mask_noop_only(L, lane); nvalid = 0; }
    }
    flags = (flags & ~F_PLAYER_M1) | (player == -1 ? F_PLAYER_M1 : 0);
    STAMP(3);   // mask generated

    // ---- rewards / dones (maenv:699-805)
    const bool done = over;
    const bool end_invalid = over && (flags & F_END_INVALID);
    float rew_p1 = 0.f, rew_m1 = 0.f;
    if (over && !end_invalid) {
        const int w = (flags & F_WIN_P1) ? 1 : (flags & F_WIN_M1) ? -1 : 0;
        rew_p1 = w == 0 ? 1e-4f : (float)w;     // impl:838-840
        rew_m1 = w == 0 ? 1e-4f : (float)-w;
    }
    if (P.mode == 0) {
        // one store instruction for the rewards (lanes 0/1) and one for the three byte flags (lanes 0..2)
        if (lane < 2 && P.io.reward_dev) P.io.reward_dev[2 * renv + lane] = lane ? rew_m1 : rew_p1;
        uint8_t *fp = lane == 0 ? P.io.done_dev : lane == 1 ? P.io.invalid_action_dev : lane == 2 ? P.io.ending_invalid_dev : nullptr;
        const uint8_t fv = lane == 0 ? (done ? 1 : 0) : lane == 1 ? (invalid_action ? 1 : 0) : (end_invalid ? 1 : 0);
        if (fp) fp[renv] = fv;
    }

    // ---- observation rendering (sgx_obs.h).  render() writes one observation.  'extended' kinds on 4-aligned boards need no
    //      second pass over memory: quads with an uncoded entry are left out of the bulk stores and written whole right after.
    //      Odd boards and 'original' kinds write single floats over lines other lanes have just stored, so they wait for the
    //      bulk stores first; render() then returns what is still to do and finish() does it.
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    const float *glut_p = P.tab->lut[(ORIG ? 4 : 0) + (raw ? 2 : 0)], *glut_f = P.tab->lut[(ORIG ? 4 : 0) + (raw ? 2 : 0) + 1];
    const uint8_t *codetab = ORIG ? nullptr : shared + tmpl_lds_bytes<G, KIND>(false) + (FULL ? tmpl_lds_bytes<G, KIND>(true) : 0);
    auto render = [&](auto spec, bool full, int q, float *dst) -> int {
        using Spec = decltype(spec);
        if constexpr (Spec::CODES) {
            const uint8_t *tmpl = full ? shared + tmpl_lds_bytes<G, KIND>(false) : shared;
            if constexpr (G::WIDE) tmpl = P.tab->tmpl[(raw ? 2 : 0) + (full ? 1 : 0)];           // (global memory, L2-resident)
            const int n_unc = build_codes<G, Spec>(L, tmpl, codetab, full ? glut_f : glut_p, q,
                                                   n_events, rp0, rp1, lane, P.piece_counts, raw, P.multi_ev != 0);
            if constexpr (RC % 4 == 0) {
                if (n_unc == 0) {
                    if (P.nt_stores) emit_codes<G, Spec, false, true>(L, dst, lane);
                    else emit_codes<G, Spec, false, false>(L, dst, lane);
                } else {
                    if (P.nt_stores) emit_codes<G, Spec, true, true>(L, dst, lane);
                    else emit_codes<G, Spec, true, false>(L, dst, lane);
                    patch_uncoded<G, Spec>(L, dst, n_unc, lane);
                }
                return 0;
            } else {
                if (P.nt_stores) emit_codes<G, Spec, false, true>(L, dst, lane);
                else emit_codes<G, Spec, false, false>(L, dst, lane);
                if (n_unc) {                                     // (rare board sizes: settle it here, L.unc_* is per rendering)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    patch_uncoded_floats(L, dst, n_unc, lane);
                }
                return 0;
            }
        } else {
            emit_obs_lut<G, Spec>(L, reinterpret_cast<const float *>(shared) + (full ? OBS_TAB_DWORDS : 0), q, dst, lane);
            if (n_events > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                emit_obs_patches<G, Spec>(L, full ? glut_f : glut_p, q, dst, n_events, lane, raw, P.multi_ev != 0, P.piece_counts);
            }
            return 0;
        }
    };

    // ---- terminal observations of both players (maenv:772-773)
    if constexpr (!NOOBS)
    if (P.mode == 0 && ended_now && P.io.final_obs_dev) {
        float *fo = P.io.final_obs_dev + env * (int64_t)(2 * RC * PS::NCH);
        render(PS{}, false, 0, fo);
        render(PS{}, false, 1, fo + RC * PS::NCH);
    }
    if constexpr (FULL)
        if (P.mode == 0 && ended_now && P.io.final_fobs_dev) {
            float *fo = P.io.final_fobs_dev + env * (int64_t)(2 * RC * FS::NCH);
            render(FS{}, true, 0, fo);
            render(FS{}, true, 1, fo + RC * FS::NCH);
        }

    // ---- auto-reset: the finished env starts its next game now
    bool wrote_reset = false;
    if (P.mode == 0 && ended_now && P.io.auto_reset) {
        game_no += 1;
        sample_boards(L, P, (uint64_t)(P.env_id_offset + env), (uint64_t)game_no, lane);
        turn = 0; flags = 0; player = 1; qi = 0; over = false;
        n_events = 0; rp0 = rp1 = 0;
        nvalid = want_bits ? gen_mask(L, 0, false, lane) : 1;          // (without mask and sampler nobody looks at the count of a fresh game)
        wrote_reset = true;
    }

    STAMP(4);   // results / terminal handling done
    // ---- outputs for the next mover
    if (lane == 0 && P.io.player_dev) P.io.player_dev[renv] = (int8_t)player;
    if constexpr (SPLIT) {
        if (lane == 0) { so->qi = qi; so->n_events = n_events; so->rp0 = rp0; so->rp1 = rp1; }
    } else {
    if (io_mask) {
        // MAPPED: the separate instantiation behind SGX_STEP_MASK_1D / SGX_STEP_MASK_STATE_COORDS (kept out of the hot kernel: its
        // 16 index computations per lane cost 30 VGPRs)
        if (MAPPED && NOOBS && (P.io.flags & SGX_STEP_MASK_1D)) emit_mask_mapped_inl(L, io_mask + env * (int64_t)AS, AS, Src1D<G>{qi}, lane);
        else if (MAPPED && NOOBS && qi) emit_mask_mapped_inl(L, io_mask + env * (int64_t)NA, NA, SrcSpatialFlipped<G>{}, lane);
        else if (MAPPED && (P.io.flags & SGX_STEP_MASK_1D)) emit_mask_mapped(L, io_mask + env * (int64_t)AS, AS, Src1D<G>{qi}, lane);
        else if (MAPPED && qi) emit_mask_mapped(L, io_mask + env * (int64_t)NA, NA, SrcSpatialFlipped<G>{}, lane);
        else if (COMPACT && (P.io.flags & SGX_STEP_COMPACT_MASK)) {   // the mask as bits: uint32 [MB_WORDS] per game, bit a = action a
            int4 *mdst = reinterpret_cast<int4 *>(reinterpret_cast<uint32_t *>(io_mask) + env * (int64_t)G::MB_WORDS);
            for (int i = lane; i < G::MB_WORDS / 4; i += G::LPG) mdst[i] = reinterpret_cast<const int4 *>(L.mbits)[i];
        }
        else emit_mask(L, io_mask + env * (int64_t)NA, lane);
    }
    STAMP(5);   // mask stores issued
    // (rendering the observation before the mask, so that its stores drain during mask generation, measured 6 % slower)
    if constexpr (!NOOBS)
    if (io_obs) {
        // compact output (opt-in, KIND bit 2): the codes themselves, 1/8 of the bytes; sgx_decode_obs expands them.  An instantiation of
        // its own: as a run-time branch it took the 8x8 hot kernel from 35 to 64 VGPRs + scratch
        bool done_compact = false;
        if constexpr (COMPACT) {
            if (P.io.flags & SGX_STEP_COMPACT_OBS) {
                compact_obs<G, PS>(P, io_obs, L, shared, codetab, glut_p, raw, qi, n_events, rp0, rp1, env, lane);
                done_compact = true;
            }
        }
        if (!done_compact) render(PS{}, false, qi, io_obs + env * (int64_t)(RC * PS::NCH));
    }
    if constexpr (FULL)
        if (io_fobs) render(FS{}, true, qi, io_fobs + env * (int64_t)(RC * FS::NCH));
    }
    STAMP(6);   // obs stores issued
    if (P.mode == 0 && P.io.next_actions_dev) {
        const int total = nvalid == 0 ? 1 : nvalid;
        const uint32_t k = rng_below(sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)game_no, STREAM_ACTION, (uint32_t)turn), (uint32_t)total);
        const int na = kth_valid(L, (int)k, lane);
        if constexpr (PERSIST) {
            carry->na = na;
            if (lane == 0 && P.traj_act_log) P.traj_act_log[env + (int64_t)carry->slot * P.traj_out_envs] = na;
        }
        if (lane == 0 && (!PERSIST || last)) P.io.next_actions_dev[env] = na;
    }

    STAMP(7);   // next action sampled
    // ---- write the record back as whole 128-byte lines: dense boards, scalars, capture events.  (Scattered stores
    //      of only the <= 9 touched bytes + 32 B of scalars are partial-line writes: measured 7 % slower.)
    bool dirty = applied || wrote_reset || (MAPPED && P.src_boards);
    if constexpr (PERSIST) {
        dirty = dirty || (!first && carry->dirty);
        carry->turn = turn; carry->flags = flags; carry->game_no = game_no; carry->max_turns = max_turns;
        carry->n_events = n_events; carry->rp0 = rp0; carry->rp1 = rp1; carry->dirty = dirty;
        if (!last) dirty = false;                  // (the boards stay in LDS; the last step of the launch writes the record)
    }
    if (dirty)
        write_record(L, rec_g, G::BIG ? (int)G::IMG_BYTES : P.rec_bytes, make_int4(turn, flags, max_turns, game_no), make_int4(n_events, rp0, rp1, 0), n_events, lane);
    STAMP(8);   // write-back issued
#ifdef SGX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(9);   // all stores acknowledged
#endif
}

// KIND bit 0: also renders the fully-observable observation (BOTH_OBSERVATIONS / FULLY_OBSERVABLE modes, maenv:477-492);
// KIND bit 1: obs_channel_mode 'original' (32/33 value channels) instead of 'extended' (67/79 one-hot channels);
// KIND bit 2 (alone: KIND 4): compact outputs -- the partial 'extended' kind whose observation / mask may leave as codes / bits
template <int R_, int C_, int KIND, bool MAPPED, int VAR = 0>
__device__ __forceinline__ void game_kernel_body(const KParams &P) {
    using G = Geo<R_, C_, VAR>;
    static_assert(VAR == 0 || (VAR == 2 && ObsKind<KIND>::NOOBS), "two games per wave: the no-observation kinds only");
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool FULL = ObsKind<KIND>::FULL, ORIG = ObsKind<KIND>::ORIG;
    __shared__ Lds<G, ObsKind<KIND>::NIB_CH> LW[G::WPB * G::GPW];
    __shared__ alignas(16) uint8_t shared[shared_table_bytes<G, KIND>()];
    __shared__ alignas(16) uint8_t obst_s[G::OBST_BYTES + COMBAT_BYTES];    // obstacle map, then the combat outcome table
    const int lane = threadIdx.x & (G::LPG - 1), slot = threadIdx.x / G::LPG;     // lane inside the game, game inside the workgroup
    const int64_t env = P.env_first + group_of_block(P) * (G::WPB * G::GPW) + slot;
    // The game's record and action are requested FIRST: the reads fly while the workgroup stages its shared tables (another
    // global round trip) and waits at the barrier -- the two round trips used to follow each other.
    const GameInput in = load_game<G, MAPPED>(P, env, lane);
    // ---- the workgroup's shared tables (L2-resident sources)
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    if constexpr (ObsKind<KIND>::NOOBS) {
        // nothing to render: only the obstacle map and the combat table below
    } else if constexpr (ORIG) {
        float *lut_s = reinterpret_cast<float *>(shared);
        const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0)]);
        for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * G::WPB) reinterpret_cast<f32x4 *>(lut_s)[i] = lsrc[i];
        build_quad_table<G, PS>(reinterpret_cast<uint32_t *>(lut_s + LUT_DWORDS), threadIdx.x, 64 * G::WPB);
        if constexpr (FULL) {
            const f32x4 *fsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0) + 1]);
            for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * G::WPB) reinterpret_cast<f32x4 *>(lut_s + OBS_TAB_DWORDS)[i] = fsrc[i];
            build_quad_table<G, FS>(reinterpret_cast<uint32_t *>(lut_s + OBS_TAB_DWORDS + LUT_DWORDS), threadIdx.x, 64 * G::WPB);
        }
    } else {
        constexpr int NP = tmpl_lds_bytes<G, KIND>(false), NF = FULL ? tmpl_lds_bytes<G, KIND>(true) : 0;
        const int4 *tp = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0]);
        for (int i = threadIdx.x; i < NP / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared)[i] = tp[i];
        if constexpr (FULL) {
            const int4 *tf = reinterpret_cast<const int4 *>(P.tab->tmpl[(raw ? 2 : 0) + 1]);
            for (int i = threadIdx.x; i < NF / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared + NP)[i] = tf[i];
        }
        const int4 *ct = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0]);
        for (int i = threadIdx.x; i < CODETAB_BYTES / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared + NP + NF)[i] = ct[i];
    }
    for (int i = threadIdx.x; i < G::S / 4; i += 64 * G::WPB) reinterpret_cast<int *>(obst_s)[i] = reinterpret_cast<const int *>(P.tab->obstacles)[i];
    for (int i = threadIdx.x; i < COMBAT_BYTES / 4; i += 64 * G::WPB)
        reinterpret_cast<int *>(obst_s + G::OBST_BYTES)[i] = reinterpret_cast<const int *>(P.tab->combat)[i];
    __syncthreads();   // from here on every wave works on its own game
    if (env < P.n_envs) env_step<R_, C_, KIND, MAPPED, false, VAR>(P, LW[slot], shared, obst_s, env, lane, in);
}

// sgx_step and sgx_observe run the same body (P.mode tells them apart at run time: specialising the body on the mode changed the
// step kernel's schedule and cost 3.7 % on Barrage); two kernel symbols, so that a kernel trace keeps the env.step() launches
// apart from the state-preserving observe launches (placement trials, reset())
template <int R_, int C_, int KIND, bool MAPPED = false, int VAR = 0>
__global__ __launch_bounds__((64 * Geo<R_, C_, VAR>::WPB), (waves_per_simd<Geo<R_, C_, VAR>, KIND, MAPPED>())) void step_kernel(const KParams P) {
    game_kernel_body<R_, C_, KIND, MAPPED, VAR>(P);
}
template <int R_, int C_, int KIND, bool MAPPED = false, int VAR = 0>
__global__ __launch_bounds__((64 * Geo<R_, C_, VAR>::WPB), (waves_per_simd<Geo<R_, C_, VAR>, KIND, MAPPED>())) void observe_kernel(const KParams P) {
    game_kernel_body<R_, C_, KIND, MAPPED, VAR>(P);
}

// ---------------------------------------------------------------------------------------------
// steps_kernel: ALL steps of an sgx_step_n / sgx_step_ring call in one launch, wave-per-game boards.  A workgroup stages its games once and
// every wave then plays its game step after step (env_step<PERSIST>): the boards stay in LDS, the scalars and the drawn action in
// registers; the record is read once and written once per launch.  No barrier after the table staging: the waves of a workgroup -- and
// of the chip -- drift out of phase within a few steps, so one wave's stores run under another's game logic (with one launch per step
// every residency round starts, plays and stores together: docs/DESIGN_rounds_4-5.md section 7).  Same results as n_steps launches of step_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int WSTEPS_MAX_SETS = 8;
struct WaveStepsParams {
    KParams k;
    int32_t n_steps, n_sets, first_set;
    // strided 1 (sgx_step_traj): the n_sets "sets" are the slots of a trajectory buffer -- slot s = the tensors of slot 0 (obs[0] / fobs[0] /
    // mask[0]) + s x these byte strides (0 for a tensor that is NULL); any number of slots.  0: the tensors of set s are obs[s] / fobs[s] /
    // mask[s] (at most WSTEPS_MAX_SETS).  2 (sgx_step_ring with MORE separate sets than fit the kernel arguments): the pointers of set s are
    // obs_tab[s] / fobs_tab[s] / mask_tab[s], tables in device memory the host filled before the launch (read with scalar loads).
    int32_t strided;
    // != 0: the waves of a workgroup meet at a barrier before every step (sgx_set_steps_barrier: rings of more than 8 sets -- the workgroup's
    // games then write ONE set at a time, 8 adjacent regions of it, instead of up to 8 sets; the host never sets it on a launch whose last
    // workgroup is partly empty)
    int32_t barrier;
    int64_t obs_slot_bytes, fobs_slot_bytes, mask_slot_bytes;
    float *const *obs_tab, *const *fobs_tab;
    uint8_t *const *mask_tab;
    float *obs[WSTEPS_MAX_SETS], *fobs[WSTEPS_MAX_SETS];      // the output tensors of set s (sgx_step_ring); one set: in place
    uint8_t *mask[WSTEPS_MAX_SETS];
};
// the output tensors of set / slot `set` (scalar arithmetic on kernel arguments)
template <class SPP>
__device__ __forceinline__ void steps_outputs_of(const SPP sp, const int set, StepCarry &carry) {
    if (sp->strided == 2) {
        carry.obs = table_entry(sp->obs_tab, set);
        carry.fobs = table_entry(sp->fobs_tab, set);
        carry.mask = table_entry(sp->mask_tab, set);
    } else if (sp->strided) {
        carry.obs = reinterpret_cast<float *>(reinterpret_cast<char *>(sp->obs[0]) + (int64_t)set * sp->obs_slot_bytes);
        carry.fobs = reinterpret_cast<float *>(reinterpret_cast<char *>(sp->fobs[0]) + (int64_t)set * sp->fobs_slot_bytes);
        carry.mask = sp->mask[0] + (int64_t)set * sp->mask_slot_bytes;
    } else {
        carry.obs = sp->obs[set];
        carry.fobs = sp->fobs[set];
        carry.mask = sp->mask[set];
    }
    carry.slot = set;
}
// occupancy promise of steps_kernel: the per-step kernel's, except on small boards of an odd cell count (5x5: two games per wave, the
// unaligned observation sweep) with an observation to render -- the 64 registers of the 8-wave promise spill inside the steps loop there,
// and a scratch access retires in order with the wave's outstanding observation stores (in-process A/B on one set of buffers, 65,536
// games: ring of three 99.8 -> 92.7 us, in place 93.5 -> 91.3, compact 42.8 -> 40.1; without an observation 8 waves stay faster)
template <class G, int KIND>
constexpr int steps_waves_per_simd() {
    constexpr int w = waves_per_simd<G, KIND, false>();
    return (G::RC <= 64 && G::RC % 4 != 0 && !ObsKind<KIND>::NOOBS && w > 6) ? 6 : w;
}
template <int R_, int C_, int KIND, int VAR = 0>
__global__ __launch_bounds__((64 * Geo<R_, C_, VAR>::WPB), (steps_waves_per_simd<Geo<R_, C_, VAR>, KIND>())) void steps_kernel(const WaveStepsParams SP) {
    using G = Geo<R_, C_, VAR>;
    static_assert(VAR == 0 || (VAR == 2 && ObsKind<KIND>::NOOBS), "two games per wave: the no-observation kinds only");
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool FULL = ObsKind<KIND>::FULL, ORIG = ObsKind<KIND>::ORIG;
    // The parameters are read through the kernel-argument segment's address IN ITS OWN ADDRESS SPACE (SGX_KERNARG, sgx_layout.h), and the loop
    // below hides that address from the optimiser once per step: otherwise every field env_step looks at is hoisted out of the loop and held
    // in scalar registers for its whole length (106 of 106 SGPRs, 324 bytes of scratch); re-read per step they cost a few scalar loads from
    // the constant cache.  (Through a generic pointer -- the first version of this kernel -- the re-reads were 17 VECTOR loads per game and
    // step, each followed by a wait for everything the wave had in flight, its observation stores included: 5x5 95 -> 76 us per step, 15x15
    // 369 -> 311, 6x6 111 -> 99, 10x10 246 -> 241 on one set of buffers, tools/lib_ab.py.)
#if defined(__HIP_DEVICE_COMPILE__)
    const SGX_KERNARG WaveStepsParams *spp = (const SGX_KERNARG WaveStepsParams *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const WaveStepsParams *spp = &SP;             // (host pass of the single-source compile: never executed)
#endif
    const SGX_KERNARG KParams &P = spp->k;
    __shared__ Lds<G, ObsKind<KIND>::NIB_CH> LW[G::WPB * G::GPW];
    __shared__ alignas(16) uint8_t shared[shared_table_bytes<G, KIND>()];
    __shared__ alignas(16) uint8_t obst_s[G::OBST_BYTES + COMBAT_BYTES];
    const int lane = threadIdx.x & (G::LPG - 1), slot = threadIdx.x / G::LPG;
    const int64_t env = P.env_first + group_of_block(P) * (G::WPB * G::GPW) + slot;
    const GameInput in = load_game<G, false>(P, env, lane);
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    if constexpr (ObsKind<KIND>::NOOBS) {
    } else if constexpr (ORIG) {
        float *lut_s = reinterpret_cast<float *>(shared);
        const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0)]);
        for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * G::WPB) reinterpret_cast<f32x4 *>(lut_s)[i] = lsrc[i];
        build_quad_table<G, PS>(reinterpret_cast<uint32_t *>(lut_s + LUT_DWORDS), threadIdx.x, 64 * G::WPB);
        if constexpr (FULL) {
            const f32x4 *fsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[4 + (raw ? 2 : 0) + 1]);
            for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * G::WPB) reinterpret_cast<f32x4 *>(lut_s + OBS_TAB_DWORDS)[i] = fsrc[i];
            build_quad_table<G, FS>(reinterpret_cast<uint32_t *>(lut_s + OBS_TAB_DWORDS + LUT_DWORDS), threadIdx.x, 64 * G::WPB);
        }
    } else {
        constexpr int NP = tmpl_lds_bytes<G, KIND>(false), NF = FULL ? tmpl_lds_bytes<G, KIND>(true) : 0;
        const int4 *tp = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0]);
        for (int i = threadIdx.x; i < NP / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared)[i] = tp[i];
        if constexpr (FULL) {
            const int4 *tf = reinterpret_cast<const int4 *>(P.tab->tmpl[(raw ? 2 : 0) + 1]);
            for (int i = threadIdx.x; i < NF / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared + NP)[i] = tf[i];
        }
        const int4 *ct = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0]);
        for (int i = threadIdx.x; i < CODETAB_BYTES / 16; i += 64 * G::WPB) reinterpret_cast<int4 *>(shared + NP + NF)[i] = ct[i];
    }
    for (int i = threadIdx.x; i < G::S / 4; i += 64 * G::WPB) reinterpret_cast<int *>(obst_s)[i] = reinterpret_cast<const int *>(P.tab->obstacles)[i];
    for (int i = threadIdx.x; i < COMBAT_BYTES / 4; i += 64 * G::WPB)
        reinterpret_cast<int *>(obst_s + G::OBST_BYTES)[i] = reinterpret_cast<const int *>(P.tab->combat)[i];
    __syncthreads();   // from here on every wave works on its own game, for all n_steps
    if (env >= P.n_envs) return;
    StepCarry carry{0, 0, 0, 0, 0, 0, 0, 0, false, nullptr, nullptr, nullptr, 0};
    int set = spp->first_set;
    const int n_steps = spp->n_steps;
    {   // the first step stages the record (its loads were issued before the table staging); its code is a copy of its own, so that the record's
        // registers are dead in the loop below
        steps_outputs_of(spp, set, carry);
        env_step<R_, C_, KIND, false, false, VAR, 1>(P, LW[slot], shared, obst_s, env, lane, in, nullptr, nullptr, &carry, n_steps == 1);
        set = set + 1 == spp->n_sets ? 0 : set + 1;
    }
    for (int t = 1; t < n_steps; ++t) {
        const SGX_KERNARG WaveStepsParams *sp = spp;
        int lane_t = lane, slot_t = slot;
        // (the step's reads of the parameters start here; and what a step derives from the lane -- dozens of cell / entry offsets -- is
        //  recomputed in every step like in a launch of its own, not hoisted out of the loop and spilled: 244 bytes of scratch otherwise)
        asm volatile("" : "+s"(sp), "+v"(lane_t), "+v"(slot_t));
        if (sp->barrier) __builtin_amdgcn_s_barrier();
        steps_outputs_of(sp, set, carry);
#ifdef SGX_MUTANT_SKIP_STORE     // test-the-tests build only (tools/mutant_check.sh): the fourth step of every launch loses its observation store
        if (t == 3) carry.obs = nullptr;
#endif
        env_step<R_, C_, KIND, false, false, VAR, 2>(sp->k, LW[slot_t], shared, obst_s, env, lane_t, in, nullptr, nullptr, &carry, t == n_steps - 1);
        set = set + 1 == sp->n_sets ? 0 : set + 1;
    }
}

// sgx_step_sync on a handful of games (the N = 1 facade, config 1): latency, not throughput.  ONE game per 512-thread workgroup: wave 0
// plays the step (env_step<SPLIT>), then all eight waves emit the mask and the observations -- 30 KiB by one wave is 4.3 of the step's
// 9.2 us (tools/phase_stamps.py barrage 1) -- and the last workgroup to finish publishes `seq` in a host-mapped word the host polls:
// the caller does not wait for the end-of-kernel processing of the queue (signal, cache write-back, wake-up: ~5 us).
// 'extended' kinds on 4-aligned one-game-per-wave boards.
constexpr int SINGLE_WAVES = 8;
template <int R_, int C_, int KIND>
__global__ __launch_bounds__(64 * SINGLE_WAVES) void single_kernel(const KParams P, uint32_t *__restrict__ done_count, uint32_t *__restrict__ flag_host,
                                                                 const uint32_t seq) {
    using G = Geo<R_, C_>;
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool FULL = ObsKind<KIND>::FULL;
    static_assert(!ObsKind<KIND>::ORIG && G::LPG == 64 && !G::WIDE && G::RC % 4 == 0, "single_kernel: 'extended' kinds, 4-aligned one-game-per-wave boards");
    constexpr int NT = 64 * SINGLE_WAVES, RC = G::RC;
    __shared__ Lds<G, ObsKind<KIND>::NIB_CH> L;
    __shared__ alignas(16) uint8_t shared[shared_table_bytes<G, KIND>()];
    __shared__ alignas(16) uint8_t obst_s[G::OBST_BYTES + COMBAT_BYTES];
    __shared__ StepOut so;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t env = P.env_first + blockIdx.x;
    GameInput in{};
    if (wave == 0) in = load_game<G, false>(P, env, lane);
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    constexpr int NP = tmpl_lds_bytes<G, KIND>(false), NF = FULL ? tmpl_lds_bytes<G, KIND>(true) : 0;
    {
        const int4 *tp = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0]);
        for (int i = tid; i < NP / 16; i += NT) reinterpret_cast<int4 *>(shared)[i] = tp[i];
        if constexpr (FULL) {
            const int4 *tf = reinterpret_cast<const int4 *>(P.tab->tmpl[(raw ? 2 : 0) + 1]);
            for (int i = tid; i < NF / 16; i += NT) reinterpret_cast<int4 *>(shared + NP)[i] = tf[i];
        }
        const int4 *ct = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0]);
        for (int i = tid; i < CODETAB_BYTES / 16; i += NT) reinterpret_cast<int4 *>(shared + NP + NF)[i] = ct[i];
        for (int i = tid; i < G::S / 4; i += NT) reinterpret_cast<int *>(obst_s)[i] = reinterpret_cast<const int *>(P.tab->obstacles)[i];
        for (int i = tid; i < COMBAT_BYTES / 4; i += NT) reinterpret_cast<int *>(obst_s + G::OBST_BYTES)[i] = reinterpret_cast<const int *>(P.tab->combat)[i];
    }
    __syncthreads();
    if (wave == 0) env_step<R_, C_, KIND, false, true>(P, L, shared, obst_s, env, lane, in, nullptr, &so);
    __syncthreads();
    if (P.io.mask_dev) emit_mask<G, NT>(L, P.io.mask_dev + env * (int64_t)G::NA, tid);
    const uint8_t *codetab = shared + NP + NF;
    auto render_all = [&](auto spec, bool full, float *dst) {
        using Spec = decltype(spec);
        if (wave == 0) {
            const int n_unc = build_codes<G, Spec>(L, full ? shared + NP : shared, codetab, P.tab->lut[(raw ? 2 : 0) + (full ? 1 : 0)], so.qi,
                                                   so.n_events, so.rp0, so.rp1, lane, P.piece_counts, raw, P.multi_ev != 0);
            if (lane == 0) so.n_unc = n_unc;
        }
        __syncthreads();
        const int n_unc = so.n_unc;
        if (n_unc == 0) emit_codes<G, Spec, false, false, NT>(L, dst, tid);
        else {
            emit_codes<G, Spec, true, false, NT>(L, dst, tid);
            if (wave == 0) patch_uncoded<G, Spec>(L, dst, n_unc, lane);
        }
        __syncthreads();                    // (the code buffer is free again)
    };
    if (P.io.obs_dev) render_all(PS{}, false, P.io.obs_dev + env * (int64_t)(RC * PS::NCH));
    if constexpr (FULL)
        if (P.io.fobs_dev) render_all(FS{}, true, P.io.fobs_dev + env * (int64_t)(RC * FS::NCH));
    // every thread's stores are visible to the host before its workgroup counts as done
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        bool last = true;
        if (gridDim.x > 1) {
            last = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
            if (last) __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace
