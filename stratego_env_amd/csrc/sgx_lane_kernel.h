// sgx_lane_kernel.h -- the step / observe kernel of boards with at most 16 cells (Micro 3x4, Tiny 4x4): ONE GAME PER LANE
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip (one translation unit).  The rules are sgx_lane.h's scalar functions
// (tested on the host against the oracle, tests/test_lane_logic_cpu.py); this file is what a wave does around them:
//
//   1. stage in     the wave's 64 records are ONE contiguous span of HBM (64 x 128 B): coalesced 16-byte loads into LDS, one record image
//                   per game at a stride of rec_bytes + 16 (conflict-free 16-byte reads by 64 lanes)
//   2. play         every lane unpacks ITS game into registers (nibble boards), decodes / validates / applies the action, generates the
//                   next mover's moves as bitboards, settles the endings, restarts finished games (auto_reset), writes the results
//                   (coalesced: lane i = env0 + i) and packs the game back into its record image
//   3. record out   the 64 record images leave as coalesced 16-byte stores
//   4. mask out     every lane renders its game's uint8 [R,C,K] mask into its row of an LDS scratch (row stride = NA bytes, an odd number
//                   of dwords on 3x4: no bank conflicts) and finds the k-th valid action on the way; the 64 rows are one contiguous span
//                   of the mask tensor: coalesced 16-byte stores
//   5. observation  in sub-batches of 16 games: the 4-bit code buffers of sgx_obs.h are built cooperatively from the record images
//                   (template copy; one lane per (game, cell) ORs the indicator codes; one lane per capture event / recent-move pair) and
//                   the sub-batch's observations -- a contiguous, line-aligned span of 16 x NOBS floats -- are swept out in 1 KiB store
//                   instructions exactly like emit_codes does for one game.  16 games at a time keep the code buffer at 6.5 KB and put the
//                   first observation stores right behind the game logic.
//
// Eligibility (host: lane_eligible): RC <= 16 and RC % 4 == 0, the 67-channel partial observation of an 'extended' channel mode, masks in
// the mover's perspective, no terminal-observation buffers, 16-byte aligned output tensors.  Everything else runs the wave-per-game kernel
// (sgx_step.h), and sgx_set_lane_kernel(h, 0) forces that one (in-process A/B; SGX_LANE=0 sets the default of new handles).
#pragma once

// (experiment knobs, -D...: observation sub-batch of an emitter, emitter waves per workgroup, quads per lane an emitter reads ahead of its
//  stores, occupancy promise -- the shipped values are the defaults; tools/lib_ab.py compares builds on the same buffers)
#ifndef SGX_KSTEP_SUB
#define SGX_KSTEP_SUB 8
#endif
#ifndef SGX_KSTEP_EMITTERS
#define SGX_KSTEP_EMITTERS 3
#endif
#ifndef SGX_LANE_EMIT_U
#define SGX_LANE_EMIT_U 8
#endif
#ifndef SGX_KSTEP_MIN_WAVES
#define SGX_KSTEP_MIN_WAVES 4
#endif

namespace {

constexpr int LANE_SUB = 16;                              // games per observation sub-batch

template <class G, int SUB = LANE_SUB>
struct LaneGeo {
    static constexpr int NCH = OBS_CH;
    static constexpr int REC_MAX = (G::EVL_OFF + G::EV_BYTES * G::EVL_MAX + 127) & ~127;      // 256 on every board of <= 16 cells
    static constexpr int NOBS = G::RC * NCH, NQ = NOBS / 4;                                  // floats / quads per observation
    static constexpr int NIBP = ((NOBS / 2) + 15) & ~15;                                     // the template of one observation's default codes, padded to 16 bytes
    // the sub-batch's codes are ONE contiguous nibble array (game gl at nibble gl * NOBS): the emission sweep then reads halfword q for
    // quad q of the span -- no division by the quads per game in the loop that issues 200 stores per wave
    static constexpr int CODES = ((SUB * NOBS / 2) + 15) & ~15;
    static constexpr int SCRATCH = (64 * G::NA > CODES ? 64 * G::NA : CODES) + 16;
    static constexpr int SLOTS = G::EVL_MAX + 4;                                             // special entries per game: events, then four pairs
    static constexpr int UNC_MAX = SUB * SLOTS;
};

template <class G>
constexpr bool lane_geometry() { return G::RC <= 16 && G::RC % 4 == 0 && !G::WIDE; }

template <class G>
struct alignas(16) LaneLds {
    using LG = LaneGeo<G>;
    alignas(16) uint8_t scratch[LG::SCRATCH];             // mask rows of the 64 games, then the code buffers of a sub-batch
    alignas(16) uint8_t tmpl[LG::NIBP];                   // default codes of one observation
    alignas(16) uint8_t codetab[CODETAB_BYTES];
    alignas(16) uint8_t combat[COMBAT_BYTES];
    alignas(16) float unc_val[LG::UNC_MAX];               // entries of the sub-batch whose value has no 4-bit code: float ...
    uint32_t unc_idx[LG::UNC_MAX];                        // ... and float index inside the sub-batch's span
    int unc_n;
};

// The observations of one sub-batch of `ng` (<= SUB) games, built from their record images (`rec_base` + game * stride, game g0 first) by
// one wave: the 4-bit code buffers of sgx_obs.h in `codes` (one contiguous nibble array: game gl at nibble gl * NOBS), then the
// sub-batch's span of the observation tensor (`dstf`: contiguous, 16-byte aligned) swept out in 1 KiB store instructions.
template <class G, int SUB>
__device__ __forceinline__ void lane_emit_obs(const uint8_t *rec_base, const int stride, const int g0, const int ng, uint8_t *codes, float *unc_val, uint32_t *unc_idx,
                                              int *unc_n, const uint8_t *tmpl, const uint8_t *codetab, const float *glut, const uint32_t obst_abs, float *dstf,
                                              const int nt, const int lane) {
    using LG = LaneGeo<G, SUB>;
    constexpr int RC = G::RC, S = G::S, NCH = LG::NCH;
    const uint8_t *lane_rec = rec_base;
    // (a) default codes: the template repeats every NOBS nibbles = NQ halfwords; dword j of the array = template halfwords
    //     (2j mod NQ, (2j + 1) mod NQ)
    {
        const uint16_t *t16 = reinterpret_cast<const uint16_t *>(tmpl);
        for (int j = lane; j < (ng * LG::NQ + 1) / 2; j += 64) {
            const int h0 = (2 * j) % LG::NQ, h1 = h0 + 1 == LG::NQ ? 0 : h0 + 1;
            reinterpret_cast<uint32_t *>(codes)[j] = (uint32_t)t16[h0] | ((uint32_t)t16[h1] << 16);
        }
    }
    if (lane == 0) *unc_n = 0;
    wave_sync<G>();
    // (b) one lane per (game, cell): the indicator entries that are set (build_codes of sgx_obs.h, from the record image)
    for (int idx = lane; idx < ng * RC; idx += 64) {
        const int gl = idx / RC, i = idx - gl * RC;
        const uint8_t *rec = lane_rec + (g0 + gl) * stride;
        const int q = (reinterpret_cast<const int32_t *>(rec + G::SC_OFF)[1] & F_PLAYER_M1) ? 1 : 0;
        unsigned int *nib = reinterpret_cast<unsigned int *>(codes);
        const int base = gl * LG::NOBS + (q ? RC - 1 - i : i) * NCH;
        const int own = rec[(B_PIECES + q) * S + i], own_po = rec[(B_PO + q) * S + i], en_po = rec[(B_PO + 1 - q) * S + i];
        const uint32_t st_own = reinterpret_cast<const uint32_t *>(rec + G::ST_OFF + q * G::SB)[0], st_en = reinterpret_cast<const uint32_t *>(rec + G::ST_OFF + (1 - q) * G::SB)[0];
        auto set_one = [&](int entry) { atomicOr(nib + (entry >> 3), (unsigned)NIB_ONE << ((entry & 7) * 4)); };
        if (own) set_one(base + PartialObs::OWN0 + own - 1);
        if (own_po) set_one(base + PartialObs::OWN_PO0 + own_po - 1);
        if (en_po) set_one(base + PartialObs::ENEMY_PO0 + en_po - 1);
        if ((obst_abs >> i) & 1u) set_one(base + PartialObs::OBST);
        if ((st_own >> i) & 1u) set_one(base + PartialObs::STILL0);
        if ((st_en >> i) & 1u) set_one(base + PartialObs::STILL0 + 1);
    }
    // (c) one lane per capture event / recent-move pair: its code replaces the channel default; a value without a code gets
    //     CODE_ESC and goes on the sub-batch's list of floats to patch in afterwards
    for (int idx = lane; idx < ng * LG::SLOTS; idx += 64) {
        const int gl = idx / LG::SLOTS, sl = idx - gl * LG::SLOTS;
        const uint8_t *rec = lane_rec + (g0 + gl) * stride;
        const int32_t *sc = reinterpret_cast<const int32_t *>(rec + G::SC_OFF);
        const int q = (sc[1] & F_PLAYER_M1) ? 1 : 0;
        const int n_events = sc[4] < (int)G::EVL_MAX ? sc[4] : (int)G::EVL_MAX;
        int cell = 0, ch = 0, v = 0, ti = 0;
        bool have = false, is_event = sl < (int)G::EVL_MAX;
        if (is_event) {
            if (sl < n_events) {
                const int e = (int)reinterpret_cast<const uint16_t *>(rec + G::EVL_OFF)[sl], b = (e >> G::CELL_BITS) & 31, pi = b >= 12 ? 1 : 0, t = b - 12 * pi;
                cell = e & G::CELL_MASK;
                ch = PartialObs::CAP0 + (pi == q ? 0 : 12) + t;
                v = (e >> G::EV_COUNT_SHIFT) + 1;
                ti = 16 * t + v;
                have = cell < RC;
            }
        } else {
            const int k = sl - (int)G::EVL_MAX, pl = k >> 1, pr = ((pl ? sc[6] : sc[5]) >> (16 * (k & 1))) & 0xFFFF;
            const int code = G::pair_code(pr);
            if (code != 0) {
                cell = G::pair_cell(pr);
                ch = PartialObs::REC0 + (pl == q ? 0 : 1);
                v = code + 3;
                ti = CODETAB_REC + v;
                have = cell < RC;
            }
        }
        if (have) {
            const int entry = (q ? RC - 1 - cell : cell) * NCH + ch;
            int now = codetab[ti];
            const int was = codetab[is_event ? (ti & ~15) : CODETAB_REC + 3];        // the default: count 0 / code 0
            if (now == CODE_NONE) {
                now = CODE_ESC;
                const int at = atomicAdd(unc_n, 1);
                unc_idx[at] = (uint32_t)(gl * LG::NOBS + entry);
                unc_val[at] = glut[lut_row(ch) + v];
            }
            const int ge = gl * LG::NOBS + entry;
            atomicXor(reinterpret_cast<unsigned int *>(codes) + (ge >> 3), (unsigned)(was ^ now) << ((ge & 7) * 4));
        }
    }
    wave_sync<G>();
    // (d) the sub-batch's span of the observation tensor, 1 KiB per store instruction on 1 KiB address boundaries (emit_codes)
    {
        f32x4 *base = reinterpret_cast<f32x4 *>(dstf);
        const int total = ng * LG::NQ;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(dstf) >> 4) & 63);
        const uint16_t *n16 = reinterpret_cast<const uint16_t *>(codes);
        // A wave emits 64 games by itself, often as the only wave of its SIMD: nothing hides a dependent LDS read -> convert ->
        // store chain, so EIGHT quads' codes are read first (one LDS round trip), then converted and stored; the store policy
        // is chosen outside the loop.
        auto sweep = [&](auto nt_tag) {
            constexpr bool NT = decltype(nt_tag)::value;
            constexpr int U = SGX_LANE_EMIT_U;
            for (int q0 = -m0; q0 < total; q0 += 64 * U) {
                unsigned x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 64 * u + lane;
                    x[u] = n16[(unsigned)q < (unsigned)total ? q : 0];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 64 * u + lane;
                    f32x4 o = {code_to_float(x[u]), code_to_float(x[u] >> 4), code_to_float(x[u] >> 8), code_to_float(x[u] >> 12)};
                    if ((unsigned)q < (unsigned)total) {
                        if constexpr (NT) __builtin_nontemporal_store(o, &base[q]);
                        else base[q] = o;
                    }
                }
            }
        };
        if (nt) sweep(std::true_type{}); else sweep(std::false_type{});
        const int n_unc = *unc_n;
        if (n_unc > 0) {                                   // (piece sets whose captured counts normalise to thirds / fifths: rare)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int k = lane; k < n_unc; k += 64) dstf[unc_idx[k]] = unc_val[k];
        }
    }
    wave_sync<G>();
}

template <int R_, int C_, bool OBSERVE>
__global__ __launch_bounds__(64) void lane_kernel(const KParams P) {
    using G = Geo<R_, C_>;
    using LG = LaneGeo<G>;
    static_assert(lane_geometry<G>(), "lane kernel: boards of at most 16 cells, a multiple of 4");
    constexpr int RC = G::RC, K = G::K, NA = G::NA;
    extern __shared__ __attribute__((aligned(16))) uint8_t lane_rec[];     // 64 record images, stride rec_bytes + 16
    __shared__ LaneLds<G> L;
    const int lane = threadIdx.x;
    const int64_t grp = group_of_block(P);
    const int64_t env0 = P.env_first + grp * 64;
    if (env0 >= P.n_envs) return;
    const int n_act = (int)((P.n_envs - env0) < 64 ? (P.n_envs - env0) : 64);
    const int64_t env = env0 + lane;
    const bool act = lane < n_act;
    const int rq = P.rec_bytes >> 4, rsh = 31 - __clz(rq), stride = P.rec_bytes + 16;   // (rec_bytes is 128 or 256: lane_eligible)
    const int mode = OBSERVE ? 1 : 0;
    const int sflags = P.io.flags;

    // ---- 1. stage in: records, action, the small tables
    {
        const int4 *src = reinterpret_cast<const int4 *>(P.boards + env0 * (int64_t)P.rec_bytes);
        for (int j = lane; j < n_act * rq; j += 64) {
            const int gl = j >> rsh, w = j - (gl << rsh);
            *reinterpret_cast<int4 *>(lane_rec + gl * stride + 16 * w) = src[j];
        }
    }
    int a_raw = 0;
    int4 pos_raw = make_int4(0, 0, 0, 0);
    if (mode == 0 && act) {
        if (sflags & SGX_STEP_ACTIONS_POSITIONS) pos_raw = reinterpret_cast<const int4 *>(P.io.actions_dev)[env];
        else a_raw = P.io.actions_dev[env];
    }
    const bool raw = (sflags & SGX_STEP_RAW_OBS) != 0;
    if (lane < LG::NIBP / 16) reinterpret_cast<int4 *>(L.tmpl)[lane] = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0])[lane];
    if (lane < CODETAB_BYTES / 16) reinterpret_cast<int4 *>(L.codetab)[lane] = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0])[lane];
    if (lane < COMBAT_BYTES / 16) reinterpret_cast<int4 *>(L.combat)[lane] = reinterpret_cast<const int4 *>(P.tab->combat)[lane];
    const uint32_t obst_abs = (uint32_t)__ballot(lane < RC && P.tab->obstacles[lane < RC ? lane : 0] != 0);
    wave_sync<G>();

    // ---- 2. play: one game per lane
    uint8_t *myrec = lane_rec + lane * stride;
    uint16_t *ev = reinterpret_cast<uint16_t *>(myrec + G::EVL_OFF);
    LaneGame g;
    lane_load<G>(g, myrec);                               // (lanes beyond n_act read their -- unwritten -- image: results unused)
    if (!act) { g.pc[0] = g.pc[1] = g.po[0] = g.po[1] = 0; g.still[0] = g.still[1] = 0; g.turn = g.flags = g.n_events = g.rp0 = g.rp1 = 0; g.game_no = 0; g.max_turns = 1; }
    int player = (g.flags & F_PLAYER_M1) ? -1 : 1;
    const int mover = player;
    LaneApplied ap{false, false};
    bool invalid_action = false;
    uint32_t V[K - 1];
    if (mode == 0) {
        const LaneMove m = lane_decode<G>(a_raw, pos_raw, sflags, player);
        bool has_moves = false;
        const bool wants_noop = m.valid && m.noop && !(g.flags & F_OVER);
        if (__any(wants_noop)) has_moves = lane_gen_moves<G>(g, player == 1 ? 0 : 1, obst_abs, false, V) != 0;     // (garbage actions only)
        ap = lane_apply<G>(g, ev, m, player, obst_abs, L.combat, P.max_events, sflags, has_moves);
        if (ap.applied) player = -player; else invalid_action = true;
    }
    int qi = player == 1 ? 0 : 1;
    int nvalid = lane_gen_moves<G>(g, qi, obst_abs, (g.flags & F_OVER) != 0, V);
    const bool over = lane_finish(g, ap, mover, nvalid);
    if (over && nvalid != 0) {
#pragma unroll
        for (int c = 0; c < K - 1; ++c) V[c] = 0;
        nvalid = 0;
    }
    g.flags = (g.flags & ~F_PLAYER_M1) | (player == -1 ? F_PLAYER_M1 : 0);
    const bool ended_now = ap.applied && over;
    if (mode == 0 && act) {                               // rewards / dones (maenv:699-805)
        const bool end_invalid = over && (g.flags & F_END_INVALID);
        float rew_p1 = 0.f, rew_m1 = 0.f;
        if (over && !end_invalid) {
            const int w = (g.flags & F_WIN_P1) ? 1 : (g.flags & F_WIN_M1) ? -1 : 0;
            rew_p1 = w == 0 ? 1e-4f : (float)w;            // impl:838-840
            rew_m1 = w == 0 ? 1e-4f : (float)-w;
        }
        if (P.io.reward_dev) reinterpret_cast<float2 *>(P.io.reward_dev)[env] = make_float2(rew_p1, rew_m1);
        if (P.io.done_dev) P.io.done_dev[env] = over ? 1 : 0;
        if (P.io.invalid_action_dev) P.io.invalid_action_dev[env] = invalid_action ? 1 : 0;
        if (P.io.ending_invalid_dev) P.io.ending_invalid_dev[env] = end_invalid ? 1 : 0;
    }
    bool wrote_reset = false;
    if (mode == 0 && P.io.auto_reset && ended_now && act) {      // the finished env starts its next game now
        g.game_no += 1;
        lane_sample_boards<G>(g, P.setups, P.n_setups, P.usable_rows, P.piece_counts, P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)g.game_no);
        g.turn = 0; g.flags = 0; g.n_events = 0; g.rp0 = g.rp1 = 0;
        player = 1; qi = 0;
        for (int i = 0; i < G::EVL_MAX; ++i)
            if (i < P.max_events) ev[i] = 0;
        nvalid = lane_gen_moves<G>(g, 0, obst_abs, false, V);
        wrote_reset = true;
    }
    if (act && P.io.player_dev) P.io.player_dev[env] = (int8_t)player;
    const bool changed = act && (ap.applied || wrote_reset);
    if (changed) lane_store<G>(g, myrec);
    wave_sync<G>();

    // ---- 3. record out (whole records, coalesced)
    if (__any(changed)) {
        int4 *dst = reinterpret_cast<int4 *>(P.boards + env0 * (int64_t)P.rec_bytes);
        for (int j = lane; j < n_act * rq; j += 64) {
            const int gl = j >> rsh, w = j - (gl << rsh);
            dst[j] = *reinterpret_cast<const int4 *>(lane_rec + gl * stride + 16 * w);
        }
    }

    // ---- 4. mask bytes (and the fused sampler)
    const bool want_next = mode == 0 && P.io.next_actions_dev != nullptr;
    if (P.io.mask_dev || want_next) {
        uint32_t *row = reinterpret_cast<uint32_t *>(L.scratch) + lane * (NA / 4);
        const int total = nvalid == 0 ? 1 : nvalid;
        const uint32_t k = rng_below(sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)g.game_no, STREAM_ACTION, (uint32_t)g.turn), (uint32_t)total);
        const int na = lane_emit_mask<G>(V, nvalid == 0, (int)k, [&](int j, uint32_t d) { row[j] = d; });
        if (want_next && act) P.io.next_actions_dev[env] = na;
        wave_sync<G>();
        if (P.io.mask_dev) {
            uint8_t *dst = P.io.mask_dev + env0 * (int64_t)NA;                   // 16-byte aligned: env0 is a multiple of 64
            const int n16 = (n_act * NA) >> 4, nd = (n_act * NA) >> 2;
            for (int j = lane; j < n16; j += 64) reinterpret_cast<int4 *>(dst)[j] = reinterpret_cast<const int4 *>(L.scratch)[j];
            if (4 * n16 + lane < nd) reinterpret_cast<uint32_t *>(dst)[4 * n16 + lane] = reinterpret_cast<const uint32_t *>(L.scratch)[4 * n16 + lane];
        }
        wave_sync<G>();
    }

    // ---- 5. observations, 16 games at a time
    if (P.io.obs_dev) {
        const float *glut = P.tab->lut[raw ? 2 : 0];
        for (int g0 = 0; g0 < n_act; g0 += LANE_SUB) {
            const int ng = n_act - g0 < LANE_SUB ? n_act - g0 : LANE_SUB;
            lane_emit_obs<G, LANE_SUB>(lane_rec, stride, g0, ng, L.scratch, L.unc_val, L.unc_idx, &L.unc_n, L.tmpl, L.codetab, glut, obst_abs,
                                       P.io.obs_dev + (env0 + g0) * (int64_t)LG::NOBS, P.nt_stores, lane);
            wave_sync<G>();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// lane_steps_kernel: K fused rollout steps of 64 games in ONE launch (sgx_step_n / sgx_step_ring on boards of at most 16 cells).
//
// Why.  One launch per step puts the game logic and the stores of 237 MB in series: every wave plays, then every wave stores (65,536
// Micro games: 13 us of logic + 31 us of stores; docs/DESIGN_rounds_4-5.md section 3.3).  Here a 256-thread workgroup keeps its 64 games in the
// REGISTERS of wave 0 for all K steps -- the lane-per-game rules of sgx_lane.h, the next action drawn in place -- and waves 1 .. 3 do
// nothing but emit observations: wave 0 plays step t + 1 while they sweep out step t.  Per step wave 0 writes what the emitters need --
// the 64 record images -- into one of TWO LDS buffers, the results and the mask rows (coalesced, by itself), and meets the emitters at
// ONE workgroup barrier; an emitter takes the sub-batches of 8 games that are its share of the step (rotating, so that the 8
// sub-batches spread evenly over 3 waves) and runs lane_emit_obs on them.  The records travel to and from HBM once per launch
// instead of once per step.  Same results as n_steps launches of lane_kernel / step_kernel: tests/test_gpu_lane_kernel.py.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KSTEP_SUB = SGX_KSTEP_SUB, KSTEP_EMITTERS = SGX_KSTEP_EMITTERS, KSTEP_MAX_SETS = 8;

struct StepsParams {
    KParams k;
    int32_t n_steps, n_sets, first_set;
    int32_t strided;                      // 1 (sgx_step_traj): set s = the tensors of set 0 + s x the byte strides below (any number of slots);
    int64_t obs_slot_bytes, mask_slot_bytes;       // 2: more separate sets than fit the kernel arguments -- pointer tables in device memory
    float *const *obs_tab;
    uint8_t *const *mask_tab;
    float *obs[KSTEP_MAX_SETS];           // the observation / mask tensor of output set s (sgx_step_ring); one set: sgx_step_n, in place
    uint8_t *mask[KSTEP_MAX_SETS];
    __device__ __forceinline__ float *obs_of(int set) const {
        return strided == 2 ? table_entry(obs_tab, set) : strided ? reinterpret_cast<float *>(reinterpret_cast<char *>(obs[0]) + (int64_t)set * obs_slot_bytes) : obs[set];
    }
    __device__ __forceinline__ uint8_t *mask_of(int set) const {
        return strided == 2 ? table_entry(mask_tab, set) : strided ? (mask[0] ? mask[0] + (int64_t)set * mask_slot_bytes : nullptr) : mask[set];
    }
};

template <class G>
struct alignas(16) StepsLds {
    using LG = LaneGeo<G, KSTEP_SUB>;
    alignas(16) uint8_t maskrows[64 * G::NA + 16];
    alignas(16) uint8_t tmpl[LG::NIBP];
    alignas(16) uint8_t codetab[CODETAB_BYTES];
    alignas(16) uint8_t combat[COMBAT_BYTES];
    struct alignas(16) Emitter {
        alignas(16) uint8_t codes[LG::CODES + 16];
        alignas(16) float unc_val[LG::UNC_MAX];
        uint32_t unc_idx[LG::UNC_MAX];
        int unc_n;
    } em[KSTEP_EMITTERS];
};

template <int R_, int C_>
__global__ __launch_bounds__(64 * (1 + KSTEP_EMITTERS), SGX_KSTEP_MIN_WAVES) void lane_steps_kernel(const StepsParams SP) {
    using G = Geo<R_, C_>;
    using LG = LaneGeo<G, KSTEP_SUB>;
    static_assert(lane_geometry<G>(), "lane kernels: boards of at most 16 cells, a multiple of 4");
    constexpr int RC = G::RC, K = G::K, NA = G::NA, NT = 64 * (1 + KSTEP_EMITTERS);
    const KParams &P = SP.k;
    extern __shared__ __attribute__((aligned(16))) uint8_t steps_rec[];    // 2 x 64 record images, stride rec_bytes + 16
    __shared__ StepsLds<G> L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t env0 = P.env_first + group_of_block(P) * 64;
    if (env0 >= P.n_envs) return;                                          // (the whole workgroup: no barrier is left behind)
    const int n_act = (int)((P.n_envs - env0) < 64 ? (P.n_envs - env0) : 64);
    const int64_t env = env0 + lane;
    const bool act = lane < n_act;
    const int rq = P.rec_bytes >> 4, rsh = 31 - __clz(rq), stride = P.rec_bytes + 16, buf_bytes = 64 * stride;
    const int sflags = P.io.flags;
    const bool raw = (sflags & SGX_STEP_RAW_OBS) != 0;
    // ---- stage in: the 64 records into BOTH buffers (so that every byte of an image is defined whichever buffer goes back to HBM), tables
    {
        const int4 *src = reinterpret_cast<const int4 *>(P.boards + env0 * (int64_t)P.rec_bytes);
        for (int j = tid; j < n_act * rq; j += NT) {
            const int gl = j >> rsh, w = j - (gl << rsh);
            const int4 v = src[j];
            *reinterpret_cast<int4 *>(steps_rec + gl * stride + 16 * w) = v;
            *reinterpret_cast<int4 *>(steps_rec + buf_bytes + gl * stride + 16 * w) = v;
        }
        for (int i = tid; i < LG::NIBP / 16; i += NT) reinterpret_cast<int4 *>(L.tmpl)[i] = reinterpret_cast<const int4 *>(P.tab->tmpl[raw ? 2 : 0])[i];
        for (int i = tid; i < CODETAB_BYTES / 16; i += NT) reinterpret_cast<int4 *>(L.codetab)[i] = reinterpret_cast<const int4 *>(P.tab->codetab[raw ? 1 : 0])[i];
        for (int i = tid; i < COMBAT_BYTES / 16; i += NT) reinterpret_cast<int4 *>(L.combat)[i] = reinterpret_cast<const int4 *>(P.tab->combat)[i];
    }
    const uint32_t obst_abs = (uint32_t)__ballot(lane < RC && P.tab->obstacles[lane < RC ? lane : 0] != 0);
    __syncthreads();

    if (wave == 0) {
        // ================= the players: one game per lane, in registers for all n_steps =================
#ifdef SGX_KSTEP_PRIO
        __builtin_amdgcn_s_setprio(SGX_KSTEP_PRIO);                        // experiment: the one wave whose latency is the step's critical path issues first
#endif
        LaneGame g;
        lane_load<G>(g, steps_rec + buf_bytes + lane * stride);            // (buffer 1: step 0 writes buffer 0)
        if (!act) { g.pc[0] = g.pc[1] = g.po[0] = g.po[1] = 0; g.still[0] = g.still[1] = 0; g.turn = g.flags = g.n_events = g.rp0 = g.rp1 = 0; g.game_no = 0; g.max_turns = 1; }
        int na = act ? P.io.actions_dev[env] : 0;
        uint32_t V[K - 1];
        for (int t = 0; t < SP.n_steps; ++t) {
            uint8_t *myrec = steps_rec + (t & 1) * buf_bytes + lane * stride;
            const uint8_t *prevrec = steps_rec + ((t & 1) ^ 1) * buf_bytes + lane * stride;
            uint16_t *ev = reinterpret_cast<uint16_t *>(myrec + G::EVL_OFF);
            {   // the capture-event list lives in the record image: it moves on to this step's buffer
                const uint16_t *pev = reinterpret_cast<const uint16_t *>(prevrec + G::EVL_OFF);
                for (int i = 0; i < G::EVL_MAX; ++i)
                    if (i < P.max_events) ev[i] = pev[i];
            }
            int player = (g.flags & F_PLAYER_M1) ? -1 : 1;
            const int mover = player;
            bool invalid_action = false;
            const LaneMove m = lane_decode<G>(na, make_int4(0, 0, 0, 0), sflags, player);
            bool has_moves = false;
            const bool wants_noop = m.valid && m.noop && !(g.flags & F_OVER);
            if (__any(wants_noop)) has_moves = lane_gen_moves<G>(g, player == 1 ? 0 : 1, obst_abs, false, V) != 0;     // (garbage actions only)
            const LaneApplied ap = lane_apply<G>(g, ev, m, player, obst_abs, L.combat, P.max_events, sflags, has_moves);
            if (ap.applied) player = -player; else invalid_action = true;
            int nvalid = lane_gen_moves<G>(g, player == 1 ? 0 : 1, obst_abs, (g.flags & F_OVER) != 0, V);
            const bool over = lane_finish(g, ap, mover, nvalid);
            if (over && nvalid != 0) {
#pragma unroll
                for (int c = 0; c < K - 1; ++c) V[c] = 0;
                nvalid = 0;
            }
            g.flags = (g.flags & ~F_PLAYER_M1) | (player == -1 ? F_PLAYER_M1 : 0);
            const bool ended_now = ap.applied && over;
            const int set = (SP.first_set + t) % SP.n_sets;
            const int64_t renv = env + (int64_t)set * P.traj_res_envs;     // (sgx_step_traj with per-slot results; else traj_res_envs = 0)
            if (act) {                                                     // rewards / dones (maenv:699-805)
                const bool end_invalid = over && (g.flags & F_END_INVALID);
                float rew_p1 = 0.f, rew_m1 = 0.f;
                if (over && !end_invalid) {
                    const int w = (g.flags & F_WIN_P1) ? 1 : (g.flags & F_WIN_M1) ? -1 : 0;
                    rew_p1 = w == 0 ? 1e-4f : (float)w;                    // impl:838-840
                    rew_m1 = w == 0 ? 1e-4f : (float)-w;
                }
                if (P.io.reward_dev) reinterpret_cast<float2 *>(P.io.reward_dev)[renv] = make_float2(rew_p1, rew_m1);
                if (P.io.done_dev) P.io.done_dev[renv] = over ? 1 : 0;
                if (P.io.invalid_action_dev) P.io.invalid_action_dev[renv] = invalid_action ? 1 : 0;
                if (P.io.ending_invalid_dev) P.io.ending_invalid_dev[renv] = end_invalid ? 1 : 0;
            }
            if (P.io.auto_reset && ended_now && act) {                     // the finished env starts its next game now
                g.game_no += 1;
                lane_sample_boards<G>(g, P.setups, P.n_setups, P.usable_rows, P.piece_counts, P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)g.game_no);
                g.turn = 0; g.flags = 0; g.n_events = 0; g.rp0 = g.rp1 = 0;
                player = 1;
                for (int i = 0; i < G::EVL_MAX; ++i)
                    if (i < P.max_events) ev[i] = 0;
                nvalid = lane_gen_moves<G>(g, 0, obst_abs, false, V);
            }
            if (act && P.io.player_dev) P.io.player_dev[renv] = (int8_t)player;
            lane_store<G>(g, myrec);                                       // (always: the emitters read this step's image)
            // mask rows (coalesced out by this wave) and the next action
            {
                uint32_t *row = reinterpret_cast<uint32_t *>(L.maskrows) + lane * (NA / 4);
                const int total = nvalid == 0 ? 1 : nvalid;
                const uint32_t k = rng_below(sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)g.game_no, STREAM_ACTION, (uint32_t)g.turn), (uint32_t)total);
                na = lane_emit_mask<G>(V, nvalid == 0, (int)k, [&](int j, uint32_t d) { row[j] = d; });
                if (act && P.traj_act_log) P.traj_act_log[env + (int64_t)set * P.traj_out_envs] = na;
                wave_sync<G>();
                if (uint8_t *mset = SP.mask_of(set)) {
                    uint8_t *dst = mset + env0 * (int64_t)NA;              // 16-byte aligned: env0 is a multiple of 64
                    const int n16 = (n_act * NA) >> 4, nd = (n_act * NA) >> 2;
                    for (int j = lane; j < n16; j += 64) reinterpret_cast<int4 *>(dst)[j] = reinterpret_cast<const int4 *>(L.maskrows)[j];
                    if (4 * n16 + lane < nd) reinterpret_cast<uint32_t *>(dst)[4 * n16 + lane] = reinterpret_cast<const uint32_t *>(L.maskrows)[4 * n16 + lane];
                }
                wave_sync<G>();
            }
            __syncthreads();                                               // barrier t: image t is complete, and emission t - 1 is over
        }
        // ---- the records and the next action go back to HBM once
        if (act && P.io.next_actions_dev) P.io.next_actions_dev[env] = na;
        {
            const uint8_t *last = steps_rec + ((SP.n_steps - 1) & 1) * buf_bytes;
            int4 *dst = reinterpret_cast<int4 *>(P.boards + env0 * (int64_t)P.rec_bytes);
            for (int j = lane; j < n_act * rq; j += 64) {
                const int gl = j >> rsh, w = j - (gl << rsh);
                dst[j] = *reinterpret_cast<const int4 *>(last + gl * stride + 16 * w);
            }
        }
    } else {
        // ================= the emitters: the observations of step t while wave 0 plays step t + 1 =================
        typename StepsLds<G>::Emitter &E = L.em[wave - 1];
        const float *glut = P.tab->lut[raw ? 2 : 0];
        for (int t = 0; t < SP.n_steps; ++t) {
            __syncthreads();                                               // barrier t
            const int set = (SP.first_set + t) % SP.n_sets;
            float *obs = SP.obs_of(set);
            const uint8_t *img = steps_rec + (t & 1) * buf_bytes;
            for (int b = 0, g0 = 0; g0 < n_act; ++b, g0 += KSTEP_SUB) {
                if ((b + t) % KSTEP_EMITTERS != wave - 1) continue;
                const int ng = n_act - g0 < KSTEP_SUB ? n_act - g0 : KSTEP_SUB;
                lane_emit_obs<G, KSTEP_SUB>(img, stride, g0, ng, E.codes, E.unc_val, E.unc_idx, &E.unc_n, L.tmpl, L.codetab, glut, obst_abs,
                                            obs + (env0 + g0) * (int64_t)LG::NOBS, P.nt_stores, lane);
            }
        }
    }
}

}  // namespace
