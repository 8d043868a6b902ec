// sgx_layout.h -- geometry, HBM record / LDS layout, kernel parameters, counter RNG, per-game wave helpers, observation channel specs
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Geometry and layout
// ---------------------------------------------------------------------------------------------
constexpr int OBS_CH = SGX_PO_OBS_CHANNELS;  // 67
constexpr int LUT_STRIDE = SGX_OBS_LUT_STRIDE;
// Device placement of the LUT rows.  A wave renders 64 consecutive float4 "quads", so the lanes of one LDS
// access hold channels 4q+j (mod 67).  Rows are placed so that bank(row(ch)) = (ch/4 + {0,16,1,17}[ch%4]) mod 32:
// the 32 lanes of an access group then hit 31-32 different banks (a dense ch*16 layout put them all on 2 banks:
// 78 % of LDS cycles were bank conflicts, profiles/r01_v1_*).  Two 16-entry rows share each 33-dword pitch.
constexpr int FOBS_CH = SGX_FO_OBS_CHANNELS;  // 79
constexpr int LUT_ROW_PITCH = 33, LUT_BLK = 673 /* >= 20*33 (79 channels), = 1 mod 32 */, LUT_DWORDS = 2 * LUT_BLK + 2;  // 1348
#ifndef SGX_WPB
#define SGX_WPB 8
#endif
#ifndef SGX_MIN_WAVES
#define SGX_MIN_WAVES 6
#endif
// waves per workgroup (WPB * Geo::GPW games; they share the observation tables): SGX_WPB wherever the games' LDS regions fit the
// 160 KiB of a CU, fewer on the longest thin boards (3 x 85: 21 KB per game) -- Geo::WPB
// 'original' channel mode only -- per observation kind in LDS: the LUT followed by the quad table (2 perspectives x NCH quads x 4
// packed entries, see emit_obs_lut)
constexpr int QTAB_DWORDS = 2 * OBS_CH * 4, OBS_TAB_DWORDS = LUT_DWORDS + QTAB_DWORDS;   // partial kind; the full kind follows it
constexpr int FOBS_TAB_DWORDS = LUT_DWORDS + 2 * FOBS_CH * 4;
// 'extended' channel mode: an observation is rendered from 4-bit codes, one per float, decoded by v_cvt_off_f32_i4 (x 4):
// 0 -> 0.0, 2 -> 0.5, 4 -> 1.0, 12 -> -1.0.  Every channel has a default code (one-hot / obstacle / never-moved: 0; recent
// moves: 0.5 normalised, 0 raw; captured counts: -1 normalised, 0 raw) held in a per-variant template; the few entries that are
// neither default nor 1.0 (captured counts >= 1, non-zero recent-move codes) are patched into the output afterwards.
constexpr int NIB_ONE = 4, CODE_NONE = 0xFF, CODETAB_REC = 192, CODETAB_BYTES = 208;
constexpr int COMBAT_BYTES = 16 * 16;
enum { COMBAT_LOSE = 0, COMBAT_TIE = 1, COMBAT_WIN = 2, COMBAT_WIN_FLAG = 3 };
// Outcome of piece type `att` attacking piece type `def` (both 1..12): the reference's first-match chain, impl:968-982 (tabulated by
// the host into DevTables::combat).
__host__ __device__ inline int combat_outcome(int att, int def) {
    if (att == 3 /* miner */ && def == 12 /* bomb */) return COMBAT_WIN;      // only the miner defuses
    if (att == 1 /* spy */ && def == 10 /* marshal */) return COMBAT_WIN;     // the spy wins only when it attacks
    if (def == 11 /* flag */) return COMBAT_WIN_FLAG;
    if (def == 12) return COMBAT_LOSE;
    if (att == def) return COMBAT_TIE;
    return att > def ? COMBAT_WIN : COMBAT_LOSE;
}
constexpr int CODE_ESC = 8;   // (-2.0, never a value of its own) marks an entry whose float has no code: see emit_codes / patch_uncoded
constexpr int TMPL_MAX_BYTES = ((SGX_MAX_CELLS * FOBS_CH / 2) + 15) & ~15;   // 40,448
__host__ __device__ constexpr int lut_row(int ch) { return ((ch >> 1) & 1) * LUT_BLK + (ch >> 2) * LUT_ROW_PITCH + (ch & 1) * 16; }

// internal board indices inside an env record (each board is S bytes, absolute coordinates)
constexpr int B_PIECES = 0;   // +pi : true pieces of player index pi (0 = player +1, 1 = player -1)   impl layers 0/1
constexpr int B_PO = 2;       // +pi : what the opponent knows of pi's pieces                           impl layers 3/4
constexpr int B_STILL = 4;    // +pi : never-moved flags                                                impl layers 32/33
constexpr int B_RECENT = 6;   // +pi : two-square bookkeeping (LDS only, rebuilt from scal)             impl layers 6/7
constexpr int N_BOARDS = 8;   // boards that make up a game in LDS (cleared by a reset)
constexpr int STORED_BOARDS = 4;  // boards 0..3 live in HBM as bytes; never-moved flags as bitmaps; the rest sparsely:
//   recent moves: at most two non-zero cells per player (impl:1013-1028)  -> two (cell, code) pairs per player in scal
//   captured counts (impl layers 8-19 / 20-31): never dense, not even in LDS -- one EVENT per (layer, cell) with a non-zero
//   count: uint16 = (count - 1) << 13 | (12 * pi + type - 1) << 8 | cell, <= 2 * pieces per side entries; a count cannot
//   exceed the 8 pieces of the most numerous type.  Boards of more than 256 cells (Geo::WIDE; the reference's
//   StrategoProceduralEnv takes any size, penv:27-36) have 10-bit cell indices: events are uint32 = (count - 1) << 15 | key << 10 | cell
//   and a recent pair keeps 6 instead of 8 bits for its code
constexpr int B_OBST = 8;     // LDS only: per-variant obstacle map (impl layer 2)
constexpr int B_ZERO = 9;     // LDS only: all zero ('original' channel mode reads the captured-count channels' defaults from it)
constexpr int N_LDS_BOARDS = 10;
constexpr int EV_COUNT_MAX = 8;          // (Geo::EV_COUNT_SHIFT / EV_KEY_MASK: the field positions depend on the cell-index width)

// record scalars (32 B at SC_OFF): {turn, flags, max_turns, game_no} {n_events, recent pairs of +1, recent pairs of -1, 0}
// a recent pair is 16 bits: cell | code << Geo::CELL_BITS (two's complement code in the remaining bits), two pairs per int (low / high
// half); code 0 = empty
constexpr int F_OVER = 1, F_WIN_P1 = 2, F_WIN_M1 = 4, F_END_INVALID = 8, F_PLAYER_M1 = 16;

enum { SP_SPY = 1, SP_SCOUT = 2, SP_MINER = 3, SP_MARSHALL = 10, SP_FLAG = 11, SP_BOMB = 12, SP_UNKNOWN = 13 };

// VAR_ = 1 ("BIG"): the general-state variant of a geometry, used ONLY inside sgx_step_states' second pass for states the packed record
// cannot carry (more than two recent-move cells per player, more capture cells than pieces, more than 8 captures on one cell): 32-bit
// capture events with room for every (layer, cell) pair and counts up to 32,768, and the two recent-move layers kept DENSE in the
// record image.  Such records never exist in HBM -- the image lives in the LDS of the fused states_kernel between import and export.
template <int R_, int C_, int VAR_ = 0>
struct Geo {
    static constexpr int R = R_, C = C_;
    static constexpr int RC = R * C;
    static constexpr bool BIG = VAR_ == 1;
    // VAR_ = 2 ("HALF"): TWO games per wave on a board that is one game per wave otherwise -- the no-observation kernel kinds only (search
    // expansion, mask-only and logic-only rollouts: sgx_step.h).  Those launches are bound by VALU issue, not by memory: 505 VALU instructions per
    // 10x10 game-step whether 64 or 32 lanes take part, so two games share every instruction (half_wave_ok<R, C>() says where it applies).
    // Measured, 65,536 games, us per step one / two games per wave (tools/half_wave_ab.py, profiles/r06_half_wave_ab.log): Barrage logic-only
    // rollout 60.6 / 33.2, mask only 72.7 / 46.6, search expansion 69.2 / 37.1 (0.95 -> 1.77 G states/s); Standard 59.9 / 37.2, 75.1 / 51.4, 68.0 / 38.1.
    // (The compact-output kind gains nothing from it -- 95 / 91 us, Standard 101 / 103: its code buffer doubles per wave -- and keeps one game per wave.)
    static constexpr bool HALF = VAR_ == 2;
    // cell-index width of the packed record: 8 bits up to 256 cells, 10 bits beyond (up to SGX_MAX_CELLS = 1024)
    static constexpr bool WIDE = RC > 256;
    static constexpr int CELL_BITS = WIDE ? 10 : 8, CELL_MASK = (1 << CELL_BITS) - 1, CODE_BITS = 16 - CELL_BITS;
    using ev_t = std::conditional_t<(WIDE || BIG), uint32_t, uint16_t>;   // capture event: (count - 1) << EV_COUNT_SHIFT | key << CELL_BITS | cell
    static constexpr int COUNT_MAX = BIG ? (1 << 15) : EV_COUNT_MAX;     // most captured pieces one event counts
    using cell_t = std::conditional_t<WIDE, uint16_t, uint8_t>;       // a cell index in LDS scratch lists
    using entry_t = std::conditional_t<WIDE, uint32_t, uint16_t>;     // an observation entry index (cell * channels + channel)
    static constexpr int EV_COUNT_SHIFT = CELL_BITS + 5, EV_KEY_MASK = (1 << EV_COUNT_SHIFT) - 1;
    __host__ __device__ static constexpr int pair_cell(int pr) { return pr & CELL_MASK; }
    __host__ __device__ static constexpr int pair_code(int pr) { return (int)((unsigned)(pr & 0xFFFF) << 16) >> (16 + CELL_BITS); }   // sign-extended
    __host__ __device__ static constexpr int make_pair(int cell, int code) { return cell | ((code & ((1 << CODE_BITS) - 1)) << CELL_BITS); }
    static constexpr int S = (RC + 3) & ~3;           // board stride (bytes)
    static constexpr int OBST_BYTES = WIDE ? ((RC + 15) & ~15) : 256;   // the workgroup's LDS copy of the obstacle map (the combat table follows it)
    static constexpr int LDS_BOARDS_BYTES = N_BOARDS * S;            // bytes of the 8 boards of a game in LDS (multiple of 16)
    // HBM record (a multiple of 128 B, so every record is read and written as whole cache lines):
    //   [0, 4S) four dense boards (true pieces, PO pieces) | zero padding to 16 | ST_OFF: never-moved bitmaps 2 x SB |
    //   SC_OFF: 32 B scalars | EVL_OFF: capture events uint16[max_events] | zero padding
    //   10x10: Barrage 512 B (4 lines), Standard 640 B (5 lines)
    static constexpr int ST_OFF = (STORED_BOARDS * S + 15) & ~15;
    static constexpr int SB = (((RC + 7) / 8) + 15) & ~15;           // bytes of one never-moved bitmap (bit i = cell i)
    static constexpr int SC_OFF = ST_OFF + 2 * SB, EVL_OFF = SC_OFF + 32;
    static constexpr int EVL_MAX = BIG ? 24 * RC : RC;               // 2 * pieces per side <= cells; BIG: every (layer, cell) pair
    static constexpr int EV_BYTES = (int)sizeof(ev_t);
    static constexpr int EVB = (EV_BYTES * EVL_MAX + 15) & ~15;      // bytes of the event list, padded to 16
    static constexpr int S_PAD = (S + 15) & ~15;
    static constexpr int RECB_OFF = EVL_OFF + EVB;                   // BIG only: the two recent-move boards, dense, S_PAD bytes each
    static constexpr int TAIL_BYTES = 2 * SB + 32 + EVB + (BIG ? 2 * S_PAD : 0);   // LDS image of the record from ST_OFF on
    static constexpr int IMG_BYTES = ST_OFF + TAIL_BYTES;            // the whole record image
    static constexpr int SPECIAL_EXTRA = BIG ? 2 * RC : 4;           // observation entries of the recent-move layers: four pairs, or every cell
    static constexpr int K = 2 * (R - 1) + 2 * (C - 1) + 1;
    static constexpr int NA = RC * K;                 // spatial actions
    static constexpr int NA_PAD = (NA + 15) & ~15;
    static constexpr int MB_WORDS = (((NA + 31) / 32 + 1) + 3) & ~3;  // mask as bits in LDS (+1 slack word), multiple of 4
    static constexpr int MPA = R + C;
    static constexpr int AS = RC * MPA + 1;           // 1-D action size (impl:252-254)
    static constexpr int NOBS = RC * OBS_CH;          // floats per observation
    // Lanes per game.  A 64-lane wave is one game on boards of more than 32 cells; toy boards share a wave between 2 or 4
    // games (each VALU instruction costs 4 cycles whether 12 or 64 lanes do useful work: one 3x4 game per wave ran the chip
    // at the VALU issue limit with 80 % of the lanes idle).  Everything below that says `lane` means the lane inside the game.
    // (twice the games per wave -- 8 / 16 / 32 lanes per game, two cells per lane up to 8x8 -- is bit-exact too but 3-5 % slower on
    // every board size: the cell loops double)
    // (The general-state variant is one game per wave on every board: it only runs in the fused states_kernel, one state per block.)
    static constexpr int LPG = BIG ? 64 : HALF ? 32 : (RC > 32 ? 64 : (RC <= 16 ? 16 : 32));
    static constexpr int GPW = 64 / LPG;              // games per wave
    // a wave may play several games in turn with the next game's reads in flight (13 more VGPRs: one-game-per-wave boards only)
    static constexpr bool PIPELINED = LPG == 64;      // observation lines written whole leave as non-temporal stores (sgx_obs.h)
    static constexpr int CPL = (RC + LPG - 1) / LPG;  // cells per lane
    static constexpr int CNT_PAD = CPL * LPG;
    // upper estimate of one game's LDS region (struct Lds with the widest code buffer) + the workgroup's shared tables
    static constexpr int LDS_GAME_EST = N_LDS_BOARDS * S + (RC * FOBS_CH / 2 + 64) + 4 * MB_WORDS + (1 + (int)sizeof(cell_t)) * CNT_PAD + S + TAIL_BYTES +
                                        (4 + (int)sizeof(entry_t)) * (EVL_MAX + SPECIAL_EXTRA + 8) + 64;
    // the workgroup's shared tables; WIDE boards read the default-code templates from global memory (L2) instead of an LDS copy
    static constexpr int LDS_SHARED_EST = (WIDE ? 0 : 2 * (RC * FOBS_CH / 2 + 32)) + 1024 + S;
    // (two games per wave: four waves per workgroup -- the same eight games per workgroup and the same LDS footprint as the board's ordinary kernels)
    static constexpr int WPB = HALF ? 4 : (SGX_WPB * GPW * LDS_GAME_EST + LDS_SHARED_EST <= 160 * 1024) ? SGX_WPB
                             : (4 * GPW * LDS_GAME_EST + LDS_SHARED_EST <= 160 * 1024) ? 4
                             : (2 * GPW * LDS_GAME_EST + LDS_SHARED_EST <= 160 * 1024) ? 2 : 1;
};

// the HALF variant of a board: one-game-per-wave boards of up to 128 cells whose record image still stages as two int4 per lane
template <int R, int C>
constexpr bool half_wave_ok() {
    using G0 = Geo<R, C>;
    return G0::LPG == 64 && !G0::WIDE && G0::RC <= 128 && (G0::ST_OFF + G0::TAIL_BYTES) / 16 <= 64;
}

struct DevTables {
    // observation LUTs, rows at lut_row(ch); index = 4 * original + 2 * raw + full:
    //   original: obs_channel_mode 'original' (32/33 channels) instead of 'extended' (67/79)
    //   raw: SGX_STEP_RAW_OBS, un-normalised channel values (penv:157-173 return raw observations)
    //   full: the fully-observable observation instead of the partial one
    float lut[8][LUT_DWORDS];
    uint8_t obstacles[SGX_MAX_CELLS];
    // 'extended' mode code templates, index = 2 * raw + full: nibble e (entry e = cell * NCH + ch of one observation, low nibble
    // first) = the default code of channel ch
    alignas(16) uint8_t tmpl[4][TMPL_MAX_BYTES];
    // codes of the values that are not channel defaults, index = raw: [16 * (type - 1) + count] for a captured count (the own and
    // the enemy block of a piece type share their normalisation), [CODETAB_REC + code + 3] for a recent-move code; CODE_NONE where
    // the float is not one of the 16 decodable values (such entries are patched into the output as floats)
    alignas(16) uint8_t codetab[2][CODETAB_BYTES];
    // combat outcome of attacker type a on defender type d (impl:968-982): [16 * a + d] = COMBAT_*
    alignas(16) uint8_t combat[COMBAT_BYTES];
};

// Kernel parameters read INSIDE a loop of steps / games are read through the kernel-argument segment's own address space: a load from
// address space 4 is a scalar load wherever the address is uniform.  Through a generic pointer the compiler cannot prove that the
// loop's global stores leave the parameters alone and falls back to VECTOR loads (+ v_readfirstlane) -- which also retire in order
// with the wave's outstanding stores, so every re-read waited for the step's own observation stores.
#if defined(__HIP_DEVICE_COMPILE__)
#define SGX_KERNARG __attribute__((address_space(4)))
#else
#define SGX_KERNARG
#endif
// An entry of a small read-only pointer table in global memory (the output sets of a ring with more sets than fit the kernel arguments), read
// through the constant address space: with a uniform index that is a scalar load, which does not queue behind the wave's own stores.
template <class T>
__device__ __forceinline__ T table_entry(T const *tab, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ((const SGX_KERNARG T *)tab)[i];
#else
    return tab[i];
#endif
}

struct KParams {
    int8_t *boards;
    const DevTables *tab;
    const uint8_t *setups;
    int32_t n_setups;
    int32_t max_turns;
    int32_t usable_rows;
    int32_t piece_counts[12];
    int32_t rec_bytes;   // bytes of one env record in HBM: EVL_OFF + 2 * max_events rounded up to 128
    int32_t max_events;
    int64_t n_envs;
    uint64_t seed;
    int64_t env_id_offset;
    sgx_step_io io;
    int32_t mode;  // 0 = step, 1 = observe
    int64_t env_first;  // this launch plays envs [env_first, n_envs) of the handle (sgx_rollout splits a batch over concurrent chains)
    int32_t map_mode, map_arg;   // experiment only: see group_of_block
    int32_t xcd_first[8], xcd_count[8];   // workgroup-groups of this launch played by XCD x: [xcd_first[x], + xcd_count[x]) (group_of_block)
    int32_t nt_stores;  // the launch's observations do not fit the Infinity Cache: whole lines leave as non-temporal stores (sgx_obs.h)
    // A variant with more than 8 pieces of one type (the reference's piece_amounts is unbounded, config.py:3-23): a capture event counts to
    // 8, so a ninth capture of one type on one cell opens ANOTHER event with the same (layer, cell) key.  Every reader of the list then
    // sums the events of a key (the first one speaks for all); 0 for every variant of the reference, whose lists never hold duplicates.
    int32_t multi_ev;
    int32_t compact_stride;   // SGX_STEP_COMPACT_OBS: bytes of one game's compact observation record (sgx_compact_obs_stride)
    // functional-API instantiation only (sgx_expand): game i is read from record src_index[i] (i when NULL) of ANOTHER handle's
    // records and written to record i of this one, whether or not the move was valid
    const int8_t *src_boards;
    const int32_t *src_index;
    // sgx_step_states: the states an import had to alter are appended here (count by atomicAdd), so that the general-state pass visits
    // only those instead of launching a block per state; NULL elsewhere
    int32_t *flag_list, *flag_count;
    // sgx_step_traj (multi-step launches only): the step that writes slot s of a trajectory buffer writes its per-step results (rewards,
    // flags, player) at env + s * traj_res_envs (0: in place) and the action it drew to traj_act_log[env + s * traj_out_envs]
    int64_t traj_out_envs, traj_res_envs;
    int32_t *traj_act_log;
#ifdef SGX_STAMPS
    unsigned long long *stamps;  // diagnostic build only: [N][16] s_memtime stamps per phase
#endif
};

#ifdef SGX_STAMPS
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (lane == 0 && P.stamps) P.stamps[env * 16 + (i)] = t_;                                  \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// per-wave LDS: one game.  NIB_CH = channels of the widest observation rendered from codes by this kernel instantiation
// (0: 'original' channel mode, no code buffer).
template <class G, int NIB_CH = OBS_CH>
struct alignas(16) Lds {
    static constexpr int NIB_BYTES = ((G::RC * NIB_CH + 1) / 2 + 15) & ~15;
    int8_t b[N_LDS_BOARDS][G::S];
    alignas(16) uint8_t nib_lead[16];                  // emit_obs_codes reads up to one halfword in front of the codes (odd boards)
    alignas(16) uint8_t nib[NIB_BYTES + 16];           // 4-bit code of every float of the observation being rendered
    alignas(16) uint32_t mbits[G::MB_WORDS];           // valid-actions mask of the next mover, one BIT per action
    alignas(16) uint8_t cnt[G::CNT_PAD];               // valid moves per perspective cell (also setup-shuffle scratch)
    alignas(16) uint8_t occ[G::S];                     // gen_mask scratch: combined occupancy byte per cell
    alignas(16) typename G::cell_t plist[G::CNT_PAD];  // gen_mask scratch: compacted list of movable cells (also setup-shuffle scratch)
    alignas(16) uint8_t tail[G::TAIL_BYTES];           // record image from ST_OFF on: bitmaps, 32 B scalars, capture-event list
    alignas(16) float unc_val[G::EVL_MAX + G::SPECIAL_EXTRA];         // entries of the observation being rendered whose value has no code:
    alignas(16) typename G::entry_t unc_entry[(G::EVL_MAX + G::SPECIAL_EXTRA + 7) & ~7];   //   the float / the entry index
};

// ---------------------------------------------------------------------------------------------
// Counter RNG of the synthetic-rollout harness; restated in oracle/stratego_oracle.c
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t sm_fin(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__host__ __device__ inline uint64_t sgx_rng(uint64_t seed, uint64_t g, uint64_t j, uint32_t stream, uint32_t t) {
    uint64_t h = sm_fin(seed + 0x9E3779B97F4A7C15ull * (g + 1));
    uint64_t ctr = ((uint64_t)stream << 32) | t;
    return sm_fin(h ^ (j * 0xD1B54A32D192ED03ull + ctr * 0x8CB92BA72F3D8DD7ull + 0x2545F4914F6CDD1Dull));
}
__host__ __device__ inline uint32_t rng_below(uint64_t r, uint32_t n) { return (uint32_t)(((r >> 32) * (uint64_t)n) >> 32); }
enum { STREAM_SETUP = 0, STREAM_ACTION = 1, STREAM_SHUFFLE_P1 = 2, STREAM_SHUFFLE_P2 = 3 };

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
template <class G>
__device__ inline int4 *rec_scal(int8_t *boards, int rec_bytes, int64_t env) {
    return reinterpret_cast<int4 *>(boards + env * (int64_t)rec_bytes + G::SC_OFF);
}

// capacity of a game's capture-event list: the handle's (2 x pieces, at most one per cell), or every (layer, cell) pair in a general-state image
template <class G, class KP>
__device__ inline int max_events_of(const KP &P) { return G::BIG ? (int)G::EVL_MAX : P.max_events; }

// A value that is the same in every lane of a game: an SGPR when the game is the whole wave, left alone otherwise.
template <class G>
__device__ inline int uni(int x) {
    if constexpr (G::LPG == 64) return __builtin_amdgcn_readfirstlane(x);
    else return x;
}
// Ballot over the lanes of this lane's game (bit i = lane i of the game).
template <class G>
__device__ inline unsigned long long gballot(bool pred) {
    const unsigned long long b = __ballot(pred);
    if constexpr (G::LPG == 64) return b;
    else return (b >> (__lane_id() & ~(G::LPG - 1))) & ((1ull << G::LPG) - 1ull);
}

// Cross-lane sums on the DPP path of the VALU (one instruction per step, no LDS round trip; __shfl_* compiles to ds_bpermute).
template <int CTRL, int ROW_MASK = 0xF>
__device__ inline int dpp_or_zero(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, true); }
// inclusive prefix sum over the LPG lanes of this lane's game (a game = one, two or four 16-lane DPP rows)
template <class G>
__device__ inline int gscan_incl(int x) {
    x += dpp_or_zero<0x111>(x);            // row_shr:1
    x += dpp_or_zero<0x112>(x);            // row_shr:2
    x += dpp_or_zero<0x114>(x);            // row_shr:4
    x += dpp_or_zero<0x118>(x);            // row_shr:8
    if constexpr (G::LPG >= 32) x += dpp_or_zero<0x142, 0xA>(x);    // row_bcast:15 -> rows 1 and 3
    if constexpr (G::LPG == 64) x += dpp_or_zero<0x143, 0xC>(x);    // row_bcast:31 -> rows 2 and 3
    return x;
}
// the value of the game's lane `l` (the same l for every lane of the game)
template <class G>
__device__ inline int glane(int x, int l) {
    if constexpr (G::LPG == 64) return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(l));
    else return __shfl(x, l, G::LPG);
}
// sum over the four lanes of a quad (lanes 4i .. 4i+3), in every lane of the quad
__device__ inline int quad_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    return x;
}

// XCD-aware block -> env map: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
// range of envs so neighbouring envs' output lines meet in one L2.
// Unequal shares (round 3).  Under a saturating write stream the odd XCD of every pair (the two XCDs of an IOD) drains its range
// ~20 % slower than the even one (per-XCD TCC_BUSY of one launch with equal eighths: 602 716 | 600 724 | 602 674 | 609 674 k cycles,
// profiles/r03_placement/class_pmc_per_xcd.txt), so with equal eighths the even XCDs idle at the end of every launch.  The host
// gives every XCD its own share of the launch's workgroup-groups: xcd_first / xcd_count (stratego_mi355x.hip: launch_shares -- the even
// XCDs 10 % more, the odd ones 10 % less where the write stream is what bounds the kernel: -3 ... -5 % launch time on 8x8 and 10x10
// boards; equal shares elsewhere, where the same skew costs 3 ... 7 %: tools/skew_ab.py, profiles/r03_skew_ab.log).  The grid is 8 x the
// largest share and surplus workgroups leave at once.  Any shares are a bijection of games onto workgroups, so results cannot depend
// on them.  (Measuring the shares per output buffer from the workgroups' own end times -- an sgx_calibrate_xcd_shares -- was built and
// dropped: it reproduces the odd / even skew (1090 / 900 per mille on every buffer, fast or slow) and nothing more, and on boards bound
// by their game logic the observe launch it measures is balanced differently from the step: Micro +8 ... +12 %, profiles/r03_calib_ab.log.)
template <class KP>
__device__ inline int64_t group_of_block(const KP &P) {
    const int64_t nb = gridDim.x, b = blockIdx.x;
    const int x = (int)(b & 7);                                      // grid is a multiple of 8
    const int64_t i = b >> 3;
    if (P.map_mode == 0) {
        if (i >= P.xcd_count[x]) return (int64_t)1 << 40;             // no work for this workgroup (beyond every env)
        return P.xcd_first[x] + i;
    }
    // experiment modes (SGX_MAP, tools/microbench/map_probe.cpp): how the launch time depends on where the eight XCDs' write fronts
    // are relative to each other
    const int64_t per = nb >> 3;
    const int map_arg = P.map_arg;
    if (P.map_mode == 1) return b;                                                        // linear: one front
    if (P.map_mode == 2) { const int64_t s = map_arg; return (i / s) * (8 * s) + x * s + (i % s); }   // stripes of s workgroups per XCD (s divides nb / 8)
    if (P.map_mode == 3) return x * per + (i + x * (int64_t)map_arg) % per;             // XCD ranges, every front started at another phase
    if (P.map_mode == 4) return x * per + (per - 1 - i);                                  // XCD ranges walked downwards
    if (P.map_mode == 5) return x * per + ((x & 1) ? per - 1 - i : i);                   // neighbouring XCDs walk towards each other
    if (P.map_mode == 6) { const int64_t f = map_arg, sub = per / f; return x * per + (i % f) * sub + (i / f); }   // f sub-fronts per XCD (f divides nb / 8)
    return x * per + i;
}

// Orders the LDS phases of ONE wave (each wave owns its game's LDS region; waves of a workgroup never exchange
// data after the LUT is staged).  DS operations of a wave execute in issue order, so only the compiler has to be
// kept from moving LDS accesses across the phase boundary.
template <class G>
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Observation channel specs: board holding channel `ch` for perspective player index qi, and the LUT index bias
// (recent-moves codes are -3..1).  Partial: impl:1306-1332; full: impl:1200-1227; perspective swap impl:645-675.
// 'extended' kinds are rendered from codes (sgx_obs.h): first channel of each block
struct PartialObs {
    static constexpr int NCH = OBS_CH;
    static constexpr bool CODES = true;
    static constexpr int OWN0 = 0, ENEMY0 = -1 /* no enemy true pieces */, OWN_PO0 = 12, ENEMY_PO0 = 25, OBST = 38, REC0 = 39, CAP0 = 41,
                         STILL0 = 65;
};
struct FullObs {
    static constexpr int NCH = FOBS_CH;
    static constexpr bool CODES = true;
    static constexpr int OWN0 = 0, ENEMY0 = 12, OWN_PO0 = 24, ENEMY_PO0 = 37, OBST = 50, REC0 = 51, CAP0 = 53, STILL0 = 77;
};
// obs_channel_mode='original' (maenv:368-375): channels hold piece VALUES; partial impl:1126-1148, full impl:1048-1070
// ('original' kinds go through (board byte, LUT row); their captured-count channels read the all-zero board and are patched)
struct OrigPartialObs {
    static constexpr int NCH = SGX_PO_OBS_CHANNELS_ORIGINAL;
    static constexpr bool CODES = false;
    static constexpr int REC0 = 4, CAP0 = 6;
    __device__ static inline int board(int ch, int qi) {
        if (ch == 0) return B_PIECES + qi;
        if (ch == 1) return B_PO + qi;
        if (ch == 2) return B_PO + (1 - qi);
        if (ch == 3) return B_OBST;
        if (ch == 4) return B_RECENT + qi;
        if (ch == 5) return B_RECENT + (1 - qi);
        if (ch < 30) return B_ZERO;
        if (ch == 30) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 4 || ch == 5) ? 3 : 0; }
};
struct OrigFullObs {
    static constexpr int NCH = SGX_FO_OBS_CHANNELS_ORIGINAL;
    static constexpr bool CODES = false;
    static constexpr int REC0 = 3, CAP0 = 7;
    __device__ static inline int board(int ch, int qi) {
        if (ch == 0) return B_PIECES + qi;
        if (ch == 1) return B_PIECES + (1 - qi);
        if (ch == 2) return B_OBST;
        if (ch == 3) return B_RECENT + qi;
        if (ch == 4) return B_RECENT + (1 - qi);
        if (ch == 5) return B_PO + qi;
        if (ch == 6) return B_PO + (1 - qi);
        if (ch < 31) return B_ZERO;
        if (ch == 31) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 3 || ch == 4) ? 3 : 0; }
};
// step-kernel observation kind: bit 0 = also render the fully-observable observation, bit 1 = 'original' channels, bit 2 (KIND 4) =
// compact outputs, bit 3 (KIND 8) = NO observation at all (launches without any observation pointer: search expansions, mask-only
// steps): no code buffer and no observation tables in LDS, so the instantiation runs at the full 8 waves per SIMD instead of 6 at 10x10
template <int KIND>
struct ObsKind {
    static constexpr bool FULL = (KIND & 1) != 0, ORIG = (KIND & 2) != 0, NOOBS = (KIND & 8) != 0;
    static_assert(!NOOBS || KIND == 8, "the no-observation kind stands alone");
    static constexpr int NIB_CH = (ORIG || NOOBS) ? 0 : (FULL ? FOBS_CH : OBS_CH);     // code buffer of the game's Lds
    using P = std::conditional_t<ORIG, OrigPartialObs, PartialObs>;
    using F = std::conditional_t<ORIG, OrigFullObs, FullObs>;
};

}  // namespace
