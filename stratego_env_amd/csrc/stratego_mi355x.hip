// stratego_mi355x.hip -- HIP kernels (gfx950 / CDNA4) and C-ABI of the batched Stratego env.
//
// One wavefront (64 lanes) per game; boards of <= 32 cells share a wave between 2 or 4 games (Geo::LPG lanes per game).
// A game's compact state record (4 dense int8 boards + bitmaps + scalars + capture events, whole 128-byte lines) is expanded
// to 32 boards in LDS, the move is applied there, the next mover's valid-actions mask is built in LDS as bits and its
// 67-channel normalised observation is rendered straight into line-aligned 16-byte global stores.
// HBM-bound integer/byte work: no MFMA.  See DESIGN.md for the data layout and byte accounting.
//
// Reference functions reproduced (paths relative to /root/reference/stratego_env):
//   game/stratego_procedural_impl.py  (impl)   stratego_multiagent_env.py (maenv)   game/util.py (util)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "stratego_mi355x.h"

#define SGX_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, const char *detail = "") {
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, detail);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(SGX_EDEVICE, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

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
constexpr int WPB = SGX_WPB;  // waves per workgroup (WPB * Geo::GPW games); they share the LUT
// per observation kind in LDS: the LUT followed by the quad table (2 perspectives x NCH quads x 4 packed entries, see emit_obs)
constexpr int QTAB_DWORDS = 2 * OBS_CH * 4, OBS_TAB_DWORDS = LUT_DWORDS + QTAB_DWORDS;   // partial kind; the full kind follows it
constexpr int FOBS_TAB_DWORDS = LUT_DWORDS + 2 * FOBS_CH * 4;
__host__ __device__ constexpr int lut_row(int ch) { return ((ch >> 1) & 1) * LUT_BLK + (ch >> 2) * LUT_ROW_PITCH + (ch & 1) * 16; }

// internal board indices inside an env record (each board is S bytes, absolute coordinates)
constexpr int B_PIECES = 0;   // +pi : true pieces of player index pi (0 = player +1, 1 = player -1)   impl layers 0/1
constexpr int B_PO = 2;       // +pi : what the opponent knows of pi's pieces                           impl layers 3/4
constexpr int B_STILL = 4;    // +pi : never-moved flags                                                impl layers 32/33
constexpr int B_RECENT = 6;   // +pi : two-square bookkeeping (LDS only, rebuilt from scal)             impl layers 6/7
constexpr int B_CAP = 8;      // +12*pi + (type-1) : captured counts (LDS only, rebuilt from events)    impl layers 8-19 / 20-31
constexpr int N_BOARDS = 32;
constexpr int STORED_BOARDS = 4;  // boards 0..3 live in HBM as bytes; never-moved flags as bitmaps; the rest sparsely:
//   recent moves: at most two non-zero cells per player (impl:1013-1028)  -> two (cell, code) pairs per player in scal
//   captured counts: one event (board - B_CAP, cell) per captured piece   -> uint16 list, <= 2 * pieces per side entries
constexpr int B_OBST = 32;    // LDS only: per-variant obstacle map (impl layer 2)

// record scalars (32 B at SC_OFF): {turn, flags, max_turns, game_no} {n_events, recent pairs of +1, recent pairs of -1, 0}
// a recent pair is cell | (code & 0xFF) << 8, two pairs per int (low / high half); code 0 = empty
constexpr int F_OVER = 1, F_WIN_P1 = 2, F_WIN_M1 = 4, F_END_INVALID = 8, F_PLAYER_M1 = 16;

enum { SP_SPY = 1, SP_SCOUT = 2, SP_MINER = 3, SP_MARSHALL = 10, SP_FLAG = 11, SP_BOMB = 12, SP_UNKNOWN = 13 };

template <int R_, int C_>
struct Geo {
    static constexpr int R = R_, C = C_;
    static constexpr int RC = R * C;
    static constexpr int S = (RC + 3) & ~3;           // board stride (bytes)
    static constexpr int LDS_BOARDS_BYTES = N_BOARDS * S;            // bytes of the 32 LDS boards (multiple of 128)
    // HBM record (a multiple of 128 B, so every record is read and written as whole cache lines):
    //   [0, 4S) four dense boards (true pieces, PO pieces) | zero padding to 16 | ST_OFF: never-moved bitmaps 2 x SB |
    //   SC_OFF: 32 B scalars | EVL_OFF: capture events uint16[max_events] | zero padding
    //   10x10: Barrage 512 B (4 lines), Standard 640 B (5 lines)
    static constexpr int ST_OFF = (STORED_BOARDS * S + 15) & ~15;
    static constexpr int SB = (((RC + 7) / 8) + 15) & ~15;           // bytes of one never-moved bitmap (bit i = cell i)
    static constexpr int SC_OFF = ST_OFF + 2 * SB, EVL_OFF = SC_OFF + 32;
    static constexpr int EVL_MAX = RC;                               // 2 * pieces per side <= cells
    static constexpr int TAIL_BYTES = 2 * SB + 32 + ((2 * EVL_MAX + 15) & ~15);   // LDS image of the record from ST_OFF on
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
    static constexpr int LPG = RC <= 16 ? 16 : (RC <= 32 ? 32 : 64);   // (6x6 with two games per wave measured 5 % slower)
    static constexpr int GPW = 64 / LPG;              // games per wave
    static constexpr int CPL = (RC + LPG - 1) / LPG;  // cells per lane
    static constexpr int CNT_PAD = CPL * LPG;
};

struct DevTables {
    // observation LUTs, rows at lut_row(ch); index = 4 * original + 2 * raw + full:
    //   original: obs_channel_mode 'original' (32/33 channels) instead of 'extended' (67/79)
    //   raw: SGX_STEP_RAW_OBS, un-normalised channel values (penv:157-173 return raw observations)
    //   full: the fully-observable observation instead of the partial one
    float lut[8][LUT_DWORDS];
    uint8_t obstacles[SGX_MAX_CELLS];
};

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

// streaming stores of the big outputs (written once per step, read by a later kernel)
#ifdef SGX_NT_STORES
template <class T> __device__ inline void stream_store(T *p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ inline void stream_store(T *p, T v) { *p = v; }
#endif

// per-wave LDS: one game
template <class G>
struct alignas(16) Lds {
    int8_t b[N_BOARDS + 1][G::S];
    alignas(16) uint32_t mbits[G::MB_WORDS];           // valid-actions mask of the next mover, one BIT per action
    alignas(16) uint8_t cnt[G::CNT_PAD];               // valid moves per perspective cell (also setup-shuffle scratch)
    alignas(16) uint8_t occ[G::S];                     // gen_mask scratch: combined occupancy byte per cell
    alignas(16) uint8_t plist[G::CNT_PAD];             // gen_mask scratch: compacted list of movable cells
    alignas(16) uint8_t tail[G::TAIL_BYTES];           // record image from ST_OFF on: bitmaps, 32 B scalars, capture-event list
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

// XCD-aware block -> env map: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
// range of envs so neighbouring envs' output lines meet in one L2.
__device__ inline int64_t group_of_block() {
    const int64_t nb = gridDim.x, b = blockIdx.x;
    return (b & 7) * (nb >> 3) + (b >> 3);   // grid is a multiple of 8
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
struct PartialObs {
    static constexpr int NCH = OBS_CH;
    __device__ static inline int board(int ch, int qi) {
        if (ch < 12) return B_PIECES + qi;
        if (ch < 25) return B_PO + qi;
        if (ch < 38) return B_PO + (1 - qi);
        if (ch == 38) return B_OBST;
        if (ch == 39) return B_RECENT + qi;
        if (ch == 40) return B_RECENT + (1 - qi);
        if (ch < 53) return B_CAP + 12 * qi + (ch - 41);
        if (ch < 65) return B_CAP + 12 * (1 - qi) + (ch - 53);
        if (ch == 65) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 39 || ch == 40) ? 3 : 0; }
};
struct FullObs {
    static constexpr int NCH = FOBS_CH;
    __device__ static inline int board(int ch, int qi) {
        if (ch < 12) return B_PIECES + qi;
        if (ch < 24) return B_PIECES + (1 - qi);
        if (ch < 37) return B_PO + qi;
        if (ch < 50) return B_PO + (1 - qi);
        if (ch == 50) return B_OBST;
        if (ch == 51) return B_RECENT + qi;
        if (ch == 52) return B_RECENT + (1 - qi);
        if (ch < 65) return B_CAP + 12 * qi + (ch - 53);
        if (ch < 77) return B_CAP + 12 * (1 - qi) + (ch - 65);
        if (ch == 77) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 51 || ch == 52) ? 3 : 0; }
};
// obs_channel_mode='original' (maenv:368-375): channels hold piece VALUES; partial impl:1126-1148, full impl:1048-1070
struct OrigPartialObs {
    static constexpr int NCH = SGX_PO_OBS_CHANNELS_ORIGINAL;
    __device__ static inline int board(int ch, int qi) {
        if (ch == 0) return B_PIECES + qi;
        if (ch == 1) return B_PO + qi;
        if (ch == 2) return B_PO + (1 - qi);
        if (ch == 3) return B_OBST;
        if (ch == 4) return B_RECENT + qi;
        if (ch == 5) return B_RECENT + (1 - qi);
        if (ch < 18) return B_CAP + 12 * qi + (ch - 6);
        if (ch < 30) return B_CAP + 12 * (1 - qi) + (ch - 18);
        if (ch == 30) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 4 || ch == 5) ? 3 : 0; }
};
struct OrigFullObs {
    static constexpr int NCH = SGX_FO_OBS_CHANNELS_ORIGINAL;
    __device__ static inline int board(int ch, int qi) {
        if (ch == 0) return B_PIECES + qi;
        if (ch == 1) return B_PIECES + (1 - qi);
        if (ch == 2) return B_OBST;
        if (ch == 3) return B_RECENT + qi;
        if (ch == 4) return B_RECENT + (1 - qi);
        if (ch == 5) return B_PO + qi;
        if (ch == 6) return B_PO + (1 - qi);
        if (ch < 19) return B_CAP + 12 * qi + (ch - 7);
        if (ch < 31) return B_CAP + 12 * (1 - qi) + (ch - 19);
        if (ch == 31) return B_STILL + qi;
        return B_STILL + (1 - qi);
    }
    __device__ static inline int bias(int ch) { return (ch == 3 || ch == 4) ? 3 : 0; }
};
// step-kernel observation kind: bit 0 = also render the fully-observable observation, bit 1 = 'original' channels
template <int KIND>
struct ObsKind {
    static constexpr bool FULL = (KIND & 1) != 0, ORIG = (KIND & 2) != 0;
    using P = std::conditional_t<ORIG, OrigPartialObs, PartialObs>;
    using F = std::conditional_t<ORIG, OrigFullObs, FullObs>;
};

// ---------------------------------------------------------------------------------------------
// Observation render: float32 [R][C][NCH], perspective of player index qi
// (partial: impl:1335-1397, full: impl:1230-1303; normalisation maenv:499-508 through the LUT)
// ---------------------------------------------------------------------------------------------
#ifndef SGX_OBS_UNROLL
#define SGX_OBS_UNROLL 4
#endif
// quad table entry of (perspective qi, quad qd, element j): LDS byte offset of the source board at the first 4-cell group
// (low 16 bits) and LUT index base (high 16 bits); built once per workgroup (build_quad_table)
template <class G, class Spec>
__device__ inline void build_quad_table(uint32_t *qtab, int tid, int nthreads) {
    constexpr int RC = G::RC, S = G::S, NCH = Spec::NCH;
    if constexpr (RC % 4 != 0) {
        // odd cell counts (5x5, 15x15): per-channel table instead -- entry (qi, ch) = LDS byte offset of the source board (low 16
        // bits) and LUT index base (high 16 bits); emit_obs adds the cell
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

// `tab` = this observation kind's LUT followed by its quad table.
// The observation is written in 1 KiB chunks aligned to 1 KiB ADDRESS boundaries (whole 128-byte lines per store
// instruction).  Chunking by 4-cell group instead (64 of a group's 67 quads per store, every store 48 bytes further off a
// line) left two partial lines per store and ran 1.5x slower in the store-pattern probe (tools/microbench/aligned_alloc.hip:
// 490 vs 333 us).  With address-aligned chunks a lane's quad changes every iteration, hence the quad table.
template <class G, class Spec>
__device__ void emit_obs(const Lds<G> &L, const float *tab, int qi, float *__restrict__ dst, int lane) {
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
            if (in) stream_store(&base[q], o);
        }
    } else {
        // odd cell counts (5x5, 15x15): an env's observation is only 4-byte aligned.  Lanes own the 16-byte slots of the
        // ADDRESS range (sweep started on a chunk boundary like above); a slot's four floats are looked up one by one
        // through the per-channel table, whole slots leave as one 16-byte store, the partial first / last slot as dwords.
        // (One dword per lane per store, the first version, reached 2.3 TB/s on 15x15.)
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
                stream_store(&reinterpret_cast<f32x4 *>(base)[k], q);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (in[j]) base[4 * k + j] = o[j];
            }
        }
    }
}

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
template <class G>
__device__ __forceinline__ void mask_noop_only(Lds<G> &L, int lane) {
    const int4 z = make_int4(0, 0, 0, 0);
    for (int i = lane; i < G::MB_WORDS / 4; i += G::LPG) reinterpret_cast<int4 *>(L.mbits)[i] = z;
    for (int i = lane; i < G::CNT_PAD / 4; i += G::LPG) reinterpret_cast<int *>(L.cnt)[i] = 0;
    wave_sync<G>();
    if (lane == 0) { L.mbits[(G::K - 1) >> 5] = 1u << ((G::K - 1) & 31); L.cnt[0] = 1; }
    wave_sync<G>();
}

template <class G>
__device__ SGX_GENMASK_INLINE int gen_mask(Lds<G> &L, int qi, bool game_over, int lane) {
    constexpr int R = G::R, C = G::C, RC = G::RC, K = G::K;
    constexpr int OCC_OWN = 1, OCC_ENEMY = 2, OCC_OBST = 4, OCC_CAME_FROM = 8;
    const int8_t *own = L.b[B_PIECES + qi], *enemy = L.b[B_PIECES + 1 - qi], *rec = L.b[B_RECENT + qi], *obst = L.b[B_OBST];
    {
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
            if (movable) L.plist[npieces + __popcll(bm & ((1ull << lane) - 1ull))] = (uint8_t)i;
            npieces += __popcll(bm);
        }
        wave_sync<G>();
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
                if (!__any(k <= lim)) break;
                if (k <= lim) {
                    e += delta;
                    const int v = L.occ[e];
                    if (v & (OCC_OWN | OCC_OBST)) {
                        lim = 0;                                                  // blocked: the ray stops
                    } else {
                        // two-square veto: this cell is skipped but the ray goes on (impl:439-445)
                        if (!(pinned && (v & OCC_CAME_FROM) && !(v & OCC_ENEMY))) {
                            const int bit = bit0 + k;
                            atomicOr(&L.mbits[bit >> 5], 1u << (bit & 31));
                            ++n;
                        }
                        if (v & OCC_ENEMY) lim = 0;                               // an attacked piece ends the ray
                    }
                }
            }
            n += __shfl_xor(n, 1);
            n += __shfl_xor(n, 2);                                                // moves of the piece = its 4 rays
            if (act && d == 0) { L.cnt[pcell] = (uint8_t)n; mine += n; }
        }
#pragma unroll
        for (int o = G::LPG / 2; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        total = uni<G>(mine);
    }
    if (total == 0 && lane == 0) {
        L.mbits[(K - 1) >> 5] = 1u << ((K - 1) & 31);  // valid_moves_mask[0, 0, -1] (impl:514-515); mbits was just zeroed
        L.cnt[0] = 1;
    }
    wave_sync<G>();
    return total;
}

// 4 mask bits -> 4 mask bytes
__device__ inline uint32_t expand4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }
// `n` (<= 32) mask bits starting at bit position p
template <class G>
__device__ inline uint32_t mask_bits(const Lds<G> &L, int p, int n) {
    const unsigned long long w = (unsigned long long)L.mbits[p >> 5] | ((unsigned long long)L.mbits[(p >> 5) + 1] << 32);
    return (uint32_t)(w >> (p & 31)) & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
}

// LDS mask bits -> global uint8 [R,C,K].  An env's mask starts at env*NA bytes: 4-byte but not 16-byte aligned at 10x10
// (3700 = 4 mod 16), byte aligned when NA is odd (5x5, 15x15); lanes own 16-byte-aligned chunks of the global range, so
// whole chunks leave as one 16-byte store per lane and only the partial first / last chunks go out as dwords (bytes when
// the base is not 4-byte aligned).
template <class G>
__device__ void emit_mask(const Lds<G> &L, uint8_t *__restrict__ dst, int lane) {
    if constexpr (G::NA % 4 != 0 && G::NA < 2048) {      // small byte-aligned masks (5x5: 425 bytes): plain byte stores measured faster
        for (int i = lane; i < G::NA; i += G::LPG) dst[i] = (uint8_t)((L.mbits[i >> 5] >> (i & 31)) & 1u);
        return;
    }
    const int A = (int)(reinterpret_cast<uintptr_t>(dst) & 15);
    const int nchunks = (A + G::NA + 15) >> 4;
    uint8_t *gbase = dst - A;                       // 16-byte aligned
    const int shift = (int)((reinterpret_cast<uintptr_t>(gbase) >> 4) & (G::LPG - 1));   // start the sweep on a 1 KiB boundary
    for (int c0 = -shift; c0 < nchunks; c0 += G::LPG) {
        const int c = c0 + lane;
        if (c < 0 || c >= nchunks) continue;
        const int lo = 16 * c - A;                  // first mask byte of this chunk
        if (lo >= 0 && lo + 16 <= G::NA) {
            const uint32_t b16 = mask_bits(L, lo, 16);
            i32x4 q4 = {(int)expand4(b16 & 15), (int)expand4((b16 >> 4) & 15), (int)expand4((b16 >> 8) & 15), (int)expand4(b16 >> 12)};
            stream_store(&reinterpret_cast<i32x4 *>(gbase)[c], q4);
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
template <class G, class F>
__device__ void emit_mask_mapped(const Lds<G> &L, uint8_t *__restrict__ dst, int n_bytes, F src, int lane) {
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
            stream_store(&reinterpret_cast<i32x4 *>(gbase)[c], q4);
        } else {
#pragma unroll                                   // (a rolled loop would index w[] dynamically: scratch memory for the whole kernel)
            for (int j = 0; j < 16; ++j)
                if (lo + j >= 0 && lo + j < n_bytes) dst[lo + j] = (uint8_t)((w[j >> 2] >> (8 * (j & 3))) & 1u);
        }
    }
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
template <class G>
__device__ int kth_valid(const Lds<G> &L, int k, int lane) {
    constexpr int K = G::K;
    int cell = 0, before = 0, run = 0;
    bool found = false;
#pragma unroll
    for (int cc = 0; cc < G::CPL; ++cc) {
        const int c0 = L.cnt[lane + G::LPG * cc];
        int incl = c0;
#pragma unroll
        for (int o = 1; o < G::LPG; o <<= 1) {
            const int v = __shfl_up(incl, o, G::LPG);
            if (lane >= o) incl += v;
        }
        const unsigned long long hit = gballot<G>(run + incl > k);
        if (!found && hit) {
            const int l = __ffsll((long long)hit) - 1;
            cell = G::LPG * cc + l;
            before = run + __shfl(incl - c0, l, G::LPG);
            found = true;
        }
        run += __shfl(incl, G::LPG - 1, G::LPG);
    }
    cell = uni<G>(cell);
    int kk = uni<G>(k - before);
    const int p = cell * K + (lane < K ? lane : 0);
    unsigned long long bits = gballot<G>(lane < K && ((L.mbits[p >> 5] >> (p & 31)) & 1u) != 0);
    for (int i = 0; i < kk; ++i) bits &= bits - 1;
    const int ch = __ffsll((long long)bits) - 1;
    return cell * K + ch;
}

// ---------------------------------------------------------------------------------------------
// Fresh game into LDS boards: _create_initial_state (impl:211-249) from own-side maps (explicit, from the
// human-setup table: util:241-275 net effect, or random back-row placement: util:13-30)
// ---------------------------------------------------------------------------------------------
template <class G>
__device__ void clear_boards(Lds<G> &L, int lane) {
    const int4 z = make_int4(0, 0, 0, 0);
    for (int i = lane; i < G::LDS_BOARDS_BYTES / 16; i += G::LPG) reinterpret_cast<int4 *>(&L.b[0][0])[i] = z;
}

// place code `t` of player index pi at absolute cell
template <class G>
__device__ inline void place(Lds<G> &L, int pi, int cell, int t) {
    L.b[B_PIECES + pi][cell] = (int8_t)t;
    L.b[B_PO + pi][cell] = t ? SP_UNKNOWN : 0;
    L.b[B_STILL + pi][cell] = t ? 1 : 0;
}

template <class G>
__device__ void sample_boards(Lds<G> &L, const KParams &P, uint64_t g, uint64_t j, int lane) {
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
        // lane 1 player -1, each in its own half of the scratch (2n <= cells <= CNT_PAD)
        const int pl = lane;
        uint8_t *loc = L.cnt + pl * n;
        for (int i = 0; i < n; ++i) loc[i] = (uint8_t)i;
        for (int i = n - 1; i > 0; --i) {
            const uint32_t k = rng_below(sgx_rng(P.seed, g, j, pl ? STREAM_SHUFFLE_P2 : STREAM_SHUFFLE_P1, (uint32_t)i), (uint32_t)(i + 1));
            const uint8_t t = loc[i]; loc[i] = loc[k]; loc[k] = t;
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

// ---------------------------------------------------------------------------------------------
// The step kernel: env.step() of N games (maenv:659-828), one wave per game
// ---------------------------------------------------------------------------------------------
// waves per SIMD this geometry can reach: LDS per workgroup = WPB game regions + the shared LUT, 160 KiB per CU
template <class G, int KIND>
constexpr int waves_per_simd() {
    constexpr bool FULL = (KIND & 1) != 0;
    constexpr int per_wg = WPB * G::GPW * (int)sizeof(Lds<G>) + 4 * (OBS_TAB_DWORDS + (FULL ? FOBS_TAB_DWORDS : 0)) + SGX_MAX_CELLS;
    constexpr int wgs = (160 * 1024) / per_wg;
    constexpr int w = wgs * WPB / 4;
    // toy boards are latency-bound (tiny per-game work): 8 waves/SIMD measured +7 %; on 10x10 forcing 64 VGPRs spills
    constexpr int want = G::RC <= 64 ? 8 : SGX_MIN_WAVES;
    return w > want ? want : (w < 1 ? 1 : w);
}

// Builds the record image from ST_OFF on in L.tail (never-moved bitmaps from the LDS still boards, the two scalar int4s,
// the event list already kept in L.tail) and writes the whole record to HBM as 16-byte stores over whole 128-byte lines.
template <class G>
__device__ inline void write_record(Lds<G> &L, int8_t *rec_g, int rec_bytes, int4 sc0, int4 sc1, int n_events, int lane) {
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
    wave_sync<G>();
    const int4 *bsrc = reinterpret_cast<const int4 *>(&L.b[0][0]), *tsrc = reinterpret_cast<const int4 *>(L.tail);
    int4 *dst = reinterpret_cast<int4 *>(rec_g);
    const int n_tail_q = (2 * G::SB + 32 + 2 * n_events + 15) >> 4;            // tail int4s that carry data
    for (int i = lane; i < rec_bytes / 16; i += G::LPG) {
        int4 v = make_int4(0, 0, 0, 0);
        if (i < G::ST_OFF / 16) {
            v = bsrc[i];
            const int keep = STORED_BOARDS * S - 16 * i;                       // bytes of this int4 that belong to the stored boards
            if (keep < 16) { if (keep <= 12) v.w = 0; if (keep <= 8) v.z = 0; if (keep <= 4) v.y = 0; if (keep <= 0) v.x = 0; }
        } else if (i < G::ST_OFF / 16 + n_tail_q) {
            v = tsrc[i - G::ST_OFF / 16];
        }
        dst[i] = v;
    }
}

// One game's env.step() by one wave (called with the wave's private LDS region).
template <int R_, int C_, int KIND, bool MAPPED>
__device__ __forceinline__ void env_step(const KParams &P, Lds<Geo<R_, C_>> &L, const float *lut_s, const uint8_t *obst_s, const int64_t env,
                                         const int lane) {
    using G = Geo<R_, C_>;
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr int R = G::R, C = G::C, RC = G::RC, S = G::S, K = G::K, NA = G::NA, MPA = G::MPA, AS = G::AS;
    STAMP(0);

    int8_t *rec_g = P.boards + env * (int64_t)P.rec_bytes;
    // ---- stage.  Every global read of the step is issued up front -- the whole record (a few 128-byte lines) as one or
    //      two int4 per lane, and the action -- so the wave pays ONE memory round trip.  The scalars, never-moved bitmaps
    //      and capture events are then read from the LDS image of the record (L.tail has the record's layout from ST_OFF
    //      on).  They used to be five dependent loads: a quarter of a toy game's lifetime.
    constexpr int Q_BOARDS = G::ST_OFF / 16, Q_REC = Q_BOARDS + G::TAIL_BYTES / 16, NLOAD = (Q_REC + G::LPG - 1) / G::LPG;
    static_assert(G::TAIL_BYTES % 16 == 0 && NLOAD <= 2, "record image must fit two int4 per lane");
    const int4 zero4 = make_int4(0, 0, 0, 0);
    int4 rq0 = zero4, rq1 = zero4;
    {
        const int4 *src = reinterpret_cast<const int4 *>(rec_g);
        const int nq = min(P.rec_bytes >> 4, Q_REC);
        if (lane < nq) rq0 = src[lane];
        if constexpr (NLOAD > 1)
            if (lane + G::LPG < nq) rq1 = src[lane + G::LPG];
    }
    int a_raw = 0;
    int4 pos_raw = zero4;
    if (P.mode == 0) {
        if (P.io.flags & SGX_STEP_ACTIONS_POSITIONS) pos_raw = reinterpret_cast<const int4 *>(P.io.actions_dev)[env];
        else a_raw = P.io.actions_dev[env];
    }
    {   // while the loads are in flight: clear the 28 rebuilt boards, copy the obstacle map (shared per workgroup)
        int4 *dst = reinterpret_cast<int4 *>(&L.b[0][0]);
        for (int i = Q_BOARDS + lane; i < G::LDS_BOARDS_BYTES / 16; i += G::LPG) dst[i] = zero4;
        for (int i = lane; i < S / 4; i += G::LPG) reinterpret_cast<int *>(L.b[B_OBST])[i] = reinterpret_cast<const int *>(obst_s)[i];
        int4 *tl = reinterpret_cast<int4 *>(L.tail);
        if (lane < Q_BOARDS) dst[lane] = rq0;
        else if (lane < Q_REC) tl[lane - Q_BOARDS] = rq0;
        if constexpr (NLOAD > 1) {
            if (lane + G::LPG < Q_BOARDS) dst[lane + G::LPG] = rq1;
            else if (lane + G::LPG < Q_REC) tl[lane + G::LPG - Q_BOARDS] = rq1;
        }
    }
    wave_sync<G>();
    const int4 sc = reinterpret_cast<const int4 *>(L.tail + 2 * G::SB)[0], sc2 = reinterpret_cast<const int4 *>(L.tail + 2 * G::SB)[1];
    int turn = uni<G>(sc.x), flags = uni<G>(sc.y), game_no = uni<G>(sc.w);
    const int max_turns = uni<G>(sc.z);
    int n_events = min(uni<G>(sc2.x), (int)G::EVL_MAX);
    int rp0 = uni<G>(sc2.y), rp1 = uni<G>(sc2.z);   // recent-move pairs of player +1 / -1 (two named scalars: a runtime-indexed
                                              // array would live in scratch memory)
    {   // ---- rebuild the 28 derived boards: never-moved bitmaps, recent-move pairs, capture events
        const uint32_t *stb = reinterpret_cast<const uint32_t *>(L.tail);
#pragma unroll
        for (int cc = 0; cc < G::CPL; ++cc) {
            const int i = lane + G::LPG * cc;
            if (i < RC) {
                L.b[B_STILL][i] = (int8_t)((stb[i >> 5] >> (i & 31)) & 1u);
                L.b[B_STILL + 1][i] = (int8_t)((stb[G::SB / 4 + (i >> 5)] >> (i & 31)) & 1u);
            }
        }
        const uint16_t *evl = reinterpret_cast<const uint16_t *>(L.tail + 2 * G::SB + 32);   // stays here for the write-back
        for (int i = lane; i < n_events; i += G::LPG) {
            const int evt = evl[i], byte = (B_CAP + (evt >> 8)) * S + (evt & 0xFF);  // event = (board - B_CAP) << 8 | cell
            atomicAdd(reinterpret_cast<unsigned int *>(&L.b[0][0]) + (byte >> 2), 1u << (8 * (byte & 3)));
        }
        if (lane < 4) {
            const int pr = (((lane >> 1) ? rp1 : rp0) >> (16 * (lane & 1))) & 0xFFFF;
            if (pr >> 8) L.b[B_RECENT + (lane >> 1)][pr & 0xFF] = (int8_t)(pr >> 8);
        }
    }
    const float *lut = lut_s;
    wave_sync<G>();
    STAMP(1);   // state staged

    int player = (flags & F_PLAYER_M1) ? -1 : 1;
    bool over = (flags & F_OVER) != 0;
    bool applied = false, invalid_action = false, noop_path = false;
    int mover = player;
    int dirty_s = -1, dirty_e = -1, dirty_cap_a = -1, dirty_cap_b = -1;   // cells / boards touched by the move

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
            int idx = (sr * C + sc_) * MPA + ((er != sr) ? er : R + ec);                     // impl:268-277
            if (player == -1 && idx != AS - 1) {                                             // impl:698-720
                const int q = fdiv_(idx, MPA), off = fmod_(idx, MPA);
                int r0 = fdiv_(q, C), c0 = fmod_(q, C), r1, c1;
                if (off >= R) { c1 = off - R; r1 = r0; } else { r1 = off; c1 = c0; }
                r0 = R - 1 - r0; r1 = R - 1 - r1; c0 = C - 1 - c0; c1 = C - 1 - c1;
                idx = (r0 * C + c0) * MPA + ((r1 != r0) ? r1 : R + c1);
            }
            if (idx == AS - 1) {
                noop_path = true;                                                            // impl:809-814
            } else {                                                                         // impl:369-383
                const int q = fdiv_(idx, MPA), off = fmod_(idx, MPA);
                sr = fdiv_(q, C); sc_ = fmod_(q, C);
                if (off >= R) { ec = off - R; er = sr; } else { er = off; ec = sc_; }
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
                const int nmoves = gen_mask(L, pi, false, lane);
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
                    const int k = lane + 1;  // lanes 0.. check the intermediate cells
                    bool blk = false;
                    if (k < dist) { const int m = s + k * stepc; blk = own[m] != 0 || enemy[m] != 0 || obst[m] != 0; }
                    if (gballot<G>(blk) != 0ull) valid = false;
                } else if (dist > 1) valid = false;
            }
            if (valid) {
                // ---- _get_next_state (impl:905-1028)
                const int moved = t;
                turn += 1;
                bool wins = false, tied = false;
                if (dest != 0) {
                    if (moved == SP_MINER && dest == SP_BOMB) wins = true;
                    else if (moved == SP_SPY && dest == SP_MARSHALL) wins = true;
                    else if (dest == SP_FLAG) { wins = true; over = true; flags |= F_OVER | (player == 1 ? F_WIN_P1 : F_WIN_M1); }
                    else if (dest != SP_BOMB) { if (moved == dest) tied = true; else if (moved > dest) wins = true; }
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
                        if (!wins) L.b[B_CAP + 12 * pi + moved - 1][e] += 1;               // impl:1001-1004
                        if (wins || tied) L.b[B_CAP + 12 * (1 - pi) + dest - 1][e] += 1;   // impl:1006-1009
                    }
                }
                dirty_s = s; dirty_e = e;
                if (dest != 0) {
                    if (!wins) dirty_cap_a = B_CAP + 12 * pi + moved - 1;
                    if (wins || tied) dirty_cap_b = B_CAP + 12 * (1 - pi) + dest - 1;
                    if (pi) rp1 = 0; else rp0 = 0;                                       // an attack wipes the mover's layer
                } else {
                    const int code = old_end == 1 ? (old_start == -2 ? -3 : -2) : -1;
                    const int pr = (s | (1 << 8)) | ((e | ((code & 0xFF) << 8)) << 16);
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
    int nvalid = gen_mask(L, qi, over, lane);
    bool ended_now = false;
    if (applied && !noop_path) {
        const bool was_over = over;
        if (nvalid == 0) { over = true; flags = (flags & ~(F_WIN_P1 | F_WIN_M1)) | F_OVER | (mover == 1 ? F_WIN_P1 : F_WIN_M1); }
        if (turn >= max_turns && !over) { over = true; flags |= F_OVER | F_END_INVALID; }
        if (over && !was_over && nvalid != 0) { mask_noop_only(L, lane); nvalid = 0; }  // finished: the no-op only
        ended_now = over;
    } else if (applied && noop_path) {
        ended_now = over;
        if (nvalid != 0) { mask_noop_only(L, lane); nvalid = 0; }
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
        if (lane < 2 && P.io.reward_dev) P.io.reward_dev[2 * env + lane] = lane ? rew_m1 : rew_p1;
        uint8_t *fp = lane == 0 ? P.io.done_dev : lane == 1 ? P.io.invalid_action_dev : lane == 2 ? P.io.ending_invalid_dev : nullptr;
        const uint8_t fv = lane == 0 ? (done ? 1 : 0) : lane == 1 ? (invalid_action ? 1 : 0) : (end_invalid ? 1 : 0);
        if (fp) fp[env] = fv;
    }

    // ---- terminal observations of both players (maenv:772-773)
    if (P.mode == 0 && ended_now && P.io.final_obs_dev) {
        float *fo = P.io.final_obs_dev + env * (int64_t)(2 * RC * PS::NCH);
        emit_obs<G, PS>(L, lut, 0, fo, lane);
        emit_obs<G, PS>(L, lut, 1, fo + RC * PS::NCH, lane);
    }
    if constexpr (ObsKind<KIND>::FULL)
        if (P.mode == 0 && ended_now && P.io.final_fobs_dev) {
            float *fo = P.io.final_fobs_dev + env * (int64_t)(2 * RC * FS::NCH);
            emit_obs<G, FS>(L, lut + OBS_TAB_DWORDS, 0, fo, lane);
            emit_obs<G, FS>(L, lut + OBS_TAB_DWORDS, 1, fo + RC * FS::NCH, lane);
        }

    // ---- auto-reset: the finished env starts its next game now
    bool wrote_reset = false;
    if (P.mode == 0 && ended_now && P.io.auto_reset) {
        game_no += 1;
        sample_boards(L, P, (uint64_t)(P.env_id_offset + env), (uint64_t)game_no, lane);
        turn = 0; flags = 0; player = 1; qi = 0; over = false;
        n_events = 0; rp0 = rp1 = 0;
        nvalid = gen_mask(L, 0, false, lane);
        wrote_reset = true;
    }

    STAMP(4);   // results / terminal handling done
    // ---- outputs for the next mover
    if (lane == 0 && P.io.player_dev) P.io.player_dev[env] = (int8_t)player;
    if (P.io.mask_dev) {
        // MAPPED: the separate instantiation behind SGX_STEP_MASK_1D / SGX_STEP_MASK_STATE_COORDS (kept out of the hot kernel: its
        // 16 index computations per lane cost 30 VGPRs)
        if (MAPPED && (P.io.flags & SGX_STEP_MASK_1D)) emit_mask_mapped(L, P.io.mask_dev + env * (int64_t)AS, AS, Src1D<G>{qi}, lane);
        else if (MAPPED && qi) emit_mask_mapped(L, P.io.mask_dev + env * (int64_t)NA, NA, SrcSpatialFlipped<G>{}, lane);
        else emit_mask(L, P.io.mask_dev + env * (int64_t)NA, lane);
    }
    STAMP(5);   // mask stores issued
    // (rendering the observation before the mask, so that its stores drain during mask generation, measured 6 % slower)
    if (P.io.obs_dev) emit_obs<G, PS>(L, lut, qi, P.io.obs_dev + env * (int64_t)(RC * PS::NCH), lane);
    if constexpr (ObsKind<KIND>::FULL)
        if (P.io.fobs_dev) emit_obs<G, FS>(L, lut + OBS_TAB_DWORDS, qi, P.io.fobs_dev + env * (int64_t)(RC * FS::NCH), lane);
    STAMP(6);   // obs stores issued
    if (P.mode == 0 && P.io.next_actions_dev) {
        const int total = nvalid == 0 ? 1 : nvalid;
        const uint32_t k = rng_below(sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)game_no, STREAM_ACTION, (uint32_t)turn), (uint32_t)total);
        const int na = kth_valid(L, (int)k, lane);
        if (lane == 0) P.io.next_actions_dev[env] = na;
    }

    STAMP(7);   // next action sampled
    // ---- write the record back as whole 128-byte lines: dense boards, scalars, capture events.  (Scattered stores
    //      of only the <= 9 touched bytes + 32 B of scalars are partial-line writes: measured 7 % slower.)
    if (applied || wrote_reset) {
        uint16_t *evl = reinterpret_cast<uint16_t *>(L.tail + 2 * G::SB + 32);
        if (!wrote_reset && dirty_s >= 0) {
            const int na = dirty_cap_a >= 0 ? 1 : 0, nb = dirty_cap_b >= 0 ? 1 : 0;
            if (lane == 0 && na && n_events < P.max_events) evl[n_events] = (uint16_t)(((dirty_cap_a - B_CAP) << 8) | dirty_e);
            if (lane == 1 && nb && n_events + na < P.max_events) evl[n_events + na] = (uint16_t)(((dirty_cap_b - B_CAP) << 8) | dirty_e);
            n_events = min(n_events + na + nb, P.max_events);
        }
        write_record(L, rec_g, P.rec_bytes, make_int4(turn, flags, max_turns, game_no), make_int4(n_events, rp0, rp1, 0), n_events, lane);
    }
    STAMP(8);   // write-back issued
#ifdef SGX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(9);   // all stores acknowledged
#endif
}

// KIND bit 0: also renders the fully-observable observation (BOTH_OBSERVATIONS / FULLY_OBSERVABLE modes, maenv:477-492);
// KIND bit 1: obs_channel_mode 'original' (32/33 value channels) instead of 'extended' (67/79 one-hot channels)
template <int R_, int C_, int KIND, bool MAPPED>
__device__ __forceinline__ void game_kernel_body(const KParams &P) {
    using G = Geo<R_, C_>;
    using PS = typename ObsKind<KIND>::P;
    using FS = typename ObsKind<KIND>::F;
    constexpr bool FULL = ObsKind<KIND>::FULL;
    constexpr int ORIG4 = ObsKind<KIND>::ORIG ? 4 : 0;
    __shared__ Lds<G> LW[WPB * G::GPW];
    __shared__ alignas(16) float lut_s[OBS_TAB_DWORDS + (FULL ? FOBS_TAB_DWORDS : 0)];
    __shared__ alignas(16) uint8_t obst_s[SGX_MAX_CELLS];
    const int lane = threadIdx.x & (G::LPG - 1), slot = threadIdx.x / G::LPG;     // lane inside the game, game inside the workgroup
    const int64_t env = group_of_block() * (WPB * G::GPW) + slot;

    // ---- the workgroup's shared normalisation LUT (L2-resident source)
    const bool raw = (P.io.flags & SGX_STEP_RAW_OBS) != 0;
    const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[ORIG4 + (raw ? 2 : 0)]);
    for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * WPB) reinterpret_cast<f32x4 *>(lut_s)[i] = lsrc[i];
    build_quad_table<G, PS>(reinterpret_cast<uint32_t *>(lut_s + LUT_DWORDS), threadIdx.x, 64 * WPB);
    if constexpr (FULL) {
        const f32x4 *fsrc = reinterpret_cast<const f32x4 *>(P.tab->lut[ORIG4 + (raw ? 2 : 0) + 1]);
        for (int i = threadIdx.x; i < LUT_DWORDS / 4; i += 64 * WPB) reinterpret_cast<f32x4 *>(lut_s + OBS_TAB_DWORDS)[i] = fsrc[i];
        build_quad_table<G, FS>(reinterpret_cast<uint32_t *>(lut_s + OBS_TAB_DWORDS + LUT_DWORDS), threadIdx.x, 64 * WPB);
    }
    for (int i = threadIdx.x; i < G::S / 4; i += 64 * WPB) reinterpret_cast<int *>(obst_s)[i] = reinterpret_cast<const int *>(P.tab->obstacles)[i];
    __syncthreads();   // from here on every wave works on its own game
    if (env < P.n_envs) env_step<R_, C_, KIND, MAPPED>(P, LW[slot], lut_s, obst_s, env, lane);
}

// sgx_step and sgx_observe run the same body (P.mode tells them apart at run time: specialising the body on the mode changed the
// step kernel's schedule and cost 3.7 % on Barrage); two kernel symbols, so that a kernel trace keeps the env.step() launches
// apart from the state-preserving observe launches (placement trials, reset())
template <int R_, int C_, int KIND, bool MAPPED = false>
__global__ __launch_bounds__(64 * WPB, (waves_per_simd<Geo<R_, C_>, KIND>())) void step_kernel(const KParams P) {
    game_kernel_body<R_, C_, KIND, MAPPED>(P);
}
template <int R_, int C_, int KIND, bool MAPPED = false>
__global__ __launch_bounds__(64 * WPB, (waves_per_simd<Geo<R_, C_>, KIND>())) void observe_kernel(const KParams P) {
    game_kernel_body<R_, C_, KIND, MAPPED>(P);
}

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
            n = __popc(mask_bits(L, cell * K, K < 32 ? K : 32));
            if constexpr (K > 32) n += __popc(mask_bits(L, cell * K + 32, K - 32));
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
        for (int i = 0; i < sc2.x; ++i) o[(8 + (ev[i] >> 8)) * RC + (ev[i] & 0xFF)] += 1;
        if (player_out) player_out[env] = (sc.y & F_PLAYER_M1) ? -1 : 1;
    }
}

// Reachable states only: at most two non-zero recent-move cells per player (impl:1013-1028) and at most
// max_events captured pieces; anything beyond that cannot come from play and is dropped.
// One 256-thread block per state: a single coalesced pass over the 34 int64 layers scatters them into LDS (dense boards
// straight into the record image, never-moved flags, recent-move codes and captured counts as bytes), the capture-event list
// is laid out with a block-wide prefix sum, and the finished record leaves as whole 128-byte lines.  (The first version
// walked the 24 captured layers with one thread: 4.6 ms per 65,536 states against 0.44 ms for the export.)
template <int R_, int C_>
__global__ __launch_bounds__(256) void import_kernel(const KParams P, const int64_t *__restrict__ in, const int8_t *__restrict__ player_in) {
    using G = Geo<R_, C_>;
    constexpr int RC = G::RC, C = G::C, S = G::S, NT = 256;
    constexpr int IMG = (G::EVL_OFF + 2 * G::EVL_MAX + 127) & ~127;      // >= rec_bytes of any piece set on this board
    constexpr int NE = 24 * RC, PER = (NE + NT - 1) / NT;
    __shared__ alignas(16) uint8_t img[IMG];
    __shared__ uint8_t cap[NE];
    __shared__ int8_t recent[2 * RC];
    __shared__ uint8_t still[2 * RC];
    __shared__ int scan[NT / 64];
    const int tid = threadIdx.x;
    const int64_t env = blockIdx.x;
    if (env >= P.n_envs) return;
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
        if (l == 2 || l == 5) continue;                               // obstacles are the variant's; scalars below
        const int64_t raw = rawv[k];
        if (l < 2 || l == 3 || l == 4) {                              // legal range of the layer; anything else -> 0
            const int b = l < 2 ? B_PIECES + l : B_PO + (l - 3), hi = l < 2 ? SP_BOMB : SP_UNKNOWN;
            img[b * S + cell] = (uint8_t)((raw >= 0 && raw <= hi) ? (int)raw : 0);
        } else if (l == 6 || l == 7) recent[(l - 6) * RC + cell] = (int8_t)((raw >= -3 && raw <= 1) ? (int)raw : 0);
        else if (l < 32) cap[(l - 8) * RC + cell] = (uint8_t)(raw <= 0 ? 0 : (raw > 12 ? 12 : (int)raw));
        else still[(l - 32) * RC + cell] = raw == 1 ? 1 : 0;
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
    // capture events in (layer, cell) order: thread t owns entries [t*PER, (t+1)*PER) of the count table
    int cnt = 0;
    for (int k = 0; k < PER; ++k) {
        const int e = tid * PER + k;
        if (e < NE) cnt += cap[e];
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
            if (e < NE)
                for (int q = cap[e]; q > 0; --q, ++at)
                    if (at < P.max_events) ev[at] = (uint16_t)(((e / RC) << 8) | (e % RC));
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
        for (int pl = 0; pl < 2; ++pl) {
            int k = 0;
            for (int cell = 0; cell < RC && k < 2; ++cell) {
                const int code = recent[pl * RC + cell];
                if (code != 0) { pairs[pl] |= (cell | ((code & 0xFF) << 8)) << (16 * k); ++k; }
            }
        }
        const int old_game = rec_scal<G>(P.boards, P.rec_bytes, env)[0].w;
        int4 *scg = reinterpret_cast<int4 *>(img + G::SC_OFF);
        scg[0] = make_int4((int)d[0], flags, (int)d[C], old_game < 0 ? 0 : old_game);
        scg[1] = make_int4(min(total_events, P.max_events), pairs[0], pairs[1], 0);
    }
    __syncthreads();
    for (int i = tid; i < P.rec_bytes / 16; i += NT) reinterpret_cast<int4 *>(rec)[i] = reinterpret_cast<const int4 *>(img)[i];
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

// =============================================================================================
// Host side: handle + C ABI
// =============================================================================================
struct sgx_env {
    sgx_config cfg;
    int64_t n_envs;
    int device;
    uint64_t seed;
    int64_t env_id_offset;
    int8_t *boards;
    DevTables *tab;
    uint8_t *setups;
    int64_t n_setups;
    int rec_bytes;
    int sc_off;
    int max_events;
    int K;
    unsigned long long *stamps;  // SGX_STAMPS builds only
};

namespace {

bool supported_geometry(int r, int c) {
    return (r == 10 && c == 10) || (r == 15 && c == 15) || (r == 8 && c == 8) || (r == 6 && c == 6) || (r == 5 && c == 5) ||
           (r == 4 && c == 4) || (r == 3 && c == 4);
}

#define DISPATCH_GEOMETRY(h, CALL)                                   \
    do {                                                             \
        const int r_ = (h)->cfg.rows, c_ = (h)->cfg.cols;            \
        if (r_ == 10 && c_ == 10) { CALL(10, 10); }                  \
        else if (r_ == 15 && c_ == 15) { CALL(15, 15); }             \
        else if (r_ == 8 && c_ == 8) { CALL(8, 8); }                 \
        else if (r_ == 6 && c_ == 6) { CALL(6, 6); }                 \
        else if (r_ == 5 && c_ == 5) { CALL(5, 5); }                 \
        else if (r_ == 4 && c_ == 4) { CALL(4, 4); }                 \
        else if (r_ == 3 && c_ == 4) { CALL(3, 4); }                 \
        else return fail(SGX_EINVAL, "unsupported board size%s");    \
    } while (0)

KParams make_params(const sgx_env *h) {
    KParams p;
    memset(&p, 0, sizeof(p));
    p.boards = h->boards;
    p.tab = h->tab;
    p.setups = h->setups;
    p.n_setups = (int32_t)h->n_setups;
    p.max_turns = h->cfg.max_turns;
    p.usable_rows = h->cfg.usable_rows;
    for (int i = 0; i < 12; ++i) p.piece_counts[i] = h->cfg.piece_counts[i];
    p.rec_bytes = h->rec_bytes;
    p.max_events = h->max_events;
    p.n_envs = h->n_envs;
    p.seed = h->seed;
    p.env_id_offset = h->env_id_offset;
#ifdef SGX_STAMPS
    p.stamps = h->stamps;
#endif
    return p;
}

unsigned grid_for(int64_t n) { return (unsigned)((n + 7) & ~(int64_t)7); }

int check_cfg(const sgx_config *cfg) {
    if (!cfg) return fail(SGX_EINVAL, "cfg is NULL%s");
    if (cfg->rows < 3 || cfg->cols < 3) return fail(SGX_EINVAL, "Both rows and columns have to be at least 3%s");
    if (cfg->rows * cfg->cols > SGX_MAX_CELLS) return fail(SGX_EINVAL, "rows*cols exceeds SGX_MAX_CELLS%s");
    if (cfg->usable_rows < 1 || cfg->usable_rows * 2 > cfg->rows) return fail(SGX_EINVAL, "usable_rows out of range%s");
    int total = 0;
    for (int i = 0; i < 12; ++i) {
        if (cfg->piece_counts[i] < 0) return fail(SGX_EINVAL, "negative piece count%s");
        total += cfg->piece_counts[i];
    }
    if (total > cfg->usable_rows * cfg->cols) return fail(SGX_EINVAL, "more pieces than usable cells%s");
    return SGX_OK;
}

}  // namespace

SGX_API int sgx_abi_version(void) { return SGX_ABI_VERSION; }
SGX_API const char *sgx_last_error(void) { return g_last_error.c_str(); }
SGX_API int64_t sgx_num_envs(const sgx_env *h) { return h ? h->n_envs : 0; }
SGX_API int sgx_spatial_channels(const sgx_env *h) { return h ? h->K : 0; }
SGX_API int64_t sgx_num_spatial_actions(const sgx_env *h) { return h ? (int64_t)h->cfg.rows * h->cfg.cols * h->K : 0; }
SGX_API int64_t sgx_action_size_1d(const sgx_env *h) {
    return h ? (int64_t)h->cfg.rows * h->cfg.cols * (h->cfg.rows + h->cfg.cols) + 1 : 0;
}

// Normalisation LUT [channels][LUT_STRIDE]: entry i of a channel's row = the float32 the reference produces for board value i
// (recent-moves channels: value i - 3).  Extended channel layout = [n_own_true][n_enemy_true][13 own PO][13 enemy PO]
// [obstacle][2 recent][12+12 captured][2 still]; partial has no enemy-true block (impl:1306-1332 vs impl:1200-1227);
// highs/lows maenv:202-313.  Original layout (value channels): partial impl:1126-1148 with maenv:146-199, full impl:1048-1070
// with maenv:87-143.  ranges/mids maenv:388-396, (x - mid) / range in float32 maenv:499-508.
static int lut_channels(bool full, bool original) {
    return original ? (full ? SGX_FO_OBS_CHANNELS_ORIGINAL : SGX_PO_OBS_CHANNELS_ORIGINAL) : (full ? FOBS_CH : OBS_CH);
}
static void build_lut(const sgx_config *cfg, bool full, float *lut, bool raw_values = false, bool original = false) {
    const int nch = lut_channels(full, original);
    enum { ONEHOT, VALUE, RECENT };
    float hi[FOBS_CH], lo[FOBS_CH];
    int kind[FOBS_CH], type[FOBS_CH];
    int rec0, cap0;
    for (int ch = 0; ch < nch; ++ch) { hi[ch] = 1.0f; lo[ch] = -1.0f; kind[ch] = VALUE; type[ch] = 0; }
    if (!original) {
        const int n_true = full ? 24 : 12, po_end = n_true + 26;
        for (int ch = 0; ch < po_end; ++ch) {          // one-hot piece channels: raw = (board value == piece type)
            kind[ch] = ONEHOT;
            type[ch] = ch < n_true ? ch % 12 + 1       // true pieces: types 1..12
                                   : (ch - n_true) % 13 + 1;   // PO pieces: types 1..13
        }
        rec0 = po_end + 1; cap0 = po_end + 3;
        for (int ch = cap0; ch < cap0 + 24; ++ch) { hi[ch] = 8.0f; lo[ch] = 0.0f; }
    } else {
        for (int ch = 0; ch < nch; ++ch) { hi[ch] = 2.0f; lo[ch] = 0.0f; }   // obstacles, captured, still
        if (full) {
            hi[0] = hi[1] = (float)SP_BOMB; hi[5] = hi[6] = (float)SP_UNKNOWN;
            rec0 = 3; cap0 = 7;
        } else {
            hi[0] = (float)SP_BOMB; hi[1] = hi[2] = (float)SP_UNKNOWN;
            rec0 = 4; cap0 = 6;
        }
    }
    kind[rec0] = kind[rec0 + 1] = RECENT;
    hi[rec0] = hi[rec0 + 1] = 1.0f; lo[rec0] = lo[rec0 + 1] = -3.0f;   // RecentMoves JUST_CAME_FROM .. JUST_ARRIVED_AND_CANT_DOUBLE_BACK
    for (int t = 1; t <= 12; ++t)
        if (cfg->piece_counts[t - 1] > 1) hi[cap0 + t - 1] = hi[cap0 + 12 + t - 1] = (float)cfg->piece_counts[t - 1];
    for (int ch = 0; ch < nch; ++ch) {
        volatile float range = (hi[ch] - lo[ch]) / 2.0f, mid = (hi[ch] + lo[ch]) / 2.0f;
        for (int i = 0; i < LUT_STRIDE; ++i) {
            float raw;
            if (kind[ch] == ONEHOT) raw = (i == type[ch]) ? 1.0f : 0.0f;
            else if (kind[ch] == RECENT) raw = (float)(i - 3);
            else raw = (float)i;
            volatile float d = raw - mid;        // two IEEE float32 roundings, as numpy does
            lut[ch * LUT_STRIDE + i] = raw_values ? raw : d / range;
        }
    }
}

SGX_API int sgx_build_obs_lut(const sgx_config *cfg, float *lut) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!lut) return fail(SGX_EINVAL, "lut is NULL%s");
    build_lut(cfg, false, lut);
    return SGX_OK;
}

SGX_API int sgx_build_full_obs_lut(const sgx_config *cfg, float *lut) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!lut) return fail(SGX_EINVAL, "lut is NULL%s");
    build_lut(cfg, true, lut);
    return SGX_OK;
}

SGX_API int sgx_build_original_obs_lut(const sgx_config *cfg, int32_t full, float *lut) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!lut) return fail(SGX_EINVAL, "lut is NULL%s");
    build_lut(cfg, full != 0, lut, false, true);
    return SGX_OK;
}

SGX_API int sgx_create(const sgx_config *cfg, int64_t n_envs, int device, uint64_t seed, int64_t env_id_offset, sgx_env **out) {
    if (!out) return fail(SGX_EINVAL, "out is NULL%s");
    *out = nullptr;
    if (int rc = check_cfg(cfg)) return rc;
    if (!supported_geometry(cfg->rows, cfg->cols)) return fail(SGX_EINVAL, "unsupported board size (built for 10x10, 15x15, 8x8, 6x6, 5x5, 4x4, 3x4)%s");
    if (n_envs <= 0 || n_envs > (int64_t)1 << 30) return fail(SGX_EINVAL, "n_envs out of range%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(SGX_EDEVICE, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(SGX_EINVAL, "device index out of range%s");
    HIP_TRY(hipSetDevice(device));
    sgx_env *h = new sgx_env();
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->n_envs = n_envs;
    h->device = device;
    h->seed = seed;
    h->env_id_offset = env_id_offset;
    const int rc_cells = cfg->rows * cfg->cols;
    {
        int pieces = 0;
        for (int i = 0; i < 12; ++i) pieces += cfg->piece_counts[i];
        h->max_events = 2 * pieces;                                   // every piece can be captured once
        const int st_off = (STORED_BOARDS * ((rc_cells + 3) & ~3) + 15) & ~15, sb = (((rc_cells + 7) / 8) + 15) & ~15;
        const int sc_off = st_off + 2 * sb;
        h->sc_off = sc_off;
        h->rec_bytes = (sc_off + 32 + 2 * h->max_events + 127) & ~127;   // whole 128-byte lines per game
    }
    h->K = 2 * (cfg->rows - 1) + 2 * (cfg->cols - 1) + 1;
    DevTables host_tab;
    memset(&host_tab, 0, sizeof(host_tab));
    {   // ABI LUTs [channels][16] -> device placement (rows at lut_row(ch), see LUT_ROW_PITCH)
        float dense[FOBS_CH * LUT_STRIDE];
        for (int k = 0; k < 8; ++k) {
            const bool original = (k & 4) != 0, raw = (k & 2) != 0, full = (k & 1) != 0;
            build_lut(cfg, full, dense, raw, original);
            for (int ch = 0; ch < lut_channels(full, original); ++ch)
                for (int i = 0; i < LUT_STRIDE; ++i) host_tab.lut[k][lut_row(ch) + i] = dense[ch * LUT_STRIDE + i];
        }
    }
    memcpy(host_tab.obstacles, cfg->obstacles, rc_cells);
    if (hipMalloc((void **)&h->boards, (size_t)n_envs * h->rec_bytes) != hipSuccess ||
        hipMalloc((void **)&h->tab, sizeof(DevTables)) != hipSuccess) {
        sgx_destroy(h);
        return fail(SGX_ENOMEM, "device allocation failed%s");
    }
#define HIP_TRY_OR_DESTROY(expr)                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) { sgx_destroy(h); return fail(SGX_EDEVICE, #expr ": %s", hipGetErrorString(e_)); } \
    } while (0)
    HIP_TRY_OR_DESTROY(hipMemset(h->boards, 0, (size_t)n_envs * h->rec_bytes));
    HIP_TRY_OR_DESTROY(hipMemcpy(h->tab, &host_tab, sizeof(DevTables), hipMemcpyHostToDevice));
#ifdef SGX_STAMPS
    HIP_TRY_OR_DESTROY(hipMalloc((void **)&h->stamps, (size_t)n_envs * 16 * sizeof(unsigned long long)));
    HIP_TRY_OR_DESTROY(hipMemset(h->stamps, 0, (size_t)n_envs * 16 * sizeof(unsigned long long)));
#endif
    init_scal_kernel<<<(unsigned)((n_envs + 255) / 256), 256>>>(h->boards, h->rec_bytes, h->sc_off, n_envs, cfg->max_turns);
    HIP_TRY_OR_DESTROY(hipGetLastError());
    HIP_TRY_OR_DESTROY(hipDeviceSynchronize());
#undef HIP_TRY_OR_DESTROY
    *out = h;
    return SGX_OK;
}

SGX_API int sgx_destroy(sgx_env *h) {
    if (!h) return SGX_OK;
    // teardown is best effort: there is nobody to report a failed free to
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->boards) (void)hipFree(h->boards);
    if (h->tab) (void)hipFree(h->tab);
    if (h->setups) (void)hipFree(h->setups);
    if (h->stamps) (void)hipFree(h->stamps);
    delete h;
    return SGX_OK;
}

SGX_API int sgx_set_setup_table(sgx_env *h, const uint8_t *table_host, int64_t n_setups) {
    if (!h || !table_host || n_setups <= 0 || n_setups > 0x7fffffff) return fail(SGX_EINVAL, "bad setup table%s");
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = (size_t)n_setups * h->cfg.usable_rows * h->cfg.cols;
    for (size_t i = 0; i < bytes; ++i)
        if (table_host[i] > 12) return fail(SGX_EINVAL, "setup table holds a piece code > 12%s");
    if (h->setups) { HIP_TRY(hipFree(h->setups)); h->setups = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->setups, bytes));
    HIP_TRY(hipMemcpy(h->setups, table_host, bytes, hipMemcpyHostToDevice));
    h->n_setups = n_setups;
    return SGX_OK;
}

SGX_API int sgx_reset(sgx_env *h, const uint8_t *env_select_dev, const int8_t *p1_maps_dev, const int8_t *p2_maps_dev, void *stream) {
    if (!h) return fail(SGX_EINVAL, "handle is NULL%s");
    if ((p1_maps_dev == nullptr) != (p2_maps_dev == nullptr)) return fail(SGX_EINVAL, "pass both piece maps or neither%s");
    HIP_TRY(hipSetDevice(h->device));
    ResetParams rp;
    rp.k = make_params(h);
    rp.select = env_select_dev;
    rp.p1_maps = p1_maps_dev;
    rp.p2_maps = p2_maps_dev;
#define CALL_RESET(R, C) reset_kernel<R, C><<<(unsigned)h->n_envs, 64, 0, (hipStream_t)stream>>>(rp)
    DISPATCH_GEOMETRY(h, CALL_RESET);
#undef CALL_RESET
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

static int launch_step(sgx_env *h, const KParams &p, void *stream) {
    const int gpw = h->cfg.rows * h->cfg.cols <= 16 ? 4 : (h->cfg.rows * h->cfg.cols <= 32 ? 2 : 1);   // Geo::GPW
    const unsigned grid = grid_for((h->n_envs + WPB * gpw - 1) / (WPB * gpw));
    const bool full = p.io.fobs_dev || p.io.final_fobs_dev, original = (p.io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
#define CALL_STEP_KIND(R, C, KIND)                                                                 \
    do {                                                                                           \
        if (p.mode) observe_kernel<R, C, KIND><<<grid, 64 * WPB, 0, (hipStream_t)stream>>>(p);     \
        else step_kernel<R, C, KIND><<<grid, 64 * WPB, 0, (hipStream_t)stream>>>(p);               \
    } while (0)
#define CALL_STEP0(R, C) CALL_STEP_KIND(R, C, 0)
#define CALL_STEP1(R, C) CALL_STEP_KIND(R, C, 1)
#define CALL_STEP2(R, C) CALL_STEP_KIND(R, C, 2)
#define CALL_STEP3(R, C) CALL_STEP_KIND(R, C, 3)
    if (p.io.flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) {
        if (full || original) return fail(SGX_EINVAL, "state-coordinate masks come with the 67-channel partial observation only%s");
#define CALL_STEP_MAPPED(R, C)                                                                     \
    do {                                                                                           \
        if (p.mode) observe_kernel<R, C, 0, true><<<grid, 64 * WPB, 0, (hipStream_t)stream>>>(p);  \
        else step_kernel<R, C, 0, true><<<grid, 64 * WPB, 0, (hipStream_t)stream>>>(p);            \
    } while (0)
        DISPATCH_GEOMETRY(h, CALL_STEP_MAPPED);
#undef CALL_STEP_MAPPED
    } else if (!original && !full) DISPATCH_GEOMETRY(h, CALL_STEP0);
    else if (!original) DISPATCH_GEOMETRY(h, CALL_STEP1);
    else if (!full) DISPATCH_GEOMETRY(h, CALL_STEP2);
    else DISPATCH_GEOMETRY(h, CALL_STEP3);
#undef CALL_STEP0
#undef CALL_STEP1
#undef CALL_STEP2
#undef CALL_STEP3
#undef CALL_STEP_KIND
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_observe(sgx_env *h, float *obs_dev, float *fobs_dev, uint8_t *mask_dev, int8_t *player_dev, int32_t flags, void *stream) {
    if (!h) return fail(SGX_EINVAL, "handle is NULL%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
    p.mode = 1;
    p.io.obs_dev = obs_dev;
    p.io.fobs_dev = fobs_dev;
    p.io.flags = flags & (SGX_STEP_RAW_OBS | SGX_STEP_ORIGINAL_CHANNELS | SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS);
    p.io.mask_dev = mask_dev;
    p.io.player_dev = player_dev;
    return launch_step(h, p, stream);
}

SGX_API int sgx_time_observe(sgx_env *h, float *obs_dev, uint8_t *mask_dev, int32_t launches, void *stream, float *microseconds) {
    if (!h || !microseconds || launches <= 0) return fail(SGX_EINVAL, "bad argument%s");
    HIP_TRY(hipSetDevice(h->device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = sgx_observe(h, obs_dev, nullptr, mask_dev, nullptr, 0, stream);       // untimed first touch
    if (rc == SGX_OK) {
        hipError_t e = hipEventRecord(e0, (hipStream_t)stream);
        for (int32_t i = 0; i < launches && rc == SGX_OK; ++i) rc = sgx_observe(h, obs_dev, nullptr, mask_dev, nullptr, 0, stream);
        if (e == hipSuccess) e = hipEventRecord(e1, (hipStream_t)stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (rc == SGX_OK && e != hipSuccess) rc = fail(SGX_EDEVICE, "timing events: %s", hipGetErrorString(e));
        *microseconds = ms * 1000.f / (float)launches;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

SGX_API int sgx_step(sgx_env *h, const sgx_step_io *io, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev) return fail(SGX_EINVAL, "actions_dev is NULL%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
    p.mode = 0;
    p.io = *io;
    return launch_step(h, p, stream);
}

SGX_API int sgx_step_n(sgx_env *h, const sgx_step_io *io, int32_t n_steps, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev || io->next_actions_dev != io->actions_dev)
        return fail(SGX_EINVAL, "sgx_step_n needs next_actions_dev == actions_dev (each step plays the action the previous one drew)%s");
    if (n_steps < 0) return fail(SGX_EINVAL, "n_steps is negative%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
    p.mode = 0;
    p.io = *io;
    for (int32_t i = 0; i < n_steps; ++i)
        if (int rc = launch_step(h, p, stream)) return rc;
    return SGX_OK;
}

SGX_API int sgx_sample_valid(sgx_env *h, const uint8_t *mask_dev, int32_t *actions_dev, void *stream) {
    if (!h || !mask_dev || !actions_dev) return fail(SGX_EINVAL, "NULL argument%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
#define CALL_SAMPLE(R, C) sample_kernel<R, C><<<(unsigned)h->n_envs, 64, 0, (hipStream_t)stream>>>(p, mask_dev, actions_dev)
    DISPATCH_GEOMETRY(h, CALL_SAMPLE);
#undef CALL_SAMPLE
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_export_state(sgx_env *h, int64_t *state_dev, int8_t *player_dev, void *stream) {
    if (!h || !state_dev) return fail(SGX_EINVAL, "NULL argument%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
#define CALL_EXPORT(R, C) export_kernel<R, C><<<(unsigned)h->n_envs, 256, 0, (hipStream_t)stream>>>(p, state_dev, player_dev)
    DISPATCH_GEOMETRY(h, CALL_EXPORT);
#undef CALL_EXPORT
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_import_state(sgx_env *h, const int64_t *state_dev, const int8_t *player_dev, void *stream) {
    if (!h || !state_dev) return fail(SGX_EINVAL, "NULL argument%s");
    HIP_TRY(hipSetDevice(h->device));
    KParams p = make_params(h);
#define CALL_IMPORT(R, C) import_kernel<R, C><<<(unsigned)h->n_envs, 256, 0, (hipStream_t)stream>>>(p, state_dev, player_dev)
    DISPATCH_GEOMETRY(h, CALL_IMPORT);
#undef CALL_IMPORT
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_get_env_info(sgx_env *h, int32_t *info_dev, void *stream) {
    if (!h || !info_dev) return fail(SGX_EINVAL, "NULL argument%s");
    HIP_TRY(hipSetDevice(h->device));
    info_kernel<<<(unsigned)((h->n_envs + 255) / 256), 256, 0, (hipStream_t)stream>>>(h->boards, h->rec_bytes, h->sc_off, info_dev, h->n_envs);
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

#ifdef SGX_STAMPS
// diagnostic build only (tools/phase_stamps.py): device pointer of the [N][16] stamp buffer
SGX_API void *sgx_debug_stamps(sgx_env *h) { return h ? (void *)h->stamps : nullptr; }
#endif
