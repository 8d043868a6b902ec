// stratego_mi355x.hip -- HIP kernels (gfx950 / CDNA4) and C-ABI of the batched Stratego env.
//
// One wavefront (64 lanes) per game; boards of <= 32 cells share a wave between 2 or 4 games (Geo::LPG lanes per game).
// A game's compact state record (4 dense int8 boards + bitmaps + scalars + capture events, whole 128-byte lines) is expanded
// to 32 boards in LDS, the move is applied there, the next mover's valid-actions mask is built in LDS as bits and its
// 67-channel normalised observation is rendered straight into line-aligned 16-byte global stores.
// HBM-bound integer/byte work: no MFMA.  See DESIGN.md for the data layout and byte accounting.
// Device code lives in the sgx_*.h headers included below (one translation unit, in this order); this file holds the error
// plumbing and the host side of the C ABI.
//
// Reference functions reproduced (paths relative to /root/reference/stratego_env):
//   game/stratego_procedural_impl.py  (impl)   stratego_multiagent_env.py (maenv)   game/util.py (util)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sched.h>
#include <time.h>
#include <string.h>
#include <string>
#include <vector>

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

// Every entry point that touches the device runs on the handle's device and puts the caller's current device back on every
// return path: a process that drives several handles on several GPUs (SURVEY 8e's other layout: one thread + stream per GPU), or
// that sits inside a torch.cuda.device(...) scope, never finds its device switched under it.
struct DeviceGuard {
    int prev = -1;
    bool restore = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            restore = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (restore) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define SGX_ON_DEVICE(dev)                                                                                      \
    DeviceGuard device_guard_(dev);                                                                             \
    if (device_guard_.err != hipSuccess) return fail(SGX_EDEVICE, "selecting the handle's device: %s", hipGetErrorString(device_guard_.err))

// Hash of the sources this binary was compiled from (stratego_env_amd/build.py: source_hash() over csrc/* and the ABI header,
// passed as -DSGX_BUILD_ID): _lib.load() refuses a library whose id differs from the sources next to it, build.needs_build()
// reads the marker straight from the file.
#ifndef SGX_BUILD_ID
#define SGX_BUILD_ID "unknown"
#endif
const char g_build_id[] = "SGX_BUILD_ID=" SGX_BUILD_ID;

}  // namespace

#include "sgx_layout.h"
#include "sgx_obs.h"
#include "sgx_mask.h"
#include "sgx_setup.h"
#include "sgx_step.h"
#include "sgx_lane.h"
#include "sgx_lane_kernel.h"
#include "sgx_aux_kernels.h"
#include "sgx_choose.h"
#include "sgx_mem.h"


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
    int map_mode, map_arg;       // SGX_MAP experiment (group_of_block)
    int xcd_skew;                // sgx_set_xcd_skew: per mille more work for the even XCDs; -1 = by the launch's output size
    int xcd_explicit;            // xcd_w was set by sgx_set_xcd_shares: it overrides the skew rule
    int32_t xcd_w[8];            // shares of the eight XCDs, per mille of the mean share (sum 8000)
    int nt_mode;                 // sgx_set_nt_stores: -1 = by the launch's output size, 0 = never, 1 = always
    int lane_mode;               // sgx_set_lane_kernel: 0 = never the lane-per-game kernel, otherwise wherever it is eligible
    float placement_target_us;   // sgx_set_placement_target: sgx_alloc_outputs searches on until a candidate is within 3 % of it (0 = its own stop rules)
    hipStream_t chain_stream[SGX_MAX_CHAINS];   // sgx_rollout: created on first use
    hipEvent_t chain_fork, chain_join[SGX_MAX_CHAINS];
    // sgx_step_sync on a handful of games (single_kernel): a host-mapped word the kernel publishes its sequence number in, and the
    // device counter of finished workgroups; created on first use
    uint32_t *sync_flag_host, *sync_flag_dev, *sync_count;
    uint32_t sync_seq;
    int no_single;               // SGX_NO_SINGLE=1: sgx_step_sync never takes the single_kernel path (A/B measurements)
    uint8_t *san_flags;          // sgx_step_states: per-state "had to be altered" flags when the caller passes none (created on first use)
    int32_t *redo_list;          // sgx_step_states: [n_envs] ids of the states the first pass had to alter, then the counter (created on first use)
    int general_states;          // sgx_set_general_states: 0 = flagged states stay sanitised, otherwise the second, general-state pass redoes them
    int no_multi_step;           // SGX_MULTI_STEP=0 (or a runtime that refuses the LDS size): sgx_step_n / sgx_step_ring never take lane_steps_kernel
    int multi_step_attr;         // lane_steps_kernel's dynamic-LDS attribute has been raised
    int last_kind;               // sgx_last_launch_kind: which kernel the last step / observe launch of this handle was
    int multi_step_wave;         // SGX_MULTI_STEP_WAVE: the multi-step launch of the wave-per-game kernels (steps_kernel) too
    int steps_barrier;           // sgx_set_steps_barrier / SGX_STEPS_BARRIER: -1 auto (long rings that render float32 observations: launch_wave_steps), 0 never, 1 always
    int half_wave;               // SGX_HALF_WAVE: launches without an observation play two games per wave where the board allows it (Geo<R, C, 2>)
    // sgx_step_ring with more output sets than fit the kernel arguments: the sets' pointers in a device table (filled through a pinned host
    // copy; the event guards the staging buffer against being rewritten before the previous upload has run)
    void **ring_tab_dev, **ring_tab_host;
    int ring_tab_cap, ring_tab_n;
    hipEvent_t ring_tab_ev;
    hipStream_t ring_tab_stream;
};

namespace {

// Board sizes compiled into this library.  Every reference variant (game/config.py) is built in; any other size (rows, cols >= 3,
// rows * cols <= SGX_MAX_CELLS) gets a library of its own, compiled from the same sources with -DSGX_EXTRA_R=<rows>
// -DSGX_EXTRA_C=<cols> -DSGX_ONLY_EXTRA (stratego_env_amd/build.py: build_geometry), like the reference's
// StrategoProceduralEnv(rows, columns) accepts any size (penv:27-36).
#ifdef SGX_ONLY_EXTRA
#define SGX_BUILTIN_GEOMETRIES(X, A)
#else
#define SGX_BUILTIN_GEOMETRIES(X, A) X(A, 10, 10) X(A, 15, 15) X(A, 8, 8) X(A, 6, 6) X(A, 5, 5) X(A, 4, 4) X(A, 3, 4)
#endif
#ifdef SGX_EXTRA_R
#define SGX_EXTRA_GEOMETRY(X, A) X(A, SGX_EXTRA_R, SGX_EXTRA_C)
static_assert(SGX_EXTRA_R >= 3 && SGX_EXTRA_C >= 3 && SGX_EXTRA_R * SGX_EXTRA_C <= SGX_MAX_CELLS, "SGX_EXTRA_R x SGX_EXTRA_C out of range");
#else
#define SGX_EXTRA_GEOMETRY(X, A)
#endif
#define SGX_ALL_GEOMETRIES(X, A) SGX_BUILTIN_GEOMETRIES(X, A) SGX_EXTRA_GEOMETRY(X, A)

bool supported_geometry(int r, int c) {
#define SGX_MATCH(A, R, C) if (r == (R) && c == (C)) return true;
    SGX_ALL_GEOMETRIES(SGX_MATCH, _)
#undef SGX_MATCH
    return false;
}

#define SGX_DISPATCH_ONE(CALL, R, C) if (!done_ && r_ == (R) && c_ == (C)) { CALL(R, C); done_ = true; }
#define DISPATCH_GEOMETRY(h, CALL)                                                     \
    do {                                                                               \
        const int r_ = (h)->cfg.rows, c_ = (h)->cfg.cols;                              \
        bool done_ = false;                                                            \
        SGX_ALL_GEOMETRIES(SGX_DISPATCH_ONE, CALL)                                     \
        if (!done_) return fail(SGX_EINVAL, "unsupported board size%s");               \
    } while (0)

// bytes of one game's compact observation record (SGX_STEP_COMPACT_OBS): 4-bit codes padded to 16 B, a 16-byte header, then room for
// every entry that can be without a code (capture events + the four recent-move pairs) as {uint32 entry, float value}; whole 128-byte lines
int compact_capacity(const sgx_env *h) { return h->max_events + 4; }
int compact_obs_stride(const sgx_env *h) {
    const int nib = ((h->cfg.rows * h->cfg.cols * OBS_CH + 1) / 2 + 15) & ~15;
    return (nib + 16 + 8 * compact_capacity(h) + 127) & ~127;
}
int mask_words(const sgx_env *h) {
    const int na = h->cfg.rows * h->cfg.cols * h->K;
    return (((na + 31) / 32 + 1) + 3) & ~3;                          // Geo::MB_WORDS
}

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
    for (int i = 0; i < 12; ++i)
        if (h->cfg.piece_counts[i] > EV_COUNT_MAX) p.multi_ev = 1;
    p.compact_stride = compact_obs_stride(h);
    p.n_envs = h->n_envs;
    p.seed = h->seed;
    p.env_id_offset = h->env_id_offset;
#ifdef SGX_STAMPS
    p.stamps = h->stamps;
#endif
    return p;
}

int check_cfg(const sgx_config *cfg) {
    if (!cfg) return fail(SGX_EINVAL, "cfg is NULL%s");
    if (cfg->rows < 3 || cfg->cols < 3) return fail(SGX_EINVAL, "Both rows and columns have to be at least 3%s");
    if (cfg->rows * cfg->cols > SGX_MAX_CELLS) return fail(SGX_EINVAL, "rows*cols exceeds SGX_MAX_CELLS%s");
    if (cfg->usable_rows < 1 || cfg->usable_rows * 2 > cfg->rows) return fail(SGX_EINVAL, "usable_rows out of range%s");
    for (int i = 0; i < 12; ++i) {
        if (cfg->piece_counts[i] < 0) return fail(SGX_EINVAL, "negative piece count%s");
        // (a capture event counts to 8 = the scouts of Standard; a variant with more pieces of one type chains events: KParams::multi_ev)
        if (cfg->piece_counts[i] > SGX_MAX_PIECES_PER_TYPE) return fail(SGX_EINVAL, "more than 127 pieces of one type per side%s");
    }
    // (more pieces than usable cells is legal for a handle that never samples random setups: an env_config that overrides
    //  'piece_amounts' changes the normalisation only; sgx_reset checks it where it matters)
    if (cfg->capture_capacity < 0 || 2 * cfg->capture_capacity > cfg->rows * cfg->cols) return fail(SGX_EINVAL, "capture_capacity out of range%s");
    return SGX_OK;
}

}  // namespace

// sampled setups without a table place piece_counts pieces on the usable back rows: they have to fit
static int check_random_setups(const sgx_env *h) {
    if (h->setups) return SGX_OK;
    int total = 0;
    for (int i = 0; i < 12; ++i) total += h->cfg.piece_counts[i];
    if (total > h->cfg.usable_rows * h->cfg.cols) return fail(SGX_EINVAL, "random setups: more pieces than usable cells%s");
    return SGX_OK;
}

SGX_API int sgx_abi_version(void) { return SGX_ABI_VERSION; }
SGX_API const char *sgx_build_id(void) { return g_build_id + 13; }
SGX_API int sgx_supports_geometry(int32_t rows, int32_t cols) { return supported_geometry(rows, cols) ? 1 : 0; }
SGX_API const char *sgx_last_error(void) { return g_last_error.c_str(); }
SGX_API int64_t sgx_num_envs(const sgx_env *h) { return h ? h->n_envs : 0; }
SGX_API int64_t sgx_record_bytes(const sgx_env *h) { return h ? h->rec_bytes : 0; }
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

// 'extended' observations are rendered from 4-bit codes (sgx_obs.h): code c decodes to sext(c) / 4.  Templates of the channel
// defaults and the codes of captured counts / recent-move codes, derived from the LUTs themselves.
static int float_code(float f) {
    for (int c = 0; c < 16; ++c)
        if (c != CODE_ESC && (float)(c < 8 ? c : c - 16) / 4.0f == f) return c;    // (CODE_ESC marks uncoded entries)
    return CODE_NONE;
}
static int build_code_tables(const sgx_config *cfg, DevTables *tab) {
    const int rc_cells = cfg->rows * cfg->cols;
    float dense[2][FOBS_CH * LUT_STRIDE];
    for (int raw = 0; raw < 2; ++raw) {
        for (int full = 0; full < 2; ++full) {
            build_lut(cfg, full != 0, dense[full], raw != 0, false);
            const int nch = full ? FOBS_CH : OBS_CH, rec0 = full ? 51 : 39;
            uint8_t *t = tab->tmpl[2 * raw + full];
            memset(t, 0, TMPL_MAX_BYTES);
            for (int ch = 0; ch < nch; ++ch) {
                const bool rec = ch == rec0 || ch == rec0 + 1;
                const int code = float_code(dense[full][ch * LUT_STRIDE + (rec ? 3 : 0)]);
                if (code == CODE_NONE) return fail(SGX_EINVAL, "a channel default has no 4-bit code%s");
                // indicator channels (one-hot blocks, obstacle, never-moved): the value of a set entry must decode from NIB_ONE
                const int n_true = full ? 24 : 12, po_end = n_true + 26;
                int set_index = -1;
                if (ch < n_true) set_index = ch % 12 + 1;
                else if (ch < po_end) set_index = (ch - n_true) % 13 + 1;
                else if (ch == po_end || ch >= nch - 2) set_index = 1;
                if (set_index >= 0 && float_code(dense[full][ch * LUT_STRIDE + set_index]) != NIB_ONE)
                    return fail(SGX_EINVAL, "an indicator channel's set value is not 1.0%s");
                for (int cell = 0; cell < rc_cells; ++cell) {
                    const int e = cell * nch + ch;
                    t[e >> 1] |= (uint8_t)(code << (4 * (e & 1)));
                }
            }
        }
        uint8_t *ct = tab->codetab[raw];
        memset(ct, CODE_NONE, CODETAB_BYTES);
        for (int t = 0; t < 12; ++t)
            for (int v = 0; v < LUT_STRIDE; ++v) {
                // own and enemy block, partial and full kind: the same normalisation per piece type (maenv:288-298 / 229-239)
                const float f = dense[0][(41 + t) * LUT_STRIDE + v];
                if (f != dense[0][(53 + t) * LUT_STRIDE + v] || f != dense[1][(53 + t) * LUT_STRIDE + v] || f != dense[1][(65 + t) * LUT_STRIDE + v])
                    return fail(SGX_EINVAL, "captured-count channels of one piece type are normalised differently%s");
                ct[16 * t + v] = (uint8_t)float_code(f);
            }
        for (int v = 0; v < 5; ++v) {
            const float f = dense[0][39 * LUT_STRIDE + v];
            if (f != dense[0][40 * LUT_STRIDE + v] || f != dense[1][51 * LUT_STRIDE + v] || f != dense[1][52 * LUT_STRIDE + v])
                return fail(SGX_EINVAL, "recent-move channels are normalised differently%s");
            ct[CODETAB_REC + v] = (uint8_t)float_code(f);
        }
    }
    return SGX_OK;
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
    if (!supported_geometry(cfg->rows, cfg->cols))
        return fail(SGX_EINVAL, "board size not compiled into this library (built in: 10x10, 15x15, 8x8, 6x6, 5x5, 4x4, 3x4; any other size: "
                                "build the same sources with -DSGX_EXTRA_R=<rows> -DSGX_EXTRA_C=<cols>, stratego_env_amd/build.py build_geometry)%s");
    if (n_envs <= 0 || n_envs > (int64_t)1 << 30) return fail(SGX_EINVAL, "n_envs out of range%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(SGX_EDEVICE, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(SGX_EINVAL, "device index out of range%s");
    SGX_ON_DEVICE(device);
    sgx_env *h = new sgx_env();
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->n_envs = n_envs;
    h->device = device;
    h->seed = seed;
    h->env_id_offset = env_id_offset;
    h->nt_mode = -1;
    if (const char *e = getenv("SGX_NT")) {           // SGX_NT=0|1|auto: the default of sgx_set_nt_stores for new handles
        if (!strcmp(e, "0")) h->nt_mode = 0;
        else if (!strcmp(e, "1")) h->nt_mode = 1;
    }
    h->lane_mode = -1;
    h->general_states = 1;
    if (const char *e = getenv("SGX_GENERAL_STATES")) h->general_states = atoi(e);
    if (const char *e = getenv("SGX_LANE")) { if (!strcmp(e, "0")) h->lane_mode = 0; else if (!strcmp(e, "1")) h->lane_mode = 1; }   // SGX_LANE=0|1|auto
    h->xcd_skew = -1;
    if (const char *e = getenv("SGX_XCD_SKEW")) { if (strcmp(e, "auto")) h->xcd_skew = atoi(e); }   // SGX_XCD_SKEW=<per mille>|auto
    if (h->xcd_skew > 900) h->xcd_skew = 900;
    if (const char *e = getenv("SGX_NO_SINGLE")) h->no_single = atoi(e);
    if (const char *e = getenv("SGX_MULTI_STEP")) h->no_multi_step = !strcmp(e, "0");
    h->multi_step_wave = 1;
    if (const char *e = getenv("SGX_MULTI_STEP_WAVE")) h->multi_step_wave = strcmp(e, "0") != 0;
    h->half_wave = 1;
    if (const char *e = getenv("SGX_HALF_WAVE")) h->half_wave = strcmp(e, "0") != 0;
    h->steps_barrier = -1;
    if (const char *e = getenv("SGX_STEPS_BARRIER")) h->steps_barrier = !strcmp(e, "0") ? 0 : !strcmp(e, "1") ? 1 : -1;
    if (const char *e = getenv("SGX_MAP")) { h->map_mode = atoi(e); if (const char *c = strchr(e, ',')) h->map_arg = atoi(c + 1); }
    const int rc_cells = cfg->rows * cfg->cols;
    {
        int pieces = 0;
        for (int i = 0; i < 12; ++i) pieces += cfg->piece_counts[i];
        if (cfg->capture_capacity > pieces) pieces = cfg->capture_capacity;
        h->max_events = 2 * pieces;                                   // every piece can be captured once
        // ... and both sides together never have more pieces than the board has cells: Geo::EVL_MAX = rows * cols bounds the LDS event
        // lists and StateIO::IMG (an env_config that overrides 'piece_amounts' for the normalisation only -- Micro with Standard's 40
        // pieces -- would ask for 80 events on 12 cells otherwise)
        if (h->max_events > rc_cells) h->max_events = rc_cells;
        const int st_off = (STORED_BOARDS * ((rc_cells + 3) & ~3) + 15) & ~15, sb = (((rc_cells + 7) / 8) + 15) & ~15;
        const int sc_off = st_off + 2 * sb;
        h->sc_off = sc_off;
        h->rec_bytes = (sc_off + 32 + (rc_cells > 256 ? 4 : 2) * h->max_events + 127) & ~127;   // whole 128-byte lines per game (Geo::ev_t events)
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
    if (int rc = build_code_tables(cfg, &host_tab)) { delete h; return rc; }
    for (int a = 0; a < 16; ++a)
        for (int d = 0; d < 16; ++d) host_tab.combat[16 * a + d] = (uint8_t)combat_outcome(a, d);
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
    DeviceGuard device_guard_(h->device);
    (void)hipDeviceSynchronize();
    if (h->boards) (void)hipFree(h->boards);
    if (h->tab) (void)hipFree(h->tab);
    if (h->setups) (void)hipFree(h->setups);
    if (h->stamps) (void)hipFree(h->stamps);
    for (int c = 0; c < SGX_MAX_CHAINS; ++c) {
        if (h->chain_stream[c]) (void)hipStreamDestroy(h->chain_stream[c]);
        if (h->chain_join[c]) (void)hipEventDestroy(h->chain_join[c]);
    }
    if (h->chain_fork) (void)hipEventDestroy(h->chain_fork);
    if (h->sync_flag_host) (void)hipHostFree(h->sync_flag_host);
    if (h->sync_count) (void)hipFree(h->sync_count);
    if (h->san_flags) (void)hipFree(h->san_flags);
    if (h->redo_list) (void)hipFree(h->redo_list);
    if (h->ring_tab_dev) (void)hipFree(h->ring_tab_dev);
    if (h->ring_tab_host) (void)hipHostFree(h->ring_tab_host);
    if (h->ring_tab_ev) (void)hipEventDestroy(h->ring_tab_ev);
    delete h;
    return SGX_OK;
}

SGX_API int sgx_set_nt_stores(sgx_env *h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(SGX_EINVAL, "sgx_set_nt_stores: mode must be -1 (auto), 0 or 1%s");
    h->nt_mode = mode;
    return SGX_OK;
}

SGX_API int sgx_set_placement_target(sgx_env *h, float target_us) {
    if (!h || !(target_us >= 0.f)) return fail(SGX_EINVAL, "sgx_set_placement_target: target_us must be >= 0%s");
    h->placement_target_us = target_us;
    return SGX_OK;
}

SGX_API int sgx_set_general_states(sgx_env *h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) return fail(SGX_EINVAL, "sgx_set_general_states: mode must be 0 or 1%s");
    h->general_states = mode;
    return SGX_OK;
}

SGX_API int sgx_last_launch_kind(const sgx_env *h) { return h ? h->last_kind : -1; }

SGX_API int sgx_set_multi_step(sgx_env *h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) return fail(SGX_EINVAL, "sgx_set_multi_step: mode must be 0 or 1%s");
    h->no_multi_step = mode ? 0 : 1;
    return SGX_OK;
}

SGX_API int sgx_set_half_wave(sgx_env *h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) return fail(SGX_EINVAL, "sgx_set_half_wave: mode must be 0 or 1%s");
    h->half_wave = mode;
    return SGX_OK;
}

SGX_API int sgx_set_steps_barrier(sgx_env *h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(SGX_EINVAL, "sgx_set_steps_barrier: mode must be -1 (auto), 0 or 1%s");
    h->steps_barrier = mode;
    return SGX_OK;
}

SGX_API int sgx_set_lane_kernel(sgx_env *h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(SGX_EINVAL, "sgx_set_lane_kernel: mode must be -1 (auto), 0 or 1%s");
    h->lane_mode = mode;
    return SGX_OK;
}

static void launch_shares(const sgx_env *h, bool streaming, int32_t *w);

SGX_API int sgx_set_xcd_skew(sgx_env *h, int32_t per_mille) {
    if (!h || per_mille < -1 || per_mille > 900) return fail(SGX_EINVAL, "sgx_set_xcd_skew: -1 (auto) or 0 .. 900 per mille%s");
    h->xcd_skew = per_mille;
    h->xcd_explicit = 0;
    return SGX_OK;
}

SGX_API int sgx_set_xcd_shares(sgx_env *h, const int32_t *per_mille) {
    if (!h) return fail(SGX_EINVAL, "handle is NULL%s");
    if (!per_mille) { h->xcd_explicit = 0; return SGX_OK; }
    for (int x = 0; x < 8; ++x)
        if (per_mille[x] < 1 || per_mille[x] > 8000) return fail(SGX_EINVAL, "sgx_set_xcd_shares: every share must be 1 .. 8000 per mille%s");
    for (int x = 0; x < 8; ++x) h->xcd_w[x] = per_mille[x];
    h->xcd_explicit = 1;
    return SGX_OK;
}

SGX_API int sgx_get_xcd_shares(sgx_env *h, int32_t *per_mille, int32_t *calibrated) {
    if (!h || !per_mille) return fail(SGX_EINVAL, "NULL argument%s");
    launch_shares(h, true, per_mille);
    if (calibrated) *calibrated = h->xcd_explicit;
    return SGX_OK;
}

SGX_API int sgx_set_setup_table(sgx_env *h, const uint8_t *table_host, int64_t n_setups) {
    if (!h || !table_host || n_setups <= 0 || n_setups > 0x7fffffff) return fail(SGX_EINVAL, "bad setup table%s");
    SGX_ON_DEVICE(h->device);
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
    if (!p1_maps_dev)
        if (int rc = check_random_setups(h)) return rc;
    SGX_ON_DEVICE(h->device);
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

// Observation bytes one launch writes; beyond what the 256 MiB Infinity Cache absorbs the kernel uses non-temporal stores for the
// lines a wave writes whole (measured crossover between 281 and 316 MB on four board sizes, sgx_obs.h).
// `sets`: output sets written round-robin (sgx_step_ring): a line is written again only after `sets` launches, so what has to fit the
// cache for plain stores to pay is the observations of all of them.
static bool launch_streams_past_cache(const sgx_env *h, const KParams &p, int sets = 1) {
    const bool original = (p.io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
    const int64_t cells = (int64_t)h->cfg.rows * h->cfg.cols;
    int64_t bytes = 0;
    if (p.io.obs_dev) bytes += h->n_envs * cells * lut_channels(false, original) * 4;
    if (p.io.fobs_dev) bytes += h->n_envs * cells * lut_channels(true, original) * 4;
    return bytes * sets > (int64_t)300 * 1000 * 1000;
}

// Workgroup-groups per XCD for `groups` groups of games from the shares w[8] (per mille of the mean share): KParams::xcd_first /
// xcd_count; returns the grid size (8 x the largest share).
#define SGX_XCD_SKEW_DEFAULT 100
static unsigned shares_for(KParams &p, int64_t groups, const int32_t *w) {
    int64_t sum = 0, cnt[8], given = 0;
    for (int x = 0; x < 8; ++x) sum += w[x];
    for (int x = 0; x < 8; ++x) { cnt[x] = groups * w[x] / sum; given += cnt[x]; }
    for (int x = 0; given < groups; x = (x + 1) & 7) { cnt[x] += 1; given += 1; }          // the remainder, one group each
    int64_t at = 0, mx = 1;
    for (int x = 0; x < 8; ++x) {
        p.xcd_first[x] = (int32_t)at;
        p.xcd_count[x] = (int32_t)cnt[x];
        at += cnt[x];
        if (cnt[x] > mx) mx = cnt[x];
    }
    return (unsigned)(8 * mx);
}
// the shares of a launch: explicit ones, else the odd-even skew where the launch is a saturating write stream
static void launch_shares(const sgx_env *h, bool streaming, int32_t *w) {
    if (h->xcd_explicit) { for (int x = 0; x < 8; ++x) w[x] = h->xcd_w[x]; return; }
    // measured in one process on one set of buffers (tools/skew_ab.py, profiles/r03_skew_ab.log): 10x10 -3 ... -5 %, 8x8 -5 %; 6x6,
    // 15x15 and Micro, where the game logic shares the critical path, +3 ... +7 %
    const int cells = h->cfg.rows * h->cfg.cols;
    const int skew = h->xcd_skew < 0 ? ((streaming && cells >= 64 && cells <= 100) ? SGX_XCD_SKEW_DEFAULT : 0) : h->xcd_skew;
    for (int x = 0; x < 8; ++x) w[x] = (x & 1) ? 1000 - skew : 1000 + skew;
}

// The lane-per-game kernel plays this launch?  (sgx_lane_kernel.h: what it covers; everything else is the wave-per-game kernel.)
static bool lane_eligible(const sgx_env *h, const KParams &p, bool full, bool original, bool multi_step = false) {
    const int cells = h->cfg.rows * h->cfg.cols;
    auto aligned = [](const void *ptr, uintptr_t a) { return (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; };
    // -1 (auto): only where it is the faster kernel -- launches that emit no observation (2 x on Micro / Tiny, docs/DESIGN_rounds_4-5.md section 3.3) and
    // the fused multi-step launches of sgx_step_n / sgx_step_ring (lane_steps_kernel: the logic of step t + 1 under the stores of step t)
    if (h->lane_mode < 0 && p.io.obs_dev && !multi_step) return false;
    return h->lane_mode != 0 && !p.multi_ev && !(p.io.flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) && cells <= 16 && cells % 4 == 0 && !full && !original && h->map_mode == 0 &&
           !(p.io.flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) && !p.src_boards && !p.io.final_obs_dev && !p.io.final_fobs_dev &&
           (h->rec_bytes == 128 || h->rec_bytes == 256) && (p.env_first & 63) == 0 &&
           aligned(p.io.obs_dev, 16) && aligned(p.io.mask_dev, 16) && aligned(p.io.reward_dev, 8) && aligned(p.io.actions_dev, 16);
}

static int check_step_io(sgx_env *h, const KParams &p) {
    if (p.mode == 0 && p.io.auto_reset) return check_random_setups(h);
    return SGX_OK;
}

// The output sets of a rollout call: sgx_step_n (one set, in place), sgx_step_ring (n_sets separate sets: ios[0 .. n_sets)), or the slots of
// an sgx_step_traj trajectory buffer (strided: ios[0] names slot 0, slot s lies s x the byte strides further; KParams::traj_* carry the
// strides of the per-step results and of the action log).
struct OutSets {
    const sgx_step_io *ios;
    int32_t n_sets;
    bool strided;
    int64_t obs_b, fobs_b, mask_b;
};

static int launch_step(sgx_env *h, const KParams &p_in, void *stream, int ring_sets = 1);

// ONE step of a rollout call as a launch of its own, writing output set / slot `set`: what the multi-step launches fall back to (calls
// they do not cover, the odd step a chunk leaves over).
static int launch_set_step(sgx_env *h, const KParams &p_in, const OutSets &sets, int32_t set, void *stream) {
    KParams p1 = p_in;
    p1.mode = 0;
    if (!sets.strided) {
        p1.io = sets.ios[set];
        return launch_step(h, p1, stream, sets.n_sets);
    }
    sgx_step_io io = sets.ios[0];
    auto at = [](auto *ptr, int64_t bytes) { return ptr ? reinterpret_cast<decltype(ptr)>(reinterpret_cast<char *>(ptr) + bytes) : ptr; };
    io.obs_dev = at(io.obs_dev, set * sets.obs_b);
    io.fobs_dev = at(io.fobs_dev, set * sets.fobs_b);
    io.mask_dev = at(io.mask_dev, set * sets.mask_b);
    const int64_t r = set * p_in.traj_res_envs;
    io.reward_dev = at(io.reward_dev, 8 * r);
    io.done_dev = at(io.done_dev, r);
    io.player_dev = at(io.player_dev, r);
    io.invalid_action_dev = at(io.invalid_action_dev, r);
    io.ending_invalid_dev = at(io.ending_invalid_dev, r);
    p1.io = io;
    if (int rc = launch_step(h, p1, stream, sets.n_sets)) return rc;
    if (p_in.traj_act_log && io.next_actions_dev)
        HIP_TRY(hipMemcpyAsync(p_in.traj_act_log + set * p_in.traj_out_envs, io.next_actions_dev, (size_t)h->n_envs * sizeof(int32_t), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    return SGX_OK;
}

// More separate output sets than the kernel arguments hold (sgx_step_ring with n_sets > 8, e.g. a ring of 64 sets that each came from its own
// placement search): the pointers of the sets go into a device table -- [obs x n][fobs x n][mask x n] -- that the multi-step kernels read with
// scalar loads.  Uploaded only when the ring changed; ordered on `stream` before the launch that reads it.
static int upload_ring_table(sgx_env *h, const OutSets &sets, hipStream_t stream, void ***tab_out) {
    const int n = sets.n_sets;
    if (n > h->ring_tab_cap) {
        if (h->ring_tab_ev) HIP_TRY(hipEventSynchronize(h->ring_tab_ev));
        if (h->ring_tab_dev) HIP_TRY(hipFree(h->ring_tab_dev));
        if (h->ring_tab_host) HIP_TRY(hipHostFree(h->ring_tab_host));
        h->ring_tab_dev = h->ring_tab_host = nullptr;
        h->ring_tab_cap = h->ring_tab_n = 0;
        HIP_TRY(hipMalloc((void **)&h->ring_tab_dev, (size_t)3 * n * sizeof(void *)));
        HIP_TRY(hipHostMalloc((void **)&h->ring_tab_host, (size_t)3 * n * sizeof(void *), hipHostMallocDefault));
        h->ring_tab_cap = n;
    }
    if (!h->ring_tab_ev) HIP_TRY(hipEventCreateWithFlags(&h->ring_tab_ev, hipEventDisableTiming));
    std::vector<void *> want((size_t)3 * n);
    for (int k = 0; k < n; ++k) {
        want[k] = sets.ios[k].obs_dev;
        want[n + k] = sets.ios[k].fobs_dev;
        want[2 * n + k] = sets.ios[k].mask_dev;
    }
    const bool same = h->ring_tab_n == n && memcmp(h->ring_tab_host, want.data(), want.size() * sizeof(void *)) == 0;
    if (!same) {
        if (h->ring_tab_n) HIP_TRY(hipEventSynchronize(h->ring_tab_ev));          // (the staging buffer's previous upload has run)
        memcpy(h->ring_tab_host, want.data(), want.size() * sizeof(void *));
        // (laid out for capacity n: a smaller ring after a larger one re-packs the three sections)
        HIP_TRY(hipMemcpyAsync(h->ring_tab_dev, h->ring_tab_host, want.size() * sizeof(void *), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(h->ring_tab_ev, stream));
        h->ring_tab_n = n;
        h->ring_tab_stream = stream;
    } else if (h->ring_tab_stream != stream) {
        HIP_TRY(hipStreamWaitEvent(stream, h->ring_tab_ev, 0));                     // (uploaded on another stream: order this launch behind it)
    }
    *tab_out = h->ring_tab_dev;
    return SGX_OK;
}

// A rollout call goes out in launches of at most SGX_STEPS_MAX_PER_LAUNCH steps (a workgroup should not own the chip for seconds: work on
// other streams, or a second rank sharing the GPU, gets in between the launches).
#define SGX_STEPS_MAX_PER_LAUNCH 256

// sgx_step_n / sgx_step_ring / sgx_step_traj on boards of at most 16 cells: the steps in launches of lane_steps_kernel (sgx_lane_kernel.h)
// where the call is eligible -- the lane kernel's conditions for every output set, flat perspective actions, observations wanted.
// *launched tells.
static int launch_lane_steps(sgx_env *h, const KParams &p_in, const OutSets &sets, int32_t first_set, int32_t n_steps, void *stream, bool *launched) {
    *launched = false;
    const int32_t n_sets = sets.n_sets;
    if (n_steps < 2 || h->lane_mode == 0 || h->no_multi_step) return SGX_OK;
    const bool table = !sets.strided && n_sets > KSTEP_MAX_SETS;        // more separate sets than the kernel arguments hold: a device table
    KParams p = p_in;
    p.mode = 0;
    p.io = sets.ios[sets.strided ? 0 : first_set];
    if (!p.io.obs_dev || (p.io.flags & (SGX_STEP_ACTIONS_1D | SGX_STEP_ACTIONS_POSITIONS))) return SGX_OK;
    StepsParams sp;
    memset(&sp, 0, sizeof(sp));
    if (sets.strided) {
        const bool full = p.io.fobs_dev || p.io.final_fobs_dev, original = (p.io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
        // every slot as aligned as slot 0 (the kernel's 16-byte stores), per-slot results as aligned as the results of slot 0
        if (!lane_eligible(h, p, full, original, true) || ((sets.obs_b | sets.mask_b) & 15) || (p.traj_res_envs & 1)) return SGX_OK;
        sp.strided = 1;
        sp.obs_slot_bytes = sets.obs_b;
        sp.mask_slot_bytes = sets.mask_b;
        sp.obs[0] = p.io.obs_dev;
        sp.mask[0] = p.io.mask_dev;
    } else
        for (int32_t k = 0; k < n_sets; ++k) {
            KParams pk = p;
            pk.io = sets.ios[k];
            const bool full = pk.io.fobs_dev || pk.io.final_fobs_dev, original = (pk.io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
            if (!pk.io.obs_dev || !lane_eligible(h, pk, full, original, true)) return SGX_OK;
            // everything but the observation / mask tensors is shared by the sets (the kernel writes the results through set first_set's pointers)
            if (pk.io.reward_dev != p.io.reward_dev || pk.io.done_dev != p.io.done_dev || pk.io.player_dev != p.io.player_dev ||
                pk.io.invalid_action_dev != p.io.invalid_action_dev || pk.io.ending_invalid_dev != p.io.ending_invalid_dev) return SGX_OK;
            if (!table) {
                sp.obs[k] = pk.io.obs_dev;
                sp.mask[k] = pk.io.mask_dev;
            }
        }
    if (int rc = check_step_io(h, p)) return rc;
    if (table) {
        void **tab = nullptr;
        if (int rc = upload_ring_table(h, sets, (hipStream_t)stream, &tab)) return rc;
        sp.strided = 2;
        sp.obs_tab = reinterpret_cast<float *const *>(tab);
        sp.mask_tab = reinterpret_cast<uint8_t *const *>(tab + 2 * n_sets);
    }
    p.map_mode = h->map_mode; p.map_arg = h->map_arg;
    const bool streaming = launch_streams_past_cache(h, p, n_sets);
    p.nt_stores = h->nt_mode < 0 ? (streaming ? 1 : 0) : h->nt_mode;
    int32_t skew[8];
    launch_shares(h, streaming, skew);
    sp.n_sets = n_sets;
    for (int32_t done = 0; done < n_steps; ) {
        const int32_t now = n_steps - done > SGX_STEPS_MAX_PER_LAUNCH ? SGX_STEPS_MAX_PER_LAUNCH : n_steps - done;
        const int32_t set = (int32_t)(((int64_t)first_set + done) % n_sets);
        if (now < 2) {                                                      // a single step left over: the ordinary launch
            if (int rc = launch_set_step(h, p_in, sets, set, stream)) return rc;
            done += now;
            continue;
        }
        sp.n_steps = now; sp.first_set = set;
        bool ok = false;
#define CALL_LANE_STEPS(R, C)                                                                                      \
    do {                                                                                                           \
        if constexpr (lane_geometry<Geo<R, C>>()) {                                                                \
            const unsigned grid = shares_for(p, (p.n_envs - p.env_first + 63) / 64, skew);                         \
            const size_t dyn = 2 * 64 * (size_t)(p.rec_bytes + 16);                                                \
            if (dyn + sizeof(StepsLds<Geo<R, C>>) > 64 * 1024 && !h->multi_step_attr) {                            \
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&lane_steps_kernel<R, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { \
                    (void)hipGetLastError();                                                                       \
                    h->no_multi_step = 1;                  /* this runtime keeps workgroups to 64 KiB: per-step launches */ \
                    break;                                                                                         \
                }                                                                                                  \
                h->multi_step_attr = 1;                                                                            \
            }                                                                                                      \
            sp.k = p;                                                                                              \
            lane_steps_kernel<R, C><<<grid, 64 * (1 + KSTEP_EMITTERS), dyn, (hipStream_t)stream>>>(sp);            \
            ok = true;                                                                                             \
        }                                                                                                          \
    } while (0)
        DISPATCH_GEOMETRY(h, CALL_LANE_STEPS);
#undef CALL_LANE_STEPS
        if (!ok) {
            if (done == 0) return SGX_OK;                                   // (not this board / this runtime: nothing was launched)
            return fail(SGX_EDEVICE, "lane_steps_kernel became unavailable in the middle of a call%s");
        }
        HIP_TRY(hipGetLastError());
        done += now;
    }
    *launched = true;
    h->last_kind = SGX_LAUNCH_MULTI_STEP;
    return SGX_OK;
}

// The same for the wave-per-game kernels (steps_kernel, sgx_step.h): the steps of an sgx_step_n / sgx_step_ring / sgx_step_traj call in launches
// of up to 256 steps -- the games' boards stay in LDS, the record travels once per launch, the waves drift out of phase -- for the 67-channel
// 'extended' kind, BOTH observations, compact outputs and launches without an observation; perspective actions and masks; at most 8
// separate output sets, or any number of slots of a trajectory buffer.
static int launch_wave_steps(sgx_env *h, const KParams &p_in, const OutSets &sets, int32_t first_set, int32_t n_steps, void *stream, bool *launched) {
    *launched = false;
    const int32_t n_sets = sets.n_sets;
    if (n_steps < 2 || h->no_multi_step || !h->multi_step_wave || h->map_mode != 0) return SGX_OK;
    const bool table = !sets.strided && n_sets > WSTEPS_MAX_SETS;       // more separate sets than the kernel arguments hold: a device table
    KParams p = p_in;
    p.mode = 0;
    p.io = sets.ios[sets.strided ? 0 : first_set];
    const sgx_step_io &io0 = p.io;
    if (io0.flags & (SGX_STEP_ACTIONS_1D | SGX_STEP_ACTIONS_POSITIONS | SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS | SGX_STEP_ORIGINAL_CHANNELS)) return SGX_OK;
    const bool compact = (io0.flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) != 0;
    const bool full = io0.fobs_dev || io0.final_fobs_dev;
    const bool no_obs = !io0.obs_dev && !io0.fobs_dev && !io0.final_obs_dev && !io0.final_fobs_dev && !compact;
    if (compact && (full || io0.final_obs_dev)) return SGX_OK;             // (launch_step refuses these: let it say so)
    if (compact && ((reinterpret_cast<uintptr_t>(io0.obs_dev) | reinterpret_cast<uintptr_t>(io0.mask_dev)) & 15)) return SGX_OK;
    // Logic-only rollouts (no observation pointer) on 3x4: the lane-per-game kernel, one launch per step, stays ahead of this kernel's multi-step
    // launch there -- 65,536 Micro games mask only 14.9 against 18.2 us per step, no outputs 13.4 against 15.7; 262,144 games 44 against 65 --
    // while on 4x4 the multi-step launch wins (Tiny 15.0 against 17.5 us, 262,144 games 56 against 74: tools/noobs_small_ab.py,
    // profiles/r06_noobs_small_ab.log)
    if (no_obs && h->cfg.rows * h->cfg.cols <= 12 && lane_eligible(h, p, false, false)) return SGX_OK;
    WaveStepsParams sp;
    memset(&sp, 0, sizeof(sp));
    if (sets.strided) {
        if (compact && ((sets.obs_b | sets.mask_b) & 15)) return SGX_OK;
        sp.strided = 1;
        sp.obs_slot_bytes = sets.obs_b; sp.fobs_slot_bytes = sets.fobs_b; sp.mask_slot_bytes = sets.mask_b;
        sp.obs[0] = io0.obs_dev; sp.fobs[0] = io0.fobs_dev; sp.mask[0] = io0.mask_dev;
    } else
        for (int32_t k = 0; k < n_sets; ++k) {
            const sgx_step_io &io = sets.ios[k];
            // the sets differ in their output tensors only, and every set has the tensors the first one has
            if (io.reward_dev != io0.reward_dev || io.done_dev != io0.done_dev || io.player_dev != io0.player_dev || io.invalid_action_dev != io0.invalid_action_dev ||
                io.ending_invalid_dev != io0.ending_invalid_dev || io.final_obs_dev != io0.final_obs_dev || io.final_fobs_dev != io0.final_fobs_dev ||
                (io.obs_dev == nullptr) != (io0.obs_dev == nullptr) || (io.fobs_dev == nullptr) != (io0.fobs_dev == nullptr) ||
                (io.mask_dev == nullptr) != (io0.mask_dev == nullptr)) return SGX_OK;
            if (compact && ((reinterpret_cast<uintptr_t>(io.obs_dev) | reinterpret_cast<uintptr_t>(io.mask_dev)) & 15)) return SGX_OK;
            if (!table) { sp.obs[k] = io.obs_dev; sp.fobs[k] = io.fobs_dev; sp.mask[k] = io.mask_dev; }
        }
    if (int rc = check_step_io(h, p)) return rc;
    if (table) {
        void **tab = nullptr;
        if (int rc = upload_ring_table(h, sets, (hipStream_t)stream, &tab)) return rc;
        sp.strided = 2;
        sp.obs_tab = reinterpret_cast<float *const *>(tab);
        sp.fobs_tab = reinterpret_cast<float *const *>(tab + n_sets);
        sp.mask_tab = reinterpret_cast<uint8_t *const *>(tab + 2 * n_sets);
    }
    p.map_mode = 0; p.map_arg = h->map_arg;
    const bool streaming = launch_streams_past_cache(h, p, n_sets);
    p.nt_stores = h->nt_mode < 0 ? (streaming ? 1 : 0) : h->nt_mode;
    int32_t skew[8];
    launch_shares(h, streaming, skew);
    const int kind = compact ? 4 : no_obs ? 8 : full ? 1 : 0;
    sp.n_sets = n_sets;
    // A barrier per step between the waves of a workgroup: on a long ring or trajectory buffer (the resident waves cycle through more sets than
    // the address translation caches hold pages for: DESIGN.md section 4.4) the workgroup's games then write one set at a time.  In-process A/B
    // of the two modes, drifting -> in step (tools/barrier_footprint_ab.py; profiles/r06_barrier_boards_ab.log, r06_barrier_footprint_ab.log):
    //   10x10  trajectory buffer of 12 / 16 / 64 slots -1.1 / -1.7 / -4.1 %, rings of 12 / 16 / 24 / 32 / 64 separate sets -1.1 / -1.9 / -2.9 / -3.3 /
    //          -4.2 %, ring of 3 (tuned buffers) 243.6 -> 242.1 us
    //   6x6    16 / 32 / 64 slots -4.2 / -6.3 / -9.3 %, rings of 12 / 24 / 32 / 64 sets +1.7 / +0.1 / -7.2 / -7.3 %
    //   8x8    16 / 32 / 64 slots -2.2 / -2.4 / -2.8 %, rings of 12 ... 64 sets 0.0 ... -0.9 %
    //   15x15  16 / 32 / 64 slots -2.0 / -2.3 / -2.4 %, rings of 24 ... 64 sets +1.0 ... -0.2 %, of 8 / 12 sets +2 ... +3 %
    //   5x5 (two games per wave) +3 ... +9 % on rings, -4 / +6 % on 64 slots by batch size;  launches that render no float32 observation
    //   (compact, mask-only, logic-only) +3 ... +8 %
    // so: float32 observations, one game per wave (36 .. 225 cells), from 9 sets / slots on 10x10 and from 16 elsewhere.  Never on a launch with a
    // partly empty last workgroup (its missing waves have left the kernel).
    const int cells = h->cfg.rows * h->cfg.cols;
    const bool pays = (kind == 0 || kind == 1) && cells >= 36 && cells <= 225 && n_sets >= (cells == 100 ? WSTEPS_MAX_SETS + 1 : 16);
    const bool want_barrier = h->steps_barrier > 0 || (h->steps_barrier < 0 && pays);
    for (int32_t done = 0; done < n_steps; ) {
        const int32_t now = n_steps - done > SGX_STEPS_MAX_PER_LAUNCH ? SGX_STEPS_MAX_PER_LAUNCH : n_steps - done;
        const int32_t set = (int32_t)(((int64_t)first_set + done) % n_sets);
        if (now < 2) {                                                      // a single step left over: the ordinary launch
            if (int rc = launch_set_step(h, p_in, sets, set, stream)) return rc;
            done += now;
            continue;
        }
        sp.n_steps = now;
        sp.first_set = set;
#define CALL_WSTEPS_K(R, C, KIND)                                                                          \
    do {                                                                                                   \
        using G_ = Geo<R, C>;                                                                              \
        const unsigned grid = shares_for(p, (p.n_envs - p.env_first + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW), skew); \
        sp.k = p;                                                                                          \
        sp.barrier = want_barrier && (p.n_envs - p.env_first) % (G_::WPB * G_::GPW) == 0;                  \
        steps_kernel<R, C, KIND><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(sp);                      \
    } while (0)
#define CALL_WSTEPS(R, C)                                                                                  \
    do {                                                                                                   \
        if (kind == 0) CALL_WSTEPS_K(R, C, 0);                                                             \
        else if (kind == 1) CALL_WSTEPS_K(R, C, 1);                                                        \
        else {                                                                                             \
            bool half_ = false;                                                                            \
            if constexpr (half_wave_ok<R, C>()) {                                                          \
                if (h->half_wave && kind == 8) {             /* two games per wave (Geo<R, C, 2>) */        \
                    using H_ = Geo<R, C, 2>;                                                               \
                    const unsigned grid = shares_for(p, (p.n_envs - p.env_first + H_::WPB * H_::GPW - 1) / (H_::WPB * H_::GPW), skew); \
                    sp.k = p;                                                                              \
                    sp.barrier = want_barrier && (p.n_envs - p.env_first) % (H_::WPB * H_::GPW) == 0;      \
                    steps_kernel<R, C, 8, 2><<<grid, 64 * H_::WPB, 0, (hipStream_t)stream>>>(sp);           \
                    half_ = true;                                                                          \
                }                                                                                          \
            }                                                                                              \
            if (!half_) { if (kind == 4) CALL_WSTEPS_K(R, C, 4); else CALL_WSTEPS_K(R, C, 8); }            \
        }                                                                                                  \
    } while (0)
        DISPATCH_GEOMETRY(h, CALL_WSTEPS);
#undef CALL_WSTEPS
#undef CALL_WSTEPS_K
        HIP_TRY(hipGetLastError());
        done += now;
    }
    *launched = true;
    h->last_kind = SGX_LAUNCH_MULTI_STEP_WAVE;
    return SGX_OK;
}

static int launch_step(sgx_env *h, const KParams &p_in, void *stream, int ring_sets) {
    KParams p = p_in;
    if (p.mode == 0 && p.io.auto_reset)
        if (int rc = check_random_setups(h)) return rc;
    p.map_mode = h->map_mode; p.map_arg = h->map_arg;
    const bool streaming = launch_streams_past_cache(h, p, ring_sets);
    p.nt_stores = h->nt_mode < 0 ? (streaming ? 1 : 0) : h->nt_mode;
    int32_t skew[8];                     // unequal XCD shares (sgx_layout.h: group_of_block)
    launch_shares(h, streaming, skew);
    const bool full = p.io.fobs_dev || p.io.final_fobs_dev, original = (p.io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
    if (p.io.flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) {
        // compact outputs: the 67-channel 'extended' observation as 4-bit codes / the mover's-perspective mask as bits, nothing else
        if (full || original || p.io.final_obs_dev || p.src_boards || (p.io.flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)))
            return fail(SGX_EINVAL, "compact outputs come with the 67-channel partial observation and the perspective mask only (no fobs / final_obs / original channels / state-coordinate masks)%s");
        if ((reinterpret_cast<uintptr_t>(p.io.obs_dev) | reinterpret_cast<uintptr_t>(p.io.mask_dev)) & 15)
            return fail(SGX_EINVAL, "compact outputs need 16-byte aligned obs_dev / mask_dev%s");
    }
    if (lane_eligible(h, p, full, original)) {
        // boards of at most 16 cells: one game per lane, 64 games per wave (sgx_lane_kernel.h)
#define CALL_LANE(R, C)                                                                                    \
    do {                                                                                                   \
        if constexpr (lane_geometry<Geo<R, C>>()) {                                                        \
            const unsigned grid = shares_for(p, (p.n_envs - p.env_first + 63) / 64, skew);                 \
            const size_t dyn = 64 * (size_t)(p.rec_bytes + 16);                                            \
            if (p.mode) lane_kernel<R, C, true><<<grid, 64, dyn, (hipStream_t)stream>>>(p);                \
            else lane_kernel<R, C, false><<<grid, 64, dyn, (hipStream_t)stream>>>(p);                      \
        }                                                                                                  \
    } while (0)
        DISPATCH_GEOMETRY(h, CALL_LANE);
#undef CALL_LANE
        HIP_TRY(hipGetLastError());
        h->last_kind = SGX_LAUNCH_LANE;
        return SGX_OK;
    }
#define CALL_STEP_KIND(R, C, KIND)                                                                 \
    do {                                                                                           \
        using G_ = Geo<R, C>;                                                                      \
        const unsigned grid = shares_for(p, (p.n_envs - p.env_first + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW), skew); \
        if (p.mode) observe_kernel<R, C, KIND><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p); \
        else step_kernel<R, C, KIND><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p);           \
    } while (0)
#define CALL_STEP0(R, C) CALL_STEP_KIND(R, C, 0)
#define CALL_STEP4(R, C) CALL_STEP_KIND(R, C, 4)
#define CALL_STEP1(R, C) CALL_STEP_KIND(R, C, 1)
#define CALL_STEP2(R, C) CALL_STEP_KIND(R, C, 2)
#define CALL_STEP3(R, C) CALL_STEP_KIND(R, C, 3)
#define CALL_STEP8(R, C)                                                                           \
    do {                                                                                           \
        bool half_ = false;                                                                        \
        if constexpr (half_wave_ok<R, C>()) {                                                      \
            if (h->half_wave) {                              /* two games per wave (Geo<R, C, 2>) */ \
                using H_ = Geo<R, C, 2>;                                                           \
                const unsigned grid = shares_for(p, (p.n_envs - p.env_first + H_::WPB * H_::GPW - 1) / (H_::WPB * H_::GPW), skew); \
                if (p.mode) observe_kernel<R, C, 8, false, 2><<<grid, 64 * H_::WPB, 0, (hipStream_t)stream>>>(p); \
                else step_kernel<R, C, 8, false, 2><<<grid, 64 * H_::WPB, 0, (hipStream_t)stream>>>(p); \
                half_ = true;                                                                      \
            }                                                                                      \
        }                                                                                          \
        if (!half_) CALL_STEP_KIND(R, C, 8);                                                       \
    } while (0)
    // no observation pointer at all (search expansions, mask-only steps, logic-only rollouts): the kind without observation tables
    const bool no_obs = !p.io.obs_dev && !p.io.fobs_dev && !p.io.final_obs_dev && !p.io.final_fobs_dev &&
                        !(p.io.flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK));
    if ((p.io.flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) || p.src_boards) {
        if (full || (original && !no_obs)) return fail(SGX_EINVAL, "state-coordinate masks and sgx_expand come with the 67-channel partial observation only%s");
#define CALL_STEP_MAPPED8(R, C)                                                                    \
    do {                                                                                           \
        bool half_ = false;                                                                        \
        if constexpr (half_wave_ok<R, C>()) {                                                      \
            if (h->half_wave) {                              /* two games per wave (Geo<R, C, 2>) */ \
                using H_ = Geo<R, C, 2>;                                                           \
                const unsigned grid = shares_for(p, (p.n_envs - p.env_first + H_::WPB * H_::GPW - 1) / (H_::WPB * H_::GPW), skew); \
                if (p.mode) observe_kernel<R, C, 8, true, 2><<<grid, 64 * H_::WPB, 0, (hipStream_t)stream>>>(p); \
                else step_kernel<R, C, 8, true, 2><<<grid, 64 * H_::WPB, 0, (hipStream_t)stream>>>(p); \
                half_ = true;                                                                      \
            }                                                                                      \
        }                                                                                          \
        if (half_) break;                                                                          \
        using G_ = Geo<R, C>;                                                                      \
        const unsigned grid = shares_for(p, (p.n_envs - p.env_first + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW), skew); \
        if (p.mode) observe_kernel<R, C, 8, true><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p);  \
        else step_kernel<R, C, 8, true><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p);        \
    } while (0)
#define CALL_STEP_MAPPED(R, C)                                                                     \
    do {                                                                                           \
        using G_ = Geo<R, C>;                                                                      \
        const unsigned grid = shares_for(p, (p.n_envs - p.env_first + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW), skew); \
        if (p.mode) observe_kernel<R, C, 0, true><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p);  \
        else step_kernel<R, C, 0, true><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(p);        \
    } while (0)
        if (no_obs) DISPATCH_GEOMETRY(h, CALL_STEP_MAPPED8);
        else DISPATCH_GEOMETRY(h, CALL_STEP_MAPPED);
#undef CALL_STEP_MAPPED
#undef CALL_STEP_MAPPED8
    } else if (p.io.flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) DISPATCH_GEOMETRY(h, CALL_STEP4);
    else if (no_obs) DISPATCH_GEOMETRY(h, CALL_STEP8);
    else if (!original && !full) DISPATCH_GEOMETRY(h, CALL_STEP0);
    else if (!original) DISPATCH_GEOMETRY(h, CALL_STEP1);
    else if (!full) DISPATCH_GEOMETRY(h, CALL_STEP2);
    else DISPATCH_GEOMETRY(h, CALL_STEP3);
#undef CALL_STEP0
#undef CALL_STEP4
#undef CALL_STEP1
#undef CALL_STEP2
#undef CALL_STEP3
#undef CALL_STEP8
#undef CALL_STEP_KIND
    HIP_TRY(hipGetLastError());
    h->last_kind = SGX_LAUNCH_WAVE;
    return SGX_OK;
}

SGX_API int sgx_observe(sgx_env *h, float *obs_dev, float *fobs_dev, uint8_t *mask_dev, int8_t *player_dev, int32_t flags, void *stream) {
    if (!h) return fail(SGX_EINVAL, "handle is NULL%s");
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
    p.mode = 1;
    p.io.obs_dev = obs_dev;
    p.io.fobs_dev = fobs_dev;
    p.io.flags = flags & (SGX_STEP_RAW_OBS | SGX_STEP_ORIGINAL_CHANNELS | SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS | SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK);
    p.io.mask_dev = mask_dev;
    p.io.player_dev = player_dev;
    return launch_step(h, p, stream);
}

SGX_API int sgx_time_observe(sgx_env *h, float *obs_dev, uint8_t *mask_dev, int32_t launches, void *stream, float *microseconds) {
    if (!h || !microseconds || launches <= 0) return fail(SGX_EINVAL, "bad argument%s");
    SGX_ON_DEVICE(h->device);
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

// ---- device-memory probe
namespace {
// GB/s of `launches` probe launches over [ptr, ptr + bytes) (one untimed first touch); e0 / e1: scratch events
int probe_range(void *ptr, int64_t bytes, int32_t launches, hipStream_t stream, hipEvent_t e0, hipEvent_t e1, float *gbps) {
    const int64_t n_seg = bytes / PROBE_SEG;
    if (n_seg < 1) return fail(SGX_EINVAL, "sgx_mem_probe: range shorter than one 26 KiB segment%s");
    const int64_t want = ((int64_t)1 << 30) / PROBE_SEG;                      // at least ~1 GiB of stores per launch
    const int64_t passes = n_seg >= want ? 1 : (want + n_seg - 1) / n_seg, n_waves = n_seg * passes;
    const unsigned grid = (unsigned)((((n_waves + 7) / 8) + 7) & ~(int64_t)7);
    mem_probe_kernel<<<grid, 512, 0, stream>>>((char *)ptr, n_seg, n_waves);
    HIP_TRY(hipEventRecord(e0, stream));
    for (int i = 0; i < launches; ++i) mem_probe_kernel<<<grid, 512, 0, stream>>>((char *)ptr, n_seg, n_waves);
    HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *gbps = (float)((double)n_waves * PROBE_SEG * launches / (ms * 1e-3) / 1e9);
    return SGX_OK;
}
}  // namespace

SGX_API int sgx_mem_probe(int device, void *ptr_dev, int64_t bytes, int32_t launches, void *stream, float *gb_per_s) {
    if (!ptr_dev || !gb_per_s || launches <= 0 || bytes <= 0) return fail(SGX_EINVAL, "sgx_mem_probe: bad argument%s");
    if ((reinterpret_cast<uintptr_t>(ptr_dev) & 1023) != 0) return fail(SGX_EINVAL, "sgx_mem_probe: the range must start on a 1 KiB boundary%s");
    SGX_ON_DEVICE(device);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int rc = probe_range(ptr_dev, bytes, launches, (hipStream_t)stream, e0, e1, gb_per_s);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// The step kernel's store stream without the game (sgx_mem.h: store_probe_kernel): what the memory takes from exactly this store shape.
SGX_API int sgx_store_probe(int device, void *ptr_dev, int64_t bytes, int32_t seg_bytes, int32_t passes, int32_t payload, int32_t nt_stores,
                            int32_t waves_per_cu, int32_t pace, int32_t persistent, int32_t dwell, int32_t ring, int32_t launches, void *stream,
                            float *microseconds_per_launch, float *gb_per_s) {
    if (!ptr_dev || !microseconds_per_launch || !gb_per_s || launches <= 0 || passes <= 0 || bytes <= 0) return fail(SGX_EINVAL, "sgx_store_probe: bad argument%s");
    if (seg_bytes < 16 || (seg_bytes & 15) || (reinterpret_cast<uintptr_t>(ptr_dev) & 15)) return fail(SGX_EINVAL, "sgx_store_probe: segments and the range are 16-byte aligned%s");
    if (payload < 0 || payload > 2) return fail(SGX_EINVAL, "sgx_store_probe: payload 0 (zeros), 1 (observation-like) or 2 (random bits)%s");
    if (pace < 0 || pace > 4096) return fail(SGX_EINVAL, "sgx_store_probe: pace 0 .. 4096%s");
    if (waves_per_cu != 0 && waves_per_cu != 8 && waves_per_cu != 16 && waves_per_cu != 24) return fail(SGX_EINVAL, "sgx_store_probe: waves_per_cu 0 (= 24), 8, 16 or 24%s");
    if (dwell < 1 || ring < 1 || ring > 8 || (dwell > 1 && !persistent)) return fail(SGX_EINVAL, "sgx_store_probe: dwell >= 1 (> 1 with persistent waves only), ring 1 .. 8%s");
    // `ring` sub-ranges of equal size (whole segments, 1 KiB aligned starts); a pass covers ONE sub-range's segments
    const int64_t ring_bytes = ring > 1 ? ((bytes / ring) & ~(int64_t)1023) : 0;
    const int64_t n_seg = (((ring > 1 ? ring_bytes : bytes) / seg_bytes)) & ~(int64_t)63;   // eight waves per workgroup, eight XCD shares
    if (n_seg < 64) return fail(SGX_EINVAL, "sgx_store_probe: the range holds fewer than 64 segments%s");
    const int64_t groups_per_pass = n_seg / 8;
    int64_t grid = groups_per_pass * passes;
    if (grid > 0x7fffffff) return fail(SGX_EINVAL, "sgx_store_probe: passes x segments exceed one grid%s");
    SGX_ON_DEVICE(device);
    if (persistent) {                                                          // the resident workgroups only: CUs x (waves per CU / 8)
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        const int64_t resident = (int64_t)prop.multiProcessorCount * ((waves_per_cu ? waves_per_cu : 24) / 8);
        if (resident < grid) grid = resident & ~(int64_t)7;
    }
    hipStream_t st = (hipStream_t)stream;
    // resident workgroups per CU through unused dynamic LDS: 160 KiB per CU / (what one workgroup asks for)
    const size_t dyn = waves_per_cu == 8 ? 100 * 1024 : waves_per_cu == 16 ? 60 * 1024 : 0;
    if (dyn > 64 * 1024) {
        hipError_t ea = hipSuccess;
        if (payload == 0) ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&store_probe_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        else if (payload == 1) ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&store_probe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        else ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&store_probe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ea != hipSuccess) return fail(SGX_EDEVICE, "sgx_store_probe: %s", hipGetErrorString(ea));
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    static uint32_t salt = 0x5EED5EEDu;
    auto go = [&]() {
        salt = salt * 1664525u + 1013904223u;
        if (payload == 0) store_probe_kernel<0><<<(unsigned)grid, 512, dyn, st>>>((char *)ptr_dev, groups_per_pass, passes, seg_bytes, nt_stores, salt, pace, persistent ? 1 : 0, dwell, ring, ring_bytes);
        else if (payload == 1) store_probe_kernel<1><<<(unsigned)grid, 512, dyn, st>>>((char *)ptr_dev, groups_per_pass, passes, seg_bytes, nt_stores, salt, pace, persistent ? 1 : 0, dwell, ring, ring_bytes);
        else store_probe_kernel<2><<<(unsigned)grid, 512, dyn, st>>>((char *)ptr_dev, groups_per_pass, passes, seg_bytes, nt_stores, salt, pace, persistent ? 1 : 0, dwell, ring, ring_bytes);
    };
    go();                                                                      // untimed first touch
    hipError_t e = hipEventRecord(e0, st);
    for (int32_t i = 0; i < launches; ++i) go();
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(SGX_EDEVICE, "sgx_store_probe: %s", hipGetErrorString(e));
    *microseconds_per_launch = ms * 1000.f / (float)launches;
    *gb_per_s = (float)((double)n_seg * seg_bytes * passes * dwell * launches / (ms * 1e-3) / 1e9);
    return SGX_OK;
}

// ---- library-owned output buffers with a bounded placement trial
namespace {

struct TrialCtx {
    sgx_env *h;
    void *stream;
    int32_t flags;
    hipEvent_t e0, e1;
};

// average launch time of 6 sgx_observe launches writing the given buffers (one untimed first touch)
int time_candidate(TrialCtx &c, float *obs, float *fobs, uint8_t *mask, float *us) {
    const int launches = 6;
    int rc = sgx_observe(c.h, obs, fobs, mask, nullptr, c.flags, c.stream);
    if (rc != SGX_OK) return rc;
    HIP_TRY(hipEventRecord(c.e0, (hipStream_t)c.stream));
    for (int i = 0; i < launches && rc == SGX_OK; ++i) rc = sgx_observe(c.h, obs, fobs, mask, nullptr, c.flags, c.stream);
    if (rc != SGX_OK) return rc;
    HIP_TRY(hipEventRecord(c.e1, (hipStream_t)c.stream));
    HIP_TRY(hipEventSynchronize(c.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c.e0, c.e1));
    *us = ms * 1000.f / (float)launches;
    return SGX_OK;
}

// Picks the fastest of up to max_trials allocations of `bytes`.  Candidate k > 0 is allocated behind a padding allocation of
// k * step bytes, which is released again before the candidate is timed, so that at most (candidate + padding) is held beyond
// the buffer that is kept.  `which` = 0: the candidate is the partial observation buffer, 1: the fully-observable one.
int pick_buffer(TrialCtx &c, int which, size_t bytes, float *obs_fixed, uint8_t *mask, int64_t max_extra, int32_t max_trials,
                float **best_out, float *trial_us, int32_t *n_trials, int64_t *peak_extra) {
    float *best = nullptr;
    if (hipMalloc((void **)&best, bytes) != hipSuccess) return fail(SGX_ENOMEM, "device allocation failed (output buffer)%s");
    *best_out = best;
    *n_trials = 0;
    const bool can_try = max_trials > 1 && max_extra >= (int64_t)bytes;
    if (!can_try) return SGX_OK;
    float best_us = 0.f;
    int rc = time_candidate(c, which ? obs_fixed : best, which ? best : nullptr, mask, &best_us);
    if (rc != SGX_OK) return rc;
    trial_us[0] = best_us;
    *n_trials = 1;
    const float target = which == 0 ? c.h->placement_target_us : 0.f;      // (the target is a partial-observation launch time)
    if (target > 0.f && best_us <= 1.03f * target) return SGX_OK;
    const size_t room = (size_t)max_extra - bytes;                        // what the padding may take
    size_t step = bytes / 8;                                              // (a big buffer leaves little room: smaller steps then)
    if (step * 12 > room) step = room / 12;
    // a generous budget is spread over all of it: with max_extra_bytes far beyond 32 x bytes / 8 the candidates sample the whole range
    const int32_t n_cand = (max_trials < SGX_OUT_MAX_TRIALS ? max_trials : SGX_OUT_MAX_TRIALS) - 1;
    const bool spread = n_cand > 0 && room / (size_t)n_cand > step;
    if (spread) step = room / (size_t)n_cand;
    step = (step + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (spread && step > ((size_t)2 << 20) && step * (size_t)n_cand > room) step -= (size_t)2 << 20;      // (the last candidate must fit too)
    int first_good = -1;
    for (int k = 1; k < max_trials && k < SGX_OUT_MAX_TRIALS; ++k) {
        const size_t pad_bytes = (size_t)k * step;
        if (pad_bytes > room) break;
        void *pad = nullptr;
        float *cand = nullptr;
        if (hipMalloc(&pad, pad_bytes) != hipSuccess) { (void)hipGetLastError(); break; }         // memory is short: settle
        if (hipMalloc((void **)&cand, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(pad); break; }
        if ((int64_t)(pad_bytes + bytes) > *peak_extra) *peak_extra = (int64_t)(pad_bytes + bytes);
        (void)hipFree(pad);
        float us = 0.f;
        rc = time_candidate(c, which ? obs_fixed : cand, which ? cand : nullptr, mask, &us);
        if (rc != SGX_OK) { (void)hipFree(cand); return rc; }
        trial_us[k] = us;
        *n_trials = k + 1;
        if (us < best_us) { (void)hipFree(best); best = cand; best_us = us; *best_out = best; }
        else (void)hipFree(cand);
        // Early stop: the classes lie >= 10 % apart; once the kept candidate beats the slowest one seen by
        // that much the fast class has been found and more candidates would only cost start-up time.  (A run of equally slow
        // candidates is no reason to stop: fast memory was found behind six and more slow candidates on several boxes.)
        // Two steps: the classes are ~255-265 / 300-310 / 320-330 / ~345 us (65,536 Barrage games): a candidate 9 % below the slowest may
        // be the middle class only, so the search goes on for up to eight more candidates and stops at once at 17 % (fast class: 272-277 us after 335-340 us;
        // 282 us, 16 %, is not the best a box has).
        if (target > 0.f) {                           // a ring of output sets: every set as fast as the first one (sgx_set_placement_target)
            if (best_us <= 1.03f * target) break;
            continue;
        }
        float worst = 0.f;
        for (int j = 0; j <= k; ++j) worst = trial_us[j] > worst ? trial_us[j] : worst;
        if (k >= 2 && best_us < 0.83f * worst) break;
        if (k >= 2 && best_us < 0.91f * worst && first_good < 0) first_good = k;
        if (first_good >= 0 && k - first_good >= 8) break;
    }
    return SGX_OK;
}

}  // namespace

SGX_API int sgx_free_outputs(sgx_env *h, sgx_outputs *out) {
    if (!out) return fail(SGX_EINVAL, "NULL argument%s");
    if (!out->obs_dev && !out->fobs_dev && !out->mask_dev) return SGX_OK;
    SGX_ON_DEVICE(h ? h->device : out->device);
    HIP_TRY(hipDeviceSynchronize());
    if (out->obs_dev) (void)hipFree(out->obs_dev);
    if (out->fobs_dev) (void)hipFree(out->fobs_dev);
    if (out->mask_dev) (void)hipFree(out->mask_dev);
    out->obs_dev = out->fobs_dev = nullptr;
    out->mask_dev = nullptr;
    return SGX_OK;
}

SGX_API int sgx_alloc_outputs(sgx_env *h, int32_t flags, int64_t max_extra_bytes, int32_t max_trials, void *stream, sgx_outputs *out) {
    if (!h || !out) return fail(SGX_EINVAL, "NULL argument%s");
    if (flags & ~(SGX_OUT_FULL_OBS | SGX_STEP_ORIGINAL_CHANNELS)) return fail(SGX_EINVAL, "sgx_alloc_outputs: unknown flag%s");
    memset(out, 0, sizeof(*out));
    out->device = h->device;
    SGX_ON_DEVICE(h->device);
    const bool original = (flags & SGX_STEP_ORIGINAL_CHANNELS) != 0, full = (flags & SGX_OUT_FULL_OBS) != 0;
    const int64_t cells = (int64_t)h->cfg.rows * h->cfg.cols;
    out->obs_bytes = h->n_envs * cells * lut_channels(false, original) * 4;
    out->fobs_bytes = full ? h->n_envs * cells * lut_channels(true, original) * 4 : 0;
    out->mask_bytes = h->n_envs * cells * h->K;
    if (hipMalloc((void **)&out->mask_dev, (size_t)out->mask_bytes) != hipSuccess) return fail(SGX_ENOMEM, "device allocation failed (mask buffer)%s");
    TrialCtx c{h, stream, flags & SGX_STEP_ORIGINAL_CHANNELS, nullptr, nullptr};
    int rc = SGX_OK;
    if (hipEventCreate(&c.e0) != hipSuccess || hipEventCreate(&c.e1) != hipSuccess) rc = fail(SGX_EDEVICE, "hipEventCreate failed%s");
    if (rc == SGX_OK)
        rc = pick_buffer(c, 0, (size_t)out->obs_bytes, nullptr, out->mask_dev, max_extra_bytes, max_trials, &out->obs_dev, out->trial_us,
                         &out->n_trials, &out->peak_extra_bytes);
    if (rc == SGX_OK && full)
        rc = pick_buffer(c, 1, (size_t)out->fobs_bytes, out->obs_dev, out->mask_dev, max_extra_bytes, max_trials, &out->fobs_dev, out->ftrial_us,
                         &out->n_ftrials, &out->peak_extra_bytes);
    if (c.e0) (void)hipEventDestroy(c.e0);
    if (c.e1) (void)hipEventDestroy(c.e1);
    if (rc != SGX_OK) {
        const std::string keep = g_last_error;
        (void)sgx_free_outputs(h, out);
        g_last_error = keep;
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    return SGX_OK;
}

SGX_API int sgx_step(sgx_env *h, const sgx_step_io *io, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev) return fail(SGX_EINVAL, "actions_dev is NULL%s");
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
    p.mode = 0;
    p.io = *io;
    return launch_step(h, p, stream);
}

// ---- single-game latency (the N = 1 facade): outputs in device-addressable pinned host memory, one call per env.step()
SGX_API int sgx_host_alloc(sgx_env *h, int64_t bytes, void **host_ptr, void **dev_ptr) {
    if (!h || !host_ptr || !dev_ptr || bytes <= 0) return fail(SGX_EINVAL, "sgx_host_alloc: bad argument%s");
    SGX_ON_DEVICE(h->device);
    void *p = nullptr, *d = nullptr;
    // fine-grained (coherent) whatever HIP_HOST_COHERENT says: sgx_step_sync's polled path returns as soon as the kernel has published
    // its completion word, usually before the kernel has retired -- the outputs in this slab must be visible to the host at that
    // moment, which coarse-grained host memory guarantees only at the end of the kernel
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return fail(SGX_ENOMEM, "pinned host allocation failed%s"); }
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return fail(SGX_EDEVICE, "hipHostGetDevicePointer failed%s"); }
    memset(p, 0, (size_t)bytes);
    *host_ptr = p;
    *dev_ptr = d;
    return SGX_OK;
}

SGX_API int sgx_host_free(sgx_env *h, void *host_ptr) {
    if (!host_ptr) return SGX_OK;
    int dev = 0;
    if (h) dev = h->device;
    else HIP_TRY(hipGetDevice(&dev));
    SGX_ON_DEVICE(dev);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipHostFree(host_ptr));
    return SGX_OK;
}

// env.step() and wait for its outputs.  A handful of games on a 4-aligned one-game-per-wave board in an 'extended' channel mode (the
// N = 1 facade) run as single_kernel -- one workgroup per game, completion published in a host-mapped word this call polls; anything
// else is sgx_step + hipStreamSynchronize.
#define SGX_SINGLE_MAX_ENVS 8
namespace {
int step_single(sgx_env *h, const KParams &p, bool full, hipStream_t stream, bool *launched) {
    *launched = false;
    if (!h->sync_count) {
        // all three resources into locals, published to the handle only when every step has succeeded (a half-initialised handle would
        // launch single_kernel with a NULL workgroup counter on the next call)
        void *hp = nullptr, *dp = nullptr;
        uint32_t *cnt = nullptr;
        hipError_t e = hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent);   // (fine-grained whatever HIP_HOST_COHERENT says: the kernel's release store must be visible while it still runs)
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, hp, 0);
        if (e == hipSuccess) { memset(hp, 0, 64); e = hipMalloc((void **)&cnt, 64); }
        if (e == hipSuccess) e = hipMemset(cnt, 0, 64);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (cnt) (void)hipFree(cnt);
            if (hp) (void)hipHostFree(hp);
            return fail(SGX_EDEVICE, "sgx_step_sync: completion word: %s", hipGetErrorString(e));
        }
        h->sync_flag_host = (uint32_t *)hp;
        h->sync_flag_dev = (uint32_t *)dp;
        h->sync_count = cnt;
    }
    const uint32_t seq = ++h->sync_seq;
#define CALL_SINGLE(R, C)                                                                                              \
    do {                                                                                                               \
        if constexpr (Geo<R, C>::LPG == 64 && !Geo<R, C>::WIDE && (R * C) % 4 == 0) {                                  \
            if (full) single_kernel<R, C, 1><<<(unsigned)h->n_envs, 64 * SINGLE_WAVES, 0, stream>>>(p, h->sync_count, h->sync_flag_dev, seq); \
            else single_kernel<R, C, 0><<<(unsigned)h->n_envs, 64 * SINGLE_WAVES, 0, stream>>>(p, h->sync_count, h->sync_flag_dev, seq);      \
            *launched = true;                                                                                          \
        }                                                                                                              \
    } while (0)
    DISPATCH_GEOMETRY(h, CALL_SINGLE);
#undef CALL_SINGLE
    if (!*launched) return SGX_OK;
    HIP_TRY(hipGetLastError());
    // poll the word the last workgroup writes after a system-scope fence; look at the stream now and then, so that a launch that
    // failed on the device ends the wait with its error instead of hanging
    volatile uint32_t *flag = h->sync_flag_host;
    // a step is ~6 us of kernel: spin with `pause` for the first ~2^14 polls (about a millisecond); then yield the core between polls
    // for ~4,000 polls; then BACK OFF -- sleep between polls, 2 us doubling up to 64 us -- so that a long wait (a queue backed up behind
    // other work, a device that hangs) costs next to nothing on the host.  From the yielding phase on the stream is looked at every 256th
    // poll (every poll once sleeping), so that a launch that failed on the device ends the wait with its error instead of hanging.
    for (uint32_t spins = 0; *flag != seq; ++spins) {
        const bool slow = spins >= (1u << 14), sleeping = spins >= (1u << 14) + 4096u;
        if (sleeping) {
            const uint32_t k = (spins - ((1u << 14) + 4096u)) >> 4;                  // 16 polls per back-off level
            struct timespec ts = {0, (long)(2000u << (k < 5u ? k : 5u))};
            nanosleep(&ts, nullptr);
        } else if (slow) sched_yield();
        else __builtin_ia32_pause();
        if (sleeping || (slow && (spins & 0xFF) == 0)) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) break;                          // the kernel has retired: its stores are visible
            if (q != hipErrorNotReady) return fail(SGX_EDEVICE, "sgx_step_sync: %s", hipGetErrorString(q));
            (void)hipGetLastError();
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return SGX_OK;
}
}  // namespace

SGX_API int sgx_step_sync(sgx_env *h, const sgx_step_io *io, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev) return fail(SGX_EINVAL, "actions_dev is NULL%s");
    const bool original = (io->flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
    if (h->n_envs <= SGX_SINGLE_MAX_ENVS && !original && !(io->flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS | SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) && h->map_mode == 0 &&
        !h->no_single) {
        SGX_ON_DEVICE(h->device);
        KParams p = make_params(h);
        p.mode = 0;
        p.io = *io;
        if (int rc = check_step_io(h, p)) return rc;
        bool launched = false;
        if (int rc = step_single(h, p, io->fobs_dev || io->final_fobs_dev, (hipStream_t)stream, &launched)) return rc;
        if (launched) return SGX_OK;
    }
    if (int rc = sgx_step(h, io, stream)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return SGX_OK;
}

// Joins the chains' streams into the caller's: records every chain's join event and makes `stream` wait for it.  Returns `rc` (the
// error of the launches, which wins) or the first error of the joining itself.
static int join_chains(sgx_env *h, int chains, hipStream_t stream, int rc) {
    const std::string keep = g_last_error;
    hipError_t first = hipSuccess;
    for (int c = 0; c < chains; ++c) {
        hipError_t e = hipEventRecord(h->chain_join[c], h->chain_stream[c]);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, h->chain_join[c], 0);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    if (rc != SGX_OK) { g_last_error = keep; return rc; }
    if (first != hipSuccess) return fail(SGX_EDEVICE, "joining the chains: %s", hipGetErrorString(first));
    return SGX_OK;
}

SGX_API int sgx_step_n(sgx_env *h, const sgx_step_io *io, int32_t n_steps, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev || io->next_actions_dev != io->actions_dev)
        return fail(SGX_EINVAL, "sgx_step_n needs next_actions_dev == actions_dev (each step plays the action the previous one drew)%s");
    if (n_steps < 0) return fail(SGX_EINVAL, "n_steps is negative%s");
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
    p.mode = 0;
    p.io = *io;
    {   // boards of at most 16 cells: the n_steps in one launch, the games in registers (lane_steps_kernel)
        bool launched = false;
        const OutSets one{io, 1, false, 0, 0, 0};
        if (int rc = launch_lane_steps(h, p, one, 0, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
        if (int rc = launch_wave_steps(h, p, one, 0, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
    }
    for (int32_t i = 0; i < n_steps; ++i)
        if (int rc = launch_step(h, p, stream)) return rc;
    return SGX_OK;
}

SGX_API int sgx_step_ring(sgx_env *h, const sgx_step_io *ios, int32_t n_sets, int32_t first_set, int32_t n_steps, void *stream) {
    if (!h || !ios) return fail(SGX_EINVAL, "handle or ios is NULL%s");
    if (n_sets < 1 || first_set < 0 || first_set >= n_sets || n_steps < 0) return fail(SGX_EINVAL, "sgx_step_ring: n_sets, first_set or n_steps out of range%s");
    for (int32_t k = 0; k < n_sets; ++k) {
        if (!ios[k].actions_dev || ios[k].next_actions_dev != ios[k].actions_dev || ios[k].actions_dev != ios[0].actions_dev)
            return fail(SGX_EINVAL, "sgx_step_ring: every set needs next_actions_dev == actions_dev == the first set's (each step plays the action the previous one drew)%s");
        if (ios[k].auto_reset != ios[0].auto_reset || ios[k].flags != ios[0].flags)
            return fail(SGX_EINVAL, "sgx_step_ring: the sets differ in auto_reset or flags%s");
    }
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
    p.mode = 0;
    {
        bool launched = false;
        const OutSets ring{ios, n_sets, false, 0, 0, 0};
        if (int rc = launch_lane_steps(h, p, ring, first_set, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
        if (int rc = launch_wave_steps(h, p, ring, first_set, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
    }
    for (int32_t i = 0; i < n_steps; ++i) {
        p.io = ios[(first_set + i) % n_sets];
        if (int rc = launch_step(h, p, stream, n_sets)) return rc;
    }
    return SGX_OK;
}

// sgx_step_n into a trajectory buffer: slot (first_slot + t) % n_slots of tensors with a leading slot axis receives step t's outputs.
SGX_API int sgx_step_traj(sgx_env *h, const sgx_traj_io *t, int32_t first_slot, int32_t n_steps, void *stream) {
    if (!h || !t) return fail(SGX_EINVAL, "handle or t is NULL%s");
    const sgx_step_io &io = t->io;
    if (!io.actions_dev || io.next_actions_dev != io.actions_dev)
        return fail(SGX_EINVAL, "sgx_step_traj needs next_actions_dev == actions_dev (each step plays the action the previous one drew)%s");
    if (t->n_slots < 1 || first_slot < 0 || first_slot >= t->n_slots || n_steps < 0) return fail(SGX_EINVAL, "sgx_step_traj: n_slots, first_slot or n_steps out of range%s");
    if (t->slot_envs < h->n_envs) return fail(SGX_EINVAL, "sgx_step_traj: slot_envs is smaller than the number of envs%s");
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
    p.mode = 0;
    p.io = io;
    p.traj_out_envs = t->slot_envs;
    p.traj_res_envs = t->results_per_slot ? t->slot_envs : 0;
    p.traj_act_log = t->actions_log_dev;
    // bytes between the slots of each tensor, by the layout the flags ask for
    const bool original = (io.flags & SGX_STEP_ORIGINAL_CHANNELS) != 0;
    const int64_t cells = (int64_t)h->cfg.rows * h->cfg.cols;
    const int64_t obs_env = (io.flags & SGX_STEP_COMPACT_OBS) ? compact_obs_stride(h) : cells * lut_channels(false, original) * 4;
    const int64_t mask_env = (io.flags & SGX_STEP_COMPACT_MASK) ? 4 * (int64_t)mask_words(h)
                           : (io.flags & SGX_STEP_MASK_1D)      ? cells * (h->cfg.rows + h->cfg.cols) + 1 : cells * h->K;
    const OutSets slots{&t->io, t->n_slots, true, io.obs_dev ? t->slot_envs * obs_env : 0, io.fobs_dev ? t->slot_envs * cells * lut_channels(true, original) * 4 : 0,
                        io.mask_dev ? t->slot_envs * mask_env : 0};
    {
        bool launched = false;
        if (int rc = launch_lane_steps(h, p, slots, first_slot, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
        if (int rc = launch_wave_steps(h, p, slots, first_slot, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
    }
    for (int32_t i = 0; i < n_steps; ++i)
        if (int rc = launch_set_step(h, p, slots, (int32_t)(((int64_t)first_slot + i) % t->n_slots), stream)) return rc;
    return SGX_OK;
}

SGX_API int sgx_rollout(sgx_env *h, const sgx_step_io *io, int32_t n_steps, int32_t chains, void *stream) {
    if (!h || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev || io->next_actions_dev != io->actions_dev)
        return fail(SGX_EINVAL, "sgx_rollout needs next_actions_dev == actions_dev (each step plays the action the previous one drew)%s");
    if (n_steps < 0 || chains < 0 || chains > SGX_MAX_CHAINS) return fail(SGX_EINVAL, "n_steps or chains out of range%s");
    // games per chain: a multiple of what eight workgroups play, so that every chain keeps the XCD-aware map
    const int cells = h->cfg.rows * h->cfg.cols;
    // chains = 0: the measured rule (profiles/r04_variant_bench.log, 65,536 games in place): two chains gain where one launch leaves
    // the chip part empty -- boards of up to 36 cells (Micro +19 %, Tiny +4 %, 5x5 +3 %, 6x6 +5 %) and boards whose cell count is no
    // multiple of 4 (15x15 +4 %) -- and lose 2-5 % on 8x8 / 10x10, whose single launch already streams at the memory rate
    SGX_ON_DEVICE(h->device);
    if (chains == 0) {
        // ... and since round 5 boards of at most 16 cells have something better than two chains of launches: all steps in ONE launch
        // (lane_steps_kernel: Micro 29 us per step against 35 with two chains) wherever the call is eligible
        KParams p0 = make_params(h);
        p0.mode = 0;
        bool launched = false;
        const OutSets one{io, 1, false, 0, 0, 0};
        if (int rc = launch_lane_steps(h, p0, one, 0, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
        // ... and the other boards the multi-step launch of the wave-per-game kernels (steps_kernel): faster than two chains of launches on
        // every board since its parameter reads are scalar loads (6x6: 97-99 against 101 us per step, 5x5: 71 against 91; profiles/r05_variant_bench.log)
        if (int rc = launch_wave_steps(h, p0, one, 0, n_steps, stream, &launched)) return rc;
        if (launched) return SGX_OK;
        chains = (cells <= 36 || cells % 4 != 0) ? 2 : 1;
    }
    const int64_t unit = 8 * 8 * (cells <= 16 ? 4 : (cells <= 32 ? 2 : 1));     // 8 workgroups x SGX_WPB waves x Geo::GPW games
    int64_t per = (h->n_envs / chains) / unit * unit;
    if (chains == 1 || per == 0 || n_steps == 0) return sgx_step_n(h, io, n_steps, stream);
    if (!h->chain_fork) HIP_TRY(hipEventCreateWithFlags(&h->chain_fork, hipEventDisableTiming));
    for (int c = 0; c < chains; ++c) {
        if (!h->chain_stream[c]) HIP_TRY(hipStreamCreateWithFlags(&h->chain_stream[c], hipStreamNonBlocking));
        if (!h->chain_join[c]) HIP_TRY(hipEventCreateWithFlags(&h->chain_join[c], hipEventDisableTiming));
    }
    KParams p = make_params(h);
    p.mode = 0;
    p.io = *io;
    HIP_TRY(hipEventRecord(h->chain_fork, (hipStream_t)stream));
    for (int c = 0; c < chains; ++c) HIP_TRY(hipStreamWaitEvent(h->chain_stream[c], h->chain_fork, 0));
    // Submission order: blocks of steps chain by chain.  (Alternating the stream with every launch costs more than the overlap
    // gains -- 65,536 Micro games: 107 us per step against 44 us with one chain; each switch of the submitting queue is a host
    // round trip.)
    const int block = 32;                                    // (4 ... 1000 measured within 3 % of each other)
    int rc = SGX_OK;
    for (int32_t i0 = 0; i0 < n_steps && rc == SGX_OK; i0 += block)
        for (int c = 0; c < chains && rc == SGX_OK; ++c) {
            KParams pc = p;
            pc.env_first = c * per;
            pc.n_envs = c == chains - 1 ? h->n_envs : (c + 1) * per;
            for (int32_t i = i0; i < n_steps && i < i0 + block && rc == SGX_OK; ++i) rc = launch_step(h, pc, (void *)h->chain_stream[c]);
        }
    // the caller's stream waits for every chain that was forked -- also when a launch failed half way: what is already enqueued still
    // writes the caller's buffers and must stay ordered before whatever the caller enqueues next
    return join_chains(h, chains, (hipStream_t)stream, rc);
}

SGX_API int64_t sgx_compact_obs_stride(const sgx_env *h) { return h ? compact_obs_stride(h) : 0; }
SGX_API int64_t sgx_compact_mask_words(const sgx_env *h) { return h ? mask_words(h) : 0; }

SGX_API int sgx_decode_obs(sgx_env *h, const uint8_t *compact_dev, float *obs_dev, void *stream) {
    if (!h || !compact_dev || !obs_dev) return fail(SGX_EINVAL, "NULL argument%s");
    if (reinterpret_cast<uintptr_t>(compact_dev) & 15) return fail(SGX_EINVAL, "sgx_decode_obs: compact_dev must be 16-byte aligned%s");
    if ((reinterpret_cast<uintptr_t>(obs_dev) & 15) && (h->cfg.rows * h->cfg.cols) % 4 == 0)
        return fail(SGX_EINVAL, "sgx_decode_obs: obs_dev must be 16-byte aligned%s");
    SGX_ON_DEVICE(h->device);
    const int64_t obs_bytes = h->n_envs * (int64_t)h->cfg.rows * h->cfg.cols * OBS_CH * 4;
    const int nt = h->nt_mode < 0 ? (obs_bytes > (int64_t)300 * 1000 * 1000 ? 1 : 0) : h->nt_mode;
#define CALL_DECODE_OBS(R, C)                                                                                              \
    do {                                                                                                                   \
        using G_ = Geo<R, C>;                                                                                              \
        const unsigned grid = (unsigned)((h->n_envs + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW));                       \
        decode_obs_kernel<R, C><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(compact_dev, compact_obs_stride(h), compact_capacity(h), obs_dev, h->n_envs, nt); \
    } while (0)
    DISPATCH_GEOMETRY(h, CALL_DECODE_OBS);
#undef CALL_DECODE_OBS
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_decode_mask(sgx_env *h, const uint32_t *bits_dev, uint8_t *mask_dev, void *stream) {
    if (!h || !bits_dev || !mask_dev) return fail(SGX_EINVAL, "NULL argument%s");
    if (reinterpret_cast<uintptr_t>(bits_dev) & 15) return fail(SGX_EINVAL, "sgx_decode_mask: bits_dev must be 16-byte aligned%s");
    SGX_ON_DEVICE(h->device);
#define CALL_DECODE_MASK(R, C)                                                                                             \
    do {                                                                                                                   \
        using G_ = Geo<R, C>;                                                                                              \
        const unsigned grid = (unsigned)((h->n_envs + G_::WPB * G_::GPW - 1) / (G_::WPB * G_::GPW));                       \
        decode_mask_kernel<R, C><<<grid, 64 * G_::WPB, 0, (hipStream_t)stream>>>(bits_dev, mask_dev, h->n_envs);           \
    } while (0)
    DISPATCH_GEOMETRY(h, CALL_DECODE_MASK);
#undef CALL_DECODE_MASK
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_sample_valid(sgx_env *h, const uint8_t *mask_dev, int32_t *actions_dev, void *stream) {
    if (!h || !mask_dev || !actions_dev) return fail(SGX_EINVAL, "NULL argument%s");
    SGX_ON_DEVICE(h->device);
    KParams p = make_params(h);
#define CALL_SAMPLE(R, C) sample_kernel<R, C><<<(unsigned)h->n_envs, 64, 0, (hipStream_t)stream>>>(p, mask_dev, actions_dev)
    DISPATCH_GEOMETRY(h, CALL_SAMPLE);
#undef CALL_SAMPLE
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

// The chooser of the reference's game loop for a batch (loop:6-31, examples/util.py:4-47): masked softmax over caller logits, one sample per game.
SGX_API int sgx_choose_actions(sgx_env *h, const float *logits_dev, const void *mask_dev, float temperature, int32_t flags, int32_t *actions_dev, void *stream) {
    if (!h || !logits_dev || !mask_dev || !actions_dev) return fail(SGX_EINVAL, "NULL argument%s");
    if (!(temperature >= 0.f) || temperature > 3.0e38f) return fail(SGX_EINVAL, "sgx_choose_actions: temperature must be finite and >= 0 (0 = argmax)%s");
    if (flags & ~SGX_STEP_COMPACT_MASK) return fail(SGX_EINVAL, "sgx_choose_actions: the only flag is SGX_STEP_COMPACT_MASK%s");
    const bool bits = (flags & SGX_STEP_COMPACT_MASK) != 0;
    if ((reinterpret_cast<uintptr_t>(logits_dev) & 3) || (bits && (reinterpret_cast<uintptr_t>(mask_dev) & 3)))
        return fail(SGX_EINVAL, "sgx_choose_actions: logits_dev (and a compact mask_dev) must be 4-byte aligned%s");
    SGX_ON_DEVICE(h->device);
    const KParams p = make_params(h);
    const float scale = temperature == 0.f ? __builtin_inff() : 1.4426950408889634f / temperature;     // (+inf: every weight below the maximum's is 0)
    const int64_t na = (int64_t)h->cfg.rows * h->cfg.cols * h->K;
    // rows of 4 actions per lane where every game's logits start on a 16-byte and its mask on a 4-byte boundary
    const bool vec4 = na % 4 == 0 && !(reinterpret_cast<uintptr_t>(logits_dev) & 15) && !(reinterpret_cast<uintptr_t>(mask_dev) & 3);
#define CALL_CHOOSE_K(R, C, V, B)                                                                                                    \
    do {                                                                                                                             \
        using CG_ = ChooseGeo<Geo<R, C>, V>;                                                                                         \
        choose_kernel<R, C, V, B><<<(unsigned)((h->n_envs + CG_::GAMES - 1) / CG_::GAMES), 64 * CG_::WAVES, 0, (hipStream_t)stream>>>( \
            p, logits_dev, mask_dev, scale, actions_dev);                                                                    \
    } while (0)
#define CALL_CHOOSE(R, C)                                                                 \
    do {                                                                                  \
        if constexpr ((Geo<R, C>::NA % 4) == 0) {                                         \
            if (vec4) { if (bits) CALL_CHOOSE_K(R, C, 4, true); else CALL_CHOOSE_K(R, C, 4, false); break; } \
        }                                                                                 \
        if (bits) CALL_CHOOSE_K(R, C, 1, true); else CALL_CHOOSE_K(R, C, 1, false);       \
    } while (0)
    DISPATCH_GEOMETRY(h, CALL_CHOOSE);
#undef CALL_CHOOSE
#undef CALL_CHOOSE_K
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

namespace {
// export / import of the envs [p.env_first, p.n_envs) of the handle
// one state per workgroup, mapped onto the XCDs like the step's workgroups (group_of_block): contiguous ranges of states per XCD
// (tools/states_ab.py, 65,536 Barrage states through sgx_step_states: 689 us against 736 us with the linear map; these kernels read as
// much as they write and the odd / even skew of the write-only step does not pay here: 697 us at 100 per mille, 702 at 150)
unsigned state_grid(const sgx_env *h, KParams &p) {
    int32_t w[8];
    launch_shares(h, false, w);
    p.map_mode = h->map_mode; p.map_arg = h->map_arg;
    return shares_for(p, p.n_envs - p.env_first, w);
}
bool states_stream_past_cache(const sgx_env *h) {
    return h->n_envs * (int64_t)SGX_STATE_LAYERS * h->cfg.rows * h->cfg.cols * 8 > (int64_t)300 * 1000 * 1000;
}
int launch_export(sgx_env *h, const KParams &p_in, int64_t *state_dev, int8_t *player_dev, hipStream_t stream) {
    // (the int64 layout is 27 KB per 10x10 state: past the Infinity Cache the layers leave as non-temporal stores, like the observations)
    const int nt = h->nt_mode < 0 ? states_stream_past_cache(h) : h->nt_mode;
    KParams p = p_in;
    const unsigned grid = state_grid(h, p);
#define CALL_EXPORT(R, C) export_kernel<R, C><<<grid, 256, 0, stream>>>(p, state_dev, player_dev, nt)
    DISPATCH_GEOMETRY(h, CALL_EXPORT);
#undef CALL_EXPORT
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}
int launch_import(sgx_env *h, const KParams &p_in, const int64_t *state_dev, const int8_t *player_dev, uint8_t *sanitised_dev, hipStream_t stream) {
    KParams p = p_in;
    const unsigned grid = state_grid(h, p);
#define CALL_IMPORT(R, C) import_kernel<R, C><<<grid, 256, 0, stream>>>(p, state_dev, player_dev, sanitised_dev)
    DISPATCH_GEOMETRY(h, CALL_IMPORT);
#undef CALL_IMPORT
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}
}  // namespace

SGX_API int sgx_export_state(sgx_env *h, int64_t *state_dev, int8_t *player_dev, void *stream) {
    if (!h || !state_dev) return fail(SGX_EINVAL, "NULL argument%s");
    SGX_ON_DEVICE(h->device);
    return launch_export(h, make_params(h), state_dev, player_dev, (hipStream_t)stream);
}

SGX_API int sgx_import_state_checked(sgx_env *h, const int64_t *state_dev, const int8_t *player_dev, uint8_t *sanitised_dev, void *stream) {
    if (!h || !state_dev) return fail(SGX_EINVAL, "NULL argument%s");
    SGX_ON_DEVICE(h->device);
    return launch_import(h, make_params(h), state_dev, player_dev, sanitised_dev, (hipStream_t)stream);
}

namespace {
// sgx_step_states' list of the states its first pass had to alter: created on first use, emptied (on the caller's stream) before every
// first pass; p.flag_list / p.flag_count make the imports of that pass append to it
int redo_list_reset(sgx_env *h, KParams &p, hipStream_t stream) {
    if (!h->redo_list) HIP_TRY(hipMalloc((void **)&h->redo_list, ((size_t)h->n_envs + 1) * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(h->redo_list + h->n_envs, 0, sizeof(int32_t), stream));
    p.flag_list = h->redo_list;
    p.flag_count = h->redo_list + h->n_envs;
    return SGX_OK;
}
// the general-state pass: a small persistent grid that walks the list
unsigned redo_grid(const sgx_env *h) { return (unsigned)(h->n_envs < 2048 ? h->n_envs : 2048); }
}  // namespace

// get_next_state (penv:148-155) and friends on caller-provided int64 states in ONE call: import -> step -> export, the batch split
// into `chains` ranges of states that run on streams of their own, so that one range's reads overlap another's writes.
SGX_API int sgx_step_states(sgx_env *h, const int64_t *state_in_dev, const int8_t *player_in_dev, uint8_t *sanitised_dev,
                            const sgx_step_io *io, int64_t *state_out_dev, int8_t *player_out_dev, int32_t chains, void *stream) {
    if (!h || !state_in_dev || !io) return fail(SGX_EINVAL, "NULL argument%s");
    if (io->auto_reset || io->next_actions_dev) return fail(SGX_EINVAL, "sgx_step_states: no auto_reset, no sampled next actions%s");
    if (io->flags & (SGX_STEP_COMPACT_OBS | SGX_STEP_COMPACT_MASK)) return fail(SGX_EINVAL, "sgx_step_states: no compact outputs%s");
    if (chains < 1 || chains > SGX_MAX_CHAINS) return fail(SGX_EINVAL, "chains out of range%s");
    // Will a general-state pass run behind this call?  (The fused one-launch path always has one; the three-launch paths have one except for
    // state-coordinate masks together with another observation kind: the same conditions as `general` / `general_small` below.)
    const bool will_redo = [&] {
        const int cells = h->cfg.rows * h->cfg.cols;
        const bool k0 = !io->fobs_dev && !io->final_fobs_dev && !(io->flags & SGX_STEP_ORIGINAL_CHANNELS);
        const bool mapped = (io->flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) != 0;
        return h->general_states != 0 && cells <= 256 && (k0 || !mapped);
    }();
    if (will_redo) {
        // The general-state pass re-reads the caller's INPUT after the first pass has written the outputs: a state stepped in place would
        // be redone from its own successor (stepped twice, or judged invalid) without anybody noticing.  (Calls behind which no such pass
        // runs may step a batch in place: every state is read completely before its successor is written.)
        auto overlap = [](const void *a, int64_t na, const void *b, int64_t nb) {
            const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), b0 = reinterpret_cast<uintptr_t>(b);
            return a && b && a0 < b0 + (uintptr_t)nb && b0 < a0 + (uintptr_t)na;
        };
        const int64_t state_bytes = h->n_envs * (int64_t)SGX_STATE_LAYERS * h->cfg.rows * h->cfg.cols * 8;
        if (overlap(state_in_dev, state_bytes, state_out_dev, state_bytes) || overlap(player_in_dev, h->n_envs, player_out_dev, h->n_envs))
            return fail(SGX_EINVAL, "sgx_step_states: state_out_dev / player_out_dev overlap the inputs, which the general-state pass reads again "
                                    "after the outputs are written; pass distinct buffers or switch the pass off (sgx_set_general_states(h, 0))%s");
    }
    SGX_ON_DEVICE(h->device);
    const int64_t unit = 64;
    int64_t per = (h->n_envs / chains) / unit * unit;
    if (per == 0) chains = 1;
    KParams p = make_params(h);
    p.mode = io->actions_dev ? 0 : 1;        // no actions: observe -- masks / observations of the given states, nothing is played
    p.io = *io;
    const bool kind0 = !io->fobs_dev && !io->final_fobs_dev && !(io->flags & SGX_STEP_ORIGINAL_CHANNELS);
    if (kind0 && h->cfg.rows * h->cfg.cols > 32 && h->cfg.rows * h->cfg.cols <= 256) {
        // one-game-per-wave boards, partial-observation kinds: ONE fused launch, the record never leaves LDS between the three steps
        if (int rc = check_step_io(h, p)) return rc;
        const int nt = h->nt_mode < 0 ? states_stream_past_cache(h) : h->nt_mode;
        const unsigned grid = state_grid(h, p);
        p.nt_stores = h->nt_mode < 0 ? (launch_streams_past_cache(h, p) ? 1 : 0) : h->nt_mode;
        const bool mapped = (io->flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) != 0;
        const bool obs = io->obs_dev || io->final_obs_dev;
        // Second pass (sgx_set_general_states, on by default): the states the first pass had to alter -- more than two recent-move cells per
        // player, more capture cells than pieces, more than 8 captures on a cell: things play cannot produce but penv's pure functions
        // accept -- are redone from the caller's int64 input on the general-state variant of the geometry (Geo<R, C, 1>); every other
        // block of that launch leaves at once.  It needs the flags even when the caller did not ask for them.
        const bool general = h->general_states != 0;
        uint8_t *flags_dev = sanitised_dev;
        if (general && !flags_dev) {
            if (!h->san_flags) HIP_TRY(hipMalloc((void **)&h->san_flags, (size_t)h->n_envs));
            flags_dev = h->san_flags;
        }
        if (general)
            if (int rc = redo_list_reset(h, p, (hipStream_t)stream)) return rc;
        KParams pbig = p;                                   // the general-state pass appends nothing
        pbig.flag_list = pbig.flag_count = nullptr;
        const unsigned grid_big = redo_grid(h);
#define CALL_STATES_K(R, C, M, O, V) states_kernel<R, C, M, O, V><<<(V) ? grid_big : grid, states_threads<M, O>(), 0, (hipStream_t)stream>>>((V) ? pbig : p, state_in_dev, player_in_dev, flags_dev, state_out_dev, player_out_dev, nt, h->redo_list, h->redo_list ? h->redo_list + h->n_envs : nullptr)
#define CALL_STATES_V(R, C, V)                                                                                              \
    do {                                                                                                                    \
        if constexpr (Geo<R, C>::LPG == 64 && !Geo<R, C>::WIDE) {                                                           \
            if (mapped && obs) CALL_STATES_K(R, C, true, true, V);                                                          \
            else if (mapped) CALL_STATES_K(R, C, true, false, V);                                                           \
            else if (obs) CALL_STATES_K(R, C, false, true, V);                                                              \
            else CALL_STATES_K(R, C, false, false, V);                                                                      \
        }                                                                                                                   \
    } while (0)
#define CALL_STATES(R, C) CALL_STATES_V(R, C, 0)
#define CALL_STATES_BIG(R, C) CALL_STATES_V(R, C, 1)
        DISPATCH_GEOMETRY(h, CALL_STATES);
        HIP_TRY(hipGetLastError());
        if (general) {
            DISPATCH_GEOMETRY(h, CALL_STATES_BIG);
            HIP_TRY(hipGetLastError());
        }
#undef CALL_STATES
#undef CALL_STATES_BIG
#undef CALL_STATES_V
#undef CALL_STATES_K
        return SGX_OK;
    }
    // What the fused path above does not cover takes the three-launch path on packed records -- boards of up to 32 cells (several games per
    // wave) and the other observation kinds (79-channel, 'original' channels).  States the packed records cannot carry are then redone,
    // like above, by the general-state variant of the fused kernel, which is one game per wave on every board and takes the observation
    // kind as a parameter.  (Not: boards of more than 256 cells; state-coordinate masks together with another observation kind.)
    const int cells_ = h->cfg.rows * h->cfg.cols;
    const bool mapped_ = (io->flags & (SGX_STEP_MASK_1D | SGX_STEP_MASK_STATE_COORDS)) != 0;
    const bool general_small = h->general_states != 0 && cells_ <= 256 && (kind0 ? cells_ <= 32 : !mapped_);
    uint8_t *flags_small = sanitised_dev;
    if (general_small && !flags_small) {
        if (!h->san_flags) HIP_TRY(hipMalloc((void **)&h->san_flags, (size_t)h->n_envs));
        flags_small = h->san_flags;
    }
    if (general_small)
        if (int rc = redo_list_reset(h, p, (hipStream_t)stream)) return rc;       // (before the chains fork: their imports append to the list)
    auto second_pass_small = [&]() -> int {
        if (!general_small) return SGX_OK;
        if (int rc = check_step_io(h, p)) return rc;
        KParams q = p;
        q.env_first = 0;
        q.n_envs = h->n_envs;
        q.flag_list = q.flag_count = nullptr;
        const int nt = h->nt_mode < 0 ? states_stream_past_cache(h) : h->nt_mode;
        const unsigned grid = redo_grid(h);
        q.nt_stores = h->nt_mode < 0 ? (launch_streams_past_cache(h, q) ? 1 : 0) : h->nt_mode;
        const bool obs = io->obs_dev || io->final_obs_dev;
        const int kindx = ((io->fobs_dev || io->final_fobs_dev) ? 1 : 0) | ((io->flags & SGX_STEP_ORIGINAL_CHANNELS) ? 2 : 0);
#define CALL_STATES_SMALL_K(R, C, M, O) states_kernel<R, C, M, O, 1><<<grid, states_threads<M, O>(), 0, (hipStream_t)stream>>>(q, state_in_dev, player_in_dev, flags_small, state_out_dev, player_out_dev, nt, h->redo_list, h->redo_list + h->n_envs)
#define CALL_STATES_KIND_K(R, C, KX) states_kernel<R, C, false, true, 1, KX><<<grid, states_threads<false, true>(), 0, (hipStream_t)stream>>>(q, state_in_dev, player_in_dev, flags_small, state_out_dev, player_out_dev, nt, h->redo_list, h->redo_list + h->n_envs)
#define CALL_STATES_SMALL(R, C)                                                                                             \
    do {                                                                                                                    \
        if constexpr (!Geo<R, C>::WIDE) {                                                                                   \
            if (kindx == 1) CALL_STATES_KIND_K(R, C, 1);                                                                    \
            else if (kindx == 2) CALL_STATES_KIND_K(R, C, 2);                                                               \
            else if (kindx == 3) CALL_STATES_KIND_K(R, C, 3);                                                               \
            else if constexpr (Geo<R, C>::LPG != 64) {                                                                      \
                if (mapped_ && obs) CALL_STATES_SMALL_K(R, C, true, true);                                                  \
                else if (mapped_) CALL_STATES_SMALL_K(R, C, true, false);                                                   \
                else if (obs) CALL_STATES_SMALL_K(R, C, false, true);                                                       \
                else CALL_STATES_SMALL_K(R, C, false, false);                                                               \
            }                                                                                                               \
        }                                                                                                                   \
    } while (0)
        DISPATCH_GEOMETRY(h, CALL_STATES_SMALL);
#undef CALL_STATES_SMALL
#undef CALL_STATES_SMALL_K
#undef CALL_STATES_KIND_K
        HIP_TRY(hipGetLastError());
        return SGX_OK;
    };
    if (chains == 1) {
        if (int rc = launch_import(h, p, state_in_dev, player_in_dev, flags_small, (hipStream_t)stream)) return rc;
        if (int rc = launch_step(h, p, stream)) return rc;
        if (state_out_dev)
            if (int rc = launch_export(h, p, state_out_dev, player_out_dev, (hipStream_t)stream)) return rc;
        return second_pass_small();
    }
    if (!h->chain_fork) HIP_TRY(hipEventCreateWithFlags(&h->chain_fork, hipEventDisableTiming));
    for (int c = 0; c < chains; ++c) {
        if (!h->chain_stream[c]) HIP_TRY(hipStreamCreateWithFlags(&h->chain_stream[c], hipStreamNonBlocking));
        if (!h->chain_join[c]) HIP_TRY(hipEventCreateWithFlags(&h->chain_join[c], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(h->chain_fork, (hipStream_t)stream));
    for (int c = 0; c < chains; ++c) HIP_TRY(hipStreamWaitEvent(h->chain_stream[c], h->chain_fork, 0));
    int rc = SGX_OK;
    for (int c = 0; c < chains && rc == SGX_OK; ++c) {
        KParams pc = p;
        pc.env_first = c * per;
        pc.n_envs = c == chains - 1 ? h->n_envs : (c + 1) * per;
        rc = launch_import(h, pc, state_in_dev, player_in_dev, flags_small, h->chain_stream[c]);
        if (rc == SGX_OK) rc = launch_step(h, pc, (void *)h->chain_stream[c]);
        if (rc == SGX_OK && state_out_dev) rc = launch_export(h, pc, state_out_dev, player_out_dev, h->chain_stream[c]);
    }
    rc = join_chains(h, chains, (hipStream_t)stream, rc);     // (also after a failed launch: see sgx_rollout)
    return rc != SGX_OK ? rc : second_pass_small();
}

SGX_API int sgx_import_state(sgx_env *h, const int64_t *state_dev, const int8_t *player_dev, void *stream) {
    return sgx_import_state_checked(h, state_dev, player_dev, nullptr, stream);
}

namespace {
int same_variant(const sgx_env *a, const sgx_env *b) {
    if (a->device != b->device) return fail(SGX_EINVAL, "the two handles live on different devices%s");
    if (memcmp(&a->cfg, &b->cfg, sizeof(sgx_config)) != 0 || a->rec_bytes != b->rec_bytes)
        return fail(SGX_EINVAL, "the two handles were created for different variants%s");
    return SGX_OK;
}
}  // namespace

SGX_API int sgx_copy_envs(sgx_env *dst, const int32_t *dst_index_dev, sgx_env *src, const int32_t *src_index_dev, int64_t n, void *stream) {
    if (!dst || !src) return fail(SGX_EINVAL, "handle is NULL%s");
    if (int rc = same_variant(dst, src)) return rc;
    if (n < 0 || (!dst_index_dev && n > dst->n_envs) || (!src_index_dev && n > src->n_envs)) return fail(SGX_EINVAL, "n out of range%s");
    // inside one pool a wave may read a record another wave of the same launch rewrites: only the identity copy is race-free
    if (dst == src && (dst_index_dev || src_index_dev))
        return fail(SGX_EINVAL, "sgx_copy_envs: an indexed copy inside one handle would race; stage through a second handle%s");
    if (n == 0 || dst == src) return SGX_OK;
    SGX_ON_DEVICE(dst->device);
    copy_records_kernel<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(dst->boards, dst_index_dev, src->boards, src_index_dev, dst->rec_bytes, n);
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

SGX_API int sgx_expand(sgx_env *dst, sgx_env *src, const int32_t *src_index_dev, const sgx_step_io *io, void *stream) {
    if (!dst || !src || !io) return fail(SGX_EINVAL, "handle or io is NULL%s");
    if (!io->actions_dev) return fail(SGX_EINVAL, "actions_dev is NULL%s");
    if (int rc = same_variant(dst, src)) return rc;
    if (!src_index_dev && src->n_envs < dst->n_envs) return fail(SGX_EINVAL, "without src_index_dev the source handle needs at least as many envs%s");
    // wave i reads record src_index[i] while another wave of the same launch rewrites that record: in-place expansion through an
    // index (a gather or a permutation) would silently mix parents and children
    if (src == dst && src_index_dev)
        return fail(SGX_EINVAL, "sgx_expand: src == dst with an index array would race; expand into a second handle%s");
    if (io->auto_reset) return fail(SGX_EINVAL, "sgx_expand does not auto-reset%s");
    if (io->fobs_dev || io->final_fobs_dev || (io->flags & SGX_STEP_ORIGINAL_CHANNELS))
        return fail(SGX_EINVAL, "sgx_expand renders the 67-channel partial observation only%s");
    SGX_ON_DEVICE(dst->device);
    KParams p = make_params(dst);
    p.mode = 0;
    p.io = *io;
    p.src_boards = src->boards;
    p.src_index = src_index_dev;
    return launch_step(dst, p, stream);
}

SGX_API int sgx_get_env_info(sgx_env *h, int32_t *info_dev, void *stream) {
    if (!h || !info_dev) return fail(SGX_EINVAL, "NULL argument%s");
    SGX_ON_DEVICE(h->device);
    info_kernel<<<(unsigned)((h->n_envs + 255) / 256), 256, 0, (hipStream_t)stream>>>(h->boards, h->rec_bytes, h->sc_off, info_dev, h->n_envs);
    HIP_TRY(hipGetLastError());
    return SGX_OK;
}

#ifdef SGX_STAMPS
// diagnostic build only (tools/phase_stamps.py): device pointer of the [N][16] stamp buffer
SGX_API void *sgx_debug_stamps(sgx_env *h) { return h ? (void *)h->stamps : nullptr; }
#endif
