/*
 * stratego_mi355x.h -- C ABI of libstratego_mi355x.so, the MI355X (gfx950) batched Stratego env.
 *
 * The reference (JBLanier/stratego_env) has no FFI: its seam is the Python class
 * StrategoProceduralEnv (stratego_env/game/stratego_procedural_env.py:20-181) under
 * StrategoMultiAgentEnv (stratego_env/stratego_multiagent_env.py:316-834).  Each entry point below
 * names the reference interface it replaces for a batch of N independent games.  Conventions:
 *   - plain C, no C++/torch types; every *_dev pointer is a DEVICE pointer owned by the caller
 *     (e.g. a torch tensor's data_ptr()); the library owns only its internal state tensor;
 *   - all device work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *     and is asynchronous; the entry points that wait for the device are sgx_create, sgx_destroy,
 *     sgx_set_setup_table, sgx_time_observe, sgx_mem_probe, sgx_store_probe, sgx_step_sync, sgx_host_free, sgx_alloc_outputs and
 *     sgx_free_outputs;
 *   - return 0 on success, a negative SGX_E* code on failure (sgx_last_error() has the text);
 *     nothing throws across the ABI; invalid *actions* are not API errors: they are reported per env
 *     in invalid_action[] with that env's state left unchanged (the reference raises ValueError,
 *     stratego_procedural_impl.py:899-902 -- the Python facade turns the flag back into ValueError);
 *   - a handle is bound to one device and is not thread-safe; different handles are independent.  Every entry point runs on its
 *     handle's device and restores the calling thread's current HIP device before it returns (also on errors): one process may drive
 *     handles on several GPUs, and a caller inside a torch.cuda.device(...) scope keeps its device;
 *
 * Data layouts (C order):
 *   obs    float32 [N][R][C][67]   normalised partial observation of the env's NEXT mover, mover's
 *                                  perspective (impl:1335-1397 + maenv:261-313,388-391,506-508)
 *   mask   uint8   [N][R][C][K]    valid-actions mask of the next mover, K = 2(R-1)+2(C-1)+1
 *                                  (impl:399-517; the reference dtype is int64 with values 0/1)
 *   action int32   [N]             flat index into (R,C,K) in the MOVER's perspective (maenv:684-689)
 *   state  int64   [N][34][R][C]   the reference's own state layout, absolute coordinates (impl:16-60)
 */
#ifndef STRATEGO_MI355X_H
#define STRATEGO_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_ABI_VERSION 15
#define SGX_MAX_CELLS 1024       /* rows*cols <= 1024 (largest reference variant: 15x15 = 225; the reference's StrategoProceduralEnv
                                    takes any size, penv:27-36: boards of more than 256 cells use 10-bit cell indices in the record) */
#define SGX_PO_OBS_CHANNELS 67   /* impl:1332 */
#define SGX_FO_OBS_CHANNELS 79   /* impl:1227 */
#define SGX_PO_OBS_CHANNELS_ORIGINAL 32   /* impl:1148, obs_channel_mode='original' (maenv:67, 368-375) */
#define SGX_FO_OBS_CHANNELS_ORIGINAL 33   /* impl:1070 */
#define SGX_STATE_LAYERS 34      /* impl:109 */
#define SGX_OBS_LUT_STRIDE 16    /* entries per channel in the normalisation LUT */
#define SGX_MAX_PIECES_PER_TYPE 127   /* config 'piece_amounts' per piece type (the reference's dict is unbounded, config.py:3-23; a side
                                         never has more pieces than usable cells anyway) */

enum {
    SGX_OK = 0,
    SGX_EINVAL = -1,   /* bad argument / unsupported board size */
    SGX_ENOMEM = -2,   /* device allocation failed */
    SGX_EDEVICE = -3,  /* HIP runtime error (no device, launch failure, ...) */
    SGX_ESTATE = -4    /* call not valid in the handle's current state (e.g. no setup table) */
};

/* Game variant = one of the reference's *_STRATEGO_CONFIG dicts (game/config.py:3-313). */
typedef struct sgx_config {
    int32_t rows, cols;               /* >= 3 each (penv:28-30), rows*cols <= SGX_MAX_CELLS */
    int32_t max_turns;                /* config 'max_turns' -> StateData.MAX_TURNS (impl:247) */
    int32_t usable_rows;              /* config 'initial_state_usable_rows' */
    int32_t piece_counts[12];         /* config 'piece_amounts' for piece codes 1..12 (SPY..BOMB), <= SGX_MAX_PIECES_PER_TYPE each */
    int32_t capture_capacity;         /* most pieces ONE side can have on the board, 0 = sum of piece_counts.  An env_config that
                                         overrides 'piece_amounts' changes the normalisation constants only (maenv:323-326, 370-382)
                                         while the setups keep the version's pieces: sizes the capture-event list */
    uint8_t obstacles[SGX_MAX_CELLS]; /* config 'obstacle_locations' as a row-major rows*cols 0/1 map */
} sgx_config;

typedef struct sgx_env sgx_env; /* opaque handle */

/* Buffers of one batched step.  Replaces StrategoMultiAgentEnv.step(action_dict) (maenv:659-828) +
 * _get_current_obs (maenv:447-497) for N envs.  Nullable members may be NULL to skip that output. */
typedef struct sgx_step_io {
    const int32_t *actions_dev;    /* [N]   in : action of each env's current mover */
    float *obs_dev;                /* [N,R,C,67] out (nullable): partial observation (ObservationModes PARTIALLY / BOTH) */
    float *fobs_dev;               /* [N,R,C,79] out (nullable): fully-observable observation (FULLY / BOTH),
                                      impl:1230-1303 + maenv:202-258, 499-501 */
    uint8_t *mask_dev;             /* [N,R,C,K]  out (nullable) */
    float *reward_dev;             /* [N,2] out: rewards[+1], rewards[-1]; 0,0 while running (maenv:769, 777-805) */
    uint8_t *done_dev;             /* [N]   out: dones["__all__"] */
    int8_t *player_dev;            /* [N]   out: the next mover, +1 / -1 (after auto-reset: +1) */
    uint8_t *invalid_action_dev;   /* [N]   out: 1 where the reference would raise; state unchanged */
    uint8_t *ending_invalid_dev;   /* [N]   out: infos['game_result_was_invalid'] (max-turn tie, maenv:777-782) */
    float *final_obs_dev;          /* [N,2,R,C,67] out (nullable): on terminal steps, the terminal observation of
                                      player +1 (slot 0) and -1 (slot 1) (maenv:772-773); untouched otherwise */
    float *final_fobs_dev;         /* [N,2,R,C,79] out (nullable): the same for the fully-observable observation */
    int32_t *next_actions_dev;     /* [N] out (nullable): a uniformly random valid action for the next mover, drawn
                                      with the handle's counter RNG keyed by (seed, global env id, game, turn);
                                      the batched counterpart of sample_random_valid_action (maenv:830-834) */
    int32_t auto_reset;            /* 1: an env that finishes starts its next game inside the same call
                                      (obs/mask/player then describe the new game's first mover) */
    int32_t flags;                 /* SGX_STEP_* bits, 0 for the env.step() path */
} sgx_step_io;

/* sgx_step_io.flags: the functional StrategoProceduralEnv API on caller-provided states (import -> step -> export) */
#define SGX_STEP_ACTIONS_1D 1         /* actions_dev holds absolute-coordinate 1-D indices (impl:262-277), as
                                         get_next_state / is_move_valid_by_1d_index take them (penv:94-99, 148-155) */
#define SGX_STEP_ALLOW_OSCILLATION 2  /* allow_piece_oscillation=True: skip the two-square check of the MOVE (impl:771-777);
                                         masks still apply it, as in the reference */
#define SGX_STEP_RAW_OBS 4            /* observations un-normalised, as penv:166-173 return them */
#define SGX_STEP_ACTIONS_POSITIONS 8   /* actions_dev is int32 [N][4] = (start_r, start_c, end_r, end_c) in absolute coordinates
                                         (is_move_valid_by_position, penv:87-92) */
/* env_config['obs_channel_mode'] == 'original' (maenv:67, 368-375, 465-467, 484-486): the deprecated value-channel
 * observations.  obs_dev / final_obs_dev then have 32 channels (impl:1126-1197) and fobs_dev / final_fobs_dev 33
 * (impl:1048-1123), normalised with the constants of maenv:87-199. */
#define SGX_STEP_ORIGINAL_CHANNELS 16
/* Masks of the functional API are indexed in the coordinates of the given STATE, not in the mover's perspective:
 * SGX_STEP_MASK_1D: mask_dev is uint8 [N][R*C*(R+C)+1], get_valid_moves_as_1d_mask (penv:74-80, impl:520-642) of the current mover;
 * SGX_STEP_MASK_STATE_COORDS: mask_dev is uint8 [N][R][C][K] as get_valid_moves_as_spatial_mask(state, player) returns it for
 * the current mover WITHOUT the perspective flip (penv:127-128, impl:399-517; identical to the default for player +1).
 * Both are also accepted by sgx_observe; they cannot be combined with fobs_dev / final_fobs_dev or SGX_STEP_ORIGINAL_CHANNELS. */
#define SGX_STEP_MASK_1D 32
#define SGX_STEP_MASK_STATE_COORDS 64

/* Compact outputs (opt-in; never the default).  The float32 observation is 85 % of a step's bytes and the step kernel already sits on
 * the write roofline, so the only way past it is to write less: with SGX_STEP_COMPACT_OBS obs_dev receives, per game, the kernel's own
 * 4-bit code buffer -- uint8 [N][sgx_compact_obs_stride(h)]: ceil(R*C*67 / 2) bytes of codes (nibble e = float e of the [R,C,67]
 * observation, low nibble first; code c decodes to sext(c) / 4, code 8 marks an entry without a code), padded to 16 bytes; a 16-byte
 * header {int32 n, 0, 0, 0}; n x {uint32 entry, float32 value} for the marked entries (captured counts that normalise to thirds /
 * fifths); padded to whole 128-byte lines (10x10: 3,584 B for Barrage instead of 26,800 B).  With SGX_STEP_COMPACT_MASK mask_dev
 * receives the mask as bits, uint32 [N][sgx_compact_mask_words(h)], bit a of the game's words = flat action a (10x10: 480 B instead of
 * 3,700 B).  sgx_decode_obs / sgx_decode_mask expand a batch of them with the step kernel's own emission code: the float32 / uint8
 * results are byte-identical to what the step writes without the flags (tests/test_gpu_compact.py).  Accepted by sgx_step, sgx_observe,
 * sgx_step_n, sgx_rollout and sgx_step_ring for the 67-channel partial observation of an 'extended' channel mode and masks in the
 * mover's perspective (no fobs_dev / final_obs_dev / original channels / state-coordinate masks; 16-byte aligned buffers).  No reference
 * counterpart: the reference's contract is the float32 observation (impl:1335-1397 + maenv:506-508), which the decode ops deliver. */
#define SGX_STEP_COMPACT_OBS 128
#define SGX_STEP_COMPACT_MASK 256
int64_t sgx_compact_obs_stride(const sgx_env *h);
int64_t sgx_compact_mask_words(const sgx_env *h);
int sgx_decode_obs(sgx_env *h, const uint8_t *compact_dev, float *obs_dev /* [N,R,C,67] */, void *stream);
int sgx_decode_mask(sgx_env *h, const uint32_t *bits_dev, uint8_t *mask_dev /* [N,R,C,K] */, void *stream);

/* Library / geometry queries (penv:32-36: action_size, spatial_action_size). */
int sgx_abi_version(void);
/* Hash of the sources the loaded binary was compiled from (the first 16 hex digits of SHA-256 over the files of stratego_env_amd/csrc and this
 * header, stratego_env_amd/build.py: source_hash; "unknown" for a build outside build.py).  The Python binding refuses a library whose
 * id differs from the sources next to it, bench.py and smoke() print it: the tested binary is tied to the tested source.  The same
 * string follows the marker "SGX_BUILD_ID=" in the file, so it can be read without loading the library. */
const char *sgx_build_id(void);
/* 1 if kernels for this board size are compiled into the library.  Every reference variant is (10x10, 15x15, 8x8, 6x6, 5x5, 4x4, 3x4);
 * the reference's StrategoProceduralEnv(rows, columns) takes ANY size >= 3 (penv:27-36): for other sizes (rows * cols <= SGX_MAX_CELLS)
 * the same sources are compiled into a library of their own with -DSGX_EXTRA_R=<rows> -DSGX_EXTRA_C=<cols> -DSGX_ONLY_EXTRA
 * (stratego_env_amd/build.py build_geometry does it on first use). */
int sgx_supports_geometry(int32_t rows, int32_t cols);
const char *sgx_last_error(void);
int64_t sgx_num_envs(const sgx_env *h);
/* Bytes of one game's packed state record in device memory (a multiple of 128: 512 for Barrage, 640 for
 * Standard).  One env.step() reads it once and writes it once: with the outputs it is the byte minimum B_min of a step that bench.py's
 * roofline is priced on.  No reference counterpart (the reference's state is int64 [34,R,C] = 27,200 B at 10x10, impl:69-163). */
int64_t sgx_record_bytes(const sgx_env *h);
int sgx_spatial_channels(const sgx_env *h);          /* K */
int64_t sgx_num_spatial_actions(const sgx_env *h);   /* R*C*K = Discrete(n) of maenv:362 */
int64_t sgx_action_size_1d(const sgx_env *h);        /* R*C*(R+C)+1 (impl:252-254) */

/* Host-only: fills lut[67*SGX_OBS_LUT_STRIDE] with the float32 bit-exact normalised value of every
 * (channel, raw integer value) pair: entry [ch*16 + clamp(v + bias_ch, 0, 15)] where bias is 3 for the two
 * recent-moves channels (raw -3..1) and 0 elsewhere.  Replaces the arithmetic of
 * normalize_p_observation (maenv:506-508) with the constants of maenv:261-313, 388-391. */
int sgx_build_obs_lut(const sgx_config *cfg, float *lut);
/* Same for the 79 fully-observable channels: lut[79*SGX_OBS_LUT_STRIDE] (maenv:202-258, 393-396, 499-501). */
int sgx_build_full_obs_lut(const sgx_config *cfg, float *lut);
/* Same for obs_channel_mode='original': lut[32*SGX_OBS_LUT_STRIDE] (full == 0; maenv:146-199) or
 * lut[33*SGX_OBS_LUT_STRIDE] (full != 0; maenv:87-143). */
int sgx_build_original_obs_lut(const sgx_config *cfg, int32_t full, float *lut);

/* StrategoMultiAgentEnv.__init__ (maenv:318-445) for a batch: allocates the device state of n_envs games on
 * `device`.  Env i of this handle has global id env_id_offset + i; all random draws are keyed by
 * (seed, global id, game number, turn), so trajectories do not depend on how the batch is sharded. */
int sgx_create(const sgx_config *cfg, int64_t n_envs, int device, uint64_t seed, int64_t env_id_offset, sgx_env **out);
int sgx_destroy(sgx_env *h);

/* Store policy of the observation writes of sgx_step / sgx_observe.  Lines a wave writes whole can leave as non-temporal stores:
 * faster when a launch's observations do not fit the 256 MiB Infinity Cache, slower when they do.
 * mode -1 (default): decided per launch from the observation bytes it writes (> 300 MB: non-temporal); 0: never; 1: always.
 * The environment variable SGX_NT=0|1|auto sets the default of handles created afterwards.  Results are identical in every
 * mode (tests/test_gpu_nt_stores.py runs the parity suites with the mode forced).  No reference counterpart. */
int sgx_set_nt_stores(sgx_env *h, int32_t mode);

/* Kernel choice on boards of at most 16 cells with a multiple of 4 cells (Micro 3x4, Tiny 4x4).  A second kernel plays ONE GAME PER LANE
 * there (64 games per wave, boards as nibbles in registers; docs/DESIGN_rounds_4-5.md section 3.3).  It is eligible for sgx_step / sgx_observe / sgx_step_n /
 * sgx_rollout / sgx_step_ring calls that ask for the 67-channel partial observation of an 'extended' channel mode (or none), masks in the
 * mover's perspective, no terminal-observation buffers and 16-byte aligned output tensors.  Measured: its game logic is twice as fast
 * (65,536 Micro games without outputs 13 against 26 us, mask only 15 against 30 us), but with the observation emitted one launch is slower
 * (51 against 42 us: one wave per SIMD, nothing overlaps its logic with its stores).  mode -1 (default): the lane kernel for eligible
 * calls that emit NO observation, the wave-per-game kernel otherwise; 0: never; 1: for every eligible call.  SGX_LANE=0|1|auto sets the
 * default of handles created afterwards.  Results are identical either way (tests/test_gpu_lane_kernel.py).  No reference counterpart. */
int sgx_set_lane_kernel(sgx_env *h, int32_t mode);

/* Multi-step launches.  A rollout call -- sgx_step_n / sgx_step_ring: n_steps >= 2 consecutive steps, each playing the action the one
 * before drew -- runs its steps in ONE launch (launches of at most 256 steps, on every board) where it is eligible: flat perspective actions,
 * masks in the mover's perspective, an 'extended' channel mode, output sets that differ in their observation / mask tensors only (any number
 * of them: beyond 8 sets sgx_step_ring passes their pointers through a small device table; a trajectory buffer in ONE allocation goes through
 * sgx_step_traj, which names its slots by a stride).
 *  - Boards of more than 16 cells (steps_kernel): a workgroup stages its games once and every wave plays its game step after step -- the
 *    dense boards, never-moved flags, recent-move codes and capture events stay in LDS, the record's scalars and the drawn action in
 *    registers; every step's outputs are written like those of a launch of its own; the record is read once and written once per LAUNCH.
 *    No barrier after the prologue: the waves drift out of phase, one wave's stores run under another's game logic.  The 67-channel kind,
 *    BOTH observations, compact outputs and calls without an observation.  65,536 Barrage games into a ring of three output sets:
 *    284-287 -> 246-249 us per step; 262,144 Standard games 1,210-1,233 -> 977-991 us (DESIGN.md sections 3 and 4).
 *  - Boards of at most 16 cells with a multiple of 4 cells, the 67-channel kind with an observation tensor (lane_steps_kernel): a 256-thread
 *    workgroup keeps 64 games in the registers of one wave, which plays step t + 1 while the other three waves store the observations of
 *    step t.  65,536 Micro games 42.1 -> 29.2 us per step.
 * Same results as n_steps launches of sgx_step (tests/test_gpu_multi_step.py, tests/test_gpu_lane_kernel.py).  mode 1 (default): on; 0: one
 * launch per step.  SGX_MULTI_STEP=0 sets the default of handles created afterwards (SGX_MULTI_STEP_WAVE=0: the first kind only off);
 * sgx_set_lane_kernel(h, 0) switches the second kind off as well.  No reference counterpart. */
int sgx_set_multi_step(sgx_env *h, int32_t mode);
/* Launches that write NO observation (sgx_expand, mask-only and logic-only steps / rollouts: no obs_dev / fobs_dev / final_*_dev, no compact
 * outputs) are bound by instruction issue, not by memory; on boards of 33 .. 128 cells they play TWO games per wave (32 lanes per game): 65,536
 * Barrage games, logic-only rollout 60.6 -> 33.7 us per step, search expansion 0.96 -> 1.75 G states/s.  Same results
 * (the parity suites of the no-observation kind run on it).  mode 1 (default): two games per wave where the board allows it; 0: one game
 * per wave everywhere (A/B measurements: tools/half_wave_ab.py).  The environment variable SGX_HALF_WAVE=0|1 sets the default of handles
 * created afterwards.  No reference counterpart. */
int sgx_set_half_wave(sgx_env *h, int32_t mode);
/* Multi-step launches of the wave-per-game kernel (sgx_step_n / sgx_step_ring / sgx_step_traj on boards of more than 16 cells): should the waves of a
 * workgroup -- 8 adjacent games -- meet at a barrier before every step?  Without it they drift apart within a few steps, which lets one wave's
 * stores run under another's game logic; on a LONG ring or trajectory buffer that drift spreads a workgroup's writes over up to 8 sets at a time,
 * and a launch whose resident waves cycle through more sets than the address translation caches hold pages for runs 1-7 %
 * faster with the waves in step (64-slot trajectory buffer of 65,536 Barrage games: 295 -> 282 us per step).  mode -1 (default): the barrier where
 * it was measured to pay -- float32 observations, one game per wave (36 .. 225 cells), from 9 sets / slots on 10x10 and from 16 elsewhere; 0: never;
 * 1: in every multi-step launch (A/B runs and the parity tests).  Same results in every mode (tests/test_gpu_multi_step.py).
 * SGX_STEPS_BARRIER=-1|0|1 sets the default of handles created afterwards.  No reference counterpart. */
int sgx_set_steps_barrier(sgx_env *h, int32_t mode);

/* Which kernel the handle's last sgx_step / sgx_observe / sgx_step_n / sgx_step_ring / sgx_rollout / sgx_expand launch was (diagnostics,
 * benchmarks that price a launch by its own bytes, tests that must not pass on another kernel): */
#define SGX_LAUNCH_WAVE 0        /* one wave per game (boards of up to 32 cells: a wave's lanes shared by 2 or 4 games), one launch per step */
#define SGX_LAUNCH_LANE 1        /* one game per lane (boards of at most 16 cells), one launch per step */
#define SGX_LAUNCH_MULTI_STEP 2  /* one game per lane, all steps of the call in one launch (sgx_set_multi_step) */
#define SGX_LAUNCH_MULTI_STEP_WAVE 3   /* one wave per game, all steps of the call in one launch: the boards stay in LDS between the steps */
int sgx_last_launch_kind(const sgx_env *h);

/* Shares of the eight XCDs in a launch of sgx_step / sgx_observe.  Under a saturating write stream the odd XCDs of MI355X drain their
 * eighth of the games ~20 % slower than the even ones, so with equal eighths the even XCDs idle at the end of every launch; the
 * library gives the even XCD of each pair `per_mille` more than the mean share and the odd one as much less (DESIGN.md section 3:
 * -3 ... -5 % launch time where the write stream bounds the kernel: 8x8 and 10x10 boards; +3 ... +7 % where the game logic shares the
 * critical path: 6x6, 15x15, Micro).  -1 (default): 100 per mille on boards of 64 .. 100 cells when a launch's observations do not fit
 * the Infinity Cache, equal shares otherwise; 0: always equal; 1 .. 900: always that.  SGX_XCD_SKEW=<per mille>|auto sets the default of
 * handles created afterwards.  Which workgroup plays which game cannot change any result.  No reference counterpart. */
int sgx_set_xcd_skew(sgx_env *h, int32_t per_mille);
/* The general form: a share per XCD, per mille of the mean share (NULL: back to the sgx_set_xcd_skew rule); sgx_get_xcd_shares returns
 * what a streaming launch of the handle uses (*explicit_out, nullable: 1 if set by sgx_set_xcd_shares). */
int sgx_set_xcd_shares(sgx_env *h, const int32_t *per_mille /* [8] or NULL */);
int sgx_get_xcd_shares(sgx_env *h, int32_t *per_mille /* [8] */, int32_t *explicit_out);

/* Upload a human-setup table (game/inits/{barrage,standard}_human_inits.py decoded to piece codes, util.py:154-180):
 * table_host is uint8 [n_setups][usable_rows*cols] in Gravon string order.  Replaces get_random_human_init_fn
 * (util.py:301-319).  Without a table, sampled resets place pieces uniformly at random in the usable rows
 * (get_random_initial_state_fn, util.py:13-53). */
int sgx_set_setup_table(sgx_env *h, const uint8_t *table_host, int64_t n_setups);

/* reset() (maenv:513-657) for the envs selected by env_select_dev (uint8 [N], NULL = all).
 * p1_maps_dev / p2_maps_dev: int8 [N][R*C] own-side piece maps, the inputs of create_initial_state
 * (penv:38-60 -> impl:211-249); pass NULL for both to sample setups (table or random placement).
 * Starts game number 0 (explicit maps) or the env's next game number (sampled).  Player +1 moves first. */
int sgx_reset(sgx_env *h, const uint8_t *env_select_dev, const int8_t *p1_maps_dev, const int8_t *p2_maps_dev, void *stream);

/* _get_current_obs (maenv:447-497) for every env's current mover, no state change.
 * obs_dev, fobs_dev, mask_dev and player_dev are laid out as in sgx_step_io; each is nullable.
 * flags: 0 or any of SGX_STEP_RAW_OBS, SGX_STEP_ORIGINAL_CHANNELS, SGX_STEP_MASK_1D, SGX_STEP_MASK_STATE_COORDS. */
int sgx_observe(sgx_env *h, float *obs_dev, float *fobs_dev, uint8_t *mask_dev, int8_t *player_dev, int32_t flags, void *stream);

/* Average duration in microseconds of `launches` sgx_observe calls writing obs_dev / mask_dev (either may be NULL), measured
 * with HIP events on `stream`; synchronises that stream.  The caller owns the output buffers, and on MI355X the same kernel
 * runs 312-400 us depending on WHICH allocation the observation buffer is: a host allocates a few
 * candidates, times each with this call and keeps the fastest (what VecStrategoEnv.tune_placement does from Python).
 * No reference counterpart. */
int sgx_time_observe(sgx_env *h, float *obs_dev, uint8_t *mask_dev, int32_t launches, void *stream, float *microseconds);

/* Write-stream rate (GB/s) of the device memory range [ptr_dev, ptr_dev + bytes) under the step kernel's store pattern: one wave
 * per 26 KiB segment, 1 KiB non-temporal store instructions, eight concurrent fronts -- the observation stream without the game.
 * OVERWRITES the range.  On MI355X device memory comes in large regions of two kinds that differ by ~25 % under this pattern (and
 * not under a sequential fill); a host that allocates its own output tensors can tell with this call which kind
 * an allocation is.  ptr_dev 1 KiB aligned; `launches` timed launches after one untimed; synchronises `stream`.  No reference
 * counterpart. */
int sgx_mem_probe(int device, void *ptr_dev, int64_t bytes, int32_t launches, void *stream, float *gb_per_s);

/* The store stream of the step kernel WITHOUT the game, for pricing the roofline against what the memory takes from exactly this store
 * shape (bench.py: roofline.store_peak_measured).  One wave per `seg_bytes` segment (a multiple of 16; 26,800 = one 10x10 observation),
 * eight waves per 512-thread workgroup at 6 waves per SIMD, workgroups mapped to segments like games to workgroups (eight contiguous XCD
 * ranges); a wave sweeps its segment in 16-byte-per-lane store instructions on 1 KiB ADDRESS boundaries, whole 128-byte lines
 * non-temporal, the two edge lines a segment shares with its neighbours through L2 -- emit_codes' pattern (sgx_obs.h).  The range
 * [ptr_dev, ptr_dev + bytes) holds floor(bytes / seg_bytes) segments; ONE launch writes all of them `passes` times, pass after pass, so a
 * launch of passes x bytes >> 288 MB (L2 + Infinity Cache) leaves a negligible share of its bytes in the caches when it ends.
 * payload: 0 = zeros, 1 = observation-like floats (0 / 1 / -1 / 0.5 from a per-lane code pattern), 2 = incompressible bits (a hash of
 * the address and the launch number).  nt_stores: 0 = plain stores, 1 = whole lines non-temporal, N >= 2 = every N-th 1 KiB sweep plain and the
 * others non-temporal (the step kernel's own mix: its mask leaves as plain stores through L2).  How many store streams the memory sees at once is the probe's other axis -- the step kernel's waves
 * do not store back to back: waves_per_cu = resident waves per CU (0 = the kernel's own 24; 16 or 8: fewer resident workgroups), pace =
 * sleeps of 64 cycles after every 1 KiB sweep (0 .. 4096), persistent != 0 = a grid of the RESIDENT workgroups only, every wave walking
 * segment after segment for the whole launch (long-lived waves, like the multi-step kernel's; 0 = one short-lived workgroup per eight
 * segments and pass, like one launch per step); dwell > 1 (persistent waves only) = a wave writes its segment `dwell` times before it moves
 * on, cycling through `ring` (1 .. 8) equal sub-ranges of the buffer -- a wave of the multi-step kernel rewrites its game's observation
 * step after step, in place or into the sets of a ring; a launch then writes passes x dwell x (one sub-range's segments).  bench.py sweeps
 * these and reports the best rate of the non-rewriting configurations as roofline.store_peak_measured.
 * OVERWRITES the range.  Returns the average duration of `launches` timed launches (HIP events on `stream`, one untimed launch first;
 * synchronises the stream) and bytes written per launch / that time.  No reference counterpart. */
#define SGX_PROBE_ZEROS 0
#define SGX_PROBE_OBS_LIKE 1
#define SGX_PROBE_RANDOM 2
int sgx_store_probe(int device, void *ptr_dev, int64_t bytes, int32_t seg_bytes, int32_t passes, int32_t payload, int32_t nt_stores,
                    int32_t waves_per_cu, int32_t pace, int32_t persistent, int32_t dwell, int32_t ring, int32_t launches, void *stream,
                    float *microseconds_per_launch, float *gb_per_s);

/* Library-owned output buffers with a bounded placement trial.  On MI355X the same launch takes
 * 313-400 us depending on which physical memory backs the big observation buffer: device memory comes in regions of two
 * kinds, a buffer lying inside one region runs at that region's rate (~350 or ~380-395 us for 65,536 Barrage games), and
 * only a buffer whose pages MIX both kinds reaches the fast class (313-325 us).  Which one a plain allocation gets depends
 * on what was allocated before it.  sgx_alloc_outputs allocates the mask buffer, then tries up to `max_trials` candidate
 * allocations of the observation buffer (and of the fully-observable one with SGX_OUT_FULL_OBS): before each candidate a
 * padding allocation of growing size (steps of an eighth of the buffer, or -- under a budget wider than `max_trials` such steps -- of
 * the budget divided by the number of candidates, so that a generous budget is sampled evenly) is made and released again afterwards,
 * which moves the candidate to other buddy blocks;
 * each candidate is timed with a few sgx_observe launches (no state change) and only the fastest so far is kept.  The trial
 * stops early once the kept candidate is >= 17 % faster than the slowest one seen (the fast class), or eight candidates after the first one
 * that is >= 9 % faster.  At no time does the trial hold more than `max_extra_bytes`
 * beyond the buffers it returns (0 or max_trials <= 1: no trial, first allocation).  Channel counts follow `flags` (SGX_STEP_ORIGINAL_CHANNELS).  Waits for the device.  No reference counterpart. */
#define SGX_OUT_FULL_OBS 1024         /* also allocate fobs_dev [N,R,C,79] (or 33) */
#define SGX_OUT_MAX_TRIALS 64
typedef struct sgx_outputs {
    float *obs_dev;                /* [N,R,C,67] (32 with SGX_STEP_ORIGINAL_CHANNELS) */
    float *fobs_dev;               /* [N,R,C,79] (33) or NULL */
    uint8_t *mask_dev;             /* [N,R,C,K] */
    int64_t obs_bytes, fobs_bytes, mask_bytes;
    int64_t peak_extra_bytes;      /* most memory the trial held beyond the returned buffers */
    int32_t n_trials, n_ftrials;   /* candidates timed for obs_dev / fobs_dev */
    float trial_us[SGX_OUT_MAX_TRIALS];    /* sgx_observe launch time with each obs candidate; [0] = the plain first allocation */
    float ftrial_us[SGX_OUT_MAX_TRIALS];   /* the same for fobs_dev */
    int32_t device;                /* the device the buffers live on (sgx_free_outputs works without the handle) */
    int32_t reserved_;
} sgx_outputs;
int sgx_alloc_outputs(sgx_env *h, int32_t flags, int64_t max_extra_bytes, int32_t max_trials, void *stream, sgx_outputs *out);
/* A target for the searches that follow: with target_us > 0 sgx_alloc_outputs keeps trying candidates (inside its budget) until one is
 * within 3 % of the target instead of applying its own stop rules -- for a RING of output sets (sgx_step_ring), whose every set should
 * be as fast as the first one (target = the observe-launch time the first search kept); 0 (default) = the stop rules above. */
int sgx_set_placement_target(sgx_env *h, float target_us);
/* Frees the buffers of `out` (h may be NULL, also after sgx_destroy of the handle that allocated them: the buffers belong to
 * whoever holds the sgx_outputs -- the Python binding ties them to the tensors that view them). */
int sgx_free_outputs(sgx_env *h, sgx_outputs *out);

/* One batched env.step(): see sgx_step_io. */
int sgx_step(sgx_env *h, const sgx_step_io *io, void *stream);

/* Single-game latency (config 1: one game behind the reference's dict API).  sgx_host_alloc returns pinned host memory the device
 * can address (*dev_ptr is its device alias): with sgx_step_io's output pointers -- and actions_dev -- pointing into it the step
 * kernel writes a game's ~30 KB of outputs straight to host memory, and sgx_step_sync (= sgx_step + wait until the outputs are
 * complete) makes env.step() ONE library call: no upload, no download, no second launch.  Meant for a handful of games; batches
 * belong in HBM.  Up to 8 games on a board of more than 32 cells with a multiple of 4 cells, 'extended' channel modes, are played by a
 * kernel of their own -- one workgroup per game: one wave plays the move, all eight emit the mask and the observations -- which
 * publishes its completion in a host-mapped word that sgx_step_sync polls -- spinning for about a millisecond (a step is ~6 us), then
 * yielding the core between polls, then backing off to sleeps of 2 ... 64 us, looking at the stream so that a failed launch ends the
 * wait: a long wait costs the host next to nothing -- and the call returns when the outputs are visible to the host,
 * typically before `stream` has retired the kernel (later work on `stream` is ordered behind it as usual).  Every other case is
 * sgx_step + hipStreamSynchronize.  Same results either way (tests/test_gpu_step_sync.py).
 * No reference counterpart (the reference is one game per object on the host, maenv:659-828). */
int sgx_host_alloc(sgx_env *h, int64_t bytes, void **host_ptr, void **dev_ptr);
int sgx_host_free(sgx_env *h, void *host_ptr);
int sgx_step_sync(sgx_env *h, const sgx_step_io *io, void *stream);

/* n_steps consecutive sgx_step calls with the same buffers, enqueued back to back without returning to the host language (ONE launch for
 * all of them where the call is eligible: sgx_set_multi_step): the random-action game loop of examples/basic_game_loop.py:34-63 (sample a valid action, step, repeat) for N games.
 * Requires io->next_actions_dev == io->actions_dev, so that every step plays the action the previous one drew; the
 * output buffers hold the last step's results afterwards.  (Toy boards finish a batched step in tens of microseconds:
 * driving them one call at a time from Python is launch-bound.) */
int sgx_step_n(sgx_env *h, const sgx_step_io *io, int32_t n_steps, void *stream);

/* sgx_step_n over a RING of output sets: step i of the call writes its outputs through ios[(first_set + i) % n_sets] -- a rollout
 * into a trajectory buffer that keeps the last n_sets steps' observations and masks (what a learner stores per step) instead of
 * overwriting one set in place.  Every set must name the same actions_dev == next_actions_dev (the one chain of drawn actions) and the
 * same auto_reset / flags.  With n_sets sets of a size whose sum exceeds the 256 MiB Infinity Cache, no line written by one step can
 * still sit in that cache when it is written again: bench.py's DRAM-side roofline figure (roofline.frac_dram) is measured this way.
 * Same game trajectories as sgx_step_n (tests/test_gpu_parity.py).  No reference counterpart beyond the loop of
 * examples/basic_game_loop.py:34-63. */
int sgx_step_ring(sgx_env *h, const sgx_step_io *ios, int32_t n_sets, int32_t first_set, int32_t n_steps, void *stream);

/* sgx_step_n into a TRAJECTORY buffer: step t of the call (t = 0 .. n_steps-1) writes its outputs into slot (first_slot + t) % n_slots of
 * tensors that carry a leading slot axis -- obs float32 [n_slots][slot_envs][R][C][67], mask uint8 [n_slots][slot_envs][R][C][K], ... --
 * named by the pointers of slot 0 (`io`) and ONE stride, `slot_envs` (>= N; = N for dense [T][N]... tensors).  This is the batched
 * counterpart of what the reference's caller gets: a FRESH observation / mask array from every env.step() (impl:905, maenv:447-497), so a
 * learner that keeps T steps keeps T arrays.  Unlike sgx_step_ring the number of slots is not limited (n_slots = n_steps = 64 ... 1024 is the
 * intended use), and with results_per_slot != 0 the per-step results are kept as well: reward float32 [n_slots][slot_envs][2], done /
 * invalid_action / ending_invalid uint8 [n_slots][slot_envs], player int8 [n_slots][slot_envs] (0: they are overwritten in place like
 * sgx_step_n's).  actions_log_dev (nullable) int32 [n_slots][slot_envs] receives the action every env DREW in the step -- the action the
 * next step plays, i.e. the action chosen from the observation / mask of the same slot: (obs_t, mask_t, action_t) line up, and
 * reward / done of slot t + 1 are what that action earned.  final_obs_dev / final_fobs_dev have no slot axis (they are written on
 * terminal steps only).  io->actions_dev == io->next_actions_dev (the chain of drawn actions, int32 [N], no slot axis); tensors that are
 * NULL in `io` are skipped.  Where the call is eligible for the multi-step kernels (sgx_set_multi_step) all steps run in ONE launch
 * (launches of at most 256 steps); everything else takes one launch per step with the same results (tests/test_gpu_trajectory.py compares
 * every slot with the oracle stepped alongside).  Compact outputs: the slot strides are those of the compact tensors
 * (uint8 [n_slots][slot_envs][sgx_compact_obs_stride], uint32 [n_slots][slot_envs][sgx_compact_mask_words]). */
typedef struct sgx_traj_io {
    sgx_step_io io;               /* the tensors of slot 0 */
    int32_t n_slots;              /* >= 1 */
    int32_t results_per_slot;     /* 0: reward / done / player / invalid_action / ending_invalid in place; != 0: [n_slots][slot_envs]... */
    int64_t slot_envs;            /* envs per slot of every tensor that has a slot axis, >= N */
    int32_t *actions_log_dev;     /* [n_slots][slot_envs] out (nullable): the action drawn in each step */
} sgx_traj_io;
int sgx_step_traj(sgx_env *h, const sgx_traj_io *t, int32_t first_slot, int32_t n_steps, void *stream);

/* sgx_step_n with the batch split into `chains` (1..SGX_MAX_CHAINS) contiguous ranges of games, each range playing its n_steps on a
 * stream of its own: games never interact, so the ranges' launches may overlap, and the ramp-up / drain of one range's step is filled
 * by the other's (a launch that lasts tens of microseconds spends a third of its time with the chip half empty).  The caller's
 * stream waits for all chains; results are identical to sgx_step_n.  Measured with chains = 2 on 65,536 games: Micro 41.3 -> 37.2 us per
 * step of all games, 5x5 107.6 -> 96.0 us, 8x8 198 -> 183 us, Barrage 322 -> 304 us (docs/DESIGN_rounds_1-3.md).  chains = 0 lets the
 * library choose by the rule measured on the current kernels (2 for boards of up to 36 cells and for boards whose cell count is no
 * multiple of 4, else 1: 8x8 and 10x10 launches already stream at the memory rate and lose 2-5 % to a second chain) -- after trying
 * the multi-step launch of sgx_set_multi_step, which is faster than any number of chains wherever the call is eligible.  No reference
 * counterpart. */
#define SGX_MAX_CHAINS 4
int sgx_rollout(sgx_env *h, const sgx_step_io *io, int32_t n_steps, int32_t chains, void *stream);

/* sample_random_valid_action (maenv:830-834) for a batch of masks laid out as mask_dev of sgx_step_io:
 * picks the k-th set byte, k drawn with the same counter RNG as next_actions_dev (identical results). */
int sgx_sample_valid(sgx_env *h, const uint8_t *mask_dev, int32_t *actions_dev, void *stream);

/* nnet_choose_action_example (examples/basic_game_loop.py:6-31 with softmax of examples/util.py:4-47) for a batch, on the device: the
 * caller's policy writes logits float32 [N][R*C*K] in the shape of the valid-actions mask; invalid actions get probability 0, the rest
 * softmax((logits - max) / temperature), and ONE action per game is drawn from it -- one wave per game, logits and mask read once.
 * mask_dev: the mask the step kernel wrote, uint8 [N][R*C*K], or with flags = SGX_STEP_COMPACT_MASK the bit mask uint32
 * [N][sgx_compact_mask_words] of a compact step.  temperature: a divisor, > 0; 0 = argmax (ties drawn uniformly).
 * The draw is keyed like next_actions_dev -- counter RNG on (seed, global env id, game number, turn) -- and sampling is fixed point:
 * weight of a valid action = floor(exp2((logit - max) * log2(e) / temperature + 23)), i.e. 2^23 for the maximum, exact integer sums,
 * inverse CDF in ascending action order.  A result depends on nothing but (logits, mask, seed, game, turn), and with equal logits it IS
 * sgx_sample_valid's action.  The truncation means an action whose probability relative to the most likely one is below 2^-23
 * (~1.2e-7) has weight 0 and is never drawn.  NaN logits count as -inf.  actions_dev[i] = -1 where NO action can be drawn: the mask
 * is empty, or every valid action's logit is -inf / NaN -- the caller must not feed that -1 to sgx_step (it is an invalid action); a
 * terminal env's mask holds the no-op only, which is drawn like any other action.
 * No policy network lives in this library: the logits are the caller's. */
int sgx_choose_actions(sgx_env *h, const float *logits_dev, const void *mask_dev, float temperature, int32_t flags, int32_t *actions_dev, void *stream);

/* INTERNAL_STATE observation component / reset(initial_state_override=...) (maenv:494-495, 551-553):
 * convert between the library's packed int8 state and the reference's int64 [N,34,R,C] layout
 * (absolute coordinates).  player_dev int8 [N] = current mover (nullable on export; NULL on import = +1). */
int sgx_export_state(sgx_env *h, int64_t *state_dev, int8_t *player_dev, void *stream);
int sgx_import_state(sgx_env *h, const int64_t *state_dev, const int8_t *player_dev, void *stream);
/* The same import with a report: the reference's pure functions accept ANY int64 [34,R,C] (impl:399-517, 894-1045), the packed
 * record only what play can produce.  sanitised_dev uint8 [N] (nullable) is set to 1 for every state the import had to alter:
 * a value outside its layer's range (-> 0), more than two non-zero recent-move cells of one player (the rest dropped), more
 * captured pieces than 2 x pieces per side (the surplus dropped), an obstacle layer that differs from the variant's (the
 * variant's is used), a player that is not +1 / -1; 0 otherwise -- results for flagged states may differ from the reference's. */
int sgx_import_state_checked(sgx_env *h, const int64_t *state_dev, const int8_t *player_dev, uint8_t *sanitised_dev, void *stream);

/* The functional operator API on caller-provided int64 states in one call: sgx_import_state_checked -> sgx_step -> sgx_export_state
 * (state_out_dev NULL: no export -- is_move_valid_*, masks and observations of the given states), i.e. get_next_state (penv:148-155)
 * for a batch.  The batch is split into `chains` (1..SGX_MAX_CHAINS) ranges of states on streams of their own, so that one range's
 * 27 KB-per-state reads overlap another's writes; the caller's stream waits for all of them.  io as in sgx_step with the SGX_STEP_*
 * flags of the functional API; no auto_reset, no next_actions_dev; io->actions_dev NULL = sgx_observe of the given states (their
 * movers' masks / observations, nothing is played).  The handle's own states are overwritten.  On boards of more than 32 cells with
 * the 67-channel observation kind the three steps are ONE launch (a 128-thread block per state, the packed record never leaves
 * LDS): 65,536 Barrage states 745 -> ~640 us; `chains` only matters on the other paths.
 * NO ALIASING while the general-state pass is on (the default, see sgx_set_general_states; boards of up to 256 cells): that pass reads
 * state_in_dev / player_in_dev again after the outputs have been written, so state_out_dev must not overlap state_in_dev and
 * player_out_dev must not overlap player_in_dev -- SGX_EINVAL otherwise.  With the pass off, stepping a batch in place is fine (every
 * state is read completely before its successor is written). */
int sgx_step_states(sgx_env *h, const int64_t *state_in_dev, const int8_t *player_in_dev, uint8_t *sanitised_dev,
                    const sgx_step_io *io, int64_t *state_out_dev, int8_t *player_out_dev, int32_t chains, void *stream);

/* General states in sgx_step_states.  The reference's pure functions accept ANY int64 [34,R,C] (penv:74-155); the packed record carries what
 * play can produce.  On boards of up to 256 cells sgx_step_states runs a second pass over the states its import had to alter (behind the
 * one-launch path above, or behind the three launches of the other paths: boards of up to 32 cells, the 79-channel and 'original'
 * observation kinds): they are redone from the caller's int64 input on a general-state variant of the kernels -- dense recent-move
 * layers, a capture event for every (layer, cell) pair, counts up to 32,768, captured-count channels beyond the 16-entry table
 * normalised by the reference's own float32 arithmetic (maenv:506-508) -- so that get_next_state, is_move_valid_*, the masks and every
 * observation kind of such states equal the reference's (not covered: boards of more than 256 cells; state-coordinate masks requested
 * together with a 79-channel or 'original' observation); sanitised_dev then reports only what still had to be altered
 * (values outside their layer's range, an obstacle layer that differs from the variant's, a player that is not +1 / -1).  mode 1 (default):
 * on; 0: off (flagged states keep the first pass' sanitised results).  SGX_GENERAL_STATES=0|1 sets the default of handles created
 * afterwards.  Costs one launch of early-exit blocks (~2 % of a get_next_state call) when no state is flagged. */
int sgx_set_general_states(sgx_env *h, int32_t mode);

/* Search callers (MCTS on get_next_state, penv:148-155) keep their nodes in the packed records instead of paying the 27 KB
 * int64 import / export per state: handles of the same variant on the same device act as node pools.
 * sgx_copy_envs: records src[src_index[i]] -> dst[dst_index[i]] for i < n (an index array may be NULL = identity).
 * sgx_expand: one env.step() per env i of `dst`, reading the game from record src_index[i] of `src` (i when NULL) and writing
 * the successor to record i of `dst`; where the action is invalid (invalid_action[i] = 1) record i becomes a copy of the
 * parent.  `io` as in sgx_step (flags SGX_STEP_ACTIONS_1D / _POSITIONS / _ALLOW_OSCILLATION / _RAW_OBS / _MASK_*; no auto_reset,
 * no fully-observable / original-channel outputs); src == dst with a NULL index is sgx_step.  Inside ONE handle an indexed
 * copy / expansion would race (a wave reads a record another wave of the same launch rewrites): sgx_expand with src == dst and
 * a non-NULL index, and sgx_copy_envs with src == dst and any index, return SGX_EINVAL -- use a second handle as the target.
 * No reference counterpart beyond get_next_state itself. */
int sgx_copy_envs(sgx_env *dst, const int32_t *dst_index_dev, sgx_env *src, const int32_t *src_index_dev, int64_t n, void *stream);
int sgx_expand(sgx_env *dst, sgx_env *src, const int32_t *src_index_dev, const sgx_step_io *io, void *stream);

/* Per-env bookkeeping: int32 [N][4] = {turn count, game number, game_over, current player}. */
int sgx_get_env_info(sgx_env *h, int32_t *info_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STRATEGO_MI355X_H */
