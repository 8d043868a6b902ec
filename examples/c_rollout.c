/*
 * c_rollout.c -- the C ABI of libstratego_mi355x.so used from plain C (no Python, no torch):
 * N concurrent Barrage games, random-valid-action rollout with auto-reset, the reference's
 * basic_game_loop (stratego_env/examples/basic_game_loop.py:34-63) for a batch.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_rollout.c \
 *       -Lstratego_env_amd/_build -lstratego_mi355x -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/stratego_env_amd/_build -Wl,-rpath,/opt/rocm/lib -o examples/c_rollout
 *   examples/c_rollout stratego_env_amd/inits/barrage_setups.npy [n_envs] [steps] [seed] [bench | traj]
 *
 * With `bench` as the fifth argument the steps are enqueued by one sgx_step_n call, nothing is copied to the host, and the program
 * prints env steps per second measured with HIP events (the bench.py figure, from C).  With `traj` the steps are ONE sgx_step_traj call
 * into a trajectory buffer [steps][n_envs]... (every step's outputs kept: the reference returns a fresh observation array per step,
 * impl:905) and the digest is computed from env 0's slots afterwards -- the same digest as the step-by-step mode.
 *
 * Prints games finished, invalid actions (must be 0) and the rolling FNV-1a digest of env 0's outputs
 * (mask, observation, rewards, done/player/ending_invalid), which tests/test_gpu_c_example.py compares with the oracle.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stratego_mi355x.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SGX_TRY(x) do { int rc_ = (x); if (rc_ != SGX_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sgx_last_error()); return 3; } } while (0)

static uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

/* the packed Gravon table: .npy v1/v2, uint8 [n][20], two piece codes per byte -> uint8 [n][40] (util.py:154-180 decoded) */
static uint8_t *load_setups(const char *path, int64_t *n_out) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    uint8_t hdr[12];
    if (fread(hdr, 1, 10, f) != 10 || memcmp(hdr, "\x93NUMPY", 6) != 0) { fclose(f); return NULL; }
    size_t hlen = hdr[8] | (hdr[9] << 8);
    if (hdr[6] >= 2) { if (fread(hdr + 10, 1, 2, f) != 2) { fclose(f); return NULL; } hlen |= ((size_t)hdr[10] << 16) | ((size_t)hdr[11] << 24); }
    fseek(f, 0, SEEK_END);
    long end = ftell(f);
    long data0 = (hdr[6] >= 2 ? 12 : 10) + (long)hlen;
    int64_t n = (end - data0) / 20;
    uint8_t *packed = (uint8_t *)malloc((size_t)n * 20), *out = (uint8_t *)malloc((size_t)n * 40);
    fseek(f, data0, SEEK_SET);
    if (fread(packed, 20, (size_t)n, f) != (size_t)n) { fclose(f); free(packed); free(out); return NULL; }
    fclose(f);
    for (int64_t i = 0; i < n * 20; i++) { out[2 * i] = packed[i] & 15; out[2 * i + 1] = packed[i] >> 4; }
    free(packed);
    *n_out = n;
    return out;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s barrage_setups.npy [n_envs] [steps] [seed]\n", argv[0]); return 1; }
    const int64_t N = argc > 2 ? atoll(argv[2]) : 4096;
    const int steps = argc > 3 ? atoi(argv[3]) : 256;
    const uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 0) : 0x5EEDull;

    /* BARRAGE_STRATEGO_CONFIG (game/config.py:183-203): 10x10, 8 pieces per side, two 2x2 lakes, max_turns 1000 */
    sgx_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1;  /* spy */      cfg.piece_counts[1] = 2;   /* scout */   cfg.piece_counts[2] = 1;  /* miner */
    cfg.piece_counts[8] = 1;  /* general */  cfg.piece_counts[9] = 1;   /* marshall */
    cfg.piece_counts[10] = 1; /* flag */     cfg.piece_counts[11] = 1;  /* bomb */
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (int i = 0; i < 8; i++) cfg.obstacles[lakes[i][0] * 10 + lakes[i][1]] = 1;

    int64_t n_setups = 0;
    uint8_t *table = load_setups(argv[1], &n_setups);
    if (!table) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }

    sgx_env *h = NULL;
    SGX_TRY(sgx_create(&cfg, N, 0, seed, 0, &h));
    SGX_TRY(sgx_set_setup_table(h, table, n_setups));
    const int K = sgx_spatial_channels(h);
    const size_t obs_n = (size_t)100 * SGX_PO_OBS_CHANNELS, mask_n = (size_t)100 * K;

    float *obs, *reward;
    uint8_t *mask, *done, *invalid, *ending_invalid;
    int8_t *player;
    int32_t *actions;
    HIP_OK(hipMalloc((void **)&obs, N * obs_n * sizeof(float)));
    HIP_OK(hipMalloc((void **)&mask, N * mask_n));
    HIP_OK(hipMalloc((void **)&reward, N * 2 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&done, N));
    HIP_OK(hipMalloc((void **)&invalid, N));
    HIP_OK(hipMalloc((void **)&ending_invalid, N));
    HIP_OK(hipMalloc((void **)&player, N));
    HIP_OK(hipMalloc((void **)&actions, N * sizeof(int32_t)));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    SGX_TRY(sgx_reset(h, NULL, NULL, NULL, stream));                       /* reset(): sampled Gravon setups */
    SGX_TRY(sgx_observe(h, obs, NULL, mask, player, 0, stream));           /* _get_current_obs */
    SGX_TRY(sgx_sample_valid(h, mask, actions, stream));                   /* sample_random_valid_action */

    sgx_step_io io;
    memset(&io, 0, sizeof(io));
    io.actions_dev = actions; io.obs_dev = obs; io.mask_dev = mask; io.reward_dev = reward; io.done_dev = done;
    io.player_dev = player; io.invalid_action_dev = invalid; io.ending_invalid_dev = ending_invalid;
    io.next_actions_dev = actions;      /* the step draws each env's next action itself */
    io.auto_reset = 1;

    if (argc > 5 && strcmp(argv[5], "traj") == 0) {
        /* the same rollout as ONE call: slot t of the trajectory tensors receives step t (sgx_step_traj: all steps in one launch per 256) */
        const int T = steps;
        float *tobs, *trew;
        uint8_t *tmask, *tdone, *tinv, *tend;
        int8_t *tpl;
        int32_t *tact;
        HIP_OK(hipMalloc((void **)&tobs, (size_t)T * N * obs_n * sizeof(float)));
        HIP_OK(hipMalloc((void **)&tmask, (size_t)T * N * mask_n));
        HIP_OK(hipMalloc((void **)&trew, (size_t)T * N * 2 * sizeof(float)));
        HIP_OK(hipMalloc((void **)&tdone, (size_t)T * N));
        HIP_OK(hipMalloc((void **)&tinv, (size_t)T * N));
        HIP_OK(hipMalloc((void **)&tend, (size_t)T * N));
        HIP_OK(hipMalloc((void **)&tpl, (size_t)T * N));
        HIP_OK(hipMalloc((void **)&tact, (size_t)T * N * sizeof(int32_t)));
        sgx_traj_io tr;
        memset(&tr, 0, sizeof(tr));
        tr.io = io;
        tr.io.obs_dev = tobs; tr.io.mask_dev = tmask; tr.io.reward_dev = trew; tr.io.done_dev = tdone; tr.io.invalid_action_dev = tinv;
        tr.io.ending_invalid_dev = tend; tr.io.player_dev = tpl;
        tr.n_slots = T; tr.results_per_slot = 1; tr.slot_envs = N; tr.actions_log_dev = tact;
        SGX_TRY(sgx_step_traj(h, &tr, 0, T, stream));
        HIP_OK(hipStreamSynchronize(stream));
        float *obs_h = (float *)malloc(obs_n * sizeof(float));
        uint8_t *mask_h = (uint8_t *)malloc(mask_n), *flag_h = (uint8_t *)malloc((size_t)N);
        uint64_t digest = 0xCBF29CE484222325ull;
        long long finished = 0, invalid_total = 0;
        for (int t = 0; t < T; t++) {
            float rw[2]; uint8_t ei; int8_t pl; uint8_t d0;
            HIP_OK(hipMemcpy(obs_h, tobs + (size_t)t * N * obs_n, obs_n * sizeof(float), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(mask_h, tmask + (size_t)t * N * mask_n, mask_n, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(rw, trew + (size_t)t * N * 2, sizeof(rw), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(&ei, tend + (size_t)t * N, 1, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(&pl, tpl + (size_t)t * N, 1, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(flag_h, tdone + (size_t)t * N, (size_t)N, hipMemcpyDeviceToHost));
            d0 = flag_h[0];
            for (int64_t i = 0; i < N; i++) finished += flag_h[i];
            HIP_OK(hipMemcpy(flag_h, tinv + (size_t)t * N, (size_t)N, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < N; i++) invalid_total += flag_h[i];
            const int32_t tail[4] = {d0, pl, ei, 0};
            digest = fnv1a(digest, mask_h, mask_n);
            digest = fnv1a(digest, obs_h, obs_n * sizeof(float));
            digest = fnv1a(digest, rw, sizeof(rw));
            digest = fnv1a(digest, tail, sizeof(tail));
        }
        printf("trajectory of %d slots: envs %lld steps %d seed 0x%llx games_finished %lld invalid_actions %lld env0_digest 0x%016llx launch_kind %d\n", T,
               (long long)N, steps, (unsigned long long)seed, finished, invalid_total, (unsigned long long)digest, sgx_last_launch_kind(h));
        SGX_TRY(sgx_destroy(h));
        return invalid_total == 0 ? 0 : 4;
    }
    if (argc > 5) {
        /* which physical memory backs the observation buffer decides up to 20 % of a step's time: let the
         * library pick its output buffers with its bounded placement trial (never more than 8 GiB held beyond what it returns) */
        sgx_outputs out;
        SGX_TRY(sgx_alloc_outputs(h, 0, (int64_t)8 << 30, 32, stream, &out));
        io.obs_dev = out.obs_dev;
        io.mask_dev = out.mask_dev;
        printf("placement trial: %d candidates, first %.1f us", out.n_trials, out.n_trials ? out.trial_us[0] : 0.f);
        float best = out.n_trials ? out.trial_us[0] : 0.f;
        for (int c = 1; c < out.n_trials; c++) best = out.trial_us[c] < best ? out.trial_us[c] : best;
        printf(", kept %.1f us, peak extra %.2f GiB\n", best, (double)out.peak_extra_bytes / (1 << 30));
        SGX_TRY(sgx_observe(h, io.obs_dev, NULL, io.mask_dev, player, 0, stream));
        SGX_TRY(sgx_sample_valid(h, io.mask_dev, actions, stream));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        SGX_TRY(sgx_step_n(h, &io, 32, stream));                           /* warm-up */
        HIP_OK(hipEventRecord(e0, stream));
        SGX_TRY(sgx_step_n(h, &io, steps, stream));
        HIP_OK(hipEventRecord(e1, stream));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        printf("envs %lld steps %d: %.1f us per batched step, %.1f M env steps/s\n", (long long)N, steps, ms / steps * 1e3,
               (double)N * steps / ms / 1e3);
        SGX_TRY(sgx_destroy(h));
        return 0;
    }
    float *obs_h = (float *)malloc(obs_n * sizeof(float));
    uint8_t *mask_h = (uint8_t *)malloc(mask_n), *done_h = (uint8_t *)malloc((size_t)N), *inv_h = (uint8_t *)malloc((size_t)N);
    uint64_t digest = 0xCBF29CE484222325ull;
    long long finished = 0, invalid_total = 0;
    for (int t = 0; t < steps; t++) {
        SGX_TRY(sgx_step(h, &io, stream));
        float rw[2]; uint8_t ei; int8_t pl;
        HIP_OK(hipMemcpyAsync(obs_h, obs, obs_n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(mask_h, mask, mask_n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(done_h, done, (size_t)N, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(inv_h, invalid, (size_t)N, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(rw, reward, sizeof(rw), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(&ei, ending_invalid, 1, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(&pl, player, 1, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int64_t i = 0; i < N; i++) { finished += done_h[i]; invalid_total += inv_h[i]; }
        const int32_t tail[4] = {done_h[0], pl, ei, 0};
        digest = fnv1a(digest, mask_h, mask_n);
        digest = fnv1a(digest, obs_h, obs_n * sizeof(float));
        digest = fnv1a(digest, rw, sizeof(rw));
        digest = fnv1a(digest, tail, sizeof(tail));
    }
    printf("envs %lld steps %d seed 0x%llx games_finished %lld invalid_actions %lld env0_digest 0x%016llx\n", (long long)N, steps,
           (unsigned long long)seed, finished, invalid_total, (unsigned long long)digest);
    SGX_TRY(sgx_destroy(h));
    return invalid_total == 0 ? 0 : 4;
}
