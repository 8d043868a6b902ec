// Prototype of a library-owned output allocation: the observation buffer is mapped from physical chunks picked out of a small
// pool (buffer + `extra` MiB), assemblies are timed with sgx_observe and the fastest is kept.
//   usage: pool_search <extra MiB> ; prints one summary line per chunk size
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;
static int8_t *player_d;
static sgx_env *h;
static unsigned rs = 99991;
static unsigned rnd() { rs = rs * 1664525u + 1013904223u; return rs >> 8; }

static float time_observe(float *obs, uint8_t *mask, int reps = 6) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask, player_d, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask, player_d, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1000.f;
}

int main(int argc, char **argv) {
    const int64_t N = 65536;
    const size_t MB = 1u << 20, extra = (argc > 1 ? atoll(argv[1]) : 4096) * MB;
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); return 1; }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    const size_t bytes = (size_t)N * 100 * 67 * 4, mbytes = (size_t)N * 3700;
    CK(hipMalloc((void **)&mask_d, mbytes)); CK(hipMalloc((void **)&player_d, N));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    {
        float *p[3];
        printf("hipMalloc x3:");
        for (int i = 0; i < 3; i++) { CK(hipMalloc((void **)&p[i], bytes)); printf(" %.1f", time_observe(p[i], mask_d)); }
        for (int i = 0; i < 3; i++) CK(hipFree(p[i]));
        printf("\n");
    }
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t xcd_range = (size_t)(N / 8) * 100 * 67 * 4;
    for (size_t cs : {32 * MB, 64 * MB, 128 * MB, xcd_range, 256 * MB}) {
        const int n = (int)((bytes + cs - 1) / cs), pool = n + (int)(extra / cs);
        std::vector<hipMemGenericAllocationHandle_t> hs(pool);
        for (int i = 0; i < pool; i++) CK(hipMemCreate(&hs[i], cs, &prop, 0));
        void *vap = nullptr;
        CK(hipMemAddressReserve(&vap, (size_t)n * cs, 2u << 20, nullptr, 0));
        char *va = (char *)vap;
        int trials = 0;
        auto timed = [&](const std::vector<int> &ids) {
            for (int i = 0; i < n; i++) CK(hipMemMap(va + (size_t)i * cs, cs, 0, hs[ids[i]], 0));
            CK(hipMemSetAccess(va, (size_t)n * cs, &acc, 1));
            const float t = time_observe((float *)va, mask_d);
            CK(hipDeviceSynchronize());
            for (int i = 0; i < n; i++) CK(hipMemUnmap(va + (size_t)i * cs, cs));
            ++trials;
            return t;
        };
        hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        CK(hipEventRecord(w0, nullptr));
        std::vector<int> ids(n), best_ids;
        for (int i = 0; i < n; i++) ids[i] = i;
        const float t_order = timed(ids);
        for (int i = 0; i < n; i++) ids[i] = (int)((long)i * pool / n);
        const float t_spread = timed(ids);
        float best = t_order < t_spread ? t_order : t_spread, rmin = 1e9, rmax = 0, rsum = 0;
        if (t_order <= t_spread) for (int i = 0; i < n; i++) ids[i] = i;
        best_ids = ids;
        const int R = 12;
        for (int rep = 0; rep < R; rep++) {
            std::vector<int> all(pool);
            for (int i = 0; i < pool; i++) all[i] = i;
            for (int i = pool - 1; i > 0; i--) std::swap(all[i], all[rnd() % (i + 1)]);
            for (int i = 0; i < n; i++) ids[i] = all[i];
            const float t = timed(ids);
            rmin = std::min(rmin, t); rmax = std::max(rmax, t); rsum += t;
            if (t < best) { best = t; best_ids = ids; }
        }
        const float after_random = best;
        // greedy refinement: 40 single-position swaps with an unused chunk, keep improvements
        std::vector<char> used(pool, 0);
        for (int i = 0; i < n; i++) used[best_ids[i]] = 1;
        for (int k = 0; k < 40 && pool > n; k++) {
            const int pos = rnd() % n;
            int c; do { c = rnd() % pool; } while (used[c]);
            ids = best_ids; ids[pos] = c;
            const float t = timed(ids);
            if (t < best - 0.5f) { used[best_ids[pos]] = 0; used[c] = 1; best = t; best_ids = ids; }
        }
        const float confirm = timed(best_ids);
        CK(hipEventRecord(w1, nullptr)); CK(hipEventSynchronize(w1));
        float wall; CK(hipEventElapsedTime(&wall, w0, w1));
        printf("chunk %4zu MiB: n %3d pool %3d | in order %.1f spread %.1f | %d random: min %.1f mean %.1f max %.1f | best after random %.1f, after greedy %.1f, confirmed %.1f | %d trials %.0f ms\n",
               cs / MB, n, pool, t_order, t_spread, R, rmin, rsum / R, rmax, after_random, best, confirm, trials, wall);
        fflush(stdout);
        for (int i = 0; i < pool; i++) CK(hipMemRelease(hs[i]));
        CK(hipMemAddressFree(vap, (size_t)n * cs));
    }
    printf("done\n");
    return 0;
}
