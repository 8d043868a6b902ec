// Round 3, third probe: is the speed class a property of the physical CHUNKS a buffer is made of (additive), of their mix, or of
// neither?  K chunks of 256 MiB created one after the other and held.
//   A1  every chunk mapped 7 times in a row (1.79 GB of virtual range over 256 MiB of memory) under the 65,536-game env with
//       non-temporal stores forced: does a chunk show a kind of its own when the Infinity Cache cannot absorb the stream?
//   A2  buffers from chunks [i .. i+6], i = 0 .. K-7 (sliding window over the creation order)
//   A3  40 random 7-subsets
//   A4  greedy: start from chunks 0..6, replace one position at a time by the best unused chunk
//   hipcc -O2 -I include tools/microbench/mix_probe3.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/mix_probe3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;
static float time_observe(sgx_env *h, float *obs, uint8_t *mask, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask, nullptr, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask, nullptr, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1000.f;
}
static hipMemAllocationProp dev_prop() {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    return prop;
}
static void set_rw(void *va, size_t n) {
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n, &acc, 1));
}
static sgx_env *make_env(int64_t N) {
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    return h;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 64;
    const size_t MB = 1u << 20, CB = 256 * MB;
    const int64_t N = 65536;
    const int PER = 7;
    sgx_env *h = make_env(N);
    CK(hipMalloc((void **)&mask_d, (size_t)N * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(K);
    for (int i = 0; i < K; i++) CK(hipMemCreate(&ch[i], CB, &prop, 0));
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, PER * CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    auto timed = [&](const int *ids, bool with_mask = true) {
        for (int i = 0; i < PER; i++) CK(hipMemMap(va + (size_t)i * CB, CB, 0, ch[ids[i]], 0));
        set_rw(va, PER * CB);
        const float t = time_observe(h, (float *)va, with_mask ? mask_d : nullptr);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < PER; i++) CK(hipMemUnmap(va + (size_t)i * CB, CB));
        return t;
    };

    sgx_set_nt_stores(h, 1);
    printf("A1 chunk mapped 7x, obs only, NT forced (us):\n");
    for (int c = 0; c < K; c++) { int ids[PER]; for (int i = 0; i < PER; i++) ids[i] = c; printf(" %.1f", timed(ids, false)); fflush(stdout); }
    printf("\nA1b the same with plain stores:\n");
    sgx_set_nt_stores(h, 0);
    for (int c = 0; c < K; c += 4) { int ids[PER]; for (int i = 0; i < PER; i++) ids[i] = c; printf(" %.1f", timed(ids, false)); fflush(stdout); }
    sgx_set_nt_stores(h, -1);
    printf("\nA2 sliding window [i..i+6] (us):\n");
    for (int s = 0; s + PER <= K; s++) { int ids[PER]; for (int i = 0; i < PER; i++) ids[i] = s + i; printf(" %.1f", timed(ids)); fflush(stdout); }
    printf("\nA3 random subsets:\n");
    unsigned seed = 99;
    for (int rep = 0; rep < 40; rep++) {
        std::vector<int> all(K);
        for (int i = 0; i < K; i++) all[i] = i;
        for (int i = K - 1; i > 0; i--) { seed = seed * 1664525u + 1013904223u; std::swap(all[i], all[(seed >> 8) % (i + 1)]); }
        printf("  [");
        for (int i = 0; i < PER; i++) printf("%d ", all[i]);
        printf("] %.1f\n", timed(all.data())); fflush(stdout);
    }
    printf("A4 greedy from 0..6:\n");
    {
        int ids[PER];
        for (int i = 0; i < PER; i++) ids[i] = i;
        float best = timed(ids);
        printf("  start %.1f\n", best);
        std::vector<char> used(K, 0);
        for (int i = 0; i < PER; i++) used[i] = 1;
        for (int pos = 0; pos < PER; pos++) {
            int keep = ids[pos];
            printf("  pos %d:", pos);
            for (int c = 0; c < K; c += 3) {
                if (used[c]) continue;
                const int old = ids[pos];
                ids[pos] = c;
                const float t = timed(ids);
                printf(" %d:%.0f", c, t);
                if (t < best - 1.5f) { best = t; keep = c; }
                ids[pos] = old;
            }
            used[ids[pos]] = 0; ids[pos] = keep; used[keep] = 1;
            printf(" -> keep %d best %.1f\n", keep, best); fflush(stdout);
        }
    }
    printf("done\n");
    return 0;
}
