// Round 3: does sgx_mem_probe (a compute-free store-pattern kernel) see the same classes as the real kernel?  Held hipMalloc'ed
// 65,536-game observation buffers: the real kernel's obs-only launch time, the probe's rate over the whole buffer and over its
// 256 MiB windows; then 256 MiB VMM chunks: probe rate of each, and buffers assembled from the fastest / slowest chunks.
//   hipcc -O2 -I include tools/microbench/probe_check.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/probe_check
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

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
static float probe(void *p, size_t n, int launches = 3) {
    float g = 0;
    if (sgx_mem_probe(0, p, (int64_t)n, launches, nullptr, &g)) { printf("%s\n", sgx_last_error()); exit(1); }
    return g;
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
    const int NB = argc > 1 ? atoi(argv[1]) : 12, K = argc > 2 ? atoi(argv[2]) : 96;
    const int64_t N = 65536;
    const size_t MB = 1u << 20, CB = 256 * MB;
    sgx_env *h = make_env(N);
    uint8_t *mask_d;
    CK(hipMalloc((void **)&mask_d, (size_t)N * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    const size_t bytes = (size_t)N * 26800;
    std::vector<float *> held;
    for (int i = 0; i < NB; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); held.push_back(p); }
    printf("Q1 held hipMalloc buffers: real obs+mask us | real obs-only us | probe GB/s whole | probe GB/s of 256 MiB windows\n");
    for (int i = 0; i < NB; i++) {
        printf("  %2d  %6.1f  %6.1f  %7.0f  |", i, time_observe(h, held[i], mask_d), time_observe(h, held[i], nullptr), probe(held[i], bytes));
        for (size_t off = 0; off + CB <= bytes; off += CB) printf(" %5.0f", probe((char *)held[i] + off, CB));
        printf("\n"); fflush(stdout);
    }
    for (auto p : held) CK(hipFree(p));

    printf("Q2 %d VMM chunks of 256 MiB (held): probe GB/s each, three passes\n", K);
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(K);
    for (int i = 0; i < K; i++) CK(hipMemCreate(&ch[i], CB, &prop, 0));
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, 7 * CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    std::vector<float> rate(K, 0.f);
    for (int pass = 0; pass < 3; pass++) {
        for (int i = 0; i < K; i++) {
            CK(hipMemMap(va, CB, 0, ch[i], 0));
            set_rw(va, CB);
            const float g = probe(va, CB);
            rate[i] += g / 3;
            printf(" %4.0f", g);
            CK(hipMemUnmap(va, CB));
        }
        printf("\n"); fflush(stdout);
    }
    std::vector<int> order(K);
    for (int i = 0; i < K; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return rate[a] > rate[b]; });
    auto assembled = [&](const int *ids, const char *name) {
        for (int i = 0; i < 7; i++) CK(hipMemMap(va + (size_t)i * CB, CB, 0, ch[ids[i]], 0));
        set_rw(va, 7 * CB);
        printf("  %-28s: obs+mask %6.1f  obs only %6.1f us   (chunk rates", name, time_observe(h, (float *)va, mask_d), time_observe(h, (float *)va, nullptr));
        for (int i = 0; i < 7; i++) printf(" %.0f", rate[ids[i]]);
        printf(")\n"); fflush(stdout);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 7; i++) CK(hipMemUnmap(va + (size_t)i * CB, CB));
    };
    printf("Q3 buffers assembled by probe rate\n");
    assembled(order.data(), "the 7 fastest chunks");
    assembled(order.data() + 7, "the next 7");
    assembled(order.data() + K - 7, "the 7 slowest chunks");
    assembled(order.data() + K / 2 - 3, "7 median chunks");
    {
        int mix[7] = {order[0], order[K - 1], order[1], order[K - 2], order[2], order[K - 3], order[3]};
        assembled(mix, "fast / slow alternating");
        int first[7] = {0, 1, 2, 3, 4, 5, 6};
        assembled(first, "chunks 0..6");
    }
    printf("done\n");
    return 0;
}
