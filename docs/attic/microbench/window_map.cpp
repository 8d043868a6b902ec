// Round 3: speed of 7-chunk windows (1.79 GB) along the creation order of 256 MiB chunks, real kernel (obs only) and sgx_mem_probe;
// then how a fast window degrades when k of its chunks are replaced by chunks of a slow window (is the class additive over chunks?).
//   hipcc -O2 -I include tools/microbench/window_map.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/window_map
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static float time_observe(sgx_env *h, float *obs, uint8_t *mask, int reps = 4) {
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
    const int KMAX = argc > 1 ? atoi(argv[1]) : 1050;
    const int64_t N = 65536;
    const size_t MB = 1u << 20, CB = 256 * MB;
    sgx_env *h = make_env(N);
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    int K = (int)std::min<size_t>((size_t)KMAX, (free_b - (6ull << 30)) / CB);
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch;
    for (int i = 0; i < K; i++) {
        hipMemGenericAllocationHandle_t hh;
        if (hipMemCreate(&hh, CB, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        ch.push_back(hh);
    }
    K = (int)ch.size();
    printf("free %zu MiB of %zu MiB; %d chunks of 256 MiB held\n", free_b / MB, total_b / MB, K);
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, 7 * CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    auto run = [&](const std::vector<int> &ids, float *real_us, float *probe_gbps) {
        for (size_t i = 0; i < 7; i++) CK(hipMemMap(va + i * CB, CB, 0, ch[ids[i]], 0));
        set_rw(va, 7 * CB);
        if (probe_gbps) *probe_gbps = probe(va, 7 * CB);
        *real_us = time_observe(h, (float *)va, nullptr);
        CK(hipDeviceSynchronize());
        for (size_t i = 0; i < 7; i++) CK(hipMemUnmap(va + i * CB, CB));
    };
    const int NW = K / 7;
    std::vector<float> wt(NW), wp(NW);
    printf("W1 windows [7w .. 7w+6]: real obs-only us (rows of 32)");
    for (int w = 0; w < NW; w++) {
        std::vector<int> ids; for (int i = 0; i < 7; i++) ids.push_back(7 * w + i);
        run(ids, &wt[w], &wp[w]);
        if (w % 32 == 0) printf("\n %4d:", w);
        printf(" %3.0f", wt[w]); fflush(stdout);
    }
    printf("\nW1p the probe's GB/s / 100 for the same windows");
    for (int w = 0; w < NW; w++) { if (w % 32 == 0) printf("\n %4d:", w); printf(" %3.0f", wp[w] / 100); }
    printf("\n");
    int wf = 0, ws = 0;
    for (int w = 0; w < NW; w++) { if (wt[w] < wt[wf]) wf = w; if (wt[w] > wt[ws]) ws = w; }
    printf("fastest window %d (%.1f us), slowest window %d (%.1f us)\n", wf, wt[wf], ws, wt[ws]);
    printf("W2 fastest window with its first k chunks replaced by the slowest window's: k = 0..7:");
    for (int k = 0; k <= 7; k++) {
        std::vector<int> ids; for (int i = 0; i < 7; i++) ids.push_back(i < k ? 7 * ws + i : 7 * wf + i);
        float t; run(ids, &t, nullptr); printf(" %.1f", t); fflush(stdout);
    }
    printf("\nW3 the same, replacing every other position first (0, 2, 4, 6, 1, 3, 5):");
    {
        const int ord[7] = {0, 2, 4, 6, 1, 3, 5};
        std::vector<int> ids; for (int i = 0; i < 7; i++) ids.push_back(7 * wf + i);
        for (int k = 0; k <= 7; k++) {
            if (k) ids[ord[k - 1]] = 7 * ws + ord[k - 1];
            float t; run(ids, &t, nullptr); printf(" %.1f", t); fflush(stdout);
        }
    }
    printf("\nW4 single chunk of the slowest window in the fastest window, position by position:");
    for (int pos = 0; pos < 7; pos++) {
        std::vector<int> ids; for (int i = 0; i < 7; i++) ids.push_back(7 * wf + i);
        ids[pos] = 7 * ws + pos;
        float t; run(ids, &t, nullptr); printf(" %.1f", t); fflush(stdout);
    }
    printf("\nW5 each chunk of the slowest window alone in the fastest window's position 3:");
    for (int c = 0; c < 7; c++) {
        std::vector<int> ids; for (int i = 0; i < 7; i++) ids.push_back(7 * wf + i);
        ids[3] = 7 * ws + c;
        float t; run(ids, &t, nullptr); printf(" %.1f", t); fflush(stdout);
    }
    printf("\ndone\n");
    return 0;
}
