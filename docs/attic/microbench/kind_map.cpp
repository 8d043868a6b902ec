// Round 3: which 256 MiB chunks of device memory are complementary?  A single chunk sustains ~5.5 TB/s under the observation
// store pattern, some 1.76 GB buffers 6.9 TB/s: the rate belongs to the COMBINATION of physical regions.  K chunks are created one
// after the other (up to almost all of device memory, held but untouched except by the probes):
//   K1  probe rate of [chunk 0 + chunk k] for every k  (who complements chunk 0?)
//   K2  with m = the best partner: [chunk 0 + chunk m + chunk k] for every 4th k  (a third kind?)
//   K3  greedy 7-chunk buffer by union probes, timed with the real kernel; the same for the first 7 chunks
//   hipcc -O2 -I include tools/microbench/kind_map.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/kind_map
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
static float probe(void *p, size_t n, int launches = 2) {
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
    const int KMAX = argc > 1 ? atoi(argv[1]) : 1000;
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
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    int K = (int)std::min<size_t>((size_t)KMAX, (free_b - (8ull << 30)) / CB);
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
    CK(hipMemAddressReserve(&vap, 8 * CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    auto union_rate = [&](const std::vector<int> &ids) {
        for (size_t i = 0; i < ids.size(); i++) CK(hipMemMap(va + i * CB, CB, 0, ch[ids[i]], 0));
        set_rw(va, ids.size() * CB);
        const float g = probe(va, ids.size() * CB);
        for (size_t i = 0; i < ids.size(); i++) CK(hipMemUnmap(va + i * CB, CB));
        return g;
    };
    printf("single chunks 0, 1, K/2, K-1: %.0f %.0f %.0f %.0f\n", union_rate({0}), union_rate({1}), union_rate({K / 2}), union_rate({K - 1}));
    printf("K1 rate of [0 + k] / 10 GB/s, k = 1 .. K-1 (rows of 64):");
    std::vector<float> r1(K, 0.f);
    for (int k = 1; k < K; k++) {
        r1[k] = union_rate({0, k});
        if ((k - 1) % 64 == 0) printf("\n %4d:", k);
        printf(" %3.0f", r1[k] / 10);
        fflush(stdout);
    }
    int m = 1;
    for (int k = 1; k < K; k++) if (r1[k] > r1[m]) m = k;
    printf("\nbest partner of chunk 0: chunk %d (%.0f GB/s)\n", m, r1[m]);
    printf("K2 rate of [0 + %d + k] / 10, every 4th k:", m);
    std::vector<float> r2(K, 0.f);
    int m2 = -1;
    for (int k = 1, c = 0; k < K; k += 4, c++) {
        if (k == m) continue;
        r2[k] = union_rate({0, m, k});
        if (c % 64 == 0) printf("\n %4d:", k);
        printf(" %3.0f", r2[k] / 10);
        if (m2 < 0 || r2[k] > r2[m2]) m2 = k;
        fflush(stdout);
    }
    printf("\nbest third: chunk %d (%.0f GB/s)\n", m2, r2[m2]);
    auto real = [&](const std::vector<int> &ids, const char *name) {
        for (size_t i = 0; i < 7; i++) CK(hipMemMap(va + i * CB, CB, 0, ch[ids[i]], 0));
        set_rw(va, 7 * CB);
        const float pr = probe(va, 7 * CB, 3);
        printf("  %-34s: probe %5.0f GB/s   real obs+mask %6.1f   obs only %6.1f us  [", name, pr, time_observe(h, (float *)va, mask_d),
               time_observe(h, (float *)va, nullptr));
        for (int i = 0; i < 7; i++) printf("%d ", ids[i]);
        printf("]\n"); fflush(stdout);
        CK(hipDeviceSynchronize());
        for (size_t i = 0; i < 7; i++) CK(hipMemUnmap(va + i * CB, CB));
    };
    printf("K3 seven-chunk buffers\n");
    real({0, 1, 2, 3, 4, 5, 6}, "chunks 0..6");
    {   // greedy by union probes over a candidate subset (every 8th chunk + the partners found above)
        std::vector<int> cand;
        for (int k = 0; k < K; k += 8) cand.push_back(k);
        cand.push_back(m); cand.push_back(m2);
        std::vector<int> sel = {0};
        while (sel.size() < 7) {
            int best = -1; float bg = 0;
            for (int c : cand) {
                if (std::find(sel.begin(), sel.end(), c) != sel.end()) continue;
                std::vector<int> t = sel; t.push_back(c);
                const float g = union_rate(t);
                if (g > bg) { bg = g; best = c; }
            }
            sel.push_back(best);
            printf("  greedy +%d -> %.0f GB/s\n", best, bg); fflush(stdout);
        }
        real(sel, "greedy by union probes");
        std::vector<int> alt = {0, m, 1, m + 1 < K ? m + 1 : m - 1, 2, m + 2 < K ? m + 2 : m - 2, 3};
        real(alt, "0, m, 1, m+1, 2, m+2, 3");
        std::vector<int> spread;
        for (int i = 0; i < 7; i++) spread.push_back((int)((int64_t)i * (K - 1) / 6));
        real(spread, "evenly spread over all chunks");
    }
    printf("done\n");
    return 0;
}
