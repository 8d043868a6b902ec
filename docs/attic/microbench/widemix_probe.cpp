// Round 3: regions of device memory sustain different rates under eight concurrent write fronts and a buffer lying in one region
// runs at that region's rate.  Does a buffer whose pages are drawn EVENLY FROM A WIDE POOL (every region a little) reach the best
// rate without knowing the regions?  Pool of 32 MiB chunks (24 GiB), real kernel, 65,536 and 262,144 games.
//   hipcc -O2 -I include tools/microbench/widemix_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/widemix_probe
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
    const size_t MB = 1u << 20, CS = (size_t)(argc > 2 ? atoi(argv[2]) : 32) * MB;
    const int P = argc > 1 ? atoi(argv[1]) : 768;
    const int64_t N = 65536, NL = 262144;
    sgx_env *h = make_env(N), *hl = make_env(NL);
    uint8_t *mask_d;
    CK(hipMalloc((void **)&mask_d, (size_t)NL * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(P);
    for (auto &x : ch) CK(hipMemCreate(&x, CS, &prop, 0));
    const size_t bytes = (size_t)N * 26800, lbytes = (size_t)NL * 26800;
    const int need = (int)((bytes + CS - 1) / CS), lneed = (int)((lbytes + CS - 1) / CS);
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, (size_t)lneed * CS, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    auto run = [&](sgx_env *env, const std::vector<int> &ids, int n, float *both, float *only) {
        for (int i = 0; i < n; i++) CK(hipMemMap(va + (size_t)i * CS, CS, 0, ch[ids[i]], 0));
        set_rw(va, (size_t)n * CS);
        *only = time_observe(env, (float *)va, nullptr);
        *both = time_observe(env, (float *)va, mask_d);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, (size_t)n * CS));
    };
    printf("pool: %d chunks of %zu MiB (%.1f GiB); 65,536 games need %d, 262,144 games %d\n", P, CS / MB, P * (double)CS / (1u << 30), need, lneed);
    printf("S1 consecutive windows of the pool, 65,536 games: obs-only us (obs+mask)\n ");
    std::vector<float> wt;
    for (int s = 0; s + need <= P; s += need) {
        std::vector<int> ids; for (int i = 0; i < need; i++) ids.push_back(s + i);
        float a, b; run(h, ids, need, &a, &b); wt.push_back(b);
        printf(" %.0f(%.0f)", b, a); fflush(stdout);
    }
    printf("\n");
    int wf = 0, ws = 0;
    for (size_t w = 0; w < wt.size(); w++) { if (wt[w] < wt[wf]) wf = (int)w; if (wt[w] > wt[ws]) ws = (int)w; }
    auto report = [&](sgx_env *env, const std::vector<int> &ids, int n, const char *name) {
        float a, b; run(env, ids, n, &a, &b);
        printf("  %-46s: obs only %7.1f   obs+mask %7.1f us\n", name, b, a); fflush(stdout);
    };
    printf("S2 65,536 games, mixtures (fastest window %d: %.0f, slowest %d: %.0f)\n", wf, wt[wf], ws, wt[ws]);
    {
        std::vector<int> ids;
        for (int i = 0; i < need; i++) ids.push_back((int)((int64_t)i * P / need));
        report(h, ids, need, "evenly strided over the whole pool");
        for (int i = 0; i < need; i++) ids[i] = (int)((int64_t)i * P / need) + 3;
        report(h, ids, need, "evenly strided, offset 3");
        std::vector<int> all(P); unsigned s = 4711;
        for (int i = 0; i < P; i++) all[i] = i;
        for (int i = P - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(all[i], all[(s >> 8) % (i + 1)]); }
        report(h, all, need, "random subset, random order");
        std::sort(all.begin(), all.begin() + need);
        report(h, all, need, "random subset, creation order");
        ids.clear();
        for (int i = 0; i < need; i++) ids.push_back((i & 1) ? ws * need + i : wf * need + i);
        report(h, ids, need, "fastest / slowest window alternating chunks");
        ids.clear();
        for (int i = 0; i < need; i++) ids.push_back(i < need / 2 ? wf * need + i : ws * need + i);
        report(h, ids, need, "first half fastest, second half slowest");
        ids.clear();
        for (int i = 0; i < need; i++) ids.push_back(i % 4 == 3 ? ws * need + i : wf * need + i);
        report(h, ids, need, "3 : 1 fastest : slowest");
    }
    printf("S3 262,144 games\n");
    {
        for (int s = 0; s + lneed <= P; s += lneed) {
            std::vector<int> ids; for (int i = 0; i < lneed; i++) ids.push_back(s + i);
            char name[64]; snprintf(name, sizeof(name), "consecutive chunks %d..%d", s, s + lneed - 1);
            report(hl, ids, lneed, name);
        }
        std::vector<int> ids;
        for (int i = 0; i < lneed; i++) ids.push_back((int)((int64_t)i * P / lneed));
        report(hl, ids, lneed, "evenly strided over the whole pool");
        std::vector<int> all(P); unsigned s = 99;
        for (int i = 0; i < P; i++) all[i] = i;
        for (int i = P - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(all[i], all[(s >> 8) % (i + 1)]); }
        report(hl, all, lneed, "random subset, random order");
        float *p; CK(hipMalloc((void **)&p, lbytes));
        printf("  %-46s: obs only %7.1f   obs+mask %7.1f us\n", "plain hipMalloc", time_observe(hl, p, nullptr), time_observe(hl, p, mask_d));
        CK(hipFree(p));
    }
    printf("done\n");
    return 0;
}
