// Round 3: do physical chunks fall into groups such that two chunks of DIFFERENT groups written together are faster than two of the
// same group?  The real kernel (65,536 games, obs only, non-temporal) on 7 x 256 MiB virtual ranges backed by ONE chunk (A A A A A A A)
// or by TWO alternating chunks (A B A B A B A).
//   hipcc -O2 -I include tools/microbench/pair_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/pair_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
static float time_observe(sgx_env *h, float *obs, int reps = 4) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, nullptr, nullptr, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, nullptr, nullptr, 0, nullptr);
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
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    return h;
}
int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 48;
    const size_t CB = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
    const int64_t N = 65536;
    const int PER = (int)(((size_t)N * 26800 + CB - 1) / CB);
    sgx_env *h = make_env(N);
    { float *w; CK(hipMalloc((void **)&w, 1u << 30)); for (int i = 0; i < 300; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr)); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    // a few plain buffers first (held) so that the chunks come from further inside device memory, and as this box's classes
    std::vector<float *> held;
    printf("plain hipMalloc buffers (held):");
    for (int i = 0; i < 8; i++) { float *p; CK(hipMalloc((void **)&p, (size_t)N * 26800)); held.push_back(p); printf(" %.0f", time_observe(h, p)); fflush(stdout); }
    printf("\n");
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(K);
    for (auto &x : ch) CK(hipMemCreate(&x, CB, &prop, 0));
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, (size_t)PER * CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    auto timed = [&](const std::vector<int> &ids) {
        for (int i = 0; i < PER; i++) CK(hipMemMap(va + (size_t)i * CB, CB, 0, ch[ids[i % ids.size()]], 0));
        set_rw(va, (size_t)PER * CB);
        const float t = time_observe(h, (float *)va);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, (size_t)PER * CB));
        return t;
    };
    printf("%d chunks of %zu MiB, %d per buffer\nT1 one chunk repeated:", K, CB >> 20, PER);
    std::vector<float> t1(K);
    for (int c = 0; c < K; c++) { t1[c] = timed({c}); printf(" %.0f", t1[c]); fflush(stdout); }
    printf("\nT2 windows of consecutive chunks:");
    for (int s = 0; s + PER <= K; s += PER) { std::vector<int> ids; for (int i = 0; i < PER; i++) ids.push_back(s + i); printf(" %.0f", timed(ids)); fflush(stdout); }
    for (int R : {0, K / 2, K - 1}) {
        printf("\nT3 chunk %d alternating with chunk k:", R);
        for (int k = 0; k < K; k++) { printf(" %.0f", timed({R, k})); fflush(stdout); }
    }
    printf("\nT4 three chunks cycling (0, K/3, 2K/3): %.0f; four (0, K/4, K/2, 3K/4): %.0f; all different, stride K/PER: ", timed({0, K / 3, 2 * K / 3}), timed({0, K / 4, K / 2, 3 * K / 4}));
    { std::vector<int> ids; for (int i = 0; i < PER; i++) ids.push_back(i * (K / PER)); printf("%.0f\n", timed(ids)); }
    printf("done\n");
    return 0;
}
