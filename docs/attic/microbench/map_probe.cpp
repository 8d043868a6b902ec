// Round 3: does the speed class of an observation buffer follow the MEMORY or the way the kernel walks it?  Twelve hipMalloc'ed
// 65,536-game observation buffers (held) x block->game maps of the kernel (SGX_MAP: 0 = XCD ranges (product), 1 = linear,
// 2,s = stripes of s workgroups per XCD, 3,r = XCD ranges with every front started at another phase, 4 = ranges walked downwards,
// 5 = neighbouring XCDs walk towards each other).  One handle per map (the map is read at sgx_create).
//   hipcc -O2 -I include tools/microbench/map_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/map_probe
#include <hip/hip_runtime.h>
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
static sgx_env *make_env(int64_t N, const char *map) {
    setenv("SGX_MAP", map, 1);
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
    const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    const int NB = argc > 2 ? atoi(argv[2]) : 12;
    const char *maps[] = {"0", "7,40", "7,60", "7,80", "7,100", "7,120", "7,150", "7,200", "0"};
    const int NM = sizeof(maps) / sizeof(maps[0]);
    std::vector<sgx_env *> hs;
    for (auto m : maps) hs.push_back(make_env(N, m));
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
    printf("%lld games; rows: buffers, columns: maps\n%-8s", (long long)N, "buffer");
    for (auto m : maps) printf(" %8s", m);
    printf("\n");
    for (int i = 0; i < NB; i++) {
        printf("%-8d", i);
        for (int m = 0; m < NM; m++) { printf(" %8.1f", time_observe(hs[m], held[i], mask_d)); fflush(stdout); }
        printf("\n");
    }
    printf("obs only (no mask):\n");
    for (int i = 0; i < NB; i++) {
        printf("%-8d", i);
        for (int m = 0; m < NM; m++) { printf(" %8.1f", time_observe(hs[m], held[i], nullptr)); fflush(stdout); }
        printf("\n");
    }
    printf("done\n");
    return 0;
}
