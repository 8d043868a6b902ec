// Round 3: what differs INSIDE the memory system between a fast and a slow observation buffer?  Twelve hipMalloc'ed 65,536-game
// buffers (held) are timed (classes printed), then each is written by exactly (3 + its index) observe launches in index order, so
// that the dispatches of a rocprofv3 --pmc trace can be attributed to the buffers by counting.  Run once plainly (times), and under
//   rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_64B -- class_pmc
//   hipcc -O2 -I include tools/microbench/class_pmc.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/class_pmc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static float time_observe(sgx_env *h, float *obs, uint8_t *mask, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask, nullptr, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1000.f;
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
    const int NB = argc > 1 ? atoi(argv[1]) : 12;
    const int64_t N = 65536;
    sgx_env *h = make_env(N);
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 300; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    const size_t bytes = (size_t)N * 26800;
    std::vector<float *> held;
    for (int i = 0; i < NB; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); held.push_back(p); }
    printf("obs-only launch us per buffer (buffer i is written by 3 + i launches):");
    for (int i = 0; i < NB; i++) { printf(" %d:%.1f", i, time_observe(h, held[i], nullptr, 3 + i)); fflush(stdout); }
    printf("\ndone\n");
    return 0;
}
