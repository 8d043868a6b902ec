// Does the class of an observation buffer depend on WHERE in device memory it lands?  A padding allocation of P bytes is made
// first (held), then the observation buffer, which is timed (sgx_observe, 65,536 Barrage games); both are freed and P grows.
// The allocator is deterministic, so P scans the buffer's physical position.
//   hipcc -O2 -I include tools/microbench/pad_sweep.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/pad_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;
static int8_t *player_d;
static sgx_env *h;

static float time_observe(float *obs, uint8_t *mask, int reps = 8) {
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
    const size_t MB = 1u << 20;
    const size_t step = (argc > 1 ? atoll(argv[1]) : 128) * MB, top = (argc > 2 ? atoll(argv[2]) : 8192) * MB;
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
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %zu MiB of %zu MiB\n", fr / MB, tot / MB);
    for (int pass = 0; pass < 2; pass++) {
        printf("pass %d: pad MiB -> us\n", pass);
        for (size_t P = 0; P <= top; P += step) {
            void *pad = nullptr; float *obs = nullptr;
            if (P) CK(hipMalloc(&pad, P));
            CK(hipMalloc((void **)&obs, bytes));
            printf(" %zu:%.1f", P / MB, time_observe(obs, mask_d)); fflush(stdout);
            CK(hipFree(obs));
            if (pad) CK(hipFree(pad));
        }
        printf("\n");
    }
    // the same with the padding made of 64 MiB pieces (small blocks fill holes first)
    printf("pad in 64 MiB pieces: pieces -> us\n");
    std::vector<void *> pieces;
    for (int k = 0; k <= 64; k++) {
        float *obs = nullptr;
        CK(hipMalloc((void **)&obs, bytes));
        printf(" %d:%.1f", k, time_observe(obs, mask_d)); fflush(stdout);
        CK(hipFree(obs));
        void *p; CK(hipMalloc(&p, 64 * MB)); pieces.push_back(p);
    }
    printf("\ndone\n");
    return 0;
}
