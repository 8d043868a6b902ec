// The observation buffer placed at growing offsets inside ONE large device allocation (a few big physically contiguous buddy
// blocks): launch time of sgx_observe (65,536 Barrage games) by offset.  A period in the offset is a period in the physical
// address.
//   hipcc -O2 -I include tools/microbench/slab_scan.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/slab_scan
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
    const int64_t N = argc > 3 ? atoll(argv[3]) : 65536;
    const size_t MB = 1u << 20;
    const size_t slab_mb = argc > 1 ? atoll(argv[1]) : 24576, step = (argc > 2 ? atoll(argv[2]) : 128) * MB;
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
    for (int s = 0; s < 2; s++) {
        char *slab = nullptr;
        CK(hipMalloc((void **)&slab, slab_mb * MB));
        printf("slab %d: %zu MiB at %p; offset MiB -> us (two passes)\n", s, slab_mb, (void *)slab);
        for (int pass = 0; pass < 2; pass++) {
            for (size_t o = 0; o + bytes <= slab_mb * MB; o += step) { printf(" %zu:%.1f", o / MB, time_observe((float *)(slab + o), mask_d)); fflush(stdout); }
            printf("\n");
        }
        if (s == 0) {   // finer scan of the first 7 GiB
            printf("fine (32 MiB steps):");
            for (size_t o = 0; o + bytes <= slab_mb * MB && o <= 7168 * MB; o += 32 * MB) { printf(" %zu:%.1f", o / MB, time_observe((float *)(slab + o), mask_d)); fflush(stdout); }
            printf("\n");
            // the mask buffer scanned the same way, observation buffer at offset 0 / the mask after it
            printf("mask offsets (obs at 0):");
            for (size_t o = 2048 * MB; o + mbytes <= slab_mb * MB && o <= 9216 * MB; o += 256 * MB) { printf(" %zu:%.1f", o / MB, time_observe((float *)slab, (uint8_t *)(slab + o))); fflush(stdout); }
            printf("\n");
        }
        // keep the slab (the next one is different memory)
    }
    printf("done\n");
    return 0;
}
