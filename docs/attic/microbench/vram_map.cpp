// Class of the observation buffer by physical position: 2 GiB physical chunks (one naturally aligned buddy block each) are
// created one after the other until device memory is (almost) full; each is timed as the observation buffer of 65,536 Barrage
// games (sgx_observe).  Output: launch time by creation order.
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

static float time_observe(float *obs, uint8_t *mask, int reps = 6) {
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
    const size_t MB = 1u << 20, cs = (argc > 1 ? atoll(argv[1]) : 2048) * MB;
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); return 1; }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    const size_t mbytes = (size_t)N * 3700;
    CK(hipMalloc((void **)&mask_d, mbytes)); CK(hipMalloc((void **)&player_d, N));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    const int n = (int)((fr - 6144 * MB) / cs);
    printf("free %zu MiB of %zu MiB: %d chunks of %zu MiB\n", fr / MB, tot / MB, n, cs / MB);
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, cs, 2u << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) {
            if (pass == 0) { hipMemGenericAllocationHandle_t x; if (hipMemCreate(&x, cs, &prop, 0) != hipSuccess) { printf(" [create failed at %d]", i); break; } hs.push_back(x); }
            if (i >= (int)hs.size()) break;
            CK(hipMemMap(vap, cs, 0, hs[i], 0));
            CK(hipMemSetAccess(vap, cs, &acc, 1));
            printf(" %.0f", time_observe((float *)vap, mask_d)); fflush(stdout);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(vap, cs));
        }
        printf("\n");
    }
    for (auto &x : hs) CK(hipMemRelease(x));
    printf("done\n");
    return 0;
}
