// Does a scrambled virtual->physical mapping change the speed class of the obs buffer?
// obs is mapped with the HIP virtual-memory API from separately created physical chunks, in allocation order or permuted,
// and sgx_observe (65,536 Barrage games) is timed writing into it; hipMalloc / hipExtMallocWithFlags(contiguous) for reference.
//   hipcc -O2 -I include tools/microbench/vmm_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x -Wl,-rpath,$PWD/stratego_env_amd/_build -o tools/microbench/vmm_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

static sgx_env *h;
static uint8_t *mask_d;
static int8_t *player_d;

static float time_observe(float *obs) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask_d, player_d, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < 6; i++) sgx_observe(h, obs, nullptr, mask_d, player_d, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 6 * 1000.f;
}

static float *vmm_alloc(size_t bytes, size_t chunk, int mode /*0 sequential, 1 permuted, 2 permuted from a 2x pool*/, unsigned seed) {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, pool = mode == 2 ? 2 * n : n;
    std::vector<hipMemGenericAllocationHandle_t> hs(pool);
    for (size_t i = 0; i < pool; i++) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
    std::vector<size_t> order(pool);
    for (size_t i = 0; i < pool; i++) order[i] = i;
    if (mode) { std::mt19937 g(seed); std::shuffle(order.begin(), order.end(), g); }
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, 2u << 20, nullptr, 0));
    for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[order[i]], 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    return (float *)va;
}

int main() {
    const int64_t N = 65536;
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); return 1; }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);          // random back-row placement (no table)
    const size_t bytes = (size_t)N * 100 * 67 * 4;
    CK(hipMalloc((void **)&mask_d, (size_t)N * 3700)); CK(hipMalloc((void **)&player_d, N));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 400; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    printf("hipMalloc          :");
    for (int i = 0; i < 6; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); printf(" %6.1f", time_observe(p)); }
    printf("\n");
    { float *p; CK(hipExtMallocWithFlags((void **)&p, bytes, hipDeviceMallocContiguous)); printf("contiguous flag    : %6.1f\n", time_observe(p)); }
    const size_t MB = 1u << 20;
    {
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t g0 = 0, g1 = 0;
        CK(hipMemGetAllocationGranularity(&g0, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&g1, &prop, hipMemAllocationGranularityRecommended));
        printf("allocation granularity: minimum %zu, recommended %zu bytes\n", g0, g1);
    }
    for (size_t chunk : {MB / 16, MB / 4, MB, 2 * MB}) {
        printf("vmm chunk %4zu MiB : seq", chunk / MB);
        for (int i = 0; i < 2; i++) printf(" %6.1f", time_observe(vmm_alloc(bytes, chunk, 0, 0)));
        printf(" | permuted");
        for (int i = 0; i < 3; i++) printf(" %6.1f", time_observe(vmm_alloc(bytes, chunk, 1, 100 + i)));
        printf(" | permuted from 2x pool");
        for (int i = 0; i < 2; i++) printf(" %6.1f", time_observe(vmm_alloc(bytes, chunk, 2, 200 + i)));
        printf("\n"); fflush(stdout);
    }
    return 0;
}
