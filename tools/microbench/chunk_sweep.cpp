// Observation buffer built with the HIP virtual-memory API from equal physical chunks: launch time of sgx_observe (65,536
// Barrage games) by chunk size, against hipMalloc'ed buffers on the same box.
//   hipcc -O2 -I include tools/microbench/chunk_sweep.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/chunk_sweep
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

struct VBuf { char *va; size_t n, cs; std::vector<hipMemGenericAllocationHandle_t> hs; };

static VBuf vmm_alloc(size_t bytes, size_t cs) {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    VBuf b; b.cs = cs; b.n = (bytes + cs - 1) / cs; b.hs.resize(b.n);
    for (size_t i = 0; i < b.n; i++) CK(hipMemCreate(&b.hs[i], cs, &prop, 0));
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, b.n * cs, 2u << 20, nullptr, 0));
    b.va = (char *)va;
    for (size_t i = 0; i < b.n; i++) CK(hipMemMap(b.va + i * cs, cs, 0, b.hs[i], 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, b.n * cs, &acc, 1));
    return b;
}
static void vmm_free(VBuf &b) {
    CK(hipMemUnmap(b.va, b.n * b.cs));
    for (auto &x : b.hs) CK(hipMemRelease(x));
    CK(hipMemAddressFree(b.va, b.n * b.cs));
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); return 1; }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    const size_t bytes = (size_t)N * 100 * 67 * 4, mbytes = (size_t)N * 3700, MB = 1u << 20;
    CK(hipMalloc((void **)&mask_d, mbytes)); CK(hipMalloc((void **)&player_d, N));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    printf("hipMalloc obs (held):");
    std::vector<float *> held;
    for (int i = 0; i < 10; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); held.push_back(p); printf(" %6.1f", time_observe(p, mask_d)); fflush(stdout); }
    printf("\n");
    for (int pass = 0; pass < 2; pass++)
        for (size_t cs : {2 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 1024 * MB, 2048 * MB}) {
            printf("pass %d  obs in %4zu MiB chunks:", pass, cs / MB);
            for (int rep = 0; rep < 3; rep++) {
                VBuf b = vmm_alloc(bytes, cs);
                printf(" %6.1f", time_observe((float *)b.va, mask_d)); fflush(stdout);
                if (rep < 2) { VBuf keep = vmm_alloc(bytes / 3, cs); (void)keep; }   // leaked on purpose: the next buffer is different memory
                vmm_free(b);
            }
            printf("\n");
        }
    // the mask buffer the same way, observation buffer = 32 MiB chunks
    VBuf ob = vmm_alloc(bytes, 32 * MB);
    printf("mask hipMalloc: %6.1f | mask in chunks:", time_observe((float *)ob.va, mask_d));
    for (size_t cs : {2 * MB, 8 * MB, 32 * MB, 128 * MB}) {
        VBuf b = vmm_alloc(mbytes, cs);
        printf("  %zu MiB %6.1f", cs / MB, time_observe((float *)ob.va, (uint8_t *)b.va)); fflush(stdout);
        vmm_free(b);
    }
    printf("\ndone\n");
    return 0;
}
