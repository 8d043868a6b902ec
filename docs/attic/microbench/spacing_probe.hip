// Round 3: does the write rate depend on the SPACING of the eight XCD fronts in (physical) memory?  One 16 GiB virtual range over
// 2 GiB physical chunks (each physically contiguous and 2 GiB aligned); XCD x writes the L bytes at x * S (one wave per 26 KiB
// segment, 26 non-temporal 1 KiB stores, like the observation stream).  Rate vs S; the same with the whole pattern shifted.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/spacing_probe.hip -o tools/microbench/spacing_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 26 * 1024;

__global__ __launch_bounds__(512) void fronts_write(char *base, long long spacing, int seg_per_front) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long long x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const long long seg = i * 8 + slot;
    if (seg >= seg_per_front) return;
    char *p = base + x * spacing + seg * SEG;
    const f32x4 v = {1.f, 0.5f, -1.f, (float)x};
#pragma unroll 2
    for (int k = 0; k < SEG / 1024; ++k) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p + k * 1024) + lane);
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

int main(int argc, char **argv) {
    const size_t MB = 1u << 20, GB = 1u << 30;
    const size_t CS = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) * MB;
    const size_t total = 16 * GB;
    const int nch = (int)(total / CS);
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(nch);
    for (auto &x : ch) CK(hipMemCreate(&x, CS, &prop, 0));
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, total, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    // chunks were probably handed out at descending addresses: map them in reverse so that the range is (mostly) one ascending run
    for (int i = 0; i < nch; i++) CK(hipMemMap(va + (size_t)i * CS, CS, 0, ch[nch - 1 - i], 0));
    set_rw(va, total);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int spf = 8192;                                        // segments per front: 8192 x 26 KiB = 208 MiB
    const size_t L = (size_t)spf * SEG;
    auto rate = [&](size_t shift, size_t S) {
        const unsigned grid = (unsigned)(spf / 8) * 8;
        fronts_write<<<grid, 512>>>(va + shift, (long long)S, spf);
        CK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < 4; r++) fronts_write<<<grid, 512>>>(va + shift, (long long)S, spf);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return 8.0 * L * 4 / (ms * 1e-3) / 1e9;
    };
    printf("chunks of %zu MiB; front length %zu MiB; GB/s by spacing S (columns: pattern shifted by 0, 64 MiB, 1 GiB, 5 GiB + 2 MiB)\n", CS / MB, L / MB);
    const size_t spacings[] = {L, 219545600, 224 * MB, 240 * MB, 256 * MB, 257 * MB, 272 * MB, 288 * MB, 320 * MB, 384 * MB, 448 * MB, 512 * MB, 513 * MB,
                               640 * MB, 768 * MB, 1024 * MB, 1025 * MB, 1088 * MB, 1280 * MB, 1536 * MB};
    for (size_t S : spacings) {
        printf("  S = %8.2f MiB:", (double)S / MB);
        for (size_t shift : {(size_t)0, 64 * MB, GB, 5 * GB + 2 * MB}) {
            if (shift + 7 * S + L > total) { printf("     n/a"); continue; }
            printf(" %7.0f", rate(shift, S)); fflush(stdout);
        }
        printf("\n");
    }
    printf("done\n");
    return 0;
}
