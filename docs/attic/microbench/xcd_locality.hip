// Round 3: is there XCD <-> memory locality?  For 256 MiB physical chunks (created one after the other) the write-stream rate
// when only the workgroups of ONE XCD write (non-temporal 1 KiB stores, one wave per 26 KiB segment).  Prints an [XCD][chunk]
// matrix in GB/s.  Workgroup -> XCD is taken from the hardware register XCC_ID and cross-checked against blockIdx % 8.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/xcd_locality.hip -o tools/microbench/xcd_locality
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 26 * 1024;

__device__ inline int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

// every workgroup whose XCD is `xcd` takes segments from a shared counter; the others leave at once
__global__ __launch_bounds__(512) void one_xcd_write(char *base, int n_seg, int xcd, int *counter, int *mismatch) {
    __shared__ int first;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int me = xcc_id();
    if (threadIdx.x == 0 && me != (int)(blockIdx.x & 7)) atomicAdd(mismatch, 1);
    if (me != xcd) return;
    const f32x4 v = {1.f, 0.5f, -1.f, (float)xcd};
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) first = atomicAdd(counter, 8);
        __syncthreads();
        const int seg = first + slot;
        if (first >= n_seg) return;
        if (seg < n_seg) {
            char *p = base + (size_t)seg * SEG;
#pragma unroll 2
            for (int k = 0; k < SEG / 1024; ++k) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p + k * 1024) + lane);
        }
    }
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
    const int K = argc > 1 ? atoi(argv[1]) : 64;
    const size_t CB = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(K);
    for (int i = 0; i < K; i++) CK(hipMemCreate(&ch[i], CB, &prop, 0));
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, CB, 2u << 20, nullptr, 0));
    char *va = (char *)vap;
    int *ctr; CK(hipMalloc((void **)&ctr, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_seg = (int)(CB / SEG);
    std::vector<std::vector<float>> rate(8, std::vector<float>(K, 0.f));
    int mism = 0;
    for (int c = 0; c < K; c++) {
        CK(hipMemMap(va, CB, 0, ch[c], 0));
        set_rw(va, CB);
        for (int x = 0; x < 8; x++) {
            const int reps = 3;
            float best = 0;
            for (int r = 0; r < reps; r++) {
                CK(hipMemsetAsync(ctr, 0, 64, nullptr));
                CK(hipEventRecord(e0, nullptr));
                one_xcd_write<<<2048, 512>>>(va, n_seg, x, ctr, ctr + 1);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) best = std::max(best, (float)((double)n_seg * SEG / (ms * 1e-3) / 1e9));
            }
            rate[x][c] = best;
        }
        int hm[2]; CK(hipMemcpy(hm, ctr, 8, hipMemcpyDeviceToHost)); mism += hm[1];
        CK(hipMemUnmap(va, CB));
    }
    printf("%d chunks of %zu MiB; XCC_ID != blockIdx %% 8 in %d workgroups\n", K, CB >> 20, mism);
    printf("rows: XCD 0..7; columns: chunks in creation order; GB/s\n");
    for (int x = 0; x < 8; x++) {
        printf("xcd %d:", x);
        for (int c = 0; c < K; c++) printf(" %4.0f", rate[x][c]);
        printf("\n");
    }
    printf("best XCD per chunk:");
    for (int c = 0; c < K; c++) { int b = 0; for (int x = 1; x < 8; x++) if (rate[x][c] > rate[b][c]) b = x; printf(" %d", b); }
    printf("\nmax/min over XCDs per chunk:");
    for (int c = 0; c < K; c++) { float lo = 1e9, hi = 0; for (int x = 0; x < 8; x++) { lo = std::min(lo, rate[x][c]); hi = std::max(hi, rate[x][c]); } printf(" %.2f", hi / lo); }
    printf("\ndone\n");
    return 0;
}
