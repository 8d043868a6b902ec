// Round 3: XCD <-> memory affinity UNDER LOAD.  All eight XCDs write at the same time, XCD x the 256 MiB chunk assigned to it
// (non-temporal 1 KiB stores, one wave per 26 KiB segment taken from a per-XCD counter); per XCD the time from its first workgroup's
// start to its last workgroup's end (wall_clock64, 100 MHz).  Rotating the assignment gives T[xcd][chunk] for every pair.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/affinity_probe.hip -o tools/microbench/affinity_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 26 * 1024;
struct Args { char *base[8]; };
__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ __launch_bounds__(512) void affinity_kernel(Args a, int n_seg, int *counter, unsigned long long *t0, unsigned long long *t1) {
    __shared__ int first;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int x = xcc_id();
    if (threadIdx.x == 0) atomicMin(&t0[x], wall_clock64());
    const f32x4 v = {1.f, 0.5f, -1.f, (float)x};
    char *base = a.base[x];
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) first = atomicAdd(&counter[x * 32], 8);
        __syncthreads();
        if (first >= n_seg) break;
        const int seg = first + slot;
        if (seg < n_seg) {
            char *p = base + (size_t)seg * SEG;
#pragma unroll 2
            for (int k = 0; k < SEG / 1024; ++k) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p + k * 1024) + lane);
        }
    }
    __threadfence();
    if (threadIdx.x == 0) atomicMax(&t1[x], wall_clock64());
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
    const int K = argc > 1 ? atoi(argv[1]) : 16;
    const size_t CB = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
    { float *w; CK(hipMalloc((void **)&w, 1u << 30)); for (int i = 0; i < 300; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr)); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    hipMemAllocationProp prop = dev_prop();
    std::vector<hipMemGenericAllocationHandle_t> ch(K);
    std::vector<char *> va(K);
    for (int i = 0; i < K; i++) {
        CK(hipMemCreate(&ch[i], CB, &prop, 0));
        void *p = nullptr; CK(hipMemAddressReserve(&p, CB, 2u << 20, nullptr, 0));
        CK(hipMemMap(p, CB, 0, ch[i], 0)); set_rw(p, CB); va[i] = (char *)p;
    }
    int *counter; unsigned long long *t0, *t1;
    CK(hipMalloc((void **)&counter, 8 * 32 * 4)); CK(hipMalloc((void **)&t0, 64)); CK(hipMalloc((void **)&t1, 64));
    const int n_seg = (int)(CB / SEG);
    std::vector<std::vector<float>> T(8, std::vector<float>(K, 0.f));
    std::vector<std::vector<int>> cnt(8, std::vector<int>(K, 0));
    for (int pass = 0; pass < 3; pass++)
        for (int r = 0; r < K; r++) {
            Args a;
            for (int x = 0; x < 8; x++) a.base[x] = va[(x * (K / 8) + r) % K];      // the eight XCDs on eight different chunks
            CK(hipMemset(counter, 0, 8 * 32 * 4)); CK(hipMemset(t0, 0xFF, 64)); CK(hipMemset(t1, 0, 64));
            affinity_kernel<<<2048, 512>>>(a, n_seg, counter, t0, t1);
            CK(hipDeviceSynchronize());
            unsigned long long h0[8], h1[8];
            CK(hipMemcpy(h0, t0, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, t1, 64, hipMemcpyDeviceToHost));
            if (pass == 0) continue;
            for (int x = 0; x < 8; x++) { const int c = (x * (K / 8) + r) % K; T[x][c] += (float)(h1[x] - h0[x]) * 0.01f; cnt[x][c]++; }
        }
    printf("us for XCD x (rows) to write chunk c (columns, creation order) while the other seven XCDs write other chunks; %zu MiB chunks\n", CB >> 20);
    for (int x = 0; x < 8; x++) {
        printf("xcd %d:", x);
        for (int c = 0; c < K; c++) printf(" %5.0f", cnt[x][c] ? T[x][c] / cnt[x][c] : 0.f);
        printf("\n");
    }
    printf("row means:");
    for (int x = 0; x < 8; x++) { float s = 0; for (int c = 0; c < K; c++) s += T[x][c] / std::max(1, cnt[x][c]); printf(" %5.0f", s / K); }
    printf("\ncolumn means:");
    for (int c = 0; c < K; c++) { float s = 0; for (int x = 0; x < 8; x++) s += T[x][c] / std::max(1, cnt[x][c]); printf(" %5.0f", s / 8); }
    printf("\ndone\n");
    return 0;
}
