// Round 3: do the eight XCDs finish their eighth of a write stream at the same time?  The observation store pattern (one wave per
// 26 KiB segment, 1 KiB non-temporal stores), XCD x writing its own contiguous range of `share[x]` segments of one buffer; per-XCD
// first-start / last-end times (wall_clock64, 100 MHz) and the launch time; then the shares are re-balanced in proportion to the
// measured rates and the launch is timed again.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/xcd_balance.hip -o tools/microbench/xcd_balance
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 26 * 1024;
struct Shares { int first[8], count[8]; };
__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ __launch_bounds__(512) void fronts_kernel(char *base, Shares s, unsigned long long *t0, unsigned long long *t1) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int seg = i * 8 + slot;
    if (threadIdx.x == 0) atomicMin(&t0[x], wall_clock64());
    if (seg < s.count[x]) {
        char *p = base + (size_t)(s.first[x] + seg) * SEG;
        const f32x4 v = {1.f, 0.5f, -1.f, (float)x};
#pragma unroll 2
        for (int k = 0; k < SEG / 1024; ++k) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p + k * 1024) + lane);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&t1[x], wall_clock64());
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 6;
    const int total = 65536;                                        // segments = "games"
    { float *w; CK(hipMalloc((void **)&w, 1u << 30)); for (int i = 0; i < 300; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr)); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    unsigned long long *t0, *t1;
    CK(hipMalloc((void **)&t0, 64)); CK(hipMalloc((void **)&t1, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<char *> bufs;
    for (int b = 0; b < nbuf; b++) { char *p; CK(hipMalloc((void **)&p, (size_t)total * SEG)); bufs.push_back(p); }
    auto run = [&](char *buf, const int *count, float *xcd_us, float *launch_us) {
        Shares s; int at = 0, mx = 0;
        for (int x = 0; x < 8; x++) { s.first[x] = at; s.count[x] = count[x]; at += count[x]; mx = std::max(mx, count[x]); }
        const unsigned grid = (unsigned)((mx + 7) / 8) * 8;
        float best = 1e9f; float acc[8] = {0};
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(t0, 0xFF, 64)); CK(hipMemset(t1, 0, 64));
            CK(hipEventRecord(e0, nullptr));
            fronts_kernel<<<grid, 512>>>(buf, s, t0, t1);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h0[8], h1[8];
            CK(hipMemcpy(h0, t0, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, t1, 64, hipMemcpyDeviceToHost));
            if (rep == 0) continue;
            unsigned long long start = h0[0]; for (int x = 1; x < 8; x++) start = std::min(start, h0[x]);
            for (int x = 0; x < 8; x++) acc[x] += (float)(h1[x] - start) * 0.01f / 4;
            best = std::min(best, ms * 1000.f);
        }
        for (int x = 0; x < 8; x++) xcd_us[x] = acc[x];
        *launch_us = best;
    };
    for (int b = 0; b < nbuf; b++) {
        int count[8]; for (int x = 0; x < 8; x++) count[x] = total / 8;
        float xu[8], lu;
        run(bufs[b], count, xu, &lu);
        printf("buffer %d equal shares: launch %.1f us; per-XCD finish (us after the first start):", b, lu);
        for (int x = 0; x < 8; x++) printf(" %.0f", xu[x]);
        printf("\n");
        for (int it = 0; it < 3; it++) {
            // new shares in proportion to the measured rates (segments per us), renormalised to the same total, multiples of 8
            double rate[8], sum = 0; for (int x = 0; x < 8; x++) { rate[x] = count[x] / xu[x]; sum += rate[x]; }
            int at = 0;
            for (int x = 0; x < 8; x++) { count[x] = (int)(total * rate[x] / sum / 8 + 0.5) * 8; at += count[x]; }
            count[7] += total - at;
            run(bufs[b], count, xu, &lu);
            printf("   rebalanced %d: launch %.1f us; shares", it + 1, lu);
            for (int x = 0; x < 8; x++) printf(" %d", count[x]);
            printf("; finish");
            for (int x = 0; x < 8; x++) printf(" %.0f", xu[x]);
            printf("\n");
        }
        fflush(stdout);
    }
    printf("done\n");
    return 0;
}
