// Probe: spread stores (compute between consecutive stores) with the kernel's chunking (64 quads from each 67-quad group:
// every 1 KiB store starts 48 B further off a 128-B line) vs chunks aligned to 1 KiB boundaries of the address space.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, int MODE>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, long n, int between) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    float acc = (float)lane;
    f32x4 v = {1.f, 0.f, 0.5f, -1.f};
    f32x4 *base = reinterpret_cast<f32x4 *>(obs + env * 6700);
    if (MODE == 0) {
        f32x4 *o = base + lane;
        for (int s = 0; s < 25; ++s) {
            for (int i = 0; i < between; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.y = acc * 1e-30f;
            o[s * 67] = v;
        }
        for (int i = 0; i < 2 * between; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
        for (int t = lane; t < 75; t += 64) base[(t / 3) * 67 + 64 + t % 3] = v;
    } else {
        const int m0 = (int)((reinterpret_cast<uintptr_t>(base) >> 4) & 63);     // quads past a 1 KiB boundary
        for (int c = 0; c < 27; ++c) {                                           // 1675 + 63 quads <= 27 chunks... +1
            for (int i = 0; i < between; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.y = acc * 1e-30f;
            const int q = 64 * c + lane - m0;
            if (q >= 0 && q < 1675) base[q] = v;
        }
        const int q = 64 * 27 + lane - m0;
        if (q >= 0 && q < 1675) base[q] = v;
    }
    if (acc == 12345.f) smem[0] = 1;
}
int main() {
    const long n = 65536;
    constexpr int WPB = 8;
    float *obs[6];
    for (int a = 0; a < 6; ++a) hipMalloc(&obs[a], n * 26800);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 40 * 1024;
    hipFuncSetAttribute((const void *)pattern<WPB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)pattern<WPB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int between : {0, 24}) {
        for (int mode = 0; mode < 2; ++mode) {
            printf("between %2d, %s:", between, mode ? "1 KiB-aligned chunks   " : "group chunks (kernel's)");
            for (int a = 0; a < 6; ++a) {
                auto launch = [&]() { if (mode) pattern<WPB, 1><<<grid, 64 * WPB, lds>>>(obs[a], n, between); else pattern<WPB, 0><<<grid, 64 * WPB, lds>>>(obs[a], n, between); };
                for (int i = 0; i < 3; ++i) launch();
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf(" %6.1f", ms / 20 * 1e3);
            }
            printf(" us\n");
        }
    }
    return 0;
}
