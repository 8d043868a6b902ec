// Probe: does placement sensitivity appear when a wave's 27 obs stores are spread over a long lifetime (compute between
// stores), as in the real kernel, vs issued back to back?  6 obs allocations x {burst, spread}.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, long n, int pre, int between, int pre_clump) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    float acc = (float)lane;
    for (int i = 0; i < pre; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    f32x4 v = {1.f, 0.f, 0.5f, -1.f};
    f32x4 *o = reinterpret_cast<f32x4 *>(obs + env * 6700) + lane;
    if (between >= 0) {
        for (int s = 0; s < 25; ++s) {
            for (int i = 0; i < between; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.y = acc * 1e-30f;
            o[s * 67] = v;
        }
    } else {   // clumps: -between fma per store, but computed for `clump` stores at once, then those stores back to back
        const int clump = pre_clump;
        for (int s0 = 0; s0 < 25; s0 += clump) {
            for (int i = 0; i < -between * clump; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.y = acc * 1e-30f;
            for (int s = s0; s < s0 + clump && s < 25; ++s) o[s * 67] = v;
        }
    }
    for (int t = lane; t < 75; t += 64) reinterpret_cast<f32x4 *>(obs + env * 6700)[(t / 3) * 67 + 64 + t % 3] = v;
    if (acc == 12345.f) smem[0] = 1;
}
int main() {
    const long n = 65536;
    constexpr int WPB = 8;
    float *obs[6];
    for (int a = 0; a < 6; ++a) hipMalloc(&obs[a], n * 26800);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 40 * 1024;   // 3-4 workgroups of 8 waves per CU, like the real kernel
    hipFuncSetAttribute((const void *)pattern<WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int pre = cfg == 1 ? 600 : 0, between = cfg == 2 ? 24 : cfg >= 3 ? -24 : 0, clump = cfg == 3 ? 9 : cfg == 4 ? 5 : 13;
        printf("%s\n", cfg == 0 ? "burst, no compute" : cfg == 1 ? "600 fma up front, then burst" : cfg == 2 ? "24 fma between consecutive stores (spread)" : cfg == 3 ? "clumps of 9 stores" : cfg == 4 ? "clumps of 5 stores" : "clumps of 13 stores");
        for (int a = 0; a < 6; ++a) {
            for (int i = 0; i < 3; ++i) pattern<WPB><<<grid, 64 * WPB, lds>>>(obs[a], n, pre, between, clump);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) pattern<WPB><<<grid, 64 * WPB, lds>>>(obs[a], n, pre, between, clump);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  alloc %d: %6.1f us\n", a, ms / 20 * 1e3);
        }
    }
    return 0;
}
