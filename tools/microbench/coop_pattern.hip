// Probe: per-wave store streams (each wave writes its own game's 26,800 B obs) vs workgroup-cooperative bursts (the 8 waves
// of a block write the block's 8 consecutive games as one contiguous range, chunk c -> wave c % 8), over several allocations.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WPB = 8, QPE = 1675;   // quads (16 B) per game observation
template <int MODE>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, long n, int delay) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    float acc = (float)lane;
    for (int i = 0; i < delay + wave * (delay / 8); ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);   // skewed per-wave "compute"
    f32x4 v = {1.f, acc * 1e-30f, 0.5f, -1.f};
    if (MODE == 0) {            // per-wave stream
        const long env = g * WPB + wave;
        if (env >= n) return;
        f32x4 *o = reinterpret_cast<f32x4 *>(obs + env * 6700);
        for (int q = lane; q < QPE; q += 64) o[q] = v;
    } else {                    // cooperative: block's 8 games = 13,400 quads, written in address order across the 8 waves
        if (MODE == 2) __syncthreads();
        f32x4 *o = reinterpret_cast<f32x4 *>(obs + g * WPB * 6700);
        for (int q = wave * 64 + lane; q < QPE * WPB; q += 64 * WPB) o[q] = v;
    }
}
template <int MODE>
float run(float *obs, long n, int delay) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int i = 0; i < 3; ++i) pattern<MODE><<<grid, 64 * WPB>>>(obs, n, delay);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) pattern<MODE><<<grid, 64 * WPB>>>(obs, n, delay);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}
int main() {
    const long n = 65536;
    float *bufs[6];
    for (int a = 0; a < 6; ++a) hipMalloc(&bufs[a], n * 26800);
    for (int delay : {0, 400}) {
        printf("delay %d fma-iterations (+ per-wave skew)\n", delay);
        for (int a = 0; a < 6; ++a)
            printf("  alloc %d: per-wave %6.1f us   coop %6.1f us   coop+barrier %6.1f us   (%.2f / %.2f / %.2f TB/s)\n", a,
                   run<0>(bufs[a], n, delay), run<1>(bufs[a], n, delay), run<2>(bufs[a], n, delay),
                   n * 26800.0 / run<0>(bufs[a], n, delay) / 1e6, n * 26800.0 / run<1>(bufs[a], n, delay) / 1e6, n * 26800.0 / run<2>(bufs[a], n, delay) / 1e6);
    }
    return 0;
}
