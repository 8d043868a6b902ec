// Ceiling probe 3: step-kernel memory mix (0.7 KB state read, 30.5 KB of stores per wave) with an artificial
// per-wave compute delay between the load and the stores, at a fixed occupancy (LDS-limited like the real kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, unsigned char *mask, int4 *state, int *sink, long n, int delay_iters, int split) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    const int4 *src = state + env * 40;            // 640 B record
    int4 s0 = lane < 40 ? src[lane] : make_int4(0, 0, 0, 0);
    float acc = (float)(s0.x & 1);
    // dependent VALU chain ~ 4 cycles per iteration per wave (plus contention)
    for (int i = 0; i < delay_iters; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    f32x4 v = {1.f + acc * 1e-30f, 0.f, 0.5f, -1.f};
    unsigned char *m = mask + env * 3700;
    const int A = (int)((env * 3700) & 15);
    int4 z = make_int4(0, 0, 0, 0);
    for (int c = lane; c < (A + 3700 + 15) / 16; c += 64) {
        if (16 * c >= A && 16 * c + 16 <= A + 3700) reinterpret_cast<int4 *>(m - A)[c] = z;
        else for (int w = 0; w < 4; ++w) { int o2 = 16 * c + 4 * w; if (o2 >= A && o2 < A + 3700) *reinterpret_cast<int *>(m - A + o2) = 0; }
    }
    f32x4 *o = reinterpret_cast<f32x4 *>(obs + env * 6700) + lane;
    for (int s = 0; s < 25; ++s) {
        if (split) for (int i = 0; i < split; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);   // compute between stores
        v.y = acc * 1e-30f;
        o[s * 67] = v;
    }
    for (int t = lane; t < 75; t += 64) reinterpret_cast<f32x4 *>(obs + env * 6700)[(t / 3) * 67 + 64 + t % 3] = v;
    if (lane == 0) reinterpret_cast<int *>(state + env * 40)[0] = s0.x + 1;
    if (acc == 12345.f) sink[0] = smem[0];
}
int main() {
    const long n = 65536;
    float *obs; unsigned char *mask; int4 *state; int *sink;
    hipMalloc(&obs, n * 26800); hipMalloc(&mask, n * 3700 + 64); hipMalloc(&state, n * 640); hipMalloc(&sink, 64);
    hipMemset(state, 1, n * 640);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    constexpr int WPB = 8;
    const int lds_cfgs[3] = {20 * 1024, 26 * 1024, 40 * 1024};     // 8 / 6 / 4 workgroups per CU -> 64(32 cap) / 48->32? see print
    for (int li = 0; li < 3; ++li) {
        hipFuncSetAttribute((const void *)pattern<WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cfgs[li]);
        int wgs = 160 * 1024 / lds_cfgs[li]; int waves = wgs * WPB > 32 ? 32 : wgs * WPB;
        for (int mode = 0; mode < 2; ++mode)
        for (int delay : {0, 1000, 2500, 5000, 10000}) {
            unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
            int pre = mode == 0 ? delay : 0, split = mode == 0 ? 0 : delay / 25;
            for (int i = 0; i < 3; ++i) pattern<WPB><<<grid, 64 * WPB, lds_cfgs[li]>>>(obs, mask, state, sink, n, pre, split);
            hipEventRecord(a);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) pattern<WPB><<<grid, 64 * WPB, lds_cfgs[li]>>>(obs, mask, state, sink, n, pre, split);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("waves/CU %2d  %s delay %5d fma  %7.1f us/launch  %.1f M steps/s\n", waves, mode ? "interleaved" : "up-front   ", delay, ms / reps * 1e3, n / (ms / reps * 1e-3) / 1e6);
        }
    }
    return 0;
}
