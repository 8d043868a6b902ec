// Probe: is the placement sensitivity reproduced by (state read + obs/mask stores) without any compute?
// 6 obs allocations x {stores only, + 640 B state read waited before the stores, + read not waited}.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, int MODE>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, unsigned char *mask, const int4 *state, int *sink, long n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    int4 s0 = make_int4(0, 0, 0, 0);
    if (MODE >= 1 && lane < 40) s0 = state[env * 40 + lane];
    float add = 0.f;
    if (MODE == 1) add = (float)(s0.x & 1);
    f32x4 v = {1.f + add, 0.f, 0.5f, -1.f};
    unsigned char *m = mask + env * 3700;
    const int A = (int)((env * 3700) & 15);
    int4 z = make_int4(0, 0, 0, 0);
    for (int c = lane; c < (A + 3700 + 15) / 16; c += 64) {
        if (16 * c >= A && 16 * c + 16 <= A + 3700) reinterpret_cast<int4 *>(m - A)[c] = z;
        else for (int w = 0; w < 4; ++w) { int o2 = 16 * c + 4 * w; if (o2 >= A && o2 < A + 3700) *reinterpret_cast<int *>(m - A + o2) = 0; }
    }
    f32x4 *o = reinterpret_cast<f32x4 *>(obs + env * 6700) + lane;
    for (int s = 0; s < 25; ++s) o[s * 67] = v;
    for (int t = lane; t < 75; t += 64) reinterpret_cast<f32x4 *>(obs + env * 6700)[(t / 3) * 67 + 64 + t % 3] = v;
    if (MODE >= 1 && s0.x == 0x12345678) sink[0] = 1;
}
template <int MODE>
float run(float *obs, unsigned char *mask, int4 *state, int *sink, long n) {
    constexpr int WPB = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int i = 0; i < 3; ++i) pattern<WPB, MODE><<<grid, 64 * WPB>>>(obs, mask, state, sink, n);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) pattern<WPB, MODE><<<grid, 64 * WPB>>>(obs, mask, state, sink, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}
int main() {
    const long n = 65536;
    float *obs[6]; unsigned char *mask; int4 *state; int *sink;
    hipMalloc(&state, n * 640); hipMemset(state, 1, n * 640); hipMalloc(&mask, n * 3700 + 64); hipMalloc(&sink, 64);
    for (int a = 0; a < 6; ++a) hipMalloc(&obs[a], n * 26800);
    for (int a = 0; a < 6; ++a)
        printf("obs alloc %d: stores only %6.1f us | + state read (waited) %6.1f us | + state read (not waited) %6.1f us\n", a,
               run<0>(obs[a], mask, state, sink, n), run<1>(obs[a], mask, state, sink, n), run<2>(obs[a], mask, state, sink, n));
    return 0;
}
