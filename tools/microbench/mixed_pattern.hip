// Ceiling probe 2: the step kernel's memory mix with no compute: per env (one wave) read the 3,200 B state record
// (+16 B scalars), then write 26,800 B obs + 3,700 B mask + 100 B recent board.  MODE 0: stores only; 1: loads issued
// first, consumed after the stores (latency fully overlapped); 2: loads waited for before the first store (what a
// wave that needs its state before rendering must do).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, int MODE>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, unsigned char *mask, int4 *state, int *sink, long n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x; { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    int4 s0 = make_int4(0,0,0,0), s1 = s0, s2 = s0, s3 = s0;
    if (MODE >= 1) {
        const int4 *src = state + env * 200;
        s0 = src[lane]; s1 = src[lane + 64]; s2 = src[lane + 128]; if (lane < 8) s3 = src[lane + 192];
    }
    float add = 0.f;
    if (MODE == 2) add = (float)((s0.x ^ s1.y ^ s2.z ^ s3.w) & 1);   // forces the wait before the stores
    f32x4 v = {1.f + add, 0.f, 0.5f, -1.f};
    f32x4 *o = reinterpret_cast<f32x4 *>(obs + env * 6700) + lane;
    for (int s = 0; s < 25; ++s) o[s * 67] = v;
    for (int t = lane; t < 75; t += 64) reinterpret_cast<f32x4 *>(obs + env * 6700)[(t / 3) * 67 + 64 + t % 3] = v;
    unsigned char *m = mask + env * 3700;
    const int A = (int)((env * 3700) & 15);
    int4 z = make_int4(0, 0, 0, 0);
    for (int c = lane; c < (A + 3700 + 15) / 16; c += 64) {
        if (16 * c >= A && 16 * c + 16 <= A + 3700) reinterpret_cast<int4 *>(m - A)[c] = z;
        else for (int w = 0; w < 4; ++w) { int o2 = 16 * c + 4 * w; if (o2 >= A && o2 < A + 3700) *reinterpret_cast<int *>(m - A + o2) = 0; }
    }
    if (MODE >= 1) {
        if (lane < 25) reinterpret_cast<int *>(state + env * 200)[100 + lane] = s0.x + 1;   // "recent board" write-back
        if ((s0.x ^ s1.y ^ s2.z ^ s3.w) == 0x12345678) sink[0] = 1;
    }
}
template <int WPB, int MODE>
void run(const char *name, float *obs, unsigned char *mask, int4 *state, int *sink, long n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int i = 0; i < 5; ++i) pattern<WPB, MODE><<<grid, 64 * WPB>>>(obs, mask, state, sink, n);
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) pattern<WPB, MODE><<<grid, 64 * WPB>>>(obs, mask, state, sink, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.1f us/launch  -> %.1f M env-steps/s\n", name, ms / reps * 1e3, n / (ms / reps * 1e-3) / 1e6);
}
int main() {
    const long n = 65536;
    float *obs; unsigned char *mask; int4 *state; int *sink;
    hipMalloc(&obs, n * 26800); hipMalloc(&mask, n * 3700 + 64); hipMalloc(&state, n * 3200); hipMalloc(&sink, 64);
    hipMemset(state, 1, n * 3200);
    for (int r = 0; r < 2; ++r) {
        run<8, 0>("WPB=8 stores only", obs, mask, state, sink, n);
        run<8, 1>("WPB=8 + state read, not waited", obs, mask, state, sink, n);
        run<8, 2>("WPB=8 + state read, waited before stores", obs, mask, state, sink, n);
        run<4, 2>("WPB=4 + state read, waited before stores", obs, mask, state, sink, n);
    }
    return 0;
}
