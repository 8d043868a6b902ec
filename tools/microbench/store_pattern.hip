// Ceiling probe: the step kernel's store pattern (one wave per env: 26,800 B obs + 3,700 B mask, 16 B per lane)
// and launch geometry (4 waves per workgroup, XCD-chunked env map) with NO compute and no loads.
// Build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, bool XCD>
__global__ __launch_bounds__(64 * WPB) void pattern(float *obs, unsigned char *mask, long n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long g = blockIdx.x;
    if (XCD) { long nb = gridDim.x, chunk = nb >> 3; g = (g & 7) * chunk + (g >> 3); }
    const long env = g * WPB + wave;
    if (env >= n) return;
    f32x4 v = {1.f, 0.f, 0.5f, -1.f};
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
}
template <int WPB, bool XCD>
void run(const char *name, float *obs, unsigned char *mask, long n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)(((n + WPB - 1) / WPB + 7) & ~7L);
    for (int i = 0; i < 5; ++i) pattern<WPB, XCD><<<grid, 64 * WPB>>>(obs, mask, n);
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) pattern<WPB, XCD><<<grid, 64 * WPB>>>(obs, mask, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)n * (26800 + 3700);
    printf("%-28s %8.1f us/launch  %.2f TB/s written\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main() {
    const long n = 65536;
    float *obs; unsigned char *mask;
    hipMalloc(&obs, n * 26800); hipMalloc(&mask, n * 3700 + 64);
    run<1, false>("WPB=1 linear", obs, mask, n);
    run<1, true>("WPB=1 xcd-chunked", obs, mask, n);
    run<4, false>("WPB=4 linear", obs, mask, n);
    run<4, true>("WPB=4 xcd-chunked", obs, mask, n);
    run<8, true>("WPB=8 xcd-chunked", obs, mask, n);
    run<16, true>("WPB=16 xcd-chunked", obs, mask, n);
    return 0;
}
