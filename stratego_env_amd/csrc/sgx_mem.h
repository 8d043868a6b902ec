// sgx_mem.h -- device-memory probe: write-stream rate of a memory range under the step kernel's store pattern
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip (one translation unit).
#pragma once

namespace {

// The step kernel's observation stream, without the game: one wave per 26 KiB segment (a 10x10 observation is 26,800 B), written as
// 26 non-temporal 1 KiB store instructions, waves mapped to segments like games to workgroups (eight XCD ranges).  `n_seg` segments
// cover the range once; the grid holds `passes` times as many waves (a wave's segment = its index modulo n_seg), so that a small
// range still gives a launch long enough to time.  Non-temporal stores bypass the Infinity Cache: the time is the memory's.
constexpr int PROBE_SEG = 26 * 1024;
__global__ __launch_bounds__(512) void mem_probe_kernel(char *__restrict__ base, int64_t n_seg, int64_t n_waves) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int64_t nb = gridDim.x, b = blockIdx.x;
    const int64_t w = ((b & 7) * (nb >> 3) + (b >> 3)) * 8 + slot;
    if (w >= n_waves) return;
    char *seg = base + (w % n_seg) * (int64_t)PROBE_SEG;
    const float x = (float)(w & 1023);
    const f32x4 v = {x, 0.5f, -1.0f, 1.0f};
#pragma unroll 2
    for (int k = 0; k < PROBE_SEG / 1024; ++k) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(seg + k * 1024) + lane);
}

}  // namespace
