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


// sgx_store_probe: the observation stream of the step kernel as emit_codes (sgx_obs.h) issues it, without the game -- one wave per segment of
// `seg_bytes` (16-byte aligned, any length: a 10x10 observation is 26,800 B, so segments start anywhere inside a 128-byte line), swept in
// 16-byte-per-lane store instructions on 1 KiB ADDRESS boundaries; the lines a wave writes whole leave as non-temporal stores (`nt`), the
// first and the last line of a segment, shared with the neighbouring segments, go through L2.  512-thread workgroups at the step kernel's
// occupancy promise (6 waves per SIMD), mapped to segments like games to workgroups: eight contiguous XCD ranges PER PASS, and the passes
// one after the other in dispatch order, so that a line is written again only after the whole range has been written in between.
// Fewer resident waves (the host requests dynamic LDS that nothing uses) and `pace` vary how many store streams the memory sees at once;
// `persistent` makes the waves long-lived (a grid of the resident workgroups, each wave walking many segments): a wave that ends keeps its
// slot until its last stores are acknowledged, so short-lived waves leave the store path idle at every tail.
// PAYLOAD 0: zeros; 1: observation-like floats (0 / 1 / -1 / 0.5 decoded from a per-quad code pattern, like the kernel's own values);
// 2: incompressible bits (a hash of the quad's address and the launch's salt).
template <int PAYLOAD>
__global__ __launch_bounds__(512, 6) void store_probe_kernel(char *__restrict__ base, const int64_t groups_per_pass, const int passes, const int seg_bytes,
                                                             const int nt, const uint32_t salt, const int pace, const int persistent, const int dwell,
                                                             const int ring, const int64_t ring_bytes) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int64_t b = blockIdx.x, x = b & 7;                           // XCD
    const int64_t per_xcd = groups_per_pass >> 3;                      // workgroups' worth of segments of one XCD in one pass
    const int64_t total = per_xcd * passes;                            // ... in the launch
    // one workgroup per (pass, group) -- short-lived waves, like one launch per step -- or, persistent, a grid of the RESIDENT workgroups
    // only, every wave walking segment after segment for the whole launch like a wave of the multi-step kernel walks its game's steps
    const int64_t istep = persistent ? (int64_t)(gridDim.x >> 3) : total;
    for (int64_t i = b >> 3; i < total; i += istep) {
        const int64_t pass = i / per_xcd, j = i - pass * per_xcd;
        const int64_t seg_i = (x * per_xcd + j) * 8 + slot;
        // dwell > 1 (persistent waves only): the wave writes its segment `dwell` times before it moves on -- what a wave of the multi-step
        // kernel does with its game's observation, step after step -- cycling through `ring` sub-ranges of the buffer (the output sets of a ring)
        for (int rep = 0; rep < dwell; ++rep) {
        char *seg = base + (int64_t)(rep % ring) * ring_bytes + seg_i * (int64_t)seg_bytes;
        const int NQ = seg_bytes >> 4;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(seg) >> 4) & 63), l0 = (int)((reinterpret_cast<uintptr_t>(seg) >> 4) & 7);
        const int first_line = l0 ? 0 : -1, last_line = ((l0 + NQ) & 7) ? (NQ - 1 + l0) >> 3 : -1;
        f32x4 *q4 = reinterpret_cast<f32x4 *>(seg);
        int sweep = 0;
#pragma unroll SGX_OBS_UNROLL
        for (int q0 = -m0; q0 < NQ; q0 += 64, ++sweep) {
            const int q = q0 + lane;
            const bool in = (unsigned)q < (unsigned)NQ;
            // nt: 0 = plain stores, 1 = non-temporal, N >= 2 = every N-th 1 KiB sweep plain and the others non-temporal -- the step kernel's mix:
            // its mask (3,700 B per 26,800 B of observation) leaves as plain stores through L2
            const bool nt_here = nt == 1 || (nt >= 2 && (sweep % nt) != 0);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PAYLOAD == 1) {
                const unsigned hq = (unsigned)q * 0x9E3779B1u;
                const unsigned xc = ((hq >> 7) & 0x4444u & (hq >> 13)) | ((hq >> 20) & 0x000Cu) | ((hq >> 3) & 0x0200u & (hq >> 11));   // nibbles: mostly 0, some 4 (1.0), 0xC (-1.0), 2 (0.5)
                o = f32x4{code_to_float(xc), code_to_float(xc >> 4), code_to_float(xc >> 8), code_to_float(xc >> 12)};
            } else if constexpr (PAYLOAD == 2) {
                uint32_t hsh = (uint32_t)(reinterpret_cast<uintptr_t>(q4 + q) >> 4) * 2654435761u ^ salt;
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12; hsh *= 0x297A2D39u; hsh ^= hsh >> 15; w[k] = hsh; hsh += 0x9E3779B9u; }
                o = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
            }
            const bool edge = ((q + l0) >> 3) == first_line || ((q + l0) >> 3) == last_line;
            if (in && (edge || !nt_here)) q4[q] = o;
            if (in && !edge && nt_here) __builtin_nontemporal_store(o, &q4[q]);
            // pacing: the step kernel's waves do not store back to back -- game logic sits between a game's bursts, and at any moment only a part
            // of the resident waves is storing; `pace` sleeps of 64 cycles after every 1 KiB sweep stand in for that
            for (int p = 0; p < pace; ++p) __builtin_amdgcn_s_sleep(1);
        }
        }
    }
}

}  // namespace
