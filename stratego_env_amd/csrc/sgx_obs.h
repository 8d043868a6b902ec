// sgx_obs.h -- observation rendering: quad / channel tables and the line-aligned LUT emission
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Observation render: float32 [R][C][NCH], perspective of player index qi
// (partial: impl:1335-1397, full: impl:1230-1303; normalisation maenv:499-508 through the LUT)
// ---------------------------------------------------------------------------------------------
#ifndef SGX_OBS_UNROLL
#define SGX_OBS_UNROLL 4
#endif
// quad table entry of (perspective qi, quad qd, element j): LDS byte offset of the source board at the first 4-cell group
// (low 16 bits) and LUT index base (high 16 bits); built once per workgroup (build_quad_table)
template <class G, class Spec>
__device__ inline void build_quad_table(uint32_t *qtab, int tid, int nthreads) {
    constexpr int RC = G::RC, S = G::S, NCH = Spec::NCH;
    if constexpr (RC % 4 != 0) {
        // odd cell counts (5x5, 15x15): per-channel table instead -- entry (qi, ch) = LDS byte offset of the source board (low 16
        // bits) and LUT index base (high 16 bits); emit_obs adds the cell
        for (int i = tid; i < 2 * NCH; i += nthreads) {
            const int qi = i / NCH, ch = i - qi * NCH;
            qtab[i] = (uint32_t)(Spec::board(ch, qi) * S) | ((uint32_t)(lut_row(ch) + Spec::bias(ch)) << 16);
        }
        return;
    }
    for (int i = tid; i < 2 * NCH * 4; i += nthreads) {
        const int qi = i / (NCH * 4), r = i - qi * (NCH * 4), f = r;            // f = 4*qd + j : float index inside a 4-cell group
        const int rc = f / NCH, ch = f - rc * NCH;
        const uint32_t boff = (uint32_t)(Spec::board(ch, qi) * S + (qi ? RC - 1 - rc : rc));
        const uint32_t lrow = (uint32_t)(lut_row(ch) + Spec::bias(ch));
        qtab[i] = boff | (lrow << 16);
    }
}

// `tab` = this observation kind's LUT followed by its quad table.
// The observation is written in 1 KiB chunks aligned to 1 KiB ADDRESS boundaries (whole 128-byte lines per store
// instruction).  Chunking by 4-cell group instead (64 of a group's 67 quads per store, every store 48 bytes further off a
// line) left two partial lines per store and ran 1.5x slower in the store-pattern probe (tools/microbench/aligned_alloc.hip:
// 490 vs 333 us).  With address-aligned chunks a lane's quad changes every iteration, hence the quad table.
template <class G, class Spec>
__device__ void emit_obs(const Lds<G> &L, const float *tab, int qi, float *__restrict__ dst, int lane) {
    constexpr int RC = G::RC, NCH = Spec::NCH;
    const int8_t *bb = &L.b[0][0];
    const float *lut = tab;
    if constexpr (RC % 4 == 0) {
        constexpr int NQ = (RC / 4) * NCH;                                           // quads (16 B) of one observation
        const uint4 *qtab = reinterpret_cast<const uint4 *>(tab + LUT_DWORDS) + qi * NCH;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & (G::LPG - 1));  // quads past a 1 KiB boundary (LPG = 64)
        f32x4 *base = reinterpret_cast<f32x4 *>(dst);
        const int gstep = qi ? -4 : 4;
#pragma unroll SGX_OBS_UNROLL
        for (int q0 = -m0; q0 < NQ; q0 += G::LPG) {
            const int q = q0 + lane;
            const bool in = (unsigned)q < (unsigned)NQ;
            const int qq = in ? q : 0, g = qq / NCH, qd = qq - g * NCH;
            const uint4 e = qtab[qd];
            const int g4 = g * gstep;
            // board bytes are legal by construction (reset, move application, sanitised import): no clamp on the LUT index
            f32x4 o;
            o.x = lut[(e.x >> 16) + bb[(e.x & 0xFFFF) + g4]];
            o.y = lut[(e.y >> 16) + bb[(e.y & 0xFFFF) + g4]];
            o.z = lut[(e.z >> 16) + bb[(e.z & 0xFFFF) + g4]];
            o.w = lut[(e.w >> 16) + bb[(e.w & 0xFFFF) + g4]];
            if (in) stream_store(&base[q], o);
        }
    } else {
        // odd cell counts (5x5, 15x15): an env's observation is only 4-byte aligned.  Lanes own the 16-byte slots of the
        // ADDRESS range (sweep started on a chunk boundary like above); a slot's four floats are looked up one by one
        // through the per-channel table, whole slots leave as one 16-byte store, the partial first / last slot as dwords.
        // (One dword per lane per store, the first version, reached 2.3 TB/s on 15x15.)
        constexpr int NF = RC * NCH;
        const uint32_t *ctab = reinterpret_cast<const uint32_t *>(tab + LUT_DWORDS) + qi * NCH;
        const int a = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);             // floats past a 16-byte boundary
        float *base = dst - a;
        const int nslots = (a + NF + 3) >> 2;
        const int m0 = (int)((reinterpret_cast<uintptr_t>(base) >> 4) & (G::LPG - 1));
#pragma unroll 2
        for (int k0 = -m0; k0 < nslots; k0 += G::LPG) {
            const int k = k0 + lane;
            const bool slot_in = k >= 0 && k < nslots;
            const int f0 = 4 * (slot_in ? k : 0) - a;
            float o[4];
            bool in[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = f0 + j;
                in[j] = slot_in && (unsigned)f < (unsigned)NF;
                const int ff = in[j] ? f : 0, pcell = ff / NCH, ch = ff - pcell * NCH;
                const uint32_t e = ctab[ch];
                o[j] = lut[(e >> 16) + bb[(e & 0xFFFF) + (qi ? RC - 1 - pcell : pcell)]];
            }
            if (in[0] && in[3]) {
                f32x4 q = {o[0], o[1], o[2], o[3]};
                stream_store(&reinterpret_cast<f32x4 *>(base)[k], q);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (in[j]) base[4 * k + j] = o[j];
            }
        }
    }
}

}  // namespace
