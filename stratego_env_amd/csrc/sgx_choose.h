// sgx_choose.h -- sgx_choose_actions: masked softmax over caller logits + one inverse-CDF sample per game
// Part of libstratego_mi355x.so; included by stratego_mi355x.hip in this order (one translation unit).
//
// The device-side counterpart of the reference's chooser (examples/basic_game_loop.py:6-31 with examples/util.py:4-47): logits in
// the shape of the valid-actions mask, invalid actions to -inf, softmax (temperature as a divisor), one sample per game.  One game
// per wave (toy boards: Geo::LPG lanes per game), HBM-bound: the logits (4 * R * C * K bytes per game) are read once, the mask
// once.
//
// Sampling is FIXED POINT, so that a draw depends on nothing but (logits, mask, seed, game, turn) -- not on the order of a
// floating-point reduction: weight(a) = floor(exp2((l_a - max) * log2(e) / T + 23)) for valid a (exactly 2^23 where l_a == max: the
// resolution of a float32 mantissa), total = the sum of the weights (exact integers), target = floor(r32 * total / 2^32) with r32 =
// the high half of the counter RNG's draw for (seed, global env id, game, STREAM_ACTION, turn), action = the first a in ascending flat
// index order whose running sum exceeds target.  With equal logits this IS sgx_sample_valid / the fused sampler's action:
// the floor(r32 * n_valid / 2^32)-th valid one.
#pragma once

namespace {

constexpr int CHOOSE_BITS = 23;
constexpr uint32_t CHOOSE_ONE = 1u << CHOOSE_BITS;

template <class G, int VEC>
struct ChooseGeo {
    static constexpr int NA = G::NA;
    static constexpr int NCH = (NA + VEC - 1) / VEC;                  // chunks of VEC actions
    static constexpr int ROWS = (NCH + G::LPG - 1) / G::LPG;          // one row = one chunk per lane of the game: <= 256 actions
    static constexpr bool CACHE = ROWS * VEC <= 72;                   // the game's logits stay in registers between the passes
    static constexpr int WAVES = 4, GAMES = WAVES * G::GPW;           // per workgroup (choose_kernel: __launch_bounds__(64 * WAVES))
};

// weight of one action; l = its logit, or -inf where the action is invalid (or the logit NaN); scale = log2(e) / T (+inf: argmax)
__device__ __forceinline__ uint32_t choose_weight(float l, float mx, float scale) {
    const uint32_t w = (uint32_t)exp2f(fmaf(l - mx, scale, (float)CHOOSE_BITS));   // l < mx: exp2(< 23) < 2^23; l = -inf: 0
    return l == mx ? CHOOSE_ONE : w;
}

// sum over the lanes of the game of a value < 2^25 (a row of <= 256 weights: < 2^31), the same in every lane of the game
template <class G>
__device__ __forceinline__ uint32_t gsum(uint32_t c) {
    if constexpr (G::LPG == 64) {
        int x = (int)c;                                  // 16-lane DPP rows first, then the four row totals on the scalar unit
        x += dpp_or_zero<0x111>(x);
        x += dpp_or_zero<0x112>(x);
        x += dpp_or_zero<0x114>(x);
        x += dpp_or_zero<0x118>(x);
        return (uint32_t)__builtin_amdgcn_readlane(x, 15) + (uint32_t)__builtin_amdgcn_readlane(x, 31) +
               (uint32_t)__builtin_amdgcn_readlane(x, 47) + (uint32_t)__builtin_amdgcn_readlane(x, 63);
    } else {
        return (uint32_t)glane<G>(gscan_incl<G>((int)c), G::LPG - 1);
    }
}

// max over the lanes of the game
template <class G>
__device__ __forceinline__ float gmax(float x) {
#pragma unroll
    for (int o = G::LPG / 2; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, G::LPG));
    return x;
}

// VEC = 4: boards whose action count is a multiple of 4 (every variant of the reference except 5x5 and 15x15) with 16-byte aligned
//          logits and 4-byte aligned mask rows -- one float4 + one mask dword per lane and row;  VEC = 1: anything else.
// BITS: mask_dev holds the compact mask (SGX_STEP_COMPACT_MASK: uint32 [MB_WORDS] per game, bit a = action a) instead of bytes.
template <int R_, int C_, int VEC, bool BITS>
__global__ __launch_bounds__(256) void choose_kernel(const KParams P, const float *__restrict__ logits, const void *__restrict__ mask, const float scale,
                                                     int32_t *__restrict__ actions) {
    using G = Geo<R_, C_>;
    using CG = ChooseGeo<G, VEC>;
    constexpr int NA = CG::NA, ROWS = CG::ROWS, LPG = G::LPG;
    __shared__ uint32_t rowtot[CG::GAMES][CG::CACHE ? 1 : ROWS];
    const int lane = threadIdx.x & (LPG - 1), slot = threadIdx.x / LPG;
    const int64_t env = blockIdx.x * (int64_t)CG::GAMES + slot;
    if (env >= P.n_envs) return;
    const float *lg = logits + env * (int64_t)NA;
    const uint8_t *mb = BITS ? nullptr : reinterpret_cast<const uint8_t *>(mask) + env * (int64_t)NA;
    const uint32_t *mw = BITS ? reinterpret_cast<const uint32_t *>(mask) + env * (int64_t)G::MB_WORDS : nullptr;
    const int4 sc = rec_scal<G>(P.boards, P.rec_bytes, env)[0];       // {turn, flags, max_turns, game_no}: the draw's key

    // one row of this lane: VEC logits, -inf where the action is invalid, lies beyond the game's actions, or the logit is NaN
    auto load_row = [&](int j, float (&v)[VEC], bool stream) {
        const int ch = j * LPG + lane, i = ch * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = -INFINITY;
        if (i >= NA) return;
        uint32_t vm;
        if constexpr (VEC == 4) {
            const f32x4 q = stream ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(lg) + ch) : reinterpret_cast<const f32x4 *>(lg)[ch];
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            if constexpr (BITS) vm = (mw[i >> 5] >> (i & 31)) & 15u;
            else {
                const uint32_t m4 = reinterpret_cast<const uint32_t *>(mb)[ch];
                vm = ((m4 & 0xFFu) ? 1u : 0u) | ((m4 & 0xFF00u) ? 2u : 0u) | ((m4 & 0xFF0000u) ? 4u : 0u) | ((m4 & 0xFF000000u) ? 8u : 0u);
            }
        } else {
            v[0] = stream ? __builtin_nontemporal_load(lg + i) : lg[i];
            if constexpr (BITS) vm = (mw[i >> 5] >> (i & 31)) & 1u;
            else vm = mb[i] ? 1u : 0u;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (!((vm >> e) & 1u) || !(v[e] == v[e])) v[e] = -INFINITY;
    };
    auto chunk_sum = [&](const float (&v)[VEC], float mx) {
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) c += choose_weight(v[e], mx, scale);
        return c;
    };

    // ---- pass 1: the largest logit among the valid actions (all of the game's loads in flight at once where they fit the registers)
    float cv[CG::CACHE ? ROWS : 1][VEC];
    float mx = -INFINITY;
    if constexpr (CG::CACHE) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) load_row(j, cv[j], true);
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, cv[j][e]);
    } else {
        for (int j = 0; j < ROWS; ++j) {
            float v[VEC];
            load_row(j, v, false);                                    // (read again below: through the caches)
#pragma unroll
            for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, v[e]);
        }
    }
    mx = gmax<G>(mx);
    if (mx == -INFINITY) {                                             // no valid action with a logit above -inf (or an empty mask)
        if (lane == 0) actions[env] = -1;
        return;
    }

    // ---- pass 2: row totals of the fixed-point weights, the draw, the row that holds it
    const uint64_t r = sgx_rng(P.seed, (uint64_t)(P.env_id_offset + env), (uint64_t)sc.w, STREAM_ACTION, (uint32_t)sc.x);
    unsigned long long total = 0, target = 0, before = 0;
    int jr = 0;
    if constexpr (CG::CACHE) {
        uint32_t t[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { t[j] = gsum<G>(chunk_sum(cv[j], mx)); total += t[j]; }
        target = __umul64hi((r >> 32) << 32, total);                   // floor(r32 * total / 2^32) < total
        bool found = false;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            if (!found && before + t[j] > target) { jr = j; found = true; }
            if (!found) before += t[j];
        }
    } else {
        for (int j = 0; j < ROWS; ++j) {
            float v[VEC];
            load_row(j, v, false);
            const uint32_t t = gsum<G>(chunk_sum(v, mx));
            if (lane == 0) rowtot[slot][j] = t;
            total += t;
        }
        wave_sync<G>();
        target = __umul64hi((r >> 32) << 32, total);
        for (int j = 0; j < ROWS; ++j) {
            const uint32_t t = rowtot[slot][j];
            if (before + t > target) { jr = j; break; }
            before += t;
        }
    }
    // ---- the lane and the action inside that row (1 KiB of the game's logits once more: a cache hit)
    float v[VEC];
    load_row(jr, v, false);
    uint32_t w[VEC], c = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { w[e] = choose_weight(v[e], mx, scale); c += w[e]; }
    const unsigned long long incl = before + (unsigned long long)(uint32_t)gscan_incl<G>((int)c);
    const unsigned long long hit = gballot<G>(incl > target);
    const int l = __ffsll((long long)hit) - 1;                         // (hit != 0: the row's total exceeds target - before)
    if (lane == l) {
        unsigned long long cum = incl - c;
        int pick = VEC - 1;
        bool found = false;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            cum += w[e];
            if (!found && cum > target) { pick = e; found = true; }
        }
        actions[env] = (jr * LPG + lane) * VEC + pick;
    }
}

}  // namespace
