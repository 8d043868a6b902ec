// Round 3: can a fast observation buffer be BUILT instead of searched for?
//   M1  1 GiB physical chunks created one after the other, each timed alone as the observation buffer of a 40,000-game env
//       (1.07 GB of observations: past the Infinity Cache): does a chunk have a kind of its own at this size, how many kinds?
//   M2  hipMemMap with a non-zero offset into a handle (so that one classified chunk can be mapped piecewise)?
//   M3  65,536-game buffers (1.76 GB) from two classified chunks: fast+fast, slow+slow, fast+slow, slow+fast; and interleaved at
//       64 / 8 / 2 MiB granularity in the ratios 1:1, 2:1, 3:1 (needs M2).
//   M4  262,144 games (Barrage and Standard records alike write 7.0 GB of observations): all-fast, all-slow, interleaved.
//   hipcc -O2 -I include tools/microbench/mix_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,$PWD/stratego_env_amd/_build -o tools/microbench/mix_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;

static float time_observe(sgx_env *h, float *obs, int reps = 6) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask_d, nullptr, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask_d, nullptr, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1000.f;
}

static hipMemAllocationProp dev_prop() {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    return prop;
}
static void set_rw(void *va, size_t n) {
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n, &acc, 1));
}
static char *reserve(size_t n) {
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n, 2u << 20, nullptr, 0));
    return (char *)va;
}

static sgx_env *make_env(int64_t N, bool standard = false) {
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = standard ? 2000 : 1000; cfg.usable_rows = 4;
    if (standard) {
        const int pc[12] = {1, 8, 5, 4, 4, 4, 3, 2, 1, 1, 1, 6};
        for (int i = 0; i < 12; i++) cfg.piece_counts[i] = pc[i];
    } else {
        cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
        cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    }
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);          // random back-row placement (no table)
    return h;
}

int main(int argc, char **argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 40;               // 1 GiB chunks held
    const size_t MB = 1u << 20, GB = 1u << 30;
    const int64_t N = 65536, NP = 40000, NL = 262144;
    sgx_env *h = make_env(N), *hp = make_env(NP);
    CK(hipMalloc((void **)&mask_d, (size_t)NL * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();

    printf("M1 %d chunks of 1 GiB, each alone under a %lld-game env (us; two passes)\n", P, (long long)NP);
    std::vector<hipMemGenericAllocationHandle_t> ch(P);
    std::vector<float> t(P);
    char *cva = reserve(GB);
    for (int i = 0; i < P; i++) CK(hipMemCreate(&ch[i], GB, &prop, 0));
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < P; i++) {
            CK(hipMemMap(cva, GB, 0, ch[i], 0));
            set_rw(cva, GB);
            const float us = time_observe(hp, (float *)cva, 8);
            if (pass == 0) t[i] = us; else t[i] = 0.5f * (t[i] + us);
            printf(" %5.1f", us);
            CK(hipMemUnmap(cva, GB));
        }
        printf("\n"); fflush(stdout);
    }
    std::vector<int> order(P);
    for (int i = 0; i < P; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return t[a] < t[b]; });
    const float tmin = t[order[0]], tmax = t[order[P - 1]], cut = 0.5f * (tmin + tmax);
    std::vector<int> F, S;
    for (int i = 0; i < P; i++) (t[i] < cut ? F : S).push_back(i);
    printf("  min %.1f max %.1f cut %.1f: %zu fast, %zu slow\n", tmin, tmax, cut, F.size(), S.size());
    const bool two_kinds = (tmax - tmin) > 0.04f * tmin && !F.empty() && !S.empty();
    if (!two_kinds) printf("  (no two kinds among the chunks: F = first half, S = second half by position)\n");
    if (!two_kinds) { F.clear(); S.clear(); for (int i = 0; i < P; i++) (i < P / 2 ? F : S).push_back(i); }

    // M2: piecewise mapping of a handle
    bool offset_ok = false;
    {
        char *va = reserve(GB);
        hipError_t e = hipMemMap(va, 512 * MB, 512 * MB, ch[0], 0);
        printf("M2 hipMemMap(size 512 MiB, offset 512 MiB) -> %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            e = hipMemMap(va + 512 * MB, 512 * MB, 0, ch[0], 0);
            printf("   second piece (offset 0) behind it -> %s\n", hipGetErrorString(e));
            if (e == hipSuccess) {
                set_rw(va, GB);
                printf("   chunk 0 with its halves swapped: %.1f us (whole: %.1f)\n", time_observe(hp, (float *)va, 8), t[0]);
                offset_ok = true;
                CK(hipMemUnmap(va + 512 * MB, 512 * MB));
            }
            CK(hipMemUnmap(va, 512 * MB));
        } else (void)hipGetLastError();
        fflush(stdout);
    }

    // generic assembler: pieces (chunk, offset, size) laid out one after the other
    struct Piece { int chunk; size_t off, size; };
    auto timed = [&](sgx_env *env, const std::vector<Piece> &pieces, int reps = 6) {
        size_t total = 0;
        for (auto &p : pieces) total += p.size;
        char *va = reserve(total);
        size_t at = 0;
        for (auto &p : pieces) { CK(hipMemMap(va + at, p.size, p.off, ch[p.chunk], 0)); at += p.size; }
        set_rw(va, total);
        const float a = time_observe(env, (float *)va, reps);
        at = 0;
        for (auto &p : pieces) { CK(hipMemUnmap(va + at, p.size)); at += p.size; }
        CK(hipMemAddressFree(va, total));
        return a;
    };
    // `gib` GiB from the chunk lists A (share num/den of every period) and B, interleaved at granularity g; every chunk is
    // used front to back, so a chunk's bytes appear at most once
    auto interleaved = [&](const std::vector<int> &A, const std::vector<int> &B, int gib, size_t g, int a_per, int b_per) {
        std::vector<Piece> pieces;
        size_t ia = 0, ib = 0, oa = 0, ob = 0, total = 0;
        const size_t want = (size_t)gib * GB;
        while (total < want) {
            for (int k = 0; k < a_per && total < want; k++) {
                if (oa == GB) { ia++; oa = 0; }
                if (ia >= A.size()) return std::vector<Piece>();
                pieces.push_back({A[ia], oa, g}); oa += g; total += g;
            }
            for (int k = 0; k < b_per && total < want; k++) {
                if (ob == GB) { ib++; ob = 0; }
                if (ib >= B.size()) return std::vector<Piece>();
                pieces.push_back({B[ib], ob, g}); ob += g; total += g;
            }
        }
        return pieces;
    };
    auto report = [&](const char *name, sgx_env *env, const std::vector<Piece> &pieces) {
        if (pieces.empty()) { printf("  %-44s: (not enough chunks)\n", name); return; }
        const float a = timed(env, pieces), b = timed(env, pieces);
        printf("  %-44s: %7.1f %7.1f us\n", name, a, b); fflush(stdout);
    };

    printf("M3 65,536 games (1.76 GB) from classified chunks\n");
    if (F.size() >= 2 && S.size() >= 2) {
        report("fast + fast", h, {{F[0], 0, GB}, {F[1], 0, GB}});
        report("slow + slow", h, {{S[0], 0, GB}, {S[1], 0, GB}});
        report("fast + slow", h, {{F[0], 0, GB}, {S[0], 0, GB}});
        report("slow + fast", h, {{S[0], 0, GB}, {F[0], 0, GB}});
        if (offset_ok) {
            for (size_t g : {256 * MB, 64 * MB, 8 * MB, 2 * MB}) {
                char name[96];
                for (auto r : {std::pair<int, int>{1, 1}, {2, 1}, {3, 1}, {1, 2}}) {
                    snprintf(name, sizeof(name), "fast:slow %d:%d at %zu MiB", r.first, r.second, g / MB);
                    report(name, h, interleaved(F, S, 2, g, r.first, r.second));
                }
            }
            report("fast:fast 1:1 at 8 MiB (two fast chunks)", h, interleaved({F[0]}, {F[1]}, 2, 8 * MB, 1, 1));
            report("slow:slow 1:1 at 8 MiB (two slow chunks)", h, interleaved({S[0]}, {S[1]}, 2, 8 * MB, 1, 1));
        }
    }

    printf("M4 262,144 games (7.0 GB of observations)\n");
    for (int standard = 0; standard < 2; standard++) {
        sgx_env *hl = make_env(NL, standard != 0);
        printf(" %s records\n", standard ? "Standard" : "Barrage");
        auto whole = [&](const std::vector<int> &ids, int n) {
            std::vector<Piece> v;
            for (int i = 0; i < n && i < (int)ids.size(); i++) v.push_back({ids[i], 0, GB});
            return (int)v.size() == n ? v : std::vector<Piece>();
        };
        {
            float *p; CK(hipMalloc((void **)&p, (size_t)NL * 26800));
            printf("  %-44s: %7.1f %7.1f us\n", "plain hipMalloc", time_observe(hl, p), time_observe(hl, p));
            CK(hipFree(p));
        }
        report("7 fast chunks", hl, whole(F, 7));
        report("7 slow chunks", hl, whole(S, 7));
        {   // whole chunks alternating F F S
            std::vector<Piece> v;
            size_t fi = 0, si = 0;
            for (int i = 0; i < 7; i++) {
                if (i % 3 == 2) { if (si < S.size()) v.push_back({S[si++], 0, GB}); }
                else if (fi < F.size()) v.push_back({F[fi++], 0, GB});
            }
            if (v.size() == 7) report("whole chunks F F S F F S F", hl, v);
        }
        if (offset_ok) {
            report("fast:slow 2:1 at 64 MiB", hl, interleaved(F, S, 7, 64 * MB, 2, 1));
            report("fast:slow 2:1 at 2 MiB", hl, interleaved(F, S, 7, 2 * MB, 2, 1));
            report("fast:slow 1:1 at 2 MiB", hl, interleaved(F, S, 7, 2 * MB, 1, 1));
        }
        sgx_destroy(hl);
    }
    printf("done\n");
    return 0;
}
