// What decides the speed class of the observation buffer: the VIRTUAL address it is mapped at, or the
// PHYSICAL memory behind it?  And do physical chunks have a class of their own, so that a fast buffer can be assembled from
// selected chunks?
//   E1  hipMalloc'ed obs-sized buffers: the box's classes.
//   E2  one physical handle mapped at several virtual addresses (2 MiB / 1 GiB / 64 GiB aligned, +2 MiB, +128 MiB): same
//       memory, different VA.  Several handles: different memory, same kind of VA.
//   E3  256 MiB physical chunks timed one by one with an 8,192-game env (its observations fill 219 MB), twice; then obs-sized
//       buffers assembled from the fastest / the slowest / every other chunk and timed with the 65,536-game env.
//   hipcc -O2 -I include tools/microbench/place_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,$PWD/stratego_env_amd/_build -o tools/microbench/place_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;
static int8_t *player_d;

static float time_observe(sgx_env *h, float *obs, int reps = 6) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask_d, player_d, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask_d, player_d, 0, nullptr);
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
// reserve `n` bytes at a VA that is `align`-aligned plus `off`
static char *reserve(size_t n, size_t align, size_t off) {
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n + off, align, nullptr, 0));
    return (char *)va + off;
}

static sgx_env *make_env(int64_t N) {
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);          // random back-row placement (no table)
    return h;
}

int main(int argc, char **argv) {
    const int n_chunks = argc > 1 ? atoi(argv[1]) : 48;
    const int64_t N = 65536, NS = 8192;
    sgx_env *h = make_env(N), *hs = make_env(NS);
    const size_t bytes = (size_t)N * 100 * 67 * 4, MB = 1u << 20, GB = 1u << 30;
    CK(hipMalloc((void **)&mask_d, (size_t)N * 3700)); CK(hipMalloc((void **)&player_d, N));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    hipMemAllocationProp prop = dev_prop();
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);

    printf("E1 hipMalloc (VA, us):\n");
    std::vector<float *> held;
    for (int i = 0; i < 8; i++) {
        float *p; CK(hipMalloc((void **)&p, bytes));
        held.push_back(p);
        printf("  %p %6.1f\n", (void *)p, time_observe(h, p)); fflush(stdout);
    }

    printf("E2 one physical handle at several VAs (rows: handles; columns: VA kinds)\n");
    const size_t hb = (bytes + 2 * MB - 1) / (2 * MB) * (2 * MB);
    struct VK { size_t align, off; const char *name; };
    const VK kinds[] = {{2 * MB, 0, "2M"}, {GB, 0, "1G"}, {64 * GB, 0, "64G"}, {GB, 2 * MB, "1G+2M"}, {GB, 128 * MB, "1G+128M"},
                        {GB, 64 * 1024, "1G+64K"}, {2 * MB, 0, "2M again"}};
    printf("  %-8s", "handle");
    for (auto &k : kinds) printf(" %9s", k.name);
    printf("\n");
    std::vector<hipMemGenericAllocationHandle_t> e2h;
    for (int i = 0; i < 5; i++) {
        hipMemGenericAllocationHandle_t ph;
        CK(hipMemCreate(&ph, hb, &prop, 0));
        e2h.push_back(ph);
        printf("  %-8d", i);
        for (auto &k : kinds) {
            if (k.off % gran) { printf(" %9s", "n/a"); continue; }
            char *va = reserve(hb, k.align, k.off);
            CK(hipMemMap(va, hb, 0, ph, 0));
            set_rw(va, hb);
            printf(" %9.1f", time_observe(h, (float *)va)); fflush(stdout);
            CK(hipMemUnmap(va, hb));
        }
        printf("\n");
    }

    printf("E3 256 MiB chunks, 8,192-game env (two passes)\n");
    const size_t cb = 256 * MB;
    std::vector<hipMemGenericAllocationHandle_t> ch(n_chunks);
    std::vector<float> t1(n_chunks), t2(n_chunks);
    char *cva = reserve(cb, 2 * MB, 0);
    for (int i = 0; i < n_chunks; i++) CK(hipMemCreate(&ch[i], cb, &prop, 0));
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < n_chunks; i++) {
            CK(hipMemMap(cva, cb, 0, ch[i], 0));
            set_rw(cva, cb);
            (pass ? t2 : t1)[i] = time_observe(hs, (float *)cva, 12);
            CK(hipMemUnmap(cva, cb));
        }
    for (int i = 0; i < n_chunks; i++) printf("  chunk %2d  %6.1f %6.1f\n", i, t1[i], t2[i]);
    std::vector<int> order(n_chunks);
    for (int i = 0; i < n_chunks; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return t1[a] + t2[a] < t1[b] + t2[b]; });
    const int per = (int)((bytes + cb - 1) / cb);
    auto assemble = [&](const std::vector<int> &ids, const char *name) {
        char *va = reserve((size_t)per * cb, 2 * MB, 0);
        for (int i = 0; i < per; i++) CK(hipMemMap(va + (size_t)i * cb, cb, 0, ch[ids[i]], 0));
        set_rw(va, (size_t)per * cb);
        const float a = time_observe(h, (float *)va), b = time_observe(h, (float *)va);
        printf("  assembled from %-28s: %6.1f %6.1f us\n", name, a, b); fflush(stdout);
        for (int i = 0; i < per; i++) CK(hipMemUnmap(va + (size_t)i * cb, cb));
    };
    if (n_chunks >= 3 * per) {
        std::vector<int> fast(order.begin(), order.begin() + per), slow(order.end() - per, order.end()), mid, seq, alt;
        for (int i = 0; i < per; i++) { mid.push_back(order[n_chunks / 2 - per / 2 + i]); seq.push_back(i); alt.push_back(2 * i); }
        assemble(fast, "the fastest chunks");
        assemble(slow, "the slowest chunks");
        assemble(mid, "the median chunks");
        assemble(seq, "chunks 0..6 in order");
        assemble(alt, "every other chunk");
        std::reverse(fast.begin(), fast.end());
        assemble(fast, "the fastest chunks, reversed");
    }

    // E4: chunk = the observation range of ONE XCD's games (65,536 / 8 games x 26,800 B = 219,545,600 B): the eight XCDs write
    //     eight fronts that advance together, one per chunk.  Which chunks sit together decides the class?
    {
        const size_t xb = (size_t)(N / 8) * 100 * 67 * 4;
        const int pool = 28;
        std::vector<hipMemGenericAllocationHandle_t> xh(pool);
        for (int i = 0; i < pool; i++) CK(hipMemCreate(&xh[i], xb, &prop, 0));
        char *va = reserve(8 * xb, 2 * MB, 0);
        auto timed = [&](const std::vector<int> &ids) {
            for (int i = 0; i < 8; i++) CK(hipMemMap(va + (size_t)i * xb, xb, 0, xh[ids[i]], 0));
            set_rw(va, 8 * xb);
            const float t = time_observe(h, (float *)va, 8);
            for (int i = 0; i < 8; i++) CK(hipMemUnmap(va + (size_t)i * xb, xb));
            return t;
        };
        printf("E4 one chunk per XCD range (%zu B), pool of %d\n", xb, pool);
        std::vector<int> ids(8);
        for (int i = 0; i < 8; i++) ids[i] = i;
        printf("  chunks 0..7 in order       : %6.1f\n", timed(ids));
        for (int i = 0; i < 8; i++) ids[i] = 7 - i;
        printf("  chunks 7..0                : %6.1f\n", timed(ids));
        for (int i = 0; i < 8; i++) ids[i] = 3 * i;
        printf("  chunks 0,3,6,..            : %6.1f\n", timed(ids));
        unsigned s = 12345;
        for (int rep = 0; rep < 10; rep++) {
            std::vector<int> all(pool);
            for (int i = 0; i < pool; i++) all[i] = i;
            for (int i = pool - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(all[i], all[(s >> 8) % (i + 1)]); }
            for (int i = 0; i < 8; i++) ids[i] = all[i];
            printf("  random draw %d [", rep);
            for (int i = 0; i < 8; i++) printf("%d ", ids[i]);
            printf("]: %6.1f\n", timed(ids)); fflush(stdout);
        }
        // greedy: from chunks 0..7, position by position try every spare chunk, keep the best
        for (int i = 0; i < 8; i++) ids[i] = i;
        float best = timed(ids);
        printf("  greedy from 0..7 (%6.1f):\n", best);
        for (int round = 0; round < 2; round++)
            for (int k = 0; k < 8; k++) {
                printf("    pos %d:", k);
                int keep = ids[k];
                for (int c = 8; c < 16; c++) {
                    bool used = false;
                    for (int j = 0; j < 8; j++) used |= ids[j] == c;
                    if (used) continue;
                    const int old = ids[k];
                    ids[k] = c;
                    const float t = timed(ids);
                    printf(" %d:%5.1f", c, t);
                    if (t < best - 1.0f) { best = t; keep = c; }
                    ids[k] = old;
                }
                ids[k] = keep;
                printf("  -> keep %d, best %6.1f\n", keep, best); fflush(stdout);
            }
        for (int i = 0; i < pool; i++) CK(hipMemRelease(xh[i]));
    }
    // E5: the obs buffer's own 2 MiB / 32 MiB chunks mapped in shuffled order (no extra memory)
    for (size_t cs : {2 * MB, 32 * MB}) {
        const int n = (int)((bytes + cs - 1) / cs);
        std::vector<hipMemGenericAllocationHandle_t> hh(n);
        for (int i = 0; i < n; i++) CK(hipMemCreate(&hh[i], cs, &prop, 0));
        char *va = reserve((size_t)n * cs, 2 * MB, 0);
        printf("E5 %zu MiB chunks of one buffer:", cs / MB);
        unsigned s = 777;
        for (int rep = 0; rep < 7; rep++) {
            std::vector<int> ord(n);
            for (int i = 0; i < n; i++) ord[i] = i;
            if (rep) for (int i = n - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(ord[i], ord[(s >> 8) % (i + 1)]); }
            for (int i = 0; i < n; i++) CK(hipMemMap(va + (size_t)i * cs, cs, 0, hh[ord[i]], 0));
            set_rw(va, (size_t)n * cs);
            printf(" %s%6.1f", rep ? "shuffled " : "in order ", time_observe(h, (float *)va, 8)); fflush(stdout);
            for (int i = 0; i < n; i++) CK(hipMemUnmap(va + (size_t)i * cs, cs));
        }
        printf("\n");
        for (int i = 0; i < n; i++) CK(hipMemRelease(hh[i]));
    }
    printf("done\n");
    return 0;
}
