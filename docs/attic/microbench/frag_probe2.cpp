// Round 3: is a physically CONTIGUOUS observation buffer the slow one and a scattered one the fast one?  65,536-game buffers from
// 2 MiB chunks: the first `need` chunks in creation order, the same chunks shuffled, every 2nd / 4th / 8th of a larger pool (pages
// scattered over 2x / 4x / 8x the range), a random subset of the 8x pool; 256 MiB and whole-buffer chunks and plain hipMalloc for
// reference.  Real kernel, obs + mask and obs only.
//   hipcc -O2 -I include tools/microbench/frag_probe.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/frag_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static uint8_t *mask_d;
static float time_observe(sgx_env *h, float *obs, uint8_t *mask, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, mask, nullptr, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, mask, nullptr, 0, nullptr);
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
static sgx_env *make_env(int64_t N) {
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    static const int lakes[8][2] = {{4, 2}, {5, 2}, {4, 3}, {5, 3}, {4, 6}, {5, 6}, {4, 7}, {5, 7}};
    for (auto &l : lakes) cfg.obstacles[l[0] * 10 + l[1]] = 1;
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    return h;
}

int main(int argc, char **argv) {
    const int64_t N = 65536;
    const size_t MB = 1u << 20;
    sgx_env *h = make_env(N);
    CK(hipMalloc((void **)&mask_d, (size_t)N * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    const size_t bytes = (size_t)N * 26800;
    hipMemAllocationProp prop = dev_prop();
    {
        printf("plain hipMalloc x4 (held):");
        std::vector<float *> held;
        for (int i = 0; i < 4; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); held.push_back(p); printf("  %.1f / %.1f", time_observe(h, p, mask_d), time_observe(h, p, nullptr)); fflush(stdout); }
        for (auto p : held) CK(hipFree(p));
        printf("\n");
    }
    for (size_t cs : {(size_t)2048 * MB, (size_t)1024 * MB, (size_t)512 * MB, (size_t)256 * MB, (size_t)128 * MB, (size_t)64 * MB, (size_t)32 * MB, (size_t)8 * MB, (size_t)2 * MB}) {
        const int need = (int)((bytes + cs - 1) / cs);
        const int mult = 1;
        const int pool = need * mult;
        std::vector<hipMemGenericAllocationHandle_t> ch(pool);
        for (int i = 0; i < pool; i++) CK(hipMemCreate(&ch[i], cs, &prop, 0));
        void *vap = nullptr;
        CK(hipMemAddressReserve(&vap, (size_t)need * cs, 2u << 20, nullptr, 0));
        char *va = (char *)vap;
        auto run = [&](const std::vector<int> &ids, const char *name) {
            for (int i = 0; i < need; i++) CK(hipMemMap(va + (size_t)i * cs, cs, 0, ch[ids[i]], 0));
            set_rw(va, (size_t)need * cs);
            const float a = time_observe(h, (float *)va, mask_d), b = time_observe(h, (float *)va, nullptr);
            printf("  %5zu MiB chunks, %-34s: obs+mask %6.1f   obs only %6.1f us\n", cs / MB, name, a, b); fflush(stdout);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va, (size_t)need * cs));
        };
        std::vector<int> ids(need);
        for (int i = 0; i < need; i++) ids[i] = i;
        run(ids, "first `need` in creation order");
        {
            std::vector<int> sh = ids; unsigned s = 7;
            for (int i = need - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(sh[i], sh[(s >> 8) % (i + 1)]); }
            run(sh, "the same chunks shuffled");
        }
        if (0) for (int stride = 2; stride <= mult; stride *= 2) {
            for (int i = 0; i < need; i++) ids[i] = i * stride;
            char name[64]; snprintf(name, sizeof(name), "every %d. chunk of the pool", stride);
            run(ids, name);
        }
        if (0) {
            std::vector<int> all(pool); unsigned s = 99;
            for (int i = 0; i < pool; i++) all[i] = i;
            for (int i = pool - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(all[i], all[(s >> 8) % (i + 1)]); }
            run(all, "random subset of the pool");
            std::sort(all.begin(), all.begin() + need);
            run(all, "random subset, in creation order");
        }
        { std::vector<int> rev(need); for (int i = 0; i < need; i++) rev[i] = need - 1 - i; run(rev, "reversed"); }
        CK(hipMemAddressFree(va, (size_t)need * cs));
        for (auto &x : ch) CK(hipMemRelease(x));
    }
    {
        const int64_t NL = 262144;
        sgx_env *hl = make_env(NL);
        uint8_t *mask_l; CK(hipMalloc((void **)&mask_l, (size_t)NL * 3700));
        const size_t lbytes = (size_t)NL * 26800;
        { float *p; CK(hipMalloc((void **)&p, lbytes)); printf("262,144 games, plain hipMalloc: obs+mask %.1f  obs only %.1f us\n", time_observe(hl, p, mask_l), time_observe(hl, p, nullptr)); CK(hipFree(p)); }
        for (size_t cs : {(size_t)2048 * MB, (size_t)256 * MB, (size_t)32 * MB, (size_t)2 * MB}) {
            const int need = (int)((lbytes + cs - 1) / cs);
            std::vector<hipMemGenericAllocationHandle_t> ch(need);
            for (int i = 0; i < need; i++) CK(hipMemCreate(&ch[i], cs, &prop, 0));
            void *vap = nullptr;
            CK(hipMemAddressReserve(&vap, (size_t)need * cs, 2u << 20, nullptr, 0));
            char *va = (char *)vap;
            for (int i = 0; i < need; i++) CK(hipMemMap(va + (size_t)i * cs, cs, 0, ch[i], 0));
            set_rw(va, (size_t)need * cs);
            printf("262,144 games, %5zu MiB chunks: obs+mask %.1f  obs only %.1f us\n", cs / MB, time_observe(hl, (float *)va, mask_l), time_observe(hl, (float *)va, nullptr)); fflush(stdout);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va, (size_t)need * cs));
            CK(hipMemAddressFree(va, (size_t)need * cs));
            for (auto &x : ch) CK(hipMemRelease(x));
        }
        // the mask buffer from small chunks too
        {
            const size_t cs = 32 * MB, mb = (size_t)NL * 3700;
            const int need = (int)((lbytes + cs - 1) / cs), mneed = (int)((mb + cs - 1) / cs);
            std::vector<hipMemGenericAllocationHandle_t> ch(need + mneed);
            for (auto &x : ch) CK(hipMemCreate(&x, cs, &prop, 0));
            void *vap = nullptr, *vam = nullptr;
            CK(hipMemAddressReserve(&vap, (size_t)need * cs, 2u << 20, nullptr, 0));
            CK(hipMemAddressReserve(&vam, (size_t)mneed * cs, 2u << 20, nullptr, 0));
            for (int i = 0; i < need; i++) CK(hipMemMap((char *)vap + (size_t)i * cs, cs, 0, ch[i], 0));
            for (int i = 0; i < mneed; i++) CK(hipMemMap((char *)vam + (size_t)i * cs, cs, 0, ch[need + i], 0));
            set_rw(vap, (size_t)need * cs); set_rw(vam, (size_t)mneed * cs);
            printf("262,144 games, obs AND mask from 32 MiB chunks: obs+mask %.1f  mask only %.1f (hipMalloc'ed mask only: %.1f) us\n",
                   time_observe(hl, (float *)vap, (uint8_t *)vam), time_observe(hl, nullptr, (uint8_t *)vam), time_observe(hl, nullptr, mask_l));
            CK(hipDeviceSynchronize());
        }
    }
    printf("done\n");
    return 0;
}
