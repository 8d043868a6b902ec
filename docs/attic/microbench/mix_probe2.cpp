// Round 3, second probe: on ONE box, the current kernel's launch time on
//   P1  successive hipMalloc'ed observation buffers (held), i.e. the classes of this box;
//   P2  buffers built with the VMM API from physical chunks of 2 GiB .. 2 MiB created one after the other, mapped in order,
//       reversed and shuffled;
//   P3  the library's own bounded trial (sgx_alloc_outputs, 8 GiB);
//   P4  what the Infinity Cache absorbs: obs-only / mask-only / both launches at 65,536 and 262,144 games;
//   P5  262,144 games on VMM buffers of several chunk sizes.
//   hipcc -O2 -I include tools/microbench/mix_probe2.cpp -L stratego_env_amd/_build -lstratego_mi355x \
//         -Wl,-rpath,'$ORIGIN/../../stratego_env_amd/_build' -o tools/microbench/mix_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)

static float time_observe(sgx_env *h, float *obs, uint8_t *mask, int reps = 6) {
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

struct VmBuf {
    char *va = nullptr;
    size_t bytes = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
};
// order: 0 = as created, 1 = reversed, 2 = shuffled, 3 = two halves interleaved (0, n/2, 1, n/2+1, ..)
static VmBuf vm_alloc(size_t bytes, size_t chunk, int order) {
    VmBuf b;
    b.chunk = chunk;
    const size_t n = (bytes + chunk - 1) / chunk;
    b.bytes = n * chunk;
    hipMemAllocationProp prop = dev_prop();
    b.hs.resize(n);
    for (size_t i = 0; i < n; i++) CK(hipMemCreate(&b.hs[i], chunk, &prop, 0));
    std::vector<size_t> ord(n);
    for (size_t i = 0; i < n; i++) ord[i] = i;
    if (order == 1) std::reverse(ord.begin(), ord.end());
    if (order == 2) { unsigned s = 4242; for (size_t i = n - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(ord[i], ord[(s >> 8) % (i + 1)]); } }
    if (order == 3) {
        const size_t hn = (n + 1) / 2;
        size_t a = 0, c = hn;
        for (size_t i = 0; i < n; i++) ord[i] = ((i & 1) && c < n) ? c++ : (a < hn ? a++ : c++);
    }
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, b.bytes, 2u << 20, nullptr, 0));
    b.va = (char *)va;
    for (size_t i = 0; i < n; i++) CK(hipMemMap(b.va + i * chunk, chunk, 0, b.hs[ord[i]], 0));
    set_rw(b.va, b.bytes);
    return b;
}
static void vm_free(VmBuf &b) {
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(b.va, b.bytes));
    for (auto &h : b.hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(b.va, b.bytes));
    b = VmBuf();
}

int main(int argc, char **argv) {
    const size_t MB = 1u << 20, GB = 1u << 30;
    const int64_t N = 65536, NL = 262144;
    sgx_env *h = make_env(N);
    uint8_t *mask_d;
    CK(hipMalloc((void **)&mask_d, (size_t)NL * 3700));
    {   // wake the GPU
        float *w; CK(hipMalloc((void **)&w, 1u << 30));
        for (int i = 0; i < 600; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr));
        CK(hipDeviceSynchronize()); CK(hipFree(w));
    }
    const size_t bytes = (size_t)N * 26800, lbytes = (size_t)NL * 26800;

    printf("P1 hipMalloc'ed buffers, held (us):");
    std::vector<float *> held;
    for (int i = 0; i < 12; i++) {
        float *p; CK(hipMalloc((void **)&p, bytes));
        held.push_back(p);
        printf(" %.1f", time_observe(h, p, mask_d)); fflush(stdout);
    }
    printf("\n   again:");
    for (auto p : held) printf(" %.1f", time_observe(h, p, mask_d));
    printf("\n");
    for (auto p : held) CK(hipFree(p));

    printf("P2 VMM buffers for 65,536 games (chunk: in order / reversed / shuffled / halves interleaved)\n");
    for (size_t chunk : {2 * GB, GB, 256 * MB, 32 * MB, 2 * MB}) {
        printf("  %5zu MiB:", chunk / MB);
        for (int order = 0; order < 4; order++) {
            VmBuf b = vm_alloc(bytes, chunk, order);
            printf("  %.1f %.1f", time_observe(h, (float *)b.va, mask_d), time_observe(h, (float *)b.va, mask_d)); fflush(stdout);
            vm_free(b);
        }
        printf("\n");
    }
    printf("   three fresh 2 MiB-chunk buffers in a row, held:");
    {
        std::vector<VmBuf> bs;
        for (int i = 0; i < 3; i++) { bs.push_back(vm_alloc(bytes, 2 * MB, 0)); printf(" %.1f", time_observe(h, (float *)bs.back().va, mask_d)); fflush(stdout); }
        for (auto &b : bs) vm_free(b);
        printf("\n");
    }

    printf("P3 sgx_alloc_outputs (8 GiB budget):");
    {
        sgx_outputs out;
        if (sgx_alloc_outputs(h, 0, (int64_t)8 << 30, 32, nullptr, &out)) { printf(" %s\n", sgx_last_error()); }
        else {
            for (int i = 0; i < out.n_trials; i++) printf(" %.1f", out.trial_us[i]);
            printf("\n   kept buffer with the probe's mask: %.1f us\n", time_observe(h, out.obs_dev, mask_d));
            sgx_free_outputs(h, &out);
        }
    }

    printf("P4 what a launch writes (us; 65,536 games | 262,144 games)\n");
    {
        sgx_env *hl = make_env(NL);
        float *o1, *o2;
        CK(hipMalloc((void **)&o1, bytes)); CK(hipMalloc((void **)&o2, lbytes));
        printf("  obs + mask : %8.1f | %8.1f\n", time_observe(h, o1, mask_d), time_observe(hl, o2, mask_d));
        printf("  obs only   : %8.1f | %8.1f\n", time_observe(h, o1, nullptr), time_observe(hl, o2, nullptr));
        printf("  mask only  : %8.1f | %8.1f\n", time_observe(h, nullptr, mask_d), time_observe(hl, nullptr, mask_d));
        printf("  neither    : %8.1f | %8.1f\n", time_observe(h, nullptr, nullptr), time_observe(hl, nullptr, nullptr));
        sgx_set_nt_stores(h, 0); sgx_set_nt_stores(hl, 0);
        printf("  obs + mask, plain stores : %8.1f | %8.1f\n", time_observe(h, o1, mask_d), time_observe(hl, o2, mask_d));
        sgx_set_nt_stores(h, -1); sgx_set_nt_stores(hl, -1);
        CK(hipFree(o1)); CK(hipFree(o2));
        printf("P5 262,144 games on VMM buffers (in order / shuffled)\n");
        for (size_t chunk : {2 * GB, 256 * MB, 2 * MB}) {
            printf("  %5zu MiB:", chunk / MB);
            for (int order : {0, 2}) {
                VmBuf b = vm_alloc(lbytes, chunk, order);
                printf("  %.1f %.1f", time_observe(hl, (float *)b.va, mask_d), time_observe(hl, (float *)b.va, mask_d)); fflush(stdout);
                vm_free(b);
            }
            printf("\n");
        }
        sgx_destroy(hl);
    }
    printf("done\n");
    return 0;
}
