// Round 3: does a buffer's class change when it is filled sequentially first (hipMemset), or over time (many launches)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stratego_mi355x.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); fflush(stdout); exit(2); } } while (0)
static float time_observe(sgx_env *h, float *obs, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sgx_observe(h, obs, nullptr, nullptr, nullptr, 0, nullptr);
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) sgx_observe(h, obs, nullptr, nullptr, nullptr, 0, nullptr);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1000.f;
}
static sgx_env *make_env(int64_t N) {
    sgx_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.rows = cfg.cols = 10; cfg.max_turns = 1000; cfg.usable_rows = 4;
    cfg.piece_counts[0] = 1; cfg.piece_counts[1] = 2; cfg.piece_counts[2] = 1; cfg.piece_counts[8] = 1; cfg.piece_counts[9] = 1;
    cfg.piece_counts[10] = 1; cfg.piece_counts[11] = 1;
    sgx_env *h = nullptr;
    if (sgx_create(&cfg, N, 0, 1, 0, &h)) { printf("%s\n", sgx_last_error()); exit(1); }
    sgx_reset(h, nullptr, nullptr, nullptr, nullptr);
    return h;
}
int main() {
    const int64_t N = 65536; const int NB = 10;
    sgx_env *h = make_env(N);
    { float *w; CK(hipMalloc((void **)&w, 1u << 30)); for (int i = 0; i < 300; i++) CK(hipMemsetAsync(w, i, 1u << 30, nullptr)); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    const size_t bytes = (size_t)N * 26800;
    std::vector<float *> held;
    for (int i = 0; i < NB; i++) { float *p; CK(hipMalloc((void **)&p, bytes)); held.push_back(p); }
    printf("fresh        :"); for (auto p : held) printf(" %.1f", time_observe(h, p, 4)); printf("\n"); fflush(stdout);
    for (auto p : held) CK(hipMemset(p, 0, bytes));
    CK(hipDeviceSynchronize());
    printf("after memset :"); for (auto p : held) printf(" %.1f", time_observe(h, p, 4)); printf("\n"); fflush(stdout);
    for (int r = 0; r < 3; r++) for (auto p : held) time_observe(h, p, 50);
    printf("after 150 launches each:"); for (auto p : held) printf(" %.1f", time_observe(h, p, 4)); printf("\n");
    // free every other buffer and allocate again: the same memory back?
    for (int i = 0; i < NB; i += 2) CK(hipFree(held[i]));
    for (int i = 0; i < NB; i += 2) CK(hipMalloc((void **)&held[i], bytes));
    printf("even ones freed and allocated again:"); for (auto p : held) printf(" %.1f", time_observe(h, p, 4)); printf("\n");
    // one big allocation of all the memory the buffers take, carved into buffers
    for (auto p : held) CK(hipFree(p));
    char *big; CK(hipMalloc((void **)&big, bytes * NB));
    printf("one %d x allocation carved up:", NB); for (int i = 0; i < NB; i++) printf(" %.1f", time_observe(h, (float *)(big + i * bytes), 4)); printf("\n");
    printf("done\n");
    return 0;
}
