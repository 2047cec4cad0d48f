// cmhsa_bench.cpp -- native (no Python, no torch) driver of the C ABI in include/cmhsa.h.
// Used for rocprofv3 counter passes (the program after `--` must not re-exec) and for quick
// A/B timing of kernel variants.  Build:  make -C tools     Run:  tools/cmhsa_bench B H T [steps] [warmup] [stamped block] [nows]
// (the seventh argument "nows" calls cmhsa_fp32_fwd, the entry point without a workspace; default: cmhsa_fp32_fwd_ws)
// Inputs are U[0,1) (the reference test's distribution, test_torch_extension.py:16-18) laid out as
// QKV column slices of one (B, T, 3*d_model) buffer, exactly what the model hands over.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include <algorithm>

#include "cmhsa.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

#ifdef CMHSA_STAMPS
extern "C" int cmhsa_debug_set_stamps(unsigned long long *buf, int block);
extern "C" int cmhsa_debug_set_block_times(unsigned long long *buf);
#endif

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, H = argc > 2 ? atoi(argv[2]) : 12, T = argc > 3 ? atoi(argv[3]) : 1024;
    const int steps = argc > 4 ? atoi(argv[4]) : 200, warmup = argc > 5 ? atoi(argv[5]) : 20;
    const int64_t d = 64LL * H, rs = 3 * d, n_in = (int64_t)B * T * rs, n_out = (int64_t)B * T * d;
    std::vector<float> h(n_in);
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (auto &x : h) x = u(rng);
    float *qkv = nullptr, *o = nullptr;
    CHECK(hipMalloc(&qkv, n_in * sizeof(float)));
    CHECK(hipMalloc(&o, n_out * sizeof(float)));
    CHECK(hipMemcpy(qkv, h.data(), n_in * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const bool use_ws = !(argc > 7 && std::string(argv[7]) == "nows");
    void *ws = nullptr;
    const size_t ws_bytes = use_ws ? cmhsa_workspace_bytes() : 0;
    if (use_ws) {
        CHECK(hipMalloc(&ws, ws_bytes));
        if (int rc = cmhsa_workspace_init(ws, ws_bytes, s)) { fprintf(stderr, "cmhsa: %s\n", cmhsa_status_string(rc)); return 1; }
    }
    {
        char desc[512];
        cmhsa_plan_describe(B, T, H, use_ws ? 1 : 0, desc, sizeof(desc));
        fprintf(stderr, "plan: %s\n", desc);
    }
    auto run = [&]() {
        return cmhsa_fp32_fwd_ws(qkv, qkv + d, qkv + 2 * d, o, rs, rs, rs, d, (int64_t)T * rs, (int64_t)T * rs, (int64_t)T * rs,
                                 (int64_t)T * d, B, T, H, 64, ws, ws_bytes, s);
    };
    for (int i = 0; i < warmup; ++i) if (int rc = run()) { fprintf(stderr, "cmhsa: %s\n", cmhsa_status_string(rc)); return 1; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < steps; ++i) run();
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = cmhsa_algorithmic_flops(B, T, H, 64), us = ms * 1e3 / steps;
    std::vector<float> ho(64);
    CHECK(hipMemcpy(ho.data(), o, 64 * sizeof(float), hipMemcpyDeviceToHost));
    printf("{\"B\": %d, \"H\": %d, \"T\": %d, \"steps\": %d, \"us_per_launch\": %.3f, \"tflops\": %.3f, \"frac_of_157.3\": %.4f, \"o00\": %.6f}\n",
           B, H, T, steps, us, flops / us / 1e6, flops / us / 1e6 / 157.3, ho[0]);
#ifdef CMHSA_STAMPS
    {   // per-workgroup entry/exit times (s_memrealtime, 10 ns ticks) of one launch: dispatch ramp and tail
        const int maxb = 1 << 16;
        unsigned long long *bt = nullptr;
        CHECK(hipMalloc(&bt, 2 * maxb * sizeof(unsigned long long)));
        CHECK(hipMemset(bt, 0, 2 * maxb * sizeof(unsigned long long)));
        if (cmhsa_debug_set_block_times(bt)) return 1;
        for (int i = 0; i < 3; ++i) run();
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hb(2 * maxb);
        CHECK(hipMemcpy(hb.data(), bt, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        cmhsa_debug_set_block_times(nullptr);
        unsigned long long t0 = ~0ull; int nb = 0;
        for (int i = 0; i < maxb; ++i) if (hb[2 * i]) { nb = i + 1; if (hb[2 * i] < t0) t0 = hb[2 * i]; }
        printf("blocks %d; per block: start_us end_us (relative to first start)\n", nb);
        for (int i = 0; i < nb; i += (nb > 64 ? nb / 48 : 1))
            printf("  blk %4d: %7.2f %7.2f\n", i, (hb[2 * i] - t0) * 0.01, (hb[2 * i + 1] - t0) * 0.01);
        double last_start = 0, last_end = 0;
        for (int i = 0; i < nb; ++i) { last_start = std::max(last_start, (hb[2 * i] - t0) * 0.01); last_end = std::max(last_end, (hb[2 * i + 1] - t0) * 0.01); }
        printf("  last start %.2f us, last end %.2f us\n", last_start, last_end);
    }
    {   // one extra launch with stamps of workgroup `argv[6]` (default 0 = a heaviest query block)
        unsigned long long *sb = nullptr;
        CHECK(hipMalloc(&sb, 8 * 128 * sizeof(unsigned long long)));
        CHECK(hipMemset(sb, 0, 8 * 128 * sizeof(unsigned long long)));
        if (cmhsa_debug_set_stamps(sb, argc > 6 ? atoi(argv[6]) : 0)) return 1;
        for (int i = 0; i < 3; ++i) run();
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hs(8 * 128);
        CHECK(hipMemcpy(hs.data(), sb, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const unsigned long long t0 = hs[0];
        for (int w = 0; w < 8; ++w) {
            const unsigned long long *p = &hs[w * 128];
            printf("wave %d: entry %lld  q_loaded %lld  loop_end %lld  done %lld | units (top, qk_start, qk_issued, pv_start):\n", w,
                   (long long)(p[0] - t0), (long long)(p[1] - t0), (long long)(p[2] - t0), (long long)(p[3] - t0));
            printf("    merge: ml_published %lld | prologue: kernargs %lld  k0_issued %lld  k0_in_lds %lld  after_barrier %lld\n", (long long)(p[120] - t0), (long long)(p[124] - t0), (long long)(p[121] - t0), (long long)(p[122] - t0), (long long)(p[123] - t0));
            for (int i = 4; i + 3 < 120 && p[i]; i += 4)
                printf("    %lld %lld %lld %lld\n", (long long)(p[i] - t0), (long long)(p[i + 1] - t0), (long long)(p[i + 2] - t0), (long long)(p[i + 3] - t0));
        }
    }
#endif
    return 0;
}
