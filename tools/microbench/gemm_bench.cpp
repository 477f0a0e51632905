// Kernel-level A/B of the GEMM tile configurations through the C ABI test hook (ezdit_test_gemm), without Python:
// every configuration is checked against a naive fp32 reference kernel first, then timed back-to-back on random data.
//
//   hipcc --offload-arch=gfx950 -O2 -x hip tools/microbench/gemm_bench.cpp -o tools/_run/gemm_bench -Iinclude -Lezaudio_amd -lezaudio_hip \
//         -Wl,-rpath,'$ORIGIN/../../ezaudio_amd'
//   tools/_run/gemm_bench [filter]
//
// variant = 256000 * VAR + 8000 * lds + 4 * tile + epi   (epi: 0 fp32 + bias, 1 split-K slabs, 2 GEGLU; lds: bf16 epilogue through LDS)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <string>

#include "ezdit.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void k_ref(const uint16_t* A, int lda, const uint16_t* W, int ldw, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += __uint_as_float((uint32_t)A[(long)m * lda + k] << 16) * __uint_as_float((uint32_t)W[(long)n * ldw + k] << 16);
    C[(long)m * N + n] = s;
}

// ---- L2-residency experiment (resid section, "l2" lines): what would a launch gain if the PREVIOUS kernel had pulled its weights into every
// XCD's L2?  k_stream evicts the L2s (not the Infinity Cache) by streaming a buffer larger than 8 x 4 MB through all XCDs; k_touch makes every
// XCD (workgroup b runs on XCD b % 8) read every 128-byte line of [p, p + bytes): afterwards each of the eight L2s holds the whole range.
__global__ void k_stream(const uint4* p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;   // never true for the buffers used here: keeps the loads
}
__global__ void k_touch(const char* p, size_t bytes, unsigned* sink) {
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    (void)xcd;
    unsigned acc = 0;
    const size_t lines = (bytes + 127) / 128;
    for (size_t i = (size_t)l * blockDim.x + threadIdx.x; i < lines; i += (size_t)per_xcd * blockDim.x) acc ^= *reinterpret_cast<const unsigned*>(p + i * 128);
    if (acc == 0x12345678u) *sink = acc;
}

struct Shape { const char* name; int M, N, K; };
struct Cfg { const char* name; int tile, epi, splitk, var, lds; };

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : "";
    const char* cfilter = argc > 2 ? argv[2] : "";
    const int iters = 40;
    const Shape shapes[] = {
        {"geglu", 1000, 9216, 1152}, {"geglu4k", 1000, 9216, 4608}, {"qkv", 1000, 3456, 1152}, {"dxd", 1000, 1152, 1152},
        {"skip", 1000, 1152, 2304}, {"mlpout", 1000, 1152, 4608}, {"dxd_b4", 4000, 1152, 1152}, {"mlpout_b4", 4000, 1152, 4608}, {"geglu_b4", 4000, 9216, 1152}, {"qkv_b4", 4000, 3456, 1152}, {"odd", 77, 288, 192}, {"odd1", 130, 576, 64}, {"odd2", 200, 432, 128}, {"odd5", 1000, 288, 320},
    };
    // which configurations run on which shape
    std::vector<Cfg> wide = {   // N >= 3456
        {"old 128x288 12w r3 (13)", 13, 2, 1, 0, 1}, {"pp 128x288 s1 r3 (60)", 60, 2, 1, 0, 1}, {"pp60 LN-algebra epilogue", 60, 2, 1, 0, 9},
        {"co 128x144 4w r2 x2/CU (66)", 66, 2, 1, 0, 1}, {"co66 LN-algebra epilogue", 66, 2, 1, 0, 9}, {"co66 half-tile phase offset", 66, 2, 1, 1, 1}, {"co66 half-lifetime offset", 66, 2, 1, 2, 1},
        {"pp 128x144 s2 r4 (61) geglu", 61, 2, 1, 0, 1}, {"pp 128x128 s1 r3 (62) geglu", 62, 2, 1, 0, 1},
        {"pp60 abl8 noMFMA", 60, 2, 1, 8, 1}, {"pp60 abl16 noReads", 60, 2, 1, 16, 1}, {"pp60 abl32 noDMA", 60, 2, 1, 32, 1},
        {"pp60 abl24 noMFMA noReads", 60, 2, 1, 24, 1}, {"pp60 abl40 noMFMA noDMA", 60, 2, 1, 40, 1}, {"pp60 abl48 noReads noDMA", 60, 2, 1, 48, 1}, {"pp60 abl56 barriers only", 60, 2, 1, 56, 1},
        {"old 128x288 f32 (13)", 13, 0, 1, 0, 0}, {"pp60 f32", 60, 0, 1, 0, 0}, {"pp61 f32", 61, 0, 1, 0, 0}, {"co66 f32", 66, 0, 1, 0, 0},
        {"pp62 f32", 62, 0, 1, 0, 0}, {"old 128x128 8w r3 f32 (9)", 9, 0, 1, 0, 0},
    };
    std::vector<Cfg> narrow = {   // N = 1152
        {"old 128x128 8w r3 split3 (9)", 9, 1, 3, 0, 0}, {"old 9 split1", 9, 1, 1, 0, 0},
        {"pp62 s1 split3", 62, 1, 3, 0, 0}, {"pp62 split2", 62, 1, 2, 0, 0}, {"pp62 split1", 62, 1, 1, 0, 0},
        {"pp61 128x144 r4 split3", 61, 1, 3, 0, 0}, {"pp61 split4", 61, 1, 4, 0, 0},
        {"co66 128x144 split1", 66, 1, 1, 0, 0}, {"co66 split2", 66, 1, 2, 0, 0}, {"co66 split3", 66, 1, 3, 0, 0},
        {"ks 48x96 f32 bias (70)", 70, 0, 1, 0, 0}, {"ks 32x96 f32 bias (72)", 72, 0, 1, 0, 0}, {"ks 48x64 f32 bias (73)", 73, 0, 1, 0, 0},
    };
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    for (const Shape& sh : shapes) {
        if (filter[0] && !strstr(sh.name, filter) && strcmp(filter, "all")) continue;
        const int M = sh.M, N = sh.N, K = sh.K;
        const int Mp = (M + 127) / 128 * 128, Np = (N + 287) / 288 * 288 + 288;
        std::vector<uint16_t> hA((size_t)Mp * K), hW((size_t)Np * K, 0);
        std::vector<float> hb(N);
        const float ws = 1.0f / sqrtf((float)K);
        for (size_t i = 0; i < (size_t)M * K; ++i) hA[i] = f2bf(frand() * 1.7f);
        for (size_t i = (size_t)M * K; i < hA.size(); ++i) hA[i] = 0;
        for (size_t i = 0; i < (size_t)N * K; ++i) hW[i] = f2bf(frand() * ws * 1.7f);
        for (int i = 0; i < N; ++i) hb[i] = frand() * 0.5f;
        uint16_t *dA, *dW; float *db, *dC, *dout;
        CHECK(hipMalloc(&dA, hA.size() * 2)); CHECK(hipMalloc(&dW, hW.size() * 2)); CHECK(hipMalloc(&db, N * 4));
        CHECK(hipMalloc(&dC, (size_t)M * N * 4));
        const size_t out_bytes = (size_t)8 * Mp * (N > 1152 ? N : 1152) * 4;
        CHECK(hipMalloc(&dout, out_bytes));
        CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        k_ref<<<dim3((N + 255) / 256, M), 256, 0, st>>>(dA, K, dW, K, dC, M, N, K);
        CHECK(hipStreamSynchronize(st));
        std::vector<float> hC((size_t)M * N);
        CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        printf("== %s  M=%d N=%d K=%d  (%.2f GFLOP)\n", sh.name, M, N, K, 2.0 * M * N * K * 1e-9);
        const std::vector<Cfg>& cfgs = N > 1152 || !strncmp(sh.name, "odd", 3) ? wide : narrow;
        std::vector<float> hout;
        for (const Cfg& c : cfgs) {
            if (c.epi == 2 && (N % 16)) continue;
            if (cfilter[0] && !strstr(c.name, cfilter)) continue;
            const int variant = 256000 * c.var + 8000 * c.lds + 4 * c.tile + c.epi;
            const int ldo = c.epi == 2 ? N / 2 : N;
            CHECK(hipMemsetAsync(dout, 0xff, out_bytes > ((size_t)1 << 28) ? ((size_t)1 << 28) : out_bytes, st));
            int rc = ezdit_test_gemm(nullptr, variant, dA, K, dW, K, c.epi == 1 ? nullptr : db, dout, ldo, M, N, K, c.splitk, st);
            if (rc) { printf("   %-34s unsupported (%s)\n", c.name, ezdit_last_error()); continue; }
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { printf("   %-34s FAILED: %s\n", c.name, hipGetErrorString(e)); return 1; }
            // ---- check
            double max_err = 0, max_ref = 0;
            long bad = 0;
            if (c.var & 56) {
                // timing ablation: results are garbage by construction
            } else if (c.epi == 2) {
                hout.resize((size_t)M * (N / 2) / 2 + 1);
                CHECK(hipMemcpy(hout.data(), dout, (size_t)M * (N / 2) * 2, hipMemcpyDeviceToHost));
                const uint16_t* o = reinterpret_cast<const uint16_t*>(hout.data());
                for (int m = 0; m < M; ++m)
                    for (int cc = 0; cc < N / 2; ++cc) {
                        const int pv = 16 * (cc / 8) + cc % 8, pg = pv + 8;
                        const double v = hC[(size_t)m * N + pv] + hb[pv], g = hC[(size_t)m * N + pg] + hb[pg];
                        const double ref = v * 0.5 * g * (1.0 + erf(g * 0.7071067811865476));
                        const double got = bf2f(o[(size_t)m * (N / 2) + cc]);
                        const double err = fabs(got - ref);
                        if (err > max_err) max_err = err;
                        if (fabs(ref) > max_ref) max_ref = fabs(ref);
                        if (!(err <= 0.01 * fabs(ref) + 2e-3)) ++bad;
                    }
            } else {
                const int S = c.epi == 1 ? c.splitk : 1;
                hout.resize((size_t)S * Mp * N);
                CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
                for (int m = 0; m < M; ++m)
                    for (int n = 0; n < N; ++n) {
                        double got = 0;
                        for (int s = 0; s < S; ++s) got += hout[((size_t)s * Mp + m) * N + n];
                        const double ref = hC[(size_t)m * N + n] + (c.epi == 0 ? hb[n] : 0.f);
                        const double err = fabs(got - ref);
                        if (err > max_err) max_err = err;
                        if (fabs(ref) > max_ref) max_ref = fabs(ref);
                        if (!(err <= 1e-3 * fabs(ref) + 1e-3)) ++bad;
                    }
            }
            // ---- time
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int i = 0; i < 5; ++i) ezdit_test_gemm(nullptr, variant, dA, K, dW, K, c.epi == 1 ? nullptr : db, dout, ldo, M, N, K, c.splitk, st);
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) ezdit_test_gemm(nullptr, variant, dA, K, dW, K, c.epi == 1 ? nullptr : db, dout, ldo, M, N, K, c.splitk, st);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("   %-34s %8.2f us  %7.1f TF  max|err| %.2e (max|ref| %.1f)%s\n", c.name, us, 2.0 * M * N * K / us * 1e-6, max_err, max_ref,
                   bad ? "  *** MISMATCH ***" : "");
            if (c.tile >= 60) {   // in-kernel stamps of one more launch
                const int NWG = 8192;
                static unsigned long long* dts = nullptr;
                if (!dts) CHECK(hipMalloc(&dts, NWG * 8 * 8));
                CHECK(hipMemsetAsync(dts, 0, NWG * 8 * 8, st));
                ezdit_debug_gemm_timestamps(dts, NWG);
                ezdit_test_gemm(nullptr, variant, dA, K, dW, K, c.epi == 1 ? nullptr : db, dout, ldo, M, N, K, c.splitk, st);
                ezdit_debug_gemm_timestamps(nullptr, 0);
                CHECK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hts(NWG * 8);
                CHECK(hipMemcpy(hts.data(), dts, NWG * 8 * 8, hipMemcpyDeviceToHost));
                unsigned long long t_first = ~0ull, t_last = 0;
                double e1s = 0, e2s = 0, e3s = 0, e4s = 0, pro = 0, loop = 0, epi = 0, pro_max = 0, loop_max = 0, epi_max = 0, start_max = 0;
                int n = 0;
                for (int w = 0; w < NWG; ++w) if (hts[8 * w] && hts[8 * w + 3]) { if (hts[8 * w] < t_first) t_first = hts[8 * w]; if (hts[8 * w + 3] > t_last) t_last = hts[8 * w + 3]; }
                for (int w = 0; w < NWG; ++w) {
                    const unsigned long long* t = &hts[8 * w];
                    if (!t[0] || !t[3]) continue;
                    ++n;
                    const double p_ = (double)(t[1] - t[0]), l_ = (double)(t[2] - t[1]), e_ = (double)(t[3] - t[2]), s_ = (double)(t[0] - t_first);
                    pro += p_; loop += l_; epi += e_;
                    if (t[4]) { e1s += (double)(t[4] - t[2]); e2s += (double)(t[5] - t[4]); e4s += (double)(t[3] - t[5]); }
                    if (p_ > pro_max) pro_max = p_; if (l_ > loop_max) loop_max = l_; if (e_ > epi_max) epi_max = e_; if (s_ > start_max) start_max = s_;
                }
                if (n) printf("      stamps (ticks; %d WGs): span %llu | prologue avg %.0f max %.0f | loop avg %.0f max %.0f (%.1f per K tile) | epilogue avg %.0f max %.0f | last start +%.0f\n",
                              n, t_last - t_first, pro / n, pro_max, loop / n, loop_max, loop / n / ((K / 64) / c.splitk), epi / n, epi_max, start_max);
                {   // per-CU timeline from the 100 MHz device clock ([6] start, tagged with the CU id; [7] end): how many workgroups share a CU at a time, how long a CU idles between them
                    struct Ev { unsigned long long t0, t1; };
                    std::vector<std::vector<Ev>> cu(4096);
                    unsigned long long r0 = ~0ull, r1 = 0; double life = 0; int m = 0;
                    for (int w = 0; w < NWG; ++w) {
                        const unsigned long long* t = &hts[8 * w];
                        if (!t[0] || !t[3] || !t[7]) continue;
                        const unsigned long long a0 = t[6] & 0xffffffffffffull, a1 = t[7] & 0xffffffffffffull;
                        cu[(t[6] >> 48) & 4095].push_back({a0, a1});
                        if (a0 < r0) r0 = a0; if (a1 > r1) r1 = a1; life += (double)(a1 - a0); ++m;
                    }
                    int ncu = 0, maxper = 0, maxconc = 0; double busy1 = 0, busy2 = 0;
                    for (auto& v : cu) {
                        if (v.empty()) continue;
                        ++ncu; if ((int)v.size() > maxper) maxper = (int)v.size();
                        // sweep: time with >= 1 and with >= 2 workgroups resident
                        std::vector<std::pair<unsigned long long, int>> ev;
                        for (auto& e : v) { ev.push_back({e.t0, 1}); ev.push_back({e.t1, -1}); }
                        std::sort(ev.begin(), ev.end());
                        int c = 0; unsigned long long last = 0;
                        for (auto& e : ev) { if (c >= 1) busy1 += (double)(e.first - last); if (c >= 2) busy2 += (double)(e.first - last); c += e.second; if (c > maxconc) maxconc = c; last = e.first; }
                    }
                    if (m) printf("      timeline (10 ns ticks): first start -> last end %.2f us | %d CUs, <= %d workgroups per CU, <= %d resident at once | mean lifetime %.2f us | a CU has >= 1 workgroup %.0f %% and >= 2 %.0f %% of the span\n",
                                  (r1 - r0) * 0.01, ncu, maxper, maxconc, life / m * 0.01, 100.0 * busy1 / ncu / (double)(r1 - r0), 100.0 * busy2 / ncu / (double)(r1 - r0));
                }
                if (n && e2s > 0) printf("      epilogue parts: math + wait + barrier %.0f | park %.0f | barrier + copy-out %.0f\n", e1s / n, e2s / n, e4s / n);
            }
            if (bad) printf("      %ld elements out of tolerance\n", bad);
            fflush(stdout);
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
        if (N == 1152 && (!cfilter[0] || strstr("resid", cfilter))) {   // un-split residual projection with LayerNorm statistics (EPI_RESID)
            float *dh, *dg, *dz; uint16_t* dzu; float* dzs;
            CHECK(hipMalloc(&dh, (size_t)Mp * N * 4)); CHECK(hipMalloc(&dg, N * 4)); CHECK(hipMalloc(&dz, N * 4));
            CHECK(hipMalloc(&dzu, (size_t)Mp * N * 2)); CHECK(hipMalloc(&dzs, (size_t)Mp * (N / 64) * 8));
            CHECK(hipMemset(dh, 0, (size_t)Mp * N * 4)); CHECK(hipMemcpy(dg, hb.data(), N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dz, hb.data(), N * 4, hipMemcpyHostToDevice));
            static char* dflush = nullptr;
            const size_t FLUSH = (size_t)768 << 20;   // > L2 + Infinity Cache
            if (!dflush) CHECK(hipMalloc(&dflush, FLUSH));
            const int tiles[] = {70, 72, 73};
            for (int tile : tiles) {
                auto run = [&]() { return ezdit_test_resid(tile, dA, K, dW, K, db, dh, dg, dz, dout, dzu, N, dzs, M, N, K, st); };
                char name[64]; snprintf(name, sizeof name, "EPI_RESID un-split tile %d", tile);
                if (run()) { printf("   %-34s unsupported (%s)\n", name, ezdit_last_error()); continue; }
                CHECK(hipStreamSynchronize(st));
                // check h_out = 0 + b * (acc + b) against the reference (gate = zg = bias vector here)
                {
                    std::vector<float> ho((size_t)M * N);
                    CHECK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
                    double max_err = 0; long bad = 0;
                    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
                        const double ref = (double)hb[n] * (hC[(size_t)m * N + n] + hb[n]);
                        const double err = fabs(ho[(size_t)m * N + n] - ref);
                        if (err > max_err) max_err = err;
                        if (!(err <= 1e-3 * fabs(ref) + 1e-3)) ++bad;
                    }
                    if (bad) printf("      *** MISMATCH *** %ld elements, max err %.3e\n", bad, max_err);
                }
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                for (int i = 0; i < 5; ++i) run();
                CHECK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) run();
                CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double warm_us = ms * 1e3 / iters;
                // cold operands: everything evicted (768 MB memset), then the activations re-written (as the producing kernel of the live step
                // would leave them: in the Infinity Cache, not in this XCD's L2); the weights come from HBM
                double cold_us = 0;
                const int citers = 10;
                std::vector<hipEvent_t> ev(2 * citers);
                for (auto& e : ev) CHECK(hipEventCreate(&e));
                for (int i = 0; i < citers; ++i) {
                    CHECK(hipMemsetAsync(dflush, i, FLUSH, st));
                    CHECK(hipMemcpyAsync(dA, dA + (size_t)Mp * K / 2, 0, hipMemcpyDeviceToDevice, st));
                    CHECK(hipMemcpyAsync(dflush, dA, (size_t)Mp * K * 2, hipMemcpyDeviceToDevice, st));
                    CHECK(hipMemcpyAsync(dA, dflush, (size_t)Mp * K * 2, hipMemcpyDeviceToDevice, st));
                    CHECK(hipEventRecord(ev[2 * i], st));
                    run();
                    CHECK(hipEventRecord(ev[2 * i + 1], st));
                }
                CHECK(hipStreamSynchronize(st));
                for (int i = 0; i < citers; ++i) { float m1 = 0; CHECK(hipEventElapsedTime(&m1, ev[2 * i], ev[2 * i + 1])); cold_us += m1 * 1e3 / citers; }
                for (auto& e : ev) CHECK(hipEventDestroy(e));
                printf("   %-34s %8.2f us warm  %8.2f us cold (event pair around one launch)  %7.1f TF warm\n", name, warm_us, cold_us, 2.0 * M * N * K / warm_us * 1e-6);
                {
                    // "as inside the step": weights and activations in the Infinity Cache, nothing in the L2s -- then the same with the weights
                    // (and, for scale, also the activations) pulled into every XCD's L2 right in front of the launch.  The difference is the
                    // most a next-kernel weight prefetch issued by the previous kernel could buy this launch.
                    static unsigned* dsink = nullptr;
                    if (!dsink) CHECK(hipMalloc(&dsink, 4));
                    const size_t EVICT = (size_t)96 << 20;   // > 8 x 4 MB of L2, < 256 MB of Infinity Cache
                    const size_t wbytes = (size_t)N * K * 2, abytes = (size_t)M * K * 2;
                    auto timed = [&](int mode) {   // 0: L2-cold, 1: + weights in every L2, 2: + weights and activations in every L2
                        double us = 0;
                        std::vector<hipEvent_t> ev2(2 * citers);
                        for (auto& e : ev2) CHECK(hipEventCreate(&e));
                        for (int i = 0; i < citers; ++i) {
                            k_touch<<<256, 512, 0, st>>>(reinterpret_cast<const char*>(dW), wbytes, dsink);          // into the Infinity Cache (and some L2s)
                            k_touch<<<256, 512, 0, st>>>(reinterpret_cast<const char*>(dA), abytes, dsink);
                            k_stream<<<1024, 256, 0, st>>>(reinterpret_cast<const uint4*>(dflush) + (size_t)(i & 3) * (EVICT / 16), EVICT / 16, dsink);   // out of the L2s
                            if (mode >= 1) k_touch<<<256, 512, 0, st>>>(reinterpret_cast<const char*>(dW), wbytes, dsink);
                            if (mode >= 2) k_touch<<<256, 512, 0, st>>>(reinterpret_cast<const char*>(dA), abytes, dsink);
                            CHECK(hipEventRecord(ev2[2 * i], st));
                            run();
                            CHECK(hipEventRecord(ev2[2 * i + 1], st));
                        }
                        CHECK(hipStreamSynchronize(st));
                        for (int i = 0; i < citers; ++i) { float m1 = 0; CHECK(hipEventElapsedTime(&m1, ev2[2 * i], ev2[2 * i + 1])); us += m1 * 1e3 / citers; }
                        for (auto& e : ev2) CHECK(hipEventDestroy(e));
                        return us;
                    };
                    const double c0 = timed(0), c1 = timed(1), c2 = timed(2);
                    printf("      l2: operands in the Infinity Cache only %8.2f us | + weights (%.1f MB) in every L2 %8.2f us | + activations too %8.2f us   (warm back-to-back %.2f)\n",
                           c0, wbytes / 1048576.0, c1, c2, warm_us);
                }
                const int NWG = 8192;
                unsigned long long* dts; CHECK(hipMalloc(&dts, NWG * 8 * 8)); CHECK(hipMemsetAsync(dts, 0, NWG * 8 * 8, st));
                ezdit_debug_gemm_timestamps(dts, NWG); run(); ezdit_debug_gemm_timestamps(nullptr, 0);
                CHECK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hts(NWG * 8);
                CHECK(hipMemcpy(hts.data(), dts, NWG * 8 * 8, hipMemcpyDeviceToHost));
                double pro = 0, loop = 0, epi = 0, red = 0; int n = 0;
                unsigned long long r0 = ~0ull, r1 = 0;
                for (int w = 0; w < NWG; ++w) { const unsigned long long* t = &hts[8 * w]; if (!t[0] || !t[3]) continue; ++n; pro += t[1] - t[0]; loop += t[2] - t[1]; epi += t[3] - t[2]; if (t[4]) red += t[4] - t[2];
                    if (t[6] < r0) r0 = t[6]; if (t[7] > r1) r1 = t[7]; }
                if (n) printf("      stamps (%d WGs, wave 0): prologue %.0f | loop %.0f (%.1f per K tile) | epilogue %.0f (park + barrier %.0f) | first start -> last end %.2f us\n", n, pro / n, loop / n, loop / n / (K / 64), epi / n, red / n, (r1 - r0) * 0.01);
                CHECK(hipFree(dts));
                CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
                fflush(stdout);
            }
            CHECK(hipFree(dh)); CHECK(hipFree(dg)); CHECK(hipFree(dz)); CHECK(hipFree(dzu)); CHECK(hipFree(dzs));
        }
        CHECK(hipFree(dA)); CHECK(hipFree(dW)); CHECK(hipFree(db)); CHECK(hipFree(dC)); CHECK(hipFree(dout));
    }
    return 0;
}
