// slide_lab.hip -- diagnostic (not product): timing and phase stamps of the sliding-window convolution (tedm_amd/csrc/conv_slide.hip)
// in one process on one device.  Variants are the product source included under its own namespace with different switches.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Itedm_amd/csrc tools/slide_lab.hip -o tools/build/slide_lab
//   run:   tools/build/slide_lab [iters] [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <utility>
#include "common.h"
#include "../../include/tedm_hip.h"
#include "pack.hip"
#include "conv_slide_common.h"

#define SL_TRACE 1      // every variant carries the stamps (dormant while ConvArgs::trace is null)
namespace v0 {
#include "conv_slide.hip"
}
namespace s128 {
int g_conv_slide = 1, g_conv_slide_rows = 0;
#include "conv_slide128.hip"
static int launch(const ConvArgs& a, int, hipStream_t st) { return conv_slide128_launch(a, st); }
}
#ifdef SL_LAB_EXTRA
namespace g1 {
#define SL_GAPMODE 1
#define SL_ABL 2
#include "conv_slide.hip"
#undef SL_GAPMODE
#undef SL_ABL
}
namespace g2 {
#define SL_GAPMODE 2
#define SL_ABL 2
#include "conv_slide.hip"
#undef SL_GAPMODE
#undef SL_ABL
}
namespace g3 {
#define SL_GAPMODE 3
#define SL_ABL 2
#include "conv_slide.hip"
#undef SL_GAPMODE
#undef SL_ABL
}
namespace x5 {
#define SL_ABL 5
#include "conv_slide.hip"
#undef SL_ABL
}
namespace x1 {
#define SL_ABL 1
#include "conv_slide.hip"
#undef SL_ABL
}
namespace x2 {
#define SL_ABL 2
#include "conv_slide.hip"
#undef SL_ABL
}
namespace x3 {
#define SL_ABL 3
#include "conv_slide.hip"
#undef SL_ABL
}
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int C1, Cout, S, gn; int C2 = 0; };

typedef int (*launch_fn)(const ConvArgs&, int, hipStream_t);

static float time_variant(const char* label, launch_fn f, const ConvArgs& a, int pro, int iters) {
    int rc = f(a, pro, 0);
    hipError_t err = hipDeviceSynchronize();
    if (rc != 0 || err != hipSuccess) { printf("variant '%s' failed: rc %d, %s\n", label, rc, hipGetErrorString(err)); fflush(stdout); exit(3); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f(a, pro, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const int B = argc > 3 ? atoi(argv[3]) : 32;
    const Shape shapes[] = {{"l0_64_64", 64, 64, 256, 0}, {"l0_64_64_gn", 64, 64, 256, 1}, {"l0_128_64", 64, 64, 256, 0, 64}};
    for (const Shape& sh : shapes) {
        const size_t n1 = (size_t)B * sh.S * sh.S * sh.C1, no = (size_t)B * sh.S * sh.S * sh.Cout;
        float *x1, *x2 = nullptr, *out, *w, *bias, *ga, *gb, *stats;
        const int Cin = sh.C1 + sh.C2;
        void* wpk;
        unsigned long long* trace;
        CK(hipMalloc(&x1, n1 * 4)); CK(hipMalloc(&out, no * 4));
        if (sh.C2) CK(hipMalloc(&x2, n1 * 4));
        CK(hipMalloc(&w, (size_t)sh.Cout * Cin * 9 * 4)); CK(hipMalloc(&bias, sh.Cout * 4));
        CK(hipMalloc(&ga, (size_t)B * sh.C1 * 4)); CK(hipMalloc(&gb, (size_t)B * sh.C1 * 4));
        const size_t E = (size_t)(sh.S / 4) * (sh.S / 32);
        CK(hipMalloc(&stats, (size_t)B * E * sh.Cout * 2 * 4));
        CK(hipMalloc(&wpk, (size_t)((sh.Cout + 31) / 32) * (Cin / 16) * 9 * 2 * 1024));
        const size_t ntr = (size_t)B * 8 * 32 * 4 * 64;              // >= grid * 4 waves * 64 stamps for every row split
        CK(hipMalloc(&trace, ntr * 8)); CK(hipMemset(trace, 0, ntr * 8));
        {
            std::vector<float> h(std::max(n1, (size_t)sh.Cout * Cin * 9));
            auto fill = [&](float* d, size_t n, float scale) {
                for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() + rand() + rand() + rand()) / (float)RAND_MAX - 2.f) * 1.7f;
                CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
            };
            fill(x1, n1, 1.f); if (x2) fill(x2, n1, 1.f);
            fill(w, (size_t)sh.Cout * Cin * 9, 1.f / sqrtf(Cin * 9.f)); fill(bias, sh.Cout, 0.1f);
            fill(ga, (size_t)B * sh.C1, 0.1f); fill(gb, (size_t)B * sh.C1, 0.1f);
        }
        if (pack_conv_weights(w, sh.Cout, Cin, 3, 3, wpk, ENC_F16F4, 0) != 0) { printf("pack failed\n"); return 1; }
        ConvArgs a{};
        a.x1 = x1; a.C1 = sh.C1; a.x2 = x2; a.C2 = sh.C2; a.B = B; a.H = sh.S; a.W = sh.S; a.Ho = sh.S; a.Wo = sh.S; a.Cout = sh.Cout;
        a.wpk = wpk; a.bias = bias; a.out = out; a.stats = stats; a.stats_mode = STATS_SUMSQ; a.nk16 = Cin / 16; a.prec = TEDM_PREC_F16F4;
        if (sh.gn) { a.pro_a = ga; a.pro_b = gb; }
        const int pro = sh.gn ? PRO_GN : PRO_NONE;
        const double flops = 2.0 * B * sh.S * sh.S * sh.Cout * Cin * 9;
        struct V { const char* label; launch_fn f; int rows; int* rows_var; };
        std::vector<V> vs;
        if (sh.C2) vs = {{"slide128 rows 256", s128::launch, 256, &s128::g_conv_slide_rows}, {"slide128 rows 128", s128::launch, 128, &s128::g_conv_slide_rows}};
        else vs = {{"slide rows 256", v0::conv_slide_launch, 256, &v0::g_conv_slide_rows},
                             {"slide rows 128", v0::conv_slide_launch, 128, &v0::g_conv_slide_rows},
#ifdef SL_LAB_EXTRA
                             {"no encoder, rows 256", x1::conv_slide_launch, 256, &x1::g_conv_slide_rows},
                             {"no stores, rows 256", x2::conv_slide_launch, 256, &x2::g_conv_slide_rows},
                             {"neither, rows 256", x3::conv_slide_launch, 256, &x3::g_conv_slide_rows},
                             {"mfma + lds reads only", x5::conv_slide_launch, 256, &x5::g_conv_slide_rows},
#endif
        };
        std::vector<std::vector<float>> t(vs.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t i = 0; i < vs.size(); ++i) {
                *vs[i].rows_var = vs[i].rows;
                t[i].push_back(time_variant(vs[i].label, vs[i].f, a, pro, iters));
            }
        for (size_t i = 0; i < vs.size(); ++i) {
            std::sort(t[i].begin(), t[i].end());
            printf("%-12s %-22s min %7.1f us  med %7.1f us  %7.1f TF/s\n", sh.name, vs[i].label, t[i][0] * 1e3, t[i][t[i].size() / 2] * 1e3,
                   flops / (t[i][t[i].size() / 2] * 1e-3) / 1e12);
        }
        // phase stamps, whole strips: per band the stamps are 0 band start, 1 MFMA loop done, 2 band start in 100 MHz ticks, 3 barrier passed
        const int grid = B * (sh.S / 32);
        for (size_t i = 0; i < vs.size(); ++i) {
            if (vs[i].rows != 256 || sh.C2) continue;
            ConvArgs at = a; at.trace = trace;
            *vs[i].rows_var = 256;
            CK(hipMemset(trace, 0, ntr * 8));
            for (int k = 0; k < 3; ++k) vs[i].f(at, pro, 0);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> ht((size_t)grid * 4 * 64);
            CK(hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> dm, dt, ck;
            for (int g = 0; g < grid; ++g)
                for (int wv = 0; wv < 4; ++wv) {
                    const unsigned long long* s = &ht[((size_t)g * 4 + wv) * 64];
                    for (int b = 2; b < 10; ++b) { dm.push_back((double)(s[b * 4 + 1] - s[b * 4])); dt.push_back((double)(s[b * 4 + 4] - s[b * 4])); }
                    ck.push_back((double)(s[10 * 4] - s[2 * 4]) / (double)(s[10 * 4 + 2] - s[2 * 4 + 2]) * 100.0);
                }
            std::sort(dm.begin(), dm.end()); std::sort(dt.begin(), dt.end()); std::sort(ck.begin(), ck.end());
            printf("%-12s %-22s per band: MFMA phase median %6.0f cycles, band median %6.0f (p90 %6.0f); clock %4.0f MHz\n", sh.name, vs[i].label,
                   dm[dm.size() / 2], dt[dt.size() / 2], dt[dt.size() * 9 / 10], ck[ck.size() / 2]);
        }
        fflush(stdout);
        hipFree(x1); if (x2) hipFree(x2); hipFree(out); hipFree(w); hipFree(bias); hipFree(ga); hipFree(gb); hipFree(stats); hipFree(wpk); hipFree(trace);
    }
    return 0;
}
