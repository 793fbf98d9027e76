// conv_lab.hip -- diagnostic (not product): A/B timing of conv_igemm_kernel VARIANTS in one process on one device
// (cdna guide rule 24).  Each variant is the product kernel source (tedm_amd/csrc/conv_igemm.inc) included under its own
// namespace with different CONV_LAB_* switches; weights are packed by the product pack kernels, inputs are random.
// No correctness check here (tests/ does that for the variant that ships) -- this measures.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Itedm_amd/csrc tools/conv_lab.hip -o tools/build/conv_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include <math.h>
#include <type_traits>
#define CONV_NO_LAUNCHERS
// -DLAB_PREC=4 measures the single-fp16-product instances (weights packed as fp16 planes)
#ifndef LAB_PREC
#define LAB_PREC 3
#endif
#define CONV_PREC LAB_PREC
#include "common.h"
#include "../../include/tedm_hip.h"
#include "pack.hip"

#define VARIANT(NS, ...) namespace NS { __VA_ARGS__ }
namespace v0 {
#include "conv_igemm.inc"
}
namespace a1 {
#define CONV_LAB_ABL 1
#include "conv_igemm.inc"
}
#undef CONV_LAB_ABL
namespace a4 {
#define CONV_LAB_ABL 4
#include "conv_igemm.inc"
}
#undef CONV_LAB_ABL
namespace a5 {
#define CONV_LAB_ABL 5
#include "conv_igemm.inc"
}
#undef CONV_LAB_ABL
namespace a7 {
#define CONV_LAB_ABL 7
#include "conv_igemm.inc"
}
#undef CONV_LAB_ABL
namespace ns {
#define CONV_LAB_NOSCALE 1
#include "conv_igemm.inc"
}
#undef CONV_LAB_NOSCALE

static size_t conv_packed_bytes_lab(int Cout, int Cin, int taps) { return (size_t)((Cout + 31) / 32) * (Cin / 16) * taps * 2 * 1024; }
#ifdef LAB_PREFETCH    // prefetch policy of the single-fp16-product / f16f4 instances on the 8-row tile (-DLAB_PREFETCH)
namespace p0 {
#define CONV_LAB_PD 0
#include "conv_igemm.inc"
#undef CONV_LAB_PD
}
namespace p2 {
#define CONV_LAB_PD 2
#include "conv_igemm.inc"
#undef CONV_LAB_PD
}
namespace s0 {
#define CONV_LAB_SPREAD 0
#include "conv_igemm.inc"
#undef CONV_LAB_SPREAD
}
namespace p2s0 {
#define CONV_LAB_PD 2
#define CONV_LAB_SPREAD 0
#include "conv_igemm.inc"
#undef CONV_LAB_SPREAD
#undef CONV_LAB_PD
}
#endif
namespace d1 {
#define CONV_LAB_DELAY 16
#include "conv_igemm.inc"
#undef CONV_LAB_DELAY
}
namespace d2 {
#define CONV_LAB_DELAY 32
#include "conv_igemm.inc"
#undef CONV_LAB_DELAY
}
namespace d3 {
#define CONV_LAB_DELAY 64
#include "conv_igemm.inc"
#undef CONV_LAB_DELAY
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int C1, C2, Cout, S, gn; int k = 3; };

// Every variant is checked on its own: launch configuration against the device limits BEFORE the launch, a full device
// synchronisation + error check AFTER the first launch (a faulting variant is named instead of taking the process down with an
// anonymous abort -- the round-2 record gpurun_out/lab8.txt ends in "dumped core" after an uncommitted variant set), events destroyed.
static const char* g_label = "";
template <typename K>
static float time_kernel(K kern, const ConvArgs& a, int threads, size_t lds, int grid, int iters) {
    if (threads < 64 || threads > 1024 || lds > 160 * 1024 || grid < 1 || !a.x1 || !a.out || !a.wpk) {
        printf("variant '%s': bad launch configuration (threads %d, lds %zu, grid %d)\n", g_label, threads, lds, grid);
        exit(2);
    }
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, a);
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) { printf("variant '%s' failed on its first launch: %s\n", g_label, hipGetErrorString(err)); fflush(stdout); exit(3); }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}

// one (variant, tile) measurement of a 3x3 stride-1 conv
#define RUN(NS, WM, WN, MT, PRO, LABEL) RUNT(NS, WM, WN, MT, PRO, 1, LABEL)
#define RUNT(NS, WM, WN, MT, PRO, TPW, LABEL)                                                                              \
    do {                                                                                                             \
        using G = NS::Geo<3, 3, 1, WM, MT, 2, 2, LAB_PREC, WM * WN * 64>;                                                   \
        ConvArgs b = a;                                                                                              \
        const int TR = WM * MT;                                                                                      \
        b.tiles_y = (b.Ho + TR - 1) / TR; b.tiles_x = (b.Wo + 31) / 32;                                              \
        b.ny = ((b.Cout + 31) / 32 + WN - 1) / WN; b.tpw = TPW; b.xcd_order = 1;                                     \
        const int grid = ((b.B * b.tiles_x * b.tiles_y + TPW - 1) / TPW) * b.ny;                                                  \
        g_label = LABEL;                                                                                             \
        const float ms = time_kernel(NS::conv_igemm_kernel<3, 3, 1, 1, 0, WM, WN, MT, 2, 2, PRO, LAB_PREC>, b, WM * WN * 64, \
                                     (size_t)G::LDS, grid, iters);                                                   \
        res.push_back({LABEL, ms});                                                                                  \
    } while (0)

// 1x1 conv (stride 1, no padding): NBUF = 1 (product) or 2 (pipelined double buffer)
#define RUN1(NS, WM, WN, MT, NBUF_, LABEL)                                                                           \
    do {                                                                                                             \
        using G = NS::Geo<1, 1, 1, WM, MT, 2, NBUF_, 3, WM * WN * 64>;                                               \
        ConvArgs b = a;                                                                                              \
        const int TR = WM * MT;                                                                                      \
        b.tiles_y = (b.Ho + TR - 1) / TR; b.tiles_x = (b.Wo + 31) / 32;                                              \
        b.ny = ((b.Cout + 31) / 32 + WN - 1) / WN; b.tpw = 1; b.xcd_order = 1;                                       \
        const int grid = b.B * b.tiles_x * b.tiles_y * b.ny;                                                         \
        const float ms = time_kernel(NS::conv_igemm_kernel<1, 1, 1, 0, 0, WM, WN, MT, 2, NBUF_, 0, 3>, b, WM * WN * 64, \
                                     (size_t)G::LDS, grid, iters);                                                   \
        res.push_back({LABEL, ms});                                                                                  \
    } while (0)

struct Res { const char* label; float ms; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const char* only = argc > 3 ? argv[3] : nullptr;            // run just this shape (profiling)
    const Shape shapes[] = {
        {"l0_64_64", 64, 0, 64, 256, 0}, {"l0_64_64_gn", 64, 0, 64, 256, 1}, {"l0_128_64", 64, 64, 64, 256, 0},
        {"l1_128_128_gn", 128, 0, 128, 128, 1}, {"l1_192_128", 128, 64, 128, 128, 0}, {"l2_128_128_gn", 128, 0, 128, 64, 1}, {"l2_256_256_gn", 256, 0, 256, 64, 1}, {"l2_384_256", 256, 128, 256, 64, 0},
        {"l3_512_512_gn", 512, 0, 512, 32, 1}, {"l3_768_512", 512, 256, 512, 32, 0},
        {"res_768_512", 512, 256, 512, 32, 0, 1}, {"res_384_256", 256, 128, 256, 64, 0, 1}, {"res_192_128", 128, 64, 128, 128, 0, 1},
        {"res_128_64", 64, 64, 64, 256, 0, 1},
    };
    const int B = 32;
    for (const Shape& sh : shapes) {
        if (only && strcmp(only, sh.name)) continue;
        const int Cin = sh.C1 + sh.C2;
        const size_t n1 = (size_t)B * sh.S * sh.S * sh.C1, n2 = (size_t)B * sh.S * sh.S * sh.C2, no = (size_t)B * sh.S * sh.S * sh.Cout;
        float *x1, *x2 = nullptr, *out, *w, *bias, *ga, *gb, *stats;
        void* wpk;
        CK(hipMalloc(&x1, n1 * 4)); if (n2) CK(hipMalloc(&x2, n2 * 4)); CK(hipMalloc(&out, no * 4));
        const int T = sh.k * sh.k;
        CK(hipMalloc(&w, (size_t)sh.Cout * Cin * 9 * 4)); CK(hipMalloc(&bias, sh.Cout * 4));
        CK(hipMalloc(&ga, (size_t)B * Cin * 4)); CK(hipMalloc(&gb, (size_t)B * Cin * 4));
        // statistics slab: entries = tiles per image * wave rows; the smallest tile here has 4 rows x 32 columns, WM <= 2
        const size_t stat_entries = (size_t)((sh.S + 3) / 4) * ((sh.S + 31) / 32) * 2;
        CK(hipMalloc(&stats, (size_t)B * stat_entries * sh.Cout * 2 * 4));
        CK(hipMalloc(&wpk, conv_packed_bytes_lab(sh.Cout, Cin, 9)));
        {   // random fill on the host (normal-ish via sum of uniforms)
            std::vector<float> h(std::max({n1, n2, (size_t)sh.Cout * Cin * 9}));
            auto fill = [&](float* d, size_t n, float scale) {
                for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() + rand() + rand() + rand()) / (float)RAND_MAX - 2.f) * 1.7f;
                CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
            };
            fill(x1, n1, 1.f); if (n2) fill(x2, n2, 1.f);
            fill(w, (size_t)sh.Cout * Cin * 9, 1.f / sqrtf(Cin * 9.f)); fill(bias, sh.Cout, 0.1f);
            fill(ga, (size_t)B * Cin, 0.1f); fill(gb, (size_t)B * Cin, 0.1f);
        }
        if (pack_conv_weights(w, sh.Cout, Cin, sh.k, sh.k, wpk, (LAB_PREC == 4 ? ENC_F16 : ENC_F16F4), 0) != 0) { printf("pack failed\n"); return 1; }
        ConvArgs a{};
        a.x1 = x1; a.C1 = sh.C1; a.x2 = x2; a.C2 = sh.C2; a.B = B; a.H = sh.S; a.W = sh.S; a.Ho = sh.S; a.Wo = sh.S; a.Cout = sh.Cout;
        a.wpk = wpk; a.bias = bias; a.out = out; a.stats = stats; a.stats_mode = STATS_SUMSQ; a.nk16 = Cin / 16; a.prec = LAB_PREC;
        if (sh.gn) { a.pro_a = ga; a.pro_b = gb; }
        const double flops = 2.0 * B * sh.S * sh.S * sh.Cout * Cin * T;
        std::vector<std::vector<Res>> all;
        for (int r = 0; r < rounds; ++r) {
            std::vector<Res> res;
            if (sh.k == 1) { if (r == 0) printf("%-14s 1x1 shape: no variant set enabled in this build (RUN1), skipped\n", sh.name); break; }
#define ALLV(WM, WN, MT, PRO, TPW, NAME)                                                          \
    RUNT(v0, WM, WN, MT, PRO, TPW, NAME " product");                                              \
    RUNT(a1, WM, WN, MT, PRO, TPW, NAME " no staging");                                           \
    RUNT(a4, WM, WN, MT, PRO, TPW, NAME " no weight reloads");                                    \
    RUNT(a5, WM, WN, MT, PRO, TPW, NAME " neither");                                              \
    RUNT(a7, WM, WN, MT, PRO, TPW, NAME " neither, no stores");                                     \
    RUNT(ns, WM, WN, MT, PRO, TPW, NAME " no scale-byte LDS read")
#ifdef LAB_TILES     // tile-shape comparison of the product kernel only (-DLAB_TILES): rows per wave 4 vs 8
#define TILES(PRO)                                                                                                     \
    if (sh.Cout < 128)      { RUNT(v0, 2, 2, 4, PRO, 4, "2x2 mt4 (product)"); RUNT(v0, 4, 2, 4, PRO, 2, "4x2 mt4 tpw2"); RUNT(v0, 4, 2, 4, PRO, 4, "4x2 mt4 tpw4"); }                                         \
    else if (sh.Cout < 256) { RUNT(v0, 2, 4, 4, PRO, 1, "2x4 mt4 (product)"); RUNT(v0, 1, 4, 8, PRO, 1, "1x4 mt8"); RUNT(v0, 1, 8, 8, PRO, 1, "1x8 mt8"); } \
    else                    { RUNT(v0, 1, 8, 4, PRO, 1, "1x8 mt4 (product)"); RUNT(v0, 1, 8, 8, PRO, 1, "1x8 mt8"); }
            if (sh.gn) { TILES(1) } else { TILES(0) }
#elif defined(LAB_PREFETCH)
#if LAB_PREC == 3   // (the rolling weight prefetch of f16f4 fixes the distance: only the staging policy varies)
#define PFV(WM, WN, MT, PRO, TPW) RUNT(v0, WM, WN, MT, PRO, TPW, "product"); RUNT(s0, WM, WN, MT, PRO, TPW, "staging at one point")
#else
#define PFV(WM, WN, MT, PRO, TPW) RUNT(v0, WM, WN, MT, PRO, TPW, "product"); RUNT(p0, WM, WN, MT, PRO, TPW, "weights just in time"); \
    RUNT(p2, WM, WN, MT, PRO, TPW, "weights two columns ahead"); RUNT(s0, WM, WN, MT, PRO, TPW, "staging at one point"); RUNT(p2s0, WM, WN, MT, PRO, TPW, "two ahead + one point")
#endif
            if (sh.Cout >= 256) { if (sh.gn) { PFV(1, 8, 8, 1, 1); } else { PFV(1, 8, 8, 0, 1); } }
            else if (sh.Cout >= 128) {
                if (sh.gn) { PFV(2, 4, 4, 1, 1); RUNT(v0, 2, 4, 8, 1, 1, "2x4x8 spread"); RUNT(s0, 2, 4, 8, 1, 1, "2x4x8 one point"); }
                else { PFV(2, 4, 4, 0, 1); RUNT(v0, 2, 4, 8, 0, 1, "2x4x8 spread"); RUNT(s0, 2, 4, 8, 0, 1, "2x4x8 one point"); }
            }
            else { if (sh.gn) { PFV(4, 2, 4, 1, 4); } else { PFV(4, 2, 4, 0, 4); } }
#elif defined(LAB_PHASE)   // two independent 4-wave workgroups per CU (2x2 tile), the one in the odd wave slot started late
            if (sh.Cout < 128) {
#define PH(PRO) RUNT(v0, 2, 2, 4, PRO, 4, "2x2 product"); RUNT(d1, 2, 2, 4, PRO, 4, "2x2 odd slot +1k cycles"); RUNT(d2, 2, 2, 4, PRO, 4, "2x2 odd slot +2k cycles"); RUNT(d3, 2, 2, 4, PRO, 4, "2x2 odd slot +4k cycles")
                if (sh.gn) { PH(1); } else { PH(0); }
            }
#else
            if (sh.Cout < 128) {
                if (sh.gn) { ALLV(2, 2, 4, 1, 4, "narrow"); } else { ALLV(2, 2, 4, 0, 4, "narrow"); }
            } else if (sh.Cout < 256) {
                if (sh.gn) { ALLV(2, 4, 4, 1, 1, "wide8w"); } else { ALLV(2, 4, 4, 0, 1, "wide8w"); }
            } else {
                if (sh.gn) { RUNT(v0, 1, 8, 4, 1, 1, "1x8 mt4"); RUNT(v0, 1, 8, 8, 1, 1, "1x8 mt8"); RUNT(v0, 1, 4, 8, 1, 1, "1x4 mt8 (1 wave/SIMD)"); }
                else       { RUNT(v0, 1, 8, 4, 0, 1, "1x8 mt4"); RUNT(v0, 1, 8, 8, 0, 1, "1x8 mt8"); RUNT(v0, 1, 4, 8, 0, 1, "1x4 mt8 (1 wave/SIMD)"); }
            }
#endif
            all.push_back(res);
        }
        for (auto& r : all)
            if (r.size() != all[0].size()) { printf("%s: rounds produced different variant counts\n", sh.name); return 4; }
        for (size_t i = 0; !all.empty() && i < all[0].size(); ++i) {
            std::vector<float> t;
            for (auto& r : all) t.push_back(r[i].ms);
            std::sort(t.begin(), t.end());
            printf("%-14s %-24s min %7.1f us  med %7.1f us  %7.1f TF/s\n", sh.name, all[0][i].label, t[0] * 1e3, t[t.size() / 2] * 1e3,
                   flops / (t[t.size() / 2] * 1e-3) / 1e12);
        }
        fflush(stdout);
        hipFree(x1); if (x2) hipFree(x2); hipFree(out); hipFree(w); hipFree(bias); hipFree(ga); hipFree(gb); hipFree(stats); hipFree(wpk);
    }
    return 0;
}
