// wgrad.hip -- weight gradients of every convolution on the UNet path (backward of
// models/unet_model.py:43,49,122,156,185,188,221,222,307,323):
//     dW[co][ci][ky][kx] = sum_{n,y,x} dY[n,y,x,co] * X[n, y*S+ky-P, x*S+kx-P, ci]
// as a GEMM whose REDUCTION index is the pixel.  On the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 the operand maps are A[i][k]: lane=(k<<5)|i and B[k][j]: lane=(k<<5)|j, so with
// i = output channel, j = input channel and k = pixel parity every lane simply reads ITS channel of one
// pixel: NHWC rows feed the MFMA without any transpose, and the result is exact fp32 (no hi/lo split).
// One workgroup = 4 waves = 64 co x 64 ci x (up to 9 taps); it walks a range of TR x 32 pixel tiles
// (dY tile + haloed X tile staged in LDS as fp32, X through the same GN+FiLM+SiLU / LayerNorm prologue
// as the forward conv), keeps the per-tap 32x32 accumulators in registers and writes ONE partial block;
// a second kernel sums the pixel splits in fixed order (deterministic, no atomics) into OIHW.
#include <string.h>

#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

struct WgradArgs {
    const float* dy; int Cout;                 // (B,Ho,Wo,Cout)
    const float* x1; int C1; const float* x2; int C2;   // conv input (virtual concat), (B,H,W,.)
    int B, H, W, Ho, Wo;
    const float* pro_a; const float* pro_b;    // PRO_GN: a,b [B][Cin] ; PRO_LN: g[Cin], stats (B*H*W,2)
    float* part;                               // [S][T][Cout][Cin]
    int S, tiles_x, tiles_y, tpw;              // pixel splits, tiles per image, tiles per workgroup
    int nci_blk;                               // ceil(Cin/64)
};

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// NTAP taps [tap0, tap0+NTAP) of a KH x KW kernel are handled per launch-z (4x4: two passes of 8).
template <int KH, int KW, int STRIDE, int PAD, int UPS, int TR, int NTAP, int PRO>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int T = KH * KW;
    constexpr int HR = (TR - 1) * STRIDE + KH, HC = 31 * STRIDE + KW;
    constexpr int NPX = TR * 32;                      // dY pixels per tile
    constexpr int NHP = HR * HC;                      // X halo pixels per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sdy = (float*)smem;                        // [NPX][64]
    float* sx = sdy + NPX * 64;                       // [NHP][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, kpar = lane >> 5;
    const int wc = wave >> 1, wi = wave & 1;
    const int Cin = a.C1 + a.C2;
    const int cob = blockIdx.y / a.nci_blk, cib = blockIdx.y - cob * a.nci_blk;
    const int co0 = cob * 64, ci0 = cib * 64;
    static_assert(NTAP % KW == 0, "a launch-z pass covers whole kernel rows");
    const int ky0 = blockIdx.z * (NTAP / KW);         // first kernel row of this pass (block-uniform)
    const int tap0 = ky0 * KW;
    const int tpi = a.tiles_x * a.tiles_y, ntile = a.B * tpi;
    const int t_beg = blockIdx.x * a.tpw, t_end = (t_beg + a.tpw < ntile) ? t_beg + a.tpw : ntile;
    const int Hin = UPS ? a.H * 2 : a.H, Win = UPS ? a.W * 2 : a.W;

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // Staging is software-pipelined: the global loads of tile t + 1 are issued into registers before the MFMAs of tile t and
    // committed to LDS (prologue, masks) after them.  (Loading and computing in turn left the kernel waiting for HBM every tile:
    // the 64 -> 384 qkv gradient took 1.6 ms at 256^2 where its fp32 MFMAs need 0.75 ms.)
    constexpr int NDI = (NPX * 16 + 255) / 256, NXI = (NHP * 16 + 255) / 256;     // float4 items per thread: dY / X
    float4 rdy[NDI], rxv[NXI];
    float2 rst[PRO == PRO_LN ? NXI : 1];
    bool okd[NDI], okx[NXI];
    auto issue = [&](int tl) __attribute__((always_inline)) {
        const int tc = tl < t_end ? tl : t_end - 1;               // past the end: an unused re-load (branch-free loop)
        const int n = tc / tpi;
        const int trem = tc - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int oy0 = ty * TR, ox0 = tx * 32;
        const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int i = tid + 256 * k;
            const int ic = i < NPX * 16 ? i : NPX * 16 - 1;
            const int px = ic >> 4, c4 = (ic & 15) * 4;
            const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
            okd[k] = oy < a.Ho && ox < a.Wo && co0 + c4 < a.Cout;
            const int oyc = oy < a.Ho ? oy : a.Ho - 1, oxc = ox < a.Wo ? ox : a.Wo - 1;
            const int cc = co0 + c4 < a.Cout ? co0 + c4 : 0;
            rdy[k] = *(const float4*)(a.dy + ((size_t)(n * a.Ho + oyc) * a.Wo + oxc) * a.Cout + cc);
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int i = tid + 256 * k;
            const int ic = i < NHP * 16 ? i : NHP * 16 - 1;
            const int hp = ic >> 4, c4 = (ic & 15) * 4;
            const int hy = hp / HC, hx = hp - hy * HC;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const int ch = ci0 + c4;
            okx[k] = iy >= 0 && iy < Hin && ix >= 0 && ix < Win && ch < Cin;
            int sy = UPS ? (iy >> 1) : iy, sxx = UPS ? (ix >> 1) : ix;
            sy = sy < 0 ? 0 : (sy >= a.H ? a.H - 1 : sy);
            sxx = sxx < 0 ? 0 : (sxx >= a.W ? a.W - 1 : sxx);
            const int chc = ch < Cin ? ch : 0;
            const size_t gp = (size_t)(n * a.H + sy) * a.W + sxx;
            const float* src = chc < a.C1 ? a.x1 + gp * a.C1 + chc : a.x2 + gp * a.C2 + (chc - a.C1);
            rxv[k] = *(const float4*)src;
            if (PRO == PRO_LN) rst[k] = *(const float2*)(a.pro_b + gp * 2);
        }
    };
    auto commit = [&](int tl) __attribute__((always_inline)) {
        const int n = tl / tpi;
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int i = tid + 256 * k;
            if (i < NPX * 16) {
                const int px = i >> 4, c4 = (i & 15) * 4;
                *(float4*)(sdy + px * 64 + c4) = okd[k] ? rdy[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int i = tid + 256 * k;
            if (i < NHP * 16) {
                const int hp = i >> 4, c4 = (i & 15) * 4;
                const int ch = ci0 + c4, chc = ch < Cin ? ch : 0;
                float4 v = rxv[k];
                if (PRO == PRO_GN) {
                    const float4 pa = *(const float4*)(a.pro_a + (size_t)n * Cin + chc);
                    const float4 pb = *(const float4*)(a.pro_b + (size_t)n * Cin + chc);
                    v.x = silu_f(fmaf(pa.x, v.x, pb.x)); v.y = silu_f(fmaf(pa.y, v.y, pb.y));
                    v.z = silu_f(fmaf(pa.z, v.z, pb.z)); v.w = silu_f(fmaf(pa.w, v.w, pb.w));
                } else if (PRO == PRO_LN) {
                    const float4 g = *(const float4*)(a.pro_a + chc);
                    const float2 st2 = rst[k];
                    v.x = g.x * (v.x * st2.x - st2.y); v.y = g.y * (v.y * st2.x - st2.y);
                    v.z = g.z * (v.z * st2.x - st2.y); v.w = g.w * (v.w * st2.x - st2.y);
                }
                if (!okx[k]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)(sx + hp * 64 + c4) = v;
            }
        }
    };
    if (t_beg < t_end) issue(t_beg);
    for (int tl = t_beg; tl < t_end; ++tl) {
        __syncthreads();                               // previous tile's MFMAs are done with the LDS
        commit(tl);
        __syncthreads();
        issue(tl + 1);                                 // in flight during this tile's MFMAs
        // ---- MFMA: for every pixel pair, A = dY[pixel][co], B = X[pixel shifted by tap][ci] ----
        const float* pa_ = sdy + wc * 32 + c;
        const float* pb_ = sx + wi * 32 + c + ky0 * HC * 64;
#pragma unroll
        for (int row = 0; row < TR; ++row) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int px = 2 * j + kpar;
                const float av = pa_[(row * 32 + px) * 64];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    constexpr int dummy = 0; (void)dummy;
                    const int kyr = t / KW, kx = t % KW;     // compile-time after unrolling
                    const float bv = pb_[((row * STRIDE + kyr) * HC + px * STRIDE + kx) * 64];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial block: part[split][tap][co][ci], coalesced over ci ----
    const int ci = ci0 + wi * 32 + c;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = tap0 + t;
        if (tap >= T) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + wc * 32 + acc_row(i, kpar);
            if (co < a.Cout && ci < Cin)
                a.part[(((size_t)blockIdx.x * T + tap) * a.Cout + co) * Cin + ci] = acc[t][i];
        }
    }
}

// dW[co][ci][tap] (+)= sum_s part[s][tap][co][ci]  (fixed order over s => deterministic).
// One block = 64 consecutive (tap,co,ci) entries x 16 split lanes.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int S, int T, int Cout, int Cin,
                                                            float* __restrict__ dw, int accumulate) {
    __shared__ float red[16][64];
    const long long per = (long long)T * Cout * Cin;
    const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + il;
    float s = 0.f;
    if (i < per)
        for (int k = sl; k < S; k += 16) s += part[(size_t)k * per + i];
    red[sl][il] = s;
    __syncthreads();
    if (sl == 0 && i < per) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][il];
        const int ci = (int)(i % Cin);
        const long long r = i / Cin;
        const int co = (int)(r % Cout), tap = (int)(r / Cout);
        float* o = dw + ((size_t)co * Cin + ci) * T + tap;
        *o = accumulate ? *o + t : t;
    }
}

template <int KH, int KW, int STRIDE, int PAD, int UPS, int TR, int NTAP, int PRO>
int launch_wgrad(WgradArgs a, hipStream_t st) {
    constexpr int HR = (TR - 1) * STRIDE + KH, HC = 31 * STRIDE + KW;
    const size_t lds = (size_t)(TR * 32 + HR * HC) * 64 * 4;
    auto kern = wgrad_kernel<KH, KW, STRIDE, PAD, UPS, TR, NTAP, PRO>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEDM_EHIP;
        attr = true;
    }
    const int Cin = a.C1 + a.C2;
    a.nci_blk = (Cin + 63) / 64;
    dim3 grid(a.S, ((a.Cout + 63) / 64) * a.nci_blk, (KH * KW + NTAP - 1) / NTAP);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return LAUNCH_OK();
}

}  // namespace

// Geometry shared by the launcher and the workspace planner.
static void wgrad_geometry(int k, int stride, int Cout, int Cin, int B, int Ho, int Wo, int* TR, int* tiles_x, int* tiles_y,
                           int* S, int* tpw) {
    *TR = stride == 2 ? 1 : (k == 1 ? 1 : 2);        // 1x1: 32-pixel tiles = 16 KB of LDS, 8 workgroups per CU hide the staging latency (TR 2: +3.5 ms per step)
    *tiles_x = (Wo + 31) / 32;
    *tiles_y = (Ho + *TR - 1) / *TR;
    const long long ntile = (long long)B * *tiles_x * *tiles_y;
    const long long blocks = (long long)((Cout + 63) / 64) * ((Cin + 63) / 64);
    long long s = (1536 + blocks - 1) / blocks;          // aim at ~1.5k workgroups
    if (s > ntile) s = ntile;
    if (s < 1) s = 1;
    const long long t = (ntile + s - 1) / s;
    *tpw = (int)t;
    *S = (int)((ntile + t - 1) / t);
}

// tedm_set_tuning("wgrad_mfma", v): 0 = exact-fp32 MFMA kernel everywhere, 1 = bf16x3 matrix-core path for 3x3 (default;
// 1x1 weight gradients are staging-bound and measured cheaper without the planar pre-pass)
static int g_wgrad_mfma = 1;
// tedm_set_tuning("wgrad_fp16", v): 1 (default) = callers that ask for the fast arithmetic (the f16f4 training mode) get ONE fp16 product
// with a per-tensor power-of-two scale on the 3x3 weight gradients (wgrad_mfma.hip); 0 = bf16x3 for them too
static int g_wgrad_fp16 = 1;
int wgrad_set_tuning(const char* key, int v) {
    if (!strcmp(key, "wgrad_mfma") && (v == 0 || v == 1)) { g_wgrad_mfma = v; return TEDM_OK; }
    if (!strcmp(key, "wgrad_fp16") && (v == 0 || v == 1)) { g_wgrad_fp16 = v; return TEDM_OK; }
    return wgrad_mfma_set_tuning(key, v);
}

size_t wgrad_partial_floats(int k, int stride, int Cout, int Cin, int B, int Ho, int Wo) {
    int TR, tx, ty, S, tpw;
    wgrad_geometry(k, stride, Cout, Cin, B, Ho, Wo, &TR, &tx, &ty, &S, &tpw);
    size_t n = (size_t)S * k * k * Cout * Cin;
    if (stride == 1 && k == 3) {                                   // sized for either path (the switch is a runtime tuning key)
        const size_t m = wgrad_mfma_floats(k, Cout, Cin, B, Ho, Wo);
        if (m > n) n = m;
    }
    return n;
}

// dw (OIHW, Cout x (C1+C2) x k x k) = [accumulate ? dw : 0] + wgrad.  ups: 1 = the conv saw the nearest-x2 upsampled input.
int op_conv_wgrad(const float* dy, int Cout, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Ho,
                  int Wo, int k, int stride, int ups, int pro, const float* pro_a, const float* pro_b, float* part,
                  float* dw, int accumulate, int fast, const unsigned* dy_amax, hipStream_t st) {
    const int Cin = C1 + C2;
    if (Cin % 4 || C1 % 4 || Cout % 4) return TEDM_EINVAL;
    WgradArgs a{};
    a.dy = dy; a.Cout = Cout; a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.pro_a = pro_a; a.pro_b = pro_b; a.part = part;
    if (stride == 1 && k == 3 && g_wgrad_mfma && (pro == PRO_NONE || pro == PRO_GN)) {
        int S = 0;
        const int rc = op_conv_wgrad_mfma(dy, Cout, x1, C1, x2, C2, B, H, W, Ho, Wo, k, ups, pro, pro_a, pro_b, part, &S, fast && g_wgrad_fp16, dy_amax, st);
        if (rc != TEDM_OK) return rc;
        const long long per = (long long)k * k * Cout * Cin;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(1024), 0, st, part, S, k * k, Cout, Cin, dw,
                           accumulate);
        return LAUNCH_OK();
    }
    int TR;
    wgrad_geometry(k, stride, Cout, Cin, B, Ho, Wo, &TR, &a.tiles_x, &a.tiles_y, &a.S, &a.tpw);
    int rc = TEDM_EINVAL;
    if (k == 3 && stride == 1 && !ups && pro == PRO_NONE) rc = launch_wgrad<3, 3, 1, 1, 0, 2, 9, PRO_NONE>(a, st);
    else if (k == 3 && stride == 1 && !ups && pro == PRO_GN) rc = launch_wgrad<3, 3, 1, 1, 0, 2, 9, PRO_GN>(a, st);
    else if (k == 3 && stride == 1 && ups == 1 && pro == PRO_NONE) rc = launch_wgrad<3, 3, 1, 1, 1, 2, 9, PRO_NONE>(a, st);
    else if (k == 4 && stride == 2 && !ups && pro == PRO_NONE) rc = launch_wgrad<4, 4, 2, 1, 0, 1, 8, PRO_NONE>(a, st);
    else if (k == 1 && stride == 1 && !ups && pro == PRO_NONE) rc = launch_wgrad<1, 1, 1, 0, 0, 1, 1, PRO_NONE>(a, st);
    else if (k == 1 && stride == 1 && !ups && pro == PRO_LN) rc = launch_wgrad<1, 1, 1, 0, 0, 1, 1, PRO_LN>(a, st);
    if (rc != TEDM_OK) return rc;
    const long long per = (long long)k * k * Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(1024), 0, st, part, a.S, k * k, Cout, Cin, dw,
                       accumulate);
    return LAUNCH_OK();
}
