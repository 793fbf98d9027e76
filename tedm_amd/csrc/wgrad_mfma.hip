// wgrad_mfma.hip -- weight gradients of the stride-1 3x3 and 1x1 convolutions on the bf16 matrix cores
// with the forward path's bf16x3 operand split (hi*hi + lo*hi + hi*lo, fp32 accumulate):
//     dW[co][ci][ky][kx] = sum_{n,y,x} dY[n,y,x,co] * X[n, y+ky-1, x+kx-1, ci]
// The reduction index of this GEMM is the PIXEL, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive
// reduction elements per lane -- 8 consecutive pixels of ONE channel, which NHWC does not provide.  So the
// work is two kernels:
//   1. planar_split_kernel: fp32 NHWC (through the forward conv's GN+FiLM+SiLU / LayerNorm prologue and
//      the optional nearest-x2 upsample) -> bf16 hi/lo planes [n][c][Hp][Wp], zero padded, pixel-contiguous.
//      The split and the prologue are paid ONCE per element instead of once per consuming workgroup.
//      X planes are written shifted by one pixel (px at element px+9) so that the three kx taps of a
//      16-pixel MFMA step are the aligned 16-byte chunk, a 16-bit funnel shift of it, and its dwords 1..4.
//   2. wgrad_mfma_kernel: workgroup = 64 co x 64 ci (4 waves of 32x32) x all taps; it walks a range of
//      2-row x 32-pixel tiles, copies the plane chunks to LDS (pure 16-byte copies, no VALU), keeps the
//      9 per-tap accumulators in registers and writes one partial block; wgrad.hip's reduce kernel sums the
//      pixel splits in fixed order (deterministic, no atomics).
// LDS channel strides are odd multiples of 16 B => the ds_read_b128 operand fetches are conflict-free.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

constexpr int TR = 2;     // output rows per tile

struct PlanarArgs {
    const float* x1; int C1; const float* x2; int C2;
    int B, H, W;             // source size
    int Hout, Wout;          // plane data size (= H,W or 2H,2W)
    int ups, lp;             // nearest-x2 upsample; left pad (element index of pixel 0)
    const float* pro_a; const float* pro_b;
    uint16_t* hi; uint16_t* lo;
    int Hp, Wp, ncb;         // padded plane size, channel blocks of 64
};

TEDM_DEVFN uint32_t bf16_bits(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v); }

// grid: (ceil(Wp/64), Hp, B*ncb).  One block = 64 plane elements of one row x 64 channels.
template <int PRO>
__global__ __launch_bounds__(256) void planar_split_kernel(PlanarArgs a) {
    constexpr int RS = 33;                            // dwords per channel row (32 pixel pairs + 1)
    __shared__ uint32_t th[64 * RS], tl[64 * RS];
    const int tid = threadIdx.x;
    const int C = a.C1 + a.C2;
    const int n = blockIdx.z / a.ncb, cb = blockIdx.z - n * a.ncb;
    const int yy = blockIdx.y, y = yy - 1;
    const int E0 = blockIdx.x * 64;
    const int c4 = (tid & 15) * 4, pp = tid >> 4;
    const int ch = cb * 64 + c4;
    const bool row_ok = y >= 0 && y < a.Hout && ch < C;
    int sy = a.ups ? (y >> 1) : y;
    sy = sy < 0 ? 0 : (sy >= a.H ? a.H - 1 : sy);
    const int chc = ch < C ? ch : 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pair = pp + 16 * it;
        float4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int px = E0 + 2 * pair + q - a.lp;
            const bool ok = row_ok && px >= 0 && px < a.Wout;
            int sx = a.ups ? (px >> 1) : px;
            sx = sx < 0 ? 0 : (sx >= a.W ? a.W - 1 : sx);
            const size_t gp = (size_t)(n * a.H + sy) * a.W + sx;
            const float* src = chc < a.C1 ? a.x1 + gp * a.C1 + chc : a.x2 + gp * a.C2 + (chc - a.C1);
            float4 t = *(const float4*)src;                       // unconditional load, clamped address
            if (PRO == PRO_GN) {
                const float4 pa = *(const float4*)(a.pro_a + (size_t)n * C + chc);
                const float4 pb = *(const float4*)(a.pro_b + (size_t)n * C + chc);
                t.x = silu_f(fmaf(pa.x, t.x, pb.x)); t.y = silu_f(fmaf(pa.y, t.y, pb.y));
                t.z = silu_f(fmaf(pa.z, t.z, pb.z)); t.w = silu_f(fmaf(pa.w, t.w, pb.w));
            } else if (PRO == PRO_LN) {
                const float4 g = *(const float4*)(a.pro_a + chc);
                const float2 s2 = *(const float2*)(a.pro_b + gp * 2);
                t.x = g.x * (t.x * s2.x - s2.y); t.y = g.y * (t.y * s2.x - s2.y);
                t.z = g.z * (t.z * s2.x - s2.y); t.w = g.w * (t.w * s2.x - s2.y);
            }
            if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
            v[q] = t;
        }
        const float e0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, e1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const __bf16 h0 = (__bf16)e0[k], h1 = (__bf16)e1[k];
            const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
            const uint32_t lb = bf16_bits(e0[k] - (float)h0) | (bf16_bits(e1[k] - (float)h1) << 16);
            th[(c4 + k) * RS + pair] = hb;
            tl[(c4 + k) * RS + pair] = lb;
        }
    }
    __syncthreads();
    const int c = tid >> 2;
    const int chn = cb * 64 + c;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int chunk = (tid & 3) + 4 * it;
        const int e = E0 + chunk * 8;
        if (chn < C && e < a.Wp) {
            const size_t off = (((size_t)n * C + chn) * a.Hp + yy) * a.Wp + e;
            const uint32_t* ph = th + c * RS + chunk * 4;
            const uint32_t* pl = tl + c * RS + chunk * 4;
            *(uint4*)(a.hi + off) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            *(uint4*)(a.lo + off) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
    }
}

struct WgArgs {
    const uint16_t *dyh, *dyl, *xh, *xl;    // planes [n][C][Hp][Wp]
    int Cout, Cin, B, Hp, Wp;
    int tiles_x, tiles_y;
    float* part;                            // [S][T][Cout][Cin]
    int S, tpw, nci_blk, nblk;
};

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
TEDM_DEVFN bf16x8 as_bf(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

// 1-D grid of 8 * ceil(S/8) * nblk workgroups; id -> (split, channel-block pair) so that the nblk workgroups
// that share one pixel range run on the same XCD (its L2 then serves the re-reads of the X / dY slices).
template <int NTAP>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgArgs a) {
    constexpr int HR = NTAP == 9 ? TR + 2 : TR;       // X rows per tile
    constexpr int XCH = NTAP == 9 ? 5 : 4;            // 16-byte chunks per X row (40 / 32 pixels)
    constexpr int XS = NTAP == 9 ? 336 : 144;         // bytes per X channel in LDS (odd multiple of 16)
    constexpr int DS = 144;                           // bytes per dY channel in LDS (2 rows x 64 B + 16)
    static_assert(HR * XCH * 16 <= XS && TR * 64 <= DS, "LDS row layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sdyh = smem;
    char* sdyl = sdyh + 64 * DS;
    char* sxh = sdyl + 64 * DS;
    char* sxl = sxh + 64 * XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wc = wave >> 1, wi = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / a.nblk), blk = slot % a.nblk;
    if (split >= a.S) return;                         // whole workgroup leaves before any barrier
    const int cob = blk / a.nci_blk, cib = blk - cob * a.nci_blk;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int tpi = a.tiles_x * a.tiles_y, ntile = a.B * tpi;
    const int t_beg = split * a.tpw, t_end = (t_beg + a.tpw < ntile) ? t_beg + a.tpw : ntile;
    const size_t plane_row = (size_t)a.Wp, plane_ch = (size_t)a.Hp * a.Wp;

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int tl_ = t_beg; tl_ < t_end; ++tl_) {
        const int n = tl_ / tpi;
        const int trem = tl_ - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int e0 = 8 + tx * 32;                   // first plane element of the tile's chunks (dY: px ox0, X 3x3: px ox0-1)
        __syncthreads();                               // previous tile's operand reads are done
        // ---- dY: 64 co x TR rows x 4 chunks, both planes ----
#pragma unroll
        for (int it = 0; it < 64 * TR * 4 / 256; ++it) {
            const int i = tid + it * 256;
            const int c = i / (TR * 4), rem = i - c * (TR * 4), row = rem >> 2, q = rem & 3;
            const int ch = co0 + c;
            const int chc = ch < a.Cout ? ch : a.Cout - 1;
            const size_t off = ((size_t)n * a.Cout + chc) * plane_ch + (size_t)(1 + ty * TR + row) * plane_row + e0 + q * 8;
            uint4 vh = *(const uint4*)(a.dyh + off), vl = *(const uint4*)(a.dyl + off);
            if (ch >= a.Cout) { vh = make_uint4(0, 0, 0, 0); vl = vh; }
            *(uint4*)(sdyh + c * DS + row * 64 + q * 16) = vh;
            *(uint4*)(sdyl + c * DS + row * 64 + q * 16) = vl;
        }
        // ---- X: 64 ci x HR rows x XCH chunks, both planes ----
        constexpr int NX = 64 * HR * XCH;
#pragma unroll
        for (int it = 0; it < (NX + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int ic = i < NX ? i : NX - 1;        // tail threads repeat the last item (same value, same address)
            const int c = ic / (HR * XCH), rem = ic - c * (HR * XCH), row = rem / XCH, q = rem - row * XCH;
            const int ch = ci0 + c;
            const int chc = ch < a.Cin ? ch : a.Cin - 1;
            const int prow = (NTAP == 9 ? 0 : 1) + ty * TR + row;
            const size_t off = ((size_t)n * a.Cin + chc) * plane_ch + (size_t)prow * plane_row + e0 + q * 8;
            uint4 vh = *(const uint4*)(a.xh + off), vl = *(const uint4*)(a.xl + off);
            if (ch >= a.Cin) { vh = make_uint4(0, 0, 0, 0); vl = vh; }
            *(uint4*)(sxh + c * XS + row * (XCH * 16) + q * 16) = vh;
            *(uint4*)(sxl + c * XS + row * (XCH * 16) + q * 16) = vl;
        }
        __syncthreads();
        // ---- MFMA: K step = 16 pixels of one row (column group cg); lane (r,h) holds pixels 8h..8h+7 of channel r ----
        const char* pdyh = sdyh + (wc * 32 + r) * DS + h * 16;
        const char* pdyl = sdyl + (wc * 32 + r) * DS + h * 16;
        const char* pxh = sxh + (wi * 32 + r) * XS + h * 16;
        const char* pxl = sxl + (wi * 32 + r) * XS + h * 16;
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            bf16x8 ah[TR], al[TR];
#pragma unroll
            for (int row = 0; row < TR; ++row) {
                ah[row] = as_bf(*(const uint4*)(pdyh + row * 64 + cg * 32));
                al[row] = as_bf(*(const uint4*)(pdyl + row * 64 + cg * 32));
            }
            if (NTAP == 1) {
#pragma unroll
                for (int row = 0; row < TR; ++row) {
                    const bf16x8 bh = as_bf(*(const uint4*)(pxh + row * 64 + cg * 32));
                    const bf16x8 bl = as_bf(*(const uint4*)(pxl + row * 64 + cg * 32));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bh, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[row], bh, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bl, acc[0], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int hy = 0; hy < HR; ++hy) {
                    // 16-pixel window starting at element 0 of chunk (2cg+h): elements 0..7 (kx=0), 1..8 (kx=1), 2..9 (kx=2)
                    const uint4 qh = *(const uint4*)(pxh + hy * (XCH * 16) + cg * 32);
                    const uint4 ql = *(const uint4*)(pxl + hy * (XCH * 16) + cg * 32);
                    const uint32_t nh = *(const uint32_t*)(pxh + hy * (XCH * 16) + cg * 32 + 16);
                    const uint32_t nl = *(const uint32_t*)(pxl + hy * (XCH * 16) + cg * 32 + 16);
                    bf16x8 bh[3], bl[3];
                    bh[0] = as_bf(qh);
                    bl[0] = as_bf(ql);
                    bh[1] = as_bf(make_uint4(__builtin_amdgcn_alignbit(qh.y, qh.x, 16), __builtin_amdgcn_alignbit(qh.z, qh.y, 16),
                                             __builtin_amdgcn_alignbit(qh.w, qh.z, 16), __builtin_amdgcn_alignbit(nh, qh.w, 16)));
                    bl[1] = as_bf(make_uint4(__builtin_amdgcn_alignbit(ql.y, ql.x, 16), __builtin_amdgcn_alignbit(ql.z, ql.y, 16),
                                             __builtin_amdgcn_alignbit(ql.w, ql.z, 16), __builtin_amdgcn_alignbit(nl, ql.w, 16)));
                    bh[2] = as_bf(make_uint4(qh.y, qh.z, qh.w, nh));
                    bl[2] = as_bf(make_uint4(ql.y, ql.z, ql.w, nl));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int row = hy - ky;         // X row hy pairs with dY row hy-ky for kernel row ky
                        if (row < 0 || row >= TR) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            f32x16& d = acc[ky * 3 + kx];
                            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bh[kx], d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[row], bh[kx], d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bl[kx], d, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- partial block: part[split][tap][co][ci], coalesced over ci ----
    const int ci = ci0 + wi * 32 + r;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + wc * 32 + acc_row(i, h);
            if (co < a.Cout && ci < a.Cin)
                a.part[(((size_t)split * NTAP + t) * a.Cout + co) * a.Cin + ci] = acc[t][i];
        }
    }
}

struct Geo { int tiles_x, tiles_y, Hp, Wp, S, tpw, nco, nci; };
Geo geometry(int Cout, int Cin, int B, int Ho, int Wo) {
    Geo g;
    g.tiles_x = (Wo + 31) / 32;
    g.tiles_y = (Ho + TR - 1) / TR;
    g.Hp = g.tiles_y * TR + 2;
    g.Wp = g.tiles_x * 32 + 16;
    g.nco = (Cout + 63) / 64;
    g.nci = (Cin + 63) / 64;
    const long long ntile = (long long)B * g.tiles_x * g.tiles_y;
    long long s = 1024 / ((long long)g.nco * g.nci);     // ~1k workgroups (2 resident per CU x 256 CUs x 2 rounds)
    if (s > ntile) s = ntile;
    if (s < 1) s = 1;
    const long long t = (ntile + s - 1) / s;
    g.tpw = (int)t;
    g.S = (int)((ntile + t - 1) / t);
    return g;
}
size_t plane_elems(const Geo& g, int B, int C) { return (size_t)B * C * g.Hp * g.Wp; }

int launch_planar(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Hout, int Wout, int ups, int lp, int pro,
                  const float* pa, const float* pb, uint16_t* hi, uint16_t* lo, const Geo& g, hipStream_t st) {
    PlanarArgs a{};
    a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Hout = Hout; a.Wout = Wout; a.ups = ups; a.lp = lp;
    a.pro_a = pa; a.pro_b = pb; a.hi = hi; a.lo = lo; a.Hp = g.Hp; a.Wp = g.Wp; a.ncb = (C1 + C2 + 63) / 64;
    dim3 grid((g.Wp + 63) / 64, g.Hp, B * a.ncb);
    if (pro == PRO_NONE) hipLaunchKernelGGL(planar_split_kernel<PRO_NONE>, grid, dim3(256), 0, st, a);
    else if (pro == PRO_GN) hipLaunchKernelGGL(planar_split_kernel<PRO_GN>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(planar_split_kernel<PRO_LN>, grid, dim3(256), 0, st, a);
    return LAUNCH_OK();
}

template <int NTAP>
int launch_mfma(WgArgs a, hipStream_t st) {
    constexpr int XS = NTAP == 9 ? 336 : 144;
    const size_t lds = (size_t)64 * (2 * 144 + 2 * XS);
    auto kern = wgrad_mfma_kernel<NTAP>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEDM_EHIP;
        attr = true;
    }
    const int grid = 8 * ((a.S + 7) / 8) * a.nblk;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return LAUNCH_OK();
}

}  // namespace

// Workspace (in floats) of the MFMA path: partials + four bf16 planes.
size_t wgrad_mfma_floats(int k, int Cout, int Cin, int B, int Ho, int Wo) {
    const Geo g = geometry(Cout, Cin, B, Ho, Wo);
    const size_t part = (size_t)g.S * k * k * Cout * Cin;
    const size_t planes = (plane_elems(g, B, Cout) + plane_elems(g, B, Cin)) * 2;    // hi + lo, 2 bytes each
    return ((part + 63) & ~(size_t)63) + (planes * 2 + 3) / 4 + 64;
}

// k in {1,3}, stride 1.  ups: the conv read the nearest-x2 upsampled input (H,W are the SOURCE size, Ho=2H).
// Fills part[S][k*k][Cout][Cin] and returns S in *S_out; the caller runs the split reduction.
int op_conv_wgrad_mfma(const float* dy, int Cout, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Ho, int Wo,
                       int k, int ups, int pro, const float* pro_a, const float* pro_b, float* ws, int* S_out, hipStream_t st) {
    const int Cin = C1 + C2;
    if ((k != 1 && k != 3) || Cin % 4 || C1 % 4 || Cout % 4) return TEDM_EINVAL;
    const Geo g = geometry(Cout, Cin, B, Ho, Wo);
    const size_t part = ((size_t)g.S * k * k * Cout * Cin + 63) & ~(size_t)63;
    uint16_t* dyh = (uint16_t*)(ws + part);
    uint16_t* dyl = dyh + plane_elems(g, B, Cout);
    uint16_t* xh = dyl + plane_elems(g, B, Cout);
    uint16_t* xl = xh + plane_elems(g, B, Cin);
    if (plane_elems(g, B, Cout) % 8 || plane_elems(g, B, Cin) % 8) return TEDM_EINVAL;   // 16-byte plane alignment
    int rc = launch_planar(dy, Cout, nullptr, 0, B, Ho, Wo, Ho, Wo, 0, 8, PRO_NONE, nullptr, nullptr, dyh, dyl, g, st);
    if (rc != TEDM_OK) return rc;
    rc = launch_planar(x1, C1, x2, C2, B, H, W, Ho, Wo, ups, k == 3 ? 9 : 8, pro, pro_a, pro_b, xh, xl, g, st);
    if (rc != TEDM_OK) return rc;
    WgArgs a{};
    a.dyh = dyh; a.dyl = dyl; a.xh = xh; a.xl = xl; a.Cout = Cout; a.Cin = Cin; a.B = B; a.Hp = g.Hp; a.Wp = g.Wp;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.part = ws; a.S = g.S; a.tpw = g.tpw; a.nci_blk = g.nci; a.nblk = g.nco * g.nci;
    *S_out = g.S;
    return k == 3 ? launch_mfma<9>(a, st) : launch_mfma<1>(a, st);
}
