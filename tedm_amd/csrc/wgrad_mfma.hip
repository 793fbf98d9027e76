// wgrad_mfma.hip -- weight gradients of the stride-1 3x3 convolutions on the bf16 matrix cores with the
// forward path's bf16x3 operand split (hi*hi + lo*hi + hi*lo, fp32 accumulate):
//     dW[co][ci][ky][kx] = sum_{n,y,x} dY[n,y,x,co] * X[n, y+ky-1, x+kx-1, ci]
// The reduction index of this GEMM is the PIXEL, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction
// elements per lane -- 8 consecutive pixels of ONE channel, which NHWC does not provide.  Two kernels:
//   1. planar_split_kernel: fp32 NHWC (through the forward conv's GN+FiLM+SiLU prologue and the optional
//      nearest-x2 upsample) -> bf16 hi / lo "tile planes"
//          plane[n][prow][tx][c][PW]      prow = image row + 1 (zero rows above / below), tx = 32-pixel column tile,
//      PW = 32 pixels for dY (pixel ox0+j) and 40 for X (pixel ox0-1+j: one halo pixel left, the rest right), zero
//      outside the image.  The split and the prologue are paid ONCE per element, and everything one workgroup of the
//      MFMA kernel stages for a tile row (64 channels x PW pixels) is ONE contiguous 4 / 5 KB run.
//      X starts one pixel early so that the three kx taps of a 16-pixel MFMA step are the aligned 16-byte chunk, a
//      16-bit funnel shift of it, and its dwords 1..4.
//   2. wgrad_mfma8_kernel: workgroup = 8 waves = 64 co x 64 ci x 9 taps; it walks a range of 2-row x 32-pixel
//      tiles; waves 0-3 take the left 16-pixel column group, waves 4-7 the right one (each a 32x32 block with the 9
//      tap accumulators in registers).  Software pipelined: the plane chunks of tile t+1 are loaded into registers
//      before the MFMAs of tile t and written to LDS after them.  One partial per (split, column group); wgrad.hip's
//      reduce kernel sums them in fixed order (deterministic, no atomics).
// LDS channel strides are odd multiples of 16 B => the ds_read_b128 operand fetches are conflict-free.
//
// F16 instances (the f16f4 training mode): ONE fp16 product instead of three bf16 ones.  The reduction runs over ~10^5..10^6 pixels
// whose operand roundings are independent, so an 11-bit significand costs ~4e-4 relative on a gradient tensor (inside that mode's
// 3e-3 bound) -- but dY has no fixed range (1e-7 and smaller right behind an L1 loss with mean reduction), so it is scaled by a
// power of two taken from its own absolute maximum (absmax_kernel: an integer atomic maximum of the bit patterns, order independent
// and hence deterministic; the maximum lands in [2^12, 2^13)) and the partial sums are scaled back exactly.  X is the forward
// pass's conv input, for which the f16f4 mode already requires |x| < 65504; the conversion saturates.  Half the plane bytes, a third
// of the MFMAs; the plane loads then run TWO tiles ahead (two register sets) to keep as many bytes in flight.
#include <string.h>

#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

constexpr int TR = 2;     // output rows per tile

struct PlanarArgs {
    const float* x1; int C1; const float* x2; int C2;
    int B, H, W;             // source size
    int Hout, Wout;          // image size the planes describe (= H,W or 2H,2W)
    int ups;                 // nearest-x2 upsample
    const float* pro_a; const float* pro_b;
    uint16_t* hi; uint16_t* lo;
    int Hp, tiles_x, ncb;    // plane rows, column tiles, channel blocks of 64
    const unsigned* amax;    // F16: bit pattern of max |value| of the tensor (null: scale 1)
};

// power-of-two scale that puts a tensor's absolute maximum (bit pattern `u`) into [2^12, 2^13); 1 for an all-zero tensor
TEDM_DEVFN float f16_plane_scale(unsigned u, bool inverse) {
    if (u == 0u) return 1.f;
    int e = (int)(u >> 23) - 127;                       // floor(log2 max)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const int k = inverse ? e - 12 : 12 - e;
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

// max |x| over n4 float4s as a bit pattern (non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ x, long long n4, unsigned* __restrict__ out) {
    unsigned m = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x[i];
        m = max(max(m, __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu), __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu);
        m = max(max(m, __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu), __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    // one atomic per workgroup, and only when it can still raise the value (same-address atomics serialise in L2)
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}

TEDM_DEVFN uint32_t bf16_bits(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v); }

// grid: (tiles_x, Hp, B*ncb).  One block = one plane row of one column tile x 64 channels (PW pixels each).
// HALO 1: X planes (40 pixels from ox0-1), 0: dY planes (32 pixels from ox0).
template <int PRO, int HALO, bool F16 = false>
__global__ __launch_bounds__(256) void planar_split_kernel(PlanarArgs a) {
    constexpr int PW = HALO ? 40 : 32, NPAIR = PW / 2, RS = NPAIR + 1;      // RS odd: conflict-light transposed access
    constexpr int NCHUNK = PW / 8;
    __shared__ uint32_t th[64 * RS], tl[F16 ? 1 : 64 * RS];
    const int tid = threadIdx.x;
    const int C = a.C1 + a.C2;
    const float fs = (F16 && a.amax) ? f16_plane_scale(*a.amax, false) : 1.f;
    const int n = blockIdx.z / a.ncb, cb = blockIdx.z - n * a.ncb;
    const int yy = blockIdx.y, y = yy - 1, tx = blockIdx.x;
    const int c4 = (tid & 15) * 4, pp = tid >> 4;
    const int ch = cb * 64 + c4;
    const bool row_ok = y >= 0 && y < a.Hout && ch < C;
    int sy = a.ups ? (y >> 1) : y;
    sy = sy < 0 ? 0 : (sy >= a.H ? a.H - 1 : sy);
    const int chc = ch < C ? ch : 0;
#pragma unroll
    for (int it = 0; it < (NPAIR + 15) / 16; ++it) {
        const int pair = pp + 16 * it;
        float4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int px = tx * 32 - HALO + 2 * pair + q;
            const bool ok = row_ok && px >= 0 && px < a.Wout && pair < NPAIR;
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
            }
            if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
            v[q] = t;
        }
        if (pair < NPAIR) {
            const float e0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, e1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (F16) {
                    const _Float16 h0 = (_Float16)fminf(fmaxf(e0[k] * fs, -65504.f), 65504.f), h1 = (_Float16)fminf(fmaxf(e1[k] * fs, -65504.f), 65504.f);
                    th[(c4 + k) * RS + pair] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                } else {
                    const __bf16 h0 = (__bf16)e0[k], h1 = (__bf16)e1[k];
                    const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                    const uint32_t lb = bf16_bits(e0[k] - (float)h0) | (bf16_bits(e1[k] - (float)h1) << 16);
                    th[(c4 + k) * RS + pair] = hb;
                    tl[(c4 + k) * RS + pair] = lb;
                }
            }
        }
    }
    __syncthreads();
    // the block's output is ONE contiguous run per plane: [64 channels][PW pixels]
    const size_t base = ((((size_t)n * a.Hp + yy) * a.tiles_x + tx) * C + (size_t)cb * 64) * PW;
    for (int i = tid; i < 64 * NCHUNK; i += 256) {
        const int c = i / NCHUNK, chunk = i - c * NCHUNK;
        if (cb * 64 + c < C) {
            const uint32_t* ph = th + c * RS + chunk * 4;
            *(uint4*)(a.hi + base + (size_t)i * 8) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            if constexpr (!F16) {
                const uint32_t* pl = tl + c * RS + chunk * 4;
                *(uint4*)(a.lo + base + (size_t)i * 8) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
        }
    }
}

struct WgArgs {
    const uint16_t *dyh, *dyl, *xh, *xl;    // tile planes
    int Cout, Cin, B, Hp;
    int tiles_x, tiles_y;
    float* part;                            // [2*S][9][Cout][Cin]
    int S, tpw, nci_blk, nblk;
};

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
TEDM_DEVFN bf16x8 as_bf(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

// 1-D grid of 8 * ceil(S/8) * nblk workgroups, one per CU; id -> (split, channel-block pair) so that the nblk
// workgroups that share one pixel range run on the same XCD (its L2 then serves the re-reads of the X / dY slices).
__global__ __launch_bounds__(512, 1) void wgrad_mfma8_kernel(WgArgs a) {
    constexpr int HR = TR + 2, XCH = 5, XS = 336, DS = 144;      // X: 4 rows x 80 B (+16) per channel; dY: 2 rows x 64 B (+16)
    constexpr int NDY = 64 * TR * 4, NX = 64 * HR * XCH;         // 16-byte chunks per plane: 512 / 1280
    static_assert(NDY == 512 && NX == 1280, "staging is written out for 512 threads: 1 dY + 3 X chunks each");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sdyh = smem;
    char* sdyl = sdyh + 64 * DS;
    char* sxh = sdyl + 64 * DS;
    char* sxl = sxh + 64 * XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cg = wave >> 2, wc = (wave >> 1) & 1, wi = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / a.nblk), blk = slot % a.nblk;
    if (split >= a.S) return;                         // whole workgroup leaves before any barrier
    const int cob = blk / a.nci_blk, cib = blk - cob * a.nci_blk;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int tpi = a.tiles_x * a.tiles_y, ntile = a.B * tpi;
    const int t_beg = split * a.tpw, t_end = (t_beg + a.tpw < ntile) ? t_beg + a.tpw : ntile;
    const unsigned dy_row = (unsigned)a.tiles_x * a.Cout * 32, x_row = (unsigned)a.tiles_x * a.Cin * 40;   // plane row strides (elements)

    // per-thread staging items (fixed over tiles) as 32-bit plane offsets relative to the tile + LDS offsets.  Item i of
    // a plane: row = i / (64*chunks), then channel-major inside the row's contiguous [64 ch][PW] run (the third X item
    // of the upper half of the threads repeats the last chunk: same value to the same address).
    auto dy_item = [&](int i, unsigned& g, int& l, uint32_t& m) {
        const int row = i >> 8, rem = i & 255, c = rem >> 2, q = rem & 3;
        const int ch = co0 + c;
        m = ch < a.Cout ? ~0u : 0u;
        g = (unsigned)(1 + row) * dy_row + (unsigned)(ch < a.Cout ? ch : a.Cout - 1) * 32 + q * 8;
        l = c * DS + row * 64 + q * 16;
    };
    auto x_item = [&](int i, unsigned& g, int& l, uint32_t& m) {
        const int ic = i < NX ? i : NX - 1;
        const int row = ic / 320, rem = ic - row * 320, c = rem / 5, q = rem - c * 5;
        const int ch = ci0 + c;
        m = ch < a.Cin ? ~0u : 0u;
        g = (unsigned)row * x_row + (unsigned)(ch < a.Cin ? ch : a.Cin - 1) * 40 + q * 8;
        l = c * XS + row * (XCH * 16) + q * 16;
    };
    unsigned g_dy0, g_x0, g_x1, g_x2;
    int l_dy0, l_x0, l_x1, l_x2;
    uint32_t m_dy0, m_x0, m_x1, m_x2;
    dy_item(tid, g_dy0, l_dy0, m_dy0);
    x_item(tid, g_x0, l_x0, m_x0);
    x_item(tid + 512, g_x1, l_x1, m_x1);
    x_item(tid + 1024, g_x2, l_x2, m_x2);
    uint4 rdh0, rdl0, rxh0, rxl0, rxh1, rxl1, rxh2, rxl2;
    // staging as macros (plain code at both sites; a `?:` on HIP's struct uint4 would select between ADDRESSES and push
    // both operands into scratch: the channel mask is applied word by word instead)
#define WG8_ISSUE(TL)                                                                                              \
    do {                                                                                                           \
        const int n_ = (TL) / tpi;                                                                                 \
        const int trem_ = (TL) - n_ * tpi;                                                                         \
        const int tx_ = trem_ / a.tiles_y, ty_ = trem_ - tx_ * a.tiles_y;   /* rows fastest: consecutive tiles share 2 of 4 halo rows (L2) */ \
        const size_t odyt = ((size_t)(n_ * a.Hp + ty_ * TR) * a.tiles_x + tx_) * a.Cout * 32;                     \
        const size_t oxt = ((size_t)(n_ * a.Hp + ty_ * TR) * a.tiles_x + tx_) * a.Cin * 40;                       \
        const uint16_t *bdyh = a.dyh + odyt, *bdyl = a.dyl + odyt, *bxh = a.xh + oxt, *bxl = a.xl + oxt;           \
        rdh0 = *(const uint4*)(bdyh + g_dy0); rdl0 = *(const uint4*)(bdyl + g_dy0);                                \
        rxh0 = *(const uint4*)(bxh + g_x0); rxl0 = *(const uint4*)(bxl + g_x0);                                    \
        rxh1 = *(const uint4*)(bxh + g_x1); rxl1 = *(const uint4*)(bxl + g_x1);                                    \
        rxh2 = *(const uint4*)(bxh + g_x2); rxl2 = *(const uint4*)(bxl + g_x2);                                    \
    } while (0)
#define WG8_PUT(BASE, OFF, V, M) *(uint4*)((BASE) + (OFF)) = make_uint4((V).x & (M), (V).y & (M), (V).z & (M), (V).w & (M))
#define WG8_COMMIT()                                                                                               \
    do {                                                                                                           \
        WG8_PUT(sdyh, l_dy0, rdh0, m_dy0); WG8_PUT(sdyl, l_dy0, rdl0, m_dy0);                                       \
        WG8_PUT(sxh, l_x0, rxh0, m_x0);    WG8_PUT(sxl, l_x0, rxl0, m_x0);                                          \
        WG8_PUT(sxh, l_x1, rxh1, m_x1);    WG8_PUT(sxl, l_x1, rxl1, m_x1);                                          \
        WG8_PUT(sxh, l_x2, rxh2, m_x2);    WG8_PUT(sxl, l_x2, rxl2, m_x2);                                          \
    } while (0)

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const char* pdyh = sdyh + (wc * 32 + r) * DS + h * 16 + cg * 32;
    const char* pdyl = sdyl + (wc * 32 + r) * DS + h * 16 + cg * 32;
    const char* pxh = sxh + (wi * 32 + r) * XS + h * 16 + cg * 32;
    const char* pxl = sxl + (wi * 32 + r) * XS + h * 16 + cg * 32;

    if (t_beg < t_end) WG8_ISSUE(t_beg);
    for (int tl_ = t_beg; tl_ < t_end; ++tl_) {
        __syncthreads();                               // every wave is done reading the previous tile
        WG8_COMMIT();
        __syncthreads();
        const int tn_ = tl_ + 1 < t_end ? tl_ + 1 : tl_;
        WG8_ISSUE(tn_);                                // in flight during this tile's MFMAs (the last iteration re-reads its own tile: branch-free loop)
        // K step = 16 pixels of one row; lane (r,h) holds pixels 8h..8h+7 of channel r
        bf16x8 ah[TR], al[TR];
#pragma unroll
        for (int row = 0; row < TR; ++row) {
            ah[row] = as_bf(*(const uint4*)(pdyh + row * 64));
            al[row] = as_bf(*(const uint4*)(pdyl + row * 64));
        }
#pragma unroll
        for (int hy = 0; hy < HR; ++hy) {
            // 16-pixel window starting at element 0 of chunk (2cg+h): elements 0..7 (kx=0), 1..8 (kx=1), 2..9 (kx=2)
            const uint4 qh = *(const uint4*)(pxh + hy * (XCH * 16));
            const uint4 ql = *(const uint4*)(pxl + hy * (XCH * 16));
            const uint32_t nh = *(const uint32_t*)(pxh + hy * (XCH * 16) + 16);
            const uint32_t nl = *(const uint32_t*)(pxl + hy * (XCH * 16) + 16);
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
                const int row = hy - ky;               // X row hy pairs with dY row hy-ky for kernel row ky
                if (row < 0 || row >= TR) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    f32x16& d = acc[ky * 3 + kx];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bh[kx], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[row], bh[kx], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[row], bl[kx], d, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);        // keep the operand reads of later halo rows from being hoisted (VGPR cap)
        }
    }
#undef WG8_ISSUE
#undef WG8_COMMIT
#undef WG8_PUT
    // ---- partial block: part[2*split + cg][tap][co][ci], coalesced over ci ----
    const int ci = ci0 + wi * 32 + r;
    const size_t sp = (size_t)split * 2 + cg;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + wc * 32 + acc_row(i, h);
            if (co < a.Cout && ci < a.Cin)
                a.part[((sp * 9 + t) * a.Cout + co) * a.Cin + ci] = acc[t][i];
        }
    }
}

// The single-fp16-product instance: same geometry, only the "hi" planes exist (dY scaled, see the header), 18 MFMAs per tile
// instead of 54 and half the bytes -- so the plane loads run TWO tiles ahead (register sets R0 / R1, the loop body is a pair of
// tiles) to keep the bytes in flight the bf16x3 kernel has with one.  Partials are scaled back by the exact inverse scale.
template <int TRF>
__global__ __launch_bounds__(512, 1) void wgrad_mfma8_f16_kernel(WgArgs a, const unsigned* __restrict__ amax) {
    // FOUR output rows per tile (the bf16x3 kernel: two): 36 MFMAs between the two barriers of a tile instead of 18, and 6 halo rows
    // of X per 4 rows instead of 4 per 2 (a quarter less X traffic)
    constexpr int TR = TRF;
    constexpr int HR = TR + 2, XCH = 5, XS = HR * XCH * 16 + 16, DS = TR * 64 + 16;     // odd multiples of 16 B: conflict-free b128 reads
    constexpr int NDY = 64 * TR * 4, NX = 64 * HR * XCH;         // 16-byte chunks per tile: 1024 / 1920
    constexpr int ND = NDY / 512, NXI = (NX + 511) / 512;        // items per thread: 2 / 4
    static_assert(NDY % 512 == 0 && (XS / 16) % 2 == 1 && (DS / 16) % 2 == 1, "staging geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sdy = smem;
    char* sx = sdy + 64 * DS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cg = wave >> 2, wc = (wave >> 1) & 1, wi = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / a.nblk), blk = slot % a.nblk;
    if (split >= a.S) return;                         // whole workgroup leaves before any barrier
    const int cob = blk / a.nci_blk, cib = blk - cob * a.nci_blk;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int tpi = a.tiles_x * a.tiles_y, ntile = a.B * tpi;
    const int t_beg = split * a.tpw, t_end = (t_beg + a.tpw < ntile) ? t_beg + a.tpw : ntile;
    const unsigned dy_row = (unsigned)a.tiles_x * a.Cout * 32, x_row = (unsigned)a.tiles_x * a.Cin * 40;

    unsigned g_dy[ND], g_x[NXI];
    int l_dy[ND], l_x[NXI];
    uint32_t m_dy[ND], m_x[NXI];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int i = tid + 512 * j;
        const int row = i >> 8, rem = i & 255, c = rem >> 2, q = rem & 3;
        const int ch = co0 + c;
        m_dy[j] = ch < a.Cout ? ~0u : 0u;
        g_dy[j] = (unsigned)(1 + row) * dy_row + (unsigned)(ch < a.Cout ? ch : a.Cout - 1) * 32 + q * 8;
        l_dy[j] = c * DS + row * 64 + q * 16;
    }
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int i = tid + 512 * j, ic = i < NX ? i : NX - 1;
        const int row = ic / 320, rem = ic - row * 320, c = rem / 5, q = rem - c * 5;
        const int ch = ci0 + c;
        m_x[j] = ch < a.Cin ? ~0u : 0u;
        g_x[j] = (unsigned)row * x_row + (unsigned)(ch < a.Cin ? ch : a.Cin - 1) * 40 + q * 8;
        l_x[j] = c * XS + row * (XCH * 16) + q * 16;
    }
    struct Regs { uint4 d[ND], x[NXI]; };
    auto issue = [&](int tl, Regs& R) __attribute__((always_inline)) {
        const int tc = tl < t_end ? tl : t_end - 1;                      // past the end: an unused re-load (branch-free loop)
        const int n_ = tc / tpi, trem_ = tc - n_ * tpi;
        const int tx_ = trem_ / a.tiles_y, ty_ = trem_ - tx_ * a.tiles_y;
        const size_t rowb = (size_t)(n_ * a.Hp + ty_ * TR) * a.tiles_x + tx_;
        const uint16_t *bdy = a.dyh + rowb * a.Cout * 32, *bx = a.xh + rowb * a.Cin * 40;
#pragma unroll
        for (int j = 0; j < ND; ++j) R.d[j] = *(const uint4*)(bdy + g_dy[j]);
#pragma unroll
        for (int j = 0; j < NXI; ++j) R.x[j] = *(const uint4*)(bx + g_x[j]);
    };
    auto put = [&](char* base, int off, const uint4& v, uint32_t m) __attribute__((always_inline)) {
        u32x4 w = {v.x & m, v.y & m, v.z & m, v.w & m};
        *(u32x4*)(base + off) = w;
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const char* pdy = sdy + (wc * 32 + r) * DS + h * 16 + cg * 32;
    const char* px = sx + (wi * 32 + r) * XS + h * 16 + cg * 32;

    auto tile = [&](int tl, Regs& R) __attribute__((always_inline)) {
        __syncthreads();                               // every wave is done reading the previous tile
#pragma unroll
        for (int j = 0; j < ND; ++j) put(sdy, l_dy[j], R.d[j], m_dy[j]);
#pragma unroll
        for (int j = 0; j < NXI; ++j) put(sx, l_x[j], R.x[j], m_x[j]);
        __syncthreads();
        issue(tl + 2, R);                              // in flight during this tile's and the next tile's MFMAs
        f16x8 av[TR];
#pragma unroll
        for (int row = 0; row < TR; ++row) av[row] = __builtin_bit_cast(f16x8, *(const u32x4*)(pdy + row * 64));
#pragma unroll
        for (int hy = 0; hy < HR; ++hy) {
            const u32x4 q = *(const u32x4*)(px + hy * (XCH * 16));
            const uint32_t nx = *(const uint32_t*)(px + hy * (XCH * 16) + 16);
            f16x8 bv[3];
            bv[0] = __builtin_bit_cast(f16x8, q);
            const u32x4 s1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                              __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(nx, q[3], 16)};
            const u32x4 s2 = {q[1], q[2], q[3], nx};
            bv[1] = __builtin_bit_cast(f16x8, s1);
            bv[2] = __builtin_bit_cast(f16x8, s2);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int row = hy - ky;
                if (row < 0 || row >= TR) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[row], bv[kx], acc[ky * 3 + kx], 0, 0, 0);
            }
        }
    };
    Regs R0, R1;
    issue(t_beg, R0);
    issue(t_beg + 1, R1);
    int tl = t_beg;
    for (; tl + 1 < t_end; tl += 2) { tile(tl, R0); tile(tl + 1, R1); }
    if (tl < t_end) tile(tl, R0);

    // The two column groups of the workgroup hold partials of the SAME (co, ci, tap) block: the right group hands its
    // accumulators to the left one through LDS (lane-contiguous rows: conflict-free) and ONE partial per pixel split is written --
    // half the partial traffic of the bf16x3 kernel, for this kernel's epilogue and for the reduce pass behind it
    __syncthreads();                                   // every wave is done with the staging buffers
    float* xs = (float*)smem + (size_t)(wave & 3) * 9 * 16 * 64;
    if (cg == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) xs[(t * 16 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (cg == 1) return;
    const float inv = amax ? f16_plane_scale(*amax, true) : 1.f;
    const int ci = ci0 + wi * 32 + r;
    const size_t sp = (size_t)split;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + wc * 32 + acc_row(i, h);
            const float v = (acc[t][i] + xs[(t * 16 + i) * 64 + lane]) * inv;
            if (co < a.Cout && ci < a.Cin)
                a.part[((sp * 9 + t) * a.Cout + co) * a.Cin + ci] = v;
        }
    }
}

int g_wgrad_splits = 512;     // tedm_set_tuning("wgrad_splits", n): pixel splits x channel blocks aimed at (n workgroups = n / 256 rounds of one per CU)
struct Geo { int tiles_x, tiles_y, Hp, S, tpw, nco, nci; };
constexpr int TR16 = 4;   // output rows per tile of the single-fp16-product kernel
Geo geometry(int Cout, int Cin, int B, int Ho, int Wo, int tr = TR) {
    Geo g;
    g.tiles_x = (Wo + 31) / 32;
    g.tiles_y = (Ho + tr - 1) / tr;
    g.Hp = g.tiles_y * tr + 2;
    g.nco = (Cout + 63) / 64;
    g.nci = (Cin + 63) / 64;
    const long long ntile = (long long)B * g.tiles_x * g.tiles_y;
    long long s = g_wgrad_splits / ((long long)g.nco * g.nci);      // default: ~2 rounds of one resident workgroup per CU x 256 CUs
    if (s > ntile) s = ntile;
    if (s < 1) s = 1;
    const long long t = (ntile + s - 1) / s;
    g.tpw = (int)t;
    g.S = (int)((ntile + t - 1) / t);
    return g;
}
size_t plane_elems(const Geo& g, int B, int C, int pw) { return (size_t)B * g.Hp * g.tiles_x * C * pw; }

int launch_planar(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Hout, int Wout, int ups, int halo, int pro,
                  const float* pa, const float* pb, uint16_t* hi, uint16_t* lo, const Geo& g, hipStream_t st, bool f16 = false,
                  const unsigned* amax = nullptr) {
    PlanarArgs a{};
    a.amax = amax;
    a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Hout = Hout; a.Wout = Wout; a.ups = ups;
    a.pro_a = pa; a.pro_b = pb; a.hi = hi; a.lo = lo; a.Hp = g.Hp; a.tiles_x = g.tiles_x; a.ncb = (C1 + C2 + 63) / 64;
    dim3 grid(g.tiles_x, g.Hp, B * a.ncb);
    if (f16) {
        if (!halo && pro == PRO_NONE) hipLaunchKernelGGL((planar_split_kernel<PRO_NONE, 0, true>), grid, dim3(256), 0, st, a);
        else if (halo && pro == PRO_NONE) hipLaunchKernelGGL((planar_split_kernel<PRO_NONE, 1, true>), grid, dim3(256), 0, st, a);
        else if (halo && pro == PRO_GN) hipLaunchKernelGGL((planar_split_kernel<PRO_GN, 1, true>), grid, dim3(256), 0, st, a);
        else return TEDM_EINVAL;
        return LAUNCH_OK();
    }
    if (!halo && pro == PRO_NONE) hipLaunchKernelGGL((planar_split_kernel<PRO_NONE, 0>), grid, dim3(256), 0, st, a);
    else if (halo && pro == PRO_NONE) hipLaunchKernelGGL((planar_split_kernel<PRO_NONE, 1>), grid, dim3(256), 0, st, a);
    else if (halo && pro == PRO_GN) hipLaunchKernelGGL((planar_split_kernel<PRO_GN, 1>), grid, dim3(256), 0, st, a);
    else return TEDM_EINVAL;
    return LAUNCH_OK();
}

}  // namespace

int wgrad_mfma_set_tuning(const char* key, int v) {
    if (!strcmp(key, "wgrad_splits") && v >= 64 && v <= 4096) { g_wgrad_splits = v; return TEDM_OK; }
    return TEDM_EINVAL;
}

// Workspace (in floats) of the MFMA path: partials + four bf16 tile planes.
size_t wgrad_mfma_floats(int k, int Cout, int Cin, int B, int Ho, int Wo) {
    if (k != 3) return 0;
    const Geo g = geometry(Cout, Cin, B, Ho, Wo);
    const size_t part = ((size_t)g.S * 2 * 9 * Cout * Cin + 63) & ~(size_t)63;
    const size_t planes = (plane_elems(g, B, Cout, 32) + plane_elems(g, B, Cin, 40)) * 2;    // hi + lo, 2 bytes each
    const size_t n2 = part + (planes * 2 + 3) / 4 + 64;
    // the single-fp16-product layout (4-row tiles, one plane per operand, one partial per split, 64 floats for the maximum)
    const Geo h = geometry(Cout, Cin, B, Ho, Wo, TR16);
    const size_t part16 = ((size_t)h.S * 9 * Cout * Cin + 63) & ~(size_t)63;
    const size_t n16 = part16 + 64 + ((plane_elems(h, B, Cout, 32) + plane_elems(h, B, Cin, 40)) * 2 + 3) / 4 + 64;
    return n2 > n16 ? n2 : n16;
}

// 3x3, stride 1, prologue NONE or GN.  ups: the conv read the nearest-x2 upsampled input (H,W are the SOURCE size, Ho=2H).
// Fills part[S'][9][Cout][Cin] and returns S' in *S_out; the caller runs the split reduction.
int op_conv_wgrad_mfma(const float* dy, int Cout, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Ho, int Wo,
                       int k, int ups, int pro, const float* pro_a, const float* pro_b, float* ws, int* S_out, int f16, const unsigned* dy_amax, hipStream_t st) {
    const int Cin = C1 + C2;
    if (k != 3 || Cin % 4 || C1 % 4 || Cout % 4 || (pro != PRO_NONE && pro != PRO_GN)) return TEDM_EINVAL;
    const Geo g = geometry(Cout, Cin, B, Ho, Wo);
    const size_t part = ((size_t)g.S * 2 * 9 * Cout * Cin + 63) & ~(size_t)63;
    const size_t ndy = plane_elems(g, B, Cout, 32), nx = plane_elems(g, B, Cin, 40);
    if (ndy >= (1ull << 31) || nx >= (1ull << 31)) return TEDM_EINVAL;        // the kernel's per-thread offsets are 32-bit
    uint16_t* dyh = (uint16_t*)(ws + part);
    uint16_t* dyl = dyh + ndy;
    uint16_t* xh = dyl + ndy;
    uint16_t* xl = xh + nx;
    if (f16) {
        // layout: [partials S x 9 x Cout x Cin | 64 floats: the maximum | dY plane | X plane], 4-row tiles
        const Geo h = geometry(Cout, Cin, B, Ho, Wo, TR16);
        const size_t part16 = ((size_t)h.S * 9 * Cout * Cin + 63) & ~(size_t)63;
        const size_t ndy16 = plane_elems(h, B, Cout, 32), nx16 = plane_elems(h, B, Cin, 40);
        if (ndy16 >= (1ull << 31) || nx16 >= (1ull << 31)) return TEDM_EINVAL;
        unsigned* am = (unsigned*)(ws + part16);
        uint16_t* dy16 = (uint16_t*)(ws + part16 + 64);
        uint16_t* x16 = dy16 + ((ndy16 + 7) & ~(size_t)7);
        const unsigned* amax = dy_amax;
        if (!amax) {                                   // the producer of dY did not track it: one more pass over dY
            if (hipMemsetAsync(am, 0, 4, st) != hipSuccess) return TEDM_EHIP;
            const long long n4 = (long long)B * Ho * Wo * Cout / 4;
            const int nb = (int)((n4 + 256 * 8 - 1) / (256 * 8) < 2048 ? (n4 + 256 * 8 - 1) / (256 * 8) : 2048);
            hipLaunchKernelGGL(absmax_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, (const float4*)dy, n4, am);
            amax = am;
        }
        int rc = launch_planar(dy, Cout, nullptr, 0, B, Ho, Wo, Ho, Wo, 0, 0, PRO_NONE, nullptr, nullptr, dy16, nullptr, h, st, true, amax);
        if (rc != TEDM_OK) return rc;
        rc = launch_planar(x1, C1, x2, C2, B, H, W, Ho, Wo, ups, 1, pro, pro_a, pro_b, x16, nullptr, h, st, true, nullptr);
        if (rc != TEDM_OK) return rc;
        WgArgs a{};
        a.dyh = dy16; a.xh = x16; a.Cout = Cout; a.Cin = Cin; a.B = B; a.Hp = h.Hp;
        a.tiles_x = h.tiles_x; a.tiles_y = h.tiles_y; a.part = ws; a.S = h.S; a.tpw = h.tpw; a.nci_blk = h.nci; a.nblk = h.nco * h.nci;
        *S_out = h.S;                                  // (the kernel merges its two column groups)
        const size_t lds = (size_t)4 * 9 * 16 * 64 * 4;        // the exchange of the column groups (147 456 B) covers the staging buffers (49 KB)
        static bool attr16[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return TEDM_EHIP;
        if (dev < 0 || dev >= 16 || !attr16[dev]) {
            if (hipFuncSetAttribute((const void*)wgrad_mfma8_f16_kernel<TR16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEDM_EHIP;
            if (dev >= 0 && dev < 16) attr16[dev] = true;
        }
        hipLaunchKernelGGL(wgrad_mfma8_f16_kernel<TR16>, dim3(8 * ((a.S + 7) / 8) * a.nblk), dim3(512), lds, st, a, amax);
        return LAUNCH_OK();
    }
    int rc = launch_planar(dy, Cout, nullptr, 0, B, Ho, Wo, Ho, Wo, 0, 0, PRO_NONE, nullptr, nullptr, dyh, dyl, g, st);
    if (rc != TEDM_OK) return rc;
    rc = launch_planar(x1, C1, x2, C2, B, H, W, Ho, Wo, ups, 1, pro, pro_a, pro_b, xh, xl, g, st);
    if (rc != TEDM_OK) return rc;
    WgArgs a{};
    a.dyh = dyh; a.dyl = dyl; a.xh = xh; a.xl = xl; a.Cout = Cout; a.Cin = Cin; a.B = B; a.Hp = g.Hp;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.part = ws; a.S = g.S; a.tpw = g.tpw; a.nci_blk = g.nci; a.nblk = g.nco * g.nci;
    *S_out = 2 * g.S;
    const size_t lds = (size_t)64 * (2 * 144 + 2 * 336);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad_mfma8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEDM_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL(wgrad_mfma8_kernel, dim3(8 * ((a.S + 7) / 8) * a.nblk), dim3(512), lds, st, a);
    return LAUNCH_OK();
}
