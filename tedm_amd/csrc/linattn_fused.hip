// linattn_fused.hip -- LinearAttention (models/unet_model.py:178-210) without the (B,384,h,w) qkv
// tensor.  The reference materialises to_qkv(LN(x)) (1.5 KB per pixel in fp32) and reads it back
// twice; here the 1x1 projection is RE-COMPUTED from x in three sweeps, each fused with what
// consumes it, so per pixel the path reads x three times and writes only the 128-channel
// attention output:
//   sweep KMAX : k = Wk LN(x)            -> per-(sample, channel) max over pixels (softmax stabiliser)
//   sweep CTX  : k, v = Wk,Wv LN(x)      -> ctx_p[d][e] = sum_n exp(k[n][d]-kmax[d]) v[n][e],  z_p[d]
//                (kmax == NULL: ONLINE form -- no KMAX sweep; every workgroup keeps a running per-channel max m_p[d],
//                 rescales its accumulators when it grows, and the finalize kernel merges the partials with
//                 exp(m_p - max_p m_p), exactly like a split softmax)
//   sweep QOUT : q = Wq LN(x)            -> o[n][e] = sum_d ctx[d][e] softmax_d(q[n])[d] * 32^-0.5
// One wave per head; a workgroup walks TPW consecutive 128-pixel tiles of one image (1x1 convolution:
// pixels are a flat index).  The projection runs on v_mfma_f32_32x32x16_{bf16,f16} in the handle's
// precision mode (bf16 hi/lo split by default), the attention products on the exact-fp32
// v_mfma_f32_32x32x2_f32.  KMAX/CTX use the accumulator as D[pixel][channel] (channel on the lane:
// k is the A operand [i=d][k=pixel], v the B operand [k=pixel][j=e] of the context MFMA, straight
// from the registers); QOUT swaps the operands to get D[channel][pixel] (pixel on the lane), so the
// softmax over d is lane-local and the result is directly the B operand of ctx^T q.
#include "common.h"
#include "ops.h"
#include <type_traits>
#include <string.h>
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

enum { LA_KMAX = 0, LA_CTX = 1, LA_QOUT = 2 };


struct LaArgs {
    const float* x; int C;             // (B, hw, C) fp32
    const float* g;                    // PreNorm LayerNorm gain [C]
    const float* lnstats;              // (B*hw, 2) = (rstd, mean*rstd)
    const void* wpk;                   // packed to_qkv weight (3*heads*32 x C, 1 tap)
    int B; long long hw; int heads;
    int tpw, S;                        // tiles per workgroup, workgroups per image
    float* kmax_part;                  // KMAX out: [B][S][heads*32]
    const float* kmax;                 // CTX in:   [B][heads*32]
    float* ctx_part;                   // CTX out:  [B][S][heads][1024+32]
    const float* ctx;                  // QOUT in:  [B][heads][32][32]
    float* o;                          // QOUT out: (B, hw, heads*32)
};

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// Registers 8s..8s+7 of a 32x32 fp32 accumulator tile (or any 8 values in that order) as the k-step-s fragment of a
// v_mfma_f32_32x32x16_f16 that sums over the tile's ROW index (cdna guide section 3, "an accumulator tile as the next MFMA's
// operand": element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3); both operands of the product must use that order).
// fp16 hi + fp16 lo (22 significand bits): hi*hi + lo*hi + hi*lo reproduces the fp32 product to ~2^-21, at 6 MFMA issues
// of 32 cycles per 32x32x32 block instead of the 16 x 64 cycles of v_mfma_f32_32x32x2_f32.
TEDM_DEVFN void split_frag(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = (_Float16)v[j];
        hi[j] = hj;
        lo[j] = (_Float16)(v[j] - (float)hj);
    }
}
TEDM_DEVFN f32x16 mfma_f16x3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
}

// LayerNorm statistics (rstd, mean * rstd) of a 64-channel pixel whose channels sit as 2 x 8 values on each lane of a quad
// (ln_stats_kernel's two-pass formula: mean, then the centred sum of squares)
TEDM_DEVFN float quad_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }
TEDM_DEVFN float2 quad_ln_stats(const float4 (&a)[2], const float4 (&b)[2]) {
    float s = ((a[0].x + a[0].y) + (a[0].z + a[0].w)) + ((a[1].x + a[1].y) + (a[1].z + a[1].w))
            + ((b[0].x + b[0].y) + (b[0].z + b[0].w)) + ((b[1].x + b[1].y) + (b[1].z + b[1].w));
    const float mean = quad_sum(s) * (1.f / 64.f);
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float t;
        t = a[i].x - mean; d2 = fmaf(t, t, d2); t = a[i].y - mean; d2 = fmaf(t, t, d2);
        t = a[i].z - mean; d2 = fmaf(t, t, d2); t = a[i].w - mean; d2 = fmaf(t, t, d2);
        t = b[i].x - mean; d2 = fmaf(t, t, d2); t = b[i].y - mean; d2 = fmaf(t, t, d2);
        t = b[i].z - mean; d2 = fmaf(t, t, d2); t = b[i].w - mean; d2 = fmaf(t, t, d2);
    }
    const float rstd = rsqrtf(quad_sum(d2) * (1.f / 64.f) + 1e-5f);
    return make_float2(rstd, mean * rstd);
}

constexpr int KS = 2, KC = 16 * KS;             // channels per staged chunk
// LDS bytes per pixel per plane (odd multiples of 16): bf16 / fp16 planes 80; F16F4 one 112-byte record
// [fp16 x32 | fp4 residual x32 | fp4 copy x32 | 2 scale bytes | pad] exactly as conv_igemm.inc stages it
constexpr int pstride(int prec) { return prec == TEDM_PREC_F16F4 ? 112 : KS * 32 + 16; }

#ifndef LA_CTX_MT
#define LA_CTX_MT 2
#endif
template <int EPI> struct LaTile { static constexpr int MT = EPI == LA_CTX ? LA_CTX_MT : 4; };   // M-tiles (32 px) per tile

// ONE (CTX sweep, F16F4, C = 64): both 32-channel chunks of a 64-pixel tile are staged in front of ONE barrier (four LDS buffers:
// tile parity x chunk), the k / v weights of both chunks stay in registers and the next tile's activations are in flight for a
// whole tile period -- half the barriers per byte of the chunk-by-chunk loop.
// ONE == 2: the same, and the PreNorm LayerNorm statistics come from the kernel itself (a.lnstats unused): a thread holds 8 + 8 of
// a pixel's 64 channels (both chunks in registers), the four threads of a quad the whole pixel -- mean and centred variance are
// two quad reductions (the formula of ln_stats_kernel), which drops the separate statistics pass over the tensor.
template <int EPI, int PREC, int ONE = 0>
__global__ __launch_bounds__(256, 2) void la_proj_kernel(LaArgs a) {
    static_assert(!ONE || (EPI == LA_CTX && PREC == TEDM_PREC_F16F4), "ONE: the C = 64 context sweep in f16f4");
    constexpr bool SELF = ONE == 2;
    constexpr int MT = LaTile<EPI>::MT, TP = MT * 32, NIT = MT / 2;   // TP*4 pieces / 256 threads
    constexpr bool F4 = PREC == TEDM_PREC_F16F4;
    constexpr int PSTRIDE = pstride(PREC);
    constexpr int REGION = TP * PSTRIDE;
    constexpr int NT = EPI == LA_CTX ? 2 : 1;                    // weight tiles per wave (k [, v])
    constexpr int NPL = PREC == TEDM_PREC_BF16X3 ? 2 : 1;        // activation planes in LDS
    constexpr int NPB = F4 ? 3 : (PREC == TEDM_PREC_BF16 ? 1 : 2);   // weight fragments per (tile, chunk) [F4] / planes per 16-channel step
    constexpr int BUFSZ = REGION * NPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (F4) fp16_saturate_on();

    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / a.S, sidx = blockIdx.x - n * a.S;
    const long long tiles = (a.hw + TP - 1) / TP;
    const long long t_beg = (long long)sidx * a.tpw, t_end = (t_beg + a.tpw < tiles) ? t_beg + a.tpw : tiles;
    const int nk16 = a.C / 16, NC = a.C / KC;
    const int hd = a.heads;
    const uint4* wq = (const uint4*)a.wpk;
    // weight tile of this wave: QOUT -> q block, KMAX -> k block, CTX -> k block then v block
    const int ct0 = (EPI == LA_QOUT ? 0 : hd) + head;

    float run_max = -INFINITY, zsum = 0.f;
    f32x16 cacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
    float kmx = 0.f;
    const bool online = EPI == LA_CTX && a.kmax == nullptr;       // block-uniform
    if (EPI == LA_CTX) kmx = online ? -INFINITY : a.kmax[(size_t)n * hd * 32 + head * 32 + r];
    // QOUT: A operand ctx^T[e = r][d] of o^T = ctx^T q' in the row order of the q' accumulator (split_frag), times hw so
    // that the entries (weighted means of v / hw, ~1e-5 at 256x256) sit in fp16's normal range; the 1/hw is applied to o
    f16x8 cth[2], ctl[2];
    if (EPI == LA_QOUT) {
        const float* cb = a.ctx + ((size_t)n * hd + head) * 1024;
        const float fhw = (float)a.hw;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float cv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cv[j] = cb[(16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) * 32 + r] * fhw;
            split_frag(cv, cth[s], ctl[s]);
        }
    }

    const int piece = tid & 3;                                    // 8-channel piece of the 32-channel chunk
    const int niter = (int)(t_end - t_beg) * NC;                  // flattened (tile, chunk) iterations

    // Software pipeline, prefetch distance 2: the raw activations of iterations it+1 and it+2 are in flight while
    // iteration it computes (two register sets, loop unrolled by two).  At 64 px x 32 ch x 4 B = 8 KB per iteration and two
    // workgroups per CU a distance of one keeps 4 MB in flight on the chip, i.e. < 2 TB/s at ~2 us loaded HBM latency:
    // the sweeps were latency-bound (2.8 us per iteration at 256x256).  VMEM returns in order, so the (L2-hit) weight loads
    // of iteration it+1 are issued BEFORE prefetch(it+2): queued after it they would wait for its HBM round trip.
    struct Raw { float4 v[NIT][2]; float2 s[NIT]; bool ok[NIT]; };
    float* gl = (float*)(smem + (ONE ? 4 : 2) * BUFSZ);          // PreNorm gain [C] (LDS: 16 fewer live registers than per-set copies)
    for (int i = tid; i < a.C; i += 256) gl[i] = a.g[i];
    __syncthreads();
    // NB (measured on gfx950 / ROCm 7.2): with these loads inside `if (pixel valid)` blocks, lanes 48-63
    // staged stale registers whenever more than one workgroup shared a CU -- the counted s_waitcnt
    // around lane-divergent VMEM blocks was not sufficient.  Keep every VMEM instruction unconditional.
    auto prefetch = [&](int it, Raw& R) __attribute__((always_inline)) {
        float4 (&rv)[NIT][2] = R.v; float2 (&rs)[NIT] = R.s; bool (&rok)[NIT] = R.ok;
        const long long t = t_beg + it / NC;
        const int cb = (it % NC) * KC + piece * 8;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            // every lane loads unconditionally (address clamped to the last pixel of the image, value
            // masked when staged): no VMEM instruction sits in a lane-divergent branch
            long long p = t * TP + (tid >> 2) + k * 64;
            rok[k] = p < a.hw;
            p = rok[k] ? p : a.hw - 1;
            const size_t gp = (size_t)n * a.hw + p;
            const float4* src = (const float4*)(a.x + gp * a.C + cb);
            rv[k][0] = src[0]; rv[k][1] = src[1];
            if (!SELF) rs[k] = *(const float2*)(a.lnstats + gp * 2);
        }
    };
    auto stage = [&](int buf, int c, const Raw& R) __attribute__((always_inline)) {
        const float4 (&rv)[NIT][2] = R.v; const float2 (&rs)[NIT] = R.s; const bool (&rok)[NIT] = R.ok;
        const float4 rg[2] = {*(const float4*)(gl + c * KC + piece * 8), *(const float4*)(gl + c * KC + piece * 8 + 4)};
        char* base = smem + buf * BUFSZ;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            float v[8];
            {
                const float lr = rok[k] ? rs[k].x : 0.f, lm = rok[k] ? rs[k].y : 0.f;   // masked pixel -> exact zeros
                v[0] = rg[0].x * (rv[k][0].x * lr - lm); v[1] = rg[0].y * (rv[k][0].y * lr - lm);
                v[2] = rg[0].z * (rv[k][0].z * lr - lm); v[3] = rg[0].w * (rv[k][0].w * lr - lm);
                v[4] = rg[1].x * (rv[k][1].x * lr - lm); v[5] = rg[1].y * (rv[k][1].y * lr - lm);
                v[6] = rg[1].z * (rv[k][1].z * lr - lm); v[7] = rg[1].w * (rv[k][1].w * lr - lm);
                if (!rok[k]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = 0.f;
                }
            }
            const int lofs = ((tid >> 2) + k * 64) * PSTRIDE + piece * 16;
            if (F4) {
                float lo[8], m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
                unsigned e_v4, e_lo;
                f4_scales(quad_max(m), e_v4, e_lo);
                const uint4 hi = split8_f16_res_sat(v, lo);
                char* rec = base + ((tid >> 2) + k * 64) * PSTRIDE;
                *(uint4*)(rec + piece * 16) = hi;
                *(unsigned*)(rec + 64 + piece * 4) = cvt8_fp4(lo, e_lo);
                *(unsigned*)(rec + 80 + piece * 4) = cvt8_fp4(v, e_v4);
                if (piece == 0) *(unsigned short*)(rec + 96) = (unsigned short)(e_lo | (e_v4 << 8));
            } else if (PREC == TEDM_PREC_FP16X2) {
                *(uint4*)(base + lofs) = cvt8_f16(v);
            } else {
                uint4 hi, lo;
                split8(v, hi, lo);
                *(uint4*)(base + lofs) = hi;
                if (NPL == 2) *(uint4*)(base + REGION + lofs) = lo;
            }
        }
    };

    f32x16 acc[MT][NT];
    uint4 bq[KS][NT][NPB];
    unsigned bsc[NT];
    auto load_w = [&](int c) __attribute__((always_inline)) {      // weights of chunk c (L2 hits)
        if (F4) {
            // conv_pack_f16f4_kernel layout of a 1x1 weight: per (cout tile, 32-channel chunk) 3 fragments, then the scale dwords
            const unsigned* wsc = (const unsigned*)(wq + (size_t)(3 * hd) * NC * 3 * 64);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[0][j][p] = wq[(((size_t)(ct0 + j * hd) * NC + c) * 3 + p) * 64 + lane];
                bsc[j] = wsc[((size_t)(ct0 + j * hd) * NC + c) * 64 + lane];
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int p = 0; p < NPB; ++p)
                        bq[ks][j][p] = wq[(((size_t)(ct0 + j * hd) * nk16 + (c * KS + ks)) * 2 + p) * 64 + lane];
        }
    };
    // Every VMEM instruction of the loop is issued UNCONDITIONALLY (past the end: a clamped, unused re-load): the waitcnt
    // pass merges the counts of all paths into a block, and one path without the younger loads turns the counted
    // vmcnt(N) in front of the staging into vmcnt(0) -- which waits for the prefetch issued one iteration ago as well.
    // epilogue of one tile of the CTX sweep (acc[mt][0] = k, acc[mt][1] = v of the tile's pixels)
    auto ctx_epilogue = [&](long long p0) __attribute__((always_inline)) {
        if constexpr (EPI == LA_CTX) {
            const bool full_tile = p0 + TP <= a.hw;
            if (online) {
                // running max of this lane's k channel over the workgroup's pixels; rescale what has been accumulated
                float tmax = -INFINITY;
                if (full_tile) {        // no per-pixel masks; v_max3 on the raw accumulators (fmaxf quiets every input first)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const float x0 = acc[mt][0][i], x1 = acc[mt][0][i + 1];
                            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(x0), "v"(x1));
                        }
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (p0 + mt * 32 + acc_row(i, h) < a.hw) tmax = fmaxf(tmax, acc[mt][0][i]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(kmx, tmax);
                const float f = (m_new == kmx) ? 1.f : __expf(kmx - m_new);     // kmx = -inf on the first tile -> 0 (accumulators are 0)
                if (__any(f != 1.f)) {                          // wave-uniform: after the first tiles the running max rarely moves
                    zsum *= f;
#pragma unroll
                    for (int s_ = 0; s_ < 16; ++s_) cacc[s_] *= __shfl(f, acc_row(s_, h));   // row d of ctx lives on lane d
                }
                kmx = m_new;
            }
            // ctx[d][e] += sum_pixel exp(k[pixel][d] - kmx[d]) v[pixel][e]: both accumulators (lane = channel, registers =
            // pixels) are operands of a product over their row index as they stand (split_frag)
            const float kl = kmx * 1.44269504088896340736f;   // exp(k - kmx) = exp2(k log2e - kmx log2e): one FMA + v_exp_f32
            auto accum = [&](auto full) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full)::value;      // whole tile inside the image: no per-pixel masks
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    float ev[8], vv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * s_ + j;
                        const bool ok = FULL || p0 + mt * 32 + acc_row(i, h) < a.hw;
                        ev[j] = ok ? __builtin_amdgcn_exp2f(fmaf(acc[mt][0][i], 1.44269504088896340736f, -kl)) : 0.f;
                        vv[j] = ok ? acc[mt][NT - 1][i] : 0.f;
                        zsum += ev[j];
                    }
                    // ONE fp16 product: the fp16 roundings of e (in [0,1]) and v are independent across the hw pixels the sum
                    // runs over, so their relative contribution to ctx shrinks like 2^-12 / sqrt(hw) -- far below the
                    // projection's own error; the normaliser z is summed in fp32 above
                    // (F16F4 mode only; the other modes keep the fp32-grade hi/lo product)
                    if (F4) {
                        f16x8 eh, vh;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { eh[j] = (_Float16)ev[j]; vh[j] = (_Float16)vv[j]; }
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, vh, cacc, 0, 0, 0);
                    } else {
                        f16x8 eh, el, vh, vl;
                        split_frag(ev, eh, el);
                        split_frag(vv, vh, vl);
                        cacc = mfma_f16x3(eh, el, vh, vl, cacc);
                    }
                }
            };
            if (full_tile) accum(std::true_type{}); else accum(std::false_type{});
        }
    };
    if constexpr (ONE) {
        // weights of both chunks resident: chunk 0 in bq[0] / bsc, chunk 1 in bq1 / bsc1
        uint4 bq1[NT][NPB]; unsigned bsc1[NT];
        load_w(1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int p = 0; p < NPB; ++p) bq1[j][p] = bq[0][j][p];
            bsc1[j] = bsc[j];
        }
        load_w(0);
        Raw Q0, Q1;
        const int ntile = (int)(t_end - t_beg);
        if (ntile > 0) { prefetch(0, Q0); prefetch(1, Q1); }
        for (int tl = 0; tl < ntile; ++tl) {
            const int b0 = (tl & 1) * 2;
            if constexpr (SELF) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const float2 st2 = quad_ln_stats(Q0.v[k], Q1.v[k]);
                    Q0.s[k] = st2; Q1.s[k] = st2;
                }
            }
            stage(b0, 0, Q0);
            stage(b0 + 1, 1, Q1);
            const int tn = tl + 1 < ntile ? tl + 1 : tl;          // past the last tile: an unused re-load of it
            prefetch(2 * tn, Q0);
            prefetch(2 * tn + 1, Q1);
            __syncthreads();       // this tile's two buffers complete; the other parity's readers finished before their barrier
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const char* sb = smem + (b0 + c) * BUFSZ;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* rec = sb + (mt * 32 + r) * PSTRIDE;
                    const f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + h * 16));
                    const f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + 32 + h * 16));
                    const uint4 aq = *(const uint4*)(rec + 64 + h * 16);
                    const int as = *(const unsigned char*)(rec + 96 + h);
                    i32x8 a4 = {};
                    a4[0] = aq.x; a4[1] = aq.y; a4[2] = aq.z; a4[3] = aq.w;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const uint4* wb = c ? bq1[j] : bq[0][j];
                        const f16x8 b0f = __builtin_bit_cast(f16x8, wb[0]), b1f = __builtin_bit_cast(f16x8, wb[1]);
                        i32x8 b4 = {};
                        b4[0] = wb[2].x; b4[1] = wb[2].y; b4[2] = wb[2].z; b4[3] = wb[2].w;
                        const int bsv = (int)(c ? bsc1[j] : bsc[j]);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0f, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1f, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, acc[mt][j], 4, 4, 0, as, 0, bsv);
                    }
                }
            }
            ctx_epilogue((t_beg + tl) * TP);
        }
    } else {
    Raw R0, R1;
    if (niter > 0) { prefetch(0, R0); load_w(0); prefetch(niter > 1 ? 1 : 0, R1); }
    auto body = [&](int it, Raw& R) __attribute__((always_inline)) {
        const int c = it % NC, buf = it & 1;
        const long long p0 = (t_beg + it / NC) * TP;
        if (c == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;
        }
        stage(buf, c, R);
        __syncthreads();          // buffer `buf` complete; every wave has finished the MFMAs of iteration it-1
        const char* sb = smem + buf * BUFSZ;
        if constexpr (F4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const char* rec = sb + (mt * 32 + r) * PSTRIDE;
                const f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + h * 16));
                const f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + 32 + h * 16));
                const uint4 aq = *(const uint4*)(rec + 64 + h * 16);
                const int as = *(const unsigned char*)(rec + 96 + h);
                i32x8 a4 = {};
                a4[0] = aq.x; a4[1] = aq.y; a4[2] = aq.z; a4[3] = aq.w;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f16x8 b0 = __builtin_bit_cast(f16x8, bq[0][j][0]), b1 = __builtin_bit_cast(f16x8, bq[0][j][1]);
                    i32x8 b4 = {};
                    b4[0] = bq[0][j][2].x; b4[1] = bq[0][j][2].y; b4[2] = bq[0][j][2].z; b4[3] = bq[0][j][2].w;
                    if (EPI == LA_QOUT) {     // D[channel][pixel]: weights are the A operand ([w4 | w_lo] x [a_lo ; a4])
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b4, a4, acc[mt][j], 4, 4, 0, (int)bsc[j], 0, as);
                    } else {                  // D[pixel][channel]
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, acc[mt][j], 4, 4, 0, as, 0, (int)bsc[j]);
                    }
                }
            }
        } else
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int off = (mt * 32 + r) * PSTRIDE + ks * 32 + h * 16;
                const uint4 ahq = *(const uint4*)(sb + off);
                uint4 alq = ahq;
                if (NPL == 2) alq = *(const uint4*)(sb + REGION + off);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (PREC == TEDM_PREC_FP16X2) {
                        const f16x8 af = __builtin_bit_cast(f16x8, ahq);
                        const f16x8 b0 = __builtin_bit_cast(f16x8, bq[ks][j][0]), b1 = __builtin_bit_cast(f16x8, bq[ks][j][NPB - 1]);
                        if (EPI == LA_QOUT) {
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, af, acc[mt][j], 0, 0, 0);
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, af, acc[mt][j], 0, 0, 0);
                        } else {
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b0, acc[mt][j], 0, 0, 0);
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b1, acc[mt][j], 0, 0, 0);
                        }
                    } else {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahq), al = __builtin_bit_cast(bf16x8, alq);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[ks][j][0]), bl = __builtin_bit_cast(bf16x8, bq[ks][j][NPB - 1]);
                        if (EPI == LA_QOUT) {     // D[channel][pixel]
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[mt][j], 0, 0, 0);
                            if (NPL == 2) {
                                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[mt][j], 0, 0, 0);
                                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[mt][j], 0, 0, 0);
                            }
                        } else {                  // D[pixel][channel]
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt][j], 0, 0, 0);
                            if (NPL == 2) {
                                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt][j], 0, 0, 0);
                                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt][j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        load_w((it + 1) % NC);                                 // this iteration's fragments are dead: next weights, then ...
        prefetch(it + 2 < niter ? it + 2 : niter - 1, R);      // ... the activations two iterations ahead into the freed set
        if (c != NC - 1) return;

        // ---- epilogue of the tile -------------------------------------------------------------
        if (EPI == LA_KMAX) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (p0 + mt * 32 + acc_row(i, h) < a.hw) run_max = fmaxf(run_max, acc[mt][0][i]);
        } else if (EPI == LA_CTX) {
            ctx_epilogue(p0);
        } else {
            const float scale = 0.17677669529663688110f;         // 32^-0.5
            const float inv_hw = 1.f / (float)a.hw;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float m = acc[mt][0][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[mt][0][i]);
                m = fmaxf(m, __shfl_xor(m, 32));
                float q[16], sum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { q[i] = __expf(acc[mt][0][i] - m); sum += q[i]; }
                sum += __shfl_xor(sum, 32);
                const float f = scale / sum;
                f32x16 oacc;
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float qv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) qv[j] = q[8 * s + j] * f;
                    f16x8 qh, ql;
                    split_frag(qv, qh, ql);
                    oacc = mfma_f16x3(cth[s], ctl[s], qh, ql, oacc);
                }
                const long long p = p0 + mt * 32 + r;
                if (p < a.hw) {
                    float* op = a.o + ((size_t)n * a.hw + p) * (hd * 32) + head * 32 + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *(float4*)(op + 8 * gq) = make_float4(oacc[4 * gq] * inv_hw, oacc[4 * gq + 1] * inv_hw, oacc[4 * gq + 2] * inv_hw,
                                                              oacc[4 * gq + 3] * inv_hw);   // the ctx fragments carry a factor hw
                }
            }
        }
    };
    // pairs only inside the loop (a conditional second half would add a path on which the first half's own prefetch is the
    // youngest load: the merged count in front of the next staging would drop to vmcnt(0)); an odd tail runs after it
    int it2 = 0;
    for (; it2 + 1 < niter; it2 += 2) { body(it2, R0); body(it2 + 1, R1); }
    if (it2 < niter) body(it2, R0);

    }

    if (EPI == LA_KMAX) {
        run_max = fmaxf(run_max, __shfl_xor(run_max, 32));
        if (h == 0) a.kmax_part[((size_t)n * a.S + sidx) * (hd * 32) + head * 32 + r] = run_max;
    } else if (EPI == LA_CTX) {
        zsum += __shfl_xor(zsum, 32);
        float* cp = a.ctx_part + (((size_t)n * a.S + sidx) * hd + head) * (online ? 1024 + 64 : 1024 + 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) cp[acc_row(i, h) * 32 + r] = cacc[i];
        if (h == 0) cp[1024 + r] = zsum;
        if (online && h == 0) cp[1056 + r] = kmx;
    }
}

// ---- QOUT fused with the rest of the block (F16F4 only, C <= 256) --------------------------------------------------
//   out = LN_c( to_out( ctx^T softmax_d(q) 32^-0.5 ) + bias ) * g_out + x          (models/unet_model.py:187-190,205-210,
//                                                                                     Residual :29-36)
// The separate path writes o (128 ch), reads it back for the 1x1 to_out conv, writes z, and reads z and x again for the
// LayerNorm + residual: 10 tensor passes of C channels at C = 64.  Here o never leaves the CU: per 64-pixel tile the four
// head-waves write their 32 output channels as F16F4 operand records into LDS (one 32-channel block per head: exactly
// the MX block), the to_out product runs as z^T[cout][pixel] = W o^T with the pixel on the lane, and the channel
// LayerNorm is lane-local sums plus one exchange between the two waves that share a pixel group.  Reads x twice (the
// second time from L2), writes out once.
struct LaTailArgs {
    const float* x; int C; const float* g; const float* lnstats; const void* wqkv; const float* ctx;
    const void* wout; const float* bias; const float* g_out; float* out;
    int B; long long hw; int heads, tpw, S;
};

// RAGGED: hw is not a multiple of the 64-pixel tile -- the output stores of the last tile are masked per lane.  The even
// case keeps every VMEM instruction of the tile loop outside conditional code (exact vmcnt counts, see la_proj_kernel).
// SELF (C = 64): PreNorm LayerNorm statistics from the kernel's own registers (quad_ln_stats), a.lnstats unused.
template <int NT2, bool RAGGED, bool SELF = false>
__global__ __launch_bounds__(256, 2) void la_qout_tail_kernel(LaTailArgs a) {
    static_assert(!SELF || NT2 == 2, "own LayerNorm statistics: both chunks of a pixel in registers (C = 64)");
    constexpr int MT = 2, TP = 64, PS = 112;                  // 64-pixel tiles, F16F4 records
    constexpr int NTH = NT2 / 2;                              // cout tiles per wave in the tail (two waves share a pixel group)
    constexpr int STG = TP * PS;                              // one staging buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* orec = smem + 2 * STG;                              // [4 heads][64 px] records of o (pixel stride 112 B: conflict-free reads)
    float2* lnx = (float2*)(orec + TP * 4 * PS);              // [64 px][2 cout halves] (sum, sumsq)
    fp16_saturate_on();
    float* bgl = (float*)(lnx + TP * 2);                      // to_out bias [C], output LayerNorm gain [C], PreNorm gain [C]

    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / a.S, sidx = blockIdx.x - n * a.S;
    const long long tiles = (a.hw + TP - 1) / TP;
    const long long t_beg = (long long)sidx * a.tpw, t_end = (t_beg + a.tpw < tiles) ? t_beg + a.tpw : tiles;
    constexpr int NC = NT2;                                   // 32-channel chunks of x == cout tiles of to_out (C / 32)
    const int hd = a.heads;
    if (threadIdx.x < a.C) {
        bgl[threadIdx.x] = a.bias[threadIdx.x]; bgl[a.C + threadIdx.x] = a.g_out[threadIdx.x]; bgl[2 * a.C + threadIdx.x] = a.g[threadIdx.x];
    }
    __syncthreads();
    const uint4* wq = (const uint4*)a.wqkv;
    const unsigned* wqs = (const unsigned*)(wq + (size_t)(3 * hd) * NC * 3 * 64);
    const uint4* wo = (const uint4*)a.wout;
    const unsigned* wos = (const unsigned*)(wo + (size_t)NT2 * 4 * 3 * 64);
    const int piece = tid & 3;

    f16x8 cth[2], ctl[2];                                      // ctx^T fragments x hw (see la_proj_kernel)
    {
        const float* cb = a.ctx + ((size_t)n * hd + head) * 1024;
        const float fhw = (float)a.hw;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float cv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cv[j] = cb[(16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) * 32 + r] * fhw;
            split_frag(cv, cth[s], ctl[s]);
        }
    }
    const float inv_hw = 1.f / (float)a.hw, scale = 0.17677669529663688110f;

    // staging of one 32-channel chunk of LN(x) for the q projection (as la_proj_kernel; one pixel piece per thread);
    // prefetch distance 2 with two register sets, weights ahead of the activations in the queue (see la_proj_kernel)
    struct Raw { float4 v[2]; float2 s; bool ok; };
    auto prefetch = [&](long long t, int c, Raw& R) __attribute__((always_inline)) {
        float4 (&rv)[2] = R.v; float2& rs = R.s; bool& rok = R.ok;
        const int cb = c * 32 + piece * 8;
        long long p = t * TP + (tid >> 2);
        rok = p < a.hw;
        p = rok ? p : a.hw - 1;                                // unconditional loads, clamped address, masked value
        const size_t gp = (size_t)n * a.hw + p;
        const float4* src = (const float4*)(a.x + gp * a.C + cb);
        rv[0] = src[0]; rv[1] = src[1];
        if (!SELF) rs = *(const float2*)(a.lnstats + gp * 2);
    };
    auto stage = [&](int buf, int c, const Raw& R) __attribute__((always_inline)) {
        const float4 (&rv)[2] = R.v; const float2& rs = R.s; const bool rok = R.ok;
        const float* gl = bgl + 2 * a.C + c * 32 + piece * 8;      // PreNorm gain of this piece's 8 channels
        const float4 rg[2] = {*(const float4*)gl, *(const float4*)(gl + 4)};
        const float lr = rok ? rs.x : 0.f, lm = rok ? rs.y : 0.f;
        float v[8] = {rg[0].x * (rv[0].x * lr - lm), rg[0].y * (rv[0].y * lr - lm), rg[0].z * (rv[0].z * lr - lm), rg[0].w * (rv[0].w * lr - lm),
                      rg[1].x * (rv[1].x * lr - lm), rg[1].y * (rv[1].y * lr - lm), rg[1].z * (rv[1].z * lr - lm), rg[1].w * (rv[1].w * lr - lm)};
        if (!rok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        float lo[8], m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
        unsigned e_v4, e_lo;
        f4_scales(quad_max(m), e_v4, e_lo);
        const uint4 hi = split8_f16_res_sat(v, lo);
        char* rec = smem + buf * STG + (tid >> 2) * PS;
        *(uint4*)(rec + piece * 16) = hi;
        *(unsigned*)(rec + 64 + piece * 4) = cvt8_fp4(lo, e_lo);
        *(unsigned*)(rec + 80 + piece * 4) = cvt8_fp4(v, e_v4);
        if (piece == 0) *(unsigned short*)(rec + 96) = (unsigned short)(e_lo | (e_v4 << 8));
    };

    // tail roles: pixel group pm (32 px), cout tiles [ch0, ch0 + NTH)
    const int pm = head & 1, ch0 = (head >> 1) * NTH;
    // to_out fragments stay in registers across tiles when a wave owns a single cout tile (C = 64)
    uint4 wfr[NTH == 1 ? 4 : 1][3]; unsigned wfs[NTH == 1 ? 4 : 1];
    if (NTH == 1) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
            for (int p = 0; p < 3; ++p) wfr[c4][p] = wo[(((size_t)ch0 * 4 + c4) * 3 + p) * 64 + lane];
            wfs[c4] = wos[((size_t)ch0 * 4 + c4) * 64 + lane];
        }
    }

    uint4 bq[3]; unsigned bs;
    auto load_w = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[p] = wq[(((size_t)head * NC + c) * 3 + p) * 64 + lane];
        bs = wqs[((size_t)head * NC + c) * 64 + lane];
    };
    Raw R0, R1;
    static_assert(NC % 2 == 0, "two register sets alternate over the chunks of a tile");
    // C = 64 (two chunks): the q weights of both chunks stay in registers for the whole workgroup and both chunks of a tile are
    // staged in front of ONE barrier (three barriers per tile instead of four, twelve MFMAs back to back, the next tile's
    // activations in flight for a whole tile period)
    constexpr bool ONE = NC == 2;
    uint4 bq1[3] = {}; unsigned bs1 = 0;
    if (ONE) {
        load_w(1);
#pragma unroll
        for (int p = 0; p < 3; ++p) bq1[p] = bq[p];
        bs1 = bs;
    }
    if (t_beg < t_end) { prefetch(t_beg, 0, R0); load_w(0); prefetch(t_beg, 1, R1); }   // all VMEM unconditional: see la_proj_kernel
    constexpr bool XPF = NTH == 1;                             // residual input of phase B fetched a whole tile early
    for (long long t = t_beg; t < t_end; ++t) {
        const long long p0 = t * TP;
        float4 xb[XPF ? NTH : 1][4];
        if (XPF) {
            long long p = p0 + pm * 32 + r;
            p = p < a.hw ? p : a.hw - 1;                       // unconditional loads, clamped address
            const float* xp = a.x + ((size_t)n * a.hw + p) * a.C;
#pragma unroll
            for (int k = 0; k < NTH; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g) xb[k][g] = *(const float4*)(xp + (ch0 + k) * 32 + 8 * g + 4 * h);
        }
        // ---- phase A: q^T[d][pixel] for this wave's head ------------------------------------------------
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        auto step = [&](int c, int buf, Raw& R) __attribute__((always_inline)) {
            stage(buf, c, R);
            __syncthreads();
            const char* sb = smem + buf * STG;
            const f16x8 b0 = __builtin_bit_cast(f16x8, bq[0]), b1 = __builtin_bit_cast(f16x8, bq[1]);
            i32x8 b4 = {};
            b4[0] = bq[2].x; b4[1] = bq[2].y; b4[2] = bq[2].z; b4[3] = bq[2].w;
            const int bsv = (int)bs;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const char* rec = sb + (mt * 32 + r) * PS;
                const f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + h * 16));
                const f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + 32 + h * 16));
                const uint4 aq = *(const uint4*)(rec + 64 + h * 16);
                const int as = *(const unsigned char*)(rec + 96 + h);
                i32x8 a4 = {};
                a4[0] = aq.x; a4[1] = aq.y; a4[2] = aq.z; a4[3] = aq.w;
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b4, a4, acc[mt], 4, 4, 0, bsv, 0, as);
            }
            // next chunk's weights first, then the activations two chunks ahead into the set just consumed
            load_w(c + 1 < NC ? c + 1 : 0);
            const bool wrap = c + 2 >= NC;
            const long long tn = wrap ? (t + 1 < t_end ? t + 1 : t) : t;       // past the last tile: an unused re-load of it
            prefetch(tn, wrap ? c + 2 - NC : c + 2, R);
        };
        if constexpr (ONE) {
            if constexpr (SELF) { R0.s = quad_ln_stats(R0.v, R1.v); R1.s = R0.s; }
            stage(0, 0, R0);
            stage(1, 1, R1);
            const long long tn = t + 1 < t_end ? t + 1 : t;    // past the last tile: an unused re-load of it
            prefetch(tn, 0, R0);
            prefetch(tn, 1, R1);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const char* sb = smem + c * STG;
                const uint4* wb = c ? bq1 : bq;
                const f16x8 b0 = __builtin_bit_cast(f16x8, wb[0]), b1 = __builtin_bit_cast(f16x8, wb[1]);
                i32x8 b4 = {};
                b4[0] = wb[2].x; b4[1] = wb[2].y; b4[2] = wb[2].z; b4[3] = wb[2].w;
                const int bsv = (int)(c ? bs1 : bs);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* rec = sb + (mt * 32 + r) * PS;
                    const f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + h * 16));
                    const f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + 32 + h * 16));
                    const uint4 aq = *(const uint4*)(rec + 64 + h * 16);
                    const int as = *(const unsigned char*)(rec + 96 + h);
                    i32x8 a4 = {};
                    a4[0] = aq.x; a4[1] = aq.y; a4[2] = aq.z; a4[3] = aq.w;
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b4, a4, acc[mt], 4, 4, 0, bsv, 0, as);
                }
            }
        } else {
#pragma unroll 1
            for (int c = 0; c < NC; c += 2) { step(c, 0, R0); step(c + 1, 1, R1); }
        }
        // softmax over d (lane-local + one exchange), o^T = ctx^T q', encoded as this head's 32-channel record
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float m = acc[mt][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[mt][i]);
            m = fmaxf(m, __shfl_xor(m, 32));
            // exp(q - m) = exp2(q log2e - m log2e): one FMA + v_exp_f32; the softmax normaliser and 32^-0.5 are per-pixel
            // scalars and the pixel sits on the lane of the o accumulator, so they multiply the 16 outputs (together with
            // 1/hw) instead of the 32 probabilities
            const float ml = m * 1.44269504088896340736f;
            float q[16], sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { q[i] = __builtin_amdgcn_exp2f(fmaf(acc[mt][i], 1.44269504088896340736f, -ml)); sum += q[i]; }
            sum += __shfl_xor(sum, 32);
            const float f = scale * inv_hw / sum;
            f32x16 oacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // probabilities (in [0, 1]) as ONE fp16 value: the rounding is independent across the 32 d of the sum; the
                // context keeps hi + lo (it is one matrix per image and head, not noise)
                f16x8 qh;
#pragma unroll
                for (int j = 0; j < 8; ++j) qh[j] = (_Float16)q[8 * s + j];
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cth[s], qh, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ctl[s], qh, oacc, 0, 0, 0);
            }
            // lane (pixel r, half h) holds o[e], e = 8 g + 4 h + c: four groups of 4 consecutive channels
            float ov[16], om = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ov[i] = oacc[i] * f; om = fmaxf(om, fabsf(ov[i])); }
            om = fmaxf(om, __shfl_xor(om, 32));
            unsigned e_v4, e_lo;
            f4_scales(om, e_v4, e_lo);
            const float s_v4 = __builtin_bit_cast(float, e_v4 << 23), s_lo = __builtin_bit_cast(float, e_lo << 23);
            char* rec = orec + (head * TP + mt * 32 + r) * PS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = 8 * g + 4 * h;
                float lo[4]; unsigned hw2[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float va = ov[4 * g + 2 * c], vb = ov[4 * g + 2 * c + 1];      // conversions saturate (fp16_saturate_on)
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw2[c]) : "v"(va), "v"(vb));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * c]) : "v"(hw2[c]), "v"(ov[4 * g + 2 * c]));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * c + 1]) : "v"(hw2[c]), "v"(ov[4 * g + 2 * c + 1]));
                }
                *(uint2*)(rec + e0 * 2) = make_uint2(hw2[0], hw2[1]);
                unsigned ul = 0, uv = 0;
                ul = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ul, lo[0], lo[1], s_lo, 0);
                ul = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ul, lo[2], lo[3], s_lo, 1);
                uv = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(uv, ov[4 * g], ov[4 * g + 1], s_v4, 0);
                uv = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(uv, ov[4 * g + 2], ov[4 * g + 3], s_v4, 1);
                *(unsigned short*)(rec + 64 + e0 / 2) = (unsigned short)ul;
                *(unsigned short*)(rec + 80 + e0 / 2) = (unsigned short)uv;
            }
            if (h == 0) *(unsigned short*)(rec + 96) = (unsigned short)(e_lo | (e_v4 << 8));
        }
        __syncthreads();                                       // o records of the tile complete

        // ---- phase B: z^T[cout][pixel] = W_out o^T, bias, channel LayerNorm, gain, residual ----------
        f32x16 z[NTH];
#pragma unroll
        for (int k = 0; k < NTH; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) z[k][i] = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const char* rec = orec + (c4 * TP + pm * 32 + r) * PS;
            const f16x8 b0 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + h * 16));
            const f16x8 b1 = __builtin_bit_cast(f16x8, *(const uint4*)(rec + 32 + h * 16));
            const uint4 bqq = *(const uint4*)(rec + 64 + h * 16);
            const int bsx = *(const unsigned char*)(rec + 96 + h);
            i32x8 b4 = {};
            b4[0] = bqq.x; b4[1] = bqq.y; b4[2] = bqq.z; b4[3] = bqq.w;
#pragma unroll
            for (int k = 0; k < NTH; ++k) {
                uint4 w0, w1, w2; unsigned ws;
                if (NTH == 1) { w0 = wfr[c4][0]; w1 = wfr[c4][1]; w2 = wfr[c4][2]; ws = wfs[c4]; }
                else {
                    const size_t fb = ((size_t)(ch0 + k) * 4 + c4);
                    w0 = wo[(fb * 3 + 0) * 64 + lane]; w1 = wo[(fb * 3 + 1) * 64 + lane]; w2 = wo[(fb * 3 + 2) * 64 + lane];
                    ws = wos[fb * 64 + lane];
                }
                i32x8 w4 = {};
                w4[0] = w2.x; w4[1] = w2.y; w4[2] = w2.z; w4[3] = w2.w;
                z[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), b0, z[k], 0, 0, 0);
                z[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w1), b1, z[k], 0, 0, 0);
                z[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w4, b4, z[k], 4, 4, 0, (int)ws, 0, bsx);
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NTH; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(bgl + (ch0 + k) * 32 + 8 * g + 4 * h);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = z[k][4 * g + c] + bb[c];
                    z[k][4 * g + c] = v;
                    s1 += v; s2 = fmaf(v, v, s2);
                }
            }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (h == 0) lnx[(pm * 32 + r) * 2 + (head >> 1)] = make_float2(s1, s2);
        __syncthreads();
        {
            const float2 u0 = lnx[(pm * 32 + r) * 2], u1 = lnx[(pm * 32 + r) * 2 + 1];
            const float invc = 1.f / (float)a.C;
            const float mean = (u0.x + u1.x) * invc;
            const float var = fmaxf((u0.y + u1.y) * invc - mean * mean, 0.f);
            const float rstd = rsqrtf(var + 1e-5f);
            const long long p = p0 + pm * 32 + r;
            if (!RAGGED || p < a.hw) {
                const size_t pb = ((size_t)n * a.hw + p) * a.C;
#pragma unroll
                for (int k = 0; k < NTH; ++k)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = (ch0 + k) * 32 + 8 * g + 4 * h;
                        const float4 gv = *(const float4*)(bgl + a.C + c0);
                        float4 xv;
                        if (XPF) xv = xb[k][g]; else xv = *(const float4*)(a.x + pb + c0);
                        float4 ov4;
                        ov4.x = (z[k][4 * g] - mean) * rstd * gv.x + xv.x;
                        ov4.y = (z[k][4 * g + 1] - mean) * rstd * gv.y + xv.y;
                        ov4.z = (z[k][4 * g + 2] - mean) * rstd * gv.z + xv.z;
                        ov4.w = (z[k][4 * g + 3] - mean) * rstd * gv.w + xv.w;
                        *(float4*)(a.out + pb + c0) = ov4;
                    }
            }
        }
    }
}

// merge of the ONLINE context partials [B][S][heads][1024 ctx | 32 z | 32 m]:
//   M[d] = max_s m_s[d];  ctx[d][e] = sum_s ctx_s[d][e] w_s[d] / (hw * sum_s z_s[d] w_s[d]),  w_s[d] = exp(m_s[d] - M[d])
__global__ __launch_bounds__(256) void la_ctx_merge_kernel(const float* __restrict__ part, int heads, int S, long long hw,
                                                           float* __restrict__ ctx) {
    // 16 blocks per (sample, head), 64 context entries each; the four waves of a block take every fourth partial (the loops are
    // latency-bound: S / 4 dependent-free loads per lane instead of S) and are combined through LDS in a fixed order
    __shared__ float sm[4][64], sc[4][64], sz[4][64];
    const int bh = blockIdx.x >> 4, b = bh / heads, head = bh - b * heads;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = (blockIdx.x & 15) * 64 + lane, d = i >> 5;
    const size_t stride = (size_t)heads * 1088;
    const float* base = part + ((size_t)b * S * heads + head) * 1088;
    float M = -INFINITY;
#pragma unroll 8
    for (int s = wv; s < S; s += 4) M = fmaxf(M, base[s * stride + 1056 + d]);
    sm[wv][lane] = M;
    __syncthreads();
    M = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
    float c = 0.f, z = 0.f;
#pragma unroll 8
    for (int s = wv; s < S; s += 4) {
        const float* cp = base + s * stride;
        const float w = __expf(cp[1056 + d] - M);
        c = fmaf(cp[i], w, c);
        z = fmaf(cp[1024 + d], w, z);
    }
    sc[wv][lane] = c; sz[wv][lane] = z;
    __syncthreads();
    if (wv == 0) {
        c = (sc[0][lane] + sc[1][lane]) + (sc[2][lane] + sc[3][lane]);
        z = (sz[0][lane] + sz[1][lane]) + (sz[2][lane] + sz[3][lane]);
        ctx[(size_t)bh * 1024 + i] = c / (z * (float)hw);
    }
}

// Tiles per workgroup / workgroups per image depend on the image size ONLY (never on B), so the
// partial-sum grouping -- and with it every output bit -- is independent of the batch composition.
struct LaSplit { int tpw, S; };
template <int EPI>
LaSplit la_split(long long hw) {
    constexpr int TP = LaTile<EPI>::MT * 32;
    const long long tiles = (hw + TP - 1) / TP;
    long long tpw = tiles / 64;                 // <= 64 workgroups per image ...
    if (tpw < 1) tpw = 1;
    if (tpw > 1024 / TP) tpw = 1024 / TP;       // ... of at most 1024 pixels each
    return LaSplit{(int)tpw, (int)((tiles + tpw - 1) / tpw)};
}

int g_la_ctx_one = 1;     // tedm_set_tuning("la_ctx_one", 0|1): single-barrier context sweep at C = 64 (f16f4)

template <int EPI>
int launch_la(const LaArgs& a, int prec, hipStream_t st) {
    dim3 grid(a.B * a.S), block(256);
    constexpr int TP = LaTile<EPI>::MT * 32;
    const size_t gl = (size_t)a.C * sizeof(float);                        // PreNorm gain
    const size_t lds1 = 2 * (size_t)TP * pstride(prec) + gl, lds2 = 4 * (size_t)TP * pstride(prec) + gl;   // one / two precision planes, double buffered
    if constexpr (EPI == LA_CTX) {
        if (prec == TEDM_PREC_F16F4 && a.C == 2 * KC && (g_la_ctx_one || !a.lnstats)) {
            if (a.lnstats) hipLaunchKernelGGL((la_proj_kernel<LA_CTX, TEDM_PREC_F16F4, 1>), grid, block, 2 * lds1 - gl, st, a);
            else hipLaunchKernelGGL((la_proj_kernel<LA_CTX, TEDM_PREC_F16F4, 2>), grid, block, 2 * lds1 - gl, st, a);
            return LAUNCH_OK();
        }
    }
    if (prec == TEDM_PREC_F16F4) hipLaunchKernelGGL((la_proj_kernel<EPI, TEDM_PREC_F16F4>), grid, block, lds1, st, a);
    else if (prec == TEDM_PREC_BF16) hipLaunchKernelGGL((la_proj_kernel<EPI, TEDM_PREC_BF16>), grid, block, lds1, st, a);
    else if (prec == TEDM_PREC_FP16X2) hipLaunchKernelGGL((la_proj_kernel<EPI, TEDM_PREC_FP16X2>), grid, block, lds1, st, a);
    else hipLaunchKernelGGL((la_proj_kernel<EPI, TEDM_PREC_BF16X3>), grid, block, lds2, st, a);
    return LAUNCH_OK();
}

}  // namespace

// sweep: 0 = KMAX, 1 = CTX, 2 = QOUT.  Requires heads == 4 (one wave per head, 256 threads) and C % 32 == 0.
int op_linattn_fused(int sweep, const float* x, int C, const float* g, const float* lnstats, const void* wpk, int B,
                     long long hw, int heads, float* kmax_part, const float* kmax, float* ctx_part, const float* ctx,
                     float* o, int prec, hipStream_t st) {
    if (heads != 4 || C % KC) return TEDM_EINVAL;
    if (!lnstats && !la_self_ln_supported(C, heads, prec, sweep)) return TEDM_EINVAL;
    LaArgs a{};
    a.x = x; a.C = C; a.g = g; a.lnstats = lnstats; a.wpk = wpk; a.B = B; a.hw = hw; a.heads = heads;
    a.kmax_part = kmax_part; a.kmax = kmax; a.ctx_part = ctx_part; a.ctx = ctx; a.o = o;
    const LaSplit sp = sweep == LA_KMAX ? la_split<LA_KMAX>(hw) : sweep == LA_CTX ? la_split<LA_CTX>(hw) : la_split<LA_QOUT>(hw);
    a.tpw = sp.tpw; a.S = sp.S;
    if (sweep == LA_KMAX) return launch_la<LA_KMAX>(a, prec, st);
    if (sweep == LA_CTX) return launch_la<LA_CTX>(a, prec, st);
    return launch_la<LA_QOUT>(a, prec, st);
}

// The context sweep and the fused tail can take the PreNorm statistics from their own registers (lnstats == nullptr)
bool la_self_ln_supported(int C, int heads, int prec, int sweep) {
    return prec == TEDM_PREC_F16F4 && heads == 4 && C == 64 && (sweep == LA_CTX || sweep == -1 /* fused tail */);
}
int la_set_tuning(const char* key, int v) {
    if (!strcmp(key, "la_ctx_one") && (v == 0 || v == 1)) { g_la_ctx_one = v; return TEDM_OK; }
    return TEDM_EINVAL;
}

// The fused q sweep + to_out + LayerNorm + residual (F16F4 weight packs; heads == 4, C in {64, 128, 256}).
bool la_tail_supported(int C, int heads, int prec) { return prec == TEDM_PREC_F16F4 && heads == 4 && (C == 64 || C == 128 || C == 256); }
int op_linattn_qout_tail(const float* x, int C, const float* g, const float* lnstats, const void* wqkv, const float* ctx,
                         const void* wout, const float* bias, const float* g_out, int B, long long hw, int heads, float* out,
                         hipStream_t st) {
    if (!la_tail_supported(C, heads, TEDM_PREC_F16F4) || !bias || !g_out) return TEDM_EINVAL;
    LaTailArgs a{};
    a.x = x; a.C = C; a.g = g; a.lnstats = lnstats; a.wqkv = wqkv; a.ctx = ctx; a.wout = wout; a.bias = bias; a.g_out = g_out;
    a.out = out; a.B = B; a.hw = hw; a.heads = heads;
    const long long tiles = (hw + 63) / 64;
    long long tpw = tiles / 64;                 // <= 64 workgroups per image, of at most 1024 pixels each (as la_split)
    if (tpw < 1) tpw = 1;
    if (tpw > 16) tpw = 16;
    a.tpw = (int)tpw; a.S = (int)((tiles + tpw - 1) / tpw);
    const size_t lds = 2 * 64 * 112 + 64 * 4 * 112 + 64 * 2 * sizeof(float2) + 3 * (size_t)C * sizeof(float);
    dim3 grid(B * a.S), block(256);
    if (!lnstats) {
        if (C != 64) return TEDM_EINVAL;
        if (hw % 64) hipLaunchKernelGGL((la_qout_tail_kernel<2, true, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((la_qout_tail_kernel<2, false, true>), grid, block, lds, st, a);
    } else if (hw % 64) {
        if (C == 64) hipLaunchKernelGGL((la_qout_tail_kernel<2, true>), grid, block, lds, st, a);
        else if (C == 128) hipLaunchKernelGGL((la_qout_tail_kernel<4, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((la_qout_tail_kernel<8, true>), grid, block, lds, st, a);
    } else {
        if (C == 64) hipLaunchKernelGGL((la_qout_tail_kernel<2, false>), grid, block, lds, st, a);
        else if (C == 128) hipLaunchKernelGGL((la_qout_tail_kernel<4, false>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((la_qout_tail_kernel<8, false>), grid, block, lds, st, a);
    }
    return LAUNCH_OK();
}

int op_linattn_ctx_merge(const float* partial, int B, int heads, int S, long long hw, float* ctx, hipStream_t st) {
    hipLaunchKernelGGL(la_ctx_merge_kernel, dim3(B * heads * 16), dim3(256), 0, st, partial, heads, S, hw, ctx);
    return LAUNCH_OK();
}

int la_fused_wgs_per_image(int sweep, long long hw) {
    return sweep == LA_KMAX ? la_split<LA_KMAX>(hw).S : sweep == LA_CTX ? la_split<LA_CTX>(hw).S : la_split<LA_QOUT>(hw).S;
}
