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

constexpr int KS = 2, KC = 16 * KS;             // channels per staged chunk
// LDS bytes per pixel per plane (odd multiples of 16): bf16 / fp16 planes 80; F16F4 one 112-byte record
// [fp16 x32 | fp4 residual x32 | fp4 copy x32 | 2 scale bytes | pad] exactly as conv_igemm.inc stages it
constexpr int pstride(int prec) { return prec == TEDM_PREC_F16F4 ? 112 : KS * 32 + 16; }

template <int EPI> struct LaTile { static constexpr int MT = EPI == LA_CTX ? 2 : 4; };   // M-tiles (32 px) per tile

template <int EPI, int PREC>
__global__ __launch_bounds__(256, 2) void la_proj_kernel(LaArgs a) {
    constexpr int MT = LaTile<EPI>::MT, TP = MT * 32, NIT = MT / 2;   // TP*4 pieces / 256 threads
    constexpr bool F4 = PREC == TEDM_PREC_F16F4;
    constexpr int PSTRIDE = pstride(PREC);
    constexpr int REGION = TP * PSTRIDE;
    constexpr int NT = EPI == LA_CTX ? 2 : 1;                    // weight tiles per wave (k [, v])
    constexpr int NPL = PREC == TEDM_PREC_BF16X3 ? 2 : 1;        // activation planes in LDS
    constexpr int NPB = F4 ? 3 : (PREC == TEDM_PREC_BF16 ? 1 : 2);   // weight fragments per (tile, chunk) [F4] / planes per 16-channel step
    constexpr int BUFSZ = REGION * NPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];

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
    float actx[16];
    if (EPI == LA_QOUT) {
        const float* cb = a.ctx + ((size_t)n * hd + head) * 1024;
#pragma unroll
        for (int s = 0; s < 16; ++s) actx[s] = cb[acc_row(s, h) * 32 + r];
    }

    const int piece = tid & 3;                                    // 8-channel piece of the 32-channel chunk
    const int niter = (int)(t_end - t_beg) * NC;                  // flattened (tile, chunk) iterations

    // software pipeline: raw activations of iteration it+1 are in flight while iteration it runs its MFMAs
    float4 rv[NIT][2]; float2 rs[NIT]; float4 rg[2]; bool rok[NIT];
    // NB (measured on gfx950 / ROCm 7.2): with these loads inside `if (pixel valid)` blocks, lanes 48-63
    // staged stale registers whenever more than one workgroup shared a CU -- the counted s_waitcnt
    // around lane-divergent VMEM blocks was not sufficient.  Keep every VMEM instruction unconditional.
    auto prefetch = [&](int it) {
        const long long t = t_beg + it / NC;
        const int cb = (it % NC) * KC + piece * 8;
        rg[0] = *(const float4*)(a.g + cb); rg[1] = *(const float4*)(a.g + cb + 4);
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
            rs[k] = *(const float2*)(a.lnstats + gp * 2);
        }
    };
    auto stage = [&](int buf) {
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
                const uint4 hi = split8_f16_res(v, lo);
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
    if (niter > 0) prefetch(0);
    for (int it = 0; it < niter; ++it) {
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
        // weights of this chunk (L2 hits): in flight across the staging + barrier below
        uint4 bq[KS][NT][NPB];
        unsigned bsc[NT];
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
        stage(buf);
        __syncthreads();          // buffer `buf` complete; every wave has finished the MFMAs of iteration it-1
        if (it + 1 < niter) prefetch(it + 1);
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
        if (c != NC - 1) continue;

        // ---- epilogue of the tile -------------------------------------------------------------
        if (EPI == LA_KMAX) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (p0 + mt * 32 + acc_row(i, h) < a.hw) run_max = fmaxf(run_max, acc[mt][0][i]);
        } else if (EPI == LA_CTX) {
            if (online) {
                // running max of this lane's k channel over the workgroup's pixels; rescale what has been accumulated
                float tmax = -INFINITY;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (p0 + mt * 32 + acc_row(i, h) < a.hw) tmax = fmaxf(tmax, acc[mt][0][i]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(kmx, tmax);
                const float f = (m_new == kmx) ? 1.f : __expf(kmx - m_new);     // kmx = -inf on the first tile -> 0 (accumulators are 0)
                zsum *= f;
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) cacc[s_] *= __shfl(f, acc_row(s_, h));   // row d of ctx lives on lane d
                kmx = m_new;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool ok = p0 + mt * 32 + acc_row(i, h) < a.hw;
                    const float e = ok ? __expf(acc[mt][0][i] - kmx) : 0.f;
                    zsum += e;
                    cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(e, ok ? acc[mt][NT - 1][i] : 0.f, cacc, 0, 0, 0);
                }
        } else {
            const float scale = 0.17677669529663688110f;         // 32^-0.5
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
                for (int s = 0; s < 16; ++s) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(actx[s], q[s] * f, oacc, 0, 0, 0);
                const long long p = p0 + mt * 32 + r;
                if (p < a.hw) {
                    float* op = a.o + ((size_t)n * a.hw + p) * (hd * 32) + head * 32 + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *(float4*)(op + 8 * gq) = make_float4(oacc[4 * gq], oacc[4 * gq + 1], oacc[4 * gq + 2], oacc[4 * gq + 3]);
                }
            }
        }
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

// merge of the ONLINE context partials [B][S][heads][1024 ctx | 32 z | 32 m]:
//   M[d] = max_s m_s[d];  ctx[d][e] = sum_s ctx_s[d][e] w_s[d] / (hw * sum_s z_s[d] w_s[d]),  w_s[d] = exp(m_s[d] - M[d])
__global__ __launch_bounds__(256) void la_ctx_merge_kernel(const float* __restrict__ part, int heads, int S, long long hw,
                                                           float* __restrict__ ctx) {
    const int bh = blockIdx.x >> 2, b = bh / heads, head = bh - b * heads;      // 4 blocks per (sample, head)
    const int i = (blockIdx.x & 3) * 256 + threadIdx.x, d = i >> 5;
    const size_t stride = (size_t)heads * 1088;
    const float* base = part + ((size_t)b * S * heads + head) * 1088;
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, base[s * stride + 1056 + d]);
    float c = 0.f, z = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* cp = base + s * stride;
        const float w = __expf(cp[1056 + d] - M);
        c = fmaf(cp[i], w, c);
        z = fmaf(cp[1024 + d], w, z);
    }
    ctx[(size_t)bh * 1024 + i] = c / (z * (float)hw);
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

template <int EPI>
int launch_la(const LaArgs& a, int prec, hipStream_t st) {
    dim3 grid(a.B * a.S), block(256);
    constexpr int TP = LaTile<EPI>::MT * 32;
    const size_t lds1 = 2 * (size_t)TP * pstride(prec), lds2 = 2 * lds1;   // one / two precision planes, double buffered
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
    LaArgs a{};
    a.x = x; a.C = C; a.g = g; a.lnstats = lnstats; a.wpk = wpk; a.B = B; a.hw = hw; a.heads = heads;
    a.kmax_part = kmax_part; a.kmax = kmax; a.ctx_part = ctx_part; a.ctx = ctx; a.o = o;
    const LaSplit sp = sweep == LA_KMAX ? la_split<LA_KMAX>(hw) : sweep == LA_CTX ? la_split<LA_CTX>(hw) : la_split<LA_QOUT>(hw);
    a.tpw = sp.tpw; a.S = sp.S;
    if (sweep == LA_KMAX) return launch_la<LA_KMAX>(a, prec, st);
    if (sweep == LA_CTX) return launch_la<LA_CTX>(a, prec, st);
    return launch_la<LA_QOUT>(a, prec, st);
}

int op_linattn_ctx_merge(const float* partial, int B, int heads, int S, long long hw, float* ctx, hipStream_t st) {
    hipLaunchKernelGGL(la_ctx_merge_kernel, dim3(B * heads * 4), dim3(256), 0, st, partial, heads, S, hw, ctx);
    return LAUNCH_OK();
}

int la_fused_wgs_per_image(int sweep, long long hw) {
    return sweep == LA_KMAX ? la_split<LA_KMAX>(hw).S : sweep == LA_CTX ? la_split<LA_CTX>(hw).S : la_split<LA_QOUT>(hw).S;
}
