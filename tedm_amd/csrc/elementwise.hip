// elementwise.hip -- the HBM-bound fp32 kernels around the MFMA convolutions (all NHWC):
//   q-sample, 7x7 stem, time-embedding MLP + fused FiLM projections, GroupNorm finalize,
//   resnet-block output (GN-apply + SiLU + residual [+ channel-LayerNorm stats]), channel
//   LayerNorm (+ residual), final 1x1 conv, per-channel norm / max finalizers, tap -> features.
// Per-pixel reductions use wave64 xor-shuffles over the lanes that share a pixel.
#include "common.h"
#include <stdlib.h>
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

// ---------------------------------------------------------------------------------------
// q-sample: models/diffusion_model.py:176-203 (+ :169-170 normalisation)
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const long long* __restrict__ t, const float* __restrict__ ca,
                                const float* __restrict__ cb, int T, float* __restrict__ xt,
                                long long chw, long long total, int normalize) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / chw);
        long long ti = t[b];
        ti = ti < 0 ? 0 : (ti >= T ? T - 1 : ti);
        float x = x0[i];
        if (normalize) x = x * 2.f - 1.f;
        xt[i] = ca[ti] * x + cb[ti] * noise[i];
    }
}

// ---------------------------------------------------------------------------------------
// stem: Conv2d(channels, init_dim, 7, padding=3), models/unet_model.py:267,334.
// NCHW input (channels is 1 for TEDM) -> NHWC output.  One thread per output pixel, 64 output
// channels per pass in registers; weights [ci][tap][co] and the haloed input tile live in LDS
// (weight reads are wave-wide broadcasts).
constexpr int STEM_TR = 8, STEM_TC = 32, STEM_HR = STEM_TR + 6, STEM_HC = STEM_TC + 6;
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int Cin, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx = (float*)smem;                               // [Cin][HR][HC]
    float* sw = sx + ((Cin * STEM_HR * STEM_HC + 3) & ~3);  // [Cin*49][64]
    const int tid = threadIdx.x;
    const int tiles_x = (W + STEM_TC - 1) / STEM_TC, tiles_y = (H + STEM_TR - 1) / STEM_TR;
    int tile = blockIdx.x;
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * STEM_TR, ox0 = tx * STEM_TC;
    for (int i = tid; i < Cin * STEM_HR * STEM_HC; i += 256) {
        const int ci = i / (STEM_HR * STEM_HC);
        const int rem = i - ci * STEM_HR * STEM_HC;
        const int hy = rem / STEM_HC, hx = rem - hy * STEM_HC;
        const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
        sx[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)(n * Cin + ci) * H + iy) * W + ix] : 0.f;
    }
    const int py = tid / STEM_TC, px = tid % STEM_TC;
    const int oy = oy0 + py, ox = ox0 + px;
    for (int cb = 0; cb < Cout; cb += 64) {
        __syncthreads();
        for (int i = tid; i < Cin * 49 * 64; i += 256) {
            const int co = i & 63, k = i >> 6;              // k = ci*49 + tap
            sw[i] = (cb + co < Cout) ? w[(size_t)(cb + co) * Cin * 49 + k] : 0.f;
        }
        __syncthreads();
        float acc[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) acc[j] = 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const float xv = sx[(ci * STEM_HR + py + ky) * STEM_HC + px + kx];
                    const float4* wr = (const float4*)(sw + (size_t)((ci * 7 + ky) * 7 + kx) * 64);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float4 wv = wr[q];
                        acc[4 * q + 0] = fmaf(xv, wv.x, acc[4 * q + 0]);
                        acc[4 * q + 1] = fmaf(xv, wv.y, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(xv, wv.z, acc[4 * q + 2]);
                        acc[4 * q + 3] = fmaf(xv, wv.w, acc[4 * q + 3]);
                    }
                }
        if (oy < H && ox < W) {
            float* o = out + ((size_t)(n * H + oy) * W + ox) * Cout + cb;
            if (cb + 64 <= Cout && (Cout & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 bv = *(const float4*)(bias + cb + 4 * q);
                    *(float4*)(o + 4 * q) = make_float4(acc[4 * q] + bv.x, acc[4 * q + 1] + bv.y,
                                                        acc[4 * q + 2] + bv.z, acc[4 * q + 3] + bv.w);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 64; ++j)
                    if (cb + j < Cout) o[j] = acc[j] + bias[cb + j];
            }
        }
    }
}

// Matrix-core stem (Cin*49 <= 128 taps, Cout in {32, 64}): out[pixel][co] = sum_tap x[pixel + tap] * w[co][tap] as a GEMM on
// the exact-fp32 MFMA with i = pixel (one 32-pixel row segment per M tile), j = output channel, k = tap parity: every lane
// gathers its pixel's sample for its tap from the haloed LDS tile.  One wave = 2 rows x all output channels.
template <int NCB>
__global__ __launch_bounds__(256) void stem_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int Cin, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = Cin * 49, K2 = (K + 1) / 2;
    float* sx = (float*)smem;                               // [Cin][HR][HC]
    float* sw = sx + ((Cin * STEM_HR * STEM_HC + 3) & ~3);  // [2*K2][Cout]  (row K zero when K is odd)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int tiles_x = (W + STEM_TC - 1) / STEM_TC, tiles_y = (H + STEM_TR - 1) / STEM_TR;
    int tile = blockIdx.x;
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * STEM_TR, ox0 = tx * STEM_TC;
    for (int i = tid; i < Cin * STEM_HR * STEM_HC; i += 256) {
        const int ci = i / (STEM_HR * STEM_HC);
        const int rem = i - ci * STEM_HR * STEM_HC;
        const int hy = rem / STEM_HC, hx = rem - hy * STEM_HC;
        const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        const float v = x[((size_t)(n * Cin + ci) * H + iyc) * W + ixc];          // unconditional load, clamped address
        sx[i] = ok ? v : 0.f;
    }
    for (int i = tid; i < 2 * K2 * Cout; i += 256) {
        const int k = i / Cout, co = i - k * Cout;
        const float v = w[(size_t)co * K + (k < K ? k : 0)];
        sw[i] = k < K ? v : 0.f;
    }
    __syncthreads();
    f32x16 acc[2][NCB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][c][i] = 0.f;
    for (int kk = 0; kk < K2; ++kk) {
        const int t = 2 * kk + lk;
        const int tc = t < K ? t : 0;
        const int ci = tc / 49, tap = tc - ci * 49, ky = tap / 7, kx = tap - ky * 7;
        const float* ax = sx + (ci * STEM_HR + 2 * wave + ky) * STEM_HC + li + kx;
        float a0 = ax[0], a1 = ax[STEM_HC];
        if (t >= K) { a0 = 0.f; a1 = 0.f; }
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const float bv = sw[t * Cout + c * 32 + li];
            acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int oy = oy0 + 2 * wave + m;
        if (oy >= H) continue;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int co = c * 32 + li;
            const float bs = bias[co];
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                const int ox = ox0 + (s_ & 3) + 8 * (s_ >> 2) + 4 * lk;
                if (ox < W) out[((size_t)(n * H + oy) * W + ox) * Cout + co] = acc[m][c][s_] + bs;
            }
        }
    }
}


// The same GEMM on v_mfma_f32_32x32x16_f16 with fp16 hi + lo operands (hi*hi + lo*hi + hi*lo: ~2^-21, fp32-grade): the K axis
// is ordered (input channel, tap row pair, 8 slots) -- k-step s of channel ci holds tap rows ky = 2s (lane half 0) and 2s+1
// (half 1), slots j = 0..6 = kx, slot 7 and row 7 zero -- so a lane's 8 operand values are 7 consecutive floats of one halo
// row at compile-time offsets.  48 MFMA issues of 32 cycles per wave (2 rows x 64 couts) instead of 100 x 64 cycles of the
// fp32 MFMA: the kernel becomes bound by its 256-byte-per-pixel output stream.
TEDM_DEVFN void stem_split(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = (_Float16)v[j];
        hi[j] = hj;
        lo[j] = (_Float16)(v[j] - (float)hj);
    }
}
template <int NCB>
__global__ __launch_bounds__(256) void stem_conv_f16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int B, int Cin, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS4 = Cin * 4;                                // k-steps
    float* sx = (float*)smem;                               // [Cin][HR + 1][HC]  (one extra zero row: tap row 7)
    uint4* swf = (uint4*)(sx + ((Cin * (STEM_HR + 1) * STEM_HC + 3) & ~3));     // [k-step][cout tile][hi|lo][64 lanes] B fragments
    float* swe = (float*)(swf + (size_t)KS4 * NCB * 2 * 64);                    // [2][Cout]: 2^-e, 2^e of each output channel's weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    // Conversions saturate at +-65504 instead of producing inf (hi = inf would make lo = v - inf = NaN), and every output
    // channel's weights are pre-scaled by a power of two into [1, 2) so that small weights do not fall into fp16's subnormal
    // range and lose the lo term (both exact: the scale is undone on the fp32 accumulator).  ADVICE r2.
    // Once per WORKGROUP, which then walks several tiles: the per-channel maximum is a cooperative LDS integer max (the first
    // version had thread co loop over its 49 weights serially: 60 us on a 180 us kernel), and the fragment build is amortised.
    fp16_saturate_on();
    unsigned* smaxu = (unsigned*)swe;
    if (tid < Cout) smaxu[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < Cout * Cin * 49; i += 256)
        atomicMax(&smaxu[i / (Cin * 49)], __float_as_uint(w[i]) & 0x7fffffffu);            // |w| as an unsigned integer: order-preserving
    __syncthreads();
    if (tid < Cout) {
        const unsigned am = smaxu[tid];
        int e = (int)((am >> 23) & 0xff) - 127;
        e = am != 0u ? (e < -100 ? -100 : (e > 100 ? 100 : e)) : 0;
        swe[tid] = ldexpf(1.f, -e);                         // (overwrites this thread's own scratch word)
        swe[Cout + tid] = ldexpf(1.f, e);
    }
    __syncthreads();
    // weight fragments: lane (cout li, half hk) of k-step (ci, s) holds w[co][ci][ky = 2 s + hk][kx = j], j < 7
    for (int f = tid; f < KS4 * NCB * 64; f += 256) {
        const int ln = f & 63, c = (f >> 6) % NCB, ks = f / (64 * NCB);
        const int ci = ks >> 2, s_ = ks & 3, co = c * 32 + (ln & 31), ky = 2 * s_ + (ln >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (j < 7 && ky < 7 && co < Cout) ? w[((size_t)co * Cin + ci) * 49 + ky * 7 + j] * swe[co] : 0.f;
        f16x8 hi, lo;
        stem_split(v, hi, lo);
        swf[((ks * NCB + c) * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, hi);
        swf[((ks * NCB + c) * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, lo);
    }
    const int tiles_x = (W + STEM_TC - 1) / STEM_TC, tiles_y = (H + STEM_TR - 1) / STEM_TR;
    const int ntiles = B * tiles_x * tiles_y;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int n = tl / (tiles_x * tiles_y);
        const int trem = tl - n * tiles_x * tiles_y;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * STEM_TR, ox0 = tx * STEM_TC;
        for (int i = tid; i < Cin * (STEM_HR + 1) * STEM_HC; i += 256) {
            const int ci = i / ((STEM_HR + 1) * STEM_HC);
            const int rem = i - ci * (STEM_HR + 1) * STEM_HC;
            const int hy = rem / STEM_HC, hx = rem - hy * STEM_HC;
            const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
            const bool ok = hy < STEM_HR && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            const float v = x[((size_t)(n * Cin + ci) * H + iyc) * W + ixc];          // unconditional load, clamped address
            sx[i] = ok ? v : 0.f;
        }
        __syncthreads();                                     // sx (and, the first time, swf) complete
        f32x16 acc[2][NCB];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][c][i] = 0.f;
        for (int ks = 0; ks < KS4; ++ks) {
            const int ci = ks >> 2, s_ = ks & 3;
            f16x8 bh[NCB], bl[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                bh[c] = __builtin_bit_cast(f16x8, swf[((ks * NCB + c) * 2 + 0) * 64 + lane]);
                bl[c] = __builtin_bit_cast(f16x8, swf[((ks * NCB + c) * 2 + 1) * 64 + lane]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // halo row of output row (2 wave + m), tap row 2 s + lk; row STEM_HR (index 14) is the zero row (tap row 7 of the last rows)
                const float* ax = sx + (ci * (STEM_HR + 1) + 2 * wave + m + 2 * s_ + lk) * STEM_HC + li;
                float av[8];
#pragma unroll
                for (int j = 0; j < 7; ++j) av[j] = ax[j];
                av[7] = 0.f;
                if (2 * s_ + lk >= 7) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) av[j] = 0.f;
                }
                f16x8 ah, al;
                stem_split(av, ah, al);
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[c], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[c], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[c], acc[m][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                     // every wave is done with sx before the next tile overwrites it
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = oy0 + 2 * wave + m;
            if (oy >= H) continue;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int co = c * 32 + li;
                const float bs = bias[co], sc = swe[Cout + co];
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) {
                    const int ox = ox0 + (s_ & 3) + 8 * (s_ >> 2) + 4 * lk;
                    if (ox < W) out[((size_t)(n * H + oy) * W + ox) * Cout + co] = fmaf(acc[m][c][s_], sc, bs);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// time MLP: SinusoidalPosEmb -> Linear -> GELU(erf) -> Linear (models/unet_model.py:76-93,287-292)
// followed by the SiLU every ResnetBlock.time_mlp applies first (:149-152).  One block per sample.
// freqs[half] is bound by the host (computed exactly as the reference does, on the CPU).
__global__ __launch_bounds__(256) void temb_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   int dim, int tdim, int learned, float* __restrict__ st /*[B][tdim]*/,
                                                   float* __restrict__ save /*training tape or NULL*/) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* emb = (float*)smem;         // [dim]
    float* hid = emb + dim;            // [tdim]
    const int b = blockIdx.x, half = learned ? (dim - 1) / 2 : dim / 2, B = gridDim.x;
    // tape sections: emb [B][dim] | pre-GELU [B][tdim] | post-GELU [B][tdim] | pre-SiLU [B][tdim]
    float* s_emb = save ? save + (size_t)b * dim : nullptr;
    float* s_t1 = save ? save + (size_t)B * dim + (size_t)b * tdim : nullptr;
    float* s_hg = save ? s_t1 + (size_t)B * tdim : nullptr;
    float* s_t2 = save ? s_hg + (size_t)B * tdim : nullptr;
    const float tv = t[b];
    if (learned) {
        // LearnedSinusoidalPosEmb (models/unet_model.py:96-114): cat(t, sin(f), cos(f)), f = t * weights * 2 * pi evaluated left
        // to right in fp32 as torch does (python scalars multiply as fp32)
        if (threadIdx.x == 0) { emb[0] = tv; if (save) s_emb[0] = tv; }
        for (int i = threadIdx.x; i < half; i += blockDim.x) {
            const float a = __fmul_rn(__fmul_rn(__fmul_rn(tv, freqs[i]), 2.f), 3.14159265358979323846f);
            emb[1 + i] = sinf(a);
            emb[1 + half + i] = cosf(a);
            if (save) { s_emb[1 + i] = emb[1 + i]; s_emb[1 + half + i] = emb[1 + half + i]; }
        }
    } else
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float a = tv * freqs[i];
        emb[i] = sinf(a);
        emb[half + i] = cosf(a);
        if (save) { s_emb[i] = emb[i]; s_emb[half + i] = emb[half + i]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tdim; j += blockDim.x) {
        float s = b1[j];
        for (int k = 0; k < dim; ++k) s = fmaf(w1[(size_t)j * dim + k], emb[k], s);
        hid[j] = 0.5f * s * (1.f + erff(s * 0.70710678118654752440f));
        if (save) { s_t1[j] = s; s_hg[j] = hid[j]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tdim; j += blockDim.x) {
        float s = b2[j];
        for (int k = 0; k < tdim; ++k) s = fmaf(w2[(size_t)j * tdim + k], hid[k], s);
        st[(size_t)b * tdim + j] = silu_f(s);
        if (save) s_t2[j] = s;
    }
}

// All ResnetBlock.time_mlp Linear(tdim -> 2*Cout) projections as ONE skinny GEMM over the
// concatenated weight (J rows): film[b][j] = wf[j] . st[b] + bf[j].  One wave per output row.
__global__ __launch_bounds__(256) void film_kernel(const float* __restrict__ st, const float* __restrict__ wf,
                                                   const float* __restrict__ bf, int B, int tdim, int J,
                                                   float* __restrict__ film) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= J) return;
    const float bj = bf[j];
    // the weight row is read once (tdim <= 256: four values per lane), the batch loop runs four samples at a time so that their
    // loads and wave reductions overlap (one sample at a time left the kernel at 48 us for 8 MB of weights)
    float wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = lane + 64 * i < tdim ? wf[(size_t)j * tdim + lane + 64 * i] : 0.f;
    if (tdim <= 256) {
        for (int b0 = 0; b0 < B; b0 += 4) {
            float s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u < B ? b0 + u : B - 1;
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) t = fmaf(wr[i], lane + 64 * i < tdim ? st[(size_t)b * tdim + lane + 64 * i] : 0.f, t);
                s[u] = t;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = wave_sum(s[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane == 0 && b0 + u < B) film[(size_t)(b0 + u) * J + j] = s[u] + bj;
        }
        return;
    }
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int k = lane; k < tdim; k += 64) s = fmaf(wf[(size_t)j * tdim + k], st[(size_t)b * tdim + k], s);
        s = wave_sum(s);
        if (lane == 0) film[(size_t)b * J + j] = s + bj;
    }
}

// ---------------------------------------------------------------------------------------
// GroupNorm finalize (nn.GroupNorm(groups, C, eps=1e-5), models/unet_model.py:123,128):
// reduce the conv epilogue's per-channel (sum, sumsq) slab in fp64, then fold normalisation,
// affine and the optional FiLM (x*(scale+1)+shift, :132) into per-(sample, channel) a, b so the
// consumer computes silu(a*x + b).   One block per (sample, group).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ slab, int E, int C, int groups,
                                                          long long hw, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ film, int film_stride,
                                                          float* __restrict__ oa, float* __restrict__ ob,
                                                          float* __restrict__ mr_out /*[B][groups][2] or null*/) {
    __shared__ double r1[4], r2[4];
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cg = C / groups;
    double s1 = 0.0, s2 = 0.0;
    const float* base = slab + ((size_t)n * E * C + (size_t)g * cg) * 2;
    if ((cg & 1) == 0) {
        // an entry's cg (sum, sumsq) pairs are cg/2 consecutive float4; four independent loads in flight per lane
        const int q = cg >> 1, total = E * q;
        auto ld = [&](int i) { const int e = i / q, j = i - e * q; return *(const float4*)(base + (size_t)e * C * 2 + j * 4); };
        int i = threadIdx.x;
        for (; i + 768 < total; i += 1024) {
            const float4 v0 = ld(i), v1 = ld(i + 256), v2 = ld(i + 512), v3 = ld(i + 768);
            s1 += ((double)v0.x + (double)v0.z) + ((double)v1.x + (double)v1.z) + (((double)v2.x + (double)v2.z) + ((double)v3.x + (double)v3.z));
            s2 += ((double)v0.y + (double)v0.w) + ((double)v1.y + (double)v1.w) + (((double)v2.y + (double)v2.w) + ((double)v3.y + (double)v3.w));
        }
        for (; i < total; i += 256) { const float4 v = ld(i); s1 += (double)v.x + (double)v.z; s2 += (double)v.y + (double)v.w; }
    } else {
        const int total = E * cg;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int e = i / cg, c = i - e * cg;
            const float2 v = *(const float2*)(base + ((size_t)e * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    s1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    s2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
    const double cnt = (double)hw * cg;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float meanf = (float)mean;
    if (mr_out && threadIdx.x == 0) { mr_out[(size_t)blockIdx.x * 2] = meanf; mr_out[(size_t)blockIdx.x * 2 + 1] = rstd; }
    for (int i = threadIdx.x; i < cg; i += 256) {
        const int c = g * cg + i;
        float a = rstd * gamma[c];
        float b = beta[c] - meanf * a;
        if (film) {
            const float sc = film[(size_t)n * film_stride + c] + 1.f;
            const float sh = film[(size_t)n * film_stride + C + c];
            a *= sc;
            b = b * sc + sh;
        }
        oa[(size_t)n * C + c] = a;
        ob[(size_t)n * C + c] = b;
    }
}

// ---------------------------------------------------------------------------------------
// Pixel-vector kernels: a pixel's C channels are spread over LPP lanes (LPP power of two <= 64),
// V float4 per lane; 64/LPP pixels per wave.
struct PixMap { int lpp, v; };

template <int V>
__global__ __launch_bounds__(256) void resblock_out_kernel(const float* __restrict__ y, const float* __restrict__ ga,
                                                           const float* __restrict__ gb, const float* __restrict__ x,
                                                           float* __restrict__ out, float* __restrict__ lnstats,
                                                           long long P, long long hw, int C, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    // U pixel groups per iteration: all loads of both groups are issued before the first SiLU (twice the bytes in flight per wave)
    constexpr int U = V <= 2 ? 2 : 1;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw * U) {
        float4 yy[U][V], xx[U][V], aa[U][V], bb[U][V];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pu = p0 + u * nwaves * ppw + sub;
            ok[u] = pu < P;
            const long long p = ok[u] ? pu : P - 1;             // unconditional loads, clamped address
            const int n = (int)(p / hw);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int c = (v * lpp + q) * 4;
                yy[u][v] = *(const float4*)(y + (size_t)p * C + c);
                xx[u][v] = *(const float4*)(x + (size_t)p * C + c);
                aa[u][v] = *(const float4*)(ga + (size_t)n * C + c);
                bb[u][v] = *(const float4*)(gb + (size_t)n * C + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + u * nwaves * ppw + sub;
            float4 val[V];
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int c = (v * lpp + q) * 4;
                val[v].x = silu_f(fmaf(aa[u][v].x, yy[u][v].x, bb[u][v].x)) + xx[u][v].x;
                val[v].y = silu_f(fmaf(aa[u][v].y, yy[u][v].y, bb[u][v].y)) + xx[u][v].y;
                val[v].z = silu_f(fmaf(aa[u][v].z, yy[u][v].z, bb[u][v].z)) + xx[u][v].z;
                val[v].w = silu_f(fmaf(aa[u][v].w, yy[u][v].w, bb[u][v].w)) + xx[u][v].w;
                if (ok[u]) *(float4*)(out + (size_t)p * C + c) = val[v];
                else val[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                s += (val[v].x + val[v].y) + (val[v].z + val[v].w);
            }
            if (lnstats) {                                         // uniform branch
                const float mean = group_sum(s, lpp) / (float)C;
                float d2 = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float dx = val[v].x - mean, dy = val[v].y - mean, dz = val[v].z - mean, dw = val[v].w - mean;
                    d2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
                const float var = group_sum(d2, lpp) / (float)C;
                const float rstd = rsqrtf(var + 1e-5f);
                if (ok[u] && q == 0) *(float2*)(lnstats + (size_t)p * 2) = make_float2(rstd, mean * rstd);
            }
        }
    }
}

// The same on a Q16-stored y (common.h): a lane owns 8 consecutive channels of a pixel (16 bytes of int16 + 8 scale bytes), C / 8
// lanes (a power of two <= 64) share the pixel.  Decoding is exact, so the result equals resblock_out on the decoded tensor.
__global__ __launch_bounds__(256) void resblock_out_q16_kernel(const short* __restrict__ y, const unsigned char* __restrict__ ysc,
                                                               const float* __restrict__ ga, const float* __restrict__ gb,
                                                               const float* __restrict__ x, float* __restrict__ out,
                                                               float* __restrict__ lnstats, long long P, long long hw, int W, int C, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        const bool ok = p < P;
        const int n = ok ? (int)(p / hw) : 0;
        const int c = q * 8;
        float val[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float s = 0.f;
        if (ok) {
            float yy[8];
            const long long row = p / W;
            q16_decode8(*(const uint4*)(y + (size_t)p * C + c), *(const uint2*)(ysc + q16_scale_index(row, (int)(p - row * W), W, C) + c), yy);
            const float4 a0 = *(const float4*)(ga + (size_t)n * C + c), a1 = *(const float4*)(ga + (size_t)n * C + c + 4);
            const float4 b0 = *(const float4*)(gb + (size_t)n * C + c), b1 = *(const float4*)(gb + (size_t)n * C + c + 4);
            const float4 x0 = *(const float4*)(x + (size_t)p * C + c), x1 = *(const float4*)(x + (size_t)p * C + c + 4);
            const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const float xx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = silu_f(fmaf(aa[e], yy[e], bb[e])) + xx[e];
            *(float4*)(out + (size_t)p * C + c) = make_float4(val[0], val[1], val[2], val[3]);
            *(float4*)(out + (size_t)p * C + c + 4) = make_float4(val[4], val[5], val[6], val[7]);
        }
        s = ((val[0] + val[1]) + (val[2] + val[3])) + ((val[4] + val[5]) + (val[6] + val[7]));
        if (lnstats) {                                         // uniform branch
            const float mean = group_sum(s, lpp) / (float)C;
            float d2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = val[e] - mean; d2 = fmaf(d, d, d2); }
            const float var = group_sum(d2, lpp) / (float)C;
            const float rstd = rsqrtf(var + 1e-5f);
            if (ok && q == 0) *(float2*)(lnstats + (size_t)p * 2) = make_float2(rstd, mean * rstd);
        }
    }
}

// fp32 NHWC <-> Q16 (test helpers and the format's reference encoder): one thread per block (row, 32-column tile, half, channel)
__global__ __launch_bounds__(256) void q16_encode_kernel(const float* __restrict__ x, short* __restrict__ q, unsigned char* __restrict__ sc,
                                                         long long rows, int W, int C) {
    const int TX = (W + 31) >> 5;
    const long long nblk = rows * TX * 2 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int hh = (int)(t & 1); t >>= 1;
        const int xt = (int)(t % TX);
        const long long row = t / TX;
        float v[16];
        unsigned mx = 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int xx = xt * 32 + (k & 3) + 8 * (k >> 2) + 4 * hh;
            v[k] = xx < W ? x[((size_t)row * W + xx) * C + c] : 0.f;
            const unsigned ab = __builtin_bit_cast(unsigned, v[k]) & 0x7fffffffu;
            mx = mx > ab ? mx : ab;
        }
        float enc;
        const unsigned eb = q16_exponent(__builtin_bit_cast(float, mx), enc);
        sc[i] = (unsigned char)eb;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int xx = xt * 32 + (k & 3) + 8 * (k >> 2) + 4 * hh;
            if (xx < W) q[((size_t)row * W + xx) * C + c] = (short)q16_word(v[k], enc);
        }
    }
}
__global__ __launch_bounds__(256) void q16_decode_kernel(const short* __restrict__ q, const unsigned char* __restrict__ sc, float* __restrict__ x,
                                                         long long rows, int W, int C) {
    const long long pieces = rows * W * (C / 8);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (C / 8)) * 8;
        const long long p = i / (C / 8), row = p / W;
        float v[8];
        q16_decode8(*(const uint4*)(q + (size_t)p * C + c), *(const uint2*)(sc + q16_scale_index(row, (int)(p - row * W), W, C) + c), v);
        *(float4*)(x + (size_t)p * C + c) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(x + (size_t)p * C + c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// channel-LayerNorm statistics only (models/unet_model.py:52-61): stats[p] = (rstd, mean*rstd)
template <int V>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ lnstats,
                                                       long long P, int C, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        const bool ok = p < P;
        float4 val[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            val[v] = ok ? *(const float4*)(x + (size_t)p * C + (v * lpp + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (val[v].x + val[v].y) + (val[v].z + val[v].w);
        }
        const float mean = group_sum(s, lpp) / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dx = val[v].x - mean, dy = val[v].y - mean, dz = val[v].z - mean, dw = val[v].w - mean;
            d2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = rsqrtf(group_sum(d2, lpp) / (float)C + 1e-5f);
        if (ok && q == 0) *(float2*)(lnstats + (size_t)p * 2) = make_float2(rstd, mean * rstd);
    }
}

// out = LayerNorm_c(z) * g + x  -- LinearAttention.to_out[1] + Residual (unet_model.py:187-190,36)
template <int V>
__global__ __launch_bounds__(256) void ln_res_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                     const float* __restrict__ x, float* __restrict__ out,
                                                     long long P, int C, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    float4 gg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) gg[v] = *(const float4*)(g + (v * lpp + q) * 4);
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        const bool ok = p < P;
        float4 val[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            val[v] = ok ? *(const float4*)(z + (size_t)p * C + (v * lpp + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (val[v].x + val[v].y) + (val[v].z + val[v].w);
        }
        const float mean = group_sum(s, lpp) / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dx = val[v].x - mean, dy = val[v].y - mean, dz = val[v].z - mean, dw = val[v].w - mean;
            d2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = rsqrtf(group_sum(d2, lpp) / (float)C + 1e-5f);
        if (ok) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const size_t o = (size_t)p * C + (v * lpp + q) * 4;
                const float4 xx = *(const float4*)(x + o);
                float4 r;
                r.x = (val[v].x - mean) * rstd * gg[v].x + xx.x;
                r.y = (val[v].y - mean) * rstd * gg[v].y + xx.y;
                r.z = (val[v].z - mean) * rstd * gg[v].z + xx.z;
                r.w = (val[v].w - mean) * rstd * gg[v].w + xx.w;
                *(float4*)(out + o) = r;
            }
        }
    }
}

// final_conv: Conv2d(dim, out_dim, 1) (models/unet_model.py:331,368). NHWC in -> NCHW out.
template <int V>
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         long long P, long long hw, int C, int OD, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        const bool ok = p < P;
        float4 val[V];
#pragma unroll
        for (int v = 0; v < V; ++v)
            val[v] = ok ? *(const float4*)(x + (size_t)p * C + (v * lpp + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int o = 0; o < OD; ++o) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float4 ww = *(const float4*)(w + (size_t)o * C + (v * lpp + q) * 4);
                s = fmaf(val[v].x, ww.x, s); s = fmaf(val[v].y, ww.y, s);
                s = fmaf(val[v].z, ww.z, s); s = fmaf(val[v].w, ww.w, s);
            }
            s = group_sum(s, lpp);
            if (ok && q == 0) {
                const long long n = p / hw, pix = p - n * hw;
                out[((size_t)n * OD + o) * hw + pix] = s + bias[o];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// final_res_block's tail fused with final_conv (inference, models/unet_model.py:166-175 + :331,368):
//   eps[o] = final_conv( res_conv(cat(x1, x2)) + silu(GN2(y2)) )[o]
//          = (Wf W_res)[o] . cat(x1, x2)  +  Wf[o] . silu(a y2 + b)  +  (Wf b_res + bf)[o]
// Both 1x1 convolutions are linear, so the 64-channel block output never has to exist: one pass reads x1, x2, y2 and writes OD
// values per pixel, in fp32 (no MFMA: 192 FMAs + 64 SiLU per pixel against 768 bytes of HBM reads).
// fold: wfold[o][0..C1+C2) = Wf[o] W_res, wfold[o][C1+C2] = Wf[o] . b_res + bf[o]        (one block)
// Row stride Cin + 4 floats (a multiple of 4: the fused kernel reads the rows with 16-byte loads; an odd stride of Cin + 1 left
// every row o >= 1 only 4-byte aligned -- ADVICE r3)
__global__ __launch_bounds__(256) void final_fold_kernel(const float* __restrict__ wf, const float* __restrict__ bf,
                                                         const float* __restrict__ wres, const float* __restrict__ bres,
                                                         int OD, int C, int Cin, float* __restrict__ wfold) {
    for (int i = threadIdx.x; i < OD * (Cin + 1); i += blockDim.x) {
        const int o = i / (Cin + 1), j = i - o * (Cin + 1);
        float s = 0.f;
        if (j < Cin) {
#pragma unroll 16
            for (int c = 0; c < C; ++c) s = fmaf(wf[o * C + c], wres[(size_t)c * Cin + j], s);     // (16 loads in flight: the loop was a chain of C latencies)
        } else { for (int c = 0; c < C; ++c) s = fmaf(wf[o * C + c], bres ? bres[c] : 0.f, s); s += bf ? bf[o] : 0.f; }
        wfold[(size_t)o * (Cin + 4) + j] = s;
    }
}

// x1, x2, y2: (P, C) NHWC with the same C; a pixel's channels over lpp lanes, V float4 per lane; two pixel groups per iteration
template <int V>
__global__ __launch_bounds__(256) void final_fused_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          const float* __restrict__ y, const float* __restrict__ ga,
                                                          const float* __restrict__ gb, const float* __restrict__ wf,
                                                          const float* __restrict__ wfold, float* __restrict__ out,
                                                          long long P, long long hw, int C, int OD, int lpp) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    constexpr int U = 2;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw * U) {
        float4 yy[U][V], xa[U][V], xb[U][V], aa[U][V], bb[U][V];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pu = p0 + u * nwaves * ppw + sub;
            ok[u] = pu < P;
            const long long p = ok[u] ? pu : P - 1;             // unconditional loads, clamped address
            const int n = (int)(p / hw);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int c = (v * lpp + q) * 4;
                yy[u][v] = *(const float4*)(y + (size_t)p * C + c);
                xa[u][v] = *(const float4*)(x1 + (size_t)p * C + c);
                xb[u][v] = *(const float4*)(x2 + (size_t)p * C + c);
                aa[u][v] = *(const float4*)(ga + (size_t)n * C + c);
                bb[u][v] = *(const float4*)(gb + (size_t)n * C + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + u * nwaves * ppw + sub;
            float4 sv[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                sv[v].x = silu_f(fmaf(aa[u][v].x, yy[u][v].x, bb[u][v].x)); sv[v].y = silu_f(fmaf(aa[u][v].y, yy[u][v].y, bb[u][v].y));
                sv[v].z = silu_f(fmaf(aa[u][v].z, yy[u][v].z, bb[u][v].z)); sv[v].w = silu_f(fmaf(aa[u][v].w, yy[u][v].w, bb[u][v].w));
            }
            for (int o = 0; o < OD; ++o) {
                const float* wo = wfold + (size_t)o * (2 * C + 4);
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int c = (v * lpp + q) * 4;
                    const float4 w0 = *(const float4*)(wf + (size_t)o * C + c);
                    const float4 w1 = *(const float4*)(wo + c), w2 = *(const float4*)(wo + C + c);
                    s = fmaf(sv[v].x, w0.x, s); s = fmaf(sv[v].y, w0.y, s); s = fmaf(sv[v].z, w0.z, s); s = fmaf(sv[v].w, w0.w, s);
                    s = fmaf(xa[u][v].x, w1.x, s); s = fmaf(xa[u][v].y, w1.y, s); s = fmaf(xa[u][v].z, w1.z, s); s = fmaf(xa[u][v].w, w1.w, s);
                    s = fmaf(xb[u][v].x, w2.x, s); s = fmaf(xb[u][v].y, w2.y, s); s = fmaf(xb[u][v].z, w2.z, s); s = fmaf(xb[u][v].w, w2.w, s);
                }
                s = group_sum(s, lpp);
                if (ok[u] && q == 0) {
                    const long long n = p / hw, pix = p - n * hw;
                    out[((size_t)n * OD + o) * hw + pix] = s + wo[2 * C];
                }
            }
        }
    }
}

// per-(sample, channel) 1/max(||.||_2 over pixels, 1e-12) from a (sum, sumsq) slab:
// F.normalize(dim=-1) of models/unet_model.py:23,234.  One block per (sample, 64-channel group):
// 4 waves split the slab entries, lanes own channels (coalesced), LDS combines the 4 partials.
__global__ __launch_bounds__(256) void colnorm_finalize_kernel(const float* __restrict__ slab, int E, int C, int B,
                                                               float* __restrict__ inv) {
    __shared__ double part[4][64];
    const int cg = (C + 63) / 64;
    const int n = blockIdx.x / cg, c = (blockIdx.x - n * cg) * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    double s = 0.0;
    if (c < C)
        for (int e = w; e < E; e += 4) s += slab[(((size_t)n * E + e) * C + c) * 2 + 1];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) {
        s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        inv[(size_t)n * C + c] = 1.f / fmaxf((float)sqrt(s), 1e-12f);
    }
}
// per-(sample, channel) max over pixels from a max slab (k.softmax(dim=-1) stabiliser, :201)
__global__ __launch_bounds__(256) void colmax_finalize_kernel(const float* __restrict__ slab, int E, int C, int B,
                                                              float* __restrict__ mx) {
    __shared__ float part[4][64];
    const int cg = (C + 63) / 64;
    const int n = blockIdx.x / cg, c = (blockIdx.x - n * cg) * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float m = -INFINITY;
    if (c < C)
        for (int e = w; e < E; e += 4) m = fmaxf(m, slab[((size_t)n * E + e) * C + c]);
    part[w][threadIdx.x & 63] = m;
    __syncthreads();
    if (w == 0 && c < C)
        mx[(size_t)n * C + c] = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]),
                                      fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
}

// nearest-resize one NHWC tap to (S,S) and scatter into the NCHW feature tensor
// (models/datasetDM_model.py:80-83).  F.interpolate(mode='nearest'): src = floor(dst * in/out).
__global__ void tap_to_features_kernel(const float* __restrict__ tap, int h, int w, int C, float* __restrict__ feats,
                                       int C_total, int c_off, int S, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % S);
        long long r = i / S;
        const int yo = (int)(r % S); r /= S;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        const int ys = (int)(((long long)yo * h) / S), xs = (int)(((long long)xo * w) / S);
        feats[(((size_t)n * C_total + c_off + c) * S + yo) * S + xo] = tap[(((size_t)n * h + ys) * w + xs) * C + c];
    }
}

// Integer-factor variant (the TEDM case: S / h in {1,2,4,8}): a block takes 32 source pixels of one row
// x 64 channels, transposes them through LDS, and writes every (channel, replicated row) as contiguous
// float4 runs of the NCHW feature tensor -- reads and writes are both fully coalesced.
__global__ __launch_bounds__(256) void tap_to_features_tiled_kernel(const float* __restrict__ tap, int h, int w, int C,
                                                                    float* __restrict__ feats, int C_total, int c_off,
                                                                    int S, int f) {
    __shared__ float tile[64][33];
    const int tid = threadIdx.x;
    const int xt = blockIdx.x, ys = blockIdx.y;
    const int cchunks = (C + 63) / 64;
    const int n = blockIdx.z / cchunks, c0 = (blockIdx.z - n * cchunks) * 64;
    const int x0 = xt * 32;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * 256;                 // 512 float4 = 32 px x 16 channel quads
        const int px = i >> 4, c4 = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x0 + px < w && c0 + c4 < C) v = *(const float4*)(tap + (((size_t)n * h + ys) * w + x0 + px) * C + c0 + c4);
        tile[c4 + 0][px] = v.x; tile[c4 + 1][px] = v.y; tile[c4 + 2][px] = v.z; tile[c4 + 3][px] = v.w;
    }
    __syncthreads();
    const int row4 = 8 * f;                          // float4 per output row segment (32*f floats)
    const int total = 64 * f * row4;                 // (channel, replicated row, float4)
    for (int i = tid; i < total; i += 256) {
        const int q = i % row4;
        const int rr = i / row4;
        const int fy = rr % f, c = rr / f;
        if (c0 + c >= C) continue;
        const int xo = q * 4;                        // output x offset inside the segment; f in {1,2,4,8}
        float4 v;
        v.x = tile[c][(xo + 0) / f]; v.y = tile[c][(xo + 1) / f]; v.z = tile[c][(xo + 2) / f]; v.w = tile[c][(xo + 3) / f];
        const int ox = x0 * f + xo;
        if (ox + 3 < S)
            *(float4*)(feats + (((size_t)n * C_total + c_off + c0 + c) * S + ys * f + fy) * S + ox) = v;
        else {
            float* o = feats + (((size_t)n * C_total + c_off + c0 + c) * S + ys * f + fy) * S;
            if (ox < S) o[ox] = v.x;
            if (ox + 1 < S) o[ox + 1] = v.y;
            if (ox + 2 < S) o[ox + 2] = v.z;
        }
    }
}

inline bool pixmap(int C, PixMap& m) {
    if (C % 4) return false;
    int v4 = C / 4, lpp = 1;
    while (lpp < 64 && (v4 % (lpp * 2)) == 0) lpp *= 2;
    m.lpp = lpp; m.v = v4 / lpp;
    return m.v >= 1 && m.v <= 4;
}
inline unsigned pix_grid(long long P, int lpp) {
    const long long waves = (P + (64 / lpp) - 1) / (64 / lpp);
    long long blocks = (waves + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

#define DISPATCH_V(mv, KERN, ...)                                                            \
    switch (mv) {                                                                              \
        case 1: hipLaunchKernelGGL(KERN<1>, __VA_ARGS__); break;                               \
        case 2: hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); break;                               \
        case 3: hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); break;                               \
        default: hipLaunchKernelGGL(KERN<4>, __VA_ARGS__); break;                              \
    }

int op_q_sample(const float* x0, const float* noise, const long long* t, const float* ca, const float* cb, int T,
                float* xt, int B, long long chw, int normalize, hipStream_t st) {
    const long long total = (long long)B * chw;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x0, noise, t, ca, cb, T, xt, chw,
                       total, normalize);
    return LAUNCH_OK();
}

int op_stem_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                 int Cout, hipStream_t st) {
    const int tiles = ((W + STEM_TC - 1) / STEM_TC) * ((H + STEM_TR - 1) / STEM_TR);
    const int K = Cin * 49;
    if (Cin <= 2 && (Cout == 32 || Cout == 64)) {                 // fp16 hi/lo matrix-core path (the model's stem: Cin = 1)
        const size_t lds = (size_t)((Cin * (STEM_HR + 1) * STEM_HC + 3) & ~3) * 4 + (size_t)Cin * 4 * (Cout / 32) * 2 * 1024 + (size_t)2 * Cout * 4;
        // persistent workgroups: at most 8 per CU (19 KB of LDS each), every one builds the weight fragments once and walks its tiles
        const int cap = 256 * 8;             // measured at 32x256x256: 1024 WGs 215 us, 2048 188-195, 4096 213, one tile per WG (8192) 243
        const int nwg = B * tiles < cap ? B * tiles : cap;
        if (Cout == 32) hipLaunchKernelGGL(stem_conv_f16_kernel<1>, dim3(nwg), dim3(256), lds, st, x, w, bias, out, B, Cin, H, W, Cout);
        else hipLaunchKernelGGL(stem_conv_f16_kernel<2>, dim3(nwg), dim3(256), lds, st, x, w, bias, out, B, Cin, H, W, Cout);
        return LAUNCH_OK();
    }
    if (K <= 128 && (Cout == 32 || Cout == 64)) {                 // exact-fp32 matrix-core path
        const size_t lds = (size_t)((Cin * STEM_HR * STEM_HC + 3) & ~3) * 4 + (size_t)2 * ((K + 1) / 2) * Cout * 4;
        if (Cout == 32) hipLaunchKernelGGL(stem_conv_mfma_kernel<1>, dim3(B * tiles), dim3(256), lds, st, x, w, bias, out, B, Cin, H, W, Cout);
        else hipLaunchKernelGGL(stem_conv_mfma_kernel<2>, dim3(B * tiles), dim3(256), lds, st, x, w, bias, out, B, Cin, H, W, Cout);
        return LAUNCH_OK();
    }
    const size_t lds = (size_t)((Cin * STEM_HR * STEM_HC + 3) & ~3) * 4 + (size_t)Cin * 49 * 64 * 4;
    if (lds > 160 * 1024) return TEDM_EINVAL;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)stem_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return TEDM_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL(stem_conv_kernel, dim3(B * tiles), dim3(256), lds, st, x, w, bias, out, B, Cin, H, W, Cout);
    return LAUNCH_OK();
}

int op_time_embed(const float* t, const float* freqs, const float* w1, const float* b1, const float* w2,
                  const float* b2, int B, int dim, int tdim, int learned, float* st_out, float* save, hipStream_t st) {
    hipLaunchKernelGGL(temb_kernel, dim3(B), dim3(256), (size_t)(dim + tdim) * 4, st, t, freqs, w1, b1, w2, b2, dim,
                       tdim, learned, st_out, save);
    return LAUNCH_OK();
}

// f = 2 pi t w  =>  d w[i] = sum_b (d sin_i cos_i - d cos_i sin_i) * 2 pi t_b   (sin, cos, t read back from the saved embedding)
__global__ __launch_bounds__(64) void learned_sinusoid_bwd_kernel(const float* __restrict__ demb, const float* __restrict__ emb, int B,
                                                                  int half, float* __restrict__ dw) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= half) return;
    const int D = 2 * half + 1;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* e = emb + (size_t)b * D;
        const float* d = demb + (size_t)b * D;
        s = fmaf(fmaf(d[1 + i], e[1 + half + i], -d[1 + half + i] * e[1 + i]), 6.28318530717958647692f * e[0], s);
    }
    dw[i] = s;
}
int op_learned_sinusoid_backward(const float* demb, const float* emb, int B, int half, float* dw, hipStream_t st) {
    hipLaunchKernelGGL(learned_sinusoid_bwd_kernel, dim3((half + 63) / 64), dim3(64), 0, st, demb, emb, B, half, dw);
    return LAUNCH_OK();
}

int op_film(const float* st_in, const float* wf, const float* bf, int B, int tdim, int J, float* film, hipStream_t st) {
    if (J <= 0) return TEDM_OK;
    hipLaunchKernelGGL(film_kernel, dim3((J + 3) / 4), dim3(256), 0, st, st_in, wf, bf, B, tdim, J, film);
    return LAUNCH_OK();
}

int op_gn_finalize(const float* slab, int E, int B, int C, int groups, long long hw, const float* gamma,
                   const float* beta, const float* film, int film_stride, float* oa, float* ob, float* mr_out,
                   hipStream_t st) {
    if (C % groups) return TEDM_EINVAL;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, st, slab, E, C, groups, hw, gamma, beta,
                       film, film_stride, oa, ob, mr_out);
    return LAUNCH_OK();
}

int op_resblock_out(const float* y, const float* ga, const float* gb, const float* x, float* out, float* lnstats,
                    int B, long long hw, int C, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    const long long P = (long long)B * hw;
    DISPATCH_V(m.v, resblock_out_kernel, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, y, ga, gb, x, out, lnstats, P, hw,
               C, m.lpp);
    return LAUNCH_OK();
}

size_t q16_bytes(int B, int H, int W, int C) { return (size_t)2 * B * H * W * C + (size_t)B * H * ((W + 31) / 32) * 2 * C; }

int op_resblock_out_q16(const void* y, const float* ga, const float* gb, const float* x, float* out, float* lnstats,
                        int B, int H, int W, int C, hipStream_t st) {
    const int lpp = C / 8;
    if (C % 8 || lpp < 1 || lpp > 64 || (lpp & (lpp - 1))) return TEDM_EINVAL;       // C in {8, 16, ..., 512}, a power of two
    const long long hw = (long long)H * W, P = (long long)B * hw;
    hipLaunchKernelGGL(resblock_out_q16_kernel, dim3(pix_grid(P, lpp)), dim3(256), 0, st, (const short*)y,
                       (const unsigned char*)y + (size_t)2 * P * C, ga, gb, x, out, lnstats, P, hw, W, C, lpp);
    return LAUNCH_OK();
}
int op_q16_encode(const float* x, void* q, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8) return TEDM_EINVAL;
    const long long rows = (long long)B * H, nblk = rows * ((W + 31) / 32) * 2 * C;
    long long blocks = (nblk + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(q16_encode_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st, x, (short*)q,
                       (unsigned char*)q + (size_t)2 * rows * W * C, rows, W, C);
    return LAUNCH_OK();
}
int op_q16_decode(const void* q, float* x, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8) return TEDM_EINVAL;
    const long long rows = (long long)B * H, pieces = rows * W * (C / 8);
    long long blocks = (pieces + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(q16_decode_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st, (const short*)q,
                       (const unsigned char*)q + (size_t)2 * rows * W * C, x, rows, W, C);
    return LAUNCH_OK();
}

int op_ln_stats(const float* x, float* lnstats, long long P, int C, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    DISPATCH_V(m.v, ln_stats_kernel, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, x, lnstats, P, C, m.lpp);
    return LAUNCH_OK();
}

int op_ln_res(const float* z, const float* g, const float* x, float* out, long long P, int C, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    DISPATCH_V(m.v, ln_res_kernel, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, z, g, x, out, P, C, m.lpp);
    return LAUNCH_OK();
}

int op_final_conv(const float* x, const float* w, const float* bias, float* out, int B, long long hw, int C, int OD,
                  hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    const long long P = (long long)B * hw;
    DISPATCH_V(m.v, final_conv_kernel, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, x, w, bias, out, P, hw, C, OD,
               m.lpp);
    return LAUNCH_OK();
}

// wfold: OD * (2C + 4) floats of scratch (written here, read by the fused kernel on the same stream)
int op_final_fused(const float* x1, const float* x2, const float* y2, const float* ga, const float* gb, const float* wres,
                   const float* bres, const float* wf, const float* bf, float* wfold, float* out_nchw, int B, long long hw,
                   int C, int OD, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m) || m.v > 2 || OD < 1) return TEDM_EINVAL;
    const long long P = (long long)B * hw;
    hipLaunchKernelGGL(final_fold_kernel, dim3(1), dim3(256), 0, st, wf, bf, wres, bres, OD, C, 2 * C, wfold);
    switch (m.v) {
        case 1: hipLaunchKernelGGL(final_fused_kernel<1>, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, x1, x2, y2, ga, gb, wf, wfold, out_nchw, P, hw, C, OD, m.lpp); break;
        default: hipLaunchKernelGGL(final_fused_kernel<2>, dim3(pix_grid(P, m.lpp)), dim3(256), 0, st, x1, x2, y2, ga, gb, wf, wfold, out_nchw, P, hw, C, OD, m.lpp); break;
    }
    return LAUNCH_OK();
}
bool final_fused_supported(int C1, int C2, int C) { PixMap m; return C1 == C && C2 == C && pixmap(C, m) && m.v <= 2; }

int op_colnorm_finalize(const float* slab, int E, int C, int B, float* inv, hipStream_t st) {
    hipLaunchKernelGGL(colnorm_finalize_kernel, dim3(B * ((C + 63) / 64)), dim3(256), 0, st, slab, E, C, B, inv);
    return LAUNCH_OK();
}
int op_colmax_finalize(const float* slab, int E, int C, int B, float* mx, hipStream_t st) {
    hipLaunchKernelGGL(colmax_finalize_kernel, dim3(B * ((C + 63) / 64)), dim3(256), 0, st, slab, E, C, B, mx);
    return LAUNCH_OK();
}

int op_tap_to_features(const float* tap, int B, int h, int w, int C, float* feats, int C_total, int c_off, int S,
                       hipStream_t st) {
    if (h == w && S % h == 0 && (S / h == 1 || S / h == 2 || S / h == 4 || S / h == 8) && C % 4 == 0 && S % 4 == 0 &&
        (long long)B * ((C + 63) / 64) <= 65535) {
        dim3 grid((w + 31) / 32, h, B * ((C + 63) / 64));
        hipLaunchKernelGGL(tap_to_features_tiled_kernel, grid, dim3(256), 0, st, tap, h, w, C, feats, C_total, c_off, S, S / h);
        return LAUNCH_OK();
    }
    const long long total = (long long)B * C * S * S;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tap_to_features_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tap, h, w, C, feats, C_total,
                       c_off, S, total);
    return LAUNCH_OK();
}
