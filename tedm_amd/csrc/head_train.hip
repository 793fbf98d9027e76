// head_train.hip -- TRAINING-mode tail of the pixel classifier (models/datasetDM_model.py:59-64 behind the first conv:
// ReLU, BatchNorm2d(C1), Conv2d(C1, 32, 1), ReLU, BatchNorm2d(32), Conv2d(32, 1, 1)) on the NHWC tensor h1 (P pixels x C1) that
// the fused first layer (tedm_amd/head.py) produces.  torch runs this as ~15 tensor passes over C1- and 32-channel tensors per
// step; here the forward is three passes and the backward three, every intermediate except z2 (P x 32) is recomputed:
//
//   forward   stats1 : per-channel (sum, sumsq) of r1 = relu(h1)                                  -> BatchNorm1 batch statistics
//             mid    : z2 = W2 (a1 r1 + b1) + bias2, per-channel (sum, sumsq) of r2 = relu(z2)    -> BatchNorm2 batch statistics
//             out    : logit = w3 . (a2 r2 + b2) + b3
//   backward  (dl = dLoss/dlogit per pixel; BatchNorm backward needs two global sums per channel before dx)
//             bwd1   : A0 = sum dl, A1[o] = sum dl r2[o]            -> dW3, db3, dgamma2, dbeta2 and the coefficients k of
//                      dz2[o] = [z2 > 0] (k0 dl + k1 + k2 z2)          (all derived on the host side: 33 numbers)
//             bwd2   : G[o][c] = sum dz2[o] r1[c], db2[o] = sum dz2[o] -> dW2, dgamma1, dbeta1 and the coefficients m of
//                      dh1[c] = [h1 > 0] (m0 du + m1 + m2 h1),  du = W2^T dz2   (dW2 = a1 G + b1 db2; sums of du and du xhat1
//                      are linear in G and db2: no third reduction pass)
//             bwd3   : dh1
// The a / b / k / m coefficient vectors and the reductions of the per-block partials are tiny and stay in torch (fp64 where
// the reference's BatchNorm is sensitive).  Partials are written per block and summed in a fixed order: deterministic, no atomics.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

constexpr int HP = 256, HL = 33;             // pixels per tile of the pixel-owner kernels, padded LDS row

// ---- forward 1: per-channel (sum, sumsq) of relu(h1) ---------------------------------------------------------------------
// Sums are taken of (x - shift[c]) (shift = the BatchNorm's running mean; null = 0): E[x^2] - mean^2 from fp32 sums loses the
// variance of a channel whose mean is large against its spread; around a shift near the mean it does not.
__global__ __launch_bounds__(256) void ht_stats1_kernel(const float* __restrict__ h1, long long P, int C1, const float* __restrict__ shift,
                                                        float* __restrict__ part) {
    __shared__ float4 red[2][256];
    const int C4 = C1 >> 2, ppb = 256 / C4;
    const int tid = threadIdx.x, c4 = tid % C4, g = tid / C4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    const float4 sh = shift ? *(const float4*)(shift + c4 * 4) : s;
    for (long long p = (long long)blockIdx.x * ppb + g; p < P; p += (long long)gridDim.x * ppb) {
        const float4 v = *(const float4*)(h1 + (size_t)p * C1 + c4 * 4);
        const float a = fmaxf(v.x, 0.f) - sh.x, b = fmaxf(v.y, 0.f) - sh.y, c = fmaxf(v.z, 0.f) - sh.z, d = fmaxf(v.w, 0.f) - sh.w;
        s.x += a; s.y += b; s.z += c; s.w += d;
        q.x = fmaf(a, a, q.x); q.y = fmaf(b, b, q.y); q.z = fmaf(c, c, q.z); q.w = fmaf(d, d, q.w);
    }
    red[0][tid] = s; red[1][tid] = q;
    __syncthreads();
    if (g == 0) {
        for (int k = 1; k < ppb; ++k) {
            const float4 a = red[0][k * C4 + c4], b = red[1][k * C4 + c4];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        float* o = part + ((size_t)blockIdx.x * C1 + c4 * 4) * 2;
        o[0] = s.x; o[1] = q.x; o[2] = s.y; o[3] = q.y; o[4] = s.z; o[5] = q.z; o[6] = s.w; o[7] = q.w;
    }
}

// ---- forward 2: z2 = W2 (a1 relu(h1) + b1) + bias2 and the statistics of relu(z2) -----------------------------------------
__global__ __launch_bounds__(256) void ht_mid_kernel(const float* __restrict__ h1, long long P, int C1, const float* __restrict__ a1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2t /*[C1][32]*/,
                                                     const float* __restrict__ bias2, const float* __restrict__ shift2 /*[32] or null*/,
                                                     float* __restrict__ z2, float* __restrict__ part2) {
    __shared__ float sa[HP * HL];
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const long long ntile = (P + HP - 1) / HP;
    float ssum = 0.f, ssq = 0.f;                         // channel tid & 31, pixel group tid >> 5
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long p0 = tile * HP;
        float acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = bias2[j];
        for (int cb = 0; cb < C1; cb += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                // 2048 float4 of the chunk, 8 per thread (8 threads per pixel: 128 B)
                const int f = tid + 256 * k, px = f >> 3, c4 = (f & 7) * 4;
                const long long p = p0 + px < P ? p0 + px : P - 1;       // unconditional load, clamped (masked at the end)
                const float4 v = *(const float4*)(h1 + (size_t)p * C1 + cb + c4);
                const float4 av = *(const float4*)(a1 + cb + c4), bv = *(const float4*)(b1 + cb + c4);
                float* d = sa + px * HL + c4;
                d[0] = fmaf(fmaxf(v.x, 0.f), av.x, bv.x); d[1] = fmaf(fmaxf(v.y, 0.f), av.y, bv.y);
                d[2] = fmaf(fmaxf(v.z, 0.f), av.z, bv.z); d[3] = fmaf(fmaxf(v.w, 0.f), av.w, bv.w);
            }
            __syncthreads();
            const float* row = sa + tid * HL;
#pragma unroll 4
            for (int c = 0; c < 32; ++c) {
                const float u = row[c];
                const float* wr = w2t + (size_t)(cb + c) * 32;          // uniform address: scalar loads
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = fmaf(u, wr[j], acc[j]);
            }
            __syncthreads();
        }
        const bool ok = p0 + tid < P;
        if (ok) {
            float4* o = (float4*)(z2 + (size_t)(p0 + tid) * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
        float* row = sa + tid * HL;
#pragma unroll
        for (int j = 0; j < 32; ++j) row[j] = ok ? fmaxf(acc[j], 0.f) - (shift2 ? shift2[j] : 0.f) : 0.f;
        __syncthreads();
        {
            const int c = tid & 31, g = tid >> 5;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) { const float r = sa[(g * 32 + i) * HL + c]; ssum += r; ssq = fmaf(r, r, ssq); }
        }
        __syncthreads();
    }
    red[0][tid] = ssum; red[1][tid] = ssq;
    __syncthreads();
    if (tid < 32) {
        float s = 0.f, q = 0.f;
        for (int g = 0; g < 8; ++g) { s += red[0][g * 32 + tid]; q += red[1][g * 32 + tid]; }
        part2[((size_t)blockIdx.x * 32 + tid) * 2] = s;
        part2[((size_t)blockIdx.x * 32 + tid) * 2 + 1] = q;
    }
}

// ---- forward 3: logit = w3 . (a2 relu(z2) + b2) + b3 ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ht_out_kernel(const float* __restrict__ z2, long long P, const float* __restrict__ a2,
                                                     const float* __restrict__ b2, const float* __restrict__ w3,
                                                     const float* __restrict__ b3, float* __restrict__ logits) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const float4* zr = (const float4*)(z2 + (size_t)p * 32);
        float s = b3[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = zr[j];
            s = fmaf(w3[4 * j], fmaf(fmaxf(v.x, 0.f), a2[4 * j], b2[4 * j]), s);
            s = fmaf(w3[4 * j + 1], fmaf(fmaxf(v.y, 0.f), a2[4 * j + 1], b2[4 * j + 1]), s);
            s = fmaf(w3[4 * j + 2], fmaf(fmaxf(v.z, 0.f), a2[4 * j + 2], b2[4 * j + 2]), s);
            s = fmaf(w3[4 * j + 3], fmaf(fmaxf(v.w, 0.f), a2[4 * j + 3], b2[4 * j + 3]), s);
        }
        logits[p] = s;
    }
}

// ---- backward 1: A0 = sum dl, A1[o] = sum dl relu(z2[o]) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void ht_bwd1_kernel(const float* __restrict__ z2, const float* __restrict__ dl, long long P,
                                                      float* __restrict__ part /*[blocks][33]*/) {
    __shared__ float sa[256 * HL];
    const int tid = threadIdx.x;
    float acc[32], a0 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + tid; p < P; p += (long long)gridDim.x * 256) {
        const float d = dl[p];
        const float4* zr = (const float4*)(z2 + (size_t)p * 32);
        a0 += d;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = zr[j];
            acc[4 * j] = fmaf(d, fmaxf(v.x, 0.f), acc[4 * j]); acc[4 * j + 1] = fmaf(d, fmaxf(v.y, 0.f), acc[4 * j + 1]);
            acc[4 * j + 2] = fmaf(d, fmaxf(v.z, 0.f), acc[4 * j + 2]); acc[4 * j + 3] = fmaf(d, fmaxf(v.w, 0.f), acc[4 * j + 3]);
        }
    }
    float* row = sa + tid * HL;
#pragma unroll
    for (int j = 0; j < 32; ++j) row[j] = acc[j];
    row[32] = a0;
    __syncthreads();
    // column c of the [256][33] tile: 8 threads per column, then one
    __shared__ float red[8][HL];
    {
        const int c = tid % HL, g = tid / HL;            // 7 full groups of 33 threads (231 threads), the rest idle
        if (g < 7) {
            float s = 0.f;
            for (int i = g; i < 256; i += 7) s += sa[i * HL + c];
            red[g][c] = s;
        }
    }
    __syncthreads();
    if (tid < HL) {
        float s = 0.f;
        for (int g = 0; g < 7; ++g) s += red[g][tid];
        part[(size_t)blockIdx.x * HL + tid] = s;        // [0..31] = A1, [32] = A0
    }
}

// dz2[o] = [z2 > 0] (k0[o] dl + k1[o] + k2[o] z2), four consecutive o
TEDM_DEVFN float4 dz2_quad(const float4 z, float d, const float* __restrict__ k, int o) {
    float4 r;
    r.x = z.x > 0.f ? fmaf(k[64 + o], z.x, fmaf(k[o], d, k[32 + o])) : 0.f;
    r.y = z.y > 0.f ? fmaf(k[64 + o + 1], z.y, fmaf(k[o + 1], d, k[32 + o + 1])) : 0.f;
    r.z = z.z > 0.f ? fmaf(k[64 + o + 2], z.z, fmaf(k[o + 2], d, k[32 + o + 2])) : 0.f;
    r.w = z.w > 0.f ? fmaf(k[64 + o + 3], z.w, fmaf(k[o + 3], d, k[32 + o + 3])) : 0.f;
    return r;
}

// ---- backward 2: G[o][c] = sum_p dz2[p][o] relu(h1[p][c]),  db2[o] = sum_p dz2[p][o] -----------------------------------------
// 64-pixel tiles; thread (og = tid >> 5, cg = tid & 31) owns the 4 x 4 block G[4 og .. +3][4 (cg + 32 i) .. +3], i < NCG = C1 / 128.
template <int NCG>
__global__ __launch_bounds__(256) void ht_bwd2_kernel(const float* __restrict__ h1, const float* __restrict__ z2,
                                                      const float* __restrict__ dl, long long P, const float* __restrict__ k /*[3][32]*/,
                                                      float* __restrict__ part /*[blocks][32 * C1 + 32]*/) {
    constexpr int C1 = 128 * NCG, TP = 64, LR = C1 + 4, LZ = 36;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sr = sm;                      // [TP][LR]  relu(h1)
    float* sz = sm + TP * LR;            // [TP][LZ]  dz2
    const int tid = threadIdx.x, og = tid >> 5, cg = tid & 31;
    float g[NCG][4][4];
#pragma unroll
    for (int i = 0; i < NCG; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) g[i][a][b] = 0.f;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);        // running sum of this thread's staged dz2 quads (o quad = tid & 7)
    const long long ntile = (P + TP - 1) / TP;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long p0 = tile * TP;
#pragma unroll
        for (int i = 0; i < (TP * C1 / 4) / 256; ++i) {   // relu(h1) tile: C1 / 4 float4 per pixel
            const int f = tid + 256 * i, px = f / (C1 / 4), c4 = (f % (C1 / 4)) * 4;
            const bool ok = p0 + px < P;
            const float4 v = *(const float4*)(h1 + (size_t)(ok ? p0 + px : P - 1) * C1 + c4);
            *(float4*)(sr + px * LR + c4) = ok ? make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f))
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // dz2 tile: 8 float4 per pixel
            const int f = tid + 256 * i, px = f >> 3, o4 = (f & 7) * 4;
            const bool ok = p0 + px < P;
            const long long p = ok ? p0 + px : P - 1;
            const float4 z = *(const float4*)(z2 + (size_t)p * 32 + o4);
            float4 d = dz2_quad(z, dl[p], k, o4);
            if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4*)(sz + px * LZ + o4) = d;
            dbs.x += d.x; dbs.y += d.y; dbs.z += d.z; dbs.w += d.w;
        }
        __syncthreads();
#pragma unroll 4
        for (int px = 0; px < TP; ++px) {
            const float4 d = *(const float4*)(sz + px * LZ + og * 4);          // same address across the 32 lanes of a group: broadcast
            const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < NCG; ++i) {
                const float4 r = *(const float4*)(sr + px * LR + (cg + 32 * i) * 4);
                const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) g[i][a][b] = fmaf(dv[a], rv[b], g[i][a][b]);
            }
        }
        __syncthreads();
    }
    float* o = part + (size_t)blockIdx.x * (32 * C1 + 32);
#pragma unroll
    for (int i = 0; i < NCG; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *(float4*)(o + (size_t)(og * 4 + a) * C1 + (cg + 32 * i) * 4) = make_float4(g[i][a][0], g[i][a][1], g[i][a][2], g[i][a][3]);
    // db2: threads with the same o quad (tid & 7) hold partial sums
    float4* red = (float4*)sm;
    red[tid] = dbs;
    __syncthreads();
    if (tid < 8) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += 8) { const float4 v = red[j]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *(float4*)(o + 32 * C1 + tid * 4) = s;
    }
}

// ---- backward 3: dh1[c] = [h1 > 0] (m0[c] du[c] + m1[c] + m2[c] h1[c]),  du = W2^T dz2 ---------------------------------------
__global__ __launch_bounds__(256) void ht_bwd3_kernel(const float* __restrict__ h1, const float* __restrict__ z2,
                                                      const float* __restrict__ dl, long long P, int C1, const float* __restrict__ k,
                                                      const float* __restrict__ w2 /*[32][C1]*/, const float* __restrict__ m /*[3][C1]*/,
                                                      float* __restrict__ dh1) {
    __shared__ float sa[HP * HL];
    const int tid = threadIdx.x;
    const long long ntile = (P + HP - 1) / HP;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long p0 = tile * HP;
        const long long pm = p0 + tid < P ? p0 + tid : P - 1;
        float dz[32];
        {
            const float d = dl[pm];
            const float4* zr = (const float4*)(z2 + (size_t)pm * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = dz2_quad(zr[j], d, k, 4 * j);
                dz[4 * j] = q.x; dz[4 * j + 1] = q.y; dz[4 * j + 2] = q.z; dz[4 * j + 3] = q.w;
            }
        }
        for (int cb = 0; cb < C1; cb += 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {                 // h1 chunk -> LDS (coalesced: 8 threads per pixel)
                const int f = tid + 256 * i, px = f >> 3, c4 = (f & 7) * 4;
                const long long p = p0 + px < P ? p0 + px : P - 1;
                const float4 v = *(const float4*)(h1 + (size_t)p * C1 + cb + c4);
                float* d = sa + px * HL + c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
            float du[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) du[j] = 0.f;
#pragma unroll 4
            for (int o = 0; o < 32; ++o) {
                const float* wr = w2 + (size_t)o * C1 + cb;             // uniform address: scalar loads
                const float dv = dz[o];
#pragma unroll
                for (int j = 0; j < 32; ++j) du[j] = fmaf(dv, wr[j], du[j]);
            }
            float* row = sa + tid * HL;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float h = row[j];
                row[j] = h > 0.f ? fmaf(m[2 * C1 + cb + j], h, fmaf(m[cb + j], du[j], m[C1 + cb + j])) : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = tid + 256 * i, px = f >> 3, c4 = (f & 7) * 4;
                if (p0 + px < P) {
                    const float* d = sa + px * HL + c4;
                    *(float4*)(dh1 + (size_t)(p0 + px) * C1 + cb + c4) = make_float4(d[0], d[1], d[2], d[3]);
                }
            }
            __syncthreads();
        }
    }
}

int grid_for(long long work_items, int per_block, int cap) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

// Every function: device pointers, P = pixels (B*S*S), `blocks` = number of per-block partials the caller allocated (>= 1; the
// launch uses exactly that many workgroups).  C1: a power of two in [32, 1024] (stats1), a multiple of 32 (mid, bwd3), 128 or 256 (bwd2).
int tedm_head_train_stats1(const float* h1, long long P, int C1, const float* shift, float* part, int blocks, void* stream) {
    if (!h1 || !part || P < 1 || blocks < 1 || C1 < 32 || C1 > 1024 || (C1 & (C1 - 1))) return TEDM_EINVAL;
    hipLaunchKernelGGL(ht_stats1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h1, P, C1, shift, part);
    return LAUNCH_OK();
}
int tedm_head_train_mid(const float* h1, long long P, int C1, const float* a1, const float* b1, const float* w2t, const float* bias2,
                        const float* shift2, float* z2, float* part2, int blocks, void* stream) {
    if (!h1 || !a1 || !b1 || !w2t || !bias2 || !z2 || !part2 || P < 1 || blocks < 1 || C1 % 32 || C1 < 32) return TEDM_EINVAL;
    hipLaunchKernelGGL(ht_mid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h1, P, C1, a1, b1, w2t, bias2, shift2, z2, part2);
    return LAUNCH_OK();
}
int tedm_head_train_out(const float* z2, long long P, const float* a2, const float* b2, const float* w3, const float* b3, float* logits,
                        void* stream) {
    if (!z2 || !a2 || !b2 || !w3 || !b3 || !logits || P < 1) return TEDM_EINVAL;
    hipLaunchKernelGGL(ht_out_kernel, dim3(grid_for(P, 256, 8192)), dim3(256), 0, (hipStream_t)stream, z2, P, a2, b2, w3, b3, logits);
    return LAUNCH_OK();
}
int tedm_head_train_bwd1(const float* z2, const float* dl, long long P, float* part, int blocks, void* stream) {
    if (!z2 || !dl || !part || P < 1 || blocks < 1) return TEDM_EINVAL;
    hipLaunchKernelGGL(ht_bwd1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z2, dl, P, part);
    return LAUNCH_OK();
}
int tedm_head_train_bwd2(const float* h1, const float* z2, const float* dl, long long P, int C1, const float* k, float* part, int blocks,
                         void* stream) {
    if (!h1 || !z2 || !dl || !k || !part || P < 1 || blocks < 1 || (C1 != 128 && C1 != 256)) return TEDM_EINVAL;
    const size_t lds = (size_t)(64 * (C1 + 4) + 64 * 36) * sizeof(float);
    if (C1 == 128) hipLaunchKernelGGL(ht_bwd2_kernel<1>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, h1, z2, dl, P, k, part);
    else {
        // (the attribute is per DEVICE and the call is cheap: no process-wide flag -- ADVICE r3)
        if (hipFuncSetAttribute((const void*)ht_bwd2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEDM_EHIP;
        hipLaunchKernelGGL(ht_bwd2_kernel<2>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, h1, z2, dl, P, k, part);
    }
    return LAUNCH_OK();
}
int tedm_head_train_bwd3(const float* h1, const float* z2, const float* dl, long long P, int C1, const float* k, const float* w2,
                         const float* m, float* dh1, void* stream) {
    if (!h1 || !z2 || !dl || !k || !w2 || !m || !dh1 || P < 1 || C1 % 32 || C1 < 32) return TEDM_EINVAL;
    hipLaunchKernelGGL(ht_bwd3_kernel, dim3(grid_for(P, HP, 4096)), dim3(256), 0, (hipStream_t)stream, h1, z2, dl, P, C1, k, w2, m, dh1);
    return LAUNCH_OK();
}

}  // extern "C"
