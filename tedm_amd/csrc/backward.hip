// backward.hip -- HBM-bound fp32 kernels of the backward pass (everything that is not a convolution
// dgrad / wgrad): 2x2 sum-pool (backward of the nearest x2 upsample), per-channel sums (bias grads),
// GroupNorm+FiLM+SiLU backward, channel-LayerNorm backward, resnet-block output backward, loss grad.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

// backward of nn.Upsample(scale_factor=2, mode='nearest') (unet_model.py:41): out[y][x] = sum of the 2x2 block
__global__ void pool2x2_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int h, int w, int C,
                                   long long total4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4)) * 4;
        long long r = i / (C / 4);
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int n = (int)(r / h);
        const float* p = g + (((size_t)n * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c4;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + C);
        const float4 c = *(const float4*)(p + (size_t)2 * w * C), d = *(const float4*)(p + (size_t)2 * w * C + C);
        *(float4*)(out + (((size_t)n * h + y) * w + x) * C + c4) =
            make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    }
}

// ---------------------------------------------------------------------------------------
// Pixel-classifier head, first layer applied at the taps' native resolution:
//   out[n][y][x][:] = bias + sum_i z_i[n][y*h_i/S][x*w_i/S][:]      (nearest upsample of every projected tap + sum)
struct UpSumArgs { const float* z[8]; int h[8]; int w[8]; int n; };
// One block row per image row: n, y and the source rows are block-uniform; per thread only 32-bit index arithmetic.
__global__ __launch_bounds__(256) void upsample_sum_kernel(UpSumArgs a, const float* __restrict__ bias, float* __restrict__ out, int B, int S, int C,
                                                           long long total4) {
    (void)total4;
    const int C4 = C >> 2, row4 = S * C4;                       // float4 per image row
    for (int row = blockIdx.y; row < B * S; row += gridDim.y) {
        const int n = row / S, y = row - n * S;
        const float* zr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < a.n) zr[k] = a.z[k] + ((size_t)n * a.h[k] + (int)((long long)y * a.h[k] / S)) * a.w[k] * C;   // F.interpolate 'nearest'
        float* orow = out + (size_t)row * S * C;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < row4; e += gridDim.x * 256) {
            const int x = e / C4, c4 = (e - x * C4) * 4;
            float4 s = bias ? *(const float4*)(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < a.n) {
                    const int sx = (int)((unsigned)x * (unsigned)a.w[k] / (unsigned)S);
                    const float4 v = *(const float4*)(zr[k] + (size_t)sx * C + c4);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            *(float4*)(orow + (size_t)e * 4) = s;
        }
    }
}
// Inference-mode tail of the pixel classifier in ONE pass (models/datasetDM_model.py:57-64 after the first conv; eval-mode
// BatchNorm folded into the following conv by the host):
//     h1 = b1 + sum_i nearest_up(z_i)                (the 128-channel tensor is never written)
//     u  = b2' + W2' relu(h1)        (128 -> 32)      logit = b3' + w3' . relu(u)        (32 -> 1)
// 256 pixels per block; per 32-channel chunk the block builds relu(h1) in LDS with coalesced loads (8 threads per pixel), then
// every thread owns one pixel and accumulates its 32 outputs with the (wave-uniform, scalar-loaded) weights.
constexpr int HT_PIX = 256, HT_LD = 33;
__global__ __launch_bounds__(256) void head_tail_kernel(UpSumArgs a, const float* __restrict__ b1, const float* __restrict__ w2t /*[C1][32]*/,
                                                        const float* __restrict__ b2, const float* __restrict__ w3,
                                                        const float* __restrict__ b3, float* __restrict__ logits, long long P, int S, int C1) {
    __shared__ float sa[HT_PIX * HT_LD];
    const int tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * HT_PIX;
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = b2[j];
    for (int cb = 0; cb < C1; cb += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {                           // 2048 float4 of the chunk, 8 per thread
            const int f = tid + 256 * k, px = f >> 3, c4 = (f & 7) * 4;
            const long long p = p0 + px;
            float4 sm = *(const float4*)(b1 + cb + c4);
            if (p < P) {
                const int x = (int)(p % S);
                const long long r = p / S;
                const int y = (int)(r % S), n = (int)(r / S);
                for (int i = 0; i < a.n; ++i) {
                    const int sy = (int)((long long)y * a.h[i] / S), sx = (int)((long long)x * a.w[i] / S);   // F.interpolate 'nearest'
                    const float4 v = *(const float4*)(a.z[i] + (((size_t)n * a.h[i] + sy) * a.w[i] + sx) * C1 + cb + c4);
                    sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
                }
            }
            float* d = sa + px * HT_LD + c4;
            d[0] = fmaxf(sm.x, 0.f); d[1] = fmaxf(sm.y, 0.f); d[2] = fmaxf(sm.z, 0.f); d[3] = fmaxf(sm.w, 0.f);
        }
        __syncthreads();
        const float* row = sa + tid * HT_LD;
#pragma unroll 4
        for (int c = 0; c < 32; ++c) {
            const float av = row[c];
            const float* wr = w2t + (size_t)(cb + c) * 32;      // uniform address: scalar loads
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] = fmaf(wr[j], av, acc[j]);
        }
        __syncthreads();
    }
    float lg = b3[0];
#pragma unroll
    for (int j = 0; j < 32; ++j) lg = fmaf(w3[j], fmaxf(acc[j], 0.f), lg);
    if (p0 + tid < P) logits[p0 + tid] = lg;
}

// its adjoint: out[n][y][x][:] = sum over the f x f block of g that nearest-upsampling maps to (y,x)   (S = f*h)
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int h, int w, int f, int C, long long total4) {
    (void)total4;
    const int C4 = C >> 2, row4 = w * C4;                       // float4 per output row; n, y block-uniform
    for (int row = blockIdx.y; row < B * h; row += gridDim.y) {
        const float* grow = g + (size_t)row * f * ((size_t)w * f) * C;          // (n * h + y) * f = first source row
        float* orow = out + (size_t)row * w * C;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < row4; e += gridDim.x * 256) {
            const int x = e / C4, c4 = (e - x * C4) * 4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) {
                    const float4 v = *(const float4*)(grow + ((size_t)dy * w * f + (size_t)x * f + dx) * C + c4);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            *(float4*)(orow + (size_t)e * 4) = s;
        }
    }
}

TEDM_DEVFN float dsilu_f(float u) {            // d/du [u * sigmoid(u)]
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return sg * (1.0f + u * (1.0f - sg));
}

// ---------------------------------------------------------------------------------------
// Backward of Block's tail: hact = silu(u), u = (xhat*gamma+beta)*(1+s)+t, xhat = (y-mean)*rstd
// (models/unet_model.py:126-135).  u is re-evaluated from the raw conv output y with the folded
// per-(n,c) coefficients of the forward (u = fa*y + fb).
// Pass A: per (sample, pixel chunk, channel) partial sums  P0 = sum du,  P1 = sum du*xhat,  P2 = sum xhat,
// with du = G * silu'(u).  Grid (E, B), one block per pixel chunk of one image.
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ G, const float* __restrict__ y,
                                                            const float* __restrict__ fa, const float* __restrict__ fb,
                                                            const float* __restrict__ mr, int groups, long long hw, int C,
                                                            int lpp, int ppb, float* __restrict__ slab /*[B][E][C][3]*/) {
    __shared__ float red[4][3][64 * 4 * V];      // wave partials: [wave][comp][channel]
    const int n = blockIdx.y, e = blockIdx.x, E = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const int cg = C / groups;
    float4 a4[V], b4[V], mu[V], rs[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int c = (v * lpp + q) * 4;
        a4[v] = *(const float4*)(fa + (size_t)n * C + c);
        b4[v] = *(const float4*)(fb + (size_t)n * C + c);
        const float* m0 = mr + ((size_t)n * groups) * 2;
        mu[v] = make_float4(m0[((c + 0) / cg) * 2], m0[((c + 1) / cg) * 2], m0[((c + 2) / cg) * 2], m0[((c + 3) / cg) * 2]);
        rs[v] = make_float4(m0[((c + 0) / cg) * 2 + 1], m0[((c + 1) / cg) * 2 + 1], m0[((c + 2) / cg) * 2 + 1], m0[((c + 3) / cg) * 2 + 1]);
    }
    float4 p0[V], p1[V], p2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { p0[v] = p1[v] = p2[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
    const long long pbeg = (long long)e * ppb, pend = (pbeg + ppb < hw) ? pbeg + ppb : hw;
    for (long long p = pbeg + wave * ppw + sub; p < pend; p += 4 * ppw) {
        const size_t base = ((size_t)n * hw + p) * C;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = (v * lpp + q) * 4;
            const float4 g = *(const float4*)(G + base + c), yy = *(const float4*)(y + base + c);
#define GN_ACC(F)                                                            \
            { const float xh = (yy.F - mu[v].F) * rs[v].F;                   \
              const float du = g.F * dsilu_f(fmaf(a4[v].F, yy.F, b4[v].F));  \
              p0[v].F += du; p1[v].F = fmaf(du, xh, p1[v].F); p2[v].F += xh; }
            GN_ACC(x) GN_ACC(y) GN_ACC(z) GN_ACC(w)
#undef GN_ACC
        }
    }
    // lanes with equal q hold the same channels: reduce over `sub`, then over the 4 waves through LDS
#pragma unroll
    for (int v = 0; v < V; ++v) {
#define GN_RED(P, K)                                                                        \
        { float4 t = P[v];                                                                  \
          for (int m = lpp; m < 64; m <<= 1) { t.x += __shfl_xor(t.x, m); t.y += __shfl_xor(t.y, m); \
                                               t.z += __shfl_xor(t.z, m); t.w += __shfl_xor(t.w, m); } \
          if (sub == 0) *(float4*)&red[wave][K][(v * lpp + q) * 4] = t; }
        GN_RED(p0, 0) GN_RED(p1, 1) GN_RED(p2, 2)
#undef GN_RED
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 3; i += 256) {
        const int k = i / C, c = i - k * C;
        slab[(((size_t)n * E + e) * C + c) * 3 + k] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// Finalize per (sample, group): totals over the pixel chunks, then
//   film grads  dshift[n,c] = P0,  dscale[n,c] = gamma*P1 + beta*P0          (Block FiLM, :130-132)
//   per-sample contributions  dgam_n[n,c] = (1+s)*P1,  dbet_n[n,c] = (1+s)*P0,
//   dbias_n[n,c] = sum_px dy = ca*P0 - hw*M0r - M1r*P2                        (conv bias grad)
//   pass-B coefficients  ca[n,c] = rstd*gamma*(1+s),  M0r = rstd*mean_grp(dxhat),  M1r = rstd*mean_grp(dxhat*xhat)
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ slab, int E, int C, int groups,
                                                             long long hw, const float* __restrict__ mr,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ film /*scale at [n*fs + c], or null*/,
                                                             int film_stride, float* __restrict__ dfilm /*same layout, or null*/,
                                                             float* __restrict__ ca, float* __restrict__ mcoef /*[B][groups][2]*/,
                                                             float* __restrict__ dgam_n, float* __restrict__ dbet_n,
                                                             float* __restrict__ dbias_n) {
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cg = C / groups, lane = threadIdx.x;
    const float rstd = mr[((size_t)n * groups + g) * 2 + 1];
    double s0 = 0.0, s1 = 0.0;
    for (int i = lane; i < cg; i += 64) {
        const int c = g * cg + i;
        double P0 = 0.0, P1 = 0.0;
        for (int e = 0; e < E; ++e) {
            const float* sp = slab + (((size_t)n * E + e) * C + c) * 3;
            P0 += sp[0]; P1 += sp[1];
        }
        const float sc = film ? film[(size_t)n * film_stride + c] + 1.f : 1.f;
        s0 += (double)gamma[c] * sc * P0;
        s1 += (double)gamma[c] * sc * P1;
    }
    for (int m = 32; m >= 1; m >>= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
    const double cnt = (double)hw * cg;
    const float M0r = (float)(s0 / cnt) * rstd, M1r = (float)(s1 / cnt) * rstd;
    if (lane == 0) { mcoef[((size_t)n * groups + g) * 2] = M0r; mcoef[((size_t)n * groups + g) * 2 + 1] = M1r; }
    for (int i = lane; i < cg; i += 64) {
        const int c = g * cg + i;
        double P0 = 0.0, P1 = 0.0, P2 = 0.0;
        for (int e = 0; e < E; ++e) {
            const float* sp = slab + (((size_t)n * E + e) * C + c) * 3;
            P0 += sp[0]; P1 += sp[1]; P2 += sp[2];
        }
        const float sc = film ? film[(size_t)n * film_stride + c] + 1.f : 1.f;
        const float cac = rstd * gamma[c] * sc;
        ca[(size_t)n * C + c] = cac;
        dgam_n[(size_t)n * C + c] = (float)(sc * P1);
        dbet_n[(size_t)n * C + c] = (float)(sc * P0);
        dbias_n[(size_t)n * C + c] = (float)(cac * P0 - (double)hw * M0r - (double)M1r * P2);
        if (dfilm) {
            dfilm[(size_t)n * film_stride + c] = (float)(gamma[c] * P1 + beta[c] * P0);      // d scale
            dfilm[(size_t)n * film_stride + C + c] = (float)P0;                             // d shift
        }
    }
}

// out[c] (+)= sum_n in[n*stride + c], n < R (partial rows of the split reductions; R up to a few thousand).
// One block = 64 columns x 16 row lanes; fixed summation order => deterministic.
__global__ __launch_bounds__(1024) void sum_over_batch_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out,
                                                              int accumulate, int stride = 0) {
    __shared__ double red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (stride == 0) stride = C;
    double s = 0.0;
    if (c < C)
        for (int n = rl; n < R; n += 16) s += in[(size_t)n * stride + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

// three [R][C] -> [C] column sums in one launch (blockIdx.y selects the array): d gamma, d beta, d bias of one GroupNorm
__global__ __launch_bounds__(1024) void sum_over_batch3_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                               const float* __restrict__ in2, int R, int C, float* __restrict__ out0,
                                                               float* __restrict__ out1, float* __restrict__ out2, int accumulate) {
    __shared__ double red[16][64];
    const float* in = blockIdx.y == 0 ? in0 : (blockIdx.y == 1 ? in1 : in2);
    float* out = blockIdx.y == 0 ? out0 : (blockIdx.y == 1 ? out1 : out2);
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < C)
        for (int n = rl; n < R; n += 16) s += in[(size_t)n * C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

// Pass B: dy = ca*du - M0r - xhat*M1r
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ y,
                                                           const float* __restrict__ fa, const float* __restrict__ fb,
                                                           const float* __restrict__ mr, const float* __restrict__ ca,
                                                           const float* __restrict__ mcoef, int groups, long long hw, int C,
                                                           int lpp, long long P, float* __restrict__ dy,
                                                           unsigned* __restrict__ amax /*nullable: max |dy| as a bit pattern*/) {
    const int lane = threadIdx.x & 63, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int cg = C / groups;
    unsigned am = 0u;
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        if (p >= P) continue;
        const int n = (int)(p / hw);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = (v * lpp + q) * 4;
            const size_t o = (size_t)p * C + c;
            const float4 g = *(const float4*)(G + o), yy = *(const float4*)(y + o);
            const float4 a4 = *(const float4*)(fa + (size_t)n * C + c), b4 = *(const float4*)(fb + (size_t)n * C + c);
            const float4 c4 = *(const float4*)(ca + (size_t)n * C + c);
            float4 r;
#define GN_APP(F, K)                                                                        \
            { const int gi = (c + K) / cg;                                                  \
              const float mu = mr[((size_t)n * groups + gi) * 2], rsd = mr[((size_t)n * groups + gi) * 2 + 1]; \
              const float xh = (yy.F - mu) * rsd;                                           \
              const float du = g.F * dsilu_f(fmaf(a4.F, yy.F, b4.F));                       \
              r.F = c4.F * du - mcoef[((size_t)n * groups + gi) * 2] - xh * mcoef[((size_t)n * groups + gi) * 2 + 1]; }
            GN_APP(x, 0) GN_APP(y, 1) GN_APP(z, 2) GN_APP(w, 3)
#undef GN_APP
            *(float4*)(dy + o) = r;
            am = max(max(am, __builtin_bit_cast(unsigned, r.x) & 0x7fffffffu), __builtin_bit_cast(unsigned, r.y) & 0x7fffffffu);
            am = max(max(am, __builtin_bit_cast(unsigned, r.z) & 0x7fffffffu), __builtin_bit_cast(unsigned, r.w) & 0x7fffffffu);
        }
    }
    if (amax) {      // (an integer maximum: order independent, the result is deterministic; consumed by the fp16 weight gradient)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o));
        // 16 K same-address atomics serialise in L2 (measured: +80 us per launch): one per workgroup, and only when it can still
        // raise the value (a stale read only costs a redundant atomic: the maximum is monotonic)
        __shared__ unsigned wmax[4];
        if (lane == 0) wmax[threadIdx.x >> 6] = am;
        __syncthreads();
        if (threadIdx.x == 0) {
            am = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
            if (am > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, am);
        }
    }
}

// ---------------------------------------------------------------------------------------
// channel-LayerNorm backward (models/unet_model.py:52-61; y = xhat * g, per pixel over channels):
//   dx = rstd * (dxhat - mean_c(dxhat) - xhat * mean_c(dxhat*xhat)) [+ add],  dxhat = dY*g
//   dg partial[c] = sum_px dY * xhat.   mean / rstd are re-derived from x (eps 1e-5), nothing is saved.
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ x,
                                                     const float* __restrict__ g, const float* __restrict__ add,
                                                     float* __restrict__ dx, float* __restrict__ dg_part /*[gridDim.x][C]*/,
                                                     long long P, int C, int lpp) {
    __shared__ float red[4][64 * 4 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    float4 gg[V], dgacc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { gg[v] = *(const float4*)(g + (v * lpp + q) * 4); dgacc[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
    for (long long p0 = wave_id * ppw; p0 < P; p0 += nwaves * ppw) {
        const long long p = p0 + sub;
        const bool ok = p < P;
        float4 xv[V], dv[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const size_t o = (size_t)(ok ? p : 0) * C + (v * lpp + q) * 4;
            xv[v] = *(const float4*)(x + o);
            dv[v] = *(const float4*)(dY + o);
            if (!ok) { xv[v] = make_float4(0.f, 0.f, 0.f, 0.f); dv[v] = xv[v]; }
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        const float mean = group_sum(s, lpp) / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float a = xv[v].x - mean, b = xv[v].y - mean, c = xv[v].z - mean, d = xv[v].w - mean;
            d2 += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = rsqrtf(group_sum(d2, lpp) / (float)C + 1e-5f);
        float m1 = 0.f, m2 = 0.f;
        float4 xh[V], dh[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xh[v] = make_float4((xv[v].x - mean) * rstd, (xv[v].y - mean) * rstd, (xv[v].z - mean) * rstd, (xv[v].w - mean) * rstd);
            dh[v] = make_float4(dv[v].x * gg[v].x, dv[v].y * gg[v].y, dv[v].z * gg[v].z, dv[v].w * gg[v].w);
            m1 += (dh[v].x + dh[v].y) + (dh[v].z + dh[v].w);
            m2 += (dh[v].x * xh[v].x + dh[v].y * xh[v].y) + (dh[v].z * xh[v].z + dh[v].w * xh[v].w);
            dgacc[v].x = fmaf(dv[v].x, xh[v].x, dgacc[v].x); dgacc[v].y = fmaf(dv[v].y, xh[v].y, dgacc[v].y);
            dgacc[v].z = fmaf(dv[v].z, xh[v].z, dgacc[v].z); dgacc[v].w = fmaf(dv[v].w, xh[v].w, dgacc[v].w);
        }
        m1 = group_sum(m1, lpp) / (float)C;
        m2 = group_sum(m2, lpp) / (float)C;
        if (ok) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const size_t o = (size_t)p * C + (v * lpp + q) * 4;
                float4 r = make_float4(rstd * (dh[v].x - m1 - xh[v].x * m2), rstd * (dh[v].y - m1 - xh[v].y * m2),
                                       rstd * (dh[v].z - m1 - xh[v].z * m2), rstd * (dh[v].w - m1 - xh[v].w * m2));
                if (add) { const float4 ad = *(const float4*)(add + o); r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w; }
                *(float4*)(dx + o) = r;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float4 t = dgacc[v];
        for (int m = lpp; m < 64; m <<= 1) { t.x += __shfl_xor(t.x, m); t.y += __shfl_xor(t.y, m); t.z += __shfl_xor(t.z, m); t.w += __shfl_xor(t.w, m); }
        if (sub == 0) *(float4*)&red[wave][(v * lpp + q) * 4] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        dg_part[(size_t)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// per-channel sum over pixels: out_part[block][c] = sum of this block's pixels (bias grads of 1x1 convs etc.)
template <int V>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, long long P, int C, int lpp) {
    __shared__ float red[4][64 * 4 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long p = wave_id * ppw + sub; p < P; p += nwaves * ppw) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 t = *(const float4*)(x + (size_t)p * C + (v * lpp + q) * 4);
            acc[v].x += t.x; acc[v].y += t.y; acc[v].z += t.z; acc[v].w += t.w;
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float4 t = acc[v];
        for (int m = lpp; m < 64; m <<= 1) { t.x += __shfl_xor(t.x, m); t.y += __shfl_xor(t.y, m); t.z += __shfl_xor(t.z, m); t.w += __shfl_xor(t.w, m); }
        if (sub == 0) *(float4*)&red[wave][(v * lpp + q) * 4] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        part[(size_t)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ---------------------------------------------------------------------------------------
// small dense layers (time MLP, FiLM projections): y[b][j] = sum_k W[j][k] x[b][k] + bias[j]
//   dW[j][k] (+)= sum_b dy[b][j] x[b][k];  db[j] (+)= sum_b dy[b][j];  dx[b][k] = sum_j dy[b][j] W[j][k]
__global__ void linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int J, int K,
                                    int dy_stride, float* __restrict__ dW, float* __restrict__ db, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)J * K) return;
    const int j = (int)(i / K), k = (int)(i - (long long)j * K);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) { const float d = dy[(size_t)b * dy_stride + j]; s = fmaf(d, x[(size_t)b * K + k], s); sb += d; }
    dW[i] = accumulate ? dW[i] + s : s;
    if (k == 0 && db) db[j] = accumulate ? db[j] + sb : sb;
}
// dx partial: block (kb, js) handles 64 columns k and the j-range [js*jpb, (js+1)*jpb) for every b (W row read once,
// reused for all B <= 32*LB_NB rows).  part[js][b][k]; summed by sum_over_batch_kernel.
constexpr int LB_MAXB = 32;
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W, int B, int J, int K,
                                                           int dy_stride, int jpb, int b0, float* __restrict__ part) {
    __shared__ float red[4][LB_MAXB][64];
    const int kl = threadIdx.x & 63, jw = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    const int jb = blockIdx.y * jpb, je = (jb + jpb < J) ? jb + jpb : J;
    const int nb = (B - b0) < LB_MAXB ? (B - b0) : LB_MAXB;
    float acc[LB_MAXB];
#pragma unroll
    for (int b = 0; b < LB_MAXB; ++b) acc[b] = 0.f;
    const int kc = k < K ? k : K - 1;
    for (int j = jb + jw; j < je; j += 4) {
        const float w = W[(size_t)j * K + kc];
#pragma unroll
        for (int b = 0; b < LB_MAXB; ++b) {
            const int bb = b < nb ? b0 + b : b0;
            acc[b] = fmaf(dy[(size_t)bb * dy_stride + j], w, acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < LB_MAXB; ++b) red[jw][b][kl] = acc[b];
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 64; i += 256) {
        const int b = i >> 6, kk = i & 63;
        if (blockIdx.x * 64 + kk < K)
            part[((size_t)blockIdx.y * B + b0 + b) * K + blockIdx.x * 64 + kk] = (red[0][b][kk] + red[1][b][kk]) + (red[2][b][kk] + red[3][b][kk]);
    }
}
// g *= act'(pre):  mode 0 SiLU (pre = input of the SiLU), mode 1 exact-erf GELU
__global__ void act_bwd_kernel(float* __restrict__ g, const float* __restrict__ pre, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    float d;
    if (mode == 0) d = dsilu_f(x);
    else d = 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
    g[i] *= d;
}

// out = a + b
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
        ((float4*)out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

// ---------------------------------------------------------------------------------------
// final_conv (1x1, C -> OD, NCHW output) backward: dX[p][c] = sum_o dY[n][o][pix] w[o][c];
// dw partial[block][o][c] = sum_p dY*X, db partial[block][o] = sum_p dY.
template <int V>
__global__ __launch_bounds__(256) void final_conv_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ x,
                                                             const float* __restrict__ w, float* __restrict__ dx,
                                                             float* __restrict__ part /*[grid][OD][C+1]*/, long long P,
                                                             long long hw, int C, int OD, int lpp) {
    __shared__ float red[4][64 * 4 * V + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & (lpp - 1), sub = lane / lpp, ppw = 64 / lpp;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (int o = 0; o < OD; ++o) {
        float4 acc[V]; float accb = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long p = wave_id * ppw + sub; p < P; p += nwaves * ppw) {
            const long long n = p / hw, pix = p - n * hw;
            const float d = dY[((size_t)n * OD + o) * hw + pix];
            if (q == 0) accb += d;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const size_t off = (size_t)p * C + (v * lpp + q) * 4;
                const float4 xx = *(const float4*)(x + off);
                acc[v].x = fmaf(d, xx.x, acc[v].x); acc[v].y = fmaf(d, xx.y, acc[v].y);
                acc[v].z = fmaf(d, xx.z, acc[v].z); acc[v].w = fmaf(d, xx.w, acc[v].w);
                const float4 ww = *(const float4*)(w + (size_t)o * C + (v * lpp + q) * 4);
                float4 r = make_float4(d * ww.x, d * ww.y, d * ww.z, d * ww.w);
                if (o > 0) { const float4 pr = *(const float4*)(dx + off); r.x += pr.x; r.y += pr.y; r.z += pr.z; r.w += pr.w; }
                *(float4*)(dx + off) = r;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float4 t = acc[v];
            for (int m = lpp; m < 64; m <<= 1) { t.x += __shfl_xor(t.x, m); t.y += __shfl_xor(t.y, m); t.z += __shfl_xor(t.z, m); t.w += __shfl_xor(t.w, m); }
            if (sub == 0) *(float4*)&red[wave][(v * lpp + q) * 4] = t;
        }
        for (int m = 1; m < 64; m <<= 1) accb += __shfl_xor(accb, m);
        if (lane == 0) red[wave][64 * 4 * V] = accb;
        __syncthreads();
        for (int c = threadIdx.x; c <= C; c += 256) {
            const int cc = c < C ? c : 64 * 4 * V;
            part[((size_t)blockIdx.x * OD + o) * (C + 1) + c] = (red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// stem (7x7, pad 3) weight gradient: dW[co][ci][ky][kx] = sum dY[n][y][x][co] * x[n][ci][y+ky-3][x+kx-3]
// (NCHW input with few channels).  Each block walks `tpb` 8x32 tiles and writes one partial [Cin*49][Cout].
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ x, int B, int Cin,
                                                         int H, int W, int Cout, int tpb, float* __restrict__ part) {
    constexpr int TR = 8, TC = 32, HR = TR + 6, HC = TC + 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sdy = (float*)smem;                       // [256][64]
    float* sx = sdy + 256 * 64;                      // [Cin][HR][HC]
    const int tid = threadIdx.x, co_l = tid & 63, tq = tid >> 6;
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR, tpi = tiles_x * tiles_y, ntile = B * tpi;
    const int t_beg = blockIdx.x * tpb, t_end = (t_beg + tpb < ntile) ? t_beg + tpb : ntile;
    const int K = Cin * 49;
    for (int cb = 0; cb < Cout; cb += 64) {
        float acc[13];                                // taps tq, tq+4, ... of up to 49*Cin (Cin <= 1 per pass of 13*4=52 >= 49)
        for (int kb = 0; kb < K; kb += 52) {
#pragma unroll
            for (int j = 0; j < 13; ++j) acc[j] = 0.f;
            for (int tl = t_beg; tl < t_end; ++tl) {
                const int n = tl / tpi, trem = tl - n * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
                const int oy0 = ty * TR, ox0 = tx * TC;
                __syncthreads();
                for (int i = tid; i < 256 * 16; i += 256) {
                    const int px = i >> 4, c4 = (i & 15) * 4;
                    const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
                    const bool ok = oy < H && ox < W && cb + c4 < Cout;
                    const int oyc = oy < H ? oy : H - 1, oxc = ox < W ? ox : W - 1, cc = cb + c4 < Cout ? cb + c4 : 0;
                    float4 v = *(const float4*)(dY + ((size_t)(n * H + oyc) * W + oxc) * Cout + cc);
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(sdy + px * 64 + c4) = v;
                }
                for (int i = tid; i < Cin * HR * HC; i += 256) {
                    const int ci = i / (HR * HC), rem = i - ci * HR * HC, hy = rem / HC, hx = rem - hy * HC;
                    const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                    const float v = x[((size_t)(n * Cin + ci) * H + iyc) * W + ixc];
                    sx[i] = ok ? v : 0.f;
                }
                __syncthreads();
                for (int px = 0; px < 256; ++px) {
                    const float d = sdy[px * 64 + co_l];
                    const int py = px >> 5, pxx = px & 31;
#pragma unroll
                    for (int j = 0; j < 13; ++j) {
                        const int k = kb + tq + 4 * j;
                        if (k < K) {
                            const int ci = k / 49, tap = k - ci * 49, ky = tap / 7, kx = tap - ky * 7;
                            acc[j] = fmaf(d, sx[(ci * HR + py + ky) * HC + pxx + kx], acc[j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                const int k = kb + tq + 4 * j;
                if (k < K && cb + co_l < Cout) part[((size_t)blockIdx.x * K + k) * Cout + cb + co_l] = acc[j];
            }
        }
    }
}
// Matrix-core version (Cout <= 64, Cin*49 <= 128): the same sum as a GEMM on the exact-fp32 MFMA with
// i = output channel, j = (ci,ky,kx) tap and k = pixel parity; each lane gathers ITS tap's shifted pixel from the haloed
// LDS tile.  Every wave owns 2 rows of the 8x32 tile and writes its own partial slice [K][Cout] (4 slices per workgroup).
template <int NCB, int NTB>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ dY, const float* __restrict__ x, int B, int Cin,
                                                              int H, int W, int Cout, int tpb, float* __restrict__ part) {
    constexpr int TR = 8, TC = 32, HR = TR + 6, HC = TC + 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sdy = (float*)smem;                       // [256][64]
    float* sx = sdy + 256 * 64;                      // [Cin][HR][HC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR, tpi = tiles_x * tiles_y, ntile = B * tpi;
    const int t_beg = blockIdx.x * tpb, t_end = (t_beg + tpb < ntile) ? t_beg + tpb : ntile;
    const int K = Cin * 49;
    int boff[NTB];                                   // LDS offset of this lane's tap inside the halo tile, -1 = no such tap
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
        const int tt = nb * 32 + li;
        const int ci = tt / 49, tap = tt - ci * 49, ky = tap / 7, kx = tap - ky * 7;
        boff[nb] = tt < K ? (ci * HR + ky) * HC + kx : -1;
    }
    f32x16 acc[NCB][NTB];
#pragma unroll
    for (int a_ = 0; a_ < NCB; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < NTB; ++b_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int n = tl / tpi, trem = tl - n * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * TR, ox0 = tx * TC;
        __syncthreads();
        for (int i = tid; i < 256 * 16; i += 256) {
            const int px = i >> 4, c4 = (i & 15) * 4;
            const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
            const bool ok = oy < H && ox < W && c4 < Cout;
            const int oyc = oy < H ? oy : H - 1, oxc = ox < W ? ox : W - 1, cc = c4 < Cout ? c4 : 0;
            float4 v = *(const float4*)(dY + ((size_t)(n * H + oyc) * W + oxc) * Cout + cc);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4*)(sdy + px * 64 + c4) = v;
        }
        for (int i = tid; i < Cin * HR * HC; i += 256) {
            const int ci = i / (HR * HC), rem = i - ci * HR * HC, hy = rem / HC, hx = rem - hy * HC;
            const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            const float v = x[((size_t)(n * Cin + ci) * H + iyc) * W + ixc];
            sx[i] = ok ? v : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
            const int py = 2 * wave + (q >> 4), pxx = 2 * (q & 15) + lk;          // this lane's pixel of the pair (k = parity)
            float av[NCB], bv[NTB];
#pragma unroll
            for (int a_ = 0; a_ < NCB; ++a_) av[a_] = sdy[(py * 32 + pxx) * 64 + a_ * 32 + li];
#pragma unroll
            for (int b_ = 0; b_ < NTB; ++b_) {
                const int o = boff[b_] < 0 ? 0 : boff[b_];
                const float t = sx[o + py * HC + pxx];
                bv[b_] = boff[b_] < 0 ? 0.f : t;
            }
#pragma unroll
            for (int a_ = 0; a_ < NCB; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < NTB; ++b_) acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a_], bv[b_], acc[a_][b_], 0, 0, 0);
        }
    }
    // partial slice of this wave: part[(block*4 + wave)][k][co]; acc reg s of lane l = (co row (s&3)+8(s>>2)+4(l>>5), tap l&31)
    float* out = part + ((size_t)blockIdx.x * 4 + wave) * K * Cout;
#pragma unroll
    for (int a_ = 0; a_ < NCB; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < NTB; ++b_)
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                const int co = a_ * 32 + (s_ & 3) + 8 * (s_ >> 2) + 4 * lk, k = b_ * 32 + li;
                if (co < Cout && k < K) out[(size_t)k * Cout + co] = acc[a_][b_][s_];
            }
}
// dW[co][k] (+)= sum_blocks part[block][k][co]
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nb, int K, int Cout,
                                                                 float* __restrict__ dW, int accumulate) {
    __shared__ float red[16][64];                     // 64 consecutive (k, co) entries x 16 slice lanes, fixed order
    const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float s = 0.f;
    if (i < K * Cout)
        for (int b = sl; b < nb; b += 16) s += part[(size_t)b * K * Cout + i];
    red[sl][il] = s;
    __syncthreads();
    if (sl == 0 && i < K * Cout) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][il];
        const int k = i / Cout, co = i - k * Cout;
        float* o = dW + (size_t)co * K + k;
        *o = accumulate ? *o + t : t;
    }
}

// ---------------------------------------------------------------------------------------
// L1 epsilon loss with p2 weighting (models/diffusion_model.py:138-143) and its gradient:
//   loss = mean_b( mean_chw |eps - target| * p2w[t_b] ),  grad = sign(eps - target) * p2w[t_b] / (B*chw)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ eps, const float* __restrict__ target,
                                                      const long long* __restrict__ t, const float* __restrict__ p2w, int T, int B,
                                                      long long chw, float* __restrict__ grad, float* __restrict__ part /*[B][gridDim.x]*/) {
    __shared__ float red[256];
    const int b = blockIdx.y;
    long long ti = t[b];
    ti = ti < 0 ? 0 : (ti >= T ? T - 1 : ti);
    const float wgt = p2w[ti], gs = wgt / ((float)B * (float)chw);
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chw; i += (long long)gridDim.x * 256) {
        const float d = eps[(size_t)b * chw + i] - target[(size_t)b * chw + i];
        s += fabsf(d);
        if (grad) grad[(size_t)b * chw + i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = red[0] * wgt;
}
__global__ __launch_bounds__(256) void l1_loss_finalize_kernel(const float* __restrict__ part, int n, float denom, float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];          // fixed assignment of entries to threads => deterministic
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(red[0] / denom);
}

struct PixMap { int lpp, v; };
inline bool pixmap(int C, PixMap& m) {
    if (C % 4) return false;
    int v4 = C / 4, lpp = 1;
    while (lpp < 64 && (v4 % (lpp * 2)) == 0) lpp *= 2;
    m.lpp = lpp; m.v = v4 / lpp;
    return m.v >= 1 && m.v <= 4;
}

}  // namespace

#define DISPATCH_V(mv, KERN, ...)                                                            \
    switch (mv) {                                                                              \
        case 1: hipLaunchKernelGGL(KERN<1>, __VA_ARGS__); break;                               \
        case 2: hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); break;                               \
        case 3: hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); break;                               \
        default: hipLaunchKernelGGL(KERN<4>, __VA_ARGS__); break;                              \
    }

// pixel chunks per image of the reduction pass: 1024-pixel chunks at the large levels, but never fewer than 16 per image
// (>= 32 pixels each) so that the 32x32 / 64x64 levels still put ~500 workgroups on the chip
int gn_bwd_chunks(long long hw) {
    long long e = hw / 1024, lo = hw / 32 < 16 ? hw / 32 : 16;
    if (e < lo) e = lo;
    return (int)(e < 1 ? 1 : e);
}

// GroupNorm(+FiLM)+SiLU backward.  Workspace (floats): slab B*E*C*3, ca B*C, mcoef B*groups*2, dgam_n/dbet_n/dbias_n B*C each.
int op_gn_silu_backward(const float* G, const float* y, const float* fa, const float* fb, const float* mr, const float* gamma,
                        const float* beta, const float* film, int film_stride, float* dfilm, int B, long long hw, int C,
                        int groups, float* slab, float* ca, float* mcoef, float* dgam_n, float* dbet_n, float* dbias_n,
                        float* dy, float* dgamma, float* dbeta, float* dbias, int accumulate, unsigned* amax, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m) || C % groups) return TEDM_EINVAL;
    if (amax && hipMemsetAsync(amax, 0, 4, st) != hipSuccess) return TEDM_EHIP;
    const int E = gn_bwd_chunks(hw);
    const int ppb = (int)((hw + E - 1) / E);
    DISPATCH_V(m.v, gn_bwd_reduce_kernel, dim3(E, B), dim3(256), 0, st, G, y, fa, fb, mr, groups, hw, C, m.lpp, ppb, slab);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B * groups), dim3(64), 0, st, slab, E, C, groups, hw, mr, gamma, beta, film,
                       film_stride, dfilm, ca, mcoef, dgam_n, dbet_n, dbias_n);
    const int cb = (C + 63) / 64;
    hipLaunchKernelGGL(sum_over_batch3_kernel, dim3(cb, dbias ? 3 : 2), dim3(1024), 0, st, dgam_n, dbet_n, dbias_n, B, C, dgamma, dbeta,
                       dbias, accumulate);
    const long long P = (long long)B * hw;
    long long waves = (P + (64 / m.lpp) - 1) / (64 / m.lpp), blocks = (waves + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    DISPATCH_V(m.v, gn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, G, y, fa, fb, mr, ca, mcoef, groups, hw, C,
               m.lpp, P, dy, amax);
    return LAUNCH_OK();
}

int ln_bwd_blocks(long long P, int C) {
    PixMap m;
    if (!pixmap(C, m)) return 0;
    long long waves = (P + (64 / m.lpp) - 1) / (64 / m.lpp), blocks = (waves + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    return (int)(blocks < 1 ? 1 : blocks);
}

// dx = LN-backward(dY; x, g) [+ add]; dg (+)= sum_px dY*xhat.  part: ln_bwd_blocks(P,C)*C floats.
int op_ln_backward(const float* dY, const float* x, const float* g, const float* add, float* dx, float* part, float* dg,
                   int accumulate, long long P, int C, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    const int nb = ln_bwd_blocks(P, C);
    DISPATCH_V(m.v, ln_bwd_kernel, dim3(nb), dim3(256), 0, st, dY, x, g, add, dx, part, P, C, m.lpp);
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, nb, C, dg, accumulate);
    return LAUNCH_OK();
}

// out[c] (+)= sum over all P pixels of x[p][c].  part: ln_bwd_blocks(P,C)*C floats.
int op_colsum(const float* x, float* part, float* out, int accumulate, long long P, int C, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    const int nb = ln_bwd_blocks(P, C);
    DISPATCH_V(m.v, colsum_kernel, dim3(nb), dim3(256), 0, st, x, part, P, C, m.lpp);
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, nb, C, out, accumulate);
    return LAUNCH_OK();
}

int linear_bwd_splits(int J) { int s = J / 64; return s < 1 ? 1 : (s > 128 ? 128 : s); }
// part: linear_bwd_splits(J)*B*K floats (only when dx is requested)
int op_linear_backward(const float* dy, int dy_stride, const float* x, const float* W, int B, int J, int K, float* dW, float* db,
                       float* dx, float* part, int accumulate, hipStream_t st) {
    if (dW) hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)(((long long)J * K + 255) / 256)), dim3(256), 0, st, dy, x, B, J, K,
                               dy_stride, dW, db, accumulate);
    if (dx) {
        if (!part) return TEDM_EINVAL;
        const int S = linear_bwd_splits(J), jpb = (J + S - 1) / S;
        for (int b0 = 0; b0 < B; b0 += LB_MAXB)
            hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((K + 63) / 64, S), dim3(256), 0, st, dy, W, B, J, K, dy_stride, jpb, b0, part);
        hipLaunchKernelGGL(sum_over_batch_kernel, dim3((B * K + 63) / 64), dim3(1024), 0, st, part, S, B * K, dx, 0, 0);
    }
    return LAUNCH_OK();
}
int op_act_backward(float* g, const float* pre, int n, int mode, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, pre, n, mode);
    return LAUNCH_OK();
}
int op_add2(const float* a, const float* b, float* out, long long n, hipStream_t st) {
    if (n % 4) return TEDM_EINVAL;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, n / 4);
    return LAUNCH_OK();
}
int final_conv_bwd_blocks(long long P) { long long b = P / 2048; return (int)(b < 1 ? 1 : (b > 512 ? 512 : b)); }
// part: final_conv_bwd_blocks(P)*OD*(C+1) floats
int op_final_conv_backward(const float* dY_nchw, const float* x, const float* w, float* dx, float* part, float* dw, float* db,
                           int B, long long hw, int C, int OD, int accumulate, hipStream_t st) {
    PixMap m;
    if (!pixmap(C, m)) return TEDM_EINVAL;
    const long long P = (long long)B * hw;
    const int nb = final_conv_bwd_blocks(P);
    DISPATCH_V(m.v, final_conv_bwd_kernel, dim3(nb), dim3(256), 0, st, dY_nchw, x, w, dx, part, P, hw, C, OD, m.lpp);
    // part rows are [o][C+1]: treat as nb x (OD*(C+1)) and sum over blocks, then split into dw / db on the fly
    for (int o = 0; o < OD; ++o) {
        hipLaunchKernelGGL(sum_over_batch_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part + (size_t)o * (C + 1), nb, C,
                           dw + (size_t)o * C, accumulate, OD * (C + 1));
        if (db) hipLaunchKernelGGL(sum_over_batch_kernel, dim3(1), dim3(1024), 0, st, part + (size_t)o * (C + 1) + C, nb, 1, db + o,
                                   accumulate, OD * (C + 1));
    }
    return LAUNCH_OK();
}
// number of partial slices [Cin*49][Cout] the stem weight gradient writes (4 per workgroup: one per wave)
int stem_wgrad_blocks(int B, int H, int W) {
    const long long ntile = (long long)B * ((W + 31) / 32) * ((H + 7) / 8);
    long long nb = ntile / 16;
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    return (int)(nb * 4);
}
// part: stem_wgrad_blocks*Cin*49*Cout floats
int op_stem_wgrad(const float* dY, const float* x_nchw, int B, int Cin, int H, int W, int Cout, float* part, float* dW,
                  int accumulate, hipStream_t st) {
    const int nb = stem_wgrad_blocks(B, H, W) / 4;
    const long long ntile = (long long)B * ((W + 31) / 32) * ((H + 7) / 8);
    const int tpb = (int)((ntile + nb - 1) / nb);
    const size_t lds = (size_t)(256 * 64 + Cin * 14 * 38) * 4;
    const int nb2 = (int)((ntile + tpb - 1) / tpb);
    const int K = Cin * 49;
    int slices = nb2;
    auto set_attr = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    if (Cout <= 64 && Cout % 4 == 0 && K <= 128) {                 // matrix-core path
        const int ncb = (Cout + 31) / 32, ntb = (K + 31) / 32;
#define STEM_CASE(A, Bk)                                                                                              \
        if (ncb == A && ntb == Bk) {                                                                                  \
            static bool attr = false;                                                                                 \
            if (!attr) { if (!set_attr((const void*)stem_wgrad_mfma_kernel<A, Bk>)) return TEDM_EHIP; attr = true; }   \
            hipLaunchKernelGGL((stem_wgrad_mfma_kernel<A, Bk>), dim3(nb2), dim3(256), lds, st, dY, x_nchw, B, Cin, H, W, Cout, tpb, part); \
        }
        STEM_CASE(1, 2) STEM_CASE(2, 2) STEM_CASE(1, 4) STEM_CASE(2, 4)
#undef STEM_CASE
        if (!((ncb == 1 || ncb == 2) && (ntb == 2 || ntb == 4))) return TEDM_EINVAL;
        slices = nb2 * 4;
    } else {
        static bool attr = false;
        if (!attr) { if (!set_attr((const void*)stem_wgrad_kernel)) return TEDM_EHIP; attr = true; }
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nb2), dim3(256), lds, st, dY, x_nchw, B, Cin, H, W, Cout, tpb, part);
    }
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((K * Cout + 63) / 64), dim3(1024), 0, st, part, slices, K, Cout, dW, accumulate);
    return LAUNCH_OK();
}
// part: B*64 floats
int op_l1_loss(const float* eps, const float* target, const long long* t, const float* p2w, int T, int B, long long chw,
               float* grad, float* part, float* loss, hipStream_t st) {
    hipLaunchKernelGGL(l1_loss_kernel, dim3(64, B), dim3(256), 0, st, eps, target, t, p2w, T, B, chw, grad, part);
    hipLaunchKernelGGL(l1_loss_finalize_kernel, dim3(1), dim3(256), 0, st, part, B * 64, (float)B * (float)chw, loss);
    return LAUNCH_OK();
}

int op_upsample_sum(const float* const* z, const int* h, const int* w, int n, int B, int S, int C, const float* bias, float* out,
                    hipStream_t st) {
    if (n < 1 || n > 8 || C % 4) return TEDM_EINVAL;
    UpSumArgs a{};
    a.n = n;
    for (int k = 0; k < n; ++k) { a.z[k] = z[k]; a.h[k] = h[k]; a.w[k] = w[k]; }
    const long long total4 = (long long)B * S * S * (C / 4);
    if ((long long)B * S >= (1ll << 31) || (long long)S * (C / 4) >= (1ll << 31)) return TEDM_EINVAL;
    const int bx = (S * (C / 4) + 255) / 256;
    const int gx = bx > 64 ? 64 : bx, gy_cap = 16384 / gx;                     // ~16 k blocks, each walking several rows
    dim3 grid(gx, (unsigned)((long long)B * S > gy_cap ? gy_cap : B * S));
    hipLaunchKernelGGL(upsample_sum_kernel, grid, dim3(256), 0, st, a, bias, out, B, S, C, total4);
    return LAUNCH_OK();
}
int op_head_tail(const float* const* z, const int* h, const int* w, int n, int B, int S, int C1, const float* b1, const float* w2t,
                 const float* b2, const float* w3, const float* b3, float* logits, hipStream_t st) {
    if (n < 1 || n > 8 || C1 % 32 || C1 < 32) return TEDM_EINVAL;
    UpSumArgs a{};
    a.n = n;
    for (int k = 0; k < n; ++k) { a.z[k] = z[k]; a.h[k] = h[k]; a.w[k] = w[k]; }
    const long long P = (long long)B * S * S;
    hipLaunchKernelGGL(head_tail_kernel, dim3((unsigned)((P + HT_PIX - 1) / HT_PIX)), dim3(256), 0, st, a, b1, w2t, b2, w3, b3, logits, P, S, C1);
    return LAUNCH_OK();
}
int op_pool_sum(const float* g, float* out, int B, int h, int w, int f, int C, hipStream_t st) {
    if (C % 4 || f < 1) return TEDM_EINVAL;
    const long long total4 = (long long)B * h * w * (C / 4);
    if ((long long)B * h >= (1ll << 31) || (long long)w * (C / 4) >= (1ll << 31)) return TEDM_EINVAL;
    const int bx = (w * (C / 4) + 255) / 256;
    const int gx = bx > 64 ? 64 : bx, gy_cap = 16384 / gx;
    dim3 grid(gx, (unsigned)((long long)B * h > gy_cap ? gy_cap : B * h));
    hipLaunchKernelGGL(pool_sum_kernel, grid, dim3(256), 0, st, g, out, B, h, w, f, C, total4);
    return LAUNCH_OK();
}
int op_pool2x2_sum(const float* g, float* out, int B, int h, int w, int C, hipStream_t st) {
    if (C % 4) return TEDM_EINVAL;
    const long long total4 = (long long)B * h * w * (C / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, out, B, h, w, C, total4);
    return LAUNCH_OK();
}
